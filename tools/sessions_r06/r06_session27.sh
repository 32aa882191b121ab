#!/bin/bash
# round 6, session 12 / 27: two-stream timeline of the captured CLIP step (tools/timeline_stats.py on a rocprofv3 kernel trace)
cd $GRAFT_REPO_ROOT; O=gpurun_out/r06s27; mkdir -p $O
python -m declip_amd.build > /dev/null 2>&1
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace -d $GRAFT_REPO_ROOT/$O/trace -o trace -- python $GRAFT_REPO_ROOT/bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-loss-delta --no-roofline > $GRAFT_REPO_ROOT/$O/trace.log 2>&1
cd $GRAFT_REPO_ROOT
DB=$(find $O/trace -name "*.db" | head -1); python tools/timeline_stats.py $DB > $O/timeline.txt 2>&1; cat $O/timeline.txt | cut -c1-200
python tools/rocpd_stats.py $DB > $O/stats.txt 2>&1; tail -12 $O/stats.txt
rm -rf $O/trace
