#!/bin/bash
# round 5: dispatch count / idle time of the DEFAULT step (two tower streams, captured graph), rocprofv3 --kernel-trace
cd $GRAFT_REPO_ROOT; O=$GRAFT_REPO_ROOT/gpurun_out/r05s12; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $O/trace -o trace -- python $GRAFT_REPO_ROOT/bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-loss-delta --no-roofline > $O/trace.log 2>&1
DB=$(find $O/trace -name "*.db" | head -1)
python $GRAFT_REPO_ROOT/tools/rocpd_stats.py $DB > $O/stats_graph.txt 2>&1
grep -n "per step" -A16 $O/stats_graph.txt | head -30
rm -rf $O/trace
