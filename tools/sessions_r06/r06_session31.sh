#!/bin/bash
# round 6, session 31: kernel-to-kernel gaps of the captured step: towers on ONE stream, captured vs eager, rocprofv3 kernel trace -> per-step span minus the sum of the kernel durations
cd $GRAFT_REPO_ROOT; O=gpurun_out/r06s31; mkdir -p $O
python -m declip_amd.build > /dev/null 2>&1
cd /tmp && export TMPDIR=/tmp
for g in 1 0; do
  DH_TOWER_STREAMS=0 rocprofv3 --kernel-trace -d $GRAFT_REPO_ROOT/$O/trace_g$g -o trace -- python $GRAFT_REPO_ROOT/bench.py --steps 6 --warmup 3 --graph $g --no-cpu-baseline --no-loss-delta --no-roofline > $GRAFT_REPO_ROOT/$O/trace_g$g.log 2>&1
  DB=$(find $GRAFT_REPO_ROOT/$O/trace_g$g -name "*.db" | head -1); python $GRAFT_REPO_ROOT/tools/rocpd_stats.py $DB > $GRAFT_REPO_ROOT/$O/stats_g$g.txt 2>&1
  echo "--- graph $g"; grep -A 12 "^per step" $GRAFT_REPO_ROOT/$O/stats_g$g.txt; tail -1 $GRAFT_REPO_ROOT/$O/trace_g$g.log | cut -c1-200
  rm -rf $GRAFT_REPO_ROOT/$O/trace_g$g
done
