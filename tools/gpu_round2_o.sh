#!/bin/bash
# GPU session O: per-step timeline of the default CLIP step (two tower streams; hipGraph on / off): idle time between kernels
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
R=$(pwd)
cd /tmp && export TMPDIR=/tmp
for g in 0 1; do
  timeout 300 rocprofv3 --kernel-trace -d $R/gpurun_out/prof_o$g -o trace -- python $R/bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-loss-delta --no-roofline --graph $g > $R/gpurun_out/prof_o$g.log 2>&1
  DB=$(find $R/gpurun_out/prof_o$g -name "*.db" | head -1)
  python $R/tools/rocpd_stats.py $DB > $R/gpurun_out/stats_o$g.txt 2>&1
  rm -rf $R/gpurun_out/prof_o$g
  echo "=== graph $g"; grep -A12 "^per step" $R/gpurun_out/stats_o$g.txt; grep '^{' $R/gpurun_out/prof_o$g.log | cut -c1-160
done
