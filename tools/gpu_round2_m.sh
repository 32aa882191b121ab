#!/bin/bash
# GPU session M: kernel tables of the DeCLIP / SLIP / FILIP steps (one stream, no graph) -- where do the multi-view models spend time?
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
R=$(pwd)
cd /tmp && export TMPDIR=/tmp
for m in declip filip; do
  DH_TOWER_STREAMS=0 timeout 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_$m -o trace -- python $R/bench.py --model $m --steps 4 --warmup 2 --no-cpu-baseline --no-loss-delta --no-roofline --graph 0 > $R/gpurun_out/prof_$m.log 2>&1
  DB=$(find $R/gpurun_out/prof_$m -name "*.db" | head -1)
  python $R/tools/rocpd_stats.py $DB > $R/gpurun_out/stats_$m.txt 2>&1
  rm -rf $R/gpurun_out/prof_$m
  echo "=== $m"; head -45 $R/gpurun_out/stats_$m.txt | cut -c1-150; grep TOTAL $R/gpurun_out/stats_$m.txt
done
