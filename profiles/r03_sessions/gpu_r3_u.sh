#!/bin/bash
# round 3, GPU session U: kernel tables of the DeCLIP and FILIP steps, the CLIP step's timeline when replayed from the graph, attention / LayerNorm PMC
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
O=$ROOT/gpurun_out/r3u
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for m in declip filip; do
  DH_TOWER_STREAMS=0 timeout 300 rocprofv3 --kernel-trace --stats -d $O/trace_$m -o trace -- python $ROOT/bench.py --model $m --steps 4 --warmup 2 --no-cpu-baseline --no-loss-delta --no-roofline --graph 0 > $O/trace_$m.log 2>&1
  DB=$(find $O/trace_$m -name "*.db" | head -1)
  [ -n "$DB" ] && python $ROOT/tools/rocpd_stats.py $DB > $O/stats_$m.txt 2>&1
  rm -rf $O/trace_$m
done
timeout 300 rocprofv3 --kernel-trace --stats -d $O/trace_graph -o trace -- python $ROOT/bench.py --steps 8 --warmup 4 --no-cpu-baseline --no-loss-delta --no-roofline --graph 1 > $O/trace_graph.log 2>&1
DB=$(find $O/trace_graph -name "*.db" | head -1)
[ -n "$DB" ] && python $ROOT/tools/rocpd_stats.py $DB | sed -n '/^TOTAL/,$p' > $O/timeline_graph.txt 2>&1
rm -rf $O/trace_graph
cd $ROOT && bash tools/pmc_attn.sh > $O/pmc_attn.txt 2>&1
head -12 $O/stats_declip.txt | cut -c1-140; head -8 $O/stats_filip.txt | cut -c1-140; cat $O/timeline_graph.txt | head -14; tail -14 $O/pmc_attn.txt | cut -c1-200
