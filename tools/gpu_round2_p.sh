#!/bin/bash
# GPU session P: the hipGraph step for DeCLIP / DeFILIP (host row counts with the batch, masked-LM selection uploaded once)
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
for m in declip defilip; do for g in 0 1; do echo "== $m --graph $g"; timeout 300 python bench.py --model $m --graph $g --steps 20 --warmup 5 --no-cpu-baseline --no-loss-delta --no-roofline > gpurun_out/p_$m$g.txt 2>&1; grep '^{' gpurun_out/p_$m$g.txt | python -c "
import json,sys
j=json.loads(sys.stdin.read()); print('   %.1f pairs/s  %.2f ms/step  loss %.4f' % (j['value'], j['ms_per_step'], j['loss']))" 2>/dev/null || grep -E "Error|error" gpurun_out/p_$m$g.txt | tail -3; done; done 2>&1 | tee gpurun_out/ab_graph_declip.txt
