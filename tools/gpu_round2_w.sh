#!/bin/bash
# GPU session W: NN queue with its write pointer on the device; DeCLIP / DeFILIP from a hipGraph by default
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
timeout 170 python -m pytest tests/test_gpu_clip.py tests/test_gpu_golden_fullwidth.py tests/test_gpu_dist.py -m gpu -q -k "declip or defilip" > gpurun_out/pytest_w.txt 2>&1; echo "rc=$?" >> gpurun_out/pytest_w.txt
tail -3 gpurun_out/pytest_w.txt
for m in declip defilip; do timeout 100 python bench.py --model $m --steps 20 --warmup 5 --no-cpu-baseline --no-loss-delta --no-roofline 2>&1 | grep '^{' | python -c "
import json,sys
j=json.loads(sys.stdin.read()); print('$m  %.1f pairs/s  %.2f ms/step  loss %.4f graph %s' % (j['value'], j['ms_per_step'], j['loss'], j['config'].get('step_graph')))"; done 2>&1 | tee gpurun_out/bench_w.txt
