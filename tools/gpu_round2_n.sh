#!/bin/bash
# GPU session N: vectorised BatchNorm1d, optimizer-written bf16 mirror (no per-step weight cast): tests + bench lines
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_clip.py tests/test_gpu_solver.py tests/test_gpu_graph.py -m gpu -q -k "bn1d or mirror or declip or solver or graph or train_steps" > gpurun_out/pytest_n.txt 2>&1; echo "rc=$?" >> gpurun_out/pytest_n.txt
tail -5 gpurun_out/pytest_n.txt
for m in clip declip filip; do for i in 1 2; do echo "== $m"; timeout 300 python bench.py --model $m --steps 20 --warmup 5 --no-cpu-baseline --no-loss-delta --no-roofline 2>&1 | grep '^{' | python -c "
import json,sys
j=json.loads(sys.stdin.read()); print('   %.1f pairs/s  %.2f ms/step  loss %.4f' % (j['value'], j['ms_per_step'], j['loss']))"; done; done
echo "== clip DH_MIRROR_TRUST=0"; DH_MIRROR_TRUST=0 timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-loss-delta --no-roofline 2>&1 | grep '^{' | python -c "
import json,sys
j=json.loads(sys.stdin.read()); print('   %.1f pairs/s  %.2f ms/step  loss %.4f' % (j['value'], j['ms_per_step'], j['loss']))"
