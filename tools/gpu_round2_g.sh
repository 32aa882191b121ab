#!/bin/bash
# GPU session G: masked-LM cross-entropy fused into the GEMM epilogue (MODE_CE_FWD / MODE_CE_BWD)
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_clip.py tests/test_gpu_golden_fullwidth.py tests/test_gpu_dist.py tests/test_gpu_solver.py -m gpu -q -k "ce_fused or declip or defilip" > gpurun_out/pytest_ce.txt 2>&1; echo "rc=$?" >> gpurun_out/pytest_ce.txt
tail -25 gpurun_out/pytest_ce.txt
rm -f gpurun_out/ab_ce.txt
for f in 0 1 0 1; do
  echo "== DeCLIP DH_CE_FUSED=$f" >> gpurun_out/ab_ce.txt
  DH_CE_FUSED=$f timeout 300 python bench.py --model declip --steps 10 --warmup 3 --no-cpu-baseline --no-roofline 2>&1 | grep '^{' | python -c "
import json,sys
j=json.loads(sys.stdin.read()); print('   %.1f pairs/s  %.2f ms/step  loss %.4f' % (j['value'], j['ms_per_step'], j['loss']))" >> gpurun_out/ab_ce.txt 2>&1
done
cat gpurun_out/ab_ce.txt
