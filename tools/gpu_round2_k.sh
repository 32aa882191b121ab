#!/bin/bash
# GPU session K: A/B of a change in gemm_v4.hip (build/old = the previous commit): tests, in-kernel trace, same-box bench
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_gemm_v4.py tests/test_gpu_clip.py tests/test_gpu_golden_fullwidth.py -m gpu -q -x > gpurun_out/pytest_k.txt 2>&1; echo "rc=$?" >> gpurun_out/pytest_k.txt
tail -3 gpurun_out/pytest_k.txt
( export LD_LIBRARY_PATH=$(pwd)/build/trace:$LD_LIBRARY_PATH
for cfg in "25600 768 768 0" "22016 2048 512 0" "25600 3072 768 1"; do
  echo "=== $cfg"; timeout 120 build/trace/gemm_trace $cfg 2>&1 | head -5
done ) > gpurun_out/trace_epilogue_pipe.txt 2>&1
cut -c1-230 gpurun_out/trace_epilogue_pipe.txt
for rep in 1 2; do for v in old new; do
  if [ $v = old ]; then export DECLIP_HIP_LIB=$(pwd)/build/old/libdeclip_hip.so; else unset DECLIP_HIP_LIB; fi
  echo "== $v"; timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-loss-delta 2>&1 | grep '^{' | python -c "
import json,sys
j=json.loads(sys.stdin.read()); r=j['roofline']; print('   %.1f pairs/s  %.2f ms/step  loss %.4f | GEMM %.0f TF frac %.3f' % (j['value'], j['ms_per_step'], j['loss'], r['achieved'], r['frac']))"
done; done 2>&1 | tee gpurun_out/ab_pipe_epilogue.txt
