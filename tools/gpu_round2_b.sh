#!/bin/bash
# GPU session B of round 2: grouped weight gradients (MODE_GROUP) + K-sliced few-tile GEMMs: kernel tests, then the whole suite
# WITHOUT -x (every failure in one pass), then same-box A/B of the grouping and its K-slice count.
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_gemm_v4.py -m gpu -q > gpurun_out/pytest_gemm_v4.txt 2>&1; echo "rc=$?" >> gpurun_out/pytest_gemm_v4.txt
tail -4 gpurun_out/pytest_gemm_v4.txt
for cfg in "0 0" "1 0" "1 2" "1 3" "1 5" "1 7"; do
  set -- $cfg
  echo "== DH_V4_GROUP=$1 DH_V4_GROUP_SPLIT=$2" >> gpurun_out/ab_group.txt
  DH_V4_GROUP=$1 DH_V4_GROUP_SPLIT=$2 DH_BENCH_GEMM_TABLE=gpurun_out/gemm_table_g$1_s$2.txt timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-loss-delta 2>/dev/null | grep '^{' | python -c "
import json,sys
j=json.loads(sys.stdin.read()); r=j['roofline']
print('%.1f pairs/s  %.2f ms/step  GEMM %.1f TF  %.2f ms/step  loss %.4f' % (j['value'], j['ms_per_step'], r['achieved'], r['gemm_ms_per_step'], j['loss']))" >> gpurun_out/ab_group.txt 2>&1
done
cat gpurun_out/ab_group.txt
timeout 1500 python -m pytest tests -m gpu -q > gpurun_out/pytest_gpu_all.txt 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_gpu_all.txt
tail -15 gpurun_out/pytest_gpu_all.txt
