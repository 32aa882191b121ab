#!/bin/bash
# GPU session E: tile order of the forward / dX GEMMs (DH_V4_GROUP_M): same-box A/B + kernel tests of the new default.
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_gemm_v4.py tests/test_gpu_graph.py -m gpu -q > gpurun_out/pytest_e.txt 2>&1; echo "rc=$?" >> gpurun_out/pytest_e.txt
tail -4 gpurun_out/pytest_e.txt
rm -f gpurun_out/ab_groupm.txt
for gm in 8 1 2 4 1 8; do
  echo "== DH_V4_GROUP_M=$gm" >> gpurun_out/ab_groupm.txt
  DH_V4_GROUP_M=$gm DH_BENCH_GEMM_TABLE=gpurun_out/gemm_table_gm$gm.txt timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-loss-delta 2>/dev/null | grep '^{' | python -c "
import json,sys
j=json.loads(sys.stdin.read()); r=j['roofline']
print('%.1f pairs/s  %.2f ms/step  GEMM %.1f TF  %.2f ms/step  loss %.4f' % (j['value'], j['ms_per_step'], r['achieved'], r['gemm_ms_per_step'], j['loss']))" >> gpurun_out/ab_groupm.txt 2>&1
done
cat gpurun_out/ab_groupm.txt
