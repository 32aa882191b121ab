#!/bin/bash
# GPU session U: how fast would the CLIP step be if parts of the persistent GEMM were free?  Ablation builds of gemm_v4.hip
# (V4_ABL: 8 = no epilogue, 1 = no LDS-DMA, 2 = no MFMA; results are numerically meaningless, only the step time is read)
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
for v in full abl_8 abl_1 abl_2; do
  if [ $v = full ]; then unset DECLIP_HIP_LIB; else export DECLIP_HIP_LIB=$(pwd)/build/$v/libdeclip_hip.so; fi
  echo "== $v"; timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-loss-delta 2>&1 | grep '^{' | python -c "
import json,sys
j=json.loads(sys.stdin.read()); r=j['roofline']; print('   %.1f pairs/s  %.2f ms/step | GEMM family %.2f ms/step (%.0f TF nominal)' % (j['value'], j['ms_per_step'], r['gemm_ms_per_step'], r['achieved']))"
done 2>&1 | tee gpurun_out/ab_ablation_step.txt
