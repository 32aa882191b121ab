#!/bin/bash
# round 3, GPU session B: the continuous K-tile stream of gemm_v4 -- parity tests, then same-box A/B against the round-2 kernel
# (build/base/libdeclip_hip.so = the library of commit 331f279) on the CLIP step
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/r3b
mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
( timeout 600 python -m pytest tests/test_gpu_gemm_v4.py -x -q -m gpu ) > $O/t_gemm.log 2>&1
echo "gemm rc=$?" >> $O/t_gemm.log
tail -3 $O/t_gemm.log
if ! grep -q "rc=0" $O/t_gemm.log; then exit 1; fi
( timeout 600 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_clip.py -x -q -m gpu -k "ce_fused or maxsim or bf16 or v4" ) > $O/t_kern.log 2>&1
echo "kern rc=$?" >> $O/t_kern.log
tail -3 $O/t_kern.log
for i in 1 2; do
  DECLIP_HIP_LIB=$PWD/build/base/libdeclip_hip.so DH_BENCH_GEMM_TABLE=$O/table_base_$i.txt timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-loss-delta > $O/bench_base_$i.log 2>&1
  DH_BENCH_GEMM_TABLE=$O/table_new_$i.txt timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-loss-delta > $O/bench_new_$i.log 2>&1
done
for f in $O/bench_*.log; do echo $f; python - "$f" <<'PY'
import json,sys
for l in open(sys.argv[1]):
    if l.startswith('{'):
        d=json.loads(l); r=d.get('roofline',{})
        print("   %.1f pairs/s  %.3f ms/step | GEMM %.3f ms/step %.1f TF/s" % (d['value'], d['ms_per_step'], r.get('gemm_ms_per_step',0), r.get('achieved',0)))
PY
done
