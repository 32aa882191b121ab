#!/bin/bash
# round 3, GPU session D: fast hand-over -- v4 parity tests, in-kernel trace, same-box A/B of the CLIP step
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/r3d
mkdir -p $O
( timeout 600 python -m pytest tests/test_gpu_gemm_v4.py -x -q -m gpu ) > $O/t_gemm.log 2>&1
echo "gemm rc=$?" >> $O/t_gemm.log
tail -2 $O/t_gemm.log
if ! grep -q "rc=0" $O/t_gemm.log; then exit 1; fi
for shape in "25600 768 768 0" "22016 2048 512 0" "25600 3072 768 1"; do
  echo "=== $shape (K-tile stream, fast hand-over)" >> $O/trace.txt
  LD_LIBRARY_PATH=$PWD/build/trace timeout 120 tools/gemm_trace $shape 0 10 2>&1 | head -8 >> $O/trace.txt
done
cut -c1-250 $O/trace.txt
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
