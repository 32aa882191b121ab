#!/bin/bash
# round 3, GPU session I: V4_SCHED=2 (piece-level balanced LDS-DMA placement) vs V4_SCHED=1 vs the round-2 kernel
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/r3i
mkdir -p $O
( timeout 600 python -m pytest tests/test_gpu_gemm_v4.py -x -q -m gpu ) > $O/t_gemm.log 2>&1
echo "gemm rc=$?" >> $O/t_gemm.log
tail -2 $O/t_gemm.log
if ! grep -q "rc=0" $O/t_gemm.log; then tail -30 $O/t_gemm.log; exit 1; fi
for v in t_q t_s2; do
  for shape in "25600 768 768 0 0" "25600 768 768 0 1" "22016 2048 512 0 0"; do
    echo "=== $v $shape" >> $O/trace.txt
    LD_LIBRARY_PATH=$PWD/build/$v timeout 120 tools/gemm_trace $shape 10 2>&1 | head -4 | tail -2 >> $O/trace.txt
  done
done
cut -c1-200 $O/trace.txt
for i in 1 2; do
  for v in base sched1 new; do
    if [ $v = new ]; then L=$PWD/declip_amd/libdeclip_hip.so; else L=$PWD/build/$v/libdeclip_hip.so; fi
    DECLIP_HIP_LIB=$L DH_BENCH_GEMM_TABLE=$O/table_${v}_$i.txt timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-loss-delta > $O/bench_${v}_$i.log 2>&1
  done
done
for f in $O/bench_*.log; do echo $f; python - "$f" <<'PY'
import json,sys
for l in open(sys.argv[1]):
    if l.startswith('{'):
        d=json.loads(l); r=d.get('roofline',{})
        print("   %.1f pairs/s  %.3f ms/step | GEMM %.3f ms/step %.1f TF/s" % (d['value'], d['ms_per_step'], r.get('gemm_ms_per_step',0), r.get('achieved',0)))
PY
done
