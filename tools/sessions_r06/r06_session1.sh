#!/bin/bash
# round 6, session 1: baseline on this round's box (headline + per-shape GEMM table), the batch sweep VERDICT r5 #3 asked for
# (b = 1024 / 2048 / 4096 on one GPU), the HBM-bound kernels, and in-kernel epilogue traces of all four bf16 epilogue flavours.
cd $GRAFT_REPO_ROOT; O=gpurun_out/r06s1; mkdir -p $O
python -m declip_amd.build > /dev/null 2>&1
DH_BENCH_GEMM_TABLE=$O/gemm_table.txt timeout 600 python bench.py --steps 20 --warmup 3 > $O/bench_clip.json 2> $O/bench_clip.err; tail -c 600 $O/bench_clip.json
for b in 1024 2048 4096; do
  timeout 900 python bench.py --batch $b --steps 6 --warmup 2 --no-cpu-baseline --no-loss-delta > $O/bench_clip_b$b.json 2> $O/bench_clip_b$b.err
  python - $O/bench_clip_b$b.json <<'PY' || tail -5 $O/bench_clip_b$b.err
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
r = d.get("roofline") or {}
print("b %5d: %9.1f pairs/s %9.3f ms/step  frac %.4f gemm_ms %.2f peak_mem %s GB graph %s" % (d["config"]["per_gpu_batch"], d["value"], d["ms_per_step"], r.get("frac", 0), r.get("gemm_ms_per_step", 0), d.get("peak_mem_gb"), d["config"].get("step_graph")))
PY
done
BENCH_SMALL=all timeout 600 python tools/bench_small.py > $O/small_kernels.txt 2>&1; cat $O/small_kernels.txt
bash tools/build_trace.sh > /dev/null 2>&1
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 -w -Iinclude tools/gemm_trace.cpp -Ldeclip_amd -ldeclip_hip -ldl -o tools/gemm_trace
for sh in "25600 3072 768 1" "25600 3072 768 2" "25600 768 3072 3" "25600 768 768 3" "25600 768 768 0" "25600 768 768 4" "22016 2048 512 1" "22016 512 512 3" "25600 2304 768 0"; do
  echo "=== $sh"; LD_LIBRARY_PATH=build/trace timeout 120 tools/gemm_trace $sh 0 10
done > $O/trace_epilogue.txt 2>&1
grep -A3 "===" $O/trace_epilogue.txt | cut -c1-260 | head -60
