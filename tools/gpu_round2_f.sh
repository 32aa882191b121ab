#!/bin/bash
# GPU session F of round 2: the whole `-m gpu` suite, the bench lines of every model family, rocprofv3 kernel table + PMC passes.
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/final
timeout 1800 python -m pytest tests -m gpu -q > gpurun_out/final/pytest_gpu.txt 2>&1; echo "pytest rc=$?" >> gpurun_out/final/pytest_gpu.txt
tail -6 gpurun_out/final/pytest_gpu.txt
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/final/smoke.txt 2>&1; echo "smoke rc=$?" >> gpurun_out/final/smoke.txt; tail -2 gpurun_out/final/smoke.txt
DH_BENCH_GEMM_TABLE=gpurun_out/final/gemm_table_clip.txt timeout 600 python bench.py --steps 20 --warmup 5 > gpurun_out/final/bench_clip.txt 2>gpurun_out/final/bench_clip.err
for m in declip slip filip defilip; do
  timeout 300 python bench.py --model $m --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/final/bench_$m.txt 2>&1
done
timeout 300 python bench.py --model clip_r50 --dtype bf16 --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/final/bench_r50_bf16.txt 2>&1
timeout 300 python bench.py --model clip_r50 --dtype fp32 --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/final/bench_r50_fp32.txt 2>&1
timeout 600 bash tools/profile_step.sh > gpurun_out/final/profile_step.log 2>&1
cp gpurun_out/prof/stats.txt gpurun_out/final/clip_kernel_stats.txt; cp gpurun_out/prof/pmc_summary.txt gpurun_out/final/clip_pmc_summary.txt
# the ResNet-50 step: kernel table
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
DH_TOWER_STREAMS=0 timeout 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_r50 -o trace -- python $R/bench.py --model clip_r50 --steps 4 --warmup 2 --no-cpu-baseline --graph 0 > $R/gpurun_out/final/prof_r50.log 2>&1
DB=$(find $R/gpurun_out/prof_r50 -name "*.db" | head -1)
if [ -n "$DB" ]; then python $R/tools/rocpd_stats.py $DB > $R/gpurun_out/final/r50_kernel_stats.txt 2>&1; fi
rm -rf $R/gpurun_out/prof_r50 $R/gpurun_out/prof/trace
python - <<'PY'
import json, glob
for f in sorted(glob.glob("/root/repo/gpurun_out/final/bench_*.txt")):
    for l in open(f):
        if l.startswith("{"):
            j = json.loads(l); r = j.get("roofline") or {}
            print(f.split("/")[-1], "%.1f pairs/s %.2f ms loss %.4f | GEMM %.0f TF %.2f ms" % (j["value"], j["ms_per_step"], j["loss"], r.get("achieved", 0), r.get("gemm_ms_per_step", 0)))
PY
