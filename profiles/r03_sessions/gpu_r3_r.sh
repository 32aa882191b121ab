#!/bin/bash
# round 3, GPU session R: the numbers of the round on ONE box -- LN forward variant, bench lines of all models, kernel table + per-step
# timeline + PMC passes of the CLIP step
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/r3r
mkdir -p $O
echo "== LN fwd: 4 rows per wave for d <= 512 (library) vs 2 (build/ln_r2)" > $O/ln.txt
BENCH_SMALL=ln python tools/bench_small.py 2>&1 | grep LN >> $O/ln.txt
echo "-- 2 rows" >> $O/ln.txt
DECLIP_HIP_LIB=build/ln_r2/libdeclip_hip.so BENCH_SMALL=ln python tools/bench_small.py 2>&1 | grep LN >> $O/ln.txt
cat $O/ln.txt
python tools/bench_small.py 2>&1 | grep -v amdgpu > $O/small.txt
( time python bench.py --steps 40 --warmup 8 ) > $O/bench_clip.txt 2>&1
python bench.py --steps 40 --warmup 8 --no-cpu-baseline --pipeline 1 > $O/bench_clip_pipeline.txt 2>&1
for m in declip slip filip defilip; do
  timeout 400 python bench.py --model $m --steps 20 --warmup 6 --no-cpu-baseline > $O/bench_$m.txt 2>&1
done
timeout 300 python bench.py --model clip_r50 --dtype fp32 --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_r50_fp32.txt 2>&1
for f in $O/bench_*.txt; do python - $f <<'PY'
import json,sys
ls=[x for x in open(sys.argv[1]) if x.startswith('{')]
if not ls: print(sys.argv[1], "NO LINE"); sys.exit()
d=json.loads(ls[-1]); print(sys.argv[1], d["value"], d["ms_per_step"], d.get("host_ms_per_step"), (d.get("roofline") or {}).get("frac"), (d.get("cpu_baseline") or {}).get("value"))
PY
done
bash tools/profile_step.sh > $O/profile.log 2>&1
cp gpurun_out/prof/stats.txt $O/stats.txt 2>/dev/null
cp gpurun_out/prof/pmc_summary.txt $O/pmc_summary.txt 2>/dev/null
grep -A12 "per step" $O/stats.txt | head -14
tail -8 $O/pmc_summary.txt | cut -c1-300
