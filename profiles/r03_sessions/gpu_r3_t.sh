#!/bin/bash
# round 3, GPU session T: final bench lines of the round with the corrected roofline brackets (CLIP default + pipeline + per-shape table)
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/r3t
mkdir -p $O
( time DH_BENCH_GEMM_TABLE=$O/gemm_table.txt python bench.py --steps 40 --warmup 8 ) > $O/bench_clip.txt 2>&1
python bench.py --steps 40 --warmup 8 --no-cpu-baseline --pipeline 1 > $O/bench_clip_pipeline.txt 2>&1
python bench.py > $O/bench_default.txt 2>&1
for f in $O/bench_*.txt; do python - $f <<'PY'
import json,sys
ls=[x for x in open(sys.argv[1]) if x.startswith('{')]
d=json.loads(ls[-1]); print(sys.argv[1], d["value"], d["ms_per_step"], d.get("host_ms_per_step"), (d.get("roofline") or {}).get("frac"), (d.get("cpu_baseline") or {}).get("value"))
PY
done
head -24 $O/gemm_table.txt
