#!/bin/bash
# session O: where does the --pipeline step lose its time?  (host profile of the bench loop)
mkdir -p gpurun_out/r3o
cd /root/repo
B="python bench.py --steps 30 --warmup 8 --no-cpu-baseline"
$B > gpurun_out/r3o/resident.txt 2>&1
$B --graph 0 > gpurun_out/r3o/resident_eager.txt 2>&1
$B --pipeline 1 > gpurun_out/r3o/pipeline.txt 2>&1
python -m cProfile -o /tmp/p.prof bench.py --steps 30 --warmup 8 --pipeline 1 --no-cpu-baseline > gpurun_out/r3o/pipeline_prof_run.txt 2>&1
python - > gpurun_out/r3o/pipeline_prof.txt 2>&1 <<'PY'
import pstats
p = pstats.Stats("/tmp/p.prof")
p.sort_stats("tottime").print_stats(35)
p.sort_stats("cumulative").print_stats(70)
PY
python -m cProfile -o /tmp/q.prof bench.py --steps 30 --warmup 8 --graph 0 --no-cpu-baseline > gpurun_out/r3o/eager_prof_run.txt 2>&1
python - > gpurun_out/r3o/eager_prof.txt 2>&1 <<'PY'
import pstats
p = pstats.Stats("/tmp/q.prof")
p.sort_stats("tottime").print_stats(35)
PY
for f in resident resident_eager pipeline pipeline_prof_run eager_prof_run; do python - gpurun_out/r3o/$f.txt <<'PY'
import json,sys
l=[x for x in open(sys.argv[1]) if x.startswith('{')][-1]
d=json.loads(l); print(sys.argv[1], d["value"], d["ms_per_step"], d["host_ms_per_step"])
PY
done
