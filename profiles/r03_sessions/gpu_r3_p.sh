#!/bin/bash
# session P: the pipeline line after the host-thread fix, beside the resident line (same box), + the default bench with the CPU baseline
mkdir -p gpurun_out/r3p
cd /root/repo
B="python bench.py --steps 40 --warmup 8"
$B --no-cpu-baseline > gpurun_out/r3p/resident.txt 2>&1
$B --no-cpu-baseline --graph 0 > gpurun_out/r3p/resident_eager.txt 2>&1
$B --no-cpu-baseline --pipeline 1 > gpurun_out/r3p/pipeline.txt 2>&1
$B --no-cpu-baseline > gpurun_out/r3p/resident2.txt 2>&1
$B --no-cpu-baseline --pipeline 1 > gpurun_out/r3p/pipeline2.txt 2>&1
( time python bench.py ) > gpurun_out/r3p/default.txt 2>&1
for f in resident resident_eager pipeline resident2 pipeline2 default; do python - gpurun_out/r3p/$f.txt <<'PY'
import json,sys
l=[x for x in open(sys.argv[1]) if x.startswith('{')][-1]
d=json.loads(l); print(sys.argv[1], d["value"], d["ms_per_step"], d["host_ms_per_step"], d.get("cpu_baseline",{}).get("value"), d.get("cpu_baseline",{}).get("cores"))
PY
done
tail -4 gpurun_out/r3p/default.txt
