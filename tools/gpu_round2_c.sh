#!/bin/bash
# GPU session C of round 2: fused FILIP max-sim (MODE_MAXSIM epilogue) -- kernel test, the FILIP goldens, FILIP / SLIP / DeFILIP lines.
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_clip.py tests/test_gpu_golden_fullwidth.py tests/test_gpu_dist.py -m gpu -q -k "maxsim or filip" > gpurun_out/pytest_filip.txt 2>&1; echo "rc=$?" >> gpurun_out/pytest_filip.txt
tail -25 gpurun_out/pytest_filip.txt
for fused in 0 1; do
  echo "== FILIP DH_MAXSIM_FUSED=$fused" >> gpurun_out/bench_filip.txt
  DH_MAXSIM_FUSED=$fused timeout 300 python bench.py --model filip --steps 10 --warmup 3 --no-cpu-baseline 2>&1 | grep '^{' >> gpurun_out/bench_filip.txt
done
timeout 300 python bench.py --model slip --steps 10 --warmup 3 --no-cpu-baseline 2>&1 | grep '^{' > gpurun_out/bench_slip.txt
timeout 300 python bench.py --model defilip --steps 10 --warmup 3 --no-cpu-baseline 2>&1 | grep '^{' > gpurun_out/bench_defilip.txt
python - <<'PY'
import json
for f in ("bench_filip", "bench_slip", "bench_defilip"):
    for l in open("gpurun_out/%s.txt" % f):
        if l.startswith("{"):
            j = json.loads(l); r = j.get("roofline") or {}
            print(f, "%.1f pairs/s %.2f ms loss %.4f | GEMM %.0f TF %.2f ms" % (j["value"], j["ms_per_step"], j["loss"], r.get("achieved", 0), r.get("gemm_ms_per_step", 0)))
        else:
            print(l.strip())
PY
