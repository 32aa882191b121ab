#!/bin/bash
# A/B of the opt-in flop-saving switches on one GPU box: same box, back to back, one JSON line each.
#   gpurun --timeout 900 -- 'bash tools/ab_switches.sh > gpurun_out/ab_switches.txt 2>&1'
# Columns: switches | pairs/s | ms/step | GEMM-family TFLOP/s in-step | GEMM ms/step
set -u
cd "$(dirname "$0")/.."
MODEL="${1:-clip}"
for sw in "" "--text-packed 2" "--text-packed 1" "--pooled-last 1" "--text-packed 1 --pooled-last 1"; do
  line=$(python bench.py --model "$MODEL" --steps 10 --warmup 3 --no-cpu-baseline $sw 2>/dev/null | grep '^{' | tail -1)
  python - "$sw" "$line" <<'PY'
import json, sys
sw, line = sys.argv[1], sys.argv[2]
if not line:
    print("%-36s  FAILED" % (sw or "(default)"))
else:
    j = json.loads(line)
    r = j.get("roofline") or {}
    print("%-36s  %9.1f pairs/s  %7.2f ms/step  %7.1f TF  %6.2f ms GEMM  loss %.4f" % (sw or "(default)", j["value"], j["ms_per_step"],
          r.get("achieved", 0.0), r.get("gemm_ms_per_step", 0.0), j["loss"]))
PY
done
