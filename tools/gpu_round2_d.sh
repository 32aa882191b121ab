#!/bin/bash
# GPU session D of round 2: HIP graph capture of the step (tests + A/B on CLIP-R50 batch 32 and CLIP ViT-B/32), two-rank DeCLIP on
# the communication stream.
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_graph.py tests/test_gpu_dist.py -m gpu -q -k "graph or declip" > gpurun_out/pytest_graph.txt 2>&1; echo "rc=$?" >> gpurun_out/pytest_graph.txt
tail -30 gpurun_out/pytest_graph.txt
for m in "clip_r50 bf16" "clip_r50 fp32" "clip bf16"; do
  set -- $m
  for g in 0 1; do
    echo "== $1 $2 graph=$g" >> gpurun_out/ab_graph.txt
    timeout 300 python bench.py --model $1 --dtype $2 --graph $g --steps 20 --warmup 5 --no-cpu-baseline --no-loss-delta --no-roofline 2>&1 | grep -E '^\{|Error|error' | cut -c1-400 >> gpurun_out/ab_graph.txt
  done
done
python - <<'PY'
import json
for l in open("gpurun_out/ab_graph.txt"):
    if l.startswith("{"):
        try:
            j = json.loads(l); print("   %.1f pairs/s %.2f ms loss %.4f" % (j["value"], j["ms_per_step"], j["loss"]))
        except Exception:
            print("   " + l[:200])
    else:
        print(l.strip())
PY
