#!/bin/bash
# GPU session V: the model-level tests that go through the batch helpers with host row counts / the cached masked-LM selection
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
timeout 400 python -m pytest tests/test_gpu_clip.py tests/test_gpu_golden_fullwidth.py tests/test_gpu_fullsize.py tests/test_gpu_graph.py -m gpu -q -k "slip or filip or declip or graph" > gpurun_out/pytest_v.txt 2>&1; echo "rc=$?" >> gpurun_out/pytest_v.txt
tail -4 gpurun_out/pytest_v.txt
for m in slip filip; do timeout 200 python bench.py --model $m --steps 10 --warmup 3 --no-cpu-baseline --no-loss-delta --no-roofline 2>&1 | grep '^{' | python -c "
import json,sys
j=json.loads(sys.stdin.read()); print('$m   %.1f pairs/s  %.2f ms/step  loss %.4f' % (j['value'], j['ms_per_step'], j['loss']))"; done
