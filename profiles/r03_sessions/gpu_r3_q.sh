#!/bin/bash
# session Q: does the K-sliced tail of the N = d GEMMs still pay for its 46 fix-up launches per step?
cd /root/repo
B="python bench.py --steps 40 --warmup 8 --no-cpu-baseline --no-loss-delta --no-roofline"
for i in 1 2; do
for t in 1 0; do
  DH_V4_TAIL=$t $B > /tmp/q_$t.txt 2>&1
  python - /tmp/q_$t.txt $t <<'PY'
import json,sys
l=[x for x in open(sys.argv[1]) if x.startswith('{')][-1]
d=json.loads(l); print("DH_V4_TAIL=%s"%sys.argv[2], d["value"], d["ms_per_step"], d["host_ms_per_step"])
PY
done
done
