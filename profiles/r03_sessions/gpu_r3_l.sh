#!/bin/bash
# round 3, GPU session L: deferred LayerNorm-gradient reduce (one launch per tower) + packed-tail zeroing inside the varlen attention
# kernels: parity tests, then same-box A/B of DH_LN_BATCH
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/r3l
mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
( timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_resnet_intake_packed.py tests/test_gpu_clip.py tests/test_gpu_graph.py -x -q -m gpu -k "layernorm or varlen or packed or clip_fp32 or bf16 or graph or three_adamw or adamw" ) > $O/t1.log 2>&1
echo "t1 rc=$?" >> $O/t1.log
tail -3 $O/t1.log
( timeout 900 python -m pytest tests/test_gpu_dist.py tests/test_gpu_golden_fullwidth.py -x -q -m gpu -k "two_ranks_on_one_gpu or one_rank or clip_vitb32 or declip_vitb32" ) > $O/t2.log 2>&1
echo "t2 rc=$?" >> $O/t2.log
tail -3 $O/t2.log
for i in 1 2; do
  for v in 0 1; do
    DH_LN_BATCH=$v timeout 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-loss-delta --no-roofline > $O/bench_ln${v}_$i.log 2>&1
  done
done
for f in $O/bench_*.log; do echo $f; python - "$f" <<'PY'
import json,sys
for l in open(sys.argv[1]):
    if l.startswith('{'):
        d=json.loads(l)
        print("   %.1f pairs/s  %.3f ms/step" % (d['value'], d['ms_per_step']))
PY
done
