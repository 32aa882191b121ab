#!/bin/bash
# round 5, session 15: AdamW with block-contiguous chunks (forward segment scan instead of a binary search per 16-byte group)
cd $GRAFT_REPO_ROOT; O=gpurun_out/r05s15; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_clip.py tests/test_gpu_solver.py -q -p no:cacheprovider -k "adamw or optim or solver or three_steps or steps" > $O/tests.txt 2>&1; tail -3 $O/tests.txt | cut -c1-200
python - <<'PY'
import os, sys, torch
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
from declip_amd import synth
from declip_amd.optim import build_adamw
from declip_amd.testing import build_clip
m = build_clip(synth.VITB32, dtype="bf16", seed=0, load_synth=False)
opt = build_adamw(m, lr=1e-4, betas=(0.9, 0.98), weight_decay=0.1)
flat = m.__dict__["_flat_store"]; flat.flat_g.normal_()
for _ in range(3): opt.step()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20): opt.step()
e1.record(); e1.synchronize()
print("adamw_seg_kernel: %.1f us per step over %d parameters" % (e0.elapsed_time(e1) * 1e3 / 20, flat.total))
PY
bash tools/ab_bench.sh $O/ab "prev:DECLIP_HIP_LIB=$GRAFT_REPO_ROOT/build/prev/libdeclip_hip.so" "new:" "prev:DECLIP_HIP_LIB=$GRAFT_REPO_ROOT/build/prev/libdeclip_hip.so" "new:" 2>&1 | tee $O/ab.txt
