"""Which Python call sites launch the small copy / fill kernels of a CLIP step?  (torch.profiler with stacks, one eager step)"""
import collections
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.profiler import ProfilerActivity, profile

from declip_amd import synth
from declip_amd.loss import ClipInfoCELoss
from declip_amd.optim import build_adamw
from declip_amd.testing import build_clip

b = 512
model = build_clip(synth.VITB32, dtype="bf16", seed=0, load_synth=False)
opt = build_adamw(model, lr=1e-4, betas=(0.9, 0.98), weight_decay=0.1)
crit = ClipInfoCELoss()
batch = {"images": synth.synth_images(b, seed=0).cuda(), "captions": synth.synth_tokens(b, seed=0).cuda()}


def step():
    opt.zero_grad()
    li, lt = model(batch)
    loss, _ = crit(li, lt)
    loss.backward()
    opt.step()


for _ in range(3):
    step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True, record_shapes=True) as prof:
    step()
    torch.cuda.synchronize()
ev = prof.events()
by = collections.Counter()
sites = collections.defaultdict(collections.Counter)
for e in ev:
    name = e.name
    if name.startswith("aten::") and any(k in name for k in ("copy_", "fill_", "zero_", "clone", "zeros", "contiguous", "cat", "add", "index", "to", "_to_copy", "mul", "where", "arange", "stack", "select", "slice")):
        st = [s for s in (e.stack or []) if "declip_amd" in s or "bench" in s]
        site = st[0] if st else "(no declip frame)"
        if e.device_time_total > 0 or name in ("aten::copy_", "aten::fill_", "aten::zero_"):
            sites[name][site] += 1
            by[name] += 1
for name, n in by.most_common(14):
    print("== %s x %d" % (name, n))
    for site, c in sites[name].most_common(8):
        print("     %3d  %s" % (c, site))
kern = collections.Counter()
for e in ev:
    if e.device_type is not None and str(e.device_type).endswith("CUDA"):
        kern[e.name[:70]] += 1
print("== device activities")
for k, c in kern.most_common(30):
    print("     %4d  %s" % (c, k))
