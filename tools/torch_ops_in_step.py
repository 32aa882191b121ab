"""Which torch (ATen) kernels are left in the CLIP training step, and which line of the host code launches each of them:
one EAGER step (same model / batch / optimizer as bench.py's default line) under torch.profiler with Python stacks.

    python tools/torch_ops_in_step.py [batch]

Every device kernel that is not one of the library's (`dh_*` launches show up under their HIP kernel names) is listed with the
innermost frames of declip_amd / bench code above it.  Used for the launch diet."""
import collections
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.profiler import ProfilerActivity, profile

from declip_amd import dist as dh_dist
from declip_amd import synth
from declip_amd.loss import ClipInfoCELoss
from declip_amd.optim import build_adamw
from declip_amd.testing import build_clip

b = int(sys.argv[1]) if len(sys.argv) > 1 else 512
torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
torch.manual_seed(1234)
model = build_clip(synth.VITB32, dtype="bf16", use_allgather=False, seed=0, load_synth=False)
batch = {"images": synth.synth_images(b, seed=0).to(dev), "captions": synth.synth_tokens(b, seed=0).to(dev)}
wrapped = dh_dist.DistModule(model, sync=False)
opt = build_adamw(model, lr=1e-4, betas=(0.9, 0.98), eps=1e-8, weight_decay=0.1)
crit = ClipInfoCELoss()


def step():
    opt.zero_grad()
    li, lt = wrapped(batch)
    loss, _ = crit(li, lt)
    loss.backward()
    wrapped.sync_gradients()
    model.logit_scale.data.clamp_(3, 6)
    opt.step()
    model.logit_scale.data.clamp_(3, 6)
    return loss.detach()


for _ in range(3):
    step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True, record_shapes=True) as prof:
    step()
    torch.cuda.synchronize()

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
rows = collections.OrderedDict()
n_kernels = 0
for ev in prof.events():
    kernels = [k for k in (ev.kernels or [])]
    if not kernels or not ev.name.startswith("aten::"):
        continue
    # only leaf aten ops (the ones that own the kernel directly): a parent op repeats its children's kernels
    if any(c.kernels for c in (ev.cpu_children or []) if c.name.startswith("aten::")):
        continue
    frames = [f for f in (ev.stack or []) if ("declip_amd" in f or "torch_ops_in_step" in f or "bench.py" in f)]
    where = " <- ".join(f.replace(ROOT + "/", "") for f in frames[:3])
    key = (ev.name, str(ev.input_shapes)[:60], where)
    rows.setdefault(key, [0, 0.0])
    rows[key][0] += 1
    rows[key][1] += sum(k.duration for k in kernels)
    n_kernels += len(kernels)

print("torch kernels in one eager CLIP step (b = %d): %d launches" % (b, n_kernels))
for (name, shapes, where), (n, us) in rows.items():
    print("%3d x %-28s %-60s %7.1f us\n        %s" % (n, name, shapes, us, where))
