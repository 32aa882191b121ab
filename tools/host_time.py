"""Host-side cost of one training step: wall time for step() to RETURN (all launches enqueued) with an idle GPU in front,
against the GPU time of the step.  If the two are close the step is launch-bound and stream overlap cannot help further."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from declip_amd import synth
from declip_amd.loss import ClipInfoCELoss
from declip_amd.optim import build_adamw
from declip_amd.testing import build_clip
b = int(os.environ.get("B", "512"))
cfg = synth.VITB32
model = build_clip(cfg, dtype="bf16", seed=0, load_synth=False)
opt = build_adamw(model, lr=1e-4, betas=(0.9, 0.98), eps=1e-8, weight_decay=0.1)
crit = ClipInfoCELoss()
batch = {"images": synth.synth_images(b, seed=0).cuda(), "captions": synth.synth_tokens(b, seed=0).cuda()}
def step():
    opt.zero_grad()
    li, lt = model(batch)
    loss, _ = crit(li, lt)
    t1 = time.perf_counter()
    loss.backward()
    t2 = time.perf_counter()
    opt.step()
    return t1, t2
for _ in range(5):
    step()
torch.cuda.synchronize()
host, fwd, bwd, gpu = [], [], [], []
for _ in range(10):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    t1, t2 = step()
    t3 = time.perf_counter()
    torch.cuda.synchronize()
    t4 = time.perf_counter()
    host.append(t3 - t0); fwd.append(t1 - t0); bwd.append(t2 - t1); gpu.append(t4 - t0)
med = lambda v: sorted(v)[len(v) // 2] * 1e3
print("b=%d host enqueue %.2f ms (forward+loss %.2f, backward %.2f, optimizer %.2f) | step incl. GPU drain %.2f ms" % (
    b, med(host), med(fwd), med(bwd), med(host) - med(fwd) - med(bwd), med(gpu)))
