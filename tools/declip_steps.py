import os, sys, time
sys.path.insert(0, "/root/repo")
import torch

from declip_amd import synth, dist as dh_dist
from declip_amd.heads import SimsiamLoss
from declip_amd.loss import ClipInfoCELoss
from declip_amd.optim import build_adamw
from declip_amd.steps import declip_loss
from declip_amd.testing import build_declip, declip_batch
cfg = synth.VITB32; b = 512
model = build_declip(cfg, dtype="bf16", seed=0, nn_size=65536, load_synth=False)
batch = declip_batch(cfg, b, seed=0, device=torch.device("cuda"))
wrapped = dh_dist.DistModule(model, sync=False)
opt = build_adamw(model, lr=1e-4, betas=(0.9, 0.98), eps=1e-8, weight_decay=0.1)
crit, sim = ClipInfoCELoss(), SimsiamLoss()
def step():
    opt.zero_grad()
    loss = declip_loss(wrapped, batch, crit, sim, None, world_size=1, with_accuracy=False)["loss"]
    loss.backward(); opt.step()
for _ in range(4): step()
torch.cuda.synchronize()
ts = []
for i in range(30):
    t0 = time.perf_counter(); step(); t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    ts.append((t1 - t0, t2 - t0))
print("host ms:", " ".join("%.0f" % (a * 1e3) for a, _ in ts))
print("step ms:", " ".join("%.0f" % (b * 1e3) for _, b in ts))
import cProfile, pstats, gc
print("gc counts", gc.get_count(), "thresholds", gc.get_threshold())
torch.cuda.synchronize()
pr = cProfile.Profile(); pr.enable()
for _ in range(5): step()
pr.disable(); torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("cumulative").print_stats(28)
