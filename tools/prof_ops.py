import os, sys
sys.path.insert(0, "/root/repo")
import torch
from torch.profiler import profile, ProfilerActivity
from declip_amd import synth
from declip_amd.loss import ClipInfoCELoss
from declip_amd.optim import build_adamw
from declip_amd.testing import build_clip
b=512; cfg=synth.VITB32
model = build_clip(cfg, dtype="bf16", seed=0, load_synth=False)
opt = build_adamw(model, lr=1e-4, betas=(0.9,0.98), eps=1e-8, weight_decay=0.1)
crit = ClipInfoCELoss()
batch = {"images": synth.synth_images(b, seed=0).cuda(), "captions": synth.synth_tokens(b, seed=0).cuda()}
def step():
    opt.zero_grad()
    li, lt = model(batch); loss,_ = crit(li, lt)
    loss.backward(); model.logit_scale.data.clamp_(3,6); opt.step(); model.logit_scale.data.clamp_(3,6)
for _ in range(3): step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    step(); torch.cuda.synchronize()
ka = prof.key_averages(group_by_stack_n=6)
rows = [e for e in ka if e.key.startswith("aten::") and e.count >= 1 and (e.device_time_total > 0)]
rows.sort(key=lambda e: -e.device_time_total)
for e in rows[:40]:
    st = [s for s in e.stack if "declip_amd" in s or "bench" in s or "prof_ops" in s][:2]
    print("%-28s n=%3d dev=%8.1fus cpu=%8.1fus | %s" % (e.key, e.count, e.device_time_total, e.cpu_time_total, " <- ".join(s.split("/")[-1] for s in st)))
