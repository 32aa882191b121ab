"""Where does the HOST spend an eager CLIP step?  cProfile of a few steps enqueued into an empty queue (b = 512, bf16).
    python tools/host_profile.py [n]"""
import cProfile
import os
import pstats
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from declip_amd import synth  # noqa: E402
from declip_amd.loss import ClipInfoCELoss  # noqa: E402
from declip_amd.optim import build_adamw  # noqa: E402
from declip_amd.testing import build_clip  # noqa: E402

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


for _ in range(4):
    step()
torch.cuda.synchronize()
pr = cProfile.Profile()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 3
for _ in range(n):
    torch.cuda.synchronize()
    pr.enable()
    step()
    pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr)
st.sort_stats("cumulative")
print("per step: total %.2f ms over %d steps" % (st.total_tt / n * 1e3, n))
st.print_stats(45)
