"""one configuration of tools/diag_graph.py as a stand-alone script (for rocgdb)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from declip_amd import synth, ops, engine
from declip_amd.graph import GraphedStep
from declip_amd.loss import ClipInfoCELoss
from declip_amd.testing import build_clip
what = sys.argv[1]
if "nomt" in what:
    torch.autograd.set_multithreading_enabled(False)      # backward on the calling thread
cfg, b = synth.TINY, 8
dtype = "bf16" if "bf16" in what else "fp32"
if what.startswith("gemm"):
    A = torch.randn(512, 256, device="cuda").to(torch.bfloat16); B = torch.randn(512, 256, device="cuda").to(torch.bfloat16)
    def fn():
        return ops.gemm(A, B).float().sum()
elif what.startswith("torchonly"):
    w = torch.randn(256, 256, device="cuda", requires_grad=True); x = torch.randn(64, 256, device="cuda")
    def fn():
        w.grad = None
        l = (x @ w).relu().sum(); l.backward(); return l.detach()
elif what.startswith("ln"):
    x = torch.randn(64, 128, device="cuda"); w = torch.ones(128, device="cuda"); bb = torch.zeros(128, device="cuda")
    def fn():
        return ops.layernorm_fwd(x, w, bb)[0].sum()
else:
    model = build_clip(cfg, dtype=dtype, seed=3, fused_loss=("v3" not in what))
    images = synth.synth_images(b, res=cfg["res"], seed=0).cuda()
    ids = synth.synth_tokens(b, ctx=cfg["ctx"], seed=0, vocab=cfg["vocab"]).cuda()
    crit = ClipInfoCELoss()
    def fn():
        if "texttower" in what:
            loss = model.encode_text(ids).float().sum(); loss.backward(); return loss.detach()
        if "vistower" in what:
            loss = model.encode_image(images).float().sum(); loss.backward(); return loss.detach()
        if "feats" in what:
            a, c = model.features(images, ids); loss = (a * c).sum(); loss.backward(); return loss.detach()
        if "v1" in what:
            from declip_amd.model.clip import LazyLogits
            a, c = model.features(images, ids); sc = model.logit_scale_value().detach()
            loss, _ = crit(LazyLogits(a, c, sc, 0), LazyLogits(c, a, sc, 0)); loss.backward(); return loss.detach()
        if "v2" in what:
            a, c = model.features(images, ids); loss = ((a @ c.t()) * model.logit_scale.exp()).logsumexp(1).mean(); loss.backward(); return loss.detach()
        if "v6" in what:
            from declip_amd.model.clip import LazyLogits
            a, c = model.features(images, ids); sc = model.logit_scale.exp()
            loss, _ = crit(LazyLogits(a, c, sc, 0), LazyLogits(c, a, sc, 0)); loss.backward(); return loss.detach()
        li, lt = model({"images": images, "captions": ids})
        loss, _ = crit(li, lt)
        if "fwd" not in what:
            loss.backward()
        return loss.detach()
if "v5" in what:
    import gc
    fn(); fn(); torch.cuda.synchronize(); gc.collect(); torch.cuda.empty_cache()
g = GraphedStep(fn, warmup=2)
for i in range(5):
    out = g()
torch.cuda.synchronize()
print("RESULT", what, "ok", float(out))
