"""Run-to-run variation of a 3-step ViT-B/32 b=256 run with and without the tower streams (tuning/diagnostic aid)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from declip_amd import synth
from declip_amd.loss import ClipInfoCELoss
from declip_amd.optim import build_adamw
from declip_amd.testing import build_clip
cfg, b, seed = synth.VITB32, 256, 5
images = synth.synth_images(b, res=cfg["res"], seed=seed).cuda()
ids = synth.synth_tokens(b, ctx=cfg["ctx"], seed=seed, vocab=cfg["vocab"]).cuda()
def run(mode):
    os.environ["DH_TOWER_STREAMS"] = mode
    model = build_clip(cfg, dtype="bf16", seed=seed)
    opt = build_adamw(model, lr=1e-4, weight_decay=0.1)
    crit = ClipInfoCELoss()
    losses = []
    g_first = None
    for i in range(3):
        li, lt = model({"images": images, "captions": ids})
        if i == 0:
            cap = {}
            li.Q.register_hook(lambda g: cap.__setitem__("d_img_n", g.detach().clone()))
            lt.Q.register_hook(lambda g: cap.__setitem__("d_txt_n", g.detach().clone()))
            cap["img_n"], cap["txt_n"] = li.Q.detach().clone(), lt.Q.detach().clone()
        loss, _ = crit(li, lt)
        opt.zero_grad()
        loss.backward()
        if i == 0:
            torch.cuda.synchronize()
            g_first = {n: p.grad.detach().clone() for n, p in model.named_parameters() if p.grad is not None}
            g_first.update({"~" + k: v for k, v in cap.items()})
        opt.step()
        losses.append(float(loss))
    torch.cuda.synchronize()
    return losses, g_first
res = {}
for tag, mode in (("a0", "0"), ("a1", "1"), ("b1", "1"), ("c1", "1"), ("b0", "0")):
    if tag == "c1":
        torch.cuda.synchronize()
        torch.cuda.empty_cache()
    res[tag] = run(mode)
    print(tag, ["%.7f" % x for x in res[tag][0]], flush=True)
for x, y in (("a0", "b0"), ("a0", "a1"), ("a0", "b1"), ("a0", "c1")):
    rel = []
    for n, g in res[x][1].items():
        d = float((g - res[y][1][n]).abs().max()) / (float(g.abs().max()) + 1e-30)
        rel.append((d, n))
    rel.sort(reverse=True)
    txt = [r for r in rel if "encode_text" in r[1]]
    vis = [r for r in rel if "visual" in r[1]]
    print(x, y, "vision: worst %.2e (%s) median %.2e nonzero %d/%d | text: worst %.2e (%s) median %.2e nonzero %d/%d" % (
        vis[0][0], vis[0][1], vis[len(vis) // 2][0], sum(r[0] > 0 for r in vis), len(vis),
        txt[0][0], txt[0][1], txt[len(txt) // 2][0], sum(r[0] > 0 for r in txt), len(txt)), flush=True)
    print("   ", {r[1]: "%.2e" % r[0] for r in rel if r[1].startswith("~")}, flush=True)
