"""Bisect: which backward op of the step cannot be captured in a hipGraph?  One process per op."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import os, sys
sys.path.insert(0, %r)
import torch
from declip_amd import ops, engine
from declip_amd.graph import GraphedStep
what = sys.argv[1]
dev = "cuda"
f32 = torch.float32
def R(*s): return torch.randn(*s, device=dev)
b, L, d, heads = 8, 16, 128, 2
if what == "ln_bwd":
    x, dy, w = R(b*L, d), R(b*L, d), torch.ones(d, device=dev); _, mean, rstd = ops.layernorm_fwd(x, w, torch.zeros(d, device=dev))
    dw, db = torch.zeros(d, device=dev), torch.zeros(d, device=dev)
    fn = lambda: ops.layernorm_bwd(dy, x, w, mean, rstd, dw, db).sum()
elif what == "gemm_acc":
    dy, x, gw, gb = R(b*L, d), R(b*L, 64), torch.zeros(d, 64, device=dev), torch.zeros(d, device=dev)
    fn = lambda: (engine.weight_grad(dy, x, gw, gb), gw.sum())[1]
elif what == "gemm_dx":
    dy, w = R(b*L, d), R(d, 64)
    fn = lambda: ops.gemm(dy, w, b_kmajor=True).sum()
elif what == "attn_bwd":
    qkv = R(b*L, 3*d); a, lse = ops.attn_fwd(qkv, b, L, heads, True); da = R(b*L, d)
    fn = lambda: ops.attn_bwd(qkv, a, da, lse, b, L, heads, True).sum()
elif what == "nce":
    from declip_amd.loss import ClipInfoCELoss
    from declip_amd.model.clip import LazyLogits
    q = torch.nn.functional.normalize(R(b, 64), dim=-1).requires_grad_(True); k = torch.nn.functional.normalize(R(b, 64), dim=-1).requires_grad_(True)
    sc = torch.tensor(10.0, device=dev, requires_grad=True)
    def fn():
        q.grad = k.grad = sc.grad = None
        loss, _ = ClipInfoCELoss()(LazyLogits(q, k, sc, 0), LazyLogits(k, q, sc, 0)); loss.backward(); return loss.detach()
elif what == "text_embed_bwd":
    ids = torch.randint(1, 1000, (b, L), device=dev); dx = R(b*L, d); gt, gp = torch.zeros(49409, d, device=dev), torch.zeros(L, d, device=dev)
    fn = lambda: (ops.text_embed_bwd(ids, dx, gt, gp, hot_ids=(0, 49407, 49408)), gp.sum())[1]
elif what == "vit_assemble_bwd":
    dx = R(b*5, d); gc, gp = torch.zeros(d, device=dev), torch.zeros(5, d, device=dev)
    fn = lambda: (ops.vit_assemble_bwd(dx, gc, gp, b, 4), gp.sum())[1]
elif what == "pool_l2":
    x = R(b, 64); y, n = ops.l2norm_fwd(x, 0.0); dy = R(b, 64)
    fn = lambda: ops.l2norm_bwd(x, n, dy, 0.0).sum() + ops.pool_rows_bwd(R(b, d), None, b, L).sum()
elif what == "zero_event":
    g = torch.zeros(1000, device=dev)
    def fn():
        g.zero_(); ev = torch.cuda.Event(); ev.record(torch.cuda.current_stream()); torch.cuda.current_stream().wait_event(ev); return g.sum()
elif what == "callback":
    w = torch.randn(16, 16, device=dev, requires_grad=True)
    class F(torch.autograd.Function):
        @staticmethod
        def forward(ctx, x): return x * 2
        @staticmethod
        def backward(ctx, g):
            torch.autograd.Variable._execution_engine.queue_callback(lambda: None); return g * 2
    def fn():
        w.grad = None; l = F.apply(w).sum(); l.backward(); return l.detach()
g = GraphedStep(fn, warmup=2)
for i in range(4):
    out = g()
torch.cuda.synchronize()
print("RESULT", what, "ok", float(out))
''' % ROOT

for what in ["ln_bwd", "gemm_acc", "gemm_dx", "attn_bwd", "nce", "text_embed_bwd", "vit_assemble_bwd", "pool_l2", "zero_event", "callback"]:
    p = subprocess.run([sys.executable, "-c", CHILD, what], capture_output=True, text=True, timeout=300)
    res = [l for l in p.stdout.splitlines() if l.startswith("RESULT")]
    tail = " | ".join((p.stderr.strip().splitlines() or [""])[-2:])[:300]
    print("%-18s rc=%4d %s %s" % (what, p.returncode, res[0] if res else "", "" if res else tail), flush=True)
