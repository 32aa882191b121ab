"""The M = 512 GEMMs of the pooled last blocks (b pooled rows against the block's weights): gemm_v4's few-tile path (tiles cut in K over the
chip + a fix-up launch) against the other kernel families on the same shapes, us per call in a captured graph of 20 dependent calls
(what the step pays: no host launch overhead, kernel-to-kernel dependency latency included).

    python tools/bench_small_gemm.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from declip_amd import engine, ops

dev = torch.device("cuda", 0)
torch.manual_seed(0)
EPI_NONE, EPI_GELU, EPI_DGELU = 0, 1, 2
# (M, N, K, b_kmajor, epilogue, residual)
SHAPES = [(512, 768, 768, 0, 0, 1), (512, 3072, 768, 0, 1, 0), (512, 768, 3072, 0, 0, 1), (512, 512, 768, 0, 0, 0),
          (512, 768, 3072, 1, 0, 0), (512, 3072, 768, 1, 2, 0), (512, 768, 768, 1, 0, 0),
          (512, 512, 512, 0, 0, 1), (512, 2048, 512, 0, 1, 0), (512, 512, 2048, 0, 0, 1), (512, 512, 2048, 1, 0, 0), (512, 2048, 512, 1, 2, 0)]
ws = engine.gemm_workspace(dev)


def run(shape, family):
    M, N, K, tb, epi, res = shape
    A = (torch.randn(M, K, device=dev) * 0.5).bfloat16()
    B = (torch.randn(K, N, device=dev) * 0.05).bfloat16() if tb else (torch.randn(N, K, device=dev) * 0.05).bfloat16()
    bias = None if tb else torch.randn(N, device=dev)
    R = (torch.randn(M, N, device=dev)).bfloat16() if res else None
    aux = torch.empty(M, N, device=dev, dtype=torch.bfloat16) if epi == EPI_GELU else ((torch.rand(M, N, device=dev)).bfloat16() if epi == EPI_DGELU else None)
    out = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    kw = dict(b_kmajor=bool(tb), bias=bias, epilogue=epi, residual=R, aux=aux, out=out)
    if family == "v4":
        kw.update(ws=ws)
    elif family == "v4_nosplit":
        pass
    else:
        kw.update(force_generic=family)

    def call():
        ops.gemm(A, B, **kw)
    try:
        for _ in range(3):
            call()
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            for _ in range(20):
                call()
        g.replay()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            g.replay()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) * 1e3 / 200
    except Exception as e:
        return float("nan")


fams = [("v4", "v4 few-tile"), ("v4_nosplit", "v4 no split"), (31, "v3 mode 31"), (32, "v3 mode 32"), (33, "v3 mode 33"), (3, "glds (v2)"), (2, "v1 mfma")]
print("%-34s" % "M N K tb epi res" + "".join("%14s" % n for _, n in fams))
tot = [0.0] * len(fams)
for s in SHAPES:
    ts = [run(s, f) for f, _ in fams]
    for i, t in enumerate(ts):
        tot[i] += t
    print("%-34s" % (" ".join(str(x) for x in s)) + "".join("%14.1f" % t for t in ts))
print("%-34s" % "sum us" + "".join("%14.1f" % t for t in tot))
