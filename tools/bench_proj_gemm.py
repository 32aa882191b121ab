"""The pooled features' projection GEMMs (b x width -> b x embed, fp32 output) and their backward, per kernel family, us per call in a captured chain.
    python tools/bench_proj_gemm.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from declip_amd import ops

dev = torch.device("cuda", 0)
torch.manual_seed(0)


def timeit(call):
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
    except Exception as e:          # noqa: BLE001
        return float("nan")


for width in (768, 512):
    b, E = 512, 512
    feat = (torch.randn(b, width, device=dev) * 0.5).bfloat16()
    proj = (torch.randn(width, E, device=dev) * 0.05).bfloat16()        # [width, E]: b_kmajor
    dout = (torch.randn(b, E, device=dev) * 0.1).bfloat16()
    gw = torch.zeros(width, E, device=dev)
    print("width %d" % width)
    for fam in (0, 31, 32, 33, 3, 2):
        t_f = timeit(lambda: ops.gemm(feat, proj, b_kmajor=True, out_dtype=torch.float32, force_generic=fam))
        t_dx = timeit(lambda: ops.gemm(dout, proj, force_generic=fam))
        t_dw = timeit(lambda: ops.gemm(feat, dout, a_kmajor=True, b_kmajor=True, out=gw, accumulate=True, force_generic=fam))
        print("  family %2d: fwd (fp32 out) %6.1f us   dfeat %6.1f us   dproj (fp32 accumulate) %6.1f us" % (fam, t_f, t_dx, t_dw))
