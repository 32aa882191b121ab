"""The M = b GEMMs of the pooled last block (512 rows): which kernel family is fastest?  v4 (256 x 256 tiles cut in K over the chip
+ fix-up launch) vs the 128 x 128 LDS-DMA kernel vs the register-staged MFMA kernel, per-launch time on an otherwise idle chip.
    python tools/bench_small_gemm.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from declip_amd import ops  # noqa: E402
from declip_amd.engine import gemm_workspace  # noqa: E402
from declip_amd.lib import EPI_DGELU, EPI_GELU, EPI_NONE  # noqa: E402

dev, bf = torch.device("cuda"), torch.bfloat16
ws = gemm_workspace(dev)
SHAPES = [(512, 768, 768, 0, 0, 0), (512, 768, 768, 0, 0, 1), (512, 3072, 768, 0, 1, 0), (512, 768, 3072, 0, 0, 1), (512, 3072, 768, 1, 2, 0),
          (512, 768, 3072, 1, 0, 0), (512, 768, 768, 1, 0, 0), (512, 512, 512, 0, 0, 1), (512, 2048, 512, 0, 1, 0), (512, 512, 2048, 0, 0, 1),
          (512, 512, 2048, 1, 0, 0), (512, 2048, 512, 1, 2, 0)]


def timed(fn, iters=50):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    e1.synchronize()
    return e0.elapsed_time(e1) * 1e3 / iters


print("%5s %5s %5s tb epi res | %10s %10s %10s   (us per launch: v4 sliced | glds 128 | auto)" % ("M", "N", "K", "v4", "glds128", "auto"))
for M, N, K, tb, epi, res in SHAPES:
    A = (torch.rand(M, K, device=dev) - 0.5).to(bf)
    B = ((torch.rand((K, N) if tb else (N, K), device=dev) - 0.5) * 0.1).to(bf)
    bias = None if epi == 2 else torch.rand(N, device=dev)
    r = (torch.rand(M, N, device=dev) - 0.5).to(bf) if res else None
    aux = torch.empty(M, N, device=dev, dtype=bf) if epi == 1 else ((torch.rand(M, N, device=dev) - 0.5).to(bf) if epi == 2 else None)
    out = torch.empty(M, N, device=dev, dtype=bf)
    row = []
    for fg in (4, 3, 0):
        def f():
            ops.gemm(A, B, b_kmajor=bool(tb), bias=bias, epilogue={0: EPI_NONE, 1: EPI_GELU, 2: EPI_DGELU}[epi], residual=r, aux=aux, out=out,
                     force_generic=fg, ws=ws)
        try:
            row.append("%10.1f" % timed(f))
        except Exception as e:  # noqa: BLE001
            row.append("%10s" % "n/a")
    print("%5d %5d %5d %2d %3d %3d | %s" % (M, N, K, tb, epi, res, " ".join(row)), flush=True)
