"""gemm_v5 (256 x 128 tiles, epilogue inside the next tile's K loop) against gemm_v4 on the tower shapes: results + time.
    DH_GEMM_V5=1 python tools/bench_v5.py"""
import os
import sys

os.environ.setdefault("DH_GEMM_V5", "1")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from declip_amd import ops  # noqa: E402

dev = torch.device("cuda")
bf = torch.bfloat16


def rnd(*s, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(*s, generator=g) * scale)


def timeit(fn, iters=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


ws = torch.empty(64 << 20, device=dev, dtype=torch.float32)
cases = [("chk.small", 512, 256, 128, False, True, True), ("chk.small.T", 768, 512, 192, True, True, False), ("chk.mid", 2816, 1280, 704, False, True, True),
         ("vis.out.fwd", 25600, 768, 768, False, True, True), ("vis.out.dX", 25600, 768, 768, True, False, False),
         ("vis.qkv.fwd", 25600, 2304, 768, False, True, False), ("vis.qkv.dX", 25600, 768, 2304, True, False, False),
         ("vis.proj.fwd", 25600, 768, 3072, False, True, True), ("vis.fc.dX", 25600, 768, 3072, True, False, False),
         ("txt.out.fwd", 22016, 512, 512, False, True, True), ("txt.out.dX", 22016, 512, 512, True, False, False),
         ("txt.qkv.fwd", 22016, 1536, 512, False, True, False), ("txt.qkv.dX", 22016, 512, 1536, True, False, False),
         ("txt.proj.fwd", 22016, 512, 2048, False, True, True), ("txt.fc.dX", 22016, 512, 2048, True, False, False)]
print("%-14s %6s %5s %5s tb bias res |  v4 us    TF |  v5 us    TF | v5/v4 | max err v5 vs v4 (ref max)" % ("case", "M", "N", "K"))
for name, M, N, K, tb, use_bias, use_res in cases:
    A = rnd(M, K, seed=1).to(bf).to(dev)
    B = (rnd(K, N, seed=2, scale=0.2) if tb else rnd(N, K, seed=2, scale=0.2)).to(bf).to(dev)
    bias = rnd(N, seed=3).to(dev) if use_bias else None
    res = rnd(M, N, seed=4).to(bf).to(dev) if use_res else None
    kw = dict(b_kmajor=tb, bias=bias, residual=res, ws=ws)
    o4 = ops.gemm(A, B, force_generic=4, **kw)
    o5 = ops.gemm(A, B, force_generic=5, **kw)
    torch.cuda.synchronize()
    err = float((o5.float() - o4.float()).abs().max())
    ref = float(o4.float().abs().max())
    # race screen: the same launch 5 times must give the same bits
    same = all(torch.equal(ops.gemm(A, B, force_generic=5, **kw), o5) for _ in range(5))
    t4 = timeit(lambda: ops.gemm(A, B, force_generic=4, **kw))
    t5 = timeit(lambda: ops.gemm(A, B, force_generic=5, **kw))
    fl = 2.0 * M * N * K
    print("%-14s %6d %5d %5d %2d %4d %3d | %6.1f %5.0f | %6.1f %5.0f | %5.2f | %.4g (%.3g) %s" % (
        name, M, N, K, tb, use_bias, use_res, t4, fl / t4 / 1e6, t5, fl / t5 / 1e6, t4 / t5, err, ref, "" if same else "NONDETERMINISTIC"))
