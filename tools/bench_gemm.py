"""Micro-benchmark of the tower GEMM shapes (per-launch TFLOP/s), v2 tile 128 vs 256 vs v1.
    python tools/bench_gemm.py [--iters 20]
"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from declip_amd import ops  # noqa: E402
from declip_amd.engine import _split_k  # noqa: E402


def run(fn, iters):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--quick", action="store_true")
    args = ap.parse_args()
    dev = torch.device("cuda")
    bf = torch.bfloat16
    towers = [("vis", 512 * 50, 768), ("txt", 512 * 77, 512)]
    cases = []
    for name, rows, d in towers:
        for lname, n, k in (("qkv", 3 * d, d), ("out", d, d), ("fc", 4 * d, d), ("proj", d, 4 * d)):
            cases.append(("%s.%s.fwd" % (name, lname), "NT", rows, n, k))
            cases.append(("%s.%s.dX" % (name, lname), "NN", rows, k, n))
            cases.append(("%s.%s.dW" % (name, lname), "TT", n, k, rows))
    cases.append(("vis.patch.fwd", "NT", 512 * 49, 768, 3072))
    if args.quick:
        cases = [c for c in cases if c[0].startswith("vis.fc") or c[0].startswith("txt.out")]
    print("%-16s %-3s %7s %6s %6s | %s" % ("case", "lay", "M", "N", "K", "  ".join("%14s" % v for v in ("v2/128 TF", "v4/256x256"))))
    tot = {}
    for name, lay, M, N, K in cases:
        if lay == "NT":
            A, B = torch.randn(M, K, device=dev).to(bf), torch.randn(N, K, device=dev).to(bf)
            kw = dict()
            out = torch.empty(M, N, device=dev, dtype=bf)
        elif lay == "NN":
            A, B = torch.randn(M, K, device=dev).to(bf), torch.randn(K, N, device=dev).to(bf)
            kw = dict(b_kmajor=True)
            out = torch.empty(M, N, device=dev, dtype=bf)
        else:
            A, B = torch.randn(K, M, device=dev).to(bf), torch.randn(K, N, device=dev).to(bf)
            kw = dict(a_kmajor=True, b_kmajor=True, accumulate=True, split_k=_split_k(M, N, K))
            out = torch.zeros(M, N, device=dev, dtype=torch.float32)
        res = []
        for variant in ("v2", "v4"):
            fg = {"v2": 3, "v4": 4}[variant]
            fn = lambda: ops.gemm(A, B, out=out, force_generic=fg, **kw)
            ms = run(fn, args.iters)
            tf = 2.0 * M * N * K / ms / 1e9
            res.append((ms, tf))
            tot[variant] = tot.get(variant, 0.0) + ms
        print("%-16s %-3s %7d %6d %6d | %s" % (name, lay, M, N, K, "  ".join("%6.3fms %6.0f" % r for r in res)))
    print("sum ms:", {k: round(v, 3) for k, v in tot.items()})


if __name__ == "__main__":
    main()
