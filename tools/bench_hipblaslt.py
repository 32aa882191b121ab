"""Vendor yardstick for the tower GEMMs: torch.matmul in bf16 (hipBLASLt / rocBLAS behind it) against
`gemm_v4` on the 18 tower shapes of the CLIP ViT-B/32 b = 512 step, same box, same process, interleaved rounds, random data.

CALIBRATION ONLY -- nothing in the product path calls a vendor GEMM; this tool lives in tools/ and is not imported by the package.

    python tools/bench_hipblaslt.py [--rounds 5] [--iters 10] [--out gpurun_out/r04_gemm_vs_hipblaslt.txt]

Columns: the in-step flavour of each shape (layout, epilogue, residual) as `profiles/r03_gemm_table_clip.txt` lists it;
`blaslt` = torch.matmul with NO epilogue (what the vendor kernel reaches on the bare product; its bias / activation / residual
would be extra), `v4 bare` = our kernel on the bare product, `v4 epi` = our kernel with the epilogue the step uses.
Medians over the rounds; TF/s = 2 M N K / time.
"""
import argparse
import os
import statistics
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from declip_amd import ops  # noqa: E402
from declip_amd.lib import EPI_DGELU, EPI_GELU, EPI_NONE  # noqa: E402

# (M, N, K, ta, tb, epi, res): forward = A[M,K] W[N,K]^T (+bias); dX = dY[M,K] W[K,N]; rows of r03_gemm_table_clip.txt, M >= 22016
SHAPES = [
    (25600, 768, 3072, 0, 0, 0, 1), (25600, 3072, 768, 0, 0, 1, 0), (25600, 3072, 768, 0, 1, 2, 0), (25600, 768, 3072, 0, 1, 0, 0),
    (25600, 768, 2304, 0, 1, 0, 0), (25600, 2304, 768, 0, 0, 0, 0), (22016, 2048, 512, 0, 1, 2, 0), (22016, 2048, 512, 0, 0, 1, 0),
    (25600, 768, 768, 0, 0, 0, 1), (22016, 512, 2048, 0, 0, 0, 1), (22016, 1536, 512, 0, 0, 0, 0), (22016, 512, 2048, 0, 1, 0, 0),
    (25600, 768, 768, 0, 1, 0, 0), (22016, 512, 1536, 0, 1, 0, 0), (22016, 512, 512, 0, 0, 0, 1), (22016, 512, 512, 0, 1, 0, 0),
    # weight gradients (single problems; the step runs them grouped): dW[N_out, K_in] = dY[rows, out]^T X[rows, in]
    (2304, 768, 25600, 1, 1, 0, 0), (3072, 768, 25600, 1, 1, 0, 0),
]


def timed(fn, iters):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    e1.synchronize()
    return e0.elapsed_time(e1) * 1e3 / iters      # us


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rounds", type=int, default=5)
    ap.add_argument("--iters", type=int, default=10)
    ap.add_argument("--out", default="")
    ap.add_argument("--no-ws", action="store_true", help="no workspace for the forward / dX launches (whole tiles only, as round 4 measured)")
    ap.add_argument("--only-n", type=int, default=0, help="only the shapes with this N (quick A/B runs)")
    args = ap.parse_args()
    dev, bf = torch.device("cuda"), torch.bfloat16
    lines = ["# torch %s, %s; rounds %d x iters %d, interleaved, uniform random [-1, 1) operands" % (
        torch.__version__, torch.cuda.get_device_name(0), args.rounds, args.iters),
        "# blaslt = torch.matmul bf16 (bare product); v4 bare = gemm_v4 bare product; v4 epi = gemm_v4 with the step's epilogue",
        "%6s %5s %6s %2s %2s %3s %3s | %9s %7s | %9s %7s | %9s %7s | %s" % (
            "M", "N", "K", "ta", "tb", "epi", "res", "blaslt us", "TF/s", "v4bare us", "TF/s", "v4epi us", "TF/s", "v4epi/blaslt")]
    tot = [0.0, 0.0, 0.0]
    for (M, N, K, ta, tb, epi, res) in SHAPES:
        if args.only_n and N != args.only_n:
            continue
        g = torch.Generator(device=dev).manual_seed(M + N + K)
        A = (torch.rand((K, M) if ta else (M, K), device=dev, generator=g) * 2 - 1).to(bf)
        B = (torch.rand((K, N) if tb else (N, K), device=dev, generator=g) * 2 - 1).to(bf)
        At, Bt = (A.t() if ta else A), (B if tb else B.t())           # views: torch picks the transposed vendor kernel
        outv = torch.empty(M, N, device=dev, dtype=bf)
        dw = bool(ta)
        out = torch.zeros(M, N, device=dev, dtype=torch.float32) if dw else torch.empty(M, N, device=dev, dtype=bf)
        bias = None if dw else torch.rand(N, device=dev) - 0.5
        resid = (torch.rand(M, N, device=dev) - 0.5).to(bf) if res else None
        aux = torch.empty(M, N, device=dev, dtype=bf) if epi == 1 else ((torch.rand(M, N, device=dev) * 2 - 1).to(bf) if epi == 2 else None)
        from declip_amd.engine import gemm_workspace
        kw = dict(a_kmajor=bool(ta), b_kmajor=bool(tb), force_generic=4)
        wsk = {} if (dw or args.no_ws) else dict(ws=gemm_workspace(dev))      # the step hands every tower GEMM the workspace (K-sliced tails)

        def f_vendor():
            torch.matmul(At, Bt, out=outv)

        def f_bare():
            if dw:
                from declip_amd.engine import _split_k
                ops.gemm(A, B, out=out, accumulate=True, split_k=_split_k(M, N, K), **kw)
            else:
                ops.gemm(A, B, out=out, **kw, **wsk)

        def f_epi():
            if dw:
                return f_bare()
            ops.gemm(A, B, out=out, bias=bias if epi != 2 else None, epilogue={0: EPI_NONE, 1: EPI_GELU, 2: EPI_DGELU}[epi],
                     residual=resid, aux=aux, **kw, **wsk)

        fns = (f_vendor, f_bare, f_epi)
        for f in fns:
            for _ in range(3):
                f()
        torch.cuda.synchronize()
        ts = [[], [], []]
        for _ in range(args.rounds):
            for i, f in enumerate(fns):
                ts[i].append(timed(f, args.iters))
        med = [statistics.median(t) for t in ts]
        tf = [2.0 * M * N * K / (m * 1e-6) / 1e12 for m in med]
        for i in range(3):
            tot[i] += med[i]
        lines.append("%6d %5d %6d %2d %2d %3d %3d | %9.1f %7.0f | %9.1f %7.0f | %9.1f %7.0f | %.2f" % (
            M, N, K, ta, tb, epi, res, med[0], tf[0], med[1], tf[1], med[2], tf[2], med[0] / med[2]))
        print(lines[-1], flush=True)
        del A, B, outv, out, resid, aux
    lines.append("# sum us: blaslt %.0f, v4 bare %.0f, v4 epi %.0f" % tuple(tot))
    print(lines[-1])
    if args.out:
        os.makedirs(os.path.dirname(args.out) or ".", exist_ok=True)
        with open(args.out, "w") as f:
            f.write("\n".join(lines) + "\n")


if __name__ == "__main__":
    main()
