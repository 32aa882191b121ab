"""gemm_v6 (round-5 experiment: two 4-wave workgroups per CU, 128 x 256 x 32 tiles; DH_GEMM_V6=1) against gemm_v4 on the forward-layout
tower shapes: correctness against an fp32 torch product first, then interleaved timing.  Tuning aid, not part of the product."""
import os
import statistics
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from declip_amd import ops  # noqa: E402
from declip_amd.lib import EPI_DGELU, EPI_GELU  # noqa: E402

dev, bf = torch.device("cuda"), torch.bfloat16
# (M, N, K, tb, epi, res): forward A[M,K] W[N,K]^T; tb: dX layout (W stored [K][N])
SHAPES = [(25600, 768, 768, 0, 0, 0), (25600, 768, 768, 0, 0, 1), (25600, 2304, 768, 0, 0, 0), (25600, 768, 3072, 0, 0, 1),
          (22016, 1536, 512, 0, 0, 0), (22016, 512, 2048, 0, 0, 1), (22016, 512, 512, 0, 0, 1), (25600, 3072, 768, 0, 1, 0), (22016, 2048, 512, 0, 1, 0),
          (25600, 768, 768, 1, 0, 0), (25600, 768, 2304, 1, 0, 0), (25600, 768, 3072, 1, 0, 0), (25600, 3072, 768, 1, 2, 0), (22016, 512, 512, 1, 0, 0),
          (22016, 512, 1536, 1, 0, 0), (22016, 512, 2048, 1, 0, 0), (22016, 2048, 512, 1, 2, 0)]


def timed(fn, iters=10):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    e1.synchronize()
    return e0.elapsed_time(e1) * 1e3 / iters


def main():
    print("%6s %5s %5s tb epi res | %8s %6s | %8s %6s | v4/v6   max rel err v6 (v4)" % ("M", "N", "K", "v4 us", "TF/s", "v6 us", "TF/s"))
    tot4 = tot6 = 0.0
    for (M, N, K, tb, epi, res) in SHAPES:
        g = torch.Generator(device=dev).manual_seed(M + N + K)
        A = (torch.rand(M, K, device=dev, generator=g) * 2 - 1).to(bf)
        B = (torch.rand((K, N) if tb else (N, K), device=dev, generator=g) * 0.2 - 0.1).to(bf)
        bias = None if epi == 2 else torch.rand(N, device=dev) - 0.5
        R = (torch.rand(M, N, device=dev) - 0.5).to(bf) if res else None
        aux = torch.empty(M, N, device=dev, dtype=bf) if epi == 1 else ((torch.rand(M, N, device=dev) * 2 - 1).to(bf) if epi == 2 else None)
        out = torch.empty(M, N, device=dev, dtype=bf)
        kw = dict(b_kmajor=bool(tb), bias=bias, residual=R, epilogue={0: 0, 1: EPI_GELU, 2: EPI_DGELU}[epi], aux=aux, out=out)

        def run(v6):
            os.environ["DH_GEMM_V6"] = "1" if v6 else "0"
            try:
                ops.gemm(A, B, force_generic=6 if v6 else 4, **kw)
            except Exception as e:       # noqa: BLE001
                return str(e)[:60]
            return None
        err6 = run(True)
        if err6:
            print("%6d %5d %5d %2d %3d %3d | v6 declined: %s" % (M, N, K, tb, epi, res, err6))
            continue
        pre = A.float() @ (B.float() if tb else B.float().t())
        if bias is not None:
            pre = pre + bias
        if epi == 1:
            ref = pre * torch.sigmoid(1.702 * pre)
        elif epi == 2:
            ref = pre * aux.float()
        else:
            ref = pre
        if R is not None:
            ref = ref + R.float()
        scale = float(ref.abs().max())
        e6 = float((out.float() - ref).abs().max()) / scale
        run(False)
        e4 = float((out.float() - ref).abs().max()) / scale
        for _ in range(3):
            run(True); run(False)
        torch.cuda.synchronize()
        t4, t6 = [], []
        for _ in range(5):
            t4.append(timed(lambda: run(False)))
            t6.append(timed(lambda: run(True)))
        m4, m6 = statistics.median(t4), statistics.median(t6)
        tot4 += m4; tot6 += m6
        fl = 2.0 * M * N * K
        print("%6d %5d %5d %2d %3d %3d | %8.1f %6.0f | %8.1f %6.0f | %5.2f    %.2e (%.2e)" % (M, N, K, tb, epi, res, m4, fl / m4 / 1e6, m6, fl / m6 / 1e6, m4 / m6, e6, e4), flush=True)
    print("# sum us: v4 %.0f, v6 %.0f" % (tot4, tot6))


if __name__ == "__main__":
    main()
