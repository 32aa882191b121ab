import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from declip_amd import ops
dev, bf = torch.device("cuda"), torch.bfloat16
os.environ["DH_GEMM_V6"] = "1"
for (M, N, K) in [(22016, 512, 512), (22016, 512, 2048), (22016, 512, 1024), (5632, 512, 2048), (25600, 768, 768), (8192, 768, 768), (33024, 256, 64)]:
    A3 = (torch.rand(M, K, device=dev) - 0.5).to(bf); B3 = (torch.rand(N, K, device=dev) - 0.5).to(bf)
    o3 = ops.gemm(A3, B3, force_generic=6).float(); r3 = A3.float() @ B3.float().t()
    o3b = ops.gemm(A3, B3, force_generic=6).float()
    err = (o3 - r3).abs()
    bad = (err > 0.05 * r3.abs().max())
    tiles = bad.view(M // 128, 128, N // 256, 256).any(dim=3).any(dim=1)
    bt = tiles.nonzero()
    print(M, N, K, "tiles", M // 128 * (N // 256), "max err %.3g of %.3g; bad elems %d; bad tiles %d; first bad tiles %s; run-to-run equal %s" % (
        float(err.max()), float(r3.abs().max()), int(bad.sum()), bt.shape[0], bt[:8].tolist(), bool(torch.equal(o3, o3b))))
    if bt.shape[0]:
        ty, tx = bt[0].tolist()
        sub = bad[ty * 128:(ty + 1) * 128, tx * 256:(tx + 1) * 256]
        print("    inside first bad tile: bad rows %s ... bad cols %s ..." % (sub.any(dim=1).nonzero().flatten()[:10].tolist(), sub.any(dim=0).nonzero().flatten()[:10].tolist()), "count", int(sub.sum()))
        # is the bad tile equal to some OTHER tile's reference?
        t = o3[ty * 128:(ty + 1) * 128, tx * 256:(tx + 1) * 256]
        for cy in range(max(0, ty - 70), min(M // 128, ty + 70)):
            for cx in range(N // 256):
                r = r3[cy * 128:(cy + 1) * 128, cx * 256:(cx + 1) * 256]
                if float((t - r).abs().max()) < 0.05 * float(r3.abs().max()):
                    print("    bad tile (%d,%d) holds the product of tile (%d,%d)" % (ty, tx, cy, cx))
