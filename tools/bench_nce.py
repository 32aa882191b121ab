"""InfoNCE kernel time vs the number of gathered columns (weak scaling: b = 512 local rows, B = 512 * world columns)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from declip_amd import ops
b, D = 512, 512
scale = torch.tensor([14.3], device="cuda")
for world in (1, 2, 4, 8):
    B = b * world
    g = torch.Generator(device="cuda").manual_seed(0)
    q = torch.nn.functional.normalize(torch.randn(b, D, device="cuda", generator=g), dim=1)
    k = torch.nn.functional.normalize(torch.randn(B, D, device="cuda", generator=g), dim=1)
    q2 = torch.nn.functional.normalize(torch.randn(b, D, device="cuda", generator=g), dim=1)
    k2 = torch.nn.functional.normalize(torch.randn(B, D, device="cuda", generator=g), dim=1)
    pairs = [(q, k), (q2, k2)]
    grow = torch.full((2, b), 1.0 / (2 * b), device="cuda")
    for _ in range(3):
        rl, lse, c1, c5, _ = ops.infonce_fwd(pairs, scale, 0)
        ops.infonce_bwd(pairs, scale, 0, lse, grow)
    e = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
    torch.cuda.synchronize()
    n = 20
    e[0].record()
    for _ in range(n):
        rl, lse, c1, c5, _ = ops.infonce_fwd(pairs, scale, 0)
    e[1].record()
    for _ in range(n):
        ops.infonce_bwd(pairs, scale, 0, lse, grow)
    e[2].record()
    torch.cuda.synchronize()
    print("world %d  B=%5d: fwd %.1f us  bwd %.1f us  (both directions of the CLIP loss)" % (world, B, e[0].elapsed_time(e[1]) / n * 1e3, e[1].elapsed_time(e[2]) / n * 1e3))
