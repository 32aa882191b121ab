"""Microbench of the HBM-bound kernels of the CLIP ViT-B/32 b = 512 step at their in-step shapes (rotating buffers, so that no
call finds its inputs in the L2 / MALL): attention fwd / bwd (image tower: L = 50 dense; text tower: packed captions, L <= 77,
causal), LayerNorm fwd / bwd (25600 x 768, 22016 x 512).  us per call, algorithmic bytes, TB/s."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from declip_amd import ops, synth
from declip_amd.engine import PackedCaptions

dev = torch.device("cuda", 0)
NSET, REP = 4, 12


def timed(fn, nbytes, label):
    for i in range(NSET):
        fn(i)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for r in range(REP):
        fn(r % NSET)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / REP * 1e3
    print("%-34s %7.1f us   %6.1f MB   %5.2f TB/s" % (label, us, nbytes / 1e6, nbytes / us / 1e6), flush=True)
    return us


def bf(*shape):
    return [(torch.randn(*shape, device=dev) * 0.5).to(torch.bfloat16) for _ in range(NSET)]


which = set((os.environ.get("BENCH_SMALL") or "attn_img,attn_txt,ln").split(","))
if "all" in which:
    which = {"attn_img", "attn_txt", "attn_txt_short", "ln"}
if "attn_img" in which:
    b, L, heads, d = 512, 50, 12, 768
    qkv, dout = bf(b * L, 3 * d), bf(b * L, d)
    outs = [ops.attn_fwd(q, b, L, heads, False) for q in qkv]
    timed(lambda i: ops.attn_fwd(qkv[i], b, L, heads, False), b * L * d * 2 * 4, "attn fwd image (512 x 50, 12 heads)")
    timed(lambda i: ops.attn_bwd(qkv[i], outs[i][0], dout[i], outs[i][1], b, L, heads, False), b * L * d * 2 * 8, "attn bwd image")
if "attn_txt" in which:
    b, L, heads, d = 512, 77, 8, 512
    ids = synth.synth_tokens(b, seed=0).to(dev)
    pk = PackedCaptions(ids, 256)
    qkv, dout = bf(pk.rows_pad, 3 * d), bf(pk.rows_pad, d)
    outs = [ops.attn_varlen_fwd(q, pk.cu, pk.rows, b, L, heads, True) for q in qkv]
    timed(lambda i: ops.attn_varlen_fwd(qkv[i], pk.cu, pk.rows, b, L, heads, True), pk.rows * d * 2 * 4, "attn fwd text (%d rows, 8 heads)" % pk.rows)
    timed(lambda i: ops.attn_varlen_bwd(qkv[i], outs[i][0], dout[i], outs[i][1], pk.cu, pk.rows, b, L, heads, True), pk.rows * d * 2 * 8, "attn bwd text")
    # round 4: the same batch in two length buckets (<= 48 tokens on the 3-key-block kernels), counts read on the device
    timed(lambda i: ops.attn_bucketed_fwd(qkv[i], pk.cu, pk.order, pk.ranges, -1, b, L, pk.L_SHORT, heads, True), pk.rows * d * 2 * 4, "attn fwd text, length buckets")
    timed(lambda i: ops.attn_bucketed_bwd(qkv[i], outs[i][0], dout[i], outs[i][1], pk.cu, pk.order, pk.ranges, -1, b, L, pk.L_SHORT, heads, True), pk.rows * d * 2 * 8,
          "attn bwd text, length buckets")
if "attn_txt_short" in which:
    # what a smaller instantiation would buy for the SHORT captions: 512 captions of 9 .. 48 tokens through the 77-token kernels
    # (five 16-key blocks, 320 threads, 68 KB of LDS) and through the 48-token ones (three blocks, 192 threads, 38 KB)
    b, heads, d = 512, 8, 512
    g = torch.Generator().manual_seed(5)
    lens = torch.randint(9, 49, (b,), generator=g)
    cu = torch.zeros(b + 1, dtype=torch.int32)
    cu[1:] = lens.cumsum(0).to(torch.int32)
    rows = int(lens.sum())
    rows_pad = (rows + 255) // 256 * 256
    cu = cu.to(dev)
    qkv, dout = bf(rows_pad, 3 * d), bf(rows_pad, d)
    for Lmax in (77, 48):
        outs = [ops.attn_varlen_fwd(q, cu, rows, b, Lmax, heads, True) for q in qkv]
        timed(lambda i: ops.attn_varlen_fwd(qkv[i], cu, rows, b, Lmax, heads, True), rows * d * 2 * 4, "attn fwd short captions, Lmax %d" % Lmax)
        timed(lambda i: ops.attn_varlen_bwd(qkv[i], outs[i][0], dout[i], outs[i][1], cu, rows, b, Lmax, heads, True), rows * d * 2 * 8, "attn bwd short captions, Lmax %d" % Lmax)
if "ln" in which:
    for rows, d in ((25600, 768), (22016, 512)):
        x, dy, dres = bf(rows, d), bf(rows, d), bf(rows, d)
        w, bb = torch.randn(d, device=dev), torch.randn(d, device=dev)
        st = [ops.layernorm_fwd(xx, w, bb) for xx in x]
        dw, db = torch.zeros(d, device=dev), torch.zeros(d, device=dev)
        part = torch.empty(ops.layernorm_bwd_ws_elems(rows, d), device=dev)
        timed(lambda i: ops.layernorm_fwd(x[i], w, bb), rows * d * 2 * 2, "LN fwd %d x %d" % (rows, d))
        timed(lambda i: ops.layernorm_bwd_part(dy[i], x[i], w, st[i][1], st[i][2], dw, db, part, dres=dres[i]), rows * d * 2 * 4, "LN bwd(+res) %d x %d" % (rows, d))
