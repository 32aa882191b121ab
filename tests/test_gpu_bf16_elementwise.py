"""Per-element bf16 gates for the non-GEMM kernels at the shapes they have in the benchmarked CLIP ViT-B/32 step.

The GEMM got `assert_bf16_close` in round 5 (tests/oracle_util.py: EVERY element within rel * magnitude + 2^-8 rms of an fp64
reference); attention, LayerNorm, the embeddings and the pooled attention were still held to `max|err| <= 1.5-3 % of the LARGEST
element` at toy shapes, which a wrong value in a small-magnitude region passes.  Here every output element of

  * LayerNorm forward / backward (+ residual-gradient add)                      25600 x 768 and 22016 x 512
  * attention forward / backward, dense                                         512 x 50 tokens x 12 heads
  * attention forward / backward, packed captions in the two length buckets     512 captions, ~22 k rows, 8 heads
  * pooled-query attention forward / backward                                   512 sequences, both towers' geometry
  * text / vision embedding assembly                                            512 x 77 x 512, 512 x 50 x 768

is compared with an fp64 evaluation of the same bf16 inputs.  `mag` is the sum of the magnitudes of the terms the kernel rounds or
adds on the way to the element (the bf16 P / dS operands of the attention matmuls, the three terms of the LayerNorm backward): an
element that is small because large terms cancel is allowed the rounding of those terms, nothing more.  Margins are recorded
(tests/golden/bf16_margins.json).

Reference: nn.MultiheadAttention / nn.LayerNorm of /root/reference/prototype/model/text_encoder/base_transformer.py:10-18,45-48.
"""
import os
import sys

import pytest
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from oracle_util import assert_bf16_close  # noqa: E402

pytestmark = pytest.mark.gpu
cuda = "cuda"
bf = torch.bfloat16


def _ops():
    from declip_amd import ops
    return ops


def rnd(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(*shape, generator=g) * scale


# ----------------------------------------------------------------------------- LayerNorm
@pytest.mark.parametrize("rows,d", [(25600, 768), (22016, 512)])
def test_layernorm_every_element_at_step_shapes(rows, d):
    ops = _ops()
    # a residual stream with a per-row offset and a few outlier channels, like the towers' (mean != 0, heavy channels)
    x = (rnd(rows, d, seed=1) + 0.5 * rnd(rows, 1, seed=2))
    x[:, ::97] *= 6.0
    x = x.to(bf)
    w, b = 1 + 0.1 * rnd(d, seed=3), 0.1 * rnd(d, seed=4)
    dy, dres = rnd(rows, d, seed=5).to(bf), rnd(rows, d, seed=6).to(bf)
    y, mean, rstd = ops.layernorm_fwd(x.to(cuda), w.to(cuda), b.to(cuda))
    dw, db = torch.zeros(d, device=cuda), torch.zeros(d, device=cuda)
    dx = ops.layernorm_bwd(dy.to(cuda), x.to(cuda), w.to(cuda), mean, rstd, dw, db, dres=dres.to(cuda))
    torch.cuda.synchronize()
    xd, wd, bd, dyd = x.double(), w.double(), b.double(), dy.double()
    mu = xd.mean(1, keepdim=True)
    rs = (xd.var(1, unbiased=False, keepdim=True) + 1e-5).rsqrt()
    xh = (xd - mu) * rs
    yr = xh * wd + bd
    assert_bf16_close("ln_fwd_%dx%d" % (rows, d), y, yr, mag=(xh * wd).abs() + bd.abs())
    g = dyd * wd
    c1, c2 = g.mean(1, keepdim=True), (g * xh).mean(1, keepdim=True)
    dxr = rs * (g - c1 - xh * c2) + dres.double()
    assert_bf16_close("ln_bwd_dx_%dx%d" % (rows, d), dx, dxr, mag=rs * (g.abs() + c1.abs() + (xh * c2).abs()) + dres.double().abs())
    # weight / bias gradients: fp32 sums over the rows (per-block partials + atomics): relative to the column's absolute sum
    dwr, dbr = (dyd * xh).sum(0), dyd.sum(0)
    aw, ab = (dyd * xh).abs().sum(0), dyd.abs().sum(0)
    assert float(((dw.double().cpu() - dwr).abs() / aw).max()) < 2e-6
    assert float(((db.double().cpu() - dbr).abs() / ab).max()) < 2e-6
    assert float((mean.double().cpu() - mu[:, 0]).abs().max()) < 1e-5 * float(xd.abs().max())
    assert float(((rstd.double().cpu() - rs[:, 0]).abs() / rs[:, 0]).max()) < 1e-5


# ----------------------------------------------------------------------------- attention (fp64 reference with the magnitudes)
def _attn_ref(q, k, v, do, causal, lens=None):
    """q, k, v, do: [n, heads, L, 64] fp64 (bf16 values); lens [n] or None.  Returns out, lse, dq, dk, dv and the per-element magnitude
    sums of their terms (what the kernel rounds: P and dS go through bf16 before the second matmuls)."""
    n, H, L, hd = q.shape
    scale = hd ** -0.5
    s = (q @ k.transpose(-1, -2)) * scale
    mask = torch.zeros(n, 1, L, L, dtype=torch.bool)
    if causal:
        mask |= torch.ones(L, L, dtype=torch.bool).triu_(1)
    if lens is not None:
        ar = torch.arange(L)
        mask = mask | (ar[None, :] >= lens[:, None])[:, None, None, :]
    s = s.masked_fill(mask, float("-inf"))
    lse = torch.logsumexp(s, -1)
    p = torch.exp(s - lse[..., None])
    out = p @ v
    m_out = p @ v.abs()
    dv = p.transpose(-1, -2) @ do
    m_dv = p.transpose(-1, -2) @ do.abs()
    dp = do @ v.transpose(-1, -2)
    D = (do * out).sum(-1, keepdim=True)
    ds = p * (dp - D) * scale
    # the magnitude that is rounded on the way to dS: dP comes from an fp32 accumulation of bf16 products, D from the bf16-ROUNDED
    # forward output -- where the two cancel (the first token of a causal row: P = 1, dP = D exactly) what is left is their rounding
    # (D itself is a sum that cancels: what the rounding of O leaves in it scales with sum |dO| |O|, not with |D|)
    ds_mag = p * (dp.abs() + (do.abs() * out.abs()).sum(-1, keepdim=True)) * scale
    dq = ds @ k
    m_dq = ds_mag @ k.abs()
    dk = ds.transpose(-1, -2) @ q
    m_dk = ds_mag.transpose(-1, -2) @ q.abs()
    return dict(out=out, lse=lse, dq=dq, dk=dk, dv=dv, m_out=m_out, m_dq=m_dq, m_dk=m_dk, m_dv=m_dv)


def _heads(x, n, L, H):
    return x.double().reshape(n, L, H, 64).transpose(1, 2)


def _rows(x):                       # [n, H, L, 64] -> [n * L, H * 64]
    n, H, L, hd = x.shape
    return x.transpose(1, 2).reshape(n * L, H * hd)


def test_dense_attention_every_element_at_the_image_tower_shape():
    ops = _ops()
    b, L, H = 512, 50, 12
    d = H * 64
    qkv = rnd(b * L, 3 * d, seed=11, scale=0.7).to(bf)
    dout = rnd(b * L, d, seed=12).to(bf)
    out, lse = ops.attn_fwd(qkv.to(cuda), b, L, H, False)
    dqkv = ops.attn_bwd(qkv.to(cuda), out, dout.to(cuda), lse, b, L, H, False)
    torch.cuda.synchronize()
    q, k, v = (_heads(t, b, L, H) for t in qkv.view(b, L, 3 * d).split(d, dim=-1))
    r = _attn_ref(q, k, v, _heads(dout.view(b, L, d), b, L, H), False)
    # (rel 2^-6: the score / probability pipeline rounds twice -- P to bf16, then the output -- on top of the bf16 inputs' own products)
    assert_bf16_close("attn_img_fwd_out", out, _rows(r["out"]), rel=2.0 ** -6, mag=_rows(r["m_out"]))
    assert float((lse.double().cpu() - r["lse"]).abs().max()) < 2e-3
    dq, dk, dv = dqkv.double().cpu().view(b * L, 3, d).unbind(1)
    assert_bf16_close("attn_img_bwd_dq", dq, _rows(r["dq"]), rel=2.0 ** -6, mag=_rows(r["m_dq"]))
    assert_bf16_close("attn_img_bwd_dk", dk, _rows(r["dk"]), rel=2.0 ** -6, mag=_rows(r["m_dk"]))
    assert_bf16_close("attn_img_bwd_dv", dv, _rows(r["dv"]), rel=2.0 ** -6, mag=_rows(r["m_dv"]))


def test_packed_attention_every_element_at_the_text_tower_shape():
    """512 captions packed to ~22 k rows, 8 heads, causal, through the two length buckets the step uses (dh_attn_bucketed_*)."""
    ops = _ops()
    from declip_amd import synth
    from declip_amd.engine import PackedCaptions
    b, L, H = 512, 77, 8
    d = H * 64
    ids = synth.synth_tokens(b, seed=0).to(cuda)
    pk = PackedCaptions(ids, 256)
    qkv = rnd(pk.rows_pad, 3 * d, seed=21, scale=0.7).to(bf)
    dout = rnd(pk.rows_pad, d, seed=22).to(bf)
    out, lse = ops.attn_bucketed_fwd(qkv.to(cuda), pk.cu, pk.order, pk.ranges, -1, b, L, pk.L_SHORT, H, True)
    dqkv = ops.attn_bucketed_bwd(qkv.to(cuda), out, dout.to(cuda), lse, pk.cu, pk.order, pk.ranges, -1, b, L, pk.L_SHORT, H, True)
    torch.cuda.synchronize()
    # the same captions in the dense [b, L] layout (rows beyond a caption: zeros, masked)
    cu = pk.cu.cpu().long()
    lens = cu[1:] - cu[:-1]
    ar = torch.arange(L)
    valid = ar[None, :] < lens[:, None]
    src = (cu[:-1, None] + ar[None, :]).clamp_(max=pk.rows - 1)

    def dense(x):
        return torch.where(valid[..., None], x.double()[src], torch.zeros((), dtype=torch.float64))
    qd, kd, vd = (_heads(t, b, L, H) for t in dense(qkv).split(d, dim=-1))
    r = _attn_ref(qd, kd, vd, _heads(dense(dout), b, L, H), True, lens=lens)

    def packed(x):                  # [b, H, L, 64] -> the valid rows, packed
        return x.transpose(1, 2).reshape(b, L, d)[valid]
    n = pk.rows
    assert_bf16_close("attn_txt_fwd_out", out[:n], packed(r["out"]), rel=2.0 ** -6, mag=packed(r["m_out"]))
    dq, dk, dv = dqkv.double().cpu().view(pk.rows_pad, 3, d).unbind(1)
    assert_bf16_close("attn_txt_bwd_dq", dq[:n], packed(r["dq"]), rel=2.0 ** -6, mag=packed(r["m_dq"]))
    assert_bf16_close("attn_txt_bwd_dk", dk[:n], packed(r["dk"]), rel=2.0 ** -6, mag=packed(r["m_dk"]))
    assert_bf16_close("attn_txt_bwd_dv", dv[:n], packed(r["dv"]), rel=2.0 ** -6, mag=packed(r["m_dv"]))
    # the rows behind the last caption are written as zeros (they meet the weight-gradient GEMMs as contraction rows)
    assert float(out[n:].abs().max()) == 0.0 and float(dqkv[n:].abs().max()) == 0.0
    lse_r = r["lse"]                                                        # [b, H, L]
    assert float((lse.double().cpu() - lse_r).masked_fill(~valid[:, None, :], 0).abs().max()) < 2e-3


@pytest.mark.parametrize("tower", ["image", "text"])
def test_pooled_attention_every_element(tower):
    """The last block's attention for the pooled row only (CLS / <|endoftext|>): one query per sequence over its keys."""
    ops = _ops()
    b = 512
    if tower == "image":
        H, L = 12, 50
        nkeys = torch.full((b,), L, dtype=torch.int32)
    else:
        H, L = 8, 77
        nkeys = torch.randint(9, 78, (b,), generator=torch.Generator().manual_seed(7)).to(torch.int32)
    d = H * 64
    row0 = torch.zeros(b, dtype=torch.int32)
    row0[1:] = nkeys.cumsum(0)[:-1].to(torch.int32)
    rows = int(nkeys.sum())
    q = rnd(b, d, seed=31, scale=0.7).to(bf)
    kv = rnd(rows, 2 * d, seed=32, scale=0.7).to(bf)
    dout = rnd(b, d, seed=33).to(bf)
    out, lse = ops.attn_pooled_fwd(q.to(cuda), kv.to(cuda), row0.to(cuda), nkeys.to(cuda), H, L)
    dq, dkv = ops.attn_pooled_bwd(q.to(cuda), kv.to(cuda), dout.to(cuda), lse, row0.to(cuda), nkeys.to(cuda), H, L)
    torch.cuda.synchronize()
    ar = torch.arange(L)
    valid = ar[None, :] < nkeys[:, None].long()
    src = (row0[:, None].long() + ar[None, :]).clamp_(max=rows - 1)
    kvd = torch.where(valid[..., None], kv.double()[src], torch.zeros((), dtype=torch.float64))      # [b, L, 2d]
    kd, vd = (t.reshape(b, L, H, 64).transpose(1, 2) for t in kvd.split(d, dim=-1))                    # [b, H, L, 64]
    qd = q.double().view(b, H, 1, 64)
    dod = dout.double().view(b, H, 1, 64)
    s = (qd @ kd.transpose(-1, -2)) * 0.125
    s = s.masked_fill(~valid[:, None, None, :], float("-inf"))
    lse_r = torch.logsumexp(s, -1)
    p = torch.exp(s - lse_r[..., None])
    o = p @ vd
    assert_bf16_close("attn_pooled_%s_out" % tower, out, o.reshape(b, d), rel=2.0 ** -6, mag=(p @ vd.abs()).reshape(b, d))
    dp = dod @ vd.transpose(-1, -2)
    D = (dod * o).sum(-1, keepdim=True)
    ds = p * (dp - D) * 0.125
    ds_mag = p * (dp.abs() + (dod.abs() * o.abs()).sum(-1, keepdim=True)) * 0.125
    assert_bf16_close("attn_pooled_%s_dq" % tower, dq, (ds @ kd).reshape(b, d), rel=2.0 ** -6, mag=(ds_mag @ kd.abs()).reshape(b, d))
    dk = (ds.transpose(-1, -2) @ qd).transpose(1, 2).reshape(b, L, d)[valid]
    dv = (p.transpose(-1, -2) @ dod).transpose(1, 2).reshape(b, L, d)[valid]
    mk = (ds_mag.transpose(-1, -2) @ qd.abs()).transpose(1, 2).reshape(b, L, d)[valid]
    mv = (p.transpose(-1, -2) @ dod.abs()).transpose(1, 2).reshape(b, L, d)[valid]
    gk, gv = dkv.double().cpu().split(d, dim=-1)
    assert_bf16_close("attn_pooled_%s_dk" % tower, gk, dk, rel=2.0 ** -6, mag=mk)
    assert_bf16_close("attn_pooled_%s_dv" % tower, gv, dv, rel=2.0 ** -6, mag=mv)


# ----------------------------------------------------------------------------- embeddings
def test_embeddings_every_element_at_step_shapes():
    ops = _ops()
    # text: token table + positions, 512 x 77 x 512
    b, L, d, V = 512, 77, 512, 49408
    g = torch.Generator().manual_seed(41)
    ids = torch.randint(0, V, (b, L), generator=g)
    table, pos = rnd(V, d, seed=42, scale=0.02), rnd(L, d, seed=43, scale=0.01)
    x = ops.text_embed_fwd(ids.to(cuda), table.to(cuda), pos.to(cuda), bf)
    ref = table.double()[ids] + pos.double()
    assert_bf16_close("text_embed_512x77x512", x.view(b, L, d), ref, mag=table.double()[ids].abs() + pos.double().abs())
    # vision: [cls | patch rows] + positions, 512 x 50 x 768
    b, npatch, d = 512, 49, 768
    patches, cls, pos = rnd(b * npatch, d, seed=44).to(bf), rnd(d, seed=45), rnd(npatch + 1, d, seed=46, scale=0.1)
    x = ops.vit_assemble_fwd(patches.to(cuda), cls.to(cuda), pos.to(cuda), b, npatch)
    core = torch.cat([cls.double().expand(b, 1, d), patches.double().view(b, npatch, d)], 1)
    assert_bf16_close("vit_assemble_512x50x768", x.view(b, npatch + 1, d), core + pos.double(), mag=core.abs() + pos.double().abs())
