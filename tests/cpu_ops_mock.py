"""TEST INFRASTRUCTURE: torch-CPU stand-ins for declip_amd.ops with identical call semantics.

Injected by tests/test_engine_cpu_mock.py ONLY, to exercise the host-side engine (flat parameter
store, block forward/backward composition, autograd plumbing, optimizer tables) without a GPU.
The product never imports this file and has no switch to reach it."""
import torch
import torch.nn.functional as F

from declip_amd.lib import EPI_DGELU, EPI_GELU, EPI_NONE


def block_native_available():
    """the C-level block calls (dh_block_fwd / dh_block_bwd) take device pointers: with the stand-ins the engine composes the
    block from the per-op functions below"""
    return False


def _gelu(x):
    return x * torch.sigmoid(1.702 * x)


def _dgelu(x):
    s = torch.sigmoid(1.702 * x)
    return s * (1 + 1.702 * x * (1 - s))


def gemm(A, B, *, a_kmajor=False, b_kmajor=False, bias=None, epilogue=EPI_NONE, residual=None, aux=None, out=None,
         out_dtype=None, accumulate=False, split_k=1, alpha=1.0, force_generic=False, a_colsum=None, pad_ok=False, dims=None, ws=None):
    a = A.float().t() if a_kmajor else A.float()
    b = B.float() if b_kmajor else B.float().t()
    if dims is not None:           # logical sizes inside padded buffers: crop / zero-extend like the kernels see them
        M, N, K = dims
        def fit(t, r, c):
            o = torch.zeros(r, c)
            rr, cc = min(r, t.shape[0]), min(c, t.shape[1])
            o[:rr, :cc] = t[:rr, :cc]
            return o
        a, b = fit(a, M, K), fit(b, K, N)
        if bias is not None:
            bb = torch.zeros(N)
            bb[:min(N, bias.numel())] = bias[:min(N, bias.numel())]
            bias = bb
    if a_colsum is not None:
        if accumulate == 2:                   # first touch: the slots are written (include/declip_hip.h)
            a_colsum.copy_(a.sum(1))
        else:
            a_colsum += a.sum(1)
    v = alpha * (a @ b)
    if accumulate == 2:
        out.copy_(v[:out.shape[0], :out.shape[1]])
        return out
    if accumulate:
        out += v[:out.shape[0], :out.shape[1]]
        return out
    if bias is not None:
        v = v + bias
    if epilogue == EPI_GELU:                  # aux = QuickGELU'(pre) (include/declip_hip.h)
        if aux is not None:
            aux.copy_(_dgelu(v))
        v = _gelu(v)
    elif epilogue == EPI_DGELU:
        v = v * aux.float()
    if residual is not None:
        v = v + residual.float()
    if out is None:
        return v.to(out_dtype or A.dtype)
    out.copy_(v)
    return out


def colsum(X, out, accumulate=True):
    s = X.float().sum(0)
    if accumulate:
        out += s
    else:
        out.copy_(s)
    return out


def layernorm_fwd(x, w, b, eps=1e-5, save_stats=True):
    xf = x.float()
    mean = xf.mean(-1)
    var = ((xf - mean[:, None]) ** 2).mean(-1)
    rstd = torch.rsqrt(var + eps)
    y = (xf - mean[:, None]) * rstd[:, None] * w + b
    return y.to(x.dtype), mean, rstd


def layernorm_bwd(dy, x, w, mean, rstd, dw, db, dres=None):
    xh = (x.float() - mean[:, None]) * rstd[:, None]
    g = dy.float() * w
    dx = rstd[:, None] * (g - g.mean(-1, keepdim=True) - xh * (g * xh).mean(-1, keepdim=True))
    if dres is not None:
        dx = dx + dres.float()
    dw += (dy.float() * xh).sum(0)
    db += dy.float().sum(0)
    return dx.to(x.dtype)


def layernorm_bwd_ws_elems(rows, d):
    return 2 * d


def layernorm_bwd_part(dy, x, w, mean, rstd, dw, db, part, dres=None):
    """the deferred form (engine.LnGradBatch): the gradient of weight / bias is left in `part` as ONE partial row, added by
    ln_reduce_many -- so the mock exercises the host-side batching (arena slices, flush points) like the kernels do."""
    d = x.shape[1]
    zw, zb = torch.zeros(d), torch.zeros(d)
    dx = layernorm_bwd(dy, x, w, mean, rstd, zw, zb, dres=dres)
    part[:d].copy_(zw)
    part[d:2 * d].copy_(zb)
    return dx, 1


def ln_reduce_many(items):
    for part, nb, d, dw, db in items:
        if nb > 0:
            dw += part[:nb * 2 * d].view(nb, 2 * d)[:, :d].sum(0)
            db += part[:nb * 2 * d].view(nb, 2 * d)[:, d:].sum(0)


def _attn(qkv, b, L, heads, causal):
    d = qkv.shape[-1] // 3
    hd = d // heads
    q, k, v = qkv.float().view(b, L, 3 * d).split(d, -1)
    q = q.reshape(b, L, heads, hd).transpose(1, 2) * hd ** -0.5
    k = k.reshape(b, L, heads, hd).transpose(1, 2)
    v = v.reshape(b, L, heads, hd).transpose(1, 2)
    s = q @ k.transpose(-1, -2)
    if causal:
        s = s + torch.full((L, L), float("-inf")).triu_(1)
    p = torch.softmax(s, -1)
    return (p @ v).transpose(1, 2).reshape(b * L, d), torch.logsumexp(s, -1)


def attn_fwd(qkv, b, Lq, heads, causal):
    out, lse = _attn(qkv, b, Lq, heads, causal)
    return out.to(qkv.dtype), lse


def attn_bwd(qkv, out, dout, lse, b, Lq, heads, causal):
    q = qkv.detach().float().requires_grad_(True)
    with torch.enable_grad():
        o, _ = _attn(q, b, Lq, heads, causal)
        o.backward(dout.float())
    return q.grad.to(qkv.dtype)


def text_embed_fwd(ids, table, pos, dtype):
    b, L = ids.shape
    return (table[ids] + pos).reshape(b * L, -1).to(dtype)


def embed_table_grad(ids, dx, dtable):
    dtable.index_add_(0, ids.reshape(-1), dx.reshape(ids.numel(), -1).float())


def text_embed_bwd(ids, dx, dtable, dpos, hot_ids=(0,)):
    b, L = ids.shape
    if dtable is not None:
        dtable.index_add_(0, ids.reshape(-1), dx.float())
    if dpos is not None:
        dpos += dx.float().view(b, L, -1).sum(0)


def im2row(images, c0, patch, dtype, out=None):
    x = images[:, c0:c0 + 3]
    b, c, H, W = x.shape
    gh, gw = H // patch, W // patch
    x = x.reshape(b, c, gh, patch, gw, patch).permute(0, 2, 4, 1, 3, 5)
    r = x.reshape(b * gh * gw, c * patch * patch).to(dtype)
    if out is not None:
        out.copy_(r)
        return out
    return r


def vit_assemble_fwd(patches, cls, pos, b, npatch):
    d = patches.shape[-1]
    x = torch.cat([cls.expand(b, 1, d), patches.float().view(b, npatch, d)], 1) + pos
    return x.reshape(b * (npatch + 1), d).to(patches.dtype)


def vit_assemble_bwd(dx, dcls, dpos, b, npatch):
    d = dx.shape[-1]
    g = dx.float().view(b, npatch + 1, d)
    dpos += g.sum(0)
    if dcls is not None:
        dcls += g[:, 0].sum(0)


def pool_rows_fwd(x, idx, b, Lq):
    xv = x.view(b, Lq, -1)
    i = idx if idx is not None else torch.zeros(b, dtype=torch.long)
    return xv[torch.arange(b), i].contiguous()


def pool_rows_bwd(dout, idx, b, Lq):
    dx = torch.zeros(b, Lq, dout.shape[-1], dtype=dout.dtype)
    i = idx if idx is not None else torch.zeros(b, dtype=torch.long)
    dx[torch.arange(b), i] = dout
    return dx.view(b * Lq, -1)


def l2norm_fwd(x, eps):
    n = x.float().norm(dim=-1)
    return x.float() / (n[:, None] + eps), n


def l2norm_bwd(x, norm, dy, eps):
    xf = x.float()
    inv = 1 / (norm + eps)
    s = (dy * xf).sum(-1)
    return (dy * inv[:, None] - xf * (s * inv * inv / norm.clamp_min(1e-30))[:, None]).to(x.dtype)


def infonce_fwd(pairs, scale, label0, want_logits=False, label0s=None, excl0s=None):
    rl, lse, c1, c5, lg = [], [], [], [], []
    for p, (q, k) in enumerate(pairs):
        logits = scale * q @ k.t()
        labels = (label0s[p] if label0s is not None else label0) + torch.arange(q.shape[0])
        if excl0s is not None and excl0s[p] >= 0:
            logits = logits.clone()
            logits[torch.arange(q.shape[0]), excl0s[p] + torch.arange(q.shape[0])] = float("-inf")
        l = torch.logsumexp(logits, -1)
        ll = logits[torch.arange(q.shape[0]), labels]
        cnt = ((logits > ll[:, None]) & (torch.arange(k.shape[0])[None, :] != labels[:, None])).sum(-1)
        rl.append(l - ll), lse.append(l), c1.append((cnt < 1).float()), c5.append((cnt < 5).float()), lg.append(logits)
    return torch.stack(rl), torch.stack(lse), torch.stack(c1), torch.stack(c5), (torch.stack(lg) if want_logits else None)


def infonce_bwd(pairs, scale, label0, row_lse, g_row, need=None, label0s=None, excl0s=None):
    outs, dscale = [], torch.zeros(1)
    for p, (q, k) in enumerate(pairs):
        dots = q @ k.t()
        P = torch.exp(scale * dots - row_lse[p][:, None])
        labels = (label0s[p] if label0s is not None else label0) + torch.arange(q.shape[0])
        if excl0s is not None and excl0s[p] >= 0:
            P[torch.arange(q.shape[0]), excl0s[p] + torch.arange(q.shape[0])] = 0
        P[torch.arange(q.shape[0]), labels] -= 1
        G = P * g_row[p][:, None]
        outs.append((scale * G @ k, scale * G.t() @ q))
        dscale += (G * dots).sum()
    return outs, dscale


def ce_rows_fwd(logits, labels):
    l = torch.logsumexp(logits, -1)
    valid = labels >= 0
    safe = labels.clamp_min(0)
    ll = logits[torch.arange(logits.shape[0]), safe]
    cnt = ((logits > ll[:, None]) & (torch.arange(logits.shape[1])[None, :] != safe[:, None])).sum(-1)
    return torch.where(valid, l - ll, torch.zeros_like(l)), l, ((cnt < 1) & valid).float(), ((cnt < 5) & valid).float()


def ce_rows_bwd(logits, labels, row_lse, g_row):
    valid = labels >= 0
    P = torch.exp(logits - row_lse[:, None])
    P[torch.arange(logits.shape[0])[valid], labels[valid]] -= 1
    return P * (g_row * valid)[:, None]


def adamw(p, g, m, v, p_bf16, lr, beta1, beta2, eps, wd, step, grad_scale=1.0):
    g = g * grad_scale
    p.mul_(1 - lr * wd)
    m.mul_(beta1).add_(g, alpha=1 - beta1)
    v.mul_(beta2).addcmul_(g, g, value=1 - beta2)
    bc1, bc2 = 1 - beta1 ** step, 1 - beta2 ** step
    p.addcdiv_(m, v.sqrt() / bc2 ** 0.5 + eps, value=-lr / bc1)
    if p_bf16 is not None:
        p_bf16.copy_(p)


def cast(src, dst):
    dst.copy_(src.view(dst.shape))
    return dst


def gemm_dw_group(problems, ws=None, first_touch=False):
    for dy, x, gw, gb in problems:
        if first_touch:                       # accumulate = 2: the slots are written (include/declip_hip.h)
            gw.copy_(dy.float().t() @ x.float())
            if gb is not None:
                gb.copy_(dy.float().sum(0))
            continue
        gw.add_(dy.float().t() @ x.float())
        if gb is not None:
            gb.add_(dy.float().sum(0))


def zero_ranges(base, table_dev, n, max_len):
    for lo, hi in table_dev[:n].tolist():
        base[lo:hi] = 0


def adamw_segmented(p, g, m, v, p_bf16, seg_start, seg_lr, seg_wd, beta1, beta2, eps, step, grad_scale=1.0):
    n = p.numel()
    starts = seg_start.tolist() + [n]
    for i in range(len(starts) - 1):
        lo, hi = starts[i], starts[i + 1]
        lr, wd = float(seg_lr[i]), float(seg_wd[i])
        if lr < 0.0:
            continue
        adamw(p[lo:hi], g[lo:hi], m[lo:hi], v[lo:hi], None, lr, beta1, beta2, eps, wd, step, grad_scale)
    if p_bf16 is not None:
        p_bf16.copy_(p)


def ce_rows_bwd_padded(logits, labels, row_lse, g_row, C, out_dtype, rows_pad, C_pad):
    rows = labels.numel()
    dl = torch.zeros(rows_pad, C_pad, dtype=out_dtype)
    dl[:rows, :C] = ce_rows_bwd(logits[:rows, :C], labels, row_lse, g_row).to(out_dtype)
    return dl


def bn1d_fwd(x, w, b, running_mean, running_var, groups, relu, training, eps=1e-5, momentum=0.1):
    rows, C = x.shape
    R = rows // groups
    xf = x.float().view(groups, R, C)
    if training:
        mean = xf.mean(1)
        var = xf.var(1, unbiased=False)
        if running_mean is not None:
            for g in range(groups):
                running_mean.mul_(1 - momentum).add_(momentum * mean[g])
                running_var.mul_(1 - momentum).add_(momentum * var[g] * (R / max(R - 1, 1)))
    else:
        mean = running_mean.expand(groups, C)
        var = running_var.expand(groups, C)
    invstd = torch.rsqrt(var + eps)
    y = (xf - mean[:, None]) * invstd[:, None] * w + b
    if relu:
        y = y.clamp_min(0)
    return y.reshape(rows, C).to(x.dtype), mean.contiguous(), invstd.contiguous()


def bn1d_bwd(dy, x, y, w, mean, invstd, dw, db, groups, relu):
    rows, C = x.shape
    R = rows // groups
    d = dy.float().view(groups, R, C)
    if relu:
        d = d * (y.float().view(groups, R, C) > 0)
    xh = (x.float().view(groups, R, C) - mean[:, None]) * invstd[:, None]
    dw += (d * xh).sum((0, 1))
    db += d.sum((0, 1))
    dx = w * invstd[:, None] * (d - d.mean(1, keepdim=True) - xh * (d * xh).mean(1, keepdim=True))
    return dx.reshape(rows, C).to(x.dtype)


def cos_rows_fwd(p, z):
    return F.cosine_similarity(p.float(), z.float(), dim=-1, eps=0)


def cos_rows_bwd(p, z, g_row):
    pf = p.detach().float().requires_grad_(True)
    with torch.enable_grad():
        c = (pf * z.float()).sum(-1) / (pf.norm(dim=-1) * z.float().norm(dim=-1))
        c.backward(g_row)
    return pf.grad.to(p.dtype)


def nn_bank_query(q, bank):
    sim = q @ bank.t()
    idx = sim.argmax(dim=1)
    return idx, bank[idx].clone()


def nn_bank_enqueue(store, ptr_dev, batch, size):
    p, b = int(ptr_dev), batch.shape[0]
    store[p:p + b] = batch
    ptr_dev.fill_(0 if p + b >= size else p + b)


def gather_rows(x, idx, n_pad=None):
    n = idx.numel()
    n_pad = n_pad or max(n, 1)
    out = torch.zeros(n_pad, x.shape[-1], dtype=x.dtype)
    out[:n] = x[idx]
    return out


def scatter_rows_add(dout, idx, dx):
    dx[idx] += dout[:idx.numel()]
    return dx


def filip_select(img_tok, txt_tok):
    cross = img_tok @ txt_tok.transpose(1, 2)
    return cross.sum(2).topk(16, dim=1)[1], cross.sum(1).topk(16, dim=1)[1]


def maxsim_reduce(S, b, B, J, scale):
    v = S[:, :B * 16].float().view(b, J, B, 16)
    mx, arg = v.max(-1)
    raw = mx.mean(1)
    return raw * scale, raw, arg.reshape(b * J, B).to(torch.uint8)


def maxsim_scatter(dlogits, arg, scale, b, B, J, dtype):
    G = torch.zeros(b * J, B, 16)
    w = (dlogits * scale / J)[:, None, :].expand(b, J, B).reshape(b * J, B)
    G.scatter_(2, arg.long()[..., None], w[..., None])
    return G.reshape(b * J, B * 16).to(dtype)


def maxsim_scatter_rows(dlogits, arg, scale, b, B, J, r0, nrows, out):
    G = maxsim_scatter(dlogits, arg[:b * J], scale, b, B, J, out.dtype)
    pad = torch.zeros(max(0, r0 + nrows - G.shape[0]), G.shape[1], dtype=G.dtype)
    out[:nrows] = torch.cat([G, pad])[r0:r0 + nrows]
    return out[:nrows]


def image_prep_u8(src, out_hw, crop_xy=None, flip=None, mean=(0.485, 0.456, 0.406), std=(0.229, 0.224, 0.225), out=None, c0=0):
    from oracle import restated
    res = restated.image_prep_u8(src, out_hw, crop_xy, flip, mean, std)
    if out is None:
        return res
    out[:, c0:c0 + 3] = res
    return out


# ---------------------------------------------------------------------------------------------
# ModifiedResNet ops: no torch stand-ins -- the REAL wrappers and C entry points run on the host-emulated build of
# csrc/resnet_ops.hip (tests/hipemu_util.py), so the engine-level CPU tests execute the kernels' own code.
# ---------------------------------------------------------------------------------------------
from declip_amd import ops as _real_ops  # noqa: E402

_RESNET_SYMS = ["dh_conv_rows", "dh_bn2d_ws_bytes", "dh_bn2d_fwd", "dh_bn2d_bwd", "dh_bn2d_sums", "dh_bn2d_fwd_apply", "dh_bn2d_bwd_apply",
                "dh_avgpool_fwd", "dh_avgpool_bwd", "dh_attnpool_tokens_fwd", "dh_attnpool_tokens_bwd"]
_RESNET_ORIG = {n: getattr(_real_ops, n) for n in ("conv_rows", "conv_rows_image", "bn2d_fwd", "bn2d_bwd", "bn2d_sums", "bn2d_fwd_apply",
                                                   "bn2d_bwd_apply", "avgpool_fwd", "avgpool_bwd", "attnpool_tokens_fwd",
                                                   "attnpool_tokens_bwd")}


def _emulated(name):
    def call(*args, **kwargs):
        from hipemu_util import emu_ops
        with emu_ops(["resnet_ops.hip"], _RESNET_SYMS):
            return _RESNET_ORIG[name](*args, **kwargs)
    call.__name__ = name
    return call


conv_rows = _emulated("conv_rows")
conv_rows_image = _emulated("conv_rows_image")
bn2d_fwd = _emulated("bn2d_fwd")
bn2d_bwd = _emulated("bn2d_bwd")
bn2d_sums = _emulated("bn2d_sums")
bn2d_fwd_apply = _emulated("bn2d_fwd_apply")
bn2d_bwd_apply = _emulated("bn2d_bwd_apply")
avgpool_fwd = _emulated("avgpool_fwd")
avgpool_bwd = _emulated("avgpool_bwd")
attnpool_tokens_fwd = _emulated("attnpool_tokens_fwd")
attnpool_tokens_bwd = _emulated("attnpool_tokens_bwd")


# ---------------------------------------------------------------------------------------------
# packed captions / pooled last block (DH_TEXT_PACKED, DH_POOLED_LAST): likewise the real wrappers on the host-emulated kernels
# ---------------------------------------------------------------------------------------------
_SEQ_SYMS = ["dh_text_embed_packed_fwd", "dh_text_embed_bwd", "dh_embed_table_grad", "dh_embed_table_grad_ws_bytes", "dh_packed_pos_grad", "dh_attn_varlen_fwd", "dh_attn_varlen_bwd",
             "dh_attn_pooled_fwd", "dh_attn_pooled_bwd"]
_SEQ_ORIG = {n: getattr(_real_ops, n) for n in ("text_embed_packed_fwd", "text_embed_packed_bwd", "attn_varlen_fwd", "attn_varlen_bwd",
                                                "attn_pooled_fwd", "attn_pooled_bwd")}


def _emulated_seq(name):
    def call(*args, **kwargs):
        from hipemu_util import emu_ops
        with emu_ops(["embed.hip", "attention.hip"], _SEQ_SYMS):
            return _SEQ_ORIG[name](*args, **kwargs)
    call.__name__ = name
    return call


text_embed_packed_fwd = _emulated_seq("text_embed_packed_fwd")
text_embed_packed_bwd = _emulated_seq("text_embed_packed_bwd")
attn_varlen_fwd = _emulated_seq("attn_varlen_fwd")
attn_varlen_bwd = _emulated_seq("attn_varlen_bwd")
attn_pooled_fwd = _emulated_seq("attn_pooled_fwd")
attn_pooled_bwd = _emulated_seq("attn_pooled_bwd")
