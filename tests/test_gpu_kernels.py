"""Kernel-level parity: every C-ABI entry point against a plain fp32/fp64 torch CPU reference of
the same op (for floating-point kernels the torch reference is the checker, see oracle/__init__)."""
import math

import pytest
import torch

pytestmark = pytest.mark.gpu

cuda = torch.device("cuda")


def _ops():
    from declip_amd import ops
    return ops


def rel_err(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return float((a - b).abs().max() / (b.abs().max() + 1e-30))


def rnd(*shape, seed=0, dtype=torch.float32, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(*shape, generator=g) * scale).to(dtype)


def quick_gelu(x):
    return x * torch.sigmoid(1.702 * x)


def quick_gelu_grad(x):
    s = torch.sigmoid(1.702 * x)
    return s * (1 + 1.702 * x * (1 - s))


# ----------------------------------------------------------------------------- GEMM
@pytest.mark.parametrize("dtype,generic", [(torch.float32, False), (torch.bfloat16, False), (torch.bfloat16, True)])
@pytest.mark.parametrize("a_km,b_km", [(False, False), (False, True), (True, True), (True, False)])
@pytest.mark.parametrize("M,N,K", [(200, 136, 192), (128, 128, 64), (400, 768, 256), (37, 264, 520)])
def test_gemm_layouts(dtype, generic, a_km, b_km, M, N, K):
    ops = _ops()
    A = rnd(M, K, seed=1).to(dtype)
    B = rnd(N, K, seed=2).to(dtype)
    ref = A.double() @ B.double().t()
    Ad = (A.t().contiguous() if a_km else A).to(cuda)
    Bd = (B.t().contiguous() if b_km else B).to(cuda)
    out = ops.gemm(Ad, Bd, a_kmajor=a_km, b_kmajor=b_km, out_dtype=torch.float32, force_generic=generic)
    tol = 2e-5 if dtype == torch.float32 else 2e-3  # bf16 inputs are exact in the reference; fp32 accumulate
    assert rel_err(out, ref) < tol


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_gemm_epilogues(dtype):
    ops = _ops()
    from declip_amd.lib import EPI_DGELU, EPI_GELU
    M, N, K = 264, 384, 128
    A, B = rnd(M, K, seed=3).to(dtype), rnd(N, K, seed=4, scale=0.1).to(dtype)
    bias = rnd(N, seed=5)
    R = rnd(M, N, seed=6).to(dtype)
    pre = A.double() @ B.double().t() + bias.double()
    tol = 1e-4 if dtype == torch.float32 else 1.5e-2
    # bias + residual
    out = ops.gemm(A.to(cuda), B.to(cuda), bias=bias.to(cuda), residual=R.to(cuda))
    assert rel_err(out, pre + R.double()) < tol
    # GELU epilogue with the derivative as side output (what DH_EPI_DGELU multiplies by)
    aux = torch.empty(M, N, device=cuda, dtype=dtype)
    out = ops.gemm(A.to(cuda), B.to(cuda), bias=bias.to(cuda), epilogue=EPI_GELU, aux=aux)
    assert rel_err(aux, quick_gelu_grad(pre)) < tol
    assert rel_err(out, quick_gelu(pre)) < tol
    # DGELU epilogue: value * aux
    U = rnd(M, N, seed=7).to(dtype)
    out = ops.gemm(A.to(cuda), B.to(cuda), epilogue=EPI_DGELU, aux=U.to(cuda))
    assert rel_err(out, (A.double() @ B.double().t()) * U.double()) < tol
    # the pair is the backward of x * sigmoid(1.702 x): d/dpre of sum(w * gelu(pre)) = w * aux
    w = rnd(M, N, seed=8).double()
    pre_t = pre.clone().requires_grad_(True)
    (w * quick_gelu(pre_t)).sum().backward()
    assert rel_err(w * aux.double().cpu(), pre_t.grad) < tol


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_gemm_weight_grad_accumulate_splitk(dtype):
    ops = _ops()
    rows, out_f, in_f = 1000, 256, 136
    dY, X = rnd(rows, out_f, seed=8).to(dtype), rnd(rows, in_f, seed=9).to(dtype)
    G0 = rnd(out_f, in_f, seed=10)
    gw = G0.clone().to(cuda)
    gb = torch.ones(out_f, device=cuda)
    ops.gemm(dY.to(cuda), X.to(cuda), a_kmajor=True, b_kmajor=True, out=gw, accumulate=True, split_k=4, a_colsum=gb)
    ref = G0.double() + dY.double().t() @ X.double()
    assert rel_err(gw, ref) < (1e-5 if dtype == torch.float32 else 2e-3)
    assert rel_err(gb, 1 + dY.double().sum(0)) < 1e-4
    # MFMA/LDS-DMA path (rows % 64 == 0) with the fused bias gradient
    rows2 = 1024
    dY2, X2 = rnd(rows2, out_f, seed=18).to(dtype), rnd(rows2, in_f, seed=19).to(dtype)
    gw2, gb2 = torch.zeros(out_f, in_f, device=cuda), torch.zeros(out_f, device=cuda)
    ops.gemm(dY2.to(cuda), X2.to(cuda), a_kmajor=True, b_kmajor=True, out=gw2, accumulate=True, split_k=2, a_colsum=gb2)
    assert rel_err(gw2, dY2.double().t() @ X2.double()) < (1e-5 if dtype == torch.float32 else 2e-3)
    assert rel_err(gb2, dY2.double().sum(0)) < 1e-4


def test_colsum():
    ops = _ops()
    for dtype in (torch.float32, torch.bfloat16):
        X = rnd(1234, 200, seed=11).to(dtype)
        out = torch.ones(200, device=cuda)
        ops.colsum(X.to(cuda), out, accumulate=True)
        assert rel_err(out, 1 + X.double().sum(0)) < 1e-4


# ----------------------------------------------------------------------------- LayerNorm
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("rows,d", [(50, 768), (51, 768), (1, 768), (7, 256), (77, 512), (13, 128), (9, 100)])     # 768 / 256: the row-pair forward kernel (odd counts: a last pair of one row)
def test_layernorm(dtype, rows, d):
    ops = _ops()
    x = rnd(rows, d, seed=12).to(dtype)
    w, b = 1 + 0.1 * rnd(d, seed=13), 0.1 * rnd(d, seed=14)
    dy, dres = rnd(rows, d, seed=15).to(dtype), rnd(rows, d, seed=16).to(dtype)
    xr = x.double().requires_grad_(True)
    wr, br = w.double().requires_grad_(True), b.double().requires_grad_(True)
    yr = torch.nn.functional.layer_norm(xr, (d,), wr, br, 1e-5)
    yr.backward(dy.double())
    y, mean, rstd = ops.layernorm_fwd(x.to(cuda), w.to(cuda), b.to(cuda))
    tol = 1e-4 if dtype == torch.float32 else 1.5e-2
    assert rel_err(y, yr.detach()) < tol
    dw, db = torch.zeros(d, device=cuda), torch.zeros(d, device=cuda)
    dx = ops.layernorm_bwd(dy.to(cuda), x.to(cuda), w.to(cuda), mean, rstd, dw, db, dres=dres.to(cuda))
    assert rel_err(dx, xr.grad + dres.double()) < tol
    assert rel_err(dw, wr.grad) < (1e-4 if dtype == torch.float32 else 2e-2)
    assert rel_err(db, br.grad) < 1e-4


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_layernorm_deferred_reduce_of_many(dtype):
    """dh_layernorm_bwd_part + dh_ln_reduce_many: the weight / bias gradient partials of several LayerNorms (different widths and
    row counts, one of them through the scalar kernel that accumulates directly) reduced by ONE launch into gradients that
    already hold values == one dh_layernorm_bwd each."""
    ops = _ops()
    cases = [(200, 768), (77, 512), (9, 100), (300, 768), (64, 128)]
    items, want, got_dx, want_dx = [], [], [], []
    for i, (rows, d) in enumerate(cases):
        x = rnd(rows, d, seed=40 + i).to(dtype).to(cuda)
        w, b = (1 + 0.1 * rnd(d, seed=50 + i)).to(cuda), (0.1 * rnd(d, seed=60 + i)).to(cuda)
        dy = rnd(rows, d, seed=70 + i).to(dtype).to(cuda)
        _, mean, rstd = ops.layernorm_fwd(x, w, b)
        dw0, db0 = rnd(d, seed=80 + i).to(cuda), rnd(d, seed=90 + i).to(cuda)
        dw_a, db_a = dw0.clone(), db0.clone()
        want_dx.append(ops.layernorm_bwd(dy, x, w, mean, rstd, dw_a, db_a))
        want.append((dw_a, db_a))
        dw_b, db_b = dw0.clone(), db0.clone()
        part = torch.full((ops.layernorm_bwd_ws_elems(rows, d) + 64,), float("nan"), device=cuda)
        dx, nb = ops.layernorm_bwd_part(dy, x, w, mean, rstd, dw_b, db_b, part)
        got_dx.append(dx)
        if nb > 0:
            assert torch.equal(dw_b, dw0) and torch.equal(db_b, db0)          # nothing reduced yet
        items.append((part, nb, d, dw_b, db_b))
    assert any(it[1] == 0 for it in items) and sum(it[1] > 0 for it in items) >= 3
    ops.ln_reduce_many(items)
    for (part, nb, d, dw_b, db_b), (dw_a, db_a), dx, dxw in zip(items, want, got_dx, want_dx):
        assert torch.equal(dx, dxw)
        assert rel_err(dw_b, dw_a) < 1e-5 and rel_err(db_b, db_a) < 1e-5


# ----------------------------------------------------------------------------- attention
def attn_ref(qkv, heads, causal):
    b, L, d3 = qkv.shape
    d = d3 // 3
    hd = d // heads
    q, k, v = qkv.split(d, dim=-1)
    q = q.reshape(b, L, heads, hd).transpose(1, 2) * hd ** -0.5
    k = k.reshape(b, L, heads, hd).transpose(1, 2)
    v = v.reshape(b, L, heads, hd).transpose(1, 2)
    s = q @ k.transpose(-1, -2)
    if causal:
        s = s + torch.full((L, L), float("-inf"), dtype=s.dtype).triu_(1)
    p = torch.softmax(s, -1)
    return (p @ v).transpose(1, 2).reshape(b, L, d), torch.logsumexp(s, -1)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("b,L,heads,causal", [(3, 50, 12, False), (2, 77, 8, True), (2, 5, 2, False), (1, 16, 2, True), (2, 33, 1, True),
                                              (1, 128, 2, True), (2, 128, 1, False), (2, 91, 2, True), (3, 1, 2, True), (700, 50, 12, False)])
def test_attention(dtype, b, L, heads, causal):
    ops = _ops()
    d = heads * 64
    if dtype == torch.float32 and L > 91:
        # fp32 validation kernels keep q, k, v (, dO) AND the [L, L] matrices in LDS: forward L <= 126, backward L <= 91 --
        # beyond that the call must fail with an error, not launch
        from declip_amd.lib import DeclipHipError
        qkv = rnd(b, L, 3 * d, seed=17).to(cuda)
        with pytest.raises(DeclipHipError):
            out, lse = ops.attn_fwd(qkv.view(b * L, 3 * d), b, L, heads, causal)
            ops.attn_bwd(qkv.view(b * L, 3 * d), out, out, lse, b, L, heads, causal)
        return
    qkv = rnd(b, L, 3 * d, seed=17).to(dtype)
    dout = rnd(b, L, d, seed=18).to(dtype)
    qr = qkv.double().requires_grad_(True)
    out_r, lse_r = attn_ref(qr, heads, causal)
    out_r.backward(dout.double())
    out, lse = ops.attn_fwd(qkv.to(cuda).view(b * L, 3 * d), b, L, heads, causal)
    tol = 1e-4 if dtype == torch.float32 else 2e-2
    assert rel_err(out.view(b, L, d), out_r.detach()) < tol
    assert rel_err(lse, lse_r.detach()) < (1e-5 if dtype == torch.float32 else 5e-3)
    dqkv = ops.attn_bwd(qkv.to(cuda).view(b * L, 3 * d), out, dout.to(cuda).view(b * L, d), lse, b, L, heads, causal)
    assert rel_err(dqkv.view(b, L, 3 * d), qr.grad) < (2e-4 if dtype == torch.float32 else 3e-2)


# ----------------------------------------------------------------------------- embeddings / pooling
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_text_embed(dtype):
    ops = _ops()
    b, L, d, V = 5, 16, 128, 1000
    g = torch.Generator().manual_seed(19)
    ids = torch.randint(0, V, (b, L), generator=g)
    ids[:, 3] = ids[0, 3]                                   # repeated ids exercise the scatter-add
    table, pos = rnd(V, d, seed=20), rnd(L, d, seed=21)
    x = ops.text_embed_fwd(ids.to(cuda), table.to(cuda), pos.to(cuda), dtype)
    ref = table[ids] + pos
    assert rel_err(x.view(b, L, d), ref) < (1e-6 if dtype == torch.float32 else 1e-2)
    dx = rnd(b * L, d, seed=22).to(dtype)
    dt, dp = torch.zeros(V, d, device=cuda), torch.zeros(L, d, device=cuda)
    ops.text_embed_bwd(ids.to(cuda), dx.to(cuda), dt, dp, hot_ids=(int(ids[0, 3]), 0))
    rt = torch.zeros(V, d, dtype=torch.float64).index_add_(0, ids.reshape(-1), dx.double())
    assert rel_err(dt, rt) < 1e-5
    assert rel_err(dp, dx.double().view(b, L, d).sum(0)) < 1e-5


@pytest.mark.parametrize("rows,d,V,dtype", [(5000, 512, 3000, torch.bfloat16), (700, 520, 90, torch.float32), (33, 768, 49408, torch.bfloat16),
                                            (22016, 512, 49408, torch.bfloat16), (900, 64, 70001, torch.float32)])
def test_embed_table_grad_sorted_segments(rows, d, V, dtype):
    """dh_embed_table_grad: counting sort by id + one wave per 16 sorted rows.  Zipf-like ids (runs far longer than a wave's chunk next
    to ids that occur once), a table that already holds gradient, ids outside the vocabulary (skipped), d not a multiple of 512
    (lanes masked off in the column loop), twice in a row through the cached workspace."""
    ops = _ops()
    g = torch.Generator().manual_seed(rows + d)
    u = torch.rand(rows, generator=g)
    ids = (V * u ** 3).long().clamp_(0, V - 1)             # cubic: a few very frequent ids, a long tail of singletons
    ids[::97] = 1                                            # one run of ~rows/97 rows
    ids[5] = -1
    ids[7] = V + 3
    dx = rnd(rows, d, seed=3).to(dtype)
    base = rnd(V, d, seed=4)
    keep = (ids >= 0) & (ids < V)
    ref = base.double().index_add_(0, ids[keep], dx.double()[keep])
    for _ in range(2):
        dt = base.clone().to(cuda)
        ops.embed_table_grad(ids.to(cuda), dx.to(cuda), dt)
        assert rel_err(dt, ref) < 2e-6                       # fp32 sums of at most a few hundred terms


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_vision_embed(dtype):
    ops = _ops()
    from oracle import restated
    b, P, res, d = 3, 32, 96, 128
    images = rnd(b, 6, res, res, seed=23)
    rows = ops.im2row(images.to(cuda), 3, P, dtype)                       # second channel-stacked view
    ref = restated.patchify(images[:, 3:6], P).reshape(-1, 3 * P * P)
    assert rel_err(rows, ref) < (1e-7 if dtype == torch.float32 else 1e-2)
    npatch = (res // P) ** 2
    patches, cls, pos = rnd(b * npatch, d, seed=24).to(dtype), rnd(d, seed=25), rnd(npatch + 1, d, seed=26)
    x = ops.vit_assemble_fwd(patches.to(cuda), cls.to(cuda), pos.to(cuda), b, npatch)
    refx = torch.cat([cls.expand(b, 1, d), patches.float().view(b, npatch, d)], 1) + pos
    assert rel_err(x.view(b, npatch + 1, d), refx) < (1e-6 if dtype == torch.float32 else 1e-2)
    dx = rnd(b * (npatch + 1), d, seed=27).to(dtype)
    dcls, dpos = torch.zeros(d, device=cuda), torch.zeros(npatch + 1, d, device=cuda)
    ops.vit_assemble_bwd(dx.to(cuda), dcls, dpos, b, npatch)
    dxr = dx.double().view(b, npatch + 1, d)
    assert rel_err(dpos, dxr.sum(0)) < 1e-5 and rel_err(dcls, dxr[:, 0].sum(0)) < 1e-5


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_pool_and_l2norm(dtype):
    ops = _ops()
    b, L, d = 6, 16, 128
    x = rnd(b * L, d, seed=28).to(dtype)
    idx = torch.tensor([0, 3, 15, 7, 7, 1])
    out = ops.pool_rows_fwd(x.to(cuda), idx.to(cuda), b, L)
    assert torch.equal(out.cpu(), x.view(b, L, d)[torch.arange(b), idx])
    out0 = ops.pool_rows_fwd(x.to(cuda), None, b, L)
    assert torch.equal(out0.cpu(), x.view(b, L, d)[:, 0])
    dx = ops.pool_rows_bwd(out, idx.to(cuda), b, L).view(b, L, d).cpu()
    ref = torch.zeros(b, L, d, dtype=dtype)
    ref[torch.arange(b), idx] = out.cpu()
    assert torch.equal(dx, ref)
    for eps in (0.0, 1e-10):
        f = rnd(b, d, seed=29).to(dtype)
        fr = f.double().requires_grad_(True)
        yr = fr / (fr.norm(dim=-1, keepdim=True) + eps)
        dy = rnd(b, d, seed=30)
        yr.backward(dy.double())
        y, norm = ops.l2norm_fwd(f.to(cuda), eps)
        assert rel_err(y, yr.detach()) < 1e-5
        dxn = ops.l2norm_bwd(f.to(cuda), norm, dy.to(cuda), eps)
        assert rel_err(dxn, fr.grad) < (1e-4 if dtype == torch.float32 else 1e-2)


# ----------------------------------------------------------------------------- losses
def _poison_lds(ops):
    """Leave NaN bit patterns in the LDS of every CU (LDS is not cleared between kernels): a kernel that reads a padding word
    it never wrote then produces NaN instead of passing by luck.  (The matrix-pipe InfoNCE kernels once did, for D % 128 != 0.)"""
    b, L, heads = 2048, 64, 2
    qkv = torch.full((b * L, 3 * heads * 64), float("nan"), device=cuda, dtype=torch.bfloat16)
    ops.attn_fwd(qkv, b, L, heads, False)


@pytest.mark.parametrize("b,B,D,label0", [(8, 8, 64, 0), (40, 120, 512, 40), (70, 70, 256, 0), (33, 99, 768, 66),
                                          (96, 1160, 128, 1000), (520, 520, 64, 0), (130, 260, 1024, 130), (40, 72, 3072, 8)])
def test_infonce(b, B, D, label0):
    """D <= 512: X rows resident in LDS; D = 768 / 1024 / 3072 (FILIP's embed_dim, the DeCLIP-88M configs): X staged per pass, dX in
    512-column chunks over blockIdx.x -- all on the fp32 matrix pipe."""
    ops = _ops()
    P = 2
    scale = torch.tensor([14.3])
    pairs, refs = [], []
    for p in range(P):
        Q = torch.nn.functional.normalize(rnd(b, D, seed=31 + p), dim=-1)
        K = torch.nn.functional.normalize(rnd(B, D, seed=41 + p), dim=-1)
        K[label0:label0 + b] += 0.5 * Q
        K = torch.nn.functional.normalize(K, dim=-1)
        pairs.append((Q, K))
    labels = label0 + torch.arange(b)
    gq = [(q.double().requires_grad_(True), k.double().requires_grad_(True)) for q, k in pairs]
    sr = scale.double().requires_grad_(True)
    g_row = rnd(P, b, seed=50).abs()
    losses, c1r, c5r, logits_r = [], [], [], []
    for (q, k) in gq:
        lg = sr * q @ k.t()
        logits_r.append(lg.detach())
        losses.append(torch.nn.functional.cross_entropy(lg, labels, reduction="none"))
        top = lg.topk(min(5, B), 1)[1]
        c1r.append((top[:, 0] == labels).double())
        c5r.append((top == labels[:, None]).any(1).double())
    total = sum((l * g_row[i].double()).sum() for i, l in enumerate(losses))
    total.backward()
    dpairs = [(q.to(cuda), k.to(cuda)) for q, k in pairs]
    _poison_lds(ops)
    row_loss, row_lse, c1, c5, logits = ops.infonce_fwd(dpairs, scale.to(cuda), label0, want_logits=True)
    ref_loss = torch.stack(losses).detach()
    # fp32 FMA chains over D, fast exp: relative to the largest row loss, with an absolute floor for batches whose losses are all
    # ~0.05 (few candidates: the fp32 rounding of logits ~7 is 6e-6 absolute whatever the loss)
    assert rel_err(row_loss, ref_loss) < (5e-5 if D <= 1024 else 1e-4) or float((row_loss.cpu().double() - ref_loss).abs().max()) < 1e-5
    assert rel_err(logits, torch.stack(logits_r)) < 1e-5
    assert torch.equal(c1.cpu().double(), torch.stack(c1r)) and torch.equal(c5.cpu().double(), torch.stack(c5r))
    _poison_lds(ops)
    outs, dscale = ops.infonce_bwd(dpairs, scale.to(cuda), label0, row_lse, g_row.to(cuda))
    for (dq, dk), (q, k) in zip(outs, gq):
        assert rel_err(dq, q.grad) < 1e-4 and rel_err(dk, k.grad) < 1e-4
    assert rel_err(dscale, sr.grad) < 1e-4


def test_ce_rows():
    ops = _ops()
    rows, C = 37, 1001
    logits = rnd(rows, C, seed=60, scale=3.0)
    labels = torch.randint(0, C, (rows,), generator=torch.Generator().manual_seed(61))
    labels[5] = -100
    lr = logits.double().requires_grad_(True)
    loss_r = torch.nn.functional.cross_entropy(lr, labels, reduction="none", ignore_index=-100)
    g = rnd(rows, seed=62).abs()
    (loss_r * g.double()).sum().backward()
    row_loss, row_lse, c1, c5 = ops.ce_rows_fwd(logits.to(cuda), labels.to(cuda))
    assert rel_err(row_loss, loss_r.detach()) < 1e-5
    dl = ops.ce_rows_bwd(logits.to(cuda), labels.to(cuda), row_lse, g.to(cuda))
    assert rel_err(dl, lr.grad) < 1e-5
    top = logits.topk(5, 1)[1]
    valid = labels >= 0
    assert torch.equal(c1.cpu().bool() & valid, (top[:, 0] == labels) & valid)


# ----------------------------------------------------------------------------- optimizer
def test_adamw_matches_torch():
    ops = _ops()
    n = 4099
    p0, g1, g2 = rnd(n, seed=70), rnd(n, seed=71), rnd(n, seed=72)
    pr = torch.nn.Parameter(p0.clone())
    opt = torch.optim.AdamW([pr], lr=1e-2, betas=(0.9, 0.98), eps=1e-8, weight_decay=0.1)
    p, m, v = p0.clone().to(cuda), torch.zeros(n, device=cuda), torch.zeros(n, device=cuda)
    pb = torch.empty(n, device=cuda, dtype=torch.bfloat16)
    for step, g in enumerate((g1, g2), 1):
        pr.grad = g.clone()
        opt.step()
        ops.adamw(p, g.to(cuda), m, v, pb, 1e-2, 0.9, 0.98, 1e-8, 0.1, step)
    assert rel_err(p, pr.detach()) < 1e-5
    assert rel_err(pb.float(), pr.detach()) < 1e-2


# ----------------------------------------------------------------------------- DeCLIP heads
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("relu", [True, False])
@pytest.mark.parametrize("G,R,C", [(2, 24, 200), (2, 24, 203), (2, 512, 1024), (3, 70, 72)])
def test_bn1d_groups(dtype, relu, G, R, C):
    """C % 8 == 0: the 16-byte kernels (8 column chunks x 32 row lanes per block); otherwise the column-per-lane ones."""
    ops = _ops()
    x = rnd(G * R, C, seed=80).to(dtype)
    w, b = 1 + 0.1 * rnd(C, seed=81), 0.1 * rnd(C, seed=82)
    dy = rnd(G * R, C, seed=83).to(dtype)
    bn = torch.nn.BatchNorm1d(C).double()
    bn.weight.data, bn.bias.data = w.double(), b.double()
    xr = x.double().requires_grad_(True)
    ys = []
    for g in range(G):                                     # the reference calls the module once per view
        y = bn(xr[g * R:(g + 1) * R])
        ys.append(torch.relu(y) if relu else y)
    yr = torch.cat(ys)
    yr.backward(dy.double())
    rm, rv = torch.zeros(C, device=cuda), torch.ones(C, device=cuda)
    y, mean, invstd = ops.bn1d_fwd(x.to(cuda), w.to(cuda), b.to(cuda), rm, rv, G, relu, True)
    tol = 1e-4 if dtype == torch.float32 else 2e-2
    assert rel_err(y, yr.detach()) < tol
    assert rel_err(rm, bn.running_mean) < 1e-4 and rel_err(rv, bn.running_var) < 1e-4
    dw, db = torch.zeros(C, device=cuda), torch.zeros(C, device=cuda)
    dx = ops.bn1d_bwd(dy.to(cuda), x.to(cuda), y, w.to(cuda), mean, invstd, dw, db, G, relu)
    assert rel_err(dx, xr.grad) < (2e-4 if dtype == torch.float32 else 3e-2)
    assert rel_err(dw, bn.weight.grad) < (1e-4 if dtype == torch.float32 else 3e-2)
    assert rel_err(db, bn.bias.grad) < (1e-4 if dtype == torch.float32 else 2e-2)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_cos_rows(dtype):
    ops = _ops()
    p, z = rnd(33, 1024, seed=84).to(dtype), rnd(33, 1024, seed=85).to(dtype)
    g = rnd(33, seed=86)
    pr = p.double().requires_grad_(True)
    cr = torch.nn.functional.cosine_similarity(pr, z.double(), dim=-1)
    (cr * g.double()).sum().backward()
    c = ops.cos_rows_fwd(p.to(cuda), z.to(cuda))
    assert rel_err(c, cr.detach()) < 1e-5
    dp = ops.cos_rows_bwd(p.to(cuda), z.to(cuda), g.to(cuda))
    assert rel_err(dp, pr.grad) < (1e-4 if dtype == torch.float32 else 1e-2)


def test_nn_bank_query_ties_and_ragged_sizes():
    """Duplicate bank rows: the lowest index wins (torch.topk on the reference's sim matrix returns the first maximum); row
    counts / bank sizes that are not multiples of the 64 x 128 tiles of the matrix-pipe kernel; a D that takes the VALU path."""
    ops = _ops()
    for rows, size, D in [(70, 4100, 512), (3, 129, 96), (33, 2049, 100)]:
        bank = torch.nn.functional.normalize(rnd(size, D, seed=90), dim=1)
        q = torch.nn.functional.normalize(rnd(rows, D, seed=91), dim=1)
        lo, hi = 5, size - 2
        bank[hi] = bank[lo]                                 # duplicates far apart (different tiles / chunks)
        q[1] = bank[lo]
        q[2] = bank[size - 1]                               # the very last (ragged) bank row
        _poison_lds(ops)
        idx, feats = ops.nn_bank_query(q.to(cuda), bank.to(cuda))
        got = idx.cpu()
        sim = q.double() @ bank.double().t()
        assert int(got[1]) == lo and int(got[2]) == size - 1, (rows, size, D, got[:3])
        chosen = sim[torch.arange(rows), got]
        assert float((sim.max(1)[0] - chosen).max()) < 2e-6
        assert torch.equal(feats.cpu(), bank[got])


@pytest.mark.parametrize("rows,size,D", [(40, 5000, 512), (7, 300, 64), (512, 65536, 512)])
def test_nn_bank_query_exact(rows, size, D):
    ops = _ops()
    bank = torch.nn.functional.normalize(rnd(size, D, seed=87), dim=1)
    q = torch.nn.functional.normalize(rnd(rows, D, seed=88), dim=1)
    q[0] = bank[size - 1]                                   # an exact hit in the last chunk
    _poison_lds(ops)
    idx, feats = ops.nn_bank_query(q.to(cuda), bank.to(cuda))
    sim = q.double() @ bank.double().t()
    ref = sim.argmax(1)
    got = idx.cpu()
    same = got == ref
    # fp32 rounding can only flip near-ties: the chosen score must equal the best score to 1e-6
    chosen = sim[torch.arange(rows), got]
    assert float((sim.max(1)[0] - chosen).max()) < 2e-6 and same.float().mean() > 0.98
    assert torch.equal(feats.cpu(), bank[got])
    assert int(got[0]) == size - 1


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_gather_scatter_rows(dtype):
    ops = _ops()
    x = rnd(50, 128, seed=89).to(dtype)
    idx = torch.tensor([3, 17, 49, 0, 8])
    out = ops.gather_rows(x.to(cuda), idx.to(cuda), 64)
    assert torch.equal(out[:5].cpu(), x[idx]) and float(out[5:].abs().max()) == 0.0
    dx = torch.zeros(50, 128, device=cuda, dtype=dtype)
    ops.scatter_rows_add(out, idx.to(cuda), dx)
    ref = torch.zeros(50, 128, dtype=dtype)
    ref[idx] = x[idx]
    assert torch.equal(dx.cpu(), ref)


def test_ce_rows_bwd_padded_layout():
    ops = _ops()
    rows, C, rows_pad, C_pad = 37, 1001, 64, 1024
    logits = rnd(rows_pad, C_pad, seed=90, scale=2.0).to(cuda)
    labels = torch.randint(0, C, (rows,), generator=torch.Generator().manual_seed(91)).to(cuda)
    view = logits[:rows, :C]
    row_loss, row_lse, _, _ = ops.ce_rows_fwd(view, labels)
    g = rnd(rows, seed=92).abs().to(cuda)
    ref = ops.ce_rows_bwd(view.contiguous(), labels, row_lse, g)
    for dt_ in (torch.float32, torch.bfloat16):
        dl = ops.ce_rows_bwd_padded(view, labels, row_lse, g, C, dt_, rows_pad, C_pad)
        assert dl.shape == (rows_pad, C_pad)
        assert float(dl[rows:].abs().max()) == 0.0 and float(dl[:, C:].abs().max()) == 0.0
        assert rel_err(dl[:rows, :C].float(), ref) < (1e-6 if dt_ == torch.float32 else 1e-2)


# ----------------------------------------------------------------------------- GEMM v3 (deep pipeline) variants
@pytest.mark.parametrize("mode", ["1", "2", "3"])                  # 256x128 / 256x256 / 128x128x3 tiles
@pytest.mark.parametrize("a_km,b_km", [(False, False), (False, True), (True, True)])
@pytest.mark.parametrize("M,N,K", [(600, 520, 96), (256, 256, 32), (1000, 768, 320), (777, 264, 64)])
def test_gemm_v3_tiles(mode, a_km, b_km, M, N, K):
    ops = _ops()
    from declip_amd.lib import EPI_GELU
    fg = 30 + int(mode)                      # dh_gemm_args.force_generic 31 / 32 / 33: gemm_v3 with this tile mode (or an error)
    if a_km and M % 8:
        M = M // 8 * 8 + 8
    A = rnd(M, K, seed=1).to(torch.bfloat16)
    B = rnd(N, K, seed=2, scale=0.2).to(torch.bfloat16)
    bias = rnd(N, seed=3)
    ref = A.double() @ B.double().t()
    Ad = (A.t().contiguous() if a_km else A).to(cuda)
    Bd = (B.t().contiguous() if b_km else B).to(cuda)
    if a_km and b_km:                                               # weight-gradient form: fp32 accumulate + fused colsum
        out = torch.zeros(M, N, device=cuda)
        cs = torch.zeros(M, device=cuda)
        ops.gemm(Ad, Bd, a_kmajor=True, b_kmajor=True, out=out, accumulate=True, split_k=2 if K >= 64 else 1, a_colsum=cs, force_generic=fg)
        assert rel_err(out, ref) < 2e-3
        assert rel_err(cs, A.double().sum(1)) < 1e-3
    else:
        aux = torch.empty(M, N, device=cuda, dtype=torch.bfloat16)
        out = ops.gemm(Ad, Bd, a_kmajor=a_km, b_kmajor=b_km, bias=bias.to(cuda), epilogue=EPI_GELU, aux=aux, force_generic=fg)
        pre = ref + bias.double()
        assert rel_err(aux, quick_gelu_grad(pre)) < 1.5e-2
        assert rel_err(out, quick_gelu(pre)) < 1.5e-2
        out32 = ops.gemm(Ad, Bd, a_kmajor=a_km, b_kmajor=b_km, out_dtype=torch.float32, force_generic=fg)
        assert rel_err(out32, ref) < 2e-3


# ----------------------------------------------------------------------------- FILIP kernels
def test_filip_select_and_maxsim():
    ops = _ops()
    b, B, J, T, D = 3, 6, 25, 24, 64
    a = torch.nn.functional.normalize(rnd(b, J, D, seed=100), dim=-1)
    t = torch.nn.functional.normalize(rnd(b, T, D, seed=101), dim=-1)
    ia, it = ops.filip_select(a.to(cuda), t.to(cuda))
    cross = a.double() @ t.double().transpose(1, 2)
    ra, rt = cross.sum(2).topk(16, dim=1)[1], cross.sum(1).topk(16, dim=1)[1]
    assert torch.equal(ia.cpu().sort(1)[0], ra.sort(1)[0]) and torch.equal(it.cpu().sort(1)[0], rt.sort(1)[0])
    K = torch.nn.functional.normalize(rnd(B * 16, D, seed=102), dim=-1)
    Q = a.reshape(b * J, D)
    S = (Q @ K.t()).contiguous()
    scale = torch.tensor([7.5])
    logits, raw, arg = ops.maxsim_reduce(S.to(cuda), b, B, J, scale.to(cuda))
    v = S.view(b, J, B, 16)
    mx, am = v.max(-1)
    assert rel_err(raw, mx.mean(1)) < 1e-6 and rel_err(logits, 7.5 * mx.mean(1)) < 1e-6
    assert torch.equal(arg.cpu().view(b, J, B).long(), am)
    dl = rnd(b, B, seed=103)
    G = ops.maxsim_scatter(dl.to(cuda), arg, scale.to(cuda), b, B, J, torch.float32)
    ref = torch.zeros(b, J, B, 16)
    ref.scatter_(3, am[..., None], (dl * 7.5 / J)[:, None, :, None].expand(b, J, B, 1))
    assert rel_err(G.view(b, J, B, 16), ref) < 1e-6


@pytest.mark.parametrize("b,B,J,D", [(16, 16, 49, 256), (32, 64, 77, 256), (5, 16, 25, 128), (256, 512, 49, 256)])
def test_maxsim_fused_forward_and_chunked_backward(b, B, J, D):
    """dh_maxsim_fused_fwd (token-similarity GEMM with max_m / mean_j in the epilogue of the persistent kernel: S is never
    written) against the explicit [b, B, J, 16] computation of filip.py:96-105 on the same bf16-rounded operands; then the whole
    autograd Function (row-chunked regeneration of G) against torch autograd on that computation."""
    ops = _ops()
    from declip_amd import engine
    bf = torch.bfloat16
    Q = torch.nn.functional.normalize(rnd(b * J, D, seed=110), dim=-1)
    K = torch.nn.functional.normalize(rnd(B * 16, D, seed=111), dim=-1)
    Qb, Kb = Q.to(bf), K.to(bf)
    rows_pad = (b * J + 255) // 256 * 256
    Qp = torch.zeros(rows_pad, D, dtype=bf)
    Qp[:b * J] = Qb
    scale = torch.tensor([9.0])
    assert ops.maxsim_fused_ok(Qp.to(cuda), Kb.to(cuda), B, J)
    logits, raw, arg = ops.maxsim_fused_fwd(Qp.to(cuda), Kb.to(cuda), b, B, J, scale.to(cuda))
    S = (Qb.double() @ Kb.double().t()).view(b, J, B, 16)
    mx, am = S.max(-1)
    assert rel_err(raw, mx.mean(1)) < 2e-6 and rel_err(logits, 9.0 * mx.mean(1)) < 2e-6
    got = arg[:b * J].cpu().view(b, J, B).long()
    picked = S.gather(3, got[..., None])[..., 0]
    assert float((mx - picked).abs().max()) <= 1e-6          # the chosen m attains the maximum (ties may resolve either way)
    assert float((got != am).double().mean()) < 1e-3
    # autograd Function vs torch autograd on the dense formula (same bf16-rounded operands)
    Qg, Kg = Qb.float().to(cuda).requires_grad_(True), Kb.float().to(cuda).requires_grad_(True)
    sg = torch.tensor(9.0, device=cuda, requires_grad=True)
    w = rnd(b, B, seed=112).to(cuda)
    out = engine.MaxSimFn.apply(sg, Qg, Kg, b, B, J, bf)
    (out * w).sum().backward()
    Qr, Kr = Qb.double().requires_grad_(True), Kb.double().requires_grad_(True)
    sr = torch.tensor(9.0, dtype=torch.float64, requires_grad=True)
    ref = sr * (Qr @ Kr.t()).view(b, J, B, 16).max(-1)[0].mean(1)
    (ref * w.double().cpu()).sum().backward()
    assert rel_err(out, ref) < 2e-6
    assert rel_err(Qg.grad, Qr.grad) < 1e-2 and rel_err(Kg.grad, Kr.grad) < 1e-2     # G is rounded to bf16 before the two GEMMs
    assert abs(float(sg.grad) - float(sr.grad)) <= 1e-5 * abs(float(sr.grad))


def test_image_prep_u8_matches_oracle():
    """uint8 HWC -> normalised fp32 CHW with crop windows and mirrors (restated.image_prep_u8 = ToTensor + Normalize + crop +
    flip); and the vision tower fed with bytes equals the tower fed with the oracle's floats."""
    from declip_amd import ops, synth
    from declip_amd.testing import build_clip
    from oracle import restated
    g = torch.Generator().manual_seed(9)
    b, Hs, Ws, H, W = 5, 300, 260, 224, 224
    src = torch.randint(0, 256, (b, Hs, Ws, 3), generator=g, dtype=torch.uint8)
    crop = torch.stack([torch.randint(0, Ws - W + 1, (b,), generator=g), torch.randint(0, Hs - H + 1, (b,), generator=g)], 1).int()
    crop[0] = torch.tensor([Ws - W, Hs - H])                  # the last admissible window
    flip = torch.tensor([0, 1, 1, 0, 1], dtype=torch.uint8)
    ref = restated.image_prep_u8(src, (H, W), crop, flip)
    out = ops.image_prep_u8(src.cuda(), (H, W), crop.cuda(), flip.cuda())
    assert float((out.cpu() - ref).abs().max()) <= 2e-6
    # into channels 3..5 of a 2-view buffer, no crop / flip tables
    buf = torch.zeros(b, 6, H, W, device="cuda")
    ops.image_prep_u8(src[:, :H, :W].contiguous().cuda(), (H, W), out=buf, c0=3)
    ref2 = restated.image_prep_u8(src[:, :H, :W], (H, W))
    assert float((buf[:, 3:].cpu() - ref2).abs().max()) <= 2e-6 and float(buf[:, :3].abs().max()) == 0.0
    from declip_amd.lib import DeclipHipError
    with pytest.raises(DeclipHipError):
        ops.image_prep_u8(src.cuda(), (H, W))                 # larger source without a crop table
    # tower on bytes == tower on floats
    cfg = synth.TINY
    model = build_clip(cfg, dtype="fp32", seed=1).eval()
    r = cfg["res"]
    small = torch.randint(0, 256, (4, r, r, 3), generator=g, dtype=torch.uint8)
    with torch.no_grad():
        f_bytes = model.encode_image(small.cuda())
        f_float = model.encode_image(restated.image_prep_u8(small, (r, r)).cuda())
    assert float((f_bytes - f_float).abs().max()) <= 1e-5 * float(f_float.abs().max())


def test_infonce_weak_scaling_shape_with_self_pair_exclusion():
    """b = 512 local rows against B = 4096 gathered columns (8 ranks, this rank = 5) on the matrix-pipe kernels, two pairs with
    per-pair label offsets; the second pair removes a self-pair column per row (NT-Xent); one gradient is not requested."""
    ops = _ops()
    b, B, D, rank = 512, 4096, 512, 5
    q1, k1 = torch.nn.functional.normalize(rnd(b, D, seed=70), dim=1), torch.nn.functional.normalize(rnd(B, D, seed=71), dim=1)
    q2, k2 = torch.nn.functional.normalize(rnd(b, D, seed=72), dim=1), torch.nn.functional.normalize(rnd(B, D, seed=73), dim=1)
    scale = torch.tensor([9.5])
    label0s, excl0s = [rank * b, 3 * b], [-1, rank * b]
    g_row = rnd(2, b, seed=74).abs() / b
    refs = []
    tensors = [t.double().requires_grad_(True) for t in (q1, k1, q2, k2)]
    sr = scale.double().requires_grad_(True)
    total = 0
    for p, (q, k) in enumerate(((tensors[0], tensors[1]), (tensors[2], tensors[3]))):
        lg = sr * q @ k.t()
        if excl0s[p] >= 0:
            mask = torch.zeros_like(lg, dtype=torch.bool)
            mask[torch.arange(b), excl0s[p] + torch.arange(b)] = True
            lg = lg.masked_fill(mask, float("-inf"))
        labels = label0s[p] + torch.arange(b)
        loss = torch.nn.functional.cross_entropy(lg, labels, reduction="none")
        refs.append(loss.detach())
        total = total + (loss * g_row[p].double()).sum()
    total.backward()
    dpairs = [(q1.to(cuda), k1.to(cuda)), (q2.to(cuda), k2.to(cuda))]
    row_loss, row_lse, c1, c5, _ = ops.infonce_fwd(dpairs, scale.to(cuda), 0, label0s=label0s, excl0s=excl0s)
    assert rel_err(row_loss, torch.stack(refs)) < 5e-5
    outs, dscale = ops.infonce_bwd(dpairs, scale.to(cuda), 0, row_lse, g_row.to(cuda), need=[(True, True), (True, False)],
                                   label0s=label0s, excl0s=excl0s)
    assert outs[1][1] is None
    assert rel_err(outs[0][0], tensors[0].grad) < 1e-4 and rel_err(outs[0][1], tensors[1].grad) < 1e-4
    assert rel_err(outs[1][0], tensors[2].grad) < 1e-4
    assert rel_err(dscale, sr.grad) < 1e-4


@pytest.mark.parametrize("n,V,K", [(300, 1000, 128), (1, 49409, 512), (5000, 49409, 512), (777, 4097, 256)])
def test_ce_fused_forward_and_backward(n, V, K):
    """dh_ce_fused_fwd / bwd (Linear + cross-entropy over the vocabulary in the epilogue of the persistent GEMM: no logits in HBM)
    against F.cross_entropy on the same bf16-rounded operands: row losses / lse, dl = g (softmax - onehot) incl. the ragged last
    column tile, the zero padding rows and the columns between V and the padded row length."""
    ops = _ops()
    bf = torch.bfloat16
    n_pad = (n + 255) // 256 * 256
    ldd = (V + 63) // 64 * 64
    X = torch.zeros(n_pad, K)
    X[:n] = rnd(n, K, seed=120)
    W, bias = rnd(V, K, seed=121, scale=0.2), rnd(V, seed=122)
    labels = torch.randint(0, V, (n,), generator=torch.Generator().manual_seed(123))
    labels[0] = V - 1                                               # a label in the ragged last tile
    Xb, Wb = X.to(bf), W.to(bf)
    assert ops.ce_fused_ok(Xb.to(cuda), Wb.to(cuda))
    row_loss, row_lse = ops.ce_fused_fwd(Xb.to(cuda), Wb.to(cuda), bias.to(cuda), labels.to(cuda), n)
    logits = Xb[:n].double() @ Wb.double().t() + bias.double()
    ref_lse = torch.logsumexp(logits, dim=1)
    ref_loss = torch.nn.functional.cross_entropy(logits, labels, reduction="none")
    assert float((row_lse.cpu().double() - ref_lse).abs().max()) <= 2e-5 * float(ref_lse.abs().max())
    assert float((row_loss.cpu().double() - ref_loss).abs().max()) <= 2e-5 * float(ref_loss.abs().max()) + 1e-5
    g = rnd(n, seed=124).abs() + 0.1
    dl = ops.ce_fused_bwd(Xb.to(cuda), Wb.to(cuda), bias.to(cuda), labels.to(cuda), row_lse, g.to(cuda), n, ldd)
    assert dl.shape == (n_pad, ldd) and dl.dtype == bf
    ref = torch.softmax(logits, dim=1)
    ref[torch.arange(n), labels] -= 1.0
    ref = ref * g.double()[:, None]
    got = dl.float().cpu().double()
    assert float((got[:n, :V] - ref).abs().max()) <= 1e-2 * float(ref.abs().max())          # bf16 storage of dl
    assert float(got[n:].abs().max() if n_pad > n else 0.0) == 0.0 and float(got[:, V:].abs().max() if ldd > V else 0.0) == 0.0
    # row sums of dl vanish (softmax - onehot), up to bf16 rounding of ~V entries
    assert float(got[:n].sum(1).abs().max()) <= 0.05 * float(g.max())


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32])
@pytest.mark.parametrize("b,L,d", [(512, 77, 512), (37, 16, 128), (1100, 12, 520)])
def test_packed_positional_gradient(dtype, b, L, d):
    """dh_packed_pos_grad: dpos[p] += sum over the captions longer than p of dx[cu[i] + p] (text_transformer.py:196: x + positional_embedding,
    on the packed rows).  bf16 takes the 16-byte-load kernel of round 6 (b > 1024: the caption offsets go through LDS in two chunks;
    d = 520: a second, partly empty column block), fp32 the scalar one."""
    ops = _ops()
    g = torch.Generator().manual_seed(b + d)
    lens = torch.randint(1, L + 1, (b,), generator=g)
    cu = torch.zeros(b + 1, dtype=torch.int32)
    cu[1:] = lens.cumsum(0).to(torch.int32)
    rows = int(cu[-1])
    dx = rnd(rows, d, seed=7).to(dtype)
    ids_p = torch.zeros(rows, dtype=torch.int64)
    base = rnd(L, d, seed=8)
    dpos = base.clone().to(cuda)
    ops.text_embed_packed_bwd(ids_p.to(cuda), cu.to(cuda), dx.to(cuda), None, dpos, rows, L)
    ref = base.double()
    pos = torch.cat([torch.arange(int(n)) for n in lens])
    ref.index_add_(0, pos, dx.double())
    assert rel_err(dpos, ref) < 1e-5
