"""ModifiedResNet tower (BASELINE.json configs[0] family) on the GPU: the tower's kernels against torch CPU references
through the C-ABI, the CLIP-R50 training step against the golden generated from the unmodified reference (fp32 at the
north_star tolerance on the forward; bf16 at the documented looser bounds), BatchNorm buffers, eval mode, and the full-size
ResNet-50 at the configs[0] batch against the oracle.

Also here, for the same reason (written after the GPU budget was gone): the on-GPU RandomResizedCrop image intake, the
variable-length attention kernels and the packed-caption text tower (DH_TEXT_PACKED), the pooled-query attention and the
last-block-for-the-pooled-rows path (DH_POOLED_LAST).

STATUS: written after round 1's GPU budget was gone and first run on the MI355X by the round-1 driver suite under a non-strict
xfail marker: 54 of 56 passed there; the two that did not were diagnosed on hardware in round 2 (tools/diag_hw.py) -- the full-size
ResNet-50 gradient bound tested the fp32 oracle's own rounding noise (see that test's docstring), and the resized crop lost ~3
digits to the fast-math reciprocal in its scale / centre arithmetic (fixed in csrc/embed.hip).  The marker is gone: every test
here must pass."""
import pytest
import torch
import torch.nn.functional as F

from oracle_util import check_grad_digests, load_golden, oracle_clip_run

pytestmark = [pytest.mark.gpu]
DTYPES = [torch.float32, torch.bfloat16]


def _tol(dtype):
    return 2e-5 if dtype == torch.float32 else 2e-2


def nhwc(x):
    return x.permute(0, 2, 3, 1).reshape(-1, x.shape[1]).contiguous()


def nchw(rows, N, H, W):
    return rows.reshape(N, H, W, -1).permute(0, 3, 1, 2).contiguous()


def close(a, b, dtype, scale=None):
    a, b = a.detach().float().cpu(), b.detach().float().cpu()
    s = float(b.abs().max()) if scale is None else scale
    assert float((a - b).abs().max()) <= _tol(dtype) * max(s, 1e-6), (float((a - b).abs().max()), s)


# ------------------------------------------------------------------------------------------------ kernels
@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("N,H,W,C,stride", [(2, 9, 7, 16, 1), (1, 12, 12, 8, 2), (4, 56, 56, 64, 1), (2, 7, 7, 512, 1)])
def test_conv_rows_nhwc(dtype, N, H, W, C, stride):
    from declip_amd import ops
    torch.manual_seed(0)
    x = torch.randn(N, C, H, W).to(dtype)
    rows, Ho, Wo = ops.conv_rows(nhwc(x).cuda(), N, H, W, C, stride=stride, pad=1)
    ref = F.unfold(x.float(), 3, padding=1, stride=stride).transpose(1, 2).reshape(N * Ho * Wo, C * 9)
    assert torch.equal(rows.float().cpu(), ref)


@pytest.mark.parametrize("dtype", DTYPES)
def test_conv_rows_image_view(dtype):
    from declip_amd import ops
    torch.manual_seed(1)
    N, H, W = 3, 224, 224
    img = torch.randn(N, 6, H, W)
    for c0 in (0, 3):
        rows, Ho, Wo = ops.conv_rows_image(img.cuda(), c0, dtype, stride=2, pad=1)
        ref = F.unfold(img[:, c0:c0 + 3], 3, padding=1, stride=2).transpose(1, 2).reshape(N * Ho * Wo, 27)
        assert (Ho, Wo) == (112, 112)
        assert torch.equal(rows[:, :27].float().cpu(), ref.to(dtype).float())
        assert float(rows[:, 27:].float().abs().max()) == 0.0


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("R,C,relu,res", [(37, 8, True, False), (3000, 16, True, True), (100352, 64, True, False), (1568, 2048, True, True),
                                          (64, 2056, False, False)])
def test_bn2d_fwd_bwd(dtype, R, C, relu, res):
    from declip_amd import ops
    torch.manual_seed(2)
    x = (torch.randn(R, C) * 1.5 + 0.3).to(dtype)
    r = torch.randn(R, C).to(dtype) if res else None
    w, b = 1 + 0.1 * torch.randn(C), 0.1 * torch.randn(C)
    rm, rv = torch.randn(C) * 0.1, torch.rand(C) + 0.5
    dy = torch.randn(R, C).to(dtype)
    xr, wr, br = x.float().requires_grad_(), w.clone().requires_grad_(), b.clone().requires_grad_()
    rm_ref, rv_ref = rm.clone(), rv.clone()
    yr = F.batch_norm(xr, rm_ref, rv_ref, wr, br, True, 0.1, 1e-5)
    if res:
        yr = yr + r.float()
    if relu:
        yr = F.relu(yr)
    rm2, rv2 = rm.cuda(), rv.cuda()
    xc, wc, bc = x.cuda(), w.cuda(), b.cuda()
    y, mean, invstd = ops.bn2d_fwd(xc, wc, bc, rm2, rv2, relu, True, residual=r.cuda() if res else None)
    close(y, yr, dtype)
    close(rm2, rm_ref, torch.float32), close(rv2, rv_ref, torch.float32)
    mask = (y.float().cpu() > 0).float() if relu else torch.ones(R, C)
    g_x, g_w, g_b = torch.autograd.grad(F.batch_norm(xr, None, None, wr, br, True, 0.1, 1e-5), (xr, wr, br), dy.float() * mask)
    dw, db = torch.full((C,), 0.5).cuda(), torch.full((C,), -0.25).cuda()
    out = ops.bn2d_bwd(dy.cuda(), xc, y, wc, mean, invstd, dw, db, relu, want_dres=res)
    dx, dres = out if res else (out, None)
    close(dx, g_x, dtype)
    close(dw - 0.5, g_w, dtype, float(g_w.abs().max()) * (10 if dtype == torch.float32 else 1))
    close(db + 0.25, g_b, dtype, float(g_b.abs().max()) * (10 if dtype == torch.float32 else 1))
    if res:
        assert torch.equal(dres.float().cpu(), (dy.float() * mask).to(dtype).float())
    # run-to-run determinism of the two-level reductions (no atomics)
    dw2, db2 = torch.full((C,), 0.5).cuda(), torch.full((C,), -0.25).cuda()
    out2 = ops.bn2d_bwd(dy.cuda(), xc, y, wc, mean, invstd, dw2, db2, relu, want_dres=res)
    assert torch.equal(dw, dw2) and torch.equal(db, db2) and torch.equal(dx, out2[0] if res else out2)
    y_eval, _, _ = ops.bn2d_fwd(xc, wc, bc, rm2, rv2, False, False)
    close(y_eval, F.batch_norm(x.float(), rm2.cpu(), rv2.cpu(), w, b, False, 0.1, 1e-5), dtype)


@pytest.mark.parametrize("dtype", DTYPES)
def test_bn2d_staged_across_two_shards(dtype):
    """synchronised BatchNorm stages (dh_bn2d_sums / _fwd_apply / _bwd_apply): two shards of a batch with added sums ==
    BatchNorm over the whole batch."""
    from declip_amd import ops
    torch.manual_seed(5)
    R, C, cut = 5000, 256, 1777
    x = (torch.randn(R, C) * 1.3 - 0.2).to(dtype)
    res, dy = torch.randn(R, C).to(dtype), torch.randn(R, C).to(dtype)
    w, b = 1 + 0.1 * torch.randn(C), 0.1 * torch.randn(C)
    xr, wr, br = x.float().requires_grad_(), w.clone().requires_grad_(), b.clone().requires_grad_()
    rm_ref, rv_ref = torch.zeros(C), torch.ones(C)
    yr = F.relu(F.batch_norm(xr, rm_ref, rv_ref, wr, br, True, 0.1, 1e-5) + res.float())
    wc, bc = w.cuda(), b.cuda()
    shards = [tuple(t[a:z].contiguous().cuda() for t in (x, res, dy)) for a, z in ((0, cut), (cut, R))]
    sums = [ops.bn2d_sums(xs) for xs, _, _ in shards]
    total = sums[0] + sums[1]
    assert float(total[-1]) == R
    outs = []
    for xs, rs, _ in shards:
        rm, rv = torch.zeros(C).cuda(), torch.ones(C).cuda()
        outs.append(ops.bn2d_fwd_apply(xs, wc, bc, total, rm, rv, True, residual=rs))
        close(rm, rm_ref, torch.float32), close(rv, rv_ref, torch.float32)
    y = torch.cat([o[0] for o in outs])
    close(y, yr, dtype)
    mask = (y.float().cpu() > 0).float()
    g_x, g_w, g_b = torch.autograd.grad(F.batch_norm(xr, None, None, wr, br, True, 0.1, 1e-5), (xr, wr, br), dy.float() * mask)
    loc = [ops.bn2d_sums(xs, dy=ds, y=o[0], mean=o[1], invstd=o[2], relu=True) for (xs, _, ds), o in zip(shards, outs)]
    glob = loc[0] + loc[1]
    dxs, dws, dbs = [], [], []
    for (xs, _, ds), o, lc in zip(shards, outs, loc):
        dw, db = torch.zeros(C).cuda(), torch.zeros(C).cuda()
        dx, _ = ops.bn2d_bwd_apply(ds, xs, o[0], wc, o[1], o[2], lc, glob, dw, db, True, want_dres=True)
        dxs.append(dx), dws.append(dw), dbs.append(db)
    close(torch.cat(dxs), g_x, dtype)
    close(dws[0] + dws[1], g_w, dtype, float(g_w.abs().max()) * (10 if dtype == torch.float32 else 1))
    close(dbs[0] + dbs[1], g_b, dtype, float(g_b.abs().max()) * (10 if dtype == torch.float32 else 1))


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("N,H,W,C,k", [(2, 8, 12, 16, 2), (1, 6, 6, 8, 3), (4, 112, 112, 64, 2)])
def test_avgpool(dtype, N, H, W, C, k):
    from declip_amd import ops
    torch.manual_seed(3)
    x = torch.randn(N, C, H, W).to(dtype)
    xr = x.float().requires_grad_()
    yr = F.avg_pool2d(xr, k)
    dy = torch.randn_like(yr).to(dtype)
    yr.backward(dy.float())
    y = ops.avgpool_fwd(nhwc(x).cuda(), N, H, W, C, k)
    dx = ops.avgpool_bwd(nhwc(dy).cuda(), N, H, W, C, k)
    close(nchw(y.cpu(), N, H // k, W // k), yr, dtype)
    close(nchw(dx.cpu(), N, H, W), xr.grad, dtype)


@pytest.mark.parametrize("dtype", DTYPES)
def test_attnpool_tokens(dtype):
    from declip_amd import ops
    torch.manual_seed(4)
    b, HW, C = 5, 49, 2048
    x = torch.randn(b, HW, C).to(dtype)
    pos = torch.randn(HW + 1, C)
    xr, pr = x.float().requires_grad_(), pos.clone().requires_grad_()
    tr = torch.cat([xr.mean(dim=1, keepdim=True), xr], dim=1) + pr
    dtok = torch.randn(b, HW + 1, C).to(dtype)
    tr.backward(dtok.float())
    tok = ops.attnpool_tokens_fwd(x.reshape(b * HW, C).cuda(), pos.cuda(), b, HW)
    dpos = torch.full((HW + 1, C), 2.0).cuda()
    dx = ops.attnpool_tokens_bwd(dtok.reshape(b * (HW + 1), C).cuda(), dpos, b, HW)
    close(tok.reshape(b, HW + 1, C), tr, dtype)
    close(dx.reshape(b, HW, C), xr.grad, dtype)
    close(dpos - 2.0, pr.grad, torch.float32 if dtype == torch.float32 else dtype)


# ------------------------------------------------------------------------------------------------ the CLIP-R50 step
def run_engine(cfg, b, seed, dtype):
    from declip_amd import synth
    from declip_amd.loss import ClipInfoCELoss
    from declip_amd.testing import build_clip
    model = build_clip(cfg, dtype=dtype, seed=seed)
    images = synth.synth_images(b, res=cfg["res"], seed=seed).cuda()
    ids = synth.synth_tokens(b, ctx=cfg["ctx"], seed=seed, vocab=cfg["vocab"]).cuda()
    li, lt = model({"images": images, "captions": ids})
    loss, _ = ClipInfoCELoss()(li, lt)
    loss.backward()
    torch.cuda.synchronize()
    grads = {n: (p.grad.detach().float().cpu() if p.grad is not None else None) for n, p in model.named_parameters()}
    return model, dict(loss=float(loss), logits_i=li.materialize().detach().float().cpu(), logits_t=lt.materialize().detach().float().cpu(),
                       grads=grads, images=images)


def _is_bn(n):
    return ".bn" in n or "downsample.1." in n


def test_clip_r50_fp32_matches_reference_golden():
    """forward at the north_star tolerance (1e-3).  Gradients: the ResNet's gradient is discontinuous in its ReLU masks and a
    single flipped mask (a different fp32 summation order is enough) moves BatchNorm affine gradients -- sums of cancelling
    terms -- by ~1 % (measured on the CPU, oracle/restated.py batch_norm2d): convolution / linear weights at 5e-3,
    BatchNorm affine parameters at 5e-2 on the norm."""
    g = load_golden("clip_r50_tiny")
    model, out = run_engine(g["cfg"], g["b"], g["seed"], "fp32")
    assert abs(out["loss"] - g["loss"]) <= 1e-3 * abs(g["loss"])
    scale = float(g["logits_i"].abs().max())
    assert float((out["logits_i"] - g["logits_i"]).abs().max()) <= 1e-3 * scale
    assert float((out["logits_t"] - g["logits_t"]).abs().max()) <= 1e-3 * scale
    check_grad_digests(g["grads"], out["grads"], rtol=5e-3, only=lambda n: not _is_bn(n))
    for n, ref in g["grads"].items():
        if _is_bn(n):
            assert abs(float(out["grads"][n].double().norm()) - ref["norm"]) <= 5e-2 * ref["norm"], n
    bufs = dict(model.named_buffers())
    for k, v in g["bn_buffers"].items():
        if k.endswith("num_batches_tracked"):
            assert int(bufs[k]) == int(v)
        else:
            assert float((bufs[k].cpu() - v).abs().max()) <= 1e-3 * max(1.0, float(v.abs().max())), k
    model.eval()
    with torch.no_grad():
        feat, dense = model.visual(out["images"], return_dense=True)
    assert float((feat.cpu() - g["eval_features"]).abs().max()) <= 1e-3 * float(g["eval_features"].abs().max())
    assert abs(float(dense.double().sum()) - g["eval_dense_sum"]) <= 1e-3 * max(1.0, abs(g["eval_dense_sum"]))


def test_clip_r50_bf16_close_to_reference():
    """bf16 activations through 17 BatchNorm layers at batch 3 (bounds from the run of this test on the host emulation, where the
    bf16 arithmetic is the kernels' own: loss 1.2e-3, cosine error 6e-3, median gradient-norm error 1 %, BatchNorm affine
    gradients -- sums of cancelling terms -- up to 24 %)."""
    g = load_golden("clip_r50_tiny")
    _, out = run_engine(g["cfg"], g["b"], g["seed"], "bf16")
    assert abs(out["loss"] - g["loss"]) <= 1e-2 * abs(g["loss"])
    logit_scale = 1 / 0.07
    assert float((out["logits_i"] - g["logits_i"]).abs().max()) <= 1.5e-2 * logit_scale          # cosine similarities within 1.5e-2
    bad = []
    gmax = max(v["norm"] for v in g["grads"].values() if v is not None)
    for n, ref in g["grads"].items():
        if ref is None or ref["norm"] < 1e-3 * gmax:
            continue
        got = float(out["grads"][n].double().norm())
        if abs(got - ref["norm"]) > (0.5 if _is_bn(n) else 0.15) * ref["norm"]:
            bad.append((n, got, ref["norm"]))
    assert len(bad) <= max(2, len(g["grads"]) // 25), bad[:8]


def test_clip_r50_full_size_fp32_against_oracle():
    """BASELINE.json configs[0]: CLIP ResNet-50 + 12-layer text transformer, fp32 (batch 8 here so that the CPU oracle
    finishes in seconds; the batch-32 step itself is exercised below).

    Forward (loss, logits): the north_star 1e-3 against the fp32 oracle (measured on the MI355X: 2e-6 / 1.4e-4).
    Gradients: this 50-layer ReLU / batch-8 BatchNorm network at random init is ill-conditioned -- the reference arithmetic
    itself, run in fp32 and in fp64 on the same inputs, disagrees by 1.7e-2...2.6e-2 on the stem / layer1 / layer2 gradients
    (median 2.7e-3 over all parameters; an fp32 rounding flips ReLU masks, and the BatchNorm backward sums cancelling terms), while
    the head (attention pool, text tower) agrees at 1e-4.  A fixed tolerance against the fp32 oracle would therefore test the
    oracle's own rounding noise (first hardware run: 2.04e-2 on `visual.conv1.weight` against a 2e-2 bound, with the forward exact
    to 2e-6).  The bar used instead: against the EXACT (fp64) gradient, the HIP path may be at most 3x as far away as the
    reference's own fp32 arithmetic is, per parameter (floor 2e-3), over EVERY parameter of the model."""
    from declip_amd import synth
    cfg, b, seed = synth.R50, 8, 3
    ref = oracle_clip_run(cfg, b, 1, seed, None)
    exact = oracle_clip_run(cfg, b, 1, seed, None, dtype=torch.float64)
    _, out = run_engine(cfg, b, seed, "fp32")
    assert abs(out["loss"] - float(ref["loss"])) <= 1e-3 * abs(float(ref["loss"]))
    li = ref["per_rank"][0][0].detach()
    assert float((out["logits_i"] - li).abs().max()) <= 1e-3 * float(li.abs().max())
    gmax = max(float(v.norm()) for v in exact["grads"].values() if v is not None)
    bad, checked = [], 0
    for n, e in exact["grads"].items():
        r, g = ref["grads"][n], out["grads"][n]
        if e is None:
            assert g is None or float(g.abs().max()) == 0.0, n
            continue
        en = float(e.norm())
        if en < 1e-6 * gmax:                      # analytically-zero gradients (a bias in front of a softmax over keys): noise only
            assert float(g.double().norm()) < 1e-4 * gmax, n
            continue
        noise = float((r.double() - e).norm()) / en
        err = float((g.double() - e).norm()) / en
        checked += 1
        if err > max(3.0 * noise, 2e-3):
            bad.append((n, err, noise))
    assert checked > 300 and not bad, bad[:8]
    for n in ("visual.attnpool.c_proj.weight", "encode_text.text_projection.weight"):      # well-conditioned: tight, vs the fp32 oracle
        r = ref["grads"][n]
        assert float((out["grads"][n] - r).norm()) <= 1e-3 * float(r.norm()), n


@pytest.mark.parametrize("dtype", ["fp32", "bf16"])
def test_clip_r50_batch32_step_trains(dtype):
    """configs[0] batch: three optimiser steps on one seeded batch must lower the loss (fp32 and bf16)."""
    from declip_amd import synth
    from declip_amd.loss import ClipInfoCELoss
    from declip_amd.optim import build_adamw
    from declip_amd.testing import build_clip
    cfg, b = synth.R50, 32
    model = build_clip(cfg, dtype=dtype, seed=0)
    opt = build_adamw(model, lr=1e-4, betas=(0.9, 0.98), eps=1e-8, weight_decay=0.1)
    crit = ClipInfoCELoss()
    batch = {"images": synth.synth_images(b, seed=0).cuda(), "captions": synth.synth_tokens(b, seed=0).cuda()}
    losses = []
    for _ in range(4):
        opt.zero_grad()
        li, lt = model(batch)
        loss, _ = crit(li, lt)
        loss.backward()
        opt.step()
        losses.append(float(loss))
    assert all(l == l for l in losses) and losses[-1] < losses[0], losses


def test_declip_r50_fp32_matches_reference_golden():
    """declip_res50 (model/declip.py:339-346): both views through the ModifiedResNet, full DeCLIP loss composition."""
    from declip_amd.heads import SimsiamLoss
    from declip_amd.loss import ClipInfoCELoss, NTXentLoss
    from declip_amd.steps import declip_loss
    from declip_amd.testing import build_declip, declip_batch
    g = load_golden("declip_r50_tiny")
    cfg, b, seed = g["cfg"], g["b"], g["seed"]
    model = build_declip(cfg, dtype="fp32", seed=seed, nn_size=g["nn_size"])
    batch = declip_batch(cfg, b, seed=seed)
    out = declip_loss(model, batch, ClipInfoCELoss(), SimsiamLoss(), NTXentLoss(b))
    out["loss"].backward()
    torch.cuda.synchronize()
    assert abs(float(out["loss"].detach()) - g["loss"]) <= 1e-3 * abs(g["loss"])
    for k in ("clip", "nn", "simsiam", "mlm", "convirt"):
        assert abs(float(out["parts"][k]) - g["parts"][k]) <= 1e-3 * max(1.0, abs(g["parts"][k])), k
    grads = {n: (p.grad.detach().float().cpu() if p.grad is not None else None) for n, p in model.named_parameters()}
    check_grad_digests(g["grads"], grads, rtol=5e-3, only=lambda n: not _is_bn(n))
    bufs = dict(model.named_buffers())
    assert int(bufs["visual.bn2.num_batches_tracked"]) == 2


def test_filip_r50_fp32_matches_reference_golden():
    """filip_res50 (model/filip.py:146-153): token-wise max-sim loss on the 7x7 feature map of the ModifiedResNet."""
    from declip_amd.loss import ClipInfoCELoss
    from declip_amd.steps import filip_loss
    from declip_amd.testing import build_filip, filip_batch
    g = load_golden("filip_r50_tiny")
    cfg, b, seed = g["cfg"], g["b"], g["seed"]
    model = build_filip(cfg, dtype="fp32", seed=seed)
    out = filip_loss(model, filip_batch(cfg, b, seed=seed), ClipInfoCELoss())
    out["loss"].backward()
    torch.cuda.synchronize()
    assert abs(float(out["loss"].detach()) - g["loss"]) <= 1e-3 * abs(g["loss"])
    dli = out["outputs"]["dense_logits"][0].detach().cpu()
    assert float((dli - g["dense_logits_i"]).abs().max()) <= 1e-3 * float(g["dense_logits_i"].abs().max())
    grads = {n: (p.grad.detach().float().cpu() if p.grad is not None else None) for n, p in model.named_parameters()}
    check_grad_digests(g["grads"], grads, rtol=5e-3, only=lambda n: not _is_bn(n))


def test_clip_r50_fc_head_fp32_matches_reference_golden():
    """64 px input -> 2x2 final map -> adaptive average pool + fc (modified_resnet.py:209-211) instead of the attention pool."""
    g = load_golden("clip_r50_fc")
    model, out = run_engine(g["cfg"], g["b"], g["seed"], "fp32")
    assert abs(out["loss"] - g["loss"]) <= 1e-3 * abs(g["loss"])
    assert float((out["logits_i"] - g["logits_i"]).abs().max()) <= 1e-3 * float(g["logits_i"].abs().max())
    check_grad_digests(g["grads"], out["grads"], rtol=5e-3, only=lambda n: not _is_bn(n))
    assert out["grads"]["visual.attnpool.q_proj.weight"] is None


def test_image_resized_crop_u8_matches_oracle():
    """on-GPU RandomResizedCrop / Resize + CenterCrop of decoded uint8 images (dh_image_resized_crop_u8) against the oracle
    restatement (torch antialiased bilinear), full-size 224 px outputs from mixed source sizes."""
    import numpy as np
    from declip_amd import augment, ops
    from oracle import restated
    g = torch.Generator().manual_seed(0)
    sizes = [(480, 640), (333, 500), (600, 400), (224, 224), (1080, 720), (256, 341)]
    canvas = torch.randint(0, 256, (len(sizes), 1080, 720, 3), generator=g, dtype=torch.uint8)
    rng = np.random.default_rng(1)
    flip = torch.tensor([0, 1, 0, 1, 1, 0], dtype=torch.uint8)
    for params in (augment.random_resized_crop_params(sizes, (224, 224), generator=rng), augment.resize_center_crop_params(sizes, 256, 224)):
        # unrounded: the filter itself, tight
        out = ops.image_resized_crop_u8(canvas.cuda(), torch.from_numpy(params).cuda(), (224, 224), flip=flip.cuda(), round_u8=False)
        ref = restated.image_resized_crop_u8(canvas, params, (224, 224), flip=flip, round_u8=False)
        assert float((out.cpu() - ref).abs().max()) <= 1e-4
        # rounded to grey levels like the PIL pipeline: identical except where the value sits on a rounding boundary
        out = ops.image_resized_crop_u8(canvas.cuda(), torch.from_numpy(params).cuda(), (224, 224), flip=flip.cuda())
        ref = restated.image_resized_crop_u8(canvas, params, (224, 224), flip=flip)
        diff = (out.cpu() - ref).abs()
        assert float(diff.max()) <= 1.01 / 255 / 0.224 and float((diff > 1e-6).float().mean()) <= 1e-3


# ------------------------------------------------------------------------------------------------ packed captions (DH_TEXT_PACKED=1)
@pytest.mark.parametrize("name", ["clip_tiny", "clip_vitb32_b8"])
def test_packed_text_tower_fp32_matches_reference_golden(monkeypatch, name):
    """the text tower on the rows up to <|endoftext|> only (engine.PackedCaptions) passes the same reference goldens."""
    import test_gpu_clip as G
    monkeypatch.setenv("DH_TEXT_PACKED", "1")
    G.test_clip_fp32_matches_reference_golden(name)


@pytest.mark.parametrize("packed_mode,pooled", [("1", "0"), ("2", "0"), ("0", "1"), ("1", "1")])
def test_packed_text_tower_bf16_full_size_equals_padded(monkeypatch, packed_mode, pooled):
    """ViT-B/32 + 12-layer text tower, batch 256, bf16: the step with the flop-saving switches on vs the padded / full-last-block
    step on the same weights and batch (the GEMMs of the packed tower see M = sum of caption lengths rounded up to whole 256-row
    tiles; the pooled last block runs its GEMMs on 256 rows)."""
    from declip_amd import synth
    from declip_amd.loss import ClipInfoCELoss
    from declip_amd.testing import build_clip
    cfg, b = synth.VITB32, 256
    images, ids = synth.synth_images(b, seed=4).cuda(), synth.synth_tokens(b, seed=4).cuda()
    res = {}
    for packed, pl in (("0", "0"), (packed_mode, pooled)):
        monkeypatch.setenv("DH_TEXT_PACKED", packed)
        monkeypatch.setenv("DH_POOLED_LAST", pl)
        model = build_clip(cfg, dtype="bf16", seed=2)
        li, lt = model({"images": images, "captions": ids})
        loss, _ = ClipInfoCELoss()(li, lt)
        loss.backward()
        torch.cuda.synchronize()
        res[(packed, pl)] = (float(loss), {n: p.grad.detach().float().cpu() for n, p in model.named_parameters() if p.grad is not None})
    (l0, g0), (l1, g1) = res[("0", "0")], res[(packed_mode, pooled)]
    assert abs(l0 - l1) <= 2e-3 * abs(l0)
    for n in g0:
        if float(g0[n].norm()) > 1e-6:
            assert abs(float(g1[n].norm()) - float(g0[n].norm())) <= 3e-2 * float(g0[n].norm()), n


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("causal", [True, False])
def test_attention_varlen_matches_per_sequence_reference(dtype, causal):
    """dh_attn_varlen_fwd / _bwd: packed rows, sequence i = rows cu[i] .. cu[i+1], against attention computed sequence by sequence."""
    from declip_amd import ops
    torch.manual_seed(0)
    heads, hd, Lmax = 8, 64, 77
    lens = [5, 77, 1, 33, 16, 64] * 40                 # 240 sequences: several pairs per persistent workgroup
    b, d = len(lens), heads * hd
    rows = sum(lens)
    rows_pad = (rows + 255) // 256 * 256
    cu = torch.tensor([0] + list(torch.tensor(lens).cumsum(0)), dtype=torch.int32)
    qkv = (torch.randn(rows_pad, 3 * d) * 0.7).to(dtype)
    dout = torch.randn(rows_pad, d).to(dtype)
    out, lse = ops.attn_varlen_fwd(qkv.cuda(), cu.cuda(), rows, b, Lmax, heads, causal)
    dqkv = ops.attn_varlen_bwd(qkv.cuda(), out, dout.cuda(), lse, cu.cuda(), rows, b, Lmax, heads, causal)
    out, dqkv, lse = out.float().cpu(), dqkv.float().cpu(), lse.cpu()
    assert float(out[rows:].abs().max()) == 0.0 and float(dqkv[rows:].abs().max()) == 0.0
    tol = 2e-5 if dtype == torch.float32 else 2e-2
    for i in list(range(8)) + [b - 1]:
        n, r0 = lens[i], int(cu[i])
        x = qkv[r0:r0 + n].float().requires_grad_()
        q, k, v = [t.reshape(n, heads, hd).transpose(0, 1) for t in x.split(d, dim=1)]
        s = (q @ k.transpose(1, 2)) * hd ** -0.5
        if causal:
            s = s + torch.full((n, n), float("-inf")).triu_(1)
        ref = (torch.softmax(s, dim=-1) @ v).transpose(0, 1).reshape(n, d)
        ref.backward(dout[r0:r0 + n].float())
        assert float((out[r0:r0 + n] - ref.detach()).abs().max()) <= tol * max(1.0, float(ref.abs().max())), (i, n)
        assert float((dqkv[r0:r0 + n] - x.grad).abs().max()) <= tol * max(1.0, float(x.grad.abs().max())) * (1 if dtype == torch.float32 else 3), (i, n)


@pytest.mark.parametrize("nseq,device_rows", [(240, False), (240, True), (7, True)])
def test_attention_length_buckets_equal_the_plain_varlen_kernels(nseq, device_rows):
    """dh_attn_bucketed_fwd / _bwd (captions of at most 48 tokens on the 3-key-block instantiation, the rest on the 5-block one;
    sequence lists and counts read on the device) against dh_attn_varlen_*: same values per sequence (extra key blocks of the long
    instantiation hold zero rows that the masks skip), the padding rows zero, also with rows = -1 and with one of the buckets empty."""
    from declip_amd import ops
    torch.manual_seed(3)
    heads, hd, Lmax, Ls = 8, 64, 77, 48
    base = [5, 77, 1, 33, 48, 64, 49, 16]
    lens = (base * ((nseq + 7) // 8))[:nseq] if nseq > 8 else [60, 77, 50, 49, 70, 55, 66][:nseq]      # (7: every caption long -> empty short bucket)
    b, d = len(lens), heads * hd
    rows = sum(lens)
    rows_pad = (rows + 255) // 256 * 256
    lt = torch.tensor(lens)
    cu = torch.tensor([0] + list(lt.cumsum(0)), dtype=torch.int32).cuda()
    short = lt <= Ls
    order = torch.sort((~short).to(torch.int32), stable=True)[1].to(torch.int32).cuda()
    ns = int(short.sum())
    ranges = torch.tensor([0, ns, ns, b - ns], dtype=torch.int32).cuda()
    qkv = (torch.randn(rows_pad, 3 * d) * 0.7).to(torch.bfloat16).cuda()
    dout = torch.randn(rows_pad, d).to(torch.bfloat16).cuda()
    r = -1 if device_rows else rows
    o0, l0 = ops.attn_varlen_fwd(qkv, cu, rows, b, Lmax, heads, True)
    g0 = ops.attn_varlen_bwd(qkv, o0, dout, l0, cu, rows, b, Lmax, heads, True)
    o1, l1 = ops.attn_bucketed_fwd(qkv, cu, order, ranges, r, b, Lmax, Ls, heads, True)
    g1 = ops.attn_bucketed_bwd(qkv, o1, dout, l1, cu, order, ranges, r, b, Lmax, Ls, heads, True)
    torch.cuda.synchronize()
    assert float(o1[rows:].float().abs().max()) == 0.0 and float(g1[rows:].float().abs().max()) == 0.0
    assert float((o1.float() - o0.float()).abs().max()) <= 1e-6 * float(o0.float().abs().max())
    assert float((g1.float() - g0.float()).abs().max()) <= 1e-6 * float(g0.float().abs().max())
    for i in range(b):                                    # lse: [b][heads][Lmax], valid entries only
        assert torch.equal(l1[i, :, :lens[i]], l0[i, :, :lens[i]])


def test_packed_text_tower_gather_mode_fp32_matches_reference_golden(monkeypatch):
    import test_gpu_clip as G
    monkeypatch.setenv("DH_TEXT_PACKED", "2")
    G.test_clip_fp32_matches_reference_golden("clip_tiny")


# ------------------------------------------------------------------------------------------------ last block for the pooled rows (DH_POOLED_LAST=1)
@pytest.mark.parametrize("ordered", [False, True])
@pytest.mark.parametrize("dtype", DTYPES)
def test_attention_pooled_query_matches_reference(dtype, ordered):
    """ordered: the launch zeroes the dkv rows no sequence owns (uninitialised buffer handed over: the allocator's free block of that
    size is poisoned with NaN first); else the caller's fill does."""
    from declip_amd import ops
    torch.manual_seed(1)
    heads, hd, L, b = 12, 64, 50, 300
    d = heads * hd
    q = (torch.randn(b, d) * 0.8).to(dtype)
    kv = (torch.randn(b * L, 2 * d) * 0.8).to(dtype)
    dout = torch.randn(b, d).to(dtype)
    row0 = (torch.arange(b) * L).to(torch.int32)
    nkeys = torch.randint(1, L + 1, (b,), generator=torch.Generator().manual_seed(2)).to(torch.int32)
    out, lse = ops.attn_pooled_fwd(q.cuda(), kv.cuda(), row0.cuda(), nkeys.cuda(), heads, L)
    kv_d = kv.cuda()
    poison = torch.full_like(kv_d, float("nan"))
    del poison
    dq, dkv = ops.attn_pooled_bwd(q.cuda(), kv_d, dout.cuda(), lse, row0.cuda(), nkeys.cuda(), heads, L, ordered=ordered)
    out, dq, dkv = out.float().cpu(), dq.float().cpu(), dkv.float().cpu()
    unowned = (torch.arange(L)[None, :] >= nkeys[:, None]).reshape(-1)
    assert float(dkv[unowned].abs().max()) == 0.0 and bool(torch.isfinite(dkv).all())
    tol = 2e-5 if dtype == torch.float32 else 2e-2
    for i in (0, 1, 17, b - 1):
        r0, n = int(row0[i]), int(nkeys[i])
        qi, kvi = q[i].float().requires_grad_(), kv[r0:r0 + n].float().requires_grad_()
        k, v = kvi[:, :d].reshape(n, heads, hd), kvi[:, d:].reshape(n, heads, hd)
        s = torch.einsum("hc,nhc->hn", qi.reshape(heads, hd), k) * hd ** -0.5
        ref = torch.einsum("hn,nhc->hc", torch.softmax(s, dim=-1), v).reshape(d)
        ref.backward(dout[i].float())
        assert float((out[i] - ref.detach()).abs().max()) <= tol * max(1.0, float(ref.abs().max()))
        assert float((dq[i] - qi.grad).abs().max()) <= 3 * tol * max(1.0, float(qi.grad.abs().max()))
        assert float((dkv[r0:r0 + n] - kvi.grad).abs().max()) <= 3 * tol * max(1.0, float(kvi.grad.abs().max()))
        assert float(dkv[r0 + n:r0 + L].abs().max() if n < L else 0.0) == 0.0


@pytest.mark.parametrize("env", [{"DH_POOLED_LAST": "1"}, {"DH_POOLED_LAST": "1", "DH_TEXT_PACKED": "1"}])
@pytest.mark.parametrize("name", ["clip_tiny", "clip_vitb32_b8"])
def test_pooled_last_block_fp32_matches_reference_golden(monkeypatch, env, name):
    import test_gpu_clip as G
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    G.test_clip_fp32_matches_reference_golden(name)
