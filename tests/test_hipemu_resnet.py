"""The ModifiedResNet kernels (csrc/resnet_ops.hip) executed on the host (tests/hipemu_util.py) through the same ops wrappers
and C-ABI entry points as on the GPU, against torch CPU references.  fp32 and bf16 storage."""
import pytest
import torch
import torch.nn.functional as F

from hipemu_util import emu_ops

SYMS = ["dh_conv_rows", "dh_bn2d_ws_bytes", "dh_bn2d_fwd", "dh_bn2d_bwd", "dh_bn2d_sums", "dh_bn2d_fwd_apply", "dh_bn2d_bwd_apply",
        "dh_avgpool_fwd", "dh_avgpool_bwd", "dh_attnpool_tokens_fwd", "dh_attnpool_tokens_bwd"]
DTYPES = [torch.float32, torch.bfloat16]


def _tol(dtype):
    return 2e-5 if dtype == torch.float32 else 2e-2


def nhwc(x):            # [N,C,H,W] -> pixel rows [N*H*W, C]
    return x.permute(0, 2, 3, 1).reshape(-1, x.shape[1]).contiguous()


def nchw(rows, N, H, W):
    return rows.reshape(N, H, W, -1).permute(0, 3, 1, 2).contiguous()


def close(a, b, dtype, scale=None):
    a, b = a.detach().float(), b.detach().float()
    s = float(b.abs().max()) if scale is None else scale
    assert float((a - b).abs().max()) <= _tol(dtype) * max(s, 1e-6), (float((a - b).abs().max()), s)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("N,H,W,C,stride", [(2, 9, 7, 16, 1), (1, 12, 12, 8, 2), (3, 5, 6, 24, 1)])
def test_conv_rows_nhwc(dtype, N, H, W, C, stride):
    torch.manual_seed(0)
    x = torch.randn(N, C, H, W).to(dtype)
    with emu_ops(["resnet_ops.hip"], SYMS) as ops:
        rows, Ho, Wo = ops.conv_rows(nhwc(x), N, H, W, C, stride=stride, pad=1)
    ref = F.unfold(x.float(), 3, padding=1, stride=stride)            # [N, C*9, L], inner order (c, ky, kx)
    assert (Ho, Wo) == ((H - 1) // stride + 1, (W - 1) // stride + 1)
    assert torch.equal(rows.float(), ref.transpose(1, 2).reshape(N * Ho * Wo, C * 9))      # a pure gather: exact
    # the GEMM it feeds == the convolution
    w = torch.randn(5, C, 3, 3)
    y = rows.float() @ w.view(5, -1).t()
    close(nchw(y, N, Ho, Wo), F.conv2d(x.float(), w, stride=stride, padding=1), torch.float32, None)


@pytest.mark.parametrize("dtype", DTYPES)
def test_conv_rows_image_view(dtype):
    torch.manual_seed(1)
    N, H, W = 2, 10, 14
    img = torch.randn(N, 6, H, W)                                      # two channel-stacked views
    for c0 in (0, 3):
        with emu_ops(["resnet_ops.hip"], SYMS) as ops:
            rows, Ho, Wo = ops.conv_rows_image(img, c0, dtype, stride=2, pad=1)
        ref = F.unfold(img[:, c0:c0 + 3], 3, padding=1, stride=2).transpose(1, 2).reshape(N * Ho * Wo, 27)
        assert (Ho, Wo) == (5, 7) and rows.shape == (N * Ho * Wo, 32)
        assert torch.equal(rows[:, :27].float(), ref.to(dtype).float())
        assert float(rows[:, 27:].float().abs().max()) == 0.0


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("R,C,relu,res", [(37, 8, True, False), (300, 16, True, True), (1000, 40, False, False), (64, 2056, True, True)])
def test_bn2d_fwd_bwd(dtype, R, C, relu, res):
    torch.manual_seed(2)
    x = (torch.randn(R, C) * 1.5 + 0.3).to(dtype)
    r = torch.randn(R, C).to(dtype) if res else None
    w, b = 1 + 0.1 * torch.randn(C), 0.1 * torch.randn(C)
    rm, rv = torch.randn(C) * 0.1, torch.rand(C) + 0.5
    dy = torch.randn(R, C).to(dtype)
    # reference (fp32 math on the stored values)
    xr, wr, br = x.float().requires_grad_(), w.clone().requires_grad_(), b.clone().requires_grad_()
    rr = r.float().requires_grad_() if res else None
    rm_ref, rv_ref = rm.clone(), rv.clone()
    yr = F.batch_norm(xr, rm_ref, rv_ref, wr, br, True, 0.1, 1e-5)
    if res:
        yr = yr + rr
    if relu:
        yr = F.relu(yr)
    with emu_ops(["resnet_ops.hip"], SYMS) as ops:
        rm2, rv2 = rm.clone(), rv.clone()
        y, mean, invstd = ops.bn2d_fwd(x, w, b, rm2, rv2, relu, True, residual=r)
        close(y, yr.detach(), dtype)
        close(rm2, rm_ref, torch.float32), close(rv2, rv_ref, torch.float32)
        close(mean, x.float().mean(0), torch.float32)
        # backward: the mask comes from OUR y (bf16 rounding can move a value across 0), so differentiate the reference with it
        mask = (y.float() > 0).float() if relu else torch.ones(R, C)
        (g_x, g_w, g_b) = torch.autograd.grad(F.batch_norm(xr, None, None, wr, br, True, 0.1, 1e-5), (xr, wr, br), dy.float() * mask)
        dw, db = torch.full((C,), 0.5), torch.full((C,), -0.25)        # accumulate-into contract
        out = ops.bn2d_bwd(dy, x, y, w, mean, invstd, dw, db, relu, want_dres=res)
        dx, dres = out if res else (out, None)
        close(dx, g_x, dtype)
        close(dw - 0.5, g_w, torch.float32 if dtype == torch.float32 else dtype, float(g_w.abs().max()))
        close(db + 0.25, g_b, torch.float32 if dtype == torch.float32 else dtype, float(g_b.abs().max()))
        if res:
            assert torch.equal(dres.float(), (dy.float() * mask).to(dtype).float())
        # eval mode: running statistics
        y_eval, _, _ = ops.bn2d_fwd(x, w, b, rm2, rv2, False, False)
        close(y_eval, F.batch_norm(x.float(), rm2, rv2, w, b, False, 0.1, 1e-5), dtype)


@pytest.mark.parametrize("dtype", DTYPES)
def test_bn2d_staged_across_two_shards(dtype):
    """the three-stage entry points of synchronised BatchNorm: two "ranks" hold the halves of a batch, their sums are added
    (what the all-reduce does) -> identical to BatchNorm over the whole batch; dw / db are per-rank partial sums."""
    torch.manual_seed(5)
    R, C = 90, 24
    x = (torch.randn(R, C) * 1.3 - 0.2).to(dtype)
    res = torch.randn(R, C).to(dtype)
    dy = torch.randn(R, C).to(dtype)
    w, b = 1 + 0.1 * torch.randn(C), 0.1 * torch.randn(C)
    cut = 37                                                       # uneven shards
    xr, wr, br = x.float().requires_grad_(), w.clone().requires_grad_(), b.clone().requires_grad_()
    rm_ref, rv_ref = torch.zeros(C), torch.ones(C)
    yr = F.relu(F.batch_norm(xr, rm_ref, rv_ref, wr, br, True, 0.1, 1e-5) + res.float())
    with emu_ops(["resnet_ops.hip"], SYMS) as ops:
        shards = [(x[:cut].contiguous(), res[:cut].contiguous(), dy[:cut].contiguous()), (x[cut:].contiguous(), res[cut:].contiguous(), dy[cut:].contiguous())]
        sums = [ops.bn2d_sums(xs) for xs, _, _ in shards]
        assert float(sums[0][-1]) == cut and float(sums[1][-1]) == R - cut
        total = sums[0] + sums[1]
        outs, rms = [], []
        for xs, rs, _ in shards:
            rm, rv = torch.zeros(C), torch.ones(C)
            outs.append(ops.bn2d_fwd_apply(xs, w, b, total, rm, rv, True, residual=rs))
            rms.append((rm, rv))
        y = torch.cat([o[0] for o in outs])
        close(y, yr.detach(), dtype)
        for rm, rv in rms:                                         # every rank ends with the global running statistics
            close(rm, rm_ref, torch.float32), close(rv, rv_ref, torch.float32)
        mask = (y.float() > 0).float()
        g_x, g_w, g_b = torch.autograd.grad(F.batch_norm(xr, None, None, wr, br, True, 0.1, 1e-5), (xr, wr, br), dy.float() * mask)
        ys = [o[0] for o in outs]
        loc = [ops.bn2d_sums(xs, dy=ds, y=yy, mean=o[1], invstd=o[2], relu=True) for (xs, _, ds), yy, o in zip(shards, ys, outs)]
        glob = loc[0] + loc[1]
        dxs, dws, dbs = [], [], []
        for (xs, _, ds), yy, o, lc in zip(shards, ys, outs, loc):
            dw, db = torch.zeros(C), torch.zeros(C)
            dx, dres = ops.bn2d_bwd_apply(ds, xs, yy, w, o[1], o[2], lc, glob, dw, db, True, want_dres=True)
            dxs.append(dx), dws.append(dw), dbs.append(db)
        close(torch.cat(dxs), g_x, dtype)
        close(dws[0] + dws[1], g_w, dtype, float(g_w.abs().max())), close(dbs[0] + dbs[1], g_b, dtype, float(g_b.abs().max()))


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("N,H,W,C,k", [(2, 8, 12, 16, 2), (1, 6, 6, 8, 3), (3, 4, 4, 32, 1)])
def test_avgpool(dtype, N, H, W, C, k):
    torch.manual_seed(3)
    x = torch.randn(N, C, H, W).to(dtype)
    xr = x.float().requires_grad_()
    yr = F.avg_pool2d(xr, k)
    dy = torch.randn_like(yr).to(dtype)
    yr.backward(dy.float())
    with emu_ops(["resnet_ops.hip"], SYMS) as ops:
        y = ops.avgpool_fwd(nhwc(x), N, H, W, C, k)
        dx = ops.avgpool_bwd(nhwc(dy), N, H, W, C, k)
    close(nchw(y, N, H // k, W // k), yr.detach(), dtype)
    close(nchw(dx, N, H, W), xr.grad, dtype)


@pytest.mark.parametrize("dtype", DTYPES)
def test_attnpool_tokens(dtype):
    torch.manual_seed(4)
    b, HW, C = 3, 49, 24
    x = torch.randn(b, HW, C).to(dtype)
    pos = torch.randn(HW + 1, C)
    xr, pr = x.float().requires_grad_(), pos.clone().requires_grad_()
    tr = torch.cat([xr.mean(dim=1, keepdim=True), xr], dim=1) + pr
    dtok = torch.randn(b, HW + 1, C).to(dtype)
    tr.backward(dtok.float())
    with emu_ops(["resnet_ops.hip"], SYMS) as ops:
        tok = ops.attnpool_tokens_fwd(x.reshape(b * HW, C), pos, b, HW)
        dpos = torch.full((HW + 1, C), 2.0)
        dx = ops.attnpool_tokens_bwd(dtok.reshape(b * (HW + 1), C), dpos, b, HW)
    close(tok.reshape(b, HW + 1, C), tr.detach(), dtype)
    close(dx.reshape(b, HW, C), xr.grad, dtype)
    close(dpos - 2.0, pr.grad, torch.float32)


def test_argument_errors_are_reported():
    from declip_amd.lib import DeclipHipError
    with emu_ops(["resnet_ops.hip"], SYMS) as ops:
        with pytest.raises(DeclipHipError, match="multiple of 8"):
            ops.bn2d_fwd(torch.randn(4, 12), torch.ones(12), torch.zeros(12), None, None, True, True)
        with pytest.raises(DeclipHipError, match="multiples of k"):
            ops.avgpool_fwd(torch.randn(2 * 5 * 5, 8), 2, 5, 5, 8, 2)
