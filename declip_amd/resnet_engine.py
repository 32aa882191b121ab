"""ModifiedResNet tower on the HIP engine (reference: model/image_encoder/modified_resnet.py:14-214).

MI355X-first arrangement: every activation is NHWC stored as pixel rows [N*H*W, C] -- the token-major layout of the
transformer towers -- so that
  * the 32 one-by-one convolutions of a ResNet-50 (conv1 / conv3 / downsample of every bottleneck: ~70 % of its MACs) are
    plain GEMMs on the activations AS STORED, forward, dX and dW, through the same MFMA kernels as the transformer blocks;
  * a 3x3 convolution is `conv_rows` (patches in the weight's own (c, ky, kx) order) + one GEMM against
    conv.weight.view(Cout, Cin*9) as stored; its input gradient is the same pair on dY with the flipped/transposed kernel,
    its weight gradient one k-major GEMM straight into the flat gradient buffer (patches are re-gathered in the backward
    pass rather than kept: 9x the activation);
  * BatchNorm statistics are deterministic column reductions, and BatchNorm + ReLU (+ the residual add of the bottleneck
    output) is one pass; the bottleneck's shortcut gradient rides on the residual input of the conv1 dX GEMM;
  * the attention pool reuses the transformer attention kernel on [b, 50, C] tokens (the last feature map as stored).

All arithmetic goes through declip_amd.ops (the C-ABI); torch handles memory and autograd plumbing only.
"""
import torch

from . import ops
from .engine import _to_act, gemm_workspace, weight_grad
from .lib import DeclipHipError


def _ws(t):
    return gemm_workspace(t.device) if t.is_cuda and t.dtype == torch.bfloat16 else None


def _w2d(flat, conv):
    """conv.weight in the compute dtype viewed [Cout, Cin*k*k] (as stored)."""
    w = flat.wview(conv.weight)
    return w.view(w.shape[0], -1)


def _w_flipped(flat, conv):
    """[Cin, Cout*9] with both taps mirrored: the kernel of the transposed 3x3 convolution (dX = conv(dY, W^T flipped)).
    A permutation copy of the compute-dtype weight (data movement only)."""
    w = flat.wview(conv.weight)                                  # [Cout, Cin, 3, 3]
    return w.flip(2, 3).permute(1, 0, 2, 3).reshape(w.shape[1], -1).contiguous()


def bn_group(tower):
    """Process group the tower's BatchNorm layers synchronise their statistics over, or None for per-rank statistics.
    `use_sync_bn: True` with `bn_group_size: g` (modified_resnet.py:118-140, simple_group_split :98-105): consecutive ranks in
    groups of g; g <= 1 is per-rank BatchNorm, g >= world the whole job.  Created once, collectively, at the first forward."""
    from . import dist as dh_dist
    import torch.distributed as tdist
    if not getattr(tower, "use_sync_bn", False) or not dh_dist.is_dist():
        return None
    cached = tower.__dict__.get("_bn_group")
    if cached is not None:
        return cached[0]
    world, rank, g = tdist.get_world_size(), tdist.get_rank(), int(tower.bn_group_size)
    grp = None
    if g >= world:
        grp = tdist.group.WORLD
    elif g > 1:
        if world % g:
            raise DeclipHipError("bn_group_size %d does not divide the world size %d" % (g, world))
        for first in range(0, world, g):                          # every rank creates every group (collective call)
            h = tdist.new_group(list(range(first, first + g)))
            if first <= rank < first + g:
                grp = h
    tower.__dict__["_bn_group"] = (grp,)
    return grp


_TRACKED = []      # num_batches_tracked buffers touched by the pass in flight: bumped together (53 tiny launches -> 1)


def _bump_tracked():
    if _TRACKED:
        torch._foreach_add_(list(_TRACKED), 1)
        _TRACKED.clear()


class _BN:
    """forward + saved state of one BatchNorm2d (+ReLU (+residual)) application; `group` != None synchronises the batch
    statistics over that process group (one all-reduce of [2C+1] doubles per direction)."""

    __slots__ = ("bn", "x", "y", "mean", "invstd", "relu", "has_res", "group")

    def __init__(self, bn, x, relu, training, residual=None, group=None):
        self.bn, self.x, self.relu, self.has_res = bn, x, relu, residual is not None
        self.group = group if training else None
        track = bn.track_running_stats and bn.running_mean is not None
        rm = bn.running_mean if (not training or track) else None
        rv = bn.running_var if rm is not None else None
        momentum = 0.1 if bn.momentum is None else bn.momentum
        if self.group is not None:
            import torch.distributed as tdist
            sums = ops.bn2d_sums(x)
            tdist.all_reduce(sums, group=self.group)
            self.y, self.mean, self.invstd = ops.bn2d_fwd_apply(x, bn.weight.data, bn.bias.data, sums, rm, rv, relu, residual=residual,
                                                                eps=bn.eps, momentum=momentum)
        else:
            self.y, self.mean, self.invstd = ops.bn2d_fwd(x, bn.weight.data, bn.bias.data, rm, rv, relu, training, residual=residual,
                                                          eps=bn.eps, momentum=momentum)
        if training and track and bn.num_batches_tracked is not None:
            _TRACKED.append(bn.num_batches_tracked)              # buffer bookkeeping (nn.BatchNorm2d.forward), one launch per pass

    def backward(self, flat, dy):
        g = flat.gview
        bn = self.bn
        if self.group is not None:
            import torch.distributed as tdist
            local = ops.bn2d_sums(self.x, dy=dy, y=self.y, mean=self.mean, invstd=self.invstd, relu=self.relu)
            glob = local.clone()
            tdist.all_reduce(glob, group=self.group)
            return ops.bn2d_bwd_apply(dy, self.x, self.y, bn.weight.data, self.mean, self.invstd, local, glob, g(bn.weight), g(bn.bias),
                                      self.relu, want_dres=self.has_res)
        return ops.bn2d_bwd(dy, self.x, self.y, bn.weight.data, self.mean, self.invstd, g(bn.weight), g(bn.bias),
                            self.relu, want_dres=self.has_res)


class _Conv3:
    """3x3 convolution (stride 1, pad 1) on pixel rows: patches + GEMM."""

    __slots__ = ("conv", "x", "geom")

    def __init__(self, conv, x, N, H, W):
        self.conv, self.x, self.geom = conv, x, (N, H, W, conv.weight.shape[1])

    def forward(self, flat):
        N, H, W, C = self.geom
        rows, _, _ = ops.conv_rows(self.x, N, H, W, C, stride=1, pad=1)
        return ops.gemm(rows, _w2d(flat, self.conv), ws=_ws(rows))

    def backward(self, flat, dy, need_dx=True):
        N, H, W, C = self.geom
        rows, _, _ = ops.conv_rows(self.x, N, H, W, C, stride=1, pad=1)       # re-gathered, not kept
        gw = flat.gview(self.conv.weight)
        weight_grad(dy, rows, gw.view(gw.shape[0], -1))
        if not need_dx:
            return None
        drows, _, _ = ops.conv_rows(dy, N, H, W, dy.shape[1], stride=1, pad=1)
        return ops.gemm(drows, _w_flipped(flat, self.conv), ws=_ws(drows))


def _conv1x1_bwd(flat, conv, dy, x, residual=None):
    gw = flat.gview(conv.weight)
    weight_grad(dy, x, gw.view(gw.shape[0], -1))
    return ops.gemm(dy, _w2d(flat, conv), b_kmajor=True, residual=residual, ws=_ws(dy))


class _Block:
    """One Bottleneck (modified_resnet.py:14-56): forward at construction, backward() returns the input gradient."""

    def __init__(self, flat, blk, x, N, H, W, training, group=None):
        self.blk, self.x, self.geom = blk, x, (N, H, W)
        s = blk.stride
        self.c1 = ops.gemm(x, _w2d(flat, blk.conv1), ws=_ws(x))
        self.b1 = _BN(blk.bn1, self.c1, True, training, group=group)
        self.k2 = _Conv3(blk.conv2, self.b1.y, N, H, W)
        self.b2 = _BN(blk.bn2, self.k2.forward(flat), True, training, group=group)
        P = self.b2.y.shape[1]
        self.p2 = ops.avgpool_fwd(self.b2.y, N, H, W, P, s) if s > 1 else self.b2.y
        c3 = ops.gemm(self.p2, _w2d(flat, blk.conv3), ws=_ws(x))
        self.bd = None
        identity = x
        if blk.downsample is not None:
            self.px = ops.avgpool_fwd(x, N, H, W, x.shape[1], s) if s > 1 else x
            cd = ops.gemm(self.px, _w2d(flat, blk.downsample[1]), ws=_ws(x))
            self.bd = _BN(blk.downsample[2], cd, False, training, group=group)
            identity = self.bd.y
        self.b3 = _BN(blk.bn3, c3, True, training, residual=identity, group=group)
        self.out = self.b3.y
        self.out_hw = (H // s, W // s)

    def params(self):
        return [p for p in self.blk.parameters()]

    def backward(self, flat, dout):
        blk = self.blk
        N, H, W = self.geom
        s = blk.stride
        dc3, did = self.b3.backward(flat, dout)                    # did = gradient of the shortcut branch
        dp2 = _conv1x1_bwd(flat, blk.conv3, dc3, self.p2)
        dy2 = ops.avgpool_bwd(dp2, N, H, W, dp2.shape[1], s) if s > 1 else dp2
        dc2 = self.b2.backward(flat, dy2)
        dy1 = self.k2.backward(flat, dc2)
        dc1 = self.b1.backward(flat, dy1)
        if self.bd is not None:
            dcd = self.bd.backward(flat, did)
            dpx = _conv1x1_bwd(flat, blk.downsample[1], dcd, self.px)
            did = ops.avgpool_bwd(dpx, N, H, W, dpx.shape[1], s) if s > 1 else dpx
        return _conv1x1_bwd(flat, blk.conv1, dc1, self.x, residual=did)       # dx = conv1^T(dc1) + shortcut gradient


class ResNetTowerFn(torch.autograd.Function):
    """forward(anchor, images, tower, c0, want_dense, n_views) -> feat [V*b, E] fp32 (, dense [V*b, 49, C] act dtype).
    n_views channel-stacked views are encoded as SEPARATE passes (BatchNorm statistics are per call in the reference:
    the views of DeCLIP / SLIP go through the tower one after the other), outputs concatenated view-major."""

    @staticmethod
    def forward(ctx, anchor, images, tower, c0, want_dense, n_views=1, want_feature=False):
        flat = tower._flat()
        training = tower.training
        save = bool(ctx.needs_input_grad[0])
        flat.tower_forward(tower, save)
        passes, feats, denses, pooleds = [], [], [], []
        for v in range(n_views):
            st = _forward_pass(flat, tower, images, c0 + 3 * v, training)
            feats.append(st["out"])
            denses.append(st["dense"])
            pooleds.append(st["pooled"])
            passes.append(st if save else None)
        ctx.tower, ctx.passes, ctx.want_dense, ctx.want_feature = tower, passes, want_dense, want_feature
        outs = [feats[0] if n_views == 1 else torch.cat(feats, dim=0)]
        if want_dense:
            outs.append(denses[0] if n_views == 1 else torch.cat(denses, dim=0))
        if want_feature:
            # the pooled trunk feature in front of the output projection ([b, width*32]: the attention pool's mean-token output
            # before c_proj, or the adaptive average pool in front of fc) -- what SLIP's `feature_dim: 2048` head consumes
            outs.append(pooleds[0] if n_views == 1 else torch.cat(pooleds, dim=0))
        return tuple(outs) if len(outs) > 1 else outs[0]

    @staticmethod
    def backward(ctx, *grads):
        tower = ctx.tower
        flat = tower._flat()
        flat.begin_backward()
        flat.tower_backward(tower)
        dout = grads[0]
        gi = 1
        ddense = dfeat = None
        if ctx.want_dense:
            ddense = grads[gi]; gi += 1
        if ctx.want_feature:
            dfeat = grads[gi]; gi += 1
        V = len(ctx.passes)
        for v in reversed(range(V)):
            st = ctx.passes[v]
            b = st["b"]
            do = dout[v * b:(v + 1) * b] if dout is not None else None
            dd = ddense[v * b:(v + 1) * b] if ddense is not None else None
            df = dfeat[v * b:(v + 1) * b] if dfeat is not None else None
            _backward_pass(flat, tower, st, do, dd, last=(v == 0), dfeat=df)
        ctx.passes = None
        return (torch.zeros_like(flat.anchor), None, None, None, None, None, None)


def _stem_conv1_weight(flat, tower, dtype, device):
    w = _w2d(flat, tower.conv1)                                   # [C1, 27]
    wp = torch.zeros(w.shape[0], 32, device=device, dtype=dtype)
    wp[:, :27].copy_(w)
    return wp


def _forward_pass(flat, tower, images, c0, training):
    dtype = flat.act_dtype
    b, _, Hi, Wi = images.shape
    st = {"b": b}
    _TRACKED.clear()
    grp = bn_group(tower) if training else None
    # ---- stem (modified_resnet.py:144-150,194-199): conv1 stride 2 on the image, conv2, conv3, avgpool(2)
    rows0, H, W = ops.conv_rows_image(images, c0, dtype, stride=2, pad=1)
    st["rows0"] = rows0
    c = ops.gemm(rows0, _stem_conv1_weight(flat, tower, dtype, images.device), ws=_ws(rows0))
    s1 = _BN(tower.bn1, c, True, training, group=grp)
    k2 = _Conv3(tower.conv2, s1.y, b, H, W)
    s2 = _BN(tower.bn2, k2.forward(flat), True, training, group=grp)
    k3 = _Conv3(tower.conv3, s2.y, b, H, W)
    s3 = _BN(tower.bn3, k3.forward(flat), True, training, group=grp)
    x = ops.avgpool_fwd(s3.y, b, H, W, s3.y.shape[1], 2)
    st["stem"] = (s1, k2, s2, k3, s3, H, W)
    H, W = H // 2, W // 2
    # ---- four stages of bottlenecks
    blocks = []
    for layer in (tower.layer1, tower.layer2, tower.layer3, tower.layer4):
        for blk in layer:
            bk = _Block(flat, blk, x, b, H, W, training, group=grp)
            x, (H, W) = bk.out, bk.out_hw
            blocks.append(bk)
    st["blocks"] = blocks
    ap = tower.attnpool
    on_attnpool = W == 7                                          # modified_resnet.py:207: `if x.size(3) == 7`
    for p_ in ap.parameters():
        p_._dh_grad_none = not on_attnpool                        # the head that is off the path keeps grad None (torch semantics)
    for p_ in tower.fc.parameters():
        p_._dh_grad_none = on_attnpool
    if not on_attnpool:
        # ---- adaptive average pool + fc (modified_resnet.py:209-211), any input size whose final map is square
        C = x.shape[1]
        if H != W or tower.fc.weight.shape[1] != C:
            raise DeclipHipError("ModifiedResNet: the adaptive-pool + fc head needs a square final map and fc.in_features == %d "
                                 "(got %dx%d, fc.in_features %d)" % (C, H, W, tower.fc.weight.shape[1]))
        pooled = ops.avgpool_fwd(x, b, H, W, C, H)                # [b, C]: mean over the whole map
        out = ops.gemm(pooled, flat.wview(tower.fc.weight), bias=tower.fc.bias.data, out_dtype=torch.float32)
        st.update(x_last=x, pooled=pooled, out=out, dense=x.view(b, H * W, C), geom=(H * W, C, 0, 0), head="fc", hw=(H, W))
        _bump_tracked()
        return st
    if H != 7 or H * W + 1 != ap.positional_embedding.shape[0]:
        raise DeclipHipError("ModifiedResNet: attention pool built for %d tokens, final map is %dx%d"
                             % (ap.positional_embedding.shape[0] - 1, H, W))
    # ---- attention pool (modified_resnet.py:59-96): tokens, q/k/v projections, attention, c_proj of the mean token
    HW, C, heads = H * W, x.shape[1], ap.num_heads
    L = HW + 1
    tok = ops.attnpool_tokens_fwd(x, ap.positional_embedding.data, b, HW)
    w = flat.wview
    ws = _ws(tok)
    q = ops.gemm(tok, w(ap.q_proj.weight), bias=ap.q_proj.bias.data, ws=ws)
    k = ops.gemm(tok, w(ap.k_proj.weight), bias=ap.k_proj.bias.data, ws=ws)
    v = ops.gemm(tok, w(ap.v_proj.weight), bias=ap.v_proj.bias.data, ws=ws)
    qkv = torch.cat([q, k, v], dim=1)                             # [b*L, 3C] (q | k | v, head-major): layout of dh_attn_*
    a, lse = ops.attn_fwd(qkv, b, L, heads, False)
    pooled = ops.pool_rows_fwd(a, None, b, L)                     # token 0 = the mean token's output (x[0], :96)
    out = ops.gemm(pooled, w(ap.c_proj.weight), bias=ap.c_proj.bias.data, out_dtype=torch.float32)
    st.update(x_last=x, tok=tok, qkv=qkv, a=a, lse=lse, pooled=pooled, out=out, dense=x.view(b, HW, C), geom=(HW, C, heads, L),
              head="attnpool")
    _bump_tracked()
    return st


def _backward_pass(flat, tower, st, dout, ddense, last, dfeat=None):
    dtype = flat.act_dtype
    g = flat.gview
    b = st["b"]
    HW, C, heads, L = st["geom"]
    ap = tower.attnpool
    w = flat.wview
    x_last = st["x_last"]
    dfeat_a = _to_act(dfeat, dtype) if dfeat is not None else None      # gradient of the pooled feature handed out (return_feature)
    if st["head"] == "fc":
        if dout is not None or dfeat_a is not None:
            dpooled = dfeat_a
            if dout is not None:
                do = _to_act(dout, dtype)
                weight_grad(do, st["pooled"], g(tower.fc.weight), g(tower.fc.bias))
                dp = ops.gemm(do, w(tower.fc.weight), b_kmajor=True)
                dpooled = dp if dpooled is None else dp.add_(dpooled)
            H, W = st["hw"]
            dx = ops.avgpool_bwd(dpooled, b, H, W, C, H)
        else:
            dx = torch.zeros_like(x_last)
    elif dout is not None or dfeat_a is not None:
        dpooled = dfeat_a
        if dout is not None:
            do = _to_act(dout, dtype)
            weight_grad(do, st["pooled"], g(ap.c_proj.weight), g(ap.c_proj.bias))
            dp = ops.gemm(do, w(ap.c_proj.weight), b_kmajor=True)
            dpooled = dp if dpooled is None else dp.add_(dpooled)
        da = ops.pool_rows_bwd(dpooled, None, b, L)
        dqkv = ops.attn_bwd(st["qkv"], st["a"], da, st["lse"], b, L, heads, False)
        dq, dk, dv = (t.contiguous() for t in dqkv.split(C, dim=1))
        tok = st["tok"]
        weight_grad(dq, tok, g(ap.q_proj.weight), g(ap.q_proj.bias))
        weight_grad(dk, tok, g(ap.k_proj.weight), g(ap.k_proj.bias))
        weight_grad(dv, tok, g(ap.v_proj.weight), g(ap.v_proj.bias))
        ws = _ws(dq)
        dtok = ops.gemm(dq, w(ap.q_proj.weight), b_kmajor=True, ws=ws)
        dtok = ops.gemm(dk, w(ap.k_proj.weight), b_kmajor=True, residual=dtok, ws=ws)
        dtok = ops.gemm(dv, w(ap.v_proj.weight), b_kmajor=True, residual=dtok, ws=ws)
        dx = ops.attnpool_tokens_bwd(dtok, g(ap.positional_embedding), b, HW)
    else:
        dx = torch.zeros_like(x_last)
    if ddense is not None:
        dx.add_(ddense.reshape(b * HW, C).to(dtype))
    if last:
        flat.grads_ready(list(ap.parameters()) if st["head"] == "attnpool" else list(tower.fc.parameters()))
    for bk in reversed(st["blocks"]):
        dx = bk.backward(flat, dx)
        if last:
            flat.grads_ready(bk.params())
    # ---- stem
    s1, k2, s2, k3, s3, H, W = st["stem"]
    dy3 = ops.avgpool_bwd(dx, b, H, W, dx.shape[1], 2)
    dc3 = s3.backward(flat, dy3)
    dy2 = k3.backward(flat, dc3)
    dc2 = s2.backward(flat, dy2)
    dy1 = k2.backward(flat, dc2)
    dc1 = s1.backward(flat, dy1)
    if tower.conv1.weight.requires_grad:
        C1 = dc1.shape[1]
        gw = torch.zeros(C1, 32, device=dc1.device, dtype=torch.float32)
        weight_grad(dc1, st["rows0"], gw)
        g(tower.conv1.weight).view(C1, 27).add_(gw[:, :27])       # the K padding of the stem patches carries no gradient
    if last:
        flat.grads_ready([tower.conv1.weight, tower.bn1.weight, tower.bn1.bias, tower.conv2.weight, tower.bn2.weight, tower.bn2.bias,
                          tower.conv3.weight, tower.bn3.weight, tower.bn3.bias])


def check_supported(tower):
    """constructor-time checks: everything the kernels assume (C multiple of 8, head dim <= 64)."""
    widths = {tower.conv1.weight.shape[0], tower.conv3.weight.shape[0]}
    if any(wd % 8 for wd in widths):
        raise DeclipHipError("ModifiedResNet width must make every channel count a multiple of 8 (width %% 16 == 0)")
    C = tower.attnpool.positional_embedding.shape[1]
    if C % tower.attnpool.num_heads or C // tower.attnpool.num_heads > 64:
        raise DeclipHipError("attention pool head dim must be <= 64 (embed %d, heads %d)" % (C, tower.attnpool.num_heads))
