"""ModifiedResNet parameter containers + engine entry point (reference: model/image_encoder/modified_resnet.py).

Same constructor arguments, module tree and state_dict names as the reference (conv1/bn1/conv2/bn2/conv3/bn3,
layer{1..4}.{i}.{conv1,bn1,conv2,bn2,conv3,bn3,downsample.{0,1}}, attnpool.{positional_embedding,k_proj,q_proj,v_proj,c_proj},
fc), so model-zoo R50 checkpoints load and the solver's isinstance-based parameter groups (bn_w / bn_b, utils/misc.py:267-412)
keep working.  None of the containers' own forward() is used: the arithmetic runs in declip_amd.resnet_engine on the HIP
kernels.  BatchNorm is per-rank (`use_sync_bn: False`, what the shipped R50 config selects,
experiments/clip_experiments/yfcc15m/yfcc15m_r50_clip/config.yaml:9) or synchronised over rank groups (`use_sync_bn: True` +
`bn_group_size`, modified_resnet.py:118-140).
"""
from collections import OrderedDict

import torch
from torch import nn

from .. import engine, resnet_engine
from ..lib import DeclipHipError
from .transformer import _Tower


def _conv(cin, cout, k, stride=1):
    return nn.Conv2d(cin, cout, kernel_size=k, stride=stride, padding=k // 2, bias=False)


def _conv_bn_chain(module, specs):
    """registers conv<i> / bn<i> pairs (i = 1, 2, ...) on `module` from (cin, cout, kernel, stride) tuples"""
    for i, (cin, cout, k, stride) in enumerate(specs, 1):
        setattr(module, "conv%d" % i, _conv(cin, cout, k, stride))
        setattr(module, "bn%d" % i, nn.BatchNorm2d(cout))


class Bottleneck(nn.Module):
    """modified_resnet.py:14-56 -- parameters and geometry only: 1x1 -> 3x3 -> 1x1 (x4 channels), every convolution stride 1,
    the block's stride taken by an average pool after the 3x3; `downsample` (avgpool + 1x1 conv + BN under the reference's
    Sequential keys "-1", "0", "1") whenever the shape changes."""
    expansion = 4

    def __init__(self, inplanes, planes, stride=1):
        super().__init__()
        wide = planes * self.expansion
        _conv_bn_chain(self, [(inplanes, planes, 1, 1), (planes, planes, 3, 1), (planes, wide, 1, 1)])
        self.stride = stride
        self.avgpool = nn.AvgPool2d(stride) if stride > 1 else nn.Identity()
        self.relu = nn.ReLU(inplace=True)
        self.downsample = None
        if stride > 1 or inplanes != wide:
            self.downsample = nn.Sequential(OrderedDict((("-1", nn.AvgPool2d(stride)), ("0", _conv(inplanes, wide, 1)),
                                                         ("1", nn.BatchNorm2d(wide)))))

    def forward(self, x):
        raise DeclipHipError("Bottleneck runs inside the HIP engine (resnet_engine); the container is never called")


class AttentionPool2d(nn.Module):
    """modified_resnet.py:59-96 (parameters: positional embedding for HW + 1 tokens, separate k / q / v projections, c_proj)."""

    def __init__(self, spacial_dim, embed_dim, num_heads, output_dim=None):
        super().__init__()
        self.num_heads = num_heads
        self.positional_embedding = nn.Parameter(torch.randn(spacial_dim ** 2 + 1, embed_dim) / embed_dim ** 0.5)
        for name, out_features in (("k_proj", embed_dim), ("q_proj", embed_dim), ("v_proj", embed_dim), ("c_proj", output_dim or embed_dim)):
            setattr(self, name, nn.Linear(embed_dim, out_features))

    def forward(self, x):
        raise DeclipHipError("AttentionPool2d runs inside the HIP engine (resnet_engine); the container is never called")


class ModifiedResNet(_Tower):
    """modified_resnet.py:106-214: 3-convolution stem (the first with stride 2) + avgpool, four bottleneck stages of widths
    width * (1, 2, 4, 8), attention pool on the final map (adaptive pool + fc when it is not 7 wide)."""

    def __init__(self, layers, embed_dim, heads, input_resolution=224, width=64, bn_group_size=1, bn_var_mode=None,
                 bn_sync_stats=False, use_sync_bn=True):
        super().__init__()
        # use_sync_bn: True synchronises the batch statistics over groups of `bn_group_size` consecutive ranks
        # (resnet_engine.bn_group; torch.nn.SyncBatchNorm semantics -- the reference's own branch, linklink's SyncBatchNorm2d with
        # `bn_var_mode` / `bn_sync_stats`, cannot be constructed in the released tree, SURVEY.md s9 quirk 19, so those two
        # arguments are accepted and ignored).  The parameter containers stay nn.BatchNorm2d: same state_dict keys.
        self.use_sync_bn, self.bn_group_size = bool(use_sync_bn), int(bn_group_size)
        self.output_dim, self.input_resolution = embed_dim, input_resolution
        half = width // 2
        _conv_bn_chain(self, [(3, half, 3, 2), (half, half, 3, 1), (half, width, 3, 1)])
        self.avgpool, self.relu = nn.AvgPool2d(2), nn.ReLU(inplace=True)
        self._inplanes = width
        for i, blocks in enumerate(layers):
            setattr(self, "layer%d" % (i + 1), self._make_layer(width << i, blocks, stride=1 if i == 0 else 2))
        self.attnpool = AttentionPool2d(input_resolution // 32, width * 32, heads, embed_dim)
        self.adaptivepool = nn.AdaptiveAvgPool2d((1, 1))
        self.fc = nn.Linear(2048, embed_dim)                    # modified_resnet.py:167 (the head when the final map is not 7 wide)
        for p in self.fc.parameters():
            p._dh_grad_none = True      # the head off the path keeps grad None and is left alone by the optimizer, as in torch
                                        # (resnet_engine sets this per forward: attention pool at 224 px, adaptive pool + fc otherwise)
        # initial distributions of the reference (modified_resnet.py:169-178)
        ap = self.attnpool
        for lin in (ap.q_proj, ap.k_proj, ap.v_proj, ap.c_proj):
            nn.init.normal_(lin.weight, std=ap.c_proj.in_features ** -0.5)
        for stage in (self.layer1, self.layer2, self.layer3, self.layer4):
            for blk in stage:
                nn.init.zeros_(blk.bn3.weight)
        resnet_engine.check_supported(self)

    def _make_layer(self, planes, blocks, stride=1):
        chain = []
        for i in range(blocks):                                  # only the first block of a stage strides / widens
            chain.append(Bottleneck(self._inplanes, planes, stride if i == 0 else 1))
            self._inplanes = planes * Bottleneck.expansion
        return nn.Sequential(*chain)

    def forward(self, x, return_dense=False, channel_offset=0, n_views=1, return_feature=False):
        """x: [b, 3*views, 224, 224] fp32 (or uint8 [b, H, W, 3]) on the GPU -> [b, embed_dim] fp32
        (, dense [b, 49, width*32]: modified_resnet.py:206)(, feature [b, width*32]).
        `return_feature` does not exist in the reference's ModifiedResNet.forward (modified_resnet.py:183): its SLIP model calls
        `self.visual(image, return_feature=True)` (slip.py:228-231), so the shipped `slip_res50` config raises a TypeError in
        the reference.  Here the keyword returns the pooled trunk feature in front of the output projection (2048 wide for
        ResNet-50 = the `feature_dim: 2048` that config hands to SLIP's projection MLP), which makes slip_res50 runnable."""
        flat = self._flat()
        if x.dtype == torch.uint8:
            x = engine.ops.image_prep_u8(x.contiguous(), (self.input_resolution, self.input_resolution))
        if x.dtype != torch.float32:
            x = x.float()
        return resnet_engine.ResNetTowerFn.apply(flat.anchor, x.contiguous(), self, channel_offset, return_dense, n_views, return_feature)


def modified_resnet_R50(**kwargs):
    """modified_resnet.py:216-228."""
    cfg = dict(layers=(3, 4, 6, 3), heads=64 * 32 // 64, input_resolution=224, width=64)
    cfg.update(kwargs)
    return ModifiedResNet(**cfg)


def modified_resnet_R101(**kwargs):
    """modified_resnet.py:230-242."""
    cfg = dict(layers=(3, 4, 23, 3), heads=64 * 32 // 64, input_resolution=224, width=64)
    cfg.update(kwargs)
    return ModifiedResNet(**cfg)
