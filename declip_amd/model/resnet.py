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


class Bottleneck(nn.Module):
    """modified_resnet.py:14-56 (parameters + geometry only)."""
    expansion = 4

    def __init__(self, inplanes, planes, stride=1):
        super().__init__()
        self.conv1 = nn.Conv2d(inplanes, planes, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(planes)
        self.conv2 = nn.Conv2d(planes, planes, 3, padding=1, bias=False)
        self.bn2 = nn.BatchNorm2d(planes)
        self.avgpool = nn.AvgPool2d(stride) if stride > 1 else nn.Identity()
        self.conv3 = nn.Conv2d(planes, planes * self.expansion, 1, bias=False)
        self.bn3 = nn.BatchNorm2d(planes * self.expansion)
        self.relu = nn.ReLU(inplace=True)
        self.downsample = None
        self.stride = stride
        if stride > 1 or inplanes != planes * Bottleneck.expansion:
            self.downsample = nn.Sequential(OrderedDict([
                ("-1", nn.AvgPool2d(stride)),
                ("0", nn.Conv2d(inplanes, planes * self.expansion, 1, stride=1, bias=False)),
                ("1", nn.BatchNorm2d(planes * self.expansion))]))

    def forward(self, x):
        raise DeclipHipError("Bottleneck runs inside the HIP engine (resnet_engine); the container is never called")


class AttentionPool2d(nn.Module):
    """modified_resnet.py:59-96."""

    def __init__(self, spacial_dim, embed_dim, num_heads, output_dim=None):
        super().__init__()
        self.positional_embedding = nn.Parameter(torch.randn(spacial_dim ** 2 + 1, embed_dim) / embed_dim ** 0.5)
        self.k_proj = nn.Linear(embed_dim, embed_dim)
        self.q_proj = nn.Linear(embed_dim, embed_dim)
        self.v_proj = nn.Linear(embed_dim, embed_dim)
        self.c_proj = nn.Linear(embed_dim, output_dim or embed_dim)
        self.num_heads = num_heads

    def forward(self, x):
        raise DeclipHipError("AttentionPool2d runs inside the HIP engine (resnet_engine); the container is never called")


class ModifiedResNet(_Tower):
    """modified_resnet.py:106-214."""

    def __init__(self, layers, embed_dim, heads, input_resolution=224, width=64, bn_group_size=1, bn_var_mode=None,
                 bn_sync_stats=False, use_sync_bn=True):
        super().__init__()
        # use_sync_bn: True synchronises the batch statistics over groups of `bn_group_size` consecutive ranks
        # (resnet_engine.bn_group; torch.nn.SyncBatchNorm semantics -- the reference's own branch, linklink's SyncBatchNorm2d with
        # `bn_var_mode` / `bn_sync_stats`, cannot be constructed in the released tree, SURVEY.md s9 quirk 19, so those two
        # arguments are accepted and ignored).  The parameter containers stay nn.BatchNorm2d: same state_dict keys.
        self.use_sync_bn, self.bn_group_size = bool(use_sync_bn), int(bn_group_size)
        self.output_dim = embed_dim
        self.input_resolution = input_resolution
        self.conv1 = nn.Conv2d(3, width // 2, kernel_size=3, stride=2, padding=1, bias=False)
        self.bn1 = nn.BatchNorm2d(width // 2)
        self.conv2 = nn.Conv2d(width // 2, width // 2, kernel_size=3, padding=1, bias=False)
        self.bn2 = nn.BatchNorm2d(width // 2)
        self.conv3 = nn.Conv2d(width // 2, width, kernel_size=3, padding=1, bias=False)
        self.bn3 = nn.BatchNorm2d(width)
        self.avgpool = nn.AvgPool2d(2)
        self.relu = nn.ReLU(inplace=True)
        self._inplanes = width
        self.layer1 = self._make_layer(width, layers[0])
        self.layer2 = self._make_layer(width * 2, layers[1], stride=2)
        self.layer3 = self._make_layer(width * 4, layers[2], stride=2)
        self.layer4 = self._make_layer(width * 8, layers[3], stride=2)
        feat_dim = width * 32
        self.attnpool = AttentionPool2d(input_resolution // 32, feat_dim, heads, embed_dim)
        self.adaptivepool = nn.AdaptiveAvgPool2d((1, 1))
        self.fc = nn.Linear(2048, embed_dim)                    # modified_resnet.py:167 (the head when the final map is not 7 wide)
        for p in self.fc.parameters():
            p._dh_grad_none = True      # the head off the path keeps grad None and is left alone by the optimizer, as in torch
                                        # (resnet_engine sets this per forward: attention pool at 224 px, adaptive pool + fc otherwise)
        std = self.attnpool.c_proj.in_features ** -0.5
        for lin in (self.attnpool.q_proj, self.attnpool.k_proj, self.attnpool.v_proj, self.attnpool.c_proj):
            nn.init.normal_(lin.weight, std=std)
        for stage in (self.layer1, self.layer2, self.layer3, self.layer4):
            for name, param in stage.named_parameters():
                if name.endswith("bn3.weight"):
                    nn.init.zeros_(param)
        resnet_engine.check_supported(self)

    def _make_layer(self, planes, blocks, stride=1):
        layers = [Bottleneck(self._inplanes, planes, stride)]
        self._inplanes = planes * Bottleneck.expansion
        for _ in range(1, blocks):
            layers.append(Bottleneck(self._inplanes, planes))
        return nn.Sequential(*layers)

    def forward(self, x, return_dense=False, channel_offset=0, n_views=1):
        """x: [b, 3*views, 224, 224] fp32 (or uint8 [b, H, W, 3]) on the GPU -> [b, embed_dim] fp32
        (, dense [b, 49, width*32]: modified_resnet.py:206)."""
        flat = self._flat()
        if x.dtype == torch.uint8:
            x = engine.ops.image_prep_u8(x.contiguous(), (self.input_resolution, self.input_resolution))
        if x.dtype != torch.float32:
            x = x.float()
        return resnet_engine.ResNetTowerFn.apply(flat.anchor, x.contiguous(), self, channel_offset, return_dense, n_views)


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
