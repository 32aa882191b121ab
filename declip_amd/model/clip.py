"""CLIP two-tower model on the HIP engine (reference: model/clip.py:53-165).

Same constructor / attribute / state_dict surface as the reference (`visual`, `encode_text`,
`logit_scale [1]`, `encode_image`, `text_parameters()` ...).  forward() returns
(logits_per_image, logits_per_text); by default these are `LazyLogits` handles -- the [b,B]
matrices are never materialised: `ClipInfoCELoss` / `accuracy` consume the handles through
the fused InfoNCE kernel.  `.materialize()` (or engine kwarg fused_loss=False) yields real
tensors for code that wants them.
"""
import contextlib
import os

import numpy as np
import torch
from torch import nn

from .. import dist as dh_dist
from .. import engine
from .resnet import modified_resnet_R50
from .transformer import text_transformers, visual_transformer_B16, visual_transformer_B32

__all__ = ["clip_vitb32", "clip_vitb16", "clip_res50", "CLIP", "LazyLogits"]


class LazyLogits:
    """scale * Q @ K^T as a handle (model/clip.py:140-141).  label of local row i = label0 + i
    (loss_functions/loss.py:38-42)."""

    def __init__(self, Q, K, scale, label0):
        self.Q, self.K, self.scale, self.label0 = Q, K, scale, label0
        self._dense = None

    @property
    def shape(self):
        return torch.Size((self.Q.shape[0], self.K.shape[0]))

    def size(self, dim=None):
        return self.shape if dim is None else self.shape[dim]

    def __len__(self):
        return self.Q.shape[0]

    @property
    def device(self):
        return self.Q.device

    def materialize(self):
        if self._dense is None:
            self._dense = engine.LogitsFn.apply(self.scale, self.Q, self.K)
        return self._dense


class CLIP(nn.Module):
    def __init__(self, image_encode, text_encode, use_allgather, dtype="bf16", fused_loss=True, scale_clamp=100.0):
        super().__init__()
        self.use_allgather = use_allgather
        self.visual = image_encode
        self.encode_text = text_encode
        self.logit_scale = nn.Parameter(torch.ones([1]))
        nn.init.constant_(self.logit_scale, np.log(1 / 0.07))
        self.fused_loss = fused_loss
        self.scale_clamp = scale_clamp          # clip.py:134 (None for SLIP/FILIP which do not clamp)
        act = torch.bfloat16 if str(dtype) in ("bf16", "bfloat16", "torch.bfloat16") else torch.float32
        self.__dict__["_flat_store"] = engine.FlatParams(self, act)
        self._adopt_towers()

    def _adopt_towers(self):
        for m in self.modules():
            if hasattr(m, "_flat") and m is not self:
                m.__dict__["_engine_root"] = self

    # ---- reference surface (clip.py:63-116) -------------------------------------------------
    def text_parameters(self):
        return [self.logit_scale, self.encode_text.positional_embedding]

    def text_modules(self):
        et = self.encode_text
        return [et.transformer, et.text_projection, et.token_embedding, et.ln_final]

    def visual_parameters(self):
        return []

    def visual_modules(self):
        return [self.visual]

    @property
    def dtype(self):
        return self.visual.conv1.weight.dtype

    def encode_image(self, image, **kw):
        self._flat_store.begin_step()
        return self.visual(image, **kw)

    def sample_captions(self, texts):
        if torch.is_tensor(texts):
            return texts[:, 0] if texts.dim() == 3 else texts
        return [t[0] for t in texts]

    def all_gather(self, x):
        return dh_dist.all_gather_cat(x)

    def logit_scale_value(self):
        """clip.py:133-134: exp() then a `.data` clamp (forward clamped, autograd sees the raw exp)."""
        s = self.logit_scale.exp()
        if self.scale_clamp is not None:
            s = s + (s.clamp(max=self.scale_clamp) - s).detach()
        return s

    def features(self, images, texts):
        """normalised (image, text) features, fp32 [b,E] (clip.py:123-130)."""
        # the towers are independent until the loss: one of them goes to a side stream.  autograd replays each tower's
        # backward on its forward stream, so the two backward passes overlap the same way.  The longer (image) tower is
        # enqueued first, on the side stream; the text tower follows on the caller's stream (+0.5 % over the other arrangements)
        side = self._fork(images)
        with self._on(side):
            img = engine.L2NormFn.apply(self.visual(images), 0.0)
        txt = engine.L2NormFn.apply(self.encode_text(texts), 1e-10)
        self._join(side, img)
        return img, txt

    # ---- two tower streams (FlatParams.side_stream): fork before the first tower, join after the second one is enqueued
    def _fork(self, images):
        if not self._two_streams(images):
            return None
        side = self._flat_store.ensure().side_stream(0)
        side.wait_stream(torch.cuda.current_stream(images.device))      # inputs + the refreshed bf16 mirror
        return side

    @staticmethod
    def _on(side):
        return torch.cuda.stream(side) if side is not None else contextlib.nullcontext()

    @staticmethod
    def _join(side, *outs):
        if side is None:
            return
        main = torch.cuda.current_stream(side.device)
        main.wait_stream(side)
        for t in outs:
            if torch.is_tensor(t):
                t.record_stream(main)                                   # allocated on the side stream, consumed on this one

    def _two_streams(self, images):
        """DH_TOWER_STREAMS: 1 (default) = image and text tower on two HIP streams, 0 = one stream.  Under torch.distributed
        the bucketed gradient all-reduce orders every bucket against both streams (dist.FlatReducer; tests/test_gpu_dist.py)."""
        return images.is_cuda and os.environ.get("DH_TOWER_STREAMS", "1") == "1"

    def forward(self, input, all_gather=False):
        self._flat_store.begin_step()
        images = input["images"]
        texts = self.sample_captions(input["captions"])
        # exp / clamp of the temperature BEFORE the towers: four one-element kernels that then run beside the towers instead of between
        # the join of the two tower streams and the loss kernel (same values; their backward nodes move behind the text tower's)
        scale = self.logit_scale_value()
        img, txt = self.features(images, texts)
        if self.training and self.use_allgather or all_gather:
            g_img, g_txt = dh_dist.all_gather_cat_many([img, txt])
            label0 = dh_dist.get_rank() * img.shape[0]
        else:
            g_img, g_txt, label0 = img, txt, 0
        li = LazyLogits(img, g_txt, scale, label0)
        lt = LazyLogits(txt, g_img, scale, label0)
        if not self.fused_loss:
            return li.materialize(), lt.materialize()
        return li, lt


def _engine_kwargs(kwargs):
    e = dict(kwargs.get("engine", {}) or {})
    return dict(dtype=e.get("dtype", "bf16"), fused_loss=e.get("fused_loss", True))


def clip_vitb32(**kwargs):
    """model/clip.py:158-165; extra optional kwargs block `engine: {dtype: bf16|fp32, fused_loss: bool}`."""
    image_encode = visual_transformer_B32(**kwargs["image_encode"])
    text_encode = text_transformers(**kwargs["text_encode"])
    return CLIP(image_encode, text_encode, **kwargs["clip"], **_engine_kwargs(kwargs))


def clip_res50(**kwargs):
    """model/clip.py:148-155 (BASELINE.json configs[0]; image_encode: {embed_dim, use_sync_bn: False, bn_*}, as in
    experiments/clip_experiments/yfcc15m/yfcc15m_r50_clip/config.yaml:2-19)."""
    image_encode = modified_resnet_R50(**kwargs["image_encode"])
    text_encode = text_transformers(**kwargs["text_encode"])
    return CLIP(image_encode, text_encode, **kwargs["clip"], **_engine_kwargs(kwargs))


def clip_vitb16(**kwargs):
    image_encode = visual_transformer_B16(**kwargs["image_encode"])
    text_encode = text_transformers(**kwargs["text_encode"])
    return CLIP(image_encode, text_encode, **kwargs["clip"], **_engine_kwargs(kwargs))
