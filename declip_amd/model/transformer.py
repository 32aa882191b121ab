"""Parameter containers for the two towers + their engine entry points.

The modules below hold parameters under EXACTLY the reference's state_dict names
(SURVEY.md s8(a) "Parameter/state_dict layout"), so model-zoo checkpoints load and
`param_group_all`-style isinstance grouping (nn.LayerNorm / nn.Linear biases) keeps working.
None of the nn containers' own forward() is ever used: all arithmetic runs in the HIP
engine (declip_amd.engine).  Reference: model/image_encoder/visual_transformer.py,
model/image_encoder/base_transformer.py, model/text_encoder/text_transformer.py.
"""
import os
from collections import OrderedDict

import torch
from torch import nn

from .. import engine
from ..lib import DeclipHipError


class LayerNorm(nn.LayerNorm):
    """base_transformer.py:10-18 (container; eps 1e-5)."""


class QuickGELU(nn.Module):
    """base_transformer.py:24-26 marker module (fused into the c_fc GEMM epilogue)."""

    def forward(self, x):
        raise DeclipHipError("QuickGELU is fused into the HIP GEMM epilogue; the container is never called")


class ResidualAttentionBlock(nn.Module):
    """base_transformer.py:29-53 (parameters only)."""

    def __init__(self, d_model, n_head, attn_mask=None, dropout=0.0):
        super().__init__()
        if dropout:
            raise DeclipHipError("dropout > 0 is not supported by the HIP engine (all shipped configs use 0)")
        self.attn = nn.MultiheadAttention(d_model, n_head)
        self.ln_1 = LayerNorm(d_model)
        self.mlp = nn.Sequential(OrderedDict([("c_fc", nn.Linear(d_model, d_model * 4)), ("gelu", QuickGELU()),
                                              ("c_proj", nn.Linear(d_model * 4, d_model))]))
        self.ln_2 = LayerNorm(d_model)
        self.causal = attn_mask is not None


class Transformer(nn.Module):
    """base_transformer.py:56-79."""

    def __init__(self, width, layers, heads, attn_mask=None, checkpoint=False, dropout=0.0, emb_dropout=0.0):
        super().__init__()
        if checkpoint:
            # base_transformer.py:66-79 recomputes the blocks in the backward pass (torch.utils.checkpoint) to save activation memory.
            # This engine keeps every block's activations (15 GB per step at b = 512 of the 288 GB) and has no
            # recompute path: refuse the option instead of silently ignoring it (no shipped config sets it).
            raise NotImplementedError("Transformer(checkpoint=True): activation recomputation is not implemented by the MI355X engine "
                                      "(activations fit in HBM at the reference's batch sizes); build the model with checkpoint=False")
        self.width, self.layers, self.heads, self.checkpoint = width, layers, heads, False
        self.resblocks = nn.Sequential(*[ResidualAttentionBlock(width, heads, attn_mask, dropout) for _ in range(layers)])


def _init_blocks(tr):
    """Same distributions as the reference (visual_transformer.py:29-38, text_transformer.py:113-127)."""
    proj_std = (tr.width ** -0.5) * ((2 * tr.layers) ** -0.5)
    attn_std = tr.width ** -0.5
    fc_std = (2 * tr.width) ** -0.5
    for blk in tr.resblocks:
        nn.init.normal_(blk.attn.in_proj_weight, std=attn_std)
        nn.init.normal_(blk.attn.out_proj.weight, std=proj_std)
        nn.init.normal_(blk.mlp.c_fc.weight, std=fc_std)
        nn.init.normal_(blk.mlp.c_proj.weight, std=proj_std)


class _Tower(nn.Module):
    def _flat(self):
        root = self.__dict__.get("_engine_root")
        if root is None:
            # standalone tower: own store
            st = self.__dict__.get("_own_flat")
            if st is None:
                st = engine.FlatParams(self, self.__dict__.get("_act_dtype", torch.bfloat16))
                self.__dict__["_own_flat"] = st
            st = st.ensure()
        else:
            st = root._flat_store.ensure()
        st.refresh_mirror()          # a tower called directly (encode_text / encode_image) must never see an unwritten bf16 mirror
        return st


class VisualTransformer(_Tower):
    """image_encoder/visual_transformer.py:6-82."""

    def __init__(self, input_resolution, patch_size, width, layers, heads, embed_dim, checkpoint=False, dropout=0,
                 emb_dropout=0):
        super().__init__()
        self.input_resolution, self.patch_size, self.width, self.heads = input_resolution, patch_size, width, heads
        self.output_dim = embed_dim
        self.freeze_conv1 = True                                   # visual_transformer.py:12
        self.conv1 = nn.Conv2d(3, width, kernel_size=patch_size, stride=patch_size, bias=False)
        scale = width ** -0.5
        self.class_embedding = nn.Parameter(scale * torch.randn(width))
        self.positional_embedding = nn.Parameter(scale * torch.randn((input_resolution // patch_size) ** 2 + 1, width))
        self.ln_pre = LayerNorm(width)
        self.transformer = Transformer(width, layers, heads, checkpoint=checkpoint, dropout=dropout, emb_dropout=emb_dropout)
        self.ln_post = LayerNorm(width)
        self.proj = nn.Parameter(scale * torch.randn(width, embed_dim))
        nn.init.normal_(self.positional_embedding, std=0.01)
        _init_blocks(self.transformer)

    def train(self, mode=True):
        super().train(mode)
        if self.freeze_conv1:                                      # quirk 6: conv1 frozen + eval on every .train()
            self.conv1.eval()
            for p in self.conv1.parameters():
                p.requires_grad = False
        return self

    def forward(self, x, return_dense=False, return_feature=False, channel_offset=0, n_views=1):
        """x: [b, 3*views, H, W] fp32 on the GPU; channel_offset selects a channel-stacked view;
        n_views > 1 encodes that many consecutive views in one pass (outputs [views*b, ...], view-major)."""
        flat = self._flat()
        if x.dtype == torch.uint8:           # decoded images as bytes [b, H, W, 3]: normalised on the GPU (4x less PCIe / HBM input)
            x = engine.ops.image_prep_u8(x.contiguous(), (self.input_resolution, self.input_resolution))
        if x.dtype != torch.float32:
            x = x.float()
        return engine.VisionTowerFn.apply(flat.anchor, x.contiguous(), self, channel_offset, return_dense, return_feature, n_views)


class TextTransformer(_Tower):
    """text_encoder/text_transformer.py:10-204 ('Transformer' branch; HF branches are out of scope,
    SURVEY.md s2 row 9)."""

    def __init__(self, embed_dim, context_length, transformer_width, transformer_heads, transformer_layers,
                 positional_embedding_flag, checkpoint, bpe_path=None, text_encode_type=None, text_model_utils=None,
                 vocab_size=49409):
        super().__init__()
        if text_encode_type != "Transformer":
            raise NotImplementedError(str(text_encode_type))
        self.context_length = context_length
        self.positional_embedding_flag = positional_embedding_flag
        self.text_encode_type = text_encode_type
        self.text_model_utils = text_model_utils or {}
        self.width, self.heads = transformer_width, transformer_heads
        self.tokenizer = None
        self._bpe_path = bpe_path
        self.vocab_size = vocab_size
        self.transformer = Transformer(transformer_width, transformer_layers, transformer_heads, attn_mask=True, checkpoint=checkpoint)
        self.token_embedding = nn.Embedding(self.vocab_size, transformer_width)
        self.positional_embedding = nn.Parameter(torch.empty(context_length, transformer_width))
        self.ln_final = LayerNorm(transformer_width)
        self.text_projection = nn.Linear(transformer_width, embed_dim)
        nn.init.normal_(self.token_embedding.weight, std=0.02)
        nn.init.normal_(self.positional_embedding, std=0.01)
        _init_blocks(self.transformer)
        nn.init.normal_(self.text_projection.weight, std=transformer_width ** -0.5)
        if not positional_embedding_flag:
            raise DeclipHipError("positional_embedding_flag=False is not supported")

    @property
    def dtype(self):
        return self.positional_embedding.dtype

    def _get_tokenizer(self):
        if self.tokenizer is None:
            from ..bpe import NativeTokenizer
            if not self._bpe_path or not os.path.exists(self._bpe_path):
                raise DeclipHipError("captions were given as strings but bpe_path %r does not exist; pass pre-tokenised "
                                     "LongTensor ids [b,%d] instead" % (self._bpe_path, self.context_length))
            self.tokenizer = NativeTokenizer(self._bpe_path)       # host-thread BPE of the C-ABI library (bpe_host.hip)
            assert len(self.tokenizer.encoder) == self.vocab_size
        return self.tokenizer

    def tokenize(self, texts, context_length=77, mask_type=None):
        """text_transformer.py:144-180 semantics (SOT, BPE ids, EOT, zero pad; over-long keeps the last token)."""
        from ..bpe import tokenize
        return tokenize(self._get_tokenizer(), texts, context_length, mask_type)

    def forward(self, text, mask_type=None, return_dense=False):
        labels = None
        if torch.is_tensor(text):
            ids = text
            if mask_type is not None:
                from ..bpe import mask_token_ids
                ids, labels = mask_token_ids(ids, self.vocab_size)
        else:
            tok = self.tokenize(text, self.context_length, mask_type)
            ids, labels = tok if mask_type is not None else (tok, None)
        flat = self._flat()
        dev = flat.flat_p.device
        want_dense = return_dense or mask_type is not None
        packed = not want_dense and engine.text_packed_mode() != 0
        host_rows = None
        if packed and not ids.is_cuda and getattr(ids, "_dh_rows", None) is None:
            host_rows = int((ids.argmax(dim=-1) + 1).sum())         # counted on the host copy: no device read in the step
        ids = engine.to_device_async(ids, dev).long().contiguous()
        if packed:
            # variable-length captions: only the rows up to <|endoftext|> are computed (engine.PackedCaptions; same outputs)
            if host_rows is not None:
                ids._dh_rows = (ids._version, host_rows)
            return engine.TextTowerPackedFn.apply(flat.anchor, ids, self)
        out = engine.TextTowerFn.apply(flat.anchor, ids, self, want_dense)
        if mask_type is not None:
            return out[0], out[1], engine.to_device_async(labels, dev)
        if return_dense:
            return out
        return out


def visual_transformer_B32(**kwargs):
    """visual_transformer.py:88-104."""
    cfg = dict(layers=12, heads=12, input_resolution=224, patch_size=32, width=768, checkpoint=False)
    cfg.update(kwargs)
    return VisualTransformer(**cfg)


def visual_transformer_B16(**kwargs):
    """visual_transformer.py:106-122."""
    cfg = dict(layers=12, heads=12, input_resolution=224, patch_size=16, width=768, checkpoint=False)
    cfg.update(kwargs)
    return VisualTransformer(**cfg)


def text_transformers(**kwargs):
    """text_transformer.py:276-288."""
    cfg = dict(context_length=77, transformer_width=512, transformer_heads=8, transformer_layers=12,
               positional_embedding_flag=True, checkpoint=False)
    cfg.update(kwargs)
    return TextTransformer(**cfg)
