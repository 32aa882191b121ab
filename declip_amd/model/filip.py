"""FILIP on the HIP engine (reference: model/filip.py:27-163): CLIP + token-wise max-similarity logits between the
49 image tokens and the 77 text tokens of every pair, with per-sample top-16 token selection.

Reference quirks kept: only view 1 is encoded (filip.py:112); the caption is MLM-masked (config text_mask_type: MLM)
yet no MLM loss is produced; select_topk must be True; top_k = 16; neither logit scale is clamped in the model;
`logit_scale_dense` is a 0-dim parameter.  The [b,B,49,16] / [b,B,77,16] tensors of the reference exist here only as
the token-similarity GEMM output that feeds dh_maxsim_reduce."""
import numpy as np
import torch
from torch import nn

from .. import dist as dh_dist
from .. import engine
from ..heads import LinearFn
from .clip import CLIP, LazyLogits, _engine_kwargs
from .transformer import text_transformers, visual_transformer_B32

__all__ = ["FILIP", "filip_vitb32", "filip_res50"]


class FILIP(CLIP):
    def __init__(self, image_encode, text_encode, use_allgather, nn_size=2 ** 16, nn_topk=1, return_dense=False,
                 return_caption=False, return_nn_bank=False, text_mask_type=None, EDA=True, feature_dim=1024, embed_dim=768,
                 forward_type="split", dense_mapping_image=2048, dense_mapping_language=512, dense_embed_dim=256,
                 mask_rate=0.75, patch_number=14, text_mae_feature=False, return_simsiam=False, two_view=False, sparse=False,
                 select_topk=False, dtype="bf16", fused_loss=True):
        super().__init__(image_encode, text_encode, use_allgather, dtype=dtype, fused_loss=fused_loss, scale_clamp=None)
        if return_caption:
            raise NotImplementedError("caption head is out of scope (SURVEY.md s2)")
        self.return_dense, self.text_mask_type, self.select_topk = return_dense, text_mask_type, select_topk
        if return_dense:
            self.image_mapping = nn.Linear(dense_mapping_image, dense_embed_dim)
            self.text_mapping = nn.Linear(dense_mapping_language, dense_embed_dim)
        self.logit_scale_dense = nn.Parameter(torch.ones([]))
        nn.init.constant_(self.logit_scale_dense, np.log(1 / 0.07))
        if text_mask_type is not None:
            enc_dim = self.encode_text.text_projection.weight.shape[-1]
            self.text_label_predictor = nn.Linear(enc_dim, self.encode_text.vocab_size)
        self._adopt_towers()

    # ---- evaluation helpers used by filip_solver.py:898,920
    def encode_text_dense(self, texts, return_dense=True):
        flat = self._flat_store
        flat.begin_step()
        _, words = self.encode_text(texts, return_dense=True)
        b, L, w = words.shape
        return LinearFn.apply(words.reshape(b * L, w), self.text_mapping, flat).view(b, L, -1)

    def encode_image_dense(self, image):
        flat = self._flat_store
        flat.begin_step()
        _, dense = self.visual(image, return_dense=True)
        b, n, w = dense.shape
        return LinearFn.apply(dense.reshape(b * n, w), self.image_mapping, flat).view(b, n, -1)

    def encode_image(self, image, return_all=False):
        self._flat_store.begin_step()
        return self.visual(image, return_dense=return_all)

    def get_weighted_dense_logits(self, itok, ttok, b, J, T, label0):
        """filip.py:71-106.  itok [b*J, D], ttok [b*T, D]: mapped (un-normalised) token features."""
        if not self.select_topk:
            raise NameError("select_topk must be True (selected_feat_* undefined otherwise, filip.py:78-93)")
        flat = self._flat_store
        D = itok.shape[-1]
        it_n = engine.L2NormFn.apply(itok, 0.0)                               # fp32 [b*J, D]
        tt_n = engine.L2NormFn.apply(ttok, 0.0)
        idx_i, idx_t = engine.ops.filip_select(it_n.detach().view(b, J, D), tt_n.detach().view(b, T, D))
        base = torch.arange(b, device=idx_i.device, dtype=torch.int64)[:, None]
        sel_i = engine.GatherTokFn.apply(it_n, (idx_i + base * J).reshape(-1))  # [b*16, D]
        sel_t = engine.GatherTokFn.apply(tt_n, (idx_t + base * T).reshape(-1))
        g_i, g_t = dh_dist.all_gather_cat_many([sel_i.view(b, 16 * D), sel_t.view(b, 16 * D)])
        B = g_i.shape[0]
        g_i, g_t = g_i.reshape(B * 16, D), g_t.reshape(B * 16, D)
        scale_d = self.logit_scale_dense.exp()
        li = engine.MaxSimFn.apply(scale_d, it_n, g_t, b, B, J, flat.act_dtype)
        lt = engine.MaxSimFn.apply(scale_d, tt_n, g_i, b, B, T, flat.act_dtype)
        return li, lt

    def forward(self, input, return_dict=False):
        if not return_dict:
            raise NotImplementedError()                                         # filip.py:142
        if not (self.training and self.use_allgather):
            raise NotImplementedError("2-View: Not Implemented")
        flat = self._flat_store
        flat.begin_step()
        images = input["images"]
        caps = input["captions"]
        et = self.encode_text
        if torch.is_tensor(caps):
            ids = caps[:, 0] if caps.dim() == 3 else caps
            if "mlm_labels" not in input and self.text_mask_type is not None:
                from ..bpe import mask_token_ids
                ids, _ = mask_token_ids(ids.cpu(), et.vocab_size)
        else:
            tok = et.tokenize(self.sample_captions(caps), et.context_length, self.text_mask_type)
            ids = tok[0] if self.text_mask_type is not None else tok
        dev = flat.flat_p.device
        ids = engine.to_device_async(ids, dev).long().contiguous()
        b = images.shape[0]
        side = self._fork(images)                                               # text tower on the side stream (clip.py)
        with self._on(side):
            txt, words = engine.TextTowerFn.apply(flat.anchor, ids, et, True)
        img, dense = self.visual(images, return_dense=True)                     # view 1 only (channels 0..2)
        self._join(side, txt, words)
        img_n = engine.L2NormFn.apply(img, 0.0)
        txt_n = engine.L2NormFn.apply(txt, 1e-10)
        scale = self.logit_scale_value()
        label0 = dh_dist.get_rank() * b if dh_dist.is_dist() else 0
        g_img, g_txt = dh_dist.all_gather_cat_many([img_n, txt_n])
        li, lt = LazyLogits(img_n, g_txt, scale, label0), LazyLogits(txt_n, g_img, scale, label0)
        ret = {"logits": (li, lt) if self.fused_loss else (li.materialize(), lt.materialize())}
        if self.return_dense:
            J, T = dense.shape[1], words.shape[1]
            itok = LinearFn.apply(dense.reshape(b * J, -1), self.image_mapping, flat)
            ttok = LinearFn.apply(words.reshape(b * T, -1), self.text_mapping, flat)
            ret["dense_logits"] = self.get_weighted_dense_logits(itok, ttok, b, J, T, label0)
        return ret


def filip_res50(**kwargs):
    """model/filip.py:146-153 (dense tokens = the 7x7 feature map, 2048 wide)."""
    from .resnet import modified_resnet_R50
    image_encode = modified_resnet_R50(**kwargs["image_encode"])
    text_encode = text_transformers(**kwargs["text_encode"])
    return FILIP(image_encode, text_encode, **kwargs["clip"], dense_mapping_image=2048, **_engine_kwargs(kwargs))


def filip_vitb32(**kwargs):
    """model/filip.py:156-163."""
    image_encode = visual_transformer_B32(**kwargs["image_encode"])
    text_encode = text_transformers(**kwargs["text_encode"])
    return FILIP(image_encode, text_encode, **kwargs["clip"], dense_mapping_image=768, **_engine_kwargs(kwargs))
