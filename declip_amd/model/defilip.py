"""DeFILIP = DeCLIP + FILIP's token-wise max-sim logits (reference: model/defilip.py:149-438).

Returns DeCLIP's dict plus 'filip' (dense logits of image view 1 x masked caption) and, with dense_aug, 'filip_aug'
(view2 x caption, view1 x augmented caption, view2 x augmented caption; defilip.py:330-345,400-403)."""
import numpy as np
import torch
from torch import nn

from ..heads import LinearFn
from .clip import _engine_kwargs
from .declip import DECLIP
from .filip import FILIP
from .transformer import text_transformers, visual_transformer_B32

__all__ = ["DEFILIP", "defilip_vitb32"]


class DEFILIP(DECLIP):
    def __init__(self, image_encode, text_encode, use_allgather, return_filip=False, dense_embed_dim=256, dense_mapping_image=768,
                 dense_mapping_language=512, dense_aug=False, **kw):
        super().__init__(image_encode, text_encode, use_allgather, **kw)
        self.return_filip, self.dense_aug = return_filip, dense_aug
        if return_filip:
            self.select_topk = True
            self.logit_scale_dense = nn.Parameter(torch.ones([]))
            nn.init.constant_(self.logit_scale_dense, np.log(1 / 0.07))
            self.image_mapping = nn.Linear(dense_mapping_image, dense_embed_dim)
            self.text_mapping = nn.Linear(dense_mapping_language, dense_embed_dim)
        self._adopt_towers()

    get_weighted_dense_logits = FILIP.get_weighted_dense_logits          # identical algorithm (defilip.py:224-269)

    def _extra_outputs(self, ret, st):
        if not self.return_filip:
            return
        flat = self._flat_store
        b, dense, words, label0 = st["b"], st["dense"], st["words"], st["label0"]
        if words is None:
            raise NotImplementedError("return_filip needs text_mask_type (word features), as in the shipped config")
        J, T = dense.shape[1], words.shape[1]
        nd = 2 if self.dense_aug else 1
        itok = LinearFn.apply(dense[:nd * b].reshape(nd * b * J, -1), self.image_mapping, flat)
        ttok = LinearFn.apply(words[:nd * b].reshape(nd * b * T, -1), self.text_mapping, flat)
        i1, t1 = itok[:b * J], ttok[:b * T]
        ret["filip"] = self.get_weighted_dense_logits(i1, t1, b, J, T, label0)
        if self.dense_aug:
            i2, t2 = itok[b * J:], ttok[b * T:]
            ret["filip_aug"] = (*self.get_weighted_dense_logits(i2, t1, b, J, T, label0),
                                *self.get_weighted_dense_logits(i1, t2, b, J, T, label0),
                                *self.get_weighted_dense_logits(i2, t2, b, J, T, label0))


def defilip_vitb32(**kwargs):
    """model/defilip.py:431-438."""
    image_encode = visual_transformer_B32(**kwargs["image_encode"])
    text_encode = text_transformers(**kwargs["text_encode"])
    return DEFILIP(image_encode, text_encode, **kwargs["clip"], **_engine_kwargs(kwargs))
