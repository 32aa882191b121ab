"""DeCLIP on the HIP engine (reference: model/declip.py:132-355): two image views, MLM-masked + augmented text,
SimSiam projector/predictor, nearest-neighbour text supervision, masked-LM head.

Differences in HOW (not what): both image views go through the vision tower in ONE pass (batch 2b, like the
reference's forward_type='image_concat'), both text variants in one pass, all feature gathers in ONE packed
collective, the NN bank lives in HBM, the 12 logits matrices are LazyLogits handles (never materialised)."""
from random import choice

import os

import torch
from torch import nn

from .. import dist as dh_dist
from .. import engine
from ..heads import NNMemoryBankModule, mlm_loss, mlm_loss_packed, prediction_MLP, projection_MLP
from .clip import CLIP, LazyLogits, _engine_kwargs
from .transformer import text_transformers, visual_transformer_B32

__all__ = ["DECLIP", "declip_vitb32", "declip_res50"]


class DECLIP(CLIP):
    def __init__(self, image_encode, text_encode, use_allgather, nn_size=2 ** 16, nn_topk=1, return_dense=False,
                 return_simsiam_text=False, return_simsiam_nn_text=False, return_caption=False, return_nn_bank=False,
                 text_mask_type=None, EDA=True, feature_dim=1024, forward_type="split", dtype="bf16", fused_loss=True,
                 global_nn_bank=False):
        super().__init__(image_encode, text_encode, use_allgather, dtype=dtype, fused_loss=fused_loss)
        if return_dense:
            raise NotImplementedError("These are bugs in the model, Please Check The Codes!")      # declip.py:157
        if return_caption:
            raise NotImplementedError("Not Available")                                             # declip.py:166-167
        if return_simsiam_text or return_simsiam_nn_text:
            raise NotImplementedError("text SimSiam branches are unreachable in the reference solver (SURVEY.md s9 #19)")
        self.projector = projection_MLP(feature_dim)
        self.predictor = prediction_MLP(1024)
        self.return_nn_bank = return_nn_bank
        self.text_mask_type = text_mask_type
        self.EDA = EDA
        self.forward_type = forward_type
        self.global_nn_bank = global_nn_bank
        self.emd = None
        if text_mask_type is not None:
            enc_dim = self.encode_text.text_projection.weight.shape[-1]
            self.text_label_predictor = nn.Linear(enc_dim, self.encode_text.vocab_size)
        if return_nn_bank:
            self.nn_replacer_img = NNMemoryBankModule(size=nn_size, topk=nn_topk)
            self.nn_replacer_text = NNMemoryBankModule(size=nn_size, topk=nn_topk)
        # the step spends two caption views and the masked-LM pass in the text tower: its packed attention in two length buckets pays
        # here (+0.6 % pairs/s, profiles/r06_attention_variants.txt), unlike in the CLIP step (engine._buckets; DH_ATTN_BUCKETS overrides)
        self.encode_text.__dict__["_dh_attn_buckets"] = True
        self._adopt_towers()

    def _adopt_towers(self):
        for m in self.modules():
            if m is not self and (hasattr(m, "_flat") or isinstance(m, (projection_MLP, prediction_MLP))):
                m.__dict__["_engine_root"] = self

    def text_modules(self):
        ret = super().text_modules()
        if self.text_mask_type is not None:
            ret.append(self.text_label_predictor)
        return ret

    def visual_modules(self):
        return super().visual_modules() + [self.predictor, self.projector]

    def encode_image(self, image, return_dense=False):
        self._flat_store.begin_step()
        return self.visual(image, return_dense=return_dense)

    # ------------------------------------------------------------------------------------------
    def _augment(self, texts):
        """EDA text augmentation (declip.py:203-212; CPU side).  Pre-tokenised input carries the augmented ids
        as a second row (captions [b, 2, ctx])."""
        if self.emd is None:
            from textaugment import EDA
            self.emd = EDA()
        out = []
        for caption in texts:
            aug = choice([self.emd.synonym_replacement, self.emd.random_swap, self.emd.random_deletion])(caption)
            out.append(" ".join(aug) if isinstance(aug, list) else aug)
        return out

    def prepare_captions(self, caps):
        """Everything forward() does to caption STRINGS (declip.py:203-230: sample the first caption, EDA-augment it, tokenise both,
        mask the original for the MLM head), as a function the data pipeline can run ahead of the step
        (prefetch.DataPrefetcher(text_prep=model.prepare_captions) does, on its worker thread): returns the batch entries
        `captions` int64 [b, 2, ctx] (masked | augmented) and `mlm_labels` [b, ctx] that forward() accepts as pre-tokenised input."""
        et = self.encode_text
        texts = self.sample_captions(caps)
        if not self.EDA:
            raise NotImplementedError("No EDA")
        texts_aug = self._augment(texts)
        tok = et.tokenize(texts, et.context_length, self.text_mask_type)
        ids, labels = tok if self.text_mask_type is not None else (tok, None)
        ids_aug = et.tokenize(texts_aug, et.context_length)
        out = {"captions": torch.stack([ids, ids_aug], dim=1)}
        if labels is not None:
            out["mlm_labels"] = labels
        return out

    def forward(self, input, return_dict=False):
        if not (self.training and self.use_allgather):
            raise NotImplementedError("2-View: Not Implemented")                                   # declip.py:301-302
        if not return_dict:
            raise NotImplementedError("Must Return A Dict")                                        # declip.py:336
        flat = self._flat_store
        flat.begin_step()
        images = input["images"]
        caps = input["captions"]
        et = self.encode_text
        # ---- text ids: (masked) caption + augmented caption
        labels = None
        if torch.is_tensor(caps):
            ids, ids_aug = (caps[:, 0], caps[:, 1]) if caps.dim() == 3 else (caps, caps)
            if "mlm_labels" in input:                      # pre-masked ids + labels supplied by the data pipeline
                labels = input["mlm_labels"]
            elif self.text_mask_type is not None:
                from ..bpe import mask_token_ids
                ids, labels = mask_token_ids(ids.cpu(), et.vocab_size)
        else:
            prep = self.prepare_captions(caps)           # string captions handed straight to forward(): the reference's in-line path
            ids, ids_aug, labels = prep["captions"][:, 0], prep["captions"][:, 1], prep.get("mlm_labels")
        dev = flat.flat_p.device
        b = images.shape[0]
        ids_cat = torch.cat([engine.to_device_async(ids, dev), engine.to_device_async(ids_aug, dev)], dim=0).long().contiguous()
        tag = (getattr(caps, "_dh_rows", None)
               if torch.is_tensor(caps) and caps.dim() == 3 and caps.shape[1] == 2 and "mlm_labels" in input else None)   # exactly the 2 variants used
        if tag is not None and tag[0] == caps._version:
            ids_cat._dh_rows = (ids_cat._version, tag[1])        # both variants of every caption: the row count the prefetcher took on the host
        want_words = self.text_mask_type is not None
        side = self._fork(images)                # text tower on the side stream, both image views on the caller's (clip.py)
        # DH_TEXT_PACKED: only the caption rows up to <|endoftext|> are computed (engine.PackedCaptions); not when a subclass needs
        # the per-token features of the padded layout (DeFILIP's token selection ranks the padding too)
        packed = engine.text_packed_mode() != 0 and not getattr(self, "return_filip", False)
        with self._on(side):
            if packed:
                tout = engine.TextTowerPackedFn.apply(flat.anchor, ids_cat, et, want_words)
            else:
                tout = engine.TextTowerFn.apply(flat.anchor, ids_cat, et, want_words)
        txt_cat, words = (tout[0], tout[1]) if want_words else (tout, None)
        # ---- both image views in one pass
        want_dense = bool(getattr(self, "return_filip", False))
        vout = self.visual(images, n_views=2, return_dense=want_dense)
        self._join(side, *(tout if isinstance(tout, (tuple, list)) else (tout,)))
        img_cat, dense_cat = (vout[0], vout[1]) if want_dense else (vout, None)   # [2b, E] fp32, view-major
        # ---- SimSiam on the UN-normalised image features (declip.py:238-241), BN statistics per view
        z = self.projector(img_cat, groups=2)
        p = self.predictor(z, groups=2)
        z1, z2 = z[:b], z[b:]
        p1, p2 = p[:b], p[b:]
        # ---- normalised features
        img_n = engine.L2NormFn.apply(img_cat, 0.0)
        txt_n = engine.L2NormFn.apply(txt_cat, 1e-10)
        i1, i2 = img_n[:b], img_n[b:]
        t, t_aug = txt_n[:b], txt_n[b:]
        scale = self.logit_scale_value()
        rank = dh_dist.get_rank()
        label0 = rank * b if dh_dist.is_dist() else 0
        # ---- NN text supervision (declip.py:281-300): search before enqueue, three calls as in the reference
        nn_feats = []
        if self.return_nn_bank:
            bank = self.nn_replacer_text
            enq_t = enq_aug = None
            if self.global_nn_bank and dh_dist.is_dist():
                # north_star: "the NN feature queue is likewise gathered" -- every rank enqueues ALL ranks' caption features, so the
                # W banks stay identical and hold W x more distinct neighbours per step (the reference keeps W independent per-rank
                # queues, memory_bank.py:66; that stays the default).  One extra small gather, off the gradient path.
                enq_t, enq_aug = dh_dist.all_gather_cat_many([t.detach(), t_aug.detach()])
                bank.spill_rows = max(bank.spill_rows, int(enq_t.shape[0]))      # the queue's storage is sized ONCE, for the gathered batch
            nn_t = bank(t, update=False)[0]
            nn_t_aug = bank(t_aug, update=True, enqueue=enq_aug)[0]
            bank(t, update=True, query=False, enqueue=enq_t)               # the reference's third call only enqueues
            nn_t = engine.L2NormFn.apply(nn_t, 1e-10)
            nn_t_aug = engine.L2NormFn.apply(nn_t_aug, 1e-10)
            nn_feats = [nn_t, nn_t_aug]
        # ---- ONE packed all-gather for everything that is gathered (reference: 4 + 2 all_gathers + 2 barriers), in flight on the
        # communication stream while the masked-LM head (three vocabulary-wide GEMMs) runs on the compute stream
        pending = dh_dist.all_gather_cat_many_async([i1, i2, t, t_aug] + nn_feats)
        ret = {}
        if self.text_mask_type is not None:
            if packed:
                ret["text_self_supervised"] = mlm_loss_packed(words, engine.packed_captions(ids_cat, flat.act_dtype), b, labels,
                                                              self.text_label_predictor, flat)
            else:
                ret["text_self_supervised"] = mlm_loss(words[:b], labels, self.text_label_predictor, flat)
        gathered = pending.result()
        g_i1, g_i2, g_t, g_t_aug = gathered[:4]
        L = lambda q, k: LazyLogits(q, k, scale, label0)
        ret["logits"] = L(i1, g_t), L(i2, g_t), L(t, g_i1), L(t, g_i2)
        ret["logits_aug"] = L(i1, g_t_aug), L(i2, g_t_aug), L(t_aug, g_i1), L(t_aug, g_i2)
        ret["simsiam_features"] = p1, p2, z1, z2
        ret["features"] = t, i1, i2
        if self.return_nn_bank:
            g_nn, g_nn_aug = gathered[4], gathered[5]
            ret["nn_text_logits"] = L(i1, g_nn), L(i2, g_nn), L(i1, g_nn_aug), L(i2, g_nn_aug)
        self._extra_outputs(ret, dict(b=b, dense=dense_cat, words=words, label0=label0))
        if not self.fused_loss:
            for k in ("logits", "logits_aug", "nn_text_logits"):
                if k in ret:
                    ret[k] = tuple(x.materialize() for x in ret[k])
        return ret


    def _extra_outputs(self, ret, st):
        """hook for DEFILIP (adds the token-wise 'filip' logits)."""
        return None


def declip_res50(**kwargs):
    """model/declip.py:339-346 (the two image views go through the ModifiedResNet one after the other: per-view BatchNorm
    statistics, as in the reference)."""
    from .resnet import modified_resnet_R50
    image_encode = modified_resnet_R50(**kwargs["image_encode"])
    text_encode = text_transformers(**kwargs["text_encode"])
    return DECLIP(image_encode, text_encode, **kwargs["clip"], **_engine_kwargs(kwargs))


def declip_vitb32(**kwargs):
    """model/declip.py:348-355."""
    image_encode = visual_transformer_B32(**kwargs["image_encode"])
    text_encode = text_transformers(**kwargs["text_encode"])
    return DECLIP(image_encode, text_encode, **kwargs["clip"], **_engine_kwargs(kwargs))
