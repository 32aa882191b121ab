"""SLIP on the HIP engine (reference: model/slip.py:209-306): CLIP on the base view + SimCLR (NT-Xent) between two
augmented views through a 768->4096->4096->256 BN-MLP on the ln_post feature.

Reference quirks kept: the text tower module is named `text_encoder` (state_dict prefix) and `encode_text` is a
METHOD (slip.py:117,237-243); `logit_scale.exp()` is NOT clamped (slip.py:265).  The three channel-stacked views go
through the vision tower in ONE pass (batch 3b); all four gathers are one packed collective."""
import torch

from .. import dist as dh_dist
from .. import engine
from ..heads import projection_MLP
from .clip import CLIP, LazyLogits, _engine_kwargs
from .transformer import text_transformers, visual_transformer_B32

__all__ = ["SLIP", "slip_vitb32", "slip_res50"]


class SLIP(CLIP):
    def __init__(self, image_encode, text_encode, use_allgather, EDA=True, feature_dim=1024, sim_dim=256,
                 forward_type="split", return_sim=False, dtype="bf16", fused_loss=True):
        super().__init__(image_encode, text_encode, use_allgather, dtype=dtype, fused_loss=fused_loss, scale_clamp=None)
        del self.encode_text                       # registered as `text_encoder` in the reference (slip.py:117)
        self.text_encoder = text_encode
        self.return_sim = return_sim
        if return_sim:
            self.predictor_sim = projection_MLP(feature_dim, hidden_dim=4096, out_dim=sim_dim, out_bn=False)
        self.forward_type = forward_type
        for m in self.modules():
            if m is not self and (hasattr(m, "_flat") or isinstance(m, projection_MLP)):
                m.__dict__["_engine_root"] = self

    def text_parameters(self):
        return [self.logit_scale, self.text_encoder.positional_embedding]

    def text_modules(self):
        et = self.text_encoder
        return [et.transformer, et.text_projection, et.token_embedding, et.ln_final]

    def visual_modules(self):
        return [self.visual, self.predictor_sim]

    def encode_image(self, image, return_dense=False, return_sim=False):
        self._flat_store.begin_step()
        out = self.visual(image, return_dense=return_dense, return_feature=return_sim)
        if return_sim:
            out = (*out[:-1], self.predictor_sim(out[-1]))
        return out

    def encode_text(self, text, text_mask_type=None, return_sim=False):
        assert not return_sim
        self._flat_store.begin_step()
        return self.text_encoder(text, mask_type=text_mask_type) if text_mask_type else self.text_encoder(text)

    def forward(self, input, return_dict=False):
        if not (self.training and self.use_allgather):
            raise NotImplementedError("2-View: Not Implemented")            # slip.py:281-282
        if not return_dict:
            raise NotImplementedError("Must Return A Dict")                 # slip.py:286
        flat = self._flat_store
        flat.begin_step()
        images = input["images"]                                            # [b, 9, H, W]: base, aug1, aug2
        texts = self.sample_captions(input["captions"])
        b = images.shape[0]
        side = self._fork(images)                                           # text tower on the side stream (clip.py)
        with self._on(side):
            txt = self.text_encoder(texts)
        proj, feat = self.visual(images, return_feature=True, n_views=3)    # [3b, E] fp32, [3b, width]
        self._join(side, txt)
        sim = self.predictor_sim(feat[b:], groups=2)                        # BN statistics per view
        sim1, sim2 = sim[:b], sim[b:]
        img_n = engine.L2NormFn.apply(proj[:b], 0.0)
        txt_n = engine.L2NormFn.apply(txt, 1e-10)
        scale = self.logit_scale_value()                                    # no clamp (slip.py:265)
        label0 = dh_dist.get_rank() * b if dh_dist.is_dist() else 0
        sim1f, sim2f = sim1.float(), sim2.float()
        g_img, g_txt, g_s1, g_s2 = dh_dist.all_gather_cat_many([img_n, txt_n, sim1f, sim2f])
        ret = {}
        li, lt = LazyLogits(img_n, g_txt, scale, label0), LazyLogits(txt_n, g_img, scale, label0)
        ret["logits"] = (li, lt) if self.fused_loss else (li.materialize(), lt.materialize())
        ret["sim_features"] = sim1f, g_s1, sim2f, g_s2
        ret["features"] = txt_n, img_n
        return ret


def slip_res50(**kwargs):
    """model/slip.py:289-297 (experiments/slip_experiments/yfcc15m/yfcc15m_r50_slip/config.yaml: image_encode {embed_dim 1024,
    bn_* / use_sync_bn}, clip {return_sim, feature_dim 2048, sim_dim 256}).  The reference's own forward of this model stops at
    `self.visual(image, return_feature=True)` -- its ModifiedResNet.forward has no such argument; see model/resnet.py."""
    from .resnet import modified_resnet_R50
    image_encode = modified_resnet_R50(**kwargs["image_encode"])
    text_encode = text_transformers(**kwargs["text_encode"])
    return SLIP(image_encode, text_encode, **kwargs["clip"], **_engine_kwargs(kwargs))


def slip_vitb32(**kwargs):
    """model/slip.py:299-306."""
    image_encode = visual_transformer_B32(**kwargs["image_encode"])
    text_encode = text_transformers(**kwargs["text_encode"])
    return SLIP(image_encode, text_encode, **kwargs["clip"], **_engine_kwargs(kwargs))
