"""Model builders shared by tests / smoke / bench (construct an engine model from a size config and
load the seeded synthetic state from declip_amd.synth)."""
import torch

from . import synth
from .model.clip import CLIP
from .model.transformer import TextTransformer, VisualTransformer


def build_clip(cfg, dtype="bf16", use_allgather=False, seed=0, logit_scale=None, fused_loss=True, device="cuda",
               load_synth=True):
    vis = VisualTransformer(input_resolution=cfg["res"], patch_size=cfg["patch"], width=cfg["v_width"],
                            layers=cfg["v_layers"], heads=cfg["v_heads"], embed_dim=cfg["embed_dim"])
    txt = TextTransformer(embed_dim=cfg["embed_dim"], context_length=cfg["ctx"], transformer_width=cfg["t_width"],
                          transformer_heads=cfg["t_heads"], transformer_layers=cfg["t_layers"],
                          positional_embedding_flag=True, checkpoint=False, bpe_path=None,
                          text_encode_type="Transformer", text_model_utils={"random": False, "freeze": False},
                          vocab_size=cfg.get("vocab", synth.VOCAB))
    model = CLIP(vis, txt, use_allgather, dtype=dtype, fused_loss=fused_loss)
    if load_synth:
        sd = synth.synth_state(synth.clip_shapes(cfg), seed=seed, logit_scale=logit_scale)
        model.load_state_dict(sd, strict=True)
    model = model.to(device)
    model.train()
    return model
