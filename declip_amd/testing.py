"""Model builders shared by tests / smoke / bench (construct an engine model from a size config and
load the seeded synthetic state from declip_amd.synth)."""
import torch

from . import synth
from .model.clip import CLIP
from .model.transformer import TextTransformer, VisualTransformer


def build_clip(cfg, dtype="bf16", use_allgather=False, seed=0, logit_scale=None, fused_loss=True, device="cuda",
               load_synth=True):
    if cfg.get("vision") == "resnet":
        from .model.resnet import ModifiedResNet
        vis = ModifiedResNet(layers=tuple(cfg["r_layers"]), embed_dim=cfg["embed_dim"], heads=cfg["r_heads"],
                             input_resolution=cfg["res"], width=cfg["r_width"], use_sync_bn=cfg.get("r_sync_bn", False),
                             bn_group_size=cfg.get("r_bn_group", 1))
    else:
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


def build_declip(cfg, dtype="bf16", seed=0, nn_size=256, fused_loss=True, device="cuda", load_synth=True, mlm=True,
                 nn_bank=True):
    from .model.declip import DECLIP
    if cfg.get("vision") == "resnet":
        from .model.resnet import ModifiedResNet
        vis = ModifiedResNet(layers=tuple(cfg["r_layers"]), embed_dim=cfg["embed_dim"], heads=cfg["r_heads"],
                             input_resolution=cfg["res"], width=cfg["r_width"], use_sync_bn=cfg.get("r_sync_bn", False),
                             bn_group_size=cfg.get("r_bn_group", 1))
    else:
        vis = VisualTransformer(input_resolution=cfg["res"], patch_size=cfg["patch"], width=cfg["v_width"],
                                layers=cfg["v_layers"], heads=cfg["v_heads"], embed_dim=cfg["embed_dim"])
    txt = TextTransformer(embed_dim=cfg["embed_dim"], context_length=cfg["ctx"], transformer_width=cfg["t_width"],
                          transformer_heads=cfg["t_heads"], transformer_layers=cfg["t_layers"],
                          positional_embedding_flag=True, checkpoint=False, bpe_path=None,
                          text_encode_type="Transformer", text_model_utils={"random": False, "freeze": False},
                          vocab_size=cfg.get("vocab", synth.VOCAB))
    model = DECLIP(vis, txt, True, nn_size=nn_size, nn_topk=1, return_nn_bank=nn_bank,
                   text_mask_type="MLM" if mlm else None, EDA=True, feature_dim=cfg["embed_dim"], dtype=dtype,
                   fused_loss=fused_loss)
    if load_synth:
        sd = synth.synth_state(synth.declip_shapes(dict(cfg, mlm=mlm)), seed=seed)
        model.load_state_dict(sd, strict=True)
    model = model.to(device)
    model.train()
    if nn_bank and load_synth:
        model.nn_replacer_text.bank = synth.synth_bank(nn_size, cfg["embed_dim"], seed=seed).to(device)
        model.nn_replacer_text.bank_ptr = 0
    return model


def _captions_to(caps, device):
    """captions to the device WITH the packed row count taken on the host copy (what declip_amd.prefetch does for a real loader):
    the step then never reads a count back from the device (no host stall in forward(), and the step stays capturable)."""
    rows = int((caps.reshape(-1, caps.shape[-1]).argmax(dim=-1) + 1).sum())
    out = caps.to(device)
    out._dh_rows = (out._version, rows)
    return out


def declip_batch(cfg, b, seed=0, device="cuda"):
    """seeded DeCLIP batch: two channel-stacked views, masked ids + labels, augmented ids."""
    images = synth.synth_images(b, views=2, res=cfg["res"], seed=seed)
    ids = synth.synth_tokens(b, ctx=cfg["ctx"], seed=seed, vocab=cfg["vocab"])
    ids_aug = synth.synth_tokens(b, ctx=cfg["ctx"], seed=seed + 50, vocab=cfg["vocab"])
    ids_masked, labels = synth.synth_mlm(ids, cfg["vocab"], seed=seed)
    caps = torch.stack([ids_masked, ids_aug], dim=1)
    return {"images": images.to(device), "captions": _captions_to(caps, device), "mlm_labels": labels}


def build_slip(cfg, dtype="bf16", seed=0, fused_loss=True, device="cuda", load_synth=True):
    from .model.slip import SLIP
    vis = VisualTransformer(input_resolution=cfg["res"], patch_size=cfg["patch"], width=cfg["v_width"],
                            layers=cfg["v_layers"], heads=cfg["v_heads"], embed_dim=cfg["embed_dim"])
    txt = TextTransformer(embed_dim=cfg["embed_dim"], context_length=cfg["ctx"], transformer_width=cfg["t_width"],
                          transformer_heads=cfg["t_heads"], transformer_layers=cfg["t_layers"],
                          positional_embedding_flag=True, checkpoint=False, bpe_path=None,
                          text_encode_type="Transformer", text_model_utils={"random": False, "freeze": False},
                          vocab_size=cfg.get("vocab", synth.VOCAB))
    model = SLIP(vis, txt, True, return_sim=True, feature_dim=cfg["v_width"], sim_dim=256, dtype=dtype, fused_loss=fused_loss)
    if load_synth:
        model.load_state_dict(synth.synth_state(synth.slip_shapes(cfg), seed=seed), strict=True)
    model = model.to(device)
    model.train()
    return model


def slip_batch(cfg, b, seed=0, device="cuda"):
    images = synth.synth_images(b, views=3, res=cfg["res"], seed=seed)
    ids = synth.synth_tokens(b, ctx=cfg["ctx"], seed=seed, vocab=cfg["vocab"])
    return {"images": images.to(device), "captions": _captions_to(ids, device)}


def build_filip(cfg, dtype="bf16", seed=0, fused_loss=True, device="cuda", load_synth=True):
    from .model.filip import FILIP
    if cfg.get("vision") == "resnet":
        from .model.resnet import ModifiedResNet
        vis = ModifiedResNet(layers=tuple(cfg["r_layers"]), embed_dim=cfg["embed_dim"], heads=cfg["r_heads"],
                             input_resolution=cfg["res"], width=cfg["r_width"], use_sync_bn=False)
    else:
        vis = VisualTransformer(input_resolution=cfg["res"], patch_size=cfg["patch"], width=cfg["v_width"],
                                layers=cfg["v_layers"], heads=cfg["v_heads"], embed_dim=cfg["embed_dim"])
    txt = TextTransformer(embed_dim=cfg["embed_dim"], context_length=cfg["ctx"], transformer_width=cfg["t_width"],
                          transformer_heads=cfg["t_heads"], transformer_layers=cfg["t_layers"],
                          positional_embedding_flag=True, checkpoint=False, bpe_path=None,
                          text_encode_type="Transformer", text_model_utils={"random": False, "freeze": False},
                          vocab_size=cfg.get("vocab", synth.VOCAB))
    model = FILIP(vis, txt, True, text_mask_type="MLM", return_dense=True, select_topk=True, feature_dim=cfg["v_width"],
                  dense_mapping_image=cfg["v_width"], dense_mapping_language=cfg["t_width"], dtype=dtype, fused_loss=fused_loss)
    if load_synth:
        model.load_state_dict(synth.synth_state(synth.filip_shapes(cfg), seed=seed), strict=True)
    model = model.to(device)
    model.train()
    return model


def filip_batch(cfg, b, seed=0, device="cuda"):
    images = synth.synth_images(b, views=2, res=cfg["res"], seed=seed)
    ids = synth.synth_tokens(b, ctx=cfg["ctx"], seed=seed, vocab=cfg["vocab"], min_len=cfg["ctx"] - 6)
    ids_masked, labels = synth.synth_mlm(ids, cfg["vocab"], seed=seed)
    return {"images": images.to(device), "captions": _captions_to(ids_masked, device), "mlm_labels": labels}


def build_defilip(cfg, dtype="bf16", seed=0, nn_size=256, device="cuda", load_synth=True, dense_aug=False):
    from .model.defilip import DEFILIP
    vis = VisualTransformer(input_resolution=cfg["res"], patch_size=cfg["patch"], width=cfg["v_width"],
                            layers=cfg["v_layers"], heads=cfg["v_heads"], embed_dim=cfg["embed_dim"])
    txt = TextTransformer(embed_dim=cfg["embed_dim"], context_length=cfg["ctx"], transformer_width=cfg["t_width"],
                          transformer_heads=cfg["t_heads"], transformer_layers=cfg["t_layers"],
                          positional_embedding_flag=True, checkpoint=False, bpe_path=None,
                          text_encode_type="Transformer", text_model_utils={"random": False, "freeze": False},
                          vocab_size=cfg.get("vocab", synth.VOCAB))
    model = DEFILIP(vis, txt, True, return_filip=True, dense_mapping_image=cfg["v_width"], dense_mapping_language=cfg["t_width"],
                    dense_aug=dense_aug, nn_size=nn_size, nn_topk=1, return_nn_bank=True, text_mask_type="MLM", EDA=True,
                    feature_dim=cfg["embed_dim"], dtype=dtype)
    if load_synth:
        model.load_state_dict(synth.synth_state(synth.defilip_shapes(cfg), seed=seed), strict=True)
    model = model.to(device)
    model.train()
    if load_synth:
        model.nn_replacer_text.bank = synth.synth_bank(nn_size, cfg["embed_dim"], seed=seed).to(device)
        model.nn_replacer_text.bank_ptr = 0
    return model


def defilip_batch(cfg, b, seed=0, device="cuda"):
    images = synth.synth_images(b, views=2, res=cfg["res"], seed=seed)
    ids = synth.synth_tokens(b, ctx=cfg["ctx"], seed=seed, vocab=cfg["vocab"], min_len=cfg["ctx"] - 6)
    ids_aug = synth.synth_tokens(b, ctx=cfg["ctx"], seed=seed + 50, vocab=cfg["vocab"], min_len=cfg["ctx"] - 6)
    ids_masked, labels = synth.synth_mlm(ids, cfg["vocab"], seed=seed)
    return {"images": images.to(device), "captions": _captions_to(torch.stack([ids_masked, ids_aug], dim=1), device), "mlm_labels": labels}
