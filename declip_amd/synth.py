"""Seeded synthetic inputs and parameters (the engine's "fake" data source, cf. the
reference's read_from: fake, base_dataset.py:81-86).  Shared by bench.py, the tests, the
oracle and the golden generator: data generation only, no model arithmetic.
Everything is a pure function of integer seeds so that both sides of a parity
check (reference/oracle on CPU, HIP engine on the GPU) see bit-identical
inputs and parameters (SURVEY.md s8(d) "Synthetic inputs").
"""
import math
from collections import OrderedDict

import torch

SOT, EOT, MASK_TOKEN, VOCAB = 49407, 49408, 49406, 49409  # simple_tokenizer.py:73-75


def synth_images(b, views=1, res=224, seed=0):
    g = torch.Generator().manual_seed(1000 + seed)
    return torch.randn(b, 3 * views, res, res, generator=g)


def synth_tokens(b, ctx=77, seed=1, vocab=VOCAB, min_len=6, max_len=None):
    """[b,ctx] int64: SOT, len~U{min..max} ids < MASK_TOKEN, EOT (= max id), zero pad
    (reference text_transformer.py:144-180 produces exactly this layout)."""
    g = torch.Generator().manual_seed(2000 + seed)
    max_len = (ctx - 2) if max_len is None else min(max_len, ctx - 2)
    min_len = min(min_len, max_len)
    ids = torch.zeros(b, ctx, dtype=torch.long)
    lens = torch.randint(min_len, max_len + 1, (b,), generator=g)
    hi = min(vocab - 3, MASK_TOKEN)
    for i in range(b):
        n = int(lens[i])
        ids[i, 0] = vocab - 2
        ids[i, 1:1 + n] = torch.randint(0, hi, (n,), generator=g)
        ids[i, 1 + n] = vocab - 1
    return ids


def vit_shapes(width, layers, patch, res, embed_dim, prefix="visual."):
    s = OrderedDict()
    n_tok = (res // patch) ** 2 + 1
    s[prefix + "class_embedding"] = (width,)
    s[prefix + "positional_embedding"] = (n_tok, width)
    s[prefix + "proj"] = (width, embed_dim)
    s[prefix + "conv1.weight"] = (width, 3, patch, patch)
    s[prefix + "ln_pre.weight"] = (width,)
    s[prefix + "ln_pre.bias"] = (width,)
    _block_shapes(s, prefix + "transformer.resblocks.", width, layers)
    s[prefix + "ln_post.weight"] = (width,)
    s[prefix + "ln_post.bias"] = (width,)
    return s


def text_shapes(width, layers, ctx, embed_dim, vocab=VOCAB, prefix="encode_text."):
    s = OrderedDict()
    s[prefix + "positional_embedding"] = (ctx, width)
    _block_shapes(s, prefix + "transformer.resblocks.", width, layers)
    s[prefix + "token_embedding.weight"] = (vocab, width)
    s[prefix + "ln_final.weight"] = (width,)
    s[prefix + "ln_final.bias"] = (width,)
    s[prefix + "text_projection.weight"] = (embed_dim, width)
    s[prefix + "text_projection.bias"] = (embed_dim,)
    return s


def _block_shapes(s, prefix, d, layers):
    for i in range(layers):
        p = "%s%d." % (prefix, i)
        s[p + "attn.in_proj_weight"] = (3 * d, d)
        s[p + "attn.in_proj_bias"] = (3 * d,)
        s[p + "attn.out_proj.weight"] = (d, d)
        s[p + "attn.out_proj.bias"] = (d,)
        s[p + "ln_1.weight"] = (d,)
        s[p + "ln_1.bias"] = (d,)
        s[p + "mlp.c_fc.weight"] = (4 * d, d)
        s[p + "mlp.c_fc.bias"] = (4 * d,)
        s[p + "mlp.c_proj.weight"] = (d, 4 * d)
        s[p + "mlp.c_proj.bias"] = (d,)
        s[p + "ln_2.weight"] = (d,)
        s[p + "ln_2.bias"] = (d,)


def synth_state(shapes, seed=0, logit_scale=None):
    """Deterministic, non-degenerate parameter values for a name->shape map.

    Not the reference's init (un-seeded, visual_transformer.py:29-38): both sides
    of every parity check load THIS state.  Each tensor has its own generator
    keyed by (seed, index) so adding tensors never shifts the others."""
    sd = OrderedDict()
    for idx, (name, shape) in enumerate(shapes.items()):
        g = torch.Generator().manual_seed(seed * 100003 + idx * 7919 + 17)
        leaf = name.rsplit(".", 1)[-1]
        is_norm = any(t in name for t in (".ln_", "ln_pre", "ln_post", "ln_final", ".bn", "downsample.1."))
        if name.endswith("logit_scale") or name.endswith("logit_scale_dense"):
            v = torch.full(shape, math.log(1 / 0.07) if logit_scale is None else logit_scale)
        elif is_norm and leaf == "weight":
            v = 1.0 + 0.1 * torch.randn(shape, generator=g)
        elif leaf in ("bias", "in_proj_bias"):
            v = 0.02 * torch.randn(shape, generator=g)
        elif leaf in ("running_mean",):
            v = torch.zeros(shape)
        elif leaf in ("running_var",):
            v = torch.ones(shape)
        elif leaf == "num_batches_tracked":
            v = torch.zeros(shape, dtype=torch.long)
        elif "positional_embedding" in name or "token_embedding" in name:
            v = 0.02 * torch.randn(shape, generator=g)
        elif "class_embedding" in name:
            v = shape[0] ** -0.5 * torch.randn(shape, generator=g)
        elif len(shape) >= 2:
            fan_in = 1
            for k in shape[1:]:
                fan_in *= k
            if leaf == "proj":  # visual.proj is [width, embed] (x @ proj)
                fan_in = shape[0]
            v = fan_in ** -0.5 * torch.randn(shape, generator=g)
        else:
            v = 0.02 * torch.randn(shape, generator=g)
        sd[name] = v
    return sd


def resnet_shapes(width, layers, res, embed_dim, prefix="visual."):
    """ModifiedResNet state_dict (image_encoder/modified_resnet.py:118-191), registration order of the reference."""
    s = OrderedDict()
    s[prefix + "conv1.weight"] = (width // 2, 3, 3, 3)
    _bn_shapes(s, prefix + "bn1.", width // 2)
    s[prefix + "conv2.weight"] = (width // 2, width // 2, 3, 3)
    _bn_shapes(s, prefix + "bn2.", width // 2)
    s[prefix + "conv3.weight"] = (width, width // 2, 3, 3)
    _bn_shapes(s, prefix + "bn3.", width)
    inplanes = width
    for li, (planes, blocks) in enumerate(zip((width, width * 2, width * 4, width * 8), layers)):
        for bi in range(blocks):
            stride = 2 if (li > 0 and bi == 0) else 1
            p = "%slayer%d.%d." % (prefix, li + 1, bi)
            s[p + "conv1.weight"] = (planes, inplanes, 1, 1)
            _bn_shapes(s, p + "bn1.", planes)
            s[p + "conv2.weight"] = (planes, planes, 3, 3)
            _bn_shapes(s, p + "bn2.", planes)
            s[p + "conv3.weight"] = (planes * 4, planes, 1, 1)
            _bn_shapes(s, p + "bn3.", planes * 4)
            if stride > 1 or inplanes != planes * 4:
                s[p + "downsample.0.weight"] = (planes * 4, inplanes, 1, 1)
                _bn_shapes(s, p + "downsample.1.", planes * 4)
            inplanes = planes * 4
    d = width * 32
    s[prefix + "attnpool.positional_embedding"] = ((res // 32) ** 2 + 1, d)
    for n in ("k_proj", "q_proj", "v_proj"):
        s[prefix + "attnpool.%s.weight" % n], s[prefix + "attnpool.%s.bias" % n] = (d, d), (d,)
    s[prefix + "attnpool.c_proj.weight"], s[prefix + "attnpool.c_proj.bias"] = (embed_dim, d), (embed_dim,)
    s[prefix + "fc.weight"], s[prefix + "fc.bias"] = (embed_dim, 2048), (embed_dim,)      # modified_resnet.py:167 (2048 hard-coded)
    return s


def clip_shapes(cfg):
    """cfg keys: v_width v_layers patch res t_width t_layers ctx embed_dim vocab
    (vision == "resnet": r_width r_layers instead of v_width v_layers patch)"""
    s = OrderedDict()
    s["logit_scale"] = (1,)
    if cfg.get("vision") == "resnet":
        s.update(resnet_shapes(cfg["r_width"], cfg["r_layers"], cfg["res"], cfg["embed_dim"]))
    else:
        s.update(vit_shapes(cfg["v_width"], cfg["v_layers"], cfg["patch"], cfg["res"], cfg["embed_dim"]))
    s.update(text_shapes(cfg["t_width"], cfg["t_layers"], cfg["ctx"], cfg["embed_dim"],
                         cfg.get("vocab", VOCAB), prefix=cfg.get("text_prefix", "encode_text.")))
    return s


def _bn_shapes(s, prefix, c):
    s[prefix + "weight"] = (c,)
    s[prefix + "bias"] = (c,)
    s[prefix + "running_mean"] = (c,)
    s[prefix + "running_var"] = (c,)
    s[prefix + "num_batches_tracked"] = ()


def declip_shapes(cfg):
    """DECLIP state_dict (model/declip.py:132-175): CLIP + projector + predictor (+ text_label_predictor)."""
    s = clip_shapes(cfg)
    fd, hid = cfg["embed_dim"], cfg.get("simsiam_hidden", 1024)
    s["projector.linear1.weight"], s["projector.linear1.bias"] = (hid, fd), (hid,)
    _bn_shapes(s, "projector.bn1.", hid)
    s["projector.linear2.weight"], s["projector.linear2.bias"] = (hid, hid), (hid,)
    _bn_shapes(s, "projector.bn2.", hid)
    s["projector.linear3.weight"], s["projector.linear3.bias"] = (hid, hid), (hid,)
    _bn_shapes(s, "projector.bn3.", hid)
    s["predictor.linear1.weight"], s["predictor.linear1.bias"] = (512, 1024), (512,)
    _bn_shapes(s, "predictor.bn1.", 512)
    s["predictor.layer2.weight"], s["predictor.layer2.bias"] = (1024, 512), (1024,)
    if cfg.get("mlm", True):
        s["text_label_predictor.weight"] = (cfg.get("vocab", VOCAB), cfg["t_width"])
        s["text_label_predictor.bias"] = (cfg.get("vocab", VOCAB),)
    return s


def slip_shapes(cfg):
    """SLIP state_dict (model/slip.py:209-217): CLIP (text prefix `text_encoder.`) + predictor_sim."""
    s = clip_shapes(dict(cfg, text_prefix="text_encoder."))
    fd, hid, sd = cfg["v_width"], 4096, cfg.get("sim_dim", 256)
    s["predictor_sim.linear1.weight"], s["predictor_sim.linear1.bias"] = (hid, fd), (hid,)
    _bn_shapes(s, "predictor_sim.bn1.", hid)
    s["predictor_sim.linear2.weight"], s["predictor_sim.linear2.bias"] = (hid, hid), (hid,)
    _bn_shapes(s, "predictor_sim.bn2.", hid)
    s["predictor_sim.linear3.weight"], s["predictor_sim.linear3.bias"] = (sd, hid), (sd,)
    _bn_shapes(s, "predictor_sim.bn3.", hid)
    return s


def filip_shapes(cfg):
    """FILIP state_dict (model/filip.py:27-59): CLIP + image/text mapping + logit_scale_dense + text_label_predictor."""
    s = clip_shapes(cfg)
    s["logit_scale_dense"] = ()
    s["image_mapping.weight"], s["image_mapping.bias"] = (256, cfg["v_width"]), (256,)
    s["text_mapping.weight"], s["text_mapping.bias"] = (256, cfg["t_width"]), (256,)
    s["text_label_predictor.weight"] = (cfg.get("vocab", VOCAB), cfg["t_width"])
    s["text_label_predictor.bias"] = (cfg.get("vocab", VOCAB),)
    return s


def defilip_shapes(cfg):
    """DEFILIP state_dict (model/defilip.py:149-207): DECLIP + logit_scale_dense + image/text mapping."""
    s = declip_shapes(cfg)
    s["logit_scale_dense"] = ()
    s["image_mapping.weight"], s["image_mapping.bias"] = (256, cfg["v_width"]), (256,)
    s["text_mapping.weight"], s["text_mapping.bias"] = (256, cfg["t_width"]), (256,)
    return s


def synth_bank(size, dim, seed=4):
    """unit-norm rows [size, dim] (NN bank randn seed 4, normalised, ptr 0)."""
    g = torch.Generator().manual_seed(4000 + seed)
    return torch.nn.functional.normalize(torch.randn(size, dim, generator=g), dim=1)


def synth_mlm(ids, vocab=VOCAB, seed=3):
    """deterministic MLM masking of padded id rows -> (masked_ids, labels)"""
    from .bpe import mask_token_ids
    g = torch.Generator().manual_seed(3000 + seed)
    return mask_token_ids(ids, vocab, generator=g)


VITB32 = dict(v_width=768, v_layers=12, v_heads=12, patch=32, res=224,
              t_width=512, t_layers=12, t_heads=8, ctx=77, embed_dim=512, vocab=VOCAB)
TINY = dict(v_width=128, v_layers=2, v_heads=2, patch=32, res=96,
            t_width=128, t_layers=2, t_heads=2, ctx=16, embed_dim=64, vocab=VOCAB)

# CLIP-R50 (BASELINE.json configs[0], experiments/clip_experiments/yfcc15m/yfcc15m_r50_clip/config.yaml) and a small
# ModifiedResNet that still ends on the 7x7 map the attention pool requires (modified_resnet.py:207)
R50 = dict(vision="resnet", r_width=64, r_layers=(3, 4, 6, 3), r_heads=32, res=224,
           t_width=512, t_layers=12, t_heads=8, ctx=77, embed_dim=1024, vocab=VOCAB)
R50_TINY = dict(vision="resnet", r_width=16, r_layers=(1, 2, 1, 1), r_heads=8, res=224,
                t_width=128, t_layers=2, t_heads=2, ctx=16, embed_dim=64, vocab=VOCAB)

# the other head of ModifiedResNet.forward (adaptive pool + fc, modified_resnet.py:209-211): any input whose final map is not 7 wide;
# fc is hard-wired to 2048 inputs, so the width must be 64
R50_FC = dict(vision="resnet", r_width=64, r_layers=(1, 1, 1, 1), r_heads=32, res=64,
              t_width=128, t_layers=2, t_heads=2, ctx=16, embed_dim=64, vocab=VOCAB)

# filip_res50: 49 dense tokens of width*32 channels; >= 16 text tokens (v_width = the dense image width the FILIP heads see)
R50_TINY_FILIP = dict(R50_TINY, ctx=24, v_width=16 * 32)

# the shipped FILIP ViT-B/32 (experiments/filip_experiments/yfcc15m/yfcc15m_vit_filip/config.yaml:5,13: embed_dim 768)
FILIP_VITB32 = dict(VITB32, embed_dim=768)

# FILIP needs >= 16 image tokens and >= 16 text tokens: 160 px / 32 = 25 patches, 24-token context
FILIP_SMALL = dict(v_width=128, v_layers=2, v_heads=2, patch=32, res=160,
                   t_width=128, t_layers=2, t_heads=2, ctx=24, embed_dim=64, vocab=VOCAB)


def synthetic_bpe_file(path):
    """A BPE merges file with the layout of the one the reference downloads (bpe_simple_vocab_16e6.txt.gz, dataset_prepare.md:36-37:
    header line + 48 894 merges -> vocabulary 49 409 with <|mask|>, simple_tokenizer.py:66-75) whose merges never apply to ordinary
    text: captions tokenise into their characters, so a caption's token count is its character count -- enough to drive the
    tokeniser at a realistic load when no real vocabulary is on the box (bench.py --pipeline)."""
    import gzip
    import os
    if not os.path.exists(path):
        n_merges = 49152 - 256 - 2
        lines = ["#version: synthetic"] + ["q%d z" % i for i in range(n_merges)] + [""]
        tmp = path + ".%d.tmp" % os.getpid()
        with gzip.open(tmp, "wb") as f:
            f.write("\n".join(lines).encode("utf-8"))
        os.replace(tmp, path)
    return path


def synth_decoded_batches(b, n_batches=6, src_hw=(256, 320), out_hw=(224, 224), seed=0, pinned=True):
    """`n_batches` host batches as a decoder + the reference's data pipeline would hand them over (imagenet_dataloader.py:36-47,
    clip_dataloader.py:47-53): decoded uint8 HWC images of varying SOURCE sizes on one (pinned) canvas, RandomResizedCrop boxes drawn on
    the host (declip_amd.augment), a mirror flag, and caption STRINGS (lengths ~ U{6..75} tokens with synthetic_bpe_file's vocabulary)."""
    import numpy as np
    import torch
    from . import augment
    g = np.random.default_rng(seed)
    alphabet = np.array(list("abcdefghijklmnopqrstuvwxyz"))
    out = []
    for _ in range(n_batches):
        hs = g.integers(src_hw[0] - 64, src_hw[0] + 1, size=b)
        ws = g.integers(src_hw[1] - 64, src_hw[1] + 1, size=b)
        canvas = torch.zeros(b, src_hw[0], src_hw[1], 3, dtype=torch.uint8, pin_memory=bool(pinned))
        canvas.copy_(torch.from_numpy(g.integers(0, 256, size=(b, src_hw[0], src_hw[1], 3), dtype=np.uint8)))
        sizes = [(int(h), int(w)) for h, w in zip(hs, ws)]
        boxes = torch.from_numpy(augment.random_resized_crop_params(sizes, out_hw, generator=g))
        flip = torch.from_numpy(g.integers(0, 2, size=b).astype(np.int32))
        caps = []
        for n in g.integers(4, 74, size=b):              # + SOT / EOT: 6 .. 75 tokens
            chars = alphabet[g.integers(0, 26, size=int(n))]
            words, i = [], 0
            while i < len(chars):                         # words of 2-8 letters (a space does not cost a token)
                k = int(g.integers(2, 9))
                words.append("".join(chars[i:i + k]))
                i += k
            caps.append([" ".join(words)])                # the reference's batches carry a LIST of captions per sample (clip.py:110-111)
        out.append({"images": canvas, "image_boxes": boxes, "image_flip": flip, "captions": caps})
    return out
