"""CPU fp32 restatement of the reference hot path (TEST INFRASTRUCTURE ONLY).

Pure-functional torch (CPU, fp32) re-statement of what the reference computes on
the contrastive training path.  Every function cites the reference file:line it
follows (paths relative to /root/reference/prototype).  Parameters come in as a
flat ``sd`` dict with the reference's state_dict names, so the same dict can be
loaded into the reference modules (oracle/gen_golden.py) and into the HIP engine.

Pinned by tests/golden/*.pt (generated from the unmodified reference by
oracle/gen_golden.py; checked in tests/test_oracle_golden.py).
"""
import torch
import torch.nn.functional as F


# ----------------------------------------------------------------------------- blocks
def layer_norm(x, w, b, eps=1e-5):
    """model/image_encoder/base_transformer.py:10-18 (nn.LayerNorm, eps 1e-5)."""
    mu = x.mean(-1, keepdim=True)
    var = ((x - mu) ** 2).mean(-1, keepdim=True)
    return (x - mu) * torch.rsqrt(var + eps) * w + b


def quick_gelu(x):
    """base_transformer.py:24-26."""
    return x * torch.sigmoid(1.702 * x)


def attention(h, w_in, b_in, w_out, b_out, heads, causal):
    """base_transformer.py:33,45-48 -> nn.MultiheadAttention(x,x,x, attn_mask):
    packed in_proj rows ordered q,k,v; q scaled by head_dim**-0.5; additive -inf
    causal mask (text_transformer.py:136-142); softmax; out_proj.  h: [b,L,d]."""
    b, L, d = h.shape
    hd = d // heads
    qkv = h @ w_in.t() + b_in
    q, k, v = qkv.split(d, dim=-1)
    q = q.reshape(b, L, heads, hd).transpose(1, 2) * (hd ** -0.5)
    k = k.reshape(b, L, heads, hd).transpose(1, 2)
    v = v.reshape(b, L, heads, hd).transpose(1, 2)
    s = q @ k.transpose(-1, -2)
    if causal:
        s = s + torch.full((L, L), float("-inf")).triu_(1)
    p = torch.softmax(s, dim=-1)
    o = (p @ v).transpose(1, 2).reshape(b, L, d)
    return o @ w_out.t() + b_out


def residual_block(x, sd, p, heads, causal):
    """base_transformer.py:50-53."""
    h = layer_norm(x, sd[p + "ln_1.weight"], sd[p + "ln_1.bias"])
    x = x + attention(h, sd[p + "attn.in_proj_weight"], sd[p + "attn.in_proj_bias"],
                      sd[p + "attn.out_proj.weight"], sd[p + "attn.out_proj.bias"], heads, causal)
    h = layer_norm(x, sd[p + "ln_2.weight"], sd[p + "ln_2.bias"])
    u = h @ sd[p + "mlp.c_fc.weight"].t() + sd[p + "mlp.c_fc.bias"]
    return x + quick_gelu(u) @ sd[p + "mlp.c_proj.weight"].t() + sd[p + "mlp.c_proj.bias"]


def transformer(x, sd, prefix, layers, heads, causal):
    """base_transformer.py:56-79 (dropout 0, no checkpointing)."""
    for i in range(layers):
        x = residual_block(x, sd, "%sresblocks.%d." % (prefix, i), heads, causal)
    return x


# ----------------------------------------------------------------------------- towers
def patchify(images, patch):
    """im2row of the stride-P conv (visual_transformer.py:14-15,56-59):
    [b,3,H,W] -> [b, gh*gw, 3*P*P] with inner order (c, ph, pw) == conv1.weight.reshape(width,-1)."""
    b, c, H, W = images.shape
    gh, gw = H // patch, W // patch
    x = images.reshape(b, c, gh, patch, gw, patch).permute(0, 2, 4, 1, 3, 5)
    return x.reshape(b, gh * gw, c * patch * patch)


def vision_tower(images, sd, cfg, prefix="visual.", return_dense=False, return_feature=False):
    """visual_transformer.py:55-82."""
    w = sd[prefix + "conv1.weight"]
    x = patchify(images, cfg["patch"]) @ w.reshape(w.shape[0], -1).t()
    cls = sd[prefix + "class_embedding"].expand(x.shape[0], 1, -1)
    x = torch.cat([cls, x], dim=1) + sd[prefix + "positional_embedding"]
    x = layer_norm(x, sd[prefix + "ln_pre.weight"], sd[prefix + "ln_pre.bias"])
    x = transformer(x, sd, prefix + "transformer.", cfg["v_layers"], cfg["v_heads"], causal=False)
    dense = x[:, 1:, :]
    feat = layer_norm(x[:, 0, :], sd[prefix + "ln_post.weight"], sd[prefix + "ln_post.bias"])
    out = feat @ sd[prefix + "proj"]
    ret = [out]
    if return_dense:
        ret.append(dense)
    if return_feature:
        ret.append(feat)
    return ret[0] if len(ret) == 1 else tuple(ret)


def text_tower(ids, sd, cfg, prefix="encode_text.", return_dense=False):
    """text_transformer.py:183-204 with pre-tokenised ids [b,ctx] (tokenize() is
    patched out on the reference side, see gen_golden.py)."""
    x = sd[prefix + "token_embedding.weight"][ids] + sd[prefix + "positional_embedding"]
    x = transformer(x, sd, prefix + "transformer.", cfg["t_layers"], cfg["t_heads"], causal=True)
    x = layer_norm(x, sd[prefix + "ln_final.weight"], sd[prefix + "ln_final.bias"])
    pooled = x[torch.arange(x.shape[0]), ids.argmax(dim=-1)]
    out = pooled @ sd[prefix + "text_projection.weight"].t() + sd[prefix + "text_projection.bias"]
    return (out, x) if return_dense else out


def batch_norm2d(x, sd, p, training=True, eps=1e-5, momentum=0.1, new_stats=None):
    """nn.BatchNorm2d (modified_resnet.py:138 with use_sync_bn False) = F.batch_norm: batch statistics over (N, H, W) with
    the biased variance in training mode, running statistics in eval mode; in training the running buffers move by `momentum`
    towards the batch mean / UNBIASED variance (returned through `new_stats`; `sd` itself is never modified).
    torch's own kernel is called rather than a hand-written mean/var formula: the ResNet's gradients are discontinuous in the
    ReLU masks (one flipped mask moves a BatchNorm bias gradient -- a sum of cancelling terms -- by ~1 %), and a formula that
    rounds differently flips a mask somewhere in 2 M activations often enough to matter at the 3e-4 pin of the golden test."""
    if not training:
        return F.batch_norm(x, sd[p + "running_mean"], sd[p + "running_var"], sd[p + "weight"], sd[p + "bias"], False, momentum, eps)
    rm, rv = sd[p + "running_mean"].detach().clone(), sd[p + "running_var"].detach().clone()
    y = F.batch_norm(x, rm, rv, sd[p + "weight"], sd[p + "bias"], True, momentum, eps)
    if new_stats is not None:
        new_stats[p + "running_mean"], new_stats[p + "running_var"] = rm, rv
    return y


def bottleneck(x, sd, p, stride, training=True, new_stats=None):
    """modified_resnet.py:14-56: 1x1 -> 3x3 -> (avgpool) -> 1x1, every stride taken by an average pool; the shortcut is
    avgpool + 1x1 conv + BN whenever the shape changes."""
    out = F.relu(batch_norm2d(F.conv2d(x, sd[p + "conv1.weight"]), sd, p + "bn1.", training, new_stats=new_stats))
    out = F.relu(batch_norm2d(F.conv2d(out, sd[p + "conv2.weight"], padding=1), sd, p + "bn2.", training, new_stats=new_stats))
    if stride > 1:
        out = F.avg_pool2d(out, stride)
    out = batch_norm2d(F.conv2d(out, sd[p + "conv3.weight"]), sd, p + "bn3.", training, new_stats=new_stats)
    identity = x
    if (p + "downsample.0.weight") in sd:
        identity = F.avg_pool2d(x, stride) if stride > 1 else x
        identity = batch_norm2d(F.conv2d(identity, sd[p + "downsample.0.weight"]), sd, p + "downsample.1.", training, new_stats=new_stats)
    return F.relu(out + identity)


def attention_pool(x, sd, p, heads):
    """modified_resnet.py:59-96: tokens = [mean token; HW tokens] + positional embedding, multi-head attention with separate
    q/k/v projections, c_proj as the output projection; only the mean token's output is returned."""
    b, c = x.shape[0], x.shape[1]
    t = x.reshape(b, c, -1).permute(0, 2, 1)                       # [b, HW, C]
    t = torch.cat([t.mean(dim=1, keepdim=True), t], dim=1) + sd[p + "positional_embedding"]
    hd = c // heads
    L = t.shape[1]
    q = (t @ sd[p + "q_proj.weight"].t() + sd[p + "q_proj.bias"]).reshape(b, L, heads, hd).transpose(1, 2) * (hd ** -0.5)
    k = (t @ sd[p + "k_proj.weight"].t() + sd[p + "k_proj.bias"]).reshape(b, L, heads, hd).transpose(1, 2)
    v = (t @ sd[p + "v_proj.weight"].t() + sd[p + "v_proj.bias"]).reshape(b, L, heads, hd).transpose(1, 2)
    o = (torch.softmax(q @ k.transpose(-1, -2), dim=-1) @ v).transpose(1, 2).reshape(b, L, c)
    return o[:, 0, :] @ sd[p + "c_proj.weight"].t() + sd[p + "c_proj.bias"]


def resnet_tower(images, sd, cfg, prefix="visual.", return_dense=False, training=True, new_stats=None):
    """modified_resnet.py:193-214: 3-conv stem (first one stride 2) + avgpool(2), four bottleneck stages, attention pool on
    the 7x7 map; `dense` = the last feature map as [b, HW, C]."""
    x = images
    for i, (stride, name) in enumerate(((2, "1"), (1, "2"), (1, "3"))):
        x = F.conv2d(x, sd[prefix + "conv%s.weight" % name], stride=stride, padding=1)
        x = F.relu(batch_norm2d(x, sd, prefix + "bn%s." % name, training, new_stats=new_stats))
    x = F.avg_pool2d(x, 2)
    for li, blocks in enumerate(cfg["r_layers"]):
        for bi in range(blocks):
            x = bottleneck(x, sd, "%slayer%d.%d." % (prefix, li + 1, bi), 2 if (li > 0 and bi == 0) else 1, training, new_stats)
    dense = x.reshape(x.shape[0], x.shape[1], -1).permute(0, 2, 1)
    if x.shape[3] == 7:                                           # modified_resnet.py:207
        out = attention_pool(x, sd, prefix + "attnpool.", cfg["r_heads"])
    else:                                                         # :209-211 (`.squeeze()` drops the batch axis at b == 1)
        out = F.adaptive_avg_pool2d(x, (1, 1)).squeeze() @ sd[prefix + "fc.weight"].t() + sd[prefix + "fc.bias"]
    return (out, dense) if return_dense else out


def image_resized_crop_u8(src, params, out_hw, flip=None, mean=(0.485, 0.456, 0.406), std=(0.229, 0.224, 0.225), round_u8=True):
    """crop box -> antialiased bilinear resize -> output window -> mirror -> ToTensor -> Normalize, one image at a time: what
    torchvision's RandomResizedCrop / RandomCropMinSize / Resize + CenterCrop do on PIL images in the reference's workers
    (data/imagenet_dataloader.py:36-47,105-111; data/transforms.py:133-157), with PIL's triangle filter restated by torch's
    antialiased bilinear (same filter in float; PIL itself is compared in tests/test_hipemu_kernels.py).
    params [b, 8]: x0, y0, w, h, Wf, Hf, ox, oy."""
    H, W = out_hw
    b = src.shape[0]
    out = torch.empty(b, 3, H, W, dtype=torch.float32)
    m, s = torch.tensor(mean).view(3, 1, 1), torch.tensor(std).view(3, 1, 1)
    for i in range(b):
        x0, y0, w, h, Wf, Hf, ox, oy = [int(v) for v in params[i]]
        crop = src[i, y0:y0 + h, x0:x0 + w].permute(2, 0, 1).float()[None]
        r = F.interpolate(crop, size=(Hf, Wf), mode="bilinear", antialias=True, align_corners=False)[0, :, oy:oy + H, ox:ox + W]
        if flip is not None and bool(flip[i]):
            r = r.flip(2)
        if round_u8:
            r = (r + 0.5).floor().clamp(0, 255)
        out[i] = (r / 255.0 - m) / s
    return out


def image_tower(images, sd, cfg, **kw):
    """the image encoder the config names: visual_transformer.py (default) or modified_resnet.py (cfg["vision"] == "resnet")."""
    if cfg.get("vision") == "resnet":
        return resnet_tower(images, sd, cfg, **kw)
    return vision_tower(images, sd, cfg, **kw)


def image_prep_u8(src, out_hw, crop_xy=None, flip=None, mean=(0.485, 0.456, 0.406), std=(0.229, 0.224, 0.225)):
    """uint8 HWC -> normalised fp32 CHW: crop window (x0, y0), horizontal flip, ToTensor (x / 255), Normalize ((x - mean) / std)
    -- the tail of the reference's input pipelines (data/transforms.py + torchvision ToTensor / Normalize as composed in
    data/clip_dataloader.py; data/nvidia_dali_dataloader.py crop_mirror_normalize does the same in one op)."""
    b = src.shape[0]
    H, W = out_hw
    out = torch.empty(b, 3, H, W, dtype=torch.float32)
    m, s = torch.tensor(mean).view(3, 1, 1), torch.tensor(std).view(3, 1, 1)
    for i in range(b):
        x0, y0 = (int(crop_xy[i, 0]), int(crop_xy[i, 1])) if crop_xy is not None else (0, 0)
        win = src[i, y0:y0 + H, x0:x0 + W, :]
        if flip is not None and bool(flip[i]):
            win = win.flip(1)
        out[i] = (win.permute(2, 0, 1).float() / 255.0 - m) / s
    return out


# ----------------------------------------------------------------------------- CLIP
def clamp_scale(log_scale, clamp_max=100.0):
    """clip.py:133-134: exp() then a `.data` clamp -- the forward value is clamped,
    autograd still differentiates the unclamped exp (saved output)."""
    s = log_scale.exp()
    if clamp_max is None:
        return s
    return s + (s.clamp(max=clamp_max) - s).detach()


def normalize_features(img, txt):
    """clip.py:129-130 (image: no eps; text: +1e-10)."""
    return (img / img.norm(dim=-1, keepdim=True),
            txt / (txt.norm(dim=-1, keepdim=True) + 1e-10))


def clip_forward(images, ids, sd, cfg, world=1, new_stats=None):
    """clip.py:118-146 for `world` emulated ranks on one process.

    images [B,3,H,W], ids [B,ctx] hold the GLOBAL batch, rank r owning rows
    [r*b,(r+1)*b).  Returns per-rank (logits_per_image, logits_per_text), each
    [b,B] (local rows x gathered columns, clip.py:136-141)."""
    if cfg.get("vision") == "resnet":
        # BatchNorm statistics are per rank (use_sync_bn False): every emulated rank runs the tower on its own rows
        # (new_stats receives rank 0's buffer updates)
        bl = images.shape[0] // world
        img = torch.cat([resnet_tower(images[r * bl:(r + 1) * bl], sd, cfg, new_stats=new_stats if r == 0 else None)
                         for r in range(world)], dim=0)
    else:
        img = vision_tower(images, sd, cfg)
    txt = text_tower(ids, sd, cfg, prefix=cfg.get("text_prefix", "encode_text."))
    img, txt = normalize_features(img, txt)
    s = clamp_scale(sd["logit_scale"], cfg.get("scale_clamp", 100.0))
    B = img.shape[0]
    b = B // world
    out = []
    for r in range(world):
        sl = slice(r * b, (r + 1) * b)
        out.append((s * img[sl] @ txt.t(), s * txt[sl] @ img.t()))
    return out, (img, txt)


def zero_shot(images, class_ids, label_num, sd, cfg):
    """solver/clip_solver.py:687-719 (evaluate): per class, encode its prompts, normalise each, mean, normalise; per image,
    encode, normalise, logits = img @ class_emb^T (no logit scale), scores = softmax(logits) @ I, prediction = top-1.
    class_ids [label_num*prompts_num, ctx], class-major (clip_dataset.py:270-276).  One class at a time, as the reference does."""
    prompts_num = class_ids.shape[0] // label_num
    prefix = cfg.get("text_prefix", "encode_text.")
    embs = []
    for i in range(label_num):
        t = text_tower(class_ids[i * prompts_num:(i + 1) * prompts_num], sd, cfg, prefix=prefix)
        t = t / t.norm(dim=-1, keepdim=True)
        t = t.mean(dim=0)
        embs.append(t / t.norm())
    class_emb = torch.stack(embs, dim=0)
    img = vision_tower(images, sd, cfg)
    img = img / img.norm(dim=-1, keepdim=True)
    logits = img @ class_emb.t()
    scores = F.softmax(logits, dim=1) @ torch.eye(label_num, dtype=logits.dtype)
    return class_emb, logits, scores, logits.topk(k=1, dim=1)[1].view(-1)


def info_nce(logits_i, logits_t, rank=0):
    """loss_functions/loss.py:37-47."""
    bs, l_bs = logits_i.shape
    labels = torch.arange(bs) if l_bs == bs else rank * bs + torch.arange(bs)
    loss = (F.cross_entropy(logits_i, labels) + F.cross_entropy(logits_t, labels)) / 2
    return loss, labels


def accuracy(output, target, topk=(1, 5)):
    """utils/misc.py:415-428 (percent)."""
    maxk = min(max(topk), output.shape[1])
    _, pred = output.topk(maxk, 1, True, True)
    correct = pred.t().eq(target.view(1, -1))
    return [correct[:min(k, maxk)].reshape(-1).float().sum() * (100.0 / target.shape[0]) for k in topk]


def clip_step_loss(images, ids, sd, cfg, world=1, new_stats=None):
    """solver/clip_solver.py:413-430: per-rank loss/world summed over ranks is what
    the SUM-all-reduced gradients correspond to (quirk 14)."""
    per_rank, feats = clip_forward(images, ids, sd, cfg, world, new_stats=new_stats)
    total = 0.0
    metrics = []
    for r, (li, lt) in enumerate(per_rank):
        loss, labels = info_nce(li, lt, r)
        total = total + loss / world
        p1, p5 = accuracy(li.detach(), labels)
        metrics.append(dict(loss=loss.detach(), top1=p1, top5=p5))
    return total, per_rank, feats, metrics


# ----------------------------------------------------------------------------- DeCLIP
def bn_train(x, w, b, eps=1e-5):
    """nn.BatchNorm1d in training mode (batch statistics, biased variance) -- declip.py:49,54,60."""
    mu = x.mean(0, keepdim=True)
    var = ((x - mu) ** 2).mean(0, keepdim=True)
    return (x - mu) * torch.rsqrt(var + eps) * w + b


def projection_mlp(x, sd, p):
    """model/declip.py:33-90 (3 layers; bn3 on the output)."""
    x = torch.relu(bn_train(x @ sd[p + "linear1.weight"].t() + sd[p + "linear1.bias"], sd[p + "bn1.weight"], sd[p + "bn1.bias"]))
    x = torch.relu(bn_train(x @ sd[p + "linear2.weight"].t() + sd[p + "linear2.bias"], sd[p + "bn2.weight"], sd[p + "bn2.bias"]))
    return bn_train(x @ sd[p + "linear3.weight"].t() + sd[p + "linear3.bias"], sd[p + "bn3.weight"], sd[p + "bn3.bias"])


def prediction_mlp(x, sd, p):
    """model/declip.py:92-130."""
    x = torch.relu(bn_train(x @ sd[p + "linear1.weight"].t() + sd[p + "linear1.bias"], sd[p + "bn1.weight"], sd[p + "bn1.bias"]))
    return x @ sd[p + "layer2.weight"].t() + sd[p + "layer2.bias"]


def neg_cos(p, z):
    """loss_functions/loss.py:49-55 D(p, z)."""
    z = z.detach()
    p = p / p.norm(dim=-1, keepdim=True)
    z = z / z.norm(dim=-1, keepdim=True)
    return (p * z).sum(dim=1).mean()


def simsiam_loss(p1, z1, p2, z2):
    """loss_functions/loss.py:70-81."""
    return -0.5 * (neg_cos(p1, z2) + neg_cos(p2, z1))


def nn_lookup(q, bank):
    """nn_memory_bank.py:53-63 with topk=1; bank here is [size, D] (the reference stores [D, size])."""
    qn = F.normalize(q.detach(), dim=1)
    bn = F.normalize(bank, dim=1)
    idx = (qn @ bn.t()).argmax(dim=1)
    return bank[idx]


def bank_enqueue(bank, ptr, batch):
    """memory_bank.py:71-87 (tail dropped on wrap, pointer reset)."""
    b, size = batch.shape[0], bank.shape[0]
    if ptr + b >= size:
        bank[ptr:] = batch[:size - ptr].detach()
        return 0
    bank[ptr:ptr + b] = batch.detach()
    return ptr + b


def convirt_ntxent(zi, zj, temperature=0.1, alpha=0.75):
    """loss_functions/nt_xent_ConVIRT.py:23-86 (one-hot soft targets == CE on the local [b,b] logits)."""
    zi, zj = F.normalize(zi, dim=1), F.normalize(zj, dim=1)
    lab = torch.arange(zi.shape[0])
    return alpha * F.cross_entropy(zi @ zj.t() / temperature, lab) + (1 - alpha) * F.cross_entropy(zj @ zi.t() / temperature, lab)


def declip_step_loss(images, ids_masked, labels, ids_aug, sd, cfg, bank, bank_ptr=0,
                     weights=(0.4, 0.2, 0.2, 0.2)):
    """DECLIP.forward (model/declip.py:196-336) + the solver's loss composition
    (solver/declip_solver.py:435-533, image_text_two_view, weights clip/nn/simsiam/mlm), one rank.
    images [b,6,H,W]; returns total loss, parts dict, updated (bank, ptr)."""
    b = images.shape[0]
    tp = cfg.get("text_prefix", "encode_text.")
    img1 = image_tower(images[:, 0:3], sd, cfg)
    img2 = image_tower(images[:, 3:6], sd, cfg)
    txt, words = text_tower(ids_masked, sd, cfg, prefix=tp, return_dense=True)
    txt_aug = text_tower(ids_aug, sd, cfg, prefix=tp)
    z1, z2 = projection_mlp(img1, sd, "projector."), projection_mlp(img2, sd, "projector.")
    p1, p2 = prediction_mlp(z1, sd, "predictor."), prediction_mlp(z2, sd, "predictor.")
    i1, t = normalize_features(img1, txt)
    i2, t_aug = normalize_features(img2, txt_aug)
    s = clamp_scale(sd["logit_scale"], 100.0)
    ce = lambda q, k: F.cross_entropy(s * q @ k.t(), torch.arange(b))
    pair = lambda li_q, li_k, lt_q, lt_k: (ce(li_q, li_k) + ce(lt_q, lt_k)) / 2
    clip_loss = (pair(i1, t, t, i1) + pair(i2, t, t, i2) + pair(i1, t_aug, t_aug, i1) + pair(i2, t_aug, t_aug, i2)) / 4
    # NN supervision (declip.py:281-300): query(t), query(t_aug)+enqueue, enqueue(t)
    bank = bank.clone()
    nn_t = nn_lookup(t, bank)
    nn_t_aug = nn_lookup(t_aug, bank)
    ptr = bank_enqueue(bank, bank_ptr, t_aug)
    ptr = bank_enqueue(bank, ptr, t)
    nn_t = nn_t / (nn_t.norm(dim=-1, keepdim=True) + 1e-10)
    nn_t_aug = nn_t_aug / (nn_t_aug.norm(dim=-1, keepdim=True) + 1e-10)
    nn_loss = (pair(i1, nn_t, i1, nn_t_aug) + pair(i2, nn_t, i2, nn_t_aug)) / 2
    sim_loss = simsiam_loss(p1, z1, p2, z2)
    # MLM (declip.py:326-334)
    sel = labels != -100
    logits = words[sel] @ sd["text_label_predictor.weight"].t() + sd["text_label_predictor.bias"]
    mlm = F.cross_entropy(logits, labels[sel])
    monitor = convirt_ntxent(i1, t) + convirt_ntxent(i2, t)
    w_clip, w_nn, w_sim, w_mlm = weights
    total = w_clip * clip_loss + w_sim * sim_loss + w_mlm * mlm + w_nn * nn_loss
    parts = dict(clip=clip_loss.detach(), nn=nn_loss.detach(), simsiam=sim_loss.detach(), mlm=mlm.detach(),
                 convirt=monitor.detach())
    return total, parts, (bank, ptr)


# ----------------------------------------------------------------------------- SLIP
def simclr_mlp(x, sd, p):
    """model/slip.py:50-109 with out_bn=False (bn3 exists but is not applied)."""
    x = torch.relu(bn_train(x @ sd[p + "linear1.weight"].t() + sd[p + "linear1.bias"], sd[p + "bn1.weight"], sd[p + "bn1.bias"]))
    x = torch.relu(bn_train(x @ sd[p + "linear2.weight"].t() + sd[p + "linear2.bias"], sd[p + "bn2.weight"], sd[p + "bn2.bias"]))
    return x @ sd[p + "linear3.weight"].t() + sd[p + "linear3.bias"]


def nt_xent(z_i, z_j, temperature):
    """loss_functions/nt_xent.py:28-44 / :62-97 for one rank (gathered == local): every row's softmax runs over
    all 2b columns except itself; positives at i <-> i+b; CE(sum) / 2b."""
    b = z_i.shape[0]
    p = F.normalize(torch.cat([z_i, z_j]), dim=1, eps=1e-8)
    sim = p @ p.t() / temperature
    sim = sim.masked_fill(torch.eye(2 * b, dtype=torch.bool), float("-inf"))
    labels = torch.cat([torch.arange(b) + b, torch.arange(b)])
    return F.cross_entropy(sim, labels, reduction="sum") / (2 * b)


def slip_step_loss(images, ids, sd, cfg, weights=(1.0, 1.0)):
    """SLIP.forward (model/slip.py:245-286) + slip_solver.py:438-527, one rank.  images [b,9,H,W]."""
    b = images.shape[0]
    img = vision_tower(images[:, 0:3], sd, cfg)
    _, f1 = vision_tower(images[:, 3:6], sd, cfg, return_feature=True)
    _, f2 = vision_tower(images[:, 6:9], sd, cfg, return_feature=True)
    txt = text_tower(ids, sd, cfg, prefix="text_encoder.")
    s1, s2 = simclr_mlp(f1, sd, "predictor_sim."), simclr_mlp(f2, sd, "predictor_sim.")
    img_n, txt_n = normalize_features(img, txt)
    s = sd["logit_scale"].exp()                                        # no clamp (slip.py:265)
    lab = torch.arange(b)
    clip = (F.cross_entropy(s * img_n @ txt_n.t(), lab) + F.cross_entropy(s * txt_n @ img_n.t(), lab)) / 2
    simclr = nt_xent(s1, s2, 0.1)
    monitor = nt_xent(img_n, txt_n, 0.5)
    total = weights[0] * clip + weights[1] * simclr
    return total, dict(clip=clip.detach(), simclr=simclr.detach(), nt_xent=monitor.detach())


# ----------------------------------------------------------------------------- FILIP
def filip_dense_logits(d1, d2, log_scale_dense, top_k=16):
    """model/filip.py:71-106 (select_topk=True), one rank.  d1 [b,J,D] image tokens, d2 [b,T,D] text tokens."""
    d1 = d1 / d1.norm(dim=-1, keepdim=True)
    d2 = d2 / d2.norm(dim=-1, keepdim=True)
    s = log_scale_dense.exp()
    cross = d1 @ d2.transpose(1, 2)
    id1 = cross.sum(2).topk(top_k, dim=1)[1]
    id2 = cross.sum(1).topk(top_k, dim=1)[1]
    b = d1.shape[0]
    sel1 = d1[torch.arange(b)[:, None], id1]                              # [b,16,D]
    sel2 = d2[torch.arange(b)[:, None], id2]

    def logits(tok, sel):
        x = s * torch.einsum("ijk,lmk->iljm", tok, sel)                   # [b, B, J, 16]
        return x.max(dim=-1)[0].mean(dim=-1)
    return logits(d1, sel2), logits(d2, sel1)


def filip_step_loss(images, ids_masked, sd, cfg, weights=(0.0, 1.0)):
    """FILIP.forward (model/filip.py:109-142) + filip_solver.py:435-532, one rank; view 1 only."""
    b = images.shape[0]
    img, dense = image_tower(images[:, 0:3], sd, cfg, return_dense=True)
    txt, words = text_tower(ids_masked, sd, cfg, return_dense=True)
    img_n, txt_n = normalize_features(img, txt)
    s = sd["logit_scale"].exp()
    lab = torch.arange(b)
    clip = (F.cross_entropy(s * img_n @ txt_n.t(), lab) + F.cross_entropy(s * txt_n @ img_n.t(), lab)) / 2
    d1 = dense @ sd["image_mapping.weight"].t() + sd["image_mapping.bias"]
    d2 = words @ sd["text_mapping.weight"].t() + sd["text_mapping.bias"]
    li, lt = filip_dense_logits(d1, d2, sd["logit_scale_dense"])
    dense_loss = (F.cross_entropy(li, lab) + F.cross_entropy(lt, lab)) / 2
    total = weights[0] * clip + weights[1] * dense_loss
    return total, dict(clip=clip.detach(), dense=dense_loss.detach()), (li.detach(), lt.detach())
