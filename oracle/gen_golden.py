"""Generate tests/golden/*.pt from the UNMODIFIED reference (build container only).

    PYTHONDONTWRITEBYTECODE=1 python -m oracle.gen_golden [--only clip_tiny ...]

Each fixture records the config + seeds (inputs/params are re-generated from
declip_amd.synth by the tests) and what the reference produced on CPU fp32:
loss, logits, normalised features, and per-parameter gradient digests
(L2 norm, first 8 elements, a seeded random projection).  TEST INFRASTRUCTURE.
"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from declip_amd import synth  # noqa: E402
from oracle import ref_harness  # noqa: E402

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")


def grad_digest(named_grads):
    out = {}
    for idx, (name, g) in enumerate(named_grads):
        if g is None:
            out[name] = None
            continue
        g = g.detach().double().flatten()
        gen = torch.Generator().manual_seed(4242 + idx)
        r = torch.randn(g.numel(), generator=gen, dtype=torch.float64)
        out[name] = dict(norm=float(g.norm()), head=g[:8].float().clone(),
                         proj=float((g * r).sum() / max(1.0, g.numel() ** 0.5)))
    return out


def logits_digest(l, label0=0):
    """compact record of a [b, B] logits matrix for the full-width fixtures (b >= 128: the matrix itself would be the bulk of the
    file): 48 x 48 corner, the label diagonal, row log-sum-exp, absolute maximum, a seeded projection."""
    l = l.detach().double()
    b, B = l.shape
    gen = torch.Generator().manual_seed(777)
    r = torch.randn(b, B, generator=gen, dtype=torch.float64)
    idx = torch.arange(b)
    return dict(shape=(b, B), corner=l[:48, :48].float().clone(), diag=l[idx, idx + label0].float().clone(),
                lse=torch.logsumexp(l, dim=1).float(), absmax=float(l.abs().max()), proj=float((l * r).sum() / (b * B) ** 0.5))


def build_ref_clip(ref, cfg, use_allgather):
    vt = ref.modules["prototype.model.image_encoder.visual_transformer"]
    tt = ref.modules["prototype.model.text_encoder.text_transformer"]
    if cfg.get("vision") == "resnet":
        # clip.py:148-155 (clip_res50) with the shipped BatchNorm switch (yfcc15m_r50_clip/config.yaml: use_sync_bn False)
        mr = ref.modules["prototype.model.image_encoder.modified_resnet"]
        vis = mr.ModifiedResNet(layers=tuple(cfg["r_layers"]), embed_dim=cfg["embed_dim"], heads=cfg["r_heads"],
                                input_resolution=cfg["res"], width=cfg["r_width"], use_sync_bn=False)
    else:
        vis = vt.VisualTransformer(input_resolution=cfg["res"], patch_size=cfg["patch"], width=cfg["v_width"],
                                   layers=cfg["v_layers"], heads=cfg["v_heads"], embed_dim=cfg["embed_dim"],
                                   checkpoint=False)
    txt = tt.TextTransformer(embed_dim=cfg["embed_dim"], context_length=cfg["ctx"],
                             transformer_width=cfg["t_width"], transformer_heads=cfg["t_heads"],
                             transformer_layers=cfg["t_layers"], positional_embedding_flag=True,
                             checkpoint=False, bpe_path=ref_harness.synthetic_bpe_path(),
                             text_encode_type="Transformer", text_model_utils={"random": False, "freeze": False})
    model = ref.modules["prototype.model.clip"].CLIP(vis, txt, use_allgather)
    return model


def patch_tokenize(text_module, ids_by_key):
    """text_transformer.py:144: replace BPE by a lookup of pre-tokenised ids.  The
    'captions' we feed are integer keys (row indices)."""
    def tokenize(texts, context_length=77, return_length=False, mask_type=None):
        rows = torch.stack([ids_by_key[int(t)] for t in texts])
        assert mask_type is None
        return rows
    text_module.tokenize = tokenize


def run_clip_rank(rank, world, cfg, b, seed, logit_scale, ret):
    """One reference rank: local rows [rank*b,(rank+1)*b) of the global batch."""
    import contextlib
    import io
    os.environ["SLURM_PROCID"] = str(rank)
    os.environ["SLURM_NTASKS"] = str(world)
    ref = ref_harness.load_reference()
    import torch.distributed as dist
    if world > 1:
        dist.init_process_group("gloo", rank=rank, world_size=world,
                                init_method="tcp://127.0.0.1:29541")
    torch.manual_seed(0)
    with contextlib.redirect_stdout(io.StringIO()):
        model = build_ref_clip(ref, cfg, use_allgather=(world > 1))
        sd = synth.synth_state(synth.clip_shapes(cfg), seed=seed, logit_scale=logit_scale)
        missing = model.load_state_dict(sd, strict=True)
        model.train()
    B = b * world
    images = synth.synth_images(B, res=cfg["res"], seed=seed)
    ids = synth.synth_tokens(B, ctx=cfg["ctx"], seed=seed, vocab=cfg["vocab"])
    sl = slice(rank * b, (rank + 1) * b)
    patch_tokenize(model.encode_text, {i: ids[i] for i in range(B)})
    batch = {"images": images[sl], "captions": [[i] for i in range(rank * b, (rank + 1) * b)]}
    logits_i, logits_t = model(batch)
    crit = ref.modules["prototype.loss_functions.loss"].ClipInfoCELoss()
    loss, labels = crit(logits_i, logits_t)
    loss = loss / world                                   # clip_solver.py:418
    loss.backward()
    grads = []
    for name, p in model.named_parameters():
        g = p.grad
        if g is not None and world > 1:
            dist.all_reduce(g)                            # utils/dist.py:71-74 (SUM)
        grads.append((name, g))
    total = loss.detach().clone()
    if world > 1:
        dist.all_reduce(total)
    if rank == 0:
        ret.update(loss=float(total), loss_rank0=float(loss.detach() * world), labels=labels.clone(), grads=grad_digest(grads))
        if b >= 128:
            ret.update(logits_i_digest=logits_digest(logits_i, int(labels[0])), logits_t_digest=logits_digest(logits_t, int(labels[0])))
        else:
            ret.update(logits_i=logits_i.detach().clone(), logits_t=logits_t.detach().clone())
        if cfg.get("vision") == "resnet":
            # BatchNorm side effects of the training forward, then the eval-mode tower (running statistics) on the same images
            bufs = dict(model.named_buffers())
            ret["bn_buffers"] = {k: bufs[k].detach().clone() for k in
                                 ("visual.bn1.running_mean", "visual.bn1.running_var", "visual.bn3.running_var",
                                  "visual.layer2.0.downsample.1.running_mean", "visual.layer4.0.bn3.running_var",
                                  "visual.bn2.num_batches_tracked")}
            model.eval()
            with torch.no_grad():
                feat, dense = model.visual(images[sl], return_dense=True)
            ret["eval_features"], ret["eval_dense_sum"] = feat.clone(), float(dense.double().sum())
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def _spawn_entry(rank, world, cfg, b, seed, logit_scale, path):
    ret = {}
    run_clip_rank(rank, world, cfg, b, seed, logit_scale, ret)
    if rank == 0:
        torch.save(ret, path)


def gen_clip(name, cfg, b, world=1, seed=0, logit_scale=None):
    if world == 1:
        ret = {}
        run_clip_rank(0, 1, cfg, b, seed, logit_scale, ret)
    else:
        import torch.multiprocessing as mp
        tmp = "/tmp/_golden_%s.pt" % name
        mp.spawn(_spawn_entry, args=(world, cfg, b, seed, logit_scale, tmp), nprocs=world, join=True)
        ret = torch.load(tmp)
        os.remove(tmp)
    ret.update(kind="clip", cfg=cfg, b=b, world=world, seed=seed, logit_scale=logit_scale,
               torch_version=torch.__version__)
    path = os.path.join(GOLDEN_DIR, name + ".pt")
    torch.save(ret, path)
    print("wrote %s  loss=%.6f  (%d KB)" % (path, ret["loss"], os.path.getsize(path) // 1024))


def run_declip_rank(rank, world, cfg, b, seed, nn_size, ret):
    """One reference DECLIP rank (model/declip.py) + the solver's loss composition (declip_solver.py:435-533, every term divided by
    world_size): local rows [rank*b, (rank+1)*b) of the global batch.  world > 1: the six gathered feature tensors (image x2, text,
    augmented text, NN text x2) span B = world*b columns, label0 = rank*b; every rank owns its NN bank (memory_bank.py:66), seeded
    per rank."""
    import contextlib
    import io
    os.environ["SLURM_PROCID"], os.environ["SLURM_NTASKS"] = str(rank), str(world)
    ref = ref_harness.load_reference()
    import torch.distributed as dist
    if world > 1:
        dist.init_process_group("gloo", rank=rank, world_size=world, init_method="tcp://127.0.0.1:29545")
        torch.set_num_threads(max(1, (os.cpu_count() or 2) // world))
    else:
        ref_harness.ensure_gloo_group()
    rd = ref.modules["prototype.model.declip"]
    vt = ref.modules["prototype.model.image_encoder.visual_transformer"]
    tt = ref.modules["prototype.model.text_encoder.text_transformer"]
    with contextlib.redirect_stdout(io.StringIO()):
        if cfg.get("vision") == "resnet":      # declip.py:339-346 (declip_res50) with the shipped BatchNorm switch
            mr = ref.modules["prototype.model.image_encoder.modified_resnet"]
            vis = mr.ModifiedResNet(layers=tuple(cfg["r_layers"]), embed_dim=cfg["embed_dim"], heads=cfg["r_heads"],
                                    input_resolution=cfg["res"], width=cfg["r_width"], use_sync_bn=False)
        else:
            vis = vt.VisualTransformer(input_resolution=cfg["res"], patch_size=cfg["patch"], width=cfg["v_width"],
                                       layers=cfg["v_layers"], heads=cfg["v_heads"], embed_dim=cfg["embed_dim"], checkpoint=False)
        txt = tt.TextTransformer(embed_dim=cfg["embed_dim"], context_length=cfg["ctx"], transformer_width=cfg["t_width"],
                                 transformer_heads=cfg["t_heads"], transformer_layers=cfg["t_layers"],
                                 positional_embedding_flag=True, checkpoint=False, bpe_path=ref_harness.synthetic_bpe_path(),
                                 text_encode_type="Transformer", text_model_utils={"random": False, "freeze": False})
        model = rd.DECLIP(vis, txt, True, nn_size=nn_size, nn_topk=1, return_nn_bank=True, text_mask_type="MLM",
                          EDA=True, feature_dim=cfg["embed_dim"])
        sd = synth.synth_state(synth.declip_shapes(cfg), seed=seed)
        model.load_state_dict(sd, strict=True)
        model.train()
    B = b * world
    images = synth.synth_images(B, views=2, res=cfg["res"], seed=seed)
    ids = synth.synth_tokens(B, ctx=cfg["ctx"], seed=seed, vocab=cfg["vocab"])
    ids_aug = synth.synth_tokens(B, ctx=cfg["ctx"], seed=seed + 50, vocab=cfg["vocab"])
    ids_masked, labels = synth.synth_mlm(ids, cfg["vocab"], seed=seed)
    bank = synth.synth_bank(nn_size, cfg["embed_dim"], seed=seed + rank)  # per-rank bank
    model.nn_replacer_text.bank = bank.t().clone()                       # reference layout [D, size]
    model.nn_replacer_text.bank_ptr = torch.LongTensor([0])
    off = ref_harness.AUG_KEY_OFFSET
    sl = slice(rank * b, (rank + 1) * b)

    def tokenize(texts, context_length=77, return_length=False, mask_type=None):
        keys = [int(t) for t in texts]
        if mask_type is not None:
            return torch.stack([ids_masked[k] for k in keys]), torch.stack([labels[k] for k in keys])
        return torch.stack([ids_aug[k - off] if k >= off else ids[k] for k in keys])
    model.encode_text.tokenize = tokenize
    out = model({"images": images[sl], "captions": [[i] for i in range(rank * b, (rank + 1) * b)]}, return_dict=True)
    L = ref.modules["prototype.loss_functions.loss"]
    crit, sim_crit = L.ClipInfoCELoss(), L.SimsiamLoss()
    ntx = ref.modules["prototype.loss_functions.nt_xent_ConVIRT"].NTXentLoss(b)
    li1, li2, lt1, lt2 = out["logits"]
    a1, a2, at1, at2 = out["logits_aug"]
    clip_loss = (crit(li1, lt1)[0] + crit(li2, lt2)[0] + crit(a1, at1)[0] + crit(a2, at2)[0]) / 4
    n1, n2, n1a, n2a = out["nn_text_logits"]
    nn_loss = (crit(n1, n1a)[0] + crit(n2, n2a)[0]) / 2
    p1, p2, z1, z2 = out["simsiam_features"]
    sim_loss = sim_crit(p1, z1, p2, z2)
    mlm = out["text_self_supervised"]
    tf, if1, if2 = out["features"]
    monitor = ntx(if1, tf) + ntx(if2, tf)
    total = (0.4 * clip_loss + 0.2 * sim_loss + 0.2 * mlm + 0.2 * nn_loss) / world   # yfcc15m_vit_declip/config.yaml:28-32; every term / world_size
    total.backward()
    grads = []
    for name, p in model.named_parameters():
        g = p.grad
        if g is not None and world > 1:
            dist.all_reduce(g)                                           # utils/dist.py:71-74 (SUM)
        grads.append((name, g))
    tot = total.detach().clone()
    parts = torch.tensor([float(clip_loss), float(nn_loss), float(sim_loss), float(mlm), float(monitor)], dtype=torch.float64)
    if world > 1:
        dist.all_reduce(tot)
        dist.all_reduce(parts)
        parts /= world
    if rank == 0:
        ret.update(loss=float(tot), parts=dict(clip=float(parts[0]), nn=float(parts[1]), simsiam=float(parts[2]), mlm=float(parts[3]),
                                                convirt=float(parts[4])),
                   logits_i1=li1.detach().clone(), nn_logits_i1=n1.detach().clone(), grads=grad_digest(grads),
                   bank_ptr=int(model.nn_replacer_text.bank_ptr), bank_sum=float(model.nn_replacer_text.bank.double().sum()),
                   bn1_running_mean=model.projector.bn1.running_mean.clone(), bn1_running_var=model.projector.bn1.running_var.clone())
        if cfg.get("vision") == "resnet":          # the tower saw two views: its BatchNorm buffers moved twice
            bufs = dict(model.named_buffers())
            ret["bn_buffers"] = {k: bufs[k].detach().clone() for k in ("visual.bn1.running_mean", "visual.layer4.0.bn3.running_var",
                                                                       "visual.bn2.num_batches_tracked")}
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def _spawn_declip(rank, world, cfg, b, seed, nn_size, path):
    ret = {}
    run_declip_rank(rank, world, cfg, b, seed, nn_size, ret)
    if rank == 0:
        torch.save(ret, path)


def gen_declip(name, cfg, b, seed=0, nn_size=256, world=1):
    """Reference DECLIP (model/declip.py) + the solver's loss composition (declip_solver.py:435-533)."""
    if world == 1:
        ret = {}
        run_declip_rank(0, 1, cfg, b, seed, nn_size, ret)
    else:
        import torch.multiprocessing as mp
        tmp = "/tmp/_golden_%s.pt" % name
        mp.spawn(_spawn_declip, args=(world, cfg, b, seed, nn_size, tmp), nprocs=world, join=True)
        ret = torch.load(tmp, weights_only=False)
        os.remove(tmp)
    ret.update(kind="declip", cfg=cfg, b=b, seed=seed, nn_size=nn_size, world=world, torch_version=torch.__version__)
    path = os.path.join(GOLDEN_DIR, name + ".pt")
    torch.save(ret, path)
    print("wrote %s  loss=%.6f parts=%s (%d KB)" % (path, ret["loss"], ret["parts"], os.path.getsize(path) // 1024))


def run_slip_rank(rank, world, cfg, b, seed, ret):
    """One reference SLIP rank (model/slip.py) + slip_solver.py loss composition: local rows [rank*b, (rank+1)*b) of the global
    batch; world > 1 exercises the gathered SimCLR features (NT_Xent_gather with positives at rank*b + i, nt_xent.py:64-83) and
    the / world_size of slip_solver.py:457,490."""
    import contextlib
    import io
    os.environ["SLURM_PROCID"], os.environ["SLURM_NTASKS"] = str(rank), str(world)
    ref = ref_harness.load_reference()
    import torch.distributed as dist
    if world > 1:
        dist.init_process_group("gloo", rank=rank, world_size=world, init_method="tcp://127.0.0.1:29544")
        torch.set_num_threads(max(1, (os.cpu_count() or 2) // world))
    else:
        ref_harness.ensure_gloo_group()
    rs = ref.modules["prototype.model.slip"]
    vt = ref.modules["prototype.model.image_encoder.visual_transformer"]
    tt = ref.modules["prototype.model.text_encoder.text_transformer"]
    with contextlib.redirect_stdout(io.StringIO()):
        vis = vt.VisualTransformer(input_resolution=cfg["res"], patch_size=cfg["patch"], width=cfg["v_width"],
                                   layers=cfg["v_layers"], heads=cfg["v_heads"], embed_dim=cfg["embed_dim"], checkpoint=False)
        txt = tt.TextTransformer(embed_dim=cfg["embed_dim"], context_length=cfg["ctx"], transformer_width=cfg["t_width"],
                                 transformer_heads=cfg["t_heads"], transformer_layers=cfg["t_layers"],
                                 positional_embedding_flag=True, checkpoint=False, bpe_path=ref_harness.synthetic_bpe_path(),
                                 text_encode_type="Transformer", text_model_utils={"random": False, "freeze": False})
        model = rs.SLIP(vis, txt, True, return_sim=True, feature_dim=cfg["v_width"], sim_dim=256)
        sd = synth.synth_state(synth.slip_shapes(cfg), seed=seed)
        model.load_state_dict(sd, strict=True)
        model.train()
    B = b * world
    images = synth.synth_images(B, views=3, res=cfg["res"], seed=seed)
    ids = synth.synth_tokens(B, ctx=cfg["ctx"], seed=seed, vocab=cfg["vocab"])
    sl = slice(rank * b, (rank + 1) * b)
    patch_tokenize(model.text_encoder, {i: ids[i] for i in range(B)})
    out = model({"images": images[sl], "captions": [[i] for i in range(rank * b, (rank + 1) * b)]}, return_dict=True)
    crit = ref.modules["prototype.loss_functions.loss"].ClipInfoCELoss()
    nx = ref.modules["prototype.loss_functions.nt_xent"]
    simclr_crit, mon_crit = nx.NT_Xent_gather(b), nx.NT_Xent(b)
    li, lt = out["logits"]
    clip_loss, _ = crit(li, lt)
    s1, g1, s2, g2 = out["sim_features"]
    simclr = simclr_crit(s1, g1, s2, g2)
    tf, imf = out["features"]
    monitor = mon_crit(imf, tf)
    total = (clip_loss + simclr) / world                                 # yfcc15m_vit_slip/config.yaml:28-30; slip_solver.py:457,490
    total.backward()
    grads = []
    for name, p in model.named_parameters():
        g = p.grad
        if g is not None and world > 1:
            dist.all_reduce(g)                                           # utils/dist.py:71-74 (SUM)
        grads.append((name, g))
    tot = total.detach().clone()
    parts = torch.tensor([float(clip_loss), float(simclr), float(monitor)], dtype=torch.float64)
    if world > 1:
        dist.all_reduce(tot)
        dist.all_reduce(parts)
        parts /= world
    if rank == 0:
        ret.update(loss=float(tot), parts=dict(clip=float(parts[0]), simclr=float(parts[1]), nt_xent=float(parts[2])),
                   grads=grad_digest(grads))
        ret.update(logits_i=li.detach().clone(), sim1=s1.detach().clone())
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def _spawn_slip(rank, world, cfg, b, seed, path):
    ret = {}
    run_slip_rank(rank, world, cfg, b, seed, ret)
    if rank == 0:
        torch.save(ret, path)


def gen_slip(name, cfg, b, seed=0, world=1):
    """Reference SLIP (model/slip.py) + slip_solver.py loss composition, one or two ranks."""
    if world == 1:
        ret = {}
        run_slip_rank(0, 1, cfg, b, seed, ret)
    else:
        import torch.multiprocessing as mp
        tmp = "/tmp/_golden_%s.pt" % name
        mp.spawn(_spawn_slip, args=(world, cfg, b, seed, tmp), nprocs=world, join=True)
        ret = torch.load(tmp, weights_only=False)
        os.remove(tmp)
    ret.update(kind="slip", cfg=cfg, b=b, seed=seed, world=world, torch_version=torch.__version__)
    path = os.path.join(GOLDEN_DIR, name + ".pt")
    torch.save(ret, path)
    print("wrote %s  loss=%.6f parts=%s (%d KB)" % (path, ret["loss"], ret["parts"], os.path.getsize(path) // 1024))


def run_filip_rank(rank, world, cfg, b, seed, ret):
    """One reference FILIP rank (model/filip.py) + filip_solver.py loss composition (clip 0.0, dense 1.0): local rows
    [rank*b, (rank+1)*b) of the global batch; world > 1 exercises B > b and label0 = rank*b != 0 in the dense logits."""
    import contextlib
    import io
    os.environ["SLURM_PROCID"], os.environ["SLURM_NTASKS"] = str(rank), str(world)
    ref = ref_harness.load_reference()
    import torch.distributed as dist
    if world > 1:
        dist.init_process_group("gloo", rank=rank, world_size=world, init_method="tcp://127.0.0.1:29543")
        torch.set_num_threads(max(1, (os.cpu_count() or 2) // world))
    else:
        ref_harness.ensure_gloo_group()
    rf = ref.modules["prototype.model.filip"]
    vt = ref.modules["prototype.model.image_encoder.visual_transformer"]
    tt = ref.modules["prototype.model.text_encoder.text_transformer"]
    with contextlib.redirect_stdout(io.StringIO()):
        if cfg.get("vision") == "resnet":      # filip.py:146-153 (filip_res50: dense tokens = the 7x7 map, width*32 channels)
            mr = ref.modules["prototype.model.image_encoder.modified_resnet"]
            vis = mr.ModifiedResNet(layers=tuple(cfg["r_layers"]), embed_dim=cfg["embed_dim"], heads=cfg["r_heads"],
                                    input_resolution=cfg["res"], width=cfg["r_width"], use_sync_bn=False)
        else:
            vis = vt.VisualTransformer(input_resolution=cfg["res"], patch_size=cfg["patch"], width=cfg["v_width"],
                                       layers=cfg["v_layers"], heads=cfg["v_heads"], embed_dim=cfg["embed_dim"], checkpoint=False)
        txt = tt.TextTransformer(embed_dim=cfg["embed_dim"], context_length=cfg["ctx"], transformer_width=cfg["t_width"],
                                 transformer_heads=cfg["t_heads"], transformer_layers=cfg["t_layers"],
                                 positional_embedding_flag=True, checkpoint=False, bpe_path=ref_harness.synthetic_bpe_path(),
                                 text_encode_type="Transformer", text_model_utils={"random": False, "freeze": False})
        model = rf.FILIP(vis, txt, True, text_mask_type="MLM", return_dense=True, select_topk=True, feature_dim=cfg["v_width"],
                         dense_mapping_image=cfg["v_width"], dense_mapping_language=cfg["t_width"])
        sd = synth.synth_state(synth.filip_shapes(cfg), seed=seed)
        model.load_state_dict(sd, strict=True)
        model.train()
    B = b * world
    images = synth.synth_images(B, views=2, res=cfg["res"], seed=seed)
    ids = synth.synth_tokens(B, ctx=cfg["ctx"], seed=seed, vocab=cfg["vocab"], min_len=cfg["ctx"] - 6)
    ids_masked, labels = synth.synth_mlm(ids, cfg["vocab"], seed=seed)
    sl = slice(rank * b, (rank + 1) * b)

    def tokenize(texts, context_length=77, return_length=False, mask_type=None):
        keys = [int(t) for t in texts]
        assert mask_type is not None
        return torch.stack([ids_masked[k] for k in keys]), torch.stack([labels[k] for k in keys])
    model.encode_text.tokenize = tokenize
    out = model({"images": images[sl], "captions": [[i] for i in range(rank * b, (rank + 1) * b)]}, return_dict=True)
    crit = ref.modules["prototype.loss_functions.loss"].ClipInfoCELoss()
    li, lt = out["logits"]
    dli, dlt = out["dense_logits"]
    clip_loss, _ = crit(li, lt)
    dense_loss, dlabels = crit(dli, dlt)
    total = (0.0 * clip_loss + 1.0 * dense_loss) / world                 # yfcc15m_vit_filip/config.yaml:32-37; filip_solver.py (/ world_size)
    total.backward()
    grads = []
    for name, p in model.named_parameters():
        g = p.grad
        if g is not None and world > 1:
            dist.all_reduce(g)                                           # utils/dist.py:71-74 (SUM)
        grads.append((name, g))
    tot = total.detach().clone()
    parts = torch.tensor([float(clip_loss), float(dense_loss)], dtype=torch.float64)
    if world > 1:
        dist.all_reduce(tot)
        dist.all_reduce(parts)
        parts /= world
    if rank == 0:
        ret.update(loss=float(tot), parts=dict(clip=float(parts[0]), dense=float(parts[1])), grads=grad_digest(grads))
        if b >= 128:
            ret.update(dense_logits_i_digest=logits_digest(dli, int(dlabels[0])), dense_logits_t_digest=logits_digest(dlt, int(dlabels[0])))
        else:
            ret.update(dense_logits_i=dli.detach().clone(), dense_logits_t=dlt.detach().clone())
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def _spawn_filip(rank, world, cfg, b, seed, path):
    ret = {}
    run_filip_rank(rank, world, cfg, b, seed, ret)
    if rank == 0:
        torch.save(ret, path)


def gen_filip(name, cfg, b, seed=0, world=1):
    if world == 1:
        ret = {}
        run_filip_rank(0, 1, cfg, b, seed, ret)
    else:
        import torch.multiprocessing as mp
        tmp = "/tmp/_golden_%s.pt" % name
        mp.spawn(_spawn_filip, args=(world, cfg, b, seed, tmp), nprocs=world, join=True)
        ret = torch.load(tmp, weights_only=False)
        os.remove(tmp)
    ret.update(kind="filip", cfg=cfg, b=b, seed=seed, world=world, torch_version=torch.__version__)
    path = os.path.join(GOLDEN_DIR, name + ".pt")
    torch.save(ret, path)
    print("wrote %s  loss=%.6f parts=%s (%d KB)" % (path, ret["loss"], ret["parts"], os.path.getsize(path) // 1024))


def run_defilip_rank(rank, world, cfg, b, seed, nn_size, ret):
    """One reference DEFILIP rank (model/defilip.py) + defilip_solver.py loss composition (declip weights + filip 0.2, every term
    divided by world_size): local rows [rank*b, (rank+1)*b) of the global batch; world > 1: the DeCLIP feature gathers AND the gathered
    top-16 token sets of the dense loss span B = world*b captions (label0 = rank*b); per-rank NN bank as in run_declip_rank."""
    import contextlib
    import io
    os.environ["SLURM_PROCID"], os.environ["SLURM_NTASKS"] = str(rank), str(world)
    ref = ref_harness.load_reference()
    import torch.distributed as dist
    if world > 1:
        dist.init_process_group("gloo", rank=rank, world_size=world, init_method="tcp://127.0.0.1:29546")
        torch.set_num_threads(max(1, (os.cpu_count() or 2) // world))
    else:
        ref_harness.ensure_gloo_group()
    rd = ref.modules["prototype.model.defilip"]
    vt = ref.modules["prototype.model.image_encoder.visual_transformer"]
    tt = ref.modules["prototype.model.text_encoder.text_transformer"]
    with contextlib.redirect_stdout(io.StringIO()):
        vis = vt.VisualTransformer(input_resolution=cfg["res"], patch_size=cfg["patch"], width=cfg["v_width"],
                                   layers=cfg["v_layers"], heads=cfg["v_heads"], embed_dim=cfg["embed_dim"], checkpoint=False)
        txt = tt.TextTransformer(embed_dim=cfg["embed_dim"], context_length=cfg["ctx"], transformer_width=cfg["t_width"],
                                 transformer_heads=cfg["t_heads"], transformer_layers=cfg["t_layers"],
                                 positional_embedding_flag=True, checkpoint=False, bpe_path=ref_harness.synthetic_bpe_path(),
                                 text_encode_type="Transformer", text_model_utils={"random": False, "freeze": False})
        model = rd.DEFILIP(vis, txt, True, nn_size=nn_size, nn_topk=1, return_nn_bank=True, text_mask_type="MLM", EDA=True,
                           feature_dim=cfg["embed_dim"], return_filip=True, dense_mapping_image=cfg["v_width"],
                           dense_mapping_language=cfg["t_width"])
        sd = synth.synth_state(synth.defilip_shapes(cfg), seed=seed)
        model.load_state_dict(sd, strict=True)
        model.train()
    B = b * world
    images = synth.synth_images(B, views=2, res=cfg["res"], seed=seed)
    ids = synth.synth_tokens(B, ctx=cfg["ctx"], seed=seed, vocab=cfg["vocab"], min_len=cfg["ctx"] - 6)
    ids_aug = synth.synth_tokens(B, ctx=cfg["ctx"], seed=seed + 50, vocab=cfg["vocab"], min_len=cfg["ctx"] - 6)
    ids_masked, labels = synth.synth_mlm(ids, cfg["vocab"], seed=seed)
    model.nn_replacer_text.bank = synth.synth_bank(nn_size, cfg["embed_dim"], seed=seed + rank).t().clone()   # per-rank bank
    model.nn_replacer_text.bank_ptr = torch.LongTensor([0])
    off = ref_harness.AUG_KEY_OFFSET
    sl = slice(rank * b, (rank + 1) * b)

    def tokenize(texts, context_length=77, return_length=False, mask_type=None):
        keys = [int(t) for t in texts]
        if mask_type is not None:
            return torch.stack([ids_masked[k] for k in keys]), torch.stack([labels[k] for k in keys])
        return torch.stack([ids_aug[k - off] if k >= off else ids[k] for k in keys])
    model.encode_text.tokenize = tokenize
    out = model({"images": images[sl], "captions": [[i] for i in range(rank * b, (rank + 1) * b)]}, return_dict=True)
    L = ref.modules["prototype.loss_functions.loss"]
    crit, sim_crit = L.ClipInfoCELoss(), L.SimsiamLoss()
    li1, li2, lt1, lt2 = out["logits"]
    a1, a2, at1, at2 = out["logits_aug"]
    clip_loss = (crit(li1, lt1)[0] + crit(li2, lt2)[0] + crit(a1, at1)[0] + crit(a2, at2)[0]) / 4
    n1, n2, n1a, n2a = out["nn_text_logits"]
    nn_loss = (crit(n1, n1a)[0] + crit(n2, n2a)[0]) / 2
    p1, p2, z1, z2 = out["simsiam_features"]
    sim_loss = sim_crit(p1, z1, p2, z2)
    mlm = out["text_self_supervised"]
    fi, ft = out["filip"]
    filip_loss = crit(fi, ft)[0]
    total = (0.4 * clip_loss + 0.2 * sim_loss + 0.2 * mlm + 0.2 * nn_loss + 0.2 * filip_loss) / world
    total.backward()
    grads = []
    for name, p in model.named_parameters():
        g = p.grad
        if g is not None and world > 1:
            dist.all_reduce(g)                                           # utils/dist.py:71-74 (SUM)
        grads.append((name, g))
    tot = total.detach().clone()
    parts = torch.tensor([float(clip_loss), float(nn_loss), float(sim_loss), float(mlm), float(filip_loss)], dtype=torch.float64)
    if world > 1:
        dist.all_reduce(tot)
        dist.all_reduce(parts)
        parts /= world
    if rank == 0:
        ret.update(loss=float(tot), parts=dict(clip=float(parts[0]), nn=float(parts[1]), simsiam=float(parts[2]), mlm=float(parts[3]),
                                                filip=float(parts[4])),
                   filip_i=fi.detach().clone(), grads=grad_digest(grads))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def _spawn_defilip(rank, world, cfg, b, seed, nn_size, path):
    ret = {}
    run_defilip_rank(rank, world, cfg, b, seed, nn_size, ret)
    if rank == 0:
        torch.save(ret, path)


def gen_defilip(name, cfg, b, seed=0, nn_size=256, world=1):
    """Reference DEFILIP (model/defilip.py) + defilip_solver.py loss composition, one or two ranks."""
    if world == 1:
        ret = {}
        run_defilip_rank(0, 1, cfg, b, seed, nn_size, ret)
    else:
        import torch.multiprocessing as mp
        tmp = "/tmp/_golden_%s.pt" % name
        mp.spawn(_spawn_defilip, args=(world, cfg, b, seed, nn_size, tmp), nprocs=world, join=True)
        ret = torch.load(tmp, weights_only=False)
        os.remove(tmp)
    ret.update(kind="defilip", cfg=cfg, b=b, seed=seed, nn_size=nn_size, world=world, torch_version=torch.__version__)
    path = os.path.join(GOLDEN_DIR, name + ".pt")
    torch.save(ret, path)
    print("wrote %s  loss=%.6f parts=%s (%d KB)" % (path, ret["loss"], ret["parts"], os.path.getsize(path) // 1024))


def gen_zeroshot(name, cfg, label_num, prompts_num, b, batches=2, seed=0):
    """The reference's own ClsSolver.evaluate (solver/clip_solver.py:675-737), called unbound on a stand-in solver object that
    carries exactly the attributes it reads (model.module, path.result_path, dist.rank, logger, fp16, config) and a stand-in
    val loader/dataset (get_label_texts / dump / evaluate).  Records scores + predictions per batch."""
    import contextlib
    import io
    import tempfile
    import types
    ref = ref_harness.load_reference()
    sys.modules.update(ref.modules)
    data_stub = types.ModuleType("prototype.data")           # prototype.data pulls torchvision (absent); evaluate() does not use it
    for n in ("build_imagenet_train_dataloader", "build_imagenet_test_dataloader", "build_clip_dataloader"):
        setattr(data_stub, n, None)
    sys.modules["prototype.data"] = data_stub
    sys.path.insert(0, ref_harness.REFERENCE_ROOT)
    try:
        import prototype.solver.clip_solver as ref_solver
    finally:
        sys.path.remove(ref_harness.REFERENCE_ROOT)
        for k in [k for k in sys.modules if k == "prototype" or k.startswith("prototype.") or k == "linklink" or k.startswith("linklink.")]:
            del sys.modules[k]
    ref_solver.broadcast_object = lambda obj, *a, **k: obj
    torch.manual_seed(0)
    with contextlib.redirect_stdout(io.StringIO()):
        model = build_ref_clip(ref, cfg, use_allgather=False)
        model.load_state_dict(synth.synth_state(synth.clip_shapes(cfg), seed=seed), strict=True)
    class_ids = synth.synth_tokens(label_num * prompts_num, ctx=cfg["ctx"], seed=seed + 77, vocab=cfg["vocab"], max_len=6)
    patch_tokenize(model.encode_text, {i: class_ids[i] for i in range(class_ids.shape[0])})
    dumped = []

    class Dataset:
        def get_label_texts(self):
            return list(range(label_num * prompts_num)), torch.eye(label_num)

        def dump(self, writer, batch):
            dumped.append(dict(prediction=batch["prediction"].clone(), score=batch["score"].clone()))

        def evaluate(self, res_file):
            return types.SimpleNamespace(metric={})

    class Loader:
        dataset = Dataset()

        def __iter__(self):
            for i in range(batches):
                yield {"images": synth.synth_images(b, res=cfg["res"], seed=seed * 1000 + i)}

    class Log:
        def info(self, *a):
            pass
        critical = info

    tmp = tempfile.mkdtemp()
    fake = types.SimpleNamespace(model=types.SimpleNamespace(module=model, eval=model.eval, train=model.train),
                                 path=types.SimpleNamespace(result_path=tmp), dist=types.SimpleNamespace(rank=0),
                                 logger=Log(), fp16=False, config={})
    ref_solver.ClsSolver.evaluate(fake, {"loader": Loader()})
    ret = dict(kind="zeroshot", cfg=cfg, label_num=label_num, prompts_num=prompts_num, b=b, batches=batches, seed=seed,
               scores=[d["score"] for d in dumped], predictions=[d["prediction"] for d in dumped],
               torch_version=torch.__version__)
    path = os.path.join(GOLDEN_DIR, name + ".pt")
    torch.save(ret, path)
    print("wrote %s  batches=%d preds0=%s (%d KB)" % (path, len(dumped), dumped[0]["prediction"].tolist(), os.path.getsize(path) // 1024))


FIXTURES = {
    "clip_tiny": lambda: gen_clip("clip_tiny", synth.TINY, b=4),
    "clip_tiny_scale5": lambda: gen_clip("clip_tiny_scale5", synth.TINY, b=4, seed=3, logit_scale=5.0),
    "clip_tiny_w2": lambda: gen_clip("clip_tiny_w2", synth.TINY, b=3, world=2, seed=5),
    "clip_vitb32_b8": lambda: gen_clip("clip_vitb32_b8", synth.VITB32, b=8, seed=1),
    "clip_r50_tiny": lambda: gen_clip("clip_r50_tiny", synth.R50_TINY, b=3, seed=9),
    "clip_r50_fc": lambda: gen_clip("clip_r50_fc", synth.R50_FC, b=3, seed=15),
    "clip_r50_tiny_w2": lambda: gen_clip("clip_r50_tiny_w2", synth.R50_TINY, b=2, world=2, seed=10),
    "declip_tiny": lambda: gen_declip("declip_tiny", synth.TINY, b=6, seed=2),
    "declip_r50_tiny": lambda: gen_declip("declip_r50_tiny", synth.R50_TINY, b=4, seed=12),
    "declip_tiny_w2": lambda: gen_declip("declip_tiny_w2", synth.TINY, b=4, seed=31, world=2),
    "filip_small_w2": lambda: gen_filip("filip_small_w2", synth.FILIP_SMALL, b=4, seed=32, world=2),
    "slip_tiny": lambda: gen_slip("slip_tiny", synth.TINY, b=5, seed=4),
    "slip_tiny_w2": lambda: gen_slip("slip_tiny_w2", synth.TINY, b=3, seed=33, world=2),
    "filip_small": lambda: gen_filip("filip_small", synth.FILIP_SMALL, b=5, seed=6),
    "filip_r50_tiny": lambda: gen_filip("filip_r50_tiny", synth.R50_TINY_FILIP, b=4, seed=13),
    "defilip_small": lambda: gen_defilip("defilip_small", synth.FILIP_SMALL, b=4, seed=7),
    "defilip_small_w2": lambda: gen_defilip("defilip_small_w2", synth.FILIP_SMALL, b=4, seed=34, world=2),
    # full-width fixtures at the smallest batches whose GEMMs are whole 256-row tiles, i.e. that run on the benchmarked
    # gemm_v4 kernel in bf16; minutes of CPU and up to ~40 GB each, logits kept as digests
    "clip_vitb32_b256": lambda: gen_clip("clip_vitb32_b256", synth.VITB32, b=256, seed=21),
    "declip_vitb32_b128": lambda: gen_declip("declip_vitb32_b128", synth.VITB32, b=128, seed=22, nn_size=4096),
    "slip_vitb32_b128": lambda: gen_slip("slip_vitb32_b128", synth.VITB32, b=128, seed=23),
    "filip_vitb32_e768_b256": lambda: gen_filip("filip_vitb32_e768_b256", synth.FILIP_VITB32, b=256, seed=24),
    "declip_vitb32_b128_w2": lambda: gen_declip("declip_vitb32_b128_w2", synth.VITB32, b=128, seed=26, nn_size=4096, world=2),
    "filip_vitb32_e768_b256_w2": lambda: gen_filip("filip_vitb32_e768_b256_w2", synth.FILIP_VITB32, b=256, seed=25, world=2),
    # round 4: the two families that had no fixture at shapes the benchmarked kernel takes
    "defilip_vitb32_b128": lambda: gen_defilip("defilip_vitb32_b128", synth.VITB32, b=128, seed=27, nn_size=4096),
    "slip_vitb32_b128_w2": lambda: gen_slip("slip_vitb32_b128_w2", synth.VITB32, b=128, seed=28, world=2),
    "zeroshot_tiny": lambda: gen_zeroshot("zeroshot_tiny", synth.TINY, label_num=7, prompts_num=3, b=5, batches=2, seed=8),
}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--only", nargs="*", default=None)
    args = ap.parse_args()
    os.makedirs(GOLDEN_DIR, exist_ok=True)
    names = args.only or list(FIXTURES)
    for n in names:
        FIXTURES[n]()


if __name__ == "__main__":
    main()
