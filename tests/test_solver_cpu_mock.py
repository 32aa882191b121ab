"""Solver loop on CPU with declip_amd.ops mocked (host logic only): config parsing, param groups (reference
param_group_all semantics), cosine/warm-up LR, logit_scale clamp, checkpoint layout and resume."""
import math
import os

import pytest
import torch

import cpu_ops_mock


@pytest.fixture()
def mocked(monkeypatch):
    from declip_amd import engine, ops
    for name in dir(cpu_ops_mock):
        if not name.startswith("_") and callable(getattr(cpu_ops_mock, name)) and hasattr(ops, name):
            monkeypatch.setattr(ops, name, getattr(cpu_ops_mock, name))
    monkeypatch.setattr(engine, "_require_gpu", lambda p, name: None)


def _config(kind="clip", max_iter=6):
    img = dict(embed_dim=32, layers=2, heads=2, width=64, input_resolution=32, patch_size=16)
    txt = dict(embed_dim=32, context_length=16, transformer_width=64, transformer_heads=2, transformer_layers=2,
               text_encode_type="Transformer", bpe_path=None, text_model_utils=dict(random=False, freeze=False), vocab_size=49409)
    clip = dict(use_allgather=False) if kind == "clip" else dict(use_allgather=True, return_sim=True, feature_dim=64, sim_dim=16)
    return dict(
        model=dict(type="%s_vitb32" % kind, kwargs=dict(image_encode=img, text_encode=txt, clip=clip, engine=dict(dtype="fp32"))),
        dist=dict(sync=False), grad_clip=dict(type="logit_scale_param_value", value=3, max_value=6),
        optimizer=dict(type="AdamW", kwargs=dict(lr=1e-4, weight_decay=0.1, betas=[0.9, 0.98], amsgrad=False, eps=1e-8),
                       pconfig={k: dict(weight_decay=0) for k in ("bn_w", "bn_b", "ln_w", "ln_b", "bias", "logit_scale")}),
        lr_scheduler=dict(type="Cosine", kwargs=dict(base_lr=1e-4, warmup_lr=1e-3, min_lr=0.0, warmup_steps=3, max_iter=max_iter)),
        data=dict(type="clip", read_from="fake", batch_size=4, input_size=32),
        saver=dict(print_freq=2, save_freq=3, pretrain=dict(auto_resume=True)))


def test_cosine_schedule_matches_reference_formula():
    from declip_amd.solver import CosineLRScheduler
    p = torch.nn.Parameter(torch.zeros(1))
    opt = torch.optim.SGD([dict(params=[p], lr=1e-4)], lr=1e-4)
    s = CosineLRScheduler(opt, max_iter=100, min_lr=0.0, base_lr=1e-4, warmup_lr=1e-3, warmup_steps=10)
    got = []
    for it in range(1, 101):
        s.step(it)
        got.append(s.get_lr()[0])
    # lr_scheduler/scheduler.py:68-84 (warm-up) and :226-236 (cosine)
    for it in (1, 5, 9):
        assert got[it - 1] == pytest.approx((1e-3 - 1e-4) / 9 * (it - 1) + 1e-4, rel=1e-6)
    for it in (10, 50, 100):
        assert got[it - 1] == pytest.approx(1e-3 * (1 + math.cos(math.pi * (it - 10) / 90)) / 2, rel=1e-6, abs=1e-12)


def test_param_groups_follow_reference_rules(mocked):
    from declip_amd.solver import AttrDict, param_groups
    from prototype.model import model_entry
    cfg = AttrDict(_config())
    model = model_entry(cfg.model)
    groups = param_groups(model, cfg.optimizer)
    names = {id(p): n for n, p in model.named_parameters()}
    all_ids = [id(p) for g in groups for p in g["params"]]
    assert len(all_ids) == len(set(all_ids)) == len(names)
    normal = {names[id(p)] for p in groups[0]["params"]}
    # MultiheadAttention.in_proj_* and embeddings stay in the default (decayed) group; out_proj.bias is a Linear bias
    assert "visual.transformer.resblocks.0.attn.in_proj_bias" in normal
    assert "visual.transformer.resblocks.0.attn.in_proj_weight" in normal
    assert "visual.class_embedding" in normal and "encode_text.token_embedding.weight" in normal
    assert "visual.transformer.resblocks.0.attn.out_proj.bias" not in normal
    assert "logit_scale" not in normal and "visual.ln_post.weight" not in normal
    # every typed group of misc.py:267-293 is emitted, also the empty ones (they carry optimizer defaults unless pconfig names them)
    assert len(groups) == 1 + 6 + 2                        # default + bn_w bn_b conv_b linear_b ln_w ln_b + logit_scale, bias
    for g in groups[1:]:
        assert g.get("weight_decay", None) == 0 or not g["params"]


@pytest.mark.parametrize("kind", ["clip", "slip"])
def test_solver_trains_saves_and_resumes(mocked, tmp_path, kind):
    import yaml
    from declip_amd.solver import ClsSolver
    cfgp = tmp_path / "config.yaml"
    cfgp.write_text(yaml.safe_dump(_config(kind)))
    s = ClsSolver(str(cfgp), device="cpu")
    l0 = None
    out = s.train(max_steps=4)
    assert torch.isfinite(out["loss"]).all()
    assert 3.0 <= float(s.model.module.logit_scale.detach()) <= 6.0            # clamp (clip_solver.py:507-508)
    ck = torch.load(os.path.join(str(tmp_path), "checkpoints", "ckpt.pth.tar"))
    assert ck["last_iter"] == 3 and all(k.startswith("module.") for k in ck["model"])
    s2 = ClsSolver(str(cfgp), device="cpu")                            # auto-resume at iter 3
    assert s2.state["last_iter"] == 3 and s2.optimizer.step_count == 3
    w1 = ck["model"]["module.visual.proj"]
    assert torch.equal(s2.model.module.visual.proj.detach().cpu(), w1)
    s2.train()
    assert s2.state["last_iter"] == 6


def test_solver_clip_res50_trains_saves_and_evaluates(mocked, tmp_path):
    """`type: clip_res50` through model_entry and the solver loop (yfcc15m_r50_clip/config.yaml keys; a narrow 4-block
    ModifiedResNet so that the host-emulated kernels stay fast): BatchNorm parameters land in bn_w / bn_b, the BatchNorm buffers
    travel through the checkpoint, zero-shot evaluate runs the tower in eval mode and restores training mode."""
    import yaml
    from declip_amd.solver import AttrDict, ClsSolver, param_groups
    cfg = _config("clip", max_iter=4)
    cfg["model"]["type"] = "clip_res50"
    cfg["model"]["kwargs"]["image_encode"] = dict(embed_dim=32, layers=[1, 1, 1, 1], width=16, heads=8, bn_group_size=32,
                                                  bn_sync_stats=True, use_sync_bn=False)
    cfg["data"].update(batch_size=2, input_size=224)
    cfg["data"]["test"] = dict(type="synthetic", label_num=4, prompts_num=2, batch_size=2, batches=1)
    cfg["saver"].update(save_freq=2)
    cfgp = tmp_path / "config.yaml"
    cfgp.write_text(yaml.safe_dump(cfg))
    s = ClsSolver(str(cfgp), device="cpu")
    model = s.model.module
    groups = param_groups(model, AttrDict(cfg).optimizer)
    names = {id(p): n for n, p in model.named_parameters()}
    normal = {names[id(p)] for p in groups[0]["params"]}
    assert "visual.layer1.0.conv2.weight" in normal and "visual.attnpool.positional_embedding" in normal
    assert "visual.bn1.weight" not in normal and "visual.layer3.0.downsample.1.bias" not in normal
    out = s.train(max_steps=2)
    assert torch.isfinite(out["loss"]).all()
    ck = torch.load(os.path.join(str(tmp_path), "checkpoints", "ckpt.pth.tar"))
    assert int(ck["model"]["module.visual.bn1.num_batches_tracked"]) == 2
    assert float(ck["model"]["module.visual.layer4.0.bn3.running_mean"].abs().max()) > 0
    m = s.evaluate()
    assert m["count"] == 2 and 0.0 <= m["top1"] <= 100.0
    assert s.model.training and int(model.visual.bn1.num_batches_tracked) == 2      # eval left the buffers alone


def test_solver_zero_shot_evaluate_synthetic(mocked, tmp_path):
    """--evaluate on the built-in synthetic set: forward only, metrics are chance-level but well formed, mode restored."""
    import yaml
    from declip_amd.solver import ClsSolver
    cfg = _config("clip")
    cfg["data"]["test"] = dict(type="synthetic", label_num=10, prompts_num=3, batch_size=6, batches=3)
    cfg["saver"]["pretrain"] = dict(auto_resume=False)
    cfgp = tmp_path / "config.yaml"
    cfgp.write_text(yaml.safe_dump(cfg))
    s = ClsSolver(str(cfgp), device="cpu")
    before = {k: v.clone() for k, v in s.model.module.state_dict().items()}
    m = s.evaluate()
    assert m["count"] == 18 and 0.0 <= m["top1"] <= m["top5"] <= 100.0 and m["images_per_s"] > 0
    assert s.model.training
    for k, v in s.model.module.state_dict().items():
        assert torch.equal(v, before[k]), k


def test_other_schedules_match_reference_formulas():
    """Step / StepDecay / Poly (lr_scheduler/scheduler.py:87-300) and the *Epoch variants (lr_scheduler/__init__.py:4-22)."""
    from declip_amd.solver import scheduler_entry

    def run(typ, **kw):
        p = torch.nn.Parameter(torch.zeros(1))
        opt = torch.optim.SGD([dict(params=[p], lr=0.1), dict(params=[torch.nn.Parameter(torch.zeros(1))], lr=0.05)], lr=0.1)
        s = scheduler_entry(dict(type=typ, kwargs=dict(optimizer=opt, base_lr=0.1, warmup_lr=0.4, warmup_steps=5, **kw)))
        out = []
        for it in range(1, 41):
            s.step(it)
            out.append(s.get_lr())
        return out
    st = run("Step", lr_steps=[10, 20], lr_mults=[0.1, 0.5], max_iter=40)
    assert st[2][0] == pytest.approx((0.4 - 0.1) / 4 * 2 + 0.1) and st[2][1] == pytest.approx(st[2][0] / 2)      # warm-up, second group scaled
    assert st[8][0] == pytest.approx(0.4) and st[9][0] == pytest.approx(0.04) and st[19][0] == pytest.approx(0.02)
    sd = run("StepDecay", step_size=10, decay=0.5, max_iter=40)
    assert sd[5][0] == pytest.approx(0.4) and sd[14][0] == pytest.approx(0.2) and sd[24][0] == pytest.approx(0.1)
    po = run("Poly", power=2.0, max_iter=40)
    assert po[19][0] == pytest.approx((1 - 15 / 40.0) ** 2 * 0.4)
    ep = scheduler_entry(dict(type="StepEpoch", kwargs=dict(optimizer=torch.optim.SGD([torch.nn.Parameter(torch.zeros(1))], lr=0.1),
                                                            base_lr=0.1, warmup_lr=0.4, warmup_epoch=1, lr_epochs=[2, 3], lr_mults=[0.1, 0.1],
                                                            max_iter=40, max_epoch=4)))
    assert ep.lr_steps == [20, 30] and ep.warmup_steps == 10
    with pytest.raises(NotImplementedError):
        scheduler_entry(dict(type="Exponential", kwargs={}))


def test_checkpoint_has_the_reference_layout_and_resumes_from_a_torch_adamw_state(mocked, tmp_path):
    """'model' / 'optimizer' / 'last_iter' (clip_solver.py:649-668) with the optimizer in torch.optim.AdamW's own layout: a
    checkpoint whose 'optimizer' was written by torch.optim.AdamW over the same param groups (what the reference saves) resumes
    with its moments and step count; saver.pretrain.path is honoured; the save is atomic (no temporary left behind)."""
    import yaml
    from declip_amd.solver import ClsSolver
    cfg = _config("clip")
    cfg["saver"].update(save_freq=2, pretrain=dict(auto_resume=False))
    cfgp = tmp_path / "config.yaml"
    cfgp.write_text(yaml.safe_dump(cfg))
    s = ClsSolver(str(cfgp), device="cpu")
    s.train(max_steps=2)
    ckdir = os.path.join(str(tmp_path), "checkpoints")
    assert sorted(os.listdir(ckdir)) == ["ckpt.pth.tar"]
    ck = torch.load(os.path.join(ckdir, "ckpt.pth.tar"), weights_only=False)
    assert set(ck) == {"model", "optimizer", "last_iter"} and ck["last_iter"] == 2
    osd = ck["optimizer"]
    assert set(osd) == {"state", "param_groups"} and all(set(v) == {"step", "exp_avg", "exp_avg_sq"} for v in osd["state"].values())
    # the same layout loads into a real torch.optim.AdamW over the same groups ...
    groups = [dict(params=list(g["params"]), **{k: v for k, v in g.items() if k != "params"}) for g in s.optimizer.param_groups]
    ref_opt = torch.optim.AdamW(groups, lr=1e-4)
    ref_opt.load_state_dict(osd)
    # ... and what torch.optim.AdamW writes resumes here (moments perturbed so that the copy is visible)
    tsd = ref_opt.state_dict()
    for v in tsd["state"].values():
        v["exp_avg"] = v["exp_avg"] + 1.0
    ck2 = dict(ck, optimizer=tsd)
    p2 = os.path.join(str(tmp_path), "zoo.pth.tar")
    torch.save(ck2, p2)
    cfg["saver"]["pretrain"] = dict(auto_resume=False, path=p2)
    cfgp.write_text(yaml.safe_dump(cfg))
    s2 = ClsSolver(str(cfgp), device="cpu")
    assert s2.state["last_iter"] == 2 and s2.optimizer.step_count == 2
    p = s2.optimizer.param_groups[0]["params"][0]
    o, n = s2.optimizer.flat.index[id(p)]
    assert torch.allclose(s2.optimizer.m[o:o + n].view(p.shape).cpu(), tsd["state"][0]["exp_avg"])
    # the LEGACY layout (checkpoints written before the empty typed groups were emitted: ADVICE r3) still resumes
    legacy = dict(state=tsd["state"], param_groups=[g for g in tsd["param_groups"] if len(g["params"]) > 0])
    assert len(legacy["param_groups"]) < len(tsd["param_groups"])
    torch.save(dict(ck, optimizer=legacy), p2)
    s3 = ClsSolver(str(cfgp), device="cpu")
    assert s3.optimizer.step_count == 2
    assert torch.allclose(s3.optimizer.m[o:o + n].view(p.shape).cpu(), tsd["state"][0]["exp_avg"])
    # a state that does not line up is refused, not dropped
    bad = dict(ck, optimizer=dict(state=tsd["state"], param_groups=tsd["param_groups"][:-1]))
    torch.save(bad, p2)
    with pytest.raises(ValueError):
        ClsSolver(str(cfgp), device="cpu")


@pytest.mark.parametrize("typ", ["norm", "value", "logit_scale_grad", "logit_scale_param", "logit_scale_param_abs_min", "constant"])
def test_grad_clip_types(mocked, tmp_path, typ):
    """every grad_clip.type of clip_solver.py:489-530 acts (round 1 honoured logit_scale_param_value only and dropped the rest)."""
    from declip_amd.solver import ClsSolver
    cfg = _config("clip")
    cfg["saver"] = dict(print_freq=100, save_freq=0, pretrain=dict(auto_resume=False))
    value = {"norm": 1e-3, "value": 1e-5, "logit_scale_grad": 1e-6, "logit_scale_param": 1e-6, "logit_scale_param_abs_min": 2.7, "constant": 0}[typ]
    cfg["grad_clip"] = dict(type=typ, value=value)
    s = ClsSolver(cfg, device="cpu")
    m = s.model.module
    flat = m.__dict__["_flat_store"]
    seen = {}
    orig_step = s.optimizer.step

    def spy(*a, **kw):
        seen["gnorm"] = float(flat.flat_g.norm())
        seen["gmax"] = float(flat.flat_g.abs().max())
        seen["dscale"] = None if m.logit_scale.grad is None else float(m.logit_scale.grad.abs().max())
        return orig_step(*a, **kw)
    s.optimizer.step = spy
    before = float(m.logit_scale.detach())
    s.train(max_steps=1)
    after = float(m.logit_scale.detach())
    if typ == "norm":
        assert seen["gnorm"] <= 1e-3 * 1.001
    elif typ == "value":
        assert seen["gmax"] <= 1e-5 * 1.001
    elif typ == "logit_scale_grad":
        assert seen["dscale"] <= 1e-6 * 1.001
    elif typ == "logit_scale_param":
        assert 0 < abs(after - before) <= 1e-6 * 1.01                  # AdamW's first step would move it by lr = 1e-4
    elif typ == "logit_scale_param_abs_min":
        assert after >= 2.7 - 1e-6
    else:
        assert after == before and not m.logit_scale.requires_grad


def test_optimizer_lr_is_base_lr_and_no_wd(mocked):
    """clip_solver.py:243 (optimizer.kwargs.lr := lr_scheduler.kwargs.base_lr) and :248-254 (optimizer.no_wd)."""
    from declip_amd.solver import ClsSolver
    cfg = _config("clip")
    cfg["saver"] = dict(print_freq=100, save_freq=0, pretrain=dict(auto_resume=False))
    cfg["optimizer"]["kwargs"]["lr"] = 0.5                                # contradicts base_lr 1e-4: must be overridden
    cfg["optimizer"].pop("pconfig")
    cfg["optimizer"]["no_wd"] = True
    s = ClsSolver(cfg, device="cpu")
    assert all(g["initial_lr"] == 1e-4 for g in s.optimizer.param_groups)
    assert s.optimizer.param_groups[0]["weight_decay"] == 0.1
    assert len(s.optimizer.param_groups) > 1 and all(g["weight_decay"] == 0.0 for g in s.optimizer.param_groups[1:])


def test_exhausted_loader_is_reported(mocked):
    from declip_amd.solver import ClsSolver
    cfg = _config("clip")
    cfg["saver"] = dict(print_freq=100, save_freq=0, pretrain=dict(auto_resume=False))
    cfg["data"]["prefetch"] = False
    probe = ClsSolver(cfg, device="cpu")
    one = probe.loader.get(1)
    s = ClsSolver(cfg, train_loader=[one], device="cpu")
    with pytest.raises(RuntimeError, match="exhausted"):
        s.train(max_steps=3)


@pytest.mark.parametrize("typ,kwargs,expect", [("FusedFP16AdamW", dict(lr=1e-4, betas=[0.9, 0.98], eps=1e-8, weight_decay=0.1), "FlatAdamW"),
                                               ("FusedFP16SGD", dict(lr=1e-3, momentum=0.9, weight_decay=1e-4, nesterov=True), "SGD"),
                                               ("AdamW", dict(lr=1e-4, amsgrad=True), "AdamW"), ("LARS", dict(lr=1.0), None)])
def test_reference_optimizer_names(mocked, typ, kwargs, expect):
    """optimizer/__init__.py:3-26: the reference's registry names.  FusedFP16SGD / FusedFP16AdamW are SGD / AdamW there when
    linklink.optim is absent; a torch optimizer (anything but the fused AdamW) steps the fp32 masters and the mirror is recast the
    step after (two training steps run, the loss moves); LARS / AdamW_SGD are refused by name."""
    from declip_amd.solver import ClsSolver
    cfg = _config("clip")
    cfg["saver"] = dict(print_freq=100, save_freq=0, pretrain=dict(auto_resume=False))
    cfg["optimizer"] = dict(type=typ, kwargs=dict(kwargs))
    if expect is None:
        with pytest.raises(NotImplementedError, match="LARS"):
            ClsSolver(cfg, device="cpu")
        return
    s = ClsSolver(cfg, device="cpu")
    assert type(s.optimizer).__name__ == expect
    before = [p.detach().clone() for p in s.model.module.parameters() if p.requires_grad][:4]
    s.train(max_steps=2)
    after = [p.detach() for p in s.model.module.parameters() if p.requires_grad][:4]
    assert any(not torch.equal(a, b) for a, b in zip(after, before))
