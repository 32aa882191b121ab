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
    for g in groups[1:]:
        assert g.get("weight_decay", None) == 0


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
