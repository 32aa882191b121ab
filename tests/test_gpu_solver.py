"""Drop-in solver on the GPU: a short run through `prototype.solver.*_solver.ClsSolver` (HIP engine underneath)."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _config(kind):
    from test_solver_cpu_mock import _config as base
    cfg = base(kind, max_iter=12)
    cfg["model"]["kwargs"]["engine"] = dict(dtype="bf16")
    cfg["model"]["kwargs"]["image_encode"].update(width=128, heads=2, input_resolution=64)
    cfg["model"]["kwargs"]["text_encode"].update(transformer_width=128, context_length=32)
    cfg["data"].update(batch_size=32, input_size=64)
    cfg["saver"] = dict(print_freq=4, save_freq=0, pretrain=dict(auto_resume=False))
    if kind == "slip":
        cfg["model"]["kwargs"]["clip"]["feature_dim"] = 128
    if kind == "declip":
        cfg["model"]["kwargs"]["clip"] = dict(use_allgather=True, text_mask_type="MLM", return_nn_bank=True, feature_dim=32, nn_size=64)
    return cfg


@pytest.mark.parametrize("kind", ["clip", "slip", "declip"])
def test_solver_reduces_loss_on_a_fixed_batch(kind):
    import importlib
    mod = importlib.import_module("prototype.solver.%s_solver" % kind)
    torch.cuda.set_device(0)
    s = mod.ClsSolver(_config(kind))
    s.loader.n = 1                                                     # one repeated batch
    first = float(s.train(max_steps=1)["loss"].detach())
    last = float(s.train()["loss"].detach())
    assert last == last and last < first
    assert 3.0 <= float(s.model.module.logit_scale.detach()) <= 6.0
