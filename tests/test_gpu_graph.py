"""Whole-step HIP graph capture (declip_amd/graph.py): a captured forward + loss + backward replays to the same loss and
gradients as the eager step, follows the contents of its static input buffers, and composes with the fused AdamW outside it."""
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("cfg_name,b,dtype", [("TINY", 8, "fp32"), ("TINY", 8, "bf16"), ("R50_TINY", 4, "bf16")])
def test_graphed_step_equals_eager_step(cfg_name, b, dtype):
    from declip_amd import synth
    from declip_amd.graph import GraphedStep
    from declip_amd.loss import ClipInfoCELoss
    from declip_amd.optim import build_adamw
    from declip_amd.testing import build_clip
    cfg = getattr(synth, cfg_name)
    crit = ClipInfoCELoss()

    def make():
        model = build_clip(cfg, dtype=dtype, seed=3)
        opt = build_adamw(model, lr=1e-3, betas=(0.9, 0.98), weight_decay=0.1)
        images = synth.synth_images(b, res=cfg["res"], seed=0).cuda()
        ids = synth.synth_tokens(b, ctx=cfg["ctx"], seed=0, vocab=cfg["vocab"]).cuda()
        batch = {"images": images, "captions": ids}

        def fwd_bwd():
            li, lt = model(batch)
            loss, _ = crit(li, lt)
            loss.backward()
            return loss.detach()
        return model, opt, batch, fwd_bwd

    def feed(batch, step):
        batch["images"].copy_(synth.synth_images(b, res=cfg["res"], seed=step).cuda())
        # same token LENGTHS for every step (packed captions size their buffers by the row count): permute the captions of batch 0
        ids0 = synth.synth_tokens(b, ctx=cfg["ctx"], seed=0, vocab=cfg["vocab"])
        batch["captions"].copy_(ids0[torch.arange(b).roll(step)].cuda())
        # the row count of the packed layout is a HOST number (it sizes buffers): the data pipeline supplies it with the batch
        # (prefetch.py does, from the host copy); under graph capture a device read-back is not possible at all
        batch["captions"]._dh_rows = (batch["captions"]._version, int((ids0.argmax(dim=-1) + 1).sum()))

    losses = {}
    grads = {}
    for mode in ("eager", "graph"):
        model, opt, batch, fwd_bwd = make()
        stepper = GraphedStep(fwd_bwd, warmup=2, enabled=(mode == "graph"))
        ls = []
        for step in range(6):                    # graph mode: 2 eager warm-up steps, 1 capture, 3 replays
            feed(batch, step)
            opt.zero_grad()
            loss = stepper()
            ls.append(float(loss))
            opt.step()
        torch.cuda.synchronize()
        if mode == "graph":
            assert stepper.graph is not None
        losses[mode] = ls
        grads[mode] = model.__dict__["_flat_store"].flat_g.clone()
    tol = 1e-5 if dtype == "fp32" else 3e-3      # two bf16 runs differ by the float-atomic noise of the loss backward
    for a, c in zip(losses["graph"], losses["eager"]):
        assert abs(a - c) <= tol * abs(c), (losses["graph"], losses["eager"])
    ga, ge = grads["graph"], grads["eager"]
    # ViT: same arithmetic in both modes up to the float-atomic noise.  ModifiedResNet at batch 4: its gradient is discontinuous in
    # the ReLU masks (oracle/restated.py batch_norm2d; tests/test_gpu_resnet_intake_packed.py), so after six optimiser steps two
    # runs that differ by atomic ordering alone sit tens of per cent apart element-wise (measured 20 %) while the loss trajectory
    # agrees to 3e-3; what a capture bug produces (a stale input buffer, a dropped node) is an unrelated gradient: ||diff|| ~ 1.4 ||g||
    gtol = 0.35 if cfg.get("vision") == "resnet" else (1e-4 if dtype == "fp32" else 2e-2)
    assert float((ga - ge).norm()) <= gtol * float(ge.norm())
