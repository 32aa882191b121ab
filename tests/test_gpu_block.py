"""dh_block_fwd / dh_block_bwd on the MI355X (csrc/block.hip; base_transformer.py:29-53): the CLIP ViT-B/32 step through one C-ABI
call per transformer block and direction against the same step composed from the per-op calls in Python -- same kernels on the same
data, so the forward (features, first loss) is bit-identical; gradients at the run-to-run noise floor of ONE mode (float atomics of
the InfoNCE backward, see test_clip_two_tower_streams_match_one_stream); and the host's share of an eager step shrinks."""
import time

import pytest
import torch

pytestmark = pytest.mark.gpu


def test_native_block_calls_match_the_per_op_composition_at_vitb32(monkeypatch):
    from declip_amd import synth
    from declip_amd.loss import ClipInfoCELoss
    from declip_amd.optim import build_adamw
    from declip_amd.testing import build_clip
    cfg, b, seed = synth.VITB32, 256, 5
    images = synth.synth_images(b, res=cfg["res"], seed=seed).cuda()
    ids = synth.synth_tokens(b, ctx=cfg["ctx"], seed=seed, vocab=cfg["vocab"]).cuda()

    def run(native):
        monkeypatch.setenv("DH_BLOCK_NATIVE", native)
        model = build_clip(cfg, dtype="bf16", seed=seed)
        opt = build_adamw(model, lr=1e-4, weight_decay=0.1)
        crit = ClipInfoCELoss()
        losses, first, host = [], None, []
        for i in range(4):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            li, lt = model({"images": images, "captions": ids})
            loss, _ = crit(li, lt)
            opt.zero_grad()
            loss.backward()
            host.append(time.perf_counter() - t0)
            if i == 0:
                torch.cuda.synchronize()
                first = {n: p.grad.detach().float().cpu() for n, p in model.named_parameters() if p.grad is not None}
                first["~img"], first["~txt"] = li.Q.detach().float().cpu(), lt.Q.detach().float().cpu()
            opt.step()
            losses.append(float(loss.detach()))
        torch.cuda.synchronize()
        return losses, first, min(host[1:])

    l1, g1, h1 = run("1")
    l0, g0, h0 = run("0")
    print("host enqueue time of forward + backward, b = 256: native blocks %.2f ms, per-op composition %.2f ms" % (h1 * 1e3, h0 * 1e3))
    assert l1[0] == l0[0] and torch.equal(g1["~img"], g0["~img"]) and torch.equal(g1["~txt"], g0["~txt"])
    for a, c in zip(l1, l0):
        assert abs(a - c) <= 3e-3 * abs(c)
    for n, g in g0.items():
        assert float((g1[n] - g).abs().max()) <= 1e-2 * float(g.abs().max()) + 1e-12, n
    assert h1 < h0
