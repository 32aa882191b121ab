"""Model-level parity of the HIP engine against (a) the golden fixtures generated from the unmodified
reference and (b) the CPU restatement on the same seeded inputs.  fp32 mode: 1e-3 (north_star);
bf16 mode: documented looser bounds."""
import pytest
import torch

from oracle_util import check_bf16_grad_directions, check_grad_digests, load_golden, oracle_clip_run

pytestmark = pytest.mark.gpu


def run_engine(cfg, b, seed, logit_scale, dtype, fused=True):
    from declip_amd import synth
    from declip_amd.loss import ClipInfoCELoss, accuracy
    from declip_amd.testing import build_clip
    model = build_clip(cfg, dtype=dtype, seed=seed, logit_scale=logit_scale, fused_loss=fused)
    images = synth.synth_images(b, res=cfg["res"], seed=seed).cuda()
    ids = synth.synth_tokens(b, ctx=cfg["ctx"], seed=seed, vocab=cfg["vocab"]).cuda()
    crit = ClipInfoCELoss()
    li, lt = model({"images": images, "captions": ids})
    loss, labels = crit(li, lt)
    p1, p5 = accuracy(li, labels, topk=(1, 5), criterion=crit)
    loss.backward()
    torch.cuda.synchronize()
    grads = {n: (p.grad.detach().float().cpu() if p.grad is not None else None) for n, p in model.named_parameters()}
    dense_i = (li.materialize() if hasattr(li, "materialize") else li).detach().float().cpu()
    dense_t = (lt.materialize() if hasattr(lt, "materialize") else lt).detach().float().cpu()
    return dict(loss=float(loss), logits_i=dense_i, logits_t=dense_t, grads=grads, top1=float(p1), top5=float(p5))


@pytest.mark.parametrize("name", ["clip_tiny", "clip_tiny_scale5", "clip_vitb32_b8"])
def test_clip_fp32_matches_reference_golden(name):
    g = load_golden(name)
    out = run_engine(g["cfg"], g["b"], g["seed"], g["logit_scale"], "fp32")
    assert abs(out["loss"] - g["loss"]) <= 1e-3 * abs(g["loss"])
    scale = float(g["logits_i"].abs().max())
    assert float((out["logits_i"] - g["logits_i"]).abs().max()) <= 1e-3 * scale
    assert float((out["logits_t"] - g["logits_t"]).abs().max()) <= 1e-3 * scale
    check_grad_digests(g["grads"], out["grads"], rtol=1e-3)


def test_clip_fp32_unfused_surface_matches_fused():
    g = load_golden("clip_tiny")
    a = run_engine(g["cfg"], g["b"], g["seed"], None, "fp32", fused=True)
    b = run_engine(g["cfg"], g["b"], g["seed"], None, "fp32", fused=False)
    assert abs(a["loss"] - b["loss"]) <= 1e-5 * abs(a["loss"])
    assert a["top1"] == b["top1"] and a["top5"] == b["top5"]
    check = [n for n in a["grads"] if a["grads"][n] is not None]
    for n in check:
        ga, gb = a["grads"][n], b["grads"][n]
        assert float((ga - gb).abs().max()) <= 1e-4 * float(ga.abs().max() + 1e-12), n


@pytest.mark.parametrize("name,loss_tol,gnorm_tol", [("clip_tiny", 1e-2, 5e-2), ("clip_vitb32_b8", 1e-2, 8e-2)])
def test_clip_bf16_close_to_reference(name, loss_tol, gnorm_tol):
    """bf16 throughput mode: loss within 1e-2 relative, per-parameter gradient norms within a few %."""
    g = load_golden(name)
    out = run_engine(g["cfg"], g["b"], g["seed"], g["logit_scale"], "bf16")
    assert abs(out["loss"] - g["loss"]) <= loss_tol * abs(g["loss"])
    scale = float(g["logits_i"].abs().max())
    assert float((out["logits_i"] - g["logits_i"]).abs().max()) <= 3e-2 * scale
    bad = []
    gmax = max(v["norm"] for v in g["grads"].values() if v is not None)
    for n, ref in g["grads"].items():
        if ref is None or ref["norm"] < 1e-3 * gmax:
            continue
        got = float(out["grads"][n].double().norm())
        if abs(got - ref["norm"]) > gnorm_tol * ref["norm"]:
            bad.append((n, got, ref["norm"]))
    assert len(bad) <= max(1, len(g["grads"]) // 50), bad[:8]
    # direction, not only size: the fixture's seeded projection of every gradient as a z-score of its relative error
    # (measured on the MI355X, round 3: rms z 0.022 / 0.029, worst |z| 0.058 / 0.085 for clip_tiny / clip_vitb32_b8)
    check_bf16_grad_directions(g["grads"], out["grads"], rms_tol=0.06, z_tol=0.25)


def test_clip_accuracy_matches_oracle():
    g = load_golden("clip_tiny")
    out = run_engine(g["cfg"], g["b"], g["seed"], None, "fp32")
    ref = oracle_clip_run(g["cfg"], g["b"], 1, g["seed"], None)
    assert abs(out["top1"] - float(ref["metrics"][0]["top1"])) < 1e-4
    assert abs(out["top5"] - float(ref["metrics"][0]["top5"])) < 1e-4


def test_train_steps_flat_adamw_matches_torch_adamw_on_oracle():
    """3 optimiser steps of the engine (fp32) vs the CPU restatement + torch.optim.AdamW."""
    from declip_amd import synth
    from declip_amd.loss import ClipInfoCELoss
    from declip_amd.optim import build_adamw
    from declip_amd.testing import build_clip
    from oracle import restated
    cfg, b, seed = synth.TINY, 4, 11
    model = build_clip(cfg, dtype="fp32", seed=seed)
    opt = build_adamw(model, lr=1e-3, betas=(0.9, 0.98), eps=1e-8, weight_decay=0.1)
    crit = ClipInfoCELoss()
    sd = synth.synth_state(synth.clip_shapes(cfg), seed=seed)
    names_decay = {n for n, p in model.named_parameters() if p.dim() > 1 and "logit_scale" not in n and not n.endswith("bias")}
    for k, v in sd.items():
        v.requires_grad_(k != "visual.conv1.weight")
    train = [k for k in sd if k != "visual.conv1.weight"]
    ref_opt = torch.optim.AdamW([dict(params=[sd[k] for k in train if k in names_decay], weight_decay=0.1),
                                 dict(params=[sd[k] for k in train if k not in names_decay], weight_decay=0.0)],
                                lr=1e-3, betas=(0.9, 0.98), eps=1e-8)
    losses, ref_losses = [], []
    for step in range(3):
        images = synth.synth_images(b, res=cfg["res"], seed=seed + step)
        ids = synth.synth_tokens(b, ctx=cfg["ctx"], seed=seed + step, vocab=cfg["vocab"])
        opt.zero_grad()
        li, lt = model({"images": images.cuda(), "captions": ids.cuda()})
        loss, _ = crit(li, lt)
        loss.backward()
        opt.step()
        losses.append(float(loss))
        ref_opt.zero_grad()
        total, _, _, _ = restated.clip_step_loss(images, ids, sd, cfg, 1)
        total.backward()
        ref_opt.step()
        ref_losses.append(float(total))
    for a, r in zip(losses, ref_losses):
        assert abs(a - r) <= 1e-3 * abs(r), (losses, ref_losses)
    got = {n: p.detach().cpu() for n, p in model.named_parameters()}
    for k in train:
        a, r = got[k].detach().cpu(), sd[k].detach()
        # Adam turns rounding noise on (near-)zero gradients into +-lr steps (e.g. the key bias, whose true
        # gradient is exactly 0): bound every element by Adam's max step and require the bulk to agree tightly.
        diff = (a - r).abs()
        assert float(diff.max()) <= 3 * 1e-3 * 3 + 2e-3 * float(r.abs().max()), k
        assert float((diff > 2e-3 * float(r.abs().max() + 1e-12)).float().mean()) <= 0.02 or k.endswith("in_proj_bias"), k


@pytest.mark.parametrize("dtype,tol", [("fp32", 1e-3), ("bf16", 2e-2)])
def test_declip_step_matches_reference_golden(dtype, tol):
    """DECLIP model + declip_solver loss composition on the HIP engine vs the reference golden."""
    from declip_amd.heads import SimsiamLoss
    from declip_amd.loss import ClipInfoCELoss, NTXentLoss
    from declip_amd.steps import declip_loss
    from declip_amd.testing import build_declip, declip_batch
    g = load_golden("declip_tiny")
    cfg, b, seed = g["cfg"], g["b"], g["seed"]
    model = build_declip(cfg, dtype=dtype, seed=seed, nn_size=g["nn_size"])
    batch = declip_batch(cfg, b, seed=seed)
    out = declip_loss(model, batch, ClipInfoCELoss(), SimsiamLoss(), NTXentLoss(b))
    out["loss"].backward()
    torch.cuda.synchronize()
    assert abs(float(out["loss"]) - g["loss"]) <= tol * abs(g["loss"])
    for k in ("clip", "nn", "mlm", "convirt"):
        assert abs(float(out["parts"][k]) - g["parts"][k]) <= tol * max(1.0, abs(g["parts"][k])), k
    assert abs(float(out["parts"]["simsiam"]) - g["parts"]["simsiam"]) <= (1e-4 if dtype == "fp32" else 2e-2)
    assert model.nn_replacer_text.bank_ptr == g["bank_ptr"]
    if dtype == "fp32":
        li1 = out["outputs"]["logits"][0].materialize().detach().cpu()
        assert float((li1 - g["logits_i1"]).abs().max()) <= 1e-3 * float(g["logits_i1"].abs().max())
        grads = {n: (p.grad.detach().float().cpu() if p.grad is not None else None) for n, p in model.named_parameters()}
        check_grad_digests(g["grads"], grads, rtol=1e-3)
        assert abs(float(model.nn_replacer_text.bank.double().sum()) - g["bank_sum"]) <= 1e-3
        assert torch.allclose(model.projector.bn1.running_mean.cpu(), g["bn1_running_mean"], rtol=1e-3, atol=1e-5)


@pytest.mark.parametrize("dtype,tol", [("fp32", 1e-3), ("bf16", 2e-2)])
def test_slip_step_matches_reference_golden(dtype, tol):
    from declip_amd.loss import ClipInfoCELoss, NT_Xent, NT_Xent_gather
    from declip_amd.steps import slip_loss
    from declip_amd.testing import build_slip, slip_batch
    g = load_golden("slip_tiny")
    cfg, b, seed = g["cfg"], g["b"], g["seed"]
    model = build_slip(cfg, dtype=dtype, seed=seed)
    out = slip_loss(model, slip_batch(cfg, b, seed=seed), ClipInfoCELoss(), NT_Xent_gather(b), NT_Xent(b))
    out["loss"].backward()
    torch.cuda.synchronize()
    assert abs(float(out["loss"]) - g["loss"]) <= tol * abs(g["loss"])
    for k in ("clip", "simclr", "nt_xent"):
        assert abs(float(out["parts"][k]) - g["parts"][k]) <= tol * max(1.0, abs(g["parts"][k])), k
    if dtype == "fp32":
        grads = {n: (p.grad.detach().float().cpu() if p.grad is not None else None) for n, p in model.named_parameters()}
        check_grad_digests(g["grads"], grads, rtol=1e-3)


@pytest.mark.parametrize("dtype,tol", [("fp32", 1e-3), ("bf16", 3e-2)])
def test_filip_step_matches_reference_golden(dtype, tol):
    from declip_amd.loss import ClipInfoCELoss
    from declip_amd.steps import filip_loss
    from declip_amd.testing import build_filip, filip_batch
    g = load_golden("filip_small")
    cfg, b, seed = g["cfg"], g["b"], g["seed"]
    model = build_filip(cfg, dtype=dtype, seed=seed)
    out = filip_loss(model, filip_batch(cfg, b, seed=seed), ClipInfoCELoss())
    out["loss"].backward()
    torch.cuda.synchronize()
    assert abs(float(out["loss"]) - g["loss"]) <= tol * abs(g["loss"])
    dli = out["outputs"]["dense_logits"][0].detach().cpu()
    assert float((dli - g["dense_logits_i"]).abs().max()) <= tol * float(g["dense_logits_i"].abs().max())
    if dtype == "fp32":
        grads = {n: (p.grad.detach().float().cpu() if p.grad is not None else None) for n, p in model.named_parameters()}
        check_grad_digests(g["grads"], grads, rtol=1e-3)


@pytest.mark.parametrize("dtype,tol", [("fp32", 1e-3), ("bf16", 3e-2)])
def test_defilip_step_matches_reference_golden(dtype, tol):
    from declip_amd.heads import SimsiamLoss
    from declip_amd.loss import ClipInfoCELoss
    from declip_amd.steps import DEFILIP_WEIGHTS, declip_loss
    from declip_amd.testing import build_defilip, defilip_batch
    g = load_golden("defilip_small")
    cfg, b, seed = g["cfg"], g["b"], g["seed"]
    model = build_defilip(cfg, dtype=dtype, seed=seed, nn_size=g["nn_size"])
    out = declip_loss(model, defilip_batch(cfg, b, seed=seed), ClipInfoCELoss(), SimsiamLoss(), None, weights=DEFILIP_WEIGHTS)
    out["loss"].backward()
    torch.cuda.synchronize()
    assert abs(float(out["loss"]) - g["loss"]) <= (tol if dtype == "fp32" else 0.05) * abs(g["loss"])
    for k in ("clip", "nn", "mlm", "filip"):
        # the nearest-neighbour lookup is a discrete choice: with bf16 towers a near-tie in the bank can resolve to a
        # different row than the fp32 reference picks, so that term is only bounded loosely outside fp32
        t = 0.15 if (k == "nn" and dtype != "fp32") else tol
        assert abs(float(out["parts"][k]) - g["parts"][k]) <= t * max(1.0, abs(g["parts"][k])), k
    if dtype == "fp32":
        grads = {n: (p.grad.detach().float().cpu() if p.grad is not None else None) for n, p in model.named_parameters()}
        check_grad_digests(g["grads"], grads, rtol=1e-3)


def test_clip_bf16_vitb32_b256_v4_gemm_matches_v2_gemm():
    """The golden fixtures use b = 8 (400 / 616 token rows: not whole 256-tiles), so they exercise the 128 x 128 kernels.  At
    b = 256 every tower GEMM runs on the persistent 256 x 256 kernel (tail slicing, split-K workspace, fused bias gradient
    included).  One full ViT-B/32 step through both GEMM families must agree to bf16 accuracy."""
    from declip_amd import lib, synth
    L = lib.load()
    cfg, b, seed = synth.VITB32, 256, 3
    prev = L.dh_gemm_v4_enable(1)
    try:
        a = run_engine(cfg, b, seed, None, "bf16")
        L.dh_gemm_v4_enable(0)
        ref = run_engine(cfg, b, seed, None, "bf16")
    finally:
        L.dh_gemm_v4_enable(prev)
    assert abs(a["loss"] - ref["loss"]) <= 2e-3 * abs(ref["loss"])
    scale = float(ref["logits_i"].abs().max())
    assert float((a["logits_i"] - ref["logits_i"]).abs().max()) <= 3e-2 * scale      # the documented bf16 bound (DESIGN_HISTORY.md s2)
    worst, min_cos = 0.0, (2.0, "")
    for n, g in ref["grads"].items():
        if g is None:
            continue
        ga = a["grads"][n]
        nr = float(g.norm())
        if nr == 0.0:
            continue
        worst = max(worst, abs(float(ga.norm()) - nr) / nr)
        # direction as well as size: cosine of the two gradients (0.999 = at most 4.5 % of the gradient off-direction; two bf16
        # GEMM families differ by the rounding of single outputs whose fp32 sums were accumulated in a different order)
        cos = float((ga.double().flatten() @ g.double().flatten()) / (ga.double().norm() * g.double().norm() + 1e-30))
        if cos < min_cos[0]:
            min_cos = (cos, n)
    print("v4 vs v2 GEMM family, bf16 step: smallest gradient cosine %.6f (%s), worst norm deviation %.4f" % (min_cos + (worst,)))
    assert min_cos[0] >= 0.999, min_cos            # measured on the MI355X (round 3): 0.99940
    assert worst < 5e-2, worst


def test_clip_bf16_v4_step_gradients_match_fp32_step_in_direction():
    """The TIMED path (bf16, every tower GEMM on gemm_v4) against the validation path (fp32: the arithmetic that meets the
    reference at 1e-3 in test_gpu_golden_fullwidth.py) on the reference fixture's inputs at b = 256: loss, and for EVERY parameter
    the full gradient tensor -- cosine and relative error, not a digest.  Stated bound for the benchmarked kernel: cosine >= 0.998
    per parameter (<= 6.3 % of a gradient off-direction), relative error <= 0.06, loss within 1e-3 relative."""
    from declip_amd import ops
    g = load_golden("clip_vitb32_b256")
    ops.gemm_stats(reset=True)
    lo = run_engine(g["cfg"], g["b"], g["seed"], g["logit_scale"], "bf16")
    stats = ops.gemm_stats()
    assert stats["v4"] >= 200 and stats["v4"] >= 0.8 * sum(stats.values()), stats
    hi = run_engine(g["cfg"], g["b"], g["seed"], g["logit_scale"], "fp32")
    assert abs(hi["loss"] - g["loss"]) <= 1e-3 * abs(g["loss"])
    assert abs(lo["loss"] - hi["loss"]) <= 1e-3 * abs(hi["loss"]), (lo["loss"], hi["loss"])
    gmax = max(float(v.norm()) for v in hi["grads"].values() if v is not None)
    worst_cos, worst_rel, named = (2.0, ""), (0.0, ""), {}
    for n, gh in hi["grads"].items():
        if gh is None or float(gh.norm()) < 1e-3 * gmax:
            continue
        gl = lo["grads"][n].double().flatten()
        gh = gh.double().flatten()
        cos = float(gl @ gh / (gl.norm() * gh.norm() + 1e-30))
        rel = float((gl - gh).norm() / gh.norm())
        named[n] = (cos, rel)
        if cos < worst_cos[0]:
            worst_cos = (cos, n)
        if rel > worst_rel[0]:
            worst_rel = (rel, n)
    print("bf16 (gemm_v4) vs fp32 step, %d parameters: smallest cosine %.6f (%s), largest relative error %.4f (%s)"
          % (len(named), worst_cos[0], worst_cos[1], worst_rel[0], worst_rel[1]))
    for n in ("visual.proj", "visual.transformer.resblocks.0.attn.in_proj_weight", "visual.transformer.resblocks.11.mlp.c_fc.weight",
              "encode_text.transformer.resblocks.0.mlp.c_proj.weight", "encode_text.token_embedding.weight", "logit_scale"):
        assert n in named, n
        print("   %-60s cos %.6f rel %.4f" % ((n,) + named[n]))
    # measured on the MI355X (round 3): smallest cosine 0.99956, largest relative error 0.0295 over 302 parameters
    assert worst_cos[0] >= 0.998, worst_cos
    assert worst_rel[0] <= 0.06, worst_rel


def test_clip_two_tower_streams_match_one_stream(monkeypatch):
    """Image and text tower on two HIP streams (CLIP.features, DH_TOWER_STREAMS) against the one-stream order, ViT-B/32
    b = 256, three optimizer steps each.  Same kernels on the same data: the forward (features, first loss) is bit-identical.
    Gradients are compared at the run-to-run noise of ONE mode (the float atomics of the InfoNCE backward perturb d(features)
    by ~1e-7, which bf16 rounding inside the towers amplifies to <= 3e-3 of a parameter's largest gradient -- measured with
    tools/stream_ab.py); a missing stream dependency (a lost or half-written gradient) is orders of magnitude above that."""
    from declip_amd import synth
    from declip_amd.loss import ClipInfoCELoss
    from declip_amd.optim import build_adamw
    from declip_amd.testing import build_clip
    cfg, b, seed = synth.VITB32, 256, 5
    images = synth.synth_images(b, res=cfg["res"], seed=seed).cuda()
    ids = synth.synth_tokens(b, ctx=cfg["ctx"], seed=seed, vocab=cfg["vocab"]).cuda()

    def run(mode):
        monkeypatch.setenv("DH_TOWER_STREAMS", mode)
        model = build_clip(cfg, dtype="bf16", seed=seed)
        opt = build_adamw(model, lr=1e-4, weight_decay=0.1)
        crit = ClipInfoCELoss()
        losses, first = [], None
        for i in range(3):
            li, lt = model({"images": images, "captions": ids})
            loss, _ = crit(li, lt)
            opt.zero_grad()
            loss.backward()
            if i == 0:
                torch.cuda.synchronize()
                first = {n: p.grad.detach().float().cpu() for n, p in model.named_parameters() if p.grad is not None}
                first["~img"], first["~txt"] = li.Q.detach().float().cpu(), lt.Q.detach().float().cpu()
            opt.step()
            losses.append(float(loss.detach()))
        torch.cuda.synchronize()
        assert (len(model._flat_store.side_streams) == 1) == (mode == "1")
        return losses, first

    l1, g1 = run("1")
    l0, g0 = run("0")
    assert l1[0] == l0[0] and torch.equal(g1["~img"], g0["~img"]) and torch.equal(g1["~txt"], g0["~txt"])
    for a, c in zip(l1, l0):
        assert abs(a - c) <= 3e-3 * abs(c)
    for n, g in g0.items():
        assert float((g1[n] - g).abs().max()) <= 1e-2 * float(g.abs().max()) + 1e-12, n


def test_bf16_mirror_written_by_the_fused_adamw_equals_a_cast_of_the_master_weights(monkeypatch):
    """begin_step() trusts the bf16 mirror that adamw_seg_kernel writes next to the master weights (no 0.9 GB cast per step): after every
    optimizer step the mirror must be bit-identical to a round-to-nearest-even cast of the fp32 parameters, and no full-size cast
    may run after the first step."""
    from declip_amd import ops, synth
    from declip_amd.loss import ClipInfoCELoss
    from declip_amd.optim import build_adamw
    from declip_amd.testing import build_clip
    cfg, b = synth.TINY, 8
    model = build_clip(cfg, dtype="bf16", seed=5)
    flat = model.__dict__["_flat_store"].ensure()
    opt = build_adamw(model, lr=1e-3, betas=(0.9, 0.98), weight_decay=0.1)
    crit = ClipInfoCELoss()
    full_casts = []
    real_cast = ops.cast

    def counting_cast(src, dst, *a, **k):
        if src.numel() == flat.total:
            full_casts.append(1)
        return real_cast(src, dst, *a, **k)
    monkeypatch.setattr(ops, "cast", counting_cast)
    for step in range(3):
        images = synth.synth_images(b, res=cfg["res"], seed=step).cuda()
        ids = synth.synth_tokens(b, ctx=cfg["ctx"], seed=step, vocab=cfg["vocab"]).cuda()
        opt.zero_grad()
        li, lt = model({"images": images, "captions": ids})
        loss, _ = crit(li, lt)
        loss.backward()
        opt.step()
        torch.cuda.synchronize()
        assert torch.equal(flat.flat_b, flat.flat_p.to(torch.bfloat16)), step
    assert len(full_casts) == 1
