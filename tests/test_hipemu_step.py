"""Model-level `-m gpu` tests executed on the CPU with EVERY kernel emulated (tests/hipemu): the engine, the flat parameter store,
the fused losses and the optimiser run on the product's own kernel sources (MFMA-builtin GEMM tiles, attention, LayerNorm, InfoNCE,
BatchNorm, ...), no torch stand-ins anywhere -- against the goldens generated from the unmodified reference, with the GPU tests'
own tolerances.  The inline-ISA GEMM families are the one part of the product path this cannot execute (they decline on the host
build and the plain-HIP tiles run instead); they are covered by tests/test_gpu_gemm_v4.py / test_gpu_kernels.py on hardware."""
import importlib
import os

import pytest

from hipemu_util import emulated_gpu

CASES = [
    ("test_gpu_clip", "test_clip_fp32_matches_reference_golden", ("clip_tiny",)),
    ("test_gpu_clip", "test_clip_fp32_matches_reference_golden", ("clip_tiny_scale5",)),
    ("test_gpu_clip", "test_clip_bf16_close_to_reference", ("clip_tiny", 1e-2, 5e-2)),
    ("test_gpu_clip", "test_clip_fp32_unfused_surface_matches_fused", ()),
    ("test_gpu_clip", "test_clip_accuracy_matches_oracle", ()),
    ("test_gpu_clip", "test_train_steps_flat_adamw_matches_torch_adamw_on_oracle", ()),
    ("test_gpu_clip", "test_declip_step_matches_reference_golden", ("fp32", 1e-3)),
    ("test_gpu_clip", "test_slip_step_matches_reference_golden", ("fp32", 1e-3)),
    ("test_gpu_clip", "test_filip_step_matches_reference_golden", ("fp32", 1e-3)),
    ("test_gpu_zeroshot", "test_zero_shot_fp32_matches_reference_evaluate", (8,)),
    ("test_gpu_resnet_intake_packed", "test_clip_r50_fp32_matches_reference_golden", ()),
    ("test_gpu_resnet_intake_packed", "test_clip_r50_fc_head_fp32_matches_reference_golden", ()),
]
# minutes each on the emulation (bf16 GEMMs = emulated MFMA tiles over 37 k pixel rows / a 49 k-word vocabulary): run on demand with
# HIPEMU_SLOW=1 (all four passed when this file was written: 690 s, 160 s, 170 s, 40 s)
SLOW = [
    ("test_gpu_resnet_intake_packed", "test_clip_r50_bf16_close_to_reference", ()),
    ("test_gpu_clip", "test_declip_step_matches_reference_golden", ("bf16", 2e-2)),
    ("test_gpu_clip", "test_defilip_step_matches_reference_golden", ("bf16", 3e-2)),
    ("test_gpu_resnet_intake_packed", "test_declip_r50_fp32_matches_reference_golden", ()),
    ("test_gpu_resnet_intake_packed", "test_filip_r50_fp32_matches_reference_golden", ()),
]


@pytest.mark.parametrize("module,name,args", CASES, ids=["%s-%d" % (c[1], i) for i, c in enumerate(CASES)])
def test_gpu_model_test_on_host_emulation(module, name, args):
    mod = importlib.import_module(module)
    with emulated_gpu():
        getattr(mod, name)(*args)


@pytest.mark.skipif(os.environ.get("HIPEMU_SLOW") != "1", reason="minutes per case on the host emulation: set HIPEMU_SLOW=1")
@pytest.mark.parametrize("module,name,args", SLOW, ids=["%s-%d" % (c[1], i) for i, c in enumerate(SLOW)])
def test_gpu_model_test_on_host_emulation_slow(module, name, args):
    mod = importlib.import_module(module)
    with emulated_gpu():
        getattr(mod, name)(*args)


def test_smoke_entry_on_host_emulation(monkeypatch):
    """__graft_entry__.smoke() -- the driver's first GPU call of a round -- end to end on the emulated kernels."""
    import torch

    import __graft_entry__ as entry
    monkeypatch.setattr(torch.cuda, "is_available", lambda: True)
    monkeypatch.setattr(torch.cuda, "set_device", lambda d: None)
    with emulated_gpu():
        entry.smoke()


PACKED = [
    ("test_gpu_clip", "test_clip_fp32_matches_reference_golden", ("clip_tiny",)),
    ("test_gpu_clip", "test_clip_fp32_matches_reference_golden", ("clip_tiny_scale5",)),
    ("test_gpu_clip", "test_clip_bf16_close_to_reference", ("clip_tiny", 1e-2, 5e-2)),
    ("test_gpu_clip", "test_train_steps_flat_adamw_matches_torch_adamw_on_oracle", ()),
    ("test_gpu_clip", "test_declip_step_matches_reference_golden", ("fp32", 1e-3)),      # packed word features feed the MLM head
    ("test_gpu_zeroshot", "test_zero_shot_fp32_matches_reference_evaluate", (8,)),          # prompt ensembles: short captions, eval mode
]


PACKED_MODES = [("1", c) for c in PACKED] + [("2", c) for c in (PACKED[0], PACKED[1])]       # 1: variable-length attention, 2: via the dense layout


@pytest.mark.parametrize("mode,case", PACKED_MODES, ids=["packed%s-%s-%d" % (m, c[1], i) for i, (m, c) in enumerate(PACKED_MODES)])
def test_packed_text_tower_passes_the_same_goldens(monkeypatch, mode, case):
    """DH_TEXT_PACKED=1|2: the text tower computes only the rows up to <|endoftext|> of every caption (engine.PackedCaptions) -- and
    has to pass the very same reference goldens / oracle comparisons as the padded layout, at the same tolerances."""
    module, name, args = case
    monkeypatch.setenv("DH_TEXT_PACKED", mode)
    mod = importlib.import_module(module)
    with emulated_gpu():
        getattr(mod, name)(*args)


@pytest.mark.parametrize("dtype,mode", [("fp32", "1"), ("bf16", "1")])
def test_packed_text_tower_equals_padded_on_edge_lengths(monkeypatch, dtype, mode):
    """captions of minimal length (SOT, EOT), of the full context and in between, batch of one: features, parameter gradients and
    the row bookkeeping of the packed tower against the padded one (same kernels, same weights)."""
    import torch

    from declip_amd import engine, synth
    from declip_amd.testing import build_clip
    cfg = synth.TINY
    ctx, V = cfg["ctx"], cfg["vocab"]

    def caption(n_words, seed):
        g = torch.Generator().manual_seed(seed)
        row = torch.zeros(ctx, dtype=torch.long)
        row[0] = V - 2
        row[1:1 + n_words] = torch.randint(0, V - 3, (n_words,), generator=g)
        row[1 + n_words] = V - 1
        return row

    for lens in ([0, ctx - 2, 5, 1, 9], [ctx - 2], [0]) if (dtype, mode) == ("fp32", "1") else ([0, ctx - 2, 5, 1, 9],):
        ids = torch.stack([caption(n, i) for i, n in enumerate(lens)])
        res = {}
        for packed in ("0", mode):
            monkeypatch.setenv("DH_TEXT_PACKED", packed)
            with emulated_gpu():
                model = build_clip(cfg, dtype=dtype, seed=3, device="cpu")
                out = model.encode_text(ids.clone())
                w = torch.linspace(-1, 1, out.numel()).view_as(out)
                (out * w).sum().backward()
                res[packed] = (out.detach().clone(), {n: p.grad.detach().clone() for n, p in model.named_parameters()
                                                      if n.startswith("encode_text.") and p.grad is not None})
        pk = engine.PackedCaptions(ids, 8)
        assert pk.rows == sum(n + 2 for n in lens) and pk.rows_pad % 8 == 0 and pk.cu.tolist()[-1] == pk.rows
        tol = 1e-5 if dtype == "fp32" else 3e-2
        (o0, g0), (o1, g1) = res["0"], res[mode]
        assert float((o0 - o1).abs().max()) <= tol * float(o0.abs().max())
        assert set(g0) == set(g1)
        for n in g0:
            assert float((g0[n] - g1[n]).abs().max()) <= tol * float(g0[n].abs().max() + 1e-12), (lens, n)
        # the positional embedding beyond the longest caption receives exactly no gradient, as in the padded layout
        longest = max(lens) + 2
        assert float(g1["encode_text.positional_embedding"][longest:].abs().max() if longest < ctx else 0.0) == 0.0


POOLED = [
    ({"DH_POOLED_LAST": "1"}, ("test_gpu_clip", "test_clip_fp32_matches_reference_golden", ("clip_tiny",))),
    ({"DH_POOLED_LAST": "1"}, ("test_gpu_clip", "test_train_steps_flat_adamw_matches_torch_adamw_on_oracle", ())),
    ({"DH_POOLED_LAST": "1", "DH_TEXT_PACKED": "1"}, ("test_gpu_clip", "test_slip_step_matches_reference_golden", ("fp32", 1e-3))),
    ({"DH_POOLED_LAST": "1", "DH_TEXT_PACKED": "1"}, ("test_gpu_clip", "test_clip_fp32_matches_reference_golden", ("clip_tiny_scale5",))),
    ({"DH_POOLED_LAST": "1", "DH_TEXT_PACKED": "1"}, ("test_gpu_clip", "test_clip_bf16_close_to_reference", ("clip_tiny", 1e-2, 5e-2))),
    ({"DH_POOLED_LAST": "1", "DH_TEXT_PACKED": "1"}, ("test_gpu_zeroshot", "test_zero_shot_fp32_matches_reference_evaluate", (8,))),
]


@pytest.mark.parametrize("env,case", POOLED, ids=["pooled-%s-%d" % (c[1], i) for i, (e, c) in enumerate(POOLED)])
def test_pooled_last_block_passes_the_same_goldens(monkeypatch, env, case):
    """DH_POOLED_LAST=1: the last block of each tower runs its query / attention / out_proj / MLP for the pooled rows only (CLS,
    <|endoftext|>; engine.block_fwd_pooled) -- alone and together with packed captions -- against the same goldens."""
    module, name, args = case
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    mod = importlib.import_module(module)
    calls = []
    with emulated_gpu() as ops:
        orig = ops.attn_pooled_fwd
        ops.attn_pooled_fwd = lambda *a, **k: (calls.append(1), orig(*a, **k))[1]
        try:
            getattr(mod, name)(*args)
        finally:
            ops.attn_pooled_fwd = orig
    assert len(calls) >= 2          # both towers took the pooled path
