"""Zero-shot evaluate on the HIP towers (declip_amd/zeroshot.py) against (a) what the reference's own ClsSolver.evaluate
produced on the same seeded weights / prompts / images (tests/golden/zeroshot_tiny.pt) and (b) the CPU restatement at ViT-B/32.
fp32 mode: 1e-3 on the class scores (north_star tolerance); bf16 mode: cosine logits within 2e-2 absolute, predictions equal
wherever the oracle's top-1 margin exceeds that bound."""
import pytest
import torch

from oracle_util import load_golden

pytestmark = pytest.mark.gpu


def _inputs(cfg, label_num, prompts_num, b, batches, seed, max_len=6):
    from declip_amd import synth
    ids = synth.synth_tokens(label_num * prompts_num, ctx=cfg["ctx"], seed=seed + 77, vocab=cfg["vocab"], max_len=max_len)
    return ids, [synth.synth_images(b, res=cfg["res"], seed=seed * 1000 + i) for i in range(batches)]


@pytest.mark.parametrize("chunk", [2048, 8])
def test_zero_shot_fp32_matches_reference_evaluate(chunk):
    from declip_amd import zeroshot
    from declip_amd.testing import build_clip
    g = load_golden("zeroshot_tiny")
    model = build_clip(g["cfg"], dtype="fp32", seed=g["seed"]).eval()
    ids, batches = _inputs(g["cfg"], g["label_num"], g["prompts_num"], g["b"], g["batches"], g["seed"])
    emb = zeroshot.class_embeddings(model, ids, g["label_num"], text_chunk=chunk)
    assert float((emb.norm(dim=-1) - 1).abs().max()) < 1e-5
    for i, images in enumerate(batches):
        out = zeroshot.classify(model, images.cuda(), emb, torch.eye(g["label_num"]))
        ref = g["scores"][i]
        assert float((out["scores"].cpu() - ref).abs().max()) <= 1e-3 * float(ref.max())
        assert torch.equal(out["prediction"].cpu(), g["predictions"][i])


@pytest.mark.parametrize("dtype,tol", [("fp32", 1e-3), ("bf16", 2e-2)])
def test_zero_shot_vitb32_matches_oracle(dtype, tol):
    """ViT-B/32, 12 classes x 3 prompts, 16 images: logits (cosines, |.| <= 1) against the CPU restatement."""
    from declip_amd import synth, zeroshot
    from declip_amd.testing import build_clip
    from oracle import restated
    cfg, seed, C, P, b = synth.VITB32, 11, 12, 3, 16
    ids, (images,) = _inputs(cfg, C, P, b, 1, seed, max_len=10)
    sd = synth.synth_state(synth.clip_shapes(cfg), seed=seed)
    with torch.no_grad():
        ref_emb, ref_logits, ref_scores, ref_pred = restated.zero_shot(images, ids, C, sd, cfg)
    model = build_clip(cfg, dtype=dtype, seed=seed).eval()
    emb = zeroshot.class_embeddings(model, ids, C)
    out = zeroshot.classify(model, images.cuda(), emb, torch.eye(C))
    assert float((emb.cpu() - ref_emb).abs().max()) <= tol
    assert float((out["logits"].cpu() - ref_logits).abs().max()) <= tol
    top2 = ref_logits.topk(2, dim=1).values
    sure = (top2[:, 0] - top2[:, 1]) > 2 * tol
    assert torch.equal(out["prediction"].cpu()[sure], ref_pred[sure])
    assert float((out["scores"].sum(1) - 1).abs().max()) < 1e-5


def test_solver_evaluate_runs_on_gpu(tmp_path):
    """`--evaluate` through the solver on the synthetic set: small bf16 towers, 100 classes x 2 prompts, 2 x 64 images."""
    import yaml
    from declip_amd.solver import ClsSolver
    from test_gpu_solver import _config
    cfg = _config("clip")
    cfg["data"]["test"] = dict(type="synthetic", label_num=100, prompts_num=2, batch_size=64, batches=2)
    p = tmp_path / "config.yaml"
    p.write_text(yaml.safe_dump(cfg))
    s = ClsSolver(str(p))
    m = s.evaluate()
    assert m["count"] == 128 and 0.0 <= m["top1"] <= m["top5"] <= 100.0
    assert s.model.training
