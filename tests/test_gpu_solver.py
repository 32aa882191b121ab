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
        cfg["data"]["train"] = dict(image_text_two_view=True)             # yfcc15m_vit_declip/config.yaml:101
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


@pytest.mark.parametrize("heads", [8, 2])
def test_solver_graphed_step_matches_the_eager_step(heads, monkeypatch):
    """model.kwargs.engine.step_graph: the CLIP solver's forward + loss + backward replayed from captured graphs (one per packed-row
    key of the caption batch; declip_amd/graph.py) while the loader hands a DIFFERENT batch every iteration (clip_solver.py:398-402),
    against the same run stepped eagerly: same loss trajectory and temperature.  heads = 2: head dimension 64 (the key is the padded
    row count, the valid count is read on the device).  heads = 8: head dimension 16 -- those kernels take the exact row count as a
    launch argument, a graph would have to be re-captured for almost every batch, so the solver keeps the EAGER step there (round 5,
    ADVICE r4: the same rule as bench.py) and the run must simply equal the eager run."""
    from declip_amd.solver import ClsSolver

    def run(graph):
        cfg = _config("clip")
        cfg["model"]["kwargs"]["text_encode"]["transformer_heads"] = heads
        cfg["model"]["kwargs"]["engine"] = dict(dtype="bf16", step_graph=graph)
        torch.manual_seed(0)
        s = ClsSolver(cfg)
        rec, orig = [], s.train_step

        def wrapped(step):
            out = orig(step)
            rec.append(out["loss"].detach().clone())
            return out
        s.train_step = wrapped
        s.train(max_steps=10)
        torch.cuda.synchronize()
        return s, [float(x) for x in rec]

    se, le = run(False)
    sg, lg = run(True)
    if heads == 2:
        assert sg.__dict__.get("_graph") is not None and sg._graph["step"].replays >= 3 and sg._graph["step"].captures >= 1
    else:
        assert sg.__dict__.get("_graph") is None and sg.__dict__.get("_graph_off") is True
    assert se.__dict__.get("_graph") is None
    for a, c in zip(lg, le):
        assert abs(a - c) <= 5e-3 * abs(c), (lg, le)
    assert abs(float(sg.model.module.logit_scale) - float(se.model.module.logit_scale)) <= 1e-3


def test_solver_prefetcher_bytes_and_strings_match_floats_and_ids(tmp_path):
    """A user loader that yields what a decoder yields -- uint8 HWC images and caption strings -- through the prefetcher
    (background tokenisation, pinned staging, copies on a side stream, bytes normalised on the GPU) gives the same first losses as
    the same data handed over as normalised floats and token ids."""
    import yaml
    from declip_amd import bpe
    from declip_amd.prefetch import DataPrefetcher
    from declip_amd.solver import ClsSolver
    from oracle import ref_harness, restated
    path = ref_harness.synthetic_bpe_path()
    cfg = _config("clip")
    cfg["model"]["kwargs"]["text_encode"]["bpe_path"] = path
    p = tmp_path / "config.yaml"
    p.write_text(yaml.safe_dump(cfg))
    g = torch.Generator().manual_seed(3)
    raw = [torch.randint(0, 256, (32, 64, 64, 3), generator=g, dtype=torch.uint8) for _ in range(4)]
    caps = [[["caption %d of batch %d, it's a photo!" % (j, i), "second caption"] for j in range(32)] for i in range(4)]

    def as_decoder():
        for im, c in zip(raw, caps):
            yield {"images": im, "captions": c}

    def as_tensors():
        tok = bpe.SimpleTokenizer(path)
        for im, c in zip(raw, caps):
            yield {"images": restated.image_prep_u8(im, (64, 64)), "captions": bpe.tokenize(tok, [x[0] for x in c], 32)}

    def run(loader):
        torch.manual_seed(0)                                  # same initial weights for both runs
        s = ClsSolver(str(p), train_loader=loader)
        rec, orig = [], s.train_step

        def wrapped(step):
            out = orig(step)
            rec.append(out["loss"].detach())
            return out
        s.train_step = wrapped
        s.train(max_steps=4)
        torch.cuda.synchronize()
        return s, [float(x) for x in rec]

    s1, l1 = run(as_decoder())
    s2, l2 = run(as_tensors())
    assert isinstance(s1._iter, DataPrefetcher) and s1._iter.next() is None
    assert len(l1) == len(l2) == 4 and l1[0] == l1[0]
    for a, c in zip(l1, l2):
        assert abs(a - c) <= 2e-2 * abs(c), (l1, l2)        # bf16 towers; the inputs agree to fp32 rounding


def test_prefetcher_leaves_mlm_labels_on_the_host():
    """The masked-LM head takes its row selection from the labels with index arithmetic on the labels' OWN device: uploaded by the
    prefetcher they turned `(labels != -100).nonzero()` into a device read-back (a host stall on the whole queue) in every
    solver-driven DeCLIP / DeFILIP step.  Everything else of the batch is on the device when next() returns."""
    from declip_amd import synth
    from declip_amd.prefetch import DataPrefetcher
    ids = synth.synth_tokens(8, ctx=16, seed=0, vocab=512)
    masked, labels = synth.synth_mlm(ids, 512, seed=0)
    batches = [{"images": synth.synth_images(8, views=2, res=32, seed=i), "captions": torch.stack([masked, ids], dim=1), "mlm_labels": labels.clone()}
               for i in range(2)]
    pf = DataPrefetcher(iter(batches), device=torch.device("cuda", 0), context_length=16)
    got = pf.next()
    assert got["images"].is_cuda and got["captions"].is_cuda
    assert not got["mlm_labels"].is_cuda and torch.equal(got["mlm_labels"], labels)
    from declip_amd.heads import _mlm_selection
    sel, lab = _mlm_selection(got["mlm_labels"], got["captions"].device)
    assert sel.is_cuda and lab.is_cuda and torch.equal(sel.cpu(), (labels.reshape(-1) != -100).nonzero().reshape(-1))
