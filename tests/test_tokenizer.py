"""Tokeniser parity against the unmodified reference (skipped where /root/reference is absent) and
padding/truncation/masking semantics (text_transformer.py:144-180, mask_tokens.py:5-29)."""
import pytest
import torch

from declip_amd import bpe
from oracle import ref_harness

CAPS = ["a photo of a cat", "Hello, World!  it's 42 degrees...", "q0 z q1 z", "naïve café — ok", "x" * 400]


@pytest.mark.skipif(not ref_harness.reference_available(), reason="reference tree not present")
def test_tokenizer_matches_reference():
    ref = ref_harness.load_reference()
    path = ref_harness.synthetic_bpe_path()
    rt = ref.modules["prototype.model.utils.text_utils.simple_tokenizer"].SimpleTokenizer(bpe_path=path)
    mine = bpe.SimpleTokenizer(path)
    assert len(mine.encoder) == len(rt.encoder) == 49409
    for k in ("<|mask|>", "<|startoftext|>", "<|endoftext|>"):
        assert mine.encoder[k] == rt.encoder[k]
    for c in CAPS:
        assert mine.encode(c) == rt.encode(c), c


def test_tokenize_layout_and_truncation():
    tok = bpe.SimpleTokenizer(ref_harness.synthetic_bpe_path())
    ids = bpe.tokenize(tok, CAPS, context_length=77)
    assert ids.shape == (len(CAPS), 77) and ids.dtype == torch.long
    assert (ids[:, 0] == 49407).all()
    eot = ids.argmax(-1)
    assert all(int(ids[i, eot[i]]) == 49408 for i in range(len(CAPS)))
    assert int(eot[-1]) == 76                       # over-long caption keeps the final EOT in the last slot
    assert all((ids[i, eot[i] + 1:] == 0).all() for i in range(len(CAPS)))


def test_mlm_masking_semantics():
    from declip_amd import synth
    ids = synth.synth_tokens(64, ctx=77, seed=3)
    g = torch.Generator().manual_seed(0)
    masked, labels = bpe.mask_token_ids(ids, 49409, generator=g)
    sel = labels != -100
    eot = ids.argmax(-1, keepdim=True)
    pos = torch.arange(77)[None]
    assert not sel[pos > eot].any() and not sel[:, 0].any() and not sel[pos == eot].any()
    assert (labels[sel] == ids[sel]).all()
    frac = sel.float().sum() / ((pos < eot) & (pos > 0)).float().sum()
    assert 0.10 < float(frac) < 0.20
    assert 0.6 < float((masked[sel] == 49406).float().mean()) < 0.95
    assert (masked[~sel] == ids[~sel]).all()
