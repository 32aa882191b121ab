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


# ---------------------------------------------------------------------------------------------------------------------
# native (C-ABI, host threads) tokeniser: bit-exact against the Python one on a vocabulary with real multi-level merges
def _trained_vocab(tmp_path, n_merges=49152 - 256 - 2):
    """A merges file with the reference's line count whose first ~600 merges are LEARNED from a small corpus by a plain BPE
    trainer (so words really merge over several levels); the rest are inert filler lines.  Test infrastructure."""
    import collections
    import gzip
    corpus = ("a photo of a cat sitting on the mat . the quick brown fox jumps over the lazy dog ! it's a dog's life , isn't it ? "
              "two dogs and three cats playing in the garden near the house with 42 windows ... photo photos photograph "
              "there they're their we've i'm you'll she'd don't can't the theatre thereafter weather whether 2021 1999 "
              "<|startoftext|> hello <|endoftext|> hello-world hello_world e-mail: someone@example.com #hashtag $9.99 (ok)") * 3
    tok = bpe.SimpleTokenizer(ref_harness.synthetic_bpe_path())
    words = collections.Counter()
    import regex
    for w in regex.findall(tok.pat, corpus.lower()):
        enc = "".join(tok.byte_encoder[b] for b in w.encode("utf-8"))
        words[tuple(enc[:-1]) + (enc[-1] + "</w>",)] += 1
    merges = []
    for _ in range(600):
        pairs = collections.Counter()
        for w, c in words.items():
            for a, b in zip(w, w[1:]):
                pairs[(a, b)] += c
        if not pairs:
            break
        (a, b), cnt = max(pairs.items(), key=lambda kv: (kv[1], kv[0]))
        if cnt < 2:
            break
        merges.append((a, b))
        new = collections.Counter()
        for w, c in words.items():
            out, i = [], 0
            while i < len(w):
                if i < len(w) - 1 and w[i] == a and w[i + 1] == b:
                    out.append(a + b)
                    i += 2
                else:
                    out.append(w[i])
                    i += 1
            new[tuple(out)] += c
        words = new
    assert len(merges) > 100
    lines = ["#version: trained-for-tests"] + ["%s %s" % m for m in merges] + ["q%d z" % i for i in range(n_merges - len(merges))] + [""]
    path = str(tmp_path / "bpe_trained.txt.gz")
    with gzip.open(path, "wb") as f:
        f.write("\n".join(lines).encode("utf-8"))
    return path


NATIVE_CAPS = CAPS + [
    "", "   ", "it's the dogs' photo, isn't it?!  they're here & we've won; i'm sure you'll agree she'd know",
    "the theatre thereafter: weather/whether 2021-1999 = 22 ... (photo) [photograph] {photos}",
    "<|startoftext|> hello <|endoftext|> <|mask|> <|start", "e-mail: someone@example.com #hashtag $9.99 100% ''quoted'' 's 't 'x",
    "tab\tseparated\nlines\r\nand \x0b vertical \x0c feed", "MiXeD Case Words ARE lowered", "ctrl \x01 char and del \x7f here",
    "emoji \U0001F600 and accents éàü and CJK 漢字 mixed with ascii photo of a cat", "a" * 300 + " " + "photo " * 100,
    "x y z " * 40, "'" , "'s", "s'", "9", "1234567890", "!!!???...", "a.b,c;d:e",
]


def test_native_tokenizer_matches_python(tmp_path):
    import random
    path = _trained_vocab(tmp_path)
    py = bpe.SimpleTokenizer(path)
    nat = bpe.NativeTokenizer(path, threads=4)
    assert nat._L.load().dh_bpe_vocab_size(nat._handle) == 49409
    rng = random.Random(0)
    alphabet = "abcdefghijklmnopqrstuvwxyz      ''.,!?-0123456789<|>#@&the photo of cat dog "
    rand_caps = ["".join(rng.choice(alphabet) for _ in range(rng.randint(0, 200))) for _ in range(400)]
    words = "a photo of the cat dog it's they're photograph theatre weather 42 ! ... <|endoftext|> hello-world".split()
    rand_caps += [" ".join(rng.choice(words) for _ in range(rng.randint(1, 90))) for _ in range(400)]
    caps = NATIVE_CAPS + rand_caps
    for ctx in (77, 16):
        ref = bpe.tokenize(py, caps, context_length=ctx)
        got = bpe.tokenize(nat, caps, context_length=ctx)
        assert got.dtype == torch.long and torch.equal(ref, got), [c for c, a, b in zip(caps, ref, got) if not torch.equal(a, b)][:3]
    # merges really happened (several levels): "photograph" is far shorter than its byte count
    assert len(py.encode("photograph")) <= 3
    # single-thread and many-thread paths agree; repeated calls hit the shared word cache
    nat.threads = 1
    assert torch.equal(bpe.tokenize(nat, caps), bpe.tokenize(py, caps))
    nat.threads = 16
    big = caps * 3                                       # >= 2048 captions: the batch is split over host threads
    assert len(big) >= 2048 and torch.equal(bpe.tokenize(nat, big), bpe.tokenize(py, caps).repeat(3, 1))


def test_native_tokenizer_flags_non_ascii_rows(tmp_path):
    import ctypes
    import numpy as np
    path = _trained_vocab(tmp_path)
    nat = bpe.NativeTokenizer(path)
    caps = [b"plain ascii caption", "café".encode("utf-8"), b"UPPER", b"ok again"]
    offs = np.zeros(len(caps) + 1, dtype=np.int64)
    np.cumsum([len(c) for c in caps], out=offs[1:])
    out = torch.full((len(caps), 8), -1, dtype=torch.long)
    status = np.full(len(caps), -1, dtype=np.int32)
    L = nat._L
    L.check(L.load().dh_bpe_encode(nat._handle, b"".join(caps), offs.ctypes.data, len(caps), 8, out.data_ptr(), status.ctypes.data, 2))
    assert status.tolist() == [0, 1, 1, 0]
    assert (out[1] == 0).all() and (out[2] == 0).all() and out[0, 0] == 49407 and out[3, 0] == 49407
    assert L.load().dh_bpe_encode(None, b"", offs.ctypes.data, 0, 8, out.data_ptr(), status.ctypes.data, 1) != 0      # bad handle -> error code
