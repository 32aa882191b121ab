"""Caption tokenisation for the text tower (CPU side, off the GPU critical path).

Semantics follow the reference (model/utils/text_utils/simple_tokenizer.py:62-134 -- the OpenAI CLIP
byte-level BPE with one extra `<|mask|>` token, so SOT = 49407, EOT = 49408 = the largest id -- and
text_encoder/text_transformer.py:144-180 for padding/truncation; MLM masking after
model/utils/text_utils/mask_tokens.py:5-29).  Written from the published BPE algorithm; the
engine also accepts pre-tokenised LongTensors, which is what throughput runs use.
"""
import gzip
import html
from functools import lru_cache

import regex as re
import torch


@lru_cache()
def _byte_table():
    """printable stand-ins for all 256 byte values (byte-level BPE)."""
    keep = list(range(33, 127)) + list(range(161, 173)) + list(range(174, 256))
    chars = keep[:]
    extra = 0
    for b in range(256):
        if b not in keep:
            keep.append(b)
            chars.append(256 + extra)
            extra += 1
    return {b: chr(c) for b, c in zip(keep, chars)}


try:                                   # resolved once: a failing `import` inside the per-caption path costs ~45 us each time
    import ftfy as _ftfy
except ImportError:
    _ftfy = None


def _unescape(text):
    """basic_clean of the reference (simple_tokenizer.py:50-53): ftfy.fix_text when available, html.unescape twice."""
    if _ftfy is not None:
        text = _ftfy.fix_text(text)
    return html.unescape(html.unescape(text))


def _clean(text):
    return re.sub(r"\s+", " ", _unescape(text).strip()).strip()


class SimpleTokenizer(object):
    def __init__(self, bpe_path):
        self.byte_encoder = _byte_table()
        self.byte_decoder = {v: k for k, v in self.byte_encoder.items()}
        lines = gzip.open(bpe_path).read().decode("utf-8").split("\n")
        merges = [tuple(l.split()) for l in lines[1:49152 - 256 - 2 + 1]]
        vocab = list(self.byte_encoder.values())
        vocab = vocab + [v + "</w>" for v in vocab] + ["".join(m) for m in merges]
        vocab += ["<|mask|>", "<|startoftext|>", "<|endoftext|>"]
        self.encoder = {tok: i for i, tok in enumerate(vocab)}
        self.decoder = {i: tok for tok, i in self.encoder.items()}
        self.ranks = {m: i for i, m in enumerate(merges)}
        self.cache = {t: t for t in ("<|mask|>", "<|startoftext|>", "<|endoftext|>")}
        self.pat = re.compile(r"<\|startoftext\|>|<\|endoftext\|>|'s|'t|'re|'ve|'m|'ll|'d|[\p{L}]+|[\p{N}]|[^\s\p{L}\p{N}]+",
                              re.IGNORECASE)

    def _bpe(self, token):
        """merge the lowest-ranked adjacent pair until none is left (byte-level BPE)."""
        if token in self.cache:
            return self.cache[token]
        word = list(token[:-1]) + [token[-1] + "</w>"]
        while len(word) > 1:
            best = min(zip(word, word[1:]), key=lambda pr: self.ranks.get(pr, float("inf")))
            if best not in self.ranks:
                break
            first, second = best
            merged, i = [], 0
            while i < len(word):
                if i < len(word) - 1 and word[i] == first and word[i + 1] == second:
                    merged.append(first + second)
                    i += 2
                else:
                    merged.append(word[i])
                    i += 1
            word = merged
        out = " ".join(word)
        self.cache[token] = out
        return out

    def encode(self, text):
        ids = []
        for tok in re.findall(self.pat, _clean(text).lower()):
            tok = "".join(self.byte_encoder[b] for b in tok.encode("utf-8"))
            ids.extend(self.encoder[p] for p in self._bpe(tok).split(" "))
        return ids

    def decode(self, ids):
        text = "".join(self.decoder[int(i)] for i in ids)
        return bytearray(self.byte_decoder[c] for c in text).decode("utf-8", errors="replace").replace("</w>", " ")


class NativeTokenizer(SimpleTokenizer):
    """Same vocabulary and results as SimpleTokenizer; batches go through the host-thread tokeniser of the C-ABI library
    (`dh_bpe_encode`, declip_amd/csrc/bpe_host.hip).  Cleaning and lower-casing stay here (C-implemented builtins); captions
    that are not pure ASCII after cleaning come back flagged and take the Python path, so results never differ."""

    def __init__(self, bpe_path, threads=None):
        super().__init__(bpe_path)
        import os
        from . import lib as L
        self._L = L
        text = gzip.open(bpe_path).read()
        n_merges = 49152 - 256 - 2
        self._handle = L.load().dh_bpe_create(text, len(text), n_merges)
        if not self._handle:
            msg = L.load().dh_last_error()
            raise L.DeclipHipError("dh_bpe_create failed: %s" % (msg.decode() if msg else "?"))
        assert L.load().dh_bpe_vocab_size(self._handle) == len(self.encoder)
        self.threads = threads or max(1, min(8, (os.cpu_count() or 2) // 2))     # used for batches of >= 2048 captions

    def __del__(self):
        h, self._handle = getattr(self, "_handle", None), None
        if h:
            try:
                self._L.load().dh_bpe_destroy(h)
            except Exception:
                pass

    def encode_batch(self, texts, context_length=77):
        """list[str] -> LongTensor [n, context_length] (SOT, ids, EOT, zero pad; over-long keeps the final EOT)."""
        import ctypes
        import numpy as np
        n = len(texts)
        out = torch.zeros(n, context_length, dtype=torch.long)
        if n == 0:
            return out
        # ASCII captions: whitespace only separates tokens, so the whitespace collapse of whitespace_clean() cannot change
        # the ids and is skipped; anything else is left empty here, comes back flagged and takes the Python path below
        blobs = []
        for t in texts:
            u = _unescape(t)
            blobs.append(u.lower().encode("ascii") if u.isascii() else b"\xff")
        offsets = np.zeros(n + 1, dtype=np.int64)
        np.cumsum([len(b) for b in blobs], out=offsets[1:])
        status = np.zeros(n, dtype=np.int32)
        blob = b"".join(blobs)
        self._L.check(self._L.load().dh_bpe_encode(self._handle, blob, offsets.ctypes.data, n, context_length, out.data_ptr(),
                                                   status.ctypes.data, self.threads if n >= 2048 else 1), "dh_bpe_encode")
        sot, eot = self.encoder["<|startoftext|>"], self.encoder["<|endoftext|>"]
        for i in np.nonzero(status)[0]:
            toks = [sot] + self.encode(texts[i]) + [eot]
            if len(toks) > context_length:
                toks = toks[:context_length - 1] + [toks[-1]]
            out[i, :len(toks)] = torch.tensor(toks, dtype=torch.long)
        return out


def mask_token_ids(ids, vocab_size, mlm_probability=0.15, generator=None):
    """BERT-style masking on padded id rows [b, ctx] (mask_tokens.py:5-29): 15 % of the non-special,
    non-pad tokens are selected; 80 % -> <|mask|>, 10 % -> random id, 10 % unchanged; labels = -100
    elsewhere.  Returns (masked_ids, labels)."""
    ids = ids.clone()
    labels = ids.clone()
    mask_tok, sot, eot = vocab_size - 3, vocab_size - 2, vocab_size - 1
    eot_pos = ids.argmax(dim=-1, keepdim=True)
    pos = torch.arange(ids.shape[1], device=ids.device)[None, :]
    special = (ids == sot) | (ids == eot) | (ids == mask_tok) | (pos > eot_pos)
    prob = torch.full(ids.shape, mlm_probability)
    prob.masked_fill_(special.cpu(), 0.0)
    chosen = torch.bernoulli(prob, generator=generator).bool().to(ids.device)
    labels[~chosen] = -100
    replaced = torch.bernoulli(torch.full(ids.shape, 0.8), generator=generator).bool().to(ids.device) & chosen
    ids[replaced] = mask_tok
    rand = torch.bernoulli(torch.full(ids.shape, 0.5), generator=generator).bool().to(ids.device) & chosen & ~replaced
    words = torch.randint(vocab_size, ids.shape, generator=generator).to(ids.device)
    ids[rand] = words[rand]
    return ids, labels


def tokenize(tokenizer, texts, context_length=77, mask_type=None):
    """text_transformer.py:144-180: [SOT] + BPE + [EOT], zero padded; over-long captions keep the
    first context_length-1 tokens and the final EOT."""
    if isinstance(texts, str):
        texts = [texts]
    if hasattr(tokenizer, "encode_batch"):
        out = tokenizer.encode_batch(list(texts), context_length)
    else:
        sot, eot = tokenizer.encoder["<|startoftext|>"], tokenizer.encoder["<|endoftext|>"]
        out = torch.zeros(len(texts), context_length, dtype=torch.long)
        for i, t in enumerate(texts):
            toks = [sot] + tokenizer.encode(t) + [eot]
            if len(toks) > context_length:
                toks = toks[:context_length - 1] + [toks[-1]]
            out[i, :len(toks)] = torch.tensor(toks, dtype=torch.long)
    if mask_type is not None:
        if mask_type != "MLM":
            raise NotImplementedError(mask_type)
        return mask_token_ids(out, len(tokenizer.encoder))
    return out
