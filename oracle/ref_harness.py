"""Import the UNMODIFIED reference (``/root/reference``) on CPU.

Only usable in the build container (``/root/reference`` does not exist on the
GPU box).  Used by ``oracle/gen_golden.py`` to generate ``tests/golden`` and by
``tests/test_oracle_vs_reference.py`` (auto-skipped when the reference is absent).

What has to be faked for the reference to import/run on CPU:
  * stub modules: ipdb, timm, ftfy, kestrel, tensorboardX, easydict, textaugment,
    torch._six (reference: model/clip.py:4,11; simple_tokenizer.py:6;
    data/image_reader.py:4; declip.py:154; utils/grad_clip.py:3)
  * identity ``Tensor.cuda`` / ``Module.cuda`` (hard-coded .cuda() at
    text_transformer.py:188, loss.py:40,42)
  * a synthetic BPE merges file giving the real vocabulary size 49409
    (simple_tokenizer.py:66-75); the real file is not in the tree
  * a 1-rank gloo process group when use_allgather=True (clip.py:34)
"""
import gzip
import os
import sys
import types

REFERENCE_ROOT = os.environ.get("DECLIP_REFERENCE_ROOT", "/root/reference")
_BPE_PATH = "/tmp/declip_oracle_bpe_simple_vocab_16e6.txt.gz"
_loaded = None


def reference_available():
    return os.path.isdir(os.path.join(REFERENCE_ROOT, "prototype", "model"))


def synthetic_bpe_path():
    """header + 48894 unique merge lines -> len(encoder) == 49409."""
    if not os.path.exists(_BPE_PATH):
        n_merges = 49152 - 256 - 2
        lines = ["#version: synthetic"] + ["q%d z" % i for i in range(n_merges)] + [""]
        with gzip.open(_BPE_PATH, "wb") as f:
            f.write("\n".join(lines).encode("utf-8"))
    return _BPE_PATH


class _EasyDict(dict):
    def __init__(self, d=None, **kw):
        super().__init__()
        d = dict(d or {}, **kw)
        for k, v in d.items():
            self[k] = v

    def __setitem__(self, k, v):
        if isinstance(v, dict) and not isinstance(v, _EasyDict):
            v = _EasyDict(v)
        super().__setitem__(k, v)

    __setattr__ = __setitem__

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError:
            raise AttributeError(k)


AUG_KEY_OFFSET = 100000


class _IdentityEDA:
    """Deterministic stand-in for textaugment.EDA (reference declip.py:203-212): strings pass through;
    integer caption KEYS (the oracle feeds row indices instead of strings) map to key + AUG_KEY_OFFSET,
    which the patched tokenize() resolves to the seeded "augmented" ids."""

    def _aug(self, s):
        return s + AUG_KEY_OFFSET if isinstance(s, int) else s

    def synonym_replacement(self, s):
        return self._aug(s)

    def random_swap(self, s):
        return self._aug(s)

    def random_deletion(self, s, p=0.1):
        return self._aug(s)


def load_reference():
    """Returns the imported reference ``prototype`` package (cached)."""
    global _loaded
    if _loaded is not None:
        return _loaded
    if not reference_available():
        raise RuntimeError("reference tree not present at %s" % REFERENCE_ROOT)
    sys.dont_write_bytecode = True  # never write under /root/reference
    import torch
    import torch.nn as nn

    def stub(name, **attrs):
        m = types.ModuleType(name)
        m.__dict__.update(attrs)
        sys.modules[name] = m
        return m

    stub("ipdb", set_trace=lambda *a, **k: None)
    stub("timm")
    stub("ftfy", fix_text=lambda s: s)
    stub("kestrel")
    stub("tensorboardX", SummaryWriter=object)
    stub("easydict", EasyDict=_EasyDict)
    stub("textaugment", EDA=_IdentityEDA)
    if "torch._six" not in sys.modules:
        six = stub("torch._six", inf=float("inf"))
        torch._six = six
    torch.Tensor.cuda = lambda self, *a, **k: self
    nn.Module.cuda = lambda self, *a, **k: self

    # another 'prototype'/'linklink' (our drop-in shims) must not shadow the reference
    for k in [k for k in sys.modules if k == "prototype" or k.startswith("prototype.")
              or k == "linklink" or k.startswith("linklink.")]:
        del sys.modules[k]
    sys.path.insert(0, REFERENCE_ROOT)
    try:
        import prototype.model as ref_model  # noqa: F401
        import prototype.loss_functions as ref_loss  # noqa: F401
        import prototype  # noqa: F401
        import linklink
    finally:
        sys.path.remove(REFERENCE_ROOT)
    linklink.barrier = lambda: None  # uses torch.cuda.IntTensor (linklink/__init__.py:32)
    _loaded = types.SimpleNamespace(
        prototype=sys.modules["prototype"], model=sys.modules["prototype.model"],
        loss=sys.modules["prototype.loss_functions"], linklink=linklink,
        modules={k: v for k, v in sys.modules.items()
                 if k.startswith("prototype") or k.startswith("linklink")})
    # un-register so that a later `import prototype` resolves to the drop-in again
    for k in list(_loaded.modules):
        del sys.modules[k]
    return _loaded


def ensure_gloo_group():
    import torch.distributed as dist
    if not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        dist.init_process_group("gloo", rank=0, world_size=1)
