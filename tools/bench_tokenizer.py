"""Captions/s of the Python tokeniser vs the host-thread tokeniser of the C-ABI library (CPU only; vocabulary = the merges
learned by tests/test_tokenizer.py's helper, since the real vocabulary file is not in the tree)."""
import os, random, sys, tempfile, time, pathlib
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from declip_amd import bpe
from test_tokenizer import _trained_vocab
path = _trained_vocab(pathlib.Path(tempfile.mkdtemp()))
rng = random.Random(1)
vocab = ("a photo of the cat dog sitting on mat quick brown fox jumps over lazy it's they're photograph theatre weather garden "
         "house windows playing near two three 42 2021 beautiful sunset beach people walking street city night").split()
extra = ["zx%dq" % i for i in range(20000)]          # a long tail of rare words (cache misses)
caps = [" ".join(rng.choice(vocab) if rng.random() < 0.9 else rng.choice(extra) for _ in range(rng.randint(5, 25))) + "." for _ in range(20000)]
py, nat = bpe.SimpleTokenizer(path), bpe.NativeTokenizer(path)
for name, tok, n in (("python", py, 4000), ("native x%d threads" % nat.threads, nat, 20000)):
    bpe.tokenize(tok, caps[:512])
    t0 = time.perf_counter()
    for i in range(0, n, 512):
        bpe.tokenize(tok, caps[i:i + 512])
    dt = time.perf_counter() - t0
    print("%-22s %8.0f captions/s" % (name, n / dt))
nat.threads = 1
t0 = time.perf_counter()
for i in range(0, 20000, 512):
    bpe.tokenize(nat, caps[i:i + 512])
print("%-22s %8.0f captions/s" % ("native x1 thread", 20000 / (time.perf_counter() - t0)))
