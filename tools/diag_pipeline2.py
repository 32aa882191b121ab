"""Which part of the input pipeline makes the step's HOST side slow?  The CLIP bf16 step of bench.py under variants of the intake:
  resident      one device batch, reused
  fresh_same    the same captions cloned (and tagged) every step: a new PackedCaptions per step, same row count
  fresh_vary    captions of 6 batches in turn: a new row count (GEMM M) per step
  pipe_notok    DataPrefetcher over pre-tokenised pinned batches (worker thread idle but for the queue)
  pipe_samestr  DataPrefetcher with tokenisation, uploads on the CURRENT stream
  pipe          DataPrefetcher as bench.py --pipeline 1 uses it
host / total ms per step, and the host time of the step's sections."""
import itertools
import os
import sys
import tempfile
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from declip_amd import bpe, dist as dh_dist, synth
from declip_amd.bpe import NativeTokenizer
from declip_amd.loss import ClipInfoCELoss
from declip_amd.optim import build_adamw
from declip_amd.prefetch import DataPrefetcher
from declip_amd.testing import build_clip

torch.cuda.set_device(0)
if os.environ.get("DIAG_THREADS"):
    torch.set_num_threads(int(os.environ["DIAG_THREADS"]))


def nthreads():
    return [l for l in open("/proc/self/status") if l.startswith("Threads")][0].strip()


print("cpu.max:", open("/sys/fs/cgroup/cpu.max").read().strip() if os.path.exists("/sys/fs/cgroup/cpu.max") else "n/a",
      " affinity:", len(os.sched_getaffinity(0)), " torch threads:", torch.get_num_threads(), nthreads())
dev = torch.device("cuda", 0)
b = 512
torch.manual_seed(1234)
model = build_clip(synth.VITB32, dtype="bf16", use_allgather=False, seed=0, load_synth=False)
wrapped = dh_dist.DistModule(model, sync=False)
opt = build_adamw(model, lr=1e-4, betas=(0.9, 0.98), eps=1e-8, weight_decay=0.1)
crit = ClipInfoCELoss()
images = synth.synth_images(b, seed=0).to(dev)
ids = synth.synth_tokens(b, seed=0).to(dev)
batch = {"images": images, "captions": ids}
pool = synth.synth_decoded_batches(b, n_batches=6, seed=0)
tok = NativeTokenizer(synth.synthetic_bpe_file(os.path.join(tempfile.gettempdir(), "dh_synthetic_bpe.txt.gz")))
tok_pool = []
for p in pool:
    q = dict(p)
    q["captions"] = bpe.tokenize(tok, [c if isinstance(c, str) else c[0] for c in p["captions"]], 77).pin_memory()
    tok_pool.append(q)
dev_caps = []
for q in tok_pool:
    c = q["captions"].to(dev)
    rows = int((q["captions"].argmax(dim=-1) + 1).sum())
    dev_caps.append((c, rows))
print("rows per batch:", [r for _, r in dev_caps], " resident synthetic:", int((ids.argmax(-1) + 1).sum()))

sec = {}


def tick(name, t):
    sec[name] = sec.get(name, 0.0) + (time.perf_counter() - t)


def run(name, feed, steps=20, warm=8):
    global sec
    for it in range(warm + steps):
        if it == warm:
            torch.cuda.synchronize()
            sec = {}
            t0 = time.perf_counter()
        t = time.perf_counter(); feed(it); tick("feed", t)
        t = time.perf_counter(); opt.zero_grad(); tick("zero", t)
        t = time.perf_counter(); li, lt = wrapped(batch); loss, _ = crit(li, lt); tick("fwd", t)
        t = time.perf_counter(); loss.backward(); tick("bwd", t)
        t = time.perf_counter(); wrapped.sync_gradients(); opt.step(); tick("opt", t)
    host = time.perf_counter() - t0
    torch.cuda.synchronize()
    tot = time.perf_counter() - t0
    print("%-13s host %6.2f  total %6.2f ms/step   | " % (name, host / steps * 1e3, tot / steps * 1e3)
          + "  ".join("%s %.2f" % (k, v / steps * 1e3) for k, v in sec.items()), flush=True)


def feed_resident(it):
    batch["images"], batch["captions"] = images, ids


def feed_fresh_same(it):
    c = dev_caps[0][0].clone()
    c._dh_rows = (c._version, dev_caps[0][1])
    batch["images"], batch["captions"] = images, c


def feed_fresh_vary(it):
    c0, rows = dev_caps[it % 6]
    c = c0.clone()
    c._dh_rows = (c._version, rows)
    batch["images"], batch["captions"] = images, c


def feeder(pf):
    def f(it):
        nxt = pf.next()
        batch["images"], batch["captions"] = nxt["images"], nxt["captions"]
    return f


run("resident", feed_resident)
run("fresh_same", feed_fresh_same)
run("fresh_vary", feed_fresh_vary)
run("resident", feed_resident)
pf = DataPrefetcher(itertools.cycle(tok_pool), dev, tokenizer=None, context_length=77, image_size=224)
run("pipe_notok", feeder(pf)); pf.close(); print(nthreads())
pf = DataPrefetcher(itertools.cycle(pool), dev, tokenizer=tok, context_length=77, image_size=224)
pf.stream = torch.cuda.current_stream()
run("pipe_samestr", feeder(pf)); pf.close()
pf = DataPrefetcher(itertools.cycle(pool), dev, tokenizer=tok, context_length=77, image_size=224)
run("pipe", feeder(pf)); pf.close()
run("resident", feed_resident)
