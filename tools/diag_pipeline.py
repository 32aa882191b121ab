"""Where does a step with the input pipeline in the loop spend its time?  (host time of next(), of the step's enqueue, GPU time of the
intake kernels)"""
import itertools
import os
import sys
import tempfile
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from declip_amd import ops, synth
from declip_amd.bpe import NativeTokenizer
from declip_amd.prefetch import DataPrefetcher, crops_on_device

b = 512
dev = torch.device("cuda", 0)
pool = synth.synth_decoded_batches(b, n_batches=4, seed=0)
tok = NativeTokenizer(synth.synthetic_bpe_file(os.path.join(tempfile.gettempdir(), "dh_synthetic_bpe.txt.gz")))
# 1. the worker's preparation alone
pf = DataPrefetcher(iter([]), dev, tokenizer=tok, context_length=77, image_size=224)
t = time.time()
for i in range(8):
    item = pf._prepare(pool[i % 4])
print("worker _prepare: %.2f ms per batch" % ((time.time() - t) / 8 * 1e3))
# 2. upload + on-GPU crop of one prepared batch (GPU time)
st = torch.cuda.Stream()
e0, e1, e2 = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
for rep in range(3):
    with torch.cuda.stream(st):
        e0.record()
        d = {k: (v.to(dev, non_blocking=True) if torch.is_tensor(v) else v) for k, v in item.items()}
        e1.record()
        d = crops_on_device(d, (224, 224))
        e2.record()
    torch.cuda.synchronize()
    print("upload %.2f ms, resized crop %.2f ms (pinned canvas: %s)" % (e0.elapsed_time(e1), e1.elapsed_time(e2), item["images"].is_pinned()))
# 3. the host side of next()
pf = DataPrefetcher(itertools.cycle(pool), dev, tokenizer=tok, context_length=77, image_size=224)
time.sleep(0.5)
ts = []
for i in range(12):
    t = time.time()
    nb = pf.next()
    ts.append((time.time() - t) * 1e3)
    torch.cuda.synchronize()
    time.sleep(0.02)
print("next() host ms:", ["%.2f" % x for x in ts])
pf.close()
