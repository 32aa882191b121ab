"""Zero-shot classification on the HIP towers (reference: solver/clip_solver.py:675-737 `evaluate`,
data/datasets/clip_dataset.py:238-288 `_get_label_text` / `get_label_texts`).

Same arithmetic as the reference, arranged for the GPU:
  * class embeddings: every prompt of every class is encoded, L2-normalised, averaged per class and re-normalised
    (clip_solver.py:692-699).  The reference encodes one class at a time (a batch of `prompts_num` texts, 1..80 rows: a
    launch-bound trickle on a 256-CU part); here the `label_num * prompts_num` prompts go through the text tower in chunks of
    `text_chunk` rows and the per-class mean is one segmented reduction over the [label_num, prompts_num, E] view.
  * images: encode -> L2-normalise -> logits = img @ class_emb^T (fp32 GEMM, no logit scale: clip_solver.py:712-714) ->
    scores = softmax(logits) @ ensemble_matrix (identity in the reference's datasets) -> top-k.
Forward only; runs under torch.no_grad().  The towers are the same HIP path as training -- there is no CPU fallback.
"""
import torch

from . import engine, ops

__all__ = ["PROMPT_SETS", "prompts_for", "label_texts", "class_embeddings", "classify", "ZeroShotMeter", "SyntheticZeroShotData"]

# Own template sets (the reference reads them from prompts/query_pattern_prompt{1,6,8,80}; a `file:<path>` ensemble reads any
# template file with one template per line and `{0}` as the class-name slot, which is how the 80-template set is supplied).
PROMPT_SETS = {
    "simple": ["a photo of a {0}."],
    "prompt6": ["a photo of a {0}.", "a photo of a big {0}.", "a photo of a small {0}.", "a picture of a {0}.",
                "a picture of a big {0}.", "a picture of a small {0}."],
    "prompt8": ["a photo of a {0}.", "a photo of the {0}.", "a picture of a {0}.", "a picture of the {0}.",
                "a close-up photo of a {0}.", "a cropped photo of a {0}.", "a bright photo of a {0}.", "a good photo of a {0}."],
}


def prompts_for(name, ensemble="simple"):
    """clip_dataset.py:238-258: the prompt texts of one class.  'cc' = the bare name; 'file:<path>' = templates from a file."""
    if ensemble == "cc":
        return [name]
    if ensemble.startswith("file:"):
        with open(ensemble[5:]) as f:
            templates = [ln.strip() for ln in f.readlines()]
        templates = [t for t in templates if t]
    elif ensemble in PROMPT_SETS:
        templates = PROMPT_SETS[ensemble]
    else:
        raise NotImplementedError(ensemble)
    return [t.replace("{0}", name) for t in templates]


def label_texts(label_to_name, ensemble="simple"):
    """clip_dataset.py:260-288: prompts of all classes in ascending label order + the ensemble matrix (identity)."""
    labels = sorted(label_to_name)
    texts = []
    for lb in labels:
        texts.extend(prompts_for(label_to_name[lb], ensemble))
    return texts, torch.eye(len(labels))


def _sync_params(model):
    """The towers compute from the bf16 mirror of the flat parameter store; training forwards refresh it, a bare
    `encode_text` call does not -- refresh it once here so evaluation sees the current master weights."""
    st = model.__dict__.get("_flat_store")
    if st is not None:
        st.begin_step()


def _encode_text_chunked(model, texts, chunk):
    n = texts.shape[0] if torch.is_tensor(texts) else len(texts)
    outs = []
    for i in range(0, n, chunk):
        outs.append(model.encode_text(texts[i:i + chunk]).float())
    return outs[0] if len(outs) == 1 else torch.cat(outs, 0)


@torch.no_grad()
def class_embeddings(model, texts, label_num, text_chunk=2048):
    """[label_num, E] fp32 unit vectors: normalise each prompt, mean over the class's prompts, normalise (clip_solver.py:692-701).
    texts: list of label_num * prompts_num strings or an int64 [label_num * prompts_num, ctx] tensor of token ids, class-major."""
    n = texts.shape[0] if torch.is_tensor(texts) else len(texts)
    if label_num <= 0 or n % label_num:
        raise ValueError("%d prompts do not divide into %d classes" % (n, label_num))
    _sync_params(model)
    feats = _encode_text_chunked(model, texts, text_chunk)
    feats = engine.L2NormFn.apply(feats.contiguous(), 0.0)
    mean = feats.view(label_num, n // label_num, feats.shape[1]).mean(dim=1)
    return engine.L2NormFn.apply(mean.contiguous(), 0.0)


@torch.no_grad()
def classify(model, images, class_emb, ensemble_matrix=None, topk=(1, 5), return_dense=False):
    """One batch: returns dict(logits [b,C], scores [b,C'], prediction [b], topk [b,max k]) (clip_solver.py:704-719)."""
    if return_dense:
        img = model.encode_image(images, return_dense=True)[0]
    else:
        img = model.encode_image(images)
    img = engine.L2NormFn.apply(img.float().contiguous(), 0.0)
    logits = ops.gemm(img, class_emb.contiguous())                    # fp32 [b, C]
    scores = torch.softmax(logits, dim=1)
    if ensemble_matrix is not None and not _is_identity(ensemble_matrix):
        scores = ops.gemm(scores.contiguous(), ensemble_matrix.to(scores).t().contiguous())
    k = min(max(topk), logits.shape[1])
    top = logits.topk(k, dim=1).indices
    return {"logits": logits, "scores": scores, "prediction": top[:, 0].contiguous(), "topk": top}


def _is_identity(m):
    return m.dim() == 2 and m.shape[0] == m.shape[1] and bool(torch.equal(m.cpu().float(), torch.eye(m.shape[0])))


class ZeroShotMeter:
    """Top-k accuracy accumulated on the device (one host read at the end; the reference dumps every sample to a text file
    and re-reads it on rank 0, imagenet_dataset-style evaluators -- file dumping is I/O pipeline, out of scope)."""

    def __init__(self, device, topk=(1, 5)):
        self.topk = tuple(topk)
        self.hits = torch.zeros(len(self.topk), device=device, dtype=torch.float64)
        self.count = torch.zeros(1, device=device, dtype=torch.float64)

    def update(self, top, labels):
        match = top.eq(labels.to(top.device).view(-1, 1))
        for i, k in enumerate(self.topk):
            self.hits[i] += match[:, :k].any(dim=1).sum()
        self.count += match.shape[0]

    def result(self, reduce=True):
        from . import dist as dh_dist
        hits, count = self.hits.clone(), self.count.clone()
        if reduce and dh_dist.is_dist():
            import torch.distributed as tdist
            packed = torch.cat([hits, count]).float()      # counts < 2^24 per job: exact in fp32 (gloo and RCCL both take it)
            tdist.all_reduce(packed)
            hits, count = packed[:-1].double(), packed[-1:].double()
        out = {"top%d" % k: float(100.0 * hits[i] / count.clamp(min=1)) for i, k in enumerate(self.topk)}
        out["count"] = int(count)
        return out


class SyntheticZeroShotData:
    """Seeded stand-in for an ImageNet-style zero-shot set (`data.test.type: synthetic`): `label_num` classes whose
    prompts are token-id rows, and batches of random images with random labels.  Accuracy on it is chance; it exists so the
    evaluate path (shapes, chunking, sharding over ranks, throughput) can be exercised without a dataset."""

    def __init__(self, label_num=1000, prompts_num=1, batch_size=256, batches=4, res=224, ctx=77, seed=0, rank=0, world=1):
        from . import synth
        self.label_num, self.prompts_num, self.batch_size, self.res = label_num, prompts_num, batch_size, res
        self.tokens = synth.synth_tokens(label_num * prompts_num, ctx=ctx, seed=seed + 77, max_len=12)
        self.batches = [b for b in range(batches) if b % world == rank]
        self.seed = seed

    def get_label_texts(self):
        return self.tokens, torch.eye(self.label_num)

    def __len__(self):
        return len(self.batches)

    def __iter__(self):
        from . import synth
        for b in self.batches:
            g = torch.Generator().manual_seed(31 * self.seed + b)
            yield {"images": synth.synth_images(self.batch_size, res=self.res, seed=self.seed * 1000 + b),
                   "labels": torch.randint(0, self.label_num, (self.batch_size,), generator=g)}
