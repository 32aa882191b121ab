"""Loss functions of the contrastive path on the HIP engine.
Reference: prototype/loss_functions/loss.py:24-47 (ClipInfoCELoss), utils/misc.py:415-428 (accuracy)."""
import weakref

import torch
from torch import nn

from . import dist as dh_dist
from . import engine


def _is_lazy(x):
    return hasattr(x, "materialize") and hasattr(x, "Q")


class ClipInfoCELoss(nn.Module):
    """(CE(logits_per_image) + CE(logits_per_text)) / 2 with labels rank*b + arange(b).

    Accepts LazyLogits handles (fused streaming kernel, nothing materialised) or plain
    tensors (row-wise CE kernel on the materialised logits).  Returns (loss, labels) like the
    reference; the per-row top-1/top-5 hits of logits_per_image are kept in `.last_correct`
    so `accuracy()` needs no second pass.  `.last_correct` identifies the logits it belongs to by a WEAK reference: a strong one
    kept the previous step's whole autograd graph alive (features -> towers -> AccumulateGrad nodes), and an AccumulateGrad node
    that survives from an eager iteration carries that iteration's stream into a later hipGraph capture (declip_amd/graph.py)."""

    def __init__(self):
        super().__init__()
        self.last_correct = None
        self._labels = {}            # (batch, first label, device) -> labels: the same tensor every step (two launches less per step)

    def _label_row(self, bs, label0, dev):
        key = (bs, int(label0), str(dev))
        lab = self._labels.get(key)
        if lab is None:
            lab = label0 + torch.arange(bs, device=dev, dtype=torch.long)
            capturing = dev.type == "cuda" and torch.cuda.is_current_stream_capturing()
            if not capturing:            # (a tensor made inside a capture lives in the graph's pool: not kept beyond it)
                if len(self._labels) > 8:
                    self._labels.clear()
                self._labels[key] = lab
        return lab

    def forward(self, logits_per_image, logits_per_text):
        bs, l_bs = logits_per_image.shape
        dev = logits_per_image.device
        if _is_lazy(logits_per_image) and _is_lazy(logits_per_text):
            li, lt = logits_per_image, logits_per_text
            label0 = li.label0
            labels = self._label_row(bs, label0, dev)
            same_scale = li.scale is lt.scale
            if same_scale:
                row_loss, c1, c5 = engine.InfoNCEFn.apply(li.scale, label0, 2, li.Q, li.K, lt.Q, lt.K)
                loss = row_loss.mean()          # mean over 2b rows == (mean_i + mean_t) / 2
                self.last_correct = (weakref.ref(li), c1[0].detach(), c5[0].detach())
            else:
                rl_i, c1, c5 = engine.InfoNCEFn.apply(li.scale, label0, 1, li.Q, li.K)
                rl_t, _, _ = engine.InfoNCEFn.apply(lt.scale, label0, 1, lt.Q, lt.K)
                loss = (rl_i.mean() + rl_t.mean()) / 2
                self.last_correct = (weakref.ref(li), c1[0].detach(), c5[0].detach())
            return loss, labels
        if _is_lazy(logits_per_image):
            logits_per_image = logits_per_image.materialize()
        if _is_lazy(logits_per_text):
            logits_per_text = logits_per_text.materialize()
        if l_bs == bs:
            labels = torch.arange(bs, device=dev, dtype=torch.long)
        else:
            labels = dh_dist.get_rank() * bs + torch.arange(bs, device=dev, dtype=torch.long)
        rl_i, c1, c5 = engine.RowCEFn.apply(logits_per_image, labels)
        rl_t, _, _ = engine.RowCEFn.apply(logits_per_text, labels)
        self.last_correct = (weakref.ref(logits_per_image), c1.detach(), c5.detach())
        return (rl_i.mean() + rl_t.mean()) / 2, labels


def accuracy(output, target, topk=(1,), criterion=None):
    """utils/misc.py:415-428: precision@k in percent.  For LazyLogits (or when the criterion
    just saw `output`) the hit flags come from the fused loss kernel; otherwise from the
    row-CE kernel on the materialised logits.  Only k in {1, 5} exist on the hot path."""
    c1 = c5 = None
    if criterion is not None and criterion.last_correct is not None and criterion.last_correct[0]() is output:
        _, c1, c5 = criterion.last_correct
    elif _is_lazy(output):
        with torch.no_grad():
            _, c1, c5 = engine.InfoNCEFn.apply(output.scale.detach(), output.label0, 1, output.Q.detach(), output.K.detach())
            c1, c5 = c1[0], c5[0]
    else:
        with torch.no_grad():
            _, c1, c5 = engine.RowCEFn.apply(output.detach(), target)
    res = []
    for k in topk:
        if k == 1:
            res.append(c1.sum().reshape(1) * (100.0 / target.size(0)))
        elif k == 5:
            res.append(c5.sum().reshape(1) * (100.0 / target.size(0)))
        else:
            raise NotImplementedError("accuracy top-%d (only top-1/top-5 are on the hot path)" % k)
    return res


class NTXentLoss(nn.Module):
    """ConVIRT-style image-text NT-Xent monitor (reference: loss_functions/nt_xent_ConVIRT.py:4-86):
    alpha * CE(zi.zj^T / T) + (1 - alpha) * CE(zj.zi^T / T) on the LOCAL [b,b] logits with one-hot targets.
    Evaluated every step by the declip/filip/defilip solvers for logging (declip_solver.py:486-488)."""

    def __init__(self, batch_size=None, temperature=0.1, use_cosine_similarity=True, alpha_weight=0.75):
        super().__init__()
        self.temperature, self.alpha_weight = temperature, alpha_weight

    def forward(self, zis, zjs, norm=True, weights=1.0):
        if norm:
            zis, zjs = engine.L2NormFn.apply(zis, 1e-12), engine.L2NormFn.apply(zjs, 1e-12)
        scale = torch.full((1,), 1.0 / self.temperature, device=zis.device, dtype=torch.float32)
        rl, _, _ = engine.InfoNCEFn.apply(scale, 0, 2, zis, zjs, zjs, zis)
        return self.alpha_weight * rl[0].mean() + (1 - self.alpha_weight) * rl[1].mean()


def _ntxent(qs, k, label0s, excl0s, temperature):
    scale = torch.full((1,), 1.0 / temperature, device=k.device, dtype=torch.float32)
    rl, _, _ = engine.InfoNCEFn.apply(scale, (label0s, excl0s), len(qs), *[t for q in qs for t in (q, k)])
    return rl


class NT_Xent(nn.Module):
    """Local SimCLR NT-Xent (reference: loss_functions/nt_xent.py:6-44): rows [z_i; z_j] against themselves,
    positive of row i is i+b (and vice versa), the row's own entry is excluded; CE(sum) / 2b.  The reference
    materialises a [2b,2b,D] cosine tensor; here it is two fused InfoNCE pairs with a self-exclusion column."""

    def __init__(self, batch_size, temperature=0.5):
        super().__init__()
        self.batch_size, self.temperature = batch_size, temperature

    def forward(self, z_i, z_j):
        b = z_i.shape[0]
        qi, qj = engine.L2NormFn.apply(z_i, 1e-8), engine.L2NormFn.apply(z_j, 1e-8)
        k = torch.cat([qi, qj], dim=0)
        rl = _ntxent([qi, qj], k, [b, 0], [0, b], self.temperature)
        return rl.sum() / (2 * b)


class NT_Xent_gather(nn.Module):
    """Cross-rank SimCLR NT-Xent (reference: loss_functions/nt_xent.py:47-97): local rows [z_i; z_j] against the
    gathered [z_ib; z_jb]; positives at (i, rank*b+i+B) and (i+b, rank*b+i), self-pairs removed; CE(sum) / 2b.
    The reference builds a [2b, 2B, D] broadcast (2.1 G elements at b=512, B=4096); here nothing above
    [2B, D] exists."""

    def __init__(self, batch_size, temperature=0.1):
        super().__init__()
        self.batch_size, self.temperature = batch_size, temperature

    def forward(self, z_i, z_ib, z_j, z_jb, temperature=None):
        bs, l_bs = z_i.shape[0], z_ib.shape[0]
        assert bs == self.batch_size
        r0 = dh_dist.get_rank() * bs if l_bs != bs else 0
        qi, qj = engine.L2NormFn.apply(z_i, 1e-8), engine.L2NormFn.apply(z_j, 1e-8)
        k = torch.cat([engine.L2NormFn.apply(z_ib, 1e-8), engine.L2NormFn.apply(z_jb, 1e-8)], dim=0)
        rl = _ntxent([qi, qj], k, [r0 + l_bs, r0], [r0, r0 + l_bs], self.temperature)
        return rl.sum() / (2 * bs)
