"""Fused flat AdamW over the engine's flat parameter buffer (one launch per step).

Reference: the shipped configs use torch.optim.AdamW through prototype/optimizer/__init__.py:18-26
with per-group weight decay from utils/misc.py:267-412; `linklink.optim.FusedFP16AdamW` is the
(missing) fused op the reference names (optimizer/__init__.py:8-15).  Update rule == torch.optim.AdamW.
"""
import torch

from . import ops


class FlatAdamW(torch.optim.Optimizer):
    """Drop-in for torch.optim.AdamW on parameters owned by an engine.FlatParams store."""

    def __init__(self, params, flat, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2, amsgrad=False):
        if amsgrad:
            raise NotImplementedError("amsgrad (all shipped configs use amsgrad: False)")
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))
        self.flat = flat.ensure()
        dev = flat.flat_p.device
        self.m = torch.zeros_like(flat.flat_p)
        self.v = torch.zeros_like(flat.flat_p)
        self.step_count = 0
        # segment table: one entry per parameter (sorted by flat offset), merged when hyper-parameters match
        self._param_group = {}
        for gi, g in enumerate(self.param_groups):
            for p in g["params"]:
                self._param_group[id(p)] = gi
        entries = sorted((flat.index[id(p)][0], id(p)) for p in flat.params)
        self._seg_pid = [pid for _, pid in entries]
        self._seg_start = torch.tensor([o for o, _ in entries], dtype=torch.int64, device=dev)
        self._seg_lr = torch.zeros(len(entries), dtype=torch.float32, device=dev)
        self._seg_wd = torch.zeros(len(entries), dtype=torch.float32, device=dev)
        self._cached_hp = None
        self._by_id = {id(p): p for p in flat.params}

    def _refresh_table(self):
        lrs, wds = [], []
        for pid in self._seg_pid:
            gi = self._param_group.get(pid)
            p = self._by_id[pid]
            if gi is None or not p.requires_grad or getattr(p, "_dh_grad_none", False):   # torch.optim.AdamW skips grad-None parameters
                lrs.append(0.0), wds.append(0.0)
            else:
                g = self.param_groups[gi]
                lrs.append(float(g["lr"])), wds.append(float(g["weight_decay"]))
        hp = (tuple(lrs), tuple(wds))
        if hp != self._cached_hp:
            self._seg_lr.copy_(torch.tensor(lrs, dtype=torch.float32), non_blocking=True)
            self._seg_wd.copy_(torch.tensor(wds, dtype=torch.float32), non_blocking=True)
            self._cached_hp = hp

    def zero_grad(self, set_to_none=True):
        # gradients live in flat_g; the engine zeroes it at the start of the next backward
        for p in self.flat.params:
            p.grad = None

    @torch.no_grad()
    def step(self, closure=None, grad_scale=1.0):
        loss = closure() if closure is not None else None
        flat = self.flat
        flat.join_streams()
        self._refresh_table()
        self.step_count += 1
        g0 = self.param_groups[0]
        b1, b2 = g0["betas"]
        ops.adamw_segmented(flat.flat_p, flat.flat_g, self.m, self.v, flat.flat_b, self._seg_start, self._seg_lr,
                            self._seg_wd, b1, b2, g0["eps"], self.step_count, grad_scale)
        flat.mirror_fresh = flat.flat_b is not None
        return loss


def build_adamw(model, flat=None, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2, no_decay=None):
    """Param groups in the spirit of the shipped `pconfig` (yfcc15m_vit_clip/config.yaml:34-47):
    LayerNorm/BatchNorm weights+biases, all biases and logit_scale get weight_decay 0."""
    flat = flat or model.__dict__["_flat_store"]
    flat.ensure()
    decay, nodecay = [], []
    for name, p in model.named_parameters():
        if not p.requires_grad:
            continue
        leaf = name.rsplit(".", 1)[-1]
        if p.dim() <= 1 or leaf.endswith("bias") or "logit_scale" in name:
            nodecay.append(p)
        else:
            decay.append(p)
    groups = [dict(params=decay, weight_decay=weight_decay), dict(params=nodecay, weight_decay=0.0)]
    return FlatAdamW(groups, flat, lr=lr, betas=betas, eps=eps, weight_decay=weight_decay)
