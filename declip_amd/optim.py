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
                lrs.append(-1.0), wds.append(0.0)       # lr < 0 = inactive segment (dh_adamw_segmented)
            else:
                g = self.param_groups[gi]
                lrs.append(float(g["lr"])), wds.append(float(g["weight_decay"]))
        hp = (tuple(lrs), tuple(wds))
        if hp != self._cached_hp:
            self._seg_lr.copy_(torch.tensor(lrs, dtype=torch.float32), non_blocking=True)
            self._seg_wd.copy_(torch.tensor(wds, dtype=torch.float32), non_blocking=True)
            self._cached_hp = hp

    # ---- checkpoint layout of torch.optim.AdamW (what the reference writes under 'optimizer', clip_solver.py:655): parameters are
    # numbered in param_groups order; per parameter 'step', 'exp_avg', 'exp_avg_sq'
    def state_dict(self):
        flat, state, groups, idx = self.flat, {}, [], 0
        for g in self.param_groups:
            ids = []
            for p in g["params"]:
                o, n = flat.index[id(p)]
                state[idx] = dict(step=torch.tensor(float(self.step_count)), exp_avg=self.m[o:o + n].view(p.shape).detach().cpu().clone(),
                                  exp_avg_sq=self.v[o:o + n].view(p.shape).detach().cpu().clone())
                ids.append(idx)
                idx += 1
            groups.append(dict({k: v for k, v in g.items() if k != "params"}, params=ids))
        return dict(state=state, param_groups=groups)

    def load_state_dict(self, sd):
        """accepts the torch.optim.AdamW layout (a reference / model-zoo checkpoint, or state_dict() above); anything that does
        not line up with this optimizer's parameters raises instead of silently dropping the moments."""
        flat = self.flat
        mine = list(self.param_groups)
        if len(sd["param_groups"]) != len(mine):
            # LEGACY layout (checkpoints this engine wrote before round 3): typed groups without parameters were not emitted.  The
            # groups that hold parameters come in the same order in both layouts: line those up, leave the empty ones alone.
            non_empty = [g for g in mine if len(g["params"]) > 0]
            if len(sd["param_groups"]) == len(non_empty) and all(len(sg["params"]) == len(g["params"]) for g, sg in zip(non_empty, sd["param_groups"])):
                mine = non_empty
            else:
                raise ValueError("optimizer state has %d param groups, this optimizer %d (%d of them non-empty): neither the reference "
                                 "layout (utils/misc.py:386-393: every typed group, empty ones too) nor the legacy one (empty groups "
                                 "dropped)" % (len(sd["param_groups"]), len(self.param_groups), len(non_empty)))
        steps = set()
        with torch.no_grad():
            for g, sg in zip(mine, sd["param_groups"]):
                if len(sg["params"]) != len(g["params"]):
                    raise ValueError("optimizer state: a param group has %d parameters, expected %d" % (len(sg["params"]), len(g["params"])))
                for k, v in sg.items():
                    if k != "params":
                        g[k] = tuple(v) if k == "betas" else v
                for p, idx in zip(g["params"], sg["params"]):
                    st = sd["state"].get(idx)
                    if st is None:                      # torch keeps no state for a parameter that never had a gradient
                        continue
                    if tuple(st["exp_avg"].shape) != tuple(p.shape):
                        raise ValueError("optimizer state %d has shape %s, parameter %s has %s" % (
                            idx, tuple(st["exp_avg"].shape), flat.names.get(id(p), "?"), tuple(p.shape)))
                    o, n = flat.index[id(p)]
                    self.m[o:o + n].view(p.shape).copy_(st["exp_avg"])
                    self.v[o:o + n].view(p.shape).copy_(st["exp_avg_sq"])
                    steps.add(int(float(st["step"])))
        if len(steps) > 1:
            raise ValueError("optimizer state with per-parameter step counts %s: the fused update keeps ONE step count" % sorted(steps))
        self.step_count = steps.pop() if steps else 0
        self._cached_hp = None

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
        for g in self.param_groups[1:]:                 # ONE fused launch: betas / eps are launch constants, lr / weight decay per segment
            if tuple(g["betas"]) != tuple(g0["betas"]) or g["eps"] != g0["eps"]:
                raise NotImplementedError("per-group betas / eps overrides (pconfig) are not supported by the fused AdamW: "
                                          "group has betas %s eps %s, group 0 betas %s eps %s" % (g["betas"], g["eps"], g0["betas"], g0["eps"]))
        b1, b2 = g0["betas"]
        ops.adamw_segmented(flat.flat_p, flat.flat_g, self.m, self.v, flat.flat_b, self._seg_start, self._seg_lr,
                            self._seg_wd, b1, b2, g0["eps"], self.step_count, grad_scale)
        flat.mirror_written_by_optimizer()
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
