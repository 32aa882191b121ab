"""Training solver for the contrastive step (CLIP / DeCLIP / SLIP / FILIP / DeFILIP) on the HIP engine.

Drop-in for `python -m prototype.solver.{clip,declip,slip,filip,defilip}_solver --config config.yaml`
(reference: solver/clip_solver.py:89-764 and siblings).  What is kept: the YAML contract (model / dist /
grad_clip / optimizer(+pconfig) / lr_scheduler / loss weights / saver), the step order
(lr step -> forward -> losses / world_size -> logit_scale clamp -> backward -> grad sync -> optimizer ->
clamp), iteration-based training, checkpoint key layout ('model' with 'module.' prefix, 'optimizer', 'last_iter').
What changes: no per-meter host sync (meters are read every print_freq), no barriers in the step, gradient
reduction is the engine's bucketed flat all-reduce.  Data: `data.read_from: fake` / `data.type: synthetic`
produce seeded synthetic batches resident on the GPU (the I/O pipeline is out of scope, DESIGN_HISTORY.md s7); any
iterable of reference-style batch dicts can be injected with `ClsSolver(config, train_loader=...)`.
"""
import argparse
import glob
import logging
import math
import os
import time

import torch
import yaml

from . import dist as dh_dist
from . import steps, synth
from .heads import SimsiamLoss
from .loss import ClipInfoCELoss, NT_Xent, NT_Xent_gather, NTXentLoss
from .meters import AverageMeter, reduce_update_packed
from .optim import FlatAdamW


class AttrDict(dict):
    """EasyDict stand-in (reference utils/misc.py:65-70 parses YAML into EasyDict)."""

    def __init__(self, d=None):
        super().__init__()
        for k, v in (d or {}).items():
            self[k] = v

    def __setitem__(self, k, v):
        if isinstance(v, dict) and not isinstance(v, AttrDict):
            v = AttrDict(v)
        super().__setitem__(k, v)

    __setattr__ = __setitem__

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError:
            raise AttributeError(k)


def parse_config(path):
    with open(path) as f:
        return AttrDict(yaml.safe_load(f))


# --------------------------------------------------------------------------------------- lr schedule
class WarmUpLRScheduler(object):
    """lr_scheduler/scheduler.py:7-84: linear warm-up base_lr -> warmup_lr over warmup_steps, then `_after_warmup()` (the target
    learning rate as a function of last_iter); every param group is scaled by target / base_lr relative to its own initial lr.
    The four shipped laws differ only in that function: Step (:87-144), StepDecay (:147-197), Cosine (:200-249), Poly (:252-300)."""

    def __init__(self, optimizer, base_lr, warmup_lr, warmup_steps, max_iter=None, last_iter=0, **law):
        assert warmup_steps >= 2 or warmup_steps == 0
        if warmup_steps == 0:
            assert base_lr == warmup_lr
        self.optimizer, self.max_iter = optimizer, max_iter
        self.base_lr, self.warmup_lr, self.warmup_steps, self.last_iter = base_lr, warmup_lr, warmup_steps, last_iter
        self.law = law
        for g in optimizer.param_groups:
            g.setdefault("initial_lr", g["lr"])
        self.base_lrs = [g["initial_lr"] for g in optimizer.param_groups]

    def _after_warmup(self):
        raise NotImplementedError

    def _target(self):
        it = self.last_iter
        if self.warmup_steps >= 2 and it < self.warmup_steps:
            return (self.warmup_lr - self.base_lr) / (self.warmup_steps - 1) * (it - 1) + self.base_lr
        return self._after_warmup()

    def step(self, this_iter=None):
        self.last_iter = self.last_iter + 1 if this_iter is None else this_iter
        scale = self._target() / self.base_lr
        for g, b in zip(self.optimizer.param_groups, self.base_lrs):
            g["lr"] = scale * b

    def get_lr(self):
        return [g["lr"] for g in self.optimizer.param_groups]


class CosineLRScheduler(WarmUpLRScheduler):
    def __init__(self, optimizer, max_iter, min_lr, base_lr, warmup_lr, warmup_steps, last_iter=0):
        super().__init__(optimizer, base_lr, warmup_lr, warmup_steps, max_iter, last_iter)
        self.min_lr = min_lr

    def _after_warmup(self):
        ratio = (self.last_iter - self.warmup_steps) / (self.max_iter - self.warmup_steps)
        return self.min_lr + (self.warmup_lr - self.min_lr) * (1 + math.cos(math.pi * ratio)) / 2


class StepLRScheduler(WarmUpLRScheduler):
    def __init__(self, optimizer, lr_steps, lr_mults, base_lr, warmup_lr, warmup_steps, max_iter, last_iter=0):
        super().__init__(optimizer, base_lr, warmup_lr, warmup_steps, max_iter, last_iter)
        if len(lr_steps) != len(lr_mults) or list(lr_steps) != sorted(lr_steps):
            raise ValueError("lr_steps must be increasing and match lr_mults: %s vs %s" % (lr_steps, lr_mults))
        self.lr_steps, self.cum = list(lr_steps), [1.0]
        for m in lr_mults:
            self.cum.append(self.cum[-1] * m)

    def _after_warmup(self):
        import bisect
        return self.warmup_lr * self.cum[bisect.bisect_right(self.lr_steps, self.last_iter)]


class StepDecayLRScheduler(WarmUpLRScheduler):
    def __init__(self, optimizer, step_size, decay, base_lr, warmup_lr, warmup_steps, max_iter, last_iter=0):
        super().__init__(optimizer, base_lr, warmup_lr, warmup_steps, max_iter, last_iter)
        self.step_size, self.decay = step_size, decay

    def _after_warmup(self):
        return self.decay ** ((self.last_iter - self.warmup_steps) // self.step_size) * self.warmup_lr


class PolynomialLRScheduler(WarmUpLRScheduler):
    def __init__(self, optimizer, power, max_iter, base_lr, warmup_lr, warmup_steps, last_iter=0):
        super().__init__(optimizer, base_lr, warmup_lr, warmup_steps, max_iter, last_iter)
        self.power = power

    def _after_warmup(self):
        return (1 - (self.last_iter - self.warmup_steps) / float(self.max_iter)) ** self.power * self.warmup_lr


_SCHEDULERS = {"Cosine": CosineLRScheduler, "Step": StepLRScheduler, "StepDecay": StepDecayLRScheduler, "Poly": PolynomialLRScheduler}


def scheduler_entry(cfg):
    """lr_scheduler/__init__.py:4-22 (incl. the *Epoch variants: epochs -> iterations by max_iter / max_epoch)."""
    typ, kw = cfg["type"], dict(cfg["kwargs"])
    if typ in ("StepEpoch", "CosineEpoch"):
        typ = typ.replace("Epoch", "")
        ratio = kw["max_iter"] / kw.pop("max_epoch")
        if "lr_epochs" in kw:
            kw["lr_steps"] = [round(e * ratio) for e in kw.pop("lr_epochs")]
        if "warmup_epoch" in kw:
            kw["warmup_steps"] = max(round(kw.pop("warmup_epoch") * ratio), 2)
    if typ not in _SCHEDULERS:
        raise NotImplementedError("lr_scheduler type %s (the reference ships %s)" % (typ, sorted(_SCHEDULERS)))
    return _SCHEDULERS[typ](**kw)


# --------------------------------------------------------------------------------------- optimizer groups
def param_groups(model, opt_cfg):
    """utils/misc.py:267-400 `param_group_all`, restated: module-type walk puts Conv2d/Linear/BatchNorm/LayerNorm
    biases into 'bias' when pconfig names it (else conv_b / linear_b / bn_b / ln_b), norm weights into bn_w / ln_w,
    Linear weights into 'linear_w' only when pconfig names it, parameters whose name contains 'logit_scale' into
    'logit_scale'; everything else (incl. MultiheadAttention.in_proj_*, embeddings, projections) is the default
    group.  Groups named in pconfig get its overrides; the others the optimizer defaults."""
    pconfig = {}
    if opt_cfg.get("no_wd", False):                                   # clip_solver.py:248-254
        for k in ("conv_b", "linear_b", "bn_w", "bn_b", "ln_w", "ln_b"):
            pconfig[k] = {"weight_decay": 0.0}
    pconfig.update(dict(opt_cfg.get("pconfig", {}) or {}))
    keys = ["bn_w", "bn_b", "conv_b", "linear_b", "ln_w", "ln_b"] + [k for k in ("linear_w", "logit_scale", "bias") if k in pconfig]
    pg, taken = {k: [] for k in keys}, set()

    def put(key, p):
        if p is not None and id(p) not in taken:
            taken.add(id(p))
            pg[key].append(p)

    any_bias = "bias" in pg
    for _, m in model.named_modules():
        if isinstance(m, torch.nn.Conv2d):
            put("bias" if any_bias else "conv_b", m.bias)
        elif isinstance(m, torch.nn.Linear):
            put("bias" if any_bias else "linear_b", m.bias)
            if "linear_w" in pg:
                put("linear_w", m.weight)
        elif isinstance(m, (torch.nn.BatchNorm1d, torch.nn.BatchNorm2d)):
            put("bn_w", m.weight)
            put("bias" if any_bias else "bn_b", m.bias)
        elif isinstance(m, torch.nn.LayerNorm):
            put("ln_w", m.weight)
            put("bias" if any_bias else "ln_b", m.bias)
    normal = []
    for name, p in model.named_parameters():
        if "logit_scale" in pg and "logit_scale" in name:
            put("logit_scale", p)
        if id(p) not in taken:
            normal.append(p)
    # EVERY typed group is emitted, the empty ones too (misc.py:386-393): a torch.optim.AdamW checkpoint records one entry per
    # group, and FlatAdamW.load_state_dict lines a reference / model-zoo checkpoint up group by group and position by position
    # (tests/test_oracle_golden.py pins names and order against the reference function itself)
    groups = [dict(params=normal)]
    for k in keys:
        groups.append(dict(params=pg[k], **dict(pconfig.get(k, {}))))
    return groups


def optim_entry(model, opt_cfg, base_lr=None):
    """optimizer/__init__.py:18-26: AdamW -> the engine's fused flat AdamW; other torch optimizers by name.  `base_lr`:
    clip_solver.py:243 overwrites optimizer.kwargs.lr with lr_scheduler.kwargs.base_lr before anything is built."""
    groups = param_groups(model, opt_cfg)
    kw = dict(opt_cfg.get("kwargs", {}))
    if base_lr is not None:
        kw["lr"] = base_lr
    typ = opt_cfg["type"]
    # the reference's own aliases when linklink.optim is absent (optimizer/__init__.py:8-15): FusedFP16SGD = SGD, FusedFP16AdamW = AdamW;
    # the FP16* wrappers of fp16_optim.py keep fp32 master weights around a half model -- the engine's masters are fp32 already
    typ = {"FusedFP16SGD": "SGD", "FP16SGD": "SGD", "FusedFP16AdamW": "AdamW", "FP16AdamW": "AdamW", "FP16RMSprop": "RMSprop"}.get(typ, typ)
    if typ == "AdamW" and not kw.get("amsgrad", False):
        kw.pop("amsgrad", None)
        kw["betas"] = tuple(kw.get("betas", (0.9, 0.999)))
        return FlatAdamW(groups, model.__dict__["_flat_store"], **kw)
    if typ in ("LARS", "AdamW_SGD", "FP16AdamW_SGD"):
        raise NotImplementedError("optimizer.type %s (optimizer/lars.py, AdamW_SGD.py) is not rebuilt: no shipped CLIP-family config uses it" % typ)
    if not hasattr(torch.optim, typ):
        raise ValueError("unknown optimizer.type %r" % (opt_cfg["type"],))
    # any torch optimizer steps the fp32 master weights (views into the flat buffer); the bf16 mirror is recast at the next step
    # because only the fused AdamW hands one over (engine.FlatParams.begin_step) -- AdamW with amsgrad: True lands here as well
    return getattr(torch.optim, typ)(groups, **kw)


# --------------------------------------------------------------------------------------- synthetic data
class SyntheticLoader(object):
    """Seeded GPU-resident batches with the reference's batch contract (clip_dataloader.py:47-53): `images`
    [b, 3*views, H, W] fp32, `captions` pre-tokenised ids (+ augmented ids / MLM labels for the DeCLIP family)."""

    def __init__(self, kind, batch_size, rank, device, res=224, ctx=77, n_distinct=4):
        self.kind, self.b, self.rank, self.device, self.res, self.ctx = kind, batch_size, rank, device, res, ctx
        self.cache, self.n = {}, n_distinct

    def get(self, step):
        key = step % self.n
        if key not in self.cache:
            seed = 1000 * self.rank + key
            views = {"clip": 1, "declip": 2, "defilip": 2, "filip": 2, "slip": 3}[self.kind]
            batch = {"images": synth.synth_images(self.b, views=views, res=self.res, seed=seed).to(self.device)}
            long_caps = dict(min_len=self.ctx - 6) if self.kind in ("filip", "defilip") else {}
            ids = synth.synth_tokens(self.b, ctx=self.ctx, seed=seed, **long_caps)
            if self.kind in ("declip", "defilip"):
                masked, labels = synth.synth_mlm(ids, seed=seed)
                aug = synth.synth_tokens(self.b, ctx=self.ctx, seed=seed + 500, **long_caps)
                batch["captions"] = torch.stack([masked, aug], dim=1).to(self.device)
                batch["mlm_labels"] = labels
            elif self.kind == "filip":
                masked, labels = synth.synth_mlm(ids, seed=seed)
                batch["captions"], batch["mlm_labels"] = masked.to(self.device), labels
            else:
                batch["captions"] = ids.to(self.device)
            self.cache[key] = batch
        return self.cache[key]


def model_kind(type_name):
    for k in ("defilip", "declip", "filip", "slip", "clip"):
        if type_name.startswith(k):
            return k
    raise NotImplementedError(type_name)


# --------------------------------------------------------------------------------------- solver
class ClsSolver(object):
    def __init__(self, config_file, train_loader=None, device="cuda"):
        self._device_kind = device
        self.config = parse_config(config_file) if isinstance(config_file, str) else AttrDict(config_file)
        self.config_file = config_file if isinstance(config_file, str) else None
        self.kind = model_kind(self.config.model.type)
        self.setup_env()
        self.build_model()
        self.build_optimizer()
        self.build_lr_scheduler()
        self.build_data(train_loader)
        self.build_criteria()

    # ---- clip_solver.py:104-165
    def setup_env(self):
        from . import hostinfo
        self.host_threads = hostinfo.limit_host_threads()      # a container's CPU quota, not the node's visible cores (hostinfo.py)
        self.rank, self.world_size = dh_dist.get_rank(), dh_dist.get_world_size()
        self.device = torch.device("cuda", torch.cuda.current_device()) if self._device_kind == "cuda" else torch.device(self._device_kind)
        saver = self.config.get("saver", AttrDict())
        base = os.path.dirname(self.config_file) if self.config_file else os.getcwd()
        self.save_dir = os.path.join(base, "checkpoints")
        self.print_freq = int(saver.get("print_freq", 10))
        self.save_freq = int(saver.get("save_freq", 0) or 0)
        self.logger = logging.getLogger("declip_amd.solver")
        if not self.logger.handlers:
            h = logging.StreamHandler()
            h.setFormatter(logging.Formatter("%(asctime)s %(message)s"))
            self.logger.addHandler(h)
        self.logger.setLevel(logging.INFO if self.rank == 0 else logging.WARNING)
        self.save_many = bool(saver.get("save_many", False))
        self.state = {"last_iter": 0}
        pre = saver.get("pretrain", None) or {}
        path = pre.get("path", None)
        if pre.get("auto_resume", False):                                        # clip_solver.py:128-133: the last checkpoint wins
            cands = sorted(glob.glob(os.path.join(self.save_dir, "ckpt*.pth.tar")), key=os.path.getmtime)
            if cands:
                path = cands[-1]
        if path:                                                                 # clip_solver.py:134-137
            self.state = torch.load(path, map_location="cpu", weights_only=False)
            self.state.setdefault("last_iter", 0)
            self.logger.info("Recovering from %s, keys=%s (iter %d)" % (path, list(self.state.keys()), self.state["last_iter"]))

    # ---- clip_solver.py:187-234
    def build_model(self):
        from prototype.model import model_entry
        self.model = model_entry(self.config.model)
        self.model.to(self.device)
        self.model.train()
        if "model" in self.state:
            sd = {k[len("module."):] if k.startswith("module.") else k: v for k, v in self.state["model"].items()}
            res = self.model.load_state_dict(sd, strict=False)                   # utils/misc.py:441-452: non-strict, every miss logged
            for k in res.missing_keys:
                self.logger.warning("missing key: %s" % k)
            for k in res.unexpected_keys:
                self.logger.warning("unexpected key: %s" % k)
            if sd and len(res.missing_keys) == len(self.model.state_dict()):
                raise RuntimeError("checkpoint 'model' shares no key with the model (wrong checkpoint?)")
        sync = bool(self.config.get("dist", {}).get("sync", False))
        self.model = dh_dist.DistModule(self.model, sync)

    def build_optimizer(self):
        base_lr = (self.config.get("lr_scheduler", None) or {}).get("kwargs", {}).get("base_lr", None)
        self.optimizer = optim_entry(self.model.module, self.config.optimizer, base_lr=base_lr)
        if "optimizer" in self.state:            # torch.optim.AdamW layout: the reference's checkpoints and our own (clip_solver.py:655)
            self.optimizer.load_state_dict(self.state["optimizer"])              # raises if it does not line up: never silently dropped
        elif "optimizer_flat" in self.state and isinstance(self.optimizer, FlatAdamW):     # round-1 checkpoints of this engine
            st = self.state["optimizer_flat"]
            self.optimizer.m.copy_(st["m"]), self.optimizer.v.copy_(st["v"])
            self.optimizer.step_count = st["step"]
        elif self.state.get("last_iter", 0) > 0:
            self.logger.warning("resuming at iter %d WITHOUT optimizer state (none in the checkpoint)" % self.state["last_iter"])

    def build_lr_scheduler(self):
        cfg = AttrDict(self.config.lr_scheduler)
        kw = dict(cfg.kwargs)
        kw["optimizer"] = self.optimizer
        kw["last_iter"] = self.state["last_iter"]
        self.max_iter = int(kw["max_iter"])
        self.lr_scheduler = scheduler_entry(dict(type=cfg.type, kwargs=kw))

    def build_data(self, train_loader):
        d = self.config.get("data", AttrDict())
        self.batch_size = int(d.get("batch_size", 128))
        if train_loader is not None:
            self.loader = train_loader
            return
        if d.get("read_from", "fake") not in ("fake", "synthetic") and d.get("type", "clip") != "synthetic":
            raise NotImplementedError(
                "only data.read_from: fake / synthetic is built in (the I/O pipeline is out of scope, DESIGN_HISTORY.md s7); "
                "pass train_loader= an iterable of {'images','captions'} batch dicts for real data")
        m = self.model.module
        ctx = int((m.text_encoder if hasattr(m, "text_encoder") else m.encode_text).context_length)
        self.loader = SyntheticLoader(self.kind, self.batch_size, self.rank, self.device, res=int(d.get("input_size", 224)), ctx=ctx)

    def build_criteria(self):
        self.criterion = ClipInfoCELoss()
        self.simsiam_criterion = SimsiamLoss()
        self.nt_xent_criterion = NTXentLoss(self.batch_size) if self.kind in ("declip", "defilip", "filip") else NT_Xent(self.batch_size)
        self.simclr_criterion = NT_Xent_gather(self.batch_size)
        self.meters = {k: AverageMeter(self.print_freq) for k in ("loss", "top1", "top5", "step_time")}

    # ---- per-model loss composition (steps.py restates the solvers)
    def _loss(self, batch):
        W = self.world_size
        if self.kind == "clip":
            return steps.clip_loss(self.model, batch, self.criterion, W)
        if self.kind in ("declip", "defilip"):
            w = dict(self.config.get("clip_simsiam_loss_weight", steps.DEFILIP_WEIGHTS if self.kind == "defilip" else steps.DECLIP_WEIGHTS))
            tv = self.config.get("data", {}).get("train", {})
            return steps.declip_loss(self.model, batch, self.criterion, self.simsiam_criterion, self.nt_xent_criterion, weights=w,
                                     world_size=W, image_text_two_view=tv.get("image_text_two_view", False),   # declip_solver.py:447-452
                                     only_image_two_view=tv.get("only_image_two_view", False))
        if self.kind == "slip":
            return steps.slip_loss(self.model, batch, self.criterion, self.simclr_criterion, self.nt_xent_criterion,
                                   weights=dict(self.config.get("loss_weight", steps.SLIP_WEIGHTS)), world_size=W)
        return steps.filip_loss(self.model, batch, self.criterion,
                                weights=dict(self.config.get("clip_simsiam_loss_weight", steps.FILIP_WEIGHTS)), world_size=W)

    # ---- grad_clip (clip_solver.py:489-530): parameter clips around the step, gradient clips before optimizer.step().  Everything
    # stays on the device (no .item()): the reference's host reads are replaced by tensor arithmetic with the same result.
    def _gc(self):
        gc = self.config.get("grad_clip", None)
        return (gc.get("type"), gc) if gc else (None, None)

    def _scales(self):
        m = self.model.module
        out = [m.logit_scale]
        if hasattr(m, "logit_scale_dense"):                 # filip_solver.py:646,661 clamps the dense temperature too
            out.append(m.logit_scale_dense)
        return out

    def _param_clip_before(self):
        typ, gc = self._gc()
        if typ == "constant":
            self.model.module.logit_scale.requires_grad = False
        elif typ == "logit_scale_param":
            self._scale_before = self.model.module.logit_scale.data.clone()
        elif typ == "logit_scale_param_abs_min":
            self.model.module.logit_scale.data.clamp_(min=gc.value)
        elif typ == "logit_scale_param_value":
            for p in self._scales():
                p.data.clamp_(min=gc.value, max=gc.max_value)

    def _param_clip_after(self):
        typ, gc = self._gc()
        if typ == "logit_scale_param":                      # the step may move the temperature by at most `value`
            p, before = self.model.module.logit_scale, self._scale_before
            p.data.copy_(torch.minimum(torch.maximum(p.data, before - gc.value), before + gc.value))
        elif typ == "logit_scale_param_abs_min":
            self.model.module.logit_scale.data.clamp_(min=gc.value)
        elif typ == "logit_scale_param_value":
            for p in self._scales():
                p.data.clamp_(min=gc.value, max=gc.max_value)

    def _grad_clip_before(self):
        typ, gc = self._gc()
        if typ not in ("norm", "value", "logit_scale_grad"):
            return
        flat = self.model.module.__dict__["_flat_store"]
        flat.join_streams()
        if typ == "norm":                                   # utils/grad_clip.py:10-44 on the flat gradient buffer (one norm, one scale)
            total = flat.flat_g.norm(2)                     # alignment padding and grad-None slots are zeros
            flat.flat_g.mul_(torch.clamp(float(gc.value) / (total + 1e-6), max=1.0))
        elif typ == "value":
            flat.flat_g.clamp_(min=-float(gc.value), max=float(gc.value))
        else:
            g = self.model.module.logit_scale.grad
            if g is not None:
                g.clamp_(min=-float(gc.value), max=float(gc.value))

    def _clamp_params(self):                                # kept for callers of the round-1 name
        self._param_clip_after()

    # ---- the captured step (declip_amd/graph.py): forward + loss + backward as ONE hipGraph launch per iteration, for a batch that
    # changes every iteration as the reference's does (clip_solver.py:398-402).  Opt-in: model.kwargs.engine.step_graph: true, or
    # DH_STEP_GRAPH=1.  CLIP only (one image view, one caption per pair); the batch is copied into static buffers, graphs are kept
    # per padded packed row count of its captions (engine.packed_key).
    def _graph_wanted(self):
        if self.__dict__.get("_graph_off"):
            return False
        eng_cfg = dict(self.config.model.get("kwargs", {}).get("engine", {}) or {})
        want = bool(eng_cfg.get("step_graph", False)) or os.environ.get("DH_STEP_GRAPH", "0") == "1"
        typ, _ = self._gc()
        # parameter clips that act BETWEEN forward and backward in the reference (constant / logit_scale_param / abs_min) keep the eager step
        ok = (want and self.kind == "clip" and self.device.type == "cuda" and typ in (None, "logit_scale_param_value", "norm", "value", "logit_scale_grad"))
        # One graph per padded row count needs a step whose launches depend on the captions through that count ALONE.  That holds for
        # padded captions (mode 0: no key at all) and for packed captions on bf16 towers with head dimension 64 (mode 1: the kernels
        # read the valid row count on the device).  Mode 2 bakes the host-side row count into the capture (pack_idx[:rows]) and the
        # fp32 / other-head-size kernels take the row count as a launch argument: there a graph would be replayed for a batch it
        # was not captured for (wrong rows gathered) or re-captured almost every step -- the eager step takes those, as bench.py does.
        if ok:
            from . import engine
            m = self.model.module
            mode = engine.text_packed_mode()
            bf16 = m.__dict__["_flat_store"].act_dtype == torch.bfloat16
            heads_dim = int(m.encode_text.width) // int(m.encode_text.heads)
            ok = mode == 0 or (mode == 1 and bf16 and heads_dim == 64)
        # Data parallelism: the graphs' key must be the same on every rank (a rank that captures while its peers replay would leave them
        # waiting inside replayed collectives).  Round 6: it is -- the padded packed row count is the MAX over the ranks (dist.RowsSync, one
        # host-side integer all-reduce per batch on the prefetcher's worker thread), every rank pads up to it -- and the step's collectives
        # run on the library communicator, the only ones GraphedStep captures (graph._check_capturable).  Without it: the eager step.
        if ok and self.world_size > 1:
            from . import dist as dh_dist
            if dh_dist.native_comm() is None:
                ok = False
                why = "world_size > 1 without the library communicator (DH_COMM_NATIVE=0 or a non-nccl process group)"
        if not ok:
            self._graph_off = True
            if want and self.rank == 0:          # (say once why a requested captured step was declined)
                self.logger.info("engine.step_graph / DH_STEP_GRAPH=1 declined, the step runs eagerly: %s" % (
                    locals().get("why") or "needs kind == clip on a GPU, packed mode 0 or 1 with bf16 towers and head dimension 64, "
                                           "and a grad_clip type that acts after backward"))
        return ok

    def _rows_sync(self):
        """dist.RowsSync of this solver (created on first use, on every rank at the same point of the step loop)."""
        rs = self.__dict__.get("_rows_sync_obj")
        if rs is None:
            from . import dist as dh_dist
            rs = self._rows_sync_obj = dh_dist.RowsSync(self.model.module.__dict__["_flat_store"].act_dtype)
        return rs

    def _graphed_loss(self, batch):
        from . import engine
        from .graph import GraphedStep
        images, caps = batch["images"], batch["captions"]
        if not (torch.is_tensor(images) and torch.is_tensor(caps) and images.is_cuda and caps.is_cuda and images.dtype == torch.float32 and caps.dim() == 2):
            self._graph_off = True               # strings / host tensors / uint8 intake outside the prefetcher: the eager step takes them
            return None
        g = self.__dict__.get("_graph")
        if g is None or g["images"].shape != images.shape or g["captions"].shape != caps.shape:
            st = {"images": torch.empty_like(images), "captions": torch.empty_like(caps)}
            m = self.model.module
            dtype = m.__dict__["_flat_store"].act_dtype
            packed = engine.text_packed_mode() == 1
            heads_dim = int(m.encode_text.width) // int(m.encode_text.heads)

            def fn():
                out = steps.clip_loss(self.model, st, self.criterion, self.world_size)
                self._param_clip_before()        # between forward and backward, where the reference has it (clip_solver.py:489-505): device-side clamps, captured with the step
                out["loss"].backward()
                return out["loss"].detach(), out["top1"].detach(), out["top5"].detach()
            key = (lambda: engine.packed_key(st["captions"], dtype, heads_dim)) if packed else None
            dist_on = self.world_size > 1

            def agree(ok):
                flag = torch.tensor([1.0 if ok else 0.0], device=self.device)
                torch.distributed.all_reduce(flag, op=torch.distributed.ReduceOp.MIN)
                return bool(float(flag) > 0.5)
            g = dict(st, step=GraphedStep(fn, warmup=2, modules=(self.model,), key=key, fallback=dist_on, agree=agree if dist_on else None))
            self._graph = g
        g["images"].copy_(images)
        g["captions"].copy_(caps)
        tag = getattr(caps, "_dh_rows", None)
        if tag is None or tag[0] != caps._version:
            rows = int((caps.argmax(dim=-1) + 1).sum())          # one read-back for a tensor that came without its host-side count
            caps._dh_rows = (caps._version, rows)
            tag = caps._dh_rows
        pad = getattr(caps, "_dh_rows_pad", None)
        pad = pad[1] if (pad is not None and pad[0] == caps._version) else None
        if pad is None and self.world_size > 1 and engine.text_packed_mode() == 1:
            pad = self._rows_sync()(tag[1])      # a batch that did not come through the prefetcher: the job-wide padded row count, here
        engine.set_rows_tag(g["captions"], tag[1], rows_pad=pad)
        loss, p1, p5 = g["step"]()
        return dict(loss=loss, top1=p1, top5=p5)

    def train_step(self, curr_step):
        if hasattr(self.loader, "get"):
            batch = self.loader.get(curr_step)
        else:
            try:
                batch = next(self._iter)
            except StopIteration:
                raise RuntimeError("train_loader is exhausted at iteration %d of %d: the solver is iteration-based, hand it an "
                                   "infinite / iteration-sized sampler (data/sampler.py DistributedGivenIterationSampler in the "
                                   "reference)" % (curr_step, self.max_iter)) from None
        self.lr_scheduler.step(curr_step)
        if self._graph_wanted():
            self.optimizer.zero_grad()
            out = self._graphed_loss(batch)
            if out is not None:
                self.model.sync_gradients()
                self._grad_clip_before()
                self.optimizer.step()
                self._param_clip_after()
                return out
        out = self._loss(batch)
        self.optimizer.zero_grad()
        self._param_clip_before()
        out["loss"].backward()
        self.model.sync_gradients()
        self._grad_clip_before()
        self.optimizer.step()
        self._param_clip_after()
        return out

    def train(self, max_steps=None):
        if not hasattr(self.loader, "get"):
            d = self.config.get("data", AttrDict())
            if d.get("train", AttrDict()).get("prefetch", d.get("prefetch", True)):
                # clip_solver.py:335-337 (`data.train.prefetch`): host batches are tokenised / pinned on a background thread
                # and copied on a side stream one step ahead (declip_amd/prefetch.py)
                from .prefetch import DataPrefetcher
                m = self.model.module
                tower = m.text_encoder if hasattr(m, "text_encoder") else m.encode_text
                tok = None
                if self.kind in ("clip", "slip") and getattr(tower, "_bpe_path", None) and os.path.exists(tower._bpe_path):
                    tok = tower._get_tokenizer()          # the other families augment / mask the caption TEXT in forward()
                # user loaders may hand decoded uint8 canvases + crop boxes (declip_amd.augment): cropped / resized on the GPU
                # DeCLIP family: caption sampling + EDA + BPE + MLM masking run on the prefetch thread as well (declip.py:203-230 has
                # them inside forward())
                prep = m.prepare_captions if (hasattr(m, "prepare_captions") and getattr(tower, "_bpe_path", None)
                                              and os.path.exists(tower._bpe_path)) else None
                from . import engine
                rows_sync = self._rows_sync() if (self.world_size > 1 and self._graph_wanted() and engine.text_packed_mode() == 1) else None
                self._iter = DataPrefetcher(self.loader, self.device, tokenizer=tok, context_length=int(tower.context_length), text_prep=prep,
                                            image_size=int(d.get("input_size", 224)), rows_sync=rows_sync)
            else:
                self._iter = iter(self.loader)
        start = self.state["last_iter"] + 1
        end = self.max_iter if max_steps is None else min(self.max_iter, start + max_steps - 1)
        t_last = time.time()
        for curr_step in range(start, end + 1):
            out = self.train_step(curr_step)
            if curr_step == start:           # everything long-lived exists now: keep it out of the cyclic collector's scans,
                import gc                    # and collect at a fixed cadence instead of whenever the allocation counters trip
                gc.collect()                 # (a generation-2 pass in mid-step is a 50-200 ms host pause)
                gc.freeze()
                gc.disable()
            elif (curr_step - start) % 100 == 0:
                gc.collect(1)
            logged = [(self.meters["loss"], out["loss"], 1)]
            if "top1" in out:
                logged += [(self.meters["top1"], out["top1"].detach() / self.world_size, 1), (self.meters["top5"], out["top5"].detach() / self.world_size, 1)]
            reduce_update_packed(logged)         # ONE collective for all logged scalars of the step (misc.py:38-40 issues one per meter)
            if curr_step % self.print_freq == 0 or curr_step == end:
                if self.device.type == "cuda":
                    torch.cuda.synchronize()
                dt = (time.time() - t_last) / max(1, min(self.print_freq, curr_step - start + 1))
                t_last = time.time()
                parts = " ".join("%s %.4f" % (k, float(v.detach())) for k, v in out.get("parts", {}).items())
                self.logger.info("Iter [%d/%d] loss %.4f (%.4f) top1 %.2f top5 %.2f lr %.6f %.1f ms/step %.0f pairs/s %s" % (
                    curr_step, self.max_iter, self.meters["loss"].val, self.meters["loss"].avg, self.meters["top1"].avg,
                    self.meters["top5"].avg, self.lr_scheduler.get_lr()[0], dt * 1e3, self.batch_size * self.world_size / dt, parts))
            if self.save_freq and curr_step % self.save_freq == 0:
                self.save(curr_step)
        self.state["last_iter"] = end
        import gc
        gc.enable()
        return out

    def save(self, curr_step):
        """clip_solver.py:649-668: {'model' ('module.'-prefixed), 'optimizer' (torch.optim.AdamW layout), 'last_iter'} to
        ckpt.pth.tar (ckpt_<iter>.pth.tar with saver.save_many).  Written to a temporary name and renamed into place: a crash in
        mid-save never leaves a truncated file where auto_resume would pick it up."""
        if self.rank != 0:
            return None
        os.makedirs(self.save_dir, exist_ok=True)
        st = {"model": {"module." + k: v.detach().cpu() for k, v in self.model.module.state_dict().items()},
              "optimizer": self.optimizer.state_dict(), "last_iter": curr_step}
        name = "ckpt_%d.pth.tar" % curr_step if self.save_many else "ckpt.pth.tar"
        path = os.path.join(self.save_dir, name)
        tmp = path + ".tmp%d" % os.getpid()
        torch.save(st, tmp)
        os.replace(tmp, path)
        return path

    @torch.no_grad()
    def evaluate(self, val_data=None):
        """Zero-shot classification (clip_solver.py:675-737).  val_data: {'loader': iterable of {'images', 'labels'} batches whose
        `.dataset` (or the loader itself) has get_label_texts() -> (prompts class-major, ensemble matrix)}; None builds the
        synthetic set from `data.test` (label_num / prompts_num / batch_size / batches).  Every rank evaluates its own batches;
        the hit counters are summed over ranks, so every rank returns the same metrics (the reference broadcasts rank 0's)."""
        from . import zeroshot
        m = self.model.module
        if val_data is None:
            t = self.config.get("data", AttrDict()).get("test", AttrDict())
            ctx = int((m.text_encoder if hasattr(m, "text_encoder") else m.encode_text).context_length)
            loader = zeroshot.SyntheticZeroShotData(
                label_num=int(t.get("label_num", 1000)), prompts_num=int(t.get("prompts_num", 1)),
                batch_size=int(t.get("batch_size", self.batch_size)), batches=int(t.get("batches", 4)),
                res=int(self.config.get("data", {}).get("input_size", 224)), ctx=ctx, rank=self.rank, world=self.world_size)
        else:
            loader = val_data["loader"] if isinstance(val_data, dict) else val_data
        source = getattr(loader, "dataset", loader)
        texts, ensemble = source.get_label_texts()
        label_num = ensemble.shape[1]
        n_texts = texts.shape[0] if torch.is_tensor(texts) else len(texts)
        self.logger.info("Use %d prompts" % (n_texts // label_num))
        was_training = self.model.training
        self.model.eval()
        class_emb = zeroshot.class_embeddings(m, texts, label_num, text_chunk=int(self.config.get("eval_text_chunk", 2048)))
        if zeroshot._is_identity(ensemble):      # the reference's datasets: scores = softmax(logits) @ I (clip_dataset.py:281)
            ensemble = None
        meter = zeroshot.ZeroShotMeter(self.device)
        t0, n_img = time.time(), 0
        for batch in loader:
            images = batch["images"].to(self.device, non_blocking=True)
            if images.dtype == torch.uint8 and batch.get("image_boxes") is not None:
                # decoded images on a uint8 canvas + boxes (augment.resize_center_crop_params = the reference's ONECROP pipeline,
                # imagenet_dataloader.py:105-111): Resize(256) + CenterCrop(224) + ToTensor + Normalize on the GPU
                from .prefetch import crops_on_device
                res = int(self.config.get("data", AttrDict()).get("input_size", 224))
                images = crops_on_device({"images": images, "image_boxes": batch["image_boxes"].to(self.device, non_blocking=True)},
                                         (res, res))["images"]
            out = zeroshot.classify(m, images, class_emb, ensemble, return_dense=bool(self.config.get("return_dense", False)))
            labels = batch["labels"] if "labels" in batch else batch["label"]
            meter.update(out["topk"], labels.view(-1).long())
            n_img += images.shape[0]
        metrics = meter.result()
        if self.device.type == "cuda":
            torch.cuda.synchronize()
        metrics["images_per_s"] = n_img * self.world_size / max(time.time() - t0, 1e-9)
        self.logger.info("zero-shot: " + " ".join("%s %.4g" % kv for kv in metrics.items()))
        if was_training:
            self.model.train()
        return metrics


def main():
    """clip_solver.py:740-764."""
    ap = argparse.ArgumentParser(description="contrastive pre-training solver (MI355X engine)")
    ap.add_argument("--config", required=True, type=str)
    ap.add_argument("--evaluate", action="store_true")
    ap.add_argument("--max-steps", type=int, default=None)
    args = ap.parse_args()
    if int(os.environ.get("WORLD_SIZE", os.environ.get("SLURM_NTASKS", "1"))) > 1:
        dh_dist.initialize("nccl")
    else:
        torch.cuda.set_device(0)
    solver = ClsSolver(args.config)
    if args.evaluate:
        solver.evaluate()
    else:
        solver.train(args.max_steps)


if __name__ == "__main__":
    main()
