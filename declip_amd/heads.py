"""DeCLIP / SLIP heads on the HIP engine: Linear + BatchNorm1d(+ReLU) MLPs (SimSiam projector / predictor,
SimCLR head), negative-cosine loss, nearest-neighbour feature bank, masked-LM head.

Reference: model/declip.py:33-130 (projection_MLP / prediction_MLP), loss_functions/loss.py:49-81 (D, SimsiamLoss),
model/utils/nnclr_modules/{memory_bank,nn_memory_bank}.py, model/declip.py:326-334 (MLM CE).
Parameter containers keep the reference's attribute names (linear1/bn1/.../layer2) so state_dicts match."""
import torch
from torch import nn

from . import engine, ops
from .lib import DeclipHipError


def _flat_of(module):
    root = module.__dict__.get("_engine_root")
    if root is None:
        raise DeclipHipError("%s is not attached to an engine model" % type(module).__name__)
    return root._flat_store.ensure()


class CastFn(torch.autograd.Function):
    """dtype cast with a cast backward (fp32 tower features -> engine activation dtype and back)."""

    @staticmethod
    def forward(ctx, x, dtype):
        ctx.src = x.dtype
        if x.dtype == dtype:
            return x
        return ops.cast(x.contiguous(), torch.empty(x.shape, device=x.device, dtype=dtype))

    @staticmethod
    def backward(ctx, g):
        if g.dtype == ctx.src:
            return g, None
        return ops.cast(g.contiguous(), torch.empty(g.shape, device=g.device, dtype=ctx.src)), None


class LinearFn(torch.autograd.Function):
    """y = x W^T + b through dh_gemm; dW / db accumulate straight into the flat gradient buffer."""

    @staticmethod
    def forward(ctx, x, lin, flat):
        x = x.contiguous()
        y = ops.gemm(x, flat.wview(lin.weight), bias=lin.bias.data if lin.bias is not None else None)
        ctx.lin, ctx.flat = lin, flat
        ctx.save_for_backward(x)
        return y

    @staticmethod
    def backward(ctx, dy):
        (x,) = ctx.saved_tensors
        lin, flat = ctx.lin, ctx.flat
        flat.begin_backward()
        dy = dy.contiguous()
        if lin.weight.requires_grad:
            engine.weight_grad(dy, x, flat.gview(lin.weight), flat.gview(lin.bias) if lin.bias is not None else None)
        dx = ops.gemm(dy, flat.wview(lin.weight), b_kmajor=True)
        return dx, None, None


class Bn1dFn(torch.autograd.Function):
    """nn.BatchNorm1d (+ReLU) with batch statistics per group of rows (one group per view)."""

    @staticmethod
    def forward(ctx, x, bn, flat, groups, relu):
        x = x.contiguous()
        training = bn.training or bn.running_mean is None
        y, mean, invstd = ops.bn1d_fwd(x, bn.weight.data, bn.bias.data, bn.running_mean, bn.running_var, groups, relu, training,
                                       eps=bn.eps, momentum=bn.momentum if bn.momentum is not None else 0.1)
        if training and bn.num_batches_tracked is not None:
            bn.num_batches_tracked += groups
        ctx.bn, ctx.flat, ctx.groups, ctx.relu, ctx.training = bn, flat, groups, relu, training
        ctx.save_for_backward(x, y, mean, invstd)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, y, mean, invstd = ctx.saved_tensors
        bn, flat = ctx.bn, ctx.flat
        if not ctx.training:
            raise DeclipHipError("BatchNorm1d backward in eval mode is not supported")
        flat.begin_backward()
        dx = ops.bn1d_bwd(dy.contiguous(), x, y, bn.weight.data, mean, invstd, flat.gview(bn.weight), flat.gview(bn.bias),
                          ctx.groups, ctx.relu)
        return dx, None, None, None, None


class _HeadBase(nn.Module):
    def _lin(self, x, lin):
        return LinearFn.apply(x, lin, _flat_of(self))

    def _bn(self, x, bn, groups, relu):
        return Bn1dFn.apply(x, bn, _flat_of(self), groups, relu)


class projection_MLP(_HeadBase):
    """model/declip.py:33-90 (DeCLIP; note bn3 = BatchNorm1d(hidden_dim) applied to the out_dim output, quirk 8)
    and model/slip.py:50-109 (SLIP: out_bn flag)."""

    def __init__(self, in_dim, hidden_dim=1024, out_dim=1024, num_layers=3, out_bn=True):
        super().__init__()
        self.num_layers, self.in_dim, self.hidden_dim, self.out_dim, self.out_bn = num_layers, in_dim, hidden_dim, out_dim, out_bn
        self.linear1 = nn.Linear(in_dim, hidden_dim)
        self.bn1 = nn.BatchNorm1d(hidden_dim)
        self.relu1 = nn.ReLU(inplace=True)
        self.linear2 = nn.Linear(hidden_dim, hidden_dim)
        self.bn2 = nn.BatchNorm1d(hidden_dim)
        if num_layers == 3:
            self.relu2 = nn.ReLU(inplace=True)
            self.linear3 = nn.Linear(hidden_dim, out_dim)
            self.bn3 = nn.BatchNorm1d(hidden_dim)          # always constructed (state_dict), applied only if out_bn

    def forward(self, x, groups=1):
        """x: [groups*b, in_dim]; BN statistics per group (== calling the reference module once per view)."""
        flat = _flat_of(self)
        x = CastFn.apply(x, flat.act_dtype)
        x = self._bn(self._lin(x, self.linear1), self.bn1, groups, True)
        x = self._lin(x, self.linear2)
        if self.num_layers == 3:
            x = self._bn(x, self.bn2, groups, True)
            x = self._lin(x, self.linear3)
            if self.out_bn:
                x = self._bn(x, self.bn3, groups, False)
        else:
            x = self._bn(x, self.bn2, groups, False)
        return x


class prediction_MLP(_HeadBase):
    """model/declip.py:92-130."""

    def __init__(self, in_dim, hidden_dim=512, out_dim=1024):
        super().__init__()
        self.in_dim, self.hidden_dim, self.out_dim = in_dim, hidden_dim, out_dim
        self.linear1 = nn.Linear(in_dim, hidden_dim)
        self.bn1 = nn.BatchNorm1d(hidden_dim)
        self.relu1 = nn.ReLU(inplace=True)
        self.layer2 = nn.Linear(hidden_dim, out_dim)

    def forward(self, x, groups=1):
        flat = _flat_of(self)
        x = CastFn.apply(x, flat.act_dtype)
        x = self._bn(self._lin(x, self.linear1), self.bn1, groups, True)
        return self._lin(x, self.layer2)


class NegCosFn(torch.autograd.Function):
    """cos(p_r, stopgrad z_r) per row (loss_functions/loss.py:49-55)."""

    @staticmethod
    def forward(ctx, p, z):
        p, z = p.contiguous(), z.detach().contiguous()
        ctx.save_for_backward(p, z)
        return ops.cos_rows_fwd(p, z)

    @staticmethod
    def backward(ctx, g):
        p, z = ctx.saved_tensors
        return ops.cos_rows_bwd(p, z, g.contiguous().float()), None


class SimsiamLoss(nn.Module):
    """loss_functions/loss.py:65-81: forward(p1, z1, p2, z2) = -0.5 (D(p1, z2) + D(p2, z1))."""

    def __init__(self, symmetry=True):
        super().__init__()
        self.symmetry = symmetry

    def forward(self, p1, z1, p2, z2, minimize_loss=False):
        if not self.symmetry or minimize_loss:
            raise NotImplementedError("only the symmetric D(p, stopgrad z) form is on the hot path")
        return -0.5 * (NegCosFn.apply(p1, z2).mean() + NegCosFn.apply(p2, z1).mean())


class NNMemoryBankModule(nn.Module):
    """Nearest-neighbour feature bank (nnclr_modules/nn_memory_bank.py:10-65, memory_bank.py:40-124).

    MI355X re-design: the bank lives in HBM as [size, D] fp32 (row-major, unit rows) and is searched in
    place by dh_nn_bank_query; the reference keeps it on the CPU as [D, size] and re-uploads 128 MiB three
    times per step.  Semantics kept: random unit-vector init on first use, FIFO enqueue that drops the
    batch tail on wrap and resets the pointer (memory_bank.py:82-87), per-rank bank (not checkpointed)."""

    def __init__(self, size=2 ** 16, topk=1):
        super().__init__()
        if size < 0:
            raise ValueError("Illegal memory bank size %d, must be non-negative." % size)
        if topk != 1:
            raise NotImplementedError("topk > 1 (all shipped configs use nn_topk=1)")
        self.size, self.topk = size, topk
        self.bank = None             # [size, D] fp32: the first `size` rows of _store (a caller may also assign a tensor directly)
        self._store = None           # [size + spill, D]: rows past `size` take the batch tail that a wrapping enqueue drops
        self._ptr = None             # [1] int64 on the bank's device: the write pointer
        self._ptr_init = 0
        self._ar = None
        self.spill_rows = 1024       # rows behind the bank for the tail a wrapping enqueue drops: >= the largest batch enqueued
        self._captured = False       # an enqueue was captured into a hipGraph: the storage must not move any more

    # The write pointer lives on the DEVICE: enqueue is dh_nn_bank_enqueue (copy + pointer update) on the stream, with no host value baked into a
    # launch -- a step captured as a hipGraph (declip_amd/graph.py) advances the queue on every replay.  `bank_ptr` (the reference's
    # attribute, memory_bank.py:66) reads it back; only tests and checkpoints do.
    @property
    def bank_ptr(self):
        return int(self._ptr.item()) if self._ptr is not None else self._ptr_init

    @bank_ptr.setter
    def bank_ptr(self, v):
        self._ptr_init = int(v)
        if self._ptr is not None:
            self._ptr.fill_(int(v))

    @torch.no_grad()
    def init_bank(self, dim, device, generator=None):
        bank = torch.randn(self.size, dim, generator=generator)
        self.bank = torch.nn.functional.normalize(bank, dim=1).to(device).contiguous()
        self.bank_ptr = 0

    @torch.no_grad()
    def _enqueue(self, batch):
        """FIFO enqueue that drops the batch tail on wrap and resets the pointer (memory_bank.py:82-87): rows ptr .. ptr+b-1 are
        written, those >= size land in the spill rows behind the bank (never searched); ptr = 0 if ptr + b >= size else ptr + b."""
        b, dev = batch.shape[0], batch.device
        if self._store is None or self.bank.data_ptr() != self._store.data_ptr() or self._store.shape[0] < self.size + b:
            # (re)allocation of the queue's storage: the first enqueue, a bank tensor assigned from outside, or a batch larger than
            # the spill region (sized once for `spill_rows`: DECLIP(global_nn_bank=True) sets it to world x batch up front).  Under
            # stream capture -- or after one: a captured query / enqueue keeps using the OLD storage -- that is refused.
            if torch.cuda.is_available() and dev.type == "cuda" and (torch.cuda.is_current_stream_capturing() or self._captured):
                raise RuntimeError("NNMemoryBankModule: the queue's storage would be re-allocated (batch of %d rows, spill region %d) "
                                   "%s a hipGraph capture of the step; size it up front with spill_rows" %
                                   (b, 0 if self._store is None else self._store.shape[0] - self.size,
                                    "during" if torch.cuda.is_current_stream_capturing() else "after"))
            store = torch.empty(self.size + max(b, self.spill_rows), self.bank.shape[1], device=dev, dtype=self.bank.dtype)
            store[:self.size].copy_(self.bank)
            self._store, self.bank = store, store[:self.size]
        elif torch.cuda.is_available() and dev.type == "cuda" and torch.cuda.is_current_stream_capturing():
            self._captured = True
        if self._ptr is None:
            self._ptr = torch.full((1,), self._ptr_init, device=dev, dtype=torch.int64)
        elif self._ptr.device != dev:
            self._ptr = self._ptr.to(dev)            # the LIVE write position moves with the queue (not the initial one)
        if self._store.dtype == torch.float32 and self._store.shape[1] % 4 == 0 and self._store.data_ptr() % 16 == 0:
            ops.nn_bank_enqueue(self._store, self._ptr, batch.float().contiguous(), self.size)       # dh_nn_bank_enqueue: copy + pointer update on the stream
        else:
            # a bank assigned from outside in another dtype, or a feature width the 16-byte kernel does not take: the
            # same FIFO through torch (index_copy_ + pointer arithmetic on the device; still no host value in a launch)
            rows = self._ptr + torch.arange(b, device=dev)
            self._store.index_copy_(0, rows, batch.to(self._store.dtype))
            self._ptr.copy_(torch.where(self._ptr + b >= self.size, torch.zeros_like(self._ptr), self._ptr + b))

    @torch.no_grad()
    def forward(self, output, update=False, query=True, enqueue=None):
        """returns [nearest neighbours [b, D] fp32]; the search sees the bank BEFORE this call's enqueue
        (memory_bank.py:117-122).  query=False skips the (discarded) search of an update-only call.  `enqueue`: rows to put into
        the queue instead of `output` (the GLOBAL queue of DECLIP(global_nn_bank=True): every rank's features, gathered)."""
        output = output.detach().float().contiguous()
        if self.bank is None:
            self.init_bank(output.shape[1], output.device)
        res = None
        if query:
            _, feats = ops.nn_bank_query(output, self.bank)
            res = [feats]
        if update:
            self._enqueue(output if enqueue is None else enqueue.detach().float().contiguous())
        return res


class MlmHeadFn(torch.autograd.Function):
    """text_label_predictor + cross-entropy on the masked positions only (model/declip.py:326-334).

    The reference runs Linear(width -> 49409) on ALL b*77 positions and then selects; here the masked rows are gathered first
    (same result, ~7x less work).
    bf16: the vocabulary-wide logits [n_masked, 49409] never reach HBM.  Forward = ONE launch of the persistent GEMM with a
    cross-entropy epilogue (per-tile max / sum-exp partials + the label logit, merged by a finalize kernel:
    ops.ce_fused_fwd); backward = the same GEMM recomputed with an epilogue that stores dl = g (softmax - onehot) as bf16
    (ops.ce_fused_bwd), which feeds the two gradient GEMMs.  Round 1 wrote fp32 logits (1.2 GB per DeCLIP step at b = 512), read
    them twice in the forward cross-entropy and once more in the backward.
    fp32 validation mode: logits materialised in a layout padded to a multiple of 64 columns, row-wise CE kernels."""

    @staticmethod
    def forward(ctx, words, idx, labels_sel, lin, flat):
        width = words.shape[-1]
        V = lin.weight.shape[0]
        Vp = (V + 63) // 64 * 64
        n = idx.numel()
        wflat = words.reshape(-1, width)
        w = flat.wview(lin.weight)
        fused = ops.ce_fused_ok(wflat, w)
        if fused:
            n_pad = (n + 255) // 256 * 256                       # whole tiles of the persistent GEMM; the padding rows are zero
            rows = ops.gather_rows(wflat, idx, n_pad)
            row_loss, row_lse = ops.ce_fused_fwd(rows, w, lin.bias.data, labels_sel, n)
            logits = None
        else:
            n_pad = max(64, (n + 63) // 64 * 64)
            rows = ops.gather_rows(wflat, idx, n_pad)
            logits = torch.empty(n_pad, Vp, device=words.device, dtype=torch.float32)
            ops.gemm(rows, w, bias=lin.bias.data, out=logits, pad_ok=True, dims=(n_pad, Vp, width))
            row_loss, row_lse, _, _ = ops.ce_rows_fwd(logits[:n, :V], labels_sel)
        ctx.lin, ctx.flat, ctx.meta = lin, flat, (n, n_pad, V, Vp, width, words.shape, fused)
        ctx.save_for_backward(rows, logits, row_lse, idx, labels_sel)
        return row_loss

    @staticmethod
    def backward(ctx, g_row):
        rows, logits, row_lse, idx, labels_sel = ctx.saved_tensors
        lin, flat = ctx.lin, ctx.flat
        n, n_pad, V, Vp, width, wshape, fused = ctx.meta
        flat.begin_backward()
        g_row = g_row.contiguous().float()
        if fused:
            dl = ops.ce_fused_bwd(rows, flat.wview(lin.weight), lin.bias.data, labels_sel, row_lse, g_row, n, Vp)
        else:
            dl = ops.ce_rows_bwd_padded(logits[:n, :V], labels_sel, row_lse, g_row, V, rows.dtype, n_pad, Vp)
        drows = ops.gemm(dl, flat.wview(lin.weight), b_kmajor=True, pad_ok=True, dims=(n_pad, width, Vp))
        if lin.weight.requires_grad:
            ws = engine.gemm_workspace(dl.device) if dl.is_cuda and dl.dtype == torch.bfloat16 else None
            ops.gemm(dl, rows, a_kmajor=True, b_kmajor=True, out=flat.gview(lin.weight), accumulate=True,
                     split_k=engine._split_k(V, width, n_pad), a_colsum=flat.gview(lin.bias), pad_ok=True, dims=(V, width, n_pad), ws=ws)
        dwords = torch.zeros(wshape[0] * wshape[1], width, device=rows.device, dtype=rows.dtype)
        ops.scatter_rows_add(drows, idx, dwords)
        return dwords.view(wshape), None, None, None, None


def _mlm_selection(labels, dev):
    """(positions, labels) of the masked tokens on the device.  The selection is index arithmetic on the labels' own device (the host,
    normally) plus two small uploads; it is kept WITH the labels tensor object (version-checked), so a batch that is stepped on
    repeatedly -- a resident synthetic batch, gradient accumulation over the same micro-batch, a captured step -- uploads once and
    forward() contains no host-to-device copy."""
    tag = getattr(labels, "_dh_mlm", None)
    if tag is not None and tag[0] == labels._version and tag[1] == str(dev):
        return tag[2], tag[3]
    lab = labels.reshape(-1)
    sel = (lab != -100).nonzero(as_tuple=False).reshape(-1)
    sel_d = engine.to_device_async(sel, dev)
    labels_sel = engine.to_device_async(lab[sel.to(lab.device)], dev)
    labels._dh_mlm = (labels._version, str(dev), sel_d, labels_sel)
    return sel_d, labels_sel


def mlm_loss(words, labels, lin, flat):
    """mean CE over positions with labels != -100 (declip.py:331-334)."""
    idx, labels_sel = _mlm_selection(labels, words.device)
    return MlmHeadFn.apply(words, idx, labels_sel, lin, flat).mean()


def mlm_loss_packed(words_p, pk, n_captions, labels, lin, flat):
    """mlm_loss on the PACKED word features [rows_pad, width] of engine.TextTowerPackedFn: the masked positions (bi, l) of the first
    `n_captions` captions are rows cu[bi] + l (a masked token lies inside its caption)."""
    L = labels.shape[-1]
    sel_d, labels_sel = _mlm_selection(labels, words_p.device)
    idx = pk.cu[:n_captions].long()[sel_d // L] + sel_d % L
    return MlmHeadFn.apply(words_p.unsqueeze(0), idx, labels_sel, lin, flat).mean()
