"""Host-side engine: flat parameter store + the fused tower / loss autograd Functions.

MI355X-first layout: all parameters live in ONE flat fp32 buffer (288 GB HBM: no
reason to scatter 300 tensors), their gradients in one flat fp32 buffer written in place by
the HIP kernels (dW GEMMs accumulate straight into it), and -- in bf16 mode -- a flat bf16
mirror refreshed by one cast kernel per step.  Gradient all-reduce then works on contiguous
slices of the flat buffer (declip_amd/dist.py) without any packing copies.

Every arithmetic step goes through declip_amd.ops (= the C-ABI); torch is used for memory,
streams and the autograd graph plumbing only.
"""
import os
import threading

import torch

from . import ops
from .lib import DeclipHipError, EPI_DGELU, EPI_GELU, EPI_NONE

ALIGN = 64  # elements; keeps every parameter 256-byte aligned inside the flat buffers
SLACK = 1 << 16  # zero elements after the last parameter: padded-operand over-reads (MLM head, pad_ok GEMMs) stay in bounds


def _require_gpu(p, name):
    if not p.is_cuda:
        raise DeclipHipError("parameter %s is not on the GPU: call model.cuda() first (no CPU path)" % name)


class FlatParams:
    """Re-homes a module's parameters into flat buffers (values, grads, bf16 mirror)."""

    def __init__(self, module, act_dtype=torch.bfloat16):
        self.module = module
        self.act_dtype = act_dtype
        self.params = []
        self.index = {}
        self.total = 0
        self.flat_p = self.flat_g = self.flat_b = None
        self.anchor = None
        self._in_backward = False
        self.mirror_fresh = False
        self.reducer = None          # set by dist.DistModule
        self.names = {}
        self.side_streams = []       # extra HIP streams that towers run on (see side_stream); joined wherever params/grads are consumed
        self._zero_event = None
        self._pending_uses = {}      # id(tower) -> forwards of this step whose backward has not run yet (see tower_forward)
        self._early_ok = True        # may grads_ready() hand ranges to the reducer in the backward that is running now?
        # The fused AdamW writes the bf16 mirror in the pass that updates the master weights (optim.FlatAdamW.step), so the next
        # step needs no cast (0.23 ms / 0.9 GB per CLIP step).  Everything ELSE that writes weights says so: params_changed() --
        # called by the load_state_dict hooks below, DistModule.broadcast_params and a re-attach; code that edits weight matrices
        # through p.data must call it too (the solver's own in-place edits are the temperature scalars, read as fp32).
        # DH_MIRROR_TRUST=0 restores the unconditional cast of every step.
        self.trust_mirror = os.environ.get("DH_MIRROR_TRUST", "1") == "1"
        self._mirror_by_opt = False  # the valid mirror was written by the fused optimizer (any other optimizer: recast every step)
        for m in module.modules():
            if hasattr(m, "register_load_state_dict_post_hook"):
                m.register_load_state_dict_post_hook(lambda mod, incompatible, store=self: store.params_changed())

    def params_changed(self):
        """Weights were written by something other than the fused optimizer: the bf16 mirror is recast at the next begin_step()."""
        self.mirror_fresh = False
        self._mirror_by_opt = False

    def mirror_written_by_optimizer(self):
        """optim.FlatAdamW.step: the bf16 mirror was written together with the master weights."""
        self.mirror_fresh = self.flat_b is not None
        self._mirror_by_opt = self.mirror_fresh

    # ------------------------------------------------------------------ construction
    def attach(self):
        seen = set()
        off = 0
        self.params, self.index = [], {}
        for name, p in self.module.named_parameters():
            if id(p) in seen:
                continue
            seen.add(id(p))
            _require_gpu(p, name)
            if p.dtype != torch.float32:
                raise DeclipHipError("parameter %s must be fp32 master weights (got %s)" % (name, p.dtype))
            n = p.numel()
            self.index[id(p)] = (off, n)
            self.names[id(p)] = name
            self.params.append(p)
            off += (n + ALIGN - 1) // ALIGN * ALIGN
        self.total = off
        dev = self.params[0].device
        self.flat_p = torch.zeros(off + SLACK, device=dev, dtype=torch.float32)[:off]
        self.flat_g = torch.zeros(off + SLACK, device=dev, dtype=torch.float32)[:off]
        for p in self.params:
            o, n = self.index[id(p)]
            view = self.flat_p[o:o + n].view(p.shape)
            view.copy_(p.data)
            p.data = view
            p.grad = None
        if self.act_dtype == torch.bfloat16:
            self.flat_b = torch.zeros(off + SLACK, device=dev, dtype=torch.bfloat16)[:off]
        self.anchor = torch.zeros(1, device=dev, dtype=torch.float32, requires_grad=True)
        self.mirror_fresh = False
        self._mirror_by_opt = False
        return self

    def attached(self):
        if self.flat_p is None:
            return False
        p = self.params[0]
        o, n = self.index[id(p)]
        return p.data_ptr() == self.flat_p.data_ptr() + 4 * o

    def ensure(self):
        if not self.attached():
            self.attach()
        return self

    # ------------------------------------------------------------------ views
    def gview(self, p):
        o, n = self.index[id(p)]
        return self.flat_g[o:o + n].view(p.shape)

    def wview(self, p):
        """weight in the compute dtype (bf16 mirror in bf16 mode, the fp32 master otherwise)."""
        if self.flat_b is None:
            return p.data
        o, n = self.index[id(p)]
        return self.flat_b[o:o + n].view(p.shape)

    def span(self, params):
        """(lo, hi) flat range covering the given parameters (for bucketed gradient reduction)."""
        lo = min(self.index[id(p)][0] for p in params)
        hi = max(self.index[id(p)][0] + self.index[id(p)][1] for p in params)
        return lo, hi

    # ------------------------------------------------------------------ tower streams
    def side_stream(self, i=0):
        """The i-th extra stream of this store.  Independent towers (image / text) are enqueued on different streams so that
        the ragged last round of one tower's persistent GEMMs, its HBM-bound LayerNorm / attention launches and the other
        tower's MFMA work fill each other's idle CUs (one 256x256 workgroup owns a CU; the hardware queues interleave at
        workgroup granularity)."""
        if not self.side_streams and hasattr(torch.autograd.graph, "set_warn_on_accumulate_grad_stream_mismatch"):
            # the shared anchor leaf is consumed by towers on different streams on purpose; its gradient is never used
            torch.autograd.graph.set_warn_on_accumulate_grad_stream_mismatch(False)
        while len(self.side_streams) <= i:
            # DH_SIDE_PRIORITY=-1: the side stream (CLIP: the longer image tower) gets its workgroups dispatched first when both
            # streams have work pending (A/B switch; default: equal priority)
            self.side_streams.append(torch.cuda.Stream(device=self.flat_p.device, priority=int(os.environ.get("DH_SIDE_PRIORITY", "0"))))
        return self.side_streams[i]

    def join_streams(self):
        """The current stream waits for everything enqueued on the side streams (stream-ordered, no host sync)."""
        if self.side_streams:
            cur = torch.cuda.current_stream(self.flat_p.device)
            for s in self.side_streams:
                if s != cur:
                    cur.wait_stream(s)

    def refresh_mirror(self):
        if self.flat_b is not None and not self.mirror_fresh:
            ops.cast(self.flat_p, self.flat_b)
            self.mirror_fresh = True
            self._mirror_by_opt = False

    def begin_step(self):
        """Call at the start of every forward in training: parameters may have changed."""
        self.ensure()
        if self._in_backward:        # a backward that raised never reached its end-of-pass callback: do not carry its state over
            self._in_backward = False
            self._zero_event = None
            self._pending_uses = {}
        self.join_streams()
        if not (self.trust_mirror and self._mirror_by_opt):
            self.mirror_fresh = False
        self.refresh_mirror()

    # ------------------------------------------------------------------ backward protocol
    def begin_backward(self):
        """First engine backward of an autograd pass: establish the accumulate-into contract."""
        if self._in_backward:
            if self._zero_event is not None:     # a tower on another stream: its gradient writes must follow the zeroing
                torch.cuda.current_stream(self.flat_p.device).wait_event(self._zero_event)
            return
        self._in_backward = True
        # gradients are being produced, so an optimizer is about to write the weights: only the fused AdamW re-asserts that it left
        # a valid mirror behind (mirror_written_by_optimizer); after any other optimizer the next begin_step() recasts
        self._mirror_by_opt = False
        base, end = self.flat_g.data_ptr(), self.flat_g.data_ptr() + 4 * self.total
        live = [p for p in self.params if p.grad is not None and base <= p.grad.data_ptr() < end]
        if live and self.reducer is not None and self.reducer.distributed():
            # these gradients were all-reduced at the end of the previous backward(): reducing the running sum again would count
            # the earlier micro-batch once per rank
            self._in_backward = False
            raise DeclipHipError("gradient accumulation over several backward() calls is not supported with data parallelism: "
                                 "call optimizer.zero_grad() before every backward()")
        self._ft_pending = None
        if not live:
            # the common case: optimizer.zero_grad() ran.  Round 5: the block weights (80 % of the buffer) are NOT cleared -- the
            # first weight-gradient GEMM of the step that reaches a block writes its slots (claim_first_touch); one launch clears
            # everything else, and _end_backward clears the slots of blocks that no backward reached.
            ft = self._first_touch_plan()
            if ft is None:
                self.flat_g.zero_()
            else:
                ops.zero_ranges(self.flat_g, ft["table"], ft["n"], ft["max_len"])
                self._ft_pending = set(ft["ids"])
        else:                                    # accumulate semantics: keep live views, zero the rest
            for p in self.params:
                if p.grad is None or not (base <= p.grad.data_ptr() < end):
                    self.gview(p).zero_()
        if self.side_streams:
            self._zero_event = torch.cuda.Event()
            self._zero_event.record(torch.cuda.current_stream(self.flat_p.device))
        if self.reducer is not None:
            self.reducer.begin()
        torch.autograd.Variable._execution_engine.queue_callback(self._end_backward)

    # ---- first-touch weight gradients (round 5; VERDICT r4 next #4: the 605 MB zero-fill and its read-back in the reduce pass)
    def register_first_touch(self, params):
        """`params` (the four weight matrices of a transformer block) get their gradient from the block's weight-gradient GEMMs
        and from nothing else: candidates for "the first GEMM of the step writes the slot" instead of zero-fill + accumulate."""
        reg = self.__dict__.setdefault("_ft_params", {})
        for p in params:
            if id(p) in self.index and p.requires_grad:
                reg[id(p)] = p
        self.__dict__.pop("_ft_plan", None)

    def _first_touch_plan(self):
        """ranges of the flat gradient buffer that begin_backward still clears = the complement of the registered slots (cached)"""
        mode = os.environ.get("DH_FIRST_TOUCH", "1")             # 0 off; 1 on for device buffers; force: also on the host (CPU tests with the mock ops)
        if mode == "0" or (not self.flat_g.is_cuda and mode != "force"):
            return None
        reg = self.__dict__.get("_ft_params")
        if not reg:
            return None
        plan = self.__dict__.get("_ft_plan")
        if plan is not None and plan["key"] == (self.flat_g.data_ptr(), len(reg)):
            return plan
        # a slot = the parameter's elements rounded up to the 16-byte vectors the clearing kernel writes: the rest of its ALIGN padding
        # belongs to the cleared complement (the norm / all-reduce passes read the whole buffer; nothing else ever writes the padding)
        slots = sorted((self.index[i][0], (self.index[i][1] + 3) // 4 * 4) for i in reg)
        ranges, cur = [], 0
        for o, n in slots:
            if o > cur:
                ranges.append((cur, o))
            cur = max(cur, o + n)
        if cur < self.total:
            ranges.append((cur, self.total))
        if any(a % 4 or b % 4 for a, b in ranges):         # dh_zero_ranges writes 16-byte vectors
            raise DeclipHipError("first-touch plan: a range of the gradient buffer is not a multiple of 4 elements: %r" % ([r for r in ranges if r[0] % 4 or r[1] % 4][:3],))
        table = torch.tensor(ranges if ranges else [(0, 0)], dtype=torch.int64, device=self.flat_g.device)
        plan = dict(key=(self.flat_g.data_ptr(), len(reg)), table=table, n=len(ranges), max_len=max([b - a for a, b in ranges] + [0]), ids=list(reg))
        self._ft_plan = plan
        return plan

    def claim_first_touch(self, params):
        """True once per step and parameter group: the caller's weight-gradient GEMMs are the first contribution to these slots
        in this backward pass and must WRITE them (accumulate = 2).  False: accumulate (a second view through the same tower,
        gradient accumulation over several backward() calls, first-touch switched off)."""
        pend = self.__dict__.get("_ft_pending")
        if not pend:
            return False
        ids = [id(p) for p in params]
        if all(i in pend for i in ids):
            pend.difference_update(ids)
            return True
        return False

    def _end_backward(self):
        self._in_backward = False
        self._zero_event = None
        self._pending_uses = {}
        self._early_ok = True
        self.join_streams()                      # gradients written on the side streams are final from here on
        pend = self.__dict__.get("_ft_pending")
        if pend:                                 # blocks that no backward reached (a frozen or unused tower): their slots were not cleared at the start
            reg = self.__dict__.get("_ft_params", {})
            for i in pend:
                self.gview(reg[i]).zero_()
        self._ft_pending = None
        for p in self.params:
            if not p.requires_grad or getattr(p, "_dh_grad_none", False):   # parameters off the path keep grad None (torch semantics)
                continue
            view = self.gview(p)
            if p.grad is None:
                p.grad = view
            elif p.grad.data_ptr() != view.data_ptr():
                view.add_(p.grad)   # a parameter that torch autograd handled itself
                p.grad = view
        if self.reducer is not None:
            self.reducer.finish()

    def before_replay(self):
        """begin_step() for a step that is about to be REPLAYED from a captured graph: the capture holds a weight cast only if one
        was due at capture time, so weights written since by anything but the fused optimizer are recast here, eagerly."""
        if not (self.trust_mirror and self._mirror_by_opt):
            self.mirror_fresh = False
        self.refresh_mirror()

    def after_replay(self):
        """A captured step (graph.GraphedStep) was replayed: its kernels and collectives wrote the flat gradient buffer, but the
        Python half of the backward protocol did not run.  Redo what a replay cannot: `p.grad` views that optimizer.zero_grad()
        set to None point at the flat buffer again (as _end_backward leaves them), and the bf16 mirror counts as written by
        whatever optimizer steps next (begin_backward's reset)."""
        self._mirror_by_opt = False
        for p in self.params:
            if not p.requires_grad or getattr(p, "_dh_grad_none", False):
                continue
            if p.grad is None:
                p.grad = self.gview(p)

    def tower_forward(self, tower, needs_grad):
        """A tower Function ran forward and will run backward: one more pending use of its parameters in this step."""
        if needs_grad:
            self._pending_uses[id(tower)] = self._pending_uses.get(id(tower), 0) + 1

    def tower_backward(self, tower):
        """Start of a tower Function's backward.  A tower that ran forward more than once in this step (encode_image called per view,
        ...) accumulates into the same gradient ranges once per use: only the LAST of those backwards may release ranges to the
        bucketed all-reduce (an earlier release would reduce a partial gradient and add the rest on top of the reduced values).
        Autograd runs the Functions of one device on one thread, so the flag set here holds for this Function's grads_ready calls."""
        left = self._pending_uses.get(id(tower), 1) - 1
        self._pending_uses[id(tower)] = left
        self._early_ok = left <= 0

    def grads_ready(self, params):
        """gradients of `params` are final for this backward pass (bucketed reduction may start).  The range handed over is the
        parameter's whole ALIGN-padded slot: slots of neighbouring parameters then touch exactly, and the reducer never has to
        bridge a gap (bridging used to sweep a not-yet-final small parameter sitting in such a gap into an early bucket)."""
        if self.reducer is not None and self._early_ok:
            for p in params:
                o, n = self.index[id(p)]
                self.reducer.ready(o, o + (n + ALIGN - 1) // ALIGN * ALIGN)


# ---------------------------------------------------------------------------------------------
# transformer blocks (base_transformer.py:29-79)
# ---------------------------------------------------------------------------------------------
class BlockRefs:
    """Tensors of one ResidualAttentionBlock: weights in compute dtype, fp32 biases/LN, grad views."""

    def __init__(self, flat, blk):
        a = blk.attn
        self.params = [a.in_proj_weight, a.in_proj_bias, a.out_proj.weight, a.out_proj.bias, blk.ln_1.weight,
                       blk.ln_1.bias, blk.mlp.c_fc.weight, blk.mlp.c_fc.bias, blk.mlp.c_proj.weight,
                       blk.mlp.c_proj.bias, blk.ln_2.weight, blk.ln_2.bias]
        w, g = flat.wview, flat.gview
        self.w_in, self.b_in, self.w_out, self.b_out = w(a.in_proj_weight), a.in_proj_bias.data, w(a.out_proj.weight), a.out_proj.bias.data
        self.ln1_w, self.ln1_b, self.ln2_w, self.ln2_b = blk.ln_1.weight.data, blk.ln_1.bias.data, blk.ln_2.weight.data, blk.ln_2.bias.data
        self.w_fc, self.b_fc, self.w_proj, self.b_proj = w(blk.mlp.c_fc.weight), blk.mlp.c_fc.bias.data, w(blk.mlp.c_proj.weight), blk.mlp.c_proj.bias.data
        self.g_w_in, self.g_b_in, self.g_w_out, self.g_b_out = g(a.in_proj_weight), g(a.in_proj_bias), g(a.out_proj.weight), g(a.out_proj.bias)
        self.g_ln1_w, self.g_ln1_b, self.g_ln2_w, self.g_ln2_b = g(blk.ln_1.weight), g(blk.ln_1.bias), g(blk.ln_2.weight), g(blk.ln_2.bias)
        self.g_w_fc, self.g_b_fc, self.g_w_proj, self.g_b_proj = g(blk.mlp.c_fc.weight), g(blk.mlp.c_fc.bias), g(blk.mlp.c_proj.weight), g(blk.mlp.c_proj.bias)
        self.eps1, self.eps2 = blk.ln_1.eps, blk.ln_2.eps
        self.trainable = all(p.requires_grad for p in self.params)
        self._cparams = None
        self.flat = flat
        self.weights = [a.in_proj_weight, a.out_proj.weight, blk.mlp.c_fc.weight, blk.mlp.c_proj.weight]
        if self.trainable:
            flat.register_first_touch(self.weights)   # their gradients come from this block's weight-gradient GEMMs only

    def cparams(self):
        """lib.BlockParams of this block (dh_block_fwd / dh_block_bwd), filled once: the views above never move while the flat
        store stays attached (block_refs re-creates the BlockRefs when it does not)."""
        if self._cparams is None:
            from .lib import BlockParams
            ptr = ops.ptr
            c = BlockParams()
            for n in ("w_in", "w_out", "w_fc", "w_proj", "b_in", "b_out", "b_fc", "b_proj", "ln1_w", "ln1_b", "ln2_w", "ln2_b",
                      "g_w_in", "g_w_out", "g_w_fc", "g_w_proj", "g_b_in", "g_b_out", "g_b_fc", "g_b_proj", "g_ln1_w", "g_ln1_b", "g_ln2_w", "g_ln2_b"):
                setattr(c, n, ptr(getattr(self, n)))
            c.eps1, c.eps2 = float(self.eps1), float(self.eps2)
            self._cparams = c
        return self._cparams


def block_refs(flat, blocks):
    """BlockRefs of a tower's blocks, cached on the block modules: building ~40 tensor views per block on every forward was
    ~1.5 ms of host time per CLIP step.  The cache key is the identity of the flat buffers (a re-attach replaces them)."""
    key = (flat.flat_p.data_ptr(), flat.flat_g.data_ptr(), 0 if flat.flat_b is None else flat.flat_b.data_ptr())
    out = []
    for blk in blocks:
        c = blk.__dict__.get("_dh_refs")
        if c is None or c[0] != key or any(p.data_ptr() != q for p, q in zip(c[1].params, c[2])):
            r = BlockRefs(flat, blk)
            c = (key, r, [p.data_ptr() for p in r.params])
            blk.__dict__["_dh_refs"] = c
        out.append(c[1])
    return out


def _split_k(mg, ng, kg):
    tiles = ((mg + 127) // 128) * ((ng + 127) // 128)
    s = max(1, min(1024 // max(tiles, 1), kg // 512))
    return s


_GEMM_WS = {}
# Scratch buffers that were replaced by larger ones.  A hipGraph captured earlier keeps the OLD address baked into its launches; if
# that buffer went back to the caching allocator it could be handed to an unrelated tensor that a later replay of the older graph
# then scribbles over.  Replaced buffers are therefore kept alive for the life of the process (a handful of growths:
# the row count of packed captions is bounded by b * L) -- but only once a capture has happened in this process (graph.GraphedStep calls
# note_capture() before its first one): an eager-only run gives outgrown buffers back to the allocator.
_RETIRED_SCRATCH = []
_CAPTURE_SEEN = False


def note_capture():
    """A stream capture is about to record launches that hold scratch addresses: from now on outgrown scratch buffers stay alive."""
    global _CAPTURE_SEEN
    _CAPTURE_SEEN = True


def _retire(buf):
    if _CAPTURE_SEEN:
        _RETIRED_SCRATCH.append(buf)


def gemm_workspace(device, nbytes=256 << 20):
    """Scratch for the split-K partial tiles of the weight-gradient GEMMs (one per device and stream; 256 MiB covers 32 slices of
    the largest tower weight, or 8 K-slices of the four weight gradients of a ViT-B block issued as one group)."""
    key = (device.index, torch.cuda.current_stream(device).cuda_stream)
    ws = _GEMM_WS.get(key)
    if ws is None or ws.numel() * 4 < nbytes:
        if ws is not None:
            _retire(ws)          # a captured graph may hold its address (see _RETIRED_SCRATCH)
        ws = torch.empty(nbytes // 4, device=device, dtype=torch.float32)
        _GEMM_WS[key] = ws
    return ws


def weight_grad(dy, x, gw, gb=None, first_touch=False):
    """gw[out,in] += dy[rows,out]^T x[rows,in]  (contraction over rows: both operands k-major);
    gb[out] += colsum(dy) fused into the same launch (bias gradient).  first_touch: gw / gb are WRITTEN (see DwGroup)."""
    ws = gemm_workspace(dy.device) if dy.is_cuda and dy.dtype == torch.bfloat16 else None
    ops.gemm(dy, x, a_kmajor=True, b_kmajor=True, out=gw, accumulate=2 if first_touch else True,
             split_k=_split_k(gw.shape[0], gw.shape[1], dy.shape[0]), a_colsum=gb, ws=ws)


class DwGroup:
    """The weight-gradient problems of one transformer block, issued together at the end of the block's backward: problems with
    the same contraction length (= row count of dy / x) go to ONE launch of the persistent GEMM (ops.gemm_dw_group, dh_gemm_group):
    a single 768 x 768 weight has 9 output tiles for 256 CUs, the four weights of a ViT-B block 108.  The dy / x tensors are
    kept alive until flush()."""

    def __init__(self, refs=None):
        """refs: the block's BlockRefs.  If this is the first backward of the step to reach the block, its weight / bias slots are
        WRITTEN by these GEMMs (accumulate = 2) -- FlatParams.begin_backward did not clear them -- otherwise accumulated into."""
        self.items = []
        self.first_touch = bool(refs is not None and refs.trainable and refs.flat.claim_first_touch(refs.weights))

    def add(self, dy, x, gw, gb=None):
        self.items.append((dy, x, gw, gb))

    def flush(self):
        by_rows = {}
        for it in self.items:
            by_rows.setdefault((it[0].shape[0], it[0].dtype), []).append(it)
        self.items = []
        for (_, dtype), grp in by_rows.items():
            ws = gemm_workspace(grp[0][0].device) if grp[0][0].is_cuda and dtype == torch.bfloat16 else None
            for i in range(0, len(grp), 4):
                chunk = grp[i:i + 4]
                if len(chunk) == 1:
                    weight_grad(*chunk[0], first_touch=self.first_touch)
                else:
                    ops.gemm_dw_group(chunk, ws=ws, first_touch=self.first_touch)


class LnGradBatch:
    """The LayerNorm weight / bias gradients of ONE tower backward, reduced together.

    Every LayerNorm backward leaves its per-block partials in a slice of a persistent arena (ops.layernorm_bwd_part) and the
    reduction into the gradient buffer is ONE launch for all of them at the end of the tower's backward (ops.ln_reduce_many) --
    the CLIP step had 51 reduce launches of ~5 us each plus their dispatch gaps.  Under data parallelism the gradients of a
    block may only be handed to the bucketed all-reduce once its LayerNorm gradients are final, so there the batch is flushed
    every `DIST_BLOCKS` blocks and `ready()` releases the parameters collected since (FlatParams.grads_ready)."""
    DIST_BLOCKS = 2

    def __init__(self, flat, tower):
        self.flat, self.key = flat, id(tower)
        self.items, self.off, self.pending, self.nblocks = [], 0, [], 0
        self.dist = flat.reducer is not None and flat.reducer.distributed()
        self.enabled = os.environ.get("DH_LN_BATCH", "1") == "1"
        # LayerNorms whose partials can be alive at once (until the next flush): all of the tower's, or DIST_BLOCKS blocks' worth
        n_ln = sum(1 for m in tower.modules() if isinstance(m, torch.nn.LayerNorm))
        self.n_live = max(2, min(n_ln, 2 * self.DIST_BLOCKS + 2) if self.dist else n_ln)

    def _slice(self, n, device):
        arenas = self.flat.__dict__.setdefault("_ln_arenas", {})
        key = (self.key, torch.cuda.current_stream(device).cuda_stream if device.type == "cuda" else 0)
        ar = arenas.get(key)
        if ar is None or self.off + n > ar.numel():
            # sized ONCE for the tower: every LayerNorm that can be pending at the same time, at the size of the one asking now
            # (the full-row ones come first in a backward; the pooled ln_post / ln_final are smaller) -- no 2x over-allocation, no
            # re-growth over several steps, and the first growth does not land in a captured step's private pool when the warm-up
            # steps ran.  (A grown arena replaces the old one; slices already handed out keep the old storage alive until the flush.)
            if ar is not None:
                _retire(ar)      # an earlier capture may hold its address
            ar = torch.empty(max(self.off + n, self.n_live * ((n + 63) // 64 * 64)), device=device, dtype=torch.float32)
            arenas[key] = ar
            self.off = 0
        part = ar[self.off:self.off + n]
        self.off += (n + 63) // 64 * 64
        return part

    def bwd(self, dy, x, w, mean, rstd, dw, db, dres=None):
        if not self.enabled:
            return ops.layernorm_bwd(dy, x, w, mean, rstd, dw, db, dres=dres)
        part = self._slice(ops.layernorm_bwd_ws_elems(x.shape[0], x.shape[1]), x.device)
        dx, nb = ops.layernorm_bwd_part(dy, x, w, mean, rstd, dw, db, part, dres=dres)
        self.items.append((part, nb, x.shape[1], dw, db))
        return dx

    def flush(self):
        if self.items:
            ops.ln_reduce_many(self.items)
        self.items, self.off = [], 0

    def ready(self, params, block=True):
        """FlatParams.grads_ready for `params`, once their LayerNorm gradients are reduced."""
        if not self.dist:
            self.flat.grads_ready(params)            # (no collective is waiting for them: the batch is flushed at the end)
            return
        self.pending += list(params)
        self.nblocks += 1 if block else 0
        if self.nblocks >= self.DIST_BLOCKS:
            self.release()

    def release(self):
        self.flush()
        if self.pending:
            self.flat.grads_ready(self.pending)
        self.pending, self.nblocks = [], 0


class _LnbSlot(threading.local):
    """The LnGradBatch of the tower backward that is running ON THIS THREAD: autograd runs the backward functions of one device
    on one thread, so two engine models on different GPUs in one process (or a backward inside a backward) keep their own."""
    cur = None


_LNB_SLOT = _LnbSlot()


def _ln_bwd(dy, x, w, mean, rstd, dw, db, dres=None):
    lnb = _LNB_SLOT.cur
    if lnb is not None:
        return lnb.bwd(dy, x, w, mean, rstd, dw, db, dres=dres)
    return ops.layernorm_bwd(dy, x, w, mean, rstd, dw, db, dres=dres)


def native_blocks():
    """One C-ABI call per transformer block and direction (dh_block_fwd / dh_block_bwd, csrc/block.hip) instead of one per kernel:
    default.  DH_BLOCK_NATIVE=0 composes the block from the per-op calls in Python (same kernels, same results; bench.py's
    per-GEMM event brackets need it)."""
    return os.environ.get("DH_BLOCK_NATIVE", "1") == "1" and ops.block_native_available()


_BWD_SCRATCH = {}


def _bwd_scratch(device, nbytes):
    """Temporaries of dh_block_bwd (du, dqkv, dh2, dx_mid, da, dh1), one buffer per device and stream, reused block after block
    (everything that reads it is enqueued on that stream before the next block overwrites it)."""
    key = (device.index, torch.cuda.current_stream(device).cuda_stream if device.type == "cuda" else 0)
    buf = _BWD_SCRATCH.get(key)
    if buf is None or buf.numel() < nbytes:
        if buf is not None:
            _retire(buf)
        buf = torch.empty(nbytes, device=device, dtype=torch.uint8)
        _BWD_SCRATCH[key] = buf
    return buf


class NativeSaved(object):
    """What dh_block_bwd needs of a block forward that ran through dh_block_fwd: the input, the activation slab, the geometry."""
    __slots__ = ("x", "act", "b", "L", "heads", "causal", "cu", "rows_valid", "buckets")

    def views(self):
        """The slab as the tensors the Python composition saves (block_fwd's tuple; no copies)."""
        from .lib import dt
        rows, d = self.x.shape
        total, off = ops.block_act_layout(dt(self.x), rows, d, self.heads, self.b, self.L)
        e = self.x.element_size()

        def t(name, shape, dtype):
            n = 1
            for v in shape:
                n *= v
            nb = n * (4 if dtype == torch.float32 else e)
            return self.act[off[name]:off[name] + nb].view(dtype).view(shape)
        dtp = self.x.dtype
        h1, qkv, a = t("h1", (rows, d), dtp), t("qkv", (rows, 3 * d), dtp), t("a", (rows, d), dtp)
        x_mid, h2, u, g = t("x_mid", (rows, d), dtp), t("h2", (rows, d), dtp), t("u", (rows, 4 * d), dtp), t("g", (rows, 4 * d), dtp)
        f = torch.float32
        mean1, rstd1, mean2, rstd2 = t("mean1", (rows,), f), t("rstd1", (rows,), f), t("mean2", (rows,), f), t("rstd2", (rows,), f)
        lse = t("lse", (self.b, self.heads, self.L), f)
        return self.x, mean1, rstd1, h1, qkv, a, lse, x_mid, mean2, rstd2, h2, u, g


def _block_cargs(x, r, b, L, heads, causal, cu, rows_valid, buckets=None):
    from .lib import BlockArgs, dt
    ptr = ops.ptr
    rows, d = x.shape
    # one struct per block and geometry, filled once (the parameter table is 26 pointers) and re-used call after call: the C side
    # reads it synchronously, forward and backward of a block never overlap on the host
    key = (x.dtype, rows, d, b, L, heads, bool(causal))
    cache = r.__dict__.setdefault("_cargs", {})
    a = cache.get(key)
    if a is None:
        a = BlockArgs()
        a.dtype, a.rows, a.d, a.heads, a.b, a.L, a.causal = dt(x), rows, d, heads, b, L, int(bool(causal))
        a.p = r.cparams()
        if len(cache) > 8:
            cache.clear()
        cache[key] = a
    a.cu, a.rows_valid = ptr(cu), int(rows_valid)
    if buckets is not None:
        a.seq_order, a.seq_ranges, a.L_short = ptr(buckets[0]), ptr(buckets[1]), int(buckets[2])
    else:
        a.seq_order, a.seq_ranges, a.L_short = None, None, 0
    a.ws, a.ws_bytes = None, 0
    if x.is_cuda and x.dtype == torch.bfloat16:
        ws = gemm_workspace(x.device)
        a.ws, a.ws_bytes = ptr(ws), ws.numel() * 4
    return a


def _block_fwd_native(x, r, b, L, heads, causal, save, cu=None, rows_valid=0, buckets=None):
    from .lib import dt
    ptr = ops.ptr
    rows, d = x.shape
    total, _ = ops.block_act_layout(dt(x), rows, d, heads, b, L)
    act = torch.empty(total, device=x.device, dtype=torch.uint8)
    x_out = torch.empty_like(x)
    a = _block_cargs(x, r, b, L, heads, causal, cu, rows_valid, buckets)
    a.save = int(bool(save))
    a.x, a.x_out, a.act, a.act_bytes = ptr(x), ptr(x_out), ptr(act), total
    ops.block_fwd(a)
    if not save:
        return x_out, None
    sv = NativeSaved()
    sv.x, sv.act, sv.b, sv.L, sv.heads, sv.causal, sv.cu, sv.rows_valid, sv.buckets = x, act, b, L, heads, causal, cu, rows_valid, buckets
    return x_out, sv


def _block_bwd_native(dx_out, r, sv):
    from .lib import dt
    ptr = ops.ptr
    x = sv.x
    rows, d = x.shape
    lnb = _LNB_SLOT.cur
    n = ops.layernorm_bwd_ws_elems(rows, d)
    part1, part2 = lnb._slice(n, x.device), lnb._slice(n, x.device)
    nscr = ops.block_bwd_scratch_bytes(dt(x), rows, d)
    scratch = _bwd_scratch(x.device, nscr)
    dx = torch.empty_like(x)
    a = _block_cargs(x, r, sv.b, sv.L, sv.heads, sv.causal, sv.cu, sv.rows_valid, sv.buckets)
    a.x, a.act, a.act_bytes = ptr(x), ptr(sv.act), sv.act.numel()
    a.dx_out, a.dx, a.scratch, a.scratch_bytes = ptr(dx_out), ptr(dx), ptr(scratch), nscr
    a.ln_part1, a.ln_part2, a.ln_part_bytes = ptr(part1), ptr(part2), n * 4
    a.dw_first_touch = int(r.trainable and r.flat.claim_first_touch(r.weights))       # the block's dW GEMMs write their slots (FlatParams.claim_first_touch)
    ops.block_bwd(a)
    lnb.items.append((part2, int(a.ln_nb2), d, r.g_ln2_w, r.g_ln2_b))
    lnb.items.append((part1, int(a.ln_nb1), d, r.g_ln1_w, r.g_ln1_b))
    return dx


def block_fwd(x, r, b, L, heads, causal, save):
    """x: [b*L, d].  Returns x_out; if `save`, also what block_bwd needs."""
    if native_blocks():
        return _block_fwd_native(x, r, b, L, heads, causal, save)
    h1, mean1, rstd1 = ops.layernorm_fwd(x, r.ln1_w, r.ln1_b, r.eps1)
    qkv = ops.gemm(h1, r.w_in, bias=r.b_in)
    a, lse = ops.attn_fwd(qkv, b, L, heads, causal)
    ws = gemm_workspace(x.device) if x.is_cuda and x.dtype == torch.bfloat16 else None   # tail-sliced tiles (v4 GEMM)
    x_mid = ops.gemm(a, r.w_out, bias=r.b_out, residual=x, ws=ws)
    h2, mean2, rstd2 = ops.layernorm_fwd(x_mid, r.ln2_w, r.ln2_b, r.eps2)
    u = torch.empty(x.shape[0], r.w_fc.shape[0], device=x.device, dtype=x.dtype) if save else None
    g = ops.gemm(h2, r.w_fc, bias=r.b_fc, epilogue=EPI_GELU, aux=u)
    x_out = ops.gemm(g, r.w_proj, bias=r.b_proj, residual=x_mid, ws=ws)
    saved = (x, mean1, rstd1, h1, qkv, a, lse, x_mid, mean2, rstd2, h2, u, g) if save else None
    return x_out, saved


def block_bwd(dx_out, r, saved, b, L, heads, causal):
    if isinstance(saved, NativeSaved):
        if _LNB_SLOT.cur is not None and _LNB_SLOT.cur.enabled and native_blocks() and dx_out.is_contiguous():
            return _block_bwd_native(dx_out, r, saved)
        saved = saved.views()
    x, mean1, rstd1, h1, qkv, a, lse, x_mid, mean2, rstd2, h2, u, g = saved
    # MLP: x_out = x_mid + gelu(h2 Wfc^T + bfc) Wproj^T + bproj
    dw = DwGroup(r)
    dw.add(dx_out, g, r.g_w_proj, r.g_b_proj)
    du = ops.gemm(dx_out, r.w_proj, b_kmajor=True, epilogue=EPI_DGELU, aux=u)
    dw.add(du, h2, r.g_w_fc, r.g_b_fc)
    ws = gemm_workspace(du.device) if du.is_cuda and du.dtype == torch.bfloat16 else None
    dh2 = ops.gemm(du, r.w_fc, b_kmajor=True, ws=ws)
    dx_mid = _ln_bwd(dh2, x_mid, r.ln2_w, mean2, rstd2, r.g_ln2_w, r.g_ln2_b, dres=dx_out)
    # attention: x_mid = x + attn(h1) Wout^T + bout
    dw.add(dx_mid, a, r.g_w_out, r.g_b_out)
    da = ops.gemm(dx_mid, r.w_out, b_kmajor=True, ws=ws)
    dqkv = ops.attn_bwd(qkv, a, da, lse, b, L, heads, causal)
    dw.add(dqkv, h1, r.g_w_in, r.g_b_in)
    dh1 = ops.gemm(dqkv, r.w_in, b_kmajor=True, ws=ws)
    dx = _ln_bwd(dh1, x, r.ln1_w, mean1, rstd1, r.g_ln1_w, r.g_ln1_b, dres=dx_mid)
    dw.flush()                         # the block's four weight gradients: one grouped launch (all inputs are final here)
    return dx


def pooled_last_block(width=None, heads=None, L=0):
    """The last block of a tower runs its query / attention / out_proj / MLP for the POOLED rows only (default since round 2:
    measured +4 % pairs/s alone, +18.6 % together with packed captions, profiles/r02_ab_switches.txt); DH_POOLED_LAST=0 restores
    the reference's dense schedule (same outputs)."""
    import os
    if os.environ.get("DH_POOLED_LAST", "1") != "1":
        return False
    # dh_attn_pooled_*: one wave per (sequence, head), lane = head dimension -> head dim 64, at most 128 keys; other geometries
    # (test-sized towers) keep the dense last block
    return width is None or (width == 64 * heads and L <= 128)


def text_packed_mode():
    """0 = padded captions as the reference computes them; 1 (default since round 2: +14 % pairs/s on the synthetic caption
    lengths, same outputs) = the text tower on the rows up to <|endoftext|> only with variable-length attention; 2 = packed rows,
    attention through the dense layout.  Environment DH_TEXT_PACKED."""
    import os
    m = os.environ.get("DH_TEXT_PACKED", "1")
    return int(m) if m in ("0", "1", "2") else 1


def block_fwd_pooled(x, r, sel, row0, nkeys, Lmax, heads, save):
    """The LAST ResidualAttentionBlock when only one row per sequence is used afterwards (CLS: visual_transformer.py:70-72; EOT:
    text_transformer.py:203): K and V are projected for every row, everything else -- the query projection, the attention of that
    one query, out_proj, both residuals, ln_2 and the MLP -- runs on the b pooled rows instead of all rows (82 % of the block's
    flops at L = 50, 81 % at L = 77).  Exactly the same values for the pooled rows as block_fwd.
    x [R, d]; sel int64 [b] = the pooled rows; row0 / nkeys int32 [b] = each sequence's key rows.  Returns x_out [b, d]."""
    d = x.shape[1]
    h1, mean1, rstd1 = ops.layernorm_fwd(x, r.ln1_w, r.ln1_b, r.eps1)
    ws = gemm_workspace(x.device) if x.is_cuda and x.dtype == torch.bfloat16 else None
    kv = ops.gemm(h1, r.w_in[d:], bias=r.b_in[d:], ws=ws)                 # k | v of every row
    h1s = ops.gather_rows(h1, sel)
    q = ops.gemm(h1s, r.w_in[:d], bias=r.b_in[:d], ws=ws)                 # the pooled rows' queries (ws: few tiles -> cut in K over the chip)
    a, lse = ops.attn_pooled_fwd(q, kv, row0, nkeys, heads, Lmax)
    xs = ops.gather_rows(x, sel)
    x_mid = ops.gemm(a, r.w_out, bias=r.b_out, residual=xs, ws=ws)
    h2, mean2, rstd2 = ops.layernorm_fwd(x_mid, r.ln2_w, r.ln2_b, r.eps2)
    u = torch.empty(x_mid.shape[0], r.w_fc.shape[0], device=x.device, dtype=x.dtype) if save else None
    g = ops.gemm(h2, r.w_fc, bias=r.b_fc, epilogue=EPI_GELU, aux=u)
    x_out = ops.gemm(g, r.w_proj, bias=r.b_proj, residual=x_mid, ws=ws)
    saved = (x, mean1, rstd1, h1, kv, h1s, q, a, lse, x_mid, mean2, rstd2, h2, u, g) if save else None
    return x_out, saved


def block_bwd_pooled(dx_out, r, saved, sel, row0, nkeys, Lmax, heads):
    """dx_out [b, d] (gradient of the pooled rows' block output) -> gradient of the block input x [R, d]."""
    x, mean1, rstd1, h1, kv, h1s, q, a, lse, x_mid, mean2, rstd2, h2, u, g = saved
    d = x.shape[1]
    dw = DwGroup(r)                    # four problems over the b pooled rows + the k|v projection over all rows
    dw.add(dx_out, g, r.g_w_proj, r.g_b_proj)
    du = ops.gemm(dx_out, r.w_proj, b_kmajor=True, epilogue=EPI_DGELU, aux=u)
    dw.add(du, h2, r.g_w_fc, r.g_b_fc)
    ws = gemm_workspace(x.device) if x.is_cuda and x.dtype == torch.bfloat16 else None
    dh2 = ops.gemm(du, r.w_fc, b_kmajor=True, ws=ws)
    dx_mid = _ln_bwd(dh2, x_mid, r.ln2_w, mean2, rstd2, r.g_ln2_w, r.g_ln2_b, dres=dx_out)       # [b, d]
    dw.add(dx_mid, a, r.g_w_out, r.g_b_out)
    da = ops.gemm(dx_mid, r.w_out, b_kmajor=True, ws=ws)
    dq, dkv = ops.attn_pooled_bwd(q, kv, da, lse, row0, nkeys, heads, Lmax, ordered=True)   # row0 = i * L (CLS, padded captions) or cu_seqlens[i] (packed)
    dw.add(dq, h1s, r.g_w_in[:d], r.g_b_in[:d])
    dw.add(dkv, h1, r.g_w_in[d:], r.g_b_in[d:])
    dh1 = ops.gemm(dkv, r.w_in[d:], b_kmajor=True, ws=ws)                 # [R, d]
    ops.scatter_rows_add(ops.gemm(dq, r.w_in[:d], b_kmajor=True, ws=ws), sel, dh1)
    dx = _ln_bwd(dh1, x, r.ln1_w, mean1, rstd1, r.g_ln1_w, r.g_ln1_b)
    ops.scatter_rows_add(dx_mid, sel, dx)                                 # the residual path of the pooled rows (x_mid = x[sel] + ...)
    dw.flush()
    return dx


def to_device_async(t, dev):
    """Host -> device without stalling the host: a pageable `.to(device)` waits for everything already enqueued on the stream
    (in the DeCLIP step: the whole text tower, ~35 ms at b=512); from pinned memory the copy is just another stream operation."""
    dev = torch.device(dev)
    if t.device == dev or dev.type != "cuda":
        return t.to(dev)
    return t.pin_memory().to(dev, non_blocking=True)


def _to_act(t, dtype):
    if t.dtype == dtype:
        return t.contiguous()
    out = torch.empty(t.shape, device=t.device, dtype=dtype)
    return ops.cast(t.contiguous(), out)


# ---------------------------------------------------------------------------------------------
# vision tower (image_encoder/visual_transformer.py:55-82)
# ---------------------------------------------------------------------------------------------
_CLS_POOL = {}


def _cls_pool(b, L, device):
    """(pooled rows, first key row, key count) of the CLS rows of b images of L tokens: constants of the geometry, built once
    (five tiny torch launches per step otherwise)."""
    key = (b, L, str(device))
    p = _CLS_POOL.get(key)
    if p is None:
        seq = torch.arange(b, device=device)
        p = ((seq * L).contiguous(), (seq * L).to(torch.int32).contiguous(), torch.full((b,), L, device=device, dtype=torch.int32))
        if len(_CLS_POOL) > 64:
            _CLS_POOL.clear()
        _CLS_POOL[key] = p
    return p


class VisionTowerFn(torch.autograd.Function):
    """forward(anchor, images, tower, c0, want_dense, want_feature, n_views) ->
         proj [V*b,E] fp32 (, dense [V*b,np,width] act dtype)(, feature [V*b,width] act dtype)
    n_views channel-stacked views (data/transforms.py:38-54) are encoded in ONE pass as a batch of V*b
    (view-major), which is arithmetically identical to V separate passes (no cross-sample op in the tower)."""

    @staticmethod
    def forward(ctx, anchor, images, tower, c0, want_dense, want_feature, n_views=1):
        flat = tower._flat()
        dtype = flat.act_dtype
        b0 = images.shape[0]
        b = b0 * n_views
        P, width, heads = tower.patch_size, tower.width, tower.heads
        npatch = (images.shape[2] // P) * (images.shape[3] // P)
        L = npatch + 1
        save = bool(ctx.needs_input_grad[0])
        flat.tower_forward(tower, save)
        if n_views == 1:
            rows = ops.im2row(images, c0, P, dtype)
        else:
            rows = torch.empty(b * npatch, 3 * P * P, device=images.device, dtype=dtype)
            for v in range(n_views):
                ops.im2row(images, c0 + 3 * v, P, dtype, out=rows[v * b0 * npatch:(v + 1) * b0 * npatch])
        wconv = flat.wview(tower.conv1.weight).view(width, -1)
        patches = ops.gemm(rows, wconv)
        x0 = ops.vit_assemble_fwd(patches, tower.class_embedding.data, tower.positional_embedding.data, b, npatch)
        x, mean0, rstd0 = ops.layernorm_fwd(x0, tower.ln_pre.weight.data, tower.ln_pre.bias.data, tower.ln_pre.eps)
        refs = block_refs(flat, tower.transformer.resblocks)
        saved_blocks = []
        pool = None
        if pooled_last_block(width, heads, L) and not want_dense and refs:
            # the last block only for the CLS rows (their keys: the L rows of the image)
            pool = _cls_pool(b, L, images.device)
        for r in (refs[:-1] if pool is not None else refs):
            x, s = block_fwd(x, r, b, L, heads, False, save=save)
            saved_blocks.append(s)
        if pool is not None:
            pooled, s = block_fwd_pooled(x, refs[-1], pool[0], pool[1], pool[2], L, heads, save)
            saved_blocks.append(s)
        else:
            pooled = ops.pool_rows_fwd(x, None, b, L)
        feat, mean_p, rstd_p = ops.layernorm_fwd(pooled, tower.ln_post.weight.data, tower.ln_post.bias.data, tower.ln_post.eps)
        out = ops.gemm(feat, flat.wview(tower.proj), b_kmajor=True, out_dtype=torch.float32)
        ctx.tower, ctx.refs, ctx.saved_blocks = tower, refs, saved_blocks
        ctx.misc = (b, L, npatch, rows if tower.conv1.weight.requires_grad else None, x0, mean0, rstd0, pooled, mean_p, rstd_p, feat, x)
        ctx.want = (want_dense, want_feature)
        ctx.pool = pool
        outs = [out]
        if want_dense:
            outs.append(x.view(b, L, width)[:, 1:, :])
        if want_feature:
            outs.append(feat)
        return tuple(outs) if len(outs) > 1 else out

    @staticmethod
    def backward(ctx, *grads):
        flat = ctx.tower._flat()
        lnb = LnGradBatch(flat, ctx.tower)
        prev, _LNB_SLOT.cur = _LNB_SLOT.cur, lnb
        try:
            out = VisionTowerFn._backward_impl(ctx, lnb, *grads)
            lnb.release()                    # ONE reduce launch for the tower's LayerNorm gradients; the rest of its parameters released
            return out
        finally:
            _LNB_SLOT.cur = prev

    @staticmethod
    def _backward_impl(ctx, lnb, *grads):
        tower = ctx.tower
        flat = tower._flat()
        flat.begin_backward()
        flat.tower_backward(tower)
        dtype = flat.act_dtype
        b, L, npatch, rows, x0, mean0, rstd0, pooled, mean_p, rstd_p, feat, x_final = ctx.misc
        width, heads = tower.width, tower.heads
        want_dense, want_feature = ctx.want
        dout = grads[0]
        gi = 1
        ddense = dfeat_extra = None
        if want_dense:
            ddense = grads[gi]; gi += 1
        if want_feature:
            dfeat_extra = grads[gi]; gi += 1
        g = flat.gview
        dfeat = None
        if dout is not None:
            dout_a = _to_act(dout, dtype)
            # out = feat @ proj  (proj [width, E])
            ops.gemm(feat, dout_a, a_kmajor=True, b_kmajor=True, out=g(tower.proj), accumulate=True)
            dfeat = ops.gemm(dout_a, flat.wview(tower.proj))          # [b,E] x proj[width,E]^T
        if dfeat_extra is not None:
            de = _to_act(dfeat_extra, dtype)
            dfeat = de if dfeat is None else dfeat.add_(de)
        pool = ctx.pool
        blocks = list(zip(ctx.refs, ctx.saved_blocks))
        if dfeat is not None:
            dpooled = _ln_bwd(dfeat, pooled, tower.ln_post.weight.data, mean_p, rstd_p, g(tower.ln_post.weight), g(tower.ln_post.bias))
            dx = ops.pool_rows_bwd(dpooled, None, b, L) if pool is None else None
        else:
            dpooled = None
            dx = torch.zeros(b * L, width, device=x_final.device, dtype=dtype)
        if ddense is not None:
            dx.view(b, L, width)[:, 1:, :].add_(ddense.to(dtype))
        lnb.ready([tower.proj, tower.ln_post.weight, tower.ln_post.bias], block=False)
        if pool is not None:
            r, s = blocks.pop()
            if dpooled is None:
                dpooled = torch.zeros(b, width, device=x_final.device, dtype=dtype)
            dx = block_bwd_pooled(dpooled, r, s, pool[0], pool[1], pool[2], L, heads)
            lnb.ready(r.params)
        for r, s in reversed(blocks):
            dx = block_bwd(dx, r, s, b, L, heads, False)
            lnb.ready(r.params)
        dx0 = _ln_bwd(dx, x0, tower.ln_pre.weight.data, mean0, rstd0, g(tower.ln_pre.weight), g(tower.ln_pre.bias))
        ops.vit_assemble_bwd(dx0, g(tower.class_embedding), g(tower.positional_embedding), b, npatch)
        if tower.conv1.weight.requires_grad:
            dpatch = dx0.view(b, L, width)[:, 1:, :].contiguous().view(b * npatch, width)
            weight_grad(dpatch, rows, g(tower.conv1.weight).view(width, -1))
        ctx.saved_blocks = ctx.misc = None
        return (torch.zeros_like(flat.anchor), None, None, None, None, None, None)


# ---------------------------------------------------------------------------------------------
# text tower (text_encoder/text_transformer.py:183-204)
# ---------------------------------------------------------------------------------------------
class TextTowerFn(torch.autograd.Function):
    """forward(anchor, ids, tower, want_dense) -> proj [b,E] fp32 (, words [b,ctx,width] act dtype)"""

    @staticmethod
    def forward(ctx, anchor, ids, tower, want_dense):
        flat = tower._flat()
        dtype = flat.act_dtype
        b, L = ids.shape
        width, heads = tower.width, tower.heads
        x = ops.text_embed_fwd(ids, tower.token_embedding.weight.data, tower.positional_embedding.data, dtype)
        refs = block_refs(flat, tower.transformer.resblocks)
        saved_blocks = []
        save = bool(ctx.needs_input_grad[0])
        flat.tower_forward(tower, save)
        eot = ids.argmax(dim=-1)                                    # text_transformer.py:203 (index arithmetic)
        pool = None
        if pooled_last_block(width, heads, L) and not want_dense and refs:
            # the last block only for the <|endoftext|> rows (their keys: the rows up to and including EOT: the causal mask)
            seq = torch.arange(b, device=ids.device)
            pool = ((seq * L + eot).contiguous(), (seq * L).to(torch.int32).contiguous(), (eot + 1).to(torch.int32).contiguous())
        for r in (refs[:-1] if pool is not None else refs):
            x, s = block_fwd(x, r, b, L, heads, True, save=save)
            saved_blocks.append(s)
        lnw, lnb = tower.ln_final.weight.data, tower.ln_final.bias.data
        if want_dense:
            words, mean_f, rstd_f = ops.layernorm_fwd(x, lnw, lnb, tower.ln_final.eps)
            feat = ops.pool_rows_fwd(words, eot, b, L)
            pooled = None
        else:
            if pool is not None:
                pooled, s = block_fwd_pooled(x, refs[-1], pool[0], pool[1], pool[2], L, heads, save)
                saved_blocks.append(s)
            else:
                pooled = ops.pool_rows_fwd(x, eot, b, L)
            feat, mean_f, rstd_f = ops.layernorm_fwd(pooled, lnw, lnb, tower.ln_final.eps)
            words = None
        tp = tower.text_projection
        out = ops.gemm(feat, flat.wview(tp.weight), bias=tp.bias.data, out_dtype=torch.float32)
        ctx.tower, ctx.refs, ctx.saved_blocks = tower, refs, saved_blocks
        ctx.misc = (b, L, ids, eot, x, pooled, mean_f, rstd_f, feat, want_dense)
        ctx.pool = pool
        if want_dense:
            return out, words.view(b, L, width)
        return out

    @staticmethod
    def backward(ctx, *grads):
        flat = ctx.tower._flat()
        lnb = LnGradBatch(flat, ctx.tower)
        prev, _LNB_SLOT.cur = _LNB_SLOT.cur, lnb
        try:
            out = TextTowerFn._backward_impl(ctx, lnb, *grads)
            lnb.release()                    # ONE reduce launch for the tower's LayerNorm gradients; the rest of its parameters released
            return out
        finally:
            _LNB_SLOT.cur = prev

    @staticmethod
    def _backward_impl(ctx, lnb, *grads):
        tower = ctx.tower
        flat = tower._flat()
        flat.begin_backward()
        flat.tower_backward(tower)
        dtype = flat.act_dtype
        b, L, ids, eot, x_final, pooled, mean_f, rstd_f, feat, want_dense = ctx.misc
        width, heads = tower.width, tower.heads
        g = flat.gview
        tp = tower.text_projection
        dout = grads[0]
        dwords = grads[1] if want_dense else None
        dfeat = None
        if dout is not None:
            dout_a = _to_act(dout, dtype)
            weight_grad(dout_a, feat, g(tp.weight), g(tp.bias))
            dfeat = ops.gemm(dout_a, flat.wview(tp.weight), b_kmajor=True)
        lnw = tower.ln_final.weight.data
        if want_dense:
            dw_total = ops.pool_rows_bwd(dfeat, eot, b, L) if dfeat is not None else torch.zeros(b * L, width, device=ids.device, dtype=dtype)
            if dwords is not None:
                dw_total.add_(dwords.reshape(b * L, width).to(dtype))
            dx = _ln_bwd(dw_total, x_final, lnw, mean_f, rstd_f, g(tower.ln_final.weight), g(tower.ln_final.bias))
        else:
            dpooled = _ln_bwd(dfeat, pooled, lnw, mean_f, rstd_f, g(tower.ln_final.weight), g(tower.ln_final.bias))
            dx = ops.pool_rows_bwd(dpooled, eot, b, L) if ctx.pool is None else None
        lnb.ready([tp.weight, tp.bias, tower.ln_final.weight, tower.ln_final.bias], block=False)
        blocks = list(zip(ctx.refs, ctx.saved_blocks))
        if ctx.pool is not None:
            r, s = blocks.pop()
            dx = block_bwd_pooled(dpooled, r, s, ctx.pool[0], ctx.pool[1], ctx.pool[2], L, heads)
            lnb.ready(r.params)
        for r, s in reversed(blocks):
            dx = block_bwd(dx, r, s, b, L, heads, True)
            lnb.ready(r.params)
        te, pe = tower.token_embedding.weight, tower.positional_embedding
        V = te.shape[0]
        ops.text_embed_bwd(ids, dx, g(te) if te.requires_grad else None, g(pe) if pe.requires_grad else None,
                           hot_ids=(0, V - 2, V - 1))          # pad, <|startoftext|>, <|endoftext|>
        ctx.saved_blocks = ctx.misc = None
        return (torch.zeros_like(flat.anchor), None, None, None)


# ---------------------------------------------------------------------------------------------
# text tower on PACKED captions (the default; DH_TEXT_PACKED=0 for the padded layout)
# ---------------------------------------------------------------------------------------------
class PackedCaptions:
    """Row bookkeeping of a caption batch in which only the tokens up to and including <|endoftext|> are rows.

    Under the causal mask (text_transformer.py:136-142) a token never attends to a later one, every other operation of the tower
    is per token, and only the EOT row is pooled (:203): the rows after EOT of the reference's [b, ctx] layout cannot reach the
    loss.  Dropping them changes no output and no gradient (the pad rows' gradients are exactly zero in the reference too) and
    removes ~45 % of the text tower's GEMM / LayerNorm work on the synthetic captions of SURVEY.md s8(d), more on real ones.
    Index arithmetic only (torch, on the device)."""

    L_SHORT = 48          # captions up to here run on the 3-key-block attention kernels (context 77: 5 blocks)

    def __init__(self, ids, tile, varlen=True):
        self.varlen = varlen                                             # attention on the packed rows (else: via the dense layout)
        b, L = ids.shape
        lens = ids.argmax(dim=-1) + 1                                    # EOT is the largest id (quirk 5)
        # the row count sizes the buffers, so the host has to know it.  It travels WITH the tensor object: set from the host copy
        # before the upload (model/transformer.py: no device read at all), or read back once for a device-resident batch that is
        # used again and again (bench.py, the synthetic loader) -- never keyed by address: allocators recycle those
        tag = getattr(ids, "_dh_rows", None)
        if tag is not None and tag[0] == ids._version:
            total = tag[1]
        else:
            total = int(lens.sum())                                      # one host read for this tensor object
            ids._dh_rows = (ids._version, total)
        self.b, self.L, self.rows = b, L, total
        self.rows_pad = rows_pad = (total + tile - 1) // tile * tile
        # a job-wide padded row count (set_rows_tag(rows_pad=...): the MAX over the ranks of a data-parallel job, dist.RowsSync) pads
        # this rank's rows further, so that every rank runs -- and captures -- a step of the same shape; the extra rows are more of
        # the zero rows behind the last caption
        forced = getattr(ids, "_dh_rows_pad", None)
        forced = forced[1] if (forced is not None and forced[0] == ids._version) else None
        if forced is not None:
            if forced < rows_pad or forced % tile:
                raise DeclipHipError("PackedCaptions: job-wide padded row count %d does not hold this rank's %d rows (tile %d)" % (forced, total, tile))
            self.rows_pad = rows_pad = forced
        if tag is not None and tag[0] == ids._version and hasattr(torch, "_assert_async"):
            # a host-side count that does not fit the device's fails loudly (device-side assert, no read-back).  What is checked is
            # the padded size, not `total` itself: a captured step is replayed for other batches of the same rows_pad
            n_dev = lens.sum()
            torch._assert_async((n_dev <= rows_pad) if forced is not None else ((n_dev <= rows_pad) & (n_dev > rows_pad - tile)))
        # Every index tensor below has a shape that depends on rows_pad only (never on `total`), and no launch argument carries
        # `total`: the whole bookkeeping is capturable, and a step captured for one batch replays for any batch with the same
        # rows_pad (graph.GraphedStep keys its graphs by it).  The rows [total, rows_pad) are a dummy run behind the last caption:
        # sequence index b, pos_idx = -1 (the embedding writes zeros there), id 0 with a zero gradient row.
        cu = torch.zeros(b + 1, device=ids.device, dtype=torch.int64)
        cu[1:] = lens.cumsum(0)
        self.cu = cu.to(torch.int32)
        lens_ext = torch.cat([lens, rows_pad - cu[b:]])
        seq = torch.repeat_interleave(torch.arange(b + 1, device=ids.device), lens_ext, output_size=rows_pad)
        valid = seq < b
        pos = torch.arange(rows_pad, device=ids.device) - cu[seq]
        self.pos_idx = torch.where(valid, pos, torch.full_like(pos, -1)).to(torch.int32).contiguous()
        self.pack_idx = torch.where(valid, seq * L + pos, torch.zeros_like(pos)).contiguous()   # dense row (bi * L + l) of every packed row (padding: row 0)
        self.ids_p = torch.where(valid, ids.reshape(-1)[self.pack_idx], torch.zeros_like(pos)).contiguous()
        l = torch.arange(L, device=ids.device)[None, :]
        # dense (bi, l) -> a packed row: its own when l < len, else the caption's first row (any finite values do: a padded query
        # is dropped again, a padded key is only seen by padded queries)
        self.unpack_idx = (cu[:-1, None] + torch.where(l < lens[:, None], l, torch.zeros_like(l))).reshape(-1).contiguous()
        self.eot_rows = (cu[1:] - 1).contiguous()
        self.row0, self.nkeys = self.cu[:-1].contiguous(), (self.cu[1:] - self.cu[:-1]).contiguous()   # the pooled last block's key rows
        # two length buckets for the attention kernels (dh_attn_bucketed_*): the captions of at most L_SHORT tokens first
        short = lens <= self.L_SHORT
        self.order = torch.sort((~short).to(torch.int32), stable=True)[1].to(torch.int32).contiguous()
        ns = short.sum().to(torch.int32).reshape(1)
        self.ranges = torch.cat([torch.zeros_like(ns), ns, ns, b - ns]).contiguous()


def _buckets(pk, x, heads):
    """(order, ranges, L_short) when the length-bucketed attention kernels are asked for (DH_ATTN_BUCKETS=1) and apply (bf16, head
    dimension 64, a context longer than the short bucket), else None.  Default OFF since round 6: the two launches per call are faster
    alone (fwd 43 -> 38 us, bwd 82 -> 73 us) and slower in the step, where the text tower's launches count through the CUs they hold
    (22.33 -> 22.27 ms, six of six interleaved pairs; profiles/r06_attention_variants.txt) -- and 22 dispatches fewer per step.  A model
    whose step spends a larger share in the text tower asks for them on its tower (`tower._dh_attn_buckets`: DeCLIP / DeFILIP, two caption
    views + the masked-LM pass: +0.6 % with buckets); the environment overrides both ways."""
    if x.dtype != torch.bfloat16 or x.shape[1] // heads != 64 or pk.L <= pk.L_SHORT or os.environ.get("DH_ATTN_BUCKETS", getattr(pk, "buckets_default", "0")) != "1":
        return None
    return pk.order, pk.ranges, pk.L_SHORT


def _rows_arg(pk, x, heads):
    """The valid row count as the attention kernels take it: -1 = "read cu_seqlens[b] on the device" wherever the kernels can
    (bf16, head dimension 64: nothing of a launch then depends on the batch's caption lengths but rows_pad), the host count else."""
    return -1 if (x.dtype == torch.bfloat16 and x.shape[1] // heads == 64) else pk.rows


def block_fwd_packed(x, r, pk, heads, save):
    """block_fwd on packed rows [rows_pad, d].  Attention: the variable-length kernels on the packed rows (pk.varlen), or -- the
    fallback that touches only long-verified kernels, DH_TEXT_PACKED=2 -- gather to the dense [b, L] layout and back."""
    if pk.varlen and native_blocks():
        return _block_fwd_native(x, r, pk.b, pk.L, heads, True, save, cu=pk.cu, rows_valid=_rows_arg(pk, x, heads), buckets=_buckets(pk, x, heads))
    h1, mean1, rstd1 = ops.layernorm_fwd(x, r.ln1_w, r.ln1_b, r.eps1)
    qkv = ops.gemm(h1, r.w_in, bias=r.b_in)
    if pk.varlen:
        bk = _buckets(pk, x, heads)
        if bk is not None:
            a, lse = ops.attn_bucketed_fwd(qkv, pk.cu, bk[0], bk[1], _rows_arg(pk, x, heads), pk.b, pk.L, bk[2], heads, True)
        else:
            a, lse = ops.attn_varlen_fwd(qkv, pk.cu, _rows_arg(pk, x, heads), pk.b, pk.L, heads, True)
        att_saved = (qkv, a)
    else:
        qkv_d = ops.gather_rows(qkv, pk.unpack_idx)
        a_d, lse = ops.attn_fwd(qkv_d, pk.b, pk.L, heads, True)
        a = ops.gather_rows(a_d, pk.pack_idx[:pk.rows], pk.rows_pad)
        att_saved = (qkv_d, a_d)
    ws = gemm_workspace(x.device) if x.is_cuda and x.dtype == torch.bfloat16 else None
    x_mid = ops.gemm(a, r.w_out, bias=r.b_out, residual=x, ws=ws)
    h2, mean2, rstd2 = ops.layernorm_fwd(x_mid, r.ln2_w, r.ln2_b, r.eps2)
    u = torch.empty(x.shape[0], r.w_fc.shape[0], device=x.device, dtype=x.dtype) if save else None
    g = ops.gemm(h2, r.w_fc, bias=r.b_fc, epilogue=EPI_GELU, aux=u)
    x_out = ops.gemm(g, r.w_proj, bias=r.b_proj, residual=x_mid, ws=ws)
    saved = (x, mean1, rstd1, h1, att_saved, a, lse, x_mid, mean2, rstd2, h2, u, g) if save else None
    return x_out, saved


def block_bwd_packed(dx_out, r, saved, pk, heads):
    if isinstance(saved, NativeSaved):
        if _LNB_SLOT.cur is not None and _LNB_SLOT.cur.enabled and native_blocks() and dx_out.is_contiguous():
            return _block_bwd_native(dx_out, r, saved)
        x, mean1, rstd1, h1, qkv, a, lse, x_mid, mean2, rstd2, h2, u, g = saved.views()
        saved = (x, mean1, rstd1, h1, (qkv, a), a, lse, x_mid, mean2, rstd2, h2, u, g)
    x, mean1, rstd1, h1, att_saved, a, lse, x_mid, mean2, rstd2, h2, u, g = saved
    dw = DwGroup(r)
    dw.add(dx_out, g, r.g_w_proj, r.g_b_proj)
    du = ops.gemm(dx_out, r.w_proj, b_kmajor=True, epilogue=EPI_DGELU, aux=u)
    dw.add(du, h2, r.g_w_fc, r.g_b_fc)
    ws = gemm_workspace(du.device) if du.is_cuda and du.dtype == torch.bfloat16 else None
    dh2 = ops.gemm(du, r.w_fc, b_kmajor=True, ws=ws)
    dx_mid = _ln_bwd(dh2, x_mid, r.ln2_w, mean2, rstd2, r.g_ln2_w, r.g_ln2_b, dres=dx_out)
    dw.add(dx_mid, a, r.g_w_out, r.g_b_out)
    da = ops.gemm(dx_mid, r.w_out, b_kmajor=True, ws=ws)
    if pk.varlen:
        qkv, a_p = att_saved
        bk = _buckets(pk, x, heads)
        if bk is not None:
            dqkv = ops.attn_bucketed_bwd(qkv, a_p, da, lse, pk.cu, bk[0], bk[1], _rows_arg(pk, x, heads), pk.b, pk.L, bk[2], heads, True)
        else:
            dqkv = ops.attn_varlen_bwd(qkv, a_p, da, lse, pk.cu, _rows_arg(pk, x, heads), pk.b, pk.L, heads, True)
    else:
        qkv_d, a_d = att_saved
        # padded queries must carry a ZERO output gradient: a later (padded) query does attend to the valid keys before it
        da_d = torch.zeros(pk.b * pk.L, da.shape[1], device=da.device, dtype=da.dtype)
        ops.scatter_rows_add(da[:pk.rows], pk.pack_idx[:pk.rows], da_d)
        dqkv_d = ops.attn_bwd(qkv_d, a_d, da_d, lse, pk.b, pk.L, heads, True)
        dqkv = ops.gather_rows(dqkv_d, pk.pack_idx[:pk.rows], pk.rows_pad)
    dw.add(dqkv, h1, r.g_w_in, r.g_b_in)
    dh1 = ops.gemm(dqkv, r.w_in, b_kmajor=True, ws=ws)
    dx = _ln_bwd(dh1, x, r.ln1_w, mean1, rstd1, r.g_ln1_w, r.g_ln1_b, dres=dx_mid)
    dw.flush()
    return dx


def packed_key(ids, dtype=torch.bfloat16, heads_dim=64):
    """What a step on these captions depends on besides the values in its input buffers: the padded row count (bf16 towers with
    head dimension 64: the kernels read the valid row count on the device), plus the row count itself otherwise.  The key of
    graph.GraphedStep for steps on packed captions; needs the host-side count tag of the tensor (prefetch.py / set_rows_tag)."""
    tag = getattr(ids, "_dh_rows", None)
    if tag is None or tag[0] != ids._version:
        raise DeclipHipError("packed_key: the caption tensor carries no host-side row count (engine.set_rows_tag)")
    tile = 256 if dtype == torch.bfloat16 else 8
    rows_pad = (tag[1] + tile - 1) // tile * tile
    forced = getattr(ids, "_dh_rows_pad", None)
    if forced is not None and forced[0] == ids._version:
        rows_pad = max(rows_pad, forced[1])              # the job-wide padded row count: the same key on every rank
    return (rows_pad,) if (dtype == torch.bfloat16 and heads_dim == 64) else (rows_pad, tag[1])


def padded_rows(rows, dtype=torch.bfloat16):
    """Padded packed row count of a batch with `rows` caption rows (whole 256-row GEMM tiles in bf16)."""
    tile = 256 if dtype == torch.bfloat16 else 8
    return (int(rows) + tile - 1) // tile * tile


def set_rows_tag(ids, rows, rows_pad=None):
    """Attach the host-side packed row count (tokens up to and including <|endoftext|>, summed over the batch) to a caption
    tensor: the text tower then never reads it back from the device (prefetch.DataPrefetcher does this for the batches it uploads).
    `rows_pad`: the job-wide padded row count (dist.RowsSync) this rank pads up to."""
    ids._dh_rows = (ids._version, int(rows))
    if rows_pad is not None:
        ids._dh_rows_pad = (ids._version, int(rows_pad))
    elif hasattr(ids, "_dh_rows_pad"):
        del ids._dh_rows_pad
    return ids


def packed_captions(ids, dtype):
    """PackedCaptions of a device id tensor (kept on the tensor object: a batch that is used again keeps its bookkeeping)."""
    import os
    tile = 256 if dtype == torch.bfloat16 else 8                        # whole tiles of the persistent GEMM in bf16
    varlen = text_packed_mode() != 2                                    # 2: attention through the dense layout (gathers)
    cached = getattr(ids, "_dh_packed", None)
    if cached is not None and cached[0] == (ids._version, tile, varlen):
        return cached[1]
    pk = PackedCaptions(ids, tile, varlen)
    ids._dh_packed = ((ids._version, tile, varlen), pk)
    return pk


class TextTowerPackedFn(torch.autograd.Function):
    """forward(anchor, ids, tower, want_words=False) -> proj [b, E] fp32 (, words [rows_pad, width] act dtype: ln_final of every
    PACKED row, for the masked-LM head): TextTowerFn on packed captions."""

    @staticmethod
    def forward(ctx, anchor, ids, tower, want_words=False):
        flat = tower._flat()
        dtype = flat.act_dtype
        pk = packed_captions(ids, dtype)
        pk.buckets_default = "1" if getattr(tower, "_dh_attn_buckets", False) else "0"      # (the model's preference: DECLIP sets it on its text tower; see _buckets)
        x = ops.text_embed_packed_fwd(pk.ids_p, pk.pos_idx, tower.token_embedding.weight.data, tower.positional_embedding.data, dtype,
                                      pk.rows_pad, pk.rows_pad)            # (validity of a row: pos_idx >= 0)
        refs = block_refs(flat, tower.transformer.resblocks)
        save = bool(ctx.needs_input_grad[0])
        flat.tower_forward(tower, save)
        saved_blocks = []
        pool = None
        if pooled_last_block(tower.width, tower.heads, ids.shape[1]) and not want_words and refs:
            pool = (pk.eot_rows, pk.row0, pk.nkeys)                     # EOT rows; keys = the caption's rows
        for r in (refs[:-1] if pool is not None else refs):
            x, s = block_fwd_packed(x, r, pk, tower.heads, save)
            saved_blocks.append(s)
        lnw, lnb = tower.ln_final.weight.data, tower.ln_final.bias.data
        if want_words:                                                  # ln_final on every row, then the EOT rows (TextTowerFn's dense branch)
            words, mean_f, rstd_f = ops.layernorm_fwd(x, lnw, lnb, tower.ln_final.eps)
            feat = ops.gather_rows(words, pk.eot_rows)
            pooled = None
        else:
            if pool is not None:
                pooled, s = block_fwd_pooled(x, refs[-1], pool[0], pool[1], pool[2], pk.L, tower.heads, save)
                saved_blocks.append(s)
            else:
                pooled = ops.gather_rows(x, pk.eot_rows)                # the EOT row of every caption (text_transformer.py:203)
            feat, mean_f, rstd_f = ops.layernorm_fwd(pooled, lnw, lnb, tower.ln_final.eps)
            words = None
        ctx.pool = pool
        tp = tower.text_projection
        out = ops.gemm(feat, flat.wview(tp.weight), bias=tp.bias.data, out_dtype=torch.float32)
        ctx.tower, ctx.refs, ctx.saved_blocks, ctx.pk = tower, refs, saved_blocks, pk
        ctx.misc = (x, pooled, mean_f, rstd_f, feat, want_words)
        return (out, words) if want_words else out

    @staticmethod
    def backward(ctx, *grads):
        flat = ctx.tower._flat()
        lnb = LnGradBatch(flat, ctx.tower)
        prev, _LNB_SLOT.cur = _LNB_SLOT.cur, lnb
        try:
            out = TextTowerPackedFn._backward_impl(ctx, lnb, *grads)
            lnb.release()                    # ONE reduce launch for the tower's LayerNorm gradients; the rest of its parameters released
            return out
        finally:
            _LNB_SLOT.cur = prev

    @staticmethod
    def _backward_impl(ctx, lnb, *grads):
        tower, pk = ctx.tower, ctx.pk
        flat = tower._flat()
        flat.begin_backward()
        flat.tower_backward(tower)
        dtype = flat.act_dtype
        x_final, pooled, mean_f, rstd_f, feat, want_words = ctx.misc
        dout = grads[0]
        dwords = grads[1] if want_words else None
        g = flat.gview
        tp = tower.text_projection
        dfeat = None
        if dout is not None:
            dout_a = _to_act(dout, dtype)
            weight_grad(dout_a, feat, g(tp.weight), g(tp.bias))
            dfeat = ops.gemm(dout_a, flat.wview(tp.weight), b_kmajor=True)
        lnw = tower.ln_final.weight.data
        if want_words:
            dw_total = torch.zeros(x_final.shape, device=x_final.device, dtype=dtype)
            if dfeat is not None:
                ops.scatter_rows_add(dfeat, pk.eot_rows, dw_total)
            if dwords is not None:
                dw_total.add_(dwords.reshape(dw_total.shape).to(dtype))
            dx = _ln_bwd(dw_total, x_final, lnw, mean_f, rstd_f, g(tower.ln_final.weight), g(tower.ln_final.bias))
        else:
            dpooled = _ln_bwd(dfeat, pooled, lnw, mean_f, rstd_f, g(tower.ln_final.weight), g(tower.ln_final.bias))
            dx = None
            if ctx.pool is None:
                dx = torch.zeros(x_final.shape, device=dpooled.device, dtype=dtype)
                ops.scatter_rows_add(dpooled, pk.eot_rows, dx)
        lnb.ready([tp.weight, tp.bias, tower.ln_final.weight, tower.ln_final.bias], block=False)
        blocks = list(zip(ctx.refs, ctx.saved_blocks))
        if ctx.pool is not None:
            r, s = blocks.pop()
            dx = block_bwd_pooled(dpooled, r, s, ctx.pool[0], ctx.pool[1], ctx.pool[2], pk.L, tower.heads)
            lnb.ready(r.params)
        for r, s in reversed(blocks):
            dx = block_bwd_packed(dx, r, s, pk, tower.heads)
            lnb.ready(r.params)
        te, pe = tower.token_embedding.weight, tower.positional_embedding
        V = te.shape[0]
        ops.text_embed_packed_bwd(pk.ids_p, pk.cu, dx, g(te) if te.requires_grad else None, g(pe) if pe.requires_grad else None,
                                  pk.rows_pad, pk.L, hot_ids=(V - 2, V - 1))    # <|startoftext|>, <|endoftext|>; the padding rows carry id 0 and an exactly zero gradient row
        ctx.saved_blocks = ctx.misc = ctx.pk = None
        return (torch.zeros_like(flat.anchor), None, None, None)


# ---------------------------------------------------------------------------------------------
# feature normalisation + fused contrastive loss
# ---------------------------------------------------------------------------------------------
class L2NormFn(torch.autograd.Function):
    """clip.py:129-130: x / (||x|| + eps) -> fp32."""

    @staticmethod
    def forward(ctx, x, eps):
        x = x.contiguous()
        y, norm = ops.l2norm_fwd(x, eps)
        ctx.save_for_backward(x, norm)
        ctx.eps = eps
        return y

    @staticmethod
    def backward(ctx, dy):
        x, norm = ctx.saved_tensors
        return ops.l2norm_bwd(x, norm, dy.contiguous().float(), ctx.eps), None


class InfoNCEFn(torch.autograd.Function):
    """Fused multi-pair InfoNCE.  forward(scale[1], label0, n_pairs, Q0, K0, Q1, K1, ...) ->
    row_loss [P,b], correct1 [P,b], correct5 [P,b] (the last two carry no gradient)."""

    @staticmethod
    def forward(ctx, scale, label0, n_pairs, *feats):
        """label0: int, or (label0s, excl0s) per-pair lists (excl0s: self-pair exclusion, NT-Xent)."""
        pairs = [(feats[2 * i].contiguous(), feats[2 * i + 1].contiguous()) for i in range(n_pairs)]
        scale = scale.detach().contiguous().float()
        label0s = excl0s = None
        if isinstance(label0, (tuple, list)):
            label0s, excl0s = label0
            label0 = int(label0s[0])
        row_loss, row_lse, c1, c5, _ = ops.infonce_fwd(pairs, scale, label0, label0s=label0s, excl0s=excl0s)
        ctx.pairs, ctx.scale, ctx.label0, ctx.row_lse = pairs, scale, label0, row_lse
        ctx.lab = (label0s, excl0s)
        ctx.need = [(bool(ctx.needs_input_grad[3 + 2 * i]), bool(ctx.needs_input_grad[4 + 2 * i])) for i in range(n_pairs)]
        ctx.mark_non_differentiable(c1, c5)
        return row_loss, c1, c5

    @staticmethod
    def backward(ctx, g_row, _g1, _g5):
        outs, dscale = ops.infonce_bwd(ctx.pairs, ctx.scale, ctx.label0, ctx.row_lse, g_row.contiguous().float(), need=ctx.need,
                                       label0s=ctx.lab[0], excl0s=ctx.lab[1])
        flat = []
        for dq, dk in outs:
            flat += [dq, dk]
        return (dscale, None, None, *flat)


class LogitsFn(torch.autograd.Function):
    """Materialised logits = scale * Q K^T (clip.py:140-141) for the API surface / parity tests."""

    @staticmethod
    def forward(ctx, scale, Q, K):
        Q, K = Q.contiguous(), K.contiguous()
        raw = ops.gemm(Q, K)                       # fp32 validation-precision kernel
        ctx.save_for_backward(scale, Q, K, raw)
        return raw * scale

    @staticmethod
    def backward(ctx, dl):
        scale, Q, K, raw = ctx.saved_tensors
        dl = dl.contiguous()
        dls = dl * scale
        dQ = ops.gemm(dls, K, b_kmajor=True)
        dK = ops.gemm(dls, Q, a_kmajor=True, b_kmajor=True)
        return (dl * raw).sum().reshape(scale.shape), dQ, dK


class RowCEFn(torch.autograd.Function):
    """F.cross_entropy(reduction='none') on materialised fp32 logits (loss.py:44-45)."""

    @staticmethod
    def forward(ctx, logits, labels):
        logits = logits.contiguous().float()
        row_loss, row_lse, c1, c5 = ops.ce_rows_fwd(logits, labels)
        ctx.save_for_backward(logits, labels, row_lse)
        ctx.mark_non_differentiable(c1, c5)
        return row_loss, c1, c5

    @staticmethod
    def backward(ctx, g_row, _a, _b):
        logits, labels, row_lse = ctx.saved_tensors
        return ops.ce_rows_bwd(logits, labels, row_lse, g_row.contiguous().float()), None


# ---------------------------------------------------------------------------------------------
# FILIP token-wise late interaction (model/filip.py:71-106)
# ---------------------------------------------------------------------------------------------
class GatherTokFn(torch.autograd.Function):
    """rows gather with scatter-add backward (selected top-16 tokens; indices unique)."""

    @staticmethod
    def forward(ctx, x, idx):
        x = x.contiguous()
        ctx.save_for_backward(idx)
        ctx.shape = x.shape
        return ops.gather_rows(x, idx)

    @staticmethod
    def backward(ctx, g):
        (idx,) = ctx.saved_tensors
        dx = torch.zeros(ctx.shape, device=g.device, dtype=g.dtype)
        ops.scatter_rows_add(g.contiguous(), idx, dx)
        return dx, None


class MaxSimFn(torch.autograd.Function):
    """logits[i,l] = scale * mean_j max_m <Q[(i,j)], K[(l,m)]> (filip.py:96-105), without the [b*J, B*16] matrices the reference
    builds (as [b, B, J, 16]; 1.6 + 2.6 GB in fp32 at the FILIP batch b = 256, B = 2048).

    forward, bf16: ONE launch -- the token-similarity GEMM on the persistent MFMA kernel with max-over-m / mean-over-j in its
    epilogue (ops.maxsim_fused_fwd); only raw [b, B] and the arg-max bytes [b*J, B] leave the chip.  fp32 validation mode (and
    shapes the fused kernel does not take): the scores exist for a chunk of captions at a time (<= 256 MB), reduced by
    dh_maxsim_reduce.
    backward: G[(i,j),(l,m)] = scale * dlogits[i,l] / J at m = argmax is regenerated from the arg-max bytes for a chunk of token
    rows at a time into ONE small buffer (<= 128 MB, cache-resident) that feeds the two MFMA GEMMs dQ_chunk = G_c K and
    dK += G_c^T Q_chunk; the full G is never allocated either."""

    S_CHUNK_BYTES = 256 << 20
    G_CHUNK_BYTES = 128 << 20
    G_ROW_QUANTUM = 256          # chunk rows are whole GEMM tiles (tests lower it to exercise several chunks on small problems)

    @staticmethod
    def forward(ctx, scale, Q, K, b, B, J, act_dtype):
        Qa = _to_act(Q, act_dtype)
        Ka = _to_act(K, act_dtype)
        sc = scale.detach().reshape(1).float().contiguous()
        rows, D = b * J, Qa.shape[1]
        if ops.maxsim_fused_ok(Qa, Ka, B, J):
            rows_pad = (rows + 255) // 256 * 256
            if rows_pad != rows:                                           # whole 256-row tiles: zero rows behind the last token
                Qp = torch.zeros(rows_pad, D, device=Qa.device, dtype=Qa.dtype)
                Qp[:rows].copy_(Qa)
                Qa = Qp
            logits, raw, arg = ops.maxsim_fused_fwd(Qa, Ka, b, B, J, sc)
        else:
            Lc = max(1, MaxSimFn.S_CHUNK_BYTES // (rows * 64))               # captions per chunk: rows x Lc x 16 fp32 scores
            Lc = Lc // 16 * 16 if Lc >= 16 else Lc
            if Lc >= B:
                S = ops.gemm(Qa, Ka, out_dtype=torch.float32)              # [b*J, B*16] (small problems: one chunk)
                logits, raw, arg = ops.maxsim_reduce(S, b, B, J, sc)
            else:
                logits = torch.empty(b, B, device=Qa.device, dtype=torch.float32)
                raw = torch.empty(b, B, device=Qa.device, dtype=torch.float32)
                arg = torch.empty(rows, B, device=Qa.device, dtype=torch.uint8)
                for l0 in range(0, B, Lc):
                    l1 = min(B, l0 + Lc)
                    S = ops.gemm(Qa, Ka[l0 * 16:l1 * 16], out_dtype=torch.float32)
                    lg, rw, ag = ops.maxsim_reduce(S, b, l1 - l0, J, sc)
                    logits[:, l0:l1], raw[:, l0:l1], arg[:, l0:l1] = lg, rw, ag
                    del S
        ctx.save_for_backward(Qa, Ka, arg, raw, sc)
        ctx.meta = (b, B, J, act_dtype, scale.shape, Q.dtype, K.dtype, rows)
        return logits

    @staticmethod
    def backward(ctx, dl):
        Qa, Ka, arg, raw, sc = ctx.saved_tensors
        b, B, J, act_dtype, sshape, qdt, kdt, rows = ctx.meta
        dl = dl.contiguous().float()
        D = Qa.shape[1]
        rows_all = Qa.shape[0]                                             # b*J, or padded to whole tiles (fused forward)
        esz = 2 if act_dtype == torch.bfloat16 else 4
        q = MaxSimFn.G_ROW_QUANTUM
        Rc = max(q, MaxSimFn.G_CHUNK_BYTES // (B * 16 * esz) // q * q)
        Rc = min(Rc, rows_all)
        Gbuf = torch.empty(Rc, B * 16, device=dl.device, dtype=act_dtype)
        dQ = torch.empty(rows_all, D, device=dl.device, dtype=torch.float32)
        dK = torch.zeros(B * 16, D, device=dl.device, dtype=torch.float32)
        ws = gemm_workspace(dl.device) if dl.is_cuda and act_dtype == torch.bfloat16 else None
        for r0 in range(0, rows_all, Rc):
            n = min(Rc, rows_all - r0)
            G = ops.maxsim_scatter_rows(dl, arg, sc, b, B, J, r0, n, Gbuf)
            ops.gemm(G, Ka, b_kmajor=True, out=dQ[r0:r0 + n], ws=ws)                                     # [n, D]
            ops.gemm(G, Qa[r0:r0 + n], a_kmajor=True, b_kmajor=True, out=dK, accumulate=True,
                     split_k=_split_k(B * 16, D, n), ws=ws)                                               # [B*16, D] +=
        dscale = (dl * raw).sum().reshape(sshape)
        return dscale, dQ[:rows].to(qdt), dK.to(kdt), None, None, None, None
