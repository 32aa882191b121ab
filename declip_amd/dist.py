"""Data-parallel plumbing over RCCL (torch.distributed backend "nccl" on ROCm) / gloo (CPU tests).

Reference: linklink/__init__.py:13-71 (backend facade), model/clip.py:25-49 (AllGather with
all-reduce backward), utils/dist.py:49-88 (DistModule: per-parameter async all-reduce hooks).

MI355X-first re-design:
  * ONE packed all-gather for all feature tensors of a step ([b, sum D_k] -> [B, sum D_k]);
    its backward is ONE reduce-scatter (W x less traffic than the reference's
    all-reduce + slice, clip.py:43-49) -- mathematically identical.
  * gradients live in one flat fp32 buffer (engine.FlatParams); FlatReducer all-reduces
    contiguous slices of it as soon as the backward of a layer group has finished, on RCCL's
    own stream (async_op=True), overlapped with the rest of the backward.  No per-parameter
    collectives, no packing copies, no host syncs.
"""
import os

import torch
import torch.distributed as tdist


def is_dist():
    """True when collectives have to run.  DH_DIST_FORCE=1 also takes the collective path in a ONE-rank group (test hook:
    exercises the RCCL calls of the step on a single GPU, tests/test_gpu_dist.py)."""
    if not (tdist.is_available() and tdist.is_initialized()):
        return False
    return tdist.get_world_size() > 1 or os.environ.get("DH_DIST_FORCE") == "1"


def get_rank():
    if tdist.is_available() and tdist.is_initialized():
        return tdist.get_rank()
    return int(os.environ.get("RANK", os.environ.get("SLURM_PROCID", 0)))


def get_world_size():
    if tdist.is_available() and tdist.is_initialized():
        return tdist.get_world_size()
    return int(os.environ.get("WORLD_SIZE", os.environ.get("SLURM_NTASKS", 1)))


def get_local_rank():
    ndev = max(1, torch.cuda.device_count() or 1)
    return int(os.environ.get("LOCAL_RANK", get_rank())) % ndev      # ranks beyond the device count share devices (gloo tests)


def initialize(backend="nccl"):
    """linklink.initialize (linklink/__init__.py:42-67) accepting torchrun env (RANK/WORLD_SIZE/LOCAL_RANK/
    MASTER_*) as well as the SLURM variables the reference requires."""
    if tdist.is_initialized():
        return
    rank = int(os.environ.get("RANK", os.environ.get("SLURM_PROCID", 0)))
    world = int(os.environ.get("WORLD_SIZE", os.environ.get("SLURM_NTASKS", 1)))
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "12345")
    os.environ["RANK"], os.environ["WORLD_SIZE"] = str(rank), str(world)
    if world > 1:
        # RCCL kernels hold some CUs while the gradient all-reduce overlaps the backward pass: let the persistent GEMM hand
        # out its tiles dynamically (a static partition loses a whole tile time per occupied CU; gemm_v4.hip / probe_contention)
        os.environ.setdefault("DH_V4_DYNAMIC", "1")
    backend = os.environ.get("DH_DIST_BACKEND", backend)     # e.g. gloo: several ranks sharing one GPU (RCCL refuses that)
    if torch.cuda.is_available():
        torch.cuda.set_device(get_local_rank())
        if "DH_V4_DYNAMIC" in os.environ:       # the library reads the environment once, at its first launch: say it explicitly
            from . import ops
            ops.set_v4_dynamic(int(os.environ["DH_V4_DYNAMIC"]))
    else:
        backend = "gloo"
    kw = {}
    # NO `device_id=` for the process group (round 4): with it ProcessGroupNCCL (torch 2.10 / RCCL 2.26) initialises the communicator
    # eagerly in a mode that costs ~0.35 ms of GPU-visible stall per collective -- measured with a one-rank group on the MI355X: the
    # CLIP step 29.3 ms with device_id against 24.1 ms without (23.7 ms with no process group at all), i.e. a fifth of the step for
    # its ~15 collectives.  The device is bound by torch.cuda.set_device above; barrier() names it explicitly (dist.barrier).
    # DH_PG_DEVICE_ID=1 restores the old call (A/B switch).
    if backend == "nccl" and torch.cuda.is_available() and os.environ.get("DH_PG_DEVICE_ID", "0") == "1":
        kw["device_id"] = torch.device("cuda", torch.cuda.current_device())
    tdist.init_process_group(backend=backend, rank=rank, world_size=world, **kw)
    if backend == "nccl" and torch.cuda.is_available() and os.environ.get("DH_COMM_NATIVE", "1") == "1":
        # the step's three collectives on the library's own communicator context (csrc/comm.hip) instead of ProcessGroupNCCL: the
        # DEFAULT since round 6 -- a step captured with them contains no ProcessGroupNCCL work object and never pulls the group's
        # internal stream into a capture (graph.GraphedStep._check_capturable); measured equal or faster than the process group on a
        # one-rank group (profiles/r05_multi_gpu_default_path.txt).  The process group stays for parameter broadcast, barriers and
        # object exchange.  DH_COMM_NATIVE=0: ProcessGroupNCCL collectives, eager step only.
        from . import comm_native
        comm_native.bootstrap(get_local_rank())


def native_comm():
    """The library-owned communicator context (declip_amd.comm_native; the default under an nccl group, DH_COMM_NATIVE=0 opts out), else None."""
    from . import comm_native
    return comm_native.context()


def step_barrier():
    """Inside the training step nothing needs linklink.barrier (linklink/__init__.py:30-34): every collective of the step is
    stream-ordered.  The engine's own call sites use this no-op."""
    return None


def barrier():
    """linklink.barrier as reference-style callers use it for HOST-side ordering (rank 0 writes a file the other ranks read,
    result dumps of evaluate(), checkpoint hand-off): a real barrier over the process group when one is initialised."""
    if tdist.is_available() and tdist.is_initialized() and tdist.get_world_size() > 1:
        if tdist.get_backend() == "nccl" and torch.cuda.is_available():
            tdist.barrier(device_ids=[torch.cuda.current_device()])      # (the group was created without device_id: name the device here)
        else:
            tdist.barrier()
    return None


def _backend_has_reduce_scatter():
    return tdist.get_backend() != "gloo"


class _AllGatherPacked(torch.autograd.Function):
    """forward: [b, D] -> [W*b, D] (rank-major, == torch.cat(all_gather) of clip.py:34-38);
    backward: reduce-scatter(SUM) of the [W*b, D] gradient (== all-reduce + slice[rank], clip.py:43-49)."""

    @staticmethod
    def forward(ctx, x):
        x = x.contiguous()
        W = tdist.get_world_size()
        out = torch.empty((W * x.shape[0],) + tuple(x.shape[1:]), device=x.device, dtype=x.dtype)
        with _timed("allgather"):
            tdist.all_gather_into_tensor(out, x)
        return out

    @staticmethod
    def backward(ctx, g):
        g = g.contiguous()
        W, r = tdist.get_world_size(), tdist.get_rank()
        b = g.shape[0] // W
        if _backend_has_reduce_scatter():
            out = torch.empty((b,) + tuple(g.shape[1:]), device=g.device, dtype=g.dtype)
            with _timed("reduce_scatter"):
                tdist.reduce_scatter_tensor(out, g, op=tdist.ReduceOp.SUM)
            return out
        g = g.clone()
        tdist.all_reduce(g)
        return g[r * b:(r + 1) * b].clone()


def all_gather_cat(x):
    """[b, ...] -> [B, ...] differentiable gather (CLIP.all_gather, clip.py:113-116)."""
    if not is_dist():
        return x
    comm = native_comm()
    if comm is not None and x.is_cuda:
        from .comm_native import AllGatherPackedNative
        out = AllGatherPackedNative.apply(comm, x)
        comm.wait(x.device)
        return out.reshape((out.shape[0],) + tuple(x.shape[1:]))
    return _AllGatherPacked.apply(x)


def all_gather_cat_many(tensors):
    """Gather several [b, D_k] feature tensors with ONE collective (packed along the feature dim)."""
    return all_gather_cat_many_async(tensors).result()


# ---- optional event timing of the step's collectives (bench.py --gpus N: `per_rank_ms`, `allreduce_exposed_ms`, `allgather_ms`) ------
# TIMING = {} switches it on for EAGER steps (events cannot be timed inside a capture): the all-gather / reduce-scatter of the
# features append (start, end) event pairs recorded on the stream they run on, FlatReducer.finish() appends the pair that brackets
# what is left of the gradient all-reduce once the backward pass has been enqueued (the part of the communication the step could NOT
# hide behind compute), and every bucket's byte count is listed.  None (default): no events, nothing recorded.
TIMING = None


def _timed(kind, stream=None):
    """context manager: event pair around a block on `stream` (default: current), appended to TIMING[kind]"""
    import contextlib

    @contextlib.contextmanager
    def cm():
        if TIMING is None or not torch.cuda.is_available():
            yield
            return
        st = stream or torch.cuda.current_stream()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(st)
        try:
            yield
        finally:
            e1.record(st)
            TIMING.setdefault(kind, []).append((e0, e1))
    return cm()


def timing_summary():
    """ms per kind (sum over the pairs recorded since TIMING was set) + bucket sizes; call after a synchronize"""
    out = {}
    if TIMING is None:
        return out
    for k, v in TIMING.items():
        if k == "bucket_bytes":
            out[k] = list(v)
        else:
            out[k + "_ms"] = sum(a.elapsed_time(b) for a, b in v)
            out[k + "_count"] = len(v)
    return out


# ---- the feature all-gather as an asynchronous step on an engine-owned communication stream --------------------------------
_COMM_STREAMS = {}


def comm_stream(device):
    """The engine's communication stream of `device`: feature all-gathers (and, through autograd, their reduce-scatter backward)
    are enqueued here, so that independent work of the step -- DeCLIP's masked-LM head, its projector / predictor MLPs, the other
    tower -- keeps the compute stream busy while the collective crosses xGMI.  Ordered against the compute streams with events
    only (wait_stream), never a host sync."""
    key = device.index if device.index is not None else torch.cuda.current_device()
    st = _COMM_STREAMS.get(key)
    if st is None:
        st = _COMM_STREAMS[key] = torch.cuda.Stream(device=device)
    return st


class GatherHandle:
    """all_gather_cat_many_async(): the packed all-gather is in flight; result() orders the CALLER's current stream behind it and
    hands out the gathered tensors (rank-major rows).  Not distributed / CPU tensors: the inputs / a synchronous gather."""

    def __init__(self, tensors=None, gathered=None, dims=None, stream=None, native=None):
        self._tensors, self._gathered, self._dims, self._stream, self._native = tensors, gathered, dims, stream, native

    def result(self):
        if self._tensors is not None:
            return self._tensors
        g = self._gathered
        if self._native is not None:
            self._native.wait(g.device)                # the library's communication stream -> this stream (one event)
        elif self._stream is not None:
            cur = torch.cuda.current_stream(g.device)
            cur.wait_stream(self._stream)
            g.record_stream(cur)                       # allocated on the comm stream, consumed on this one
        self._tensors = list(torch.split(g, self._dims, dim=1)) if len(self._dims) > 1 else [g]
        return self._tensors


def all_gather_cat_many_async(tensors):
    """all_gather_cat_many whose collective runs on the communication stream; call .result() where the gathered rows are first
    needed and put the work that does not need them in between."""
    tensors = list(tensors)
    if not is_dist():
        return GatherHandle(tensors=tensors)
    dims = [t[0].numel() for t in tensors]
    comm = native_comm()
    if comm is not None and tensors[0].is_cuda:
        from .comm_native import AllGatherPackedNative
        gathered = AllGatherPackedNative.apply(comm, *[t.reshape(t.shape[0], -1) for t in tensors])
        return GatherHandle(gathered=gathered, dims=dims, native=comm)
    packed = tensors[0].reshape(tensors[0].shape[0], -1) if len(tensors) == 1 else torch.cat([t.reshape(t.shape[0], -1) for t in tensors], dim=1)
    if not packed.is_cuda or os.environ.get("DH_COMM_STREAM", "1") == "0":
        return GatherHandle(gathered=_AllGatherPacked.apply(packed), dims=dims)
    comm = comm_stream(packed.device)
    comm.wait_stream(torch.cuda.current_stream(packed.device))
    with torch.cuda.stream(comm):
        packed.record_stream(comm)
        gathered = _AllGatherPacked.apply(packed)      # autograd replays the backward (reduce-scatter) on this stream as well
    return GatherHandle(gathered=gathered, dims=dims, stream=comm)


class FlatReducer:
    """Bucketed SUM all-reduce of FlatParams.flat_g, launched as backward progresses.

    ready(lo, hi) marks a flat range whose gradients are final; adjacent ranges coalesce and are
    launched (async, on RCCL's stream) once a contiguous run reaches the bucket size.  finish()
    reduces everything that is left (small/odd ranges, parameters handled by torch autograd) as
    maximal contiguous runs and waits for all collectives (stream-ordered: no host sync)."""

    # ranges arrive as whole ALIGN-padded parameter slots (FlatParams.grads_ready): neighbours touch exactly, nothing is bridged

    def __init__(self, flat, bucket_bytes=48 << 20, grad_dtype=None):
        self.flat = flat
        self.bucket_elems = max(1, bucket_bytes // 4)
        # grad_dtype torch.bfloat16 (DistModule(grad_dtype=...) / DH_GRAD_BF16=1): every bucket crosses xGMI as bf16 -- cast, SUM
        # all-reduce, cast back into the fp32 gradient (half the 605 MB per step of CLIP ViT-B/32; the sum over W ranks is
        # rounded to 8 mantissa bits once per hop, so this is opt-in: ~2e-3 relative error on a gradient)
        if grad_dtype is None and os.environ.get("DH_GRAD_BF16", "0") == "1":
            grad_dtype = torch.bfloat16
        self.grad_dtype = grad_dtype
        self.staged = []     # (work, lo, hi, low-precision copy): written back in finish()
        self.done = []       # launched [lo, hi)
        self.runs = []       # coalesced ready-but-not-launched [lo, hi)
        self.works = []
        self.events = []     # (lo, hi, stream id, event): ranges finished on a tower side stream (FlatParams.side_stream)
        self.native = None   # the library's communicator context, once a bucket went through it this step

    def begin(self):
        self.done, self.runs, self.works, self.events, self.staged, self.native = [], [], [], [], [], None

    @staticmethod
    def distributed():
        return is_dist()

    def _launch(self, lo, hi):
        if hi <= lo:
            return
        self.done.append((lo, hi))
        if self.events:
            # the collective is ordered after the CURRENT stream only: wait for pieces of this run produced on another one
            cur = torch.cuda.current_stream(self.flat.flat_g.device)
            for a, b, sid, ev in self.events:
                if sid != cur.cuda_stream and a < hi and b > lo:
                    cur.wait_event(ev)
        if is_dist():
            seg = self.flat.flat_g[lo:hi]
            if TIMING is not None:
                TIMING.setdefault("bucket_bytes", []).append(int(hi - lo) * (2 if self.grad_dtype == torch.bfloat16 else 4))
            comm = native_comm()
            if comm is not None and seg.is_cuda:
                comm.allreduce_bucket(seg, bf16=(self.grad_dtype == torch.bfloat16))
                self.native = comm
            elif self.grad_dtype is not None and self.grad_dtype != seg.dtype:
                low = seg.to(self.grad_dtype)
                self.staged.append((tdist.all_reduce(low, op=tdist.ReduceOp.SUM, async_op=True), lo, hi, low))
            else:
                self.works.append(tdist.all_reduce(seg, op=tdist.ReduceOp.SUM, async_op=True))

    def ready(self, lo, hi):
        if getattr(self.flat, "side_streams", None):
            cur = torch.cuda.current_stream(self.flat.flat_g.device)
            ev = torch.cuda.Event()
            ev.record(cur)
            self.events.append((lo, hi, cur.cuda_stream, ev))
        runs = sorted(self.runs + [(lo, hi)])
        merged = [list(runs[0])]
        for a, b in runs[1:]:
            if a <= merged[-1][1]:
                merged[-1][1] = max(merged[-1][1], b)
            else:
                merged.append([a, b])
        self.runs = []
        for a, b in merged:
            if b - a >= self.bucket_elems:
                self._launch(a, b)
            else:
                self.runs.append((a, b))

    def finish(self):
        # (TIMING: the event pair brackets everything from "backward fully enqueued" to "all gradients reduced" on the compute
        # stream = the part of the all-reduce that compute did not hide: the buckets launched here + the waits)
        with _timed("allreduce_exposed"):
            self.runs = []
            cur = 0
            for lo, hi in sorted(self.done):
                if lo > cur:
                    self._launch(cur, lo)
                cur = max(cur, hi)
            if cur < self.flat.total:
                self._launch(cur, self.flat.total)
            for w in self.works:
                w.wait()
            self.works = []
            for w, lo, hi, low in self.staged:
                w.wait()
                self.flat.flat_g[lo:hi].copy_(low)
            self.staged = []
            if self.native is not None:
                self.native.wait(self.flat.flat_g.device)
                self.native = None


class DistModule(torch.nn.Module):
    """utils/dist.py:49-88 surface: wraps the model, broadcasts parameters from rank 0 and
    arranges gradient averaging (the loss is pre-divided by world size, clip_solver.py:418, so
    SUM == mean).  `sync` is accepted for config compatibility; both modes reduce flat buckets."""

    def __init__(self, module, sync=False, bucket_bytes=48 << 20, grad_dtype=None):
        super().__init__()
        if os.environ.get("DH_BUCKET_MB"):                 # A/B knob: size of the gradient all-reduce buckets
            bucket_bytes = int(float(os.environ["DH_BUCKET_MB"]) * (1 << 20))
        self.module = module
        self.sync = sync
        flat = module.__dict__.get("_flat_store")
        if flat is None:
            raise RuntimeError("DistModule expects a declip_amd engine model (with a flat parameter store)")
        self._flat = flat
        if torch.cuda.is_available() and next(module.parameters()).is_cuda:
            flat.ensure()
            self.broadcast_params()
            flat.reducer = FlatReducer(flat, bucket_bytes, grad_dtype)
        else:
            flat.reducer = FlatReducer(flat, bucket_bytes, grad_dtype)

    def forward(self, *inputs, **kwargs):
        return self.module(*inputs, **kwargs)

    def sync_gradients(self):
        """Reference: device sync / blocking all-reduce.  Here the reducer already waited on its
        collectives inside the autograd callback (stream-ordered, no host sync)."""
        return None

    def broadcast_params(self):
        if not is_dist():
            return
        tdist.broadcast(self._flat.flat_p, 0)          # one flattened broadcast instead of ~300
        self._flat.params_changed()
        for b in self.module.buffers():
            tdist.broadcast(b, 0)


class RowsSync(object):
    """Rank-uniform padded row count of a packed caption batch: MAX over the ranks of each rank's own padded count, as ONE host-side
    all-reduce of one integer on a gloo group of its own -- issued by the input prefetcher's worker thread one batch ahead of the
    step (prefetch.DataPrefetcher(rows_sync=...)), so it never sits on the step's critical path and never touches a GPU stream.
    Every rank pads its packed rows up to that count (engine.PackedCaptions), which makes the shape of the step -- the key of its
    captured graph (engine.packed_key) -- the same on all ranks: they capture and replay in lock-step (VERDICT r5 #4b; the
    reference's ranks run identical eager steps, utils/dist.py:63-88).  Not distributed: the identity."""

    def __init__(self, dtype=torch.bfloat16):
        self.dtype = dtype
        self.group = None
        if is_dist():
            self.group = tdist.new_group(backend="gloo")      # (collective: every rank constructs its RowsSync at the same point)
        self.calls = 0

    def __call__(self, rows):
        from .engine import padded_rows
        mine = padded_rows(rows, self.dtype)
        self.calls += 1
        if self.group is None:
            return mine
        t = torch.tensor([mine], dtype=torch.int64)
        tdist.all_reduce(t, op=tdist.ReduceOp.MAX, group=self.group)
        return int(t[0])

    def close(self):
        """Tear the side group down (collective; after the prefetcher that used it was closed).  A process that exits with the group's
        transport thread still alive can abort in a C++ destructor: destroy it, or the whole process group, before exiting."""
        if self.group is not None:
            tdist.destroy_process_group(self.group)
            self.group = None


def broadcast_object(obj, src=0):
    """utils/dist.py:111-126."""
    if not is_dist():
        return obj
    lst = [obj]
    tdist.broadcast_object_list(lst, src=src)
    return lst[0]
