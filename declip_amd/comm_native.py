"""The step's collectives through the library's own communicator context (include/declip_hip.h: dh_init, dh_allgather_packed,
dh_reducescatter_packed, dh_allreduce_bucket -- RCCL on a library-owned communication stream, csrc/comm.hip).

The default under an nccl process group since round 6 (`DH_COMM_NATIVE=0` opts out; read by declip_amd.dist.initialize): the alternative
runs the same three collectives through torch.distributed's ProcessGroupNCCL (= RCCL) and is never captured into a step graph.  What this path changes: the feature tensors are packed by one kernel
straight into the own slot of the gathered buffer (no torch.cat, in-place all-gather), the gradient split after the
reduce-scatter is one kernel, the bf16 bucket casts run on the communication stream, and the ordering against the compute
streams is the context's two events instead of ProcessGroupNCCL's bookkeeping.  The process group is still used to hand the
ncclUniqueId from rank 0 to the others and for the host-side barrier / object broadcast of the solver.

Reference: linklink/__init__.py:13-71, model/clip.py:25-49 (AllGather), utils/dist.py:49-88 (DistModule).
"""
import ctypes

import torch
import torch.distributed as tdist

from . import lib as L

_CTX = None


def _ptr(t):
    return ctypes.c_void_p(t.data_ptr())


def _cur(device):
    return ctypes.c_void_p(torch.cuda.current_stream(device).cuda_stream)


class NativeComm(object):
    """One per process.  Every method only enqueues; `wait()` orders the current torch stream behind the collectives."""

    def __init__(self, rank, world, local_rank, unique_id):
        self.lib = L.load()
        buf = ctypes.create_string_buffer(bytes(unique_id), 128)
        self.ctx = self.lib.dh_init(int(rank), int(world), int(local_rank), buf)
        if not self.ctx:
            msg = self.lib.dh_last_error()
            raise L.DeclipHipError("dh_init failed: %s" % (msg.decode() if msg else "?"))
        self.rank, self.world, self.device = int(rank), int(world), torch.device("cuda", int(local_rank))
        # torch's caching allocator has to know that blocks handed to the collectives are in use on the library's stream
        self.stream = torch.cuda.ExternalStream(self.lib.dh_comm_stream(self.ctx), device=self.device)

    @staticmethod
    def unique_id():
        lib = L.load()
        buf = ctypes.create_string_buffer(128)
        L.check(lib.dh_comm_unique_id(buf, 128), "dh_comm_unique_id")
        return buf.raw

    def close(self):
        if self.ctx:
            L.check(self.lib.dh_finalize(self.ctx), "dh_finalize")
            self.ctx = None

    def wait(self, device=None):
        L.check(self.lib.dh_comm_wait(self.ctx, _cur(device or self.device)), "dh_comm_wait")

    # ---- feature gather -------------------------------------------------------------------------------------------------
    def _table(self, tensors):
        n = len(tensors)
        if not (1 <= n <= 8):
            raise L.DeclipHipError("packed gather of %d tensors (1..8)" % n)
        t0 = tensors[0]
        for t in tensors:
            if t.dtype != t0.dtype or t.shape[0] != t0.shape[0] or not t.is_cuda or not t.is_contiguous():
                raise L.DeclipHipError("packed gather: tensors must be contiguous CUDA tensors of one dtype and row count")
        if t0.dtype not in (torch.float32, torch.bfloat16):
            raise L.DeclipHipError("packed gather: dtype %s" % t0.dtype)
        cols = [t[0].numel() for t in tensors]
        return (ctypes.c_void_p * n)(*[t.data_ptr() for t in tensors]), (ctypes.c_int * n)(*cols), cols, t0.element_size()

    def all_gather_packed(self, tensors):
        """n tensors [rows, ...] -> gathered [world * rows, sum cols] (rank-major rows; in flight on the communication stream)."""
        ptrs, ccols, cols, es = self._table(tensors)
        rows = tensors[0].shape[0]
        out = torch.empty((self.world * rows, sum(cols)), device=tensors[0].device, dtype=tensors[0].dtype)
        from . import dist as dh_dist
        with dh_dist._timed("allgather", stream=self.stream):      # (bench.py's attribution fields: event pair on the communication stream, eager steps only)
            L.check(self.lib.dh_allgather_packed(self.ctx, ptrs, ccols, len(tensors), rows, es, _ptr(out), _cur(out.device)),
                    "dh_allgather_packed")
        for t in tensors:
            t.record_stream(self.stream)
        out.record_stream(self.stream)
        return out, cols

    def reduce_scatter_packed(self, g, shapes, cols):
        """g [world * rows, sum cols] -> the own rows of the rank-summed gradient, split into tensors of `shapes`."""
        g = g.contiguous()
        rows = g.shape[0] // self.world
        outs = [torch.empty(s, device=g.device, dtype=g.dtype) for s in shapes]
        scratch = torch.empty((rows, g.shape[1]), device=g.device, dtype=g.dtype)
        n = len(outs)
        ptrs, ccols = (ctypes.c_void_p * n)(*[o.data_ptr() for o in outs]), (ctypes.c_int * n)(*cols)
        from . import dist as dh_dist
        with dh_dist._timed("reduce_scatter", stream=self.stream):
            L.check(self.lib.dh_reducescatter_packed(self.ctx, _ptr(g), ptrs, ccols, n, rows, g.element_size(), _ptr(scratch),
                                                     _cur(g.device)), "dh_reducescatter_packed")
        for t in outs + [g, scratch]:
            t.record_stream(self.stream)
        return outs

    # ---- gradient buckets -----------------------------------------------------------------------------------------------
    def allreduce_bucket(self, seg, bf16=False):
        """SUM all-reduce of a contiguous fp32 slice of the flat gradient buffer, in place, in flight after the call."""
        if seg.dtype != torch.float32 or not seg.is_contiguous():
            raise L.DeclipHipError("gradient bucket must be a contiguous fp32 slice")
        stage = torch.empty(seg.numel(), device=seg.device, dtype=torch.bfloat16) if bf16 else None
        L.check(self.lib.dh_allreduce_bucket(self.ctx, _ptr(seg), seg.numel(), _ptr(stage) if bf16 else None, _cur(seg.device)),
                "dh_allreduce_bucket")
        if stage is not None:
            stage.record_stream(self.stream)


class AllGatherPackedNative(torch.autograd.Function):
    """forward: n x [b, ...] -> packed [W*b, sum cols]; backward: reduce-scatter(SUM) + split (== clip.py:25-49's gradient)."""

    @staticmethod
    def forward(ctx, comm, *tensors):
        tensors = [t.contiguous() for t in tensors]
        out, cols = comm.all_gather_packed(tensors)
        ctx.comm, ctx.cols, ctx.shapes = comm, cols, [tuple(t.shape) for t in tensors]
        return out

    @staticmethod
    def backward(ctx, g):
        outs = ctx.comm.reduce_scatter_packed(g, ctx.shapes, ctx.cols)
        ctx.comm.wait(g.device)                      # the tower backward needs these gradients now
        return (None,) + tuple(outs)


def context():
    return _CTX


def bootstrap(local_rank):
    """Create the process-wide context: rank 0's ncclUniqueId travels over the (already initialised) torch process group."""
    global _CTX
    if _CTX is not None:
        return _CTX
    rank, world = tdist.get_rank(), tdist.get_world_size()
    box = [NativeComm.unique_id() if rank == 0 else None]
    if world > 1:
        tdist.broadcast_object_list(box, src=0)
    _CTX = NativeComm(rank, world, local_rank, box[0])
    return _CTX


def shutdown():
    global _CTX
    if _CTX is not None:
        _CTX.close()
        _CTX = None
