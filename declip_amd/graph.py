"""Whole-step HIP graph capture (`torch.cuda.CUDAGraph` over the engine's ctypes launches).

Why: the contrastive step is ~700-1100 kernel launches.  At the ViT-B/32 batch of 512 the host enqueues them in ~11 ms of a
26 ms step, so the GPU never starves; at the ResNet-50 batch of 32 (BASELINE.json configs[0]) the same launch count stands
against ~8 ms of kernel time and the step is bound by the Python / ctypes enqueue (14 us per launch).  Captured once, the
forward + loss + backward is ONE graph launch per step; the fused AdamW stays outside (its bias-correction constants are
launch arguments that change every step) as one more launch.

What makes the step capturable: every engine kernel is launched on torch's current stream with arguments that do not depend on
the data beyond its SHAPE (packed captions: everything depends on the batch through the padded row count alone -- the valid
row count is read on the device -- and GraphedStep keeps one graph per padded row count; labels of the masked-LM head are host tensors);
scratch (split-K workspace, scheduler slots) is created during the eager warm-up steps; the gradient buffer is zeroed by a
captured memset; the two tower streams fork from and join the capture stream with events.  Inputs live in STATIC buffers that
the caller refreshes (copy_) before each replay.  Not capturable (and refused): a distributed step whose collectives run on a
gloo group (host-side collectives), host-side branching on device data.

Host-side bookkeeping of a backward pass does not run on a replay -- `FlatParams._end_backward` (which points every `p.grad` at
its slice of the flat gradient buffer) and the reducer's begin / finish are Python.  The kernels and collectives they enqueue ARE
in the graph (the RCCL all-gather / reduce-scatter / bucket all-reduces are captured like any other launch of the step), so what
has to be redone per replay is only the `p.grad` views that `optimizer.zero_grad()` set to None: pass the models as `modules` and
`after_replay()` restores them, which keeps torch optimizers, gradient clipping and anything else that reads `p.grad` working on
replayed steps (the fused FlatAdamW reads the flat buffer directly and needs nothing).
"""
import collections

import torch


def _flat_stores(modules):
    stores = []
    for m in modules or ():
        st = m.__dict__.get("_flat_store")
        if st is None and hasattr(m, "module"):          # dist.DistModule around an engine model
            st = m.module.__dict__.get("_flat_store")
        if st is None:
            raise ValueError("GraphedStep(modules=...): %s has no flat parameter store" % type(m).__name__)
        stores.append(st)
    return stores


class GraphedStep(object):
    """graphed = GraphedStep(fn, modules=(model,)); loss = graphed()  -- `fn()` runs forward + loss + backward on static input
    buffers and returns a tensor (or tuple of tensors); the first `warmup` calls run eagerly (lazy initialisation, allocator
    warm-up), the next one is captured, every later one replays the graph.  The returned tensors are the graph's static outputs:
    read them before the next call.  `modules`: the engine models (or their DistModule wrappers) that `fn` steps through; their
    `p.grad` views are restored after every replay and a distributed reducer is checked for capturability.

    `key`: a callable returning a hashable value that names the SHAPE of the step about to run -- for a text tower on packed
    captions the padded row count of the batch in the static buffers (`engine.packed_key`): every launch argument and every
    tensor shape of the step depends on the batch through that value alone (the valid row count is read on the device), so one
    captured graph serves every batch with the same key.  Graphs are kept per key (LRU, `max_graphs`); they share one memory
    pool (only one of them runs at a time and a step's tensors are dead when it ends).  The reference's step takes a new batch
    every iteration (solver/clip_solver.py:398-402): this is what lets the captured step be the one training uses."""

    def __init__(self, fn, warmup=2, enabled=True, modules=None, key=None, max_graphs=8, fallback=False, agree=None):
        """`fallback=True`: a capture (or first replay) that raises does not end the run -- the step is executed eagerly from then on
        and `fallback_reason` says why.  `agree`: a callable bool -> bool that makes that decision UNIFORM across the ranks of a
        data-parallel job (an all-reduce MIN of the flag over the process group, outside any capture): one rank replaying RCCL
        collectives from a graph while its peer issues them eagerly is fine, but the ranks must not disagree on whether a
        capture's collectives were ever launched."""
        self.fallback, self.agree, self.fallback_reason = bool(fallback), agree, None
        self.fn, self.warmup, self.enabled = fn, int(warmup), bool(enabled)
        self.calls, self.graph, self.out = 0, None, None
        self.stores = _flat_stores(modules)
        if modules is None and self.enabled:
            import warnings
            warnings.warn("GraphedStep(modules=None): after optimizer.zero_grad() + a replay the parameters' .grad stay None (the fused "
                          "FlatAdamW reads the flat buffer and does not care; torch optimizers and gradient clipping would see no "
                          "gradients) and a distributed reducer is not checked -- pass the engine models", stacklevel=2)
        self.key, self.max_graphs = key, int(max_graphs)
        self.graphs = collections.OrderedDict()         # key -> (graph, static outputs)
        self.pool = None
        self.captures, self.replays = 0, 0

    @staticmethod
    def _capture_mode():
        """Stream-capture error mode.  With a process group alive, ProcessGroupNCCL's watchdog THREAD polls the events of earlier
        collectives (hipEventQuery); under the default `global` mode any such call from any thread while a capture is open is an
        error -- and the watchdog turns it into std::terminate ("operation not permitted when stream is capturing"; seen in one of
        four runs of a one-rank RCCL step on the MI355X, round 5).  `thread_local` confines the restriction to the capturing thread."""
        import torch.distributed as tdist
        return "thread_local" if (tdist.is_available() and tdist.is_initialized()) else "global"

    @staticmethod
    def _abandon_capture(before):
        """A capture that was INVALIDATED (an operation that is illegal under capture ran) makes torch.cuda.graph's exit raise from
        capture_end() -- before it restores the current stream, and with the capture still open on the side stream it used.  End
        that capture (dh_stream_abandon_capture) and go back to the stream the step was on, so that the eager re-run is an eager run."""
        from . import lib as L
        cur = torch.cuda.current_stream()
        try:
            L.load().dh_stream_abandon_capture(cur.cuda_stream)
        except Exception:                       # noqa: BLE001
            pass
        if cur.cuda_stream != before.cuda_stream:
            torch.cuda.set_stream(before)

    def _check_capturable(self):
        """A data-parallel step is captured only when its three collectives run on the LIBRARY's communicator (csrc/comm.hip through
        declip_amd.comm_native: RCCL on a library-owned stream, ordered with two events).  Then no ProcessGroupNCCL work object is
        created inside the capture and the group's internal stream is never pulled into it -- which is what made the watchdog
        thread's event polling fatal in round 5 (hipErrorCapturedEvent, 1 run in ~10; it was papered over with a one-second sleep
        before every capture, removed in round 6).  Collectives on a ProcessGroup (gloo: host-side; nccl without the library
        communicator) are refused: the step runs eagerly."""
        import torch.distributed as tdist
        from . import dist as dh_dist
        for st in self.stores:
            red = getattr(st, "reducer", None)
            if red is not None and red.distributed() and tdist.is_initialized():
                if tdist.get_backend() != "nccl":
                    raise RuntimeError("GraphedStep: the gradient collectives of this step run on a '%s' process group (host-side "
                                       "collectives cannot be captured in a HIP graph); run the step eagerly" % tdist.get_backend())
                if dh_dist.native_comm() is None:
                    raise RuntimeError("GraphedStep: a data-parallel step is captured only with the library communicator "
                                       "(DH_COMM_NATIVE=1, the default under an nccl group); ProcessGroupNCCL collectives are not captured")

    def __call__(self):
        if not self.enabled:
            return self.fn()
        self.calls += 1
        k = self.key() if self.key is not None else None
        hit = self.graphs.get(k)
        if hit is not None:
            self.graphs.move_to_end(k)
            self.graph, self.out = hit
            for st in self.stores:
                st.before_replay()
            self.graph.replay()
            self.replays += 1
            for st in self.stores:
                st.after_replay()
            return self.out
        if self.calls <= self.warmup:
            return self.fn()
        from . import engine
        engine.note_capture()                   # launches recorded from here on hold scratch addresses: outgrown scratch stays alive
        if not self.fallback:
            self._check_capturable()
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            if self.pool is None:
                self.pool = torch.cuda.graph_pool_handle()
            # the capture stream is a fresh side stream (torch.cuda.graph's default): the engine's own side streams fork from it
            with torch.cuda.graph(g, pool=self.pool, capture_error_mode=self._capture_mode()):
                out = self.fn()
            g.replay()                          # the captured launches did not execute during capture
        else:
            # fallback mode (every rank of a data-parallel job): the capture AND the first replay (RCCL errors surface at launch, not at
            # capture) sit inside the try, and the ranks agree on the outcome after each of the two -- a rank whose first graph launch
            # fails must not leave its peers waiting in replayed collectives
            g, out, err = None, None, None

            def _agreed(ok, what):
                if self.agree is None:
                    return ok, None
                try:
                    torch.cuda.synchronize()
                except Exception:               # noqa: BLE001
                    pass
                all_ok = bool(self.agree(ok))
                return all_ok, ("a peer rank could not %s its step" % what if ok and not all_ok else None)

            before = torch.cuda.current_stream()
            try:
                self._check_capturable()
                torch.cuda.synchronize()
                g = torch.cuda.CUDAGraph()
                if self.pool is None:
                    self.pool = torch.cuda.graph_pool_handle()
                with torch.cuda.graph(g, pool=self.pool, capture_error_mode=self._capture_mode()):
                    out = self.fn()
            except Exception as e:              # noqa: BLE001  (whatever the step raises: the eager step is the answer to it)
                err = "capture: %s: %s" % (type(e).__name__, str(e).splitlines()[0][:200] if str(e) else "")
                if torch.cuda.current_stream().cuda_stream != before.cuda_stream or torch.cuda.is_current_stream_capturing():
                    # the capture itself was INVALIDATED (an operation that is illegal under capture ran): torch.cuda.graph's exit raised
                    # from capture_end() before it restored the stream and told the allocator that the capture is over.  There is no
                    # eager step to fall back to in this process (measured: the next synchronisation aborts it) -- end the HIP-side
                    # capture so that teardown is clean, and fail loudly; bench.py re-executes the rank with --graph 0.
                    self._abandon_capture(before)
                    self.enabled, self.fallback_reason = False, "capture invalidated (%s)" % err
                    raise RuntimeError("GraphedStep: the stream capture of the step was invalidated and cannot be recovered from in this "
                                       "process: %s" % err) from e
            ok, peer = _agreed(err is None, "capture")
            err = err or peer
            if ok:
                try:
                    g.replay()                  # the captured launches did not execute during capture
                    torch.cuda.synchronize()
                except Exception as e:          # noqa: BLE001
                    err = "first replay: %s: %s" % (type(e).__name__, str(e).splitlines()[0][:200] if str(e) else "")
                ok, peer = _agreed(err is None, "replay")
                err = err or peer
                if not ok and err.startswith("a peer"):
                    # this rank's replay DID run (its gradients and outputs are those of one step): hand them out, go eager from the next call on
                    self.enabled, self.fallback_reason = False, "first replay failed on a peer rank, step runs eagerly from now on (%s)" % err
                    for st in self.stores:
                        st.after_replay()
                    return out
                if not ok:
                    # THIS rank's first graph launch failed after its peers may have run the step's collectives: an eager re-run here would
                    # enter collectives nobody else is in.  Fail loudly (the launcher ends the job) instead of hanging.
                    self.enabled, self.fallback_reason = False, "first replay failed (%s)" % err
                    raise RuntimeError("GraphedStep: the first launch of the captured data-parallel step failed on this rank: %s" % err)
            if not ok:
                self.enabled, self.fallback_reason = False, "capture failed, step runs eagerly (%s)" % err
                for st in self.stores:          # a backward that died inside the capture never reached its end-of-pass callback
                    st._in_backward = False
                if torch.cuda.is_current_stream_capturing():
                    raise RuntimeError("GraphedStep: the stream is still capturing after a failed capture (%s)" % err)
                return self.fn()
        self.captures += 1
        self.graphs[k] = (g, out)
        while len(self.graphs) > self.max_graphs:
            self.graphs.popitem(last=False)              # least recently used
        self.graph, self.out = g, out
        for st in self.stores:
            st.after_replay()
        return self.out
