"""Whole-step HIP graph capture (`torch.cuda.CUDAGraph` over the engine's ctypes launches).

Why: the contrastive step is ~700-1100 kernel launches.  At the ViT-B/32 batch of 512 the host enqueues them in ~11 ms of a
26 ms step, so the GPU never starves; at the ResNet-50 batch of 32 (BASELINE.json configs[0]) the same launch count stands
against ~8 ms of kernel time and the step is bound by the Python / ctypes enqueue (14 us per launch).  Captured once, the
forward + loss + backward is ONE graph launch per step; the fused AdamW stays outside (its bias-correction constants are
launch arguments that change every step) as one more launch.

What makes the step capturable: every engine kernel is launched on torch's current stream with arguments that do not depend on
the data (row counts of packed captions are taken from the host copy of a batch, labels of the masked-LM head are host tensors);
scratch (split-K workspace, scheduler slots) is created during the eager warm-up steps; the gradient buffer is zeroed by a
captured memset; the two tower streams fork from and join the capture stream with events.  Inputs live in STATIC buffers that
the caller refreshes (copy_) before each replay.  Not capturable (and refused): a distributed step whose collectives run on a
gloo group, host-side branching on device data.
"""
import torch


class GraphedStep(object):
    """graphed = GraphedStep(fn); loss = graphed()  -- `fn()` runs forward + loss + backward on static input buffers and returns a
    tensor (or tuple of tensors); the first `warmup` calls run eagerly (lazy initialisation, allocator warm-up), the next one is
    captured, every later one replays the graph.  The returned tensors are the graph's static outputs: read them before the
    next call."""

    def __init__(self, fn, warmup=2, enabled=True):
        self.fn, self.warmup, self.enabled = fn, int(warmup), bool(enabled)
        self.calls, self.graph, self.out = 0, None, None

    def __call__(self):
        if not self.enabled:
            return self.fn()
        self.calls += 1
        if self.graph is not None:
            self.graph.replay()
            return self.out
        if self.calls <= self.warmup:
            return self.fn()
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        # the capture stream is a fresh side stream (torch.cuda.graph's default): the engine's own side streams fork from it
        with torch.cuda.graph(g):
            out = self.fn()
        self.graph, self.out = g, out
        g.replay()                              # the captured launches did not execute during capture
        return self.out
