"""AverageMeter (reference utils/misc.py:22-56).  reduce_update all-reduces like the reference but
keeps the value on the device until someone reads `.val/.avg` (no per-meter host sync in the step);
reduce_update_packed reduces ALL logged scalars of a step with ONE collective (the reference issues one tiny
all-reduce per meter, misc.py:38-40: 3-10 latency-bound collectives per step next to the gradient buckets)."""
import numpy as np
import torch
import torch.distributed as tdist


class AverageMeter(object):
    def __init__(self, length=0):
        self.length = length
        self.reset()

    def reset(self):
        self.history = []
        self.count = 0
        self._sum = 0.0
        self._pending = []
        self._val = 0.0

    def reduce_update(self, tensor, num=1):
        if tdist.is_available() and tdist.is_initialized() and tdist.get_world_size() > 1:
            tdist.all_reduce(tensor)
        self._pending.append((tensor.detach(), num))

    def update(self, val, num=1):
        self._pending.append((val, num))

    def _flush(self):
        for val, num in self._pending:
            v = float(val.item()) if (torch.is_tensor(val) or isinstance(val, _Slot)) else float(val)
            if self.length > 0:
                self.history.append(v)
                if len(self.history) > self.length:
                    del self.history[0]
            else:
                self._sum += v * num
                self.count += num
            self._val = v
        self._pending = []

    @property
    def val(self):
        self._flush()
        return self._val

    @property
    def avg(self):
        self._flush()
        if self.length > 0:
            return float(np.mean(self.history)) if self.history else 0.0
        return self._sum / max(1, self.count)


def reduce_update_packed(items):
    """items: [(meter, scalar tensor, num)].  The scalars are packed into one [k] tensor, SUM all-reduced ONCE (asynchronously, on
    the collective's own stream: the step does not wait for it), and every meter keeps a view of its slot until `.val/.avg` is read.
    Same values as k calls of AverageMeter.reduce_update (misc.py:38-40)."""
    items = [(m, t, n) for m, t, n in items if t is not None]
    if not items:
        return None
    packed = torch.stack([t.detach().reshape(()).float() for _, t, _ in items])
    work = None
    if tdist.is_available() and tdist.is_initialized() and tdist.get_world_size() > 1:
        work = tdist.all_reduce(packed, async_op=True)
    for i, (m, _, n) in enumerate(items):
        m._pending.append((_Slot(packed, i, work), n))
    return packed


class _Slot(object):
    """One scalar of a packed reduction; read (and waited for) only when the meter is flushed."""

    def __init__(self, packed, i, work):
        self.packed, self.i, self.work = packed, i, work

    def item(self):
        if self.work is not None:
            self.work.wait()
            self.work = None
        return self.packed[self.i].item()
