"""AverageMeter (reference utils/misc.py:22-56).  reduce_update all-reduces like the reference but
keeps the value on the device until someone reads `.val/.avg` (no per-meter host sync in the step)."""
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
            v = float(val.item()) if torch.is_tensor(val) else float(val)
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
