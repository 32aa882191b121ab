"""linklink.fp16 surface (reference linklink/fp16.py:1-18 is a stub; mixed precision here is the
engine's bf16 mode: fp32 master weights + bf16 mirror, no loss scaling needed)."""


def register_float_module(*a, **k):
    return None


def init():
    return None


class FP16_Optimizer(object):
    def __init__(self, *a, **k):
        raise NotImplementedError("use engine dtype bf16 (declip_amd) instead of FP16_Optimizer")
