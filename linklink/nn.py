"""linklink.nn surface (reference linklink/nn.py:5-10)."""
import torch

SyncBatchNorm2d = torch.nn.BatchNorm1d  # the reference shim aliases exactly this


class syncbnVarMode_t(object):
    L1 = None
    L2 = None
