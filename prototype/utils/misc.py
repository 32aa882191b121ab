"""Step-adjacent helpers (reference: prototype/utils/misc.py:22-56,415-428)."""
from declip_amd.loss import accuracy  # noqa: F401
from declip_amd.meters import AverageMeter  # noqa: F401
