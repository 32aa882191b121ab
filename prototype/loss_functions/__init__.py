"""reference: prototype/loss_functions/__init__.py:1-3."""
from declip_amd.heads import SimsiamLoss  # noqa: F401
from declip_amd.loss import ClipInfoCELoss, NT_Xent, NT_Xent_gather, NTXentLoss  # noqa: F401
