"""reference: prototype/loss_functions/__init__.py:1-3."""
from declip_amd.loss import ClipInfoCELoss  # noqa: F401
