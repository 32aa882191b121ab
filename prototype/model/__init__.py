"""Model registry (reference: prototype/model/__init__.py:15-21)."""
from declip_amd.model.clip import clip_res50, clip_vitb16, clip_vitb32  # noqa: F401
from declip_amd.model.declip import declip_res50, declip_vitb32  # noqa: F401
from declip_amd.model.defilip import defilip_vitb32  # noqa: F401
from declip_amd.model.filip import filip_res50, filip_vitb32  # noqa: F401
from declip_amd.model.slip import slip_res50, slip_vitb32  # noqa: F401


def model_entry(config):
    if config["type"] not in globals():
        from prototype.spring import PrototypeHelper
        return PrototypeHelper.external_model_builder[config["type"]](**config["kwargs"])
    return globals()[config["type"]](**config["kwargs"])
