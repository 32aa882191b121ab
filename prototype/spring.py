"""Extension hook the reference looks up but does not ship (model/__init__.py:17-19)."""


class PrototypeHelper(object):
    external_model_builder = {}
