"""TEST INFRASTRUCTURE: the parameter order of the reference's optimizer, as a small fixture.

A `torch.optim.AdamW` checkpoint numbers its per-parameter state by position inside `param_groups` (what the reference saves
under 'optimizer', clip_solver.py:655).  The groups come from `param_group_all` (utils/misc.py:267-412): a module-type walk
fills the typed groups, a second walk over named_parameters() the default group, and EVERY typed group is emitted -- also the
empty ones.  This script runs the UNMODIFIED reference function on the UNMODIFIED reference models (CPU, tiny widths: the order
depends on the module tree, not on the sizes) with the shipped pconfig (yfcc15m_vit_clip/config.yaml:34-47) and writes the
parameter NAMES of every group, in order, to tests/golden/param_group_order.json.  tests/test_oracle_golden.py compares the
engine's declip_amd.solver.param_groups against it, so that FlatAdamW.load_state_dict maps a reference checkpoint's moments onto
the right parameters.

    python -m oracle.gen_param_order
"""
import contextlib
import io
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

SHIPPED_PCONFIG = {k: {"weight_decay": 0} for k in ("bn_w", "bn_b", "ln_w", "ln_b", "bias", "logit_scale")}   # yfcc15m_vit_clip/config.yaml:34-47
PCONFIGS = {"shipped": SHIPPED_PCONFIG, "none": {}, "linear_w": dict(SHIPPED_PCONFIG, linear_w={"weight_decay": 0.05})}


def _import_reference_module(ref, ref_harness, name):
    """One more module of the reference package (the harness imports the model and loss packages only): its already-imported
    siblings are put back into sys.modules for the duration of the import, and taken out again (the drop-in `prototype` of this
    repository must stay importable afterwards)."""
    import importlib
    if name in ref.modules:
        return ref.modules[name]
    shadow = {k: sys.modules.pop(k) for k in list(sys.modules) if k.split(".")[0] in ("prototype", "linklink")}
    sys.modules.update(ref.modules)
    sys.path.insert(0, ref_harness.REFERENCE_ROOT)
    try:
        mod = importlib.import_module(name)
    finally:
        sys.path.remove(ref_harness.REFERENCE_ROOT)
        for k in [k for k in sys.modules if k.split(".")[0] in ("prototype", "linklink")]:
            ref.modules.setdefault(k, sys.modules[k])
            del sys.modules[k]
        sys.modules.update(shadow)
    return mod


def reference_orders():
    from declip_amd import synth
    from oracle import gen_golden, ref_harness
    ref = ref_harness.load_reference()
    ref_harness.ensure_gloo_group()
    misc = _import_reference_module(ref, ref_harness, "prototype.utils.misc")
    out = {}
    with contextlib.redirect_stdout(io.StringIO()):
        models = {
            "clip_vit": gen_golden.build_ref_clip(ref, synth.TINY, use_allgather=False),
            "clip_r50": gen_golden.build_ref_clip(ref, synth.R50_TINY, use_allgather=False),
        }
        for mname, model in models.items():
            names = {id(p): n for n, p in model.named_parameters()}
            for pname, pconfig in PCONFIGS.items():
                groups, _ = misc.param_group_all(model, dict(pconfig))
                out["%s/%s" % (mname, pname)] = [[names[id(p)] for p in g["params"]] for g in groups]
    return out


def main():
    out = reference_orders()
    path = os.path.join(ROOT, "tests", "golden", "param_group_order.json")
    with open(path, "w") as fh:
        json.dump(out, fh, indent=0, sort_keys=True)
    for k, groups in out.items():
        print(k, [len(g) for g in groups])
    print("wrote", path)


if __name__ == "__main__":
    main()
