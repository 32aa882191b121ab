"""dh_block_fwd / dh_block_bwd (csrc/block.hip: one C-ABI call per ResidualAttentionBlock and direction) on the host emulation
(tests/hipemu; TEST INFRASTRUCTURE): the step through the C-level block calls is BIT-IDENTICAL to the step composed from the per-op
calls in Python (same kernels, same order, same arguments), for the dense image tower and the packed text tower, fp32 and bf16.
Reference arithmetic: prototype/model/image_encoder/base_transformer.py:29-53."""
import pytest
import torch

from hipemu_util import emulated_gpu


def _step(dtype, native, monkeypatch, counts):
    from declip_amd import ops, synth, testing
    from declip_amd.loss import ClipInfoCELoss
    monkeypatch.setenv("DH_BLOCK_NATIVE", "1" if native else "0")
    monkeypatch.setenv("DH_POOLED_LAST", "1")
    monkeypatch.setenv("DH_TEXT_PACKED", "1")
    cfg, b, seed = synth.TINY, 5, 3
    f0, b0 = ops.block_fwd, ops.block_bwd

    def cf(a):
        counts["fwd"] += 1
        return f0(a)

    def cb(a):
        counts["bwd"] += 1
        return b0(a)
    monkeypatch.setattr(ops, "block_fwd", cf)
    monkeypatch.setattr(ops, "block_bwd", cb)
    model = testing.build_clip(cfg, dtype=dtype, seed=seed)
    images = synth.synth_images(b, res=cfg["res"], seed=seed)
    ids = synth.synth_tokens(b, ctx=cfg["ctx"], seed=seed, vocab=cfg["vocab"])
    li, lt = model({"images": images, "captions": ids})
    loss, _ = ClipInfoCELoss()(li, lt)
    loss.backward()
    grads = {n: p.grad.detach().clone() for n, p in model.named_parameters() if p.grad is not None}
    monkeypatch.setattr(ops, "block_fwd", f0)
    monkeypatch.setattr(ops, "block_bwd", b0)
    return float(loss), grads


@pytest.mark.parametrize("dtype", ["fp32", "bf16"])
def test_native_block_calls_are_bit_identical_to_the_per_op_composition(monkeypatch, dtype):
    with emulated_gpu():
        cn, cp = dict(fwd=0, bwd=0), dict(fwd=0, bwd=0)
        ln, gn = _step(dtype, True, monkeypatch, cn)
        lp, gp = _step(dtype, False, monkeypatch, cp)
    # TINY: 2 layers per tower, the last one pooled (Python composition) -> one native block per tower and direction
    assert cn == dict(fwd=2, bwd=2) and cp == dict(fwd=0, bwd=0), (cn, cp)
    assert ln == lp
    assert gn.keys() == gp.keys()
    for n in gn:
        assert torch.equal(gn[n], gp[n]), n


def test_block_entry_points_check_their_arguments():
    """the REAL library, no GPU needed: every check precedes the first launch"""
    import ctypes

    from declip_amd import lib
    L = lib.load()
    a = lib.BlockArgs()
    assert L.dh_block_fwd(ctypes.byref(a), None) == -1 and b"dh_block_fwd" in L.dh_last_error()
    a.dtype, a.rows, a.d, a.heads, a.b, a.L = 1, 100, 64, 1, 3, 50
    assert L.dh_block_fwd(ctypes.byref(a), None) == -1 and b"rows == b * L" in L.dh_last_error()
    a.rows = 150
    assert L.dh_block_fwd(ctypes.byref(a), None) == -1 and b"slab too small" in L.dh_last_error()
    assert L.dh_block_act_bytes(1, 150, 64, 1, 3, 50) >= 150 * 64 * 2 * 15
    assert L.dh_block_bwd_scratch_bytes(1, 150, 64) >= 150 * 64 * 2 * 11
    assert L.dh_block_bwd(ctypes.byref(a), None) == -1
