"""The C-ABI library loads (CPU container, no GPU needed) and exports every symbol include/declip_hip.h declares."""
import ctypes
import os
import re

from conftest import ROOT


def _header_functions():
    src = open(os.path.join(ROOT, "include", "declip_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(dh_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    from declip_amd import build, lib
    if not os.path.exists(lib.LIB_PATH):
        build.build(verbose=False)
    names = _header_functions()
    assert len(names) >= 25
    so = ctypes.CDLL(lib.LIB_PATH)
    for n in names:
        assert hasattr(so, n), "missing export " + n
    assert set(names) == set(lib.exported_symbols()), set(names) ^ set(lib.exported_symbols())
    L = lib.load()
    assert L.dh_version() >= 100


def test_argument_errors_are_reported_not_thrown():
    from declip_amd import lib
    L = lib.load()
    rc = L.dh_layernorm_fwd(0, None, None, None, None, None, None, 0, 0, 1e-5, None)
    assert rc == -1 and b"dh_layernorm_fwd" in L.dh_last_error()
    args = lib.GemmArgs()
    assert L.dh_gemm(ctypes.byref(args), None) == -1


def test_resnet_entry_points_check_their_arguments():
    """the REAL library (no GPU needed: every check precedes the first launch): error codes + messages, never an exception"""
    from declip_amd import lib
    L = lib.load()
    one = ctypes.c_void_p(16)          # any non-NULL pointer: the calls below must fail before touching memory
    assert L.dh_conv_rows(1, one, 0, 64, 0, one, 1, 8, 8, 64, 5, 1, 2, 25 * 64, None) == -1 and b"3x3" in L.dh_last_error()
    assert L.dh_conv_rows(1, one, 0, 12, 0, one, 1, 8, 8, 12, 3, 1, 1, 108, None) == -1 and b"C % 8" in L.dh_last_error()
    assert L.dh_conv_rows(1, one, 1, 3, 2, one, 1, 8, 8, 3, 3, 2, 1, 32, None) == -1 and b"channel window" in L.dh_last_error()
    assert L.dh_bn2d_fwd(1, one, None, one, one, one, one, one, None, None, 10, 12, 1e-5, 0.1, 1, 1, one, 1 << 20, None) == -1
    assert b"multiple of 8" in L.dh_last_error()
    assert L.dh_bn2d_fwd(1, one, None, one, one, one, one, one, None, None, 10, 16, 1e-5, 0.1, 1, 1, one, 8, None) == -1
    assert b"workspace too small" in L.dh_last_error()
    assert L.dh_bn2d_fwd(1, one, None, one, one, one, one, one, None, None, 10, 16, 1e-5, 0.1, 1, 0, one, 1 << 20, None) == -1
    assert b"eval needs running stats" in L.dh_last_error()
    assert L.dh_bn2d_bwd(1, one, one, None, one, one, one, one, None, one, one, 10, 16, 1, one, 1 << 20, None) == -1      # relu without y
    assert L.dh_bn2d_sums(1, 1, one, None, None, None, None, 0, 10, 16, one, one, 1 << 20, None) == -1 and b"mode 1" in L.dh_last_error()
    assert L.dh_avgpool_fwd(1, one, one, 1, 7, 7, 16, 2, None) == -1 and b"multiples of k" in L.dh_last_error()
    assert L.dh_attnpool_tokens_fwd(1, one, one, one, 0, 49, 64, None) == -1
    assert L.dh_image_resized_crop_u8(one, 1, 8, 8, None, None, (ctypes.c_float * 3)(), (ctypes.c_float * 3)(1, 1, 1), one, 3, 0, 4, 4, 1, None) == -1
    assert L.dh_bn2d_ws_bytes(100352, 64) == 4 * (393 * 2 * 64 + 2 * 64) or L.dh_bn2d_ws_bytes(100352, 64) > 0


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    from declip_amd import lib
    monkeypatch.setattr(lib, "_lib", None)
    monkeypatch.setattr(lib, "LIB_PATH", str(tmp_path / "nope.so"))
    import pytest
    with pytest.raises(lib.DeclipHipError):
        lib.load()


def test_cpu_tensors_are_rejected():
    import pytest
    import torch
    from declip_amd import lib, ops
    with pytest.raises(lib.DeclipHipError):
        ops.layernorm_fwd(torch.zeros(4, 8), torch.ones(8), torch.zeros(8))


def test_communicator_context_fails_loudly_without_a_device():
    """The comm entry points (csrc/comm.hip) bind RCCL at run time and never fall back: on a box without a GPU dh_init returns
    NULL with the reason in dh_last_error(), bad arguments are refused, and nothing crashes.  (dh_comm_unique_id needs no device:
    it proves that RCCL was found and bound.)"""
    import ctypes
    from declip_amd import lib as L
    lib = L.load()
    buf = ctypes.create_string_buffer(128)
    assert lib.dh_comm_unique_id(buf, 16) == -1 and b"128" in lib.dh_last_error()
    rc = lib.dh_comm_unique_id(buf, 128)
    if rc != 0:                                   # no RCCL on this machine: that, too, must be a clean error
        assert b"RCCL" in lib.dh_last_error()
        return
    assert buf.raw != b"\x00" * 128
    assert lib.dh_init(3, 2, 0, buf) is None and b"rank 3 of 2" in lib.dh_last_error()
    import torch
    if not torch.cuda.is_available():
        assert lib.dh_init(0, 1, 0, buf) is None and b"visible devices" in lib.dh_last_error()
    assert lib.dh_allgather_packed(None, None, None, 1, 1, 2, None, None) == -1
    assert lib.dh_reducescatter_packed(None, None, None, None, 1, 1, 2, None, None) == -1
    assert lib.dh_allreduce_bucket(None, None, 0, None, None) == -1
    assert lib.dh_comm_wait(None, None) == -1 and lib.dh_finalize(None) == -1
