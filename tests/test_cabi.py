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
