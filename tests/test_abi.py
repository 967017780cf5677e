"""CPU: the C-ABI library builds, loads, and exports every symbol include/artiboost_hip.h declares."""
import ctypes
import os


def test_library_exports_every_declared_symbol():
    from artiboost_amd import _lib as L
    from artiboost_amd import build
    build.build()
    names = L.declared_symbols()
    assert "ab_softargmax3d_fwd" in names and len(names) >= 4
    lib = ctypes.CDLL(L.LIB_PATH)
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/artiboost_hip.h but not exported"
    assert L.lib().ab_abi_version() >= 1


def test_ops_fail_loudly_without_device_tensors():
    import pytest
    import torch
    from artiboost_amd.head import softargmax3d
    with pytest.raises(RuntimeError):
        softargmax3d(torch.zeros(1, 2, 2, 6), 2, 3)
