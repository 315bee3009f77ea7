"""CPU-side checks of the drop-in boundary: the C-ABI library loads and exports
exactly what include/adell_hip.h declares (no compute calls here)."""
import ctypes
import os
import re

import pytest

from adell_mri_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    src = open(os.path.join(ROOT, "include", "adell_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(adell_[a-z0-9_]+)\s*\(", src)))


def test_header_declares_something():
    syms = _declared_symbols()
    assert "adell_conv3d_fwd" in syms and len(syms) >= 10


def test_library_exports_every_declared_symbol():
    if not os.path.exists(_lib.LIB_PATH):
        pytest.fail(f"{_lib.LIB_PATH} missing: run __graft_entry__.build()")
    h = ctypes.CDLL(_lib.LIB_PATH)
    missing = [s for s in _declared_symbols() if not hasattr(h, s)]
    assert not missing, f"declared in adell_hip.h but not exported: {missing}"


def test_binding_table_matches_header():
    assert sorted(_lib.SIGNATURES) == _declared_symbols()


def test_abi_version_and_error_string():
    h = _lib.lib()
    assert h.adell_abi_version() == 1
    assert isinstance(h.adell_last_error(), bytes)


def test_bad_descriptor_is_rejected_without_gpu():
    # argument validation happens before any HIP call
    d = _lib.ConvDesc(1, 8, 8, 8, 4, 0, 4, 3, 3, 3, 1, 1, 1, 1, 1, 1, 7, 8, 8)  # Do wrong
    rc = _lib.lib().adell_conv3d_fwd_ntiles(ctypes.byref(d))
    assert rc == _lib.E_BADARG
    with pytest.raises(_lib.AdellHipError):
        _lib.check(rc)
