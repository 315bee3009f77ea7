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


def test_library_exports_nothing_but_the_declared_symbols():
    """-fvisibility=hidden + csrc/exports.map: the dynamic symbol table of the library is the header,
    no internal helper, no kernel stub (nm -D minus header = empty)."""
    import subprocess

    out = subprocess.run(["nm", "-D", "--defined-only", _lib.LIB_PATH], capture_output=True, text=True,
                         check=True).stdout
    exported = sorted({ln.split()[-1] for ln in out.splitlines() if ln.strip()})
    assert exported == _declared_symbols(), sorted(set(exported) ^ set(_declared_symbols()))


def test_binding_table_matches_header():
    assert sorted(_lib.SIGNATURES) == _declared_symbols()


def test_abi_version_and_error_string():
    h = _lib.lib()
    assert h.adell_abi_version() == 2
    assert isinstance(h.adell_last_error(), bytes)


def test_bad_descriptor_is_rejected_without_gpu():
    # argument validation happens before any HIP call
    d = _lib.ConvDesc(1, 8, 8, 8, 4, 0, 4, 3, 3, 3, 1, 1, 1, 1, 1, 1, 7, 8, 8)  # Do wrong
    rc = _lib.lib().adell_conv3d_fwd_ntiles(ctypes.byref(d))
    assert rc == _lib.E_BADARG
    with pytest.raises(_lib.AdellHipError):
        _lib.check(rc)


def _desc(N, size, C0, C1, Cout, k=3, s=1, p=1):
    o = (size + 2 * p - k) // s + 1
    return _lib.ConvDesc(N, size, size, size, C0, C1, Cout, k, k, k, s, s, s, p, p, p, o, o, o)


def test_partial_sum_buffers_carry_their_row_count():
    """ABI version 2: every entry point that writes per-block partial sums takes the rows per item
    the caller sized the buffer for and refuses -- before any HIP call, so this runs without a GPU
    -- when the launch plan of the call writes a different number (the row count is the buffer's
    stride too: more rows are as wrong as fewer). Round 3's fault was a row count cached across an
    adell_set_tuning flip."""
    h = _lib.lib()
    buf = (ctypes.c_float * 64)()            # never dereferenced: the calls fail before a launch
    p = ctypes.cast(buf, ctypes.c_void_p)
    d = _desc(1, 32, 32, 0, 32)
    nt = h.adell_conv3d_fwd_ntiles_f16x3_ws(ctypes.byref(d))
    assert nt > 0
    for rows in (nt - 1, nt + 1, 0):
        rc = h.adell_conv3d_fwd_f16x3_ws(ctypes.byref(d), p, None, p, p, None, None, p, p, rows,
                                         None, None, 0, None)
        assert rc == _lib.E_BADARG, rows
        assert b"rows" in h.adell_last_error()
        rc = h.adell_conv3d_fwd_f16x3(ctypes.byref(d), p, None, p, p, None, None, p, p, rows, None,
                                      None)
        assert rc == _lib.E_BADARG
    nt32 = h.adell_conv3d_fwd_ntiles(ctypes.byref(d))
    assert h.adell_conv3d_fwd(ctypes.byref(d), p, None, p, None, None, p, p, nt32 + 1,
                              None) == _lib.E_BADARG
    # the fused norm / dropout / activation backward epilogue
    na = h.adell_conv3d_bwd_data_f16x3_adn_ntiles(ctypes.byref(d))
    assert na > 0
    site = _lib.AdnSite(p, p, p, None, 0.0, 0.0, 1)
    rc = h.adell_conv3d_bwd_data_f16x3_adn(ctypes.byref(d), p, p, p, None, p, None, None,
                                           ctypes.byref(site), None, p, na * 2, None)
    assert rc == _lib.E_BADARG
    # stride-2 fused forward, narrow-input forwards
    d2 = _desc(1, 32, 32, 0, 32, s=2)
    n2 = h.adell_conv3d_fwd_s2_fused_ntiles(ctypes.byref(d2))
    assert n2 > 0
    assert h.adell_conv3d_fwd_s2_fused(ctypes.byref(d2), p, p, p, None, p, p, n2 - 1, None,
                                       None) == _lib.E_BADARG
    dc = _desc(1, 32, 2, 0, 32)
    nc = h.adell_conv_cinfold_ntiles(ctypes.byref(dc))
    assert nc > 0
    assert h.adell_conv_cinfold_fwd(ctypes.byref(dc), p, p, None, p, p, nc + 3, None) == _lib.E_BADARG
    assert h.adell_conv_cinfold_fwd_f16x3(ctypes.byref(dc), p, p, None, p, p, nc + 3,
                                          None) == _lib.E_BADARG
    ds = _desc(1, 32, 2, 0, 2)
    ns = h.adell_conv_cin_small_ntiles(ctypes.byref(ds))
    assert ns > 0
    assert h.adell_conv_cin_small_fwd(ctypes.byref(ds), p, p, None, p, p, ns + 1, None) == _lib.E_BADARG


def test_set_tuning_bumps_the_plan_epoch_and_changes_row_counts():
    h = _lib.lib()
    d = _desc(2, 64, 32, 0, 32)
    e0 = h.adell_plan_epoch()
    rows = h.adell_conv3d_fwd_ntiles_f16x3_ws(ctypes.byref(d))
    with _lib.tuning(igemm_no8=1):
        assert h.adell_plan_epoch() > e0
        other = h.adell_conv3d_fwd_ntiles_f16x3_ws(ctypes.byref(d))
        assert other != rows         # 8x8x4 bricks instead of 8x8x8: twice the rows
        buf = (ctypes.c_float * 64)()
        p = ctypes.cast(buf, ctypes.c_void_p)
        # a buffer sized under the old plan is refused under the new one (no launch, no GPU needed)
        rc = h.adell_conv3d_fwd_f16x3_ws(ctypes.byref(d), p, None, p, p, None, None, p, p, rows,
                                         None, None, 0, None)
        assert rc == _lib.E_BADARG
    e1 = h.adell_plan_epoch()
    assert e1 > e0
    with _lib.tuning(igemm_no8=0):       # no change of value: no new epoch
        assert h.adell_plan_epoch() == e1
