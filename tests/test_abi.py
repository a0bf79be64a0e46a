"""The C-ABI library loads and exports every symbol include/occ_hip.h declares (no compute calls)."""
import ctypes
import os

import pytest

from occm_amd import _lib


def test_header_parses_and_has_the_core_entry_points():
    protos = _lib.parse_header()
    for name in ("occ_last_error", "occ_version", "occ_arch", "occ_rawboost_fir_bank", "occ_rawboost_center_norm",
                 "occ_compactness_loss", "occ_ce_loss", "occ_adam_multi", "occ_gemm", "occ_layernorm"):
        assert name in protos, name
    assert protos["occ_last_error"][0] is ctypes.c_char_p
    assert len(protos["occ_rawboost_fir_bank"][1]) == 11


def test_library_exports_every_declared_symbol():
    if not os.path.exists(_lib.LIB_PATH):
        import __graft_entry__
        __graft_entry__.build()
    handle = _lib.lib()
    for name in _lib.parse_header():
        assert hasattr(handle, name), "libocc_hip.so does not export %s" % name
    assert handle.occ_version() >= 100
    assert handle.occ_arch() == b"gfx950"


def test_host_argument_validation_needs_no_gpu():
    handle = _lib.lib()
    rc = handle.occ_rawboost_fir_bank(None, 0, None, None, None, 1, 1, 1, 1, 0, None)
    assert rc == -1 and b"null pointer" in handle.occ_last_error()
    with pytest.raises(_lib.OccError):
        _lib.check(rc, "occ_rawboost_fir_bank")
