"""CPU test: the C-ABI shared library builds for gfx950, loads, and exports every symbol that
include/clsr_hip.h declares (no kernel is launched here)."""
import os

from clsr_amd import _lib


def test_library_exports_every_declared_symbol():
    if not os.path.exists(_lib.LIB_PATH):
        from clsr_amd.build import build

        build(verbose=False)
    protos = _lib.parse_header()
    assert len(protos) >= 40
    lib = _lib.load()
    for name in protos:
        assert hasattr(lib, name), name
    assert lib.clsr_version() >= 100
    # host-only queries work without a GPU
    assert lib.clsr_pgemm_dw_workspace_floats(1024, 80, 80) > 0
    assert lib.clsr_pgemm_stats_parts(100000) > 0


def test_invalid_arguments_are_reported_not_thrown():
    lib = _lib.load()
    rc = lib.clsr_pgemm(None, 0, 0, 0, None, 0, None, None, 0, None, 0, None, None, 0, None, 0, None, 0, 0,
                        None, 16, 16, 16, None)
    assert rc == -1
    assert "invalid argument" in _lib.last_error()
