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


def test_descriptor_structs_match_their_ctypes_mirrors():
    """Every descriptor struct that crosses the ABI by address has a ctypes mirror in clsr_amd/ops.py: same size as the C
    side reports (a field added on one side only would shift every later field silently)."""
    import ctypes

    from clsr_amd import ops

    lib = _lib.load()
    pairs = [("clsr_sizeof_heads_desc", ops.HeadsDesc), ("clsr_sizeof_segsum_desc", ops.SegsumDesc),
             ("clsr_sizeof_dwjob", ops.DwJob), ("clsr_sizeof_pack_desc", getattr(ops, "PackDesc", None))]
    checked = 0
    for fn, mirror in pairs:
        if mirror is None or not hasattr(lib, fn):
            continue
        assert ctypes.sizeof(mirror) == getattr(lib, fn)(), fn
        checked += 1
    assert checked >= 3
    # host-only queries of the round-4 entry points
    assert lib.clsr_heads_fused_workspace_bytes() > lib.clsr_heads_fused_counter_bytes() > 0
    assert lib.clsr_heads_comm_buffer_bytes() > 0 and lib.clsr_heads_comm_max_world() == 8
    assert lib.clsr_pgemm_dw_wide_supported(204800, 128, 1536) == 1 and lib.clsr_pgemm_dw_wide_supported(204800, 40, 360) == 0
    assert lib.clsr_pgemm_dw_wide_workspace_floats(204800, 128, 1536) >= 128 * 1536
    assert lib.clsr_comm_buffer_bytes() > 0 and lib.clsr_comm_max_doubles() == 256
