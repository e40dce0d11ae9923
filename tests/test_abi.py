"""The C ABI library: loads, exports every symbol include/hsw.h declares, and
its host-side arithmetic agrees with the oracle.  No compute calls (no GPU)."""
import ctypes as C
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_functions():
    src = open(os.path.join(ROOT, "include", "hsw.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    names = re.findall(r"\b(hsw_[a-z0-9_]+)\s*\(", src)
    return sorted(set(names))


def test_header_symbols_all_exported(hsw):
    lib = hsw._native.lib()
    declared = _declared_functions()
    assert len(declared) >= 14
    for name in declared:
        assert hasattr(lib, name), "libhsw.so does not export %s" % name
    assert set(declared) == set(hsw._native.SYMBOLS), "binding list out of sync with hsw.h"
    assert lib.hsw_abi_version() == 3


def test_every_bound_function_has_a_signature(hsw):
    """ctypes passes Python ints as 32-bit C ints unless argtypes says otherwise: a binding without a
    signature truncates 64-bit pointers (a host-side crash waiting to happen)."""
    lib = hsw._native.lib()
    for name in hsw._native.SYMBOLS:
        f = getattr(lib, name)
        assert f.argtypes is not None or name == "hsw_abi_version", "no argtypes for %s" % name


def test_library_is_in_tree_and_has_gfx950_code(hsw):
    path = hsw._native.LIB_PATH
    assert path.startswith(ROOT) and os.path.exists(path)
    blob = open(path, "rb").read()
    assert b"gfx950" in blob, "no gfx950 code object embedded"
    assert b"hsw_expand_kernel" in blob


@pytest.mark.parametrize("bits,ncols", [(8, 2), (16, 1), (4, 3), (2, 2), (1, 7), (8, 9)])
def test_shape_query_matches_oracle_measurement(hsw, oracle, bits, ncols):
    s = hsw.shape_query(bits, ncols)
    g, lc = oracle.measure_shape(bits, ncols)
    assert s.gate_cells_per_block == g and s.limb_calls_per_block == lc
    assert s.chip_cells_per_block == 2 * lc and s.limbs_per_spread == 16 // bits
    assert s.algorithmic_bytes_per_block == (g + 2 * lc) * 32 + 128
    assert s.off_feed + 80 == g
    assert s.off_rounds + 64 * s.cells_per_round == s.off_feed


def test_survey_numbers_at_reference_config(hsw):
    s = hsw.shape_query(8, 2).as_dict()
    assert s["gate_cells_per_block"] == 66308 and s["chip_cells_per_block"] == 8240
    assert s["algorithmic_bytes_per_block"] == 2385664
    assert (s["cells_per_state_spread"], s["cells_per_sigma"], s["cells_per_ch"], s["cells_per_maj"],
            s["cells_per_sched_step"], s["cells_per_round"]) == (46, 138, 228, 112, 340, 760)


@pytest.mark.parametrize("bits,ncols", [(0, 2), (3, 2), (5, 1), (32, 2), (8, 0)])
def test_bad_shapes_are_hard_errors(hsw, bits, ncols):
    """spread.rs:37 debug_assert -> HSW_ERR_SHAPE."""
    with pytest.raises(hsw.HswError) as ei:
        hsw.shape_query(bits, ncols)
    assert ei.value.status == hsw._native.HSW_ERR_SHAPE


def test_chip_rows(hsw):
    lib = hsw._native.lib()
    s = hsw.shape_query(8, 2)
    assert lib.hsw_chip_rows(C.byref(s), 0, 1) == 2060
    assert lib.hsw_chip_rows(C.byref(s), 1, 1) == 2061      # straddles a row on both ends
    assert lib.hsw_chip_rows(C.byref(s), 0, 4096) == 2060 * 4096
    s3 = hsw.shape_query(8, 3)
    assert lib.hsw_chip_rows(C.byref(s3), 0, 1) == (4120 + 2) // 3
    assert lib.hsw_chip_rows(C.byref(s3), 2, 2) == (2 + 8240 + 2) // 3


def test_strerror_and_null_handling(hsw):
    lib = hsw._native.lib()
    for st in range(0, 8):
        assert lib.hsw_strerror(st)
    assert lib.hsw_strerror(99) == b"unknown status"
    assert lib.hsw_last_error(None) == b""
    assert lib.hsw_shape_query(8, 2, None) == hsw._native.HSW_ERR_INVALID_ARG
    assert lib.hsw_engine_synchronize(None) == hsw._native.HSW_ERR_INVALID_ARG
    lib.hsw_engine_destroy(None)


def test_no_cpu_fallback(hsw):
    """Without a HIP device the engine refuses to exist -- it must not silently
    compute on the CPU."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    h = C.c_void_p()
    rc = hsw._native.lib().hsw_engine_create(0, None, 8, 2, C.byref(h))
    assert rc == hsw._native.HSW_ERR_NO_DEVICE and not h.value
    with pytest.raises(hsw.HswError):
        hsw.WitnessEngine(0, 8, 2)


def test_product_never_imports_oracle():
    pkg = os.path.join(ROOT, "halo2-dynamic-sha256_amd")
    for dp, _, fs in os.walk(pkg):
        for f in fs:
            if f.endswith((".py", ".cpp", ".hip", ".h", ".hpp")) or f == "Makefile":
                txt = open(os.path.join(dp, f), errors="ignore").read()
                assert "liboracle" not in txt and "hsw_oracle" not in txt, f
                assert not re.search(r"^\s*(from|import)\s+oracle", txt, flags=re.M), f


def test_rust_binding_source_lists_every_symbol():
    """rust/hsw-sys (source only -- no Rust toolchain here) must declare every
    function include/hsw.h declares, so the two cannot drift silently."""
    rs = open(os.path.join(ROOT, "rust", "hsw-sys", "src", "lib.rs")).read()
    bound = set(re.findall(r"pub fn (hsw_[a-z0-9_]+)\s*\(", rs))
    assert bound == set(_declared_functions())


@pytest.mark.parametrize("bits", [1, 2, 4, 8, 16])
def test_spread_table_matches_oracle(hsw, oracle, bits):
    """SpreadConfig::load (spread.rs:165-194)."""
    d, s = hsw._native.spread_table(bits)
    ref = oracle.spread_table(bits)
    assert [(int(a), int(b)) for a, b in zip(d, s)] == ref
    with pytest.raises(hsw.HswError):
        hsw._native.spread_table(3)
