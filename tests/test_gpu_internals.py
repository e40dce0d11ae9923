"""Placement adaptor on the GPU: HSW_MODE_HALO2_INTERNALS gate
stream (range_check(32) cells included), the lookup-advice column stream, and
FlexGate column packing -- bit-exact against the oracle and a Python model of
the packing rule.  All of it rests on assumption A3."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _inputs(n, seed):
    rng = np.random.default_rng(seed)
    return (rng.integers(0, 256, (n, 64), dtype=np.uint8),
            rng.integers(0, 2**32, (n, 8), dtype=np.uint64).astype(np.uint32))


@pytest.fixture(scope="module", params=["default", "main-kernel"])
def eng_int(hsw, request):
    """Tiny batches go to the small-batch kernel by default (hsw_small.hpp); "main-kernel" pins the same
    tests to hsw_expand_kernel so that neither loses its coverage."""
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no HIP device")
    e = hsw.WitnessEngine(0, 8, 2, mode=hsw._native.HSW_MODE_HALO2_INTERNALS)
    e.set_option("split", -1 if request.param == "default" else 0)
    yield e
    e.close()


@pytest.mark.parametrize("mont", [False, True])
def test_internals_stream_and_lookup_column(eng_int, oracle, hsw, mont):
    import torch
    blocks, pre = _inputs(4, 77)
    blocks[0] = 0xFF
    pre[0] = 0xFFFFFFFF
    ref = oracle.Oracle(8, 2, check=True, internals=True).witness_blocks(blocks, pre, cursor0=2)
    assert eng_int.G == 69348 == ref["gate_cells_per_block"] and eng_int.lookup_cells == 3184
    out = eng_int.witness_blocks_ex(torch.from_numpy(blocks).cuda(), torch.from_numpy(pre.view(np.int32)).cuda(),
                                    cursor0=2, want_lookup=True, flags=hsw.HSW_REPR_MONTGOMERY if mont else 0)
    eng_int.synchronize()
    conv = oracle.to_montgomery if mont else (lambda x: x)
    g = out["gate"].cpu().numpy().view(np.uint64)
    exp = conv(ref["gate"])
    if not np.array_equal(g, exp):
        bad = np.nonzero((g != exp).any(axis=1))[0]
        raise AssertionError("gate differs at %d cells, first %d: %s vs %s" % (len(bad), bad[0], g[bad[0]], exp[bad[0]]))
    assert np.array_equal(out["lookup"].cpu().numpy().view(np.uint64), conv(ref["lookup"]))
    assert np.array_equal(out["dense"].cpu().numpy().view(np.uint64), conv(ref["dense"]))
    assert np.array_equal(out["next_states"].cpu().numpy().view(np.uint32), ref["next_states"])
    # every lookup value is a 16-bit range-table entry
    assert (ref["lookup"][:, 0] < 65536).all() and not ref["lookup"][:, 1:].any()


@pytest.mark.parametrize("tile,parts", [(32, 1), (32, 8), (64, 2), (64, 4), (128, 4), (128, 16), (0, 0)])
@pytest.mark.parametrize("flags_name", ["canonical", "montgomery", "compact"])
def test_internals_every_tile_shape(eng_int, oracle, hsw, tile, parts, flags_name):
    """The internals kernels are built for the same three tile shapes as the default mode."""
    import torch
    N = hsw._native
    flags = {"canonical": 0, "montgomery": N.HSW_REPR_MONTGOMERY, "compact": N.HSW_REPR_COMPACT64}[flags_name]
    eng_int.set_option("tile", tile)
    eng_int.set_option("parts", parts)
    try:
        blocks, pre = _inputs(3, 1000 + tile + parts)
        ref = oracle.Oracle(8, 2, check=True, internals=True).witness_blocks(blocks, pre, cursor0=5)
        out = eng_int.witness_blocks_ex(torch.from_numpy(blocks).cuda(), torch.from_numpy(pre.view(np.int32)).cuda(),
                                        cursor0=5, want_lookup=True, flags=flags)
        eng_int.synchronize()
        if flags_name == "compact":
            g = out["gate"].cpu().numpy().view(np.uint64).reshape(-1)[: 3 * eng_int.G]
            exp = ref["gate"][:, 0].copy()
            neg = np.nonzero(ref["gate"][:, 1:].any(axis=1))[0]
            exp[neg] = np.uint64(0x43e1f593f0000001) - exp[neg]          # the cell holds x where the value is p - x
            assert np.array_equal(g, exp)
            lk = out["lookup"].cpu().numpy().view(np.uint64).reshape(-1)[: 3 * eng_int.lookup_cells]
            assert np.array_equal(lk, ref["lookup"][:, 0])
        else:
            conv = oracle.to_montgomery if flags_name == "montgomery" else (lambda x: x)
            assert np.array_equal(out["gate"].cpu().numpy().view(np.uint64), conv(ref["gate"]))
            assert np.array_equal(out["lookup"].cpu().numpy().view(np.uint64), conv(ref["lookup"]))
            assert np.array_equal(out["dense"].cpu().numpy().view(np.uint64), conv(ref["dense"]))
    finally:
        eng_int.set_option("tile", 0)
        eng_int.set_option("parts", 0)


@pytest.mark.parametrize("parts", [1, 4, 16])
def test_internals_with_split_blocks(eng_int, oracle, parts):
    import torch
    eng_int.set_option("parts", parts)
    try:
        blocks, pre = _inputs(2, 5 + parts)
        ref = oracle.Oracle(8, 2, check=True, internals=True).witness_blocks(blocks, pre)
        out = eng_int.witness_blocks_ex(torch.from_numpy(blocks).cuda(), torch.from_numpy(pre.view(np.int32)).cuda(),
                                        want_lookup=True)
        eng_int.synchronize()
        assert np.array_equal(out["gate"].cpu().numpy().view(np.uint64), ref["gate"])
        assert np.array_equal(out["lookup"].cpu().numpy().view(np.uint64), ref["lookup"])
    finally:
        eng_int.set_option("parts", 0)


def _model_positions(lens, n_blocks, start_row, max_rows):
    col, row, out = 0, start_row, []
    for _ in range(n_blocks):
        for ln in lens.tolist():
            if row + ln >= max_rows:
                col, row = col + 1, 0
            out.extend((col, row + k) for k in range(ln))
            row += ln
    return out


@pytest.mark.parametrize("internals,n,start_row,max_rows", [
    (True, 3, 1000, 100000),      # one break, in the middle of block 1
    (True, 4, 69000, 70000),      # a break in every block
    (False, 5, 17, 40009),        # two breaks inside one block (default mode, 66,308 cells)
    (True, 16, 4242, 131063)])    # the reference's bench circuit: 16 blocks at k = 17
def test_flexgate_column_packing(hsw, oracle, eng_int, engine_factory, internals, n, start_row, max_rows):
    import torch
    eng = eng_int if internals else engine_factory(8, 2)
    blocks, pre = _inputs(n, 900 + n)
    ref = oracle.Oracle(8, 2, check=False, internals=internals).witness_blocks(blocks, pre)
    out = eng.witness_blocks_ex(torch.from_numpy(blocks).cuda(), torch.from_numpy(pre.view(np.int32)).cuda(),
                                start_row=start_row, max_rows=max_rows)
    eng.synchronize()
    plan = out["plan"]
    flat = out["gate"].cpu().numpy().view(np.uint64)
    assert flat.shape[0] == plan.span_cells
    lens = hsw._native.gate_tape(eng.shape)
    pos = _model_positions(lens, n, start_row, max_rows)
    assert len(pos) == ref["gate"].shape[0]
    idx = np.array([c * max_rows + r - start_row for c, r in pos], dtype=np.int64)
    assert idx.max() == plan.span_cells - 1 and len(np.unique(idx)) == len(idx)
    assert np.array_equal(flat[idx], ref["gate"]), "cells are not where halo2-base's assign_region would put them"
    # the unused tail rows of every closed column were not touched (buffer was pre-filled with -1)
    mask = np.ones(flat.shape[0], dtype=bool)
    mask[idx] = False
    assert (flat[mask] == np.uint64(2**64 - 1)).all()
    assert plan.columns_touched == max(c for c, _ in pos) + 1


def _break_right_after_a_skewed_block_start(shape, n, rng, window):
    """(start_row, max_rows) whose plan puts a column break within `window` cells after the start of a block
    that does not start on a 128-byte line -- the corner tests/fuzz_parity.py seed 77031 found."""
    import importlib
    N = importlib.import_module("halo2-dynamic-sha256_amd._native")
    G = int(shape.gate_cells_per_block)
    for _ in range(20000):
        max_rows = int(rng.integers(G // 2 + 16, 3 * G))
        start_row = int(rng.integers(0, max_rows))
        try:
            plan = N.pack_plan(shape, n, start_row, max_rows)
        except N.HswError:
            continue
        gap = 0
        for k in range(plan.n_breaks):
            bc = int(plan.break_cell[k])
            if 0 < bc % G <= window and ((bc - bc % G) + gap) % 4 != 0:
                return start_row, max_rows, bc // G, bc % G
            gap += int(plan.break_gap[k])
    raise AssertionError("no such layout found")


@pytest.mark.parametrize("tile,parts,window", [(0, 0, 60), (128, 8, 150), (128, 4, 150), (64, 4, 100), (32, 2, 40)])
@pytest.mark.parametrize("mont", [False, True])
def test_column_break_right_after_a_skewed_block_start(hsw, oracle, tile, parts, window, mont):
    """A break that lands inside the first flush of a block whose stream does not start on a 128-byte line:
    the realigning write-out's bounds must not wrap below the block's first cell (they did: the flush was
    taken to lie past the break and shifted by its gap).  The fuzz's own failing case first, then layouts
    searched for per tile shape."""
    import ctypes as C
    import torch
    N = hsw._native
    eng = hsw.WitnessEngine(0, 8, 4, mode=N.HSW_MODE_HALO2_INTERNALS)
    n = 13
    rng = np.random.default_rng(77031 + tile + parts)
    layouts = [(70699, 109190)] if tile == 128 else []
    for _ in range(3):
        layouts.append(_break_right_after_a_skewed_block_start(eng.shape, n, rng, window)[:2])
    blocks, pre = _inputs(n, 4100 + tile)
    ref = oracle.Oracle(8, 4, check=False, internals=True).witness_blocks(blocks, pre)
    exp = oracle.to_montgomery(ref["gate"]) if mont else ref["gate"]
    lens = N.gate_tape(eng.shape)
    tb, tp = torch.from_numpy(blocks).cuda(), torch.from_numpy(pre.view(np.int32)).cuda()
    eng.set_option("tile", tile)
    eng.set_option("parts", parts)
    eng.set_option("split", 0)                          # the streaming kernel (the small-batch one never realigns)
    for start_row, max_rows in layouts:
        plan = N.pack_plan(eng.shape, n, start_row, max_rows)
        gate = torch.full((int(plan.span_cells), 4), -1, dtype=torch.int64, device="cuda")
        rows = eng.chip_rows(0, n)
        dense = torch.zeros((4, rows, 4), dtype=torch.int64, device="cuda")
        spread = torch.zeros((4, rows, 4), dtype=torch.int64, device="cuda")
        lookup = torch.empty((n * eng.lookup_cells, 4), dtype=torch.int64, device="cuda")
        a = N.WitnessArgs()
        a.d_blocks, a.d_pre_states, a.n_blocks, a.spread_cursor0 = tb.data_ptr(), tp.data_ptr(), n, 0
        a.d_gate, a.d_chip_dense, a.d_chip_spread, a.chip_col_stride = gate.data_ptr(), dense.data_ptr(), spread.data_ptr(), rows
        a.d_next_states, a.d_lookup, a.pack = None, lookup.data_ptr(), C.pointer(plan)
        a.flags = N.HSW_REPR_MONTGOMERY if mont else 0
        rc = eng.lib.hsw_witness_blocks_ex(eng.h, C.byref(a))
        assert rc == 0, eng.lib.hsw_last_error(eng.h)
        eng.synchronize()
        pos = _model_positions(lens, n, start_row, max_rows)
        idx = np.array([c * max_rows + r - start_row for c, r in pos], dtype=np.int64)
        flat = gate.cpu().numpy().view(np.uint64)
        bad = np.nonzero((flat[idx] != exp).any(axis=1))[0]
        assert len(bad) == 0, (start_row, max_rows, len(bad), int(bad[0]) // eng.G, int(bad[0]) % eng.G)
        mask = np.ones(flat.shape[0], dtype=bool)
        mask[idx] = False
        assert (flat[mask] == np.uint64(2**64 - 1)).all(), (start_row, max_rows)
    eng.close()


@pytest.mark.parametrize("flags_name", ["HSW_REPR_MONTGOMERY", "HSW_REPR_COMPACT64"])
def test_packing_combined_with_other_representations(hsw, oracle, eng_int, flags_name):
    """Column packing + internals + a non-default cell representation in one call."""
    import torch
    flags = getattr(hsw, flags_name)
    n, start_row, max_rows = 3, 777, 90001
    blocks, pre = _inputs(n, 31)
    ref = oracle.Oracle(8, 2, check=False, internals=True).witness_blocks(blocks, pre)
    plan = hsw._native.pack_plan(eng_int.shape, n, start_row, max_rows)
    width = 1 if flags == hsw.HSW_REPR_COMPACT64 else 4
    gate = torch.full((int(plan.span_cells), width), -1, dtype=torch.int64, device="cuda")
    lookup = torch.empty((n * 3184, width), dtype=torch.int64, device="cuda")
    rows = eng_int.chip_rows(0, n)
    dense = torch.zeros((2, rows, width), dtype=torch.int64, device="cuda")
    spread = torch.zeros((2, rows, width), dtype=torch.int64, device="cuda")
    import ctypes as C
    a = hsw._native.WitnessArgs()
    tb, tp = torch.from_numpy(blocks).cuda(), torch.from_numpy(pre.view(np.int32)).cuda()
    a.d_blocks, a.d_pre_states, a.n_blocks, a.spread_cursor0 = tb.data_ptr(), tp.data_ptr(), n, 0
    a.d_gate, a.d_chip_dense, a.d_chip_spread, a.chip_col_stride = gate.data_ptr(), dense.data_ptr(), spread.data_ptr(), rows
    a.d_next_states, a.d_lookup, a.flags, a.pack = None, lookup.data_ptr(), flags, C.pointer(plan)
    rc = eng_int.lib.hsw_witness_blocks_ex(eng_int.h, C.byref(a))
    assert rc == 0, eng_int.lib.hsw_last_error(eng_int.h)
    eng_int.synchronize()
    lens = hsw._native.gate_tape(eng_int.shape)
    pos = _model_positions(lens, n, start_row, max_rows)
    idx = np.array([c * max_rows + r - start_row for c, r in pos], dtype=np.int64)
    flat = gate.cpu().numpy().view(np.uint64)
    if flags == hsw.HSW_REPR_MONTGOMERY:
        assert np.array_equal(flat[idx], oracle.to_montgomery(ref["gate"]))
        assert np.array_equal(lookup.cpu().numpy().view(np.uint64), oracle.to_montgomery(ref["lookup"]))
    else:
        neg = hsw._native.neg_cells(eng_int.shape).astype(np.int64)
        exp = ref["gate"][:, 0].copy().reshape(n, -1)
        wide = (ref["gate"][:, 1:] != 0).any(axis=1).reshape(n, -1)
        x = (np.uint64(0x43e1f593f0000001) - exp[:, neg]).astype(np.uint64)
        exp[:, neg] = np.where(wide[:, neg], x, exp[:, neg])
        assert np.array_equal(flat[idx, 0], exp.reshape(-1))
        assert np.array_equal(lookup.cpu().numpy().view(np.uint64)[:, 0], ref["lookup"][:, 0])
    mask = np.ones(flat.shape[0], dtype=bool)
    mask[idx] = False
    assert (flat[mask] == np.uint64(2**64 - 1)).all()


@pytest.mark.parametrize("bits,ncols", [(16, 2), (4, 2), (4, 3), (2, 2), (1, 3)])
def test_internals_other_table_widths(hsw, oracle, bits, ncols):
    """Internals mode (range_check rows + lookup column), digest frames and on-device verification for every
    other table width the reference admits (16 % num_bits_lookup == 0)."""
    import hashlib
    import torch
    N = hsw._native
    eng = hsw.WitnessEngine(0, bits, ncols, mode=N.HSW_MODE_HALO2_INTERNALS)
    try:
        blocks, pre = _inputs(3, 40 + bits)
        ref = oracle.Oracle(bits, ncols, check=True, internals=True).witness_blocks(blocks, pre, cursor0=5)
        tb, tp = torch.from_numpy(blocks).cuda(), torch.from_numpy(pre.view(np.int32)).cuda()
        out = eng.witness_blocks_ex(tb, tp, cursor0=5, want_lookup=True)
        eng.synchronize()
        assert np.array_equal(out["gate"].cpu().numpy().view(np.uint64), ref["gate"])
        assert np.array_equal(out["lookup"].cpu().numpy().view(np.uint64), ref["lookup"])
        assert np.array_equal(out["dense"].cpu().numpy().view(np.uint64), ref["dense"])
        assert eng.verify_blocks(tb, tp, out, cursor0=5, lookup=out["lookup"])["violations"] == 0
        # whole digests
        msgs, sizes = [b"abc", bytes(range(100))], [64, 128]
        cfg = hsw.Sha256DynamicConfig(eng, sizes, is_input_range_check=True, whole_digest=True)
        res = cfg.digest_batch(msgs, [None, None])
        assert cfg.verify()["violations"] == 0
        st = cfg.streams()
        cfg.close()
        refd = oracle.digest_cells(msgs, sizes, None, True, num_bits_lookup=bits, num_advice_columns=ncols)
        assert np.array_equal(st["gate"], refd["gate"]) and np.array_equal(st["lookup"], refd["lookup"])
        assert np.array_equal(st["dense"], refd["dense"][:, : st["rows"]])
        for m, r in zip(msgs, res):
            assert r.output_bytes == hashlib.sha256(m).digest()
    finally:
        eng.close()
    if bits <= 2:                                             # ... in 32-byte cells; the compact form is not built there
        eng = hsw.WitnessEngine(0, bits, ncols, mode=N.HSW_MODE_HALO2_INTERNALS)
        with pytest.raises(hsw.HswError) as ei:
            eng.witness_blocks(tb, tp, flags=N.HSW_REPR_COMPACT64)
        assert ei.value.status == N.HSW_ERR_UNSUPPORTED
        out = eng.witness_blocks(tb, tp, cursor0=5, flags=N.HSW_REPR_MONTGOMERY)
        eng.synchronize()
        assert np.array_equal(out["gate"].cpu().numpy().view(np.uint64), oracle.to_montgomery(ref["gate"]))
        eng.close()
