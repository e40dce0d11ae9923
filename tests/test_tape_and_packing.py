"""Placement adaptor, host side: the call-length tape and the
FlexGate column-packing plan of libhsw.so, against the oracle's per-cell tape
and a straightforward Python model of halo2-lib v0.2.x assign_region (A3)."""
import numpy as np
import pytest


def _oracle_call_lens(oracle, bits, internals):
    """Call lengths recovered from the oracle's per-cell kinds: a 0 is a
    load_witness call; a 1 starts a gate row of 4 cells (internals rows of a
    2-limb range check are one call of 4 cells as well)."""
    k = oracle.gate_tape(bits, 2, internals)
    lens, i = [], 0
    while i < len(k):
        if k[i] == 0:
            lens.append(1)
            i += 1
        else:
            assert list(k[i:i + 4]) == [1, 2, 3, 4], (i, k[i:i + 4])
            lens.append(4)
            i += 4
    return np.array(lens, dtype=np.uint8)


@pytest.mark.parametrize("bits,internals", [(8, False), (8, True), (16, False), (4, False), (4, True), (2, False)])
def test_tape_matches_oracle(hsw, oracle, bits, internals):
    mode = hsw._native.HSW_MODE_HALO2_INTERNALS if internals else 0
    s = hsw.shape_query(bits, 2, mode)
    lens = hsw._native.gate_tape(s)
    ref = _oracle_call_lens(oracle, bits, internals)
    assert np.array_equal(lens, ref)
    assert int(lens.sum()) == s.gate_cells_per_block == oracle.measure_shape(bits, 2, internals)[0]
    assert s.gate_calls_per_block == len(lens)
    if bits == 8:
        assert s.lookup_cells_per_block == 3184 == oracle.lookup_cells_per_block(8, 2)
        # 12,268 load_witness + 13,510 gates (+ 760 range_check(32) rows with internals)
        assert (lens == 1).sum() == 12268 and (lens == 4).sum() == 13510 + (760 if internals else 0)
        assert s.gate_cells_per_block == (69348 if internals else 66308)


def _model_pack(lens, n_blocks, start_row, max_rows):
    """halo2-lib v0.2.x FlexGate::assign_region: `if row + len >= max_rows { row = 0; column += 1 }`."""
    col, row, pos = 0, start_row, []
    for _ in range(n_blocks):
        for ln in lens.tolist():
            if row + ln >= max_rows:
                col, row = col + 1, 0
            pos.append((col, row))
            row += ln
    return pos, col, row


@pytest.mark.parametrize("n_blocks,start_row,max_rows,internals", [
    (1, 0, 1 << 20, False), (2, 100, 131000, True), (16, 5000, 131063, True),      # bench circuit: 16 blocks, k = 17
    (4, 0, 92000, False), (3, 69000, 70000, True), (5, 17, 20011, False)])
def test_pack_plan_matches_model(hsw, n_blocks, start_row, max_rows, internals):
    mode = hsw._native.HSW_MODE_HALO2_INTERNALS if internals else 0
    s = hsw.shape_query(8, 2, mode)
    lens = hsw._native.gate_tape(s)
    G = int(s.gate_cells_per_block)
    try:
        plan = hsw._native.pack_plan(s, n_blocks, start_row, max_rows)
    except hsw.HswError as e:
        assert e.status == hsw._native.HSW_ERR_TOO_LARGE       # more than 8 column breaks
        _, cols, _ = _model_pack(lens, n_blocks, start_row, max_rows)
        assert cols > 8
        return
    pos, cols, end_row = _model_pack(lens, n_blocks, start_row, max_rows)
    assert plan.n_breaks == cols and plan.columns_touched == cols + 1 and plan.end_row == end_row
    # flat position of call j = start cell index + gaps so far == (col * max_rows + row) - start_row
    starts = np.concatenate([[0], np.cumsum(np.tile(lens, n_blocks).astype(np.int64))[:-1]])
    bc = np.array(list(plan.break_cell)[: plan.n_breaks], dtype=np.int64)
    bg = np.array(list(plan.break_gap)[: plan.n_breaks], dtype=np.int64)
    for j in list(range(0, len(starts), 997)) + [len(starts) - 1]:
        flat = starts[j] + bg[bc <= starts[j]].sum()
        c, r = pos[j]
        assert flat == c * max_rows + r - start_row, j
    assert plan.span_cells == n_blocks * G + bg.sum()


def test_pack_plan_argument_errors(hsw):
    s = hsw.shape_query(8, 2)
    N = hsw._native
    for args in [(1, 10, 10), (1, 0, 4)]:
        with pytest.raises(hsw.HswError) as ei:
            N.pack_plan(s, *args)
        assert ei.value.status == N.HSW_ERR_INVALID_ARG
    with pytest.raises(hsw.HswError) as ei:
        N.pack_plan(s, 64, 0, 70000)          # 64 blocks over 70,000-row columns: > 8 breaks
    assert ei.value.status == N.HSW_ERR_TOO_LARGE
