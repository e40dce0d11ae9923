"""The recorded constraint structure itself (CPU): counts per SURVEY 8a, every
stream cell accounted for, and the oracle's own streams pass the checker."""
import numpy as np
import pytest

from tests.constraint_check import check_block_batch, chip_in_call_order


@pytest.mark.parametrize("internals", [False, True])
def test_structure_counts(oracle, internals):
    cs = oracle.constraint_system(8, 2, internals)
    G = cs["G"]
    assert G == (69348 if internals else 66308)
    assert len(cs["gate_starts"]) == 13510 + (760 if internals else 0)
    assert len(cs["chip"]) == 4120 and len(cs["lookup_src"]) == 3184
    assert len(cs["range"]) == 1664 + 760                               # RC16 + RC32
    # every cell of the stream is a load_witness cell, a gate output / free witness, or constrained:
    # gate inputs are Existing (copy) or Constant (fixed); nothing else exists
    kinds = cs["kinds"]
    n_lw = int((kinds == 0).sum())
    assert n_lw == 12268
    constrained = set(cs["eq"][:, 0].tolist()) | set(cs["const"][:, 0].tolist())
    gate_inputs = [int(s) + k for s in cs["gate_starts"] for k in range(3)]
    free = [c for c in gate_inputs if c not in constrained]
    # free gate inputs: neg's output sits at position 1 (128 of them); with internals the two limbs of
    # each range-check row (positions 0, 1) are halo2-base witnesses
    assert len(free) == 128 + (2 * 760 if internals else 0)
    # 3,850 assert_equal + one copy per Existing gate input (+ 760 constrain_equal(a, acc) with internals)
    # (neg rows [a, -a, 1, 0] also fix their LAST cell to the constant 0: 128 constants outside positions 0..2)
    const_inputs = len(set(cs["const"][:, 0].tolist()) & set(gate_inputs))
    assert len(cs["const"]) == const_inputs + 128
    n_existing = len(gate_inputs) - len(free) - const_inputs
    assert len(cs["eq"]) == 3850 + n_existing + (760 if internals else 0)
    # external cells referenced: 64 input bytes, 8 pre-state words, the zero cell
    ext = set(int(c) for c in cs["eq"].reshape(-1) if c < 0) - {oracle.CELL_HIDDEN}
    assert ext == set(range(-64, 0)) | set(range(-107, -99)) | {oracle.CELL_ZERO}


@pytest.mark.parametrize("bits,ncols,internals", [(8, 2, False), (8, 2, True), (16, 1, False), (4, 3, False)])
def test_oracle_streams_satisfy_the_structure(oracle, bits, ncols, internals):
    rng = np.random.default_rng(2)
    n = 3
    blocks = rng.integers(0, 256, (n, 64), dtype=np.uint8)
    blocks[0] = 0xFF
    pre = rng.integers(0, 2**32, (n, 8), dtype=np.uint64).astype(np.uint32)
    pre[0] = 0xFFFFFFFF
    w = oracle.Oracle(bits, ncols, check=True, internals=internals).witness_blocks(blocks, pre)
    cs = oracle.constraint_system(bits, ncols, internals)
    gate = w["gate"].view(np.int64).reshape(n, cs["G"], 4)
    dl = chip_in_call_order(np, w["dense"].view(np.int64), n, cs["LC"], ncols)
    sl = chip_in_call_order(np, w["spread"].view(np.int64), n, cs["LC"], ncols)
    tab = [s for _, s in oracle.spread_table(bits)]
    lookup = w["lookup"].view(np.int64)[:, 0].reshape(n, cs["LK"]) if internals else None
    cnt = check_block_batch(np, cs, gate, blocks.astype(np.int64), pre.astype(np.int64), dl, sl,
                            w["next_states"].astype(np.int64), lookup, tab, bits)
    assert cnt > n * 40000


def test_checker_catches_a_single_wrong_cell(oracle):
    """Flip one cell at a time (a witness, a gate output, a constant, a chip cell): the checker must object."""
    rng = np.random.default_rng(3)
    blocks = rng.integers(0, 256, (1, 64), dtype=np.uint8)
    pre = oracle.INIT_STATE.reshape(1, 8).copy()
    w = oracle.Oracle(8, 2, check=True).witness_blocks(blocks, pre)
    cs = oracle.constraint_system(8, 2, False)
    tab = [s for _, s in oracle.spread_table(8)]
    base_gate = w["gate"].view(np.int64).reshape(1, cs["G"], 4)
    dl = chip_in_call_order(np, w["dense"].view(np.int64), 1, cs["LC"], 2)
    sl = chip_in_call_order(np, w["spread"].view(np.int64), 1, cs["LC"], 2)
    args = (blocks.astype(np.int64), pre.astype(np.int64))
    check_block_batch(np, cs, base_gate, *args, dl, sl, w["next_states"].astype(np.int64), None, tab, 8)
    for cell in [0, 3, 257, 1000, 17588 + 5, 40000, 66307] + rng.integers(0, cs["G"], 25).tolist():
        g = base_gate.copy()
        g[0, cell, 0] ^= 1
        with pytest.raises(AssertionError):
            check_block_batch(np, cs, g, *args, dl, sl, w["next_states"].astype(np.int64), None, tab, 8)
    d2 = dl.copy()
    d2[0, 77] ^= 1
    with pytest.raises(AssertionError):
        check_block_batch(np, cs, base_gate, *args, d2, sl, w["next_states"].astype(np.int64), None, tab, 8)
