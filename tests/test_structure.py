"""hsw_block_structure (the product's own, value-free walk of the reference's call sequence) against
the oracle's recorder, which derives the same constraint structure while computing values: two
independent restatements must agree cell for cell."""
import numpy as np
import pytest


@pytest.mark.parametrize("bits,internals", [(8, False), (8, True), (16, False), (4, False), (2, False), (1, False)])
def test_block_structure_matches_oracle(hsw, oracle, bits, internals):
    N = hsw._native
    mode = N.HSW_MODE_HALO2_INTERNALS if internals else 0
    shape = N.shape_query(bits, 2, mode)
    st = N.block_structure(shape)
    cs = oracle.constraint_system(bits, 2, internals)
    G = int(shape.gate_cells_per_block)
    assert len(st["kind"]) == G == cs["G"]
    # gate rows
    assert np.array_equal(st["gate_rows"].astype(np.int64), cs["gate_starts"])
    # constants: (cell, value)
    kc = np.nonzero(st["kind"] == N.HSW_KIND_CONSTANT if hasattr(N, "HSW_KIND_CONSTANT") else st["kind"] == 1)[0]
    assert np.array_equal(np.stack([kc, st["ref"][kc]], axis=1), cs["const"])
    # copy constraints: Existing cells + assert_equal pairs == the oracle's eq list (it interleaves both)
    ke = np.nonzero(st["kind"] == 2)[0]
    mine = sorted(map(tuple, np.stack([ke, st["ref"][ke]], axis=1).tolist())) + []
    mine = sorted(mine + list(map(tuple, st["assert_eq"].tolist())))
    assert mine == sorted(map(tuple, cs["eq"].tolist()))
    # witnesses are exactly the remaining cells
    assert int((st["kind"] == 0).sum()) == G - len(kc) - len(ke)
    assert np.array_equal(st["range"], cs["range"])
    assert np.array_equal(st["lookup_src"], cs["lookup_src"])
    assert np.array_equal(st["chip"], cs["chip"])
    assert np.array_equal(st["next_state"], cs["next_state_cells"])
    # the tape (call lengths) is consistent with the kinds: every load_witness call is a WITNESS cell
    lens = N.gate_tape(shape)
    starts = np.concatenate([[0], np.cumsum(lens.astype(np.int64))[:-1]]).astype(np.int64)
    assert (st["kind"][starts[lens == 1]] == 0).all()
    assert set(st["gate_rows"].tolist()) == set(starts[lens == 4].tolist())


def test_block_structure_errors(hsw):
    N = hsw._native
    import ctypes as C
    s = N.shape_query(8, 2)
    assert N.lib().hsw_block_structure(None, None, None, None, None, None, None, None, None, None) == N.HSW_ERR_INVALID_ARG
    c = N.StructureCounts()
    assert N.lib().hsw_block_structure(C.byref(s), C.byref(c), None, None, None, None, None, None, None, None) == 0
    assert (c.gate_cells, c.gate_rows, c.assert_eq, c.limb_calls) == (66308, 13510, 3850, 4120)
    assert c.ranges == 1664 + 760 and c.lookups == 3184


@pytest.mark.parametrize("msgs,sizes,rc", [([b"abc"], [64], False), ([b"abc", b""], [128, 64], True)])
def test_whole_digest_structure_assembled_from_the_product_matches_oracle(hsw, oracle, msgs, sizes, rc):
    """Prologue + zero cell + blocks + epilogue structures of libhsw.so, linked through their external
    ids (input bytes, pre-states / next states, target round, zero cell), give exactly the constraint
    system the oracle records for the whole digest() calls."""
    N = hsw._native
    shape = N.shape_query(8, 2, N.HSW_MODE_HALO2_INTERNALS)
    blk = N.block_structure(shape)
    G = int(shape.gate_cells_per_block)
    ref = oracle.digest_cells(msgs, sizes, None, rc, record=True)
    cs = ref["cs"]
    eq, const, rng, lk, chip, rows = [], [], [], [], [], []
    zero_abs = None
    for h, (mx, lay) in enumerate(zip(sizes, ref["layouts"])):
        nb = mx // 64
        pro, epi = N.frame_structure(shape, mx, rc, 0), N.frame_structure(shape, mx, rc, 1)
        g0 = lay["gate0"]
        P = len(pro["kind"])
        assert P == lay["prologue_cells"] and len(epi["kind"]) == lay["epilogue_cells"]
        if h == 0:
            zero_abs = g0 + P
            const.append((zero_abs, 0))
        B = [g0 + P + lay["zero_cells"] + b * G for b in range(nb)]
        E = B[-1] + G

        def state_cell(n, i):
            return g0 + 38 + i if n == 0 else B[n - 1] + int(blk["next_state"][i])

        def emit(st, base, ext):
            for c, (k, r) in enumerate(zip(st["kind"].tolist(), st["ref"].tolist())):
                if k == 1:
                    const.append((base + c, r))
                elif k == 2:
                    eq.append((base + c, ext(r)))
            rows.extend((base + st["gate_rows"].astype(np.int64)).tolist())
            eq.extend((ext(a), ext(b)) for a, b in st["assert_eq"].tolist())
            rng.extend((ext(c), b) for c, b in st["range"].tolist())
            lk.extend(ext(c) for c in st["lookup_src"].tolist())

        def pro_ext(r):
            assert r >= 0
            return g0 + r
        emit(pro, g0, pro_ext)
        const.extend((g0 + c, k) for c, k in pro["assert_const"].tolist())
        for b in range(nb):
            def blk_ext(r, b=b):
                if r >= 0:
                    return B[b] + r
                if -64 <= r <= -1:
                    return g0 + 46 + 64 * b + (-1 - r)
                if -107 <= r <= -100:
                    return state_cell(b, -100 - r)
                assert r == -1000, r
                return zero_abs
            emit(blk, B[b], blk_ext)
            chip.extend((blk_ext(a), blk_ext(c)) for a, c in blk["chip"].tolist())

        def epi_ext(r):
            if r >= 0:
                return E + r
            if r == -1000:
                return zero_abs
            if r == -3000:
                return g0 + 34
            n, i = divmod(-4000 - r, 8)
            return state_cell(n, i)
        emit(epi, E, epi_ext)
        assert not len(epi["assert_const"])
    assert sorted(rows) == sorted(ref["gate_rows"].tolist())
    assert sorted(eq) == sorted(map(tuple, cs["eq"].tolist()))
    assert sorted(const) == sorted(map(tuple, cs["const"].tolist()))
    assert rng == list(map(tuple, cs["range"].tolist()))
    assert lk == cs["lookup_src"].tolist()
    assert chip == list(map(tuple, cs["chip"].tolist()))
