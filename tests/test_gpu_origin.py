"""hsw_gadget_set_origin: the whole-region gadget starts wherever the caller's halo2-base Context stands.

The reference's digest() works on any Context it is handed (lib.rs:71-76, 351-360): a circuit that has used the
gate / range chips before its first digest has ctx.advice_alloc[0] = (column, row) != (0, 0), usually a cached
zero cell (ctx.zero_cell, A4-iii) and cells queued for the lookup column (ctx.cells_to_lookup).  The column
image, the lookup stream and the reported positions are compared with the oracle's streams laid out by a Python
model of FlexGate::assign_region started at that (column, row); nothing above the origin row and nothing before
the queued lookups may be touched, on the device or in the caller's host buffers."""
import ctypes as C
import hashlib

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
MAX_ROWS = (1 << 17) - 9          # k = 17, lib.rs:491 / benches/digest.rs:106


@pytest.fixture(scope="module")
def eng_int(hsw):
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no HIP device")
    e = hsw.WitnessEngine(0, 8, 2, mode=hsw._native.HSW_MODE_HALO2_INTERNALS)
    yield e
    e.close()


def model_columns(call_lens, gate, max_rows, start_row=0):
    """halo2-lib v0.2.x FlexGate::assign_region over the oracle's call tape (A3-iii), the Context standing at
    row start_row of its current column (image column 0): `if row + len >= max_rows { column += 1; row = 0 }`.
    Returns the image, a mask of the cells the gadget assigns, and (last image column, next free row)."""
    cols, mask, col, row, pos = [np.zeros((max_rows, 4), dtype=np.uint64)], [np.zeros(max_rows, dtype=bool)], 0, start_row, 0
    for ln in call_lens.tolist():
        if row + ln >= max_rows:
            cols.append(np.zeros((max_rows, 4), dtype=np.uint64))
            mask.append(np.zeros(max_rows, dtype=bool))
            col, row = col + 1, 0
        cols[col][row:row + ln] = gate[pos:pos + ln]
        mask[col][row:row + ln] = True
        row += ln
        pos += ln
    assert pos == len(gate)
    return np.stack(cols), np.stack(mask), (col, row)


ORIGINS = [  # (column, row, zero cell loaded, lookups queued)
    (0, 17, False, 0),
    (2, 131000, False, 5),           # the first column break falls inside the first prologue
    (1, 40000, True, 1234),          # a Context that has already called load_zero and queued lookups
    (0, MAX_ROWS - 1, False, 0),     # not even one cell fits: the region starts on the next column
    (5, 0, True, 0),
]


@pytest.mark.parametrize("origin", ORIGINS, ids=lambda o: "c%d_r%d_z%d_l%d" % (o[0], o[1], int(o[2]), o[3]))
@pytest.mark.parametrize("mont", [False, True], ids=["canonical", "montgomery"])
def test_region_from_any_context_origin(hsw, oracle, eng_int, origin, mont):
    N = hsw._native
    col0, row0, zero, lq = origin
    msgs, sizes = [b"abc", b""], [128, 128]                    # the reference's TestCircuit (lib.rs:455-466)
    ref = oracle.digest_cells(msgs, sizes, None, True, zero_cell_loaded=zero)
    conv = oracle.to_montgomery if mont else (lambda x: x)
    img, mask, (last_col, end_row) = model_columns(ref["call_lens"], conv(ref["gate"]), MAX_ROWS, row0)
    assert len(ref["gate"]) == 279797 - (1 if zero else 0)     # one cell shorter without the zero cell

    cfg = hsw.Sha256DynamicConfig(eng_int, sizes, is_input_range_check=True, whole_digest=True)
    if mont:
        cfg.set_repr(N.HSW_REPR_MONTGOMERY)
    cfg.set_origin(col0, row0, zero, lq)
    assert cfg.set_columns(MAX_ROWS) == img.shape[0]
    v = cfg.view()
    assert (int(v.origin_column), int(v.origin_row), int(v.origin_lookups), int(v.origin_zero_loaded)) == (col0, row0, lq, int(zero))
    assert int(v.lookup_cells) == lq and int(v.gate_cells) == 0
    for trial in range(2):                                     # the origin survives a reset (next synthesis pass)
        res = [cfg.digest(m) for m in msgs]
        assert [r.output_bytes for r in res] == [hashlib.sha256(m).digest() for m in msgs]
        rep = cfg.verify()
        assert rep["violations"] == 0 and rep["checks"] > 4 * 69348 // 4, rep
        st = cfg.streams()
        bad = np.nonzero((st["gate"] != img).any(axis=2))
        assert len(bad[0]) == 0, "first differing (image column, row): %s" % [(int(c), int(r)) for c, r in zip(*bad)][:6]
        assert not st["gate"][0, :row0].any()                  # rows above the origin: the caller's, never written
        assert np.array_equal(st["lookup"][lq:], conv(ref["lookup"])) and not st["lookup"][:lq].any()
        assert np.array_equal(st["dense"], conv(ref["dense"])[:, : st["rows"]])
        # positions are FlexGate columns; sections where the oracle has them
        assert cfg.cell_position(res[-1].end_cell - 1) == (col0 + last_col, end_row - 1)
        if row0 + 1 < MAX_ROWS:
            assert cfg.cell_position(0) == (col0, row0)
        else:
            assert cfg.cell_position(0) == (col0 + 1, 0)
        for r, lay in zip(res, ref["layouts"]):
            assert r.prologue_cell == lay["gate0"] and r.prologue_lookup == lq + lay["lookup0"]
            assert r.block_cell == lay["gate0"] + lay["prologue_cells"] + lay["zero_cells"]
        assert ref["layouts"][0]["zero_cells"] == (0 if zero else 1)
        rc = N.ResultCells()
        cfg._ok(cfg.lib.hsw_gadget_result_cells(cfg.h, 1, C.byref(rc)))
        c, r_ = int(rc.output_byte_pos[0][0]), int(rc.output_byte_pos[0][1])
        assert int(st["gate"][c - col0, r_, 0]) == (int(conv(np.array([[hashlib.sha256(b"").digest()[0], 0, 0, 0]], dtype=np.uint64))[0, 0]))
        if trial == 0:
            # host delivery: only the gadget's cells travel -- what the caller's buffers hold elsewhere stays
            sentinel = np.uint64(0xDEADBEEFCAFEF00D)
            gate_h = np.full(img.shape, sentinel, dtype=np.uint64)
            look_h = np.full((int(cfg.view().lookup_cells), 4), sentinel, dtype=np.uint64)
            dst = N.RegionHost(gate_h.ctypes.data, look_h.ctypes.data, None, None)
            cfg._ok(cfg.lib.hsw_gadget_download_region(cfg.h, C.byref(dst)))
            assert np.array_equal(gate_h[mask], img[mask]) and (gate_h[~mask] == sentinel).all()
            assert np.array_equal(look_h[lq:], conv(ref["lookup"])) and (look_h[:lq] == sentinel).all()
            if not mont:
                bufs, n_wide = cfg.download_region_compact()
                bufs["gate"][:] = sentinel
                bufs["lookup"][:] = sentinel
                bufs, n_wide = cfg.download_region_compact(bufs)
                g8 = bufs["gate"].reshape(img.shape[0], MAX_ROWS)
                assert (g8[0, :row0] == sentinel).all() and (bufs["lookup"][:lq] == sentinel).all()
                wide = cfg.widen(bufs["gate"], 0, bufs["wide"], n_wide).reshape(img.shape)
                first = row0 if row0 + 1 < MAX_ROWS else MAX_ROWS        # (all of image column 0 may be the caller's)
                used = np.zeros(img.shape[:2], dtype=bool)
                used.reshape(-1)[first: last_col * MAX_ROWS + end_row] = True
                assert np.array_equal(wide[used], img[used])
            cfg.reset()
            assert int(cfg.view().lookup_cells) == lq and int(cfg.view().gate_cells) == 0
    cfg.close()


def test_linear_stream_in_a_context_with_a_zero_cell(hsw, oracle, eng_int):
    """Without a column image the origin only decides whether a zero cell is assigned and where the lookups
    start; bench-circuit shape (one 16-block digest, benches/digest.rs:93-129)."""
    msg = bytes([1] * 56)
    ref = oracle.digest_cells([msg], [1024], None, True, zero_cell_loaded=True)
    cfg = hsw.Sha256DynamicConfig(eng_int, [1024], is_input_range_check=True, whole_digest=True)
    cfg.set_origin(3, 1000, True, 9)
    r = cfg.digest(msg)
    st = cfg.streams()
    assert int(cfg.view().gate_cells) == 1116315 - 1 == len(ref["gate"])
    assert np.array_equal(st["gate"], ref["gate"])
    assert np.array_equal(st["lookup"][9:], ref["lookup"]) and not st["lookup"][:9].any()
    assert cfg.verify()["violations"] == 0
    assert cfg.cell_position(7) == (3, 1007)
    assert r.block_cell == r.prologue_cell + ref["layouts"][0]["prologue_cells"]      # no zero cell in between
    cfg.close()


def test_origin_with_large_batches_and_seek(hsw, oracle, eng_int):
    """140 single-block digests (the two-launch path: streaming kernel + hsw_frame_kernel) from a non-zero origin,
    dealt to two gadgets with hsw_gadget_seek: the union is the single-gadget image."""
    rng = np.random.default_rng(0x0516)
    n = 140
    msgs = [rng.integers(0, 256, int(rng.integers(0, 56)), dtype=np.uint8).tobytes() for _ in range(n)]
    sizes = [64] * n
    row0, lq, rows = 77777, 31, 700001
    ref = oracle.digest_cells(msgs, sizes, None, False, zero_cell_loaded=False)
    img, _, (last_col, end_row) = model_columns(ref["call_lens"], ref["gate"], rows, row0)

    def run(first, last):
        cfg = hsw.Sha256DynamicConfig(eng_int, sizes, is_input_range_check=False, whole_digest=True)
        cfg.set_columns(rows)
        cfg.set_origin(4, row0, False, lq)                    # after set_columns: the image is laid out again
        if first:
            cfg.seek(first)
        res = cfg.digest_batch(msgs[first:last])
        assert cfg.verify()["violations"] == 0
        v = cfg.view()
        out = np.zeros((int(v.columns) * rows, 4), dtype=np.uint64)
        cfg._ok(cfg.lib.hsw_download(eng_int.h, out.ctypes.data, v.d_gate, out.nbytes))
        lk = np.zeros((int(v.lookup_capacity), 4), dtype=np.uint64)
        cfg._ok(cfg.lib.hsw_download(eng_int.h, lk.ctypes.data, v.d_lookup, lk.nbytes))
        pos = cfg.cell_position(res[-1].end_cell - 1)
        cfg.close()
        return out.reshape(-1, rows, 4), lk, res, pos

    full, lk, res, pos = run(0, n)
    assert full.shape == img.shape and np.array_equal(full, img)
    assert pos == (4 + last_col, end_row - 1)
    assert np.array_equal(lk[lq:], ref["lookup"]) and not lk[:lq].any()
    assert [r.output_bytes for r in res] == [hashlib.sha256(m).digest() for m in msgs]
    a, lka, _, _ = run(0, 60)
    b, lkb, _, _ = run(60, n)
    assert np.array_equal(a | b, img) and not (a.any(axis=2) & b.any(axis=2)).any()
    assert np.array_equal(lka | lkb, lk)


def test_set_origin_argument_errors(hsw, eng_int, engine_factory):
    N = hsw._native
    cfg = hsw.Sha256DynamicConfig(eng_int, [64, 64], whole_digest=True)
    cfg.set_columns(MAX_ROWS)
    with pytest.raises(hsw.HswError):
        cfg.set_origin(0, MAX_ROWS, False, 0)                 # the next free row lies inside the column
    cfg.set_origin(0, 5, False, 0)
    cfg.digest(b"x")
    with pytest.raises(hsw.HswError):
        cfg.set_origin(0, 6, False, 0)                        # only before the first digest of a pass
    cfg.reset()
    cfg.set_origin(1, 6, True, 3)                             # ... or after a reset
    assert cfg.cell_position(0) == (1, 6)
    cfg.close()
    plain = engine_factory(8, 2)
    blocks_only = hsw.Sha256DynamicConfig(plain, [64])        # block-stream contexts have no region
    with pytest.raises(hsw.HswError):
        blocks_only.set_origin(0, 1, False, 0)
    blocks_only.close()
    # a layout that no longer fits 17 columns from the new row is refused and the old one kept
    many = hsw.Sha256DynamicConfig(eng_int, [64] * 16, whole_digest=True)
    rows = 69348 + 16
    while True:                                               # the smallest column height (in steps) that fits 17 columns
        try:
            assert many.set_columns(rows) == 17
            break
        except hsw.HswError as e:
            assert e.status == N.HSW_ERR_TOO_LARGE
            rows += 97
    with pytest.raises(hsw.HswError) as ei:
        many.set_origin(0, rows - 2, False, 0)
    assert ei.value.status == N.HSW_ERR_TOO_LARGE
    assert many.cell_position(0) == (0, 0) and int(many.view().columns) == 17
    many.close()


def test_compact_staging_follows_the_geometry(hsw, oracle, eng_int):
    """the 8-byte staging of hsw_gadget_download_region_compact was sized once; a reset followed by
    set_columns / set_origin with a larger image then wrote past it.  Now it is dropped with the geometry."""
    sizes, msgs = [128, 64], [b"compact", b"z"]
    cfg = hsw.Sha256DynamicConfig(eng_int, sizes, is_input_range_check=True, whole_digest=True)
    cfg.digest_batch(msgs)
    bufs, n_wide = cfg.download_region_compact()               # staging sized for the linear stream
    lin = cfg.widen(bufs["gate"], 0, bufs["wide"], n_wide)
    assert np.array_equal(lin, cfg.download_region(pinned=False)["gate"])
    cfg.reset()
    cfg.set_columns(70000)                                     # 4 columns x 70,000 rows > the linear stream
    cfg.set_origin(0, 69000, False, 0)
    cfg.digest_batch(msgs)
    full = cfg.download_region(pinned=False)
    bufs, n_wide = cfg.download_region_compact()
    img = cfg.widen(bufs["gate"], 0, bufs["wide"], n_wide).reshape(full["gate"].shape)
    assert np.array_equal(img, full["gate"])
    assert cfg.verify()["violations"] == 0
    cfg.close()


@pytest.mark.parametrize("layout", ["linear", "columns", "columns-at-origin", "independent"])
@pytest.mark.parametrize("mont", [False, True], ids=["canonical", "montgomery"])
def test_distinct_value_delivery_rebuilds_the_region(hsw, oracle, eng_int, layout, mont):
    """hsw_gadget_download_region_distinct + hsw_gadget_replay_region: only the new witnesses (~40 % of the cells)
    cross PCIe; the tape (input independent: copies resolved to their root witness or to a constant) rebuilds the
    gate image, the lookup column and the chip columns on the host -- bit-equal to hsw_gadget_download_region."""
    N = hsw._native
    sizes, msgs = [128, 64, 192], [b"distinct values", b"", bytes(range(150))]
    cfg = hsw.Sha256DynamicConfig(eng_int, sizes, is_input_range_check=True, whole_digest=True, independent=layout == "independent")
    if mont:
        cfg.set_repr(N.HSW_REPR_MONTGOMERY)
    if layout.startswith("columns"):
        cfg.set_columns(100003)
    if layout == "columns-at-origin":
        cfg.set_origin(3, 99990, True, 11)                  # the first break a few cells into the first prologue
    bufs = None
    for trial, batch in enumerate((msgs[:2], msgs)):        # after two digests, then after all three (and a reset)
        cfg.digest_batch(batch)
        full = cfg.download_region(pinned=False)
        got = cfg.download_region_distinct(threads=1 + 3 * trial, bufs=bufs)
        bufs = got["bufs"]
        assert 0.25 < got["n_distinct"] / int(cfg.view().gate_cells) < 0.5
        assert np.array_equal(got["gate"], full["gate"]), layout
        lq = int(cfg.view().origin_lookups)
        assert np.array_equal(got["lookup"][lq:], full["lookup"][lq:])
        assert np.array_equal(got["dense"], full["dense"]) and np.array_equal(got["spread"], full["spread"])
        cfg.reset()
    # the tape alone: every cell is value(code[i])
    cfg.digest_batch(msgs)
    tape = N.RegionTape()
    cfg._ok(cfg.lib.hsw_gadget_region_tape(cfg.h, C.byref(tape)))
    st = cfg.streams()
    got = cfg.download_region_distinct(bufs=bufs)
    code = np.ctypeslib.as_array(tape.gate_code, shape=(int(tape.gate_cells),))
    consts = np.ctypeslib.as_array(C.cast(tape.consts, C.POINTER(C.c_uint64)), shape=(int(tape.n_consts), 4))
    is_c = (code & np.uint32(N.HSW_TAPE_CONST)) != 0
    vals = np.where(is_c[:, None], consts[np.where(is_c, code & np.uint32(0x7fffffff), 0)], got["distinct"][np.where(is_c, 0, code)])
    if layout.startswith("columns"):
        lin = np.zeros_like(vals)
        for i in range(0, len(vals), 4099):                 # a sample of positions through cell_position
            c_, r_ = cfg.cell_position(i)
            assert np.array_equal(st["gate"][c_ - int(cfg.view().origin_column), r_], vals[i]), i
    else:
        assert np.array_equal(st["gate"], vals)
    cfg.close()


def test_the_tape_survives_synthesis_passes(hsw, oracle, eng_int):
    """A prover synthesizes the same circuit pass after pass: reset + set_origin at the same place must not cost a
    new tape (1.1 M cells walked on the host for the bench circuit).  The codes number STREAM cells: a new origin
    row or column height only moves the witnesses' image positions; only a zero cell that comes or goes changes
    the stream itself.  Every delivery stays bit-equal to hsw_gadget_download_region."""
    N = hsw._native
    sizes, msgs = [128, 128], [b"abc", b""]
    cfg = hsw.Sha256DynamicConfig(eng_int, sizes, is_input_range_check=True, whole_digest=True)
    cfg.set_repr(N.HSW_REPR_MONTGOMERY)
    cfg.set_origin(1, 500, False, 3)
    cfg.set_columns(MAX_ROWS)

    def one_pass(origin):
        cfg.reset()
        cfg.set_origin(*origin)
        cfg.digest_batch(msgs)
        tape = N.RegionTape()
        cfg._ok(cfg.lib.hsw_gadget_region_tape(cfg.h, C.byref(tape)))
        full = cfg.download_region(pinned=False)
        got = cfg.download_region_distinct(threads=2)
        lq = origin[3]
        assert np.array_equal(got["gate"], full["gate"]) and np.array_equal(got["lookup"][lq:], full["lookup"][lq:])
        assert np.array_equal(got["dense"], full["dense"]) and np.array_equal(got["spread"], full["spread"])
        return C.cast(tape.gate_code, C.c_void_p).value, (int(tape.distinct_capacity), int(tape.gate_cells))

    p0, n0 = one_pass((1, 500, False, 3))
    p1, n1 = one_pass((1, 500, False, 3))                   # the usual case: the same place again
    assert (p1, n1) == (p0, n0)
    p2, n2 = one_pass((4, 500, False, 77))                  # another column, more lookups queued: offsets only
    assert (p2, n2) == (p0, n0)
    p3, n3 = one_pass((0, 130990, False, 0))                # another row: the breaks move, the stream does not
    assert (p3, n3) == (p0, n0)
    cfg.reset()
    cfg.set_origin(0, 130, False, 0)                        # (row 130,990 does not exist in the shorter columns)
    cfg.set_columns(MAX_ROWS - 1000)                         # between passes: another column height
    p4, n4 = one_pass((0, 130, False, 0))
    assert (p4, n4) == (p0, n0)
    p5, n5 = one_pass((0, 130, True, 0))                    # the Context now caches its zero cell: the stream is one
    assert n5 == (n0[0], n0[1] - 1)                         # cell shorter (a constant: as many distinct values), new tape
    cfg.close()
