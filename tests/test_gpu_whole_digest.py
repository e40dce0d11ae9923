"""SURVEY 8 f4 on the GPU: a gadget context created with HSW_GADGET_WHOLE_DIGEST emits
EVERY advice cell Sha256DynamicConfig::digest allocates -- prologue (lib.rs:122-178), the
Context's zero cell, the blocks, the epilogue (lib.rs:294-341) -- as one stream, plus the
lookup-advice stream.  Checked bit for bit against the oracle (assumptions A1-A4) and,
independently of any oracle value, against the recorded constraint structure."""
import ctypes as C
import hashlib
import json
import os

import numpy as np
import pytest

from tests.constraint_check import check_whole_stream

pytestmark = pytest.mark.gpu
KATS = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "reference_kats.json")))


@pytest.fixture(scope="module")
def eng_int(hsw):
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no HIP device")
    e = hsw.WitnessEngine(0, 8, 2, mode=hsw._native.HSW_MODE_HALO2_INTERNALS)
    yield e
    e.close()


CASES = [  # (messages, max sizes, precomputed, input range checks, one batch call?)
    ([b"abc"], [64], None, False, True),
    ([b"abc", b""], [128, 128], None, True, False),               # the reference's TestCircuit (lib.rs:455-466)
    ([b"abc", b""], [128, 128], None, True, True),                # same, as ONE launch (equal sizes -> framed run)
    ([bytes([1] * 56)], [192], None, True, True),
    ([bytes(range(200))], [128], [128], False, True),             # precomputed prefix
    ([bytes(range(119)), b"xy", b"q" * 70], [128, 64, 128], None, False, True),   # three runs in one batch
    ([bytes([7] * 200)], [128], [192], False, True),              # target_round = 1 after 3 precomputed rounds
    ([bytes(range(256)) * 14], [4096], None, True, True),         # 64 blocks: the frame kernel runs 16 workgroups
]


def _run(hsw, eng, msgs, sizes, pre, rc, batch, repr_flag=0):
    cfg = hsw.Sha256DynamicConfig(eng, sizes, is_input_range_check=rc, whole_digest=True)
    if repr_flag:
        cfg.set_repr(repr_flag)
    pre = pre or [None] * len(msgs)
    res = cfg.digest_batch(msgs, pre) if batch else [cfg.digest(m, p) for m, p in zip(msgs, pre)]
    st = cfg.streams()
    v = cfg.view()
    cfg.close()
    return res, st, v


@pytest.mark.parametrize("msgs,sizes,pre,rc,batch", CASES)
def test_whole_digest_stream_matches_oracle(hsw, oracle, eng_int, msgs, sizes, pre, rc, batch):
    res, st, v = _run(hsw, eng_int, msgs, sizes, pre, rc, batch)
    ref = oracle.digest_cells(msgs, sizes, pre, rc, record=True)
    for m, r, d in zip(msgs, res, ref["digests"]):
        assert r.output_bytes == hashlib.sha256(m).digest() == d
    assert int(v.gate_cells) == len(ref["gate"]) == int(v.gate_capacity)
    assert int(v.lookup_cells) == len(ref["lookup"]) == int(v.lookup_capacity)
    # sections where the oracle has them
    for r, lay in zip(res, ref["layouts"]):
        assert r.prologue_cell == lay["gate0"]
        assert r.block_cell == lay["gate0"] + lay["prologue_cells"] + lay["zero_cells"]
        assert r.epilogue_cell == r.block_cell + lay["block_cells"]
        assert r.end_cell == r.epilogue_cell + lay["epilogue_cells"]
        assert r.prologue_lookup == lay["lookup0"]
        assert r.block_lookup == lay["lookup0"] + lay["prologue_lookups"]
        assert r.epilogue_lookup == r.block_lookup + lay["block_lookups"]
    bad = np.nonzero((st["gate"] != ref["gate"]).any(axis=1))[0]
    assert len(bad) == 0, "first differing gate cells: %s" % bad[:8]
    assert np.array_equal(st["lookup"], ref["lookup"])
    assert np.array_equal(st["dense"], ref["dense"][:, : st["rows"]])
    assert np.array_equal(st["spread"], ref["spread"][:, : st["rows"]])
    # and without trusting any oracle VALUE: the GPU stream satisfies the recorded structure
    n = check_whole_stream(ref, st["gate"], st["lookup"], st["dense"], st["spread"], hsw._native.spread_table(8))
    assert n > 80000 * sum(s // 64 for s in sizes)
    # ... and carries the public facts of these inputs
    g = st["gate"]
    for m, r, p in zip(msgs, res, pre or [0] * len(msgs)):
        assert int(g[r.prologue_cell, 0]) == len(m)                                  # AssignedHashResult.input_len
        b0 = r.prologue_cell + 46
        assert bytes(g[b0: b0 + r.n_blocks * 64, 0].astype(np.uint8)) == r.input_bytes
        out0 = r.epilogue_cell + 76 * (r.n_blocks + 1)
        digest_cells = [out0 + 36 * w + 5 * i for w in range(8) for i in range(4)]   # AssignedHashResult.output_bytes
        assert bytes(g[digest_cells, 0].astype(np.uint8)) == hashlib.sha256(m).digest()


def test_whole_digest_reference_kats(hsw, oracle, eng_int):
    """Every digest-level vector the reference's tests hold, through the whole-digest path,
    two per context like TestCircuit (lib.rs:496-611)."""
    vecs = KATS["vectors"]
    for i in range(0, len(vecs), 2):
        pair = [vecs[i], vecs[(i + 1) % len(vecs)]]
        msgs = [bytes.fromhex(v["input_hex"]) for v in pair]
        pres = [v.get("precomputed_input_len", 0) for v in pair]
        sizes = [max(128, ((len(m) + 9 + 63) // 64) * 64 - p) for m, p in zip(msgs, pres)]
        res, st, _ = _run(hsw, eng_int, msgs, sizes, pres, True, True)
        for v, r in zip(pair, res):
            assert r.output_bytes.hex() == v["digest_hex"]
        ref = oracle.digest_cells(msgs, sizes, pres, True)
        assert np.array_equal(st["gate"], ref["gate"]) and np.array_equal(st["lookup"], ref["lookup"])


def test_whole_digest_montgomery(hsw, oracle, eng_int):
    N = hsw._native
    msgs, sizes = [b"abc", bytes(range(100))], [64, 192]
    res, st, _ = _run(hsw, eng_int, msgs, sizes, None, True, True, N.HSW_REPR_MONTGOMERY)
    ref = oracle.digest_cells(msgs, sizes, None, True)
    assert np.array_equal(st["gate"], oracle.to_montgomery(ref["gate"]))
    assert np.array_equal(st["lookup"], oracle.to_montgomery(ref["lookup"]))
    assert np.array_equal(st["dense"], oracle.to_montgomery(ref["dense"])[:, : st["rows"]])
    for m, r in zip(msgs, res):
        assert r.output_bytes == hashlib.sha256(m).digest()


def test_whole_digest_bench_circuit(hsw, oracle, eng_int):
    """benches/digest.rs:103-129: one 56-byte message, max 1024 B (16 blocks), input range checks:
    1,116,315 gate cells + 53,059 lookup cells, in 9 columns' worth of rows."""
    msg = bytes([1] * 56)
    res, st, v = _run(hsw, eng_int, [msg], [1024], None, True, True)
    assert int(v.gate_cells) == 1116315 and int(v.lookup_cells) == 53059
    ref = oracle.digest_cells([msg], [1024], None, True)
    assert np.array_equal(st["gate"], ref["gate"]) and np.array_equal(st["lookup"], ref["lookup"])
    assert res[0].output_bytes == hashlib.sha256(msg).digest()


def test_whole_digest_many_small_digests(hsw, oracle, eng_int):
    """512 single-block digests as one framed launch + one frame launch (batch path)."""
    rng = np.random.default_rng(0xF4)
    msgs = [rng.integers(0, 256, int(rng.integers(0, 56)), dtype=np.uint8).tobytes() for _ in range(512)]
    res, st, v = _run(hsw, eng_int, msgs, [64] * 512, None, False, True)
    for m, r in zip(msgs, res):
        assert r.output_bytes == hashlib.sha256(m).digest()
    ref = oracle.digest_cells(msgs[:3], [64] * 3, None, False)
    n3 = len(ref["gate"])
    assert np.array_equal(st["gate"][:n3], ref["gate"])
    # every digest's stream is the same function of its message: compare digest #400 with a fresh context
    one = oracle.digest_cells([msgs[400]], [64], None, False)
    r = res[400]
    a = st["gate"][r.prologue_cell: r.end_cell]
    b = np.delete(one["gate"], 110, axis=0)            # a fresh context has its zero cell after the 110 prologue cells
    # chip-independent comparison: gate cells do not depend on the cursor
    assert np.array_equal(a, b)


def test_frame_argument_errors(hsw, eng_int, engine_factory):
    N = hsw._native
    L = eng_int.lib
    d = N.FrameDesc()
    d.input_len, d.n_blocks, d.num_round, d.precomputed_round = 3, 1, 2, 0          # num_round must be 1
    d.zero_cell = N.NO_CELL
    import torch
    buf = torch.zeros(1 << 16, dtype=torch.int64, device="cuda")
    p = buf.data_ptr()
    assert L.hsw_witness_frames(eng_int.h, C.byref(d), 1, p, p, p, p, p, None, 0) == N.HSW_ERR_INVALID_ARG
    d.num_round, d.n_blocks = 1, 0
    assert L.hsw_witness_frames(eng_int.h, C.byref(d), 1, p, p, p, p, p, None, 0) == N.HSW_ERR_UNSUPPORTED
    d.n_blocks, d.input_len, d.num_round = 1, 100, 2                                 # needs 2 blocks, max is 1
    assert L.hsw_witness_frames(eng_int.h, C.byref(d), 1, p, p, p, p, p, None, 0) == N.HSW_ERR_TOO_LARGE
    rep = N.VerifyReport()                                                            # the verifier refuses it too
    assert L.hsw_verify_frames(eng_int.h, C.byref(d), 1, p, p, p, p, p, None, 0, C.byref(rep)) == N.HSW_ERR_INVALID_ARG
    d.input_len, d.num_round = 3, 1
    assert L.hsw_witness_frames(eng_int.h, C.byref(d), 1, p, p, p, p, p, None, N.HSW_REPR_COMPACT64) == N.HSW_ERR_UNSUPPORTED
    plain = engine_factory(8, 2)                                                     # not in internals mode
    assert L.hsw_witness_frames(plain.h, C.byref(d), 1, p, p, p, p, p, None, 0) == N.HSW_ERR_INVALID_ARG
    with pytest.raises(N.HswError):
        hsw.Sha256DynamicConfig(plain, [64], whole_digest=True)
    cfg = hsw.Sha256DynamicConfig(eng_int, [64], whole_digest=True)
    with pytest.raises(N.HswError):
        cfg.set_repr(N.HSW_REPR_COMPACT64)
    cfg.close()


def _model_columns(call_lens, gate, max_rows):
    """halo2-lib v0.2.x FlexGate::assign_region over the oracle's call tape (A3-iii):
    `if row + len >= max_rows { column += 1; row = 0 }`.  Returns the column image."""
    cols, col, row, pos = [np.zeros((max_rows, 4), dtype=np.uint64)], 0, 0, 0
    for ln in call_lens.tolist():
        if row + ln >= max_rows:
            cols.append(np.zeros((max_rows, 4), dtype=np.uint64))
            col, row = col + 1, 0
        cols[col][row:row + ln] = gate[pos:pos + ln]
        row += ln
        pos += ln
    assert pos == len(gate)
    return np.stack(cols), (col, row)


@pytest.mark.parametrize("msgs,sizes,rc,max_rows,want_cols", [
    ([b"abc", b""], [128, 128], True, (1 << 17) - 9, 3),           # TestCircuit: NUM_ADVICE = 3 (lib.rs:487-493)
    ([bytes([1] * 56)], [1024], True, (1 << 17) - 9, 9),           # bench circuit: NUM_ADVICE = 9 (benches/digest.rs:103-108)
    ([b"abc", b"de", b"f" * 60], [64, 64, 128], False, 69500, None),   # a break in (nearly) every block
])
def test_whole_digest_as_flexgate_columns(hsw, oracle, eng_int, msgs, sizes, rc, max_rows, want_cols):
    """f2 + f4 together: the gadget writes the literal FlexGate advice-column image of the whole
    region (every assign_region call kept inside one column, unassigned tail rows zero)."""
    cfg = hsw.Sha256DynamicConfig(eng_int, sizes, is_input_range_check=rc, whole_digest=True)
    ncols = cfg.set_columns(max_rows)
    assert want_cols is None or ncols == want_cols
    res = cfg.digest_batch(msgs, [None] * len(msgs))
    st = cfg.streams()
    ref = oracle.digest_cells(msgs, sizes, None, rc)
    img, (last_col, end_row) = _model_columns(ref["call_lens"], ref["gate"], max_rows)
    assert img.shape[0] == ncols == st["gate"].shape[0]
    bad = np.nonzero((st["gate"] != img).any(axis=2))
    assert len(bad[0]) == 0, "first differing (column, row): %s" % [(int(c), int(r)) for c, r in zip(*bad)][:6]
    assert np.array_equal(st["lookup"], ref["lookup"])
    # cell_position is the same map
    r = res[-1]
    assert cfg.cell_position(r.end_cell - 1) == (last_col, end_row - 1)
    assert cfg.cell_position(0) == (0, 0)
    for m, rr in zip(msgs, res):
        assert rr.output_bytes == hashlib.sha256(m).digest()
    with pytest.raises(hsw.HswError):
        cfg.set_columns(max_rows)                  # only before the first digest
    cfg.close()


def test_set_columns_errors(hsw, eng_int):
    cfg = hsw.Sha256DynamicConfig(eng_int, [64] * 40, whole_digest=True)
    with pytest.raises(hsw.HswError):
        cfg.set_columns(1000)                      # a block must fit two columns
    with pytest.raises(hsw.HswError):
        cfg.set_columns(69348 + 16)                # > HSW_MAX_BREAKS + 1 columns
    assert cfg.set_columns(1 << 20) == 3
    cfg.close()


def test_gadget_reset_is_a_new_synthesis_pass(hsw, oracle, eng_int):
    """lib.rs:440 / benches/digest.rs:78: the harnesses clone the config per synthesis to reset
    cur_hash_idx and the SpreadConfig cursors; hsw_gadget_reset does that on the same buffers."""
    cfg = hsw.Sha256DynamicConfig(eng_int, [128, 64], is_input_range_check=True, whole_digest=True)
    cfg.set_columns(150000)
    cfg.digest_batch([b"first pass", b"x"], [None, None])
    a = cfg.streams()
    cfg.reset()
    assert cfg.view().cur_hash_idx == 0 and cfg.view().num_limb_sum == 0 and cfg.view().gate_cells == 0
    res = cfg.digest_batch([b"second pass, other data", b"y"], [None, None])
    b = cfg.streams()
    ref = oracle.digest_cells([b"second pass, other data", b"y"], [128, 64], None, True)
    img, _ = _model_columns(ref["call_lens"], ref["gate"], 150000)
    assert np.array_equal(b["gate"], img) and not np.array_equal(a["gate"], b["gate"])
    assert np.array_equal(b["lookup"], ref["lookup"])
    assert np.array_equal(b["dense"], ref["dense"][:, : b["rows"]])
    assert res[0].output_bytes == hashlib.sha256(b"second pass, other data").digest()
    cfg.close()


def test_whole_digest_randomized_contexts(hsw, oracle, eng_int):
    """Random contexts: 1-4 digests of random maximum sizes, message lengths up to the maximum,
    random precomputed prefixes, both range-check settings, batched or one by one, optionally as a
    column image with a random max_rows -- always bit-exact against the oracle."""
    rng = np.random.default_rng(20261004)
    for trial in range(14):
        k = int(rng.integers(1, 5))
        sizes, msgs, pres = [], [], []
        for _ in range(k):
            nb = int(rng.integers(1, 5))
            pre_rounds = int(rng.integers(0, 3)) if rng.random() < 0.4 else 0
            # total padded rounds must satisfy pre_rounds <= num_round <= pre_rounds + nb
            num_round = int(rng.integers(max(pre_rounds, 1), pre_rounds + nb + 1))
            ln = int(rng.integers(max(0, 64 * (num_round - 1) - 8), 64 * num_round - 8))   # ceil((ln+9)/64) == num_round
            sizes.append(64 * nb)
            msgs.append(rng.integers(0, 256, ln, dtype=np.uint8).tobytes())
            pres.append(64 * pre_rounds)
        rc = bool(rng.integers(0, 2))
        batch = bool(rng.integers(0, 2))
        max_rows = int(rng.integers(69348 + 16, 200000)) if rng.random() < 0.5 else None
        cfg = hsw.Sha256DynamicConfig(eng_int, sizes, is_input_range_check=rc, whole_digest=True)
        try:
            if max_rows is not None:
                try:
                    cfg.set_columns(max_rows)
                except hsw.HswError:
                    max_rows = None                      # > 17 columns: keep the linear stream
            res = cfg.digest_batch(msgs, pres) if batch else [cfg.digest(m, p) for m, p in zip(msgs, pres)]
            st = cfg.streams()
        finally:
            cfg.close()
        ref = oracle.digest_cells(msgs, sizes, pres, rc)
        for m, r in zip(msgs, res):
            assert r.output_bytes == hashlib.sha256(m).digest(), (trial, len(m))
        exp = _model_columns(ref["call_lens"], ref["gate"], max_rows)[0] if max_rows else ref["gate"]
        assert np.array_equal(st["gate"], exp), (trial, sizes, [len(m) for m in msgs], pres, rc, batch, max_rows)
        assert np.array_equal(st["lookup"], ref["lookup"]), trial
        assert np.array_equal(st["dense"], ref["dense"][:, : st["rows"]]), trial


@pytest.mark.parametrize("columns", [False, True])
def test_seek_deals_one_circuit_to_two_gadgets(hsw, oracle, eng_int, columns):
    """hsw_gadget_seek: two gadgets (two GPUs in production) with the same configuration assign
    disjoint digests of one circuit into the positions the serial reference would use; the union of
    their images is the single-gadget image.  No exchange is needed: positions depend on
    max_variable_byte_sizes only."""
    sizes = [128, 64, 192, 64]
    msgs = [b"a" * 100, b"bc", bytes(range(150)), b""]
    max_rows = 140000 if columns else None

    def run(first, last):
        cfg = hsw.Sha256DynamicConfig(eng_int, sizes, is_input_range_check=True, whole_digest=True)
        if max_rows:
            cfg.set_columns(max_rows)
        if first:
            cfg.seek(first)
        res = cfg.digest_batch(msgs[first:last], [None] * (last - first))
        assert cfg.verify()["violations"] == 0            # also right after a seek: only what this gadget wrote
        # the gadget's view only covers what it has written; read the whole buffers
        v = cfg.view()
        n_gate = int(v.max_rows * v.columns) if max_rows else int(v.gate_capacity)

        def grab(ptr, n_cells):
            a = np.zeros((n_cells, 4), dtype=np.uint64)
            cfg._ok(cfg.lib.hsw_download(eng_int.h, a.ctypes.data, ptr, n_cells * 32))
            return a
        out = dict(gate=grab(v.d_gate, n_gate), lookup=grab(v.d_lookup, int(v.lookup_capacity)),
                   dense=grab(v.d_chip_dense, 2 * int(v.chip_col_stride)), res=res,
                   end=(int(v.gate_cells), int(v.lookup_cells), int(v.num_limb_sum), int(v.cur_hash_idx)))
        cfg.close()
        return out

    full = run(0, 4)
    a, b = run(0, 2), run(2, 4)
    assert b["end"] == full["end"]                                   # the cursors after the last digest agree
    ref = oracle.digest_cells(msgs, sizes, None, True)
    g_end = len(ref["gate"])
    if not columns:
        # linear stream: hipMalloc'd, so only the written ranges are defined -- compare those
        cut = a["end"][0]
        assert np.array_equal(a["gate"][:cut], ref["gate"][:cut])
        assert np.array_equal(b["gate"][cut:g_end], ref["gate"][cut:g_end])
    else:
        # column image: zero-initialised, the two halves are disjoint
        assert np.array_equal(a["gate"] | b["gate"], full["gate"])
        assert not (a["gate"].any(axis=1) & b["gate"].any(axis=1)).any()
    lc = a["end"][1]
    assert np.array_equal(a["lookup"][:lc], ref["lookup"][:lc]) and np.array_equal(b["lookup"][lc:], ref["lookup"][lc:])
    assert np.array_equal(a["dense"] | b["dense"], full["dense"])
    for m, r in zip(msgs[2:], b["res"]):
        assert r.output_bytes == hashlib.sha256(m).digest()
    assert [r.prologue_cell for r in b["res"]] == [r.prologue_cell for r in full["res"][2:]]


@pytest.mark.parametrize("columns", [False, True])
def test_download_region_to_host(hsw, oracle, eng_int, columns):
    """hsw_gadget_download_region: the region in host memory (pinned), used rows only, equal to the
    device image."""
    sizes, msgs = [128, 64], [b"host delivery", b"z"]
    cfg = hsw.Sha256DynamicConfig(eng_int, sizes, is_input_range_check=True, whole_digest=True)
    if columns:
        cfg.set_columns(100000)
    cfg.digest_batch(msgs, [None, None])
    dev = cfg.streams()
    for pinned in (True, False):
        host = cfg.download_region(pinned=pinned)
        assert np.array_equal(host["gate"], dev["gate"])
        assert np.array_equal(host["lookup"], dev["lookup"])
        assert np.array_equal(host["dense"], dev["dense"]) and np.array_equal(host["spread"], dev["spread"])
    ref = oracle.digest_cells(msgs, sizes, None, True)
    exp = _model_columns(ref["call_lens"], ref["gate"], 100000)[0] if columns else ref["gate"]
    assert np.array_equal(host["gate"], exp)
    cfg.close()
    # block-stream contexts deliver too
    eng = hsw.WitnessEngine(0, 8, 2)
    plain = hsw.Sha256DynamicConfig(eng, [128], True)
    plain.digest(b"abc")
    d, h = plain.streams(), plain.download_region()
    assert np.array_equal(h["gate"], d["gate"]) and np.array_equal(h["dense"], d["dense"]) and "lookup" not in h
    plain.close()
    eng.close()


def test_whole_region_montgomery_column_image(hsw, oracle, eng_int):
    """What a Rust shim would take: the column image in Montgomery form (halo2curves' in-memory Fr) --
    TestCircuit shape, 3 columns -- equals the oracle's canonical cells times 2^256 mod p, gaps zero."""
    N = hsw._native
    msgs, sizes = [b"abc", b""], [128, 128]
    cfg = hsw.Sha256DynamicConfig(eng_int, sizes, is_input_range_check=True, whole_digest=True)
    cfg.set_repr(N.HSW_REPR_MONTGOMERY)
    assert cfg.set_columns((1 << 17) - 9) == 3
    res = cfg.digest_batch(msgs, [None, None])
    st = cfg.streams()
    host = cfg.download_region()
    cfg.close()
    ref = oracle.digest_cells(msgs, sizes, None, True)
    img, _ = _model_columns(ref["call_lens"], oracle.to_montgomery(ref["gate"]), (1 << 17) - 9)
    assert np.array_equal(st["gate"], img) and np.array_equal(host["gate"], img)
    assert np.array_equal(st["lookup"], oracle.to_montgomery(ref["lookup"]))
    assert np.array_equal(st["dense"], oracle.to_montgomery(ref["dense"])[:, : st["rows"]])
    for m, r in zip(msgs, res):
        assert r.output_bytes == hashlib.sha256(m).digest()


@pytest.mark.parametrize("ncols", [1, 3])
def test_whole_digest_other_spread_column_counts(hsw, oracle, ncols):
    """SpreadConfig with 1 or 3 advice column pairs (spread.rs:25): the chip cursor wraps differently,
    the gate / lookup streams do not change."""
    N = hsw._native
    eng = hsw.WitnessEngine(0, 8, ncols, mode=N.HSW_MODE_HALO2_INTERNALS)
    msgs, sizes = [b"abc", bytes(range(90))], [64, 128]
    cfg = hsw.Sha256DynamicConfig(eng, sizes, is_input_range_check=True, whole_digest=True)
    res = cfg.digest_batch(msgs, [None, None])
    assert cfg.verify()["violations"] == 0
    st = cfg.streams()
    cfg.close()
    eng.close()
    ref = oracle.digest_cells(msgs, sizes, None, True, num_advice_columns=ncols)
    assert np.array_equal(st["gate"], ref["gate"]) and np.array_equal(st["lookup"], ref["lookup"])
    assert np.array_equal(st["dense"], ref["dense"][:, : st["rows"]]) and np.array_equal(st["spread"], ref["spread"][:, : st["rows"]])
    for m, r in zip(msgs, res):
        assert r.output_bytes == hashlib.sha256(m).digest()


@pytest.mark.parametrize("sizes", [[7680], [1024] * 7 + [512]], ids=["1x120blocks", "8digests"])
@pytest.mark.parametrize("mont", [False, True], ids=["canonical", "montgomery"])
def test_k20_region_of_120_blocks(hsw, oracle, eng_int, sizes, mont):
    """BASELINE configs[4] substitute: the region of a k = 20 circuit -- max_rows = 2^20 - 9
    (lib.rs:351-360), ~120 blocks at 9 advice columns, input range checks on (create_proof itself needs the
    Rust prover: not runnable here).  One 120-block digest, and eight digests of two sizes in one context:
    the linear stream + lookup + chip columns and the FlexGate column image, canonical and Montgomery,
    against the oracle and a Python model of assign_region, and verified on the device."""
    N = hsw._native
    max_rows = (1 << 20) - 9
    msgs = [bytes(((i * 7 + 3 * k) % 256) for i in range(s - 9 - 5 * k)) for k, s in enumerate(sizes)]
    assert sum(sizes) // 64 == 120
    ref = oracle.digest_cells(msgs, sizes, None, True)
    conv = oracle.to_montgomery if mont else (lambda x: x)
    img, (last_col, end_row) = _model_columns(ref["call_lens"], conv(ref["gate"]), max_rows)
    for columns in (False, True):
        cfg = hsw.Sha256DynamicConfig(eng_int, sizes, is_input_range_check=True, whole_digest=True)
        if mont:
            cfg.set_repr(N.HSW_REPR_MONTGOMERY)
        if columns:
            assert cfg.set_columns(max_rows) == img.shape[0] == 8      # 8.4 M cells in 2^20-row columns (9 configured)
        res = cfg.digest_batch(msgs)
        assert [r.output_bytes for r in res] == [hashlib.sha256(m).digest() for m in msgs] == ref["digests"]
        rep = cfg.verify()
        assert rep["violations"] == 0 and rep["checks"] > 120 * 69348 // 4, rep
        st = cfg.streams()
        if columns:
            bad = np.nonzero((st["gate"] != img).any(axis=2))
            assert len(bad[0]) == 0, "first differing (column, row): %s" % [(int(c), int(r)) for c, r in zip(*bad)][:6]
            assert cfg.cell_position(res[-1].end_cell - 1) == (last_col, end_row - 1)
            host = cfg.download_region(pinned=True)
            assert np.array_equal(host["gate"], img) and np.array_equal(host["lookup"], conv(ref["lookup"]))
        else:
            assert np.array_equal(st["gate"], conv(ref["gate"]))
        assert np.array_equal(st["lookup"], conv(ref["lookup"]))
        assert np.array_equal(st["dense"], conv(ref["dense"])[:, : st["rows"]])
        assert np.array_equal(st["spread"], conv(ref["spread"])[:, : st["rows"]])
        cfg.close()


@pytest.mark.parametrize("columns", [False, True])
def test_compact_delivery_of_a_whole_region(hsw, oracle, eng_int, columns):
    """hsw_gadget_download_region_compact: the region crosses PCIe as 8-byte cells plus a side list of the cells
    wider than 64 bits (ch negations, -2^16, negative differences, is_zero inverses); widened on the host
    (hsw_region_widen) it is the 32-byte image bit for bit.  TestCircuit shape and a three-digest context with a
    target round below the maximum (negative n - target, inverses)."""
    N = hsw._native
    for sizes, msgs in (([128, 128], [b"abc", b""]), ([256, 64, 192], [b"q" * 70, b"", bytes(range(100))])):
        cfg = hsw.Sha256DynamicConfig(eng_int, sizes, is_input_range_check=True, whole_digest=True)
        if columns:
            cfg.set_columns(100000)
        cfg.digest_batch(msgs)
        full = cfg.download_region(pinned=False)
        bufs, n_wide = cfg.download_region_compact()
        bufs, n_wide2 = cfg.download_region_compact(bufs)            # buffers reusable, list rebuilt from scratch
        assert n_wide == n_wide2
        nblk = sum(sizes) // 64
        assert 0 < n_wide <= 256 * nblk + 64 * len(sizes)
        wide = bufs["wide"]
        v = cfg.view()
        g = cfg.widen(bufs["gate"], N.HSW_STREAM_GATE if hasattr(N, "HSW_STREAM_GATE") else 0, wide, n_wide)
        assert np.array_equal(g.reshape(full["gate"].shape), full["gate"])
        lk = cfg.widen(bufs["lookup"][: int(v.lookup_cells)], 1, wide, n_wide)
        assert np.array_equal(lk, full["lookup"])
        stride = int(v.chip_col_stride)
        d = cfg.widen(bufs["dense"], 2, wide, n_wide).reshape(2, stride, 4)[:, : full["rows"]]
        s = cfg.widen(bufs["spread"], 3, wide, n_wide).reshape(2, stride, 4)[:, : full["rows"]]
        assert np.array_equal(d, full["dense"]) and np.array_equal(s, full["spread"])
        # the side list holds exactly the cells with a non-zero upper limb
        flat = full["gate"].reshape(-1, 4)
        n_expected = int(flat[:, 1:].any(axis=1).sum())
        assert n_wide == n_expected
        # too small a side list is reported, with the size needed
        small = N.RegionCompact(bufs["gate"].ctypes.data, None, None, None, bufs["wide"].ctypes.data, 3, 0)
        assert cfg.lib.hsw_gadget_download_region_compact(cfg.h, C.byref(small)) == N.HSW_ERR_TOO_LARGE and small.n_wide == n_wide
        cfg.set_repr(N.HSW_REPR_MONTGOMERY)
        with pytest.raises(hsw.HswError):
            cfg.download_region_compact(bufs)
        cfg.close()


@pytest.mark.parametrize("k,mont", [(8, False), (8, True), (9, True)], ids=["K8", "K8-montgomery", "K9-two-launch-path"])
def test_k_independent_syntheses_of_the_bench_circuit_in_one_batch(hsw, oracle, eng_int, k, mont):
    """HSW_GADGET_INDEPENDENT: K proofs of the reference's bench circuit (one 16-block digest each,
    benches/digest.rs:93-129) in flight -- one hsw_gadget_digest_batch call (K = 8: 128 blocks + 8 frames in ONE
    launch), every digest a Context of its own: its slice of the streams is cell for cell what a fresh
    single-digest gadget writes (own zero cell, lookup entries and chip rows from 0)."""
    N = hsw._native
    rng = np.random.default_rng(0xB0 + k)
    msgs = [bytes([1] * 56)] + [rng.integers(0, 256, int(rng.integers(0, 1016)), dtype=np.uint8).tobytes() for _ in range(k - 1)]
    cfg = hsw.Sha256DynamicConfig(eng_int, [1024] * k, is_input_range_check=True, whole_digest=True, independent=True)
    if mont:
        cfg.set_repr(N.HSW_REPR_MONTGOMERY)
    with pytest.raises(hsw.HswError):
        cfg.set_columns((1 << 17) - 9)                       # K regions in one stream: linear only
    res = cfg.digest_batch(msgs)
    assert eng_int.last_launch()["split"] == (2 if k * 16 <= 128 else 0)
    rep = cfg.verify()
    assert rep["violations"] == 0, rep
    st = cfg.streams()
    v = cfg.view()
    assert int(v.gate_cells) == k * 1116315 == int(v.gate_capacity) and int(v.lookup_cells) == k * 53059
    conv = oracle.to_montgomery if mont else (lambda x: x)
    rows_per = 16 * 4120 // 2
    for h in (0, 1, k - 1):
        one = oracle.digest_cells([msgs[h]], [1024], None, True)
        r = res[h]
        assert r.output_bytes == hashlib.sha256(msgs[h]).digest()
        assert r.end_cell - r.prologue_cell == len(one["gate"]) == 1116315 and r.prologue_cell == h * 1116315
        assert r.block_cell == r.prologue_cell + one["layouts"][0]["prologue_cells"] + 1      # its own zero cell
        assert np.array_equal(st["gate"][r.prologue_cell: r.end_cell], conv(one["gate"]))
        assert np.array_equal(st["lookup"][r.prologue_lookup: r.prologue_lookup + 53059], conv(one["lookup"]))
        row0 = r.first_block * 4120 // 2
        assert np.array_equal(st["dense"][:, row0: row0 + rows_per], conv(one["dense"]))
        assert np.array_equal(st["spread"][:, row0: row0 + rows_per], conv(one["spread"]))
    cfg.reset()
    res2 = cfg.digest_batch(msgs[::-1])                      # next round of proofs on the same buffers
    assert [r.output_bytes for r in res2] == [hashlib.sha256(m).digest() for m in msgs[::-1]]
    cfg.close()
