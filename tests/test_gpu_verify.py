"""hsw_verify_blocks: the product's own on-device MockProver-style check (gate rows, copy constraints,
constants, ranges, chip cells + spread table, lookup copies, next states) -- accepts what the kernels
write, and pins down single corrupted cells of every constraint class."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _inputs(n, seed):
    rng = np.random.default_rng(seed)
    return (rng.integers(0, 256, (n, 64), dtype=np.uint8),
            rng.integers(0, 2**32, (n, 8), dtype=np.uint64).astype(np.uint32))


def _gen(eng, n, seed, cursor0=0):
    import torch
    blocks, pre = _inputs(n, seed)
    tb, tp = torch.from_numpy(blocks).cuda(), torch.from_numpy(pre.view(np.int32)).cuda()
    out = eng.witness_blocks(tb, tp, cursor0=cursor0)
    eng.synchronize()
    return tb, tp, out


def test_verify_accepts_generated_streams(engine_factory, hsw):
    eng = engine_factory(8, 2)
    tb, tp, out = _gen(eng, 6, 11, cursor0=4)
    rep = eng.verify_blocks(tb, tp, out, cursor0=4)
    # per block: 13,638 constants + 30,550 - 3,850 Existing copies ... counted by the library; at least these
    assert rep["violations"] == 0 and rep["checks"] >= 6 * (13510 + 3850 + 2424 + 4 * 4120 + 8 + 40000)
    # chip / next-state checks are optional
    rep2 = eng.verify_blocks(tb, tp, out, cursor0=4, check_chip=False, check_next=False)
    assert rep2["violations"] == 0 and rep2["checks"] == rep["checks"] - 6 * (4 * 4120 + 8)


@pytest.mark.parametrize("bits,ncols", [(16, 1), (4, 3), (8, 5)])
def test_verify_other_shapes(engine_factory, bits, ncols):
    eng = engine_factory(bits, ncols)
    tb, tp, out = _gen(eng, 3, 5 + bits, cursor0=7)
    assert eng.verify_blocks(tb, tp, out, cursor0=7)["violations"] == 0


def test_verify_pins_single_corruptions(engine_factory, oracle, hsw):
    """Flip one cell of each constraint class; the report names block, cell and class."""
    import torch
    eng = engine_factory(8, 2)
    tb, tp, out = _gen(eng, 4, 99)
    st = hsw._native.block_structure(eng.shape)
    G = eng.G
    kind, ref = st["kind"], st["ref"]
    const_cell = int(np.nonzero(kind == 1)[0][100])
    copy_cell = int(np.nonzero((kind == 2) & (ref >= 0))[0][500])
    row = int(st["gate_rows"][7000])
    witness_out = row + 3                                   # the output of a gate row: a pure witness
    range_cell = int(st["range"][50][0])
    cases = [(2, const_cell, "constant"), (1, copy_cell, "copy"), (3, witness_out, None), (0, range_cell, None)]
    for blk, cell, want in cases:
        g = out["gate"]
        saved = g[blk * G + cell].clone()
        g[blk * G + cell, 0] += 1
        rep = eng.verify_blocks(tb, tp, out)
        assert rep["violations"] >= 1, (blk, cell)
        assert rep["first_block"] == blk
        if want:      # the earliest failing cell: the cell itself, or the start of the gate row it sits in
            assert rep["first_class"] in (want, "gate row") and cell - 3 <= rep["first_cell"] <= cell
        g[blk * G + cell] = saved
        assert eng.verify_blocks(tb, tp, out)["violations"] == 0
    # a chip cell, a next-state word, an input byte, a pre-state word
    d = out["dense"]
    saved = d[1, 37].clone(); d[1, 37, 0] ^= 1
    rep = eng.verify_blocks(tb, tp, out)
    assert rep["violations"] >= 1 and rep["first_class"] == "chip"
    d[1, 37] = saved
    ns = out["next_states"]
    ns[2, 3] += 1
    rep = eng.verify_blocks(tb, tp, out)
    assert rep["violations"] == 1 and rep["first_class"] == "next state" and rep["first_block"] == 2
    ns[2, 3] -= 1
    tb2 = tb.clone(); tb2[1, 17] ^= 0x40
    rep = eng.verify_blocks(tb2, tp, out)
    assert rep["violations"] >= 1 and rep["first_block"] == 1 and rep["first_class"] in ("copy", "gate row")
    tp2 = tp.clone(); tp2[3, 5] ^= 1
    rep = eng.verify_blocks(tb, tp2, out)
    assert rep["violations"] >= 1 and rep["first_block"] == 3
    assert eng.verify_blocks(tb, tp, out)["violations"] == 0


def test_verify_internals_mode_with_lookup_column(hsw):
    import torch
    N = hsw._native
    eng = hsw.WitnessEngine(0, 8, 2, mode=N.HSW_MODE_HALO2_INTERNALS)
    blocks, pre = _inputs(5, 3)
    tb, tp = torch.from_numpy(blocks).cuda(), torch.from_numpy(pre.view(np.int32)).cuda()
    out = eng.witness_blocks_ex(tb, tp, cursor0=2, want_lookup=True)
    eng.synchronize()
    rep = eng.verify_blocks(tb, tp, out, cursor0=2, lookup=out["lookup"])
    assert rep["violations"] == 0
    lk = out["lookup"]
    lk[3 * eng.lookup_cells + 1000, 0] += 1
    rep = eng.verify_blocks(tb, tp, out, cursor0=2, lookup=lk)
    assert rep["violations"] == 1 and rep["first_class"] == "lookup" and rep["first_block"] == 3 and rep["first_cell"] == 1000
    eng.close()


def test_verify_rejects_what_it_cannot_check(engine_factory, hsw):
    import ctypes as C
    N = hsw._native
    eng = engine_factory(8, 2)
    tb, tp, out = _gen(eng, 1, 1)
    a = N.WitnessArgs()
    a.d_blocks, a.d_pre_states, a.n_blocks, a.d_gate = tb.data_ptr(), tp.data_ptr(), 1, out["gate"].data_ptr()
    rep = N.VerifyReport()
    a.flags = N.HSW_REPR_COMPACT64                # 8-byte cells cannot hold the negations
    assert eng.lib.hsw_verify_blocks(eng.h, C.byref(a), C.byref(rep)) == N.HSW_ERR_UNSUPPORTED
    a.flags = 0
    a.frame_every = 1                             # digest frames exist in internals mode only
    assert eng.lib.hsw_verify_blocks(eng.h, C.byref(a), C.byref(rep)) == N.HSW_ERR_INVALID_ARG
    a.frame_every = 0
    a.d_lookup = out["gate"].data_ptr()
    assert eng.lib.hsw_verify_blocks(eng.h, C.byref(a), C.byref(rep)) == N.HSW_ERR_INVALID_ARG
    a.d_lookup = None
    assert eng.lib.hsw_verify_blocks(eng.h, C.byref(a), C.byref(rep)) == 0 and rep.violations == 0


def test_verify_full_batch(engine_factory):
    """configs[2]: all 4,096 blocks (9.77 GB of cells), checked in HBM in a few milliseconds."""
    eng = engine_factory(8, 2)
    tb, tp, out = _gen(eng, 4096, 0xC3)
    rep = eng.verify_blocks(tb, tp, out)
    assert rep["violations"] == 0 and rep["checks"] > 2.5e8
    assert rep["kernel_ms"] < 50
    out["gate"][4000 * eng.G + 12345, 0] ^= 1
    rep = eng.verify_blocks(tb, tp, out)
    assert rep["violations"] >= 1 and rep["first_block"] == 4000 and rep["first_cell"] in (12345, 12344, 12343, 12342)


@pytest.fixture(scope="module")
def eng_int(hsw):
    e = hsw.WitnessEngine(0, 8, 2, mode=hsw._native.HSW_MODE_HALO2_INTERNALS)
    yield e
    e.close()


@pytest.mark.parametrize("columns", [None, 100003])
def test_gadget_verify_whole_regions(hsw, eng_int, columns):
    """hsw_gadget_verify on whole-digest contexts: blocks + frames (full field arithmetic for the rows with
    full-width cells) + the links between prologue, blocks and epilogue; linear streams and column images."""
    sizes = [128, 128, 64, 192]
    msgs = [b"abc", b"", b"x" * 40, bytes(range(150))]
    cfg = hsw.Sha256DynamicConfig(eng_int, sizes, is_input_range_check=True, whole_digest=True)
    if columns:
        cfg.set_columns(columns)
    res = cfg.digest_batch(msgs[:3], [None] * 3) + [cfg.digest(msgs[3])]
    rep = cfg.verify()
    assert rep["violations"] == 0 and rep["checks"] > 8 * 80000
    v = cfg.view()
    import ctypes as C
    hip = C.CDLL("libamdhip64.so")
    hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]

    def poke(cell, delta=1):
        """add delta to limb 0 of stream cell `cell`, in place in the gadget's HBM buffer"""
        col, row = cfg.cell_position(cell)
        at = (col * int(v.max_rows) + row) if columns else cell
        buf = np.zeros(4, dtype=np.uint64)
        cfg._ok(cfg.lib.hsw_download(eng_int.h, buf.ctypes.data, v.d_gate + at * 32, 32))
        buf[0] = np.uint64((int(buf[0]) + delta) % (1 << 64))
        assert hip.hipMemcpy(C.c_void_p(v.d_gate + at * 32), C.c_void_p(buf.ctypes.data), 32, 1) == 0   # host -> device

    r1 = res[1]
    probes = [
        (r1.prologue_cell + 0, "copy"),                     # input_len
        (r1.prologue_cell + 18, None),                      # the -2^16 constant
        (r1.prologue_cell + 27, "free"),                    # inverse witness of is_zero(0): unconstrained
        (r1.prologue_cell + 46 + 5, None),                  # an input byte
        (r1.block_cell + 12345, None),                      # inside a block
        (r1.epilogue_cell + 6, None),                       # (n - target)^-1 of candidate 0
        (r1.epilogue_cell + 12 + 7, None),                  # a select output
        (r1.end_cell - 1, None),                            # last cell of the digest's recomposition
    ]
    for cell, want in probes:
        poke(cell, 1)
        rep = cfg.verify()
        if want == "free":
            assert rep["violations"] == 0, cell
        else:
            assert rep["violations"] >= 1, cell
        poke(cell, -1)
        assert cfg.verify()["violations"] == 0, cell
    cfg.close()


def test_gadget_verify_block_stream_contexts(engine_factory, hsw):
    eng = engine_factory(8, 2)
    cfg = hsw.Sha256DynamicConfig(eng, [1024, 64, 4096], True)
    cfg.digest(bytes(range(200)) * 4)            # 16 blocks: inputs read in place from pinned memory
    cfg.digest(b"q")                              # 1 block
    cfg.digest(b"long " * 700)                    # 64 blocks: copied
    rep = cfg.verify()
    assert rep["violations"] == 0 and rep["checks"] > 81 * 70000
    cfg.close()


def test_gadget_verify_after_a_representation_change(engine_factory, hsw):
    """set_repr between digests (same cell size): verify checks each batch in the form it was written in."""
    eng = engine_factory(8, 2)
    cfg = hsw.Sha256DynamicConfig(eng, [128, 128, 64], False)
    cfg.digest(b"canonical cells")
    cfg.set_repr(hsw._native.HSW_REPR_MONTGOMERY)
    cfg.digest(b"montgomery cells")
    cfg.set_repr(0)
    cfg.digest(b"canonical again")
    rep = cfg.verify()
    assert rep["violations"] == 0 and rep["checks"] > 5 * 70000
    cfg.close()


def test_verify_shard_of_configs3_and_beyond(engine_factory):
    """BASELINE configs[3]'s per-GPU shard is 8,192 blocks (19.5 GB); twice that (39 GB, byte offsets far
    past 2^32 and cell indices past 2^30) generated at an odd cursor and checked in HBM by the product's
    own verifier -- no host copy of the streams is ever made."""
    import hashlib
    import torch
    eng = engine_factory(8, 2)
    n = 16384
    rng = np.random.default_rng(0xC4)
    msgs = rng.integers(0, 256, (n, 55), dtype=np.uint8)
    blocks = np.zeros((n, 64), dtype=np.uint8)
    blocks[:, :55] = msgs
    blocks[:, 55] = 0x80
    blocks[:, 62], blocks[:, 63] = (55 * 8) >> 8, (55 * 8) & 0xFF
    iv = np.array([0x6a09e667, 0xbb67ae85, 0x3c6ef372, 0xa54ff53a, 0x510e527f, 0x9b05688c, 0x1f83d9ab, 0x5be0cd19], dtype=np.uint32)
    tb = torch.from_numpy(blocks).cuda()
    tp = torch.from_numpy(np.tile(iv, (n, 1)).view(np.int32)).cuda()
    cursor0 = 3 * 4120 + 1                       # odd: first / last chip rows are shared with neighbours
    out = eng.witness_blocks(tb, tp, cursor0=cursor0)
    eng.synchronize()
    rep = eng.verify_blocks(tb, tp, out, cursor0=cursor0)
    assert rep["violations"] == 0 and rep["checks"] > 1.2e9
    ns = out["next_states"].cpu().numpy().view(np.uint32)
    for i in (0, 1, 8191, 8192, n - 1):
        assert b"".join(int(x).to_bytes(4, "big") for x in ns[i]) == hashlib.sha256(msgs[i].tobytes()).digest()
    # a corruption in the very last block is found, at the right place
    out["gate"][(n - 1) * eng.G + 66000, 0] ^= 1
    rep = eng.verify_blocks(tb, tp, out, cursor0=cursor0)
    assert rep["violations"] >= 1 and rep["first_block"] == n - 1
    del out
    torch.cuda.empty_cache()


def test_verify_montgomery_streams(engine_factory, hsw, eng_int):
    """Montgomery cells are reduced on load and checked in the canonical domain: block streams and whole
    regions (column image), and a flipped cell is still found."""
    import ctypes as C
    import torch
    N = hsw._native
    eng = engine_factory(8, 2)
    blocks, pre = _inputs(5, 21)
    tb, tp = torch.from_numpy(blocks).cuda(), torch.from_numpy(pre.view(np.int32)).cuda()
    out = eng.witness_blocks(tb, tp, cursor0=6, flags=N.HSW_REPR_MONTGOMERY)
    eng.synchronize()
    a = N.WitnessArgs()
    a.d_blocks, a.d_pre_states, a.n_blocks, a.spread_cursor0 = tb.data_ptr(), tp.data_ptr(), 5, 6
    a.d_gate, a.d_chip_dense, a.d_chip_spread = out["gate"].data_ptr(), out["dense"].data_ptr(), out["spread"].data_ptr()
    a.chip_col_stride, a.d_next_states, a.flags = out["dense"].shape[1], out["next_states"].data_ptr(), N.HSW_REPR_MONTGOMERY
    rep = N.VerifyReport()
    assert eng.lib.hsw_verify_blocks(eng.h, C.byref(a), C.byref(rep)) == 0 and rep.violations == 0
    a.flags = 0                                   # the same bytes read as canonical cells are garbage
    assert eng.lib.hsw_verify_blocks(eng.h, C.byref(a), C.byref(rep)) == 0 and rep.violations > 100000
    a.flags = N.HSW_REPR_MONTGOMERY
    out["gate"][3 * eng.G + 777, 2] ^= 1 << 40
    assert eng.lib.hsw_verify_blocks(eng.h, C.byref(a), C.byref(rep)) == 0 and rep.violations >= 1 and rep.first_block == 3
    # whole region, Montgomery, column image
    cfg = hsw.Sha256DynamicConfig(eng_int, [128, 128], is_input_range_check=True, whole_digest=True)
    cfg.set_repr(N.HSW_REPR_MONTGOMERY)
    cfg.set_columns((1 << 17) - 9)
    cfg.digest_batch([b"abc", b""], [None, None])
    rep2 = cfg.verify()
    assert rep2["violations"] == 0 and rep2["checks"] > 4 * 80000
    cfg.close()


def test_verify_rejects_non_canonical_montgomery_encodings_and_bad_arguments(engine_factory, hsw):
    """A Montgomery cell is an encoding m < p.  m + p reduces to the same value, so a verifier that only
    reduced on load would accept it: the stored limbs themselves must be < p.  And the argument checks of
    hsw_witness_blocks_ex hold for the verifier too (a short stride / misaligned pointer must be an error
    code, not a device fault)."""
    import ctypes as C
    import torch
    N = hsw._native
    P = [0x43e1f593f0000001, 0x2833e84879b97091, 0xb85045b68181585d, 0x30644e72e131a029]
    eng = engine_factory(8, 2)
    blocks, pre = _inputs(3, 1234)
    tb, tp = torch.from_numpy(blocks).cuda(), torch.from_numpy(pre.view(np.int32)).cuda()
    out = eng.witness_blocks(tb, tp, flags=N.HSW_REPR_MONTGOMERY)
    eng.synchronize()
    a = N.WitnessArgs()
    a.d_blocks, a.d_pre_states, a.n_blocks, a.spread_cursor0 = tb.data_ptr(), tp.data_ptr(), 3, 0
    a.d_gate, a.d_chip_dense, a.d_chip_spread = out["gate"].data_ptr(), out["dense"].data_ptr(), out["spread"].data_ptr()
    a.chip_col_stride, a.d_next_states, a.flags = out["dense"].shape[1], out["next_states"].data_ptr(), N.HSW_REPR_MONTGOMERY
    rep = N.VerifyReport()
    assert eng.lib.hsw_verify_blocks(eng.h, C.byref(a), C.byref(rep)) == 0 and rep.violations == 0

    def add_p(t, idx):          # m -> m + p on the four little-endian limbs of cell idx (m + p < 2^255: no overflow)
        limbs = [int(x) & 0xFFFFFFFFFFFFFFFF for x in t[idx].cpu().numpy().view(np.uint64)]
        carry, res = 0, []
        for l, pl in zip(limbs, P):
            s = l + pl + carry
            res.append(s & 0xFFFFFFFFFFFFFFFF)
            carry = s >> 64
        assert carry == 0
        t[idx] = torch.from_numpy(np.array(res, dtype=np.uint64).view(np.int64)).to(t.device)
        return limbs

    cell = 2 * eng.G + 40000
    saved = add_p(out["gate"], cell)
    assert eng.lib.hsw_verify_blocks(eng.h, C.byref(a), C.byref(rep)) == 0
    assert rep.violations >= 1 and rep.first_block == 2 and rep.first_cell == 40000
    out["gate"][cell] = torch.from_numpy(np.array(saved, dtype=np.uint64).view(np.int64)).cuda()
    saved = add_p(out["dense"][1], 17)                     # a chip cell
    assert eng.lib.hsw_verify_blocks(eng.h, C.byref(a), C.byref(rep)) == 0 and rep.violations >= 1
    out["dense"][1][17] = torch.from_numpy(np.array(saved, dtype=np.uint64).view(np.int64)).cuda()
    assert eng.lib.hsw_verify_blocks(eng.h, C.byref(a), C.byref(rep)) == 0 and rep.violations == 0
    # arguments
    a.chip_col_stride = out["dense"].shape[1] - 1
    assert eng.lib.hsw_verify_blocks(eng.h, C.byref(a), C.byref(rep)) == N.HSW_ERR_INVALID_ARG
    a.chip_col_stride = out["dense"].shape[1]
    a.d_gate = out["gate"].data_ptr() + 8
    assert eng.lib.hsw_verify_blocks(eng.h, C.byref(a), C.byref(rep)) == N.HSW_ERR_INVALID_ARG
    a.d_gate = out["gate"].data_ptr()
    a.d_chip_spread = out["spread"].data_ptr() + 4
    assert eng.lib.hsw_verify_blocks(eng.h, C.byref(a), C.byref(rep)) == N.HSW_ERR_INVALID_ARG
