"""SURVEY 8 f4 on the host: the oracle's whole-digest stream (prologue | zero cell |
blocks | epilogue, assumptions A1-A4) satisfies the constraint structure it records,
hashes correctly, and its section sizes / tape agree with libhsw.so's host arithmetic
(hsw_frame_query / hsw_frame_tape / hsw_gate_tape)."""
import hashlib

import numpy as np
import pytest

from tests.constraint_check import check_whole_stream, cells_to_int, P_INT


CASES = [  # (messages, max sizes, precomputed, input range checks)
    ([b"abc"], [64], None, False),
    ([b"abc", b""], [128, 128], None, True),                       # the reference's TestCircuit shape (lib.rs:455-466, 487-493)
    ([bytes([1] * 56)], [192], None, True),                        # two real rounds, one zero block after them
    ([bytes(range(200))], [128], [128], False),                    # precomputed prefix (lib.rs:153-160)
    ([bytes(range(119)), b"xy"], [128, 64], None, False),          # exactly fills max; different sizes in one context
]


@pytest.mark.parametrize("msgs,sizes,pre,rc", CASES)
def test_oracle_whole_digest_is_consistent(hsw, oracle, msgs, sizes, pre, rc):
    ref = oracle.digest_cells(msgs, sizes, pre, rc, record=True)
    for m, d in zip(msgs, ref["digests"]):
        assert d == hashlib.sha256(m).digest()
    n = check_whole_stream(ref, ref["gate"], ref["lookup"], ref["dense"], ref["spread"],
                           hsw._native.spread_table(8))
    assert n > 80000 * sum(s // 64 for s in sizes)
    # the recorded structure really constrains: any flipped cell breaks something
    rng = np.random.default_rng(7)
    lay = ref["layouts"][0]
    # (cell 27, the inverse witness of is_zero(0), is the one genuinely free cell of the prologue:
    #  z + 0*inv = 1 holds for any inv; halo2-base assigns 1)
    picks = [0, 5, 14, 18, 21, 26, 38, 46, lay["prologue_cells"] + 3]
    epi = lay["gate0"] + lay["prologue_cells"] + lay["zero_cells"] + lay["block_cells"]
    picks += [epi, epi + 6, epi + 12, epi + 19, epi + 76 * (lay["n_blocks"] + 1) + 2, len(ref["gate"]) - 1]
    picks += rng.integers(0, len(ref["gate"]), 6).tolist()
    for c in picks:
        bad = ref["gate"].copy()
        bad[c, 0] ^= np.uint64(1)
        with pytest.raises(AssertionError):
            check_whole_stream(ref, bad, ref["lookup"], ref["dense"], ref["spread"], hsw._native.spread_table(8))


@pytest.mark.parametrize("msgs,sizes,pre,rc", CASES)
def test_frame_shape_and_tape_match_oracle(hsw, oracle, msgs, sizes, pre, rc):
    N = hsw._native
    s = N.shape_query(8, 2, N.HSW_MODE_HALO2_INTERNALS)
    ref = oracle.digest_cells(msgs, sizes, pre, rc)
    block_tape = N.gate_tape(s)
    tape = []
    for i, (mx, lay) in enumerate(zip(sizes, ref["layouts"])):
        fs = N.frame_query(s, mx, rc).as_dict()
        assert fs["n_blocks"] == lay["n_blocks"] == mx // 64
        assert fs["prologue_cells"] == lay["prologue_cells"] and fs["epilogue_cells"] == lay["epilogue_cells"]
        assert fs["prologue_lookups"] == lay["prologue_lookups"] and fs["epilogue_lookups"] == lay["epilogue_lookups"]
        assert fs["digest_cells"] == lay["prologue_cells"] + lay["block_cells"] + lay["epilogue_cells"]
        assert fs["digest_lookups"] == lay["prologue_lookups"] + lay["block_lookups"] + lay["epilogue_lookups"]
        assert lay["zero_cells"] == (1 if i == 0 else 0)            # Context.zero_cell: first load_zero only
        pro, epi = N.frame_tape(s, mx, rc, 0), N.frame_tape(s, mx, rc, 1)
        assert len(pro) == fs["prologue_calls"] and len(epi) == fs["epilogue_calls"]
        assert int(pro.sum()) == fs["prologue_cells"] and int(epi.sum()) == fs["epilogue_cells"]
        tape += [pro] + ([np.array([1], dtype=np.uint8)] if i == 0 else []) + [block_tape] * (mx // 64) + [epi]
    tape = np.concatenate(tape)
    assert np.array_equal(tape, ref["call_lens"])
    assert int(tape.sum()) == len(ref["gate"])


def test_frame_wide_cells_are_where_expected(oracle):
    """The frame's only full-width cells: -2^16 of is_less_than, n - target < 0 and the is_zero
    inverses of is_equal, and select differences state_n - state_target < 0."""
    ref = oracle.digest_cells([b"abc"], [192], None, False)
    lay = ref["layouts"][0]
    g = ref["gate"]
    wide = np.nonzero(g[:, 1:].any(axis=1))[0]
    pro_wide = wide[wide < lay["prologue_cells"]]
    assert pro_wide.tolist() == [18]                                                # P_LT + 4
    assert cells_to_int(g[18:19])[0] == P_INT - 65536
    epi0 = lay["prologue_cells"] + lay["zero_cells"] + lay["block_cells"]
    epi_wide = wide[wide >= epi0] - epi0
    # candidate 0 (n = 0 < target = 1): diff at +0, +5, +9 and its inverse at +6
    assert {0, 5, 6, 9} <= set(epi_wide.tolist())
    # target candidate (n = 1): nothing wide in its is_equal
    assert not (set(range(76, 88)) & set(epi_wide.tolist()))


def test_column_count_sanity(hsw):
    """Under A1-A4 the reference's two circuits need exactly the advice columns they configure:
    TestCircuit (2 x 128 B, k = 17, NUM_ADVICE = 3, lib.rs:487-493) and the bench circuit
    (1 x 1024 B, k = 17, NUM_ADVICE = 9, benches/digest.rs:103-108).  Usable rows of a
    2^17-row halo2 column: 2^17 - 9 blinding rows (MockProver / create_proof minimum)."""
    N = hsw._native
    s = N.shape_query(8, 2, N.HSW_MODE_HALO2_INTERNALS)
    usable = (1 << 17) - 9
    test_cells = 1 + 2 * N.frame_query(s, 128, True).digest_cells
    bench_cells = 1 + N.frame_query(s, 1024, True).digest_cells
    assert test_cells == 279797 and bench_cells == 1116315
    assert -(-test_cells // usable) == 3
    assert -(-bench_cells // usable) == 9
    # the lookup-advice column (NUM_LOOKUP_ADVICE = 1) holds them too
    assert 2 * N.frame_query(s, 128, True).digest_lookups <= usable
    assert N.frame_query(s, 1024, True).digest_lookups == 53059
