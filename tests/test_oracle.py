"""The oracle against every vector the reference's own tests hold (digest-level
KATs, lib.rs:497-611), FIPS/hashlib, SURVEY 8a tallies, and its own committed
stream fingerprints.  CPU only."""
import hashlib
import json
import os

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
KATS = json.load(open(os.path.join(HERE, "golden", "reference_kats.json")))
FPS = json.load(open(os.path.join(HERE, "golden", "stream_fingerprints.json")))


@pytest.mark.parametrize("vec", KATS["vectors"], ids=lambda v: v["cite"].split(" ")[0])
def test_reference_kat_digests(oracle, vec):
    o = oracle.Oracle(KATS["num_bits_lookup"], KATS["num_advice_columns"], check=True)
    r = o.digest(bytes.fromhex(vec["input_hex"]), KATS["max_variable_byte_size"],
                 vec["precomputed_input_len"])
    assert r["digest"].hex() == vec["digest_hex"]
    # the reference always synthesises max/64 blocks (lib.rs:180)
    assert r["blocks"].shape[0] == KATS["max_variable_byte_size"] // 64
    st = o.stats()
    assert st["gate_cells"] == 2 * 66308 and st["assert_equal"] == 2 * 3850


def test_reference_test_circuit_two_hashes_share_cursor(oracle):
    """TestCircuit hashes two messages with ONE SpreadConfig (lib.rs:455-468):
    the chip cursor runs on from the first digest into the second."""
    o = oracle.Oracle(8, 2, check=True)
    o.digest(b"abc", 128)
    assert o.cursor == 2 * 4120
    r = o.digest(b"", 128, want_streams=True)
    assert r["row_base"] == 2 * 2060 and o.cursor == 4 * 4120
    assert r["digest"].hex() == KATS["vectors"][1]["digest_hex"]


def test_random_192_bytes_with_precomputed_prefix(oracle):
    """lib.rs:587-611 shape: 192-byte inputs, 128 precomputed, max 128."""
    rng = np.random.default_rng(20240611)
    for _ in range(4):
        m = rng.integers(0, 256, KATS["random_case"]["input_len"], dtype=np.uint8).tobytes()
        r = oracle.Oracle(8, 2, check=True).digest(m, 128, KATS["random_case"]["precomputed_input_len"])
        assert r["digest"] == hashlib.sha256(m).digest()
        # first pre-state is the plain-SHA state after the 2 prefix blocks (lib.rs:153-160)
        st = oracle.INIT_STATE.copy()
        for b in range(2):
            st = oracle.plain_compress(st, np.frombuffer(m[64 * b:64 * b + 64], dtype=np.uint8))
        assert np.array_equal(r["pre_states"][0], st)


def test_bench_workload_16_blocks(oracle):
    """benches/digest.rs: [0x01; 56] at MAX_BYTE_SIZE 1024 -> 16 compressions."""
    m = b"\x01" * 56
    o = oracle.Oracle(8, 2, check=True)
    r = o.digest(m, KATS["bench_case"]["max_variable_byte_size"])
    assert r["digest"] == hashlib.sha256(m).digest()
    assert r["blocks"].shape[0] == 16 and not r["blocks"][2:].any()
    assert o.stats()["gate_cells"] == 16 * 66308


@pytest.mark.parametrize("n", [0, 1, 54, 55, 56, 63, 64, 118, 119])
def test_padding_boundaries_vs_hashlib(oracle, n):
    m = bytes((7 * i + 3) & 0xFF for i in range(n))
    r = oracle.Oracle(8, 2, check=True).digest(m, 128)
    assert r["digest"] == hashlib.sha256(m).digest()


def test_reference_asserts_become_errors(oracle):
    o = oracle.Oracle(8, 2)
    with pytest.raises(ValueError):       # lib.rs:90 padded size exceeds max
        o.digest(b"x" * 120, 128)
    with pytest.raises(ValueError):       # lib.rs:89 precomputed len not a multiple of 64
        o.digest(b"x" * 100, 128, 32)
    with pytest.raises(ValueError):       # lib.rs:57-59 max not a multiple of 64
        o.digest(b"x", 100)
    with pytest.raises(ValueError):       # spread.rs:37
        oracle.Oracle(5, 2)


def test_survey_tallies(oracle):
    """SURVEY 8a per-block counts, measured by running the call sequence."""
    o = oracle.Oracle(8, 2, check=True)
    rng = np.random.default_rng(1)
    o.witness_blocks(rng.integers(0, 256, (1, 64), dtype=np.uint8),
                     rng.integers(0, 2**32, (1, 8), dtype=np.uint64).astype(np.uint32))
    st = o.stats()
    assert (st["load_witness"], st["add"], st["neg"], st["mul_add"]) == (12268, 1368, 128, 12014)
    assert st["load_zero"] == 4362 and st["assert_equal"] == 3850
    assert st["spread_calls"] == 2060 and st["spread_limb_calls"] == 4120
    assert st["range_check16"] == 1664 and st["range_check32"] == 760 and st["even_odd_calls"] == 832
    assert st["gate_cells"] == 12268 + 4 * (1368 + 128 + 12014) == 66308
    assert st["chip_cells"] == 8240


def test_spread_table(oracle):
    """SpreadConfig::load rows (spread.rs:165-194)."""
    t = oracle.spread_table(8)
    assert len(t) == 256 and t[0] == (0, 0) and t[1] == (1, 1) and t[2] == (2, 4) and t[255] == (255, 0x5555)
    for i, s in t:
        assert sum(((s >> (2 * b)) & 1) << b for b in range(16)) == i and s & 0xAAAAAAAA == 0


def test_chip_layout_rule(oracle):
    """spread.rs:202-231: limb call n -> column n % ncols, row n / ncols."""
    rng = np.random.default_rng(5)
    blocks = rng.integers(0, 256, (2, 64), dtype=np.uint8)
    pre = rng.integers(0, 2**32, (2, 8), dtype=np.uint64).astype(np.uint32)
    a = oracle.Oracle(8, 2).witness_blocks(blocks, pre)          # 2 columns
    b = oracle.Oracle(8, 1).witness_blocks(blocks, pre)          # same limbs, 1 column
    flat2 = a["dense"][:, :, 0].T.reshape(-1)                    # (row, col) order = call order
    assert np.array_equal(flat2, b["dense"][0, :, 0])
    assert (a["dense"][..., 1:] == 0).all() and (a["dense"][..., 0] < 256).all()
    # spread column is the table image of the dense column
    tab = np.array([s for _, s in oracle.spread_table(8)], dtype=np.uint64)
    assert np.array_equal(a["spread"][..., 0], tab[a["dense"][..., 0]])


def test_neg_cells_are_field_negations(oracle):
    """ch's two neg gates (compression.rs:320-321) are the only >64-bit cells."""
    P = 0x30644e72e131a029b85045b68181585d2833e84879b9709143e1f593f0000001
    rng = np.random.default_rng(9)
    w = oracle.Oracle(8, 2).witness_blocks(rng.integers(0, 256, (1, 64), dtype=np.uint8),
                                           oracle.INIT_STATE.reshape(1, 8))
    g = w["gate"]
    wide = np.nonzero(g[:, 2] | g[:, 3])[0]
    assert 0 < len(wide) <= 256          # 128 neg outputs, each reused once as an add operand
    for i in wide[:16]:
        v = sum(int(g[i, k]) << (64 * k) for k in range(4))
        x = P - v
        assert 0 < x <= 0x55555555 and x & 0xAAAAAAAA == 0


@pytest.mark.parametrize("case", FPS["cases"], ids=lambda c: "%s-b%d-c%d-@%d" % (
    c["name"], c["num_bits_lookup"], c["num_advice_columns"], c["cursor0"]))
def test_oracle_matches_committed_fingerprints(oracle, case):
    from tests.golden.make_golden import fingerprint, golden_inputs
    blocks, pre = golden_inputs(case["name"])
    w = oracle.Oracle(case["num_bits_lookup"], case["num_advice_columns"], check=True).witness_blocks(
        blocks, pre, cursor0=case["cursor0"])
    fp = fingerprint(w)
    for k, v in fp.items():
        assert case[k] == v, k


@pytest.mark.parametrize("bits", [1, 2, 4, 8, 16])
def test_all_table_widths_selfcheck(oracle, bits):
    g, lc = oracle.measure_shape(bits, 2)
    L = 16 // bits
    assert g == 25108 + 2060 * 10 * L and lc == 2060 * L


def test_to_montgomery_matches_bigint(oracle):
    """x -> x * 2^256 mod p (halo2curves Fr memory form), vs Python integers."""
    P = 0x30644e72e131a029b85045b68181585d2833e84879b9709143e1f593f0000001
    R = (1 << 256) % P
    assert R == 0x0e0a77c19a07df2f666ea36f7879462e36fc76959f60cd29ac96341c4ffffffb   # halo2curves bn256 fr.rs R
    rng = np.random.default_rng(3)
    vals = [0, 1, 2, 255, 0x55555555, 0xFFFFFFFF, 1 << 32, (1 << 64) - 1, P - 1, P - 0x55555555]
    vals += [int(rng.integers(0, 2**63)) << int(rng.integers(0, 2)) for _ in range(50)]
    a = np.array([[(v >> (64 * k)) & (2**64 - 1) for k in range(4)] for v in vals], dtype=np.uint64)
    m = oracle.to_montgomery(a)
    for v, row in zip(vals, m):
        assert sum(int(row[k]) << (64 * k) for k in range(4)) == (v * R) % P
