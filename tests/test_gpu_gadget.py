"""Gadget front-end on the GPU: Sha256DynamicConfig.digest through the C ABI,
driven exactly like the reference's TestCircuit / bench (two digests sharing
one SpreadConfig; 16-block bench message), compared with the oracle."""
import hashlib
import json
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
KATS = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "reference_kats.json")))


def _oracle_two(oracle, msgs, maxes, pres):
    o = oracle.Oracle(8, 2, check=True)
    outs = [o.digest(m, mx, p, want_streams=True) for m, mx, p in zip(msgs, maxes, pres)]
    return outs


@pytest.mark.parametrize("pair", [(0, 1), (2, 1), (3, 4)])
def test_reference_test_circuit_pairs(engine_factory, hsw, oracle, pair):
    """lib.rs:496-584: TestCircuit hashes two messages (max 128 B each) in one region."""
    eng = engine_factory(8, 2)
    vecs = [KATS["vectors"][i] for i in pair]
    msgs = [bytes.fromhex(v["input_hex"]) for v in vecs]
    cfg = hsw.Sha256DynamicConfig(eng, [128, 128], True)
    r0 = cfg.digest(msgs[0], None)
    r1 = cfg.digest(msgs[1], None)
    assert r0.output_bytes.hex() == vecs[0]["digest_hex"] and r1.output_bytes.hex() == vecs[1]["digest_hex"]
    assert (r0.first_block, r1.first_block) == (0, 2) and r1.spread_cursor0 == 2 * 4120
    assert cfg.view().cur_hash_idx == 2
    ref = _oracle_two(oracle, msgs, [128, 128], [0, 0])
    st = cfg.streams()
    assert np.array_equal(st["gate"], np.concatenate([ref[0]["gate"], ref[1]["gate"]]))
    assert np.array_equal(st["dense"], np.concatenate([ref[0]["dense"], ref[1]["dense"]], axis=1))
    assert np.array_equal(st["spread"], np.concatenate([ref[0]["spread"], ref[1]["spread"]], axis=1))
    assert r0.input_bytes == ref[0]["blocks"].tobytes() and r0.input_len == len(msgs[0])
    with pytest.raises(hsw.HswError):           # a third digest: max_variable_byte_sizes[2] does not exist
        cfg.digest(b"", None)
    cfg.close()


def test_random_192_with_precomputed_prefix(engine_factory, hsw, oracle):
    """lib.rs:587-611."""
    eng = engine_factory(8, 2)
    rng = np.random.default_rng(4)
    msgs = [rng.integers(0, 256, 192, dtype=np.uint8).tobytes() for _ in range(2)]
    cfg = hsw.Sha256DynamicConfig(eng, [128, 128], True)
    rs = cfg.digest_batch(msgs, [128, 128])
    for m, r in zip(msgs, rs):
        assert r.output_bytes == hashlib.sha256(m).digest()
        assert r.num_round == 4 and r.target_round == 2
    ref = _oracle_two(oracle, msgs, [128, 128], [128, 128])
    st = cfg.streams()
    assert np.array_equal(st["gate"], np.concatenate([ref[0]["gate"], ref[1]["gate"]]))
    assert np.array_equal(st["dense"], np.concatenate([ref[0]["dense"], ref[1]["dense"]], axis=1))
    cfg.close()


def test_bench_circuit_16_blocks(engine_factory, hsw, oracle):
    """benches/digest.rs: [0x01; 56] at MAX_BYTE_SIZE 1024 -> 16 compressions, 14 of them on zero blocks."""
    eng = engine_factory(8, 2)
    m = b"\x01" * 56
    cfg = hsw.Sha256DynamicConfig(eng, [1024, 1024], True)
    r = cfg.digest(m, None)
    assert r.output_bytes == hashlib.sha256(m).digest() and r.n_blocks == 16 and r.target_round == 2
    ref = oracle.Oracle(8, 2, check=True).digest(m, 1024, want_streams=True)
    st = cfg.streams()
    assert np.array_equal(st["gate"], ref["gate"])
    assert np.array_equal(st["dense"], ref["dense"]) and np.array_equal(st["spread"], ref["spread"])
    cfg.close()


def test_ragged_max_sizes_and_batch_equals_sequential(engine_factory, hsw, oracle):
    eng = engine_factory(8, 2)
    rng = np.random.default_rng(12)
    maxes = [64, 256, 128, 192]
    msgs = [rng.integers(0, 256, n, dtype=np.uint8).tobytes() for n in (10, 200, 0, 100)]
    a = hsw.Sha256DynamicConfig(eng, maxes, False)
    ra = a.digest_batch(msgs)
    b = hsw.Sha256DynamicConfig(eng, maxes, False)
    rb = [b.digest(m) for m in msgs]
    for m, x, y in zip(msgs, ra, rb):
        assert x.output_bytes == y.output_bytes == hashlib.sha256(m).digest()
        assert (x.first_block, x.spread_cursor0) == (y.first_block, y.spread_cursor0)
    sa, sb = a.streams(), b.streams()
    assert np.array_equal(sa["gate"], sb["gate"]) and np.array_equal(sa["dense"], sb["dense"])
    ref = _oracle_two(oracle, msgs, maxes, [0] * 4)
    assert np.array_equal(sa["gate"], np.concatenate([r["gate"] for r in ref]))
    a.close()
    b.close()


def test_gadget_errors(engine_factory, hsw):
    eng = engine_factory(8, 2)
    with pytest.raises(hsw.HswError) as ei:
        hsw.Sha256DynamicConfig(eng, [100], True)              # lib.rs:57-59
    assert ei.value.status == hsw._native.HSW_ERR_SHAPE
    cfg = hsw.Sha256DynamicConfig(eng, [128], True)
    with pytest.raises(hsw.HswError) as ei:
        cfg.digest(b"x" * 120)                                   # lib.rs:90
    assert ei.value.status == hsw._native.HSW_ERR_TOO_LARGE
    assert cfg.view().cur_hash_idx == 0 and cfg.view().blocks_done == 0   # nothing committed
    assert cfg.digest(b"abc").output_bytes == hashlib.sha256(b"abc").digest()
    cfg.close()


def test_gadget_montgomery_repr(engine_factory, hsw, oracle):
    """hsw_gadget_set_repr(HSW_REPR_MONTGOMERY): same digests, cells in halo2curves' memory form."""
    eng = engine_factory(8, 2)
    cfg = hsw.Sha256DynamicConfig(eng, [128], True)
    cfg.set_repr(hsw.HSW_REPR_MONTGOMERY)
    r = cfg.digest(b"abc")
    assert r.output_bytes == hashlib.sha256(b"abc").digest()
    ref = oracle.Oracle(8, 2, check=True).digest(b"abc", 128, want_streams=True)
    st = cfg.streams()
    assert np.array_equal(st["gate"], oracle.to_montgomery(ref["gate"]))
    assert np.array_equal(st["dense"], oracle.to_montgomery(ref["dense"]))
    cfg.close()


def test_large_batch_uses_gpu_chain_and_matches(engine_factory, hsw):
    """> 2,048 blocks in one batch: the ragged GPU chain kernel path (small batches chain on the host)."""
    eng = engine_factory(8, 2)
    rng = np.random.default_rng(21)
    n = 700
    maxes = [192 if i % 3 else 256 for i in range(n)]          # 3- and 4-block hashes, 2,334 blocks
    msgs = [rng.integers(0, 256, int(rng.integers(0, mx - 9)), dtype=np.uint8).tobytes() for mx in maxes]
    cfg = hsw.Sha256DynamicConfig(eng, maxes, False)
    rs = cfg.digest_batch(msgs)
    assert sum(r.n_blocks for r in rs) > 2048
    for m, r in zip(msgs, rs):
        assert r.output_bytes == hashlib.sha256(m).digest()
    cfg.close()
