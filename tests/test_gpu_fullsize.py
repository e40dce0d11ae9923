"""BASELINE.json's full single-GPU size (configs[2]: 4,096 single-block
messages, 9.77 GB of cells) checked through size-independent properties:

  1. every digest recomputed from next_states equals hashlib's;
  2. every 4-cell gate row [x0,x1,x2,x3] of every block satisfies the FlexGate
     equation x0 + x1*x2 = x3 (what MockProver::verify checks, lib.rs:525-526),
     evaluated on the GPU over all 55 M gate rows;
  3. every chip row is a (x, spread(x)) table row (the "spread lookup",
     spread.rs:56-62);
  4. a strided sample of blocks is bit-compared with the oracle;
  5. the run is deterministic (two passes give identical bytes).
"""
import hashlib

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

N_MSG = 4096


def _workload():
    rng = np.random.default_rng(0xC3)
    msgs = rng.integers(0, 256, (N_MSG, 55), dtype=np.uint8)
    blocks = np.zeros((N_MSG, 64), dtype=np.uint8)
    blocks[:, :55] = msgs
    blocks[:, 55] = 0x80
    blocks[:, 62] = (55 * 8) >> 8
    blocks[:, 63] = (55 * 8) & 0xFF
    return msgs, blocks


def test_full_batch_properties(engine_factory, oracle):
    import torch
    eng = engine_factory(8, 2)
    msgs, blocks = _workload()
    pre = np.tile(oracle.INIT_STATE, (N_MSG, 1))
    tb = torch.from_numpy(blocks).cuda()
    tp = torch.from_numpy(pre.view(np.int32)).cuda()
    out = eng.witness_blocks(tb, tp)
    eng.synchronize()
    G = eng.G

    # 1. digests
    ns = out["next_states"].cpu().numpy().view(np.uint32)
    be = ns.astype(">u4").tobytes()
    for i in range(N_MSG):
        assert be[32 * i:32 * i + 32] == hashlib.sha256(msgs[i].tobytes()).digest(), i

    # 2. gate equation on every gate row (chunked to bound temporary memory)
    tape = oracle.gate_tape(8, 2)
    starts = torch.from_numpy(np.nonzero(tape == 1)[0].astype(np.int64)).cuda()
    assert starts.numel() == 13510
    gate = out["gate"].view(N_MSG, G, 4)
    P = [0x43e1f593f0000001, 0x2833e84879b97091, 0xb85045b68181585d, 0x30644e72e131a029]
    Pt = torch.tensor([p - (1 << 64) if p >= (1 << 63) else p for p in P], dtype=torch.int64, device="cuda")
    CH = 256
    wide_rows_total = 0
    for lo in range(0, N_MSG, CH):
        g = gate[lo:lo + CH]
        x = [g[:, starts + k, :] for k in range(4)]              # each (CH, 13510, 4)
        narrow = torch.ones(x[0].shape[:2], dtype=torch.bool, device="cuda")
        for k in range(4):
            narrow &= (x[k][..., 1:] == 0).all(dim=-1)
        # all four values < 2^64: x0 + x1*x2 == x3 holds in Z, hence mod 2^64
        lhs = x[0][..., 0] + x[1][..., 0] * x[2][..., 0]
        assert bool(((lhs == x[3][..., 0]) | ~narrow).all()), "gate equation violated (narrow rows)"
        # x1*x2 must not have wrapped: x2 is a constant < 2^61 or x1 is
        wide = ~narrow
        wide_rows_total += int(wide.sum())
        # wide rows are ch's neg rows [a, p-a, 1, 0] and [M, p-a, 1, M-a] (compression.rs:320-335)
        w0, w1, w2, w3 = (x[k][wide] for k in range(4))
        assert bool((w2[:, 0] == 1).all() and (w2[:, 1:] == 0).all())
        assert bool((w1[:, 1:] == Pt[1:]).all()), "wide cell is not p - small"
        a = Pt[0] - w1[:, 0]                                      # a = p0 - (p0 - a), no borrow
        assert bool(((a > 0) & (a <= 0x55555555)).all())
        is_neg_row = (w3[:, 0] == 0) & (w3[:, 1:] == 0).all(dim=-1)
        assert bool((w0[:, 0][is_neg_row] == a[is_neg_row]).all())
        assert bool(((w0[:, 0] - a)[~is_neg_row] == w3[:, 0][~is_neg_row]).all())
        assert bool((w0[:, 0][~is_neg_row] == 0x55555555).all())
    assert 0 < wide_rows_total <= 256 * N_MSG

    # 3. chip rows are spread-table rows
    d = out["dense"][..., 0]
    s = out["spread"][..., 0]
    assert bool((out["dense"][..., 1:] == 0).all() and (out["spread"][..., 1:] == 0).all())
    assert bool(((d >= 0) & (d < 256)).all())
    tab = torch.tensor([v for _, v in oracle.spread_table(8)], dtype=torch.int64, device="cuda")
    assert bool((tab[d] == s).all())
    assert d.shape[1] == 2060 * N_MSG

    # 4. strided sample vs the oracle, bit-exact
    idx = np.arange(0, N_MSG, 64)
    ref = oracle.Oracle(8, 2, check=True).witness_blocks(blocks[idx], pre[idx])
    got = gate[torch.from_numpy(idx).cuda()].cpu().numpy().view(np.uint64).reshape(-1, 4)
    assert np.array_equal(got, ref["gate"])
    for j, b in enumerate(idx):
        rows = slice(2060 * int(b), 2060 * int(b) + 2060)
        assert np.array_equal(out["dense"][:, rows].cpu().numpy().view(np.uint64), ref["dense"][:, 2060 * j:2060 * j + 2060])

    # 5. determinism
    h1 = [int(out["gate"].sum()), int(out["dense"].sum()), int(out["spread"].sum())]
    first = out["gate"][:G].clone()
    out2 = eng.witness_blocks(tb, tp)
    eng.synchronize()
    assert h1 == [int(out2["gate"].sum()), int(out2["dense"].sum()), int(out2["spread"].sum())]
    assert bool((out2["gate"][:G] == first).all())


def _check_on_gpu(oracle, eng, blocks, pre, out, internals, chunk=256):
    """MockProver-style verification of the GPU streams against the reference's
    constraint STRUCTURE (tests/constraint_check.py): no oracle-computed value is used."""
    import torch
    from tests.constraint_check import check_block_batch, chip_in_call_order
    n = blocks.shape[0]
    cs = oracle.constraint_system(8, 2, internals)
    G, LC, LK = cs["G"], cs["LC"], cs["LK"]
    assert eng.G == G
    gate = out["gate"].view(n, G, 4)
    dl = chip_in_call_order(torch, out["dense"], n, LC, 2)
    sl = chip_in_call_order(torch, out["spread"], n, LC, 2)
    tab = [s for _, s in oracle.spread_table(8)]
    tb = torch.from_numpy(blocks.astype(np.int64)).cuda()
    tp = torch.from_numpy(pre.astype(np.int64)).cuda()
    ns = out["next_states"].to(torch.int64) & 0xFFFFFFFF
    lk = out["lookup"][:, 0].view(n, LK) if internals else None
    total = 0
    for lo in range(0, n, chunk):
        hi = min(n, lo + chunk)
        total += check_block_batch(torch, cs, gate[lo:hi], tb[lo:hi], tp[lo:hi], dl[lo:hi], sl[lo:hi], ns[lo:hi],
                                   lk[lo:hi] if internals else None, tab, 8)
    return total


def test_full_batch_satisfies_the_reference_constraint_system(engine_factory, oracle):
    """All 4,096 blocks of BASELINE configs[2]: every gate row, every copy constraint
    (assert_equal + Existing gate inputs), every fixed constant, every range bound, every
    chip-cell tie and spread lookup, and the digest -- what MockProver::verify checks
    (lib.rs:525-526) -- evaluated on the GPU output without any oracle value."""
    import torch
    eng = engine_factory(8, 2)
    msgs, blocks = _workload()
    pre = np.tile(oracle.INIT_STATE, (N_MSG, 1))
    out = eng.witness_blocks(torch.from_numpy(blocks).cuda(), torch.from_numpy(pre.view(np.int32)).cuda())
    eng.synchronize()
    total = _check_on_gpu(oracle, eng, blocks, pre, out, internals=False)
    assert total > N_MSG * 70000
    # inputs are tied through the external cells: a stream computed for OTHER inputs must fail
    other = blocks.copy()
    other[5, 7] ^= 0x10
    with pytest.raises(AssertionError):
        _check_on_gpu(oracle, eng, other[:256], pre[:256], {k: (v[:256 * eng.G] if k == "gate" else
                      (v[:, :256 * 2060] if k in ("dense", "spread") else v[:256])) for k, v in out.items()
                      if k in ("gate", "dense", "spread", "next_states")}, internals=False)


def test_internals_mode_satisfies_constraints_including_lookup_column(hsw, oracle):
    """HSW_MODE_HALO2_INTERNALS on 512 blocks: the range_check rows are real gate rows now, their
    limbs are tied to the lookup column, and constrain_equal(a, acc) is a recorded copy constraint."""
    import torch
    eng = hsw.WitnessEngine(0, 8, 2, mode=hsw._native.HSW_MODE_HALO2_INTERNALS)
    n = 512
    rng = np.random.default_rng(0xC3)
    blocks = rng.integers(0, 256, (n, 64), dtype=np.uint8)
    pre = rng.integers(0, 2**32, (n, 8), dtype=np.uint64).astype(np.uint32)
    out = eng.witness_blocks_ex(torch.from_numpy(blocks).cuda(), torch.from_numpy(pre.view(np.int32)).cuda(),
                                want_lookup=True)
    eng.synchronize()
    total = _check_on_gpu(oracle, eng, blocks, pre, out, internals=True)
    assert total > n * 75000
    eng.close()
