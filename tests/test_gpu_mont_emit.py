"""Montgomery cells converted at EMIT time (hsw_expand.hpp Em::M32: tiles of finished 32-byte cells, every
distinct value of a unit converted once, byte tables for the limbs and 16-bit witnesses) against the oracle and
against the write-out conversion (one multiply + Barrett step per cell) -- same bytes either way.  The engine
uses it for streaming launches of 1,536 blocks or more in default mode (one wave per block: smaller launches
would not fill the chip); "mont_emit" = 2 forces it for any size and in internals mode too, which is what
exercises its realigning write-out, column breaks and the lookup column."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _inputs(n, seed):
    rng = np.random.default_rng(seed)
    blocks = rng.integers(0, 256, (n, 64), dtype=np.uint8)
    pre = rng.integers(0, 2**32, (n, 8), dtype=np.uint64).astype(np.uint32)
    blocks[0] = 0xFF; pre[0] = 0xFFFFFFFF           # all-ones words: r_spread = 2^64 - 1
    if n > 1:
        blocks[1] = 0; pre[1] = 0                   # neg(0) stays 0
    return blocks, pre


@pytest.mark.parametrize("n,cursor0,ncols", [(3, 0, 2), (150, 7, 2), (131, 4121, 3)])
def test_emit_time_conversion_default_mode(hsw, oracle, n, cursor0, ncols):
    import torch
    N = hsw._native
    blocks, pre = _inputs(n, 31 + n)
    ref = oracle.Oracle(8, ncols, check=False).witness_blocks(blocks, pre, cursor0=cursor0)
    tb, tp = torch.from_numpy(blocks).cuda(), torch.from_numpy(pre.view(np.int32)).cuda()
    outs = []
    for emit in (2, 0):
        eng = hsw.WitnessEngine(0, 8, ncols)
        eng.set_option("split", 0)                  # the streaming kernel also for the 3-block batch
        eng.set_option("mont_emit", emit)
        out = eng.witness_blocks(tb, tp, cursor0=cursor0, flags=N.HSW_REPR_MONTGOMERY)
        eng.synchronize()
        assert eng.last_launch()["repr"] == (3 if emit else 1)
        outs.append({k: out[k].cpu().numpy() for k in ("gate", "dense", "spread", "next_states")})
        eng.close()
    g = outs[0]["gate"].view(np.uint64)
    exp = oracle.to_montgomery(ref["gate"])
    bad = np.nonzero((g != exp).any(axis=1))[0]
    assert len(bad) == 0, "first differing cells %s (block-relative %s)" % (bad[:6], bad[:6] % 66308)
    assert np.array_equal(outs[0]["dense"].view(np.uint64), oracle.to_montgomery(ref["dense"]))
    assert np.array_equal(outs[0]["spread"].view(np.uint64), oracle.to_montgomery(ref["spread"]))
    assert np.array_equal(outs[0]["next_states"].view(np.uint32), ref["next_states"])
    for k in outs[0]:
        assert np.array_equal(outs[0][k], outs[1][k]), k          # identical to the write-out conversion


@pytest.mark.parametrize("n,start_row,max_rows", [
    (3, 1001, 100000),       # misaligned stream (skew 1): realigned write-out, one column break
    (4, 69002, 70000),       # a break in every block, skew 2
    (140, 3, 1 << 20)])      # a long batch, skew 3
def test_emit_time_conversion_internals_mode(hsw, oracle, n, start_row, max_rows):
    """mont_emit = 2: lookup column, range_check rows, column breaks and the realigned write-out of 32-byte tiles."""
    import torch
    N = hsw._native
    blocks, pre = _inputs(n, 57 + n)
    ref = oracle.Oracle(8, 2, check=False, internals=True).witness_blocks(blocks, pre, cursor0=2)
    tb, tp = torch.from_numpy(blocks).cuda(), torch.from_numpy(pre.view(np.int32)).cuda()
    got = []
    for emit in (2, 0):
        eng = hsw.WitnessEngine(0, 8, 2, mode=N.HSW_MODE_HALO2_INTERNALS)
        eng.set_option("split", 0)
        eng.set_option("mont_emit", emit)
        out = eng.witness_blocks_ex(tb, tp, cursor0=2, flags=N.HSW_REPR_MONTGOMERY, want_lookup=True,
                                    start_row=start_row, max_rows=max_rows)
        eng.synchronize()
        assert eng.last_launch()["repr"] == (3 if emit else 1)
        got.append({k: out[k].cpu().numpy() for k in ("gate", "dense", "spread", "lookup")})
        eng.close()
    for k in got[0]:
        assert np.array_equal(got[0][k], got[1][k]), k            # cell for cell, the untouched (-1) gap rows included
    assert np.array_equal(got[0]["lookup"].view(np.uint64), oracle.to_montgomery(ref["lookup"]))
    flat = got[0]["gate"].view(np.uint64)
    assert (flat == np.uint64(2**64 - 1)).all(axis=1).sum() == flat.shape[0] - ref["gate"].shape[0]   # only the gaps are untouched
    used = ~(flat == np.uint64(2**64 - 1)).all(axis=1)
    assert np.array_equal(flat[used], oracle.to_montgomery(ref["gate"]))


def test_default_engages_emit_time_conversion_from_1536_blocks(hsw):
    """Default options: launches of 1,536 blocks or more take the emit-time kernel, smaller ones the write-out
    conversion (which spreads a block over up to 16 waves); both give the same bytes."""
    import torch
    N = hsw._native
    blocks, pre = _inputs(1536, 99)
    tb, tp = torch.from_numpy(blocks).cuda(), torch.from_numpy(pre.view(np.int32)).cuda()
    eng = hsw.WitnessEngine(0, 8, 2)
    out = eng.witness_blocks(tb, tp, flags=N.HSW_REPR_MONTGOMERY)
    eng.synchronize()
    assert eng.last_launch()["repr"] == 3
    big = {k: out[k].clone() for k in ("gate", "dense", "spread", "next_states")}
    del out
    small = eng.witness_blocks(tb[:1535], tp[:1535], flags=N.HSW_REPR_MONTGOMERY)
    eng.synchronize()
    assert eng.last_launch()["repr"] == 1
    G = eng.G
    assert torch.equal(small["gate"].view(-1, 4)[: 1535 * G], big["gate"].view(-1, 4)[: 1535 * G])
    assert torch.equal(small["next_states"], big["next_states"][:1535])
    rows = small["dense"].shape[1] - 1               # (the last chip row of the shorter batch may be partial)
    assert torch.equal(small["dense"][:, :rows], big["dense"][:, :rows]) and torch.equal(small["spread"][:, :rows], big["spread"][:, :rows])
    eng.close()
