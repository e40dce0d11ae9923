"""HIP path vs CPU oracle, bit-exact, through the C ABI (include/hsw.h).

Integer / byte work: the bar is bit-exact equality of every cell.
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _rand_inputs(n, seed):
    rng = np.random.default_rng(seed)
    blocks = rng.integers(0, 256, (n, 64), dtype=np.uint8)
    pre = rng.integers(0, 2**32, (n, 8), dtype=np.uint64).astype(np.uint32)
    return blocks, pre


def _run_gpu(eng, blocks, pre, cursor0=0, flags=0):
    import torch
    tb = torch.from_numpy(blocks).cuda()
    tp = torch.from_numpy(pre.view(np.int32)).cuda()
    out = eng.witness_blocks(tb, tp, cursor0=cursor0, flags=flags)
    eng.synchronize()
    return {k: (v.cpu().numpy() if hasattr(v, "cpu") else v) for k, v in out.items()}


def _assert_same(gpu, ref):
    g = gpu["gate"].view(np.uint64)
    if not np.array_equal(g, ref["gate"]):
        bad = np.nonzero((g != ref["gate"]).any(axis=1))[0]
        raise AssertionError("gate stream differs at %d cells, first cell %d: gpu %s ref %s" % (
            len(bad), bad[0], g[bad[0]], ref["gate"][bad[0]]))
    assert np.array_equal(gpu["dense"].view(np.uint64), ref["dense"]), "chip dense columns differ"
    assert np.array_equal(gpu["spread"].view(np.uint64), ref["spread"]), "chip spread columns differ"
    assert np.array_equal(gpu["next_states"].view(np.uint32), ref["next_states"]), "next states differ"


@pytest.mark.parametrize("bits,ncols", [(8, 2), (16, 1), (4, 3), (8, 1), (8, 5), (16, 2), (4, 2)])
def test_random_blocks_all_shapes(engine_factory, oracle, bits, ncols):
    eng = engine_factory(bits, ncols)
    blocks, pre = _rand_inputs(5, 100 + bits * 10 + ncols)
    ref = oracle.Oracle(bits, ncols, check=True).witness_blocks(blocks, pre)
    assert ref["gate_cells_per_block"] == eng.G
    _assert_same(_run_gpu(eng, blocks, pre), ref)


@pytest.mark.parametrize("cursor0", [0, 1, 7, 4120, 123457])
def test_cursor_positions(engine_factory, oracle, cursor0):
    """SpreadConfig.num_limb_sum carries across calls (spread.rs:26,228-231)."""
    for bits, ncols in [(8, 2), (8, 3)]:
        eng = engine_factory(bits, ncols)
        blocks, pre = _rand_inputs(3, 7 + cursor0)
        ref = oracle.Oracle(bits, ncols, check=True).witness_blocks(blocks, pre, cursor0=cursor0)
        _assert_same(_run_gpu(eng, blocks, pre, cursor0=cursor0), ref)


def test_edge_words(engine_factory, oracle):
    """All-zero / all-one words: neg(0) = 0 cells, r_spread = 2^64-1, carries."""
    eng = engine_factory(8, 2)
    blocks = np.zeros((4, 64), dtype=np.uint8)
    blocks[1] = 0xFF
    blocks[2, ::2] = 0xAA
    blocks[3, 1::2] = 0x55
    pre = np.zeros((4, 8), dtype=np.uint32)
    pre[1] = 0xFFFFFFFF
    pre[2] = oracle.INIT_STATE
    pre[3] = 0x80000000
    ref = oracle.Oracle(8, 2, check=True).witness_blocks(blocks, pre)
    _assert_same(_run_gpu(eng, blocks, pre), ref)


def test_batch_64_blocks(engine_factory, oracle):
    eng = engine_factory(8, 2)
    blocks, pre = _rand_inputs(64, 4242)
    ref = oracle.Oracle(8, 2, check=True).witness_blocks(blocks, pre)
    _assert_same(_run_gpu(eng, blocks, pre), ref)
