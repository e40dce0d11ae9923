"""The small-batch kernel (csrc/hsw_small.hpp): what every launch of <= 128 blocks uses -- the reference's own
bench circuit (benches/digest.rs: ONE 16-block digest) first of all.  37 waves per block, each running one
SUB-UNIT program over 16 units with a register-latched chain; frames of whole digests in the same launch.
Same cells as hsw_expand_kernel, bit for bit, and as the oracle."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _inputs(n, seed):
    rng = np.random.default_rng(seed)
    return (rng.integers(0, 256, (n, 64), dtype=np.uint8),
            rng.integers(0, 2**32, (n, 8), dtype=np.uint64).astype(np.uint32))


def _run(eng, blocks, pre, cursor0=0, flags=0):
    import torch
    out = eng.witness_blocks(torch.from_numpy(blocks).cuda(), torch.from_numpy(pre.view(np.int32)).cuda(),
                             cursor0=cursor0, flags=flags)
    eng.synchronize()
    return {k: (v.cpu().numpy() if hasattr(v, "cpu") else v) for k, v in out.items()}


@pytest.mark.parametrize("n", [1, 2, 16, 31, 33, 128, 129])
@pytest.mark.parametrize("ncols,cursor0", [(2, 0), (3, 5)])
def test_small_kernel_gives_identical_streams(hsw, oracle, n, ncols, cursor0):
    """Default engine: <= 128 blocks -> hsw_small_kernel, 129 -> hsw_expand_kernel; both against the oracle."""
    eng = hsw.WitnessEngine(0, 8, ncols)
    blocks, pre = _inputs(n, 900 + n)
    blocks[0] = 0
    pre[0] = 0                                    # neg(0) cells, zero words
    if n > 1:
        blocks[1] = 0xFF
        pre[1] = 0xFFFFFFFF                       # r_spread = 2^64 - 1, every carry
    ref = oracle.Oracle(8, ncols, check=True).witness_blocks(blocks, pre, cursor0=cursor0)
    got = _run(eng, blocks, pre, cursor0=cursor0)
    li = eng.last_launch()
    assert li["split"] == (2 if n <= 128 else 0) and ("hsw_small_kernel" in li["kernel"]) == (n <= 128)
    assert li["grid"] == (37 * n if n <= 128 else li["parts"] * n)
    g = got["gate"].view(np.uint64)
    if not np.array_equal(g, ref["gate"]):
        bad = np.nonzero((g != ref["gate"]).any(axis=1))[0]
        raise AssertionError("gate differs at %d cells, first %d (block %d, cell %d): %s vs %s" % (
            len(bad), bad[0], bad[0] // eng.G, bad[0] % eng.G, g[bad[0]], ref["gate"][bad[0]]))
    assert np.array_equal(got["dense"].view(np.uint64), ref["dense"])
    assert np.array_equal(got["spread"].view(np.uint64), ref["spread"])
    assert np.array_equal(got["next_states"].view(np.uint32), ref["next_states"])
    eng.close()


@pytest.mark.parametrize("flags_name", ["montgomery", "compact"])
def test_small_kernel_representations(hsw, oracle, flags_name):
    N = hsw._native
    flags = {"montgomery": N.HSW_REPR_MONTGOMERY, "compact": N.HSW_REPR_COMPACT64}[flags_name]
    eng = hsw.WitnessEngine(0, 8, 2)
    blocks, pre = _inputs(5, 4711)
    ref = oracle.Oracle(8, 2, check=True).witness_blocks(blocks, pre, cursor0=7)
    got = _run(eng, blocks, pre, cursor0=7, flags=flags)
    assert eng.last_launch()["split"] == 2
    if flags_name == "compact":
        g = got["gate"].view(np.uint64).reshape(-1)[: 5 * eng.G]
        exp = ref["gate"][:, 0].copy()
        neg = np.nonzero(ref["gate"][:, 1:].any(axis=1))[0]
        exp[neg] = np.uint64(0x43e1f593f0000001) - exp[neg]
        assert np.array_equal(g, exp)
        assert np.array_equal(got["dense"].view(np.uint64).reshape(2, -1), ref["dense"][:, :, 0])
        assert np.array_equal(got["spread"].view(np.uint64).reshape(2, -1), ref["spread"][:, :, 0])
    else:
        assert np.array_equal(got["gate"].view(np.uint64), oracle.to_montgomery(ref["gate"]))
        assert np.array_equal(got["dense"].view(np.uint64), oracle.to_montgomery(ref["dense"]))
        assert np.array_equal(got["spread"].view(np.uint64), oracle.to_montgomery(ref["spread"]))
    assert np.array_equal(got["next_states"].view(np.uint32), ref["next_states"])
    eng.close()


def test_small_kernel_on_a_large_batch_and_both_kernels_agree(hsw, oracle):
    """"split" = 2 forces the small-batch kernel for any batch size: 70 blocks, against hsw_expand_kernel
    ("split" = 0) on the same inputs, bit for bit, and a sample against the oracle."""
    eng = hsw.WitnessEngine(0, 8, 2)
    blocks, pre = _inputs(70, 2718)
    eng.set_option("split", 2)
    a = _run(eng, blocks, pre, cursor0=1)
    assert eng.last_launch()["split"] == 2 and eng.last_launch()["grid"] == 70 * 37
    eng.set_option("split", 0)
    b = _run(eng, blocks, pre, cursor0=1)
    assert eng.last_launch()["split"] == 0
    for k in ("gate", "dense", "spread", "next_states"):
        assert np.array_equal(a[k], b[k]), k
    ref = oracle.Oracle(8, 2, check=False).witness_blocks(blocks[:3], pre[:3], cursor0=1)
    assert np.array_equal(a["gate"].view(np.uint64)[: 3 * eng.G], ref["gate"])
    eng.close()


def test_small_kernel_skip_flags_and_neighbours(hsw, oracle):
    """HSW_SKIP_GATE / HSW_SKIP_CHIP and the cells of the first / last chip row that belong to the neighbouring
    calls (cursor not a multiple of ncols) -- the scatter writes of the sub-unit waves must respect both."""
    import torch
    N = hsw._native
    eng = hsw.WitnessEngine(0, 8, 3)
    blocks, pre = _inputs(2, 55)
    ref = oracle.Oracle(8, 3).witness_blocks(blocks, pre, cursor0=2)
    tb, tp = torch.from_numpy(blocks).cuda(), torch.from_numpy(pre.view(np.int32)).cuda()
    out = eng.alloc_outputs(2, cursor0=2)
    for k in ("gate", "dense", "spread"):
        out[k].fill_(-7)
    eng.witness_blocks(tb, tp, cursor0=2, out=out, flags=N.HSW_SKIP_GATE)
    eng.synchronize()
    assert eng.last_launch()["split"] == 2
    assert (out["gate"] == -7).all()
    d = out["dense"].cpu().numpy().view(np.uint64)
    assert (d[0, 0] == np.uint64(-7 & 0xFFFFFFFFFFFFFFFF)).all() and (d[1, 0] == np.uint64(-7 & 0xFFFFFFFFFFFFFFFF)).all()
    assert np.array_equal(d[2, 0], ref["dense"][2, 0])
    last = (2 + 2 * eng.limb_calls) % 3           # cells of the last row owned by this call
    rows = ref["rows"]
    for c in range(3):
        if last and c >= last:
            assert (d[c, rows - 1] == np.uint64(-7 & 0xFFFFFFFFFFFFFFFF)).all()
    out["dense"].fill_(-7)
    eng.witness_blocks(tb, tp, cursor0=2, out=out, flags=N.HSW_SKIP_CHIP)
    eng.synchronize()
    assert (out["dense"] == -7).all()
    assert np.array_equal(out["gate"].cpu().numpy().view(np.uint64), ref["gate"])
    eng.close()


@pytest.mark.parametrize("mont", [False, True])
def test_one_launch_whole_digest_equals_two_launches(hsw, oracle, mont):
    """The reference's bench circuit (1 x 1,024 B, input range checks, 9 columns of 131,063 rows) and its
    TestCircuit (2 x 128 B, 3 columns) as whole regions: by default ONE launch (frame waves ride on the
    small-batch kernel's grid, inputs read in place, next states written into pinned memory) -- against the
    same context with "split" = 0 (expansion launch + hsw_frame_kernel + copies), every stream bit for bit,
    and verified on the device."""
    import hashlib
    N = hsw._native
    for sizes, msgs, rows in (([1024], [bytes([1] * 56)], (1 << 17) - 9),
                              ([128, 128], [b"abc", b""], (1 << 17) - 9),
                              ([256, 256, 256], [bytes(range(200)), b"x" * 100, b""], None)):
        got = []
        for split in (-1, 0):
            eng = hsw.WitnessEngine(0, 8, 2, mode=N.HSW_MODE_HALO2_INTERNALS)
            eng.set_option("split", split)
            cfg = hsw.Sha256DynamicConfig(eng, sizes, True, whole_digest=True)
            if mont:
                cfg.set_repr(N.HSW_REPR_MONTGOMERY)
            if rows:
                cfg.set_columns(rows)
            rs = cfg.digest_batch(msgs)
            li = eng.last_launch()
            assert li["split"] == (2 if split < 0 else 0)
            if split < 0:
                assert li["grid"] > 37 * li["n_blocks"]          # frame waves in the same grid
            assert [r.output_bytes for r in rs] == [hashlib.sha256(m).digest() for m in msgs]
            rep = cfg.verify()
            assert rep["violations"] == 0, rep
            got.append(cfg.streams())
            cfg.close()
            eng.close()
        for k in ("gate", "lookup", "dense", "spread"):
            assert np.array_equal(got[0][k], got[1][k]), (sizes, k)


@pytest.mark.parametrize("helpers", [0, 1, 2, 3, 4])
@pytest.mark.parametrize("internals", [False, True])
@pytest.mark.parametrize("flags_name", ["canonical", "montgomery", "compact"])
def test_helper_waves(hsw, oracle, helpers, internals, flags_name):
    """A workgroup of the small-batch kernel is "helpers" waves (default 4 for Montgomery cells, else 4 / 2 / 1 up to 16 / 64 / 128 blocks): wave 0 emits, and all of them share
    the conversion and write-out of every tile.  Any count gives the cells of the streaming kernel, and of
    the oracle."""
    import torch
    N = hsw._native
    flags = {"canonical": 0, "montgomery": N.HSW_REPR_MONTGOMERY, "compact": N.HSW_REPR_COMPACT64}[flags_name]
    mode = N.HSW_MODE_HALO2_INTERNALS if internals else N.HSW_MODE_DEFAULT
    eng = hsw.WitnessEngine(0, 8, 3, mode=mode)
    eng.set_option("helpers", helpers)
    n = 18
    blocks, pre = _inputs(n, 8100 + helpers)
    blocks[0] = 0
    pre[0] = 0
    blocks[1] = 0xFF
    pre[1] = 0xFFFFFFFF
    tb, tp = torch.from_numpy(blocks).cuda(), torch.from_numpy(pre.view(np.int32)).cuda()

    def run():
        out = eng.witness_blocks_ex(tb, tp, cursor0=4, flags=flags, want_lookup=internals)
        eng.synchronize()
        return {k: v.cpu().numpy() for k, v in out.items() if hasattr(v, "cpu")}

    got = run()
    li = eng.last_launch()
    h = helpers if helpers else (4 if flags_name == "montgomery" else 2)      # 18 blocks: 2 (17 .. 64) unless Montgomery
    assert li["split"] == 2 and li["parts"] == 37 * h and li["grid"] == 37 * n
    eng.set_option("split", 0)
    ref = run()
    assert eng.last_launch()["split"] == 0
    for k in ("gate", "dense", "spread", "next_states") + (("lookup",) if internals else ()):
        assert np.array_equal(got[k], ref[k]), k
    if flags_name != "compact":
        o = oracle.Oracle(8, 3, check=False, internals=internals).witness_blocks(blocks[:2], pre[:2], cursor0=4)
        exp = oracle.to_montgomery(o["gate"]) if flags_name == "montgomery" else o["gate"]
        assert np.array_equal(got["gate"].view(np.uint64)[: 2 * eng.G], exp)
    eng.close()


@pytest.mark.parametrize("nblk,flags_name", [(1, "canonical"), (16, "canonical"), (16, "montgomery"), (32, "canonical")])
def test_chained_single_launch(hsw, oracle, nblk, flags_name):
    """HSW_CHAINED: the blocks are ONE message and d_pre_states holds its initial state only -- every wave
    walks the message to its own block, no chain pre-pass, no second launch (BASELINE configs[1]: one
    16-block message).  Streams, chip columns and every block's next state against the oracle's digest() run."""
    import hashlib
    import torch
    N = hsw._native
    eng = hsw.WitnessEngine(0, 8, 2)
    m = bytes(((i * 131 + 7) % 256) for i in range(64 * nblk - 9))
    ref = oracle.Oracle(8, 2, check=True).digest(m, 64 * nblk, want_streams=True)
    assert ref["digest"] == hashlib.sha256(m).digest()
    flags = N.HSW_CHAINED | (N.HSW_REPR_MONTGOMERY if flags_name == "montgomery" else 0)
    conv = oracle.to_montgomery if flags_name == "montgomery" else (lambda x: x)
    tb = torch.from_numpy(ref["blocks"].copy()).cuda()
    init = torch.from_numpy(oracle.INIT_STATE.view(np.int32).copy()).cuda()
    out = eng.witness_blocks(tb, init, flags=flags)
    eng.synchronize()
    assert eng.last_launch()["split"] == 2
    assert np.array_equal(out["gate"].cpu().numpy().view(np.uint64), conv(ref["gate"]))
    assert np.array_equal(out["dense"].cpu().numpy().view(np.uint64), conv(ref["dense"]))
    assert np.array_equal(out["spread"].cpu().numpy().view(np.uint64), conv(ref["spread"]))
    assert np.array_equal(out["next_states"].cpu().numpy().view(np.uint32), ref["next_states"])
    # a custom initial state (precomputed prefix, lib.rs:153-160)
    init2 = np.arange(8, dtype=np.uint32) * np.uint32(0x01234567)
    st = init2.copy()
    nxt = []
    for j in range(min(nblk, 3)):
        st = oracle.plain_compress(st, ref["blocks"][j])
        nxt.append(st.copy())
    out2 = eng.witness_blocks(tb[:3] if nblk >= 3 else tb, torch.from_numpy(init2.view(np.int32).copy()).cuda(), flags=N.HSW_CHAINED)
    eng.synchronize()
    assert np.array_equal(out2["next_states"].cpu().numpy().view(np.uint32)[: len(nxt)], np.stack(nxt))
    # not a small-batch launch: refused, with a pointer to the pre-pass
    big = torch.zeros((40, 64), dtype=torch.uint8, device="cuda")
    with pytest.raises(hsw.HswError) as ei:
        eng.witness_blocks(big, init, flags=N.HSW_CHAINED)
    assert ei.value.status == N.HSW_ERR_UNSUPPORTED
    eng.close()
