"""Lifetimes on the real runtime (the CPU sanitizer build with a stand-in runtime covers the host bookkeeping:
tests/test_host_sanitizers.py): an engine destroyed with launches still queued, and hsw_witness_digests' public
entry -- bit-exact against the oracle, next states delivered into pinned memory by the kernel itself, and an
unpinned host_next_states refused on EVERY call (a cached translation once let a pointer near an
earlier pinned one through)."""
import ctypes as C
import hashlib

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_engine_destroyed_with_work_in_flight(hsw, oracle):
    """hsw_engine_destroy drains the engine's streams before it frees what queued launches use."""
    import torch
    rng = np.random.default_rng(7)
    n = 1024
    blocks = rng.integers(0, 256, (n, 64), dtype=np.uint8)
    pre = np.tile(oracle.INIT_STATE, (n, 1))
    tb, tp = torch.from_numpy(blocks).cuda(), torch.from_numpy(pre.view(np.int32)).cuda()
    ref = oracle.Oracle(8, 2, check=False).witness_blocks(blocks[:2], pre[:2])
    for _ in range(3):
        eng = hsw.WitnessEngine(0, 8, 2)
        out = eng.witness_blocks(tb, tp)          # asynchronous: ~0.5 ms of kernel
        eng.witness_blocks(tb, tp, out=out)
        eng.close()                               # no synchronize() first
        torch.cuda.synchronize()
        assert np.array_equal(out["gate"][: 2 * 66308].cpu().numpy().view(np.uint64), ref["gate"])
        assert np.array_equal(out["next_states"][:2].cpu().numpy().view(np.uint32), ref["next_states"])


def test_witness_digests_public_entry_and_pinned_pointer_check(hsw, oracle):
    import torch
    N = hsw._native
    eng = hsw.WitnessEngine(0, 8, 2, mode=N.HSW_MODE_HALO2_INTERNALS)
    msg, mx = b"abc", 128
    ref = oracle.digest_cells([msg], [mx], None, True)
    blocks_b, init, info = N.digest_prepare(msg, mx)
    nb = mx // 64
    blocks = np.frombuffer(bytes(blocks_b), dtype=np.uint8).reshape(nb, 64).copy()
    pre = np.zeros((nb, 8), dtype=np.uint32)
    st = np.array(init, dtype=np.uint32)
    for b in range(nb):
        pre[b] = st
        st = np.array(oracle.plain_compress(st, blocks[b]), dtype=np.uint32)
    fs = N.FrameShape()
    assert eng.lib.hsw_frame_query(C.byref(eng.shape), mx, 1, C.byref(fs)) == N.HSW_OK
    G, LK = eng.G, eng.lookup_cells
    t = torch
    tb, tp = t.from_numpy(blocks).cuda(), t.from_numpy(pre.view(np.int32)).cuda()
    gate = t.zeros((int(fs.digest_cells) + 1, 4), dtype=t.int64, device="cuda")
    lookup = t.zeros((int(fs.digest_lookups), 4), dtype=t.int64, device="cuda")
    rows = eng.chip_rows(0, nb)
    dense, spread = (t.zeros((2, rows, 4), dtype=t.int64, device="cuda") for _ in range(2))
    nxt = t.zeros((nb, 8), dtype=t.int32, device="cuda")
    d = N.FrameDesc()
    d.input_len, d.first_block, d.n_blocks, d.num_round, d.precomputed_round, d.is_input_range_check = len(msg), 0, nb, 1, 0, 1
    d.prologue_cell, d.zero_cell = 0, int(fs.prologue_cells)
    block_cell = int(fs.prologue_cells) + 1
    d.epilogue_cell = block_cell + nb * G
    d.prologue_lookup, d.epilogue_lookup = 0, int(fs.prologue_lookups) + nb * LK
    a = N.DigestsArgs()
    a.blocks.d_blocks, a.blocks.d_pre_states, a.blocks.n_blocks = tb.data_ptr(), tp.data_ptr(), nb
    a.blocks.d_gate = gate.data_ptr() + 32 * block_cell
    a.blocks.d_chip_dense, a.blocks.d_chip_spread, a.blocks.chip_col_stride = dense.data_ptr(), spread.data_ptr(), rows
    a.blocks.d_next_states = nxt.data_ptr()
    a.blocks.d_lookup = lookup.data_ptr() + 32 * int(fs.prologue_lookups)
    a.descs, a.n_digests = C.addressof(d), 1
    a.d_blocks0, a.d_pre_states0, a.d_next_states0 = tb.data_ptr(), tp.data_ptr(), nxt.data_ptr()
    a.d_gate0, a.d_lookup0 = gate.data_ptr(), lookup.data_ptr()
    pinned = eng.host_empty((64,))
    pinned[:] = 0
    a.host_next_states = pinned.ctypes.data + 64                      # inside a pinned allocation, at an offset
    assert eng.lib.hsw_witness_digests(eng.h, C.byref(a)) == N.HSW_OK
    eng.synchronize()
    assert eng.last_launch()["split"] == 2                             # ONE launch: frames rode on the small-batch kernel
    assert np.array_equal(gate.cpu().numpy().view(np.uint64), ref["gate"])
    assert np.array_equal(lookup.cpu().numpy().view(np.uint64), ref["lookup"])
    assert np.array_equal(dense.cpu().numpy().view(np.uint64), ref["dense"][:, :rows])
    host_next = pinned.view(np.uint32)[16: 16 + 8 * nb].reshape(nb, 8)
    assert np.array_equal(host_next, nxt.cpu().numpy().view(np.uint32))
    sel = host_next[int(info["target_round"]) - 1]
    assert b"".join(int(w).to_bytes(4, "big") for w in sel) == hashlib.sha256(msg).digest()
    # ordinary heap memory right after a pinned pointer has been used: refused, nothing launched
    heap = np.zeros(64, dtype=np.uint32)
    a.host_next_states = heap.ctypes.data
    assert eng.lib.hsw_witness_digests(eng.h, C.byref(a)) == N.HSW_ERR_INVALID_ARG
    assert b"pinned" in eng.lib.hsw_last_error(eng.h)
    a.host_next_states = None
    assert eng.lib.hsw_witness_digests(eng.h, C.byref(a)) == N.HSW_OK
    eng.synchronize()
    eng.close()


def test_gadget_place_keeps_results_and_one_set_of_buffers(hsw, oracle):
    """hsw_gadget_place: candidate allocations of the chip columns, the gadget's own batch timed on each, the fastest
    kept.  Whatever it keeps, the gadget writes the same region afterwards; refused once digests are assigned."""
    import hashlib
    N = hsw._native
    eng = hsw.WitnessEngine(0, 8, 2, mode=N.HSW_MODE_HALO2_INTERNALS)
    sizes, msgs = [128, 64, 192], [b"place", b"", bytes(range(150))]
    cfg = hsw.Sha256DynamicConfig(eng, sizes, is_input_range_check=True, whole_digest=True)
    before = (int(cfg.view().d_chip_dense), int(cfg.view().d_chip_spread))
    ms, kept = cfg.place(4)
    assert len(ms) == 4 and all(m > 0 for m in ms) and 0 <= kept < 4
    v = cfg.view()
    assert int(v.gate_cells) == 0 and int(v.num_limb_sum) == 0                     # left reset
    assert (kept == 0) == ((int(v.d_chip_dense), int(v.d_chip_spread)) == before)
    res = cfg.digest_batch(msgs)
    assert [r.output_bytes for r in res] == [hashlib.sha256(m).digest() for m in msgs]
    st = cfg.streams()
    ref = oracle.digest_cells(msgs, sizes, None, True)
    assert np.array_equal(st["gate"], ref["gate"]) and np.array_equal(st["lookup"], ref["lookup"])
    assert np.array_equal(st["dense"], ref["dense"][:, : st["rows"]]) and np.array_equal(st["spread"], ref["spread"][:, : st["rows"]])
    assert cfg.verify()["violations"] == 0
    with pytest.raises(hsw.HswError):
        cfg.place(2)                                                               # digests assigned in this pass
    cfg.reset()
    assert cfg.place(1)[1] == 0                                                    # one candidate = what is there
    cfg.close()
    eng.close()


def test_alloc_outputs_placed_changes_where_not_what(hsw, oracle):
    """WitnessEngine.alloc_outputs_placed: candidate allocations of the chip columns, the launch timed on each, the
    fastest pair kept -- the same bytes as with any other buffers."""
    import torch
    rng = np.random.default_rng(8)
    n = 40
    blocks = rng.integers(0, 256, (n, 64), dtype=np.uint8)
    pre = rng.integers(0, 2**32, (n, 8), dtype=np.uint64).astype(np.uint32)
    tb, tp = torch.from_numpy(blocks).cuda(), torch.from_numpy(pre.view(np.int32)).cuda()
    eng = hsw.WitnessEngine(0, 8, 2)
    out, rep = eng.alloc_outputs_placed(tb, tp, cursor0=6, candidates=3, spacer_bytes=1 << 20, gate_candidates=2)
    assert rep["gate_candidates"] == ["range of 4 GiB pieces", "plain"] and len(rep["chip_candidates"]) == 3
    assert len(rep["kernel_ms"]) == 2 and all(len(row) == 3 for row in rep["kernel_ms"])
    assert 0 <= rep["kept"][0] < 2 and 0 <= rep["kept"][1] < 3
    eng.witness_blocks(tb, tp, cursor0=6, out=out)
    eng.synchronize()
    ref = oracle.Oracle(8, 2, check=False).witness_blocks(blocks, pre, cursor0=6)
    assert np.array_equal(out["gate"].cpu().numpy().view(np.uint64), ref["gate"])
    assert np.array_equal(out["dense"].cpu().numpy().view(np.uint64), ref["dense"])
    assert np.array_equal(out["spread"].cpu().numpy().view(np.uint64), ref["spread"])
    eng.close()


def test_device_ranges_hold_witness_streams(hsw, oracle):
    """hsw_device_alloc: one virtual range backed by several physical allocations -- a witness launch writes through
    the piece boundaries like anywhere else."""
    import torch
    N = hsw._native
    rng = np.random.default_rng(12)
    n = 24                                                   # 24 x 66,308 cells x 32 B = 50.9 MB: pieces of 4 MiB -> 13 allocations
    blocks = rng.integers(0, 256, (n, 64), dtype=np.uint8)
    pre = rng.integers(0, 2**32, (n, 8), dtype=np.uint64).astype(np.uint32)
    tb, tp = torch.from_numpy(blocks).cuda(), torch.from_numpy(pre.view(np.int32)).cuda()
    eng = hsw.WitnessEngine(0, 8, 2)
    out = eng.alloc_outputs(n)
    out["gate"] = eng.device_empty(out["gate"].shape, chunk_bytes=4 << 20)
    out["dense"] = eng.device_empty(out["dense"].shape, chunk_bytes=2 << 20).zero_()
    eng.witness_blocks(tb, tp, out=out)
    eng.synchronize()
    ref = oracle.Oracle(8, 2, check=False).witness_blocks(blocks, pre)
    assert np.array_equal(out["gate"].cpu().numpy().view(np.uint64), ref["gate"])
    assert np.array_equal(out["dense"].cpu().numpy().view(np.uint64), ref["dense"])
    del out
    p = C.c_void_p()
    assert eng.lib.hsw_device_alloc(0, 0, 0, C.byref(p)) == N.HSW_ERR_INVALID_ARG
    on_heap = np.zeros(4, dtype=np.uint64)
    assert eng.lib.hsw_device_free(C.c_void_p(on_heap.ctypes.data)) == N.HSW_ERR_INVALID_ARG      # not a range of ours
    # (freeing a live range is exercised under the stub runtime, tests/cpp/host_lifecycle.cpp: on this ROCm a probe
    #  that unmapped ranges between launches ended in GPU memory faults, so the GPU suite never does)
    eng.close()
