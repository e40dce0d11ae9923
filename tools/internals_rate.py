#!/usr/bin/env python3
"""Throughput of the halo2-internals mode (range-check cells + lookup column) and of whole-digest batches."""
import ctypes as C, importlib, os, sys, time, json, hashlib
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
hsw = importlib.import_module("halo2-dynamic-sha256_amd")
N = hsw._native
n = 4096
rng = np.random.default_rng(0xC3)
blocks = torch.from_numpy(rng.integers(0, 256, (n, 64), dtype=np.uint8)).cuda()
pre = torch.from_numpy(rng.integers(0, 2**31, (n, 8), dtype=np.int64).astype(np.int32)).cuda()
for mode, name in [(0, "default"), (N.HSW_MODE_HALO2_INTERNALS, "internals")]:
    eng = hsw.WitnessEngine(0, 8, 2, mode=mode)
    eng.set_timing(True)
    for flags, fname in [(0, "canonical"), (N.HSW_REPR_MONTGOMERY, "montgomery")]:
        out = eng.alloc_outputs(n, 0, flags)
        lookup = torch.zeros((n * eng.lookup_cells, 4), dtype=torch.int64, device="cuda") if mode else None
        ms = []
        for i in range(8):
            if mode:
                a = N.WitnessArgs()
                a.d_blocks, a.d_pre_states, a.n_blocks = blocks.data_ptr(), pre.data_ptr(), n
                a.spread_cursor0, a.d_gate = 0, out["gate"].data_ptr()
                a.d_chip_dense, a.d_chip_spread = out["dense"].data_ptr(), out["spread"].data_ptr()
                a.chip_col_stride = out["dense"].shape[1]
                a.d_next_states, a.d_lookup, a.flags = out["next_states"].data_ptr(), lookup.data_ptr(), flags
                assert eng.lib.hsw_witness_blocks_ex(eng.h, C.byref(a)) == 0
            else:
                eng.witness_blocks(blocks, pre, out=out, flags=flags)
            if i >= 2:
                ms.append(eng.last_kernel_ms())
        m = float(np.median(ms))
        cells = eng.G + 8240 + (eng.lookup_cells if mode else 0)
        print(json.dumps({"mode": name, "repr": fname, "ms": round(m, 4), "blocks_per_s": round(n / m * 1e3),
                          "GBps": round(cells * 32 * n / m / 1e6)}), flush=True)
        del out, lookup
    eng.close()

# whole-digest batch: 4096 single-block digests
eng = hsw.WitnessEngine(0, 8, 2, mode=N.HSW_MODE_HALO2_INTERNALS)
msgs = [rng.integers(0, 256, 55, dtype=np.uint8).tobytes() for _ in range(n)]
cfg = hsw.Sha256DynamicConfig(eng, [64] * n, False, whole_digest=True)
for _ in range(2):
    cfg.reset(); res = cfg.digest_batch(msgs)
t0 = time.perf_counter()
for _ in range(5):
    cfg.reset(); res = cfg.digest_batch(msgs)
dt = (time.perf_counter() - t0) / 5
assert res[77].output_bytes == hashlib.sha256(msgs[77]).digest()
print(json.dumps({"whole_digest_batch_4096x1": {"ms": round(dt * 1e3, 3), "digests_per_s": round(n / dt)}}))

# the same through the C ABI only (no per-result Python marshalling)
bufs = [(C.c_uint8 * len(m)).from_buffer_copy(m) for m in msgs]
ptrs = (C.c_void_p * n)(*[C.addressof(b) for b in bufs])
lens = (C.c_size_t * n)(*[len(m) for m in msgs])
pres = (C.c_size_t * n)(*([0] * n))
resv = (N.HashResult * n)()
for whole in (True, False):
    c2 = hsw.Sha256DynamicConfig(eng, [64] * n, False, whole_digest=whole)
    ts = []
    for i in range(7):
        eng.lib.hsw_gadget_reset(c2.h)
        t0 = time.perf_counter()
        assert eng.lib.hsw_gadget_digest_batch(c2.h, n, ptrs, lens, pres, resv) == 0
        ts.append(time.perf_counter() - t0)
    dt = float(np.median(ts[2:]))
    assert bytes(resv[77].output_bytes) == hashlib.sha256(msgs[77]).digest()
    print(json.dumps({"c_abi_digest_batch_4096x1": {"whole_digest": whole, "ms": round(dt * 1e3, 3), "digests_per_s": round(n / dt)}}))
    c2.close()
