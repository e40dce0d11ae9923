#!/usr/bin/env python3
"""Fine batch-size sweep, several libs interleaved (tuning aid)."""
import ctypes as C, sys, json
import numpy as np, torch
libs = []
for spec in sys.argv[2:]:
    path, parts, tile = (spec.split(":") + ["", ""])[:3]
    L = C.CDLL(path)
    L.hsw_engine_create.argtypes = [C.c_int, C.c_void_p, C.c_uint32, C.c_uint32, C.POINTER(C.c_void_p)]
    L.hsw_witness_blocks.argtypes = [C.c_void_p] * 3 + [C.c_size_t, C.c_uint64] + [C.c_void_p] * 3 + [C.c_size_t, C.c_void_p, C.c_uint32]
    L.hsw_last_kernel_ms.argtypes = [C.c_void_p, C.POINTER(C.c_float)]
    L.hsw_set_timing.argtypes = [C.c_void_p, C.c_int]
    h = C.c_void_p(); assert L.hsw_engine_create(0, None, 8, 2, C.byref(h)) == 0
    L.hsw_set_timing(h, 1)
    if parts:
        L.hsw_engine_set_option.argtypes = [C.c_void_p, C.c_char_p, C.c_int64]
        assert L.hsw_engine_set_option(h, b"parts", int(parts)) == 0
        if tile:
            assert L.hsw_engine_set_option(h, b"tile", int(tile)) == 0
    libs.append((spec.split("/")[-1], L, h))
nmax = 8192
rng = np.random.default_rng(0xC3)
blocks = torch.from_numpy(rng.integers(0, 256, (nmax, 64), dtype=np.uint8)).cuda()
pre = torch.from_numpy(rng.integers(0, 2**31, (nmax, 8), dtype=np.int64).astype(np.int32)).cuda()
gate = torch.empty((nmax * 66308, 4), dtype=torch.int64, device="cuda")
dense = torch.zeros((2, 2060 * nmax, 4), dtype=torch.int64, device="cuda")
spread = torch.zeros((2, 2060 * nmax, 4), dtype=torch.int64, device="cuda")
def run(L, h, n):
    assert L.hsw_witness_blocks(h, blocks.data_ptr(), pre.data_ptr(), n, 0, gate.data_ptr(), dense.data_ptr(), spread.data_ptr(), 2060 * nmax, None, 0) == 0
    ms = C.c_float(); assert L.hsw_last_kernel_ms(h, C.byref(ms)) == 0
    return ms.value
for n in [int(x) for x in sys.argv[1].split(",")]:
    row = {"n": n}
    for name, L, h in libs:
        run(L, h, n)
    ts = {name: [] for name, _, _ in libs}
    for _ in range(7):
        for name, L, h in libs:
            ts[name].append(run(L, h, n))
    for name in ts:
        row[name] = round(2385664 * n / 1e6 / float(np.median(ts[name])))
    print(json.dumps(row), flush=True)
