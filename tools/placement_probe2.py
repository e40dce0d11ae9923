#!/usr/bin/env python3
"""Kernel time of the 4,096-block launch vs where the chip columns sit RELATIVE to the gate stream, both carved
from ONE allocation (tools/placement_probe.py found: the same launch takes 1.64 or 1.77 ms depending on the pair of
buffers, the pure fill and the launch without chip columns do not care)."""
import ctypes as C, sys, os
import numpy as np, torch
L = C.CDLL("halo2-dynamic-sha256_amd/libhsw.so")
L.hsw_engine_create_ex.argtypes = [C.c_int, C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.POINTER(C.c_void_p)]
L.hsw_witness_blocks.argtypes = [C.c_void_p] * 3 + [C.c_size_t, C.c_uint64] + [C.c_void_p] * 3 + [C.c_size_t, C.c_void_p, C.c_uint32]
L.hsw_last_kernel_ms.argtypes = [C.c_void_p, C.POINTER(C.c_float)]
L.hsw_set_timing.argtypes = [C.c_void_p, C.c_int]
h = C.c_void_p(); assert L.hsw_engine_create_ex(0, None, 8, 2, 0, C.byref(h)) == 0
L.hsw_set_timing(h, 1)
n = 4096
rng = np.random.default_rng(0xC3)
blocks = torch.from_numpy(rng.integers(0, 256, (n, 64), dtype=np.uint8)).cuda()
pre = torch.from_numpy(np.tile(np.array([0x6a09e667, 0xbb67ae85, 0x3c6ef372, 0xa54ff53a, 0x510e527f, 0x9b05688c, 0x1f83d9ab, 0x5be0cd19], dtype=np.uint32).view(np.int32), (n, 1))).cuda()
nxt = torch.empty((n, 8), dtype=torch.int32, device="cuda")
GATE = n * 66308 * 32
COL = 2 * 2060 * n * 32                       # one chip array: 2 columns
TOTAL = 24 << 30
big = torch.empty((TOTAL,), dtype=torch.uint8, device="cuda")
base = big.data_ptr()
assert base % (2 << 20) == 0, hex(base)
def run(gate_off, dense_off, spread_off):
    for o, sz in ((gate_off, GATE), (dense_off, COL), (spread_off, COL)):
        assert 0 <= o and o + sz <= TOTAL, (o, sz)
    ts = []
    for i in range(9):
        assert L.hsw_witness_blocks(h, blocks.data_ptr(), pre.data_ptr(), n, 0, base + gate_off, base + dense_off, base + spread_off, 2060 * n, nxt.data_ptr(), 0) == 0
        ms = C.c_float(); L.hsw_last_kernel_ms(h, C.byref(ms)); ts.append(ms.value)
    return float(np.median(ts[2:]))
MiB = 1 << 20
print("gate stream at offset 0 (%.2f GB); dense columns at D, spread columns right behind" % (GATE / 1e9))
D0 = ((GATE + 2 * MiB - 1) // (2 * MiB)) * 2 * MiB + 512 * MiB
step = int(sys.argv[1]) if len(sys.argv) > 1 else 2 * MiB
cnt = int(sys.argv[2]) if len(sys.argv) > 2 else 160
res = []
for i in range(cnt):
    D = D0 + i * step
    if D + 2 * COL + 2 * MiB > TOTAL: break
    res.append((D, run(0, D, D + COL + 2 * MiB)))
print("step %d KiB from D0 = %d MiB:" % (step >> 10, D0 >> 20))
print(" ".join("%.3f" % t for _, t in res))
# the two chip arrays moved independently: dense fixed at D0, spread at D0 + COL + k * step
res2 = [run(0, D0, D0 + COL + 2 * MiB + i * step) for i in range(min(cnt, 64))]
print("dense fixed, spread moving by the same step:")
print(" ".join("%.3f" % t for t in res2))
