#!/usr/bin/env python3
"""Interleaved A/B timing of libhsw builds in ONE process (tuning aid, not product).
usage: ab.py n_blocks reps lib.so[:parts[:tile[:flags[:opt=value,opt=value]]]] ...   (flags: HSW_REPR_* bits)"""
import ctypes as C, sys, json
import numpy as np, torch
import os
n = int(sys.argv[1]); reps = int(sys.argv[2])
MODE = int(os.environ.get("HSW_AB_MODE", "0"))        # 1 = HSW_MODE_HALO2_INTERNALS (G = 69,348 cells per block)
variants = []
for spec in sys.argv[3:]:
    path, parts, tile, vflags, opts = (spec.split(":") + ["", "", "", ""])[:5]
    L = C.CDLL(path)
    L.hsw_engine_create.argtypes = [C.c_int, C.c_void_p, C.c_uint32, C.c_uint32, C.POINTER(C.c_void_p)]
    L.hsw_witness_blocks.argtypes = [C.c_void_p] * 3 + [C.c_size_t, C.c_uint64] + [C.c_void_p] * 3 + [C.c_size_t, C.c_void_p, C.c_uint32]
    L.hsw_last_kernel_ms.argtypes = [C.c_void_p, C.POINTER(C.c_float)]
    L.hsw_set_timing.argtypes = [C.c_void_p, C.c_int]
    h = C.c_void_p()
    L.hsw_engine_create_ex.argtypes = [C.c_int, C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.POINTER(C.c_void_p)]
    assert L.hsw_engine_create_ex(0, None, 8, 2, MODE, C.byref(h)) == 0
    L.hsw_set_timing(h, 1)
    if parts:
        L.hsw_engine_set_option.argtypes = [C.c_void_p, C.c_char_p, C.c_int64]
        assert L.hsw_engine_set_option(h, b"parts", int(parts)) == 0
        if tile:
            assert L.hsw_engine_set_option(h, b"tile", int(tile)) == 0
    for kv in filter(None, opts.split(",")):
        L.hsw_engine_set_option.argtypes = [C.c_void_p, C.c_char_p, C.c_int64]
        k, v = kv.split("=")
        assert L.hsw_engine_set_option(h, k.encode(), int(v)) == 0, kv
    variants.append((spec, L, h, int(vflags or 0)))
rng = np.random.default_rng(0xC3)
blocks = torch.from_numpy(rng.integers(0, 256, (n, 64), dtype=np.uint8)).cuda()
pre = torch.from_numpy(np.tile(np.array([0x6a09e667, 0xbb67ae85, 0x3c6ef372, 0xa54ff53a, 0x510e527f, 0x9b05688c, 0x1f83d9ab, 0x5be0cd19], dtype=np.uint32).view(np.int32), (n, 1))).cuda()
G = 69348 if MODE else 66308
if os.environ.get("HSW_AB_RANGED") == "1":       # the gate stream in an hsw_device_alloc range (4 GiB physical pieces)
    class _Ptr:
        def __init__(self, p, cells): self.p, self.cells = p, cells
        def data_ptr(self): return self.p
        def numel(self): return self.cells * 4
    _p = C.c_void_p()
    _L = variants[0][1]
    _L.hsw_device_alloc.argtypes = [C.c_int, C.c_size_t, C.c_size_t, C.POINTER(C.c_void_p)]
    assert _L.hsw_device_alloc(0, n * G * 32, 0, C.byref(_p)) == 0
    gate = _Ptr(_p.value, n * G)
else:
    gate = torch.empty((n * G, 4), dtype=torch.int64, device="cuda")  # sized for 32-byte cells; compact runs use a quarter
dense = torch.zeros((2, 2060 * n, 4), dtype=torch.int64, device="cuda")
spread = torch.zeros((2, 2060 * n, 4), dtype=torch.int64, device="cuda")
nxt = torch.empty((n, 8), dtype=torch.int32, device="cuda")
def run(L, h, fl=0):
    rc = L.hsw_witness_blocks(h, blocks.data_ptr(), pre.data_ptr(), n, 0, gate.data_ptr(), dense.data_ptr(), spread.data_ptr(), 2060 * n, nxt.data_ptr(), fl)
    assert rc == 0, rc
    ms = C.c_float(); assert L.hsw_last_kernel_ms(h, C.byref(ms)) == 0
    return ms.value
for _ in range(3):
    for _, L, h, fl in variants: run(L, h, fl)
times = {spec: [] for spec, _, _, _ in variants}
order = list(range(len(variants)))
shuffler = np.random.default_rng(7)
for _ in range(reps):
    shuffler.shuffle(order)            # a fixed A, B, A, B order biases the later variant by ~2 % (measured)
    for k in order:
        spec, L, h, fl = variants[k]
        times[spec].append(run(L, h, fl))
L0, h0 = variants[-1][1], variants[-1][2]
if hasattr(L0, "hsw_fill_calibrate"):
    L0.hsw_fill_calibrate.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.POINTER(C.c_float)]
    fm = []
    for _ in range(5):
        ms = C.c_float(); assert L0.hsw_fill_calibrate(h0, gate.data_ptr(), gate.numel() * 8, C.byref(ms)) == 0
        fm.append(ms.value)
    print(json.dumps(dict(variant="plain fill of the gate buffer", GBps_median=gate.numel() * 8 / 1e6 / float(np.median(fm)), GBps_best=gate.numel() * 8 / 1e6 / min(fm))))
for spec in times:
    t = np.array(times[spec]); gb = 2385664 * n / 1e6
    print(json.dumps(dict(variant=spec, n=n, median_ms=float(np.median(t)), min_ms=float(t.min()), GBps_median=gb / float(np.median(t)), GBps_best=gb / float(t.min()))))
