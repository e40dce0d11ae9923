#!/usr/bin/env python3
"""Three launches each of the 4,096-block batch in canonical form, Montgomery converted at write-out, and
Montgomery converted at emit time -- for rocprofv3 counter passes over the three kernels.
usage: mont_once.py [lib.so]"""
import ctypes as C, sys
import numpy as np, torch
path = sys.argv[1] if len(sys.argv) > 1 else "halo2-dynamic-sha256_amd/libhsw.so"
n = 4096
L = C.CDLL(path)
L.hsw_engine_create.argtypes = [C.c_int, C.c_void_p, C.c_uint32, C.c_uint32, C.POINTER(C.c_void_p)]
L.hsw_witness_blocks.argtypes = [C.c_void_p] * 3 + [C.c_size_t, C.c_uint64] + [C.c_void_p] * 3 + [C.c_size_t, C.c_void_p, C.c_uint32]
L.hsw_engine_set_option.argtypes = [C.c_void_p, C.c_char_p, C.c_int64]
L.hsw_engine_synchronize.argtypes = [C.c_void_p]
rng = np.random.default_rng(0xC3)
blocks = torch.from_numpy(rng.integers(0, 256, (n, 64), dtype=np.uint8)).cuda()
pre = torch.from_numpy(np.tile(np.array([0x6a09e667, 0xbb67ae85, 0x3c6ef372, 0xa54ff53a, 0x510e527f, 0x9b05688c, 0x1f83d9ab, 0x5be0cd19], dtype=np.uint32).view(np.int32), (n, 1))).cuda()
gate = torch.empty((n * 66308, 4), dtype=torch.int64, device="cuda")
dense = torch.zeros((2, 2060 * n, 4), dtype=torch.int64, device="cuda")
spread = torch.zeros((2, 2060 * n, 4), dtype=torch.int64, device="cuda")
nxt = torch.empty((n, 8), dtype=torch.int32, device="cuda")
for flags, emit in ((0, 0), (1, 0), (1, 1)):
    h = C.c_void_p()
    assert L.hsw_engine_create(0, None, 8, 2, C.byref(h)) == 0
    assert L.hsw_engine_set_option(h, b"mont_emit", emit) == 0
    for _ in range(3):
        assert L.hsw_witness_blocks(h, blocks.data_ptr(), pre.data_ptr(), n, 0, gate.data_ptr(), dense.data_ptr(), spread.data_ptr(), 2060 * n, nxt.data_ptr(), flags) == 0
    L.hsw_engine_synchronize(h)
print("ok")
