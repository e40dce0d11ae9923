#!/usr/bin/env python3
"""Does the placement of the output buffer matter?  Same process, one big pool,
the gate stream written at different base offsets / with different neighbours."""
import ctypes as C, importlib, os, sys, json
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
hsw = importlib.import_module("halo2-dynamic-sha256_amd")
eng = hsw.WitnessEngine(0, 8, 2)
eng.set_timing(True)
n = 4096
G = eng.G
rng = np.random.default_rng(0xC3)
blocks = torch.from_numpy(rng.integers(0, 256, (n, 64), dtype=np.uint8)).cuda()
pre = torch.from_numpy(rng.integers(0, 2**31, (n, 8), dtype=np.int64).astype(np.int32)).cuda()
pool = torch.empty(24 * 2**30, dtype=torch.uint8, device="cuda")
dense = torch.zeros((2, 2060 * n, 4), dtype=torch.int64, device="cuda")
spread = torch.zeros((2, 2060 * n, 4), dtype=torch.int64, device="cuda")
base = pool.data_ptr()
print("pool base %x (mod 2MiB = %d)" % (base, base % (2 << 20)))
def run(off):
    ms = []
    for i in range(8):
        rc = eng.lib.hsw_witness_blocks(eng.h, blocks.data_ptr(), pre.data_ptr(), n, 0, base + off, dense.data_ptr(),
                                        spread.data_ptr(), 2060 * n, None, 0)
        assert rc == 0
        if i >= 2:
            ms.append(eng.last_kernel_ms())
    return float(np.median(ms))
offs = [0, 32, 64, 96, 128, 160, 256, 4096, 65536, 1 << 20, 2 << 20, (2 << 20) + 4096, 1 << 30, (1 << 30) + 12345 * 16, 5 << 30, 10 << 30, 13 << 30]
for rep in range(2):
    for off in offs:
        print(json.dumps({"offset": off, "ms": round(run(off), 4)}), flush=True)
