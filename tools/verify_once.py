#!/usr/bin/env python3
"""Three hsw_verify_blocks launches over 4,096 blocks -- for profiling that kernel alone, e.g.
rocprofv3 --kernel-trace --pmc TCP_TCC_READ_REQ_sum -- python3 tools/verify_once.py"""
import importlib, os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
hsw = importlib.import_module("halo2-dynamic-sha256_amd")
eng = hsw.WitnessEngine(0, 8, 2)
rng = np.random.default_rng(1)
n = 4096
tb = torch.from_numpy(rng.integers(0, 256, (n, 64), dtype=np.uint8)).cuda()
tp = torch.from_numpy(rng.integers(0, 2**31, (n, 8), dtype=np.int64).astype(np.int32)).cuda()
out = eng.witness_blocks(tb, tp)
for _ in range(3):
    rep = eng.verify_blocks(tb, tp, out)
    print(rep["violations"], round(rep["kernel_ms"], 3))
