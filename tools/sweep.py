#!/usr/bin/env python3
"""Tuning sweep (not part of the product): kernel time vs batch size / parts."""
import importlib, sys, os, json
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
hsw = importlib.import_module("halo2-dynamic-sha256_amd")
eng = hsw.WitnessEngine(0, 8, 2)
eng.set_timing(True)
alg = int(eng.shape.algorithmic_bytes_per_block)
rng = np.random.default_rng(1)
res = []
for n, parts_list in [(16, [1, 2, 4, 8, 16]), (256, [1, 2, 4, 8]), (1792, [1, 2]), (2048, [1, 2]), (3584, [1, 2]), (4096, [1, 2, 4]), (8192, [1, 2]), (16384, [1])]:
    blocks = torch.from_numpy(rng.integers(0, 256, (n, 64), dtype=np.uint8)).cuda()
    pre = torch.from_numpy(rng.integers(0, 2**31, (n, 8), dtype=np.int64).astype(np.int32)).cuda()
    out = eng.alloc_outputs(n)
    for parts in parts_list:
        eng.set_option("parts", parts)
        for _ in range(3):
            eng.witness_blocks(blocks, pre, out=out)
        ms = []
        for _ in range(8):
            eng.witness_blocks(blocks, pre, out=out)
            ms.append(eng.last_kernel_ms())
        m = float(np.median(ms))
        r = dict(n=n, parts=parts, ms=m, blocks_per_s=n / m * 1e3, GBps=alg * n / m / 1e6)
        res.append(r)
        print(json.dumps(r), flush=True)
    del out
    torch.cuda.empty_cache()
