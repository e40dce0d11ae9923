#!/usr/bin/env python3
"""hsw_verify_blocks time for 4,096 blocks vs workgroups per block ("verify_slices"), canonical and Montgomery."""
import importlib, os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
hsw = importlib.import_module("halo2-dynamic-sha256_amd")
eng = hsw.WitnessEngine(0, 8, 2)
rng = np.random.default_rng(1)
n = 4096
tb = torch.from_numpy(rng.integers(0, 256, (n, 64), dtype=np.uint8)).cuda()
tp = torch.from_numpy(rng.integers(0, 2**31, (n, 8), dtype=np.int64).astype(np.int32)).cuda()
for name, flags in (("canonical", 0), ("montgomery", hsw.HSW_REPR_MONTGOMERY)):
    out = eng.witness_blocks(tb, tp, flags=flags)
    for s in [1, 2, 4, 8, 16]:
        eng.set_option("verify_slices", s)
        reps = [eng.verify_blocks(tb, tp, out, flags=flags) for _ in range(5)]
        assert all(r["violations"] == 0 for r in reps)
        print(name, "slices", s, "ms", round(float(np.median([r["kernel_ms"] for r in reps[1:]])), 3), flush=True)
