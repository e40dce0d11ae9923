#!/usr/bin/env python3
"""Soak: back-to-back launches for ~60 s, all three representations, checksums must never change."""
import importlib, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
hsw = importlib.import_module("halo2-dynamic-sha256_amd")
eng = hsw.WitnessEngine(0, 8, 2)
n = 2048
rng = np.random.default_rng(7)
blocks = torch.from_numpy(rng.integers(0, 256, (n, 64), dtype=np.uint8)).cuda()
pre = torch.from_numpy(rng.integers(0, 2**31, (n, 8), dtype=np.int64).astype(np.int32)).cuda()
outs = {f: eng.alloc_outputs(n, 0, f) for f in (0, hsw.HSW_REPR_MONTGOMERY, hsw.HSW_REPR_COMPACT64)}
ref = {}
t0 = time.time(); launches = 0
secs = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
while time.time() - t0 < secs:
    for f, out in outs.items():
        for _ in range(20):
            eng.witness_blocks(blocks, pre, out=out, flags=f)
            launches += 1
        sig = (int(out["gate"].sum()), int(out["dense"].sum()), int(out["spread"].sum()), int(out["next_states"].sum()))
        if f not in ref:
            ref[f] = sig
        assert sig == ref[f], ("checksum changed", f, launches)
print("soak ok: %d launches in %.1f s, checksums stable" % (launches, time.time() - t0))
