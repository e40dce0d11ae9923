#!/usr/bin/env python3
"""Soak of the realigned write-out and the frame path: whole-digest batches (every block stream starts off a
128-byte line), canonical and Montgomery, linear and as column images, each verified on the device.  Any
violation would point at a race in the tile carry / head hold-back logic.  usage: soak_region.py [seconds] [small]"""
import importlib, os, sys, time, json
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
hsw = importlib.import_module("halo2-dynamic-sha256_amd")
N = hsw._native
secs = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
only_small = len(sys.argv) > 2 and sys.argv[2] == "small"      # the small-batch launches only
eng = hsw.WitnessEngine(0, 8, 2, mode=N.HSW_MODE_HALO2_INTERNALS)
rng = np.random.default_rng(5)
configs = []
# the last four are small-batch launches (the very last of mixed sizes: one fused launch per run of equal sizes) (<= 128 blocks): workgroups of up to 4 waves sharing one tile, frames in the same grid
for nd, sizes in [(512, [64] * 512), (96, [192] * 96), (40, [64, 128, 256, 64] * 10),
                  (1, [1024]), (8, [960] * 8), (2, [128, 128]), (6, [64, 128, 256, 64, 64, 192])]:
    if only_small and nd > 8:
        continue
    for rep in (0, N.HSW_REPR_MONTGOMERY):
        for max_rows in (None, 1_000_003 if nd > 8 else 131_063):
            cfg = hsw.Sha256DynamicConfig(eng, sizes, is_input_range_check=bool(rep), whole_digest=True)
            if rep:
                cfg.set_repr(rep)
            if max_rows:
                try:
                    cfg.set_columns(max_rows)
                except hsw.HswError:
                    cfg.close()
                    continue
            configs.append((cfg, sizes))
t0 = time.time()
iters = launches = checks = 0
while time.time() - t0 < secs:
    for cfg, sizes in configs:
        msgs = [rng.integers(0, 256, int(rng.integers(0, s - 8)), dtype=np.uint8).tobytes() for s in sizes]
        cfg.reset()
        cfg.digest_batch(msgs)
        rep = cfg.verify()
        assert rep["violations"] == 0, (iters, rep)
        checks += rep["checks"]
        launches += 1
    iters += 1
    if iters % 5 == 0:
        print(json.dumps({"iterations": iters, "batches": launches, "checks": checks, "seconds": round(time.time() - t0, 1)}), flush=True)
print(json.dumps({"iterations": iters, "batches": launches, "checks": checks, "violations": 0}))
