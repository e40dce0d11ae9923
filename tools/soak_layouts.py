#!/usr/bin/env python3
"""Random whole-digest LAYOUTS, each verified on the device: digests of random sizes, input range checks on /
off, canonical / Montgomery cells, FlexGate columns of a random height (so the column breaks fall anywhere in
the frames and block streams), streaming kernel ("split" = 0) or the engine's own choice.  The verifier
(hsw_gadget_verify) checks every gate row, copy, range bound and chip tie at the place the constraint structure
expects it, independently of the write-out logic -- a misplaced flush shows up as violations.
usage: soak_layouts.py [seconds] [seed] [num_advice_columns of the spread chip, default 2]"""
import importlib, json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
hsw = importlib.import_module("halo2-dynamic-sha256_amd")
N = hsw._native
secs = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 11)
ncols = int(sys.argv[3]) if len(sys.argv) > 3 else 2
eng = hsw.WitnessEngine(0, 8, ncols, mode=N.HSW_MODE_HALO2_INTERNALS)
G = eng.G
t0 = time.time()
layouts = checks = skipped = 0
last = t0
while time.time() - t0 < secs:
    nd = int(rng.integers(1, 25))
    same = rng.random() < 0.5
    sizes = [64 * int(rng.integers(1, 9))] * nd if same else [64 * int(rng.integers(1, 9)) for _ in range(nd)]
    rc = bool(rng.integers(0, 2))
    mont = bool(rng.integers(0, 2))
    split = int(rng.choice([0, 0, -1]))
    max_rows = int(rng.integers(G + 16, 6 * G))
    desc = dict(sizes=sizes, rc=rc, mont=mont, split=split, max_rows=max_rows)
    eng.set_option("split", split)
    cfg = hsw.Sha256DynamicConfig(eng, sizes, is_input_range_check=rc, whole_digest=True)
    try:
        if mont:
            cfg.set_repr(N.HSW_REPR_MONTGOMERY)
        try:
            cfg.set_columns(max_rows)
        except hsw.HswError:
            skipped += 1
            continue
        msgs = [rng.integers(0, 256, int(rng.integers(0, s - 8)), dtype=np.uint8).tobytes() for s in sizes]
        cfg.digest_batch(msgs)
        rep = cfg.verify()
        assert rep["violations"] == 0, (desc, rep)
        checks += rep["checks"]
        layouts += 1
    finally:
        cfg.close()
        eng.set_option("split", -1)
    if time.time() - last > 20:
        last = time.time()
        print(json.dumps({"layouts": layouts, "checks": checks, "skipped": skipped, "seconds": round(last - t0, 1)}), flush=True)
print(json.dumps({"layouts": layouts, "checks": checks, "skipped": skipped, "violations": 0, "seconds": round(time.time() - t0, 1)}))
