import importlib, os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
hsw = importlib.import_module("halo2-dynamic-sha256_amd")
eng = hsw.WitnessEngine(0, 8, 2)
rng = np.random.default_rng(1)
n = 4096
tb = torch.from_numpy(rng.integers(0, 256, (n, 64), dtype=np.uint8)).cuda()
tp = torch.from_numpy(rng.integers(0, 2**31, (n, 8), dtype=np.int64).astype(np.int32)).cuda()
out = eng.witness_blocks(tb, tp)
for name, kw in (("full", {}), ("no chip", dict(check_chip=False)), ("no chip no next", dict(check_chip=False, check_next=False))):
    reps = [eng.verify_blocks(tb, tp, out, **kw) for _ in range(5)]
    print(name, round(float(np.median([r["kernel_ms"] for r in reps[1:]])), 3), reps[0]["checks"])
