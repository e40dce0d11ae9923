import importlib, sys, numpy as np, torch
sys.path.insert(0, '/root/repo')
hsw = importlib.import_module("halo2-dynamic-sha256_amd")
eng = hsw.WitnessEngine(0, 8, 2)
rng = np.random.default_rng(1)
n = 4096
tb = torch.from_numpy(rng.integers(0, 256, (n, 64), dtype=np.uint8)).cuda()
tp = torch.from_numpy(rng.integers(0, 2**31, (n, 8), dtype=np.int64).astype(np.int32)).cuda()
out = eng.witness_blocks(tb, tp)
for s in [1, 2, 4, 8, 16, 32, 64, 128]:
    eng.set_option("verify_slices", s)
    ms = [eng.verify_blocks(tb, tp, out)["kernel_ms"] for _ in range(4)]
    print(s, round(float(np.median(ms[1:])), 3), flush=True)
