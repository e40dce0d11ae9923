import importlib, sys, os, numpy as np, torch
sys.path.insert(0, os.getcwd())
hsw = importlib.import_module("halo2-dynamic-sha256_amd")
eng = hsw.WitnessEngine(0, 8, 2)
n = 4096
rng = np.random.default_rng(1)
blocks = torch.from_numpy(rng.integers(0, 256, (n, 64), dtype=np.uint8)).cuda()
pre = torch.from_numpy(rng.integers(0, 2**31, (n, 8), dtype=np.int64).astype(np.int32)).cuda()
out, rep = eng.alloc_outputs_placed(blocks, pre, candidates=int(sys.argv[1]) if len(sys.argv) > 1 else 8, spacer_bytes=int(float(sys.argv[2]) * 2**30) if len(sys.argv) > 2 else 6 << 30)
print(" ".join("%.3f" % t for t in rep["kernel_ms_each"]), "kept", rep["kept"])
