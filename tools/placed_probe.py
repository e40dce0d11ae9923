import importlib, sys, os, numpy as np, torch
sys.path.insert(0, os.getcwd())
hsw = importlib.import_module("halo2-dynamic-sha256_amd")
eng = hsw.WitnessEngine(0, 8, 2)
n = 4096
rng = np.random.default_rng(1)
blocks = torch.from_numpy(rng.integers(0, 256, (n, 64), dtype=np.uint8)).cuda()
pre = torch.from_numpy(rng.integers(0, 2**31, (n, 8), dtype=np.int64).astype(np.int32)).cuda()
out, rep = eng.alloc_outputs_placed(blocks, pre, candidates=int(sys.argv[1]) if len(sys.argv) > 1 else 8, spacer_bytes=int(float(sys.argv[2]) * 2**30) if len(sys.argv) > 2 else 6 << 30,
                                    ranged=(sys.argv[3] == "1") if len(sys.argv) > 3 else True)
for kind, row in zip(rep["gate_candidates"], rep["kernel_ms"]):
    print("%-24s" % kind, " ".join("%.3f" % x for x in row))
print("kept", rep["kept"], "=", rep["kernel_ms"][rep["kept"][0]][rep["kept"][1]])
