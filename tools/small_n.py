"""Kernel time of the small-batch kernel ("split" = 2, by "helpers") against the streaming kernel ("split" = 0,
its own choice of waves per block) by batch size -- where the engine's automatic switch (<= 32 blocks) should sit.
Tuning aid; usage: python tools/small_n.py [montgomery]"""
import importlib, sys, numpy as np, torch
sys.path.insert(0, '/root/repo')
hsw = importlib.import_module("halo2-dynamic-sha256_amd")
flags = hsw._native.HSW_REPR_MONTGOMERY if len(sys.argv) > 1 else 0
eng = hsw.WitnessEngine(0, 8, 2)
eng.set_timing(True)
rng = np.random.default_rng(1)
for n in [8, 16, 32, 48, 64, 96, 128, 192, 256, 512]:
    blocks = torch.from_numpy(rng.integers(0, 256, (n, 64), dtype=np.uint8)).cuda()
    pre = torch.zeros((n, 8), dtype=torch.int32, device="cuda")
    out = eng.alloc_outputs(n, 0, 0)
    res = {}
    for name, split, helpers in [("stream", 0, 0), ("small/1", 2, 1), ("small/2", 2, 2), ("small/4", 2, 4)]:
        eng.set_option("split", split)
        eng.set_option("helpers", helpers)
        ms = []
        for i in range(10):
            eng.witness_blocks(blocks, pre, out=out, flags=flags)
            if i >= 3:
                ms.append(eng.last_kernel_ms())
        res[name] = round(float(np.median(ms)) * 1e3, 1)
    print(n, res, flush=True)
