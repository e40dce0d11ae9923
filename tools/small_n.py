import importlib, sys, time, numpy as np, torch
sys.path.insert(0, '/root/repo')
hsw = importlib.import_module("halo2-dynamic-sha256_amd")
eng = hsw.WitnessEngine(0, 8, 2)
eng.set_timing(True)
rng = np.random.default_rng(1)
for n in [1, 2, 4, 8, 16, 32, 64, 128, 256, 512]:
    blocks = torch.from_numpy(rng.integers(0, 256, (n, 64), dtype=np.uint8)).cuda()
    pre = torch.zeros((n, 8), dtype=torch.int32, device="cuda")
    out = eng.alloc_outputs(n, 0, 0)
    res = {}
    for parts in [0, 4, 8, 16, 32]:
        eng.set_option("parts", parts)
        if parts == 32: eng.set_option("tile", 32)
        ms = []
        for i in range(8):
            eng.witness_blocks(blocks, pre, out=out)
            if i >= 3: ms.append(eng.last_kernel_ms())
        res[parts] = round(float(np.median(ms)) * 1e3, 1)
        eng.set_option("tile", 0)
    eng.set_option("parts", 0)
    print(n, res, flush=True)
