import importlib, time, numpy as np, torch, hashlib, sys
sys.path.insert(0, ".")
hsw = importlib.import_module("halo2-dynamic-sha256_amd")
from oracle import oracle as O
eng = hsw.WitnessEngine(0, 8, 2)
for nm, bpm in ((1, 16), (256, 16), (2048, 2), (4096, 1), (3000, 3)):
    rng = np.random.default_rng(nm)
    blocks = rng.integers(0, 256, (nm * bpm, 64), dtype=np.uint8)
    tb = torch.from_numpy(blocks).cuda()
    pre = eng.sha256_chain(tb, nm, bpm)
    torch.cuda.synchronize()
    # check against plain compress
    p = pre.cpu().numpy().view(np.uint32).reshape(nm, bpm, 8)
    for m in (0, nm - 1):
        st = O.INIT_STATE.copy()
        for j in range(bpm):
            assert np.array_equal(p[m, j], st), (nm, bpm, m, j)
            st = O.plain_compress(st, blocks[m * bpm + j])
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(20)]
    for a, b in ev:
        a.record(); eng.sha256_chain(tb, nm, bpm); b.record()
    torch.cuda.synchronize()
    print(nm, bpm, "chain kernel us:", round(float(np.median([a.elapsed_time(b) for a, b in ev])) * 1e3, 1))
