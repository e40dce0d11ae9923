#!/usr/bin/env python3
"""Latency of the digest-frame path (hsw_witness_frames / whole-digest gadget)."""
import ctypes as C, importlib, os, sys, time, json
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
hsw = importlib.import_module("halo2-dynamic-sha256_amd")
N = hsw._native
eng = hsw.WitnessEngine(0, 8, 2, mode=N.HSW_MODE_HALO2_INTERNALS)
s = eng.shape


def frames_only(n_digests, nb, rc):
    fs = N.frame_query(s, 64 * nb, rc)
    G, LK = eng.G, eng.lookup_cells
    descs = (N.FrameDesc * n_digests)()
    gc = lc = 0
    for i in range(n_digests):
        d = descs[i]
        d.input_len, d.first_block, d.n_blocks, d.num_round, d.precomputed_round = 3, i * nb, nb, 1, 0
        d.is_input_range_check = 1 if rc else 0
        d.prologue_cell, d.prologue_lookup = gc, lc
        gc += fs.prologue_cells; lc += fs.prologue_lookups
        d.zero_cell = N.NO_CELL
        gc += nb * G; lc += nb * LK
        d.epilogue_cell, d.epilogue_lookup = gc, lc
        gc += fs.epilogue_cells; lc += fs.epilogue_lookups
    blocks = torch.zeros(n_digests * nb * 64, dtype=torch.uint8, device="cuda")
    pre = torch.zeros(n_digests * nb * 8, dtype=torch.int32, device="cuda")
    nxt = torch.ones(n_digests * nb * 8, dtype=torch.int32, device="cuda")
    gate = torch.zeros(gc * 4, dtype=torch.int64, device="cuda")
    look = torch.zeros(lc * 4, dtype=torch.int64, device="cuda")
    def call():
        rc_ = eng.lib.hsw_witness_frames(eng.h, descs, n_digests, blocks.data_ptr(), pre.data_ptr(), nxt.data_ptr(),
                                         gate.data_ptr(), look.data_ptr(), None, 0)
        assert rc_ == 0
    for _ in range(5):
        call()
    torch.cuda.synchronize()
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(20)]
    t0 = time.perf_counter()
    for a, b in evs:
        a.record(); call(); b.record()
    torch.cuda.synchronize()
    wall = (time.perf_counter() - t0) / len(evs)
    dev = float(np.median([a.elapsed_time(b) for a, b in evs]))
    return {"digests": n_digests, "blocks_each": nb, "rc": rc, "device_ms": round(dev, 4), "wall_ms_per_call": round(wall * 1e3, 4)}


for cfg in [(1, 16, True), (1, 1, False), (64, 1, False), (4096, 1, False), (4096, 1, True)]:
    print(json.dumps(frames_only(*cfg)), flush=True)

# gadget, whole region of the bench circuit
import hashlib
cfgw = hsw.Sha256DynamicConfig(eng, [1024], True, whole_digest=True)
cfgw.set_columns((1 << 17) - 9)
m = bytes([1] * 56)
for _ in range(5):
    cfgw.reset(); cfgw.digest(m)
t0 = time.perf_counter()
for _ in range(100):
    cfgw.reset(); r = cfgw.digest(m)
print(json.dumps({"bench_circuit_whole_region_ms": round((time.perf_counter() - t0) / 100 * 1e3, 4)}))
assert r.output_bytes == hashlib.sha256(m).digest()
