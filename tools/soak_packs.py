#!/usr/bin/env python3
"""Random FlexGate column packings of raw block launches (hsw_witness_blocks_ex with a pack plan, internals
mode), each verified on the device with hsw_verify_blocks under the same plan: tile shape, waves per block,
kernel choice, helper waves, cell form, batch size, start row and column height are drawn at random; half of
the layouts aim a column break at the first or last 160 cells of a block (the realigning write-out's corner
cases).  No oracle on the CPU, so ~1 ms per case.  usage: soak_packs.py [seconds] [seed] [num_advice_columns] [num_bits_lookup]"""
import ctypes as C, importlib, json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
hsw = importlib.import_module("halo2-dynamic-sha256_amd")
N = hsw._native
import torch
secs = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 3)
ncols = int(sys.argv[3]) if len(sys.argv) > 3 else 2
bits = int(sys.argv[4]) if len(sys.argv) > 4 else 8
eng = hsw.WitnessEngine(0, bits, ncols, mode=N.HSW_MODE_HALO2_INTERNALS)
G, LK = eng.G, eng.lookup_cells
KNOBS = [(0, 0), (32, 1), (32, 4), (32, 32), (64, 2), (64, 4), (64, 16), (128, 4), (128, 8), (128, 32), (0, 1), (0, 8)]
t0 = time.time()
last = t0
cases = checks = skipped = aimed = 0
while time.time() - t0 < secs:
    n = int(rng.choice([1, 2, 3, 5, 8, 13, 17, 31, 33, 40, 65, 128, 129] if bits >= 8 else [1, 2, 3, 5, 8]))
    mont = bool(rng.integers(0, 2))
    tile, parts = KNOBS[int(rng.integers(0, len(KNOBS)))]
    split = int(rng.choice([-1, 0, 0, 2]))
    helpers = int(rng.integers(0, 5))
    aim = rng.random()
    plan = None
    for _ in range(400 if aim < 0.5 else 1):
        max_rows = int(rng.integers(G // 2 + 16, 3 * G))
        start_row = int(rng.integers(0, max_rows))
        try:
            pl = N.pack_plan(eng.shape, n, start_row, max_rows)
        except N.HswError:
            continue
        at = [int(pl.break_cell[k]) % G for k in range(pl.n_breaks)]
        if aim >= 0.5 or any((0 < c <= 160) if aim < 0.25 else (c >= G - 160) for c in at):
            plan = pl
            aimed += aim < 0.5
            break
    if plan is None:
        skipped += 1
        continue
    desc = dict(n=n, mont=mont, tile=tile, parts=parts, split=split, helpers=helpers, start_row=start_row, max_rows=max_rows)
    blocks = torch.from_numpy(rng.integers(0, 256, (n, 64), dtype=np.uint8)).cuda()
    pre = torch.from_numpy(rng.integers(0, 2**32, (n, 8), dtype=np.uint64).astype(np.uint32).view(np.int32)).cuda()
    rows = eng.chip_rows(0, n)
    gate = torch.full((int(plan.span_cells), 4), -1, dtype=torch.int64, device="cuda")
    dense = torch.zeros((ncols, rows, 4), dtype=torch.int64, device="cuda")
    spread = torch.zeros((ncols, rows, 4), dtype=torch.int64, device="cuda")
    lookup = torch.empty((n * LK, 4), dtype=torch.int64, device="cuda")
    nxt = torch.empty((n, 8), dtype=torch.int32, device="cuda")
    a = N.WitnessArgs()
    a.d_blocks, a.d_pre_states, a.n_blocks, a.spread_cursor0 = blocks.data_ptr(), pre.data_ptr(), n, 0
    a.d_gate, a.d_chip_dense, a.d_chip_spread, a.chip_col_stride = gate.data_ptr(), dense.data_ptr(), spread.data_ptr(), rows
    a.d_next_states, a.d_lookup, a.pack = nxt.data_ptr(), lookup.data_ptr(), C.pointer(plan)
    a.flags = N.HSW_REPR_MONTGOMERY if mont else 0
    for k, v in (("tile", tile), ("parts", parts), ("split", split), ("helpers", helpers)):
        eng.set_option(k, v)
    try:
        rc = eng.lib.hsw_witness_blocks_ex(eng.h, C.byref(a))
        if rc == N.HSW_ERR_UNSUPPORTED:          # e.g. more than two column breaks inside one block
            skipped += 1
            continue
        assert rc == 0, (desc, eng.lib.hsw_last_error(eng.h))
        rep = N.VerifyReport()
        rc = eng.lib.hsw_verify_blocks(eng.h, C.byref(a), C.byref(rep))
        assert rc == 0, (desc, eng.lib.hsw_last_error(eng.h))
        assert rep.violations == 0, (desc, int(rep.violations), int(rep.first_block), int(rep.first_cell), int(rep.first_class))
        checks += int(rep.checks)
        cases += 1
    finally:
        for k, v in (("tile", 0), ("parts", 0), ("split", -1), ("helpers", 0)):
            eng.set_option(k, v)
    if time.time() - last > 20:
        last = time.time()
        print(json.dumps({"cases": cases, "aimed": int(aimed), "checks": checks, "skipped": skipped, "seconds": round(last - t0, 1)}), flush=True)
print(json.dumps({"cases": cases, "aimed": int(aimed), "checks": checks, "skipped": skipped, "violations": 0, "seconds": round(time.time() - t0, 1)}))
