#!/usr/bin/env python3
"""Completeness sweep of the on-device verifier (hsw_verify_blocks): corrupt EVERY cell of a block, one at a
time -- all gate cells, both chip columns, the lookup column, the next state, every input byte and pre-state
word -- and require a violation each time.  A cell whose corruption passes would be a witness the constraint
system (as recorded from the reference's source) leaves free, or a gap in the verifier.
Test infrastructure.  usage: python tests/flip_sweep.py [bits] [ncols] [montgomery 0/1]"""
import importlib
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def sweep(bits=8, ncols=2, mont=False, internals=True, limb=0, stride=1, seed=5, log=None):
    import torch
    hsw = importlib.import_module("halo2-dynamic-sha256_amd")
    N = hsw._native
    eng = hsw.WitnessEngine(0, bits, ncols, mode=N.HSW_MODE_HALO2_INTERNALS if internals else N.HSW_MODE_DEFAULT)
    rng = np.random.default_rng(seed)
    n, blk = 2, 1                                        # corrupt block 1 of 2 (its pre-state has a neighbour)
    blocks = rng.integers(0, 256, (n, 64), dtype=np.uint8)
    pre = rng.integers(0, 2**32, (n, 8), dtype=np.uint64).astype(np.uint32)
    tb, tp = torch.from_numpy(blocks).cuda(), torch.from_numpy(pre.view(np.int32)).cuda()
    flags = N.HSW_REPR_MONTGOMERY if mont else 0
    out = eng.witness_blocks_ex(tb, tp, cursor0=0, flags=flags, want_lookup=internals)
    eng.synchronize()
    lk = out["lookup"] if internals else None

    def verify(b=tb, p=tp):
        return eng.verify_blocks(b, p, out, cursor0=0, lookup=lk, flags=flags)

    rep = verify()
    assert rep["violations"] == 0, rep
    missed = {}
    t0 = time.time()
    G, LC, LK = eng.G, eng.limb_calls, eng.lookup_cells

    def run(name, tensor, index_of, count):
        bad = []
        for k in range(0, count, stride):
            idx = index_of(k)
            saved = tensor[idx].clone()
            tensor[idx][limb] ^= 1 if not mont else 0x10        # Montgomery: any change of the limb moves the value
            if verify()["violations"] == 0:
                bad.append(k)
            tensor[idx] = saved
        missed[name] = bad
        if log:
            print("%-12s %6d cells, %d undetected  (%.0f s)" % (name, (count + stride - 1) // stride, len(bad), time.time() - t0),
                  file=log, flush=True)

    run("gate", out["gate"], lambda k: blk * G + k, G)
    run("chip dense", out["dense"], lambda k: ((blk * LC + k) % ncols, (blk * LC + k) // ncols), LC)
    run("chip spread", out["spread"], lambda k: ((blk * LC + k) % ncols, (blk * LC + k) // ncols), LC)
    if internals:
        run("lookup", lk, lambda k: blk * LK + k, LK)
    # next state, inputs, pre-state: one word / byte at a time
    bad = []
    for w in range(8):
        out["next_states"][blk, w] ^= 1
        if verify()["violations"] == 0:
            bad.append(w)
        out["next_states"][blk, w] ^= 1
    missed["next state"] = bad
    bad = []
    for i in range(64):
        b2 = tb.clone(); b2[blk, i] ^= 1
        if verify(b=b2)["violations"] == 0:
            bad.append(i)
    missed["input byte"] = bad
    bad = []
    for w in range(8):
        p2 = tp.clone(); p2[blk, w] ^= 1
        if verify(p=p2)["violations"] == 0:
            bad.append(w)
    missed["pre-state"] = bad
    assert verify()["violations"] == 0
    eng.close()
    return missed


class _DevCells:
    """(n, 4) int64 view of device memory owned by libhsw, for torch.as_tensor."""
    def __init__(self, ptr, n):
        self.__cuda_array_interface__ = {"shape": (n, 4), "typestr": "<i8", "data": (int(ptr), False), "version": 2}


def sweep_frames(sizes=(128, 64), rc=True, mont=False, columns=None, limb=0, log=None):
    """The same for the digest frames (hsw_gadget_verify): every prologue / epilogue cell and the zero cell of a
    whole-digest context, every lookup-column entry of the frames.  Expected free: the one cell the reference
    itself leaves unconstrained (is_zero's inverse witness when its input IS zero -- any value satisfies the row)."""
    import torch
    hsw = importlib.import_module("halo2-dynamic-sha256_amd")
    N = hsw._native
    eng = hsw.WitnessEngine(0, 8, 2, mode=N.HSW_MODE_HALO2_INTERNALS)
    cfg = hsw.Sha256DynamicConfig(eng, list(sizes), is_input_range_check=rc, whole_digest=True)
    if mont:
        cfg.set_repr(N.HSW_REPR_MONTGOMERY)
    if columns:
        cfg.set_columns(columns)
    msgs = [bytes(range(60)), b"abc"][: len(sizes)]
    res = cfg.digest_batch(msgs, [0] * len(msgs))
    assert cfg.verify()["violations"] == 0
    v = cfg.view()
    total = int(v.max_rows * v.columns) if columns else int(v.gate_cells)
    gate = torch.as_tensor(_DevCells(v.d_gate, total), device="cuda")
    lookup = torch.as_tensor(_DevCells(v.d_lookup, int(v.lookup_cells)), device="cuda")
    missed = {"frame gate": [], "frame lookup": []}
    t0 = time.time()
    tested = 0
    for d, r in enumerate(res):
        spans = [(r.prologue_cell, r.block_cell), (r.epilogue_cell, r.end_cell)]
        for a, b in spans:
            for cell in range(a, b):
                col, row = cfg.cell_position(cell) if columns else (0, cell)
                at = col * columns + row if columns else cell
                saved = gate[at].clone()
                gate[at][limb] ^= 1 if not mont else 0x10
                if cfg.verify()["violations"] == 0:
                    missed["frame gate"].append((d, cell - (a if a == r.prologue_cell else r.epilogue_cell), "prologue" if a == r.prologue_cell else "epilogue"))
                gate[at] = saved
                tested += 1
        for a, b in [(r.prologue_lookup, r.block_lookup), (r.epilogue_lookup, r.epilogue_lookup + 64)]:
            for k in range(a, b):
                saved = lookup[k].clone()
                lookup[k][limb] ^= 1 if not mont else 0x10
                if cfg.verify()["violations"] == 0:
                    missed["frame lookup"].append((d, k - a))
                lookup[k] = saved
                tested += 1
    assert cfg.verify()["violations"] == 0
    if log:
        print("frames       %6d cells, %d undetected  (%.0f s)" % (tested, sum(len(x) for x in missed.values()), time.time() - t0),
              file=log, flush=True)
    cfg.close()
    eng.close()
    return missed


if __name__ == "__main__":
    bits = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    ncols = int(sys.argv[2]) if len(sys.argv) > 2 else 2
    mont = bool(int(sys.argv[3])) if len(sys.argv) > 3 else False
    limb = int(sys.argv[4]) if len(sys.argv) > 4 else 0
    internals = bool(int(sys.argv[5])) if len(sys.argv) > 5 else True
    m = sweep(bits, ncols, mont, internals=internals, limb=limb, log=sys.stdout)
    if bits == 8 and ncols == 2 and internals:
        m.update(sweep_frames(mont=mont, limb=limb, log=sys.stdout))
        m.update({k + " (columns)": v for k, v in sweep_frames(mont=mont, limb=limb, columns=100003, log=sys.stdout).items()})
    print(json.dumps({"bits": bits, "ncols": ncols, "montgomery": mont, "limb": limb, "internals": internals,
                      "undetected": {k: v[:50] for k, v in m.items()}, "undetected_total": sum(len(v) for v in m.values())}))
