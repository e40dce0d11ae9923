#!/usr/bin/env python3
"""Generates tests/golden/stream_fingerprints.json from the CPU oracle.

The reference holds no cell-level golden vectors, and cannot be
run here, so these fingerprints are NOT reference outputs: they are regression
anchors of the oracle's streams (whose value-exactness is argued in
oracle/hsw_oracle.h), so that (a) the oracle cannot drift silently and (b) the
GPU box can check the HIP path against committed data.

Run from the repo root:  python tests/golden/make_golden.py
"""
import hashlib
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import oracle as O  # noqa: E402


def golden_inputs(case):
    """Deterministic inputs, reproducible without the oracle."""
    if case == "iv_abc":          # 'abc' padded, pre-state IV (lib.rs:501)
        blk = np.zeros((1, 64), dtype=np.uint8)
        blk[0, :3] = [0x61, 0x62, 0x63]
        blk[0, 3] = 0x80
        blk[0, 63] = 24
        return blk, O.INIT_STATE.reshape(1, 8).copy()
    if case == "zeros":           # the zero block the reference compresses after the padding (lib.rs:111-113)
        return np.zeros((1, 64), dtype=np.uint8), O.INIT_STATE.reshape(1, 8).copy()
    if case == "ones":
        return np.full((1, 64), 0xFF, dtype=np.uint8), np.full((1, 8), 0xFFFFFFFF, dtype=np.uint32)
    if case == "lcg8":            # 8 blocks from a fixed LCG
        x = np.uint64(0x9E3779B97F4A7C15)
        vals = []
        for _ in range(8 * 64 + 8 * 8 * 4):
            x = np.uint64((int(x) * 6364136223846793005 + 1442695040888963407) % 2**64)
            vals.append(int(x) >> 56)
        b = np.array(vals[:512], dtype=np.uint8).reshape(8, 64)
        p = np.array(vals[512:], dtype=np.uint8).reshape(8, 8, 4)
        pre = (p[..., 0].astype(np.uint32) << 24) | (p[..., 1].astype(np.uint32) << 16) | \
              (p[..., 2].astype(np.uint32) << 8) | p[..., 3].astype(np.uint32)
        return b, pre
    raise KeyError(case)


CASES = [("iv_abc", 8, 2, 0), ("zeros", 8, 2, 0), ("ones", 8, 2, 0), ("lcg8", 8, 2, 0),
         ("lcg8", 8, 2, 12345), ("lcg8", 16, 1, 0), ("lcg8", 4, 3, 5)]


def fingerprint(w):
    return {
        "gate_sha256": hashlib.sha256(w["gate"].tobytes()).hexdigest(),
        "dense_sha256": hashlib.sha256(w["dense"].tobytes()).hexdigest(),
        "spread_sha256": hashlib.sha256(w["spread"].tobytes()).hexdigest(),
        "next_states_sha256": hashlib.sha256(w["next_states"].tobytes()).hexdigest(),
        "gate_cells": int(w["gate"].shape[0]),
        "rows": int(w["rows"]),
        "gate_first8_lo": [int(x) for x in w["gate"][:8, 0]],
        "gate_last8_lo": [int(x) for x in w["gate"][-8:, 0]],
    }


def main():
    out = {"_comment": "oracle-generated regression fingerprints (not reference outputs); see make_golden.py",
           "cases": []}
    for name, bits, ncols, cursor0 in CASES:
        blocks, pre = golden_inputs(name)
        w = O.Oracle(bits, ncols, check=True).witness_blocks(blocks, pre, cursor0=cursor0)
        out["cases"].append(dict(name=name, num_bits_lookup=bits, num_advice_columns=ncols,
                                 cursor0=cursor0, **fingerprint(w)))
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "stream_fingerprints.json")
    json.dump(out, open(path, "w"), indent=1)
    print("wrote", path)


if __name__ == "__main__":
    main()
