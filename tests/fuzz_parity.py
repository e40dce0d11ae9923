#!/usr/bin/env python3
"""Randomised differential run: HIP path (through the C ABI) vs the CPU oracle, bit for bit.

Every iteration draws a point of the cross product the parametrised tests only sample:
  block streams   mode (default / halo2 internals) x table width x advice columns x batch size (both
                  sides of the 32-block split-phase threshold) x chip cursor x cell representation x
                  tile / waves-per-block / split knobs x gate stream 0..3 cells off a 128-byte line x
                  FlexGate column packing with a random start row / column height
  gadget streams  Sha256DynamicConfig.digest / digest_batch in random groupings (both chaining sides, zero-copy
                  or staged inputs, reset), canonical / Montgomery
  chain pre-pass  hsw_sha256_chain around the 64-lane / 256-thread boundaries, FIPS IV or given prefix states
  host delivery   hsw_witness_blocks_host around its 128-block chunking, both cursor alignments, pinned or not
  whole digests   random message lists (lengths, maximum sizes, precomputed prefixes, input range checks,
                  one batch or one call per digest, canonical / Montgomery, linear stream or column image)
and compares every output cell with the oracle, plus the bytes around the outputs (must stay untouched).

Test infrastructure (imports oracle/).  usage: python tests/fuzz_parity.py [seconds] [seed]
tests/test_gpu_fuzz.py runs a short, fixed-seed slice of it in the GPU suite."""
import ctypes as C
import hashlib
import importlib
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

FILL = np.uint64(0xFFFFFFFFFFFFFFFF)
P0 = 0x43e1f593f0000001


class Fuzzer:
    def __init__(self, seed=1):
        import torch
        self.torch = torch
        self.hsw = importlib.import_module("halo2-dynamic-sha256_amd")
        self.N = self.hsw._native
        self.O = importlib.import_module("oracle.oracle")
        self.rng = np.random.default_rng(seed)
        self.engines = {}
        self.stats = dict(block_runs=0, block_cells=0, digest_runs=0, digest_cells=0, skipped=0)

    def engine(self, bits, ncols, internals):
        key = (bits, ncols, internals)
        if key not in self.engines:
            mode = self.N.HSW_MODE_HALO2_INTERNALS if internals else self.N.HSW_MODE_DEFAULT
            self.engines[key] = self.hsw.WitnessEngine(0, bits, ncols, mode=mode)
        return self.engines[key]

    def close(self):
        for e in self.engines.values():
            e.close()
        self.engines = {}

    # ---------------------------------------------------------------- block streams
    def _expected(self, ref, eng, n, flags, internals):
        """Oracle streams in the representation `flags` asks for: (gate, dense, spread, lookup)."""
        N, O = self.N, self.O
        lookup = ref.get("lookup") if internals else None
        if flags == N.HSW_REPR_MONTGOMERY:
            return (O.to_montgomery(ref["gate"]), O.to_montgomery(ref["dense"]), O.to_montgomery(ref["spread"]),
                    O.to_montgomery(lookup) if lookup is not None else None)
        if flags == N.HSW_REPR_COMPACT64:
            G = eng.G
            g = ref["gate"][:, 0].copy().reshape(n, G)
            neg = N.neg_cells(eng.shape).astype(np.int64)
            wide = (ref["gate"][:, 1:] != 0).any(axis=1).reshape(n, G)
            mask = np.zeros(G, dtype=bool)
            mask[neg] = True
            assert not wide[:, ~mask].any()
            vals = g[:, neg]
            g[:, neg] = np.where(wide[:, neg], (np.uint64(P0) - vals).astype(np.uint64), vals)
            return (g.reshape(-1, 1), ref["dense"][..., :1], ref["spread"][..., :1],
                    lookup[:, :1] if lookup is not None else None)
        return ref["gate"], ref["dense"], ref["spread"], lookup

    def block_case(self, override=None):
        """override: a FAILED-case dict to replay (its layout parameters; the data is drawn afresh)."""
        rng, N, t = self.rng, self.N, self.torch
        internals = bool(rng.integers(0, 2))
        bits = int(rng.choice([16, 8, 8, 4, 2, 1]))
        ncols = int(rng.integers(1, 7))
        n = int(rng.choice([1, 2, 3, 5, 8, 13, 17, 31, 32, 33, 40, 65, 128, 129]))
        if bits <= 2:
            n = min(n, 8)
        cursor0 = int(rng.choice([0, 1, int(rng.integers(0, 10**6)), int(rng.integers(0, 2**40))]))
        flags = int(rng.choice([0, N.HSW_REPR_MONTGOMERY, N.HSW_REPR_COMPACT64]))
        tile, parts = [(0, 0), (32, 1), (32, 4), (32, 32), (64, 2), (64, 4), (64, 16), (128, 4), (128, 8), (128, 32),
                       (0, 1), (0, 8)][int(rng.integers(0, 12))]
        split = int(rng.choice([-1, -1, 0, 1, 2]))
        helpers = int(rng.integers(0, 5))                          # waves per workgroup of the small-batch kernel
        mont_emit = int(rng.choice([1, 1, 0, 2]))                  # Montgomery cells: converted at emit time (default mode / always) or at write-out
        shift = int(rng.choice([0, 0, 1, 2, 3]))
        pack = rng.random() < 0.4
        chunk = int(rng.choice([1 << 20, 1 << 20, 1, 3, 5]))      # blocks per launch: reach the multi-launch loop
        if override:
            internals, bits, ncols, n, cursor0, flags = (override[k] for k in ("internals", "bits", "ncols", "n", "cursor0", "flags"))
            tile, parts, split, shift, pack, chunk = (override[k] for k in ("tile", "parts", "split", "shift", "pack", "chunk"))
            helpers = override.get("helpers", 0)
            mont_emit = override.get("mont_emit", 1)
        desc = dict(kind="blocks", internals=internals, bits=bits, ncols=ncols, n=n, cursor0=cursor0, flags=flags,
                    tile=tile, parts=parts, split=split, helpers=helpers, mont_emit=mont_emit, shift=shift, pack=pack, chunk=chunk)
        self.current = desc
        eng = self.engine(bits, ncols, internals)
        G, LK = eng.G, eng.lookup_cells
        plan, start_row, max_rows = None, 0, 0
        if pack:
            max_rows = int(rng.integers(G // 2 + 16, 3 * G))
            start_row = int(rng.integers(0, max_rows))
            aim = rng.random()
            if aim < 0.5 and not override:
                # half of the packed cases aim a column break at the first / last cells of a block (the first and
                # the last flush of a block are special cases of the realigning write-out)
                for _ in range(400):
                    mr, sr = int(rng.integers(G // 2 + 16, 3 * G)), 0
                    sr = int(rng.integers(0, mr))
                    try:
                        pl = N.pack_plan(eng.shape, n, sr, mr)
                    except N.HswError:
                        continue
                    at = [int(pl.break_cell[k]) % G for k in range(pl.n_breaks)]
                    if any((0 < c <= 160) if aim < 0.25 else (c >= G - 160) for c in at):
                        max_rows, start_row = mr, sr
                        break
            if override:
                start_row, max_rows = override["start_row"], override["max_rows"]
            desc.update(start_row=start_row, max_rows=max_rows)
            try:
                plan = N.pack_plan(eng.shape, n, start_row, max_rows)
            except N.HswError:
                self.stats["skipped"] += 1
                return desc
        blocks = rng.integers(0, 256, (n, 64), dtype=np.uint8)
        pre = rng.integers(0, 2**32, (n, 8), dtype=np.uint64).astype(np.uint32)
        if rng.random() < 0.1:
            blocks[0] = 0 if rng.random() < 0.5 else 255
            pre[0] = 0 if rng.random() < 0.5 else 0xFFFFFFFF
        w = 1 if flags == N.HSW_REPR_COMPACT64 else 4
        span = int(plan.span_cells) if plan is not None else n * G
        # compact cells are 8 bytes: keep the stream 16-byte aligned (two cells per shift step)
        lead = shift * (2 if w == 1 else 1)
        big = t.full((span + lead + 8, w), -1, dtype=t.int64, device="cuda")
        rows = eng.chip_rows(cursor0, n)
        dense = t.full((ncols, rows + 2, w), -1, dtype=t.int64, device="cuda")
        spread = t.full((ncols, rows + 2, w), -1, dtype=t.int64, device="cuda")
        nxt = t.empty((n, 8), dtype=t.int32, device="cuda")
        lookup = t.full((n * LK + 4, w), -1, dtype=t.int64, device="cuda") if internals else None
        tb, tp = t.from_numpy(blocks).cuda(), t.from_numpy(pre.view(np.int32)).cuda()
        a = N.WitnessArgs()
        a.d_blocks, a.d_pre_states, a.n_blocks, a.spread_cursor0 = tb.data_ptr(), tp.data_ptr(), n, cursor0
        a.d_gate = big.data_ptr() + lead * 8 * w
        a.d_chip_dense, a.d_chip_spread, a.chip_col_stride = dense.data_ptr(), spread.data_ptr(), rows + 2
        a.d_next_states = nxt.data_ptr()
        a.d_lookup = lookup.data_ptr() if internals else None
        a.flags = flags
        a.pack = C.pointer(plan) if plan is not None else None
        eng.set_option("tile", tile)
        eng.set_option("parts", parts)
        eng.set_option("split", split)
        if helpers:
            eng.set_option("helpers", helpers)
        eng.set_option("chunk_blocks", chunk)
        eng.set_option("mont_emit", mont_emit)
        try:
            rc = eng.lib.hsw_witness_blocks_ex(eng.h, C.byref(a))
            if rc == N.HSW_ERR_UNSUPPORTED:      # e.g. more than two column breaks inside one block
                self.stats["skipped"] += 1
                return desc
            assert rc == 0, (rc, eng.lib.hsw_last_error(eng.h))
            eng.synchronize()
        finally:
            eng.set_option("tile", 0)
            eng.set_option("parts", 0)
            eng.set_option("split", -1)
            if helpers:
                eng.set_option("helpers", 0)
            eng.set_option("chunk_blocks", 1 << 20)
            eng.set_option("mont_emit", 1)
        ref = self.O.Oracle(bits, ncols, check=False, internals=internals).witness_blocks(blocks, pre, cursor0=cursor0)
        eg, ed, es, el = self._expected(ref, eng, n, flags, internals)
        flat = big.cpu().numpy().view(np.uint64)
        if plan is not None:
            lens = N.gate_tape(eng.shape).astype(np.int64)
            # position of every cell under "row + len >= max_rows -> next column" (assumption A3-iii)
            idx = np.empty(n * G, dtype=np.int64)
            col, row, k = 0, start_row, 0
            for _ in range(n):
                if row + G + 4 < max_rows:
                    idx[k:k + G] = col * max_rows + row - start_row + np.arange(G)
                    row += G
                    k += G
                    continue
                for ln in lens.tolist():
                    if row + ln >= max_rows:
                        col, row = col + 1, 0
                    idx[k:k + ln] = col * max_rows + row - start_row + np.arange(ln)
                    row += ln
                    k += ln
            idx += lead
        else:
            idx = lead + np.arange(n * G, dtype=np.int64)
        got = flat[idx]
        if not np.array_equal(got, eg):
            bad = np.nonzero((got != eg).any(axis=1))[0]
            stray = np.ones(flat.shape[0], dtype=bool)
            stray[idx] = False
            stray = np.nonzero(stray & (flat != FILL).any(axis=1))[0]      # written, but not a place of the stream
            raise AssertionError("gate stream differs at %d cells, first %d (block %d cell %d; all: %s); got %s expected %s; "
                                 "buffer places of the bad cells %s; written outside the stream: %s" % (
                len(bad), bad[0], bad[0] // G, bad[0] % G, [int(b) for b in bad[:48]], got[bad[0]], eg[bad[0]],
                [int(i) for i in idx[bad[:48]]], [int(i) for i in stray[:48]]))
        mask = np.ones(flat.shape[0], dtype=bool)
        mask[idx] = False
        assert (flat[mask] == FILL).all(), "cells outside the stream were written"
        d, s = dense.cpu().numpy().view(np.uint64), spread.cpu().numpy().view(np.uint64)
        # rows the call owns; cells of the first / last row that belong to neighbouring calls keep the fill
        first, last = cursor0 % ncols, (cursor0 + n * eng.limb_calls - 1) % ncols
        own = np.ones((ncols, rows), dtype=bool)
        own[:first, 0] = False
        own[last + 1:, rows - 1] = False
        assert np.array_equal(d[:, :rows][own], ed[:, :rows][own]), "chip dense cells differ"
        assert np.array_equal(s[:, :rows][own], es[:, :rows][own]), "chip spread cells differ"
        assert (d[:, :rows][~own] == FILL).all() and (s[:, :rows][~own] == FILL).all(), "neighbour's chip cells written"
        assert (d[:, rows:] == FILL).all() and (s[:, rows:] == FILL).all()
        assert np.array_equal(nxt.cpu().numpy().view(np.uint32), ref["next_states"]), "next states differ"
        if internals:
            lk = lookup.cpu().numpy().view(np.uint64)
            assert np.array_equal(lk[: n * LK], el), "lookup column differs"
            assert (lk[n * LK:] == FILL).all()
        self.stats["block_runs"] += 1
        self.stats["block_cells"] += n * (G + 2 * eng.limb_calls)
        return desc

    # ---------------------------------------------------------------- gadget block streams
    def _random_digests(self, nd, nb_choices, equal=False):
        rng = self.rng
        sizes, msgs, pres = [], [], []
        for _ in range(nd):
            nb = int(rng.choice(nb_choices)) if not (equal and sizes) else sizes[0] // 64
            pre_rounds = int(rng.integers(0, 3)) if rng.random() < 0.4 else 0
            # total padded rounds must satisfy max(pre_rounds, 1) <= num_round <= pre_rounds + nb (lib.rs:89-90)
            num_round = int(rng.integers(max(pre_rounds, 1), pre_rounds + nb + 1))
            lo, hi = max(0, 64 * (num_round - 1) - 8), 64 * num_round - 9           # ceil((ln + 9) / 64) == num_round
            ln = int(rng.integers(lo, hi + 1)) if rng.random() < 0.8 else int(rng.choice([lo, hi]))
            sizes.append(64 * nb)
            pres.append(64 * pre_rounds)
            msgs.append(rng.integers(0, 256, ln, dtype=np.uint8).tobytes())
        return sizes, msgs, pres

    def gadget_case(self):
        """Sha256DynamicConfig.digest as block streams (the f1 front-end): random groupings of digest /
        digest_batch calls in one context (host- or GPU-side chaining, zero-copy or staged inputs), optionally
        a reset and a second pass; the concatenated streams against the oracle's."""
        rng, N, hsw = self.rng, self.N, self.hsw
        internals = rng.random() < 0.3
        bits = int(rng.choice([8, 8, 16, 4]))
        ncols = int(rng.integers(1, 5))
        nd = int(rng.integers(1, 7))
        sizes, msgs, pres = self._random_digests(nd, [1, 1, 2, 3, 4, 8, 16, 40])
        rc = bool(rng.integers(0, 2))
        mont = rng.random() < 0.4
        again = rng.random() < 0.25
        self.current = dict(kind="gadget", internals=internals, bits=bits, ncols=ncols, sizes=sizes,
                            lens=[len(m) for m in msgs], pres=pres, rc=rc, mont=mont, again=again)
        eng = self.engine(bits, ncols, internals)
        cfg = hsw.Sha256DynamicConfig(eng, sizes, is_input_range_check=rc)
        try:
            if mont:
                cfg.set_repr(N.HSW_REPR_MONTGOMERY)
            for rnd in range(2 if again else 1):
                if rnd:
                    cfg.reset()
                res, i = [], 0
                while i < nd:
                    k = int(rng.integers(1, nd - i + 1))
                    if k == 1 and rng.random() < 0.5:
                        res.append(cfg.digest(msgs[i], pres[i]))
                    else:
                        res += cfg.digest_batch(msgs[i:i + k], pres[i:i + k])
                    i += k
            st = cfg.streams()
            rep = cfg.verify()
        finally:
            cfg.close()
        o = self.O.Oracle(bits, ncols, check=False, internals=internals)
        ds = [o.digest(m, mx, p) for m, mx, p in zip(msgs, sizes, pres)]
        blocks = np.concatenate([d["blocks"][: mx // 64] for d, mx in zip(ds, sizes)])
        pre = np.concatenate([d["pre_states"][: mx // 64] for d, mx in zip(ds, sizes)])
        ref = self.O.Oracle(bits, ncols, check=False, internals=internals).witness_blocks(blocks, pre, cursor0=0)
        conv = self.O.to_montgomery if mont else (lambda x: x)
        for m, r, d in zip(msgs, res, ds):
            assert r.output_bytes == hashlib.sha256(m).digest() == d["digest"], "digest differs"
        assert np.array_equal(st["gate"], conv(ref["gate"])), "gadget gate stream differs"
        assert np.array_equal(st["dense"], conv(ref["dense"][:, : st["rows"]])), "gadget chip dense differs"
        assert np.array_equal(st["spread"], conv(ref["spread"][:, : st["rows"]])), "gadget chip spread differs"
        assert rep["violations"] == 0, rep
        self.stats["gadget_runs"] = self.stats.get("gadget_runs", 0) + 1
        self.stats["block_cells"] += len(ref["gate"])

    # ---------------------------------------------------------------- host delivery
    def host_case(self):
        """hsw_witness_blocks_host: numpy in, numpy out -- the pipelined path (cursor a multiple of the column
        count: 128-block chunks through two staging slots) and the single-shot one."""
        rng, N = self.rng, self.N
        bits = int(rng.choice([8, 8, 16, 4]))
        ncols = int(rng.integers(1, 6))
        n = int(rng.choice([1, 7, 127, 128, 129, 200, 257, 300]))
        cursor0 = int(rng.integers(0, 10**5))
        if rng.random() < 0.6:
            cursor0 -= cursor0 % ncols
        flags = int(rng.choice([0, N.HSW_REPR_MONTGOMERY, N.HSW_REPR_COMPACT64]))
        pinned = bool(rng.integers(0, 2))
        register = (not pinned) and bool(rng.integers(0, 2))      # HSW_HOST_REGISTER: pin the caller's pageable buffers in place
        self.current = dict(kind="host", bits=bits, ncols=ncols, n=n, cursor0=cursor0, flags=flags, pinned=pinned, register=register)
        eng = self.engine(bits, ncols, False)
        blocks = rng.integers(0, 256, (n, 64), dtype=np.uint8)
        pre = rng.integers(0, 2**32, (n, 8), dtype=np.uint64).astype(np.uint32)
        got = eng.witness_blocks_host(blocks, pre, cursor0=cursor0, flags=flags | (N.HSW_HOST_REGISTER if register else 0), pinned=pinned)
        ref = self.O.Oracle(bits, ncols, check=False).witness_blocks(blocks, pre, cursor0=cursor0)
        eg, ed, es, _ = self._expected(ref, eng, n, flags, False)
        assert np.array_equal(got["gate"], eg), "host gate stream differs"
        assert np.array_equal(got["dense"], ed) and np.array_equal(got["spread"], es), "host chip columns differ"
        assert np.array_equal(got["next_states"], ref["next_states"])
        self.stats["host_runs"] = self.stats.get("host_runs", 0) + 1
        self.stats["block_cells"] += n * (eng.G + 2 * eng.limb_calls)

    # ---------------------------------------------------------------- chain pre-pass
    def chain_case(self):
        """hsw_sha256_chain: the plain SHA-256 chain that makes the blocks of a message independent
        (lib.rs:188,236), one lane per message, from the FIPS IV or from given prefix states (lib.rs:153-160)."""
        rng, t = self.rng, self.torch
        nm = int(rng.choice([1, 2, 63, 64, 65, 255, 256, 257, 1000]))
        bpm = int(rng.choice([1, 2, 3, 16, 33]))
        custom = bool(rng.integers(0, 2))
        self.current = dict(kind="chain", n_messages=nm, blocks_per_message=bpm, custom_init=custom)
        eng = self.engine(8, 2, False)
        blocks = rng.integers(0, 256, (nm * bpm, 64), dtype=np.uint8)
        init = rng.integers(0, 2**32, (nm, 8), dtype=np.uint64).astype(np.uint32) if custom else None
        pre = eng.sha256_chain(t.from_numpy(blocks).cuda(), nm, bpm,
                               t.from_numpy(init.view(np.int32)).cuda() if custom else None).cpu().numpy().view(np.uint32)
        # a handful of messages, chained with the oracle's plain compression (first, last, some in between)
        for mi in sorted({0, nm - 1, nm // 2, int(rng.integers(0, nm))}):
            st = init[mi].copy() if custom else self.O.INIT_STATE.copy()
            for j in range(bpm):
                assert np.array_equal(pre[mi * bpm + j], st), "pre-state of message %d block %d differs" % (mi, j)
                st = self.O.plain_compress(st, blocks[mi * bpm + j])
        self.stats["chain_runs"] = self.stats.get("chain_runs", 0) + 1

    # ---------------------------------------------------------------- whole digests
    def digest_case(self):
        rng, N, hsw = self.rng, self.N, self.hsw
        bits = int(rng.choice([8, 8, 8, 16, 4, 2]))
        ncols = int(rng.choice([2, 2, 1, 3]))
        nd = int(rng.integers(1, 5))
        sizes, msgs, pres = self._random_digests(nd, [1, 2, 3, 4, 8], equal=rng.random() < 0.4)
        rc = bool(rng.integers(0, 2))
        batch = bool(rng.integers(0, 2))
        mont = rng.random() < 0.35
        columns = rng.random() < 0.4
        # hsw_gadget_set_origin: the Context stands at (column, row), may cache its zero cell and have lookups queued
        origin = None
        if rng.random() < 0.35:
            origin = (int(rng.integers(0, 4)), 0, bool(rng.integers(0, 2)), int(rng.choice([0, 0, 1, int(rng.integers(0, 3000))])))
        # HSW_GADGET_INDEPENDENT: every digest a Context of its own (linear streams, no origin)
        independent = origin is None and not columns and rng.random() < 0.25
        distinct = rng.random() < 0.3
        desc = dict(kind="digests", bits=bits, ncols=ncols, sizes=sizes, lens=[len(m) for m in msgs], pres=pres, rc=rc,
                    batch=batch, mont=mont, columns=columns, origin=origin, independent=independent, distinct=distinct)
        self.current = desc
        eng = self.engine(bits, ncols, True)
        try:
            cfg = hsw.Sha256DynamicConfig(eng, sizes, is_input_range_check=rc, whole_digest=True, independent=independent)
        except hsw.HswError as ex:
            if independent and ex.status == N.HSW_ERR_UNSUPPORTED:       # a context's chip rows would not start on a row boundary
                self.stats["skipped"] += 1
                return desc
            raise
        max_rows = None
        try:
            if mont:
                cfg.set_repr(N.HSW_REPR_MONTGOMERY)
            if columns:
                max_rows = int(rng.integers(eng.G + 16, 4 * eng.G))
                desc["max_rows"] = max_rows
            if origin is not None:
                if columns:
                    r0 = int(rng.choice([0, 1, max_rows - 1, int(rng.integers(0, max_rows))]))
                    origin = (origin[0], r0, origin[2], origin[3])
                    desc["origin"] = origin
                    self.current = desc
                cfg.set_origin(*origin)
            if columns:
                try:
                    cfg.set_columns(max_rows)
                except hsw.HswError as ex:
                    if ex.status in (N.HSW_ERR_TOO_LARGE, N.HSW_ERR_UNSUPPORTED):
                        self.stats["skipped"] += 1
                        return desc
                    raise
            res = cfg.digest_batch(msgs, pres) if batch else [cfg.digest(m, p) for m, p in zip(msgs, pres)]
            st = cfg.streams()
            rep = cfg.verify()
            if rng.random() < 0.3:                    # hsw_gadget_download_region: the same image in host memory
                host = cfg.download_region(pinned=bool(rng.integers(0, 2)))
                for k in ("gate", "lookup", "dense", "spread"):
                    assert np.array_equal(host[k], st[k]), "download_region %s differs from the device image" % k
                self.stats["downloads"] = self.stats.get("downloads", 0) + 1
            if distinct:                              # distinct-value delivery + host replay: the same image again
                got = cfg.download_region_distinct(threads=int(rng.integers(1, 5)))
                for k in ("gate", "lookup", "dense", "spread"):
                    assert np.array_equal(got[k], st[k]), "distinct-value delivery: %s differs from the device image" % k
                assert 0 < got["n_distinct"] < int(cfg.view().gate_cells)
                self.stats["distinct"] = self.stats.get("distinct", 0) + 1
        finally:
            cfg.close()
        for m, r in zip(msgs, res):
            assert r.output_bytes == hashlib.sha256(m).digest(), "digest differs from SHA-256"
        conv = self.O.to_montgomery if mont else (lambda x: x)
        if independent:
            # every digest's slice of the streams is what a fresh single-digest gadget writes
            at_g = at_l = 0
            lc = int(eng.shape.limb_calls_per_block)
            for m, mx, pz, r in zip(msgs, sizes, pres, res):
                one = self.O.digest_cells([m], [mx], [pz], rc, num_bits_lookup=bits, num_advice_columns=ncols)
                ng, nl = len(one["gate"]), len(one["lookup"])
                assert r.prologue_cell == at_g and r.end_cell == at_g + ng, "independent context: positions differ"
                assert np.array_equal(st["gate"][at_g: at_g + ng], conv(one["gate"])), "independent context: gate slice differs"
                assert np.array_equal(st["lookup"][at_l: at_l + nl], conv(one["lookup"])), "independent context: lookup slice differs"
                row0 = r.first_block * lc // ncols    # (a context's chip rows start on a row of their own, or the gadget is refused)
                for k in ("dense", "spread"):
                    c1 = conv(one[k])
                    assert np.array_equal(st[k][:, row0: row0 + c1.shape[1]], c1), "independent context: chip %s rows differ" % k
                at_g, at_l = at_g + ng, at_l + nl
            assert at_g == len(st["gate"]) and rep["violations"] == 0, rep
            self.stats["independent"] = self.stats.get("independent", 0) + 1
            self.stats["digest_runs"] += 1
            self.stats["digest_cells"] += at_g
            return desc
        if nd >= 2 and origin is None and rng.random() < 0.25:
            self._seek_split(eng, sizes, msgs, pres, rc, mont, max_rows if columns else None, st, int(rng.integers(1, nd)))
        zero = bool(origin[2]) if origin is not None else False
        lq = origin[3] if origin is not None else 0
        ref = self.O.digest_cells(msgs, sizes, pres, rc, num_bits_lookup=bits, num_advice_columns=ncols, zero_cell_loaded=zero)
        assert not st["lookup"][:lq].any(), "lookup entries queued by the caller were written"
        st = dict(st, lookup=st["lookup"][lq:])
        if origin is not None:
            self.stats["origins"] = self.stats.get("origins", 0) + 1
        if columns:
            # halo2-lib v0.2.x FlexGate::assign_region over the oracle's call tape (assumption A3-iii)
            g = conv(ref["gate"])
            cols, col, row, at = [np.zeros((max_rows, 4), dtype=np.uint64)], 0, (origin[1] if origin is not None else 0), 0
            for ln in ref["call_lens"].tolist():
                if row + ln >= max_rows:
                    cols.append(np.zeros((max_rows, 4), dtype=np.uint64))
                    col, row = col + 1, 0
                cols[col][row:row + ln] = g[at:at + ln]
                row += ln
                at += ln
            assert at == len(g)
            assert np.array_equal(st["gate"], np.stack(cols)), "column image differs"
        else:
            assert np.array_equal(st["gate"], conv(ref["gate"])), "whole-digest gate stream differs"
        assert np.array_equal(st["lookup"], conv(ref["lookup"])), "lookup stream differs"
        assert np.array_equal(st["dense"], conv(ref["dense"][:, : st["rows"]]))
        assert np.array_equal(st["spread"], conv(ref["spread"][:, : st["rows"]]))
        assert rep["violations"] == 0, rep
        self.stats["digest_runs"] += 1
        self.stats["digest_cells"] += len(ref["gate"])
        return desc

    def _seek_split(self, eng, sizes, msgs, pres, rc, mont, max_rows, full, k):
        """hsw_gadget_seek: gadget A assigns digests [0, k), gadget B seeks to k and assigns the rest; together
        they must give the image one gadget writes (`full`), each touching only its own cells."""
        hsw, N = self.hsw, self.N
        parts = []
        for first, last in ((0, k), (k, len(msgs))):
            cfg = hsw.Sha256DynamicConfig(eng, sizes, is_input_range_check=rc, whole_digest=True)
            try:
                if mont:
                    cfg.set_repr(N.HSW_REPR_MONTGOMERY)
                if max_rows:
                    cfg.set_columns(max_rows)
                if first:
                    cfg.seek(first)
                res = cfg.digest_batch(msgs[first:last], pres[first:last])
                assert cfg.verify()["violations"] == 0
                v = cfg.view()
                n_gate = int(v.max_rows * v.columns) if max_rows else int(v.gate_capacity)

                def grab(ptr, n_cells):
                    a = np.zeros((n_cells, 4), dtype=np.uint64)
                    cfg._ok(cfg.lib.hsw_download(eng.h, a.ctypes.data, ptr, n_cells * 32))
                    return a
                parts.append(dict(gate=grab(v.d_gate, n_gate), lookup=grab(v.d_lookup, int(v.lookup_capacity)),
                                  cells=int(v.gate_cells), lookups=int(v.lookup_cells), res=res))
            finally:
                cfg.close()
        a, b = parts
        for m, r in zip(msgs[k:], b["res"]):
            assert r.output_bytes == hashlib.sha256(m).digest()
        if max_rows:
            fg = full["gate"].reshape(-1, 4)
            # the column image is zero-initialised: the two parts are disjoint and their union is the whole
            assert np.array_equal(a["gate"] | b["gate"], fg), "seek: union of the two gadgets' images differs"
            assert not (a["gate"].any(axis=1) & b["gate"].any(axis=1)).any(), "seek: the two gadgets overlap"
        else:
            cut, end = a["cells"], b["cells"]
            assert end == len(full["gate"])
            assert np.array_equal(a["gate"][:cut], full["gate"][:cut]) and np.array_equal(b["gate"][cut:end], full["gate"][cut:end])
        lc, le = a["lookups"], b["lookups"]
        assert np.array_equal(a["lookup"][:lc], full["lookup"][:lc]) and np.array_equal(b["lookup"][lc:le], full["lookup"][lc:le])
        self.stats["seek_splits"] = self.stats.get("seek_splits", 0) + 1

    # The case about to run is also written to a one-line trace file: a GPU fault ends the process without a
    # Python exception, and the file then names the case that was on the device (HSW_FUZZ_TRACE overrides the path).
    @property
    def current(self):
        return getattr(self, "_current", None)

    @current.setter
    def current(self, v):
        self._current = v
        if v is None:
            return
        if not hasattr(self, "_trace"):
            path = os.environ.get("HSW_FUZZ_TRACE") or (os.path.join(ROOT, "gpurun_out", "fuzz_last_case.txt")
                                                        if os.path.isdir(os.path.join(ROOT, "gpurun_out")) else None)
            self._trace = open(path, "w") if path else None
        if self._trace:
            self._trace.seek(0)
            self._trace.truncate()
            self._trace.write(repr(v) + "\n")
            self._trace.flush()

    def run(self, seconds=None, iterations=None, log=None):
        t0 = time.time()
        it = 0
        last = t0
        while (seconds is None or time.time() - t0 < seconds) and (iterations is None or it < iterations):
            self.current = None
            try:
                u = self.rng.random()
                (self.block_case() if u < 0.55 else self.digest_case() if u < 0.8 else self.gadget_case() if u < 0.93
                 else self.host_case() if u < 0.98 else self.chain_case())
            except Exception:
                print("FAILED case:", self.current, file=sys.stderr, flush=True)
                raise
            it += 1
            if log and time.time() - last > 20:
                last = time.time()
                print("%6.0f s  %s" % (last - t0, self.stats), file=log, flush=True)
        return dict(self.stats, iterations=it, seconds=round(time.time() - t0, 1))


if __name__ == "__main__":
    if len(sys.argv) > 2 and sys.argv[1] == "replay":      # python tests/fuzz_parity.py replay "<FAILED case dict>" [seed]
        import ast
        f = Fuzzer(int(sys.argv[3]) if len(sys.argv) > 3 else 1)
        case = ast.literal_eval(sys.argv[2])
        assert case["kind"] == "blocks", "only block cases replay"
        try:
            f.block_case(override=case)
            print("replay: clean")
        finally:
            f.close()
        sys.exit(0)
    secs = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else int(time.time())
    f = Fuzzer(seed)
    try:
        out = f.run(seconds=secs, log=sys.stdout)
    finally:
        f.close()
    import json
    print(json.dumps(dict(out, seed=seed, violations=0)))
