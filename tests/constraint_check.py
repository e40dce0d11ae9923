"""MockProver-style check of a witness against the constraint STRUCTURE the
reference's source specifies (oracle.constraint_system): gate rows, copy
constraints, fixed constants, range bounds, chip-cell ties, both lookup tables.
It uses no oracle-computed VALUE, so a stream that passes is -- by the
uniqueness argument of SURVEY 8c -- the witness for its inputs.

Works on numpy arrays or torch tensors (same indexing API); `xp` is the module."""
import numpy as np

P_LIMBS = [0x43e1f593f0000001, 0x2833e84879b97091, 0xb85045b68181585d, 0x30644e72e131a029]


def _s64(x):
    return x - (1 << 64) if x >= (1 << 63) else x


def check_block_batch(xp, cs, gate, blocks, pre, dense_limb, spread_limb, next_states, lookup=None,
                      spread_table=None, num_bits_lookup=8):
    """gate: (n, G, 4) int64 (canonical limbs); blocks (n,64) ints; pre (n,8) ints (u32 values);
    dense_limb / spread_limb: (n, LC) int64 chip values in limb-call order; next_states (n,8) u32 values;
    lookup: (n, LK) int64 or None.  Raises AssertionError on the first violated constraint class.
    Returns the number of individual constraints checked."""
    n = gate.shape[0]
    checked = 0
    as_t = (lambda a: xp.asarray(a)) if xp is np else (lambda a: xp.as_tensor(a, device=gate.device))

    def cellval(ids):
        """(n, len(ids), 4) values of cells `ids` (block-relative stream cells or external cells)."""
        ids_np = np.asarray(ids)
        out = xp.zeros((n, len(ids_np), 4), dtype=gate.dtype) if xp is np else xp.zeros(
            (n, len(ids_np), 4), dtype=gate.dtype, device=gate.device)
        inside = ids_np >= 0
        if inside.any():
            out[:, as_t(np.nonzero(inside)[0])] = gate[:, as_t(ids_np[inside])]
        for j in np.nonzero(~inside)[0]:
            cid = int(ids_np[j])
            if -64 <= cid <= -1:
                out[:, j, 0] = blocks[:, -1 - cid]
            elif -107 <= cid <= -100:
                out[:, j, 0] = pre[:, -100 - cid]
            elif cid == -1000:
                pass                                   # the zero cell
            else:
                raise AssertionError("unexpected external cell %d" % cid)
        return out

    # 1. gate rows: x0 + x1*x2 = x3 (mod p).  Narrow rows: exact in Z, hence mod 2^64.
    st = cs["gate_starts"]
    x = [gate[:, as_t(st + k)] for k in range(4)]
    narrow = (x[0][..., 1:] == 0).all(-1) & (x[1][..., 1:] == 0).all(-1) & (x[2][..., 1:] == 0).all(-1) & \
             (x[3][..., 1:] == 0).all(-1)
    lhs = x[0][..., 0] + x[1][..., 0] * x[2][..., 0]
    assert bool(((lhs == x[3][..., 0]) | ~narrow).all()), "gate equation violated"
    wide = ~narrow
    if bool(wide.any()):
        # the only rows with a > 64-bit cell: [a, p-a, 1, 0] and [M, p-a, 1, M-a] (compression.rs:320-335)
        w0, w1, w2, w3 = (x[k][wide] for k in range(4))
        pl = as_t(np.array([_s64(v) for v in P_LIMBS], dtype=np.int64))
        assert bool((w2[:, 0] == 1).all() and (w2[:, 1:] == 0).all())
        assert bool((w1[:, 1:] == pl[1:]).all()), "wide cell is not p - small"
        a = pl[0] - w1[:, 0]
        assert bool(((a > 0) & (a <= 0x55555555)).all())
        assert bool(((w0[:, 0] - a) == w3[:, 0]).all() and (w0[:, 1:] == 0).all() and (w3[:, 1:] == 0).all())
    checked += n * len(st)

    # 2. copy constraints
    eq = cs["eq"]
    keep = (eq[:, 0] != -2000) & (eq[:, 1] != -2000)           # halo2-base-internal witnesses outside the stream
    a, b = cellval(eq[keep, 0]), cellval(eq[keep, 1])
    bad = (a != b).any(-1)
    assert not bool(bad.any()), "copy constraint violated (first pair index %d)" % int(np.nonzero(
        (bad.any(0).cpu().numpy() if xp is not np else bad.any(0)))[0][0])
    checked += n * int(keep.sum())

    # 3. constants
    kc = cs["const"]
    v = cellval(kc[:, 0])
    want = as_t(kc[:, 1].astype(np.int64))
    assert bool((v[..., 0] == want).all() and (v[..., 1:] == 0).all()), "constant cell differs"
    checked += n * len(kc)

    # 4. range bounds (cells in the stream)
    rg = cs["range"]
    rg = rg[rg[:, 0] != -2000]
    v = cellval(rg[:, 0])
    bits = as_t(rg[:, 1].astype(np.int64))
    assert bool((v[..., 1:] == 0).all() and (v[..., 0] >= 0).all() and ((v[..., 0] >> bits) == 0).all()), "range bound violated"
    checked += n * len(rg)

    # 5. chip cells tied to stream cells + the spread lookup (spread.rs:56-62, 209-210, 226-227)
    ch = cs["chip"]
    d_src, s_src = cellval(ch[:, 0]), cellval(ch[:, 1])
    assert bool((d_src[..., 0] == dense_limb).all() and (d_src[..., 1:] == 0).all()), "chip dense cell != limb cell"
    assert bool((s_src[..., 0] == spread_limb).all() and (s_src[..., 1:] == 0).all()), "chip spread cell != witness cell"
    tab = as_t(np.asarray(spread_table, dtype=np.int64))
    assert bool(((dense_limb >= 0) & (dense_limb < (1 << num_bits_lookup))).all())
    assert bool((tab[dense_limb] == spread_limb).all()), "(dense, spread) is not a row of the spread table"
    checked += 3 * n * len(ch)

    # 6. next states
    ns = cellval(cs["next_state_cells"])
    assert bool((ns[..., 0] == next_states).all() and (ns[..., 1:] == 0).all()), "next_state cells differ"
    checked += 8 * n

    # 7. lookup-advice column (internals mode): entry j copies its source cell and is < 2^16
    if lookup is not None:
        src = cs["lookup_src"]
        assert (src != -2000).all(), "lookup sources are hidden without internals"
        v = cellval(src)
        assert bool((v[..., 0] == lookup).all() and (v[..., 1:] == 0).all()), "lookup column entry != source cell"
        assert bool(((lookup >= 0) & (lookup < 65536)).all())
        checked += 2 * n * len(src)
    return checked


def chip_in_call_order(xp, col_array, n_blocks, LC, ncols):
    """(ncols, rows, 4) column image written from cursor 0 -> (n_blocks, LC) low limbs in limb-call order."""
    rows = col_array.shape[1]
    flat = col_array[..., 0].transpose(0, 1) if xp is not np else col_array[..., 0].T      # (rows, ncols)
    flat = flat.reshape(rows * ncols)[: n_blocks * LC]
    return flat.reshape(n_blocks, LC)


# ---------------------------------------------------------------------------
# Whole-digest streams: prologue | zero cell | blocks | epilogue.
P_INT = sum(v << (64 * i) for i, v in enumerate(P_LIMBS))


def cells_to_int(a):
    """(n,4) uint64 canonical limbs -> list of Python ints."""
    a = np.asarray(a, dtype=np.uint64)
    l0, l1, l2, l3 = (a[:, i].tolist() for i in range(4))
    return [w | (x << 64) | (y << 128) | (z << 192) for w, x, y, z in zip(l0, l1, l2, l3)]


def check_whole_stream(ref, gate, lookup, dense, spread, spread_table, num_bits_lookup=8, ncols=2):
    """MockProver-style check of a whole-digest witness (gate stream, lookup-advice stream,
    chip columns) against the constraint STRUCTURE recorded by oracle.digest_cells(record=True)
    (`ref`): every gate row, copy constraint, constant, range bound, chip tie, both lookup
    tables.  No oracle-computed VALUE is used.  Returns the number of constraints checked."""
    cs = ref["cs"]
    g = cells_to_int(gate)
    assert all(v < P_INT for v in g), "non-canonical cell"
    n = 0
    for r in ref["gate_rows"].tolist():
        assert (g[r] + g[r + 1] * g[r + 2] - g[r + 3]) % P_INT == 0, "gate row at cell %d violated" % r
    n += len(ref["gate_rows"])
    for a, b in cs["eq"].tolist():
        assert a >= 0 and b >= 0, "whole-digest streams have no cells outside the stream"
        assert g[a] == g[b], "copy constraint (%d, %d) violated" % (a, b)
    n += len(cs["eq"])
    for c, k in cs["const"].tolist():
        assert g[c] == (k if k >= 0 else P_INT + k), "constant at cell %d differs" % c
    n += len(cs["const"])
    for c, bits in cs["range"].tolist():
        assert g[c] < (1 << bits), "range bound at cell %d violated" % c
    n += len(cs["range"])
    # chip columns: limb call j -> (column j % ncols, row j // ncols); tied to gate cells; a table row
    tab = {int(d): int(s) for d, s in zip(*spread_table)}
    dl, sl = np.asarray(dense), np.asarray(spread)
    for j, (dc, sc) in enumerate(cs["chip"].tolist()):
        dv, sv = dl[j % ncols, j // ncols], sl[j % ncols, j // ncols]
        assert not dv[1:].any() and not sv[1:].any()
        assert int(dv[0]) == g[dc] and int(sv[0]) == g[sc], "chip cell of limb call %d not tied" % j
        assert tab[int(dv[0])] == int(sv[0]), "(dense, spread) of limb call %d is not a table row" % j
    n += 3 * len(cs["chip"])
    lk = cells_to_int(lookup)
    assert len(lk) == len(cs["lookup_src"])
    for j, src in enumerate(cs["lookup_src"].tolist()):
        assert lk[j] == g[src] and lk[j] < 65536, "lookup entry %d" % j
    n += 2 * len(lk)
    return n
