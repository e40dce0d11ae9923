"""ctypes binding of the CPU oracle (oracle/hsw_oracle.c).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg -- never by the product package.  See
hsw_oracle.h for what is restated, the assumptions and how it is pinned.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "liboracle.so")

CELL_DTYPE = np.uint64  # one cell = 4 little-endian u64 limbs (canonical Fr)


class Stats(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in (
        "load_witness", "add", "neg", "mul_add", "load_zero",
        "assert_equal", "range_check16", "range_check32", "range_check_other",
        "spread_calls", "spread_limb_calls", "even_odd_calls",
        "gate_cells", "chip_cells")]

    def as_dict(self):
        return {n: int(getattr(self, n)) for n, _ in self._fields_}


def _cpu_signature():
    """ISA flags of this host: liboracle.so is built with -march=native, so a
    library built on another machine (the build container vs the GPU box) must
    be rebuilt before it is loaded."""
    try:
        import hashlib
        for line in open("/proc/cpuinfo"):
            if line.startswith("flags"):
                return hashlib.sha256(line.encode()).hexdigest()[:16]
    except Exception:
        pass
    return "unknown"


def build(force=False):
    """Compile liboracle.so with the committed Makefile (gcc only)."""
    src = os.path.join(_HERE, "hsw_oracle.c")
    stamp = os.path.join(_HERE, ".build_host")
    sig = _cpu_signature()
    try:
        same_host = open(stamp).read().strip() == sig
    except Exception:
        same_host = False
    if (force or not same_host or not os.path.exists(_LIB_PATH)
            or os.path.getmtime(_LIB_PATH) < os.path.getmtime(src)):
        subprocess.check_call(["make", "-C", _HERE, "-s", "clean"])
        subprocess.check_call(["make", "-C", _HERE, "-s"])
        with open(stamp, "w") as f:
            f.write(sig)
    return _LIB_PATH


class DigestLayout(C.Structure):
    """oracle_digest_layout_t"""
    _fields_ = [(n, C.c_size_t) for n in (
        "prologue_cells", "zero_cells", "block_cells", "epilogue_cells",
        "prologue_lookups", "block_lookups", "epilogue_lookups",
        "num_round", "target_round", "n_blocks")] + [("input_len_cell", C.c_int64)]

    def as_dict(self):
        return {n: int(getattr(self, n)) for n, _ in self._fields_}


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_LIB_PATH)
        L.oracle_create.restype = C.c_void_p
        L.oracle_create.argtypes = [C.c_int, C.c_int, C.c_int]
        L.oracle_destroy.argtypes = [C.c_void_p]
        L.oracle_set_outputs.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p,
                                         C.c_void_p, C.c_size_t, C.c_uint64]
        L.oracle_set_cursor.argtypes = [C.c_void_p, C.c_uint64]
        L.oracle_set_kinds.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
        L.oracle_record_constraints.argtypes = [C.c_void_p, C.c_void_p]
        L.oracle_set_internals.argtypes = [C.c_void_p, C.c_int]
        L.oracle_set_lookup_output.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
        L.oracle_lookup_len.restype = C.c_size_t
        L.oracle_lookup_len.argtypes = [C.c_void_p]
        L.oracle_get_cursor.restype = C.c_uint64
        L.oracle_get_cursor.argtypes = [C.c_void_p]
        L.oracle_gate_len.restype = C.c_size_t
        L.oracle_gate_len.argtypes = [C.c_void_p]
        L.oracle_get_stats.argtypes = [C.c_void_p, C.POINTER(Stats)]
        L.oracle_failed.restype = C.c_int
        L.oracle_failed.argtypes = [C.c_void_p, C.POINTER(C.c_char_p)]
        L.oracle_sha256_compression.restype = C.c_int
        L.oracle_sha256_compression.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.oracle_witness_blocks.restype = C.c_int
        L.oracle_witness_blocks.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]
        L.oracle_digest.restype = C.c_int
        L.oracle_digest.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_size_t, C.c_size_t,
                                    C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.oracle_digest_cells.restype = C.c_int
        L.oracle_digest_cells.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_size_t, C.c_size_t, C.c_int,
                                          C.c_void_p, C.POINTER(DigestLayout)]
        L.oracle_set_context.argtypes = [C.c_void_p, C.c_int]
        L.oracle_set_tape.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t]
        L.oracle_tape_calls.restype = C.c_size_t
        L.oracle_tape_calls.argtypes = [C.c_void_p]
        L.oracle_tape_rows.restype = C.c_size_t
        L.oracle_tape_rows.argtypes = [C.c_void_p]
        L.oracle_plain_compress.argtypes = [C.c_void_p, C.c_void_p]
        L.oracle_spread_table_entry.restype = C.c_uint64
        L.oracle_spread_table_entry.argtypes = [C.c_uint32]
        L.oracle_to_montgomery.argtypes = [C.c_void_p, C.c_size_t]
        L.oracle_measure_shape.restype = C.c_int
        L.oracle_measure_shape.argtypes = [C.c_int, C.c_int, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
        _lib = L
    return _lib


INIT_STATE = np.array([0x6a09e667, 0xbb67ae85, 0x3c6ef372, 0xa54ff53a,
                       0x510e527f, 0x9b05688c, 0x1f83d9ab, 0x5be0cd19], dtype=np.uint32)


_shape_cache = {}


def measure_shape(num_bits_lookup=8, num_advice_columns=2, internals=False):
    """(gate cells, limb calls) per block, measured by running one block.  With
    internals=True the gate count includes halo2-base's range_check cells (A3)."""
    key = (num_bits_lookup, num_advice_columns, bool(internals))
    if key not in _shape_cache:
        L = lib()
        h = L.oracle_create(num_bits_lookup, num_advice_columns, 1)
        if not h:
            raise ValueError("bad shape")
        L.oracle_set_internals(h, 1 if internals else 0)
        L.oracle_set_outputs(h, None, 0, None, None, 0, 0)
        blk = (np.arange(64, dtype=np.uint8) * 37 + 11).astype(np.uint8)
        nxt = np.zeros(8, dtype=np.uint32)
        L.oracle_sha256_compression(h, blk.ctypes.data, INIT_STATE.ctypes.data, nxt.ctypes.data)
        bad = L.oracle_failed(h, None)
        res = (int(L.oracle_gate_len(h)), int(L.oracle_get_cursor(h)), int(L.oracle_lookup_len(h)))
        L.oracle_destroy(h)
        if bad:
            raise RuntimeError("oracle self-check failed while measuring shape")
        _shape_cache[key] = res
    return _shape_cache[key][:2]


def lookup_cells_per_block(num_bits_lookup=8, num_advice_columns=2):
    measure_shape(num_bits_lookup, num_advice_columns, False)
    return _shape_cache[(num_bits_lookup, num_advice_columns, False)][2]


class Oracle:
    """One SpreadConfig + gate context, mirroring the reference's mutable cursors."""

    def __init__(self, num_bits_lookup=8, num_advice_columns=2, check=True, internals=False):
        self.L = lib()
        self.bits, self.ncols = num_bits_lookup, num_advice_columns
        self.internals = bool(internals)
        self.h = self.L.oracle_create(num_bits_lookup, num_advice_columns, 1 if check else 0)
        if not self.h:
            raise ValueError("bad shape: 16 %% num_bits_lookup must be 0 (spread.rs:37)")
        self.L.oracle_set_internals(self.h, 1 if internals else 0)
        self._keep = None

    def close(self):
        if self.h:
            self.L.oracle_destroy(self.h)
            self.h = None

    def __del__(self):
        self.close()

    def _check(self):
        msg = C.c_char_p()
        if self.L.oracle_failed(self.h, C.byref(msg)):
            raise AssertionError("oracle constraint self-check failed: " + msg.value.decode())

    def stats(self):
        s = Stats()
        self.L.oracle_get_stats(self.h, C.byref(s))
        return s.as_dict()

    @property
    def cursor(self):
        return int(self.L.oracle_get_cursor(self.h))

    def witness_blocks(self, blocks, pre_states, cursor0=0, want_streams=True):
        """blocks: (n,64) u8; pre_states: (n,8) u32.  Returns dict with
        gate (n*G,4) u64, dense/spread (ncols, rows, 4) u64, next_states (n,8) u32,
        row_base (absolute chip row of buffer row 0)."""
        blocks = np.ascontiguousarray(blocks, dtype=np.uint8).reshape(-1, 64)
        pre_states = np.ascontiguousarray(pre_states, dtype=np.uint32).reshape(-1, 8)
        n = blocks.shape[0]
        assert pre_states.shape[0] == n
        G, LC = measure_shape(self.bits, self.ncols, self.internals)
        LK = lookup_cells_per_block(self.bits, self.ncols)
        row_base = cursor0 // self.ncols
        rows = (cursor0 % self.ncols + LC * n + self.ncols - 1) // self.ncols
        nxt = np.zeros((n, 8), dtype=np.uint32)
        self.L.oracle_set_cursor(self.h, cursor0)
        lookup = None
        if want_streams:
            lookup = np.zeros((n * LK, 4), dtype=np.uint64)
            self.L.oracle_set_lookup_output(self.h, lookup.ctypes.data, lookup.shape[0])
            gate = np.zeros((n * G, 4), dtype=np.uint64)
            dense = np.zeros((self.ncols, max(rows, 1), 4), dtype=np.uint64)
            spread = np.zeros((self.ncols, max(rows, 1), 4), dtype=np.uint64)
            self.L.oracle_set_outputs(self.h, gate.ctypes.data, gate.shape[0], dense.ctypes.data,
                                      spread.ctypes.data, dense.shape[1], row_base)
        else:
            gate = dense = spread = None
            self.L.oracle_set_lookup_output(self.h, None, 0)
            self.L.oracle_set_outputs(self.h, None, 0, None, None, 0, 0)
        self.L.oracle_witness_blocks(self.h, blocks.ctypes.data, pre_states.ctypes.data, n, nxt.ctypes.data)
        self.L.oracle_set_lookup_output(self.h, None, 0)
        self._check()
        return dict(gate=gate, dense=dense, spread=spread, lookup=lookup, next_states=nxt, row_base=row_base,
                    rows=rows, gate_cells_per_block=G, limb_calls_per_block=LC, lookup_cells_per_block=LK)

    def digest(self, message: bytes, max_variable_byte_size: int, precomputed_input_len: int = 0,
               want_streams=False, cursor0=None):
        """lib.rs:71-349 on values.  Returns dict(digest, blocks, pre_states, next_states[, streams])."""
        nblk = max_variable_byte_size // 64
        G, LC = measure_shape(self.bits, self.ncols, self.internals)
        msg = np.frombuffer(bytes(message), dtype=np.uint8).copy()
        dig = np.zeros(32, dtype=np.uint8)
        blocks = np.zeros((max(nblk, 1), 64), dtype=np.uint8)
        pre = np.zeros((max(nblk, 1), 8), dtype=np.uint32)
        nxt = np.zeros((max(nblk, 1), 8), dtype=np.uint32)
        if cursor0 is not None:
            self.L.oracle_set_cursor(self.h, cursor0)
        c0 = self.cursor
        out = {}
        if want_streams:
            row_base = c0 // self.ncols
            rows = (c0 % self.ncols + LC * nblk + self.ncols - 1) // self.ncols
            gate = np.zeros((nblk * G, 4), dtype=np.uint64)
            dense = np.zeros((self.ncols, max(rows, 1), 4), dtype=np.uint64)
            spread = np.zeros((self.ncols, max(rows, 1), 4), dtype=np.uint64)
            self.L.oracle_set_outputs(self.h, gate.ctypes.data, gate.shape[0], dense.ctypes.data,
                                      spread.ctypes.data, dense.shape[1], row_base)
            out.update(gate=gate, dense=dense, spread=spread, row_base=row_base, rows=rows)
        else:
            self.L.oracle_set_outputs(self.h, None, 0, None, None, 0, 0)
        rc = self.L.oracle_digest(self.h, msg.ctypes.data if len(msg) else None, len(msg),
                                  precomputed_input_len, max_variable_byte_size, dig.ctypes.data,
                                  blocks.ctypes.data, pre.ctypes.data, nxt.ctypes.data)
        if rc >= 10:
            raise ValueError("reference assert would fire (oracle_digest rc=%d)" % rc)
        self._check()
        out.update(digest=dig.tobytes(), blocks=blocks[:nblk], pre_states=pre[:nblk],
                   next_states=nxt[:nblk])
        return out


def digest_cells(messages, max_sizes, precomputed=None, is_input_range_check=False, record=False,
                 num_bits_lookup=8, num_advice_columns=2, zero_cell_loaded=False):
    """len(messages) consecutive Sha256DynamicConfig::digest calls in ONE Context -- fresh, or
    (zero_cell_loaded) one that already caches its zero cell, as after other halo2-base calls --
    (the reference's TestCircuit makes two, lib.rs:455-466) with every cell digest()
    allocates, under assumptions A1-A4 (hsw_oracle.h: oracle_digest_cells).
    Returns dict: gate (N,4) u64, lookup (M,4) u64, dense/spread (ncols, rows, 4), digests,
    layouts (one dict per digest, plus gate0/lookup0 = where its sections start),
    call_lens (u8, the assign_region tape), gate_rows (stream index of every gate row) and,
    with record=True, cs = the constraint structure on absolute cell ids."""
    precomputed = precomputed or [0] * len(messages)
    G, LC = measure_shape(num_bits_lookup, num_advice_columns, True)
    LK = lookup_cells_per_block(num_bits_lookup, num_advice_columns)
    nblk = [m // 64 for m in max_sizes]
    rc = 4 if is_input_range_check else 0
    gcap = sum(64 + (1 + rc) * m + b * G + 76 * (b + 1) + 288 for m, b in zip(max_sizes, nblk))
    lcap = sum(8 + 2 * m + b * LK + 64 for m, b in zip(max_sizes, nblk))
    tot_blk = sum(nblk)
    rows = (LC * tot_blk + num_advice_columns - 1) // num_advice_columns
    o = Oracle(num_bits_lookup, num_advice_columns, check=True, internals=True)
    gate = np.zeros((gcap, 4), dtype=np.uint64)
    lookup = np.zeros((lcap, 4), dtype=np.uint64)
    dense = np.zeros((num_advice_columns, max(rows, 1), 4), dtype=np.uint64)
    spread = np.zeros((num_advice_columns, max(rows, 1), 4), dtype=np.uint64)
    call_lens = np.zeros(gcap, dtype=np.uint8)
    gate_rows = np.zeros(gcap, dtype=np.uint64)
    L = o.L
    L.oracle_set_outputs(o.h, gate.ctypes.data, gcap, dense.ctypes.data, spread.ctypes.data, dense.shape[1], 0)
    L.oracle_set_lookup_output(o.h, lookup.ctypes.data, lcap)
    L.oracle_set_tape(o.h, call_lens.ctypes.data, gcap, gate_rows.ctypes.data, gcap)
    if zero_cell_loaded:
        L.oracle_set_context(o.h, 1)
    cs = None
    if record:
        cap = gcap
        eq = np.zeros((cap, 2), dtype=np.int64)
        konst = np.zeros((cap, 2), dtype=np.int64)
        rng_ = np.zeros((cap, 2), dtype=np.int64)
        chip = np.zeros((LC * tot_blk, 2), dtype=np.int64)
        lk = np.zeros(lcap, dtype=np.int64)
        nsc = np.zeros(8, dtype=np.int64)
        cst = _Constraints(eq.ctypes.data, cap, 0, konst.ctypes.data, cap, 0, rng_.ctypes.data, cap, 0,
                           chip.ctypes.data, LC * tot_blk, 0, lk.ctypes.data, lcap, 0, nsc.ctypes.data)
        L.oracle_record_constraints(o.h, C.byref(cst))
    digests, layouts = [], []
    for msg, mx, pre in zip(messages, max_sizes, precomputed):
        m = np.frombuffer(bytes(msg), dtype=np.uint8).copy()
        dig = np.zeros(32, dtype=np.uint8)
        lay = DigestLayout()
        g0, l0 = int(L.oracle_gate_len(o.h)), int(L.oracle_lookup_len(o.h))
        r = L.oracle_digest_cells(o.h, m.ctypes.data if len(m) else None, len(m), pre, mx,
                                  1 if is_input_range_check else 0, dig.ctypes.data, C.byref(lay))
        if r >= 10:
            raise ValueError("reference assert would fire (oracle_digest_cells rc=%d)" % r)
        o._check()
        d = lay.as_dict()
        d.update(gate0=g0, lookup0=l0)
        layouts.append(d)
        digests.append(dig.tobytes())
    n_gate, n_lk = int(L.oracle_gate_len(o.h)), int(L.oracle_lookup_len(o.h))
    n_calls, n_rows = int(L.oracle_tape_calls(o.h)), int(L.oracle_tape_rows(o.h))
    assert n_gate <= gcap and n_lk <= lcap and n_calls <= gcap and n_rows <= gcap
    if record:
        L.oracle_record_constraints(o.h, None)
        assert cst.n_eq <= cap and cst.n_const <= cap and cst.n_range <= cap
        cs = dict(eq=eq[:cst.n_eq].copy(), const=konst[:cst.n_const].copy(), range=rng_[:cst.n_range].copy(),
                  chip=chip[:cst.n_chip].copy(), lookup_src=lk[:cst.n_lookup].copy())
    L.oracle_set_tape(o.h, None, 0, None, 0)
    L.oracle_set_lookup_output(o.h, None, 0)
    L.oracle_set_outputs(o.h, None, 0, None, None, 0, 0)
    return dict(gate=gate[:n_gate].copy(), lookup=lookup[:n_lk].copy(), dense=dense, spread=spread,
                digests=digests, layouts=layouts, call_lens=call_lens[:n_calls].copy(),
                gate_rows=gate_rows[:n_rows].astype(np.int64), cs=cs, G=G, LC=LC, LK=LK)


CELL_ZERO = -1000
CELL_INPUT_BYTE0 = -1
CELL_PRE_STATE0 = -100
CELL_HIDDEN = -2000


class _Constraints(C.Structure):
    _fields_ = [("eq", C.c_void_p), ("eq_cap", C.c_size_t), ("n_eq", C.c_size_t),
                ("konst", C.c_void_p), ("const_cap", C.c_size_t), ("n_const", C.c_size_t),
                ("range", C.c_void_p), ("range_cap", C.c_size_t), ("n_range", C.c_size_t),
                ("chip", C.c_void_p), ("chip_cap", C.c_size_t), ("n_chip", C.c_size_t),
                ("lookup_src", C.c_void_p), ("lookup_cap", C.c_size_t), ("n_lookup", C.c_size_t),
                ("next_state_cells", C.c_void_p)]


_cs_cache = {}


def constraint_system(num_bits_lookup=8, num_advice_columns=2, internals=False):
    """The constraint STRUCTURE of one block as the reference's source specifies it
    (hsw_oracle.h oracle_constraints_t): dict of int64 numpy arrays
      eq (n,2), const (n,2), range (n,2), chip (limb calls,2), lookup_src (n,), next_state_cells (8,)
    plus gate_starts (first cell of every 4-cell gate row).  Input independent."""
    key = (num_bits_lookup, num_advice_columns, bool(internals))
    if key in _cs_cache:
        return _cs_cache[key]
    G, LC = measure_shape(num_bits_lookup, num_advice_columns, internals)
    LK = lookup_cells_per_block(num_bits_lookup, num_advice_columns)
    o = Oracle(num_bits_lookup, num_advice_columns, check=True, internals=internals)
    cap = 4 * G
    eq = np.zeros((cap, 2), dtype=np.int64)
    konst = np.zeros((cap, 2), dtype=np.int64)
    rng_ = np.zeros((cap, 2), dtype=np.int64)
    chip = np.zeros((LC, 2), dtype=np.int64)
    lk = np.zeros(LK, dtype=np.int64)
    nsc = np.zeros(8, dtype=np.int64)
    cs = _Constraints(eq.ctypes.data, cap, 0, konst.ctypes.data, cap, 0, rng_.ctypes.data, cap, 0,
                      chip.ctypes.data, LC, 0, lk.ctypes.data, LK, 0, nsc.ctypes.data)
    kinds = np.zeros(G, dtype=np.uint8)
    o.L.oracle_set_outputs(o.h, None, 0, None, None, 0, 0)
    o.L.oracle_set_kinds(o.h, kinds.ctypes.data, G)
    o.L.oracle_record_constraints(o.h, C.byref(cs))
    blk = (np.arange(64, dtype=np.uint8) * 13 + 5).astype(np.uint8)
    nxt = np.zeros(8, dtype=np.uint32)
    o.L.oracle_sha256_compression(o.h, blk.ctypes.data, INIT_STATE.ctypes.data, nxt.ctypes.data)
    o.L.oracle_set_kinds(o.h, None, 0)
    o._check()
    assert cs.n_eq <= cap and cs.n_const <= cap and cs.n_range <= cap and cs.n_chip == LC and cs.n_lookup == LK
    res = dict(eq=eq[:cs.n_eq].copy(), const=konst[:cs.n_const].copy(), range=rng_[:cs.n_range].copy(),
               chip=chip.copy(), lookup_src=lk.copy(), next_state_cells=nsc.copy(),
               gate_starts=np.nonzero(kinds == 1)[0].astype(np.int64), kinds=kinds, G=G, LC=LC, LK=LK)
    _cs_cache[key] = res
    return res


def gate_tape(num_bits_lookup=8, num_advice_columns=2, internals=False):
    """Per-cell tags of ONE block's gate stream (input independent): uint8 array
    of length G, 0 = load_witness cell, 1..4 = position in a gate row whose
    constraint is x0 + x1*x2 == x3."""
    G, _ = measure_shape(num_bits_lookup, num_advice_columns, internals)
    o = Oracle(num_bits_lookup, num_advice_columns, check=True, internals=internals)
    kinds = np.zeros(G, dtype=np.uint8)
    o.L.oracle_set_outputs(o.h, None, 0, None, None, 0, 0)
    o.L.oracle_set_kinds(o.h, kinds.ctypes.data, G)
    blk = np.arange(64, dtype=np.uint8)
    nxt = np.zeros(8, dtype=np.uint32)
    o.L.oracle_sha256_compression(o.h, blk.ctypes.data, INIT_STATE.ctypes.data, nxt.ctypes.data)
    o.L.oracle_set_kinds(o.h, None, 0)
    o._check()
    return kinds


def to_montgomery(cells):
    """Copy of a (..., 4) uint64 cell array converted to Montgomery form."""
    a = np.ascontiguousarray(cells, dtype=np.uint64).copy()
    lib().oracle_to_montgomery(a.ctypes.data, a.size // 4)
    return a


def plain_compress(state, block):
    st = np.ascontiguousarray(state, dtype=np.uint32).copy()
    blk = np.ascontiguousarray(block, dtype=np.uint8)
    lib().oracle_plain_compress(st.ctypes.data, blk.ctypes.data)
    return st


def spread_table(num_bits_lookup=8):
    """SpreadConfig::load (spread.rs:165-194): rows (i, spread(i))."""
    L = lib()
    return [(i, int(L.oracle_spread_table_entry(i))) for i in range(1 << num_bits_lookup)]
