"""Python view of the C++ gadget front-end (csrc/hsw_gadget.hpp): the same
names as the reference's `Sha256DynamicConfig` (src/lib.rs:38-369).  All logic
lives in libhsw.so; this file only marshals arguments."""
import ctypes as C

from . import _native as N


class AssignedHashResult:
    """lib.rs:31-36 on values."""

    def __init__(self, r, input_bytes):
        self.input_len = int(r.input_len)
        self._input_bytes = input_bytes          # bytes, or a callable fetching them on first use
        self.output_bytes = bytes(r.output_bytes)
        self.first_block = int(r.first_block)
        self.n_blocks = int(r.n_blocks)
        self.spread_cursor0 = int(r.spread_cursor0)
        self.num_round = int(r.num_round)
        self.target_round = int(r.target_round)
        # whole-digest contexts: where the sections of this digest start (cells), else 0
        for k in ("prologue_cell", "block_cell", "epilogue_cell", "end_cell",
                  "prologue_lookup", "block_lookup", "epilogue_lookup"):
            setattr(self, k, int(getattr(r, k)))


    @property
    def input_bytes(self):
        """assigned_input_bytes (lib.rs:170-173): the padded variable part, max_variable_byte_size bytes."""
        if callable(self._input_bytes):
            self._input_bytes = self._input_bytes()
        return self._input_bytes


class Sha256DynamicConfig:
    """configure (lib.rs:49-69) + new_context (lib.rs:351-360) in one object."""

    def __init__(self, engine, max_variable_byte_sizes, is_input_range_check=True, whole_digest=False, independent=False):
        """whole_digest: also emit the cells digest() itself allocates (lib.rs:122-178, 294-341;
        SURVEY 8 f4, assumption A4) -- needs an engine in HSW_MODE_HALO2_INTERNALS.
        independent: every digest is a synthesis of its own (HSW_GADGET_INDEPENDENT: K proofs in one launch)."""
        self.engine = engine
        self.whole_digest = bool(whole_digest)
        self.lib = engine.lib
        self.max_variable_byte_sizes = list(max_variable_byte_sizes)
        arr = (C.c_size_t * max(len(self.max_variable_byte_sizes), 1))(*self.max_variable_byte_sizes)
        h = C.c_void_p()
        rc = self.lib.hsw_gadget_create_ex(engine.h, arr, len(self.max_variable_byte_sizes),
                                           1 if is_input_range_check else 0,
                                           (N.HSW_GADGET_WHOLE_DIGEST if whole_digest else 0) |
                                           (N.HSW_GADGET_INDEPENDENT if independent else 0), C.byref(h))
        if rc != N.HSW_OK:
            raise N.HswError(rc, self.lib.hsw_last_error(engine.h).decode())
        self.h = h
        self._n = 0
        self._pending = []      # results whose input_bytes have not been fetched yet

    def _resolve_pending(self):
        for r in self._pending:
            r.input_bytes               # fetch while the gadget still holds them
        self._pending = []

    def close(self):
        if getattr(self, "h", None):
            self._resolve_pending()
            self.lib.hsw_gadget_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _ok(self, rc):
        if rc != N.HSW_OK:
            raise N.HswError(rc, self.lib.hsw_last_error(self.engine.h).decode())

    def _input_bytes(self, idx):
        n = C.c_size_t()
        self._ok(self.lib.hsw_gadget_input_bytes(self.h, idx, None, 0, C.byref(n)))
        buf = (C.c_uint8 * max(n.value, 1))()
        self._ok(self.lib.hsw_gadget_input_bytes(self.h, idx, buf, n.value, None))
        return bytes(buf[: n.value])

    def digest(self, message: bytes, precomputed_input_len=None):
        """lib.rs:71-349; precomputed_input_len None == the reference's Option::None."""
        return self.digest_batch([message], [precomputed_input_len])[0]

    def digest_batch(self, messages, precomputed_input_lens=None):
        n = len(messages)
        keep = [bytes(m) for m in messages]
        bufs = [(C.c_uint8 * max(len(m), 1)).from_buffer_copy(m if m else b"\0") for m in keep]
        ptrs = (C.c_void_p * n)(*[C.addressof(b) for b in bufs])
        lens = (C.c_size_t * n)(*[len(m) for m in keep])
        pl = precomputed_input_lens or [None] * n
        pre = (C.c_size_t * n)(*[int(p or 0) for p in pl])
        res = (N.HashResult * n)()
        self._ok(self.lib.hsw_gadget_digest_batch(self.h, n, ptrs, lens, pre, res))
        base = self._n
        out = [AssignedHashResult(res[i], (lambda k=base + i: self._input_bytes(k))) for i in range(n)]
        self._n += n
        self._pending.extend(out)
        return out

    def set_columns(self, max_rows):
        """Lay the whole-digest stream out as FlexGate advice columns of max_rows rows (before the
        first digest).  Returns the number of columns."""
        n = C.c_uint64()
        self._ok(self.lib.hsw_gadget_set_columns(self.h, max_rows, C.byref(n)))
        return int(n.value)

    def set_origin(self, column=0, row=0, zero_cell_loaded=False, lookups_queued=0):
        """Where the caller's halo2-base Context stands when the gadget takes over (hsw_gadget_set_origin):
        ctx.advice_alloc[0] = (column, row), ctx.zero_cell.is_some(), ctx.cells_to_lookup.len()."""
        self._ok(self.lib.hsw_gadget_set_origin(self.h, column, row, 1 if zero_cell_loaded else 0, lookups_queued))

    def reset(self):
        """Next synthesis pass: all cursors back to their start, buffers and layout kept
        (the reference clones the config per synthesis, lib.rs:440)."""
        self._resolve_pending()
        self._ok(self.lib.hsw_gadget_reset(self.h))
        self._n = 0

    def download_region(self, pinned=True):
        """hsw_gadget_download_region into (pinned) host arrays: dict of numpy uint64 arrays shaped like
        streams() -- gate (columns, max_rows, 4) or (cells, 4); lookup; dense / spread (ncols, stride, 4)."""
        import numpy as np
        v = self.view()
        ncols = self.engine.ncols
        img = self.whole_digest and int(v.max_rows)
        n_gate = int(v.max_rows) * int(v.columns) if img else (
            int(v.gate_cells) if self.whole_digest else int(v.blocks_done) * self.engine.G)
        stride = int(v.chip_col_stride)

        def buf(cells):
            if pinned:
                return self.engine.host_empty((max(cells, 1), 4))
            return np.zeros((max(cells, 1), 4), dtype=np.uint64)
        gate, dense, spread = buf(n_gate), buf(ncols * stride), buf(ncols * stride)
        lookup = buf(int(v.lookup_cells)) if self.whole_digest else None
        if img:
            gate[:] = 0                                    # unassigned tail rows of a column stay zero
        dst = N.RegionHost(gate.ctypes.data, lookup.ctypes.data if lookup is not None else None,
                           dense.ctypes.data, spread.ctypes.data)
        self._ok(self.lib.hsw_gadget_download_region(self.h, C.byref(dst)))
        rows = (int(v.num_limb_sum) + ncols - 1) // ncols
        out = dict(gate=gate[:n_gate].reshape(int(v.columns), int(v.max_rows), 4) if img else gate[:n_gate],
                   dense=dense.reshape(ncols, stride, 4)[:, :rows], spread=spread.reshape(ncols, stride, 4)[:, :rows],
                   rows=rows)
        if lookup is not None:
            out["lookup"] = lookup[: int(v.lookup_cells)]
        return out

    def place(self, candidates=3):
        """hsw_gadget_place: try `candidates` allocations of the chip columns, keep the one the gadget's own batch
        runs fastest on (fresh or reset gadget).  Returns (batch ms of every candidate, index kept)."""
        ms = (C.c_float * candidates)()
        kept = C.c_uint()
        self._ok(self.lib.hsw_gadget_place(self.h, candidates, ms, C.byref(kept)))
        return [float(x) for x in ms], int(kept.value)

    def download_region_distinct(self, threads=8, bufs=None):
        """Distinct-value delivery: only the new witnesses cross PCIe (hsw_gadget_download_region_distinct into
        pinned memory), the image is rebuilt on the host (hsw_gadget_replay_region).  Returns the same dict as
        download_region plus "distinct" (n, 4) and "bufs" (pass back in to reuse the host buffers)."""
        import numpy as np
        tape = N.RegionTape()
        self._ok(self.lib.hsw_gadget_region_tape(self.h, C.byref(tape)))
        v = self.view()
        ncols = self.engine.ncols
        img = int(v.max_rows)
        n_gate = int(v.max_rows) * int(v.columns) if img else int(v.gate_cells)
        stride = int(v.chip_col_stride)
        if bufs is None:          # sized for the whole gadget, so that they can be reused as more digests are assigned
            bufs = dict(distinct=self.engine.host_empty((max(int(tape.distinct_capacity), 1), 4)),
                        gate=np.zeros((max(n_gate if img else int(v.gate_capacity), 1), 4), dtype=np.uint64), lookup=np.zeros((max(int(v.lookup_capacity), 1), 4), dtype=np.uint64),
                        dense=np.zeros((ncols * stride, 4), dtype=np.uint64), spread=np.zeros((ncols * stride, 4), dtype=np.uint64))
        n = C.c_size_t()
        self._ok(self.lib.hsw_gadget_download_region_distinct(self.h, bufs["distinct"].ctypes.data, bufs["distinct"].shape[0], C.byref(n)))
        dst = N.RegionHost(bufs["gate"].ctypes.data, bufs["lookup"].ctypes.data, bufs["dense"].ctypes.data, bufs["spread"].ctypes.data)
        self._ok(self.lib.hsw_gadget_replay_region(self.h, bufs["distinct"].ctypes.data, C.byref(dst), threads))
        rows = (int(v.num_limb_sum) + ncols - 1) // ncols
        return dict(gate=bufs["gate"][:n_gate].reshape(int(v.columns), int(v.max_rows), 4) if img else bufs["gate"][:n_gate],
                    lookup=bufs["lookup"][: int(v.lookup_cells)], dense=bufs["dense"].reshape(ncols, stride, 4)[:, :rows],
                    spread=bufs["spread"].reshape(ncols, stride, 4)[:, :rows], rows=rows, distinct=bufs["distinct"][: n.value],
                    n_distinct=int(n.value), bufs=bufs)

    def download_region_compact(self, bufs=None):
        """hsw_gadget_download_region_compact into pinned host arrays: 8-byte cells + the side list of the cells
        wider than 64 bits.  Returns (bufs, n_wide); pass `bufs` back in to reuse the buffers.  widen() rebuilds
        the 32-byte streams on the host (hsw_region_widen)."""
        import numpy as np
        v = self.view()
        ncols = self.engine.ncols
        img = self.whole_digest and int(v.max_rows)
        n_gate = int(v.max_rows) * int(v.columns) if img else (
            int(v.gate_cells) if self.whole_digest else int(v.blocks_done) * self.engine.G)
        stride = int(v.chip_col_stride)
        if bufs is None:
            cap = 300 * max(int(v.capacity_blocks), 1) + 4096
            bufs = dict(gate=self.engine.host_empty((max(n_gate, 1),)), dense=self.engine.host_empty((ncols * stride,)),
                        spread=self.engine.host_empty((ncols * stride,)), wide=self.engine.host_empty((cap, 6)), cap=cap)
            bufs["lookup"] = self.engine.host_empty((max(int(v.lookup_cells), 1),)) if self.whole_digest else None
            if img:
                bufs["gate"][:] = 0
        dst = N.RegionCompact(bufs["gate"].ctypes.data, bufs["lookup"].ctypes.data if bufs["lookup"] is not None else None,
                              bufs["dense"].ctypes.data, bufs["spread"].ctypes.data, bufs["wide"].ctypes.data, bufs["cap"], 0)
        self._ok(self.lib.hsw_gadget_download_region_compact(self.h, C.byref(dst)))
        return bufs, int(dst.n_wide)

    def widen(self, compact, stream_id, wide, n_wide):
        """hsw_region_widen: one stream's 32-byte canonical cells from its compact form + the side list."""
        import numpy as np
        compact = np.ascontiguousarray(compact, dtype=np.uint64).reshape(-1)
        out = np.zeros((compact.size, 4), dtype=np.uint64)
        self._ok(self.lib.hsw_region_widen(compact.ctypes.data, compact.size, stream_id, wide.ctypes.data, n_wide,
                                           out.ctypes.data))
        return out

    def verify(self):
        """hsw_gadget_verify: everything written so far against the constraint system, on the device."""
        rep = N.VerifyReport()
        self._ok(self.lib.hsw_gadget_verify(self.h, C.byref(rep)))
        return dict(violations=int(rep.violations), checks=int(rep.checks), first_block=int(rep.first_block),
                    first_cell=int(rep.first_cell), first_class=N.VerifyReport.CLASSES.get(int(rep.first_class)),
                    kernel_ms=float(rep.kernel_ms))

    def seek(self, hash_idx):
        """Continue at digest #hash_idx as if the earlier ones had been assigned (their positions
        follow from max_variable_byte_sizes alone): lets several GPUs share one circuit's digests."""
        self._resolve_pending()
        self._ok(self.lib.hsw_gadget_seek(self.h, hash_idx))
        self._n = hash_idx

    def cell_position(self, cell):
        c, r = C.c_uint64(), C.c_uint64()
        self._ok(self.lib.hsw_gadget_cell_position(self.h, cell, C.byref(c), C.byref(r)))
        return int(c.value), int(r.value)

    def set_repr(self, repr_flag):
        self._ok(self.lib.hsw_gadget_set_repr(self.h, repr_flag))

    def view(self):
        v = N.GadgetView()
        self._ok(self.lib.hsw_gadget_streams(self.h, C.byref(v)))
        return v

    def streams(self):
        """Copies of the streams written so far, as numpy uint64 arrays."""
        import numpy as np
        v = self.view()
        G = self.engine.G
        ncols = self.engine.ncols

        def grab(ptr, n_cells):
            a = np.zeros((n_cells, 4), dtype=np.uint64)
            self._ok(self.lib.hsw_download(self.engine.h, a.ctypes.data, ptr, n_cells * 32))
            return a

        rows = (int(v.num_limb_sum) + ncols - 1) // ncols
        if self.whole_digest and int(v.max_rows):
            gate = grab(v.d_gate, int(v.max_rows) * int(v.columns)).reshape(int(v.columns), int(v.max_rows), 4)
        else:
            gate = grab(v.d_gate, int(v.gate_cells) if self.whole_digest else int(v.blocks_done) * G)
        dense = np.stack([grab(v.d_chip_dense + c * int(v.chip_col_stride) * 32, rows) for c in range(ncols)])
        spread = np.stack([grab(v.d_chip_spread + c * int(v.chip_col_stride) * 32, rows) for c in range(ncols)])
        out = dict(gate=gate, dense=dense, spread=spread, rows=rows)
        if self.whole_digest:
            out["lookup"] = grab(v.d_lookup, int(v.lookup_cells))
        return out
