"""Device-memory plumbing over the C ABI (include/hsw.h).

PyTorch is used here only for HBM allocations, streams and (in bench.py)
torch.distributed; all computation happens in libhsw.so's gfx950 kernels.
Tensors that carry u32 words are int32, tensors that carry field cells are
int64 with a trailing dimension of 4 (little-endian limbs): same bits, and
`.view(np.uint64)` on the host side recovers the unsigned view.
"""
import ctypes as C

from . import _native as N


class _Pinned:
    """Owner of one hsw_host_alloc allocation."""

    def __init__(self, lib, ptr):
        self.lib, self.ptr = lib, ptr

    def __del__(self):
        try:
            self.lib.hsw_host_free(self.ptr)
        except Exception:
            pass


class _DeviceRange:
    """One hsw_device_alloc range, exposed to torch through __cuda_array_interface__ (zero copy).

    Ranges are NOT returned while the process runs: on ROCm 7.2 a probe that unmapped and re-created ranges between
    launches (tools/vmmprobe) ended in GPU memory faults in three of three processes after a handful of cycles,
    while allocating and using them never did.  They go back with the process (free() for callers that know the
    device is idle and stays so)."""
    _live = []

    def __init__(self, lib, ptr, shape, typestr):
        self.lib, self.ptr = lib, ptr
        self.__cuda_array_interface__ = dict(shape=tuple(shape), typestr=typestr, data=(ptr, False), version=2, strides=None)
        _DeviceRange._live.append(self)

    def free(self):
        if self.ptr:
            self.lib.hsw_device_free(self.ptr)
            self.ptr = None
            _DeviceRange._live.remove(self)


class WitnessEngine:
    """One hsw_engine bound to (device, stream).

    Mirrors how the reference is driven: SpreadConfig::configure's two shape
    parameters (spread.rs:32-36) are fixed at construction; the mutable
    num_limb_sum cursor (spread.rs:26) is an explicit argument of every call.
    """

    def __init__(self, device=0, num_bits_lookup=8, num_advice_columns=2, stream=None,
                 mode=N.HSW_MODE_DEFAULT):
        import torch  # plumbing only
        self.torch = torch
        self.lib = N.lib()
        if not torch.cuda.is_available():
            raise N.HswError(N.HSW_ERR_NO_DEVICE, "torch sees no HIP device; the witness engine has no CPU path")
        self.device = torch.device("cuda", device)
        if stream is None:
            stream = torch.cuda.current_stream(self.device)
        self.stream = stream
        h = C.c_void_p()
        rc = self.lib.hsw_engine_create_ex(device, C.c_void_p(stream.cuda_stream), num_bits_lookup,
                                           num_advice_columns, mode, C.byref(h))
        if rc != N.HSW_OK:
            raise N.HswError(rc)
        self.h = h
        s = N.Shape()
        self._ok(self.lib.hsw_engine_shape(self.h, C.byref(s)))
        self.shape = s
        self.G = int(s.gate_cells_per_block)
        self.ncols = int(s.num_advice_columns)
        self.limb_calls = int(s.limb_calls_per_block)
        self.lookup_cells = int(s.lookup_cells_per_block)
        self.mode = mode

    def close(self):
        if getattr(self, "h", None):
            self.lib.hsw_engine_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _ok(self, rc):
        if rc != N.HSW_OK:
            raise N.HswError(rc, self.lib.hsw_last_error(self.h).decode())

    # ---- sizes -----------------------------------------------------------
    def chip_rows(self, cursor0, n_blocks):
        return int(self.lib.hsw_chip_rows(C.byref(self.shape), cursor0, n_blocks))

    def device_empty(self, shape, chunk_bytes=0):
        """An int64 tensor in an hsw_device_alloc range (one virtual range backed by physical allocations of up to
        chunk_bytes, 0 = 4 GiB): where the witness launch's gate stream ran fastest (tools/vmmprobe)."""
        n = 1
        for d in shape:
            n *= int(d)
        p = C.c_void_p()
        self._ok(self.lib.hsw_device_alloc(self.device.index or 0, max(n, 1) * 8, chunk_bytes, C.byref(p)))
        return self.torch.as_tensor(_DeviceRange(self.lib, p.value, shape, "<i8"), device=self.device)

    def alloc_outputs(self, n_blocks, cursor0=0, flags=0, ranged=False):
        """Allocate the three output buffers for n_blocks blocks in HBM (cells are
        4 x int64, or 1 x int64 with HSW_REPR_COMPACT64).  ranged: the gate stream in an hsw_device_alloc range."""
        t = self.torch
        rows = self.chip_rows(cursor0, n_blocks)
        w = 1 if (flags & N.HSW_REPR_COMPACT64) else 4
        gate = None
        if ranged:
            try:
                gate = self.device_empty((n_blocks * self.G, w))
            except N.HswError:                  # no virtual memory management on this stack: a plain buffer
                gate = None
        if gate is None:
            gate = t.empty((n_blocks * self.G, w), dtype=t.int64, device=self.device)
        dense = t.zeros((self.ncols, max(rows, 1), w), dtype=t.int64, device=self.device)
        spread = t.zeros((self.ncols, max(rows, 1), w), dtype=t.int64, device=self.device)
        nxt = t.empty((n_blocks, 8), dtype=t.int32, device=self.device)
        return dict(gate=gate, dense=dense, spread=spread, next_states=nxt, rows=rows)

    def alloc_outputs_placed(self, blocks, pre_states, cursor0=0, flags=0, candidates=6, spacer_bytes=6 << 30, ranged=True,
                             gate_candidates=3):
        """alloc_outputs with a measured choice of WHERE the gate stream and the chip columns live.

        On MI355X the same launch takes between 1.58 and 1.78 ms per 4,096 blocks depending on the allocations its
        gate stream and its chip columns live in (profiles/r03_placement_probe.log: the pure fill of either buffer
        and the launch without chip columns do not care; two regions of ONE plain allocation are always a slow
        case; a gate stream in an hsw_device_alloc range of 4 GiB physical pieces was the fastest in most processes,
        yet not in all).  Nothing in user space says which combinations are fast, so a caller that allocates its
        witness buffers once and reuses them -- a prover does -- times the launch itself on a few candidates and
        keeps the best: `gate_candidates` gate buffers (hsw_device_alloc ranges and plain ones alternating when
        `ranged`) x `candidates` (dense, spread) pairs (plain ones and ranges of smaller pieces alternating, spaced
        out by throw-away allocations of `spacer_bytes`).  Returns (outputs, report)."""
        t = self.torch
        n = blocks.numel() // 64
        timing_was = getattr(self, "_timing", False)
        self.set_timing(True)
        gates, kinds = [], []
        for g in range(max(gate_candidates, 1)):
            try:
                o = self.alloc_outputs(n, cursor0, flags, ranged=ranged and g % 2 == 0)
            except (RuntimeError, N.HswError):
                break
            gates.append(o)
            kinds.append("range of 4 GiB pieces" if ranged and g % 2 == 0 else "plain")
        out = gates[0]
        pairs, spacers, pair_kinds = [(out["dense"], out["spread"])], [], ["plain"]
        for k in range(1, max(candidates, 1)):
            try:
                spacers.append(t.empty((spacer_bytes,), dtype=t.uint8, device=self.device))
                if ranged and k % 2 == 1:     # every other candidate in hsw_device_alloc ranges of 32 MiB / 256 MiB / 2 GiB pieces
                    chunk = (32 << 20) << (3 * ((k // 2) % 3))
                    pairs.append((self.device_empty(out["dense"].shape, chunk).zero_(), self.device_empty(out["spread"].shape, chunk).zero_()))
                    pair_kinds.append("ranges of %d MiB pieces" % (chunk >> 20))
                else:
                    pairs.append((t.zeros_like(out["dense"]), t.zeros_like(out["spread"])))
                    pair_kinds.append("plain")
            except (RuntimeError, N.HswError):          # out of memory: keep what we have
                break

        def time_on(gate_out, dense, spread):
            o = dict(gate_out, dense=dense, spread=spread)
            ms = []
            for _ in range(4):
                self.witness_blocks(blocks, pre_states, cursor0=cursor0, out=o, flags=flags)
                ms.append(self.last_kernel_ms())
            return float(min(ms[1:]))

        table = [[time_on(go, d, s_) for d, s_ in pairs] for go in gates]
        bg, bp = min(((g, p) for g in range(len(gates)) for p in range(len(pairs))), key=lambda gp: table[gp[0]][gp[1]])
        out = dict(gates[bg], dense=pairs[bp][0], spread=pairs[bp][1])
        del pairs, spacers, gates
        self.set_timing(timing_was)
        return out, dict(gate_candidates=kinds, chip_candidates=pair_kinds, kernel_ms=table, kept=[bg, bp],
                         candidates=len(kinds) * len(pair_kinds), kernel_ms_each=[x for row in table for x in row])

    # ---- the hot path ----------------------------------------------------
    def witness_blocks(self, blocks, pre_states, cursor0=0, out=None, flags=0):
        """blocks: cuda uint8 (n,64); pre_states: cuda int32 (n,8).  Asynchronous
        on the engine's stream.  Returns the dict of output tensors."""
        t = self.torch
        assert blocks.is_cuda and pre_states.is_cuda and blocks.is_contiguous() and pre_states.is_contiguous()
        assert blocks.dtype == t.uint8 and pre_states.dtype == t.int32
        n = blocks.numel() // 64
        assert pre_states.numel() == (8 if flags & N.HSW_CHAINED else 8 * n)
        if out is None:
            out = self.alloc_outputs(n, cursor0, flags)
        gate, dense, spread, nxt = out["gate"], out["dense"], out["spread"], out["next_states"]
        rc = self.lib.hsw_witness_blocks(
            self.h, blocks.data_ptr(), pre_states.data_ptr(), n, cursor0,
            gate.data_ptr() if gate is not None else None,
            dense.data_ptr() if dense is not None else None,
            spread.data_ptr() if spread is not None else None,
            dense.shape[1] if dense is not None else 0,
            nxt.data_ptr() if nxt is not None else None, flags)
        self._ok(rc)
        return out

    def witness_blocks_ex(self, blocks, pre_states, cursor0=0, flags=0, want_lookup=False,
                          start_row=0, max_rows=None):
        """hsw_witness_blocks_ex: optional lookup-column stream (internals mode) and
        FlexGate column packing.  With max_rows the gate output is a flat buffer of
        plan.span_cells cells whose cell 0 is (first column, start_row)."""
        t = self.torch
        n = blocks.numel() // 64
        rows = self.chip_rows(cursor0, n)
        plan = None
        gate_cells = n * self.G
        if max_rows is not None:
            plan = N.pack_plan(self.shape, n, start_row, max_rows)
            gate_cells = int(plan.span_cells)
        gate = t.full((gate_cells, 4), -1, dtype=t.int64, device=self.device)
        dense = t.zeros((self.ncols, max(rows, 1), 4), dtype=t.int64, device=self.device)
        spread = t.zeros((self.ncols, max(rows, 1), 4), dtype=t.int64, device=self.device)
        nxt = t.empty((n, 8), dtype=t.int32, device=self.device)
        lookup = t.empty((n * self.lookup_cells, 4), dtype=t.int64, device=self.device) if want_lookup else None
        a = N.WitnessArgs()
        a.d_blocks, a.d_pre_states, a.n_blocks = blocks.data_ptr(), pre_states.data_ptr(), n
        a.spread_cursor0, a.d_gate = cursor0, gate.data_ptr()
        a.d_chip_dense, a.d_chip_spread, a.chip_col_stride = dense.data_ptr(), spread.data_ptr(), dense.shape[1]
        a.d_next_states = nxt.data_ptr()
        a.d_lookup = lookup.data_ptr() if want_lookup else None
        a.flags = flags
        a.pack = C.pointer(plan) if plan is not None else None
        self._ok(self.lib.hsw_witness_blocks_ex(self.h, C.byref(a)))
        return dict(gate=gate, dense=dense, spread=spread, next_states=nxt, lookup=lookup, plan=plan, rows=rows)

    def witness_blocks_host(self, blocks, pre_states, cursor0=0, flags=0, pinned=True):
        """Host delivery (hsw_witness_blocks_host): numpy in, numpy out.  With
        pinned=True the output arrays live in page-locked memory from
        hsw_host_alloc (freed when the returned dict's "_keep" is dropped)."""
        import numpy as np
        blocks = np.ascontiguousarray(blocks, dtype=np.uint8).reshape(-1, 64)
        pre_states = np.ascontiguousarray(pre_states, dtype=np.uint32).reshape(-1, 8)
        n = blocks.shape[0]
        rows = self.chip_rows(cursor0, n)
        w = 1 if (flags & N.HSW_REPR_COMPACT64) else 4
        shapes = dict(gate=(n * self.G, w), dense=(self.ncols, rows, w), spread=(self.ncols, rows, w))
        out, keep = {}, []
        for k, shp in shapes.items():
            nbytes = int(np.prod(shp)) * 8
            if pinned:
                p = C.c_void_p()
                self._ok(self.lib.hsw_host_alloc(nbytes, C.byref(p)))
                keep.append(_Pinned(self.lib, p))
                buf = (C.c_uint8 * nbytes).from_address(p.value)
                out[k] = np.frombuffer(buf, dtype=np.uint64).reshape(shp)
            else:
                out[k] = np.zeros(shp, dtype=np.uint64)
        nxt = np.zeros((n, 8), dtype=np.uint32)
        self._ok(self.lib.hsw_witness_blocks_host(
            self.h, blocks.ctypes.data, pre_states.ctypes.data, n, cursor0, out["gate"].ctypes.data,
            out["dense"].ctypes.data, out["spread"].ctypes.data, rows, nxt.ctypes.data, flags))
        out["next_states"] = nxt
        out["_keep"] = keep
        return out

    def verify_blocks(self, blocks, pre_states, out, cursor0=0, lookup=None, check_chip=True, check_next=True, flags=0):
        """hsw_verify_blocks: on-device check of the streams in `out` (as written by witness_blocks for
        these inputs) against the gadget's constraint system.  Returns the report as a dict."""
        a = N.WitnessArgs()
        n = blocks.numel() // 64
        a.d_blocks, a.d_pre_states, a.n_blocks = blocks.data_ptr(), pre_states.data_ptr(), n
        a.spread_cursor0, a.d_gate = cursor0, out["gate"].data_ptr()
        if check_chip:
            a.d_chip_dense, a.d_chip_spread = out["dense"].data_ptr(), out["spread"].data_ptr()
            a.chip_col_stride = out["dense"].shape[1]
        if check_next and out.get("next_states") is not None:
            a.d_next_states = out["next_states"].data_ptr()
        if lookup is not None:
            a.d_lookup = lookup.data_ptr()
        a.flags = flags                      # the representation the streams were written in (canonical / Montgomery)
        rep = N.VerifyReport()
        self._ok(self.lib.hsw_verify_blocks(self.h, C.byref(a), C.byref(rep)))
        return dict(violations=int(rep.violations), checks=int(rep.checks), first_block=int(rep.first_block),
                    first_cell=int(rep.first_cell), first_class=N.VerifyReport.CLASSES.get(int(rep.first_class)),
                    kernel_ms=float(rep.kernel_ms))

    def host_empty(self, shape):
        """numpy uint64 array in page-locked memory (hsw_host_alloc); the allocation lives as long as
        the array (it is kept alive through the array's base object)."""
        import numpy as np
        nbytes = int(np.prod(shape)) * 8
        p = C.c_void_p()
        self._ok(self.lib.hsw_host_alloc(max(nbytes, 8), C.byref(p)))
        owner = _Pinned(self.lib, p)
        buf = (C.c_uint8 * max(nbytes, 8)).from_address(p.value)
        buf._owner = owner                      # numpy keeps `buf` as the array's base
        return np.frombuffer(buf, dtype=np.uint64)[: nbytes // 8].reshape(shape)

    def sha256_chain(self, blocks, n_messages, blocks_per_message, init_states=None):
        """Plain SHA-256 chain pre-pass: pre-state of every block (lib.rs:188,236)."""
        t = self.torch
        pre = t.empty((n_messages * blocks_per_message, 8), dtype=t.int32, device=self.device)
        rc = self.lib.hsw_sha256_chain(self.h, blocks.data_ptr(), n_messages, blocks_per_message,
                                       init_states.data_ptr() if init_states is not None else None,
                                       pre.data_ptr())
        self._ok(rc)
        return pre

    def set_option(self, name, value):
        self._ok(self.lib.hsw_engine_set_option(self.h, name.encode(), int(value)))

    def fill_calibrate(self, tensor):
        """Plain streaming fill of `tensor`'s bytes; returns milliseconds."""
        ms = C.c_float()
        self._ok(self.lib.hsw_fill_calibrate(self.h, tensor.data_ptr(), tensor.numel() * tensor.element_size(),
                                             C.byref(ms)))
        return float(ms.value)

    def set_timing(self, on=True):
        self._ok(self.lib.hsw_set_timing(self.h, 1 if on else 0))
        self._timing = bool(on)

    def last_kernel_ms(self):
        ms = C.c_float()
        self._ok(self.lib.hsw_last_kernel_ms(self.h, C.byref(ms)))
        return float(ms.value)

    def last_launch(self):
        """hsw_last_launch: the expansion kernel instantiation and work split of the most recent launch."""
        li = N.LaunchInfo()
        self._ok(self.lib.hsw_last_launch(self.h, C.byref(li)))
        d = li.as_dict()
        d["kernel"] = li.kernel_name()
        return d

    def synchronize(self):
        self._ok(self.lib.hsw_engine_synchronize(self.h))
