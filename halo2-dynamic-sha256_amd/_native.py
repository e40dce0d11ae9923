"""ctypes binding of libhsw.so -- the C ABI declared in include/hsw.h.

The library is built in-tree by csrc/Makefile (hipcc, gfx950).  There is no
fallback: if it is missing the import of the engine fails loudly.
"""
import ctypes as C
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("HSW_LIB_OVERRIDE") or os.path.join(_HERE, "libhsw.so")   # override: A/B tuning only
CSRC = os.path.join(_HERE, "csrc")

HSW_OK = 0
HSW_ERR_INVALID_ARG = 1
HSW_ERR_SHAPE = 2
HSW_ERR_NO_DEVICE = 3
HSW_ERR_HIP = 4
HSW_ERR_UNSUPPORTED = 5
HSW_ERR_TOO_LARGE = 6
HSW_ERR_NOMEM = 7

HSW_REPR_CANONICAL = 0
HSW_REPR_MONTGOMERY = 1
HSW_SKIP_GATE = 2
HSW_SKIP_CHIP = 4
HSW_HOST_REGISTER = 8
HSW_REPR_COMPACT64 = 16
HSW_CHAINED = 32
HSW_MODE_DEFAULT = 0
HSW_MODE_HALO2_INTERNALS = 1
HSW_MAX_BREAKS = 16
HSW_CELL_BYTES = 32
HSW_GADGET_WHOLE_DIGEST = 1
HSW_GADGET_INDEPENDENT = 2
NO_CELL = (1 << 64) - 1


class Shape(C.Structure):
    _fields_ = [(n, C.c_uint32) for n in (
        "num_bits_lookup", "num_advice_columns", "limbs_per_spread", "cells_per_spread",
        "cells_per_state_spread", "cells_per_sigma", "cells_per_ch", "cells_per_maj",
        "cells_per_sched_step", "cells_per_round", "off_words", "off_msg_spread", "off_sched",
        "off_state_spread", "off_rounds", "off_feed", "gate_cells_per_block",
        "spread_calls_per_block", "limb_calls_per_block", "chip_cells_per_block")] + [
        ("algorithmic_bytes_per_block", C.c_uint64), ("mode", C.c_uint32),
        ("lookup_cells_per_block", C.c_uint32), ("gate_calls_per_block", C.c_uint32),
        ("reserved_", C.c_uint32)]

    def as_dict(self):
        return {n: int(getattr(self, n)) for n, _ in self._fields_}


class PackPlan(C.Structure):
    _fields_ = [("n_breaks", C.c_uint32), ("columns_touched", C.c_uint32),
                ("break_cell", C.c_uint64 * HSW_MAX_BREAKS), ("break_gap", C.c_uint64 * HSW_MAX_BREAKS),
                ("span_cells", C.c_uint64), ("end_row", C.c_uint64)]


class WitnessArgs(C.Structure):
    _fields_ = [("d_blocks", C.c_void_p), ("d_pre_states", C.c_void_p), ("n_blocks", C.c_size_t),
                ("spread_cursor0", C.c_uint64), ("d_gate", C.c_void_p), ("d_chip_dense", C.c_void_p),
                ("d_chip_spread", C.c_void_p), ("chip_col_stride", C.c_size_t), ("d_next_states", C.c_void_p),
                ("d_lookup", C.c_void_p), ("flags", C.c_uint32), ("pack", C.POINTER(PackPlan)),
                ("frame_every", C.c_uint64), ("frame_cells", C.c_uint64), ("frame_lookups", C.c_uint64)]


class LaunchInfo(C.Structure):
    _fields_ = [(n, C.c_uint32) for n in ("limbs", "tile_cells", "tile_rows", "repr", "internals", "parts", "split",
                                          "reserved_")] + [("n_blocks", C.c_uint64), ("grid", C.c_uint64)]

    def as_dict(self):
        return {n: int(getattr(self, n)) for n, _ in self._fields_ if n != "reserved_"}

    def kernel_name(self):
        if self.split == 2:
            return "hsw::hsw_small_kernel<%d, %d, %s>" % (self.limbs, self.repr, "true" if self.internals else "false")
        return "hsw::hsw_expand_kernel<%d, %d, %d, %d, %s>" % (self.limbs, self.tile_cells, self.tile_rows, self.repr,
                                                              "true" if self.internals else "false")


class WideCell(C.Structure):
    _fields_ = [("stream", C.c_uint64), ("index", C.c_uint64), ("value", C.c_uint64 * 4)]


class RegionCompact(C.Structure):
    _fields_ = [("gate", C.c_void_p), ("lookup", C.c_void_p), ("chip_dense", C.c_void_p), ("chip_spread", C.c_void_p),
                ("wide", C.c_void_p), ("wide_cap", C.c_size_t), ("n_wide", C.c_size_t)]


class ResultCells(C.Structure):
    _fields_ = [("input_len_cell", C.c_uint64), ("input_bytes_cell0", C.c_uint64), ("n_input_bytes", C.c_uint64),
                ("output_byte_cells", C.c_uint64 * 32), ("input_len_pos", C.c_uint64 * 2),
                ("input_bytes_pos0", C.c_uint64 * 2), ("output_byte_pos", (C.c_uint64 * 2) * 32)]


class DigestsArgs(C.Structure):
    """hsw_digests_args (descs: pointer to FrameDesc, declared below -> void pointer here)."""
    _fields_ = [("blocks", WitnessArgs), ("descs", C.c_void_p), ("n_digests", C.c_size_t), ("d_blocks0", C.c_void_p),
                ("d_pre_states0", C.c_void_p), ("d_next_states0", C.c_void_p), ("d_gate0", C.c_void_p),
                ("d_lookup0", C.c_void_p), ("frame_pack", C.POINTER(PackPlan)), ("host_next_states", C.c_void_p)]


class FrameShape(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in (
        "n_blocks", "prologue_cells", "epilogue_cells", "prologue_lookups", "epilogue_lookups",
        "prologue_calls", "epilogue_calls", "digest_cells", "digest_lookups")]

    def as_dict(self):
        return {n: int(getattr(self, n)) for n, _ in self._fields_}


class FrameDesc(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in (
        "input_len", "first_block", "prologue_cell", "epilogue_cell", "prologue_lookup",
        "epilogue_lookup", "zero_cell")] + [(n, C.c_uint32) for n in (
        "n_blocks", "num_round", "precomputed_round", "is_input_range_check")]


class DigestInfo(C.Structure):
    _fields_ = [("num_round", C.c_size_t), ("precomputed_round", C.c_size_t),
                ("target_round", C.c_size_t), ("n_blocks", C.c_size_t)]


class HashResult(C.Structure):
    _fields_ = [("input_len", C.c_uint64), ("first_block", C.c_size_t), ("n_blocks", C.c_size_t),
                ("spread_cursor0", C.c_uint64), ("num_round", C.c_size_t), ("target_round", C.c_size_t),
                ("output_bytes", C.c_uint8 * 32)] + [(n, C.c_uint64) for n in (
        "prologue_cell", "block_cell", "epilogue_cell", "end_cell",
        "prologue_lookup", "block_lookup", "epilogue_lookup")]


class StructureCounts(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in ("gate_cells", "gate_rows", "assert_eq", "ranges", "lookups", "limb_calls")]


class VerifyReport(C.Structure):
    _fields_ = [("violations", C.c_uint64), ("checks", C.c_uint64), ("first_block", C.c_uint64),
                ("first_cell", C.c_int64), ("first_class", C.c_uint32), ("kernel_ms", C.c_float)]
    CLASSES = {1: "constant", 2: "copy", 3: "gate row", 4: "assert_equal", 5: "range", 6: "chip", 7: "lookup",
               8: "next state"}


class FrameStructureCounts(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in ("cells", "gate_rows", "assert_eq", "assert_const", "ranges", "lookups")]


class RegionHost(C.Structure):
    _fields_ = [("gate", C.c_void_p), ("lookup", C.c_void_p), ("chip_dense", C.c_void_p), ("chip_spread", C.c_void_p)]


class RegionTape(C.Structure):
    _fields_ = [("n_distinct", C.c_uint64), ("distinct_capacity", C.c_uint64), ("gate_cells", C.c_uint64),
                ("lookup_cells", C.c_uint64), ("limb_calls", C.c_uint64), ("gate_code", C.POINTER(C.c_uint32)),
                ("lookup_code", C.POINTER(C.c_uint32)), ("chip_dense_code", C.POINTER(C.c_uint32)),
                ("chip_spread_code", C.POINTER(C.c_uint32)), ("consts", C.c_void_p), ("n_consts", C.c_uint64)]


HSW_TAPE_CONST = 0x80000000


class GadgetView(C.Structure):
    _fields_ = [("d_gate", C.c_void_p), ("d_chip_dense", C.c_void_p), ("d_chip_spread", C.c_void_p),
                ("d_next_states", C.c_void_p), ("chip_col_stride", C.c_size_t), ("blocks_done", C.c_size_t),
                ("capacity_blocks", C.c_size_t), ("num_limb_sum", C.c_uint64), ("cur_hash_idx", C.c_size_t),
                ("gate_cells", C.c_uint64), ("gate_capacity", C.c_uint64), ("d_lookup", C.c_void_p),
                ("lookup_cells", C.c_uint64), ("lookup_capacity", C.c_uint64),
                ("max_rows", C.c_uint64), ("columns", C.c_uint64),
                ("origin_column", C.c_uint64), ("origin_row", C.c_uint64), ("origin_lookups", C.c_uint64),
                ("origin_zero_loaded", C.c_uint32), ("reserved_", C.c_uint32)]


# every symbol include/hsw.h declares (tests check the library exports them all)
SYMBOLS = (
    "hsw_shape_query", "hsw_chip_rows", "hsw_engine_create", "hsw_engine_destroy",
    "hsw_engine_shape", "hsw_engine_synchronize", "hsw_witness_blocks", "hsw_sha256_chain",
    "hsw_witness_blocks_host", "hsw_last_kernel_ms", "hsw_set_timing", "hsw_strerror",
    "hsw_last_error", "hsw_abi_version", "hsw_engine_set_option", "hsw_fill_calibrate",
    "hsw_engine_stream", "hsw_digest_prepare", "hsw_gadget_create", "hsw_gadget_destroy",
    "hsw_gadget_digest", "hsw_gadget_digest_batch", "hsw_gadget_streams", "hsw_gadget_input_bytes",
    "hsw_gadget_set_repr", "hsw_download", "hsw_host_alloc", "hsw_host_free",
    "hsw_shape_query_ex", "hsw_engine_create_ex", "hsw_pack_plan_query", "hsw_gate_tape",
    "hsw_witness_blocks_ex", "hsw_spread_table", "hsw_cell_bytes", "hsw_neg_cells",
    "hsw_frame_query", "hsw_frame_tape", "hsw_witness_frames", "hsw_gadget_create_ex",
    "hsw_gadget_set_columns", "hsw_gadget_cell_position", "hsw_gadget_reset", "hsw_gadget_seek", "hsw_gadget_place", "hsw_device_alloc", "hsw_device_free", "hsw_gadget_download_region",
    "hsw_block_structure", "hsw_frame_structure", "hsw_verify_blocks",
    "hsw_verify_frames", "hsw_gadget_verify", "hsw_last_launch", "hsw_witness_digests",
    "hsw_gadget_download_region_compact", "hsw_region_widen", "hsw_gadget_result_cells",
    "hsw_gadget_set_origin", "hsw_gadget_region_tape", "hsw_gadget_download_region_distinct", "hsw_gadget_replay_region",
)


def build(force=False):
    """Compile libhsw.so for gfx950 (hipcc cross-compiles without a GPU)."""
    if force:
        subprocess.check_call(["make", "-C", CSRC, "-s", "clean"])
    subprocess.check_call(["make", "-C", CSRC, "-s", "-j8"])
    if not os.path.exists(LIB_PATH):
        raise RuntimeError("libhsw.so was not produced by csrc/Makefile")
    return LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            "libhsw.so is missing: build it with `make -C %s` (hipcc, gfx950). "
            "There is no CPU fallback for the witness engine." % CSRC)
    L = C.CDLL(LIB_PATH)
    vp, u8p, u32p = C.c_void_p, C.c_void_p, C.c_void_p
    L.hsw_abi_version.restype = C.c_uint32
    L.hsw_strerror.restype = C.c_char_p
    L.hsw_strerror.argtypes = [C.c_int]
    L.hsw_last_error.restype = C.c_char_p
    L.hsw_last_error.argtypes = [vp]
    L.hsw_shape_query.restype = C.c_int
    L.hsw_shape_query.argtypes = [C.c_uint32, C.c_uint32, C.POINTER(Shape)]
    L.hsw_chip_rows.restype = C.c_uint64
    L.hsw_chip_rows.argtypes = [C.POINTER(Shape), C.c_uint64, C.c_uint64]
    L.hsw_engine_create.restype = C.c_int
    L.hsw_engine_create.argtypes = [C.c_int, vp, C.c_uint32, C.c_uint32, C.POINTER(vp)]
    L.hsw_engine_destroy.restype = None
    L.hsw_engine_destroy.argtypes = [vp]
    L.hsw_engine_shape.restype = C.c_int
    L.hsw_engine_shape.argtypes = [vp, C.POINTER(Shape)]
    L.hsw_engine_synchronize.restype = C.c_int
    L.hsw_engine_synchronize.argtypes = [vp]
    L.hsw_witness_blocks.restype = C.c_int
    L.hsw_witness_blocks.argtypes = [vp, u8p, u32p, C.c_size_t, C.c_uint64, vp, vp, vp,
                                     C.c_size_t, u32p, C.c_uint32]
    L.hsw_sha256_chain.restype = C.c_int
    L.hsw_sha256_chain.argtypes = [vp, u8p, C.c_size_t, C.c_size_t, u32p, u32p]
    L.hsw_witness_blocks_host.restype = C.c_int
    L.hsw_witness_blocks_host.argtypes = [vp, u8p, u32p, C.c_size_t, C.c_uint64, vp, vp, vp,
                                          C.c_size_t, u32p, C.c_uint32]
    L.hsw_last_kernel_ms.restype = C.c_int
    L.hsw_last_kernel_ms.argtypes = [vp, C.POINTER(C.c_float)]
    if hasattr(L, "hsw_engine_set_option"):   # absent only in pre-ABI-1 A/B builds
        L.hsw_engine_set_option.restype = C.c_int
        L.hsw_engine_set_option.argtypes = [vp, C.c_char_p, C.c_int64]
    if hasattr(L, "hsw_fill_calibrate"):
        L.hsw_fill_calibrate.restype = C.c_int
        L.hsw_fill_calibrate.argtypes = [vp, vp, C.c_size_t, C.POINTER(C.c_float)]
    L.hsw_set_timing.restype = C.c_int
    L.hsw_set_timing.argtypes = [vp, C.c_int]
    if hasattr(L, "hsw_gadget_create"):
        L.hsw_engine_stream.restype = C.c_int
        L.hsw_engine_stream.argtypes = [vp, C.POINTER(vp), C.POINTER(C.c_int)]
        L.hsw_digest_prepare.restype = C.c_int
        L.hsw_digest_prepare.argtypes = [vp, C.c_size_t, C.c_size_t, C.c_size_t, vp, vp, C.POINTER(DigestInfo)]
        L.hsw_gadget_create.restype = C.c_int
        L.hsw_gadget_create.argtypes = [vp, C.POINTER(C.c_size_t), C.c_size_t, C.c_int, C.POINTER(vp)]
        L.hsw_gadget_destroy.restype = None
        L.hsw_gadget_destroy.argtypes = [vp]
        L.hsw_gadget_digest.restype = C.c_int
        L.hsw_gadget_digest.argtypes = [vp, vp, C.c_size_t, C.c_size_t, C.POINTER(HashResult)]
        L.hsw_gadget_digest_batch.restype = C.c_int
        L.hsw_gadget_digest_batch.argtypes = [vp, C.c_size_t, C.POINTER(vp), C.POINTER(C.c_size_t),
                                              C.POINTER(C.c_size_t), C.POINTER(HashResult)]
        L.hsw_gadget_streams.restype = C.c_int
        L.hsw_gadget_streams.argtypes = [vp, C.POINTER(GadgetView)]
        L.hsw_gadget_input_bytes.restype = C.c_int
        L.hsw_gadget_input_bytes.argtypes = [vp, C.c_size_t, vp, C.c_size_t, C.POINTER(C.c_size_t)]
        L.hsw_shape_query_ex.restype = C.c_int
        L.hsw_shape_query_ex.argtypes = [C.c_uint32, C.c_uint32, C.c_uint32, C.POINTER(Shape)]
        L.hsw_engine_create_ex.restype = C.c_int
        L.hsw_engine_create_ex.argtypes = [C.c_int, vp, C.c_uint32, C.c_uint32, C.c_uint32, C.POINTER(vp)]
        L.hsw_pack_plan_query.restype = C.c_int
        L.hsw_pack_plan_query.argtypes = [C.POINTER(Shape), C.c_size_t, C.c_uint64, C.c_uint64, C.POINTER(PackPlan)]
        L.hsw_gate_tape.restype = C.c_int
        L.hsw_gate_tape.argtypes = [C.POINTER(Shape), vp, C.c_size_t, C.POINTER(C.c_size_t)]
        L.hsw_witness_blocks_ex.restype = C.c_int
        L.hsw_witness_blocks_ex.argtypes = [vp, C.POINTER(WitnessArgs)]
        L.hsw_cell_bytes.restype = C.c_uint32
        L.hsw_cell_bytes.argtypes = [C.c_uint32]
        L.hsw_neg_cells.restype = C.c_int
        L.hsw_neg_cells.argtypes = [C.POINTER(Shape), vp, C.c_size_t, C.POINTER(C.c_size_t)]
        L.hsw_spread_table.restype = C.c_int
        L.hsw_spread_table.argtypes = [C.c_uint32, vp, vp]
        L.hsw_host_alloc.restype = C.c_int
        L.hsw_host_alloc.argtypes = [C.c_size_t, C.POINTER(vp)]
        L.hsw_host_free.restype = None
        L.hsw_host_free.argtypes = [vp]
        L.hsw_download.restype = C.c_int
        L.hsw_download.argtypes = [vp, vp, vp, C.c_size_t]
        L.hsw_gadget_set_repr.restype = C.c_int
        L.hsw_gadget_set_repr.argtypes = [vp, C.c_uint32]
        L.hsw_frame_query.restype = C.c_int
        L.hsw_frame_query.argtypes = [C.POINTER(Shape), C.c_size_t, C.c_int, C.POINTER(FrameShape)]
        L.hsw_frame_tape.restype = C.c_int
        L.hsw_frame_tape.argtypes = [C.POINTER(Shape), C.c_size_t, C.c_int, C.c_int, vp, C.c_size_t,
                                     C.POINTER(C.c_size_t)]
        L.hsw_witness_frames.restype = C.c_int
        L.hsw_witness_frames.argtypes = [vp, C.POINTER(FrameDesc), C.c_size_t, vp, vp, vp, vp, vp,
                                         C.POINTER(PackPlan), C.c_uint32]
        L.hsw_gadget_set_columns.restype = C.c_int
        L.hsw_gadget_set_columns.argtypes = [vp, C.c_uint64, C.POINTER(C.c_uint64)]
        L.hsw_gadget_region_tape.restype = C.c_int
        L.hsw_gadget_region_tape.argtypes = [vp, C.POINTER(RegionTape)]
        L.hsw_gadget_download_region_distinct.restype = C.c_int
        L.hsw_gadget_download_region_distinct.argtypes = [vp, vp, C.c_size_t, C.POINTER(C.c_size_t)]
        L.hsw_gadget_replay_region.restype = C.c_int
        L.hsw_gadget_replay_region.argtypes = [vp, vp, C.POINTER(RegionHost), C.c_uint]
        L.hsw_gadget_set_origin.restype = C.c_int
        L.hsw_gadget_set_origin.argtypes = [vp, C.c_uint64, C.c_uint64, C.c_int, C.c_uint64]
        L.hsw_verify_frames.restype = C.c_int
        L.hsw_verify_frames.argtypes = [vp, C.POINTER(FrameDesc), C.c_size_t, vp, vp, vp, vp, vp, C.POINTER(PackPlan),
                                        C.c_uint32, C.POINTER(VerifyReport)]
        L.hsw_gadget_verify.restype = C.c_int
        L.hsw_gadget_verify.argtypes = [vp, C.POINTER(VerifyReport)]
        L.hsw_verify_blocks.restype = C.c_int
        L.hsw_verify_blocks.argtypes = [vp, C.POINTER(WitnessArgs), C.POINTER(VerifyReport)]
        L.hsw_frame_structure.restype = C.c_int
        L.hsw_frame_structure.argtypes = [C.POINTER(Shape), C.c_size_t, C.c_int, C.c_int,
                                          C.POINTER(FrameStructureCounts)] + [vp] * 7
        L.hsw_block_structure.restype = C.c_int
        L.hsw_block_structure.argtypes = [C.POINTER(Shape), C.POINTER(StructureCounts)] + [vp] * 8
        L.hsw_gadget_download_region.restype = C.c_int
        L.hsw_gadget_download_region.argtypes = [vp, C.POINTER(RegionHost)]
        L.hsw_gadget_seek.restype = C.c_int
        L.hsw_gadget_seek.argtypes = [vp, C.c_size_t]
        L.hsw_device_alloc.restype = C.c_int
        L.hsw_device_alloc.argtypes = [C.c_int, C.c_size_t, C.c_size_t, C.POINTER(C.c_void_p)]
        L.hsw_device_free.restype = C.c_int
        L.hsw_device_free.argtypes = [vp]
        L.hsw_gadget_place.restype = C.c_int
        L.hsw_gadget_place.argtypes = [vp, C.c_uint, C.POINTER(C.c_float), C.POINTER(C.c_uint)]
        L.hsw_gadget_reset.restype = C.c_int
        L.hsw_gadget_reset.argtypes = [vp]
        L.hsw_gadget_cell_position.restype = C.c_int
        L.hsw_gadget_cell_position.argtypes = [vp, C.c_uint64, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
        L.hsw_gadget_create_ex.restype = C.c_int
        L.hsw_gadget_create_ex.argtypes = [vp, C.POINTER(C.c_size_t), C.c_size_t, C.c_int, C.c_uint32,
                                           C.POINTER(vp)]
        L.hsw_last_launch.restype = C.c_int
        L.hsw_last_launch.argtypes = [vp, C.POINTER(LaunchInfo)]
        L.hsw_witness_digests.restype = C.c_int
        L.hsw_witness_digests.argtypes = [vp, C.POINTER(DigestsArgs)]
        L.hsw_gadget_download_region_compact.restype = C.c_int
        L.hsw_gadget_download_region_compact.argtypes = [vp, C.POINTER(RegionCompact)]
        L.hsw_gadget_result_cells.restype = C.c_int
        L.hsw_gadget_result_cells.argtypes = [vp, C.c_size_t, C.POINTER(ResultCells)]
        L.hsw_region_widen.restype = C.c_int
        L.hsw_region_widen.argtypes = [vp, C.c_size_t, C.c_uint64, vp, C.c_size_t, vp]
    _lib = L
    return L


class HswError(RuntimeError):
    def __init__(self, status, detail=""):
        self.status = status
        msg = lib().hsw_strerror(status).decode()
        super().__init__("hsw status %d (%s)%s" % (status, msg, (": " + detail) if detail else ""))


def digest_prepare(message: bytes, max_variable_byte_size: int, precomputed_input_len: int = 0):
    """Host-only lib.rs:77-160: returns (blocks bytes, init_state list, DigestInfo dict)."""
    import numpy as np
    msg = bytes(message)
    buf = (C.c_uint8 * max(len(msg), 1)).from_buffer_copy(msg if msg else b"\0")
    blocks = np.zeros(max(max_variable_byte_size, 1), dtype=np.uint8)
    init = np.zeros(8, dtype=np.uint32)
    info = DigestInfo()
    rc = lib().hsw_digest_prepare(C.addressof(buf) if msg else None, len(msg), precomputed_input_len,
                                  max_variable_byte_size, blocks.ctypes.data, init.ctypes.data, C.byref(info))
    if rc != HSW_OK:
        raise HswError(rc)
    return blocks[:max_variable_byte_size], init, {k: int(getattr(info, k)) for k, _ in DigestInfo._fields_}


def shape_query(num_bits_lookup=8, num_advice_columns=2, mode=HSW_MODE_DEFAULT):
    s = Shape()
    rc = lib().hsw_shape_query_ex(num_bits_lookup, num_advice_columns, mode, C.byref(s))
    if rc != HSW_OK:
        raise HswError(rc)
    return s


def spread_table(num_bits_lookup=8):
    """SpreadConfig::load rows (spread.rs:165-194) as two numpy uint64 arrays."""
    import numpy as np
    n = 1 << num_bits_lookup
    d, s = np.zeros(n, dtype=np.uint64), np.zeros(n, dtype=np.uint64)
    rc = lib().hsw_spread_table(num_bits_lookup, d.ctypes.data, s.ctypes.data)
    if rc != HSW_OK:
        raise HswError(rc)
    return d, s


def neg_cells(shape):
    """Block-relative gate cells holding field negations (hsw_neg_cells), numpy uint32."""
    import numpy as np
    out = np.zeros(256, dtype=np.uint32)
    n = C.c_size_t()
    rc = lib().hsw_neg_cells(C.byref(shape), out.ctypes.data, 256, C.byref(n))
    if rc != HSW_OK:
        raise HswError(rc)
    return out[: n.value]


def gate_tape(shape):
    """assign_region call lengths of one block (numpy uint8), hsw_gate_tape."""
    import numpy as np
    n = C.c_size_t()
    rc = lib().hsw_gate_tape(C.byref(shape), None, 0, C.byref(n))
    if rc != HSW_OK:
        raise HswError(rc)
    lens = np.zeros(n.value, dtype=np.uint8)
    rc = lib().hsw_gate_tape(C.byref(shape), lens.ctypes.data, lens.size, None)
    if rc != HSW_OK:
        raise HswError(rc)
    return lens


def pack_plan(shape, n_blocks, start_row, max_rows):
    p = PackPlan()
    rc = lib().hsw_pack_plan_query(C.byref(shape), n_blocks, start_row, max_rows, C.byref(p))
    if rc != HSW_OK:
        raise HswError(rc)
    return p


def frame_query(shape, max_variable_byte_size, is_input_range_check=False):
    """hsw_frame_query: cell counts of the digest frame."""
    fs = FrameShape()
    rc = lib().hsw_frame_query(C.byref(shape), max_variable_byte_size, 1 if is_input_range_check else 0, C.byref(fs))
    if rc != HSW_OK:
        raise HswError(rc)
    return fs


def frame_tape(shape, max_variable_byte_size, is_input_range_check, section):
    """assign_region call lengths of the prologue (section 0) / epilogue (1), numpy uint8."""
    import numpy as np
    n = C.c_size_t()
    rc = lib().hsw_frame_tape(C.byref(shape), max_variable_byte_size, 1 if is_input_range_check else 0, section,
                              None, 0, C.byref(n))
    if rc != HSW_OK:
        raise HswError(rc)
    lens = np.zeros(n.value, dtype=np.uint8)
    rc = lib().hsw_frame_tape(C.byref(shape), max_variable_byte_size, 1 if is_input_range_check else 0, section,
                              lens.ctypes.data, lens.size, None)
    if rc != HSW_OK:
        raise HswError(rc)
    return lens


def block_structure(shape):
    """hsw_block_structure as a dict of numpy arrays: kind (G,) u8, ref (G,) i64, gate_rows (n,) u32,
    assert_eq (n,2), range (n,2), lookup_src (LK,), chip (LC,2), next_state (8,) -- all i64 cell ids."""
    import numpy as np
    c = StructureCounts()
    rc = lib().hsw_block_structure(C.byref(shape), C.byref(c), None, None, None, None, None, None, None, None)
    if rc != HSW_OK:
        raise HswError(rc)
    kind = np.zeros(c.gate_cells, dtype=np.uint8)
    ref = np.zeros(c.gate_cells, dtype=np.int64)
    rows = np.zeros(c.gate_rows, dtype=np.uint32)
    aeq = np.zeros((c.assert_eq, 2), dtype=np.int64)
    rng = np.zeros((c.ranges, 2), dtype=np.int64)
    lk = np.zeros(c.lookups, dtype=np.int64)
    chip = np.zeros((c.limb_calls, 2), dtype=np.int64)
    ns = np.zeros(8, dtype=np.int64)
    rc = lib().hsw_block_structure(C.byref(shape), None, kind.ctypes.data, ref.ctypes.data, rows.ctypes.data,
                                   aeq.ctypes.data, rng.ctypes.data, lk.ctypes.data, chip.ctypes.data, ns.ctypes.data)
    if rc != HSW_OK:
        raise HswError(rc)
    return dict(kind=kind, ref=ref, gate_rows=rows, assert_eq=aeq, range=rng, lookup_src=lk, chip=chip, next_state=ns)


def frame_structure(shape, max_variable_byte_size, is_input_range_check, section):
    """hsw_frame_structure as a dict of numpy arrays (section 0 = prologue, 1 = epilogue)."""
    import numpy as np
    c = FrameStructureCounts()
    args = (C.byref(shape), max_variable_byte_size, 1 if is_input_range_check else 0, section)
    rc = lib().hsw_frame_structure(*args, C.byref(c), None, None, None, None, None, None, None)
    if rc != HSW_OK:
        raise HswError(rc)
    kind = np.zeros(c.cells, dtype=np.uint8)
    ref = np.zeros(c.cells, dtype=np.int64)
    rows = np.zeros(c.gate_rows, dtype=np.uint32)
    aeq = np.zeros((c.assert_eq, 2), dtype=np.int64)
    ac = np.zeros((c.assert_const, 2), dtype=np.int64)
    rng = np.zeros((c.ranges, 2), dtype=np.int64)
    lk = np.zeros(c.lookups, dtype=np.int64)
    rc = lib().hsw_frame_structure(*args, None, kind.ctypes.data, ref.ctypes.data, rows.ctypes.data, aeq.ctypes.data,
                                   ac.ctypes.data, rng.ctypes.data, lk.ctypes.data)
    if rc != HSW_OK:
        raise HswError(rc)
    return dict(kind=kind, ref=ref, gate_rows=rows, assert_eq=aeq, assert_const=ac, range=rng, lookup_src=lk)
