//! Raw bindings of `include/hsw.h` (ABI version 1).  One item per C declaration;
//! struct layouts are `#[repr(C)]` mirrors.  Not compiled in the build image.
#![allow(non_camel_case_types)]
use std::os::raw::{c_char, c_int, c_uint, c_void};

#[repr(C)] pub struct hsw_engine { _private: [u8; 0] }
#[repr(C)] pub struct hsw_gadget { _private: [u8; 0] }

pub const HSW_OK: c_int = 0;
pub const HSW_ERR_INVALID_ARG: c_int = 1;
pub const HSW_ERR_SHAPE: c_int = 2;
pub const HSW_ERR_NO_DEVICE: c_int = 3;
pub const HSW_ERR_HIP: c_int = 4;
pub const HSW_ERR_UNSUPPORTED: c_int = 5;
pub const HSW_ERR_TOO_LARGE: c_int = 6;
pub const HSW_ERR_NOMEM: c_int = 7;

pub const HSW_REPR_CANONICAL: u32 = 0;
pub const HSW_REPR_MONTGOMERY: u32 = 1;
pub const HSW_SKIP_GATE: u32 = 2;
pub const HSW_SKIP_CHIP: u32 = 4;
pub const HSW_HOST_REGISTER: u32 = 8;
pub const HSW_REPR_COMPACT64: u32 = 16;
/// The blocks are ONE message; `d_pre_states` holds its initial state only (small-batch launches).
pub const HSW_CHAINED: u32 = 32;
pub const HSW_MODE_DEFAULT: u32 = 0;
pub const HSW_MODE_HALO2_INTERNALS: u32 = 1;
pub const HSW_MAX_BREAKS: usize = 16;
pub const HSW_CELL_BYTES: usize = 32;
pub const HSW_GADGET_WHOLE_DIGEST: u32 = 1;
pub const HSW_GADGET_INDEPENDENT: u32 = 2;

#[repr(C)]
#[derive(Default, Clone, Copy, Debug)]
pub struct hsw_shape {
    pub num_bits_lookup: u32,
    pub num_advice_columns: u32,
    pub limbs_per_spread: u32,
    pub cells_per_spread: u32,
    pub cells_per_state_spread: u32,
    pub cells_per_sigma: u32,
    pub cells_per_ch: u32,
    pub cells_per_maj: u32,
    pub cells_per_sched_step: u32,
    pub cells_per_round: u32,
    pub off_words: u32,
    pub off_msg_spread: u32,
    pub off_sched: u32,
    pub off_state_spread: u32,
    pub off_rounds: u32,
    pub off_feed: u32,
    pub gate_cells_per_block: u32,
    pub spread_calls_per_block: u32,
    pub limb_calls_per_block: u32,
    pub chip_cells_per_block: u32,
    pub algorithmic_bytes_per_block: u64,
    pub mode: u32,
    pub lookup_cells_per_block: u32,
    pub gate_calls_per_block: u32,
    pub reserved_: u32,
}

#[repr(C)]
#[derive(Clone, Copy, Debug)]
pub struct hsw_pack_plan {
    pub n_breaks: u32,
    pub columns_touched: u32,
    pub break_cell: [u64; HSW_MAX_BREAKS],
    pub break_gap: [u64; HSW_MAX_BREAKS],
    pub span_cells: u64,
    pub end_row: u64,
}

/// `hsw_last_launch`: the kernel instantiation and work split of the most recent expansion launch.
#[repr(C)]
#[derive(Clone, Copy, Debug, Default)]
pub struct hsw_launch_info {
    pub limbs: u32,
    pub tile_cells: u32,
    pub tile_rows: u32,
    pub repr: u32,
    pub internals: u32,
    pub parts: u32,
    pub split: u32,
    pub reserved_: u32,
    pub n_blocks: u64,
    pub grid: u64,
}

#[repr(C)]
pub struct hsw_witness_args {
    pub d_blocks: *const u8,
    pub d_pre_states: *const u32,
    pub n_blocks: usize,
    pub spread_cursor0: u64,
    pub d_gate: *mut c_void,
    pub d_chip_dense: *mut c_void,
    pub d_chip_spread: *mut c_void,
    pub chip_col_stride: usize,
    pub d_next_states: *mut u32,
    pub d_lookup: *mut c_void,
    pub flags: u32,
    pub pack: *const hsw_pack_plan,
    /// whole-digest streams: blocks per digest / cells and lookup cells skipped between digests (0 = off)
    pub frame_every: u64,
    pub frame_cells: u64,
    pub frame_lookups: u64,
}

/// Cell counts of the frame `digest` puts around its block loop (reference src/lib.rs:122-178, 294-341).
#[repr(C)]
#[derive(Default, Clone, Copy, Debug)]
pub struct hsw_frame_shape {
    pub n_blocks: u64,
    pub prologue_cells: u64,
    pub epilogue_cells: u64,
    pub prologue_lookups: u64,
    pub epilogue_lookups: u64,
    pub prologue_calls: u64,
    pub epilogue_calls: u64,
    pub digest_cells: u64,
    pub digest_lookups: u64,
}

/// One `digest()` call for `hsw_witness_frames`.
#[repr(C)]
#[derive(Default, Clone, Copy, Debug)]
pub struct hsw_frame_desc {
    pub input_len: u64,
    pub first_block: u64,
    pub prologue_cell: u64,
    pub epilogue_cell: u64,
    pub prologue_lookup: u64,
    pub epilogue_lookup: u64,
    pub zero_cell: u64,
    pub n_blocks: u32,
    pub num_round: u32,
    pub precomputed_round: u32,
    pub is_input_range_check: u32,
}

/// `hsw_witness_digests`: block streams + frames of n equally sized digests in one call.
#[repr(C)]
pub struct hsw_digests_args {
    pub blocks: hsw_witness_args,
    pub descs: *const hsw_frame_desc,
    pub n_digests: usize,
    pub d_blocks0: *const u8,
    pub d_pre_states0: *const u32,
    pub d_next_states0: *const u32,
    pub d_gate0: *mut c_void,
    pub d_lookup0: *mut c_void,
    pub frame_pack: *const hsw_pack_plan,
    pub host_next_states: *mut u32,
}

pub const HSW_STREAM_GATE: u64 = 0;
pub const HSW_STREAM_LOOKUP: u64 = 1;
pub const HSW_STREAM_CHIP_DENSE: u64 = 2;
pub const HSW_STREAM_CHIP_SPREAD: u64 = 3;

/// One cell wider than 64 bits in the compact delivery of a region.
#[repr(C)]
#[derive(Default, Clone, Copy, Debug)]
pub struct hsw_wide_cell {
    pub stream: u64,
    pub index: u64,
    pub value: [u64; 4],
}

/// `hsw_gadget_result_cells`: where the cells of one digest's `AssignedHashResult` sit.
#[repr(C)]
#[derive(Clone, Copy, Debug)]
pub struct hsw_result_cells {
    pub input_len_cell: u64,
    pub input_bytes_cell0: u64,
    pub n_input_bytes: u64,
    pub output_byte_cells: [u64; 32],
    pub input_len_pos: [u64; 2],
    pub input_bytes_pos0: [u64; 2],
    pub output_byte_pos: [[u64; 2]; 32],
}

#[repr(C)]
pub struct hsw_region_compact {
    pub gate: *mut u64,
    pub lookup: *mut u64,
    pub chip_dense: *mut u64,
    pub chip_spread: *mut u64,
    pub wide: *mut hsw_wide_cell,
    pub wide_cap: usize,
    pub n_wide: usize,
}

#[repr(C)]
#[derive(Default, Clone, Copy, Debug)]
pub struct hsw_digest_info {
    pub num_round: usize,
    pub precomputed_round: usize,
    pub target_round: usize,
    pub n_blocks: usize,
}

/// `AssignedHashResult` (reference src/lib.rs:31-36) on values.
#[repr(C)]
#[derive(Clone, Copy, Debug)]
pub struct hsw_hash_result {
    pub input_len: u64,
    pub first_block: usize,
    pub n_blocks: usize,
    pub spread_cursor0: u64,
    pub num_round: usize,
    pub target_round: usize,
    pub output_bytes: [u8; 32],
    /// HSW_GADGET_WHOLE_DIGEST: section starts of this digest in the gate / lookup streams (cells)
    pub prologue_cell: u64,
    pub block_cell: u64,
    pub epilogue_cell: u64,
    pub end_cell: u64,
    pub prologue_lookup: u64,
    pub block_lookup: u64,
    pub epilogue_lookup: u64,
}

pub const HSW_CELL_INPUT_BYTE0: i64 = -1;
pub const HSW_CELL_PRE_STATE0: i64 = -100;
pub const HSW_CELL_ZERO: i64 = -1000;
pub const HSW_CELL_HIDDEN: i64 = -2000;
pub const HSW_KIND_WITNESS: u8 = 0;
pub const HSW_KIND_CONSTANT: u8 = 1;
pub const HSW_KIND_EXISTING: u8 = 2;

/// Result of `hsw_verify_blocks`.
#[repr(C)]
#[derive(Default, Clone, Copy, Debug)]
pub struct hsw_verify_report {
    pub violations: u64,
    pub checks: u64,
    pub first_block: u64,
    pub first_cell: i64,
    pub first_class: u32,
    pub kernel_ms: f32,
}

pub const HSW_CELL_TARGET: i64 = -3000;
pub const HSW_CELL_STATE0: i64 = -4000;

/// Sizes of the arrays `hsw_frame_structure` fills.
#[repr(C)]
#[derive(Default, Clone, Copy, Debug)]
pub struct hsw_frame_structure_counts {
    pub cells: u64,
    pub gate_rows: u64,
    pub assert_eq: u64,
    pub assert_const: u64,
    pub ranges: u64,
    pub lookups: u64,
}

/// Sizes of the arrays `hsw_block_structure` fills.
#[repr(C)]
#[derive(Default, Clone, Copy, Debug)]
pub struct hsw_structure_counts {
    pub gate_cells: u64,
    pub gate_rows: u64,
    pub assert_eq: u64,
    pub ranges: u64,
    pub lookups: u64,
    pub limb_calls: u64,
}

/// Host destinations of `hsw_gadget_download_region` (any may be null).
#[repr(C)]
pub struct hsw_region_host {
    pub gate: *mut c_void,
    pub lookup: *mut c_void,
    pub chip_dense: *mut c_void,
    pub chip_spread: *mut c_void,
}

pub const HSW_TAPE_CONST: u32 = 0x8000_0000;
#[repr(C)]
pub struct hsw_region_tape {
    pub n_distinct: u64,
    pub distinct_capacity: u64,
    pub gate_cells: u64,
    pub lookup_cells: u64,
    pub limb_calls: u64,
    pub gate_code: *const u32,
    pub lookup_code: *const u32,
    pub chip_dense_code: *const u32,
    pub chip_spread_code: *const u32,
    pub consts: *const c_void,
    pub n_consts: u64,
}

#[repr(C)]
pub struct hsw_gadget_view {
    pub d_gate: *mut c_void,
    pub d_chip_dense: *mut c_void,
    pub d_chip_spread: *mut c_void,
    pub d_next_states: *mut u32,
    pub chip_col_stride: usize,
    pub blocks_done: usize,
    pub capacity_blocks: usize,
    pub num_limb_sum: u64,
    pub cur_hash_idx: usize,
    pub gate_cells: u64,
    pub gate_capacity: u64,
    pub d_lookup: *mut c_void,
    pub lookup_cells: u64,
    pub lookup_capacity: u64,
    pub max_rows: u64,
    pub columns: u64,
    pub origin_column: u64,
    pub origin_row: u64,
    pub origin_lookups: u64,
    pub origin_zero_loaded: u32,
    pub reserved_: u32,
}

extern "C" {
    pub fn hsw_abi_version() -> u32;
    pub fn hsw_strerror(status: c_int) -> *const c_char;
    pub fn hsw_last_error(e: *const hsw_engine) -> *const c_char;

    pub fn hsw_shape_query(num_bits_lookup: u32, num_advice_columns: u32, out: *mut hsw_shape) -> c_int;
    pub fn hsw_shape_query_ex(num_bits_lookup: u32, num_advice_columns: u32, mode: u32, out: *mut hsw_shape) -> c_int;
    pub fn hsw_spread_table(num_bits_lookup: u32, dense_out: *mut u64, spread_out: *mut u64) -> c_int;
    pub fn hsw_cell_bytes(flags: u32) -> u32;
    pub fn hsw_neg_cells(shape: *const hsw_shape, out: *mut u32, cap: usize, n: *mut usize) -> c_int;
    pub fn hsw_chip_rows(shape: *const hsw_shape, spread_cursor0: u64, n_blocks: u64) -> u64;

    pub fn hsw_engine_create(device: c_int, hip_stream: *mut c_void, num_bits_lookup: u32,
                             num_advice_columns: u32, out: *mut *mut hsw_engine) -> c_int;
    pub fn hsw_engine_create_ex(device: c_int, hip_stream: *mut c_void, num_bits_lookup: u32,
                                num_advice_columns: u32, mode: u32, out: *mut *mut hsw_engine) -> c_int;
    pub fn hsw_engine_destroy(e: *mut hsw_engine);
    pub fn hsw_engine_shape(e: *const hsw_engine, out: *mut hsw_shape) -> c_int;
    pub fn hsw_engine_synchronize(e: *mut hsw_engine) -> c_int;
    pub fn hsw_engine_stream(e: *const hsw_engine, hip_stream: *mut *mut c_void, device: *mut c_int) -> c_int;
    pub fn hsw_engine_set_option(e: *mut hsw_engine, name: *const c_char, value: i64) -> c_int;
    pub fn hsw_last_launch(e: *const hsw_engine, out: *mut hsw_launch_info) -> c_int;
    /// Block streams and frames of n equally sized digests in one call (one kernel launch up to 128 blocks, in digests of up to 32).
    pub fn hsw_witness_digests(e: *mut hsw_engine, args: *const hsw_digests_args) -> c_int;
    pub fn hsw_gadget_result_cells(g: *const hsw_gadget, hash_idx: usize, out: *mut hsw_result_cells) -> c_int;
    pub fn hsw_gadget_download_region_compact(g: *mut hsw_gadget, dst: *mut hsw_region_compact) -> c_int;
    pub fn hsw_region_widen(compact: *const u64, n_cells: usize, stream_id: u64, wide: *const hsw_wide_cell,
                            n_wide: usize, cells32: *mut c_void) -> c_int;

    /// Replaces the block loop of reference src/lib.rs:180-238 over src/compression.rs:19-25.
    pub fn hsw_witness_blocks(e: *mut hsw_engine, d_blocks: *const u8, d_pre_states: *const u32,
                              n_blocks: usize, spread_cursor0: u64, d_gate: *mut c_void,
                              d_chip_dense: *mut c_void, d_chip_spread: *mut c_void,
                              chip_col_stride: usize, d_next_states: *mut u32, flags: u32) -> c_int;
    pub fn hsw_witness_blocks_ex(e: *mut hsw_engine, args: *const hsw_witness_args) -> c_int;
    pub fn hsw_witness_blocks_host(e: *mut hsw_engine, blocks: *const u8, pre_states: *const u32,
                                   n_blocks: usize, spread_cursor0: u64, gate: *mut c_void,
                                   chip_dense: *mut c_void, chip_spread: *mut c_void,
                                   chip_col_stride: usize, next_states: *mut u32, flags: u32) -> c_int;
    pub fn hsw_sha256_chain(e: *mut hsw_engine, d_blocks: *const u8, n_messages: usize,
                            blocks_per_message: usize, d_init_states: *const u32,
                            d_pre_states: *mut u32) -> c_int;

    pub fn hsw_pack_plan_query(shape: *const hsw_shape, n_blocks: usize, start_row: u64, max_rows: u64,
                               out: *mut hsw_pack_plan) -> c_int;
    pub fn hsw_gate_tape(shape: *const hsw_shape, lens_out: *mut u8, cap: usize, n_calls: *mut usize) -> c_int;

    pub fn hsw_digest_prepare(input: *const u8, input_len: usize, precomputed_input_len: usize,
                              max_variable_byte_size: usize, blocks_out: *mut u8,
                              init_state_out: *mut u32, info: *mut hsw_digest_info) -> c_int;
    pub fn hsw_gadget_create(e: *mut hsw_engine, max_variable_byte_sizes: *const usize, n_hashes: usize,
                             is_input_range_check: c_int, out: *mut *mut hsw_gadget) -> c_int;
    pub fn hsw_gadget_create_ex(e: *mut hsw_engine, max_variable_byte_sizes: *const usize, n_hashes: usize,
                                is_input_range_check: c_int, flags: u32, out: *mut *mut hsw_gadget) -> c_int;
    pub fn hsw_gadget_destroy(g: *mut hsw_gadget);
    pub fn hsw_gadget_set_columns(g: *mut hsw_gadget, max_rows: u64, n_columns: *mut u64) -> c_int;
    /// Distinct-value delivery: the input-independent tape, the packed new witnesses, the host replay.
    pub fn hsw_gadget_region_tape(g: *mut hsw_gadget, out: *mut hsw_region_tape) -> c_int;
    pub fn hsw_gadget_download_region_distinct(g: *mut hsw_gadget, distinct: *mut c_void, cap_cells: usize,
                                               n_cells: *mut usize) -> c_int;
    pub fn hsw_gadget_replay_region(g: *mut hsw_gadget, distinct: *const c_void, dst: *const hsw_region_host,
                                    threads: u32) -> c_int;
    /// Where the caller's `Context` stands: `ctx.advice_alloc[0]`, `ctx.zero_cell.is_some()`, `ctx.cells_to_lookup.len()`.
    pub fn hsw_gadget_set_origin(g: *mut hsw_gadget, column: u64, row: u64, zero_cell_loaded: c_int,
                                 lookups_already_queued: u64) -> c_int;
    pub fn hsw_gadget_reset(g: *mut hsw_gadget) -> c_int;
    pub fn hsw_gadget_seek(g: *mut hsw_gadget, hash_idx: usize) -> c_int;
    /// Buffer placement: try `candidates` allocations of the chip columns, keep the one the gadget's own batch runs fastest on.
    pub fn hsw_gadget_place(g: *mut hsw_gadget, candidates: c_uint, ms_each: *mut f32, kept: *mut c_uint) -> c_int;
    pub fn hsw_verify_frames(e: *mut hsw_engine, descs: *const hsw_frame_desc, n: usize, d_blocks: *const u8,
                             d_pre_states: *const u32, d_next_states: *const u32, d_gate: *const c_void,
                             d_lookup: *const c_void, pack: *const hsw_pack_plan, flags: u32,
                             report: *mut hsw_verify_report) -> c_int;
    pub fn hsw_gadget_verify(g: *mut hsw_gadget, report: *mut hsw_verify_report) -> c_int;
    pub fn hsw_verify_blocks(e: *mut hsw_engine, args: *const hsw_witness_args, report: *mut hsw_verify_report) -> c_int;
    pub fn hsw_frame_structure(shape: *const hsw_shape, max_variable_byte_size: usize, is_input_range_check: c_int,
                               section: c_int, counts: *mut hsw_frame_structure_counts, cell_kind: *mut u8,
                               cell_ref: *mut i64, gate_rows: *mut u32, assert_eq: *mut i64,
                               assert_const: *mut i64, range: *mut i64, lookup_src: *mut i64) -> c_int;
    pub fn hsw_block_structure(shape: *const hsw_shape, counts: *mut hsw_structure_counts, cell_kind: *mut u8,
                               cell_ref: *mut i64, gate_rows: *mut u32, assert_eq: *mut i64, range: *mut i64,
                               lookup_src: *mut i64, chip: *mut i64, next_state: *mut i64) -> c_int;
    pub fn hsw_gadget_download_region(g: *mut hsw_gadget, dst: *const hsw_region_host) -> c_int;
    pub fn hsw_gadget_cell_position(g: *const hsw_gadget, cell: u64, column: *mut u64, row: *mut u64) -> c_int;
    pub fn hsw_frame_query(shape: *const hsw_shape, max_variable_byte_size: usize, is_input_range_check: c_int,
                           out: *mut hsw_frame_shape) -> c_int;
    pub fn hsw_frame_tape(shape: *const hsw_shape, max_variable_byte_size: usize, is_input_range_check: c_int,
                          section: c_int, lens_out: *mut u8, cap: usize, n_calls: *mut usize) -> c_int;
    pub fn hsw_witness_frames(e: *mut hsw_engine, descs: *const hsw_frame_desc, n: usize, d_blocks: *const u8,
                              d_pre_states: *const u32, d_next_states: *const u32, d_gate: *mut c_void,
                              d_lookup: *mut c_void, pack: *const hsw_pack_plan, flags: u32) -> c_int;
    pub fn hsw_gadget_digest(g: *mut hsw_gadget, input: *const u8, input_len: usize,
                             precomputed_input_len: usize, result: *mut hsw_hash_result) -> c_int;
    pub fn hsw_gadget_digest_batch(g: *mut hsw_gadget, n: usize, inputs: *const *const u8,
                                   input_lens: *const usize, precomputed_input_lens: *const usize,
                                   results: *mut hsw_hash_result) -> c_int;
    pub fn hsw_gadget_streams(g: *mut hsw_gadget, view: *mut hsw_gadget_view) -> c_int;
    pub fn hsw_gadget_input_bytes(g: *mut hsw_gadget, hash_idx: usize, out: *mut u8, cap: usize,
                                  len: *mut usize) -> c_int;
    pub fn hsw_gadget_set_repr(g: *mut hsw_gadget, repr: u32) -> c_int;

    pub fn hsw_download(e: *mut hsw_engine, host_dst: *mut c_void, d_src: *const c_void, bytes: usize) -> c_int;
    pub fn hsw_host_alloc(bytes: usize, out: *mut *mut c_void) -> c_int;
    pub fn hsw_host_free(p: *mut c_void);
    /// Device memory for witness streams: one virtual range backed by physical allocations of up to `chunk_bytes` (0 = 4 GiB).
    pub fn hsw_device_alloc(device: c_int, bytes: usize, chunk_bytes: usize, out: *mut *mut c_void) -> c_int;
    pub fn hsw_device_free(ptr: *mut c_void) -> c_int;
    pub fn hsw_fill_calibrate(e: *mut hsw_engine, d_buf: *mut c_void, bytes: usize, ms: *mut f32) -> c_int;
    pub fn hsw_last_kernel_ms(e: *mut hsw_engine, ms: *mut f32) -> c_int;
    pub fn hsw_set_timing(e: *mut hsw_engine, enabled: c_int) -> c_int;
}
