// hsw_gadget.hpp -- host-side mirror of the reference's gadget front-end.
//
// Mirrors, by name and argument meaning, reference src/lib.rs:
//   AssignedHashResult            lib.rs:31-36
//   Sha256DynamicConfig           lib.rs:38-45
//     ::configure                 lib.rs:49-69
//     ::digest                    lib.rs:71-349
//     ::new_context               lib.rs:351-360
//     ::load                      lib.rs:366-368  (spread table, spread.rs:165-194)
// What differs is only where the per-block work happens: the block loop of
// lib.rs:180-238 becomes ONE hsw_witness_blocks call (HIP) for all blocks of
// a digest -- or of a whole batch of digests.
//
// SURVEY 8 f4 -- the cells digest() itself allocates around the block loop
// (length constraints lib.rs:122-151, inputs :162-178, is_equal/select :294-310,
// output bytes :311-341) -- are emitted by a context created with
// HSW_GADGET_WHOLE_DIGEST (hsw_frame.hpp / hsw_frame_kernel, assumption A4);
// without it only their *values* are produced (AssignedHashResult).
#ifndef HSW_GADGET_HPP
#define HSW_GADGET_HPP

#include <cstddef>
#include <cstdint>
#include <vector>

#include "../../include/hsw.h"

namespace hsw {

// Value-level result of lib.rs:77-160 for one message.
struct DigestPlan {
    std::vector<uint8_t> blocks;   // padded_inputs[precomputed_input_len..]: max_variable_byte_size bytes
    uint32_t init_state[8];        // INIT_STATE after compress256 over the precomputed prefix (lib.rs:153-160)
    size_t num_round = 0;          // lib.rs:80-84
    size_t precomputed_round = 0;  // lib.rs:93
    size_t target_round = 0;       // num_round - precomputed_round (lib.rs:147-151): index into the state list
    size_t max_variable_round = 0; // lib.rs:87: compressions synthesised, always the maximum
};

// lib.rs:77-117,153-160.  Returns HSW_OK or the status the reference's
// assert!/debug_assert! maps to (HSW_ERR_SHAPE / HSW_ERR_TOO_LARGE).
int digest_prepare(const uint8_t *input, size_t input_byte_size, size_t precomputed_input_len,
                   size_t max_variable_byte_size, DigestPlan *plan);

// lib.rs:31-36 on values, plus where this hash's streams live.
struct AssignedHashResult {
    uint64_t input_len = 0;              // assigned_input_byte_size (lib.rs:124-125)
    std::vector<uint8_t> input_bytes;    // assigned_input_bytes (lib.rs:170-173): the padded variable part
    uint8_t output_bytes[32] = {0};      // lib.rs:311-341
    size_t first_block = 0;              // index of this hash's first block in the context's streams
    size_t n_blocks = 0;                 // max_variable_byte_size / 64
    uint64_t spread_cursor0 = 0;         // SpreadConfig.num_limb_sum when this digest started
    size_t num_round = 0, target_round = 0;
    // whole-digest contexts: where the sections of this digest start (cells)
    uint64_t prologue_cell = 0, block_cell = 0, epilogue_cell = 0, end_cell = 0;
    uint64_t prologue_lookup = 0, block_lookup = 0, epilogue_lookup = 0;
};

class Context;

class Sha256DynamicConfig {
  public:
    std::vector<size_t> max_variable_byte_sizes;   // lib.rs:40
    size_t cur_hash_idx = 0;                       // lib.rs:43
    uint32_t num_bits_lookup = 8;                  // SpreadConfig (spread.rs:24-25)
    uint32_t num_advice_columns = 2;
    bool is_input_range_check = false;             // lib.rs:44 (the 8-bit range checks are halo2-base cells: emitted
                                                   // by whole-digest contexts only)

    // lib.rs:49-69.  HSW_ERR_SHAPE if a size is not a multiple of 64 (lib.rs:57-59)
    // or the spread shape is invalid (spread.rs:37).
    static int configure(const std::vector<size_t> &max_variable_byte_sizes, uint32_t num_bits_lookup,
                         uint32_t num_advice_columns, bool is_input_range_check, Sha256DynamicConfig *out);

    // lib.rs:351-360: a context sized for every hash this config will assign.
    // whole_digest: also lay out digest()'s own cells (HSW_GADGET_WHOLE_DIGEST)
    int new_context(hsw_engine *engine, Context **out, bool whole_digest = false, bool independent = false) const;

    // lib.rs:71-349.  precomputed_input_len = 0 is the reference's None.
    int digest(Context &ctx, const uint8_t *input, size_t input_len, size_t precomputed_input_len,
               AssignedHashResult *result);
    // n consecutive digest() calls with one kernel launch; results[i] as if called in order.
    int digest_batch(Context &ctx, size_t n, const uint8_t *const *inputs, const size_t *input_lens,
                     const size_t *precomputed_input_lens, AssignedHashResult *results);

    // lib.rs:366-368 -> spread.rs:165-194: the (dense, spread) lookup table rows.
    std::vector<std::pair<uint64_t, uint64_t>> load() const;
};

// The Region-owning context of lib.rs:351-360, re-imagined for HBM: it owns the
// device buffers the streams are written to and SpreadConfig's mutable cursor.
class Context {
  public:
    ~Context();
    hsw_engine *engine = nullptr;
    hsw_shape shape{};
    size_t capacity_blocks = 0;      // sum(max_variable_byte_sizes) / 64
    size_t blocks_done = 0;
    uint64_t num_limb_sum = 0;       // SpreadConfig.num_limb_sum (spread.rs:26), starts at 0 (spread.rs:70)
    size_t chip_col_stride = 0;      // rows per chip column buffer
    void *d_gate = nullptr;          // capacity_blocks * G cells
    void *d_chip_dense = nullptr;    // ncols * chip_col_stride cells
    void *d_chip_spread = nullptr;
    uint32_t *d_next_states = nullptr;   // capacity_blocks * 8
    uint8_t *d_blocks = nullptr;         // staging: capacity_blocks * 64
    uint32_t *d_pre_states = nullptr;    // capacity_blocks * 8
    uint32_t *d_init_states = nullptr;   // one per hash in flight
    uint32_t *d_offsets = nullptr;       // first block of every hash in flight (+ 1): hsw_chain_var_kernel
    size_t init_capacity = 0;
    // small batches: pinned, device-mapped host staging the kernels read directly (no H2D copies) and
    // the next states are copied back into (a truly asynchronous D2H): capacity_blocks * (64 + 32 + 32) bytes
    uint8_t *hp_blocks = nullptr;        // host views ...
    uint32_t *hp_pre = nullptr, *hp_next = nullptr;
    uint8_t *dp_blocks = nullptr;        // ... and the device addresses of the same memory
    uint32_t *dp_pre = nullptr, *dp_next = nullptr;
    // compact delivery (hsw_gadget_download_region_compact): 8-byte staging of the streams, side list, its counter
    void *d_c_gate = nullptr, *d_c_lookup = nullptr, *d_c_dense = nullptr, *d_c_spread = nullptr, *d_wide = nullptr;
    uint32_t *d_wide_count = nullptr, *hp_wide_count = nullptr;
    size_t wide_cap = 0;
    uint32_t repr_flags = HSW_REPR_CANONICAL;
    // HSW_GADGET_WHOLE_DIGEST: d_gate is one stream (prologue | zero cell | blocks | epilogue per
    // digest, back to back) and d_lookup the lookup-advice stream next to it
    bool whole = false;
    bool independent = false;        // HSW_GADGET_INDEPENDENT: every digest is a Context of its own (K proofs in flight)
    bool zero_loaded = false;        // Context.zero_cell (first load_zero: compression.rs:34 of the first block)
    uint64_t gate_cursor = 0, gate_capacity = 0;       // cells
    void *d_lookup = nullptr;
    uint64_t lookup_cursor = 0, lookup_capacity = 0;
    // what every digest_batch call wrote, for hsw_gadget_verify
    struct BatchRecord { size_t first_digest, n_digests, first_block, n_blocks; bool inputs_in_pinned; uint32_t repr_flags; };
    std::vector<BatchRecord> batches;
    // FlexGate column image (set_columns): d_gate is `columns` advice columns of max_rows cells;
    // stream cell i sits at i + the gaps of all breaks at or before i (assumption A3-iii)
    uint64_t max_rows = 0, columns = 0;
    std::vector<uint64_t> break_cell, break_gap;
    // Where the caller's halo2-base Context stood when it handed the region to the gadget (hsw_gadget_set_origin;
    // the reference's digest takes whatever Context it is given, lib.rs:71-76,351-360): stream cell 0 lands at
    // (origin_column, origin_row) = ctx.advice_alloc[0]; the Context may already cache its zero cell
    // (ctx.zero_cell, A4-iii: then no digest of this gadget assigns one) and may have queued
    // origin_lookups cells for the lookup-advice column (ctx.cells_to_lookup.len()).  With a column image,
    // image column k is FlexGate column origin_column + k and rows [0, origin_row) of image column 0 are
    // the caller's: never written, never delivered.
    uint64_t origin_column = 0, origin_row = 0, origin_lookups = 0;
    bool origin_zero_loaded = false;
    uint64_t own_lookup_capacity = 0;                  // lookup_capacity - origin_lookups
    int set_origin(uint64_t column, uint64_t row, bool zero_cell_loaded, uint64_t lookups_queued);
    // device address of stream cell 0 (32-byte cells: whole-digest contexts have no compact form)
    void *gate_stream() const {
        return static_cast<uint8_t *>(d_gate) + (size_t)(max_rows ? origin_row : 0) * HSW_CELL_BYTES;
    }
    // Lay the whole-digest stream out as FlexGate (Vertical) advice columns of max_rows usable rows.
    // Only before the first digest.  HSW_ERR_TOO_LARGE: more than HSW_MAX_BREAKS + 1 columns.
    int set_columns(const std::vector<size_t> &max_variable_byte_sizes, bool is_input_range_check, uint64_t max_rows);
    // (column, row) of stream cell i
    void position(uint64_t cell, uint64_t *column, uint64_t *row) const;
    void free_compact_staging();                       // the 8-byte staging follows the geometry: dropped when it changes
};

// The input-independent half of a whole region (hsw_replay.cpp): which cells are NEW witnesses, and for every other
// cell of the gate / lookup / chip streams which witness or constant it repeats.
struct RegionTape;
void free_region_tape(RegionTape *t);
void drop_region_tape_positions(RegionTape *t);

}  // namespace hsw

struct hsw_gadget {
    hsw::Sha256DynamicConfig cfg;
    hsw::Context *ctx = nullptr;
    std::vector<hsw::AssignedHashResult> results;   // one per digest so far (input_bytes kept for queries)
    hsw::RegionTape *tape = nullptr;                // built on first use, dropped when the layout changes
    ~hsw_gadget() { hsw::free_region_tape(tape); }
};
#endif
