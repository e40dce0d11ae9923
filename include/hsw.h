/*
 * hsw.h -- C ABI of the MI355X SHA-256 witness engine ("halo2 sha witness").
 *
 * This is the drop-in boundary for the hot path of zhmolly/halo2-dynamic-sha256:
 * the reference has NO FFI of its own (it is plain Rust calling halo2-base), so
 * the boundary is new and sits exactly where `Sha256DynamicConfig::digest`
 * calls `sha256_compression` once per 64-byte block:
 *
 *     reference src/lib.rs:180-189          the block loop (caller)
 *     reference src/compression.rs:19-25    sha256_compression(ctx, range,
 *                                           spread_config, bytes[64], pre_state[8])
 *     reference src/spread.rs:196-233       spread_limb (chip column placement)
 *
 * A Rust shim keeps `Sha256DynamicConfig`'s surface, pads/chains on the host,
 * calls hsw_witness_blocks() once for all blocks of a digest (or a batch of
 * digests) and replays the returned streams into `Context`/`Region`
 * (INTEGRATION.md shows the `extern "C"` block).
 *
 * Streams (DESIGN.md "Streams"):
 *   gate cells   per block G cells (hsw_shape.gate_cells_per_block; 66,308 at
 *                the reference's configuration), in the order the reference
 *                issues halo2-base gate calls, 4 cells per add/neg/mul_add gate
 *                and 1 per load_witness.
 *   chip columns the SpreadConfig advice columns denses[c] / spreads[c]
 *                (spread.rs:20-21): limb call #n (counted from
 *                SpreadConfig.num_limb_sum) lands in column n % ncols at row
 *                n / ncols (spread.rs:202-231).
 *   next states  8 u32 words per block (compression.rs:197-212).
 * One cell = 32 bytes = one BN254 scalar-field element, 4 little-endian 64-bit
 * limbs, canonical (HSW_REPR_CANONICAL) or Montgomery (HSW_REPR_MONTGOMERY,
 * the in-memory form of halo2curves' Fr) form; both are built.
 * Beyond the block loop: in HSW_MODE_HALO2_INTERNALS the engine also emits the
 * cells halo2-base / digest() allocate around it ("placement adaptor", "digest
 * frame" below), up to the literal advice-column image of the whole region.
 *
 * All entry points return an int status (HSW_OK = 0); nothing unwinds across
 * the ABI.  Shape violations that the reference would `assert!`/`debug_assert!`
 * on (lib.rs:57-59,89-90; spread.rs:37; compression.rs:26-27) are hard errors.
 */
#ifndef HSW_H
#define HSW_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define HSW_ABI_VERSION 3   /* 3: hsw_gadget_view grew the origin fields; hsw_gadget_set_origin */

/* ---- status codes ---- */
#define HSW_OK                 0
#define HSW_ERR_INVALID_ARG    1   /* NULL / misaligned / inconsistent argument */
#define HSW_ERR_SHAPE          2   /* 16 % num_bits_lookup != 0, ncols == 0, max % 64 != 0 ... */
#define HSW_ERR_NO_DEVICE      3   /* no usable gfx950 device / HIP runtime */
#define HSW_ERR_HIP            4   /* a HIP call failed; see hsw_last_error() */
#define HSW_ERR_UNSUPPORTED    5   /* valid request this build does not implement */
#define HSW_ERR_TOO_LARGE      6   /* message does not fit max_variable_byte_size (lib.rs:90) */
#define HSW_ERR_NOMEM          7

/* ---- flags for hsw_witness_blocks ---- */
#define HSW_REPR_CANONICAL     0u  /* cells hold the canonical integer, LE limbs */
#define HSW_REPR_MONTGOMERY    1u  /* cells hold x*2^256 mod p (halo2curves Fr memory form) */
#define HSW_REPR_COMPACT64    16u  /* transport form: 8-byte cells holding the low 64 bits of the canonical
                                      value.  Every cell of the path fits 64 bits except the field negations
                                      of ch (compression.rs:320-335; 256 cells per block at fixed positions,
                                      hsw_neg_cells): those hold x where the cell's value is -x = p - x.
                                      4x fewer bytes for PCIe-bound consumers; buffers are sized with
                                      hsw_cell_bytes(flags) = 8 instead of 32 */
#define HSW_REPR_MASK          (1u | 16u)
#define HSW_SKIP_GATE          2u  /* do not write the gate stream (d_gate may be NULL) */
#define HSW_SKIP_CHIP          4u  /* do not write the chip columns (pointers may be NULL) */

#define HSW_HOST_REGISTER       8u  /* DEPRECATED (to be removed at the next ABI version).  hsw_witness_blocks_host only.
                                      Accepted and IGNORED since ABI 2: the library
                                      never pins memory it does not own (hipHostRegister on a caller's heap buffers
                                      ended in GPU memory faults twice: a user-pointer registration does not survive
                                      the allocator trimming and re-growing its heap).  Pageable buffers take the
                                      runtime's staged copies; for the PCIe rate allocate with hsw_host_alloc. */
#define HSW_CHAINED            32u  /* hsw_witness_blocks(_ex): the n_blocks blocks are ONE message (lib.rs:180-238) and
                                      d_pre_states holds its initial state only (8 words); every wave derives its
                                      block's pre-state itself, so no chain pre-pass (hsw_sha256_chain) and no second
                                      launch is needed.  Small-batch launches only (<= 32 blocks at the 8-bit table),
                                      HSW_ERR_UNSUPPORTED otherwise.  d_next_states receives every block's output. */

#define HSW_CELL_BYTES         32u

typedef struct hsw_engine hsw_engine;

/* Shape of the streams for SpreadConfig::configure(num_bits_lookup,
 * num_advice_columns) (spread.rs:32-74).  All counts are per 64-byte block. */
typedef struct hsw_shape {
    uint32_t num_bits_lookup;        /* 16 % it == 0 (spread.rs:37) */
    uint32_t num_advice_columns;     /* >= 1 */
    uint32_t limbs_per_spread;       /* 16 / num_bits_lookup (spread.rs:84) */
    uint32_t cells_per_spread;       /* gate cells of one SpreadConfig::spread call */
    uint32_t cells_per_state_spread; /* state_to_spread_u32 (compression.rs:215-246) */
    uint32_t cells_per_sigma;        /* sigma_generic (compression.rs:702-882) */
    uint32_t cells_per_ch;           /* compression.rs:297-405 */
    uint32_t cells_per_maj;          /* compression.rs:460-519 */
    uint32_t cells_per_sched_step;   /* one idx of compression.rs:57-96 */
    uint32_t cells_per_round;        /* one idx of compression.rs:125-196 */
    /* offsets (in cells) of the six regions of one block's gate stream */
    uint32_t off_words;              /* compression.rs:31-47   16 x 16 cells */
    uint32_t off_msg_spread;         /* compression.rs:53-56   16 state_to_spread_u32 */
    uint32_t off_sched;              /* compression.rs:57-96   48 steps */
    uint32_t off_state_spread;       /* compression.rs:109-115 6 state_to_spread_u32 */
    uint32_t off_rounds;             /* compression.rs:125-196 64 rounds */
    uint32_t off_feed;               /* compression.rs:197-212 8 x 10 cells */
    uint32_t gate_cells_per_block;   /* G */
    uint32_t spread_calls_per_block; /* 2,060 */
    uint32_t limb_calls_per_block;   /* spread_calls * limbs_per_spread (cursor advance) */
    uint32_t chip_cells_per_block;   /* 2 * limb_calls_per_block */
    uint64_t algorithmic_bytes_per_block; /* (G + chip cells) * 32 + 64 + 32 + 32 */
    uint32_t mode;                   /* HSW_MODE_* this shape was computed for */
    uint32_t lookup_cells_per_block; /* entries of the lookup-advice column per block (3,184) */
    uint32_t gate_calls_per_block;   /* halo2-base assign_region calls per block (the tape length) */
    uint32_t reserved_;
} hsw_shape;

/* ---- engine modes ---- */
#define HSW_MODE_DEFAULT          0u
/* Also emit the cells halo2-base itself allocates inside the path -- the 4-cell
 * inner product [limb0, limb1, 2^16, a] of every range_check(a, 32) (760 per
 * block => G = 69,348) -- and make the lookup-advice column stream available
 * (what RangeConfig::finalize copies, lib.rs:469: 3,184 cells per block).
 * These follow halo2-lib v0.2.x (DESIGN.md assumption A3); the fork the
 * reference pins is not in its tree, so A3 is unpinned.  Every table width; with the 2- and 1-bit
 * tables only 32-byte cells (canonical / Montgomery), not HSW_REPR_COMPACT64. */
#define HSW_MODE_HALO2_INTERNALS  1u

/* Fill *out for the given SpreadConfig parameters.  Pure host arithmetic. */
int hsw_shape_query(uint32_t num_bits_lookup, uint32_t num_advice_columns, hsw_shape *out);
int hsw_shape_query_ex(uint32_t num_bits_lookup, uint32_t num_advice_columns, uint32_t mode,
                       hsw_shape *out);

/* SpreadConfig::load (spread.rs:165-194): the 2^num_bits_lookup rows
 * (i, spread(i)) of the lookup table, as u64 values.  Either output may be NULL. */
int hsw_spread_table(uint32_t num_bits_lookup, uint64_t *dense_out, uint64_t *spread_out);

/* Bytes per cell for a flags word: 8 with HSW_REPR_COMPACT64, else 32. */
uint32_t hsw_cell_bytes(uint32_t flags);
/* Block-relative gate-stream indices of the cells that hold a field negation
 * (4 per round, compression.rs:320-335), ascending; needed to decode
 * HSW_REPR_COMPACT64.  out may be NULL to query the count (256). */
int hsw_neg_cells(const hsw_shape *shape, uint32_t *out, size_t cap, size_t *n);

/* Number of rows every chip column buffer must hold for n_blocks blocks whose
 * first limb call is #spread_cursor0: buffer row 0 is absolute chip row
 * spread_cursor0 / ncols.  Returns 0 on a bad shape. */
uint64_t hsw_chip_rows(const hsw_shape *shape, uint64_t spread_cursor0, uint64_t n_blocks);

/* Engine bound to one HIP device and stream (hip_stream: a hipStream_t, or
 * NULL for the device's default stream).  One engine per host thread/stream;
 * calls on one engine are issued asynchronously in order on that stream. */
int hsw_engine_create(int device, void *hip_stream, uint32_t num_bits_lookup,
                      uint32_t num_advice_columns, hsw_engine **out);
int hsw_engine_create_ex(int device, void *hip_stream, uint32_t num_bits_lookup,
                         uint32_t num_advice_columns, uint32_t mode, hsw_engine **out);
void hsw_engine_destroy(hsw_engine *e);
int hsw_engine_shape(const hsw_engine *e, hsw_shape *out);
int hsw_engine_synchronize(hsw_engine *e);

/* Replaces n calls of sha256_compression (compression.rs:19-213).
 *
 *   d_blocks      n*64 message bytes            (device memory)
 *   d_pre_states  n*8 u32 pre-state words       (device memory)
 *   spread_cursor0  SpreadConfig.num_limb_sum before the first block
 *                 (spread.rs:26,202); block j starts at cursor0 + j*limb_calls_per_block
 *   d_gate        n*G cells (device memory); must be 16-byte aligned and should be
 *                 32-byte (cell) aligned: at 16 mod 32 every canonical cell straddles
 *                 two sectors and the launch runs ~45 % slower.  Any cell-aligned
 *                 position is fine -- the kernel realigns its write-out to 128-byte
 *                 lines (a few % slower than a line-aligned stream)
 *   d_chip_dense / d_chip_spread
 *                 ncols columns each; column c starts at base + c*chip_col_stride
 *                 cells; hsw_chip_rows() rows are written per column (cells of
 *                 the first/last row that belong to neighbouring calls are left
 *                 untouched when the cursor is not a multiple of ncols)
 *   d_next_states n*8 u32                       (device memory, may be NULL)
 *   flags         HSW_REPR_* | HSW_SKIP_*
 *
 * Asynchronous on the engine's stream. */
int hsw_witness_blocks(hsw_engine *e, const uint8_t *d_blocks, const uint32_t *d_pre_states,
                       size_t n_blocks, uint64_t spread_cursor0, void *d_gate,
                       void *d_chip_dense, void *d_chip_spread, size_t chip_col_stride,
                       uint32_t *d_next_states, uint32_t flags);

/* ------------------------------------------------------------------------
 * Placement adaptor: from streams to FlexGate advice columns.
 * halo2-lib v0.2.x FlexGate (Vertical strategy) fills ONE advice column after
 * another: every assign_region call of `len` cells goes to the current column
 * at the current row, or -- if row + len >= max_rows -- to row 0 of the next
 * column (assumption A3).  So advice columns are the linear gate stream with a
 * gap of unused tail rows at every column break.  hsw_pack_plan_query computes
 * the breaks for n_blocks blocks whose first cell lands at `start_row` of some
 * column; hsw_witness_blocks_ex applies them while writing.
 * ------------------------------------------------------------------------ */
#define HSW_MAX_BREAKS 16
typedef struct hsw_pack_plan {
    uint32_t n_breaks;
    uint32_t columns_touched;            /* 1 + n_breaks */
    uint64_t break_cell[HSW_MAX_BREAKS]; /* linear stream index of the first cell after break k */
    uint64_t break_gap[HSW_MAX_BREAKS];  /* unused tail rows of the column that break k closes */
    uint64_t span_cells;                 /* cells from the first written one to one past the last, gaps included */
    uint64_t end_row;                    /* row after the last cell in the last column (advice_alloc.1) */
} hsw_pack_plan;
int hsw_pack_plan_query(const hsw_shape *shape, size_t n_blocks, uint64_t start_row, uint64_t max_rows,
                        hsw_pack_plan *out);
/* The assign_region call lengths of one block (1 or 4 each), in stream order:
 * the tape a shim walks to replay the stream into halo2-base.  lens_out may be
 * NULL to query the count. */
int hsw_gate_tape(const hsw_shape *shape, uint8_t *lens_out, size_t cap, size_t *n_calls);

/* The constraint STRUCTURE of one block's gate stream (input independent): for every cell which
 * QuantumCell the reference hands to halo2-base there, plus everything else a replayer needs to
 * rebuild the gadget's constraint system around the value streams (what halo2 key generation
 * records): gate rows, assert_equal pairs, range_check bounds, lookup sources, spread-chip ties.
 * Built on the host by walking the reference's call sequence on symbolic cells
 * (csrc/hsw_structure.hpp); follows the engine mode of `shape` (HSW_MODE_HALO2_INTERNALS adds the
 * range_check rows).  Cell ids: >= 0 = block-relative stream index; < 0 = outside the block:
 * HSW_CELL_INPUT_BYTE0 - k (input byte k), HSW_CELL_PRE_STATE0 - i (pre-state word i),
 * HSW_CELL_ZERO (the Context's zero cell), HSW_CELL_HIDDEN (a halo2-base witness not in the stream). */
#define HSW_CELL_INPUT_BYTE0 (-1)
#define HSW_CELL_PRE_STATE0  (-100)
#define HSW_CELL_ZERO        (-1000)
#define HSW_CELL_HIDDEN      (-2000)
#define HSW_KIND_WITNESS  0
#define HSW_KIND_CONSTANT 1   /* cell_ref = the constant */
#define HSW_KIND_EXISTING 2   /* cell_ref = the cell it is copy-constrained to */
typedef struct hsw_structure_counts {
    uint64_t gate_cells, gate_rows, assert_eq, ranges, lookups, limb_calls;
} hsw_structure_counts;
/* Any output pointer may be NULL.  Sizes: cell_kind / cell_ref gate_cells; gate_rows gate_rows (first
 * cell of each row x0 + x1*x2 = x3); assert_eq 2*assert_eq; range 2*ranges (cell, bits); lookup_src
 * lookups; chip 2*limb_calls (cell tied to the dense chip cell, cell tied to the spread chip cell);
 * next_state 8 (the cells holding the block's output words). */
int hsw_block_structure(const hsw_shape *shape, hsw_structure_counts *counts, uint8_t *cell_kind,
                        int64_t *cell_ref, uint32_t *gate_rows, int64_t *assert_eq, int64_t *range,
                        int64_t *lookup_src, int64_t *chip, int64_t *next_state);

/* On-device check of a block witness against the gadget's constraint system -- what MockProver::verify
 * checks for the path (lib.rs:525-526): every gate row, copy constraint (Existing cells, assert_equal),
 * fixed constant, range_check bound, spread-chip cell (tied to its gate cell; (dense, spread) a row of the
 * spread table), lookup-column copy and next-state word, with the block bytes and pre-states entering
 * only through the cells they are copy-constrained to.  No value is recomputed from the inputs, so a
 * stream that passes IS the witness of its inputs (every cell is forced by those constraints).  Takes
 * the buffers and layout of a hsw_witness_blocks(_ex) call (canonical or Montgomery cells -- the latter
 * are reduced on load; pack and frame_* as given
 * there; chip, lookup and next-state pointers may be NULL = not checked).  Synchronous. */
#define HSW_VERIFY_CONSTANT   1u
#define HSW_VERIFY_COPY       2u
#define HSW_VERIFY_GATE_ROW   3u
#define HSW_VERIFY_ASSERT_EQ  4u
#define HSW_VERIFY_RANGE      5u
#define HSW_VERIFY_CHIP       6u
#define HSW_VERIFY_LOOKUP     7u
#define HSW_VERIFY_NEXT_STATE 8u
typedef struct hsw_verify_report {
    uint64_t violations;       /* 0 = the stream satisfies the constraint system */
    uint64_t checks;           /* individual constraints evaluated */
    uint64_t first_block;      /* earliest failure (valid if violations != 0) */
    int64_t first_cell;        /* block-relative gate cell (lookup entry for HSW_VERIFY_LOOKUP) */
    uint32_t first_class;      /* HSW_VERIFY_* */
    float kernel_ms;
} hsw_verify_report;
struct hsw_witness_args;
int hsw_verify_blocks(hsw_engine *e, const struct hsw_witness_args *args, hsw_verify_report *report);
struct hsw_frame_desc;
struct hsw_pack_plan;
/* The same for the digest frames hsw_witness_frames wrote (one call = equally shaped digests: same
 * n_blocks and range-check setting): prologue and epilogue cells against hsw_frame_structure -- full
 * field arithmetic for the rows with full-width cells -- plus the facts and links of each digest (input
 * length, precomputed rounds, initial state, input bytes, pre-state of block b = next state of block
 * b - 1, the epilogue's candidate states).  first_cell: section-relative, bit 30 set for the epilogue. */
int hsw_verify_frames(hsw_engine *e, const struct hsw_frame_desc *descs, size_t n, const uint8_t *d_blocks,
                      const uint32_t *d_pre_states, const uint32_t *d_next_states, const void *d_gate,
                      const void *d_lookup, const struct hsw_pack_plan *pack, uint32_t flags,
                      hsw_verify_report *report);

typedef struct hsw_witness_args {
    const uint8_t *d_blocks;       /* as hsw_witness_blocks */
    const uint32_t *d_pre_states;
    size_t n_blocks;
    uint64_t spread_cursor0;
    void *d_gate;                  /* where stream cell 0 lands: column base + start_row cells; columns are
                                      max_rows cells apart, so hsw_pack_plan.span_cells cells are addressed */
    void *d_chip_dense, *d_chip_spread;
    size_t chip_col_stride;
    uint32_t *d_next_states;
    void *d_lookup;                /* n_blocks * lookup_cells_per_block cells (HSW_MODE_HALO2_INTERNALS), or NULL */
    uint32_t flags;                /* HSW_REPR_* | HSW_SKIP_* */
    const hsw_pack_plan *pack;     /* NULL = plain linear stream */
    /* Whole-digest streams (HSW_MODE_HALO2_INTERNALS; see "digest frame" below): the blocks are
     * those of consecutive digests of frame_every blocks each, and between the block streams of two
     * digests the gate stream skips frame_cells cells and the lookup stream frame_lookups cells (one
     * digest's epilogue + the next one's prologue, written by hsw_witness_frames).  0 = off. */
    uint64_t frame_every, frame_cells, frame_lookups;
} hsw_witness_args;
int hsw_witness_blocks_ex(hsw_engine *e, const hsw_witness_args *args);

/* Plain SHA-256 chain pre-pass (what makes the blocks of one message
 * independent; lib.rs:188,236): message m has blocks_per_message consecutive
 * blocks in d_blocks; its first pre-state is d_init_states[m*8..] (NULL = the
 * FIPS IV, compression.rs:1003-1012).  Writes the pre-state of every block to
 * d_pre_states (n_messages*blocks_per_message*8 u32).  Asynchronous. */
int hsw_sha256_chain(hsw_engine *e, const uint8_t *d_blocks, size_t n_messages,
                     size_t blocks_per_message, const uint32_t *d_init_states,
                     uint32_t *d_pre_states);

/* Host delivery: inputs and outputs are HOST memory.  Stages the inputs H2D, then
 * expands chunks of 128 blocks on the engine's stream into two device staging
 * slots while the previous chunk drains D2H on a second stream (kernel || copy
 * overlap), and synchronizes.  Any output pointer may be NULL (stream skipped).
 * Fastest with pinned output buffers (hsw_host_alloc, or HSW_HOST_REGISTER);
 * PCIe-bound either way: 2.39 MB per block.  (If spread_cursor0 is not a
 * multiple of num_advice_columns the call falls back to one unpipelined pass.)
 * As on the device, chip cells of the first / last row that belong to the
 * neighbouring calls keep what the caller's buffers hold. */
int hsw_witness_blocks_host(hsw_engine *e, const uint8_t *blocks, const uint32_t *pre_states,
                            size_t n_blocks, uint64_t spread_cursor0, void *gate,
                            void *chip_dense, void *chip_spread, size_t chip_col_stride,
                            uint32_t *next_states, uint32_t flags);

/* Pinned (page-locked) host memory for the buffers of hsw_witness_blocks_host. */
int hsw_host_alloc(size_t bytes, void **out);
void hsw_host_free(void *p);

/* Device memory for witness streams: one contiguous virtual range of `bytes` bytes on `device`, backed by separate
 * physical allocations of up to chunk_bytes each (0 = 4 GiB) through HIP's virtual memory management.  On MI355X the
 * write rate of an HBM-bound launch depends on where its output buffers sit; a gate stream in such a
 * range ran 2-4 % faster than in one plain hipMalloc buffer in every process tried, and was much less sensitive to
 * where the chip columns are (tools/vmmprobe).  Any device pointer works with every entry point of this library;
 * this is an offer, not a requirement.  hsw_device_free waits for the device, unmaps and releases the range;
 * HSW_ERR_INVALID_ARG for a pointer that did not come from hsw_device_alloc.  CAUTION (ROCm 7.2): a probe that
 * unmapped and re-created ranges between launches ended in GPU memory faults after a handful of cycles
 * (tools/vmmprobe), allocating and using ranges never did -- allocate them once, free them at shutdown. */
int hsw_device_alloc(int device, size_t bytes, size_t chunk_bytes, void **out);
int hsw_device_free(void *ptr);

/* Duration in milliseconds of the most recent expansion kernel launched by
 * hsw_witness_blocks on this engine, measured with HIP events recorded on the
 * engine's stream around that launch.  Synchronizes on the stop event. */
int hsw_last_kernel_ms(hsw_engine *e, float *ms);
/* Enable/disable the per-launch event pair (off by default: no overhead). */
int hsw_set_timing(hsw_engine *e, int enabled);

/* The stream / device an engine was created on. */
int hsw_engine_stream(const hsw_engine *e, void **hip_stream, int *device);

/* What the most recent expansion launch of this engine was: the kernel instantiation
 * hsw_expand_kernel<limbs, tile_cells, tile_rows, repr, internals> and its work split.  Lets a
 * measurement name the kernel it timed instead of assuming one (bench.py roofline.kernel). */
typedef struct hsw_launch_info {
    uint32_t limbs;        /* 16 / num_bits_lookup */
    uint32_t tile_cells;   /* cells per tile row = contiguous run of one unit */
    uint32_t tile_rows;    /* units one wave expands per phase */
    uint32_t repr;         /* 0 canonical, 1 Montgomery, 2 compact (8-byte cells) */
    uint32_t internals;    /* engine mode HSW_MODE_HALO2_INTERNALS */
    uint32_t parts;        /* waves per block */
    uint32_t split;        /* 0: every wave takes a share of every phase; 1: one phase program per wave;
                              2: one sub-unit program per wave (tiny batches) */
    uint32_t reserved_;
    uint64_t n_blocks;     /* blocks of that launch */
    uint64_t grid;         /* workgroups (= waves) of that launch */
} hsw_launch_info;
int hsw_last_launch(const hsw_engine *e, hsw_launch_info *out);

/* ------------------------------------------------------------------------
 * Gadget front-end: the host side of Sha256DynamicConfig::digest
 * (reference src/lib.rs:71-349) -- SHA-256 padding, zero fill up to the
 * FIXED maximum size, prefix pre-hash, chaining, "select state #n" -- over
 * the engine.  csrc/hsw_gadget.hpp holds the C++ class of the same name.
 * ------------------------------------------------------------------------ */

typedef struct hsw_digest_info {
    size_t num_round;          /* real rounds incl. the precomputed ones (lib.rs:80-84) */
    size_t precomputed_round;  /* lib.rs:93 */
    size_t target_round;       /* num_round - precomputed_round (lib.rs:147-151) */
    size_t n_blocks;           /* max_variable_byte_size / 64: compressions synthesised (lib.rs:87,180) */
} hsw_digest_info;

/* lib.rs:77-117,153-160 on the host, no GPU: pads `input`, returns the
 * max_variable_byte_size bytes fed to the circuit (blocks_out, may be NULL)
 * and the state after the precomputed prefix (init_state_out, may be NULL).
 * HSW_ERR_SHAPE: max or precomputed length not a multiple of 64 (lib.rs:57-59,89);
 * HSW_ERR_TOO_LARGE: padded message does not fit (lib.rs:90). */
int hsw_digest_prepare(const uint8_t *input, size_t input_len, size_t precomputed_input_len,
                       size_t max_variable_byte_size, uint8_t *blocks_out,
                       uint32_t init_state_out[8], hsw_digest_info *info);

/* ------------------------------------------------------------------------
 * Digest frame: the cells Sha256DynamicConfig::digest itself
 * allocates around its block loop -- the prologue (lib.rs:122-178: lengths,
 * is_less_than_safe, the initial state, the input bytes and their optional
 * 8-bit range checks) and the epilogue (lib.rs:294-341: is_equal/select of the
 * state after round #target, the 32 digest bytes with their range checks and
 * recomposition).  With them the gate stream of one digest() call is, in order,
 *     prologue | [the Context's zero cell, at its first use] | n_blocks x G | epilogue
 * and the lookup-advice stream  prologue_lookups | n_blocks x 3,184 | 64.
 * Cell layout of the halo2-base calls involved (mul, sub, is_zero, is_equal,
 * select, is_less_than, range_check(.,8), the first load_zero) follows
 * halo2-lib v0.2.x: ASSUMPTION A4 (csrc/hsw_frame.hpp), unpinned
 * like A1-A3.  Needs an engine in HSW_MODE_HALO2_INTERNALS; canonical or
 * Montgomery cells (a frame has full-width cells, so no COMPACT64).
 * ------------------------------------------------------------------------ */
typedef struct hsw_frame_shape {
    uint64_t n_blocks;            /* max_variable_byte_size / 64 */
    uint64_t prologue_cells;      /* 46 + max (+ 4*max with is_input_range_check) */
    uint64_t epilogue_cells;      /* 76*(n_blocks + 1) + 288 */
    uint64_t prologue_lookups;    /* 3 (+ 2*max) */
    uint64_t epilogue_lookups;    /* 64 */
    uint64_t prologue_calls;      /* assign_region calls (tape entries) */
    uint64_t epilogue_calls;
    uint64_t digest_cells;        /* prologue + n_blocks*G + epilogue; the zero cell is not counted */
    uint64_t digest_lookups;
} hsw_frame_shape;
/* HSW_ERR_SHAPE: max not a multiple of 64 (lib.rs:57-59); HSW_ERR_TOO_LARGE: max above 2^32 bytes
 * (a frame counts its blocks in 32 bits). */
int hsw_frame_query(const hsw_shape *shape, size_t max_variable_byte_size, int is_input_range_check,
                    hsw_frame_shape *out);
/* The assign_region call lengths of the prologue (section 0) or epilogue
 * (section 1), like hsw_gate_tape.  lens_out may be NULL to query the count. */
int hsw_frame_tape(const hsw_shape *shape, size_t max_variable_byte_size, int is_input_range_check,
                   int section, uint8_t *lens_out, size_t cap, size_t *n_calls);

/* Constraint structure of the prologue (section 0) / epilogue (section 1), like hsw_block_structure.
 * Cell ids are section-relative; negative ids name cells of other sections: HSW_CELL_ZERO,
 * HSW_CELL_TARGET (assigned_target_round: prologue cell 34), HSW_CELL_STATE0 - (8 n + i) (word i of
 * candidate state n: n = 0 the prologue's cells 38..45, n >= 1 the next_state cells of block n - 1).
 * assert_const: pairs (cell, k) from assert_is_const.  Constants: -k stands for p - k. */
#define HSW_CELL_TARGET (-3000)
#define HSW_CELL_STATE0 (-4000)
typedef struct hsw_frame_structure_counts {
    uint64_t cells, gate_rows, assert_eq, assert_const, ranges, lookups;
} hsw_frame_structure_counts;
int hsw_frame_structure(const hsw_shape *shape, size_t max_variable_byte_size, int is_input_range_check,
                        int section, hsw_frame_structure_counts *counts, uint8_t *cell_kind, int64_t *cell_ref,
                        uint32_t *gate_rows, int64_t *assert_eq, int64_t *assert_const, int64_t *range,
                        int64_t *lookup_src);

typedef struct hsw_frame_desc {   /* one digest() call */
    uint64_t input_len;           /* lib.rs:77 */
    uint64_t first_block;         /* this digest's first block in d_blocks / d_pre_states / d_next_states */
    uint64_t prologue_cell;       /* gate-stream cell index where its prologue starts */
    uint64_t epilogue_cell;
    uint64_t prologue_lookup;     /* lookup-stream cell indices */
    uint64_t epilogue_lookup;
    uint64_t zero_cell;           /* gate-stream index of the Context's zero cell if this digest is the
                                     first to call load_zero in its Context, else UINT64_MAX */
    uint32_t n_blocks;            /* >= 1 */
    uint32_t num_round;           /* ceil((input_len + 9) / 64), lib.rs:80-84 (checked) */
    uint32_t precomputed_round;   /* lib.rs:93; num_round - precomputed_round <= n_blocks (lib.rs:90) */
    uint32_t is_input_range_check;
} hsw_frame_desc;
/* Writes the frames of n digests.  d_pre_states / d_next_states are the ones the
 * block expansion read / wrote (the candidate states of lib.rs:296 are
 * pre_states[first_block] and next_states[first_block .. first_block+n_blocks-1]);
 * descs is HOST memory.  d_lookup may be NULL.  pack (may be NULL): FlexGate column
 * breaks in the same absolute gate-stream indices as prologue_cell / epilogue_cell;
 * a frame cell at stream index i is written at i + the gaps of all breaks <= i.
 * Asynchronous on the engine's stream, ordered after earlier hsw_witness_blocks calls. */
int hsw_witness_frames(hsw_engine *e, const hsw_frame_desc *descs, size_t n, const uint8_t *d_blocks,
                       const uint32_t *d_pre_states, const uint32_t *d_next_states, void *d_gate,
                       void *d_lookup, const hsw_pack_plan *pack, uint32_t flags);

/* Block streams AND frames of n equally sized digests in one call: what hsw_witness_blocks_ex (with
 * frame_every = the digests' block count) followed by hsw_witness_frames do.  For up to 128 blocks in digests of
 * up to 32 -- the reference's own bench circuit is one 16-block digest -- it is ONE kernel launch: the frame cells are
 * written by extra waves of the expansion's grid, which take the candidate states from the chain inputs
 * (pre-state of block b + 1 = next state of block b) and compute the last block's output themselves, so
 * nothing waits for the expansion.  Larger batches: the two launches above.
 *   blocks            the block streams, exactly as for hsw_witness_blocks_ex
 *   descs, n_digests  as for hsw_witness_frames; first_block / cell indices relative to the *0 bases
 *   frame_pack        column breaks in absolute cells of d_gate0 (blocks.pack: relative to blocks.d_gate)
 *   host_next_states  optional: pinned host memory (hsw_host_alloc), blocks.n_blocks * 8 words; receives a
 *                     copy of the next states without a separate copy launch.  Valid after the stream has
 *                     been synchronized. */
typedef struct hsw_digests_args {
    hsw_witness_args blocks;
    const hsw_frame_desc *descs;
    size_t n_digests;
    const uint8_t *d_blocks0;
    const uint32_t *d_pre_states0;
    const uint32_t *d_next_states0;
    void *d_gate0, *d_lookup0;
    const hsw_pack_plan *frame_pack;
    uint32_t *host_next_states;
} hsw_digests_args;
int hsw_witness_digests(hsw_engine *e, const hsw_digests_args *args);

typedef struct hsw_gadget hsw_gadget;   /* Sha256DynamicConfig + its Context */

typedef struct hsw_hash_result {        /* AssignedHashResult (lib.rs:31-36) on values */
    uint64_t input_len;                 /* lib.rs:124-125 */
    size_t first_block;                 /* this hash's first block in the gadget's streams */
    size_t n_blocks;
    uint64_t spread_cursor0;            /* SpreadConfig.num_limb_sum when this digest began */
    size_t num_round, target_round;
    uint8_t output_bytes[32];           /* lib.rs:311-341 */
    /* HSW_GADGET_WHOLE_DIGEST only (else 0): where this digest's sections start in the
     * gadget's gate / lookup streams, in cells */
    uint64_t prologue_cell, block_cell, epilogue_cell, end_cell;
    uint64_t prologue_lookup, block_lookup, epilogue_lookup;
} hsw_hash_result;

typedef struct hsw_gadget_view {
    void *d_gate;                       /* capacity_blocks * G cells (device) */
    void *d_chip_dense, *d_chip_spread; /* ncols columns, chip_col_stride cells apart, row 0 = chip row 0 */
    uint32_t *d_next_states;
    size_t chip_col_stride;
    size_t blocks_done, capacity_blocks;
    uint64_t num_limb_sum;              /* spread.rs:26 */
    size_t cur_hash_idx;                /* lib.rs:43 */
    /* HSW_GADGET_WHOLE_DIGEST: d_gate is ONE stream of gate_cells cells (of gate_capacity), the
     * digests back to back with their frames; d_lookup the lookup-advice stream.  Else 0 / NULL. */
    uint64_t gate_cells, gate_capacity;
    void *d_lookup;
    uint64_t lookup_cells, lookup_capacity;
    uint64_t max_rows, columns;         /* hsw_gadget_set_columns: d_gate is columns x max_rows cells; else 0 */
    /* hsw_gadget_set_origin (all 0 by default): image column k is FlexGate column origin_column + k, stream
     * cell 0 sits at row origin_row of image column 0, the gadget's lookup entries start at d_lookup cell
     * origin_lookups (lookup_cells / lookup_capacity count from cell 0 of the buffer) */
    uint64_t origin_column, origin_row, origin_lookups;
    uint32_t origin_zero_loaded, reserved_;
} hsw_gadget_view;

/* Sha256DynamicConfig::configure (lib.rs:49-69) + new_context (lib.rs:351-360):
 * allocates HBM for sum(max_variable_byte_sizes)/64 blocks of streams. */
int hsw_gadget_create(hsw_engine *e, const size_t *max_variable_byte_sizes, size_t n_hashes,
                      int is_input_range_check, hsw_gadget **out);
/* flags: HSW_GADGET_WHOLE_DIGEST = also emit the digest frames (engine must be in
 * HSW_MODE_HALO2_INTERNALS): the gadget's gate stream is then every advice cell the
 * reference's digest() calls allocate, in allocation order. */
#define HSW_GADGET_WHOLE_DIGEST 1u
/* With HSW_GADGET_WHOLE_DIGEST: every digest of the gadget is a synthesis OF ITS OWN -- K proofs of one circuit in
 * flight (the reference's bench circuit is one digest per proof, benches/digest.rs:93-129): each digest gets a
 * fresh Context (its own [Constant(0)] cell, A4-iii; its lookup entries and chip rows start at 0 of its own
 * columns), and hsw_gadget_digest_batch still expands all of them in ONE launch.  The K regions lie back to back
 * in the gadget's streams: digest h's gate cells are [prologue_cell, end_cell) and its lookup entries
 * [prologue_lookup, epilogue_lookup + 64) of hsw_hash_result -- cell for cell what a single-digest gadget writes
 * from cell 0 -- and its chip rows are rows [first_block * limb_calls_per_block / ncols, ...) of the chip columns
 * (needs n_blocks * limb_calls_per_block to be a multiple of num_advice_columns: HSW_ERR_UNSUPPORTED otherwise).
 * Linear streams only: hsw_gadget_set_columns / hsw_gadget_set_origin return HSW_ERR_UNSUPPORTED. */
#define HSW_GADGET_INDEPENDENT  2u
int hsw_gadget_create_ex(hsw_engine *e, const size_t *max_variable_byte_sizes, size_t n_hashes,
                         int is_input_range_check, uint32_t flags, hsw_gadget **out);
void hsw_gadget_destroy(hsw_gadget *g);
/* HSW_GADGET_WHOLE_DIGEST, before the first digest: lay the gate stream out as the
 * FlexGate (Vertical) advice columns themselves -- column c = cells [c*max_rows,
 * (c+1)*max_rows) of d_gate, every assign_region call placed by the v0.2.x rule
 * `row + len >= max_rows -> next column` (assumption A3-iii), unassigned tail rows 0.
 * max_rows = the gate's usable rows (RangeConfig.gate.max_rows, lib.rs:355).  The
 * layout depends only on max_variable_byte_sizes, never on the messages.
 * HSW_ERR_TOO_LARGE: more than HSW_MAX_BREAKS + 1 columns. */
int hsw_gadget_set_columns(hsw_gadget *g, uint64_t max_rows, uint64_t *n_columns);
/* HSW_GADGET_WHOLE_DIGEST, before the first digest of a synthesis pass (fresh, or after hsw_gadget_reset;
 * before or after hsw_gadget_set_columns): where the caller's halo2-base Context stands when it hands
 * the region to the gadget.  The reference's digest() works on whatever Context it is given
 * (lib.rs:71-76, 351-360) -- a circuit that has used the gate / range chips before its first digest is
 * the normal case outside the reference's two harnesses.
 *   column, row              ctx.advice_alloc[0]: the FlexGate column in use and its next free row.
 *                            Stream cell 0 lands there and the column breaks follow from it (A3-iii);
 *                            with a column image, image column k = FlexGate column `column + k`, rows
 *                            [0, row) of image column 0 belong to the caller: they are never written by
 *                            the gadget and never touched in the caller's host buffers by the downloads.
 *                            hsw_gadget_cell_position / hsw_gadget_result_cells report FlexGate columns.
 *   zero_cell_loaded         ctx.zero_cell.is_some(): the Context already caches its [Constant(0)] cell
 *                            (A4-iii), so no digest of this gadget assigns one -- the stream is one cell
 *                            shorter and hsw_frame_desc.zero_cell is never set.
 *   lookups_already_queued   ctx.cells_to_lookup.len(): RangeConfig::finalize (lib.rs:469) copies the queue
 *                            into the lookup-advice column in order, so the gadget's entries start at that
 *                            index: d_lookup is reallocated with that many leading cells (zero, the
 *                            caller's), every *_lookup index of hsw_hash_result counts from cell 0.
 * The origin survives hsw_gadget_reset (the next synthesis of the same circuit starts at the same place).
 * HSW_ERR_INVALID_ARG: not a whole-digest context, digests already assigned in this pass, or row >= max_rows;
 * HSW_ERR_TOO_LARGE: the layout from that row needs more than HSW_MAX_BREAKS + 1 columns (the previous
 * origin and layout are kept). */
int hsw_gadget_set_origin(hsw_gadget *g, uint64_t column, uint64_t row, int zero_cell_loaded,
                          uint64_t lookups_already_queued);
/* Start the next synthesis pass with the same buffers and layout: every cursor back to
 * its initial value (cur_hash_idx, num_limb_sum, the stream cursors, the Context's zero
 * cell).  What the reference's harnesses do by cloning the config per synthesis
 * (lib.rs:440, benches/digest.rs:78).  Waits for outstanding work on the engine. */
int hsw_gadget_reset(hsw_gadget *g);
/* Host delivery of what the gadget has written so far (a CPU prover reads advice columns from host
 * memory).  Every pointer may be NULL (skipped).  Layouts equal the device ones: `gate` is the column
 * image (columns x max_rows cells; only the used rows of each column are copied) or, without
 * hsw_gadget_set_columns, the linear stream; `lookup` the lookup-advice stream (whole-digest contexts);
 * `chip_dense` / `chip_spread` ncols columns chip_col_stride cells apart (used rows copied).  One pass
 * of asynchronous copies on the engine's stream, then a synchronize; fastest into pinned memory
 * (hsw_host_alloc).  PCIe-bound: the bench circuit's region is 37 MB. */
typedef struct hsw_region_host {
    void *gate, *lookup, *chip_dense, *chip_spread;
} hsw_region_host;
int hsw_gadget_download_region(hsw_gadget *g, const hsw_region_host *dst);
/* The same delivery in the compact transport form: canonical cells travel as their low 64 bits (a quarter
 * of the bytes over PCIe) and the few cells of a region that do not fit -- the field negations of ch, -2^16,
 * the negative differences and is_zero inverses of the frames -- as a side list of (stream, cell index, four
 * limbs).  Layouts as in hsw_gadget_download_region with 8-byte cells (a column image travels in one piece:
 * the unassigned rows at the end of a column arrive as the zeros they are on the device); `wide` receives up to wide_cap entries
 * (in no particular order), n_wide how many the region has (HSW_ERR_TOO_LARGE if more than wide_cap: a block
 * has at most 256, a frame a few dozen).  hsw_region_widen rebuilds one stream's 32-byte cells on the host.
 * Canonical representation only (Montgomery cells are all full width). */
#define HSW_STREAM_GATE        0u
#define HSW_STREAM_LOOKUP      1u
#define HSW_STREAM_CHIP_DENSE  2u
#define HSW_STREAM_CHIP_SPREAD 3u
typedef struct hsw_wide_cell {
    uint64_t stream;               /* HSW_STREAM_* */
    uint64_t index;                /* cell index in that stream's buffer */
    uint64_t value[4];             /* canonical little-endian limbs */
} hsw_wide_cell;
typedef struct hsw_region_compact {
    uint64_t *gate, *lookup, *chip_dense, *chip_spread;     /* host buffers, 8 bytes per cell; NULL = skipped */
    hsw_wide_cell *wide;
    size_t wide_cap;
    size_t n_wide;                                          /* out */
} hsw_region_compact;
int hsw_gadget_download_region_compact(hsw_gadget *g, hsw_region_compact *dst);
/* Host helper: cells32[i] = compact[i] widened to 32 bytes, then the side-list entries of `stream_id`
 * (indices relative to the same buffer) copied over them. */
int hsw_region_widen(const uint64_t *compact, size_t n_cells, uint64_t stream_id, const hsw_wide_cell *wide,
                     size_t n_wide, void *cells32);
/* ---- distinct-value delivery of a whole region (HSW_GADGET_WHOLE_DIGEST contexts, 32-byte cells) ----
 * About 60 % of a region's cells repeat an earlier cell (QuantumCell::Existing, lookup-column copies, the
 * spread-chip cells tied to gate cells: spread.rs:209-210,226-227) or hold a gate constant, at positions that do not
 * depend on the input.  So only the NEW witnesses need to cross PCIe:
 *   hsw_gadget_region_tape              the input-independent half, built once per layout on the host: for every
 *                                       cell of the gate / lookup / chip streams assigned so far either
 *                                       HSW_TAPE_CONST | k (consts[k]) or the index of a distinct value -- copy
 *                                       chains already resolved, so cell i is simply  value(code[i])
 *   hsw_gadget_download_region_distinct packs the distinct values on the device (one gather launch) and copies
 *                                       them to `distinct` (cap_cells 32-byte cells, ideally pinned: hsw_host_alloc),
 *                                       in the gadget's representation (Montgomery for a halo2 prover)
 *   hsw_gadget_replay_region            rebuilds what hsw_gadget_download_region delivers -- column image (or
 *                                       linear stream), lookup column, chip columns; layouts and the
 *                                       "never touch the caller's cells" rules as there -- from the distinct
 *                                       values, with `threads` host threads (pure host work; a consumer that
 *                                       walks the region cell by cell anyway can read value(code[i]) itself)
 * The tape numbers STREAM cells, so it survives hsw_gadget_reset, a new column height and a new origin column /
 * row / lookup count (a prover that synthesizes one circuit pass after pass builds it once); its pointers stay
 * valid until hsw_gadget_set_origin is given a different zero_cell_loaded or the gadget is destroyed.  HSW_ERR_TOO_LARGE: cap_cells too small (*n_cells says how many), or a region of 2^30 cells or more. */
#define HSW_TAPE_CONST 0x80000000u
typedef struct hsw_region_tape {
    uint64_t n_distinct;                 /* distinct values of the digests assigned so far in this pass */
    uint64_t distinct_capacity;          /* ... of all digests of the gadget: size a reusable buffer with this */
    uint64_t gate_cells, lookup_cells, limb_calls;   /* entries of the code arrays that are assigned so far */
    const uint32_t *gate_code;           /* per gate-stream cell (hsw_gadget_cell_position gives its column / row) */
    const uint32_t *lookup_code;         /* per lookup entry of the gadget (entry j sits at d_lookup cell origin_lookups + j) */
    const uint32_t *chip_dense_code, *chip_spread_code;   /* per limb call n: column n % ncols, row n / ncols */
    const void *consts;                  /* n_consts 32-byte cells in the gadget's current representation */
    uint64_t n_consts;
} hsw_region_tape;
int hsw_gadget_region_tape(hsw_gadget *g, hsw_region_tape *out);
int hsw_gadget_download_region_distinct(hsw_gadget *g, void *distinct, size_t cap_cells, size_t *n_cells);
int hsw_gadget_replay_region(hsw_gadget *g, const void *distinct, const hsw_region_host *dst, unsigned threads);

/* Buffer placement.  On MI355X the same HBM-bound batch runs up to 8 % faster or slower depending on which pair of
 * allocations its gate stream and its chip columns live in, and nothing in user space predicts it.
 * On a fresh or reset gadget: allocate the chip columns up to `candidates` (1..16) times, time the gadget's own batch
 * (every digest an empty message, through hsw_gadget_digest_batch) on each and keep the fastest; the others are
 * freed, the gadget is left reset.  ms_each (candidates floats, may be NULL) receives every candidate's batch time,
 * *kept (may be NULL) the index kept.  Worth calling once for gadgets of a few hundred blocks or more; a
 * latency-bound single digest does not care. */
int hsw_gadget_place(hsw_gadget *g, unsigned candidates, float *ms_each, unsigned *kept);

/* Position the context as if digests #0 .. #hash_idx-1 had already been assigned: every cursor
 * (cur_hash_idx, num_limb_sum, the gate / lookup stream cursors, the zero cell) takes the value it
 * would have then.  All of them follow from max_variable_byte_sizes alone -- never from the
 * messages -- so the digests of one circuit can be dealt to several GPUs (one gadget each, same
 * configuration): rank r seeks to its first digest and assigns its share into the same positions
 * the serial reference would use; no exchange is needed to agree on the layout. */
int hsw_gadget_seek(hsw_gadget *g, size_t hash_idx);
/* Check everything the gadget has written so far against the constraint system, on the device:
 * hsw_verify_blocks on every run of blocks + hsw_verify_frames on their frames (whole-digest contexts;
 * linear stream or column image), or hsw_verify_blocks alone (block-stream contexts).  Canonical or
 * Montgomery cells. */
int hsw_gadget_verify(hsw_gadget *g, hsw_verify_report *report);
/* (column, row) of gate-stream cell `cell` (identity on row without set_columns). */
int hsw_gadget_cell_position(const hsw_gadget *g, uint64_t cell, uint64_t *column, uint64_t *row);
/* Sha256DynamicConfig::digest (lib.rs:71-349); precomputed_input_len 0 = None.
 * Synchronous: returns once the streams of this hash are in HBM. */
int hsw_gadget_digest(hsw_gadget *g, const uint8_t *input, size_t input_len,
                      size_t precomputed_input_len, hsw_hash_result *result);
/* n consecutive digest() calls as ONE kernel launch (results as if sequential). */
int hsw_gadget_digest_batch(hsw_gadget *g, size_t n, const uint8_t *const *inputs,
                            const size_t *input_lens, const size_t *precomputed_input_lens,
                            hsw_hash_result *results);
int hsw_gadget_streams(hsw_gadget *g, hsw_gadget_view *view);
/* Where the cells of digest #hash_idx's AssignedHashResult (lib.rs:31-36, 342-346) sit -- what a shim
 * needs to hand back to the circuit (the reference's TestCircuit constrains output_bytes to its instance
 * column, lib.rs:480-482).  HSW_GADGET_WHOLE_DIGEST contexts only.  Cells are gate-stream indices;
 * positions (column, row) of the same cells in the FlexGate image (identity column 0 without set_columns). */
typedef struct hsw_result_cells {
    uint64_t input_len_cell;            /* load_witness(input_byte_size), lib.rs:124-125 */
    uint64_t input_bytes_cell0;         /* assigned_input_bytes[i] = cell input_bytes_cell0 + i (lib.rs:170-173) */
    uint64_t n_input_bytes;             /* max_variable_byte_size */
    uint64_t output_byte_cells[32];     /* the load_witness cells of the digest bytes (lib.rs:317-324) */
    uint64_t input_len_pos[2];          /* (column, row) */
    uint64_t input_bytes_pos0[2];       /* of input byte 0; byte i follows at i rows below unless a column break
                                           lies in between: hsw_gadget_cell_position(input_bytes_cell0 + i) */
    uint64_t output_byte_pos[32][2];
} hsw_result_cells;
int hsw_gadget_result_cells(const hsw_gadget *g, size_t hash_idx, hsw_result_cells *out);
/* AssignedHashResult.input_bytes of digest #hash_idx (the padded variable part). */
int hsw_gadget_input_bytes(hsw_gadget *g, size_t hash_idx, uint8_t *out, size_t cap, size_t *len);
/* HSW_REPR_CANONICAL (default) or HSW_REPR_MONTGOMERY for subsequent digests. */
int hsw_gadget_set_repr(hsw_gadget *g, uint32_t repr);

/* Synchronous device-to-host copy on the engine's stream (for callers that hold
 * device pointers from hsw_gadget_streams but have no HIP binding of their own). */
int hsw_download(hsw_engine *e, void *host_dst, const void *d_src, size_t bytes);

/* Calibration: overwrites `bytes` of d_buf with a plain 16-byte-per-lane
 * streaming fill and returns its duration -- the practical HBM write ceiling
 * of this device/allocation, which bench.py reports next to the 8 TB/s spec. */
int hsw_fill_calibrate(hsw_engine *e, void *d_buf, size_t bytes, float *ms);

/* Tuning knobs (never change results).  "parts": waves per block, 0 = chosen
 * from the batch size (default), or 1, 2, 4, 8, 16.  "tile": cells per
 * contiguous run of one unit, 0 = chosen by the engine (default), 32, 64 or 128
 * (also 6416 = [16 rows][64 cells], the default of the compact form).  "split": how tiny batches
 * are dealt to waves -- -1 = the small-batch kernel (37 waves per block, one sub-unit program each) for
 * batches of <= 128 blocks (default), 0 = never, 1 = one phase program per wave (32 waves per block),
 * 2 = the small-batch kernel always.  "helpers": waves per workgroup (= role) of the small-batch kernel --
 * they share the write-out of every tile; 0 = chosen by the engine (default: 4 for Montgomery cells, else 4 / 2 / 1
 * up to 16 / 64 / 128 blocks), 1..4.  "verify_slices": workgroups per block of hsw_verify_blocks, 0 = default.
 * "mont_emit": where Montgomery cells of the streaming kernel are converted (8-bit table only) -- 0 = at
 * write-out always, 1 = at emit time for default-mode launches of 1,536 blocks or more (default; fewer VALU
 * instructions than the canonical kernel, but one wave per block), 2 = at emit time always (also internals mode).
 * "chunk_blocks": blocks per kernel launch of a long batch (default and maximum 2^20; a test knob). */
int hsw_engine_set_option(hsw_engine *e, const char *name, int64_t value);

const char *hsw_strerror(int status);
/* Detail of the last failure on this engine ("" if none); never NULL. */
const char *hsw_last_error(const hsw_engine *e);
uint32_t hsw_abi_version(void);

#ifdef __cplusplus
}
#endif
#endif /* HSW_H */
