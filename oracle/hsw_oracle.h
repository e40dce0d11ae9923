/*
 * hsw_oracle.h -- CPU restatement of the halo2-dynamic-sha256 witness path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product:
 * only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load this library, and there only as the checker / the reported CPU
 * baseline.  The product path (include/hsw.h -> libhsw.so, HIP) never links,
 * loads or calls anything here.
 *
 * What it restates (all citations are paths under the reference tree):
 *   src/compression.rs:19-213   sha256_compression  (call sequence)
 *   src/compression.rs:215-246  state_to_spread_u32
 *   src/compression.rs:266-295  mod_u32
 *   src/compression.rs:297-405  ch
 *   src/compression.rs:460-519  maj
 *   src/compression.rs:521-530  three_add
 *   src/compression.rs:594-882  sigma_upper0/1, sigma_lower0/1, sigma_generic
 *   src/compression.rs:990-1012 ROUND_CONSTANTS, INIT_STATE
 *   src/spread.rs:76-123        SpreadConfig::spread
 *   src/spread.rs:139-163       decompose_even_and_odd_unchecked
 *   src/spread.rs:196-233       spread_limb   (chip column / row placement)
 *   src/utils.rs:6-29           fe_to_bits_le / bits_le_to_fe
 *   src/lib.rs:71-349           Sha256DynamicConfig::digest (padding, prefix
 *                               pre-hash, block loop, output selection)
 *
 * Arithmetic that lives in third-party code absent from the reference tree
 * (halo2-base / halo2-ecc, git github.com/zkmove/halo2-lib rev 40ba7e3;
 * sha2 0.10.6 compress256) is restated from its published semantics:
 *   - field = BN254 scalar field Fr, canonical little-endian 4x64-bit limbs;
 *   - FlexGate vertical gate q*(a + b*c - d) = 0 over 4 consecutive cells,
 *     cell orders (ASSUMPTION A1, halo2-lib v0.2.x):
 *         load_witness(v)  -> [v]
 *         add(a,b)         -> [a, b, 1, a+b]
 *         neg(a)           -> [a, -a, 1, 0]
 *         mul_add(a,b,c)   -> [c, a, b, a*b+c]
 *   - load_zero is cached by the Context and costs no stream cell
 *     (ASSUMPTION A2); assert_equal / range_check add constraints only and
 *     their halo2-base-internal cells are outside the stream.
 *
 * PINNING.  The reference holds no cell-level golden vectors and cannot be
 * built here (no Rust toolchain, un-vendored git deps).  The oracle is pinned
 * by (i) every digest-level known-answer vector of the reference's own tests
 * (src/lib.rs:497-611), (ii) FIPS 180-4 / hashlib on random inputs, and
 * (iii) self-checking every constraint the gadget emits (3,850 assert_equal,
 * every range_check bound, every spread-table lookup per block) -- each
 * witness cell is forced to a unique value by those constraints, so a
 * restatement that satisfies all of them in source order is value-exact.
 * Cell PLACEMENT inside FlexGate columns (halo2-base internals) is NOT pinned:
 * "placement parity unpinned"; see DESIGN.md.
 */
#ifndef HSW_ORACLE_H
#define HSW_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* One field element, canonical (non-Montgomery) little-endian limbs. */
typedef struct { uint64_t l[4]; } ofe_t;

/* Call / constraint tallies of one oracle context (SURVEY 8a counts). */
typedef struct {
    uint64_t load_witness, add, neg, mul_add, load_zero;
    uint64_t assert_equal, range_check16, range_check32, range_check_other;
    uint64_t spread_calls, spread_limb_calls, even_odd_calls;
    uint64_t gate_cells, chip_cells;
} oracle_stats_t;

typedef struct oracle_ctx oracle_ctx;

/* The constraint STRUCTURE of one block, as the reference's source specifies it
 * (QuantumCell::Existing / Constant / Witness per gate input, assert_equal,
 * range_check, spread_limb's constrain_equal): independent of any input, and of
 * any value the oracle computes.  Together with the gate rows of the tape
 * (x0 + x1*x2 = x3) and the two lookup tables this is everything
 * MockProver::verify checks for the path (lib.rs:525-526), so a stream that
 * satisfies it is THE witness for its inputs (uniqueness).
 * Cell ids: >= 0 = block-relative index in the gate stream; negative = cells
 * outside it (below).  Arrays may be NULL to only count. */
#define ORACLE_CELL_ZERO        (-1000)        /* the cached load_zero cell (value 0)        */
#define ORACLE_CELL_INPUT_BYTE0 (-1)           /* input byte k lives in cell -1 - k          */
#define ORACLE_CELL_PRE_STATE0  (-100)         /* pre-state word i lives in cell -100 - i    */
#define ORACLE_CELL_HIDDEN      (-2000)        /* a halo2-base witness that is not in the stream
                                                  (range-check limbs without internals)     */
typedef struct {
    int64_t *eq;     size_t eq_cap, n_eq;         /* pairs (a, b): cells copy-constrained equal          */
    int64_t *konst;  size_t const_cap, n_const;   /* pairs (cell, k): cell fixed to the constant k < 2^64 */
    int64_t *range;  size_t range_cap, n_range;   /* pairs (cell, bits): cell < 2^bits                    */
    int64_t *chip;   size_t chip_cap, n_chip;     /* limb call n: (cell equal to chip dense cell n,
                                                     cell equal to chip spread cell n)                  */
    int64_t *lookup_src; size_t lookup_cap, n_lookup;   /* lookup-column entry j copies this cell        */
    int64_t *next_state_cells;                    /* 8 cells holding the next state words, or NULL        */
} oracle_constraints_t;
/* Record the structure while the NEXT block is expanded (then detaches itself). */
void oracle_record_constraints(oracle_ctx *c, oracle_constraints_t *rec);

/* Create a context for SpreadConfig::configure(num_bits_lookup,
 * num_advice_columns) (spread.rs:32-74).  check!=0 turns on every
 * assert_equal / range / lookup self-check.  Returns NULL on bad shape. */
oracle_ctx *oracle_create(int num_bits_lookup, int num_advice_columns, int check);
void oracle_destroy(oracle_ctx *c);

/* Attach output buffers.  gate: capacity in cells (may be NULL to discard).
 * dense/spread: chip columns, column c at base + c*col_stride cells, buffer
 * row 0 == absolute chip row `row_base`.  May be NULL to discard. */
void oracle_set_outputs(oracle_ctx *c, ofe_t *gate, size_t gate_cap,
                        ofe_t *dense, ofe_t *spread, size_t col_stride,
                        uint64_t row_base);
/* Optional tag per gate-stream cell: 0 = load_witness cell, 1..4 = position
 * inside a 4-cell gate row [x0,x1,x2,x3] whose constraint is x0 + x1*x2 = x3
 * (add: [a,b,1,out]; neg: [a,-a,1,0]; mul_add: [c,a,b,out]).  This is the
 * "tape" a placement adaptor needs, and what the gate-equation property test
 * walks.  Indexed like the gate buffer (reset by oracle_set_outputs). */
void oracle_set_kinds(oracle_ctx *c, uint8_t *kinds, size_t cap);
/* ASSUMPTION A3 (halo2-lib v0.2.x internals; source absent from the reference
 * tree): with internals on, range_check(a, 32) also appends its own 4 cells
 * [limb0, limb1, 2^16, a] to the gate stream at the call position
 * (range_check(a, 16) adds none).  Independently, every looked-up cell is
 * appended to the lookup stream -- the content of the lookup-advice column
 * that RangeConfig::finalize() fills (lib.rs:469), in enable_lookup order. */
void oracle_set_internals(oracle_ctx *c, int on);
void oracle_set_lookup_output(oracle_ctx *c, ofe_t *lookup, size_t cap);
size_t oracle_lookup_len(const oracle_ctx *c);
/* Set SpreadConfig.num_limb_sum (row_offset follows: spread.rs:228-231). */
void oracle_set_cursor(oracle_ctx *c, uint64_t num_limb_sum);
uint64_t oracle_get_cursor(const oracle_ctx *c);
size_t oracle_gate_len(const oracle_ctx *c);
void oracle_get_stats(const oracle_ctx *c, oracle_stats_t *out);
/* 0 if no self-check has failed so far, else nonzero; msg describes the first. */
int oracle_failed(const oracle_ctx *c, const char **msg);

/* compression.rs:19-213.  Appends one block's gate cells and chip cells. */
int oracle_sha256_compression(oracle_ctx *c, const uint8_t block[64],
                              const uint32_t pre_state[8], uint32_t next_state[8]);

/* n independent blocks, each with its own pre-state (batch driver used by the
 * parity tests and by the CPU baseline).  Streams are appended block after
 * block; the chip cursor runs on across blocks exactly as in the reference. */
int oracle_witness_blocks(oracle_ctx *c, const uint8_t *blocks,
                          const uint32_t *pre_states, size_t n,
                          uint32_t *next_states);

/* lib.rs:71-349 restated on values: pads `input`, pre-hashes the first
 * precomputed_input_len bytes with plain SHA-256 (sha2::compress256), runs
 * max_variable_byte_size/64 in-circuit compressions (zero blocks included) and
 * selects the state after round (num_round - precomputed_round).
 * Outputs: digest[32]; padded blocks fed to the circuit (n_blocks*64 bytes,
 * optional); pre/next states per block (optional).  Returns 0, or nonzero if
 * a reference assert would have fired (lib.rs:89-90,94-97,110,114-117). */
int oracle_digest(oracle_ctx *c, const uint8_t *input, size_t input_len,
                  size_t precomputed_input_len, size_t max_variable_byte_size,
                  uint8_t digest[32], uint8_t *blocks_out,
                  uint32_t *pre_states_out, uint32_t *next_states_out);

/* lib.rs:71-349 with EVERY cell digest() itself allocates appended
 * to the gate / lookup streams: prologue (lib.rs:122-178) | the Context's zero
 * cell, at its first use in a context | max/64 blocks | epilogue (lib.rs:294-341).
 * Needs oracle_set_internals(1) (returns 20 otherwise): the frame is made of
 * halo2-base calls whose cells are ASSUMPTION A4 (halo2-lib v0.2.x, unpinned):
 *   mul(a,b) -> [0,a,b,ab]        sub(a,b) -> [a-b,b,1,a]
 *   is_zero(a) -> [z,a,inv,1,0,a,z,0]            (rows at 0 and 4)
 *   is_equal(a,b) -> [a-b,1,b,a] + is_zero        select(a,b,s) -> [a-b,1,b,a,b,s,a-b,out]
 *   is_less_than(a,b,n) -> [a+2^pb-b,b,1,a+2^pb,-2^pb,1,a] (rows at 0 and 3)
 *                          + range_check(.,pb+16) + is_zero(top limb)
 *   range_check(a,8) -> lookup a; [0,a,2^8,a*2^8]; lookup the last cell
 *   load_zero: first call in a Context assigns one cell [0], later calls none.
 * With a constraint recorder attached (oracle_record_constraints) cell ids are
 * ABSOLUTE stream indices and the recorder stays attached for the whole digest;
 * constants that are p - k are recorded as -k. */
typedef struct {
    size_t prologue_cells, zero_cells, block_cells, epilogue_cells;   /* sections of the gate stream, in order */
    size_t prologue_lookups, block_lookups, epilogue_lookups;         /* sections of the lookup stream */
    size_t num_round, target_round, n_blocks;
    int64_t input_len_cell;                                           /* AssignedHashResult.input_len */
} oracle_digest_layout_t;
int oracle_digest_cells(oracle_ctx *c, const uint8_t *input, size_t input_len,
                        size_t precomputed_input_len, size_t max_variable_byte_size,
                        int is_input_range_check, uint8_t digest[32], oracle_digest_layout_t *lay);
/* The state of the halo2-base Context the next oracle_digest_cells call is handed (the reference's
 * digest() takes whatever Context it is given, lib.rs:71-76,351-360): zero_cell_loaded != 0 = the Context
 * already caches its [Constant(0)] cell (ctx.zero_cell.is_some(), A4-iii), so no digest assigns one -- its
 * cell id is then ORACLE_CELL_ZERO (outside the stream).  Where the first cell lands
 * (ctx.advice_alloc[0]) and how many cells are already queued for the lookup column only shift positions:
 * the tests apply them to the tape / lookup stream this oracle returns. */
void oracle_set_context(oracle_ctx *c, int zero_cell_loaded);
/* Optional recorders of the halo2-base call structure: the length of every
 * assign_region call (the replay tape) and the stream index of every enabled
 * gate row (x0 + x1*x2 = x3 on cells [i, i+3]).  Reset by this call. */
void oracle_set_tape(oracle_ctx *c, uint8_t *call_lens, size_t call_cap, uint64_t *gate_rows, size_t rows_cap);
size_t oracle_tape_calls(const oracle_ctx *c);
size_t oracle_tape_rows(const oracle_ctx *c);

/* Plain SHA-256 compression (FIPS 180-4; what sha2::compress256 computes). */
void oracle_plain_compress(uint32_t state[8], const uint8_t block[64]);

/* SpreadConfig::load table row i (spread.rs:165-194): (i, spread(i)). */
uint64_t oracle_spread_table_entry(uint32_t i);

/* In-place canonical -> Montgomery form (x * 2^256 mod p), the in-memory form of
 * halo2curves' Fr; checks HSW_REPR_MONTGOMERY output. */
void oracle_to_montgomery(ofe_t *cells, size_t n);

/* Shape numbers derived by *running* one block (not from formulas). */
int oracle_measure_shape(int num_bits_lookup, int num_advice_columns,
                         uint64_t *gate_cells_per_block,
                         uint64_t *limb_calls_per_block);

#ifdef __cplusplus
}
#endif
#endif
