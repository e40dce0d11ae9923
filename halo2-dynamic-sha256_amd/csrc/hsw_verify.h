// hsw_verify.h -- launch interface of the on-device constraint check (hsw_verify.hip).
#ifndef HSW_VERIFY_H
#define HSW_VERIFY_H

#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

namespace hsw {

enum : uint32_t {      // mirrors HSW_VERIFY_* of include/hsw.h
    VERIFY_CONSTANT = 1, VERIFY_COPY = 2, VERIFY_GATE_ROW = 3, VERIFY_ASSERT_EQ = 4, VERIFY_RANGE = 5,
    VERIFY_CHIP = 6, VERIFY_LOOKUP = 7, VERIFY_NEXT_STATE = 8,
};

enum : int64_t { FS_ZERO = -1000, FS_TARGET = -3000, FS_STATE0 = -4000 };   // FrameStructure's external ids

struct VerifyReport {              // device memory
    uint64_t violations;
    uint64_t first_key;            // (block << 36) | (cell << 4) | class of the earliest failure; ~0 = none
    uint64_t pad_;
};

struct VerifyParams {
    // the witness and its inputs (as given to hsw_witness_blocks)
    const void *gate;              // n_blocks * gate_cells canonical cells
    const void *chip_dense, *chip_spread;   // may be null (skipped)
    const void *lookup;            // may be null
    const uint8_t *blocks;
    const uint32_t *pre_states;
    const uint32_t *next_states;   // may be null
    uint64_t cursor0;
    uint64_t chip_col_stride;
    uint32_t ncols, num_bits_lookup;
    uint32_t slices;               // workgroups per block
    uint32_t montgomery;           // cells are x * 2^256 mod p: reduced on load
    // whole-digest streams / column images: block b of the launch starts at stream cell
    // gate_cell0 + b*gate_cells + (b / frame_every)*frame_cells (lookups alike); a stream cell i sits at
    // i + the gaps of all breaks at or before it
    uint64_t gate_cell0, lookup_cell0, frame_every, frame_cells, frame_lookups;
    uint32_t n_breaks;
    uint64_t break_cell[16], break_gap[16];
    // the structure (device copies of hsw::BlockStructure)
    uint32_t gate_cells, n_rows, n_assert_eq, n_range, limb_calls, lookup_cells;
    const uint8_t *kind;
    const int64_t *ref;
    const uint32_t *gate_rows;
    const int64_t *assert_eq, *range, *chip, *lookup_src, *next_state_cells;
    VerifyReport *report;
};

hipError_t launch_verify(const VerifyParams &p, size_t n_blocks, hipStream_t stream);

struct FrameDesc;
// The frames of `n` equally shaped digests (hsw_frame.hpp) against their structure (hsw_structure.hpp:
// FrameStructure of the prologue and of the epilogue, uploaded by the host).
struct FrameVerifyParams {
    const FrameDesc *descs;
    const void *gate, *lookup;     // stream origins (lookup may be null)
    const uint8_t *blocks;
    const uint32_t *pre_states, *next_states;
    uint32_t n_breaks, montgomery;
    uint64_t break_cell[16], break_gap[16];
    struct Section {
        uint32_t cells, n_rows, n_assert_eq, n_assert_const, n_range, n_lookup;
        const uint8_t *kind;
        const int64_t *ref;
        const uint32_t *gate_rows;
        const int64_t *assert_eq, *assert_const, *range, *lookup_src;
    } pro, epi;
    VerifyReport *report;
};
hipError_t launch_verify_frames(const FrameVerifyParams &p, size_t n_digests, hipStream_t stream);

}  // namespace hsw
#endif
