// hsw_frame.hpp -- cell arithmetic of the "frame" Sha256DynamicConfig::digest puts
// around its block loop: the prologue (reference lib.rs:122-178) and
// the epilogue (lib.rs:294-341).  Shared by the host (hsw_frame_query, the tape)
// and the device (hsw_frame_kernel).
//
// Every count follows from the halo2-base calls digest() makes, with their cells
// laid out as halo2-lib v0.2.x does (ASSUMPTION A4 -- source absent from the
// reference tree, unpinned like A1-A3):
//   load_witness(v)      [v]
//   mul(a, b)            [0, a, b, a*b]
//   add(a, b)            [a, b, 1, a+b]
//   sub(a, b)            [a-b, b, 1, a]
//   is_zero(a)           [z, a, inv, 1, 0, a, z, 0]              gate rows at 0 and 4
//   is_equal(a, b)       [a-b, 1, b, a] + is_zero(a-b)
//   select(a, b, s)      [a-b, 1, b, a, b, s, a-b, out]          gate rows at 0 and 4
//   is_less_than_safe(a, 64) with lookup_bits = 16:
//                        range_check(a, 16): lookup a, no cell
//                        [a+2^16-64, 64, 1, a+2^16, -2^16, 1, a]  gate rows at 0 and 3
//                        range_check(., 32): [limb0, limb1, 2^16, .], lookup both limbs
//                        is_zero(limb1)
//   range_check(a, 8)    lookup a; [0, a, 2^8, a*2^8]; lookup the last cell
//   mul_add(a, b, c)     [c, a, b, a*b+c]
//   load_zero            the first call in a Context assigns one cell [0] (here: right
//                        before the first block of the first digest), later calls none
#ifndef HSW_FRAME_HPP
#define HSW_FRAME_HPP

#include <stddef.h>
#include <stdint.h>

namespace hsw {
namespace frame {

// ---- prologue: offsets of the halo2-base calls of lib.rs:122-165 ----
enum : uint32_t {
    P_LEN = 0,        // :124-125  load_witness(input_byte_size)
    P_NROUND = 1,     // :126      load_witness(num_round)
    P_MUL = 2,        // :127-131  mul(num_round, 64)
    P_ADD = 6,        // :132-136  add(input_byte_size, 9)
    P_SUB = 10,       // :137-141  sub(padded_size, input_with_9_size)
    P_LT = 14,        // :142-143  is_less_than_safe: the 7-cell region
    P_RC32 = 21,      //           range_check(shifted, 32)
    P_ISZ = 25,       //           is_zero(limb1)
    P_PRE = 33,       // :145-146  load_witness(precomputed_round)
    P_TGT = 34,       // :147-151  sub(num_round, precomputed_round)
    P_STATE = 38,     // :162-165  8 x load_witness(state word)
    P_BYTES = 46,     // :170-173  max x load_witness(byte); then :174-178 max x range_check(byte, 8)
    P_FIXED_CALLS = 18,   // 1,1,4,4,4,7,4,8,1,4 and eight 1s
    P_FIXED_LOOKUPS = 3,  // padding_size, limb0, limb1
};
// ---- epilogue ----
enum : uint32_t {
    E_STATE = 76,     // per candidate state: is_equal (4 + 8) + 8 x select (8)   :296-310
    E_WORD = 36,      // per output word: 4 x (load_witness + range_check 8) + 4 x mul_add   :311-341
    E_WORDS = 8 * 36,
    E_LOOKUPS = 64,   // 32 output bytes x (byte, byte*2^8)
};

inline uint64_t prologue_cells(uint64_t max_bytes, bool rc) { return P_BYTES + max_bytes * (rc ? 5u : 1u); }
inline uint64_t prologue_lookups(uint64_t max_bytes, bool rc) { return P_FIXED_LOOKUPS + (rc ? 2u * max_bytes : 0u); }
inline uint64_t prologue_calls(uint64_t max_bytes, bool rc) { return P_FIXED_CALLS + max_bytes * (rc ? 2u : 1u); }
inline uint64_t epilogue_cells(uint64_t n_blocks) { return (uint64_t)E_STATE * (n_blocks + 1) + E_WORDS; }
inline uint64_t epilogue_calls(uint64_t n_blocks) { return 10u * (n_blocks + 1) + 8u * 12u; }

}  // namespace frame

// One digest() call, as the frame kernel sees it (device copy of hsw_frame_desc).
struct FrameDesc {
    uint64_t input_len;
    uint64_t first_block;       // into blocks / pre_states / next_states
    uint64_t prologue_cell, epilogue_cell;       // gate-stream cell indices
    uint64_t prologue_lookup, epilogue_lookup;   // lookup-stream indices
    uint64_t zero_cell;         // where this digest loads the Context's zero cell, or ~0
    uint32_t n_blocks;
    uint32_t num_round, precomputed_round;
    uint32_t range_check_inputs;
};

// FlexGate column breaks applied to frame cells (absolute gate-stream indices), like
// ExpandParams::break_cell / break_gap.
struct FrameBreaks {      // entries past n: cell = UINT64_MAX, gap = 0 (the device code walks all 16)
    uint32_t n;
    uint64_t cell[16];
    uint64_t gap[16];
};

// Whole-digest launches of the small-batch kernel (hsw_small.hpp): the frames are written by waves of the
// same grid as the expansion.  Block indices in the descriptors are relative to blocks0 / pre0.
// A frame wave stages its digest's candidate states (n_blocks + 1 of them) in LDS and prefetches the chain inputs
// with four loads per lane: digests of up to this many blocks.  Checked where the kernel is launched (launch_small_L).
constexpr uint32_t SMALL_FRAME_MAX_BLOCKS = 32;
struct SmallFrames {
    const FrameDesc *descs;       // n_frames digests; may live in pinned, device-mapped host memory
    const uint64_t *inv_tbl;      // k^-1 table in the output representation (launch_frames)
    const uint8_t *blocks0;       // block bytes / pre-states the descriptors' first_block indexes
    const uint32_t *pre0;
    void *gate0, *lookup0;        // bases of the whole gate / lookup streams (frame cells are absolute indices)
    uint32_t n_frames;
    uint32_t state_waves;         // waves per digest on the cells that look at state words (they run the last block's chain)
    uint32_t byte_waves;          // waves per digest on the input-byte cells
    uint32_t max_frame_blocks;    // largest n_blocks of any descriptor (host-computed; <= SMALL_FRAME_MAX_BLOCKS)
    FrameBreaks brk;              // column breaks in absolute gate-stream cells
    FrameDesc d0;                 // descs[0] by value: a single digest needs no read of host memory to get going
};

}  // namespace hsw
#endif
