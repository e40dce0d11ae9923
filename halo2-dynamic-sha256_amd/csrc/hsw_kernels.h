// hsw_kernels.h -- internal launch interface between the C ABI (hsw_api.cpp)
// and the gfx950 kernels (hsw_expand.hpp, hsw_kernels.hip).  Not part of the public boundary.
#ifndef HSW_KERNELS_H
#define HSW_KERNELS_H

#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

namespace hsw {

enum : uint32_t {
    HSW_K_MONTGOMERY = 1u,   // mirrors HSW_REPR_MONTGOMERY
    HSW_K_SKIP_GATE = 2u,    // mirrors HSW_SKIP_GATE
    HSW_K_SKIP_CHIP = 4u,    // mirrors HSW_SKIP_CHIP
    HSW_K_COMPACT = 8u,      // HSW_REPR_COMPACT64: 8-byte cells
    HSW_K_INTERNALS = 16u,   // engine mode HSW_MODE_HALO2_INTERNALS: range_check cells + lookup stream
    HSW_K_SPLIT = 32u,       // 32 waves per block, each running one phase program (tiny batches: latency)
    HSW_K_CHAINED = 64u,     // small-batch kernel: the blocks are ONE message, pre_states holds its initial state only
    HSW_K_ROLE_MAJOR = 128u, // small-batch kernel: grid = role x block instead of block x role (hsw_small.hpp)
    HSW_K_M32 = 256u,        // with HSW_K_MONTGOMERY, streaming kernel, 8-bit table: Montgomery form built at EMIT time, tiles of
                             // finished 32-byte cells (hsw_expand.hpp Em::M32); else one conversion per cell at write-out
};
enum { HSW_K_MAX_BREAKS = 16 };
enum { HSW_M32_TILE = 8 };                     // cells per tile row of the emit-time Montgomery kernels (Em::M32): [64][8] x 32 B = 16 KiB
enum { HSW_CHAIN_WAVE_MAX_MESSAGES = 2048 };   // chain pre-pass: up to this many messages get a wave each (8 waves per CU), beyond
                                               // that one lane per message has the parallelism anyway
enum { HSW_SMALL_WAVES_PER_BLOCK = 37 };   // roles of the small-batch kernel (hsw_small.hpp)
enum { HSW_SMALL_AUTO_BLOCKS = 128 };      // the engine picks the small-batch kernel up to this many blocks per launch
                                           // (tools/small_n.py: faster than the streaming kernel up to ~190 canonical, ~250 Montgomery)
#ifndef HSW_SMALL_MAX_HELPERS
#define HSW_SMALL_MAX_HELPERS 4            // waves per role (workgroup): the emitter + up to 3 helper waves; a launch
                                           // bound of 512 threads made every wave of the kernel crawl (57 vs 33 us)
#endif

struct ExpandParams {
    const uint8_t *blocks;        // n_blocks * 64 bytes
    const uint32_t *pre_states;   // n_blocks * 8
    void *gate;                   // n_blocks * G cells of 32 B
    void *chip_dense;             // ncols columns, chip_col_stride cells apart
    void *chip_spread;
    uint32_t *next_states;        // n_blocks * 8, may be null
    size_t n_blocks;
    size_t chip_col_stride;       // cells
    uint64_t cursor0;             // SpreadConfig.num_limb_sum before block 0
    uint32_t ncols;               // num_advice_columns
    uint32_t flags;               // HSW_K_*
    uint32_t parts;               // waves per block: 1, 2, 4, 8 or 16
    void *lookup;                 // n_blocks * LOOKUP_CELLS cells (internals mode), may be null
    // FlexGate column packing: gate cell i is written at i + sum of break_gap[k] over break_cell[k] <= i
    uint32_t n_breaks;
    uint64_t break_cell[HSW_K_MAX_BREAKS];
    uint64_t break_gap[HSW_K_MAX_BREAKS];
    // whole-digest streams (internals mode): after every frame_every blocks the gate stream skips
    // frame_cells cells and the lookup stream frame_lookups cells (a digest's epilogue and the
    // next digest's prologue, written by hsw_frame_kernel); frame_every = 0: off
    uint64_t frame_every, frame_cells, frame_lookups;
    uint32_t *next_states_host;   // small-batch kernel only: a second copy of next_states, in pinned host memory (may be null)
    const void *mont_tab;         // HSW_K_M32 only: 3 x 256 Montgomery-form cells -- i, spread(i), i << 8 for i < 256 (hsw_api.cpp)
};

// limbs = 16 / num_bits_lookup.  Returns hipErrorInvalidValue for a limb count
// this build has no instantiation for.
// tile = cells per tile row: 32 (64 units per wave), 64 (32 units, parts >= 2) or
// 128 (16 units, parts >= 4).
hipError_t launch_expand(const ExpandParams &p, int limbs, int tile, hipStream_t stream);
hipError_t launch_chain(const uint8_t *blocks, size_t n_messages, size_t blocks_per_message,
                        const uint32_t *init_states, uint32_t *pre_states, hipStream_t stream);

hipError_t launch_chain_var(const uint8_t *blocks, size_t n_messages, const uint32_t *offsets,
                            const uint32_t *init_states, uint32_t *pre_states, hipStream_t stream);
hipError_t launch_fill(void *dst, size_t bytes, hipStream_t stream);
// 32-byte canonical cells -> 8-byte cells + side list of the cells wider than 64 bits (6 u64 per entry)
hipError_t launch_pack64(const void *src32, void *dst8, size_t n_cells, uint64_t stream_id, uint64_t index0,
                         void *wide, uint32_t wide_cap, uint32_t *wide_count, hipStream_t stream);

// dst[w] = the 32-byte cell at image position pos[w] (distinct-value delivery, hsw_replay.cpp)
hipError_t launch_gather32(const void *image, const uint32_t *pos, void *dst, size_t n, hipStream_t stream);

struct FrameDesc;   // hsw_frame.hpp
struct FrameBreaks;
struct SmallFrames;
// The small-batch kernel (hsw_small.hpp; 8-bit table only): 37 waves per block, one sub-unit program each;
// with `frames` the digest frames are written by extra waves of the same launch.
hipError_t launch_small(const ExpandParams &p, const SmallFrames *frames, int limbs, hipStream_t stream);
// d_inv_tbl: k^-1 mod p for k = 0..(largest n_blocks), 4 x u64 each, in the output representation
hipError_t launch_frames(const FrameDesc *d_descs, size_t n, const uint8_t *blocks, const uint32_t *pre_states,
                         const uint32_t *next_states, const uint64_t *d_inv_tbl, void *gate, void *lookup,
                         const FrameBreaks &brk, unsigned slices, bool montgomery, hipStream_t stream);

}  // namespace hsw
#endif
