// hsw_frame.hip -- the cells Sha256DynamicConfig::digest allocates around its block
// loop: prologue lib.rs:122-178, epilogue lib.rs:294-341, and the
// Context's zero cell.  Cell layout: hsw_frame.hpp (assumption A4).
//
// A few hundred cells per digest next to 69,348 per block, so this kernel is
// about being in HBM next to the block streams (no host round trip), not about
// bandwidth: one 256-thread workgroup per digest, every thread computes whole
// halo2-base calls (a select, an is_equal, an output word ...) from the u32 state
// words the expansion kernel left in next_states and stores their cells.
#include "hsw_frame_body.hpp"

namespace hsw {

template <bool MONT>
__global__ __launch_bounds__(256) void hsw_frame_kernel(const FrameDesc *descs, const uint8_t *blocks,
                                                        const u32 *pre_states, const u32 *next_states,
                                                        const u64 *inv_tbl, uint4 *gate, uint4 *lookup,
                                                        FrameBreaks brk, u32 slices) {
    // `slices` workgroups share one digest: all of them stride through the fixed cells, the input bytes
    // and the candidate states (a 16 KiB digest has 82 k byte cells)
    const u32 slice = blockIdx.x % slices;
    const FrameDesc d = descs[blockIdx.x / slices];
    framedev::frame_cells<MONT>(d, blocks, inv_tbl, gate, lookup, brk, framedev::FRAME_ALL, slice * blockDim.x + threadIdx.x,
                                slices * blockDim.x, [&](u32 n, u32 i) -> u32 {
                                    return n == 0 ? pre_states[8 * d.first_block + i]
                                                  : next_states[8 * (d.first_block + n - 1) + i];
                                });
}

hipError_t launch_frames(const FrameDesc *d_descs, size_t n, const uint8_t *blocks, const uint32_t *pre_states,
                         const uint32_t *next_states, const uint64_t *d_inv_tbl, void *gate, void *lookup,
                         const FrameBreaks &brk, unsigned slices, bool montgomery, hipStream_t stream) {
    if (n == 0) return hipSuccess;
    if (slices == 0) slices = 1;
    const dim3 grid((unsigned)(n * slices)), block(256);
    if (montgomery)
        hipLaunchKernelGGL(hsw_frame_kernel<true>, grid, block, 0, stream, d_descs, blocks, pre_states, next_states,
                           reinterpret_cast<const u64 *>(d_inv_tbl), reinterpret_cast<uint4 *>(gate),
                           reinterpret_cast<uint4 *>(lookup), brk, slices);
    else
        hipLaunchKernelGGL(hsw_frame_kernel<false>, grid, block, 0, stream, d_descs, blocks, pre_states, next_states,
                           reinterpret_cast<const u64 *>(d_inv_tbl), reinterpret_cast<uint4 *>(gate),
                           reinterpret_cast<uint4 *>(lookup), brk, slices);
    return hipGetLastError();
}

}  // namespace hsw
