// hsw_kernels.hip -- chain pre-pass, calibration fill and launch dispatch.
// The expansion kernel template lives in hsw_expand.hpp and is instantiated per
// table width in hsw_expand_l{1,2,4,8,16}.hip (parallel compilation).
#include "hsw_expand.hpp"

namespace hsw {

// Plain SHA-256 chain of one message: writes the pre-state of each of its
// blocks (what makes blocks independent for the expansion; lib.rs:188,236).
DEV void chain_message(const uint8_t *blocks, size_t first_blk, size_t nblk, const u32 *init, u32 *pre_states) {
    u32 st[8];
#pragma unroll
    for (int i = 0; i < 8; i++) st[i] = init ? init[i] : IV256[i];
    for (size_t j = 0; j < nblk; j++) {
        const size_t blk = first_blk + j;
        const u32 *bw = reinterpret_cast<const u32 *>(blocks + 64 * blk);
#pragma unroll
        for (int i = 0; i < 8; i++) pre_states[8 * blk + i] = st[i];
        u32 w[16];
#pragma unroll
        for (int i = 0; i < 16; i++) w[i] = __builtin_bswap32(bw[i]);
        u32 a = st[0], b = st[1], c = st[2], d = st[3], e = st[4], f = st[5], g = st[6], h = st[7];
#pragma unroll
        for (int t = 0; t < 64; t++) {
            if (t >= 16)
                w[t & 15] = w[t & 15] + sha_s0(w[(t - 15) & 15]) + w[(t - 7) & 15] + sha_s1(w[(t - 2) & 15]);
            const u32 t1 = h + K256[t] + w[t & 15] + sha_S1(e) + sha_ch(e, f, g);
            const u32 t2 = sha_S0(a) + sha_maj(a, b, c);
            h = g; g = f; f = e; e = d + t1; d = c; c = b; b = a; a = t1 + t2;
        }
        st[0] += a; st[1] += b; st[2] += c; st[3] += d;
        st[4] += e; st[5] += f; st[6] += g; st[7] += h;
    }
}

// The same with one WAVE per message, for a few long messages (the reference's bench circuit is one 16-block
// message; K proofs in flight are K of them): the message schedules of up to 32 blocks are expanded side by side --
// they do not depend on the state, lane = block, K[t] + W[t] goes to LDS -- and only the 64-round recurrence runs
// block after block, wave-uniform: ~12 instructions per round instead of ~24 with the schedule in its way
// (a single wave issues one instruction every ~2.1 ns whatever its type: 1.7 us instead of 3.2 us per block).
__global__ __launch_bounds__(64) void hsw_chain_wave_kernel(const uint8_t *blocks, size_t n_messages, size_t bpm,
                                                            const u32 *offsets, const u32 *init_states, u32 *pre_states) {
    constexpr u32 CH = 32;                                 // blocks per chunk
    __shared__ u32 s_kw[CH][64 + 4];                       // +4: rows of consecutive lanes start in different banks
    const size_t m = blockIdx.x;
    if (m >= n_messages) return;
    const u32 lane = threadIdx.x;
    const size_t first = offsets ? offsets[m] : m * bpm;
    const size_t nblk = offsets ? (size_t)(offsets[m + 1] - offsets[m]) : bpm;
    const u32 *init = init_states ? init_states + 8 * m : nullptr;
    u32 st[8];
#pragma unroll
    for (int i = 0; i < 8; i++) st[i] = init ? init[i] : IV256[i];
    for (size_t c0 = 0; c0 < nblk; c0 += CH) {
        const u32 nb = (u32)(nblk - c0 < CH ? nblk - c0 : CH);
        if (lane < nb) {                                   // lane = block: its 64 schedule words, + K
            const u32 *bw = reinterpret_cast<const u32 *>(blocks + 64 * (first + c0 + lane));
            u32 w[16];
#pragma unroll
            for (int i = 0; i < 16; i++) { w[i] = __builtin_bswap32(bw[i]); s_kw[lane][i] = w[i] + K256[i]; }
#pragma unroll
            for (int t = 16; t < 64; t++) {
                w[t & 15] = w[t & 15] + sha_s0(w[(t - 15) & 15]) + w[(t - 7) & 15] + sha_s1(w[(t - 2) & 15]);
                s_kw[lane][t] = w[t & 15] + K256[t];
            }
        }
        __syncthreads();
        for (u32 j = 0; j < nb; j++) {                     // the recurrence: wave-uniform, K + W from LDS four rounds at a time
            if (lane < 8) {
                u32 v = st[0];
#pragma unroll
                for (int i = 1; i < 8; i++) v = lane == (u32)i ? st[i] : v;
                pre_states[8 * (first + c0 + j) + lane] = v;
            }
            u32 a = st[0], b = st[1], c = st[2], d = st[3], e = st[4], f = st[5], g = st[6], h = st[7];
#pragma unroll
            for (int t = 0; t < 64; t += 4) {
                const uint4 kw = *reinterpret_cast<const uint4 *>(&s_kw[j][t]);
                const u32 q[4] = {kw.x, kw.y, kw.z, kw.w};
#pragma unroll
                for (int u = 0; u < 4; u++) {
                    const u32 t1 = h + q[u] + sha_S1(e) + sha_ch(e, f, g);
                    const u32 t2 = sha_S0(a) + sha_maj(a, b, c);
                    h = g; g = f; f = e; e = d + t1; d = c; c = b; b = a; a = t1 + t2;
                }
            }
            st[0] += a; st[1] += b; st[2] += c; st[3] += d;
            st[4] += e; st[5] += f; st[6] += g; st[7] += h;
        }
        __syncthreads();
    }
}

// One thread per message, uniform blocks per message (hsw.h hsw_sha256_chain).
__global__ __launch_bounds__(64) void hsw_chain_kernel(const uint8_t *blocks, size_t n_messages,
                                                       size_t bpm, const u32 *init_states,
                                                       u32 *pre_states) {
    const size_t m = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (m >= n_messages) return;
    chain_message(blocks, m * bpm, bpm, init_states ? init_states + 8 * m : nullptr, pre_states);
}

// Ragged variant: message m owns blocks [offsets[m], offsets[m+1]) (gadget front-end:
// hashes of one circuit may have different max_variable_byte_sizes, lib.rs:40,86).
__global__ __launch_bounds__(64) void hsw_chain_var_kernel(const uint8_t *blocks, size_t n_messages,
                                                           const u32 *offsets, const u32 *init_states,
                                                           u32 *pre_states) {
    const size_t m = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (m >= n_messages) return;
    chain_message(blocks, offsets[m], offsets[m + 1] - offsets[m],
                  init_states ? init_states + 8 * m : nullptr, pre_states);
}

// Calibration fill: every wave streams its own contiguous 64 KiB chunks, 16 B per lane -- the best pure-write pattern
// found on this part (tools/fillbench: 5.7-5.8 TB/s on a 9.77 GB buffer; a grid-stride fill reaches 4.9-5.3).  What
// bench.py reports as `roofline.calibrated_fill_GBps`, next to the 8 TB/s spec and the guide's ~6.3 TB/s achievable.
__global__ __launch_bounds__(64) void hsw_fill_kernel(uint4 *dst, size_t n16, uint4 v) {
    constexpr size_t CHUNK = 4096;                       // 16-byte pieces per chunk (64 KiB)
    const size_t nwaves = gridDim.x;
    for (size_t c = blockIdx.x; c * CHUNK < n16; c += nwaves) {
        const size_t base = c * CHUNK;
        const size_t end = base + CHUNK < n16 ? base + CHUNK : n16;
        for (size_t i = base + threadIdx.x; i < end; i += 64) dst[i] = v;
    }
}

// Compact transport of a finished region (hsw.h hsw_gadget_download_region_compact): cell i of a 32-byte
// canonical stream becomes its low 64 bits; the few cells that do not fit (field negations, -2^16, negative
// differences, is_zero inverses) are appended -- stream id, cell index, all four limbs -- to a side list.
__global__ __launch_bounds__(256) void hsw_pack64_kernel(const uint4 *src, u64 *dst, size_t n_cells, u64 stream_id,
                                                         u64 index0, u64 *wide, u32 wide_cap, u32 *wide_count) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_cells; i += stride) {
        const uint4 lo = src[2 * i], hi = src[2 * i + 1];
        dst[i] = (u64)lo.x | ((u64)lo.y << 32);
        if (lo.z | lo.w | hi.x | hi.y | hi.z | hi.w) {
            const u32 k = atomicAdd(wide_count, 1u);
            if (k < wide_cap) {
                u64 *w = wide + 6 * (size_t)k;
                w[0] = stream_id; w[1] = index0 + i;
                w[2] = (u64)lo.x | ((u64)lo.y << 32); w[3] = (u64)lo.z | ((u64)lo.w << 32);
                w[4] = (u64)hi.x | ((u64)hi.y << 32); w[5] = (u64)hi.z | ((u64)hi.w << 32);
            }
        }
    }
}

hipError_t launch_pack64(const void *src32, void *dst8, size_t n_cells, uint64_t stream_id, uint64_t index0,
                         void *wide, uint32_t wide_cap, uint32_t *wide_count, hipStream_t stream) {
    if (n_cells == 0) return hipSuccess;
    const size_t want = (n_cells + 255) / 256;
    const unsigned grid = (unsigned)(want < 2048 ? want : 2048);
    hipLaunchKernelGGL(hsw_pack64_kernel, dim3(grid), dim3(256), 0, stream, reinterpret_cast<const uint4 *>(src32),
                       reinterpret_cast<u64 *>(dst8), n_cells, stream_id, index0, reinterpret_cast<u64 *>(wide), wide_cap,
                       wide_count);
    return hipGetLastError();
}

// Distinct-value delivery (hsw_replay.cpp): lane pairs fetch the two halves of the new-witness cells of a region,
// in stream order -- consecutive witnesses are a few cells apart, so the reads stay within a few lines.
__global__ __launch_bounds__(256) void hsw_gather32_kernel(const uint4 *img, const u32 *pos, uint4 *dst, size_t n) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < 2 * n; i += stride)
        dst[i] = img[2 * (size_t)pos[i >> 1] + (i & 1)];
}
hipError_t launch_gather32(const void *image, const uint32_t *pos, void *dst, size_t n, hipStream_t stream) {
    if (n == 0) return hipSuccess;
    const size_t want = (2 * n + 255) / 256;
    hipLaunchKernelGGL(hsw_gather32_kernel, dim3((unsigned)(want < 4096 ? want : 4096)), dim3(256), 0, stream,
                       reinterpret_cast<const uint4 *>(image), pos, reinterpret_cast<uint4 *>(dst), n);
    return hipGetLastError();
}

hipError_t launch_fill(void *dst, size_t bytes, hipStream_t stream) {
    const size_t n16 = bytes / 16;
    if (n16 == 0) return hipSuccess;
    uint4 v; v.x = 0x01010101u; v.y = 0; v.z = 0; v.w = 0;
    hipLaunchKernelGGL(hsw_fill_kernel, dim3(256 * 16), dim3(64), 0, stream, reinterpret_cast<uint4 *>(dst), n16, v);
    return hipGetLastError();
}

// ------------------------------------------------------------------ launch
template <int L> hipError_t launch_expand_L(const ExpandParams &p, int tile, hipStream_t stream);
extern template hipError_t launch_expand_L<1>(const ExpandParams &, int, hipStream_t);
extern template hipError_t launch_expand_L<2>(const ExpandParams &, int, hipStream_t);
extern template hipError_t launch_expand_L<4>(const ExpandParams &, int, hipStream_t);
extern template hipError_t launch_expand_L<8>(const ExpandParams &, int, hipStream_t);
extern template hipError_t launch_expand_L<16>(const ExpandParams &, int, hipStream_t);

hipError_t launch_expand(const ExpandParams &p, int limbs, int tile, hipStream_t stream) {
    switch (limbs) {
        case 1: return launch_expand_L<1>(p, tile, stream);
        case 2: return launch_expand_L<2>(p, tile, stream);
        case 4: return launch_expand_L<4>(p, tile, stream);
        case 8: return launch_expand_L<8>(p, tile, stream);
        case 16: return launch_expand_L<16>(p, tile, stream);
        default: return hipErrorInvalidValue;
    }
}

hipError_t launch_chain(const uint8_t *blocks, size_t n_messages, size_t bpm,
                        const uint32_t *init_states, uint32_t *pre_states, hipStream_t stream) {
    if (n_messages == 0 || bpm == 0) return hipSuccess;
    if (n_messages <= HSW_CHAIN_WAVE_MAX_MESSAGES && bpm >= 2) {       // few long messages: a wave each
        hipLaunchKernelGGL(hsw_chain_wave_kernel, dim3((unsigned)n_messages), dim3(64), 0, stream, blocks, n_messages, bpm,
                           static_cast<const u32 *>(nullptr), init_states, pre_states);
        return hipGetLastError();
    }
    const unsigned grid = (unsigned)((n_messages + 63) / 64);
    hipLaunchKernelGGL(hsw_chain_kernel, dim3(grid), dim3(64), 0, stream, blocks, n_messages, bpm,
                       init_states, pre_states);
    return hipGetLastError();
}

hipError_t launch_chain_var(const uint8_t *blocks, size_t n_messages, const uint32_t *offsets,
                            const uint32_t *init_states, uint32_t *pre_states, hipStream_t stream) {
    if (n_messages == 0) return hipSuccess;
    if (n_messages <= HSW_CHAIN_WAVE_MAX_MESSAGES) {                    // (single-block messages lose nothing: ~2 us either way)
        hipLaunchKernelGGL(hsw_chain_wave_kernel, dim3((unsigned)n_messages), dim3(64), 0, stream, blocks, n_messages, (size_t)0,
                           offsets, init_states, pre_states);
        return hipGetLastError();
    }
    const unsigned grid = (unsigned)((n_messages + 63) / 64);
    hipLaunchKernelGGL(hsw_chain_var_kernel, dim3(grid), dim3(64), 0, stream, blocks, n_messages, offsets,
                       init_states, pre_states);
    return hipGetLastError();
}

}  // namespace hsw
