// In-kernel timeline of small expansion launches (tuning aid; not part of the product).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -DHSW_STAMPS -I../../halo2-dynamic-sha256_amd/csrc probe.hip -o probe
// Every wave stamps the 100 MHz wall clock at: 0 entry, 1 chain done, 2 seeds pulled, 3 rounds phase done,
// 4 exit.  Prints, per launch configuration, the kernel's event-timed duration and the distribution of the
// stamps relative to the earliest entry.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "hsw_small.hpp"

namespace hsw {
template hipError_t launch_expand_L<2>(const ExpandParams &, int, hipStream_t);
template hipError_t launch_expand_m32<2>(const ExpandParams &, hipStream_t);
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__global__ void empty_kernel(int *p) { if (p && threadIdx.x == 1000) *p = 1; }

int main(int argc, char **argv) {
    const size_t NB = argc > 1 ? atoi(argv[1]) : 16;
    const size_t G = hsw::Lay<2, false>::GATE_CELLS, LC = hsw::Lay<2, false>::LIMB_CALLS;
    uint8_t *d_blocks; uint32_t *d_pre, *d_next; void *d_gate, *d_cd, *d_cs; unsigned long long *d_st;
    CK(hipMalloc(&d_blocks, NB * 64)); CK(hipMalloc(&d_pre, NB * 32)); CK(hipMalloc(&d_next, NB * 32));
    CK(hipMalloc(&d_gate, NB * (G + 4000) * 32)); CK(hipMalloc(&d_cd, NB * LC * 32)); CK(hipMalloc(&d_cs, NB * LC * 32));
    CK(hipMalloc(&d_st, NB * 64 * 16 * 8));
    CK(hipMemset(d_blocks, 0x5a, NB * 64)); CK(hipMemset(d_pre, 0x33, NB * 32));
    CK(hipMemcpyToSymbol(HIP_SYMBOL(hsw::g_hsw_stamps), &d_st, sizeof d_st));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    {   // launch floor: an empty kernel between two events
        std::vector<float> ms;
        for (int i = 0; i < 20; i++) {
            CK(hipEventRecord(e0, 0)); hipLaunchKernelGGL(empty_kernel, dim3(512), dim3(64), 0, 0, nullptr); CK(hipEventRecord(e1, 0));
            CK(hipEventSynchronize(e1)); float m; CK(hipEventElapsedTime(&m, e0, e1)); ms.push_back(m);
        }
        std::sort(ms.begin(), ms.end());
        printf("empty kernel (512 waves) between two events: median %.1f us, min %.1f us\n", ms[10] * 1e3, ms[0] * 1e3);
    }
    struct Cfg { const char *name; unsigned parts; unsigned flags; int tile; };
    const Cfg cfgs[] = {{"small-batch kernel (37 roles/block), 1 wave per role", 1, 0, -1},
                        {"small-batch kernel, 2 waves per role", 2, 0, -1},
                        {"small-batch kernel, 3 waves per role", 3, 0, -1},
                        {"small-batch kernel, 4 waves per role", 4, 0, -1},
                        {"small-batch kernel, Montgomery cells, 1 wave per role", 1, hsw::HSW_K_MONTGOMERY, -1},
                        {"small-batch kernel, Montgomery cells, 2 waves per role", 2, hsw::HSW_K_MONTGOMERY, -1},
                        {"small-batch kernel, Montgomery cells, 4 waves per role", 4, hsw::HSW_K_MONTGOMERY, -1},
                        {"split (32 waves/block, one phase each)", 32, hsw::HSW_K_SPLIT, 32},
                        {"16 waves/block", 16, 0, 64}, {"4 waves/block", 4, 0, 64}};
    for (const Cfg &c : cfgs) {
        if (c.tile < 0 && c.parts > HSW_SMALL_MAX_HELPERS) continue;
        hsw::ExpandParams p{};
        p.blocks = d_blocks; p.pre_states = d_pre; p.gate = d_gate; p.chip_dense = d_cd; p.chip_spread = d_cs;
        p.next_states = d_next; p.n_blocks = NB; p.chip_col_stride = NB * LC / 2; p.cursor0 = 0; p.ncols = 2;
        p.flags = c.flags | (c.tile < 0 ? hsw::HSW_K_ROLE_MAJOR : 0u); p.parts = c.parts;
        std::vector<float> ms;
        for (int i = 0; i < 12; i++) {
            CK(hipMemset(d_st, 0, NB * 64 * 16 * 8));
            CK(hipDeviceSynchronize());
            CK(hipEventRecord(e0, 0));
            if (c.tile < 0) CK(hsw::launch_small_L<2>(p, nullptr, 0));
            else CK(hsw::launch_expand_L<2>(p, c.tile, 0));
            CK(hipEventRecord(e1, 0));
            CK(hipEventSynchronize(e1)); float m; CK(hipEventElapsedTime(&m, e0, e1)); ms.push_back(m);
        }
        const unsigned wpb = c.tile < 0 ? 37u : c.parts;       // workgroups per block (wave 0 of each stamps)
        const size_t waves = NB * wpb;
        std::vector<unsigned long long> st(waves * 16);
        CK(hipMemcpy(st.data(), d_st, waves * 16 * 8, hipMemcpyDeviceToHost));
        std::sort(ms.begin(), ms.end());
        unsigned long long t0 = ~0ull;
        for (size_t w = 0; w < waves; w++) t0 = std::min(t0, st[w * 16]);
        printf("%s, %zu blocks: kernel (events) median %.1f us min %.1f us\n", c.name, NB, ms[6] * 1e3, ms[0] * 1e3);
        const char *names[5] = {"entry", "chain done", "seeds pulled", "rounds done", "exit"};
        for (int k = 0; k < 5; k++) {
            std::vector<double> v;
            for (size_t w = 0; w < waves; w++) v.push_back((double)(st[w * 16 + k] - t0) * 0.01);   // 100 MHz -> us
            std::sort(v.begin(), v.end());
            printf("   %-13s min %6.2f  median %6.2f  max %6.2f us\n", names[k], v.front(), v[v.size() / 2], v.back());
        }
        // per-wave durations of the split-mode programs of block 0
        if (c.tile < 0) {
            const size_t NBs = NB;      // the small-batch kernel's grid is slot-major: (block 0, role w) is workgroup (size_t)w * NB
            printf("   block 0, roles 0..32: emission+full flushes / last flush / chip scatter / rest (us)\n   ");
            for (unsigned w = 0; w < 33; w++)
                printf("[%u] %.1f/%.1f/%.1f/%.1f  ", w, (st[w * NBs * 16 + 5] - st[w * NBs * 16 + 2]) * 0.01, (st[w * NBs * 16 + 6] - st[w * NBs * 16 + 5]) * 0.01,
                       (st[w * NBs * 16 + 7] - st[w * NBs * 16 + 6]) * 0.01, (st[w * NBs * 16 + 4] - st[w * NBs * 16 + 7]) * 0.01);
            printf("\n");
        }
        if ((c.flags & hsw::HSW_K_SPLIT) || c.tile < 0) {
            printf("   block 0, per wave: chain / seeds / program (us)\n   ");
            for (unsigned w = 0; w < wpb; w++)
            {
                const size_t x = c.tile < 0 ? (size_t)w * NB : w;
                printf("[%u] %.1f/%.1f/%.1f  ", w, (st[x * 16 + 1] - st[x * 16]) * 0.01, (st[x * 16 + 2] - st[x * 16 + 1]) * 0.01,
                       (st[x * 16 + 4] - st[x * 16 + 2]) * 0.01);
            }
            printf("\n");
        }
    }
    for (int mont = 0; mont < 2; mont++)
    {   // whole-digest launch: one digest of NB blocks (1,024-byte class), frames in the same grid, 9-column breaks
        using LYR = hsw::Lay<2, true>;
        const unsigned SW = (unsigned)std::min<size_t>(16, (192 + 76 * (NB + 1) + 352 + 127) / 128);
        const size_t GR = LYR::GATE_CELLS, PL = hsw::frame::prologue_cells(64 * NB, true), EL = hsw::frame::epilogue_cells(NB);
        const size_t cells = PL + 1 + NB * GR + EL, rows = 131063;
        void *d_gate2, *d_lk; unsigned long long *d_inv;
        CK(hipMalloc(&d_gate2, (cells + 20 * rows) * 32)); CK(hipMalloc(&d_lk, (NB * LYR::LOOKUP_CELLS + 4 * 64 * NB + 128) * 32));
        CK(hipMalloc(&d_inv, 64 * 32)); CK(hipMemset(d_inv, 1, 64 * 32));
        hsw::ExpandParams p{};
        p.blocks = d_blocks; p.pre_states = d_pre; p.gate = (char *)d_gate2 + (PL + 1) * 32; p.chip_dense = d_cd; p.chip_spread = d_cs;
        p.next_states = d_next; p.n_blocks = NB; p.chip_col_stride = NB * LC / 2; p.cursor0 = 0; p.ncols = 2;
        p.flags = hsw::HSW_K_INTERNALS | hsw::HSW_K_ROLE_MAJOR | (mont ? hsw::HSW_K_MONTGOMERY : 0u); p.parts = 4;
        p.lookup = (char *)d_lk + (3 + 2 * 64 * NB) * 32;
        p.frame_every = NB; p.frame_cells = PL + EL; p.frame_lookups = 3 + 2 * 64 * NB + 64;
        hsw::SmallFrames fr{};
        fr.n_frames = 1; fr.state_waves = SW; fr.byte_waves = (unsigned)NB; fr.blocks0 = d_blocks; fr.pre0 = d_pre;
        fr.gate0 = d_gate2; fr.lookup0 = d_lk; fr.inv_tbl = (const uint64_t *)d_inv;
        for (int k = 0; k < 16; k++) fr.brk.cell[k] = ~0ull;
        // column breaks every ~131,063 cells (relative to the block streams for the expansion, absolute for the frames)
        unsigned nb = 0;
        for (size_t c = rows - 50; c < cells && nb < 16; c += rows - 50, nb++) {
            fr.brk.cell[nb] = c; fr.brk.gap[nb] = 50;
            if (c > PL + 1) { p.break_cell[p.n_breaks] = c - (PL + 1); p.break_gap[p.n_breaks] = 50; p.n_breaks++; }
        }
        fr.brk.n = nb;
        fr.d0.input_len = 56; fr.d0.first_block = 0; fr.d0.prologue_cell = 0; fr.d0.epilogue_cell = PL + 1 + NB * GR;
        fr.d0.prologue_lookup = 0; fr.d0.epilogue_lookup = 3 + 2 * 64 * NB + NB * LYR::LOOKUP_CELLS; fr.d0.zero_cell = PL;
        fr.d0.n_blocks = (unsigned)NB; fr.d0.num_round = 2; fr.d0.precomputed_round = 0; fr.d0.range_check_inputs = 1;
        std::vector<float> ms;
        const size_t waves = NB * 37 + SW + NB;
        for (int i = 0; i < 12; i++) {
            CK(hipMemset(d_st, 0, (NB * 64) * 16 * 8));
            CK(hipDeviceSynchronize());
            CK(hipEventRecord(e0, 0));
            CK(hsw::launch_small_L<2>(p, &fr, 0));
            CK(hipEventRecord(e1, 0));
            CK(hipEventSynchronize(e1)); float m; CK(hipEventElapsedTime(&m, e0, e1)); ms.push_back(m);
        }
        std::vector<unsigned long long> st(waves * 16);
        CK(hipMemcpy(st.data(), d_st, waves * 16 * 8, hipMemcpyDeviceToHost));
        std::sort(ms.begin(), ms.end());
        unsigned long long t0 = ~0ull;
        for (size_t w = 0; w < waves; w++) t0 = std::min(t0, st[w * 16]);
        printf("whole digest%s, %zu blocks + frames in one launch (%u breaks): kernel (events) median %.1f us min %.1f us\n", mont ? " (Montgomery cells, 4 waves per role)" : " (4 waves per role)", NB, nb, ms[6] * 1e3, ms[0] * 1e3);
        double mx = 0;
        for (size_t w = 0; w < NB * 37; w++) mx = std::max(mx, (st[w * 16 + 4] - t0) * 0.01);
        printf("   block 0, per wave: chain / seeds / program (us)\n   ");
        for (unsigned w = 0; w < 37; w++)
        {
            const size_t x = (size_t)w * NB;       // slot-major grid
            printf("[%u] %.1f/%.1f/%.1f  ", w, (st[x * 16 + 1] - st[x * 16]) * 0.01, (st[x * 16 + 2] - st[x * 16 + 1]) * 0.01,
                   (st[x * 16 + 4] - st[x * 16 + 2]) * 0.01);
        }
        printf("\n   block %zu, per wave: chain / seeds / program (us)\n   ", NB - 1);
        for (unsigned w = 0; w < 37; w++) {
            const size_t x = (size_t)w * NB + (NB - 1);
            printf("[%u] %.1f/%.1f/%.1f  ", w, (st[x * 16 + 1] - st[x * 16]) * 0.01, (st[x * 16 + 2] - st[x * 16 + 1]) * 0.01,
                   (st[x * 16 + 4] - st[x * 16 + 2]) * 0.01);
        }
        {
            std::vector<std::pair<double, size_t>> ex;
            std::vector<double> en;
            for (size_t w = 0; w < NB * 37; w++) { ex.push_back({(st[w * 16 + 4] - t0) * 0.01, w}); en.push_back((st[w * 16] - t0) * 0.01); }
            std::sort(ex.begin(), ex.end()); std::sort(en.begin(), en.end());
            printf("\n   entry: median %.2f  p90 %.2f  max %.2f us;  latest exits:", en[en.size() / 2], en[en.size() * 9 / 10], en.back());
            for (size_t i = ex.size() - 6; i < ex.size(); i++) {
                const size_t w = ex[i].second;
                printf("  [blk %zu role %zu: entry %.1f chain %.1f prog %.1f exit %.1f]", w % NB, w / NB, (st[w * 16] - t0) * 0.01,
                       (st[w * 16 + 1] - st[w * 16]) * 0.01, (st[w * 16 + 4] - st[w * 16 + 2]) * 0.01, ex[i].first);
            }
        }
        printf("\n   expansion waves: last exit %.2f us\n   state waves (entry / chain+states done / exit): ", mx);
        for (size_t w = NB * 37; w < NB * 37 + SW; w++)
            printf("[%.1f %.1f %.1f] ", (st[w * 16] - t0) * 0.01, (st[w * 16 + 1] - t0) * 0.01, (st[w * 16 + 4] - t0) * 0.01);
        printf("\n   byte waves (entry / exit): ");
        for (size_t w = NB * 37 + SW; w < waves; w++) printf("[%.1f %.1f] ", (st[w * 16] - t0) * 0.01, (st[w * 16 + 4] - t0) * 0.01);
        printf("\n");
    }
    return 0;
}
