// Timeline of a full-size streaming launch (tuning aid; not part of the product).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -DHSW_STAMPS -I../../halo2-dynamic-sha256_amd/csrc timeline.hip -o timeline
// Every wave stamps the 100 MHz wall clock at entry / chain done / seeds pulled / rounds done / exit and records
// the XCD and CU it ran on.  Printed per configuration: resident waves and waves that have not stored anything
// yet per time bin, how the launch drains (the tail), and the spread over XCDs.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <vector>

#include "hsw_expand.hpp"

namespace hsw {
template hipError_t launch_expand_L<2>(const ExpandParams &, int, hipStream_t);
template hipError_t launch_expand_m32<2>(const ExpandParams &, hipStream_t);
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

int main(int argc, char **argv) {
    const size_t NB = argc > 1 ? atoi(argv[1]) : 4096;
    const size_t G = hsw::Lay<2, false>::GATE_CELLS, LC = hsw::Lay<2, false>::LIMB_CALLS;
    uint8_t *d_blocks; uint32_t *d_pre, *d_next; void *d_gate, *d_cd, *d_cs, *d_tab; unsigned long long *d_st;
    CK(hipMalloc(&d_blocks, NB * 64)); CK(hipMalloc(&d_pre, NB * 32)); CK(hipMalloc(&d_next, NB * 32));
    CK(hipMalloc(&d_gate, NB * G * 32)); CK(hipMalloc(&d_cd, NB * LC * 32)); CK(hipMalloc(&d_cs, NB * LC * 32));
    CK(hipMalloc(&d_tab, 768 * 32)); CK(hipMemset(d_tab, 0x11, 768 * 32));      // timing only: any table content
    const size_t max_waves = NB * 16;
    CK(hipMalloc(&d_st, max_waves * 16 * 8));
    {
        std::vector<uint8_t> hb(NB * 64);
        srand(7);
        for (auto &b : hb) b = (uint8_t)rand();
        CK(hipMemcpy(d_blocks, hb.data(), hb.size(), hipMemcpyHostToDevice));
    }
    CK(hipMemset(d_pre, 0x33, NB * 32));
    CK(hipMemcpyToSymbol(HIP_SYMBOL(hsw::g_hsw_stamps), &d_st, sizeof d_st));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    struct Cfg { const char *name; unsigned parts; unsigned flags; int tile; };
    const Cfg cfgs[] = {{"canonical, [32][64] tiles, 4 waves per block", 4, 0, 64},
                        {"canonical, [64][32] tiles, 1 wave per block", 1, 0, 32},
                        {"canonical, [64][32] tiles, 2 waves per block", 2, 0, 32},
                        {"Montgomery at write-out, [16][128] tiles, 4 waves per block", 4, hsw::HSW_K_MONTGOMERY, 128},
                        {"Montgomery at emit time, [64][8] tiles of 32-byte cells, 1 wave per block", 1, hsw::HSW_K_MONTGOMERY | hsw::HSW_K_M32, 0}};
    for (const Cfg &c : cfgs) {
        hsw::ExpandParams p{};
        p.blocks = d_blocks; p.pre_states = d_pre; p.gate = d_gate; p.chip_dense = d_cd; p.chip_spread = d_cs;
        p.next_states = d_next; p.n_blocks = NB; p.chip_col_stride = NB * LC / 2; p.cursor0 = 0; p.ncols = 2;
        p.flags = c.flags; p.parts = c.parts; p.mont_tab = d_tab;
        std::vector<float> ms;
        for (int i = 0; i < 6; i++) {
            CK(hipMemset(d_st, 0, max_waves * 16 * 8));
            CK(hipDeviceSynchronize());
            CK(hipEventRecord(e0, 0));
            CK(hsw::launch_expand_L<2>(p, c.tile, 0));
            CK(hipEventRecord(e1, 0));
            CK(hipEventSynchronize(e1)); float m; CK(hipEventElapsedTime(&m, e0, e1)); ms.push_back(m);
        }
        const size_t waves = NB * c.parts;
        std::vector<unsigned long long> st(waves * 16);
        CK(hipMemcpy(st.data(), d_st, waves * 16 * 8, hipMemcpyDeviceToHost));
        const float last_ms = ms.back();
        std::sort(ms.begin(), ms.end());
        unsigned long long t0 = ~0ull, t1 = 0;
        for (size_t w = 0; w < waves; w++) { t0 = std::min(t0, st[w * 16]); t1 = std::max(t1, st[w * 16 + 4]); }
        const double span = (double)(t1 - t0) * 0.01;
        printf("\n%s, %zu blocks: kernel (events) median %.1f us, this launch %.1f us; first entry -> last exit %.1f us\n", c.name, NB, ms[3] * 1e3,
               last_ms * 1e3, span);
        // wave lifetimes
        std::vector<double> life, chain;
        for (size_t w = 0; w < waves; w++) { life.push_back((double)(st[w * 16 + 4] - st[w * 16]) * 0.01); chain.push_back((double)(st[w * 16 + 2] - st[w * 16]) * 0.01); }
        std::sort(life.begin(), life.end()); std::sort(chain.begin(), chain.end());
        printf("   wave lifetime us: min %.1f  p10 %.1f  median %.1f  p90 %.1f  max %.1f;  entry -> seeds pulled (no store yet): median %.1f  p90 %.1f\n", life.front(),
               life[waves / 10], life[waves / 2], life[waves * 9 / 10], life.back(), chain[waves / 2], chain[waves * 9 / 10]);
        // timeline in 20 bins
        const int NBIN = 24;
        const double bw = span / NBIN;
        printf("   bin us   resident  not-yet-storing  exits\n");
        for (int b = 0; b < NBIN; b++) {
            const double ta = b * bw + 0.5 * bw;                     // sample at the bin centre
            size_t res = 0, pre = 0, ex = 0;
            for (size_t w = 0; w < waves; w++) {
                const double en = (double)(st[w * 16] - t0) * 0.01, sd = (double)(st[w * 16 + 2] - t0) * 0.01, xt = (double)(st[w * 16 + 4] - t0) * 0.01;
                if (en <= ta && ta < xt) { res++; if (ta < sd) pre++; }
                if (xt >= b * bw && xt < (b + 1) * bw) ex++;
            }
            printf("   %7.1f  %6zu  %6zu  %6zu\n", ta, res, pre, ex);
        }
        // the drain: when does the number of resident waves fall below 75 / 50 / 25 % of its maximum
        {
            std::vector<std::pair<double, int>> ev;
            for (size_t w = 0; w < waves; w++) { ev.push_back({(double)(st[w * 16] - t0) * 0.01, +1}); ev.push_back({(double)(st[w * 16 + 4] - t0) * 0.01, -1}); }
            std::sort(ev.begin(), ev.end());
            int cur = 0, mx = 0;
            for (auto &e : ev) { cur += e.second; mx = std::max(mx, cur); }
            double t75 = 0, t50 = 0, t25 = 0; cur = 0;
            for (auto &e : ev) { cur += e.second; if (cur >= mx * 3 / 4) t75 = e.first; if (cur >= mx / 2) t50 = e.first; if (cur >= mx / 4) t25 = e.first; }
            printf("   resident waves: max %d; last time at >= 75 %% of it: %.1f us, >= 50 %%: %.1f, >= 25 %%: %.1f, empty: %.1f\n", mx, t75, t50, t25, span);
        }
        // per XCD: waves, last exit
        {
            std::map<unsigned, std::pair<size_t, double>> xcd;
            std::map<unsigned, size_t> cu;
            for (size_t w = 0; w < waves; w++) {
                const unsigned long long id = st[w * 16 + 15];
                const unsigned x = (unsigned)(id >> 32) & 15u, hw = (unsigned)id;
                auto &e = xcd[x];
                e.first++; e.second = std::max(e.second, (double)(st[w * 16 + 4] - t0) * 0.01);
                cu[(x << 16) | (hw & 0xff00u)]++;               // cu_id[11:8], sh_id[12], se_id[15:13]
            }
            printf("   per XCD (waves, last exit us):");
            for (auto &e : xcd) printf("  [%u] %zu, %.1f", e.first, e.second.first, e.second.second);
            size_t mn = ~0ull, mxw = 0;
            for (auto &e : cu) { mn = std::min(mn, e.second); mxw = std::max(mxw, e.second); }
            printf("\n   CUs seen: %zu, waves per CU min %zu max %zu\n", cu.size(), mn, mxw);
        }
    }
    return 0;
}
