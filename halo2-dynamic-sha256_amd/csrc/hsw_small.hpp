// hsw_small.hpp -- the expansion kernel for SMALL batches (<= 128 blocks per launch by default: the
// reference's own bench circuit, benches/digest.rs:93,102-109,129, is ONE 56-byte message = 16 blocks).
//
// A 16-block launch writes ~40 MB: it is bound by latency -- launch + the 64-round chain of one block + the
// longest program a wave runs -- not by HBM.  hsw_expand_kernel (hsw_expand.hpp) deals whole units to lanes
// (lane = round), which is the right shape for 4,096 blocks and the wrong one for 16: with 16 waves per
// block on the rounds, 4 of 64 lanes work through a 760-cell straight-line program (17 us measured, after
// 6.4 us of chain and 0.7 us of seed staging through LDS; tools/latency_probe).  Here instead:
//
//  * a workgroup = one ROLE: a SUB-UNIT program (a round is six of them -- Sigma1 | ch, first half | ch, second
//    half + T1 | Sigma0 | maj + T2 | the two state updates, independent once the chain seeds are known,
//    compression.rs:125-196; a schedule step is three) over 16 consecutive units, lane = unit.  37 workgroups
//    per block, the longest program is a sigma's 138 cells instead of a round's 760, [16 rows][128 cells] tiles.
//  * up to 4 waves per workgroup (Em::HELPERS, hsw_expand.hpp): wave 0 emits, and all of them share the
//    conversion and write-out of every tile -- that, not the emission, is most of a role's program.
//  * the chain is recomputed by every emitter wave, wave-uniform, but only as far as the wave needs it (rounds
//    0..15 need 15 rounds of it, schedule waves none), straight through registers: lane l LATCHES the a / e /
//    W value born at its own index (one compare + select per value) and fetches its neighbours' with
//    ds_bpermute -- no LDS staging, no stores, and the bit operations are v_bitop3 / v_add3.
//  * whole-digest launches (engine mode HSW_MODE_HALO2_INTERNALS): the digest frame (hsw_frame_body.hpp) is
//    written by extra waves of the SAME launch; they take the candidate states from the chain inputs
//    (pre-state of block b + 1 = next state of block b) and compute the last block's output themselves.
//
// Every cell comes out of the same gate functions as in hsw_expand_kernel (sigma_generic, ch_gadget, ...), so
// the streams are identical bit for bit (tests: test_small_kernel_gives_identical_streams).
#ifndef HSW_SMALL_HPP
#define HSW_SMALL_HPP
#include "hsw_expand.hpp"
#include "hsw_frame_body.hpp"

namespace hsw {

// Sub-units of one round / one schedule step: cells, spread calls and lookup entries, in stream order.
template <int L, bool RC>
struct SmallPlan {
    using LY = Lay<L, RC>;
    static constexpr int CH_A = 40 + 2 * (2 * LY::S + 4);    // ch_part_a: 2 add, 2 neg, 4 add, 8 witnesses, 2 re-checks
    static constexpr int CH_B = LY::CH - CH_A;               // ch_part_b: 2 re-checks, 2 add, 1 mul_add
    // round: Sigma1(e) | ch A | ch B + 4 add + mod_u32 (T1) | Sigma0(a) | maj + add + mod_u32 (T2) | 2 x (add + mod_u32 + s2s)
    static constexpr int NR = 6;
    static constexpr int R_CELLS[NR] = {LY::SIGMA, CH_A, CH_B + 16 + LY::MOD, LY::SIGMA, LY::MAJ + 4 + LY::MOD,
                                        2 * (4 + LY::MOD + LY::S2S)};
    static constexpr int R_CALLS[NR] = {LY::CALLS_SIGMA, 4, 4, LY::CALLS_SIGMA, 4, 2 * LY::CALLS_S2S};
    static constexpr int R_LK[NR] = {LY::LK_SIGMA, LY::LK_CH, LY::LK_MOD, LY::LK_SIGMA, LY::LK_MAJ + LY::LK_MOD, 2 * LY::LK_MOD};
    // schedule step: sigma1(W[i-2]) | sigma0(W[i-15]) | 3 add + mod_u32 + s2s
    static constexpr int NS = 3;
    static constexpr int S_CELLS[NS] = {LY::SIGMA, LY::SIGMA, 12 + LY::MOD + LY::S2S};
    static constexpr int S_CALLS[NS] = {LY::CALLS_SIGMA, LY::CALLS_SIGMA, LY::CALLS_S2S};
    static constexpr int S_LK[NS] = {LY::LK_SIGMA, LY::LK_SIGMA, LY::LK_MOD};
    static constexpr int off(const int *a, int k) { int s = 0; for (int i = 0; i < k; i++) s += a[i]; return s; }
    static_assert(off(R_CELLS, NR) == LY::ROUND && off(R_CALLS, NR) == LY::CALLS_ROUND && off(R_LK, NR) == LY::LK_ROUND, "round sub-units");
    static_assert(off(S_CELLS, NS) == LY::SCHED && off(S_CALLS, NS) == LY::CALLS_SCHED && off(S_LK, NS) == LY::LK_SCHED, "schedule sub-units");
    static constexpr int MAX_CALLS = 4;   // spread calls of the largest sub-unit
    static constexpr int MAX_LK = 8;
};

// Roles of the workgroups of one block.
enum : u32 {
    SMALL_ROUND_ROLES = 24,   // role = type * 4 + (3 - range): 6 sub-unit types x 4 ranges of 16 rounds
    SMALL_SCHED_ROLES = 9,    // role - 24 = type * 3 + (2 - range): 3 sub-unit types x 3 ranges of 16 steps
    SMALL_ROLE_FEED = 33, SMALL_ROLE_WORDS = 34, SMALL_ROLE_MSG = 35, SMALL_ROLE_STATE = 36,
    SMALL_ROLES = 37,
    SMALL_ROWS = 16, SMALL_TILE = 128,
};
static_assert(SMALL_ROLES == HSW_SMALL_WAVES_PER_BLOCK, "hsw_kernels.h");
// The plain SHA-256 recurrence of one block, wave-uniform, through registers.  Index convention: A_t / E_t
// are the a / e words at the START of round t (A_0 = a, A_-1 = b, A_-2 = c, A_-3 = d of the pre-state; round
// t works on a..d = A_t..A_t-3, e..h = E_t..E_t-3), W_t the schedule word of round t.  Runs rounds
// 0 .. steps-1 (steps: wave-uniform, rounded up to 16) and leaves in
//   lA, lE   A_tl / E_tl of this lane's own index tl_state (-3 .. 64)
//   lW       W_tl of this lane's own index tl_w (0 .. 63)
// -- one v_cmp + v_cndmask per value and round instead of staging every value in LDS.  ROUNDS = false:
// the message schedule only.
template <bool ROUNDS>
DEV void chain_latch(const u32 *bw, const u32 *ps, int steps, int tl_state, int tl_w, u32 &lA, u32 &lE, u32 &lW) {
    u32 w[16];
#pragma unroll
    for (int j = 0; j < 16; j++) w[j] = __builtin_bswap32(bw[j]);            // big-endian words (compression.rs:31-47)
    u32 a = 0, b = 0, c = 0, d = 0, e = 0, f = 0, g = 0, h = 0;
    lA = lE = lW = 0;
    if constexpr (ROUNDS) {
        a = ps[0]; b = ps[1]; c = ps[2]; d = ps[3]; e = ps[4]; f = ps[5]; g = ps[6]; h = ps[7];
        lA = tl_state == 0 ? a : tl_state == -1 ? b : tl_state == -2 ? c : d;
        lE = tl_state == 0 ? e : tl_state == -1 ? f : tl_state == -2 ? g : h;
    }
    for (int base = 0; base < steps; base += 16) {                           // wave-uniform trip count
        const int dw = tl_w - base, ds = tl_state - base;
#pragma unroll
        for (int j = 0; j < 16; j++) {
            if (base != 0)                                                   // W_t, t >= 16, into ring slot t & 15
                w[j] = w[j] + sha_s0(w[(j + 1) & 15]) + w[(j + 9) & 15] + sha_s1(w[(j + 14) & 15]);
            lW = dw == j ? w[j] : lW;
            if constexpr (ROUNDS) {
                const u32 t1 = h + K256[base + j] + w[j] + sha_S1(e) + sha_ch(e, f, g);
                const u32 t2 = sha_S0(a) + sha_maj(a, b, c);
                h = g; g = f; f = e; e = d + t1; d = c; c = b; b = a; a = t1 + t2;
                lA = ds == j + 1 ? a : lA;                                   // A_(base+j+1)
                lE = ds == j + 1 ? e : lE;
            }
        }
    }
}

DEV u32 lane_get(u32 v, u32 src_lane) { return (u32)__builtin_amdgcn_ds_bpermute((int)(src_lane << 2), (int)v); }

// A sub-unit phase: row r of the tile is unit unit_lo + r, its cells start sub_off cells into the unit; the
// spread calls / lookup entries it stages are scattered to their places afterwards (sub_end).
template <class EM>
DEV void sub_begin(EM &em, u32 unit_lo, u32 nrows, u32 unit_cells, u32 phase_off, u32 sub_off, u32 sub_calls,
                   u32 sub_lk) {
    const u32 lane = lane_id();
    em.nrows = nrows;
    em.unit_cells = unit_cells;                     // row stride in the stream
    em.cell_base = phase_off + unit_lo * unit_cells + sub_off;
    em.active = lane < nrows;
    const u32 r = lane < nrows ? lane : nrows - 1u;
    em.unit = unit_lo + r;
    em.call = r * sub_calls;
    em.lk = r * sub_lk;
    em.call_first = 0; em.calls = 0; em.lk_first = 0; em.lks = 0;      // unused: the scatter below replaces flush_chip / flush_lookup
    em.skew = 0; em.carry_neg = 0;
    em.row = em.row0;
}

// Chip cells of the spread calls a sub-unit phase staged (spread.rs:196-233): row r's calls are the block's
// calls call0 + r * calls_per_unit + [0, sub_calls); limb call n lands in column n % ncols at row n / ncols.
template <int L, class EM>
DEV void scatter_chip(const EM &em, const ExpandParams &p, u64 block_first_limb, u32 call0, u32 calls_per_unit,
                      u32 sub_calls) {
    constexpr int B = 16 / L;
    constexpr u32 MASK = (1u << B) - 1u;
    if (sub_calls == 0 || (p.flags & HSW_K_SKIP_CHIP)) return;
    em_sync<EM>();                                                     // d16 staged by all lanes
    const u32 lane = lane_id();
    const u32 per_row = sub_calls * (u32)L, total = em.nrows * per_row;
    // one 64-bit division per wave (scalar), 32-bit ones per limb: limb n0 + x sits in column (c0 + x) % ncols
    // at row r0 + (c0 + x) / ncols, x < 2^17
    const u64 n0 = block_first_limb + (u64)call0 * L;
    const u64 r0 = n0 / p.ncols;
    const u32 c0 = (u32)(n0 - r0 * p.ncols), ncols = p.ncols;
    const size_t rbase = (size_t)(r0 - p.cursor0 / p.ncols);
    constexpr u32 CB = EM::COMPACT ? 8u : 32u;
    const u32 hs = em.hsel, hn = em.hcnt;             // all waves of the workgroup, the emitter included
    for (u32 k = lane + 64u * hs; k < total; k += 64u * hn) {
        const u32 r = k / per_row, j = k - r * per_row;
        const u32 limb = ((u32)em.d16[r * sub_calls + j / (u32)L] >> (B * (j % (u32)L))) & MASK;
        const u32 x = c0 + r * calls_per_unit * (u32)L + j;
        const u32 row = x / ncols;
        const size_t cell = (size_t)(x - row * ncols) * p.chip_col_stride + rbase + row;
        char *cd = reinterpret_cast<char *>(p.chip_dense) + cell * CB;
        char *cs = reinterpret_cast<char *>(p.chip_spread) + cell * CB;
        if constexpr (EM::COMPACT) {
            store8(cd, 0, limb);
            store8(cs, 0, spread16(limb));
        } else if constexpr (EM::MONT_OUT) {
            const Fe8 md = mont_from_u64<false>(limb, 0), ms = mont_from_u64<false>(spread16(limb), 0);
            store16(cd, 0, make_uint4(md.l[0], md.l[1], md.l[2], md.l[3]));
            store16(cd, 16, make_uint4(md.l[4], md.l[5], md.l[6], md.l[7]));
            store16(cs, 0, make_uint4(ms.l[0], ms.l[1], ms.l[2], ms.l[3]));
            store16(cs, 16, make_uint4(ms.l[4], ms.l[5], ms.l[6], ms.l[7]));
        } else {
            store16(cd, 0, make_uint4(limb, 0u, 0u, 0u));
            store16(cd, 16, make_uint4(0u, 0u, 0u, 0u));
            store16(cs, 0, make_uint4(spread16(limb), 0u, 0u, 0u));
            store16(cs, 16, make_uint4(0u, 0u, 0u, 0u));
        }
    }
    em_sync<EM>();
}

// Lookup-advice entries of a sub-unit phase (RC only): row r's are lk0 + r * lk_per_unit + [0, sub_lk).
template <class EM>
DEV void scatter_lookup(const EM &em, const ExpandParams &p, size_t lookup_block_base, u32 lk0, u32 lk_per_unit,
                        u32 sub_lk) {
    if (sub_lk == 0 || p.lookup == nullptr) return;
    em_sync<EM>();
    const u32 lane = lane_id();
    const u32 total = em.nrows * sub_lk;
    const u32 hs = em.hsel, hn = em.hcnt;             // all waves of the workgroup, the emitter included
    for (u32 k = lane + 64u * hs; k < total; k += 64u * hn) {
        const u32 r = k / sub_lk, j = k - r * sub_lk;
        const u32 v = em.lk16[k];
        const size_t at = lookup_block_base + lk0 + r * lk_per_unit + j;
        if constexpr (EM::COMPACT) {
            reinterpret_cast<u64 *>(p.lookup)[at] = v;
        } else {
            uint4 *out = reinterpret_cast<uint4 *>(p.lookup) + at * 2u;
            if constexpr (EM::MONT_OUT) {
                const Fe8 m = mont_from_u64<false>(v, 0);
                out[0] = make_uint4(m.l[0], m.l[1], m.l[2], m.l[3]);
                out[1] = make_uint4(m.l[4], m.l[5], m.l[6], m.l[7]);
            } else {
                out[0] = make_uint4(v, 0u, 0u, 0u);
                out[1] = make_uint4(0u, 0u, 0u, 0u);
            }
        }
    }
    em_sync<EM>();
}

template <int L, class EM, class C>
DEV void sub_end(C, EM &em, const ExpandParams &p, u64 block_first_limb, size_t lookup_block_base, u32 call0,
                 u32 calls_per_unit, u32 sub_calls, u32 lk0, u32 lk_per_unit, u32 sub_lk) {
    HSW_STAMP(5);
    if constexpr (C::pos != 0) {
        if constexpr (EM::M32) flush_tile32<EM, false>(em, C::pos, C::fl);
        else flush_tile<EM, false>(em, C::pos, C::fl, C::na, C::nb, C::nc, C::nd, C::cn);
    }
    HSW_STAMP(6);
    scatter_chip<L>(em, p, block_first_limb, call0, calls_per_unit, sub_calls);
    HSW_STAMP(7);
    if constexpr (EM::RC) scatter_lookup(em, p, lookup_block_base, lk0, lk_per_unit, sub_lk);
}

// One role's program (kernel below): EM is the emitter type, or its EMITS = false twin for helper waves.
template <int L, int REPR, bool RC, class EM>
DEV void small_role(const ExpandParams &p, const u32 *bw, const u32 (&ps)[8], size_t blk, u32 role, u64 *s_tile,
                    u16 *s_d16, u16 *s_lk16) {
    using LY = Lay<L, RC>;
    using SP = SmallPlan<L, RC>;
    const u32 lane = lane_id();
    EM em;
    em.lk16 = s_lk16;
    em.tile = s_tile;
    em.row0 = s_tile + (lane < (u32)SMALL_ROWS ? lane : (u32)SMALL_ROWS) * EM::STRIDE_W;
    em.row = em.row0;
    em.skew = 0;
    em.carry_neg = 0;
    em.head = nullptr;
    em.d16 = s_d16;
    em.tab = nullptr;
    em.hsel = (u32)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));   // wave-uniform (the flush loops stay scalar); blockDim.x / 64 waves share the role
    em.hcnt = blockDim.x >> 6;

    {   // FlexGate column packing: gaps of the breaks at or before this block, and the (<= 2) inside it
        u64 first = (u64)blk * (u64)LY::GATE_CELLS;
        if constexpr (RC)
            if (p.frame_every) first += (u64)(blk / p.frame_every) * p.frame_cells;
        u64 gap0 = 0;
        em.brk1 = em.brk2 = 0xffffffffu;
        em.gap1 = em.gap2 = 0;
        for (u32 k = 0; k < p.n_breaks; k++) {
            const u64 bc = p.break_cell[k];
            if (bc <= first) gap0 += p.break_gap[k];
            else if (bc < first + (u64)LY::GATE_CELLS) {
                if (em.brk1 == 0xffffffffu) { em.brk1 = (u32)(bc - first); em.gap1 = (u32)p.break_gap[k]; }
                else { em.brk2 = (u32)(bc - first); em.gap2 = (u32)p.break_gap[k]; }
            }
        }
        if constexpr (REPR == 2)
            em.out = reinterpret_cast<uint4 *>(reinterpret_cast<u64 *>(p.gate) + (size_t)(first + gap0));
        else
            em.out = reinterpret_cast<uint4 *>(p.gate) + (size_t)(first + gap0) * 2u;
    }
    size_t lk_blk = (size_t)blk * (size_t)LY::LOOKUP_CELLS;
    if constexpr (RC)
        if (p.frame_every) lk_blk += (size_t)(blk / p.frame_every) * (size_t)p.frame_lookups;
    em.write_gate = (p.flags & HSW_K_SKIP_GATE) == 0u;
    const u64 blk_limb0 = p.cursor0 + (u64)blk * (u64)LY::LIMB_CALLS;

    if (role < SMALL_ROUND_ROLES) {
        // ---- one sub-unit of rounds 16 * range .. + 15: compression.rs:125-196 --------------------------
        const u32 type = role / 4u, range = 3u - role % 4u, unit_lo = 16u * range;
        u32 lA, lE, lW;
        chain_latch<true>(bw, ps, (int)unit_lo + 15, (int)(unit_lo + lane) - 3, (int)(unit_lo + lane), lA, lE, lW);
        HSW_STAMP(1);
        // lane i < 16 expands round r = unit_lo + i: lanes i .. i + 3 hold A_(r-3) .. A_r
        const u32 a = lane_get(lA, lane + 3), b = lane_get(lA, lane + 2), c = lane_get(lA, lane + 1), d = lA;
        const u32 e = lane_get(lE, lane + 3), f = lane_get(lE, lane + 2), g = lane_get(lE, lane + 1), h = lE;
        const u32 wr = lW, kr = K256[(unit_lo + lane) & 63u];
        HSW_STAMP(2);
        const u32 call0 = LY::CALL_ROUNDS + unit_lo * LY::CALLS_ROUND, lk0 = LY::LK_OFF_ROUNDS + unit_lo * LY::LK_ROUND;
        auto begin = [&](int k) {
            sub_begin(em, unit_lo, 16, LY::ROUND, LY::OFF_ROUNDS, SP::off(SP::R_CELLS, k), SP::R_CALLS[k], SP::R_LK[k]);
        };
        auto end = [&](auto cur, int k) {
            sub_end<L>(cur, em, p, blk_limb0, lk_blk, call0 + SP::off(SP::R_CALLS, k), LY::CALLS_ROUND, SP::R_CALLS[k],
                       lk0 + SP::off(SP::R_LK, k), LY::LK_ROUND, SP::R_LK[k]);
        };
        W<EM> sig1, chv, t1, sig0, mjv, t2, e_new, a_new, s;
        if (type == 0) {
            begin(0);
            end(sigma_generic<SigmaUpper1, L>(CurStart{}, em, e, sig1), 0);                 // :130
        } else if (type == 1) {
            begin(1);
            const ChVals cv = ch_values(e, f, g);
            end(ch_part_a<L>(CurStart{}, em, cv, ch_witnesses(em, cv)), 1);                 // :131 (:309-365)
        } else if (type == 2) {
            begin(2);
            const ChVals cv = ch_values(e, f, g);
            auto c2 = ch_part_b<L>(CurStart{}, em, cv, ch_witnesses(em, cv), chv);          // :131 (:366-403)
            auto c3 = g_add(c2, em, w32<EM>(h), w32<EM>(sha_S1(e)), s);                     // :138-142
            auto c4 = g_add(c3, em, s, chv, s);                                             // :143-147
            auto c5 = g_add(c4, em, s, wround_constant<EM>(unit_lo + lane), s);             // :148-152
            auto c6 = g_add(c5, em, s, w32<EM>(wr), s);                                     // :153-157
            end(mod_u32(c6, em, s, t1), 2);                                                 // :158
        } else if (type == 3) {
            begin(3);
            end(sigma_generic<SigmaUpper0, L>(CurStart{}, em, a, sig0), 3);                 // :164
        } else if (type == 4) {
            begin(4);
            auto c9 = maj_gadget<L>(CurStart{}, em, a, b, c, mjv);                          // :165
            auto c10 = g_add(c9, em, w32<EM>(sha_S0(a)), mjv, s);                           // :166-170
            end(mod_u32(c10, em, s, t2), 4);                                                // :171
        } else {
            begin(5);
            t1 = w32<EM>(h + sha_S1(e) + sha_ch(e, f, g) + kr + wr);
            t2 = w32<EM>(sha_S0(a) + sha_maj(a, b, c));
            auto c12 = g_add(CurStart{}, em, w32<EM>(d), t1, s);                            // :181
            auto c13 = mod_u32(c12, em, s, e_new);                                          // :182
            auto c14 = state_to_spread<L>(c13, em, e_new);                                  // :184
            auto c15 = g_add(c14, em, t1, t2, s);                                           // :192
            auto c16 = mod_u32(c15, em, s, a_new);                                          // :193
            end(state_to_spread<L>(c16, em, a_new), 5);                                     // :195
        }
        HSW_STAMP(3);
    } else if (role < SMALL_ROUND_ROLES + SMALL_SCHED_ROLES) {
        // ---- one sub-unit of schedule steps 16 * range .. + 15 (idx = 16 + step): compression.rs:57-96 ---
        const u32 type = (role - SMALL_ROUND_ROLES) / 3u, range = 2u - (role - SMALL_ROUND_ROLES) % 3u, unit_lo = 16u * range;
        u32 lA, lE, lW;
        chain_latch<false>(bw, ps, (int)unit_lo + 30, 0, (int)(unit_lo + lane), lA, lE, lW);   // lane l: W_(unit_lo + l)
        HSW_STAMP(1);
        // step s = unit_lo + i, idx = s + 16: W[idx-16] = W_s, W[idx-15] = W_(s+1), W[idx-7] = W_(s+9), W[idx-2] = W_(s+14)
        const u32 w16 = lW, w15 = lane_get(lW, lane + 1), w7 = lane_get(lW, lane + 9), w2 = lane_get(lW, lane + 14);
        HSW_STAMP(2);
        const u32 call0 = LY::CALL_SCHED + unit_lo * LY::CALLS_SCHED, lk0 = LY::LK_OFF_SCHED + unit_lo * LY::LK_SCHED;
        sub_begin(em, unit_lo, 16, LY::SCHED, LY::OFF_SCHED, SP::off(SP::S_CELLS, (int)type), SP::S_CALLS[type], SP::S_LK[type]);
        auto end = [&](auto cur, int k) {
            sub_end<L>(cur, em, p, blk_limb0, lk_blk, call0 + SP::off(SP::S_CALLS, k), LY::CALLS_SCHED, SP::S_CALLS[k],
                       lk0 + SP::off(SP::S_LK, k), LY::LK_SCHED, SP::S_LK[k]);
        };
        W<EM> term1, term3, new_w, sum;
        if (type == 0) {
            end(sigma_generic<SigmaLower1, L>(CurStart{}, em, w2, term1), 0);               // :60
        } else if (type == 1) {
            end(sigma_generic<SigmaLower0, L>(CurStart{}, em, w15, term3), 1);              // :61
        } else {
            term1 = w32<EM>(sha_s1(w2)); term3 = w32<EM>(sha_s0(w15));
            auto c3 = g_add(CurStart{}, em, term1, w32<EM>(w7), sum);                       // :65-69
            auto c4 = g_add(c3, em, sum, term3, sum);                                       // :70-74
            auto c5 = g_add(c4, em, sum, w32<EM>(w16), sum);                                // :75-79
            auto c6 = mod_u32(c5, em, sum, new_w);                                          // :80
            end(state_to_spread<L>(c6, em, new_w), 2);                                      // :90
        }
        HSW_STAMP(3);
    } else if (role == SMALL_ROLE_FEED) {
        // ---- feed-forward: compression.rs:197-212, 8 units; this wave also owns next_states ------------
        u32 lA, lE, lW;
        chain_latch<true>(bw, ps, 64, 64 - (int)(lane & 3u), -1, lA, lE, lW);       // lanes 0..3: A_64..A_61 / E_64..E_61
        HSW_STAMP(1);
        u32 fy = ps[0];
#pragma unroll
        for (int i = 1; i < 8; i++) fy = (lane & 7u) == (u32)i ? ps[i] : fy;
        const u32 fx = lane < 4 ? lA : lE;
        if (lane < 8 && em.hsel == 0u) {
            if (p.next_states != nullptr) p.next_states[8 * blk + lane] = fy + fx;
            if (p.next_states_host != nullptr) p.next_states_host[8 * blk + lane] = fy + fx;
        }
        HSW_STAMP(2);
        if (phase_begin(em, 0, 1, 8, LY::FEED, LY::OFF_FEED, 0, 0, LY::LK_OFF_FEED, LY::LK_FEED)) {
            W<EM> s, lo;
            auto c1 = g_add(CurStart{}, em, w32<EM>(fx), w32<EM>(fy), s);
            auto c2 = mod_u32(c1, em, s, lo);
            phase_end<L>(c2, em, p, blk_limb0, lk_blk);
        }
        HSW_STAMP(3);
    } else if (role == SMALL_ROLE_WORDS) {
        // ---- words: compression.rs:31-47, 16 units of 4 mul_add -----------------------------------------
        const u32 word = __builtin_bswap32(bw[lane & 15u]);
        HSW_STAMP(1); HSW_STAMP(2);
        if (phase_begin(em, 0, 1, 16, LY::WORD, LY::OFF_WORDS, 0, 0)) {
            phase_end<L>(word_unit(CurStart{}, em, word), em, p, blk_limb0, lk_blk);
        }
        HSW_STAMP(3);
    } else if (role == SMALL_ROLE_MSG) {
        // ---- 16 x state_to_spread_u32(W[i]): compression.rs:53-56 ---------------------------------------
        const u32 word = __builtin_bswap32(bw[lane & 15u]);
        HSW_STAMP(1); HSW_STAMP(2);
        if (phase_begin(em, 0, 1, 16, LY::S2S, LY::OFF_MSG, LY::CALL_MSG, LY::CALLS_S2S)) {
            auto c1 = state_to_spread<L>(CurStart{}, em, w32<EM>(word));
            phase_end<L>(c1, em, p, blk_limb0, lk_blk);
        }
        HSW_STAMP(3);
    } else {
        // ---- 6 x state_to_spread_u32 of a,b,c,e,f,g: compression.rs:109-115 (d and h are never spread) --
        const u32 u = lane < 6 ? lane : 5u, wi = u < 3 ? u : u + 1;      // a, b, c, e, f, g
        u32 word = ps[0];
#pragma unroll
        for (int i = 1; i < 7; i++) word = wi == (u32)i ? ps[i] : word;
        HSW_STAMP(1); HSW_STAMP(2);
        if (phase_begin(em, 0, 1, 6, LY::S2S, LY::OFF_STATE, LY::CALL_STATE, LY::CALLS_S2S)) {
            auto c1 = state_to_spread<L>(CurStart{}, em, w32<EM>(word));
            phase_end<L>(c1, em, p, blk_limb0, lk_blk);
        }
        HSW_STAMP(3);
    }
}

template <int L, int REPR, bool RC>
__global__ __launch_bounds__(64 * HSW_SMALL_MAX_HELPERS) void hsw_small_kernel(ExpandParams p, SmallFrames fr) {
    using SP = SmallPlan<L, RC>;
    using EM = Em<SMALL_TILE, SMALL_ROWS, REPR, RC, true>;
    __shared__ __attribute__((aligned(16))) u64 s_tile[(SMALL_ROWS + 1) * EM::STRIDE_W];   // +1 scratch row for lanes >= 16
    __shared__ u16 s_d16[SMALL_ROWS * SP::MAX_CALLS];
    __shared__ u16 s_lk16[RC ? SMALL_ROWS * SP::MAX_LK : 1];
    HSW_STAMP(0);
    const u32 lane = lane_id();
    const u32 n_expand = (u32)p.n_blocks * SMALL_ROLES;

    // ---- frame waves (whole-digest launches): hsw_frame_body.hpp ------------------------------------
    if constexpr (RC && REPR != 2) {
        if (blockIdx.x >= n_expand) {
            if (threadIdx.x >= 64u) return;                      // (frame workgroups use their first wave only)
            const u32 wpf = fr.state_waves + fr.byte_waves;
            const u32 fw = blockIdx.x - n_expand, fi = fw / wpf, slice = fw % wpf;
            const FrameDesc d = fi == 0u ? fr.d0 : fr.descs[fi];
            uint4 *gate = reinterpret_cast<uint4 *>(fr.gate0), *lookup = reinterpret_cast<uint4 *>(fr.lookup0);
            const u64 *inv = reinterpret_cast<const u64 *>(fr.inv_tbl);
            if (slice >= fr.state_waves) {       // the input-byte cells: lib.rs:170-178
                framedev::frame_cells<REPR == 1>(d, fr.blocks0, inv, gate, lookup, fr.brk, framedev::FRAME_BYTES,
                                                 (slice - fr.state_waves) * 64u + lane, fr.byte_waves * 64u,
                                                 [](u32, u32) -> u32 { return 0u; });
                HSW_STAMP(4);
                return;
            }
            // candidate state n >= 1 = output of block n - 1 = pre-state of block n (the chain inputs); the
            // last block's output comes from the recurrence itself, computed here
            const u32 *ps0 = fr.pre0 + 8 * d.first_block, *ps_last = ps0 + 8 * (d.n_blocks - 1);
            // (the pre-states may sit in pinned host memory: fetch them now, they arrive while the chain runs)
            // (digests of at most SMALL_FRAME_MAX_BLOCKS blocks each -- enforced by launch_small_L; the launch itself
            // may hold up to 128 blocks, or any number with split = 2)
            __shared__ u32 s_states[8 * (SMALL_FRAME_MAX_BLOCKS + 2)];
            static_assert(8 * SMALL_FRAME_MAX_BLOCKS <= 4 * 64, "four prefetch loads per lane cover the chain inputs");
            const u32 nw = 8u * d.n_blocks;
            u32 pre_w[4];
#pragma unroll
            for (u32 k = 0; k < 4; k++) pre_w[k] = lane + 64u * k < nw ? ps0[lane + 64u * k] : 0u;
            u32 lA, lE, lW;
            chain_latch<true>(reinterpret_cast<const u32 *>(fr.blocks0 + 64 * (d.first_block + d.n_blocks - 1)), ps_last,
                              64, 64 - (int)(lane & 3u), -1, lA, lE, lW);
#pragma unroll
            for (u32 k = 0; k < 4; k++) if (lane + 64u * k < nw) s_states[lane + 64u * k] = pre_w[k];
            if (lane < 8) s_states[nw + lane] = ps_last[lane] + (lane < 4 ? lA : lE);   // compression.rs:197-212
            __syncthreads();
            HSW_STAMP(1);
            framedev::frame_cells<REPR == 1>(
                d, fr.blocks0, inv, gate, lookup, fr.brk, framedev::FRAME_STATES, slice * 64u + lane, fr.state_waves * 64u,
                [&](u32 n, u32 i) -> u32 { return s_states[8u * n + i]; });
            HSW_STAMP(4);
            return;
        }
    }

    // Grid order.  Block-major, or (HSW_K_ROLE_MAJOR, launches of <= 16 blocks) role-major: the workgroups of one
    // role -- the same instructions -- then sit next to each other.  Same-box A/B through the C ABI: 16
    // Montgomery blocks 34.5 -> 32.6 us, canonical unchanged, 32 Montgomery blocks 47.1 -> 48.9 us (hence the
    // limit); a longest-roles-first permutation on top of it changed nothing (workgroups do not start in grid order).
    const bool role_major = (p.flags & HSW_K_ROLE_MAJOR) != 0u;
    const u32 role = role_major ? blockIdx.x / (u32)p.n_blocks : blockIdx.x % SMALL_ROLES;
    const size_t blk = role_major ? blockIdx.x - role * (u32)p.n_blocks : blockIdx.x / SMALL_ROLES;
    const u32 *bw = reinterpret_cast<const u32 *>(p.blocks + 64 * blk);
    u32 ps[8];                                   // this block's pre-state (wave-uniform)
    if (p.flags & HSW_K_CHAINED) {
        // ONE message: pre_states holds its initial state only; block b's pre-state is b compressions away
        // (the roles that never look at the state skip the walk)
#pragma unroll
        for (int i = 0; i < 8; i++) ps[i] = p.pre_states[i];
        const bool needs_state = role < SMALL_ROUND_ROLES || role == SMALL_ROLE_FEED || role == SMALL_ROLE_STATE;
        if (needs_state && blk != 0) {
            // the message schedules of the blocks before this one do not depend on the state: lane l expands
            // block l's (all at once), K + W goes through LDS (the tile is not in use yet; rows 65 words apart:
            // no bank conflicts), and only the 64-round recurrence of each block remains serial
            u32 *s_kw = reinterpret_cast<u32 *>(s_tile);
            static_assert(sizeof(s_tile) >= 32 * 65 * 4, "K + W of 31 blocks must fit the tile");
            // (helper waves never look at the state: they only keep the barriers company -- walking along would
            //  cost the emitters' SIMDs a third of their issue slots, 66 vs 47 us per 16 blocks)
            const bool emitter = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)) == 0;
            if (emitter && lane < (u32)blk) {
                const u32 *bl = reinterpret_cast<const u32 *>(p.blocks + 64 * (size_t)lane);
                u32 w[16];
#pragma unroll
                for (int j = 0; j < 16; j++) w[j] = __builtin_bswap32(bl[j]);
#pragma unroll
                for (int t = 0; t < 64; t++) {
                    if (t >= 16)
                        w[t & 15] = w[t & 15] + sha_s0(w[(t + 1) & 15]) + w[(t + 9) & 15] + sha_s1(w[(t + 14) & 15]);
                    s_kw[lane * 65u + (u32)t] = w[t & 15] + K256[t];
                }
            }
            __syncthreads();
            for (u32 b = 0; emitter && b < (u32)blk; b++) {
                u32 a = ps[0], bb = ps[1], c = ps[2], d = ps[3], e = ps[4], f = ps[5], g = ps[6], h = ps[7];
                const u32 *kw = s_kw + b * 65u;
#pragma unroll 16
                for (int t = 0; t < 64; t++) {
                    const u32 t1 = h + kw[t] + sha_S1(e) + sha_ch(e, f, g);
                    const u32 t2 = sha_S0(a) + sha_maj(a, bb, c);
                    h = g; g = f; f = e; e = d + t1; d = c; c = bb; bb = a; a = t1 + t2;
                }
                ps[0] += a; ps[1] += bb; ps[2] += c; ps[3] += d; ps[4] += e; ps[5] += f; ps[6] += g; ps[7] += h;
            }
            __syncthreads();                       // the tile takes the memory over
        }
    } else {
#pragma unroll
        for (int i = 0; i < 8; i++) ps[i] = p.pre_states[8 * blk + i];
    }

    // wave 0 of the workgroup emits, the others (if any) only take their share of every flush: the same role
    // program instantiated without the staging stores (Em::HELPERS)
    using EMH = Em<SMALL_TILE, SMALL_ROWS, REPR, RC, true, false>;
    if (__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)) != 0)
        small_role<L, REPR, RC, EMH>(p, bw, ps, blk, role, s_tile, s_d16, s_lk16);
    else
        small_role<L, REPR, RC, EM>(p, bw, ps, blk, role, s_tile, s_d16, s_lk16);
    HSW_STAMP(4);
}

// 37 workgroups per block, of p.parts waves each (the emitter + p.parts - 1 helper waves, Em::HELPERS);
// fr == nullptr: no frame waves.
template <int L>
hipError_t launch_small_L(const ExpandParams &p, const SmallFrames *fr, hipStream_t stream) {
    if (p.n_blocks == 0) return hipSuccess;
    const unsigned helpers = p.parts;
    if (helpers == 0 || helpers > (unsigned)HSW_SMALL_MAX_HELPERS) return hipErrorInvalidValue;
    SmallFrames f{};
    if (fr) f = *fr;
    const bool rc = (p.flags & HSW_K_INTERNALS) != 0u;
    if (f.n_frames && (!rc || (p.flags & HSW_K_COMPACT))) return hipErrorInvalidValue;
    // the frame waves' LDS staging and prefetch are sized for digests of <= SMALL_FRAME_MAX_BLOCKS blocks
    if (f.n_frames && (f.max_frame_blocks == 0 || f.max_frame_blocks > SMALL_FRAME_MAX_BLOCKS ||
                       f.d0.n_blocks > f.max_frame_blocks)) return hipErrorInvalidValue;
    const dim3 grid((unsigned)(p.n_blocks * SMALL_ROLES + (size_t)f.n_frames * (f.state_waves + f.byte_waves))), block(64 * helpers);
    if (rc) {
        if (p.flags & HSW_K_MONTGOMERY) hipLaunchKernelGGL((hsw_small_kernel<L, 1, true>), grid, block, 0, stream, p, f);
        else if (p.flags & HSW_K_COMPACT) hipLaunchKernelGGL((hsw_small_kernel<L, 2, true>), grid, block, 0, stream, p, f);
        else hipLaunchKernelGGL((hsw_small_kernel<L, 0, true>), grid, block, 0, stream, p, f);
    } else {
        if (p.flags & HSW_K_MONTGOMERY) hipLaunchKernelGGL((hsw_small_kernel<L, 1, false>), grid, block, 0, stream, p, f);
        else if (p.flags & HSW_K_COMPACT) hipLaunchKernelGGL((hsw_small_kernel<L, 2, false>), grid, block, 0, stream, p, f);
        else hipLaunchKernelGGL((hsw_small_kernel<L, 0, false>), grid, block, 0, stream, p, f);
    }
    return hipGetLastError();
}

}  // namespace hsw
#endif
