// hsw_layout.h -- cell-count arithmetic of one block's witness streams.
//
// Single source of truth for host (hsw_shape_query) and device (kernel
// template constants).  Every count is the number of halo2-base gate cells the
// reference's call sequence allocates (1 per load_witness, 4 per
// add/neg/mul_add; DESIGN.md assumption A1), derived per function:
//
//   spread            spread.rs:76-123        L lw + L mul_add + L*(lw + mul_add) = 10 L
//   state_to_spread   compression.rs:215-246  2 lw + mul_add + 2 spread           = 6 + 2 S
//   mod_u32           compression.rs:266-295  2 lw + mul_add                      = 6
//   sigma_generic     compression.rs:702-882  4 lw + 3 ma + ma + 4 ma + 2 lw + ma
//                                             + 4 lw + 2*(2 spread + ma) + ma     = 58 + 4 S
//   ch                compression.rs:297-405  2 add + 2 neg + 4 add + 8 lw
//                                             + 4*(2 spread + ma) + 2 add + ma    = 68 + 8 S
//   maj               compression.rs:460-519  4 add + 4 lw + 2*(2 spread + ma) + ma = 32 + 4 S
//   schedule step     compression.rs:57-96    2 sigma + 3 add + mod + s2s         = 140 + 10 S
//   round             compression.rs:125-196  2 sigma + ch + maj + 7 add + 4 mod
//                                             + 2 s2s                             = 280 + 24 S
// with L = 16 / num_bits_lookup limbs per spread and S = 10 L.
//
// RC = true adds halo2-base's own cells of every range_check(a, 32) at the call
// position: [limb0, limb1, 2^16, a] (DESIGN.md assumption A3; range_check(a, 16)
// adds none at lookup_bits = 16): +4 per mod_u32, +8 per sigma_generic.
// Independently of RC, LK_* count the cells queued for the lookup-advice column
// (2 limbs per range_check 32, 1 value per range_check 16), in call order.
#ifndef HSW_LAYOUT_H
#define HSW_LAYOUT_H

namespace hsw {

template <int L, bool RC = false>
struct Lay {
    static_assert(L == 1 || L == 2 || L == 4 || L == 8 || L == 16, "16 % num_bits_lookup == 0");
    static constexpr int LIMBS = L;
    static constexpr int LIMB_BITS = 16 / L;
    static constexpr int S = 10 * L;            // cells per spread()
    static constexpr int S2S = 6 + 2 * S;       // state_to_spread_u32
    static constexpr int RC32 = RC ? 4 : 0;     // halo2-base cells of one range_check(a, 32)
    static constexpr int MOD = 6 + RC32;        // mod_u32
    static constexpr int SIGMA = 58 + 4 * S + 2 * RC32;
    static constexpr int CH = 68 + 8 * S;
    static constexpr int MAJ = 32 + 4 * S;
    static constexpr int SCHED = 2 * SIGMA + 12 + MOD + S2S;
    static constexpr int ROUND = 2 * SIGMA + CH + MAJ + 7 * 4 + 4 * MOD + 2 * S2S;
    static constexpr int WORD = 16;             // 4 mul_add per message word
    static constexpr int FEED = 4 + MOD;        // add + mod_u32 per state word

    // regions of one block's gate stream, in stream order
    static constexpr int OFF_WORDS = 0;
    static constexpr int OFF_MSG = OFF_WORDS + 16 * WORD;
    static constexpr int OFF_SCHED = OFF_MSG + 16 * S2S;
    static constexpr int OFF_STATE = OFF_SCHED + 48 * SCHED;
    static constexpr int OFF_ROUNDS = OFF_STATE + 6 * S2S;
    static constexpr int OFF_FEED = OFF_ROUNDS + 64 * ROUND;
    static constexpr int GATE_CELLS = OFF_FEED + 8 * FEED;

    // spread() calls, in call order (each stages one 16-bit dense value)
    static constexpr int CALLS_S2S = 2;
    static constexpr int CALLS_SIGMA = 4;
    static constexpr int CALLS_SCHED = 2 * CALLS_SIGMA + CALLS_S2S;                 // 10
    static constexpr int CALLS_ROUND = 2 * CALLS_SIGMA + 8 + 4 + 2 * CALLS_S2S;     // 24
    static constexpr int CALL_MSG = 0;
    static constexpr int CALL_SCHED = CALL_MSG + 16 * CALLS_S2S;                    // 32
    static constexpr int CALL_STATE = CALL_SCHED + 48 * CALLS_SCHED;                // 512
    static constexpr int CALL_ROUNDS = CALL_STATE + 6 * CALLS_S2S;                  // 524
    static constexpr int SPREAD_CALLS = CALL_ROUNDS + 64 * CALLS_ROUND;             // 2060
    static constexpr int LIMB_CALLS = SPREAD_CALLS * L;
    static constexpr int CHIP_CELLS = 2 * LIMB_CALLS;

    // lookup-advice column entries, in enable_lookup order
    static constexpr int LK_MOD = 2;                                  // range_check(lo, 32)
    static constexpr int LK_SIGMA = 2 * 2 + 4;                        // 2 x range_check 32 + 4 x range_check 16
    static constexpr int LK_CH = 8, LK_MAJ = 4;                       // even/odd range_check 16
    static constexpr int LK_SCHED = 2 * LK_SIGMA + LK_MOD;            // 18
    static constexpr int LK_ROUND = 2 * LK_SIGMA + LK_CH + LK_MAJ + 4 * LK_MOD;   // 36
    static constexpr int LK_FEED = LK_MOD;
    static constexpr int LK_OFF_SCHED = 0;
    static constexpr int LK_OFF_ROUNDS = LK_OFF_SCHED + 48 * LK_SCHED;            // 864
    static constexpr int LK_OFF_FEED = LK_OFF_ROUNDS + 64 * LK_ROUND;             // 3168
    static constexpr int LOOKUP_CELLS = LK_OFF_FEED + 8 * LK_FEED;                // 3184
};

static_assert(Lay<2>::ROUND == 760 && Lay<2>::SCHED == 340 && Lay<2>::GATE_CELLS == 66308,
              "SURVEY 8a per-block tallies at num_bits_lookup = 8");
static_assert(Lay<2>::SPREAD_CALLS == 2060 && Lay<2>::CHIP_CELLS == 8240, "SURVEY 8a");
static_assert(Lay<2, true>::GATE_CELLS == 66308 + 760 * 4, "760 range_check(32) per block (SURVEY 8a)");
static_assert(Lay<2>::LOOKUP_CELLS == 3184, "SURVEY 8c: ~3,184 lookup-column copies per block");

}  // namespace hsw
#endif
