// hsw_frame_body.hpp -- device code of the digest frame: the cells Sha256DynamicConfig::digest
// allocates around its block loop -- prologue lib.rs:122-178, epilogue lib.rs:294-341, the Context's zero
// cell.  Cell layout: hsw_frame.hpp (assumption A4).  Shared by hsw_frame_kernel (hsw_frame.hip: its own
// launch, reads the next states the expansion left in HBM) and by the frame waves of the small-batch kernel
// (hsw_small.hpp: same launch as the expansion, next states from the chain inputs).
#ifndef HSW_FRAME_BODY_HPP
#define HSW_FRAME_BODY_HPP
#include "hsw_expand.hpp"
#include "hsw_frame.hpp"

namespace hsw {
namespace framedev {


// A field element as eight 32-bit limbs in the output representation.
template <bool MONT>
DEV Fe8 fe_small(u64 v) {
    if constexpr (MONT) {
        return mont_from_u64<true>((u32)v, (u32)(v >> 32));
    } else {
        Fe8 r;
        r.l[0] = (u32)v; r.l[1] = (u32)(v >> 32);
#pragma unroll
        for (int j = 2; j < 8; j++) r.l[j] = 0;
        return r;
    }
}
DEV Fe8 fe_zero() {
    Fe8 r;
#pragma unroll
    for (int j = 0; j < 8; j++) r.l[j] = 0;
    return r;
}
// sign * mag for a small magnitude: the field element mag, or p - mag
template <bool MONT>
DEV Fe8 fe_signed(bool negative, u64 mag) {
    if (mag == 0) return fe_zero();
    const Fe8 m = fe_small<MONT>(mag);
    return negative ? fe_neg_nonzero(m) : m;
}
DEV Fe8 fe_load(const u64 *t) {
    Fe8 r;
#pragma unroll
    for (int j = 0; j < 4; j++) { r.l[2 * j] = (u32)t[j]; r.l[2 * j + 1] = (u32)(t[j] >> 32); }
    return r;
}
DEV void put(uint4 *gate, u64 cell, const Fe8 &v) {
    uint4 a, b;
    a.x = v.l[0]; a.y = v.l[1]; a.z = v.l[2]; a.w = v.l[3];
    b.x = v.l[4]; b.y = v.l[5]; b.z = v.l[6]; b.w = v.l[7];
    gate[2 * cell] = a;
    gate[2 * cell + 1] = b;
}

template <bool MONT>
struct Out {
    uint4 *gate;
    uint4 *lookup;      // may be null
    const FrameBreaks *brk;
    // FlexGate column packing (A3-iii): stream cell `at` sits at `at` + the gaps of all breaks at or before it.
    // The walk over the break table (scalar loads, one round trip each) is remembered as the column segment
    // [seg_lo, seg_hi) it found: a thread's cells nearly always share one.
    mutable u64 seg_lo = 1, seg_hi = 0, seg_gap = 0;
    DEV u64 place(u64 at) const {
        if (at >= seg_lo && at < seg_hi) return at + seg_gap;
        u64 gap = 0, lo = 0, hi = ~0ull;
        for (u32 k = 0; k < brk->n; k++) {
            const u64 c = brk->cell[k];
            if (c <= at) { gap += brk->gap[k]; lo = c; }
            else if (c < hi) hi = c;
        }
        seg_lo = lo; seg_hi = hi; seg_gap = gap;
        return at + gap;
    }
    DEV void cell(u64 at, u64 v) const { put(gate, place(at), fe_small<MONT>(v)); }
    DEV void cell_signed(u64 at, bool neg, u64 mag) const { put(gate, place(at), fe_signed<MONT>(neg, mag)); }
    DEV void cell_fe(u64 at, const Fe8 &v) const { put(gate, place(at), v); }
    DEV void look(u64 at, u64 v) const { if (lookup) put(lookup, at, fe_small<MONT>(v)); }
    // one conversion site for a gate cell or a lookup-column entry (divergent callers pay for it once)
    DEV void any(bool is_look, u64 at, bool neg, u64 mag) const {
        if (is_look && !lookup) return;
        put(is_look ? lookup : gate, is_look ? at : place(at), fe_signed<MONT>(neg, mag));
    }
    // range_check(byte, 8): lookup byte; [0, byte, 2^8, byte * 2^8]; lookup the product
    DEV void range_check8(u64 at, u64 lk, u32 byte) const {
        cell(at, 0); cell(at + 1, byte); cell(at + 2, 256); cell(at + 3, (u64)byte << 8);
        look(lk, byte); look(lk + 1, (u64)byte << 8);
    }
};


// All frame cells of one digest, dealt to `nthreads` threads (gtid = this thread's index among them).
// state_word(n, i): word i of candidate state n (lib.rs:162-165, 236): 0 = the state after the precomputed
// prefix, n >= 1 = the output of block n - 1.
// inv_tbl[k] (k = 1..max |n - target|): k^-1 mod p in the OUTPUT representation, 4 x u64 each
// parts: which of the three groups of cells this call writes -- FRAME_BYTES the input-byte cells (they need
// nothing but the bytes), FRAME_STATES everything that looks at a state word (the fixed prologue cells, the
// state selection, the digest bytes); gtid / nthreads count the threads of the SAME group.
enum : u32 { FRAME_BYTES = 1u, FRAME_STATES = 2u, FRAME_ALL = 3u };
template <bool MONT, class StateWord>
DEV void frame_cells(const FrameDesc &d, const uint8_t *blocks, const u64 *inv_tbl, uint4 *gate, uint4 *lookup,
                     const FrameBreaks &brk, u32 parts, u32 gtid, u32 nthreads, StateWord state_word) {
    using namespace frame;
    const Out<MONT> o{gate, lookup, &brk};      // (its segment cache is per thread)
    const u64 P0 = d.prologue_cell, E0 = d.epilogue_cell;
    const u32 N = d.n_blocks;
    const u64 max_bytes = (u64)N * 64u;
    const u32 target = d.num_round - d.precomputed_round;               // lib.rs:147-151
    const uint8_t *bytes = blocks + 64 * d.first_block;
    // ---- everything that looks at a state word, ONE CELL per work item and one conversion site for all of
    //      them (a wave's lanes sit in different halo2-base calls; dealt call by call, the lanes of the
    //      longest call -- 12 cells, ~200 instructions each in Montgomery form -- held the whole launch up):
    //        [0, 192)                  prologue, fixed part: lib.rs:124-165; item t < 46 = cell t, item 64 = the
    //                                  fixed lookups and the zero cell
    //        [192, +76 (N + 1))        epilogue, state selection: lib.rs:294-310; candidate n = j / 76, cell j % 76:
    //                                  is_equal(n, target) (12 cells), then select of word i (8 cells each)
    //        [.., +32 * 11)            epilogue, digest bytes: lib.rs:311-341; (word, byte) = j / 11, cell j % 11:
    //                                  the byte, range_check 8 (4 cells), the mul_add (4 cells), 2 lookup entries
    const u32 n_sel = (u32)E_STATE * (N + 1), n_items = 192u + n_sel + 32u * 11u;
    const u64 len = d.input_len, nr = d.num_round, pre = d.precomputed_round;
    const u64 padded = 64 * nr, with9 = len + 9, pad = padded - with9;          // pad < 64 (lib.rs:142-144; host-checked)
    // is_less_than_safe(padding_size, 64)
    const u64 shift_a = pad + 65536, shifted = shift_a - 64;
    const u64 limb0 = shifted & 0xffff, limb1 = shifted >> 16;                  // limb1 = 0 <=> pad < 64
    const u64 z = limb1 == 0 ? 1 : 0;
    for (u32 t = (parts & FRAME_STATES) ? gtid : n_items; t < n_items; t += nthreads) {
        bool is_look = false, neg = false;
        u64 at = 0, v = 0;
        if (t < 192u) {
            const u32 tid = t;
            if (tid < P_STATE) {
                switch (tid) {
                    case P_LEN: v = len; break;
                    case P_NROUND: v = nr; break;
                    case P_MUL: v = 0; break;      case P_MUL + 1: v = nr; break;     case P_MUL + 2: v = 64; break;  case P_MUL + 3: v = padded; break;
                    case P_ADD: v = len; break;    case P_ADD + 1: v = 9; break;      case P_ADD + 2: v = 1; break;   case P_ADD + 3: v = with9; break;
                    case P_SUB: v = pad; break;    case P_SUB + 1: v = with9; break;  case P_SUB + 2: v = 1; break;   case P_SUB + 3: v = padded; break;
                    case P_LT: v = shifted; break; case P_LT + 1: v = 64; break;      case P_LT + 2: v = 1; break;    case P_LT + 3: v = shift_a; break;
                    case P_LT + 4: v = 65536; break;   /* written as -2^16 */         case P_LT + 5: v = 1; break;    case P_LT + 6: v = pad; break;
                    case P_RC32: v = limb0; break; case P_RC32 + 1: v = limb1; break; case P_RC32 + 2: v = 65536; break; case P_RC32 + 3: v = shifted; break;
                    case P_ISZ: v = z; break;      case P_ISZ + 1: v = limb1; break;
                    case P_ISZ + 2: v = 1; break;  // is_zero's inv witness of 0 is 1
                    case P_ISZ + 3: v = 1; break;  case P_ISZ + 4: v = 0; break;      case P_ISZ + 5: v = limb1; break;
                    case P_ISZ + 6: v = z; break;  case P_ISZ + 7: v = 0; break;
                    case P_PRE: v = pre; break;
                    case P_TGT: v = nr - pre; break; case P_TGT + 1: v = pre; break;  case P_TGT + 2: v = 1; break;   case P_TGT + 3: v = nr; break;
                    default: break;
                }
                at = P0 + tid; neg = tid == P_LT + 4;
            } else if (tid < P_BYTES) {
                at = P0 + tid; v = state_word(0, tid - P_STATE);                 // lib.rs:162-165
            } else if (tid >= 64u && tid < 67u) {
                is_look = true; at = d.prologue_lookup + (tid - 64u);
                v = tid == 64u ? pad : tid == 65u ? limb0 : limb1;
            } else if (tid == 67u && d.zero_cell != ~0ull) {
                at = d.zero_cell; v = 0;                                         // Context.zero_cell
            } else {
                continue;
            }
        } else if (t < 192u + n_sel) {
            const u32 j = t - 192u, n = j / (u32)E_STATE, k = j % (u32)E_STATE;
            const bool sel = n == target;
            at = E0 + j;
            if (k < 12u) {                                                       // is_equal(n, target)
                const bool lt = n < target;
                const u64 mag = lt ? target - n : n - target;
                if (k == 6u && !sel) {                                           // (n - target)^-1 = -(target - n)^-1
                    const Fe8 inv = fe_load(inv_tbl + 4 * mag);
                    o.cell_fe(at, lt ? fe_neg_nonzero(inv) : inv);
                    continue;
                }
                switch (k) {
                    case 0: case 5: case 9: neg = lt; v = mag; break;
                    case 1: case 6: case 7: v = 1; break;
                    case 2: v = target; break;
                    case 3: v = n; break;
                    case 4: case 10: v = sel ? 1 : 0; break;
                    default: v = 0; break;                                       // 8, 11
                }
            } else {                                                             // select of word i
                const u32 i = (k - 12u) / 8u, q = (k - 12u) % 8u;
                const u64 a = state_word(n, i);                                  // assigned_state[i]
                const u64 b = n > target ? state_word(target, i) : 0;            // output_h_out[i] so far
                const bool lt = a < b;
                const u64 mag = lt ? b - a : a - b;
                switch (q) {
                    case 0: case 6: neg = lt; v = mag; break;
                    case 1: v = 1; break;
                    case 2: case 4: v = b; break;
                    case 3: v = a; break;
                    case 5: v = sel ? 1 : 0; break;
                    default: v = sel ? a : b; break;                             // 7
                }
            }
        } else {
            const u32 j = t - 192u - n_sel, wi = j / 11u, c = j % 11u, w = wi / 4u, idx = wi % 4u;
            const u32 word = target <= N ? state_word(target, w) : 0;
            const u64 e = E0 + (u64)E_STATE * (N + 1) + (u64)E_WORD * w;
            const u32 byte = (word >> (24 - 8 * idx)) & 0xffu;
            const u32 sum_before = idx == 0 ? 0 : (word >> (32 - 8 * idx)) << (32 - 8 * idx);
            // range_check(byte, 8): lookup byte; [0, byte, 2^8, byte * 2^8]; lookup the product
            switch (c) {
                case 0: at = e + 5 * idx; v = byte; break;
                case 1: at = e + 5 * idx + 1; v = 0; break;
                case 2: at = e + 5 * idx + 2; v = byte; break;
                case 3: at = e + 5 * idx + 3; v = 256; break;
                case 4: at = e + 5 * idx + 4; v = (u64)byte << 8; break;
                case 5: at = e + 20 + 4 * idx; v = sum_before; break;
                case 6: at = e + 20 + 4 * idx + 1; v = byte; break;
                case 7: at = e + 20 + 4 * idx + 2; v = 1ull << (24 - 8 * idx); break;
                case 8: at = e + 20 + 4 * idx + 3; v = (u64)sum_before + ((u64)byte << (24 - 8 * idx)); break;
                case 9: is_look = true; at = d.epilogue_lookup + 8 * w + 2 * idx; v = byte; break;
                default: is_look = true; at = d.epilogue_lookup + 8 * w + 2 * idx + 1; v = (u64)byte << 8; break;
            }
        }
        o.any(is_look, at, neg, v);
    }

    // ---- prologue, input bytes: lib.rs:170-178 ----
    for (u64 i = (parts & FRAME_BYTES) ? (u64)gtid : max_bytes; i < max_bytes; i += nthreads) {
        const u32 b = bytes[i];
        o.cell(P0 + P_BYTES + i, b);
        if (d.range_check_inputs)
            o.range_check8(P0 + P_BYTES + max_bytes + 4 * i, d.prologue_lookup + P_FIXED_LOOKUPS + 2 * i, b);
    }

}

}  // namespace framedev
}  // namespace hsw
#endif
