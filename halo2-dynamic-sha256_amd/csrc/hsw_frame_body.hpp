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
    // ---- prologue, fixed part: lib.rs:124-165; work item t < 46 writes cell t, item 64 the fixed lookups
    //      and the zero cell (items 128..159: the digest bytes, below) ----
    for (u32 tid = (parts & FRAME_STATES) ? gtid : 192u; tid < 192u; tid += nthreads) {
        const u64 len = d.input_len, nr = d.num_round, pre = d.precomputed_round;
        const u64 padded = 64 * nr, with9 = len + 9, pad = padded - with9;      // pad < 64 (lib.rs:142-144; host-checked)
        // is_less_than_safe(padding_size, 64)
        const u64 shift_a = pad + 65536, shifted = shift_a - 64;
        const u64 limb0 = shifted & 0xffff, limb1 = shifted >> 16;              // limb1 = 0 <=> pad < 64
        const u64 z = limb1 == 0 ? 1 : 0;
        if (tid < P_STATE) {
            u64 v = 0;
            switch (tid) {
                case P_LEN: v = len; break;
                case P_NROUND: v = nr; break;
                case P_MUL: v = 0; break;      case P_MUL + 1: v = nr; break;     case P_MUL + 2: v = 64; break;  case P_MUL + 3: v = padded; break;
                case P_ADD: v = len; break;    case P_ADD + 1: v = 9; break;      case P_ADD + 2: v = 1; break;   case P_ADD + 3: v = with9; break;
                case P_SUB: v = pad; break;    case P_SUB + 1: v = with9; break;  case P_SUB + 2: v = 1; break;   case P_SUB + 3: v = padded; break;
                case P_LT: v = shifted; break; case P_LT + 1: v = 64; break;      case P_LT + 2: v = 1; break;    case P_LT + 3: v = shift_a; break;
                case P_LT + 4: v = 65536; break;   /* written as -2^16 below */   case P_LT + 5: v = 1; break;    case P_LT + 6: v = pad; break;
                case P_RC32: v = limb0; break; case P_RC32 + 1: v = limb1; break; case P_RC32 + 2: v = 65536; break; case P_RC32 + 3: v = shifted; break;
                case P_ISZ: v = z; break;      case P_ISZ + 1: v = limb1; break;
                case P_ISZ + 2: v = 1; break;  // is_zero's inv witness of 0 is 1
                case P_ISZ + 3: v = 1; break;  case P_ISZ + 4: v = 0; break;      case P_ISZ + 5: v = limb1; break;
                case P_ISZ + 6: v = z; break;  case P_ISZ + 7: v = 0; break;
                case P_PRE: v = pre; break;
                case P_TGT: v = nr - pre; break; case P_TGT + 1: v = pre; break;  case P_TGT + 2: v = 1; break;   case P_TGT + 3: v = nr; break;
                default: break;
            }
            o.cell_signed(P0 + tid, tid == P_LT + 4, v);
        } else if (tid < P_BYTES) {
            o.cell(P0 + tid, state_word(0, tid - P_STATE));                      // lib.rs:162-165
        } else if (tid == 64) {
            o.look(d.prologue_lookup, pad);
            o.look(d.prologue_lookup + 1, limb0);
            o.look(d.prologue_lookup + 2, limb1);
            if (d.zero_cell != ~0ull) o.cell(d.zero_cell, 0);                   // Context.zero_cell
        }
    }

    // ---- prologue, input bytes: lib.rs:170-178 ----
    for (u64 i = (parts & FRAME_BYTES) ? (u64)gtid : max_bytes; i < max_bytes; i += nthreads) {
        const u32 b = bytes[i];
        o.cell(P0 + P_BYTES + i, b);
        if (d.range_check_inputs)
            o.range_check8(P0 + P_BYTES + max_bytes + 4 * i, d.prologue_lookup + P_FIXED_LOOKUPS + 2 * i, b);
    }

    // ---- epilogue, state selection: lib.rs:294-310 ----
    // work item j: candidate n = j / 9; part 0 = is_equal(n, target), parts 1..8 = select of word part - 1
    for (u32 j = (parts & FRAME_STATES) ? gtid : 9u * (N + 1); j < 9u * (N + 1); j += nthreads) {
        const u32 n = j / 9u, part = j % 9u;
        const u64 at = E0 + (u64)E_STATE * n;
        const bool sel = n == target;
        if (part == 0) {
            const bool neg = n < target;
            const u64 mag = neg ? target - n : n - target;
            o.cell_signed(at, neg, mag); o.cell(at + 1, 1); o.cell(at + 2, target); o.cell(at + 3, n);
            o.cell(at + 4, sel ? 1 : 0); o.cell_signed(at + 5, neg, mag);
            if (sel) o.cell(at + 6, 1);
            else {
                const Fe8 inv = fe_load(inv_tbl + 4 * mag);                     // (n - target)^-1 = -(target - n)^-1
                o.cell_fe(at + 6, neg ? fe_neg_nonzero(inv) : inv);
            }
            o.cell(at + 7, 1); o.cell(at + 8, 0); o.cell_signed(at + 9, neg, mag);
            o.cell(at + 10, sel ? 1 : 0); o.cell(at + 11, 0);
        } else {
            const u32 i = part - 1;
            const u64 a = state_word(n, i);                                     // assigned_state[i]
            const u64 b = n > target ? state_word(target, i) : 0;               // output_h_out[i] so far
            const u64 s = at + 12 + 8 * i;
            const bool neg = a < b;
            const u64 mag = neg ? b - a : a - b;
            o.cell_signed(s, neg, mag); o.cell(s + 1, 1); o.cell(s + 2, b); o.cell(s + 3, a);
            o.cell(s + 4, b); o.cell(s + 5, sel ? 1 : 0); o.cell_signed(s + 6, neg, mag);
            o.cell(s + 7, sel ? a : b);
        }
    }

    // ---- epilogue, digest bytes: lib.rs:311-341 ----
    for (u32 tid = (parts & FRAME_STATES) ? 128u + gtid : 160u; tid < 160u; tid += nthreads) {
        const u32 w = (tid - 128) / 4, idx = (tid - 128) % 4;
        const u32 word = target <= N ? state_word(target, w) : 0;
        const u64 at = E0 + (u64)E_STATE * (N + 1) + (u64)E_WORD * w;
        const u32 byte = (word >> (24 - 8 * idx)) & 0xffu;
        o.cell(at + 5 * idx, byte);
        o.range_check8(at + 5 * idx + 1, d.epilogue_lookup + 8 * w + 2 * idx, byte);
        const u32 sum_before = idx == 0 ? 0 : (word >> (32 - 8 * idx)) << (32 - 8 * idx);
        const u64 m = at + 20 + 4 * idx;
        o.cell(m, sum_before); o.cell(m + 1, byte); o.cell(m + 2, 1ull << (24 - 8 * idx));
        o.cell(m + 3, (u64)sum_before + ((u64)byte << (24 - 8 * idx)));
    }
}

}  // namespace framedev
}  // namespace hsw
#endif
