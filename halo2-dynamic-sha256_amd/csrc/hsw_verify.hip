// hsw_verify.hip -- on-device check of a block witness against the gadget's constraint system.
//
// What MockProver::verify checks for the path (reference lib.rs:525-526), evaluated in HBM right where the
// witness was written: every gate row x0 + x1*x2 = x3, every copy constraint (QuantumCell::Existing and
// assert_equal), every fixed constant, every range_check bound, the spread-chip cells (tied to their gate
// cells, and (dense, spread) a row of the spread table), the lookup-column copies, the next-state words --
// with the block bytes and the pre-state entering only through the external cells they are
// copy-constrained to.  The structure is the product's own (csrc/hsw_structure.hpp); no value is
// recomputed from the inputs, so a stream that passes is the witness of its inputs by the uniqueness
// argument of SURVEY 8c.  One pass over the gate rows (four lanes per row, one cell each) checks the row
// equation and, on the same loads, the constants and copies among the row's cells; then the assert_equal
// pairs, range bounds, chip ties and lookup copies.  ~3.0 ms for 4,096 blocks (1.8x the time it took to
// write them).  `slices` workgroups may share a block: small batches are sliced to fill the chip.
#include "hsw_expand.hpp"
#include "hsw_frame.hpp"
#include "hsw_verify.h"

namespace hsw {

namespace {

struct Cell { u64 l[4]; };

DEV Cell load_cell(const uint4 *gate, u64 idx) {
    const uint4 a = gate[2 * idx], b = gate[2 * idx + 1];
    Cell c;
    c.l[0] = (u64)a.x | ((u64)a.y << 32); c.l[1] = (u64)a.z | ((u64)a.w << 32);
    c.l[2] = (u64)b.x | ((u64)b.y << 32); c.l[3] = (u64)b.z | ((u64)b.w << 32);
    return c;
}
// HSW_REPR_MONTGOMERY streams are checked in the canonical domain: every loaded cell m = x * 2^256 mod p is
// reduced to x on the fly (one Montgomery reduction) wherever its VALUE is needed -- gate equation, constants, ranges;
// copies of stream cells are compared as stored.  3.1-3.3 ms per 4,096 blocks against 3.0 ms canonical.
DEV Cell from_mont(const Cell &a) {
    const u64 P[4] = {0x43e1f593f0000001ull, 0x2833e84879b97091ull, 0xb85045b68181585dull, 0x30644e72e131a029ull};
    const u64 INV = 0xc2e1f593efffffffull;                       // -p^-1 mod 2^64
    u64 t[5] = {a.l[0], a.l[1], a.l[2], a.l[3], 0};
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const u64 m = t[0] * INV;
        unsigned __int128 s = (unsigned __int128)m * P[0] + t[0];
        u64 carry = (u64)(s >> 64);
#pragma unroll
        for (int j = 1; j < 4; j++) {
            s = (unsigned __int128)m * P[j] + t[j] + carry;
            t[j - 1] = (u64)s;
            carry = (u64)(s >> 64);
        }
        s = (unsigned __int128)t[4] + carry;
        t[3] = (u64)s;
        t[4] = (u64)(s >> 64);
    }
    Cell r;
    r.l[0] = t[0]; r.l[1] = t[1]; r.l[2] = t[2]; r.l[3] = t[3];
    // r < 2p: one conditional subtraction
    bool ge = t[4] != 0;
    if (!ge) { ge = true; for (int i = 3; i >= 0; i--) { if (r.l[i] > P[i]) break; if (r.l[i] < P[i]) { ge = false; break; } } }
    if (ge) { u64 br = 0; for (int i = 0; i < 4; i++) { const unsigned __int128 d = (unsigned __int128)r.l[i] - P[i] - br; r.l[i] = (u64)d; br = (u64)(d >> 64) & 1; } }
    return r;
}
template <bool MONT>
DEV Cell load_value(const uint4 *base, u64 idx) {
    const Cell c = load_cell(base, idx);
    if constexpr (MONT) return from_mont(c); else return c;
}
DEV bool narrow(const Cell &c) { return (c.l[1] | c.l[2] | c.l[3]) == 0; }
DEV bool same(const Cell &a, const Cell &b) { return a.l[0] == b.l[0] && a.l[1] == b.l[1] && a.l[2] == b.l[2] && a.l[3] == b.l[3]; }
DEV Cell small(u64 v) { Cell c; c.l[0] = v; c.l[1] = c.l[2] = c.l[3] = 0; return c; }

// FlexGate column packing: where stream cell i sits
template <class P>
DEV u64 place(const P &p, u64 i) {
    u64 gap = 0;
    for (u32 k = 0; k < p.n_breaks; k++) gap += p.break_cell[k] <= i ? p.break_gap[k] : 0;
    return i + gap;
}

// ---- generic field arithmetic for the few full-width rows of a digest frame (is_zero's z + a*inv = 1, the
// differences of is_equal / select): canonical 4 x u64 limbs, double-and-add -- slow and simple; a frame has
// a few hundred such rows, a block none beyond the negation pattern handled inline.
DEV bool geq_p(const Cell &a) {
    const u64 P[4] = {0x43e1f593f0000001ull, 0x2833e84879b97091ull, 0xb85045b68181585dull, 0x30644e72e131a029ull};
    for (int i = 3; i >= 0; i--) { if (a.l[i] > P[i]) return true; if (a.l[i] < P[i]) return false; }
    return true;
}
DEV Cell sub_p(const Cell &a) {
    const u64 P[4] = {0x43e1f593f0000001ull, 0x2833e84879b97091ull, 0xb85045b68181585dull, 0x30644e72e131a029ull};
    Cell r; u64 br = 0;
    for (int i = 0; i < 4; i++) { const unsigned __int128 d = (unsigned __int128)a.l[i] - P[i] - br; r.l[i] = (u64)d; br = (u64)(d >> 64) & 1; }
    return r;
}
DEV Cell add_mod(const Cell &a, const Cell &b) {          // a, b < p < 2^254: no carry out of 256 bits
    Cell r; u64 cy = 0;
    for (int i = 0; i < 4; i++) { const unsigned __int128 s = (unsigned __int128)a.l[i] + b.l[i] + cy; r.l[i] = (u64)s; cy = (u64)(s >> 64); }
    return geq_p(r) ? sub_p(r) : r;
}
DEV Cell neg_mod(const Cell &a) {                           // p - a (0 stays 0)
    if ((a.l[0] | a.l[1] | a.l[2] | a.l[3]) == 0) return a;
    const u64 P[4] = {0x43e1f593f0000001ull, 0x2833e84879b97091ull, 0xb85045b68181585dull, 0x30644e72e131a029ull};
    Cell r; u64 br = 0;
    for (int i = 0; i < 4; i++) { const unsigned __int128 d = (unsigned __int128)P[i] - a.l[i] - br; r.l[i] = (u64)d; br = (u64)(d >> 64) & 1; }
    return r;
}
// a * k for a 64-bit k: double-and-add from k's top bit
DEV Cell mul_small(const Cell &a, u64 k) {
    Cell r = small(0);
    if (k == 0) return r;
    for (int bit = 63 - __builtin_clzll(k); bit >= 0; bit--) {
        r = add_mod(r, r);
        if ((k >> bit) & 1) r = add_mod(r, a);
    }
    return r;
}
// a * b: through the operand that is small, or the negation of something small (the differences n - target,
// state_n - state_target, -2^16 of a frame are); both full width only for corrupted cells
DEV Cell mul_mod(const Cell &a, const Cell &b) {
    if (narrow(b)) return mul_small(a, b.l[0]);
    if (narrow(a)) return mul_small(b, a.l[0]);
    const Cell nb = neg_mod(b);
    if (narrow(nb)) return neg_mod(mul_small(a, nb.l[0]));
    const Cell na = neg_mod(a);
    if (narrow(na)) return neg_mod(mul_small(b, na.l[0]));
    Cell r = small(0);
    for (int bit = 255; bit >= 0; bit--) {
        r = add_mod(r, r);
        if ((b.l[bit >> 6] >> (bit & 63)) & 1) r = add_mod(r, a);
    }
    return r;
}
// x0 + x1*x2 == x3 (mod p) for canonical cells
DEV bool row_holds(const Cell x[4]) {
    if (narrow(x[0]) && narrow(x[1]) && narrow(x[2]) && narrow(x[3])) {
        const unsigned __int128 s = (unsigned __int128)x[1].l[0] * x[2].l[0] + x[0].l[0];
        return (u64)(s >> 64) == 0 && (u64)s == x[3].l[0];
    }
    if (geq_p(x[0]) || geq_p(x[1]) || geq_p(x[2]) || geq_p(x[3])) return false;     // not canonical
    return same(add_mod(x[0], mul_mod(x[1], x[2])), x[3]);
}

}  // namespace

template <bool MONT>
__global__ __launch_bounds__(256) void hsw_verify_kernel(VerifyParams p) {
    const u64 blk = blockIdx.x / p.slices;
    const u32 tid = (blockIdx.x % p.slices) * blockDim.x + threadIdx.x, nt = p.slices * blockDim.x;
    const uint4 *gate = reinterpret_cast<const uint4 *>(p.gate);
    const u64 dg = p.frame_every ? blk / p.frame_every : 0;
    const u64 g0 = p.gate_cell0 + blk * (u64)p.gate_cells + dg * p.frame_cells;
    const bool packed = p.n_breaks != 0;
    auto gcell = [&](u64 idx) -> Cell { return load_value<MONT>(gate, packed ? place(p, idx) : idx); };
    const uint8_t *bytes = p.blocks + 64 * blk;
    const u32 *pre = p.pre_states + 8 * blk;
    u32 bad = 0;
    u32 first = 0xffffffffu, first_class = 0;

    // a cell by structure id: stream cell or one of the cells outside the block's stream
    auto cell_of = [&](int64_t id, bool &known) -> Cell {
        known = true;
        if (id >= 0) return gcell(g0 + (u64)id);
        if (id >= -64) return small(bytes[-1 - id]);                        // input byte k = -1 - id
        if (id <= -100 && id >= -107) return small(pre[-100 - id]);          // pre-state word
        if (id == -1000) return small(0);                                    // the Context's zero cell
        known = false;                                                       // a halo2-base witness outside the stream
        return small(0);
    };
    auto fail = [&](u32 cls, u32 at) { bad++; if (at < first) { first = at; first_class = cls; } };

    // 1 + 2. gate rows x0 + x1*x2 = x3 (mod p), and -- on the same four loads -- what each of the row's cells
    //    must be: a fixed constant or a QuantumCell::Existing copy (every such cell sits in a gate row; the
    //    host checks that when it uploads the structure).  All-narrow rows are exact in 128 bits; the only
    //    rows with a full-width cell are the negations of ch: [a, p-a, 1, 0] and [M, p-a, 1, M-a]
    //    (compression.rs:320-335)
    //    Copies of stream cells are compared as stored (raw to raw: equal values have equal encodings, and a
    //    Montgomery stream needs no reduction for them).
    //    Work split: FOUR lanes per gate row, one per cell.  A quad then reads its row as 128 contiguous bytes
    //    and a wave instruction covers 16 rows in 16 lines -- with one lane per row every load instruction
    //    touched 64 different lines, 16 bytes of each, and the L1 had to keep them all until the row's eighth
    //    load (measured: 3.9 L2 requests per line of the stream, the kernel stalled on them 75 % of the time).
    //    Each lane checks its own cell (constant / copy: one source load per lane, all in flight together);
    //    the row equation gets the other three cells' low limbs by DPP quad broadcasts.
    auto raw_cell = [&](u64 idx) -> Cell { return load_cell(gate, packed ? place(p, idx) : idx); };
    auto quad64 = [](u64 v, int q) -> u64 {            // lane q of the quad's value, in every lane of the quad
        const int lo = (int)(u32)v, hi = (int)(u32)(v >> 32);
        int rl, rh;
        switch (q) {
            case 0: rl = __builtin_amdgcn_mov_dpp(lo, 0x00, 0xF, 0xF, true); rh = __builtin_amdgcn_mov_dpp(hi, 0x00, 0xF, 0xF, true); break;
            case 1: rl = __builtin_amdgcn_mov_dpp(lo, 0x55, 0xF, 0xF, true); rh = __builtin_amdgcn_mov_dpp(hi, 0x55, 0xF, 0xF, true); break;
            case 2: rl = __builtin_amdgcn_mov_dpp(lo, 0xAA, 0xF, 0xF, true); rh = __builtin_amdgcn_mov_dpp(hi, 0xAA, 0xF, 0xF, true); break;
            default: rl = __builtin_amdgcn_mov_dpp(lo, 0xFF, 0xF, 0xF, true); rh = __builtin_amdgcn_mov_dpp(hi, 0xFF, 0xF, 0xF, true); break;
        }
        return (u64)(u32)rl | ((u64)(u32)rh << 32);
    };
    const u32 j4 = threadIdx.x & 3u;                   // this lane's cell of the row
    const u32 slot = tid >> 2, nslots = nt >> 2;       // row slots of the launch slice (nt is a multiple of 4)
    for (u32 rb = 0; rb < p.n_rows; rb += nslots) {    // the same trip count in every lane: DPP needs whole quads
        const u32 r = rb + slot;
        const bool act = r < p.n_rows;
        const u32 c = p.gate_rows[act ? r : 0u];
        const u32 cell = c + j4;
        const Cell raw = raw_cell(g0 + cell);
        const uint8_t k = p.kind[cell];
        const int64_t rf = p.ref[cell];
        Cell w = raw;
        if (act && k == 2 && rf >= 0) w = raw_cell(g0 + (u64)rf);
        Cell x;
        if constexpr (MONT) x = from_mont(raw); else x = raw;
        // ---- the row: x0 + x1*x2 = x3
        const u64 l0 = quad64(x.l[0], 0), l1 = quad64(x.l[0], 1), l2 = quad64(x.l[0], 2), l3 = quad64(x.l[0], 3);
        const u64 up = x.l[1] | x.l[2] | x.l[3];        // 0 <=> this cell is narrow
        const u64 up0 = quad64(up, 0), up1 = quad64(up, 1), up2 = quad64(up, 2), up3 = quad64(up, 3);
        bool ok;
        if ((up0 | up1 | up2 | up3) == 0) {
            const unsigned __int128 s128 = (unsigned __int128)l1 * l2 + l0;
            ok = (u64)(s128 >> 64) == 0 && (u64)s128 == l3;
        } else {
            // the only rows with a full-width cell: [a, p-a, 1, 0] and [M, p-a, 1, M-a] (compression.rs:320-335)
            const u64 P0 = 0x43e1f593f0000001ull, P1 = 0x2833e84879b97091ull, P2 = 0xb85045b68181585dull, P3 = 0x30644e72e131a029ull;
            const u64 x11 = quad64(x.l[1], 1), x12 = quad64(x.l[2], 1), x13 = quad64(x.l[3], 1);
            const u64 a = P0 - l1;                                            // x1 = p - a
            ok = (up0 | up2 | up3) == 0 && l2 == 1 && x11 == P1 && x12 == P2 && x13 == P3 && a >= 1 && a <= 0x55555555ull &&
                 l0 >= a && l0 - a == l3;
        }
        if (act && !ok && j4 == 0) fail(VERIFY_GATE_ROW, c);
        // ---- this lane's cell
        if constexpr (MONT)            // a Montgomery cell is an encoding m < p: m + p reduces to the same value and would pass everything below
            if (act && geq_p(raw)) fail(VERIFY_RANGE, cell);
        if (act) {
            if (k == 1) { if (!same(x, small((u64)rf))) fail(VERIFY_CONSTANT, cell); }
            else if (k == 2) {
                if (rf >= 0) { if (!same(raw, w)) fail(VERIFY_COPY, cell); }
                else { bool known; const Cell e = cell_of(rf, known); if (known && !same(x, e)) fail(VERIFY_COPY, cell); }
            }
        }
    }
    // 3. assert_equal / range_check accumulator copies
    for (u32 i = tid; i < p.n_assert_eq; i += nt) {
        bool ka, kb;
        const Cell a = cell_of(p.assert_eq[2 * i], ka), b = cell_of(p.assert_eq[2 * i + 1], kb);
        if (ka && kb && !same(a, b)) fail(VERIFY_ASSERT_EQ, (u32)(p.assert_eq[2 * i] >= 0 ? p.assert_eq[2 * i] : p.assert_eq[2 * i + 1]));
    }
    // 4. range_check bounds
    for (u32 i = tid; i < p.n_range; i += nt) {
        bool known;
        const Cell v = cell_of(p.range[2 * i], known);
        const int64_t bits = p.range[2 * i + 1];
        if (known && !(narrow(v) && (bits >= 64 || (v.l[0] >> bits) == 0))) fail(VERIFY_RANGE, (u32)p.range[2 * i]);
    }
    // 5. spread chip: limb call n of this block is absolute call N = cursor0 + blk*LC + n -> column N % ncols,
    //    row N / ncols (spread.rs:202-231); the cells are tied to gate cells and form a row of the spread table.
    //    Two lanes per limb call -- the dense pair and the spread pair -- each with one chip cell and one gate cell
    //    to load (compared as stored); the table relation takes the partner's low limb by a DPP swap.
    if (p.chip_dense) {
        const uint4 *cd = reinterpret_cast<const uint4 *>(p.chip_dense), *csp = reinterpret_cast<const uint4 *>(p.chip_spread);
        const u64 row0 = p.cursor0 / p.ncols;
        const u32 half = threadIdx.x & 1u;                                    // 0: dense, 1: spread
        auto swap32 = [](u32 v) -> u32 { return (u32)__builtin_amdgcn_mov_dpp((int)v, 0xB1, 0xF, 0xF, true); };   // quad_perm [1,0,3,2]
        for (u32 nb = 0; nb < p.limb_calls; nb += nt >> 1) {                  // the same trip count in every lane
            const u32 n = nb + (tid >> 1);
            const bool act = n < p.limb_calls;
            const u32 nn = act ? n : 0u;
            const u64 N = p.cursor0 + blk * (u64)p.limb_calls + nn;
            const u64 at = (N % p.ncols) * (u64)p.chip_col_stride + (N / p.ncols - row0);
            const Cell rv = load_cell(half ? csp : cd, at);
            const int64_t id = p.chip[2 * nn + half];
            bool tied;
            Cell v;
            if constexpr (MONT) v = from_mont(rv); else v = rv;
            if (id >= 0) tied = same(rv, raw_cell(g0 + (u64)id));
            else { bool known; const Cell e = cell_of(id, known); tied = !known || same(v, e); }
            const u32 lo = (u32)v.l[0], hi = (u32)(v.l[0] >> 32);
            const u32 plo = swap32(lo), phi = swap32(hi);                     // the partner's low limb
            bool ok = tied && narrow(v);
            if constexpr (MONT) ok = ok && !geq_p(rv);                          // canonical encoding only
            if (half) ok = ok && phi == 0 && (u64)spread16(plo) == v.l[0];    // (dense, spread) is a row of the table
            else ok = ok && v.l[0] < (1ull << p.num_bits_lookup);
            const u32 both = (ok ? 1u : 0u) & swap32(ok ? 1u : 0u);
            if (act && half == 0 && !both) fail(VERIFY_CHIP, (u32)p.chip[2 * nn + 1]);
        }
    }
    // 6. lookup-advice column: entry j copies its source cell and is a 16-bit range-table entry
    if (p.lookup) {
        const uint4 *lk = reinterpret_cast<const uint4 *>(p.lookup);
        for (u32 j = tid; j < p.lookup_cells; j += nt) {
            bool known;
            const Cell src = cell_of(p.lookup_src[j], known);
            const Cell rv = load_cell(lk, p.lookup_cell0 + blk * (u64)p.lookup_cells + dg * p.frame_lookups + j);
            Cell v = rv;
            bool enc = true;
            if constexpr (MONT) { v = from_mont(rv); enc = !geq_p(rv); }
            if (!(enc && narrow(v) && v.l[0] < 65536 && (!known || same(v, src)))) fail(VERIFY_LOOKUP, j);
        }
    }
    // 7. next-state words
    if (p.next_states && tid < 8) {
        bool known;
        const Cell v = cell_of(p.next_state_cells[tid], known);
        if (!same(v, small(p.next_states[8 * blk + tid]))) fail(VERIFY_NEXT_STATE, (u32)p.next_state_cells[tid]);
    }
    if (bad) {
        atomicAdd(reinterpret_cast<unsigned long long *>(&p.report->violations), (unsigned long long)bad);
        // first failing (block, cell, class): smallest packed key wins
        const unsigned long long key = ((unsigned long long)blk << 36) | ((unsigned long long)first << 4) | first_class;
        atomicMin(reinterpret_cast<unsigned long long *>(&p.report->first_key), key);
    }
}

// ---------------------------------------------------------------- digest frames
// One workgroup per digest: prologue and epilogue cells against their structure, plus the links a
// replayer would make by copy constraints and this check makes through the arrays both sides were checked
// against: input length / rounds, the initial state, the input bytes, pre-state of block b = next state of
// block b - 1, the candidate states of the epilogue.
template <bool MONT>
__global__ __launch_bounds__(256) void hsw_verify_frame_kernel(FrameVerifyParams p) {
    const FrameDesc d = p.descs[blockIdx.x];
    const u32 tid = threadIdx.x, nt = blockDim.x;
    const uint4 *gate = reinterpret_cast<const uint4 *>(p.gate);
    const uint4 *lk = reinterpret_cast<const uint4 *>(p.lookup);
    const bool packed = p.n_breaks != 0;
    auto gcell = [&](u64 idx) -> Cell { return load_value<MONT>(gate, packed ? place(p, idx) : idx); };
    u32 bad = 0, first = 0xffffffffu, first_class = 0;
    auto fail = [&](u32 cls, u32 at) { bad++; if (at < first) { first = at; first_class = cls; } };
    const u32 N = d.n_blocks;
    const u32 target = d.num_round - d.precomputed_round;
    auto state_word = [&](u32 n, u32 i) -> u64 {
        return n == 0 ? p.pre_states[8 * d.first_block + i] : p.next_states[8 * (d.first_block + n - 1) + i];
    };

    for (int sec = 0; sec < 2; sec++) {
        const FrameVerifyParams::Section &S = sec ? p.epi : p.pro;
        const u64 base = sec ? d.epilogue_cell : d.prologue_cell;
        const u64 lbase = sec ? d.epilogue_lookup : d.prologue_lookup;
        const u32 tag = sec ? 0x40000000u : 0u;            // reported cell: section-relative, epilogue flagged
        // a cell by structure id (section-relative, or a cell of another section)
        auto cell_of = [&](int64_t id) -> Cell {
            if (id >= 0) return gcell(base + (u64)id);
            if (id == FS_ZERO) return small(0);
            if (id == FS_TARGET) return gcell(d.prologue_cell + frame::P_TGT);
            const u32 q = (u32)(FS_STATE0 - id);                       // 8 n + i
            return q < 8 ? gcell(d.prologue_cell + frame::P_STATE + q) : small(state_word(q / 8, q % 8));
        };
        for (u32 r = tid; r < S.n_rows; r += nt) {
            const u32 c = S.gate_rows[r];
            Cell x[4];
            for (int j = 0; j < 4; j++) x[j] = gcell(base + c + j);
            if (!row_holds(x)) fail(VERIFY_GATE_ROW, tag | c);
        }
        for (u32 c = tid; c < S.cells; c += nt) {
            const uint8_t k = S.kind[c];
            if (k == 0) continue;
            const Cell v = gcell(base + c);
            if (k == 1) {
                const int64_t kv = S.ref[c];
                Cell want = small((u64)(kv >= 0 ? kv : -kv));
                if (kv < 0) {                                                       // p - |k|, |k| < 2^62: only limb 0 borrows
                    want.l[0] = 0x43e1f593f0000001ull - want.l[0];
                    want.l[1] = 0x2833e84879b97091ull; want.l[2] = 0xb85045b68181585dull; want.l[3] = 0x30644e72e131a029ull;
                }
                if (!same(v, want)) fail(VERIFY_CONSTANT, tag | c);
            } else if (!same(v, cell_of(S.ref[c]))) fail(VERIFY_COPY, tag | c);
        }
        for (u32 i = tid; i < S.n_assert_eq; i += nt)
            if (!same(cell_of(S.assert_eq[2 * i]), cell_of(S.assert_eq[2 * i + 1]))) fail(VERIFY_ASSERT_EQ, tag | (u32)S.assert_eq[2 * i + 1]);
        for (u32 i = tid; i < S.n_assert_const; i += nt)
            if (!same(cell_of(S.assert_const[2 * i]), small((u64)S.assert_const[2 * i + 1]))) fail(VERIFY_CONSTANT, tag | (u32)S.assert_const[2 * i]);
        for (u32 i = tid; i < S.n_range; i += nt) {
            const Cell v = cell_of(S.range[2 * i]);
            if (!(narrow(v) && (v.l[0] >> S.range[2 * i + 1]) == 0)) fail(VERIFY_RANGE, tag | (u32)S.range[2 * i]);
        }
        if (lk)
            for (u32 j = tid; j < S.n_lookup; j += nt) {
                const Cell v = load_value<MONT>(lk, lbase + j);
                if (!(narrow(v) && v.l[0] < 65536 && same(v, cell_of(S.lookup_src[j])))) fail(VERIFY_LOOKUP, tag | j);
            }
    }
    // ---- the facts of this digest and the links between the sections ----
    const u64 P0 = d.prologue_cell;
    if (tid == 0) {
        if (!same(gcell(P0 + frame::P_LEN), small(d.input_len))) fail(VERIFY_COPY, frame::P_LEN);           // AssignedHashResult.input_len
        if (!same(gcell(P0 + frame::P_PRE), small(d.precomputed_round))) fail(VERIFY_COPY, frame::P_PRE);
        if (d.zero_cell != ~0ull && !same(gcell(d.zero_cell), small(0))) fail(VERIFY_CONSTANT, frame::P_BYTES - 1);
        (void)target;
    }
    if (tid < 8 && !same(gcell(P0 + frame::P_STATE + tid), small(state_word(0, tid)))) fail(VERIFY_COPY, frame::P_STATE + tid);
    for (u32 i = tid; i < 64u * N; i += nt)                                                                  // input bytes
        if (!same(gcell(P0 + frame::P_BYTES + i), small(p.blocks[64 * d.first_block + i]))) fail(VERIFY_COPY, frame::P_BYTES + i);
    for (u32 i = tid; i < 8u * (N - 1); i += nt)                                                             // the chain
        if (p.pre_states[8 * (d.first_block + 1) + i] != p.next_states[8 * d.first_block + i]) fail(VERIFY_NEXT_STATE, i);
    if (bad) {
        atomicAdd(reinterpret_cast<unsigned long long *>(&p.report->violations), (unsigned long long)bad);
        const unsigned long long key = ((unsigned long long)d.first_block << 36) | ((unsigned long long)first << 4) | first_class;
        atomicMin(reinterpret_cast<unsigned long long *>(&p.report->first_key), key);
    }
}

hipError_t launch_verify_frames(const FrameVerifyParams &p, size_t n_digests, hipStream_t stream) {
    if (n_digests == 0) return hipSuccess;
    if (p.montgomery) hipLaunchKernelGGL(hsw_verify_frame_kernel<true>, dim3((unsigned)n_digests), dim3(256), 0, stream, p);
    else hipLaunchKernelGGL(hsw_verify_frame_kernel<false>, dim3((unsigned)n_digests), dim3(256), 0, stream, p);
    return hipGetLastError();
}

hipError_t launch_verify(const VerifyParams &p, size_t n_blocks, hipStream_t stream) {
    if (n_blocks == 0) return hipSuccess;
    if (p.montgomery) hipLaunchKernelGGL(hsw_verify_kernel<true>, dim3((unsigned)(n_blocks * p.slices)), dim3(256), 0, stream, p);
    else hipLaunchKernelGGL(hsw_verify_kernel<false>, dim3((unsigned)(n_blocks * p.slices)), dim3(256), 0, stream, p);
    return hipGetLastError();
}

}  // namespace hsw
