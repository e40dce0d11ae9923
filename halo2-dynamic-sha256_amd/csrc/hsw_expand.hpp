// hsw_expand.hpp -- gfx950 (MI355X / CDNA4) kernels of the SHA-256 witness engine.
//
// One 64-lane wavefront per work item (workgroup = one wave); a block is one
// work item, or is dealt to 2..32 of them (`parts`): each takes a share of the
// units of every phase, or -- tiny batches, "split" mode -- the units of one phase.
//
//  chain phase   the plain SHA-256 recurrence of the block (W[0..63] and the a/e
//                value born in every round) is computed once, wave-uniform, and
//                staged in LDS; every lane then pulls the seeds of ITS units
//                into registers.  From these seeds every unit of the gadget (a
//                schedule step, a round, ...) is independent.
//  expand phase  lane = unit.  Every lane runs the same straight-line program --
//                the reference's gate-call sequence for that unit
//                (compression.rs:57-96 for a schedule step, :125-196 for a
//                round) -- and appends each gate cell to its own row of an
//                [R][T] LDS tile of 64-bit values.  WHERE a cell goes is a
//                compile-time cursor type (Cur<POS, FLUSHES, NEG0..NEG3, CN>)
//                threaded through every gate function, so an emitted cell is
//                one ds_write_b64 at an immediate offset and each flush point
//                is an `if constexpr`.  Spread/dense conversions are shift/mask
//                bit interleaves, no table reads.
//  write-out     when a tile is full the wave transposes it to HBM: for every
//                row, lanes store consecutive 16-byte pieces, i.e. each
//                global_store_dwordx4 wave-instruction writes 1 KiB contiguous
//                (canonical form).  Montgomery form (x * 2^256 mod p,
//                halo2curves' in-memory Fr) does one 64x256-bit
//                multiply + Barrett reduce per cell, one lane per cell.
//                A stream that does not start on a 128-byte line (digest frames,
//                column breaks) is realigned: skewed tile columns, carried cells
//                and held-back unit heads keep every run on whole lines (flush_tile).
//  chip columns  the 16-bit dense input of every SpreadConfig::spread call is
//                staged in LDS; at the end of each phase the chip columns
//                denses[c] / spreads[c] (spread.rs:196-233) receive one
//                contiguous run of rows per column.
//
// Pure 32/64-bit integer work, write-streaming: the bound is HBM write
// bandwidth (DESIGN.md "Roofline").  No MFMA.
#ifndef HSW_EXPAND_HPP
#define HSW_EXPAND_HPP
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "hsw_flush_bounds.hpp"

#include "hsw_kernels.h"
#include "hsw_layout.h"
#include "hsw_mont.hpp"

namespace hsw {

typedef unsigned int u32;
typedef unsigned long long u64;
typedef unsigned short u16;

// FIPS 180-4 round constants (reference compression.rs:992-1001: K enters the
// circuit as a gate constant).
static __constant__ u32 K256[64] = {
    0x428a2f98, 0x71374491, 0xb5c0fbcf, 0xe9b5dba5, 0x3956c25b, 0x59f111f1, 0x923f82a4, 0xab1c5ed5,
    0xd807aa98, 0x12835b01, 0x243185be, 0x550c7dc3, 0x72be5d74, 0x80deb1fe, 0x9bdc06a7, 0xc19bf174,
    0xe49b69c1, 0xefbe4786, 0x0fc19dc6, 0x240ca1cc, 0x2de92c6f, 0x4a7484aa, 0x5cb0a9dc, 0x76f988da,
    0x983e5152, 0xa831c66d, 0xb00327c8, 0xbf597fc7, 0xc6e00bf3, 0xd5a79147, 0x06ca6351, 0x14292967,
    0x27b70a85, 0x2e1b2138, 0x4d2c6dfc, 0x53380d13, 0x650a7354, 0x766a0abb, 0x81c2c92e, 0x92722c85,
    0xa2bfe8a1, 0xa81a664b, 0xc24b8b70, 0xc76c51a3, 0xd192e819, 0xd6990624, 0xf40e3585, 0x106aa070,
    0x19a4c116, 0x1e376c08, 0x2748774c, 0x34b0bcb5, 0x391c0cb3, 0x4ed8aa4a, 0x5b9cca4f, 0x682e6ff3,
    0x748f82ee, 0x78a5636f, 0x84c87814, 0x8cc70208, 0x90befffa, 0xa4506ceb, 0xbef9a3f7, 0xc67178f2};

static __device__ __constant__ u32 IV256[8] = {0x6a09e667, 0xbb67ae85, 0x3c6ef372, 0xa54ff53a,
                                        0x510e527f, 0x9b05688c, 0x1f83d9ab, 0x5be0cd19};

// BN254 Fr modulus p: HSW_P0..7, 32-bit little-endian limbs (hsw_mont.hpp).  -x for 0 < x <= 0x55555555
// only touches limb 0: p[0] = 0xf0000001 > x, so there is no borrow.

#define DEV __device__ __forceinline__

// Tuning aid (tools/latency_probe): with -DHSW_STAMPS every wave records the 100 MHz wall clock at its
// phase boundaries.  Never defined in the product build.
#ifdef HSW_STAMPS
__device__ unsigned long long *g_hsw_stamps = nullptr;
// slot 15: where the wave ran -- XCC_ID (hwreg 20) << 32 | HW_ID (hwreg 4: wave, simd, cu, sh, se)
#define HSW_STAMP(i) do { if (threadIdx.x == 0 && g_hsw_stamps) { g_hsw_stamps[(size_t)blockIdx.x * 16 + (i)] = wall_clock64(); \
    if ((i) == 0) g_hsw_stamps[(size_t)blockIdx.x * 16 + 15] = ((unsigned long long)__builtin_amdgcn_s_getreg(63508) << 32) | __builtin_amdgcn_s_getreg(63492); } } while (0)
#else
#define HSW_STAMP(i) do { } while (0)
#endif

DEV u32 rotr32(u32 x, int n) { return __builtin_amdgcn_alignbit(x, x, n); }

// dense 16 bits -> 32 bits with bit i at position 2i (the "spread" form,
// reference spread.rs:211-218 / table rows of :165-194), by shifts and masks.
DEV u32 spread16(u32 x) {
    x &= 0xffffu;
    x = (x | (x << 8)) & 0x00ff00ffu;
    x = (x | (x << 4)) & 0x0f0f0f0fu;
    x = (x | (x << 2)) & 0x33333333u;
    x = (x | (x << 1)) & 0x55555555u;
    return x;
}
DEV u64 spread32(u32 x) { return (u64)spread16(x) | ((u64)spread16(x >> 16) << 32); }
// even-position bits of a 32-bit value packed into 16 bits
// (decompose_even_and_odd_unchecked, spread.rs:146-157; odd = even_bits(x >> 1)).
DEV u32 even_bits(u32 x) {
    x &= 0x55555555u;
    x = (x | (x >> 1)) & 0x33333333u;
    x = (x | (x >> 2)) & 0x0f0f0f0fu;
    x = (x | (x >> 4)) & 0x00ff00ffu;
    x = (x | (x >> 8)) & 0x0000ffffu;
    return x;
}

// The SHA-256 functions on dense words, three-input boolean ops as one v_bitop3 each.
DEV u32 sha_S1(u32 e) { return __builtin_amdgcn_bitop3_b32(rotr32(e, 6), rotr32(e, 11), rotr32(e, 25), 0x96); }
DEV u32 sha_S0(u32 a) { return __builtin_amdgcn_bitop3_b32(rotr32(a, 2), rotr32(a, 13), rotr32(a, 22), 0x96); }
DEV u32 sha_s0(u32 w) { return __builtin_amdgcn_bitop3_b32(rotr32(w, 7), rotr32(w, 18), w >> 3, 0x96); }
DEV u32 sha_s1(u32 w) { return __builtin_amdgcn_bitop3_b32(rotr32(w, 17), rotr32(w, 19), w >> 10, 0x96); }
DEV u32 sha_ch(u32 e, u32 f, u32 g) { return __builtin_amdgcn_bitop3_b32(e, f, g, 0xca); }    // e ? f : g
DEV u32 sha_maj(u32 a, u32 b, u32 c) { return __builtin_amdgcn_bitop3_b32(a, b, c, 0xe8); }

// ---------------------------------------------------------- Montgomery form
// HSW_REPR_MONTGOMERY: a cell holds x * 2^256 mod p, halo2curves' in-memory Fr.
// For x = lo + hi*2^32 < 2^64:  x*R mod p = lo*R + hi*R32 (mod p) with the
// constants R = 2^256 mod p (~0.29 p) and R32 = 2^288 mod p (~0.14 p), so
// t = lo*R + hi*R32 < 0.43 * 2^32 * p: the quotient fits 32 bits, and one
// Barrett step with mu = floor(2^288/p) on the top 64 bits of t gives q^ = q or
// q - 1 (t/p - th*mu/2^64 < 2^-29 + 0.43), i.e. r = t - q^ p < 2p: exactly one
// conditional subtraction.  32-bit limbs with explicit carry chains
// (v_addc_co / v_subb_co): the u64 formulation cost 316 instructions per cell,
// this one ~170.
#define HSW_MU 0x54a474626ull      /* floor(2^288 / p) */

template <bool HAS_HI>
DEV Fe8 mont_from_u64(u32 lo, u32 hi) {
    const u32 P[8] = {HSW_P0, HSW_P1, HSW_P2, HSW_P3, HSW_P4, HSW_P5, HSW_P6, HSW_P7};
    const u32 RR[8] = {0x4ffffffbu, 0xac96341cu, 0x9f60cd29u, 0x36fc7695u,       // 2^256 mod p
                       0x7879462eu, 0x666ea36fu, 0x9a07df2fu, 0x0e0a77c1u};
    const u32 R32[8] = {0x15b8b9dau, 0x93e78865u, 0xb05ea154u, 0x16df2426u,      // 2^288 mod p
                        0x302ab839u, 0x1271b743u, 0xec6c226eu, 0x06bc037eu};
    // Every product below is a 32 x 32 + 64-bit multiply-add (v_mad_u64_u32), the carry travelling in the high
    // word of the accumulator: (2^32 - 1)^2 + 2^32 - 1 < 2^64, so a chain never overflows.
    u32 t[9];
    {   // t = lo * R
        u64 acc = 0;
#pragma unroll
        for (int j = 0; j < 8; j++) { acc = (u64)lo * RR[j] + (acc >> 32); t[j] = (u32)acc; }
        t[8] = (u32)(acc >> 32);
    }
    if (HAS_HI) {   // t += hi * R32   (t < 2^33 p < 2^288 afterwards: nine limbs hold it)
        u32 u[9];
        u64 acc = 0;
#pragma unroll
        for (int j = 0; j < 8; j++) { acc = (u64)hi * R32[j] + (acc >> 32); u[j] = (u32)acc; }
        u[8] = (u32)(acc >> 32);
        u32 cy = 0;
#pragma unroll
        for (int j = 0; j < 8; j++) t[j] = __builtin_addc(t[j], u[j], cy, &cy);
        t[8] += u[8] + cy;
    }
    // q = floor((t >> 224) * floor(2^288 / p) / 2^64), the multiplier being 5 * 2^32 + MU0      (q < 0.43 * 2^32)
    constexpr u32 MU0 = (u32)HSW_MU;
    static_assert((HSW_MU >> 32) == 5ull, "Barrett multiplier");
    u64 mid = (u64)t[7] * MU0;
    mid = (u64)t[8] * MU0 + (mid >> 32);
    mid = (u64)t[7] * 5u + mid;
    u32 q = t[8] * 5u + (u32)(mid >> 32);
    asm("" : "+v"(q));      // a 32-bit value from here on (otherwise the sum is widened and q * p becomes 64 x 256 bits)
    // r = (t - q*p) mod 2^256 (the true value is < 2p < 2^256)
    Fe8 r;
    {
        u64 acc = 0;
        u32 br = 0;
#pragma unroll
        for (int j = 0; j < 8; j++) {
            acc = (u64)q * P[j] + (acc >> 32);
            r.l[j] = __builtin_subc(t[j], (u32)acc, br, &br);
        }
    }
    u32 br;
    u32 sub[8];
    br = 0;
#pragma unroll
    for (int j = 0; j < 8; j++) sub[j] = __builtin_subc(r.l[j], P[j], br, &br);
#pragma unroll
    for (int j = 0; j < 8; j++) r.l[j] = br ? r.l[j] : sub[j];
    return r;
}
// K[r] * 2^256 mod p: the round constants as gate constants in Montgomery form (compression.rs:151)
static __device__ const Fe8 K256M[64] = {
#define HSW_KM(a, b, c, d, e, f, g, h) mont_const(a), mont_const(b), mont_const(c), mont_const(d), mont_const(e), mont_const(f), mont_const(g), mont_const(h)
    HSW_KM(0x428a2f98, 0x71374491, 0xb5c0fbcf, 0xe9b5dba5, 0x3956c25b, 0x59f111f1, 0x923f82a4, 0xab1c5ed5),
    HSW_KM(0xd807aa98, 0x12835b01, 0x243185be, 0x550c7dc3, 0x72be5d74, 0x80deb1fe, 0x9bdc06a7, 0xc19bf174),
    HSW_KM(0xe49b69c1, 0xefbe4786, 0x0fc19dc6, 0x240ca1cc, 0x2de92c6f, 0x4a7484aa, 0x5cb0a9dc, 0x76f988da),
    HSW_KM(0x983e5152, 0xa831c66d, 0xb00327c8, 0xbf597fc7, 0xc6e00bf3, 0xd5a79147, 0x06ca6351, 0x14292967),
    HSW_KM(0x27b70a85, 0x2e1b2138, 0x4d2c6dfc, 0x53380d13, 0x650a7354, 0x766a0abb, 0x81c2c92e, 0x92722c85),
    HSW_KM(0xa2bfe8a1, 0xa81a664b, 0xc24b8b70, 0xc76c51a3, 0xd192e819, 0xd6990624, 0xf40e3585, 0x106aa070),
    HSW_KM(0x19a4c116, 0x1e376c08, 0x2748774c, 0x34b0bcb5, 0x391c0cb3, 0x4ed8aa4a, 0x5b9cca4f, 0x682e6ff3),
    HSW_KM(0x748f82ee, 0x78a5636f, 0x84c87814, 0x8cc70208, 0x90befffa, 0xa4506ceb, 0xbef9a3f7, 0xc67178f2)
#undef HSW_KM
};

// Lane within the wave.  Workgroups are one wave, except the small-batch kernel's Montgomery instantiation,
// whose workgroup is up to 8 waves sharing one tile (Em::HELPERS); with __launch_bounds__(64) the mask folds away.
DEV u32 lane_id() { return threadIdx.x & 63u; }
// Makes the LDS writes of a tile's emitter visible to whoever writes the tile out.
// Workgroup barrier that orders LDS only.  __syncthreads() is a workgroup-scope fence over ALL memory: the wave
// first waits for vmcnt(0), i.e. until every store it has issued has landed in HBM -- in a write-streaming wave
// that is the one thing it must never wait for.
DEV void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
template <class EM>
DEV void em_sync() {
#ifdef HSW_SYNCTHREADS_EVERYWHERE          /* A/B: the round-2 behaviour */
    __syncthreads();
    return;
#endif
    if constexpr (EM::HELPERS) {
        lds_barrier();                     // several waves share the tile (small-batch kernel)
    } else {
        // one wave owns the tile (the only wave of its workgroup): only the compiler needs telling -- the LDS
        // unit executes the instructions of one wave in order
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
}

// ------------------------------------------------------------------ emitter
// Emission state of one lane.  `row`, `active`, `call`, `unit` are per lane; the
// rest is wave-uniform.  WHERE a cell goes inside the tile is not state at all:
// it is the compile-time cursor type below, so every emitted cell is one
// ds_write_b64 at an immediate offset and every flush point is an `if constexpr`.
// REPR: 0 = canonical 32-byte cells, 1 = Montgomery 32-byte cells converted at write-out (one lane per cell),
// 2 = compact 8-byte cells (low 64 bits; the negation cells hold x where the field value is -x -- hsw.h
// HSW_REPR_COMPACT64), 3 = Montgomery 32-byte cells converted at EMIT time (M32): the tile holds finished
// 32-byte cells, every distinct value of a unit is converted once by the lane that owns the unit, copies and
// constants cost no arithmetic (hsw_mont.hpp), and the write-out is the plain transposing copy of the canonical
// form.  Same bytes in HBM as REPR 1.
template <int T, int R, int REPR_, bool RC_, bool NO_REALIGN_ = false, bool EMITS_ = true>
struct Em {
    static constexpr int TILE = T, ROWS = R;
    static constexpr int REPR = REPR_;
    static constexpr bool MONT = REPR_ == 1;
    static constexpr bool COMPACT = REPR_ == 2;
    static constexpr bool M32 = REPR_ == 3;
    static constexpr bool RC = RC_;   // halo2-base internals: range_check cells + lookup-column stream (A3)
    // Realigned write-out (flush_tile).  Free for the HBM-bound canonical kernels; the Montgomery kernels
    // are issue-bound and pay ~4 % for it even on aligned streams, so they realign only in internals mode,
    // where misaligned streams are the rule (digest frames, column images); compact cells never do.
    // (NO_REALIGN_: the small-batch kernel of hsw_small.hpp -- latency-bound launches, rows are sub-units whose
    //  tails and heads are not neighbours in the stream.)
    static constexpr bool REALIGN = !NO_REALIGN_ && (REPR_ == 0 || ((REPR_ == 1 || REPR_ == 3) && RC_));
    static constexpr bool MONT_OUT = MONT || M32;             // chip / lookup cells are converted where they are written

    static constexpr int STRIDE = REALIGN ? T + 3 : T + 1;   // u64 per tile row: T cells + up to 3 carried ones (odd: no bank conflicts)
    // M32: a tile cell is 32 bytes = CW words; a row is its cells (+ 3 carried ones) + 16 bytes, so that the
    // rows of consecutive lanes start in different 16-byte bank groups (ds_write_b128 / ds_read_b128)
    static constexpr int CW = M32 ? 4 : 1;
    static constexpr int STRIDE_W = M32 ? 4 * (REALIGN ? T + 3 : T) + 2 : STRIDE;   // u64 words per tile row
    static constexpr int HEAD_W = M32 ? 12 : 3;               // words per row of the held-back heads
    // Helper waves (small-batch kernel): the workgroup is hcnt waves sharing one tile.  Wave 0 is the emitter;
    // the others run the same role program instantiated with EMITS = false -- nothing is staged, so the
    // compiler drops the arithmetic and what remains is the role's sequence of flushes.  At a flush every wave,
    // the emitter included, takes every hcnt-th row / 64-cell run: a launch this small is paced by one wave's
    // instruction count, and the write-out is most of it (two thirds in canonical form, more in Montgomery
    // form).  hsel = this wave's index (wave-uniform: the flush loops stay scalar).
    // Measured and dropped: helper waves as extra workgroups (slower beyond 2: workgroup dispatch, ~5 ns each,
    // paces the launch); helpers that repeat the emission (27 vs 32 us per 16 Montgomery blocks: the VALUs of
    // 592 workgroups are the bound); double-buffered tiles with a dedicated emitter wave (the emission is the
    // smaller part, so a wave that only emits is a flusher lost: 25.0 vs 22.8 us); a launch bound of 512
    // threads (every wave of the kernel crawled, 57 vs 33 us).
    static constexpr bool HELPERS = NO_REALIGN_;
    static constexpr bool EMITS = EMITS_;
    u32 hsel, hcnt;
    u64 *row0;         // this lane's tile row (LDS), column 0
    u32 skew;          // 0..3, wave-uniform: cells by which this phase's units start past a 128-byte line
    u32 carry_neg;     // bit j: carried column j holds a field negation
    u64 *head;         // [R][3] (LDS): the first 4 - skew cells of every unit, held back until the last flush
    u64 *row;          // row0 + skew: where cell POS of the current tile goes
    u64 *tile;         // tile base (LDS)
    u16 *d16;          // staged dense inputs of spread() calls (LDS)
    uint4 *out;        // this block's gate stream, in 16-byte pieces
    u32 nrows;         // units (rows) of the current phase-part
    u32 unit_cells;    // gate cells per unit
    u32 cell_base;     // cell index (in block) of the phase-part's first unit, cell 0
    u32 call;          // this lane's next spread-call slot (index into d16, phase-local)
    u32 unit;          // the unit (word / schedule step / round ...) this lane expands
    u32 call_first;    // block-relative index of the phase-part's first spread call
    u32 calls;         // spread calls staged by this phase-part (nrows * calls_per_unit)
    bool active;       // lane < nrows
    bool write_gate;   // HSW_SKIP_GATE not set
    // FlexGate column packing: cells at block-local index >= brk1 / brk2 are shifted
    // by gap1 / gap2 more cells (the unused tail rows of a column); 0xffffffff = none
    u32 brk1, gap1, brk2, gap2;
    // lookup-advice column staging (RC only)
    u16 *lk16;         // LDS
    u32 lk;            // this lane's next slot (phase-local)
    u32 lk_first;      // block-relative index of the phase-part's first lookup cell
    u32 lks;           // lookup cells staged by this phase-part
    const uint4 *tab;  // M32: ExpandParams.mont_tab
};

// entry idx of ExpandParams.mont_tab (M32): Montgomery form of i | spread(i) | i << 8 at 0 | 256 | 512 + i
template <class EM> DEV Fe8 tab_entry(const EM &em, u32 idx) {
    const uint4 a = em.tab[2u * idx], b = em.tab[2u * idx + 1u];
    return Fe8{{a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w}};
}

// Compile-time emission cursor: POS = cells already in the current tile, FL =
// tiles flushed so far in this phase, NA..ND = the (at most four: ch has four per
// round, compression.rs:320-335) tile positions holding a field negation, -1 = none.
// CN: the previous tile had a negation among its last three positions, i.e. one may have been carried
// into this tile by the realigning flush (only then does the flush look at the runtime carry mask).
template <int POS, int FL, int NA = -1, int NB = -1, int NC = -1, int ND = -1, bool CN = false>
struct Cur {
    static constexpr int pos = POS, fl = FL;
    static constexpr int na = NA, nb = NB, nc = NC, nd = ND;
    static constexpr bool cn = CN;
};
using CurStart = Cur<0, 0>;

// Transpose `ncells` tile columns out to HBM: row r goes to cells
// [cell_base + r*unit_cells + seg, +ncells).  Canonical form: lane pairs cover one
// 32-byte cell (low / high 16 bytes), so a wave-wide store instruction writes
// 1 KiB contiguous.  Montgomery / compact form: one lane per cell.
// All addressing is a wave-uniform base pointer plus a 32-bit byte offset (one
// block's stream spans far less than 4 GiB even with column-break gaps), so a store costs a
// handful of VALU instructions instead of 64-bit pointer arithmetic per lane.
template <class EM>
DEV u32 packed_cell(const EM &em, u32 cl) {       // FlexGate column packing: add the gaps of the breaks passed
    return packed_cell_of(BlockBreaks{em.brk1, em.gap1, em.brk2, em.gap2}, cl);
}
DEV void store16(char *base, u32 byte_off, uint4 v) { *reinterpret_cast<uint4 *>(base + (size_t)byte_off) = v; }
DEV void store8(char *base, u32 byte_off, u64 v) { *reinterpret_cast<u64 *>(base + (size_t)byte_off) = v; }

// Gate cell -> Montgomery form.  8 % of a block's cells hold a value of 2^32 or more, and half of the 64-cell runs
// a wave converts at a time hold none: those take the one-multiplicand conversion (wave-uniform choice, 70 instead
// of 93 VALU instructions).
template <class EM>
DEV Fe8 mont_cell(u64 v) {
    if (__builtin_amdgcn_ballot_w64((u32)(v >> 32) != 0u) == 0ull) return mont_from_u64<false>((u32)v, 0u);
    return mont_from_u64<true>((u32)v, (u32)(v >> 32));
}

// Realignment.  HBM writes run at full rate only when every contiguous run covers whole
// 128-byte lines (measured: a stream starting 32 / 64 / 96 bytes past a line boundary runs
// 47 / 26 / 47 % slower, tools/align_probe.py) -- but where a block's stream starts is dictated
// by the circuit layout (digest frames, column breaks, odd max_rows).  So a unit that starts
// `skew` cells past a line boundary keeps its tile in LDS columns [skew, skew + T): a flush
// writes columns [0, T) -- a line-aligned window of the stream -- and carries the last `skew`
// cells over into columns [0, skew) of the next tile.  Only the first and last few cells of a
// unit are partial lines, and those meet the neighbouring unit's in L2 within the same flush.
template <class EM, bool FULL>
DEV void flush_tile(EM &em, u32 ncells, int fl, int na, int nb, int nc, int nd, bool cn) {
    constexpr int T = EM::TILE;
    constexpr int S = EM::STRIDE;
    const u32 lane = lane_id();
    char *base = reinterpret_cast<char *>(em.out);
    const u32 skew = EM::REALIGN ? em.skew : 0u;
    // helper waves (Em::HELPERS): wave hs of hn takes rows / 64-cell runs hs, hs + hn, ...
    const u64 *tile = em.tile;
    u32 hs = 0, hn = 1;
    if constexpr (EM::HELPERS) { hs = em.hsel; hn = em.hcnt; }
    // FlexGate column breaks inside this block (wave-uniform; none unless a pack plan is in force): a flush
    // whose cells all lie on one side of them is only SHIFTED by the gaps it has passed and keeps the fast
    // paths; only a flush that straddles a break places every piece on its own (`packed`).
    // (hsw_flush_bounds.hpp: plain integer functions, property-tested on the CPU -- tests/test_flush_bounds.py)
    const BlockBreaks bb{em.brk1, em.gap1, em.brk2, em.gap2};
    const u32 lo_c = flush_lo_cell(em.cell_base, (u32)fl, (u32)T, skew);          // LDS column 0 of row 0
    bool packed = false;
    const u32 shift = flush_shift(bb, lo_c, em.nrows, em.unit_cells, (u32)T, packed);
    // ... and inside a straddling flush every ROW is again either shifted as a whole or (at most two of them)
    // straddles a break itself: returns the row's shift, `strad` = place the row piece by piece
    auto row_shift = [&](u32 r, bool &strad) -> u32 { return flush_row_shift(bb, lo_c, r, em.unit_cells, (u32)T, strad); };
    // A skewed unit shares its first line with the previous unit's tail, which is written at the END of
    // the phase: the first 4 - skew cells of every unit but the wave's first are held back in em.head and
    // appended to the previous row's tail in the last flush, so that the shared line is completed within
    // one flush (a partial line costs a read-modify-write in HBM, and in a write-only stream the bus
    // turnaround of that read is worth ~10 lines).
    const bool hold_heads = FULL && fl == 0 && skew != 0u;
    const u32 hc = (!FULL && fl != 0 && skew != 0u) ? 4u - skew : 0u;    // head cells appended per row now
    const u32 hi = FULL ? (u32)T : ncells + skew;        // one past the last tile column written now
    if (hold_heads && lane >= 1u && lane < em.nrows)
        for (u32 j = skew; j < 4u; j++) em.head[lane * 3u + j - skew] = em.row0[j];
    if (hc != 0u && lane >= 1u && lane < em.nrows)       // memory after row r's tail is row r+1's head
        for (u32 j = 0; j < hc; j++) em.tile[(lane - 1u) * S + hi + j] = em.head[lane * 3u + j];
    em_sync<EM>();
    const u32 lo0 = fl == 0 ? skew : 0u;                 // first LDS column written now: row 0 ...
    const u32 lo = hold_heads ? 4u : lo0;                // ... and the other rows
    const u32 seg = (u32)fl * (u32)T - skew;             // column c holds unit cell seg + c (never used below lo)
    const u32 cneg = cn ? em.carry_neg : 0u;             // cn is a compile-time constant at every call site
    // is LDS column p a field negation?  tile position = p - skew; carried columns: bit mask
    auto is_neg = [&](u32 p) -> bool {
        if (skew == 0u)                                  // wave-uniform: the aligned case stays 4 compares
            return na >= 0 && (p == (u32)na || p == (u32)nb || p == (u32)nc || p == (u32)nd);
        if (p < skew) return fl != 0 && ((cneg >> p) & 1u);
        const u32 q = p - skew;
        return na >= 0 && (q == (u32)na || q == (u32)nb || q == (u32)nc || q == (u32)nd);
    };
    const bool any_neg = na >= 0 || cneg != 0u;
    const bool first_skewed = FULL && fl == 0 && skew != 0u;      // wave-uniform: some leading columns are not written
    if (em.write_gate && hi + hc > lo0) {
        // partial flushes: every row writes columns [lo0, hi + hc), the last row only [lo0, hi)
        const u32 ncols = hi + hc - lo0;
        const u32 total_cells = em.nrows * ncols - hc;
        // a partial flush writes ncells columns per row, or ncells + 4 when skewed (skew carried + 4 - skew
        // appended): both are compile-time constants after inlining, so the row of piece i is two
        // divisions by constants and a select instead of a runtime division
        const bool wide_rows = hc != 0u;
        auto row_of = [&](u32 i, u32 per_cell) -> u32 {
            return wide_rows ? i / ((ncells + 4u) * per_cell) : i / ((ncells ? ncells : 1u) * per_cell);
        };
        // an unskewed partial flush without column breaks (every flush of the small-batch kernel's last
        // tile): rows x [0, ncells) with constant strides, no division per piece
        const bool plain_partial = !FULL && skew == 0u;
        const u32 cell_sh = em.cell_base + seg + shift;     // (shift = 0 whenever packed)
        if (plain_partial) {
            const u32 row_bytes = em.unit_cells * (EM::COMPACT ? 8u : 32u);
            if constexpr (EM::COMPACT) {
                const u64 *src = tile + lane + hs * S;
                u32 cl0 = em.cell_base + seg + lane + hs * em.unit_cells;
                for (u32 r = hs; r < em.nrows; r += hn, src += hn * S, cl0 += hn * em.unit_cells)
                    for (u32 q = 0; lane + q < ncells; q += 64)
                        store8(base, (packed ? packed_cell(em, cl0 + q) : cl0 + q + shift) * 8u, src[q]);
            } else if constexpr (EM::MONT) {
                const u64 *src = tile + lane + hs * S;
                u32 cl0 = em.cell_base + seg + lane + hs * em.unit_cells;
                for (u32 r = hs; r < em.nrows; r += hn, src += hn * S, cl0 += hn * em.unit_cells)
                    for (u32 q = 0; lane + q < ncells; q += 64) {
                        const u32 off = (packed ? packed_cell(em, cl0 + q) : cl0 + q + shift) * 32u - q * 32u;
                        const u64 v = src[q];
                        Fe8 m = mont_cell<EM>(v);
                        if (any_neg) {
                            if (is_neg(lane + q) && v != 0ull) m = fe_neg_nonzero(m);
                        }
                        store16(base, off + q * 32u, make_uint4(m.l[0], m.l[1], m.l[2], m.l[3]));
                        store16(base, off + q * 32u + 16u, make_uint4(m.l[4], m.l[5], m.l[6], m.l[7]));
                    }
            } else {
                // like the full-tile path below: all T / 32 LDS reads of a row first (columns past ncells hold
                // stale cells of the same row -- read, never stored), then the stores
                const u32 h = lane & 1u, p0 = lane >> 1;
                const u64 *src = tile + p0 + hs * S;
                const u32 cell0 = em.cell_base + seg + p0;
                u32 off = ((cell0 + shift) * 2u + h) * 16u + hs * row_bytes;
                for (u32 r = hs; r < em.nrows; r += hn, src += hn * S, off += hn * row_bytes) {
                    bool strad = false;
                    u32 offr = off;
                    if (packed) offr = ((cell0 + r * em.unit_cells + row_shift(r, strad)) * 2u + h) * 16u;
                    u64 vv[T / 32];
#pragma unroll
                    for (int k = 0; k < T / 32; k++) vv[k] = src[32 * k];
#pragma unroll
                    for (int k = 0; k < T / 32; k++) {
                        const u64 v = vv[k];
                        const u32 vlo = (u32)v, vhi = (u32)(v >> 32);
                        uint4 o = make_uint4(h ? 0u : vlo, h ? 0u : vhi, 0u, 0u);
                        if (any_neg) {
                            if (is_neg(p0 + 32u * (u32)k) && v != 0ull)
                                o = h ? make_uint4(HSW_P4, HSW_P5, HSW_P6, HSW_P7)
                                      : make_uint4(HSW_P0 - vlo, HSW_P1, HSW_P2, HSW_P3);
                        }
                        u32 o_off = offr + 1024u * (u32)k;
                        if (strad) o_off = (packed_cell(em, cell0 + r * em.unit_cells + 32u * (u32)k) * 2u + h) * 16u;
                        if (p0 + 32u * (u32)k < ncells) store16(base, o_off, o);
                    }
                }
            }
        } else if constexpr (EM::COMPACT) {
            // 8-byte cells: one lane per cell, 512 B contiguous per wave-instruction.
            // Negation cells keep x (their positions are static: hsw_neg_cells).  (skew is always 0 here)
#pragma unroll 4
            for (u32 i = lane + 64u * hs; i < total_cells; i += 64u * hn) {
                const u32 r = FULL ? i / (u32)T : row_of(i, 1u);
                const u32 p = lo0 + i - r * ncols;
                u32 cl = cell_sh + r * em.unit_cells + p;
                if (packed) cl = packed_cell(em, cl);
                store8(base, cl * 8u, tile[r * S + p]);
            }
        } else if constexpr (EM::MONT) {
            // full tiles index by the compile-time T (a shift) and skip the < 4 empty / held-back columns
            // of a skewed unit's first tile; partial ones divide by the run length
            const u32 total = FULL ? em.nrows * (u32)T : total_cells;
            for (u32 i = lane + 64u * hs; i < total; i += 64u * hn) {
                const u32 r = FULL ? i / (u32)T : row_of(i, 1u);
                const u32 p = FULL ? i % (u32)T : lo0 + i - r * ncols;
                if (first_skewed && p < (r == 0u ? lo0 : lo)) continue;
                const u64 v = tile[r * S + p];
                Fe8 m = mont_cell<EM>(v);
                if (any_neg) {                        // compile-time false for most tiles
                    if (is_neg(p) && v != 0ull) m = fe_neg_nonzero(m);
                }
                u32 cl = cell_sh + r * em.unit_cells + p;
                if (packed) cl = packed_cell(em, cl);
                store16(base, cl * 32u, make_uint4(m.l[0], m.l[1], m.l[2], m.l[3]));
                store16(base, cl * 32u + 16u, make_uint4(m.l[4], m.l[5], m.l[6], m.l[7]));
            }
        } else if (FULL) {
            // the common case: a full tile.  Lane l owns the 16-byte piece (l & 1) of LDS column
            // (l >> 1) + 32 k of every row; LDS and HBM addresses advance by constants (a row next to a
            // column break is re-based, one that straddles it placed piece by piece).
            const u32 h = lane & 1u, p0 = lane >> 1;
            const u32 row_bytes = em.unit_cells * 32u;
            const u64 *src = tile + p0 + hs * S;
            const u32 cell0 = em.cell_base + seg + p0;
            u32 off = ((cell0 + shift) * 2u + h) * 16u + hs * row_bytes;
            for (u32 r = hs; r < em.nrows; r += hn, src += hn * S, off += hn * row_bytes) {
                const bool skip0 = first_skewed && p0 < (r == 0u ? lo0 : lo);
                bool strad = false;
                u32 offr = off;
                if (packed) offr = ((cell0 + r * em.unit_cells + row_shift(r, strad)) * 2u + h) * 16u;
#pragma unroll
                for (int k = 0; k < T / 32; k++) {
                    const u64 v = src[32 * k];
                    const u32 vlo = (u32)v, vhi = (u32)(v >> 32);
                    uint4 o = make_uint4(h ? 0u : vlo, h ? 0u : vhi, 0u, 0u);
                    if (any_neg) {
                        if (is_neg(p0 + 32u * (u32)k) && v != 0ull)   // cell holds p - x (neg gate, compression.rs:320-321)
                            o = h ? make_uint4(HSW_P4, HSW_P5, HSW_P6, HSW_P7)
                                  : make_uint4(HSW_P0 - vlo, HSW_P1, HSW_P2, HSW_P3);
                    }
                    u32 o_off = offr + 1024u * (u32)k;
                    if (strad) o_off = (packed_cell(em, cell0 + r * em.unit_cells + 32u * (u32)k) * 2u + h) * 16u;
                    if (k != 0 || !skip0) store16(base, o_off, o);
                }
            }
        } else {
            // (skewed partial flushes: every piece placed on its own)
            const u32 ppr = FULL ? 2u * (u32)T : 2u * ncols;        // 16-byte pieces per row
            const u32 total = FULL ? em.nrows * ppr : 2u * total_cells;
            for (u32 i = lane; i < total; i += 64) {
                const u32 r = FULL ? i / (2u * (u32)T) : row_of(i, 2u);
                const u32 q = i - r * ppr;
                const u32 p = (FULL ? 0u : lo0) + (q >> 1), h = q & 1u;
                if (first_skewed && p < (r == 0u ? lo0 : lo)) continue;
                const u64 v = tile[r * S + p];
                const u32 vlo = (u32)v, vhi = (u32)(v >> 32);
                uint4 o = make_uint4(h ? 0u : vlo, h ? 0u : vhi, 0u, 0u);
                if (any_neg) {
                    if (is_neg(p) && v != 0ull)
                        o = h ? make_uint4(HSW_P4, HSW_P5, HSW_P6, HSW_P7)
                              : make_uint4(HSW_P0 - vlo, HSW_P1, HSW_P2, HSW_P3);
                }
                u32 cl = cell_sh + r * em.unit_cells + p;
                if (packed) cl = packed_cell(em, cl);
                store16(base, cl * 32u + h * 16u, o);
            }
        }
    }
    if (FULL && skew != 0u) {
        // carry the last `skew` cells (LDS columns [T, T + skew)) over to columns [0, skew)
        em_sync<EM>();                                 // every lane's stores have read the tile
        u32 m = 0;
        for (u32 j = 0; j < skew; j++) {
            em.row0[j] = em.row0[T + j];
            const u32 q = (u32)T + j - skew;             // tile position of the carried cell
            if (na >= 0 && (q == (u32)na || q == (u32)nb || q == (u32)nc || q == (u32)nd)) m |= 1u << j;
        }
        em.carry_neg = (u32)__builtin_amdgcn_readfirstlane((int)m);     // wave-uniform: keep it scalar
    }
    em_sync<EM>();
}

// The same write-out for tiles of finished 32-byte cells (Em::M32): a transposing copy.  Cell (r, p) sits at
// tile + r * STRIDE_W + 4 p words; lanes take consecutive 16-byte pieces, so a store instruction writes 1 KiB
// contiguous (T >= 32) or the 32 T bytes of 64 / (2 T) rows (T = 16).  Realignment, held-back heads, column
// breaks and helper waves exactly as in flush_tile; no negation bookkeeping (a cell is stored as it was emitted).
template <class EM, bool FULL>
DEV void flush_tile32(EM &em, u32 ncells, int fl) {
    constexpr int T = EM::TILE, SW = EM::STRIDE_W, HW = EM::HEAD_W;
    constexpr u32 PPR = 2u * (u32)T;                     // 16-byte pieces of a full row
    static_assert(T >= 4 && (T & (T - 1)) == 0, "M32 tiles: 4, 8, 16, 32, 64 ... cells per row");
    const u32 lane = lane_id();
    char *base = reinterpret_cast<char *>(em.out);
    const u32 skew = EM::REALIGN ? em.skew : 0u;
    const u64 *tile = em.tile;
    u32 hs = 0, hn = 1;
    if constexpr (EM::HELPERS) { hs = em.hsel; hn = em.hcnt; }
    const BlockBreaks bb{em.brk1, em.gap1, em.brk2, em.gap2};
    const u32 lo_c = flush_lo_cell(em.cell_base, (u32)fl, (u32)T, skew);
    bool packed = false;
    const u32 shift = flush_shift(bb, lo_c, em.nrows, em.unit_cells, (u32)T, packed);
    const bool hold_heads = FULL && fl == 0 && skew != 0u;
    const u32 hc = (!FULL && fl != 0 && skew != 0u) ? 4u - skew : 0u;
    const u32 hi = FULL ? (u32)T : ncells + skew;
    auto copy_cell = [](u64 *dst, const u64 *src) {
        const uint4 a = reinterpret_cast<const uint4 *>(src)[0], b = reinterpret_cast<const uint4 *>(src)[1];
        reinterpret_cast<uint4 *>(dst)[0] = a;
        reinterpret_cast<uint4 *>(dst)[1] = b;
    };
    if (hold_heads && lane >= 1u && lane < em.nrows)
        for (u32 j = skew; j < 4u; j++) copy_cell(em.head + lane * HW + (j - skew) * 4u, em.row0 + j * 4u);
    if (hc != 0u && lane >= 1u && lane < em.nrows)       // memory after row r's tail is row r+1's head
        for (u32 j = 0; j < hc; j++) copy_cell(em.tile + (lane - 1u) * SW + (hi + j) * 4u, em.head + lane * HW + j * 4u);
    em_sync<EM>();
    const u32 lo0 = fl == 0 ? skew : 0u;                 // first tile column written now: row 0 ...
    const u32 lo = hold_heads ? 4u : lo0;                // ... and the other rows
    const u32 seg = (u32)fl * (u32)T - skew;             // column c holds unit cell seg + c
    const bool first_skewed = FULL && fl == 0 && skew != 0u;
    if (em.write_gate && hi + hc > lo0) {
        const u32 row_bytes = em.unit_cells * 32u;
        if (FULL && !packed) {
            const u32 cell_sh = em.cell_base + seg + shift;
            // (eight LDS reads in flight before the first store: with one wave per SIMD nothing else hides the
            //  LDS latency of a read-then-store loop)
            if constexpr (PPR >= 64u) {
                const u32 h = lane & 1u, p0 = lane >> 1;
                const u64 *src = tile + p0 * 4u + h * 2u + hs * SW;
                u32 off = ((cell_sh + p0) * 2u + h) * 16u + hs * row_bytes;
                constexpr u32 K = PPR / 64u, UR = K >= 8u ? 1u : 8u / K;          // rows per batch
                for (u32 r = hs; r < em.nrows; r += hn * UR, src += hn * UR * SW, off += hn * UR * row_bytes) {
                    uint4 v[UR * K];
#pragma unroll
                    for (u32 u = 0; u < UR; u++)
#pragma unroll
                        for (u32 k = 0; k < K; k++)
                            if (r + u * hn < em.nrows) v[u * K + k] = *reinterpret_cast<const uint4 *>(src + u * hn * SW + 128u * k);
#pragma unroll
                    for (u32 u = 0; u < UR; u++) {
                        const u32 rr = r + u * hn;
                        const bool skip0 = first_skewed && p0 < (rr == 0u ? lo0 : lo);
#pragma unroll
                        for (u32 k = 0; k < K; k++)
                            if (rr < em.nrows && (k != 0u || !skip0)) store16(base, off + u * hn * row_bytes + 1024u * k, v[u * K + k]);
                    }
                }
            } else {
                constexpr u32 RPI = 64u / PPR;           // rows per store instruction
                constexpr u32 UR = 8u;                   // store instructions per batch
                const u32 dr = lane / PPR, q = lane % PPR, p = q >> 1, h = q & 1u;
                const u64 *src = tile + p * 4u + h * 2u + (hs * RPI + dr) * SW;
                u32 off = ((cell_sh + p) * 2u + h) * 16u + (hs * RPI + dr) * row_bytes;
                for (u32 r0 = hs * RPI; r0 < em.nrows; r0 += hn * RPI * UR, src += hn * RPI * UR * SW, off += hn * RPI * UR * row_bytes) {
                    uint4 v[UR];
#pragma unroll
                    for (u32 u = 0; u < UR; u++)
                        if (r0 + u * hn * RPI + dr < em.nrows) v[u] = *reinterpret_cast<const uint4 *>(src + u * hn * RPI * SW);
#pragma unroll
                    for (u32 u = 0; u < UR; u++) {
                        const u32 r = r0 + u * hn * RPI + dr;
                        if (r < em.nrows && !(first_skewed && p < (r == 0u ? lo0 : lo)))
                            store16(base, off + u * hn * RPI * row_bytes, v[u]);
                    }
                }
            }
        } else {
            // every piece placed on its own: the partial tile at the end of a phase, and flushes that straddle a
            // FlexGate column break.  Row r writes columns [lo0, hi + hc), the last row only [lo0, hi).
            const u32 ncols = hi + hc - lo0;
            const bool wide_rows = hc != 0u;             // ncols = ncells + 4 then, else ncells: divisions by constants
            const u32 total = FULL ? em.nrows * PPR : 2u * (em.nrows * ncols - hc);
            for (u32 i = lane + 64u * hs; i < total; i += 64u * hn) {
                u32 r, p;
                const u32 h = i & 1u;
                if (FULL) {
                    r = i / PPR;
                    p = (i % PPR) >> 1;
                    if (first_skewed && p < (r == 0u ? lo0 : lo)) continue;
                } else {
                    const u32 c = i >> 1;
                    r = wide_rows ? c / (ncells + 4u) : c / (ncells ? ncells : 1u);
                    p = lo0 + c - r * ncols;
                }
                u32 cl = em.cell_base + seg + r * em.unit_cells + p;
                cl = packed ? packed_cell(em, cl) : cl + shift;
                store16(base, cl * 32u + h * 16u, *reinterpret_cast<const uint4 *>(tile + r * SW + p * 4u + h * 2u));
            }
        }
    }
    if (FULL && skew != 0u) {
        em_sync<EM>();                                 // every lane's stores have read the tile
        for (u32 j = 0; j < skew; j++) copy_cell(em.row0 + j * 4u, em.row0 + ((u32)T + j) * 4u);
    }
    em_sync<EM>();
}

// ---------------------------------------------------------------- witness values
// What travels through the gate functions: the integer value of a cell and, for M32 emitters only, its
// Montgomery form.  For every other emitter W is a bare u64 and all of this folds away.
template <bool M> struct Wv;
template <> struct Wv<false> { u64 v; };
template <> struct Wv<true> { u64 v; Fe8 m; };
template <class EM> using W = Wv<EM::M32>;

// Out of line on purpose: a unit program holds hundreds of conversions, and inlined they make ~400 KB of
// straight-line code per kernel -- every wave then runs at the speed of instruction-cache misses (64 KB per two
// CUs; measured 2.32 ms per 4,096 blocks inlined against 1.91 ms out of line, profiles/README.md round 3).  The
// price of a call on gfx9-family parts: the callee starts with s_waitcnt vmcnt(0), i.e. the first conversion after
// a write-out waits until the wave's stores have landed.
__device__ __attribute__((noinline)) Fe8 mont32_call(u32 x) { return mont_from_u64<false>(x, 0u); }
__device__ __attribute__((noinline)) Fe8 mont64_call(u32 lo, u32 hi) { return mont_from_u64<true>(lo, hi); }
template <class EM> DEV W<EM> w32(u32 x) {               // a new value < 2^32
    W<EM> w;
    w.v = x;
    if constexpr (EM::M32) w.m = mont32_call(x);
    return w;
}
template <class EM> DEV W<EM> w64(u64 x) {               // a new value < 2^64
    W<EM> w;
    w.v = x;
    if constexpr (EM::M32) w.m = mont64_call((u32)x, (u32)(x >> 32));
    return w;
}
// New values below 2^8 / spreads of them / below 2^16 (M32): the limbs of every spread() call and the 16-bit
// witnesses are most of a unit's new values; their Montgomery forms come from a 24 KiB table (L1 / L2 resident)
// instead of a 32 x 256-bit multiply + Barrett step -- x * R mod p is additive in x, so 16 bits are two
// table entries and one field addition.
template <class EM> DEV W<EM> w8(const EM &em, u32 x) {          // x < 2^8
    W<EM> w;
    w.v = x;
    if constexpr (EM::M32) w.m = tab_entry(em, x);
    return w;
}
template <class EM> DEV W<EM> wspread8(const EM &em, u32 x) {    // spread(x), x < 2^8
    W<EM> w;
    w.v = spread16(x);
    if constexpr (EM::M32) w.m = tab_entry(em, 256u + x);
    return w;
}
template <class EM> DEV W<EM> w16(const EM &em, u32 x) {         // x < 2^16
    W<EM> w;
    w.v = x;
    if constexpr (EM::M32) w.m = fe_add(tab_entry(em, x & 255u), tab_entry(em, 512u + (x >> 8)));
    return w;
}
template <class EM, u64 K> DEV W<EM> wk() {              // a gate constant
    W<EM> w;
    w.v = K;
    if constexpr (EM::M32) w.m = mont_k<K>();
    return w;
}
template <class EM> DEV W<EM> wadd(const W<EM> &a, const W<EM> &b) {   // a + b (the integer sum stays below 2^64 on this path)
    W<EM> w;
    w.v = a.v + b.v;
    if constexpr (EM::M32) w.m = fe_add(a.m, b.m);
    return w;
}
// K[r] as a gate constant (compression.rs:151): lane-dependent, so its Montgomery form comes from a table
template <class EM> DEV W<EM> wround_constant(u32 r) {
    W<EM> w;
    w.v = K256[r & 63u];
    if constexpr (EM::M32) w.m = K256M[r & 63u];
    return w;
}

template <class EM, class C>
DEV auto emit(C, EM &em, const W<EM> &w) {
    if constexpr (EM::EMITS) {
        if constexpr (EM::M32) {
            uint4 *q = reinterpret_cast<uint4 *>(em.row + 4 * C::pos);
            q[0] = make_uint4(w.m.l[0], w.m.l[1], w.m.l[2], w.m.l[3]);
            q[1] = make_uint4(w.m.l[4], w.m.l[5], w.m.l[6], w.m.l[7]);
        } else {
            em.row[C::pos] = w.v;
        }
    }
    if constexpr (C::pos + 1 == EM::TILE) {
        if constexpr (EM::M32) flush_tile32<EM, true>(em, EM::TILE, C::fl);
        else flush_tile<EM, true>(em, EM::TILE, C::fl, C::na, C::nb, C::nc, C::nd, C::cn);
        constexpr int T3 = EM::TILE - 3;
        constexpr bool carried = C::na >= T3 || C::nb >= T3 || C::nc >= T3 || C::nd >= T3;
        return Cur<0, C::fl + 1, -1, -1, -1, -1, carried>{};
    } else {
        return Cur<C::pos + 1, C::fl, C::na, C::nb, C::nc, C::nd, C::cn>{};
    }
}
// cell whose field value is -x (x small; compression.rs:320-321).  u64 tiles store x and append the position to
// the cursor's neg list (the write-out expands it); M32 tiles store p - x in Montgomery form right away.
template <class EM, class C>
DEV auto emit_neg(C c, EM &em, const W<EM> &x) {
    if constexpr (EM::M32) {
        W<EM> n;
        n.v = x.v;
        const Fe8 neg = fe_neg_nonzero(x.m);
#pragma unroll
        for (int j = 0; j < 8; j++) n.m.l[j] = x.v != 0ull ? neg.l[j] : 0u;      // -0 = 0
        return emit(c, em, n);
    } else {
        static_assert(C::nd < 0, "more than four neg cells in one tile");
        if constexpr (C::na < 0) return emit(Cur<C::pos, C::fl, C::pos, -1, -1, -1, C::cn>{}, em, x);
        else if constexpr (C::nb < 0) return emit(Cur<C::pos, C::fl, C::na, C::pos, -1, -1, C::cn>{}, em, x);
        else if constexpr (C::nc < 0) return emit(Cur<C::pos, C::fl, C::na, C::nb, C::pos, -1, C::cn>{}, em, x);
        else return emit(Cur<C::pos, C::fl, C::na, C::nb, C::nc, C::pos, C::cn>{}, em, x);
    }
}

// Units of a phase dealt to `parts` waves: part k expands the contiguous range
// [k*per, min((k+1)*per, n_units)) with per = ceil(n_units / parts).
DEV void part_units(u32 part, u32 parts, u32 n_units, u32 &unit_lo, u32 &nrows) {
    const u32 per = (n_units + parts - 1) / parts;
    unit_lo = part * per;
    nrows = unit_lo >= n_units ? 0u : (n_units - unit_lo < per ? n_units - unit_lo : per);
}
// Unit expanded by this lane (idle lanes shadow the last active one).
DEV u32 lane_unit(u32 part, u32 parts, u32 n_units) {
    const u32 lane = lane_id();
    u32 nrows, unit_lo;
    part_units(part, parts, n_units, unit_lo, nrows);
    return (nrows ? unit_lo : 0u) + (lane < nrows ? lane : (nrows ? nrows - 1 : 0));
}

// A phase is `n_units` independent units of `unit_cells` gate cells each.
// Returns false if this wave has nothing to do in the phase (wave-uniform).
template <class EM>
DEV bool phase_begin(EM &em, u32 part, u32 parts, u32 n_units, u32 unit_cells, u32 phase_off,
                     u32 call_base, u32 calls_per_unit, u32 lk_base = 0, u32 lk_per_unit = 0) {
    const u32 lane = lane_id();
    u32 nrows, unit_lo;
    part_units(part, parts, n_units, unit_lo, nrows);
    if (nrows == 0) unit_lo = 0;
    em.nrows = nrows;
    em.unit_cells = unit_cells;
    em.cell_base = phase_off + unit_lo * unit_cells;
    em.active = lane < nrows;
    const u32 r = lane < nrows ? lane : (nrows ? nrows - 1 : 0);
    em.unit = unit_lo + r;
    em.call = r * calls_per_unit;
    em.call_first = call_base + unit_lo * calls_per_unit;
    em.calls = nrows * calls_per_unit;
    em.lk = r * lk_per_unit;
    em.lk_first = lk_base + unit_lo * lk_per_unit;
    em.lks = nrows * lk_per_unit;
    // realignment (flush_tile): only where every unit of the phase starts at the same offset within
    // a 128-byte line (unit_cells % 4 == 0: words, schedule steps, rounds -- 98 % of the cells)
    u32 skew = 0;
    if constexpr (EM::REALIGN) {
        if ((unit_cells & 3u) == 0u) {
            u32 cl = em.cell_base;
            if (em.brk1 != 0xffffffffu) cl = packed_cell(em, cl);
            skew = (cl + (u32)(reinterpret_cast<size_t>(em.out) >> 5)) & 3u;
        }
    }
    em.skew = (u32)__builtin_amdgcn_readfirstlane((int)skew);           // wave-uniform: keep it scalar
    em.carry_neg = 0;
    em.row = em.row0 + skew * (u32)EM::CW;
    return nrows != 0;
}

// Chip columns of the spread calls staged by this phase-part (spread.rs:196-233):
// limb call n (absolute, counted from SpreadConfig.num_limb_sum = 0) lands in
// column n % ncols at row n / ncols; buffer row 0 = row cursor0 / ncols.  Every
// column receives one contiguous run of rows.
template <int L, class EM>
DEV void flush_chip(const EM &em, const ExpandParams &p, u64 block_first_limb) {
    constexpr int B = 16 / L;
    constexpr u32 MASK = (1u << B) - 1u;
    if (em.calls == 0 || (p.flags & HSW_K_SKIP_CHIP)) return;
    em_sync<EM>();                                                   // d16 staged by all lanes
    const u32 lane = lane_id();
    const u64 ncols = p.ncols;
    const u64 first = block_first_limb + (u64)em.call_first * L;       // first limb call of this run
    const u64 last = first + (u64)em.calls * L - 1;
    const u64 row0 = p.cursor0 / ncols;
    for (u64 c = 0; c < ncols; c++) {
        if (last < c) continue;
        const u64 row_lo = (first + ncols - 1 - c) / ncols;            // first row with row*ncols + c >= first
        const u64 row_hi = (last - c) / ncols;                         // last row with row*ncols + c <= last
        if (row_hi < row_lo) continue;
        const u32 count = (u32)(row_hi - row_lo + 1);
        const u32 n0 = (u32)(row_lo * ncols + c - first);              // run-relative limb index of row_lo
        // wave-uniform column run base + 32-bit lane offsets
        const size_t cell0 = (size_t)c * p.chip_col_stride + (size_t)(row_lo - row0);
        constexpr u32 CB = EM::COMPACT ? 8u : 32u;
        char *cdb = reinterpret_cast<char *>(p.chip_dense) + cell0 * CB;
        char *csb = reinterpret_cast<char *>(p.chip_spread) + cell0 * CB;
        if constexpr (EM::COMPACT) {
            for (u32 k = lane; k < count; k += 64) {
                const u32 n = n0 + k * (u32)ncols;
                const u32 call = n / L, j = n % L;
                const u32 limb = ((u32)em.d16[call] >> (B * j)) & MASK;
                store8(cdb, k * 8u, limb);
                store8(csb, k * 8u, spread16(limb));
            }
        } else if constexpr (EM::MONT_OUT) {
            for (u32 k = lane; k < count; k += 64) {
                const u32 n = n0 + k * (u32)ncols;
                const u32 call = n / L, j = n % L;
                const u32 limb = ((u32)em.d16[call] >> (B * j)) & MASK;
                Fe8 md, ms;
                if constexpr (EM::M32 && B <= 8) {           // both cells straight from the byte tables
                    md = tab_entry(em, limb);
                    ms = B == 8 ? tab_entry(em, 256u + limb) : tab_entry(em, spread16(limb));
                } else {
                    md = mont_from_u64<false>(limb, 0);
                    ms = mont_from_u64<false>(spread16(limb), 0);
                }
                store16(cdb, k * 32u, make_uint4(md.l[0], md.l[1], md.l[2], md.l[3]));
                store16(cdb, k * 32u + 16u, make_uint4(md.l[4], md.l[5], md.l[6], md.l[7]));
                store16(csb, k * 32u, make_uint4(ms.l[0], ms.l[1], ms.l[2], ms.l[3]));
                store16(csb, k * 32u + 16u, make_uint4(ms.l[4], ms.l[5], ms.l[6], ms.l[7]));
            }
        } else {
            for (u32 i = lane; i < 2u * count; i += 64) {
                const u32 k = i >> 1, hpart = i & 1u;
                const u32 n = n0 + k * (u32)ncols;
                const u32 call = n / L, j = n % L;
                const u32 limb = ((u32)em.d16[call] >> (B * j)) & MASK;
                store16(cdb, i * 16u, make_uint4(hpart ? 0u : limb, 0u, 0u, 0u));
                store16(csb, i * 16u, make_uint4(hpart ? 0u : spread16(limb), 0u, 0u, 0u));
            }
        }
    }
    em_sync<EM>();
}

// Lookup-advice column (RC only): the values queued by enable_lookup, in queue
// order, as one contiguous run per phase-part (RangeConfig::finalize, lib.rs:469).
template <class EM>
DEV void flush_lookup(const EM &em, const ExpandParams &p, size_t lookup_block_base) {
    if (em.lks == 0 || p.lookup == nullptr) return;
    em_sync<EM>();
    const u32 lane = lane_id();
    uint4 *out = reinterpret_cast<uint4 *>(p.lookup) + (lookup_block_base + em.lk_first) * 2u;
    if constexpr (EM::COMPACT) {
        u64 *out64 = reinterpret_cast<u64 *>(p.lookup) + lookup_block_base + em.lk_first;
        for (u32 k = lane; k < em.lks; k += 64) out64[k] = em.lk16[k];
    } else if constexpr (EM::MONT_OUT) {
        for (u32 k = lane; k < em.lks; k += 64) {
            const u32 x = em.lk16[k];
            Fe8 m;
            if constexpr (EM::M32) m = fe_add(tab_entry(em, x & 255u), tab_entry(em, 512u + (x >> 8)));
            else m = mont_from_u64<false>(x, 0);
            out[2 * k] = make_uint4(m.l[0], m.l[1], m.l[2], m.l[3]);
            out[2 * k + 1] = make_uint4(m.l[4], m.l[5], m.l[6], m.l[7]);
        }
    } else {
        for (u32 i = lane; i < 2u * em.lks; i += 64) {
            uint4 o;
            o.x = (i & 1u) ? 0u : (u32)em.lk16[i >> 1];
            o.y = 0; o.z = 0; o.w = 0;
            out[i] = o;
        }
    }
    em_sync<EM>();
}

template <int L, class EM, class C>
DEV void phase_end(C, EM &em, const ExpandParams &p, u64 block_first_limb, size_t lookup_block_base) {
    if constexpr (EM::M32) {
        if constexpr (C::pos != 0) flush_tile32<EM, false>(em, C::pos, C::fl);
        else if (EM::REALIGN && em.skew != 0u) flush_tile32<EM, false>(em, 0u, C::fl);                    // the carried cells only
    } else {
        if constexpr (C::pos != 0) flush_tile<EM, false>(em, C::pos, C::fl, C::na, C::nb, C::nc, C::nd, C::cn);
        else if (EM::REALIGN && em.skew != 0u) flush_tile<EM, false>(em, 0u, C::fl, -1, -1, -1, -1, C::cn);   // the carried cells only
    }
    flush_chip<L>(em, p, block_first_limb);
    if constexpr (EM::RC) flush_lookup(em, p, lookup_block_base);
}

// enable_lookup: queue a (<= 16-bit) value for the lookup-advice column
template <class EM>
DEV void lookup16(EM &em, u32 v) {
    if constexpr (EM::RC) {
        if constexpr (EM::EMITS)
            if (em.active) em.lk16[em.lk] = (u16)v;
        em.lk++;
    }
}
// range_check(a, 32) at lookup_bits = 16: halo2-base lays out [limb0, limb1, 2^16, a]
// (inner_product_left of the two limbs with [1, 2^16]) and looks both limbs up (A3).
template <class EM, class C>
DEV auto range_check32(C c, EM &em, const W<EM> &a) {
    const u32 l0 = (u32)a.v & 0xffffu, l1 = (u32)a.v >> 16;
    lookup16(em, l0);
    lookup16(em, l1);
    if constexpr (EM::RC) {
        auto c1 = emit(c, em, w16(em, l0));
        auto c2 = emit(c1, em, w16(em, l1));
        auto c3 = emit(c2, em, wk<EM, (1ull << 16)>());
        return emit(c3, em, a);
    } else {
        return c;
    }
}

// ---------------------------------------------------- halo2-base gate cells
// (cell orders: DESIGN.md assumption A1).  Every function takes the cursor and
// returns the advanced cursor; runtime results come back through references.
// Values are W<EM> (above): built once per distinct value (w32 / w64 / wk / wadd), then only copied.
template <class EM, class C>
DEV auto g_lw(C c, EM &em, const W<EM> &v) { return emit(c, em, v); }            // [v]
template <class EM, class C>
DEV auto g_add(C c, EM &em, const W<EM> &a, const W<EM> &b, W<EM> &out) {        // [a, b, 1, a+b]
    const W<EM> sum = wadd<EM>(a, b);            // (out may alias a or b)
    auto c1 = emit(c, em, a);
    auto c2 = emit(c1, em, b);
    auto c3 = emit(c2, em, wk<EM, 1>());
    out = sum;
    return emit(c3, em, sum);
}
// mul_add(a, b, c) = a*b + c -> [c, a, b, out]; `out` is passed in (every b, or a, is a gate constant:
// a power of two or a 3-term sum of them, so the caller has the product by shifts).
template <class EM, class C>
DEV auto g_mul_add(C c, EM &em, const W<EM> &a, const W<EM> &b, const W<EM> &cc, const W<EM> &out) {
    auto c1 = emit(c, em, cc);
    auto c2 = emit(c1, em, a);
    auto c3 = emit(c2, em, b);
    return emit(c3, em, out);
}

// ------------------------------------------------------- spread.rs mirrors
// SpreadConfig::spread (spread.rs:76-123) on a 16-bit dense value.  The L limbs and their spreads are the
// call's new values; the running sums are the first limb, partial sums, and finally `dense` / the result.
template <int L, class EM>
struct SpreadVals {
    W<EM> limb[L], sl[L];
};
template <int L, int J, class EM, class C>
DEV auto spread_limbs_lw(C c, EM &em, const SpreadVals<L, EM> &sv) {             // :86-88
    if constexpr (J == L) return c;
    else return spread_limbs_lw<L, J + 1>(g_lw(c, em, sv.limb[J]), em, sv);
}
template <int L, int J, class EM, class C>
DEV auto spread_limbs_sum(C c, EM &em, const SpreadVals<L, EM> &sv, const W<EM> &dense, const W<EM> &sum) {   // :91-98
    if constexpr (J == L) return c;
    else {
        constexpr int B = 16 / L;
        // sum + limb_J * 2^(B J): limb 0 itself, the dense value at the end, a new partial sum in between
        W<EM> ns;
        if constexpr (J == 0) ns = sv.limb[0];
        else if constexpr (J == L - 1) ns = dense;
        else ns = w32<EM>((u32)dense.v & ((1u << (B * (J + 1))) - 1u));
        auto c1 = g_mul_add(c, em, sv.limb[J], wk<EM, (1ull << (B * J))>(), sum, ns);
        return spread_limbs_sum<L, J + 1>(c1, em, sv, dense, ns);
    }
}
template <int L, int J, class EM, class C>
DEV auto spread_limbs_acc(C c, EM &em, const SpreadVals<L, EM> &sv, const W<EM> &acc, const W<EM> &spread) {   // :112-121
    if constexpr (J == L) return c;
    else {
        constexpr int B = 16 / L;
        auto c1 = g_lw(c, em, sv.sl[J]);                                         // spread_limb :225
        W<EM> na;
        if constexpr (J == 0) na = sv.sl[0];
        else if constexpr (J == L - 1) na = spread;
        else na = w32<EM>((u32)spread.v & (u32)((1ull << (2 * B * (J + 1))) - 1ull));
        auto c2 = g_mul_add(c1, em, sv.sl[J], wk<EM, (1ull << (2 * B * J))>(), acc, na);
        return spread_limbs_acc<L, J + 1>(c2, em, sv, na, spread);
    }
}
template <int L, class EM, class C>
DEV auto sc_spread(C c, EM &em, const W<EM> &dense, W<EM> &spread_out) {
    constexpr int B = 16 / L;
    const u32 d = (u32)dense.v;
    if constexpr (EM::EMITS)
        if (em.active) em.d16[em.call] = (u16)d;       // chip cells are produced by flush_chip
    em.call++;
    SpreadVals<L, EM> sv;
#pragma unroll
    for (int j = 0; j < L; j++) {
        const u32 limb = (d >> (B * j)) & ((1u << B) - 1u);
        if constexpr (B <= 8) sv.limb[j] = w8(em, limb);
        else sv.limb[j] = w16(em, limb);
        if constexpr (B == 8) sv.sl[j] = wspread8(em, limb);
        else if constexpr (B < 8) sv.sl[j] = w8(em, spread16(limb));
        else sv.sl[j] = w32<EM>(spread16(limb));
    }
    spread_out = L == 1 ? sv.sl[0] : w32<EM>(spread16(d));
    auto c1 = spread_limbs_lw<L, 0>(c, em, sv);
    auto c2 = spread_limbs_sum<L, 0>(c1, em, sv, dense, wk<EM, 0>());
    return spread_limbs_acc<L, 0>(c2, em, sv, wk<EM, 0>(), spread_out);
}

// state_to_spread_u32 (compression.rs:215-246); x = the (already built) value of the word
template <int L, class EM, class C>
DEV auto state_to_spread(C c, EM &em, const W<EM> &x) {
    const W<EM> lo = w16(em, (u32)x.v & 0xffffu), hi = w16(em, (u32)x.v >> 16);
    W<EM> unused;
    auto c1 = g_lw(c, em, lo);                               // :230
    auto c2 = g_lw(c1, em, hi);                              // :231
    auto c3 = g_mul_add(c2, em, hi, wk<EM, (1ull << 16)>(), lo, x);   // :232-237
    auto c4 = sc_spread<L>(c3, em, lo, unused);              // :243
    return sc_spread<L>(c4, em, hi, unused);                 // :244
}

// mod_u32 (compression.rs:266-295); x < 2^35
template <class EM, class C>
DEV auto mod_u32(C c, EM &em, const W<EM> &x, W<EM> &lo_out) {
    const W<EM> lo = w32<EM>((u32)x.v), hi = w8(em, (u32)(x.v >> 32));
    auto c1 = g_lw(c, em, lo);                               // :280
    auto c2 = g_lw(c1, em, hi);                              // :281
    auto c3 = range_check32(c2, em, lo);                     // :282
    auto c4 = g_mul_add(c3, em, hi, wk<EM, (1ull << 32)>(), lo, x);   // :283-288
    lo_out = lo;                                             // (may alias x)
    return c4;
}

// { spread(even); spread(odd); 2*odd_spread + even_spread } (compression.rs:344-354 etc.)
template <int L, class EM, class C>
DEV auto recheck_even_odd(C c, EM &em, const W<EM> &even, const W<EM> &odd) {
    W<EM> es, os;
    auto c1 = sc_spread<L>(c, em, even, es);
    auto c2 = sc_spread<L>(c1, em, odd, os);
    return g_mul_add(c2, em, wk<EM, 2>(), os, es, wadd<EM>(es, wadd<EM>(os, os)));
}

// sigma_generic (compression.rs:702-882).  S1..S3 are STARTS[1..3]; C0..C3 the
// coeffs (sigma_lower drops the wrapped piece: two terms in C0, :658,:685).
struct SigmaUpper0 {   // :600-608
    static constexpr int S1 = 2, S2 = 13, S3 = 22;
    static constexpr u64 C0 = (1ull << 60) + (1ull << 38) + (1ull << 20);
    static constexpr u64 C1 = (1ull << 0) + (1ull << 42) + (1ull << 24);
    static constexpr u64 C2 = (1ull << 22) + (1ull << 0) + (1ull << 46);
    static constexpr u64 C3 = (1ull << 40) + (1ull << 18) + (1ull << 0);
};
struct SigmaUpper1 {   // :627-635
    static constexpr int S1 = 6, S2 = 11, S3 = 25;
    static constexpr u64 C0 = (1ull << 52) + (1ull << 42) + (1ull << 14);
    static constexpr u64 C1 = (1ull << 0) + (1ull << 54) + (1ull << 26);
    static constexpr u64 C2 = (1ull << 10) + (1ull << 0) + (1ull << 36);
    static constexpr u64 C3 = (1ull << 38) + (1ull << 28) + (1ull << 0);
};
struct SigmaLower0 {   // :654-662
    static constexpr int S1 = 3, S2 = 7, S3 = 18;
    static constexpr u64 C0 = (1ull << 50) + (1ull << 28);
    static constexpr u64 C1 = (1ull << 0) + (1ull << 56) + (1ull << 34);
    static constexpr u64 C2 = (1ull << 8) + (1ull << 0) + (1ull << 42);
    static constexpr u64 C3 = (1ull << 30) + (1ull << 22) + (1ull << 0);
};
struct SigmaLower1 {   // :681-689
    static constexpr int S1 = 10, S2 = 17, S3 = 19;
    static constexpr u64 C0 = (1ull << 30) + (1ull << 26);
    static constexpr u64 C1 = (1ull << 0) + (1ull << 50) + (1ull << 46);
    static constexpr u64 C2 = (1ull << 14) + (1ull << 0) + (1ull << 60);
    static constexpr u64 C3 = (1ull << 18) + (1ull << 4) + (1ull << 0);
};

// a new value below 2^BITS: the narrow conversion where it is enough
template <class EM, int BITS>
DEV W<EM> wbits(u64 x) {
    if constexpr (BITS <= 32) return w32<EM>((u32)x);
    else return w64<EM>(x);
}

template <class SG, int L, class EM, class C>
DEV auto sigma_generic(C c0, EM &em, u32 x, W<EM> &out) {
    const u64 X = spread32(x);                               // x_spread.1 * 2^32 + x_spread.0
    // :719-734 the four pieces, spread bits [2*start, 2*end) shifted to 0
    const W<EM> pa = wbits<EM, 2 * SG::S1>(X & ((1ull << (2 * SG::S1)) - 1));
    const W<EM> pb = wbits<EM, 2 * (SG::S2 - SG::S1)>((X >> (2 * SG::S1)) & ((1ull << (2 * (SG::S2 - SG::S1))) - 1));
    const W<EM> pc = wbits<EM, 2 * (SG::S3 - SG::S2)>((X >> (2 * SG::S2)) & ((1ull << (2 * (SG::S3 - SG::S2))) - 1));
    const W<EM> pd = wbits<EM, 2 * (32 - SG::S3)>(X >> (2 * SG::S3));
    auto c1 = g_lw(c0, em, pa);
    auto c2 = g_lw(c1, em, pb);
    auto c3 = g_lw(c2, em, pc);
    auto c4 = g_lw(c3, em, pd);
    // :736-754 recomposition: partial sums of the spread, the last one is X itself
    const W<EM> s1 = wbits<EM, 2 * SG::S2>(pa.v + (pb.v << (2 * SG::S1)));
    const W<EM> s2 = wbits<EM, 2 * SG::S3>(s1.v + (pc.v << (2 * SG::S2)));
    const W<EM> s3 = w64<EM>(X);
    auto c5 = g_mul_add(c4, em, pb, wk<EM, (1ull << (2 * SG::S1))>(), pa, s1);
    auto c6 = g_mul_add(c5, em, pc, wk<EM, (1ull << (2 * SG::S2))>(), s1, s2);
    auto c7 = g_mul_add(c6, em, pd, wk<EM, (1ull << (2 * SG::S3))>(), s2, s3);
    // :755-760 x_composed
    auto c8 = g_mul_add(c7, em, w32<EM>((u32)(X >> 32)), wk<EM, (1ull << 32)>(), w32<EM>((u32)X), s3);
    // :780-808 r_spread = sum coeff_i * piece_i  (< 2^64 by construction)
    const W<EM> r1 = w64<EM>(SG::C0 * pa.v);
    const W<EM> r2 = w64<EM>(r1.v + SG::C1 * pb.v);
    const W<EM> r3 = w64<EM>(r2.v + SG::C2 * pc.v);
    const W<EM> r = w64<EM>(r3.v + SG::C3 * pd.v);
    auto c9 = g_mul_add(c8, em, wk<EM, SG::C0>(), pa, wk<EM, 0>(), r1);
    auto c10 = g_mul_add(c9, em, wk<EM, SG::C1>(), pb, r1, r2);
    auto c11 = g_mul_add(c10, em, wk<EM, SG::C2>(), pc, r2, r3);
    auto c12 = g_mul_add(c11, em, wk<EM, SG::C3>(), pd, r3, r);
    // :811-836
    const W<EM> r_lo = w32<EM>((u32)r.v), r_hi = w32<EM>((u32)(r.v >> 32));
    auto c13 = g_lw(c12, em, r_lo);                          // :820
    auto c14a = g_lw(c13, em, r_hi);                         // :821
    auto c14b = range_check32(c14a, em, r_lo);               // :822
    auto c14 = range_check32(c14b, em, r_hi);                // :823
    auto c15 = g_mul_add(c14, em, r_hi, wk<EM, (1ull << 32)>(), r_lo, r);
    // :843-846
    const u32 rl = (u32)r.v, rh = (u32)(r.v >> 32);
    const W<EM> lo_even = w16(em, even_bits(rl)), lo_odd = w16(em, even_bits(rl >> 1));
    const W<EM> hi_even = w16(em, even_bits(rh)), hi_odd = w16(em, even_bits(rh >> 1));
    auto c16 = g_lw(c15, em, lo_even);
    auto c17 = g_lw(c16, em, lo_odd);
    auto c18 = g_lw(c17, em, hi_even);
    auto c19 = g_lw(c18, em, hi_odd);
    lookup16(em, (u32)lo_even.v); lookup16(em, (u32)lo_odd.v);        // range_check 16, spread.rs:160-161
    lookup16(em, (u32)hi_even.v); lookup16(em, (u32)hi_odd.v);
    auto c20 = recheck_even_odd<L>(c19, em, lo_even, lo_odd);     // :852-862
    auto c21 = recheck_even_odd<L>(c20, em, hi_even, hi_odd);     // :863-873
    out = w32<EM>(((u32)hi_even.v << 16) | (u32)lo_even.v);
    return g_mul_add(c21, em, hi_even, wk<EM, (1ull << 16)>(), lo_even, out);   // :874-879
}

// ch (compression.rs:297-405); x, y, z are the dense words e, f, g.  In two halves so that the small-batch
// kernel (hsw_small.hpp) can give them to different waves: A = the sums, the negations, the eight even / odd
// witnesses and the re-checks of p; B = the re-checks of q and the result.
struct ChVals {
    u32 x_lo, x_hi, y_lo, y_hi, z_lo, z_hi;
    u32 p_lo_even, p_lo_odd, p_hi_even, p_hi_odd, q_lo_even, q_lo_odd, q_hi_even, q_hi_odd;
};
DEV ChVals ch_values(u32 x, u32 y, u32 z) {
    ChVals v;
    v.x_lo = spread16(x); v.x_hi = spread16(x >> 16);
    v.y_lo = spread16(y); v.y_hi = spread16(y >> 16);
    v.z_lo = spread16(z); v.z_hi = spread16(z >> 16);
    const u32 MASK_EVEN_32 = 0x55555555u;
    const u32 p_lo = v.x_lo + v.y_lo, p_hi = v.x_hi + v.y_hi;                         // :309-318
    const u32 q_lo = MASK_EVEN_32 - v.x_lo + v.z_lo, q_hi = MASK_EVEN_32 - v.x_hi + v.z_hi;   // :319-335 (values < 2^32)
    // :336-343 four even/odd splits before any re-check
    v.p_lo_even = even_bits(p_lo); v.p_lo_odd = even_bits(p_lo >> 1);
    v.p_hi_even = even_bits(p_hi); v.p_hi_odd = even_bits(p_hi >> 1);
    v.q_lo_even = even_bits(q_lo); v.q_lo_odd = even_bits(q_lo >> 1);
    v.q_hi_even = even_bits(q_hi); v.q_hi_odd = even_bits(q_hi >> 1);
    return v;
}
// the eight even / odd witnesses as values: built once, used by both halves
template <class EM>
struct ChW {
    W<EM> p_lo_even, p_lo_odd, p_hi_even, p_hi_odd, q_lo_even, q_lo_odd, q_hi_even, q_hi_odd;
};
template <class EM>
DEV ChW<EM> ch_witnesses(const EM &em, const ChVals &v) {
    ChW<EM> w;
    w.p_lo_even = w16(em, v.p_lo_even); w.p_lo_odd = w16(em, v.p_lo_odd);
    w.p_hi_even = w16(em, v.p_hi_even); w.p_hi_odd = w16(em, v.p_hi_odd);
    w.q_lo_even = w16(em, v.q_lo_even); w.q_lo_odd = w16(em, v.q_lo_odd);
    w.q_hi_even = w16(em, v.q_hi_even); w.q_hi_odd = w16(em, v.q_hi_odd);
    return w;
}
template <int L, class EM, class C>
DEV auto ch_part_a(C c0, EM &em, const ChVals &v, const ChW<EM> &w) {
    const W<EM> x_lo = w32<EM>(v.x_lo), x_hi = w32<EM>(v.x_hi);
    const W<EM> mask = wk<EM, 0x55555555ull>(), one = wk<EM, 1>(), zero = wk<EM, 0>();
    W<EM> p_lo, p_hi, q_lo, q_hi;
    auto c1 = g_add(c0, em, x_lo, w32<EM>(v.y_lo), p_lo);   // :309-313
    auto c2 = g_add(c1, em, x_hi, w32<EM>(v.y_hi), p_hi);   // :314-318
    // neg: [a, -a, 1, 0]                                       :320-321
    auto c3 = emit(c2, em, x_lo);
    auto c4 = emit_neg(c3, em, x_lo);
    auto c5 = emit(c4, em, one);
    auto c6 = emit(c5, em, zero);
    auto c7 = emit(c6, em, x_hi);
    auto c8 = emit_neg(c7, em, x_hi);
    auto c9 = emit(c8, em, one);
    auto c10 = emit(c9, em, zero);
    // three_add(Constant(MASK), -x, z)                         :322-335, :521-530
    const W<EM> t_lo = w32<EM>(0x55555555u - v.x_lo), t_hi = w32<EM>(0x55555555u - v.x_hi);
    auto c11 = emit(c10, em, mask);
    auto c12 = emit_neg(c11, em, x_lo);
    auto c13 = emit(c12, em, one);
    auto c14 = emit(c13, em, t_lo);
    auto c15 = g_add(c14, em, t_lo, w32<EM>(v.z_lo), q_lo);
    auto c16 = emit(c15, em, mask);
    auto c17 = emit_neg(c16, em, x_hi);
    auto c18 = emit(c17, em, one);
    auto c19 = emit(c18, em, t_hi);
    auto c20 = g_add(c19, em, t_hi, w32<EM>(v.z_hi), q_hi);
    // :336-343 four even/odd splits before any re-check
    auto c21 = g_lw(c20, em, w.p_lo_even);
    auto c22 = g_lw(c21, em, w.p_lo_odd);
    auto c23 = g_lw(c22, em, w.p_hi_even);
    auto c24 = g_lw(c23, em, w.p_hi_odd);
    auto c25 = g_lw(c24, em, w.q_lo_even);
    auto c26 = g_lw(c25, em, w.q_lo_odd);
    auto c27 = g_lw(c26, em, w.q_hi_even);
    auto c28 = g_lw(c27, em, w.q_hi_odd);
    lookup16(em, v.p_lo_even); lookup16(em, v.p_lo_odd); lookup16(em, v.p_hi_even); lookup16(em, v.p_hi_odd);
    lookup16(em, v.q_lo_even); lookup16(em, v.q_lo_odd); lookup16(em, v.q_hi_even); lookup16(em, v.q_hi_odd);
    auto c29 = recheck_even_odd<L>(c28, em, w.p_lo_even, w.p_lo_odd);     // :344-354
    return recheck_even_odd<L>(c29, em, w.p_hi_even, w.p_hi_odd);         // :355-365
}
template <int L, class EM, class C>
DEV auto ch_part_b(C c30, EM &em, const ChVals &v, const ChW<EM> &w, W<EM> &out) {
    auto c31 = recheck_even_odd<L>(c30, em, w.q_lo_even, w.q_lo_odd);     // :366-376
    auto c32 = recheck_even_odd<L>(c31, em, w.q_hi_even, w.q_hi_odd);     // :377-387
    W<EM> out_lo, out_hi;
    auto c33 = g_add(c32, em, w.p_lo_odd, w.q_lo_odd, out_lo);   // :388-392
    auto c34 = g_add(c33, em, w.p_hi_odd, w.q_hi_odd, out_hi);   // :393-397
    out = w32<EM>(((u32)out_hi.v << 16) + (u32)out_lo.v);
    (void)v;
    return g_mul_add(c34, em, out_hi, wk<EM, (1ull << 16)>(), out_lo, out);    // :398-403
}
template <int L, class EM, class C>
DEV auto ch_gadget(C c0, EM &em, u32 x, u32 y, u32 z, W<EM> &out) {
    const ChVals v = ch_values(x, y, z);
    const ChW<EM> w = ch_witnesses(em, v);
    return ch_part_b<L>(ch_part_a<L>(c0, em, v, w), em, v, w, out);
}

// maj (compression.rs:460-519)
template <int L, class EM, class C>
DEV auto maj_gadget(C c0, EM &em, u32 x, u32 y, u32 z, W<EM> &out) {
    const W<EM> x_lo = w32<EM>(spread16(x)), x_hi = w32<EM>(spread16(x >> 16));
    const W<EM> y_lo = w32<EM>(spread16(y)), y_hi = w32<EM>(spread16(y >> 16));
    const W<EM> z_lo = w32<EM>(spread16(z)), z_hi = w32<EM>(spread16(z >> 16));
    W<EM> t, m_lo64, m_hi64;
    auto c1 = g_add(c0, em, x_lo, y_lo, t);
    auto c2 = g_add(c1, em, t, z_lo, m_lo64);                // :472-478
    auto c3 = g_add(c2, em, x_hi, y_hi, t);
    auto c4 = g_add(c3, em, t, z_hi, m_hi64);                // :479-485
    const u32 m_lo = (u32)m_lo64.v, m_hi = (u32)m_hi64.v;
    const W<EM> m_lo_even = w16(em, even_bits(m_lo)), m_lo_odd = w16(em, even_bits(m_lo >> 1));
    const W<EM> m_hi_even = w16(em, even_bits(m_hi)), m_hi_odd = w16(em, even_bits(m_hi >> 1));
    auto c5 = g_lw(c4, em, m_lo_even);                       // :486-487
    auto c6 = g_lw(c5, em, m_lo_odd);
    auto c7 = g_lw(c6, em, m_hi_even);                       // :488-489
    auto c8 = g_lw(c7, em, m_hi_odd);
    lookup16(em, (u32)m_lo_even.v); lookup16(em, (u32)m_lo_odd.v); lookup16(em, (u32)m_hi_even.v); lookup16(em, (u32)m_hi_odd.v);
    auto c9 = recheck_even_odd<L>(c8, em, m_lo_even, m_lo_odd);      // :490-500
    auto c10 = recheck_even_odd<L>(c9, em, m_hi_even, m_hi_odd);     // :501-511
    out = w32<EM>(((u32)m_hi_odd.v << 16) | (u32)m_lo_odd.v);
    return g_mul_add(c10, em, m_hi_odd, wk<EM, (1ull << 16)>(), m_lo_odd, out);    // :512-517
}

// one message word from its four bytes: compression.rs:31-47, bytes[3 - idx] * 2^(8 idx) + sum
template <class EM, class C>
DEV auto word_unit(C c0, EM &em, u32 word) {
    const W<EM> b0 = w8(em, word & 0xffu), b1 = w8(em, (word >> 8) & 0xffu), b2 = w8(em, (word >> 16) & 0xffu), b3 = w8(em, word >> 24);
    const W<EM> s1 = w16(em, word & 0xffffu), s2 = w32<EM>(word & 0xffffffu), s3 = w32<EM>(word);
    auto c1 = g_mul_add(c0, em, b0, wk<EM, 1>(), wk<EM, 0>(), b0);
    auto c2 = g_mul_add(c1, em, b1, wk<EM, (1ull << 8)>(), b0, s1);
    auto c3 = g_mul_add(c2, em, b2, wk<EM, (1ull << 16)>(), s1, s2);
    return g_mul_add(c3, em, b3, wk<EM, (1ull << 24)>(), s2, s3);
}

// Which waves of a block expand which phase.  Normally every wave takes a share of every phase.  In
// split mode (tiny batches, 32 waves per block) each wave runs ONE phase program -- rounds on 16 waves,
// schedule steps on 8, words / message spreads / state spreads / feed-forward on 2 each -- so the
// latency of a launch is the chain plus the longest program (a round), not the sum of all six.
enum { PH_WORDS = 0, PH_MSG, PH_SCHED, PH_STATE, PH_ROUNDS, PH_FEED };
DEV bool phase_window(bool split, int phase, u32 part, u32 parts, u32 &wpart, u32 &wparts) {
    if (!split) { wpart = part; wparts = parts; return true; }
    const u32 first = phase == PH_ROUNDS ? 0u : phase == PH_SCHED ? 16u : phase == PH_WORDS ? 24u
                    : phase == PH_MSG ? 26u : phase == PH_STATE ? 28u : 30u;
    const u32 count = phase == PH_ROUNDS ? 16u : phase == PH_SCHED ? 8u : 2u;
    const bool in = part >= first && part < first + count;
    wpart = in ? part - first : 0u;
    wparts = count;
    return in;
}

// --------------------------------------------------------------- the kernel
// T = tile width in cells (contiguous run per row = 32*T bytes), R = tile rows =
// units one wave expands per phase; a block needs parts >= 64/R waves.
// One wave's program for its block (or its share of one).
template <int L, int T, int R, int REPR, bool RC, bool EMITS>
DEV void expand_block(const ExpandParams &p, u64 *s_tile, u64 *s_head, u16 *s_d16, u16 *s_lk16) {
    using LY = Lay<L, RC>;
    using EM = Em<T, R, REPR, RC, false, EMITS>;
    static_assert(R * EM::STRIDE_W * 8 >= 800, "tile must be able to hold the chain seeds");
    u32 *sW = reinterpret_cast<u32 *>(s_tile);   // [64]
    u32 *sA = sW + 64;         // [68] sA[k] = a-value A[k-3]: A[-3..0] = d,c,b,a of the pre-state
    u32 *sE = sA + 68;         // [68] sE[k] = e-value E[k-3]: E[-3..0] = h,g,f,e of the pre-state

    HSW_STAMP(0);
    const u32 lane = lane_id();
    const u32 parts = p.parts;                   // waves per block (power of two <= 16)
    const size_t blk = blockIdx.x / parts;
    const u32 part = blockIdx.x % parts;

    // ---- chain phase: plain SHA-256 of this block, wave-uniform -------------
    if constexpr (EMITS) {
        const u32 *bw = reinterpret_cast<const u32 *>(p.blocks + 64 * blk);
        const u32 *ps = p.pre_states + 8 * blk;
        u32 w[16];
#pragma unroll
        for (int i = 0; i < 16; i++) {
            w[i] = __builtin_bswap32(bw[i]);                 // big-endian words (compression.rs:31-47)
            if (lane == 0) sW[i] = w[i];
        }
        u32 a = ps[0], b = ps[1], c = ps[2], d = ps[3], e = ps[4], f = ps[5], g = ps[6], h = ps[7];
        if (lane == 0) {
            sA[0] = d; sA[1] = c; sA[2] = b; sA[3] = a;
            sE[0] = h; sE[1] = g; sE[2] = f; sE[3] = e;
        }
#pragma unroll
        for (int t = 0; t < 64; t++) {
            if (t >= 16) {
                const u32 w15 = w[(t - 15) & 15], w2 = w[(t - 2) & 15];
                const u32 s0 = rotr32(w15, 7) ^ rotr32(w15, 18) ^ (w15 >> 3);
                const u32 s1 = rotr32(w2, 17) ^ rotr32(w2, 19) ^ (w2 >> 10);
                w[t & 15] = w[t & 15] + s0 + w[(t - 7) & 15] + s1;
                if (lane == 0) sW[t] = w[t & 15];
            }
            const u32 S1 = rotr32(e, 6) ^ rotr32(e, 11) ^ rotr32(e, 25);
            const u32 chv = (e & f) ^ (~e & g);
            const u32 t1 = h + S1 + chv + K256[t] + w[t & 15];
            const u32 S0 = rotr32(a, 2) ^ rotr32(a, 13) ^ rotr32(a, 22);
            const u32 mj = (a & b) ^ (a & c) ^ (b & c);
            const u32 t2 = S0 + mj;
            h = g; g = f; f = e; e = d + t1; d = c; c = b; b = a; a = t1 + t2;
            if (lane == 0) { sA[t + 4] = a; sE[t + 4] = e; }
        }
        if (p.next_states != nullptr && lane == 0 && part == 0) {
            u32 *ns = p.next_states + 8 * blk;               // compression.rs:197-212
            ns[0] = ps[0] + a; ns[1] = ps[1] + b; ns[2] = ps[2] + c; ns[3] = ps[3] + d;
            ns[4] = ps[4] + e; ns[5] = ps[5] + f; ns[6] = ps[6] + g; ns[7] = ps[7] + h;
        }
    }
    __syncthreads();
    HSW_STAMP(1);

    // ---- every lane pulls the seeds of its units into registers -------------
    auto pre_word = [&](u32 i) -> u32 { return i < 4 ? sA[3 - i] : sE[7 - i]; };   // a..d = sA[3..0], e..h = sE[3..0]
    const bool split = (p.flags & HSW_K_SPLIT) != 0u;        // parts == 32 then
    u32 wp[6], wn[6];
    bool in_phase[6];
#pragma unroll
    for (int ph = 0; ph < 6; ph++) in_phase[ph] = phase_window(split, ph, part, parts, wp[ph], wn[ph]);
    const u32 uw = lane_unit(wp[PH_WORDS], wn[PH_WORDS], 16), um = lane_unit(wp[PH_MSG], wn[PH_MSG], 16);
    const u32 us = lane_unit(wp[PH_SCHED], wn[PH_SCHED], 48), ust = lane_unit(wp[PH_STATE], wn[PH_STATE], 6);
    const u32 ur = lane_unit(wp[PH_ROUNDS], wn[PH_ROUNDS], 64), uf = lane_unit(wp[PH_FEED], wn[PH_FEED], 8);
    const u32 seed_word = sW[uw], seed_word_msg = sW[um];
    const u32 seed_w2 = sW[us + 14], seed_w15 = sW[us + 1], seed_w7 = sW[us + 9], seed_w16 = sW[us];
    const u32 seed_state = pre_word(ust < 3 ? ust : ust + 1);
    const u32 seed_a = sA[ur + 3], seed_b = sA[ur + 2], seed_c = sA[ur + 1], seed_d = sA[ur];
    const u32 seed_e = sE[ur + 3], seed_f = sE[ur + 2], seed_g = sE[ur + 1], seed_h = sE[ur];
    const u32 seed_wr = sW[ur];
    const u32 seed_fx = uf < 4 ? sA[67 - uf] : sE[71 - uf], seed_fy = pre_word(uf);
    __syncthreads();           // seeds are in registers: the tile may now overwrite them
    HSW_STAMP(2);

    EM em;
    em.lk16 = s_lk16;
    em.tile = s_tile;
    em.row0 = s_tile + (lane < (u32)R ? lane : (u32)R) * EM::STRIDE_W;   // lanes >= R never flush: scratch row
    em.row = em.row0;

    em.skew = 0;
    em.carry_neg = 0;
    em.head = s_head;
    em.d16 = s_d16;
    em.tab = static_cast<const uint4 *>(p.mont_tab);
    {   // FlexGate column packing: gaps of the breaks at or before this block, and the (<= 2) inside it
        u64 first = (u64)blk * (u64)LY::GATE_CELLS;
        if constexpr (RC)          // whole-digest streams: every frame_every blocks a digest frame sits in between
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
        if constexpr (REPR == 2)   // compact: 8-byte cells
            em.out = reinterpret_cast<uint4 *>(reinterpret_cast<u64 *>(p.gate) + (size_t)(first + gap0));
        else
            em.out = reinterpret_cast<uint4 *>(p.gate) + (size_t)(first + gap0) * 2u;
    }
    size_t lk_blk = (size_t)blk * (size_t)LY::LOOKUP_CELLS;
    if constexpr (RC)
        if (p.frame_every) lk_blk += (size_t)(blk / p.frame_every) * (size_t)p.frame_lookups;
    em.write_gate = (p.flags & HSW_K_SKIP_GATE) == 0u;
    const u64 blk_limb0 = p.cursor0 + (u64)blk * (u64)LY::LIMB_CALLS;   // first limb call of this block

    // ---- words: compression.rs:31-47, 16 units of 4 mul_add ----------------
    if (in_phase[PH_WORDS] && phase_begin(em, wp[PH_WORDS], wn[PH_WORDS], 16, LY::WORD, LY::OFF_WORDS, 0, 0)) {
        phase_end<L>(word_unit(CurStart{}, em, seed_word), em, p, blk_limb0, lk_blk);
    }

    // ---- 16 x state_to_spread_u32(W[i]): compression.rs:53-56 --------------
    if (in_phase[PH_MSG] && phase_begin(em, wp[PH_MSG], wn[PH_MSG], 16, LY::S2S, LY::OFF_MSG, LY::CALL_MSG, LY::CALLS_S2S)) {
        auto c1 = state_to_spread<L>(CurStart{}, em, w32<EM>(seed_word_msg));
        phase_end<L>(c1, em, p, blk_limb0, lk_blk);
    }

    // ---- schedule: compression.rs:57-96, 48 units --------------------------
    if (in_phase[PH_SCHED] && phase_begin(em, wp[PH_SCHED], wn[PH_SCHED], 48, LY::SCHED, LY::OFF_SCHED, LY::CALL_SCHED, LY::CALLS_SCHED,
                    LY::LK_OFF_SCHED, LY::LK_SCHED)) {
        const u32 w2 = seed_w2, w15 = seed_w15, w7 = seed_w7, w16 = seed_w16;   // W[idx-2], [idx-15], [idx-7], [idx-16]
        W<EM> term1, term3, new_w, sum;
        auto c1 = sigma_generic<SigmaLower1, L>(CurStart{}, em, w2, term1);    // :60
        auto c2 = sigma_generic<SigmaLower0, L>(c1, em, w15, term3);           // :61
        auto c3 = g_add(c2, em, term1, w32<EM>(w7), sum);                      // :65-69
        auto c4 = g_add(c3, em, sum, term3, sum);                              // :70-74
        auto c5 = g_add(c4, em, sum, w32<EM>(w16), sum);                       // :75-79
        auto c6 = mod_u32(c5, em, sum, new_w);                                 // :80
        auto c7 = state_to_spread<L>(c6, em, new_w);                           // :90
        phase_end<L>(c7, em, p, blk_limb0, lk_blk);
    }

    // ---- 6 x state_to_spread_u32 of a,b,c,e,f,g: compression.rs:109-115 ----
    if (in_phase[PH_STATE] && phase_begin(em, wp[PH_STATE], wn[PH_STATE], 6, LY::S2S, LY::OFF_STATE, LY::CALL_STATE, LY::CALLS_S2S)) {
        auto c1 = state_to_spread<L>(CurStart{}, em, w32<EM>(seed_state));
        phase_end<L>(c1, em, p, blk_limb0, lk_blk);
    }

    // ---- 64 rounds: compression.rs:125-196 ---------------------------------
    if (in_phase[PH_ROUNDS] && phase_begin(em, wp[PH_ROUNDS], wn[PH_ROUNDS], 64, LY::ROUND, LY::OFF_ROUNDS, LY::CALL_ROUNDS, LY::CALLS_ROUND,
                    LY::LK_OFF_ROUNDS, LY::LK_ROUND)) {
        const u32 a = seed_a, b = seed_b, c = seed_c, d = seed_d;
        const u32 e = seed_e, f = seed_f, g = seed_g, h = seed_h;
        W<EM> sig1, chv, t1, sig0, mjv, t2, e_new, a_new, s;
        auto c1 = sigma_generic<SigmaUpper1, L>(CurStart{}, em, e, sig1);      // :130
        auto c2 = ch_gadget<L>(c1, em, e, f, g, chv);                          // :131
        auto c3 = g_add(c2, em, w32<EM>(h), sig1, s);                          // :138-142
        auto c4 = g_add(c3, em, s, chv, s);                                    // :143-147
        auto c5 = g_add(c4, em, s, wround_constant<EM>(ur), s);                // :148-152
        auto c6 = g_add(c5, em, s, w32<EM>(seed_wr), s);                       // :153-157
        auto c7 = mod_u32(c6, em, s, t1);                                      // :158
        auto c8 = sigma_generic<SigmaUpper0, L>(c7, em, a, sig0);              // :164
        auto c9 = maj_gadget<L>(c8, em, a, b, c, mjv);                         // :165
        auto c10 = g_add(c9, em, sig0, mjv, s);                                // :166-170
        auto c11 = mod_u32(c10, em, s, t2);                                    // :171
        auto c12 = g_add(c11, em, w32<EM>(d), t1, s);                          // :181
        auto c13 = mod_u32(c12, em, s, e_new);                                 // :182
        auto c14 = state_to_spread<L>(c13, em, e_new);                         // :184
        auto c15 = g_add(c14, em, t1, t2, s);                                  // :192
        auto c16 = mod_u32(c15, em, s, a_new);                                 // :193
        auto c17 = state_to_spread<L>(c16, em, a_new);                         // :195
        phase_end<L>(c17, em, p, blk_limb0, lk_blk);
    }
    HSW_STAMP(3);

    // ---- feed-forward: compression.rs:197-212, 8 units ---------------------
    if (in_phase[PH_FEED] && phase_begin(em, wp[PH_FEED], wn[PH_FEED], 8, LY::FEED, LY::OFF_FEED, 0, 0, LY::LK_OFF_FEED, LY::LK_FEED)) {
        W<EM> s, lo;
        auto c1 = g_add(CurStart{}, em, w32<EM>(seed_fx), w32<EM>(seed_fy), s);
        auto c2 = mod_u32(c1, em, s, lo);
        phase_end<L>(c2, em, p, blk_limb0, lk_blk);
    }
    HSW_STAMP(4);
}

template <int L, int T, int R, int REPR, bool RC>
__global__ __launch_bounds__(64) void hsw_expand_kernel(ExpandParams p) {
    using LY = Lay<L, RC>;
    using EM = Em<T, R, REPR, RC>;
    // The chain seeds live in LDS only until every lane has pulled its own into
    // registers; the tile then reuses the same bytes (keeps the workgroup at
    // <= 20 KiB of LDS = 8 waves per CU, so 4,096 blocks are exactly 2 waves of
    // residency on 256 CUs).
    __shared__ __attribute__((aligned(16))) u64 s_tile[(R + (R < 64 ? 1 : 0)) * EM::STRIDE_W];   // +1 scratch row for lanes >= R
    __shared__ __attribute__((aligned(16))) u64 s_head[EM::REALIGN ? R * EM::HEAD_W : 2];   // realignment: held-back first cells of every unit (flush_tile)
    __shared__ u16 s_d16[R * LY::CALLS_ROUND];    // largest phase-part: R rounds x 24 spread calls
    __shared__ u16 s_lk16[RC ? R * LY::LK_ROUND : 1];   // lookup-column staging (internals mode only)
    expand_block<L, T, R, REPR, RC, true>(p, s_tile, s_head, s_d16, s_lk16);
}

// ------------------------------------------------------------------ launch
template <int L, int T, int R, bool RC>
static hipError_t launch_expand_LTR(const ExpandParams &p, hipStream_t stream) {
    if ((p.flags & HSW_K_SPLIT) ? (p.parts != 32u || R < 8) : (p.parts * (unsigned)R < 64u))
        return hipErrorInvalidValue;                                 // every unit needs a row
    const dim3 grid((unsigned)(p.n_blocks * p.parts)), block(64);
    if (p.flags & HSW_K_MONTGOMERY)
        hipLaunchKernelGGL((hsw_expand_kernel<L, T, R, 1, RC>), grid, block, 0, stream, p);
    else if (p.flags & HSW_K_COMPACT)
        hipLaunchKernelGGL((hsw_expand_kernel<L, T, R, 2, RC>), grid, block, 0, stream, p);
    else
        hipLaunchKernelGGL((hsw_expand_kernel<L, T, R, 0, RC>), grid, block, 0, stream, p);
    return hipGetLastError();
}

// halo2-base internals for the 2- and 1-bit tables (8 / 16 limbs per spread): these kernels compile for ~30 s
// each, so they live in translation units of their own (hsw_expand_l8_rc.hip, hsw_expand_l16_rc.hip) and only
// the two 32-byte representations are built (no HSW_REPR_COMPACT64).
template <int L>
hipError_t launch_expand_L_internals_wide(const ExpandParams &p, hipStream_t stream) {
    if (p.flags & (HSW_K_COMPACT | HSW_K_SPLIT)) return hipErrorInvalidValue;
    if (p.parts * 64u < 64u) return hipErrorInvalidValue;
    const dim3 grid((unsigned)(p.n_blocks * p.parts)), block(64);
    if (p.flags & HSW_K_MONTGOMERY)
        hipLaunchKernelGGL((hsw_expand_kernel<L, 32, 64, 1, true>), grid, block, 0, stream, p);
    else
        hipLaunchKernelGGL((hsw_expand_kernel<L, 32, 64, 0, true>), grid, block, 0, stream, p);
    return hipGetLastError();
}
extern template hipError_t launch_expand_L_internals_wide<8>(const ExpandParams &, hipStream_t);
extern template hipError_t launch_expand_L_internals_wide<16>(const ExpandParams &, hipStream_t);

// Montgomery cells built at emit time (Em::M32): [64 rows][8 cells] tiles of 32-byte cells (20 KiB of LDS: 7 waves per
// CU), one wave per block.  Built for the reference's 8-bit table, in a translation unit of its own
// (hsw_expand_l2_m32.hip).
template <int L>
hipError_t launch_expand_m32(const ExpandParams &p, hipStream_t stream) {
    if ((p.flags & (HSW_K_COMPACT | HSW_K_SPLIT)) || !(p.flags & HSW_K_MONTGOMERY) || p.parts != 1u) return hipErrorInvalidValue;
    const dim3 grid((unsigned)p.n_blocks), block(64);
    if (p.flags & HSW_K_INTERNALS) hipLaunchKernelGGL((hsw_expand_kernel<L, HSW_M32_TILE, 64, 3, true>), grid, block, 0, stream, p);
    else hipLaunchKernelGGL((hsw_expand_kernel<L, HSW_M32_TILE, 64, 3, false>), grid, block, 0, stream, p);
    return hipGetLastError();
}
extern template hipError_t launch_expand_m32<2>(const ExpandParams &, hipStream_t);

template <int L>
hipError_t launch_expand_L(const ExpandParams &p, int tile, hipStream_t stream) {
    if (p.n_blocks == 0) return hipSuccess;
    if (p.flags & HSW_K_M32) {
        if constexpr (L == 2) return launch_expand_m32<2>(p, stream);
        else return hipErrorInvalidValue;
    }
    if (p.flags & HSW_K_INTERNALS) {
        // halo2-base internals (A3): the wider tiles for the reference's 8-bit table only
        if constexpr (L == 2) {
            switch (tile) {
                case 64: return launch_expand_LTR<L, 64, 32, true>(p, stream);
                case 128: return launch_expand_LTR<L, 128, 16, true>(p, stream);
                default: return launch_expand_LTR<L, 32, 64, true>(p, stream);
            }
        } else if constexpr (L <= 4) return launch_expand_LTR<L, 32, 64, true>(p, stream);
        else return launch_expand_L_internals_wide<L>(p, stream);
    }
    switch (tile) {
        case 6416: if constexpr (L == 2) return launch_expand_LTR<L, 64, 16, false>(p, stream); else return hipErrorInvalidValue;
        case 32: return launch_expand_LTR<L, 32, 64, false>(p, stream);
        case 64: if constexpr (L == 2) return launch_expand_LTR<L, 64, 32, false>(p, stream); else return hipErrorInvalidValue;
        case 128: if constexpr (L == 2) return launch_expand_LTR<L, 128, 16, false>(p, stream); else return hipErrorInvalidValue;
        // [8][256], [4][512] and [2][768] tiles (8-24 KiB runs) were measured too: they need 8-32 waves
        // per block, and the redundant chain work then costs more than the longer runs gain
        // (1.80 / 1.99 / 2.55 ms against 1.72-1.78 ms).
        default: return hipErrorInvalidValue;
    }
}

}  // namespace hsw
#endif
