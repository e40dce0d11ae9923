// hsw_structure.hpp -- the constraint STRUCTURE of one block's gate stream, built on the host.
//
// The value streams say what every advice cell holds; a replayer that rebuilds the gadget's
// constraint system around them (selectors, fixed constants, copy constraints: what halo2's key
// generation records) also needs to know, for every cell, which QuantumCell the reference passed to
// halo2-base at that position:
//     Witness            a fresh value (the stream cell is the only source)
//     Constant(k)        fixed to k
//     Existing(cell)     copy-constrained to an earlier cell
// plus the gate rows (x0 + x1*x2 = x3 on cells [r, r+3]), the assert_equal pairs, the range_check
// bounds, which cell every lookup-column entry copies, and which cells the spread-chip cells are tied
// to.  All of it is input independent.  This builder walks the reference's call sequence
// (compression.rs:19-213 and what it calls; same citations as the kernel and hsw_tape.hpp) on
// symbolic cells only -- no values -- and is checked against the oracle's recorder, which derives the
// same structure independently while computing values (tests/test_structure.py).
//
// Cell ids: >= 0 block-relative stream index; negative = cells outside the block's stream:
//   -1 - k    input byte k of the block (assigned by digest(), lib.rs:170-173)
//   -100 - i  pre-state word i          (lib.rs:162-165 / the previous block's output)
//   -1000     the Context's cached zero cell
//   -2000     a halo2-base witness that is not in the stream (range-check limbs without internals)
#ifndef HSW_STRUCTURE_HPP
#define HSW_STRUCTURE_HPP

#include <cstdint>
#include <vector>

namespace hsw {

struct BlockStructure {
    enum : uint8_t { WITNESS = 0, CONSTANT = 1, EXISTING = 2 };
    enum : int64_t { INPUT_BYTE0 = -1, PRE_STATE0 = -100, ZERO = -1000, HIDDEN = -2000 };
    std::vector<uint8_t> kind;         // per gate cell
    std::vector<int64_t> ref;          // CONSTANT: the constant (k < 2^63; -k stands for p - k); EXISTING: the cell copied
    std::vector<uint32_t> gate_rows;   // first cell of every enabled gate row
    std::vector<int64_t> assert_eq;    // pairs (a, b): constrain_equal outside cell assignment (assert_equal, range_check acc)
    std::vector<int64_t> range;        // pairs (cell, bits): range_check(cell, bits)
    std::vector<int64_t> lookup_src;   // lookup-column entry j copies this cell
    std::vector<int64_t> chip;         // limb call n: (cell tied to the dense chip cell, cell tied to the spread chip cell)
    int64_t next_state[8];
};

class StructureBuilder {
  public:
    StructureBuilder(int limbs, bool internals) : L(limbs), bits(16 / limbs), rc(internals) {}

    BlockStructure block() {                                 // compression.rs:19-213
        s = BlockStructure();
        AV bytes[64], pre[8];
        for (int i = 0; i < 64; i++) bytes[i] = ext(BlockStructure::INPUT_BYTE0 - i);
        for (int i = 0; i < 8; i++) pre[i] = ext(BlockStructure::PRE_STATE0 - i);
        AV w32[64];
        SP wsp[64];
        for (int w = 0; w < 16; w++) {                       // :31-47
            AV sum = zero();
            for (int idx = 0; idx < 4; idx++) sum = mul_add(bytes[4 * w + 3 - idx], K(1ull << (8 * idx)), sum);
            w32[w] = sum;
        }
        for (int w = 0; w < 16; w++) wsp[w] = state_to_spread(w32[w]);   // :53-56
        for (int idx = 16; idx < 64; idx++) {                // :57-96
            AV term1 = sigma(wsp[idx - 2], SIGMA_LOWER1);
            AV term3 = sigma(wsp[idx - 15], SIGMA_LOWER0);
            AV sum = add(term1, w32[idx - 7]);
            sum = add(sum, term3);
            sum = add(sum, w32[idx - 16]);
            w32[idx] = mod_u32(sum);
            wsp[idx] = state_to_spread(w32[idx]);
        }
        AV a = pre[0], b = pre[1], c = pre[2], d = pre[3], e = pre[4], f = pre[5], g = pre[6], h = pre[7];
        SP as = state_to_spread(a), bs = state_to_spread(b), cs = state_to_spread(c);   // :109-111
        SP es = state_to_spread(e), fs = state_to_spread(f), gs = state_to_spread(g);   // :113-115
        for (int idx = 0; idx < 64; idx++) {                 // :125-196
            AV t1, t2;
            {
                AV sg = sigma(es, SIGMA_UPPER1);
                AV chv = ch(es, fs, gs);
                AV s1 = add(h, sg);
                AV s2 = add(s1, chv);
                AV s3 = add(s2, K(ROUND_K[idx]));
                AV s4 = add(s3, w32[idx]);
                t1 = mod_u32(s4);
            }
            {
                AV sg = sigma(as, SIGMA_UPPER0);
                AV mj = maj(as, bs, cs);
                t2 = mod_u32(add(sg, mj));
            }
            h = g;
            g = f; gs = fs;
            f = e; fs = es;
            e = mod_u32(add(d, t1));
            es = state_to_spread(e);
            d = c;
            c = b; cs = bs;
            b = a; bs = as;
            a = mod_u32(add(t1, t2));
            as = state_to_spread(a);
        }
        const AV ns[8] = {a, b, c, d, e, f, g, h};
        for (int i = 0; i < 8; i++) s.next_state[i] = mod_u32(add(ns[i], pre[i])).cell;   // :197-212
        return s;
    }

  private:
    struct AV { int64_t cell; bool is_const; uint64_t k; };       // an AssignedValue or a QuantumCell::Constant
    struct SP { AV lo, hi; };                                     // SpreadU32
    enum Sigma { SIGMA_UPPER0, SIGMA_UPPER1, SIGMA_LOWER0, SIGMA_LOWER1 };
    int L, bits;
    bool rc;
    BlockStructure s;
    static const uint32_t ROUND_K[64];

    static AV ext(int64_t id) { return AV{id, false, 0}; }
    static AV K(uint64_t k) { return AV{0, true, k}; }
    AV zero() const { return ext(BlockStructure::ZERO); }          // load_zero: cached by the Context (A2)

    // one advice cell holding QuantumCell q (Witness if `witness`)
    AV put(const AV &q, bool witness) {
        const int64_t idx = (int64_t)s.kind.size();
        if (witness) { s.kind.push_back(BlockStructure::WITNESS); s.ref.push_back(0); }
        else if (q.is_const) { s.kind.push_back(BlockStructure::CONSTANT); s.ref.push_back((int64_t)q.k); }
        else { s.kind.push_back(BlockStructure::EXISTING); s.ref.push_back(q.cell); }
        return AV{idx, false, 0};
    }
    void row() { s.gate_rows.push_back((uint32_t)s.kind.size()); }
    AV load_witness() { return put(AV{}, true); }
    AV add(const AV &a, const AV &b) { row(); put(a, false); put(b, false); put(K(1), false); return put(AV{}, true); }
    AV neg(const AV &a) {                                          // [a, -a, 1, 0]
        row(); put(a, false);
        const AV out = put(AV{}, true);
        put(K(1), false); put(K(0), false);
        return out;
    }
    AV mul_add(const AV &a, const AV &b, const AV &c) { row(); put(c, false); put(a, false); put(b, false); return put(AV{}, true); }
    void assert_equal(const AV &a, const AV &b) { s.assert_eq.push_back(a.cell); s.assert_eq.push_back(b.cell); }
    void lookup(const AV &v) { s.lookup_src.push_back(v.cell); }
    void range_check(const AV &a, int nbits) {                     // lookup_bits = 16 (A3)
        s.range.push_back(a.cell); s.range.push_back(nbits);
        if (nbits <= 16) { lookup(a); return; }                    // 16: `a` itself is looked up
        AV limb0 = ext(BlockStructure::HIDDEN), limb1 = ext(BlockStructure::HIDDEN);
        if (rc) {                                                  // [limb0, limb1, 2^16, acc]
            row();
            limb0 = put(AV{}, true);
            limb1 = put(AV{}, true);
            put(K(1ull << 16), false);
            const AV acc = put(AV{}, true);
            assert_equal(acc, a);
        }
        lookup(limb0); lookup(limb1);
    }
    AV spread_limb(const AV &limb) {                               // spread.rs:196-233
        const AV sp = load_witness();
        s.chip.push_back(limb.cell); s.chip.push_back(sp.cell);
        return sp;
    }
    AV spread(const AV &dense) {                                   // spread.rs:76-123
        AV limbs[16];
        for (int i = 0; i < L; i++) limbs[i] = load_witness();
        AV sum = zero();
        for (int i = 0; i < L; i++) sum = mul_add(limbs[i], K(1ull << (bits * i)), sum);
        assert_equal(sum, dense);
        AV acc = zero();
        for (int i = 0; i < L; i++) {
            const AV sl = spread_limb(limbs[i]);
            acc = mul_add(sl, K(1ull << (2 * bits * i)), acc);
        }
        return acc;
    }
    void even_odd(AV &even, AV &odd) {                             // spread.rs:139-163
        even = load_witness(); odd = load_witness();
        range_check(even, 16); range_check(odd, 16);
    }
    SP state_to_spread(const AV &x) {                              // compression.rs:215-246
        const AV lo = load_witness(), hi = load_witness();
        const AV composed = mul_add(hi, K(1ull << 16), lo);
        assert_equal(x, composed);
        SP r;
        r.lo = spread(lo);
        r.hi = spread(hi);
        return r;
    }
    AV mod_u32(const AV &x) {                                      // compression.rs:266-295
        const AV lo = load_witness(), hi = load_witness();
        range_check(lo, 32);
        const AV composed = mul_add(hi, K(1ull << 32), lo);
        assert_equal(x, composed);
        return lo;
    }
    void recheck(const AV &even, const AV &odd, const AV &whole) { // compression.rs:344-354 and siblings
        const AV es = spread(even), os = spread(odd);
        assert_equal(mul_add(K(2), os, es), whole);
    }
    AV ch(const SP &x, const SP &y, const SP &z) {                 // compression.rs:297-405
        const AV p_lo = add(x.lo, y.lo), p_hi = add(x.hi, y.hi);
        const AV xn_lo = neg(x.lo), xn_hi = neg(x.hi);
        const AV q_lo = add(add(K(0x55555555ull), xn_lo), z.lo);
        const AV q_hi = add(add(K(0x55555555ull), xn_hi), z.hi);
        AV ple, plo, phe, pho, qle, qlo, qhe, qho;
        even_odd(ple, plo); even_odd(phe, pho); even_odd(qle, qlo); even_odd(qhe, qho);
        recheck(ple, plo, p_lo); recheck(phe, pho, p_hi); recheck(qle, qlo, q_lo); recheck(qhe, qho, q_hi);
        const AV out_lo = add(plo, qlo), out_hi = add(pho, qho);
        return mul_add(out_hi, K(1ull << 16), out_lo);
    }
    AV maj(const SP &x, const SP &y, const SP &z) {                // compression.rs:460-519
        const AV m_lo = add(add(x.lo, y.lo), z.lo), m_hi = add(add(x.hi, y.hi), z.hi);
        AV mle, mlo, mhe, mho;
        even_odd(mle, mlo); even_odd(mhe, mho);
        recheck(mle, mlo, m_lo); recheck(mhe, mho, m_hi);
        return mul_add(mho, K(1ull << 16), mlo);
    }
    AV sigma(const SP &x, Sigma which) {                           // compression.rs:594-882
        static const int STARTS[4][4] = {{0, 2, 13, 22}, {0, 6, 11, 25}, {0, 3, 7, 18}, {0, 10, 17, 19}};
#define HSW_P2(n) (1ull << (n))
        static const uint64_t COEFFS[4][4] = {
            {HSW_P2(60) + HSW_P2(38) + HSW_P2(20), HSW_P2(0) + HSW_P2(42) + HSW_P2(24), HSW_P2(22) + HSW_P2(0) + HSW_P2(46), HSW_P2(40) + HSW_P2(18) + HSW_P2(0)},
            {HSW_P2(52) + HSW_P2(42) + HSW_P2(14), HSW_P2(0) + HSW_P2(54) + HSW_P2(26), HSW_P2(10) + HSW_P2(0) + HSW_P2(36), HSW_P2(38) + HSW_P2(28) + HSW_P2(0)},
            {HSW_P2(50) + HSW_P2(28), HSW_P2(0) + HSW_P2(56) + HSW_P2(34), HSW_P2(8) + HSW_P2(0) + HSW_P2(42), HSW_P2(30) + HSW_P2(22) + HSW_P2(0)},
            {HSW_P2(30) + HSW_P2(26), HSW_P2(0) + HSW_P2(50) + HSW_P2(46), HSW_P2(14) + HSW_P2(0) + HSW_P2(60), HSW_P2(18) + HSW_P2(4) + HSW_P2(0)}};
#undef HSW_P2
        const int *st = STARTS[which];
        AV piece[4];
        for (int i = 0; i < 4; i++) piece[i] = load_witness();                      // :719-734
        AV sum = piece[0];                                                           // :736-754
        for (int i = 1; i < 4; i++) sum = mul_add(piece[i], K(1ull << (2 * st[i])), sum);
        const AV x_composed = mul_add(x.hi, K(1ull << 32), x.lo);                    // :755-760
        assert_equal(x_composed, sum);
        AV r = zero();                                                               // :780-808
        for (int i = 0; i < 4; i++) r = mul_add(K(COEFFS[which][i]), piece[i], r);
        const AV r_lo = load_witness(), r_hi = load_witness();                       // :820-821
        range_check(r_lo, 32); range_check(r_hi, 32);                                // :822-823
        assert_equal(r, mul_add(r_hi, K(1ull << 32), r_lo));                         // :824-834
        AV le, lo_, he, ho;
        even_odd(le, lo_); even_odd(he, ho);                                         // :843-846
        recheck(le, lo_, r_lo); recheck(he, ho, r_hi);                               // :852-873
        return mul_add(he, K(1ull << 16), le);                                       // :874-879
    }
};

// The same for the frame digest() puts around the block loop (hsw_frame.hpp, assumption A4).  Cell ids
// are section-relative (>= 0); negative ids name cells of other sections:
//   ZERO                 the Context's zero cell
//   TARGET               assigned_target_round (prologue cell P_TGT)
//   STATE0 - (8 n + i)   word i of candidate state n: n = 0 the prologue's state cells, n >= 1 the
//                        next_state cells of block n - 1
struct FrameStructure {
    enum : int64_t { ZERO = -1000, TARGET = -3000, STATE0 = -4000 };
    std::vector<uint8_t> kind;
    std::vector<int64_t> ref;
    std::vector<uint32_t> gate_rows;
    std::vector<int64_t> assert_eq;     // pairs of cells
    std::vector<int64_t> assert_const;  // pairs (cell, k): assert_is_const
    std::vector<int64_t> range;         // pairs (cell, bits)
    std::vector<int64_t> lookup_src;
};

class FrameStructureBuilder {
  public:
    FrameStructure prologue(uint64_t max_bytes, bool rc_inputs) {          // lib.rs:122-178
        s = FrameStructure();
        const int64_t len = W(), nround = W();                              // :124-126
        row(); C(0); E(nround); C(64); const int64_t padded = W();          // :127-131 mul
        row(); E(len); C(9); C(1); const int64_t with9 = W();               // :132-136 add
        row(); const int64_t pad = W(); E(with9); C(1); E(padded);          // :137-141 sub
        // :142-143 is_less_than_safe(padding_size, 64): range_check(., 16); is_less_than
        s.range.push_back(pad); s.range.push_back(16); s.lookup_src.push_back(pad);
        const uint32_t lt0 = (uint32_t)s.kind.size();                       // the 7-cell region: gate rows at 0 and 3
        s.gate_rows.push_back(lt0); s.gate_rows.push_back(lt0 + 3);
        const int64_t shifted = W(); C(64); C(1); W(); C(-65536); C(1); E(pad);
        s.range.push_back(shifted); s.range.push_back(32);
        row(); const int64_t limb0 = W(), limb1 = W(); C(65536); const int64_t acc = W();
        s.assert_eq.push_back(acc); s.assert_eq.push_back(shifted);
        s.lookup_src.push_back(limb0); s.lookup_src.push_back(limb1);
        const int64_t lt = is_zero(limb1);
        s.assert_const.push_back(lt); s.assert_const.push_back(1);          // :144 assert_is_const
        const int64_t pre = W();                                            // :145-146
        row(); W(); E(pre); C(1); E(nround);                                // :147-151 sub -> target round
        for (int i = 0; i < 8; i++) W();                                    // :162-165
        const int64_t byte0 = (int64_t)s.kind.size();
        for (uint64_t i = 0; i < max_bytes; i++) W();                       // :170-173
        if (rc_inputs)
            for (uint64_t i = 0; i < max_bytes; i++) range_check8(byte0 + (int64_t)i);   // :174-178
        return s;
    }
    FrameStructure epilogue(uint64_t n_blocks) {                            // lib.rs:294-341
        s = FrameStructure();
        int64_t out[8];
        for (int i = 0; i < 8; i++) out[i] = FrameStructure::ZERO;          // :294-295
        for (uint64_t n = 0; n <= n_blocks; n++) {                          // :296-310
            row(); const int64_t diff = W(); C(1); E(FrameStructure::TARGET); C((int64_t)n);   // is_equal
            const int64_t sel = is_zero(diff);
            for (int i = 0; i < 8; i++) {                                   // select(state, out, sel)
                const int64_t st = FrameStructure::STATE0 - (int64_t)(8 * n + (uint64_t)i);
                row(); const int64_t d = W(); C(1); E(out[i]); E(st);
                row(); E(out[i]); E(sel); E(d); out[i] = W();
            }
        }
        for (int w = 0; w < 8; w++) {                                       // :311-341
            int64_t bytes[4];
            for (int idx = 0; idx < 4; idx++) { bytes[idx] = W(); range_check8(bytes[idx]); }
            int64_t sum = FrameStructure::ZERO;                             // :325
            for (int idx = 0; idx < 4; idx++) { row(); E(sum); E(bytes[idx]); C((int64_t)1 << (24 - 8 * idx)); sum = W(); }
            s.assert_eq.push_back(out[w]); s.assert_eq.push_back(sum);      // :334-338
        }
        return s;
    }

  private:
    FrameStructure s;
    int64_t put(uint8_t kind, int64_t ref) {
        s.kind.push_back(kind); s.ref.push_back(ref);
        return (int64_t)s.kind.size() - 1;
    }
    int64_t W() { return put(BlockStructure::WITNESS, 0); }
    int64_t C(int64_t k) { return put(BlockStructure::CONSTANT, k); }
    int64_t E(int64_t cell) { return put(BlockStructure::EXISTING, cell); }
    void row() { s.gate_rows.push_back((uint32_t)s.kind.size()); }
    int64_t is_zero(int64_t a) {                                            // [z, a, inv, 1, 0, a, z, 0]
        row(); const int64_t z = W(); E(a); W(); C(1);
        row(); C(0); E(a); E(z); C(0);
        return z;
    }
    void range_check8(int64_t a) {                                          // lookup a; [0, a, 2^8, a*2^8]; lookup the product
        s.range.push_back(a); s.range.push_back(8);
        s.lookup_src.push_back(a);
        row(); C(0); E(a); C(256); const int64_t prod = W();
        s.lookup_src.push_back(prod);
    }
};

inline const uint32_t StructureBuilder::ROUND_K[64] = {
    0x428a2f98, 0x71374491, 0xb5c0fbcf, 0xe9b5dba5, 0x3956c25b, 0x59f111f1, 0x923f82a4, 0xab1c5ed5,
    0xd807aa98, 0x12835b01, 0x243185be, 0x550c7dc3, 0x72be5d74, 0x80deb1fe, 0x9bdc06a7, 0xc19bf174,
    0xe49b69c1, 0xefbe4786, 0x0fc19dc6, 0x240ca1cc, 0x2de92c6f, 0x4a7484aa, 0x5cb0a9dc, 0x76f988da,
    0x983e5152, 0xa831c66d, 0xb00327c8, 0xbf597fc7, 0xc6e00bf3, 0xd5a79147, 0x06ca6351, 0x14292967,
    0x27b70a85, 0x2e1b2138, 0x4d2c6dfc, 0x53380d13, 0x650a7354, 0x766a0abb, 0x81c2c92e, 0x92722c85,
    0xa2bfe8a1, 0xa81a664b, 0xc24b8b70, 0xc76c51a3, 0xd192e819, 0xd6990624, 0xf40e3585, 0x106aa070,
    0x19a4c116, 0x1e376c08, 0x2748774c, 0x34b0bcb5, 0x391c0cb3, 0x4ed8aa4a, 0x5b9cca4f, 0x682e6ff3,
    0x748f82ee, 0x78a5636f, 0x84c87814, 0x8cc70208, 0x90befffa, 0xa4506ceb, 0xbef9a3f7, 0xc67178f2};

}  // namespace hsw
#endif
