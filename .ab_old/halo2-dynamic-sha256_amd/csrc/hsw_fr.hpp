// hsw_fr.hpp -- host-side BN254 scalar-field (Fr) arithmetic, used once per gadget to
// tabulate k^-1 for the is_zero witnesses of the digest epilogue (is_equal(n_round,
// target_round), lib.rs:297-301): at most max_variable_byte_size/64 small inversions.
// Not a compute path: every cell still comes from the GPU (hsw_frame_kernel), which
// reads this table.
#ifndef HSW_FR_HPP
#define HSW_FR_HPP

#include <stdint.h>

namespace hsw {
namespace fr {

typedef unsigned __int128 u128;
struct Fe { uint64_t l[4]; };

static const Fe P = {{0x43e1f593f0000001ULL, 0x2833e84879b97091ULL, 0xb85045b68181585dULL, 0x30644e72e131a029ULL}};

inline bool geq(const Fe &a, const Fe &b) {
    for (int i = 3; i >= 0; i--) {
        if (a.l[i] > b.l[i]) return true;
        if (a.l[i] < b.l[i]) return false;
    }
    return true;
}
inline Fe sub_raw(const Fe &a, const Fe &b) {
    Fe r; u128 borrow = 0;
    for (int i = 0; i < 4; i++) {
        const u128 d = (u128)a.l[i] - b.l[i] - borrow;
        r.l[i] = (uint64_t)d;
        borrow = (d >> 64) & 1;
    }
    return r;
}
inline Fe add(const Fe &a, const Fe &b) {        // a, b < p < 2^254
    Fe r; u128 carry = 0;
    for (int i = 0; i < 4; i++) {
        const u128 s = (u128)a.l[i] + b.l[i] + carry;
        r.l[i] = (uint64_t)s;
        carry = s >> 64;
    }
    return geq(r, P) ? sub_raw(r, P) : r;
}
// -p^-1 mod 2^64 by Newton iteration on the low limb
inline uint64_t neg_inv64() {
    uint64_t x = 1;
    for (int i = 0; i < 6; i++) x *= 2 - P.l[0] * x;
    return (uint64_t)0 - x;
}
// Montgomery product a * b / 2^256 mod p (CIOS)
inline Fe mont_mul(const Fe &a, const Fe &b) {
    static const uint64_t INV = neg_inv64();
    uint64_t t[6] = {0, 0, 0, 0, 0, 0};
    for (int i = 0; i < 4; i++) {
        u128 carry = 0;
        for (int j = 0; j < 4; j++) {
            const u128 s = (u128)a.l[j] * b.l[i] + t[j] + carry;
            t[j] = (uint64_t)s;
            carry = s >> 64;
        }
        u128 s = (u128)t[4] + carry;
        t[4] = (uint64_t)s;
        t[5] = (uint64_t)(s >> 64);
        const uint64_t m = t[0] * INV;
        carry = ((u128)m * P.l[0] + t[0]) >> 64;
        for (int j = 1; j < 4; j++) {
            const u128 s2 = (u128)m * P.l[j] + t[j] + carry;
            t[j - 1] = (uint64_t)s2;
            carry = s2 >> 64;
        }
        s = (u128)t[4] + carry;
        t[3] = (uint64_t)s;
        t[4] = t[5] + (uint64_t)(s >> 64);
    }
    Fe r = {{t[0], t[1], t[2], t[3]}};
    return (t[4] || geq(r, P)) ? sub_raw(r, P) : r;
}
// 2^(256 k) mod p by doubling
inline Fe pow2_256k(int k) {
    Fe r = {{1, 0, 0, 0}};
    for (int i = 0; i < 256 * k; i++) r = add(r, r);
    return r;
}
inline Fe to_mont(const Fe &a) { static const Fe R2 = pow2_256k(2); return mont_mul(a, R2); }
inline Fe from_mont(const Fe &a) { const Fe one = {{1, 0, 0, 0}}; return mont_mul(a, one); }
// (canonical k)^-1 as a Montgomery-form element: k^(p-2)
inline Fe inv_mont(uint64_t k) {
    const Fe km = to_mont(Fe{{k, 0, 0, 0}});
    Fe e = P; e.l[0] -= 2;
    Fe r = to_mont(Fe{{1, 0, 0, 0}}), base = km;
    for (int bit = 0; bit < 254; bit++) {
        if ((e.l[bit / 64] >> (bit % 64)) & 1) r = mont_mul(r, base);
        base = mont_mul(base, base);
    }
    return r;
}

}  // namespace fr
}  // namespace hsw
#endif
