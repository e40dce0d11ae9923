// hsw_mont.hpp -- BN254 Fr in Montgomery form (x * 2^256 mod p, halo2curves' in-memory Fr) on 32-bit limbs:
// what the emit-time conversion of hsw_expand.hpp needs beyond the u64 -> Montgomery multiply it already had.
//
// The witness values of the path are small integers (<= 64 bits) and every cell of the stream is
// either a NEW value, a COPY of an earlier one, or a CONSTANT (a spread() call emits 20 cells from 6 values and
// 4 constants).  With lane = unit each of the three costs what it should:
//   constant   mont_k<K>()   the limbs are compile-time literals (constexpr doubling below), no arithmetic
//   copy       the eight limbs are already in registers
//   sum        fe_add        one 256-bit add + conditional subtract (24 VALU) instead of a fresh conversion
//   new value  mont_from_u64 one 32x256 / 64x256 multiply + Barrett step (~90 / ~150 VALU), once per value
#ifndef HSW_MONT_HPP
#define HSW_MONT_HPP
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace hsw {

struct Fe8 { uint32_t l[8]; };

// p, little-endian 32-bit limbs
#define HSW_P0 0xf0000001u
#define HSW_P1 0x43e1f593u
#define HSW_P2 0x79b97091u
#define HSW_P3 0x2833e848u
#define HSW_P4 0x8181585du
#define HSW_P5 0xb85045b6u
#define HSW_P6 0xe131a029u
#define HSW_P7 0x30644e72u

// k * 2^256 mod p for a compile-time k < 2^64: 256 modular doublings, evaluated by the compiler.
constexpr Fe8 mont_const(uint64_t k) {
    const uint32_t P[8] = {HSW_P0, HSW_P1, HSW_P2, HSW_P3, HSW_P4, HSW_P5, HSW_P6, HSW_P7};
    uint32_t x[8] = {(uint32_t)k, (uint32_t)(k >> 32), 0, 0, 0, 0, 0, 0};
    for (int i = 0; i < 256; i++) {
        uint32_t carry = 0;                                      // x <<= 1 (x < p < 2^254: no bit is lost)
        for (int j = 0; j < 8; j++) {
            const uint32_t nc = x[j] >> 31;
            x[j] = (x[j] << 1) | carry;
            carry = nc;
        }
        bool ge = true;                                          // x >= p ?
        for (int j = 7; j >= 0; j--) {
            if (x[j] != P[j]) { ge = x[j] > P[j]; break; }
        }
        if (ge) {
            uint64_t borrow = 0;
            for (int j = 0; j < 8; j++) {
                const uint64_t d = (uint64_t)x[j] - P[j] - borrow;
                x[j] = (uint32_t)d;
                borrow = (d >> 32) & 1u;
            }
        }
    }
    return Fe8{{x[0], x[1], x[2], x[3], x[4], x[5], x[6], x[7]}};
}
template <uint64_t K>
struct MontK { static constexpr Fe8 value = mont_const(K); };
template <uint64_t K>
__device__ __forceinline__ Fe8 mont_k() {
    constexpr Fe8 c = MontK<K>::value;
    return Fe8{{c.l[0], c.l[1], c.l[2], c.l[3], c.l[4], c.l[5], c.l[6], c.l[7]}};
}
static_assert(mont_const(1).l[0] == 0x4ffffffbu && mont_const(1).l[7] == 0x0e0a77c1u, "R = 2^256 mod p");
static_assert(mont_const(1ull << 32).l[0] == 0x15b8b9dau && mont_const(1ull << 32).l[7] == 0x06bc037eu, "2^288 mod p");

// a + b mod p for a, b < p
__device__ __forceinline__ Fe8 fe_add(const Fe8 &a, const Fe8 &b) {
    const uint32_t P[8] = {HSW_P0, HSW_P1, HSW_P2, HSW_P3, HSW_P4, HSW_P5, HSW_P6, HSW_P7};
    uint32_t s[8], d[8];
    uint32_t cy = 0;
#pragma unroll
    for (int j = 0; j < 8; j++) s[j] = __builtin_addc(a.l[j], b.l[j], cy, &cy);   // < 2p < 2^255: no carry out
    uint32_t br = 0;
#pragma unroll
    for (int j = 0; j < 8; j++) d[j] = __builtin_subc(s[j], P[j], br, &br);
    Fe8 r;
#pragma unroll
    for (int j = 0; j < 8; j++) r.l[j] = br ? s[j] : d[j];
    return r;
}
// p - m for a non-zero m < p
__device__ __forceinline__ Fe8 fe_neg_nonzero(const Fe8 &m) {
    const uint32_t P[8] = {HSW_P0, HSW_P1, HSW_P2, HSW_P3, HSW_P4, HSW_P5, HSW_P6, HSW_P7};
    Fe8 o;
    uint32_t br = 0;
#pragma unroll
    for (int j = 0; j < 8; j++) o.l[j] = __builtin_subc(P[j], m.l[j], br, &br);
    return o;
}

}  // namespace hsw
#endif
