// BN254 scalar field Fr for CDNA4 lanes: one element per lane, 8 x u32 little-endian limbs,
// Montgomery form with R = 2^256. The reference's production field is ffiasm's 4x64 x86-64
// Montgomery Fr (reference tools/helpers/actions.js:207-215, buildZqField(p,"Fr")); on gfx950 a
// 64x64 multiply lowers to v_mad_u64_u32 chains, so the native limb is 32 bits.
// Every function is __host__ __device__ so the host-side batch builder shares the arithmetic.
#pragma once
#include <stdint.h>
#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define HZ_HD __host__ __device__ __forceinline__
// The 256-bit Montgomery product is ~700 instructions: kept out of line so that callers' loops
// stay unrollable and the hot code fits the instruction cache.
#define HZ_HD_NOINLINE __host__ __device__ __attribute__((noinline))
#else
#define HZ_HD inline
#define HZ_HD_NOINLINE inline
#endif

namespace hz {

struct Fr {
    uint32_t v[8];
};

// r = 21888242871839275222246405745257275088548364400416034343698204186575808495617
#define HZ_P0 0xf0000001u
#define HZ_P1 0x43e1f593u
#define HZ_P2 0x79b97091u
#define HZ_P3 0x2833e848u
#define HZ_P4 0x8181585du
#define HZ_P5 0xb85045b6u
#define HZ_P6 0xe131a029u
#define HZ_P7 0x30644e72u
#define HZ_INV32 0xefffffffu  // -r^-1 mod 2^32

HZ_HD constexpr uint32_t fr_p(int i) {
    return i == 0 ? HZ_P0 : i == 1 ? HZ_P1 : i == 2 ? HZ_P2 : i == 3 ? HZ_P3 : i == 4 ? HZ_P4 : i == 5 ? HZ_P5 : i == 6 ? HZ_P6 : HZ_P7;
}
// R mod r (Montgomery one) and R^2 mod r
HZ_HD constexpr uint32_t fr_r1(int i) {
    constexpr uint32_t k[8] = {0x4ffffffbu, 0xac96341cu, 0x9f60cd29u, 0x36fc7695u, 0x7879462eu, 0x666ea36fu, 0x9a07df2fu, 0x0e0a77c1u};
    return k[i];
}
HZ_HD constexpr uint32_t fr_r2(int i) {
    constexpr uint32_t k[8] = {0xae216da7u, 0x1bb8e645u, 0xe35c59e3u, 0x53fe3ab1u, 0x53bb8085u, 0x8c49833du, 0x7f4e44a5u, 0x0216d0b1u};
    return k[i];
}

HZ_HD Fr fr_zero() {
    Fr r;
#pragma unroll
    for (int i = 0; i < 8; i++) r.v[i] = 0;
    return r;
}
HZ_HD Fr fr_one() {  // Montgomery 1
    Fr r;
#pragma unroll
    for (int i = 0; i < 8; i++) r.v[i] = fr_r1(i);
    return r;
}
HZ_HD bool fr_is_zero(const Fr& a) {
    uint32_t o = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) o |= a.v[i];
    return o == 0;
}
HZ_HD bool fr_eq(const Fr& a, const Fr& b) {
    uint32_t o = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) o |= a.v[i] ^ b.v[i];
    return o == 0;
}
// a >= p ? (limb compare, raw integers)
HZ_HD bool fr_geq_p(const uint32_t* a) {
    // compute a - p borrow
    uint64_t br = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        uint64_t d = (uint64_t)a[i] - fr_p(i) - br;
        br = (d >> 63) & 1;
    }
    return br == 0;
}
HZ_HD void fr_cond_sub_p(uint32_t* t) {
    uint32_t s[8];
    uint64_t br = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        uint64_t d = (uint64_t)t[i] - fr_p(i) - br;
        s[i] = (uint32_t)d;
        br = (d >> 63) & 1;
    }
    if (br == 0) {
#pragma unroll
        for (int i = 0; i < 8; i++) t[i] = s[i];
    }
}
HZ_HD Fr fr_add(const Fr& a, const Fr& b) {
    Fr r;
    uint64_t c = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        c += (uint64_t)a.v[i] + b.v[i];
        r.v[i] = (uint32_t)c;
        c >>= 32;
    }
    fr_cond_sub_p(r.v);  // a+b < 2p < 2^256, no carry out
    return r;
}
HZ_HD Fr fr_sub(const Fr& a, const Fr& b) {
    Fr r;
    uint64_t br = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        uint64_t d = (uint64_t)a.v[i] - b.v[i] - br;
        r.v[i] = (uint32_t)d;
        br = (d >> 63) & 1;
    }
    if (br) {
        uint64_t c = 0;
#pragma unroll
        for (int i = 0; i < 8; i++) {
            c += (uint64_t)r.v[i] + fr_p(i);
            r.v[i] = (uint32_t)c;
            c >>= 32;
        }
    }
    return r;
}
HZ_HD Fr fr_neg(const Fr& a) { return fr_sub(fr_zero(), a); }
HZ_HD Fr fr_dbl(const Fr& a) { return fr_add(a, a); }

// Montgomery product a*b/R mod p. CIOS, interleaved, using the "no final carry" shortcut that
// holds because the top limb of p is < 2^31.
// Out of line on the device unless HZ_FR_MUL_INLINE is defined: the product is ~700 instructions and
// the witness kernels call it from hundreds of sites; one shared body keeps them inside the
// instruction cache (and compiles in seconds instead of minutes).
#if defined(__HIP_DEVICE_COMPILE__) && !defined(HZ_FR_MUL_INLINE)
static __device__ __attribute__((noinline)) Fr fr_mul(const Fr a, const Fr b) {
#else
HZ_HD Fr fr_mul(const Fr& a, const Fr& b) {
#endif
    uint32_t t[8];
#pragma unroll
    for (int i = 0; i < 8; i++) t[i] = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        const uint32_t bi = b.v[i];
        uint64_t A = (uint64_t)a.v[0] * bi + t[0];
        const uint32_t t0 = (uint32_t)A;
        A >>= 32;
        const uint32_t m = t0 * HZ_INV32;
        uint64_t C = (uint64_t)m * fr_p(0) + t0;
        C >>= 32;
#pragma unroll
        for (int j = 1; j < 8; j++) {
            A += (uint64_t)a.v[j] * bi + t[j];
            C += (uint64_t)m * fr_p(j) + (uint32_t)A;
            A >>= 32;
            t[j - 1] = (uint32_t)C;
            C >>= 32;
        }
        t[7] = (uint32_t)(C + A);
    }
    fr_cond_sub_p(t);
    Fr r;
#pragma unroll
    for (int i = 0; i < 8; i++) r.v[i] = t[i];
    return r;
}
HZ_HD Fr fr_sqr(const Fr& a) { return fr_mul(a, a); }

// canonical (plain integer, < p) <-> Montgomery
HZ_HD Fr fr_from_canon(const Fr& a) {
    Fr r2;
#pragma unroll
    for (int i = 0; i < 8; i++) r2.v[i] = fr_r2(i);
    return fr_mul(a, r2);
}
#if defined(__HIP_DEVICE_COMPILE__) && !defined(HZ_FR_MUL_INLINE)
static __device__ __attribute__((noinline)) Fr fr_to_canon(const Fr a) {
#else
HZ_HD Fr fr_to_canon(const Fr& a) {
#endif
    // Montgomery reduction of a (multiply by 1): 8 rounds of m*p accumulation only
    uint32_t t[8];
#pragma unroll
    for (int i = 0; i < 8; i++) t[i] = a.v[i];
#pragma unroll
    for (int i = 0; i < 8; i++) {
        const uint32_t m = t[0] * HZ_INV32;
        uint64_t C = (uint64_t)m * fr_p(0) + t[0];
        C >>= 32;
#pragma unroll
        for (int j = 1; j < 8; j++) {
            C += (uint64_t)m * fr_p(j) + t[j];
            t[j - 1] = (uint32_t)C;
            C >>= 32;
        }
        t[7] = (uint32_t)C;
    }
    fr_cond_sub_p(t);
    Fr r;
#pragma unroll
    for (int i = 0; i < 8; i++) r.v[i] = t[i];
    return r;
}
// small integer -> Montgomery
HZ_HD Fr fr_from_u64(uint64_t x) {
    Fr c = fr_zero();
    c.v[0] = (uint32_t)x;
    c.v[1] = (uint32_t)(x >> 32);
    return fr_from_canon(c);
}
HZ_HD Fr fr_from_bit(uint32_t b) {  // 0 or Montgomery 1, branch-free
    Fr r;
#pragma unroll
    for (int i = 0; i < 8; i++) r.v[i] = b ? fr_r1(i) : 0u;
    return r;
}
HZ_HD Fr fr_select(bool c, const Fr& a, const Fr& b) {  // c ? a : b
    Fr r;
#pragma unroll
    for (int i = 0; i < 8; i++) r.v[i] = c ? a.v[i] : b.v[i];
    return r;
}

// a^e for a public 256-bit exponent given as 8 LE limbs (square-and-multiply, MSB first).
HZ_HD Fr fr_pow(const Fr& a, const uint32_t* e) {
    Fr r = fr_one();
    bool started = false;
    for (int i = 255; i >= 0; i--) {
        if (started) r = fr_sqr(r);
        if ((e[i >> 5] >> (i & 31)) & 1u) {
            r = started ? fr_mul(r, a) : a;
            started = true;
        }
    }
    return r;
}
// Inverse by Fermat (a^(p-2)); inverse of 0 is 0, which is the convention the witness needs for
// `x != 0 ? 1/x : 0` (circomlib comparators IsZero) and for division by zero in
// `<--` expressions. The exponent is public, so control flow is wave-uniform.
HZ_HD Fr fr_inv(const Fr& a) {
    const uint32_t e[8] = {HZ_P0 - 2u, HZ_P1, HZ_P2, HZ_P3, HZ_P4, HZ_P5, HZ_P6, HZ_P7};
    Fr r = a;  // top bit (bit 253) of p-2 is set
#pragma unroll 1
    for (int i = 252; i >= 0; i--) {
        r = fr_sqr(r);
        if ((e[i >> 5] >> (i & 31)) & 1u) r = fr_mul(r, a);
    }
    return r;
}

}  // namespace hz
