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
// Fermat inverse a^(p-2) (kept as the cross-check of fr_inv in the self tests).
HZ_HD Fr fr_inv_fermat(const Fr& a) {
    const uint32_t e[8] = {HZ_P0 - 2u, HZ_P1, HZ_P2, HZ_P3, HZ_P4, HZ_P5, HZ_P6, HZ_P7};
    Fr r = a;  // top bit (bit 253) of p-2 is set
#pragma unroll 1
    for (int i = 252; i >= 0; i--) {
        r = fr_sqr(r);
        if ((e[i >> 5] >> (i & 31)) & 1u) r = fr_mul(r, a);
    }
    return r;
}

// ---- modular inverse by constant-time Bernstein-Yang division steps ("safegcd") -----------------
// 20 batches of 30 half-delta divsteps on signed 30-bit limbs (9 limbs). Control flow does not
// depend on the data: all 64 lanes of a wavefront run the same instruction stream, which is what
// makes this the right inversion for SIMT (a Fermat inverse costs ~380 Montgomery products, this
// one about 20 products' worth of instructions). inverse(0) = 0, the convention the witness needs
// for `x != 0 ? 1/x : 0` (circomlib IsZero) and for `<--` divisions by zero.
struct Fr30 {
    int32_t v[9];
};
#define HZ_M30 0x3fffffff
HZ_HD constexpr int32_t fr_p30(int i) {
    constexpr int32_t k[9] = {0x30000001, 0x0f87d64f, 0x1b970914, 0x0cfa121e, 0x01585d28, 0x0116da06, 0x1a029b85, 0x139cb84c, 0x00003064};
    return k[i];
}
#define HZ_P_INV30 0x10000001u  // p^-1 mod 2^30

// 30 divsteps on the low limbs; returns the new zeta and the transition matrix (u,v;q,r)
HZ_HD int32_t fr_divsteps_30(int32_t zeta, uint32_t f0, uint32_t g0, int32_t* t) {
    uint32_t u = 1, v = 0, q = 0, r = 1, f = f0, g = g0;
#pragma unroll 1
    for (int i = 0; i < 30; ++i) {
        uint32_t mask1 = (uint32_t)(zeta >> 31);
        const uint32_t mask2 = 0u - (g & 1u);
        const uint32_t x = (f ^ mask1) - mask1, y = (u ^ mask1) - mask1, z = (v ^ mask1) - mask1;
        g += x & mask2;
        q += y & mask2;
        r += z & mask2;
        mask1 &= mask2;
        zeta = (int32_t)(((uint32_t)zeta ^ mask1) - 1u);
        f += g & mask1;
        u += q & mask1;
        v += r & mask1;
        g >>= 1;
        u <<= 1;
        v <<= 1;
    }
    t[0] = (int32_t)u; t[1] = (int32_t)v; t[2] = (int32_t)q; t[3] = (int32_t)r;
    return zeta;
}
HZ_HD void fr_update_fg_30(Fr30& f, Fr30& g, const int32_t* t) {
    const int64_t u = t[0], v = t[1], q = t[2], r = t[3];
    int64_t cf = u * f.v[0] + v * g.v[0];
    int64_t cg = q * f.v[0] + r * g.v[0];
    cf >>= 30;
    cg >>= 30;
#pragma unroll
    for (int i = 1; i < 9; ++i) {
        const int64_t fi = f.v[i], gi = g.v[i];
        cf += u * fi + v * gi;
        cg += q * fi + r * gi;
        f.v[i - 1] = (int32_t)cf & HZ_M30; cf >>= 30;
        g.v[i - 1] = (int32_t)cg & HZ_M30; cg >>= 30;
    }
    f.v[8] = (int32_t)cf;
    g.v[8] = (int32_t)cg;
}
HZ_HD void fr_update_de_30(Fr30& d, Fr30& e, const int32_t* t) {
    const int32_t u = t[0], v = t[1], q = t[2], r = t[3];
    const int32_t sd = d.v[8] >> 31, se = e.v[8] >> 31;
    int32_t md = (u & sd) + (v & se), me = (q & sd) + (r & se);
    int64_t cd = (int64_t)u * d.v[0] + (int64_t)v * e.v[0];
    int64_t ce = (int64_t)q * d.v[0] + (int64_t)r * e.v[0];
    md -= (int32_t)((HZ_P_INV30 * (uint32_t)cd + (uint32_t)md) & HZ_M30);
    me -= (int32_t)((HZ_P_INV30 * (uint32_t)ce + (uint32_t)me) & HZ_M30);
    cd += (int64_t)fr_p30(0) * md;
    ce += (int64_t)fr_p30(0) * me;
    cd >>= 30;
    ce >>= 30;
#pragma unroll
    for (int i = 1; i < 9; ++i) {
        const int64_t di = d.v[i], ei = e.v[i];
        cd += (int64_t)u * di + (int64_t)v * ei + (int64_t)fr_p30(i) * md;
        ce += (int64_t)q * di + (int64_t)r * ei + (int64_t)fr_p30(i) * me;
        d.v[i - 1] = (int32_t)cd & HZ_M30; cd >>= 30;
        e.v[i - 1] = (int32_t)ce & HZ_M30; ce >>= 30;
    }
    d.v[8] = (int32_t)cd;
    e.v[8] = (int32_t)ce;
}
// plain (non-Montgomery) inverse of the canonical integer x (< p): x^-1 mod p, 0 for x = 0
HZ_HD Fr fr_inv_plain(const Fr& x) {
    Fr30 d, e, f, g;
#pragma unroll
    for (int i = 0; i < 9; i++) { d.v[i] = 0; e.v[i] = 0; f.v[i] = fr_p30(i); }
    e.v[0] = 1;
    // 8 x 32-bit limbs -> 9 x 30-bit limbs
#pragma unroll
    for (int i = 0; i < 9; i++) {
        const int bit = 30 * i, w = bit >> 5, sh = bit & 31;
        uint64_t lo = x.v[w];
        if (w + 1 < 8) lo |= (uint64_t)x.v[w + 1] << 32;
        g.v[i] = (int32_t)((lo >> sh) & HZ_M30);
    }
    int32_t zeta = -1;
#pragma unroll 1
    for (int it = 0; it < 20; ++it) {
        int32_t t[4];
        zeta = fr_divsteps_30(zeta, (uint32_t)f.v[0], (uint32_t)g.v[0], t);
        fr_update_de_30(d, e, t);
        fr_update_fg_30(f, g, t);
    }
    // normalise d: add p if negative, negate if f is negative, add p if negative again
    int32_t rr[9];
    const int32_t cond_add = d.v[8] >> 31, cond_neg = f.v[8] >> 31;
#pragma unroll
    for (int i = 0; i < 9; i++) {
        rr[i] = d.v[i] + (fr_p30(i) & cond_add);
        rr[i] = (rr[i] ^ cond_neg) - cond_neg;
    }
#pragma unroll
    for (int i = 0; i < 8; i++) { rr[i + 1] += rr[i] >> 30; rr[i] &= HZ_M30; }
    const int32_t cond_add2 = rr[8] >> 31;
#pragma unroll
    for (int i = 0; i < 9; i++) rr[i] += fr_p30(i) & cond_add2;
#pragma unroll
    for (int i = 0; i < 8; i++) { rr[i + 1] += rr[i] >> 30; rr[i] &= HZ_M30; }
    // 9 x 30 -> 8 x 32
    Fr o;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        const int bit = 32 * i, l = bit / 30, sh = bit % 30;
        uint64_t acc = (uint64_t)(uint32_t)rr[l] >> sh;
        acc |= (uint64_t)(uint32_t)rr[l + 1] << (30 - sh);
        if (l + 2 < 9) acc |= (uint64_t)(uint32_t)rr[l + 2] << (60 - sh);
        o.v[i] = (uint32_t)acc;
    }
    return o;
}
// Montgomery-domain inverse: (aR)^-1 * R^3 / R = a^-1 R
#if defined(__HIP_DEVICE_COMPILE__) && !defined(HZ_FR_MUL_INLINE)
static __device__ __attribute__((noinline)) Fr fr_inv(const Fr a) {
#else
HZ_HD Fr fr_inv(const Fr& a) {
#endif
    constexpr uint32_t r3[8] = {0xb4bf0040u, 0x5e94d8e1u, 0x1cfbb6b8u, 0x2a489cbeu, 0xa19fcfedu, 0x893cc664u, 0x7fcc657cu, 0x0cf8594bu};
    Fr k;
#pragma unroll
    for (int i = 0; i < 8; i++) k.v[i] = r3[i];
    return fr_mul(fr_inv_plain(a), k);
}

}  // namespace hz
