// BN254 scalar field Fr for CDNA4 lanes -- one element per lane.
//
// The reference's production field is ffiasm's 4 x 64-bit x86-64 Montgomery Fr (reference
// tools/helpers/actions.js:207-215, buildZqField(p,"Fr")). gfx950 has no cheap carry chain for a
// 64x64 multiply; its multiplier primitive is v_mad_u64_u32 (32x32 + 64-bit accumulate). The
// representation here is therefore 9 limbs of 29 bits with lazy carries: every column of the
// schoolbook product and of the Montgomery reduction is a pure v_mad_u64_u32 accumulation chain
// (9 terms of < 2^58 fit a 64-bit accumulator), so a product is 162 multiply-accumulates and a few
// shifts/masks -- measured 3x faster per dependent product than the 8 x 32-bit CIOS form
// (tools/microbench/mulbench.hip, DESIGN.md "Field arithmetic").
//
//   Fr  Montgomery domain, R = 2^261, value in [0, 2p), limbs v[0..7] < 2^29 ("normalised")
//   Fc  canonical integer, 8 x u32 little-endian: the 32-byte element format of the ABI/witness
//
// Every function is __host__ __device__: the host-side batch builder shares the arithmetic.
#pragma once
#include <stdint.h>
#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define HZ_HD __host__ __device__ __forceinline__
#else
#define HZ_HD inline
#endif
// Heavy routines stay out of line on the device unless HZ_FR_INLINE is defined (one shared body, seconds to compile). Every
// kernel on the RollupMain path defines it: measured, the call overhead (operands copied through the argument registers) costs
// more than the larger code -- k_smt, k_hash4, the signature kernels (-8 %) and k_main_front (-11 %). Only the fee / withdraw /
// SHA kernels (fee_kernels.hip: latency or store bound, no difference) and the gadget mains keep the out-of-line form.
#if defined(__HIP_DEVICE_COMPILE__) && !defined(HZ_FR_INLINE)
#define HZ_HD_HEAVY static __device__ __attribute__((noinline))
#define HZ_HEAVY_ARG(T) const T
#else
#define HZ_HD_HEAVY HZ_HD
#define HZ_HEAVY_ARG(T) const T&
#endif

namespace hz {

struct Fr {
    uint32_t v[9];
};
struct Fc {
    uint32_t v[8];
};

#define HZ_M29 0x1fffffffu
#define HZ_INV29 0x0fffffffu  // -p^-1 mod 2^29

// r = 21888242871839275222246405745257275088548364400416034343698204186575808495617
HZ_HD constexpr uint32_t fr_p29(int i) {
    constexpr uint32_t k[9] = {0x10000001u, 0x1f0fac9fu, 0x0e5c2450u, 0x07d090f3u, 0x1585d283u, 0x02db40c0u, 0x00a6e141u, 0x0e5c2634u, 0x0030644eu};
    return k[i];
}
HZ_HD constexpr uint32_t fr_2p29(int i) {
    constexpr uint32_t k[9] = {0x00000002u, 0x1e1f593fu, 0x1cb848a1u, 0x0fa121e6u, 0x0b0ba506u, 0x05b68181u, 0x014dc282u, 0x1cb84c68u, 0x0060c89cu};
    return k[i];
}
HZ_HD constexpr uint32_t fr_4p29(int i) {
    constexpr uint32_t k[9] = {0x00000004u, 0x1c3eb27eu, 0x19709143u, 0x1f4243cdu, 0x16174a0cu, 0x0b6d0302u, 0x029b8504u, 0x197098d0u, 0x00c19139u};
    return k[i];
}
HZ_HD constexpr uint32_t fr_r1(int i) {  // R mod p
    constexpr uint32_t k[9] = {0x0fffff57u, 0x1ea70ab4u, 0x052c068bu, 0x17504f49u, 0x0aa8075bu, 0x1d4240ceu, 0x11d54c07u, 0x052ac7a8u, 0x000dc836u};
    return k[i];
}
HZ_HD constexpr uint32_t fr_r2(int i) {  // R^2 mod p
    constexpr uint32_t k[9] = {0x05b69bd4u, 0x06170a5au, 0x020cddceu, 0x1db6310bu, 0x0e54d0ffu, 0x1cf855e3u, 0x1c15e103u, 0x07d09161u, 0x000a054au};
    return k[i];
}
HZ_HD constexpr uint32_t fr_r3(int i) {  // R^3 mod p
    constexpr uint32_t k[9] = {0x001fddb2u, 0x17d30b63u, 0x1a2600eeu, 0x09507c47u, 0x1496b29bu, 0x0b00a268u, 0x15b645ebu, 0x1f9fcb3du, 0x001baa96u};
    return k[i];
}
// canonical modulus, 8 x u32
HZ_HD constexpr uint32_t fc_p(int i) {
    constexpr uint32_t k[8] = {0xf0000001u, 0x43e1f593u, 0x79b97091u, 0x2833e848u, 0x8181585du, 0xb85045b6u, 0xe131a029u, 0x30644e72u};
    return k[i];
}

// ---- canonical (Fc) helpers ------------------------------------------------------------------------
HZ_HD Fc fc_zero() {
    Fc r;
#pragma unroll
    for (int i = 0; i < 8; i++) r.v[i] = 0;
    return r;
}
// t -= p if t >= p (plain 256-bit integer)
HZ_HD void fc_cond_sub_p(uint32_t* t) {
    uint32_t s[8];
    uint64_t br = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        const uint64_t d = (uint64_t)t[i] - fc_p(i) - br;
        s[i] = (uint32_t)d;
        br = (d >> 63) & 1;
    }
    if (br == 0) {
#pragma unroll
        for (int i = 0; i < 8; i++) t[i] = s[i];
    }
}
HZ_HD bool fc_is_zero(const Fc& a) {
    uint32_t o = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) o |= a.v[i];
    return o == 0;
}

// ---- basic Fr ------------------------------------------------------------------------------------------
HZ_HD Fr fr_zero() {
    Fr r;
#pragma unroll
    for (int i = 0; i < 9; i++) r.v[i] = 0;
    return r;
}
HZ_HD Fr fr_one() {  // Montgomery 1
    Fr r;
#pragma unroll
    for (int i = 0; i < 9; i++) r.v[i] = fr_r1(i);
    return r;
}
HZ_HD Fr fr_from_bit(uint32_t b) {  // 0 or Montgomery 1
    Fr r;
#pragma unroll
    for (int i = 0; i < 9; i++) r.v[i] = b ? fr_r1(i) : 0u;
    return r;
}
HZ_HD Fr fr_select(bool c, const Fr& a, const Fr& b) {
    Fr r;
#pragma unroll
    for (int i = 0; i < 9; i++) r.v[i] = c ? a.v[i] : b.v[i];
    return r;
}
// propagate carries: limbs 0..7 back below 2^29
HZ_HD void fr_norm(uint32_t* t) {
#pragma unroll
    for (int i = 0; i < 8; i++) {
        t[i + 1] += t[i] >> 29;
        t[i] &= HZ_M29;
    }
}
// t (normalised, value < 4p) -> t - 2p if t >= 2p
HZ_HD void fr_cond_sub_2p(uint32_t* t) {
    int32_t d[9];
    int32_t c = 0;
#pragma unroll
    for (int i = 0; i < 9; i++) {
        const int32_t x = (int32_t)t[i] - (int32_t)fr_2p29(i) + c;
        d[i] = (i < 8) ? (x & (int32_t)HZ_M29) : x;
        c = x >> 29;
    }
    const bool ge = d[8] >= 0;   // selects, not a branch: no exec-masked stores
#pragma unroll
    for (int i = 0; i < 9; i++) t[i] = ge ? (uint32_t)d[i] : t[i];
}
// a (normalised, value < 8p) -> a - 4p if a >= 4p; branch-free (selects), by value
HZ_HD Fr fr_cond_sub_4p(const Fr& a) {
    int32_t d[9];
    int32_t c = 0;
#pragma unroll
    for (int i = 0; i < 9; i++) {
        const int32_t x = (int32_t)a.v[i] - (int32_t)fr_4p29(i) + c;
        d[i] = (i < 8) ? (x & (int32_t)HZ_M29) : x;
        c = x >> 29;
    }
    const bool ge = d[8] >= 0;
    Fr r;
#pragma unroll
    for (int i = 0; i < 9; i++) r.v[i] = ge ? (uint32_t)d[i] : a.v[i];
    return r;
}
HZ_HD Fr fr_add(const Fr& a, const Fr& b) {
    Fr r;
#pragma unroll
    for (int i = 0; i < 9; i++) r.v[i] = a.v[i] + b.v[i];
    fr_norm(r.v);
    fr_cond_sub_2p(r.v);
    return r;
}
HZ_HD Fr fr_sub(const Fr& a, const Fr& b) {
    // a - b + 2p in [0, 4p), signed limb arithmetic with arithmetic carries
    Fr r;
    int32_t c = 0;
#pragma unroll
    for (int i = 0; i < 9; i++) {
        const int32_t x = (int32_t)a.v[i] - (int32_t)b.v[i] + (int32_t)fr_2p29(i) + c;
        r.v[i] = (i < 8) ? (uint32_t)(x & (int32_t)HZ_M29) : (uint32_t)x;
        c = x >> 29;
    }
    fr_cond_sub_2p(r.v);
    return r;
}
// ---- lazily reduced sums (the signature ladder, eddsa_kernels.hip). A product accepts operands below 2^257 (> 10 p), so a difference
// that only feeds products needs no conditional subtraction, and a chain of differences that is stored needs one reduction, not
// one per link. Every routine states the range of its operands and of its result; limbs stay normalised.
// a - b + 2p in (0, A + 2p) for a in [0, A), b in [0, 2p): not reduced
HZ_HD Fr fr_sub_lazy(const Fr& a, const Fr& b) {
    Fr r;
    int32_t c = 0;
#pragma unroll
    for (int i = 0; i < 9; i++) {
        const int32_t x = (int32_t)a.v[i] - (int32_t)b.v[i] + (int32_t)fr_2p29(i) + c;
        r.v[i] = (i < 8) ? (uint32_t)(x & (int32_t)HZ_M29) : (uint32_t)x;
        c = x >> 29;
    }
    return r;
}
// 2a in [0, 2A) for a in [0, A): not reduced (for a below p this IS fr_dbl(a))
HZ_HD Fr fr_dbl_lazy(const Fr& a) {
    Fr r;
#pragma unroll
    for (int i = 0; i < 9; i++) r.v[i] = a.v[i] << 1;
    fr_norm(r.v);
    return r;
}
// a + b, not reduced: below A + B
HZ_HD Fr fr_add_lazy(const Fr& a, const Fr& b) {
    Fr r;
#pragma unroll
    for (int i = 0; i < 9; i++) r.v[i] = a.v[i] + b.v[i];
    fr_norm(r.v);
    return r;
}
// a - b - c + 4p in (0, A + 4p) for a in [0, A), b + c < 4p: not reduced
HZ_HD Fr fr_sub2_lazy(const Fr& a, const Fr& b, const Fr& c2) {
    Fr r;
    int32_t c = 0;
#pragma unroll
    for (int i = 0; i < 9; i++) {
        const int32_t x = (int32_t)a.v[i] - (int32_t)b.v[i] - (int32_t)c2.v[i] + (int32_t)fr_4p29(i) + c;
        r.v[i] = (i < 8) ? (uint32_t)(x & (int32_t)HZ_M29) : (uint32_t)x;
        c = x >> 29;
    }
    return r;
}
// m - a - b - c reduced to [0, 2p); m in [0, 2p), a, b, c in [0, 2p) with a + b + c < 4p
HZ_HD Fr fr_sub3(const Fr& m, const Fr& a, const Fr& b, const Fr& c3) {
    Fr t;
    int32_t c = 0;
#pragma unroll
    for (int i = 0; i < 9; i++) {   // m - a - b - c + 4p in (0, 6p): per limb within [-3 * 2^29, 2 * 2^29]
        const int32_t x = (int32_t)m.v[i] - (int32_t)a.v[i] - (int32_t)b.v[i] - (int32_t)c3.v[i] + (int32_t)fr_4p29(i) + c;
        t.v[i] = (i < 8) ? (uint32_t)(x & (int32_t)HZ_M29) : (uint32_t)x;
        c = x >> 29;
    }
    Fr r = fr_cond_sub_4p(t);   // [0, 4p)
    fr_cond_sub_2p(r.v);
    return r;
}
// s - a - 2x reduced to [0, 2p); s, a, x in [0, 2p)
HZ_HD Fr fr_sub_a_2x(const Fr& s, const Fr& a, const Fr& x2) {
    Fr t;
    int32_t c = 0;
#pragma unroll
    for (int i = 0; i < 9; i++) {   // s - a - 2x + 6p in (0, 8p): per limb within [-3 * 2^29, 3 * 2^29]
        const int32_t x = (int32_t)s.v[i] - (int32_t)a.v[i] - 2 * (int32_t)x2.v[i] + (int32_t)fr_2p29(i) + (int32_t)fr_4p29(i) + c;
        t.v[i] = (i < 8) ? (uint32_t)(x & (int32_t)HZ_M29) : (uint32_t)x;
        c = x >> 29;
    }
    Fr r = fr_cond_sub_4p(t);   // t < 8p -> [0, 4p)
    fr_cond_sub_2p(r.v);
    return r;
}
// 3a + b + c, not reduced: below 3A + B + C for a in [0, A), b in [0, B), c in [0, C) (limbs: 3 * 2^29 + 2^29 + 2^29 fit 32 bits)
HZ_HD Fr fr_3a_b_c_lazy(const Fr& a, const Fr& b, const Fr& c) {
    Fr r;
#pragma unroll
    for (int i = 0; i < 9; i++) r.v[i] = 3u * a.v[i] + b.v[i] + c.v[i];
    fr_norm(r.v);
    return r;
}
HZ_HD Fr fr_neg(const Fr& a) { return fr_sub(fr_zero(), a); }
HZ_HD Fr fr_dbl(const Fr& a) { return fr_add(a, a); }
// value in {0, p} (both represent zero; limbs are normalised, so each integer has one encoding)
HZ_HD bool fr_is_zero(const Fr& a) {
    uint32_t z = 0, e = 0;
#pragma unroll
    for (int i = 0; i < 9; i++) {
        z |= a.v[i];
        e |= a.v[i] ^ fr_p29(i);
    }
    return z == 0 || e == 0;
}
HZ_HD bool fr_eq(const Fr& a, const Fr& b) { return fr_is_zero(fr_sub(a, b)); }

// shared tail of the product routines: Montgomery-reduce the column sums t[0..17] (t[17]: top limb of an addend, or 0).
// Column by column with ONE running accumulator: the carry out of column k heads column k+1's multiply-accumulate chain, so
// there are no separate carry additions and no 18 live column sums (measured, tools/microbench/mulbench.hip variant 4: -12 %
// latency of a dependent product, +1 % throughput against reducing row by row). The empty asm keeps the compiler from
// re-associating the carry back into independent partial sums joined by 64-bit additions.
HZ_HD Fr fr_reduce_cols(uint64_t* t) {
    uint32_t m[9];
    Fr r;
    uint64_t acc = 0;
#pragma unroll
    for (int k = 0; k < 9; k++) {
        acc += t[k];
#pragma unroll
        for (int i = 0; i < k; i++) acc += (uint64_t)m[i] * fr_p29(k - i);
        m[k] = ((uint32_t)acc * HZ_INV29) & HZ_M29;
        acc += (uint64_t)m[k] * fr_p29(0);
        acc >>= 29;
#if defined(__HIP_DEVICE_COMPILE__)
        asm("" : "+v"(acc));
#endif
    }
#pragma unroll
    for (int k = 9; k < 17; k++) {
        acc += t[k];
#pragma unroll
        for (int i = k - 8; i < 9; i++) acc += (uint64_t)m[i] * fr_p29(k - i);
        r.v[k - 9] = (uint32_t)acc & HZ_M29;
        acc >>= 29;
#if defined(__HIP_DEVICE_COMPILE__)
        asm("" : "+v"(acc));
#endif
    }
    r.v[8] = (uint32_t)(acc + t[17]);
    return r;
}
// Montgomery product a*b/R mod p; inputs normalised with value < 2^257, output < 1.03 p, normalised.
HZ_HD_HEAVY Fr fr_mul(HZ_HEAVY_ARG(Fr) a, HZ_HEAVY_ARG(Fr) b) {
    uint64_t t[18];
#pragma unroll
    for (int k = 0; k < 17; k++) {
        uint64_t acc = 0;
#pragma unroll
        for (int i = 0; i < 9; i++) {
            const int j = k - i;
            if (j < 0 || j > 8) continue;
            acc += (uint64_t)a.v[i] * b.v[j];
        }
        t[k] = acc;
    }
    t[17] = 0;
    return fr_reduce_cols(t);
}
// a^2 / R: 36 doubled cross products + 9 squares instead of 81 products
HZ_HD_HEAVY Fr fr_sqr(HZ_HEAVY_ARG(Fr) a) {
    uint32_t d[9];
#pragma unroll
    for (int i = 0; i < 9; i++) d[i] = a.v[i] << 1;   // < 2^30
    uint64_t t[18];
#pragma unroll
    for (int k = 0; k < 17; k++) {
        uint64_t acc = 0;
#pragma unroll
        for (int i = 0; i < 9; i++) {
            const int j = k - i;
            if (j < 0 || j > 8 || i > j) continue;
            acc += (i == j) ? (uint64_t)a.v[i] * a.v[i] : (uint64_t)d[i] * a.v[j];
        }
        t[k] = acc;
    }
    t[17] = 0;
    return fr_reduce_cols(t);
}
// (sum_{n<N} a[n]*b[n] + addend) / R with ONE Montgomery reduction (N <= 6: 9N+9 terms of < 2^58 and
// one 29-bit limb fit 64 bits). `addend` is a value < p in 29-bit limbs, i.e. the result gains addend/R.
// The MDS mix of Poseidon is t such dot products per round.
template <int N>
HZ_HD Fr fr_dot(const Fr* a, const Fr* b, const Fr* addend = nullptr) {
    static_assert(N >= 1 && N <= 6, "fr_dot: at most 6 products per reduction");
    uint64_t t[18];
#pragma unroll
    for (int k = 0; k < 17; k++) {
        uint64_t acc = (addend && k < 9) ? addend->v[k] : 0;
#pragma unroll
        for (int n = 0; n < N; n++) {
#pragma unroll
            for (int i = 0; i < 9; i++) {
                const int j = k - i;
                if (j < 0 || j > 8) continue;
                acc += (uint64_t)a[n].v[i] * b[n].v[j];
            }
        }
        t[k] = acc;
    }
    t[17] = 0;
    return fr_reduce_cols(t);
}
// (a*b + s*R) / R = a*b/R + s with one reduction: s enters the upper nine columns. a < p, b and s
// normalised with b < 2^257, s < 8p; the result is normalised and < s + 1.01 p.
HZ_HD Fr fr_muladd(const Fr& a, const Fr& b, const Fr& s) {
    uint64_t t[18];
#pragma unroll
    for (int k = 0; k < 18; k++) {
        uint64_t acc = k >= 9 ? s.v[k - 9] : 0;
#pragma unroll
        for (int i = 0; i < 9; i++) {
            const int j = k - i;
            if (j < 0 || j > 8) continue;
            acc += (uint64_t)a.v[i] * b.v[j];
        }
        t[k] = acc;
    }
    return fr_reduce_cols(t);
}

// (a0*b0 + a1*b1 + s*R) / R with one reduction; a0, a1 < p, b0, b1 normalised < 2^257, s < 8p.
HZ_HD Fr fr_muladd2(const Fr& a0, const Fr& b0, const Fr& a1, const Fr& b1, const Fr& s) {
    uint64_t t[18];
#pragma unroll
    for (int k = 0; k < 18; k++) {
        uint64_t acc = k >= 9 ? s.v[k - 9] : 0;
#pragma unroll
        for (int i = 0; i < 9; i++) {
            const int j = k - i;
            if (j < 0 || j > 8) continue;
            acc += (uint64_t)a0.v[i] * b0.v[j];
            acc += (uint64_t)a1.v[i] * b1.v[j];
        }
        t[k] = acc;
    }
    return fr_reduce_cols(t);
}

// canonical -> Montgomery (also accepts any 256-bit integer)
HZ_HD Fr fr_unpack(const Fc& c) {
    Fr x;
#pragma unroll
    for (int i = 0; i < 9; i++) {
        const int bit = 29 * i, w = bit >> 5, sh = bit & 31;
        uint64_t lo = c.v[w];
        if (w + 1 < 8) lo |= (uint64_t)c.v[w + 1] << 32;
        x.v[i] = (uint32_t)(lo >> sh) & HZ_M29;
    }
    return x;
}
HZ_HD Fr fr_from_canon(const Fc& c) {
    Fr r2;
#pragma unroll
    for (int i = 0; i < 9; i++) r2.v[i] = fr_r2(i);
    return fr_mul(fr_unpack(c), r2);
}
// Montgomery -> canonical (< p)
HZ_HD_HEAVY Fc fr_to_canon(HZ_HEAVY_ARG(Fr) a) {
    uint64_t t[18];
#pragma unroll
    for (int i = 0; i < 9; i++) t[i] = a.v[i];
#pragma unroll
    for (int i = 9; i < 18; i++) t[i] = 0;
    const Fr rr = fr_reduce_cols(t);
    uint32_t r[9];
#pragma unroll
    for (int i = 0; i < 9; i++) r[i] = rr.v[i];
    // (a + m p)/R <= p: the only non-canonical outcome is exactly p
    uint32_t e = 0;
#pragma unroll
    for (int i = 0; i < 9; i++) e |= r[i] ^ fr_p29(i);
    Fc o;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        const int bit = 32 * i, l = bit / 29, sh = bit % 29;
        uint64_t acc = (uint64_t)r[l] >> sh;
        acc |= (uint64_t)r[l + 1] << (29 - sh);
        if (l + 2 < 9) acc |= (uint64_t)r[l + 2] << (58 - sh);
        o.v[i] = e ? (uint32_t)acc : 0u;
    }
    return o;
}
// ---- canonical values in 29-bit limbs (the witness form of the Poseidon S-box, poseidon.h) ----------------
// t (normalised limbs, value < 2p) -> t - p if t >= p: the unique representative in [0, p)
HZ_HD Fr fr_cond_sub_p(const Fr& a) {
    int32_t d[9];
    int32_t c = 0;
#pragma unroll
    for (int i = 0; i < 9; i++) {
        const int32_t x = (int32_t)a.v[i] - (int32_t)fr_p29(i) + c;
        d[i] = (i < 8) ? (x & (int32_t)HZ_M29) : x;
        c = x >> 29;
    }
    const bool ge = d[8] >= 0;
    Fr r;
#pragma unroll
    for (int i = 0; i < 9; i++) r.v[i] = ge ? (uint32_t)d[i] : a.v[i];
    return r;
}
// The same for values that are almost always below p already -- a Montgomery product of reduced operands is < p (1 + p/R), i.e.
// >= p for one lane in ~256, and a bare reduction yields exactly p or less: when no lane of the wavefront has a top limb that
// reaches p's, nobody can be >= p and the 36-instruction subtract-and-select is skipped (wave-uniform branch; three of these per
// Poseidon S-box were 12 % of its instructions).
HZ_HD Fr fr_cond_sub_p_rare(const Fr& a) {
#if defined(__HIP_DEVICE_COMPILE__)
    if (!__any(a.v[8] >= fr_p29(8))) return a;
#endif
    return fr_cond_sub_p(a);
}
// Montgomery -> canonical, kept in 29-bit limbs (a usable multiplicand: x * canon(y) / R = xy/R^... see poseidon_sbox)
HZ_HD Fr fr_canon_limbs(const Fr& a) {
    uint64_t t[18];
#pragma unroll
    for (int i = 0; i < 9; i++) t[i] = a.v[i];
#pragma unroll
    for (int i = 9; i < 18; i++) t[i] = 0;
    return fr_cond_sub_p_rare(fr_reduce_cols(t));   // (a + m p)/R <= p
}
// canonical value in 29-bit limbs (< p) -> 8 x u32
HZ_HD Fc fr_pack_canon(const Fr& r) {
    Fc o;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        const int bit = 32 * i, l = bit / 29, sh = bit % 29;
        uint64_t acc = (uint64_t)r.v[l] >> sh;
        acc |= (uint64_t)r.v[l + 1] << (29 - sh);
        if (l + 2 < 9) acc |= (uint64_t)r.v[l + 2] << (58 - sh);
        o.v[i] = (uint32_t)acc;
    }
    return o;
}
// small integer -> Montgomery
HZ_HD Fr fr_from_u64(uint64_t x) {
    Fc c = fc_zero();
    c.v[0] = (uint32_t)x;
    c.v[1] = (uint32_t)(x >> 32);
    return fr_from_canon(c);
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
    uint32_t e[8];
#pragma unroll
    for (int i = 0; i < 8; i++) e[i] = fc_p(i);
    e[0] -= 2u;
    return fr_pow(a, e);
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

// 30 division steps (plain delta = 1 form, eta = -delta), several per iteration: the zeros at the bottom of g are shifted
// out together (count-trailing-zeros up to the steps that are left), and when g is odd the multiple w of f that cancels its
// bottom min(eta + 1, steps left, 8) bits is added at once -- no sign flip of eta can happen inside those steps -- with
// w = -g / f mod 2^8 from two Newton steps on f. About 10 iterations of ~35 instructions instead of 30 x 23; the lanes of a
// wavefront finish at different iterations and wait for the slowest (SIMT loop). The transition matrix is the product of the same
// division steps, so d, e, f, g come out as in the one-step-at-a-time form of this delta rule.
HZ_HD int32_t fr_divsteps_30_var(int32_t eta, uint32_t f0, uint32_t g0, int32_t* t) {
    uint32_t u = 1, v = 0, q = 0, r = 1, f = f0, g = g0;
    int i = 30;
    for (;;) {
        const uint32_t sentinel = g | (0xffffffffu << i);   // i in 1..30
#if defined(__HIP_DEVICE_COMPILE__)
        const int zeros = __ffs((int)sentinel) - 1;
#else
        const int zeros = __builtin_ctz(sentinel);
#endif
        g >>= zeros; u <<= zeros; v <<= zeros;
        eta -= zeros;
        i -= zeros;
        if (i == 0) break;
        if (eta < 0) {
            eta = -eta;
            uint32_t tmp = f; f = g; g = 0u - tmp;
            tmp = u; u = q; q = 0u - tmp;
            tmp = v; v = r; r = 0u - tmp;
        }
        int limit = eta + 1 > i ? i : eta + 1;
        if (limit > 8) limit = 8;
        const uint32_t m = 0xffffffffu >> (32 - limit);
        uint32_t finv = f;                  // f * f = 1 mod 8: 3 bits
        finv *= 2u - f * finv;              // 6 bits
        finv *= 2u - f * finv;              // 12 bits
        const uint32_t w = (0u - g * finv) & m;
        g += f * w; q += u * w; r += v * w;
    }
    t[0] = (int32_t)u; t[1] = (int32_t)v; t[2] = (int32_t)q; t[3] = (int32_t)r;
    return eta;
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
// value of a (< 2p, normalised) reduced to [0, p) and re-cut into 30-bit limbs
HZ_HD void fr_to_limbs30(const Fr& a, Fr30& g) {
    int32_t d[9];
    int32_t c = 0;
#pragma unroll
    for (int i = 0; i < 9; i++) {
        const int32_t x = (int32_t)a.v[i] - (int32_t)fr_p29(i) + c;
        d[i] = (i < 8) ? (x & (int32_t)HZ_M29) : x;
        c = x >> 29;
    }
    uint32_t w[9];
    const bool ge = d[8] >= 0;
#pragma unroll
    for (int i = 0; i < 9; i++) w[i] = ge ? (uint32_t)d[i] : a.v[i];
#pragma unroll
    for (int i = 0; i < 9; i++) {
        const int bit = 30 * i, l = bit / 29, sh = bit % 29;
        uint64_t acc = 0;
        if (l < 9) acc = (uint64_t)w[l] >> sh;
        if (l + 1 < 9) acc |= (uint64_t)w[l + 1] << (29 - sh);
        if (l + 2 < 9) acc |= (uint64_t)w[l + 2] << (58 - sh);
        g.v[i] = (int32_t)((uint32_t)acc & HZ_M30);
    }
}
HZ_HD Fr fr_from_limbs30(const int32_t* rr) {  // rr: 9 non-negative 30-bit limbs, value < p
    Fr o;
#pragma unroll
    for (int i = 0; i < 9; i++) {
        const int bit = 29 * i, l = bit / 30, sh = bit % 30;
        uint64_t acc = (uint64_t)(uint32_t)rr[l] >> sh;
        if (l + 1 < 9) acc |= (uint64_t)(uint32_t)rr[l + 1] << (30 - sh);
        o.v[i] = (uint32_t)acc & HZ_M29;
    }
    return o;
}
// Montgomery-domain inverse: (aR)^-1 * R^3 / R = a^-1 R
HZ_HD_HEAVY Fr fr_inv(HZ_HEAVY_ARG(Fr) a) {
    Fr30 d, e, f, g;
#pragma unroll
    for (int i = 0; i < 9; i++) { d.v[i] = 0; e.v[i] = 0; f.v[i] = fr_p30(i); }
    e.v[0] = 1;
    fr_to_limbs30(a, g);
    int32_t zeta = -1;
    // batches: the plain delta = 1 rule needs at most 735 division steps (25 batches); the loop leaves as soon as every lane of the
    // wavefront has g = 0 (random operands: 17..19 batches). (The one-step-at-a-time half-delta form of round 1 -- 20 batches of
    // 30 x 23 instructions -- gave the same inverses on 20 000 operands and 1.89 instead of 2.25 G inversions/s.)
    constexpr int kBatches = 25;
#pragma unroll 1
    for (int it = 0; it < kBatches; ++it) {
        int32_t t[4];
        zeta = fr_divsteps_30_var(zeta, (uint32_t)f.v[0], (uint32_t)g.v[0], t);
        fr_update_de_30(d, e, t);
        fr_update_fg_30(f, g, t);
        // g = 0: f = +-1 and d = +-1/a already (the invariants d*a = f, e*a = g hold after every batch); 600 division steps are
        // the proven bound, random operands need 495..530 of them. Left when every lane of the wavefront is done (18 batches).
        if (it >= 15) {
            uint32_t nz = 0;
#pragma unroll
            for (int i = 0; i < 9; i++) nz |= (uint32_t)g.v[i];
#if defined(__HIP_DEVICE_COMPILE__)
            if (__all(nz == 0)) break;
#else
            if (nz == 0) break;
#endif
        }
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
    Fr k;
#pragma unroll
    for (int i = 0; i < 9; i++) k.v[i] = fr_r3(i);
    return fr_mul(fr_from_limbs30(rr), k);
}

}  // namespace hz
