// Micro-benchmark of Montgomery-product formulations on gfx950 (build: hipcc -O3 --offload-arch=gfx950 mulbench.hip -o mulbench)
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <vector>
#define HZ_FR_INLINE 1
#include "../../circuits_amd/csrc/fr.h"
using namespace hz;
#define P(i) fc_p(i)
#define INV 0xefffffffu

__device__ __forceinline__ void mac(uint64_t& acc, uint32_t& hi, uint32_t a, uint32_t b) {
    asm volatile("v_mad_u64_u32 %0, vcc, %2, %3, %0\n\tv_addc_co_u32 %1, vcc, 0, %1, vcc" : "+v"(acc), "+v"(hi) : "v"(a), "v"(b) : "vcc");
}
__device__ __forceinline__ void mac_s(uint64_t& acc, uint32_t& hi, uint32_t a, uint32_t b_const) {
    asm volatile("v_mad_u64_u32 %0, vcc, %2, %3, %0\n\tv_addc_co_u32 %1, vcc, 0, %1, vcc" : "+v"(acc), "+v"(hi) : "v"(a), "s"(b_const) : "vcc");
}
// V2: product scanning with carry-out of v_mad_u64_u32
__device__ __forceinline__ Fc mul_v2(const Fc& a, const Fc& b) {
    uint32_t t[16];
    uint64_t acc = 0; uint32_t hi = 0;
#pragma unroll
    for (int k = 0; k < 15; k++) {
#pragma unroll
        for (int i = 0; i < 8; i++) { const int j = k - i; if (j < 0 || j > 7) continue; mac(acc, hi, a.v[i], b.v[j]); }
        t[k] = (uint32_t)acc; acc = (acc >> 32) | ((uint64_t)hi << 32); hi = 0;
    }
    t[15] = (uint32_t)acc;
    uint32_t m[8], r[8];
    acc = 0; hi = 0;
#pragma unroll
    for (int k = 0; k < 16; k++) {
        asm volatile("v_add_co_u32 %0, vcc, %2, %0\n\tv_addc_co_u32 %1, vcc, 0, %1, vcc" : "+v"(*(uint32_t*)&acc), "+v"(*((uint32_t*)&acc + 1)) : "v"(t[k]) : "vcc");
#pragma unroll
        for (int i = 0; i < 8; i++) { const int j = k - i; if (j < 0 || j > 7) continue; if (k < 8 && i >= k) continue; mac_s(acc, hi, m[i], P(j)); }
        if (k < 8) { m[k] = (uint32_t)acc * INV; mac_s(acc, hi, m[k], P(0)); } else { r[k - 8] = (uint32_t)acc; }
        acc = (acc >> 32) | ((uint64_t)hi << 32); hi = 0;
    }
    fc_cond_sub_p(r);
    Fc o;
#pragma unroll
    for (int i = 0; i < 8; i++) o.v[i] = r[i];
    return o;
}
// V3: 9 x 29-bit limbs, lazy carries: pure v_mad_u64_u32 accumulation chains
struct F29 { uint32_t v[9]; };
#define M29 0x1fffffffu
__device__ __forceinline__ constexpr uint32_t p29(int i) {
    constexpr uint32_t k[9] = {0x10000001, 0x1f0fac9f, 0x0e5c2450, 0x07d090f3, 0x1585d283, 0x02db40c0, 0x00a6e141, 0x0e5c2634, 0x0030644e};
    return k[i];
}
#define INV29 0x0fffffffu
__device__ __forceinline__ F29 mul_v3(const F29& a, const F29& b) {
    uint64_t t[18];
#pragma unroll
    for (int k = 0; k < 17; k++) {
        uint64_t acc = 0;
#pragma unroll
        for (int i = 0; i < 9; i++) { const int j = k - i; if (j < 0 || j > 8) continue; acc += (uint64_t)a.v[i] * b.v[j]; }
        t[k] = acc;
    }
    t[17] = 0;
#pragma unroll
    for (int i = 0; i < 9; i++) {
        const uint32_t m = ((uint32_t)t[i] * INV29) & M29;
#pragma unroll
        for (int j = 0; j < 9; j++) t[i + j] += (uint64_t)m * p29(j);
        t[i + 1] += t[i] >> 29;
    }
    F29 r;
#pragma unroll
    for (int k = 0; k < 9; k++) { r.v[k] = (uint32_t)t[9 + k] & M29; if (k < 8) t[10 + k] += t[9 + k] >> 29; else r.v[8] = (uint32_t)t[17]; }
    return r;
}

// V4: the same limbs, product and reduction fused column by column (one running accumulator whose carry feeds the next
// column's multiply-accumulate chain: no separate carry additions, no 18 live column sums)
__device__ __forceinline__ F29 mul_v4(const F29& a, const F29& b) {
    uint32_t m[9];
    F29 r;
    uint64_t acc = 0;
#pragma unroll
    for (int k = 0; k < 9; k++) {
#pragma unroll
        for (int i = 0; i <= k; i++) acc += (uint64_t)a.v[i] * b.v[k - i];
#pragma unroll
        for (int i = 0; i < k; i++) acc += (uint64_t)m[i] * p29(k - i);
        m[k] = ((uint32_t)acc * INV29) & M29;
        acc += (uint64_t)m[k] * p29(0);
        acc >>= 29;
        asm("" : "+v"(acc));   // keep the carry as the head of the next column's accumulation chain
    }
#pragma unroll
    for (int k = 9; k < 17; k++) {
#pragma unroll
        for (int i = k - 8; i < 9; i++) acc += (uint64_t)a.v[i] * b.v[k - i];
#pragma unroll
        for (int i = k - 8; i < 9; i++) acc += (uint64_t)m[i] * p29(k - i);
        r.v[k - 9] = (uint32_t)acc & M29;
        acc >>= 29;
        asm("" : "+v"(acc));
    }
    r.v[8] = (uint32_t)acc;
    return r;
}

// V5: V4 with the multiply-accumulates written as instructions, so that the compiler cannot re-associate the column sums
// back into independent partial sums joined by 64-bit additions
__device__ __forceinline__ void mad_vv(uint64_t& acc, uint32_t a, uint32_t b) { asm("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b) : "vcc"); }
__device__ __forceinline__ void mad_vs(uint64_t& acc, uint32_t a, uint32_t b) { asm("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(acc) : "v"(a), "s"(b) : "vcc"); }
__device__ __forceinline__ F29 mul_v5(const F29& a, const F29& b) {
    uint32_t m[9];
    F29 r;
    uint64_t acc = 0;
#pragma unroll
    for (int k = 0; k < 9; k++) {
#pragma unroll
        for (int i = 0; i <= k; i++) mad_vv(acc, a.v[i], b.v[k - i]);
#pragma unroll
        for (int i = 0; i < k; i++) mad_vs(acc, m[i], p29(k - i));
        m[k] = ((uint32_t)acc * INV29) & M29;
        mad_vs(acc, m[k], p29(0));
        acc >>= 29;
    }
#pragma unroll
    for (int k = 9; k < 17; k++) {
#pragma unroll
        for (int i = k - 8; i < 9; i++) mad_vv(acc, a.v[i], b.v[k - i]);
#pragma unroll
        for (int i = k - 8; i < 9; i++) mad_vs(acc, m[i], p29(k - i));
        r.v[k - 9] = (uint32_t)acc & M29;
        acc >>= 29;
    }
    r.v[8] = (uint32_t)acc;
    return r;
}
// V7: V4 with every partial sum given a second (empty) use, which stops the re-association without opaque producers
#define KEEP(x) asm volatile("" :: "v"(x))
__device__ __forceinline__ F29 mul_v7(const F29& a, const F29& b) {
    uint32_t m[9];
    F29 r;
    uint64_t acc = 0;
#pragma unroll
    for (int k = 0; k < 9; k++) {
#pragma unroll
        for (int i = 0; i <= k; i++) { acc += (uint64_t)a.v[i] * b.v[k - i]; KEEP(acc); }
#pragma unroll
        for (int i = 0; i < k; i++) { acc += (uint64_t)m[i] * p29(k - i); KEEP(acc); }
        m[k] = ((uint32_t)acc * INV29) & M29;
        acc += (uint64_t)m[k] * p29(0);
        acc >>= 29;
    }
#pragma unroll
    for (int k = 9; k < 17; k++) {
#pragma unroll
        for (int i = k - 8; i < 9; i++) { acc += (uint64_t)a.v[i] * b.v[k - i]; KEEP(acc); }
#pragma unroll
        for (int i = k - 8; i < 9; i++) { acc += (uint64_t)m[i] * p29(k - i); KEEP(acc); }
        r.v[k - 9] = (uint32_t)acc & M29;
        acc >>= 29;
    }
    r.v[8] = (uint32_t)acc;
    return r;
}
// V6: two chains per column (products / reduction terms) joined by one addition: half the dependent chain length of V5
__device__ __forceinline__ F29 mul_v6(const F29& a, const F29& b) {
    uint32_t m[9];
    F29 r;
    uint64_t carry = 0;
#pragma unroll
    for (int k = 0; k < 17; k++) {
        uint64_t pa = carry, pm = 0;
#pragma unroll
        for (int i = 0; i < 9; i++) { const int j = k - i; if (j < 0 || j > 8) continue; mad_vv(pa, a.v[i], b.v[j]); }
#pragma unroll
        for (int i = 0; i < 9; i++) { const int j = k - i; if (j < 1 || j > 8 || i >= k) continue; mad_vs(pm, m[i], p29(j)); }
        uint64_t acc = pa + pm;
        if (k < 9) {
            m[k] = ((uint32_t)acc * INV29) & M29;
            mad_vs(acc, m[k], p29(0));
        } else {
            r.v[k - 9] = (uint32_t)acc & M29;
        }
        carry = acc >> 29;
    }
    r.v[8] = (uint32_t)carry;
    return r;
}


// V8: 5 x 52-bit limbs held as doubles (R = 2^260); every limb product by the two-FMA split in round-toward-zero (Emmart, Zheng,
// Weems, ARITH 2018): hi = fma_rz(a, b, 2^104) = 2^104 + floor(ab / 2^52) 2^52, lo = fma_rz(a, b, (2^104 + 2^52) - hi) = 2^52 + ab mod 2^52;
// the mantissa fields of hi / lo ARE the two 52-bit halves, so the column sums are integer additions of the bit patterns (the exponent
// fields, a known count per column, are subtracted up front). Per limb product: 2 v_fma_f64 + 1 v_add_f64 + 2 v_lshl_add_u64 = 5
// instructions for 52 x 52 bit-products; v_mad_u64_u32 does 29 x 29 in ONE. N products share one reduction (the fr_dot form).
struct F52 { double v[5]; };
#define V8_C1 0x1p104
#define V8_C2 (0x1p104 + 0x1p52)
#define V8_LOB 0x4330000000000000ull   // bit pattern of 2^52
#define V8_HIB 0x4670000000000000ull   // bit pattern of 2^104
#define V8_M52 0xfffffffffffffull
__device__ __forceinline__ constexpr double p52(int i) {
    constexpr double k[5] = {(double)0x1f593f0000001ull, (double)0x4879b9709143eull, (double)0x181585d2833e8ull, (double)0xa029b85045b68ull, (double)0x30644e72e131ull};
    return k[i];
}
#define V8_PINV ((double)0x1f593efffffffull)   // -p^-1 mod 2^52
__device__ __forceinline__ void v8_mac(uint64_t* col, int k, double a, double b) {
    const double hi = __builtin_fma(a, b, V8_C1);          // MODE.fp_round (f64) = toward zero, set once by the kernel
    const double lo = __builtin_fma(a, b, V8_C2 - hi);
    col[k] += (uint64_t)__double_as_longlong(lo);
    col[k + 1] += (uint64_t)__double_as_longlong(hi);
}
__device__ __forceinline__ constexpr int v8_pairs(int k) { return (k < 0 || k > 8) ? 0 : (k < 5 ? k + 1 : 9 - k); }
template <int N>
__device__ __forceinline__ F52 mul_v8(const F52* a, const F52* b) {
    uint64_t col[11];
#pragma unroll
    for (int k = 0; k < 11; k++)   // exponent fields of every lo / hi this column will receive: N products + one reduction
        col[k] = 0ull - (uint64_t)(N + 1) * ((uint64_t)v8_pairs(k) * V8_LOB + (uint64_t)v8_pairs(k - 1) * V8_HIB);
#pragma unroll
    for (int n = 0; n < N; n++)
#pragma unroll
        for (int i = 0; i < 5; i++)
#pragma unroll
            for (int j = 0; j < 5; j++) v8_mac(col, i + j, a[n].v[i], b[n].v[j]);
#pragma unroll
    for (int i = 0; i < 5; i++) {
        const uint64_t tl = col[i] & V8_M52;
        const double td = __longlong_as_double((long long)(tl | V8_LOB)) - 0x1p52;
        const double qh = __builtin_fma(td, V8_PINV, V8_C1);
        const double ql = __builtin_fma(td, V8_PINV, V8_C2 - qh);   // 2^52 + (t * pinv mod 2^52)
        const double q = ql - 0x1p52;
#pragma unroll
        for (int j = 0; j < 5; j++) v8_mac(col, i + j, q, p52(j));
        col[i + 1] += col[i] >> 52;   // col[i] is a multiple of 2^52 now
    }
    F52 r;
#pragma unroll
    for (int k = 0; k < 5; k++) {
        r.v[k] = __longlong_as_double((long long)((col[5 + k] & V8_M52) | V8_LOB)) - 0x1p52;
        if (k < 4) col[6 + k] += col[5 + k] >> 52;
    }
    return r;
}
template <int N>
__global__ __launch_bounds__(64) void kbench8(uint32_t* x, int n, uint64_t* dbg) {
    asm volatile("s_setreg_imm32_b32 hwreg(HW_REG_MODE, 2, 2), 3");   // f64 rounding: toward zero
    const int tid = blockIdx.x * blockDim.x + threadIdx.x;
    F52 a[N], b[N];
    for (int i = 0; i < 5; i++) {
        const uint64_t la = ((uint64_t)x[tid * 18 + 2 * i] | ((uint64_t)x[tid * 18 + 2 * i + 1] << 32)) & (i == 4 ? 0x1fffffffffffull : V8_M52);
        const uint64_t lb = ((uint64_t)x[tid * 18 + 8 + 2 * i] | ((uint64_t)x[tid * 18 + 9 + 2 * i] << 32)) & (i == 4 ? 0x1fffffffffffull : V8_M52);
        for (int q = 0; q < N; q++) { a[q].v[i] = (double)(la ^ (uint64_t)q); b[q].v[i] = (double)(lb ^ (uint64_t)(3 * q)); }
    }
    if (tid == 777 && dbg) for (int i = 0; i < 5; i++) { dbg[i] = (uint64_t)a[0].v[i]; dbg[5 + i] = (uint64_t)b[0].v[i]; }
    for (int i = 0; i < n; i++) {
        a[0] = mul_v8<N>(a, b);
        for (int q = 1; q < N; q++) a[q].v[0] = a[0].v[1];   // the other products follow the chain
    }
    if (tid == 777 && dbg) for (int i = 0; i < 5; i++) dbg[10 + i] = (uint64_t)a[0].v[i];
    for (int i = 0; i < 5; i++) { x[tid * 18 + 2 * i] = (uint32_t)(uint64_t)a[0].v[i]; }
}
// the adopted 9 x 29-bit form with N products per reduction (fr.h fr_dot<N>), same dependent chain
template <int N>
__global__ __launch_bounds__(64) void kbench_dot(uint32_t* x, int n) {
    const int tid = blockIdx.x * blockDim.x + threadIdx.x;
    Fr a[N], b[N];
    for (int q = 0; q < N; q++)
        for (int i = 0; i < 9; i++) { a[q].v[i] = (x[tid * 18 + i] ^ q) & HZ_M29; b[q].v[i] = (x[tid * 18 + 9 + i] ^ (3 * q)) & HZ_M29; }
    for (int q = 0; q < N; q++) { a[q].v[8] &= 0xffff; b[q].v[8] &= 0xffff; }
    for (int i = 0; i < n; i++) {
        a[0] = fr_dot<N>(a, b);
        for (int q = 1; q < N; q++) a[q].v[0] = a[0].v[1];
    }
    for (int i = 0; i < 9; i++) x[tid * 18 + i] = a[0].v[i];
}


// V11 (round 5, the latency question of SURVEY 8a' "alternative lane mapping"): ONE product spread over the lanes of a quad -- lane q
// (0..2) of the quad holds limbs 3q .. 3q+2 of each operand, lane 3 holds zeros (the "fourth block" that makes every shift uniform).
// Block Montgomery in radix B = 2^87 (R = B^3 = 2^261, the same domain as fr.h): for j = 0..2
//     every lane:  acc += A_q * B_j                 (B_j = lane j's three limbs, three DPP quad broadcasts; nine multiply-accumulates)
//     lane 0:      M = -acc_low / p mod B           (three limbs, the interleaved reduction inside the low block)
//     every lane:  acc += M * P_q                   (M from lane 0 by three DPP broadcasts; nine multiply-accumulates)
//     every lane:  acc = acc >> 87 + (low block of lane q + 1)   (six DPP moves of raw 64-bit column sums; lane 0 adds its carry)
// then two rounds of carry normalisation that ripple across the lanes. Same integer as fr_mul (the quotient m is unique).
struct Q3 { uint32_t v[3]; };
template <int CTRL>
__device__ __forceinline__ uint32_t dppq(uint32_t x) { return (uint32_t)__builtin_amdgcn_update_dpp((int)x, (int)x, CTRL, 0xF, 0xF, false); }   // every lane of a quad has a source: `old` is never kept
__device__ __forceinline__ uint64_t dppq64_next(uint64_t x) {   // from lane q + 1 of the quad (quad_perm [1,2,3,3]: the zero lane keeps reading itself and stays zero)
    const uint32_t lo = dppq<0xF9>((uint32_t)x), hi = dppq<0xF9>((uint32_t)(x >> 32));
    return (uint64_t)lo | ((uint64_t)hi << 32);
}
template <int J>
__device__ __forceinline__ void v11_step(uint64_t* c, const Q3& a, const Q3& b, const Q3& pq, uint64_t lane0_mask) {
    const uint32_t b0 = dppq<J * 0x55>(b.v[0]), b1 = dppq<J * 0x55>(b.v[1]), b2 = dppq<J * 0x55>(b.v[2]);
    c[0] += (uint64_t)a.v[0] * b0;
    c[1] += (uint64_t)a.v[0] * b1 + (uint64_t)a.v[1] * b0;
    c[2] += (uint64_t)a.v[0] * b2 + (uint64_t)a.v[1] * b1 + (uint64_t)a.v[2] * b0;
    c[3] += (uint64_t)a.v[1] * b2 + (uint64_t)a.v[2] * b1;
    c[4] += (uint64_t)a.v[2] * b2;
    // the quotient block from the low block (meaningful in lane 0; the other lanes run the same instructions on their own columns)
    uint64_t t0 = c[0];
    const uint32_t m0 = ((uint32_t)t0 * INV29) & M29;
    t0 += (uint64_t)m0 * p29(0);
    uint64_t t1 = c[1] + (t0 >> 29) + (uint64_t)m0 * p29(1);
    const uint32_t m1 = ((uint32_t)t1 * INV29) & M29;
    t1 += (uint64_t)m1 * p29(0);
    uint64_t t2 = c[2] + (t1 >> 29) + (uint64_t)m0 * p29(2) + (uint64_t)m1 * p29(1);
    const uint32_t m2 = ((uint32_t)t2 * INV29) & M29;
    t2 += (uint64_t)m2 * p29(0);
    const uint64_t carry = (t2 >> 29) & lane0_mask;
    const uint32_t M0 = dppq<0x00>(m0), M1 = dppq<0x00>(m1), M2 = dppq<0x00>(m2);
    c[3] += (uint64_t)M1 * pq.v[2] + (uint64_t)M2 * pq.v[1];
    c[4] += (uint64_t)M2 * pq.v[2];
    // columns 0..2 of lanes 1.. go down one lane as they are; lane 0's have become its carry
    const uint64_t l0 = c[0] + (uint64_t)M0 * pq.v[0];
    const uint64_t l1 = c[1] + (uint64_t)M0 * pq.v[1] + (uint64_t)M1 * pq.v[0];
    const uint64_t l2 = c[2] + (uint64_t)M0 * pq.v[2] + (uint64_t)M1 * pq.v[1] + (uint64_t)M2 * pq.v[0];
    c[0] = c[3] + dppq64_next(l0) + carry;
    c[1] = c[4] + dppq64_next(l1);
    c[2] = dppq64_next(l2);
    c[3] = 0; c[4] = 0;
}
__device__ __forceinline__ Q3 mul_v11(const Q3& a, const Q3& b, const Q3& pq, uint64_t lane0_mask) {
    uint64_t c[5] = {0, 0, 0, 0, 0};
    v11_step<0>(c, a, b, pq, lane0_mask);
    v11_step<1>(c, a, b, pq, lane0_mask);
    v11_step<2>(c, a, b, pq, lane0_mask);
    // carries: inside the lane, then the lane's carry-out to lane q + 1 (quad_perm [3,0,1,2]: lane 0 receives lane 3's zero), twice
#pragma unroll
    for (int round = 0; round < 2; round++) {
        c[1] += c[0] >> 29; c[0] &= M29;
        c[2] += c[1] >> 29; c[1] &= M29;
        const uint64_t out = c[2] >> 29; c[2] &= M29;
        const uint32_t lo = dppq<0x93>((uint32_t)out), hi = dppq<0x93>((uint32_t)(out >> 32));
        c[0] += (uint64_t)lo | ((uint64_t)hi << 32);
    }
    Q3 r;
    r.v[0] = (uint32_t)c[0]; r.v[1] = (uint32_t)c[1]; r.v[2] = (uint32_t)c[2];
    return r;
}
// dependent chain a <- a * b on quads; the first launch also checks every product of the chain's first steps against fr_mul
__global__ __launch_bounds__(64) void kbench11(uint32_t* x, int n, unsigned int* bad) {
    const int tid = blockIdx.x * blockDim.x + threadIdx.x;
    const int q = threadIdx.x & 3, quad = tid >> 2;
    // operands of the quad: nine limbs each, the same derivation as kbench<4> from x[quad * 18 ..]
    uint32_t A[9], B[9];
    for (int i = 0; i < 9; i++) { A[i] = x[quad * 18 + i] & M29; B[i] = x[quad * 18 + 9 + i] & M29; }
    A[8] &= 0xffff; B[8] &= 0xffff;
    Q3 a, b, pq;
    for (int i = 0; i < 3; i++) {
        a.v[i] = q < 3 ? A[3 * q + i] : 0u;
        b.v[i] = q < 3 ? B[3 * q + i] : 0u;
        pq.v[i] = q == 0 ? p29(i) : q == 1 ? p29(3 + i) : q == 2 ? p29(6 + i) : 0u;
    }
    const uint64_t lane0_mask = q == 0 ? ~0ull : 0ull;
    if (bad) {   // verification: 64 dependent products, each compared with the one-lane product of the gathered operands
        F29 fa, fb;
        for (int i = 0; i < 9; i++) { fa.v[i] = A[i]; fb.v[i] = B[i]; }
        for (int it = 0; it < 64; it++) {
            a = mul_v11(a, b, pq, lane0_mask);
            fa = mul_v4(fa, fb);
            // normalise the reference the same way (mul_v4 leaves limb 8 with the top carry, limbs 0..7 below 2^29: already unique)
            for (int i = 0; i < 3; i++)
                if (q < 3 && a.v[i] != fa.v[3 * q + i]) atomicAdd(bad, 1u);
        }
        return;
    }
    for (int i = 0; i < n; i++) { a = mul_v11(a, b, pq, lane0_mask); b.v[0] ^= a.v[0] & 1; }
    for (int i = 0; i < 3; i++) x[tid * 3 + i] = a.v[i];
}

template <int V>
__global__ __launch_bounds__(64) void kbench(uint32_t* x, int n) {
    const int tid = blockIdx.x * blockDim.x + threadIdx.x;
    if (V >= 3) {
        F29 a, b;
        for (int i = 0; i < 9; i++) { a.v[i] = x[tid * 18 + i] & M29; b.v[i] = x[tid * 18 + 9 + i] & M29; }
        a.v[8] &= 0xffff; b.v[8] &= 0xffff;
        for (int i = 0; i < n; i++) { a = (V == 3) ? mul_v3(a, b) : (V == 4) ? mul_v4(a, b) : (V == 5) ? mul_v5(a, b) : (V == 6) ? mul_v6(a, b) : mul_v7(a, b); b.v[0] ^= a.v[0] & 1; }
        for (int i = 0; i < 9; i++) x[tid * 18 + i] = a.v[i];
    } else {
        Fc a, b;
        for (int i = 0; i < 8; i++) { a.v[i] = x[tid * 18 + i]; b.v[i] = x[tid * 18 + 9 + i]; }
        a.v[7] &= 0x0fffffff; b.v[7] &= 0x0fffffff;
        if (V == 0) {
            Fr am = fr_unpack(a), bm = fr_unpack(b);
            for (int i = 0; i < n; i++) { am = fr_mul(am, bm); bm.v[0] ^= am.v[0] & 1; }
            for (int i = 0; i < 8; i++) x[tid * 18 + i] = am.v[i];
        } else {
            for (int i = 0; i < n; i++) { a = mul_v2(a, b); b.v[0] ^= a.v[0] & 1; }
            for (int i = 0; i < 8; i++) x[tid * 18 + i] = a.v[i];
        }
    }
}

int main() {
    const int n = 20000;
    for (int waves : {256 * 4, 256 * 4 * 2, 256 * 4 * 4, 256 * 4 * 8}) {
        const int threads = waves * 64;
        std::vector<uint32_t> h(threads * 18);
        for (size_t i = 0; i < h.size(); i++) h[i] = (uint32_t)(i * 2654435761u + 12345);
        uint32_t* d;
        hipMalloc(&d, h.size() * 4);
        uint64_t* dbg;
        hipMalloc(&dbg, 15 * 8);
        if (waves == 256 * 4) {   // V11 against the one-lane product, 64 dependent products per quad
            unsigned int* bad; unsigned int hb = 0;
            hipMalloc(&bad, 4); hipMemset(bad, 0, 4);
            hipMemcpy(d, h.data(), h.size() * 4, hipMemcpyHostToDevice);
            hipLaunchKernelGGL(kbench11, dim3(waves), dim3(64), 0, 0, d, 64, bad);
            hipMemcpy(&hb, bad, 4, hipMemcpyDeviceToHost);
            printf("V11CHECK %u limb mismatches in %d quads x 64 products (0 expected)\n", hb, threads / 4);
            hipFree(bad);
        }
        for (int v : {4, 11, 7, 8, 10, 9}) {
            hipMemcpy(d, h.data(), h.size() * 4, hipMemcpyHostToDevice);
            hipEvent_t e0, e1;
            hipEventCreate(&e0); hipEventCreate(&e1);
            auto launch = [&]() {
                if (v == 0) hipLaunchKernelGGL(kbench<0>, dim3(waves), dim3(64), 0, 0, d, n);
                if (v == 2) hipLaunchKernelGGL(kbench<2>, dim3(waves), dim3(64), 0, 0, d, n);
                if (v == 3) hipLaunchKernelGGL(kbench<3>, dim3(waves), dim3(64), 0, 0, d, n);
                if (v == 4) hipLaunchKernelGGL(kbench<4>, dim3(waves), dim3(64), 0, 0, d, n);
                if (v == 5) hipLaunchKernelGGL(kbench<5>, dim3(waves), dim3(64), 0, 0, d, n);
                if (v == 6) hipLaunchKernelGGL(kbench<6>, dim3(waves), dim3(64), 0, 0, d, n);
                if (v == 7) hipLaunchKernelGGL(kbench<7>, dim3(waves), dim3(64), 0, 0, d, n);
                if (v == 8) hipLaunchKernelGGL(kbench8<1>, dim3(waves), dim3(64), 0, 0, d, n, dbg);       // double-FMA limbs, one product
                if (v == 9) hipLaunchKernelGGL(kbench8<3>, dim3(waves), dim3(64), 0, 0, d, n, (uint64_t*)nullptr);   // three products, one reduction
                if (v == 10) hipLaunchKernelGGL(kbench_dot<3>, dim3(waves), dim3(64), 0, 0, d, n);          // fr_dot<3>, 9 x 29
                if (v == 11) hipLaunchKernelGGL(kbench11, dim3(waves), dim3(64), 0, 0, d, n, (unsigned int*)nullptr);   // one product per QUAD of lanes
            };
            launch();
            hipDeviceSynchronize();
            if (v == 8 && waves == 256 * 4) {   // operands and result of one lane, for tools/microbench/check_v8.py (a0 * b^n / R^n mod p, R = 2^260)
                uint64_t h8[15];
                hipMemcpy(h8, dbg, sizeof h8, hipMemcpyDeviceToHost);
                printf("V8CHECK n=%d", n);
                for (int i = 0; i < 15; i++) printf(" %llx", (unsigned long long)h8[i]);
                printf("\n");
                hipMemcpy(d, h.data(), h.size() * 4, hipMemcpyHostToDevice);
            }
            hipEventRecord(e0);
            launch();
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms;
            hipEventElapsedTime(&ms, e0, e1);
            uint32_t chk[9];
            hipMemcpy(chk, d + 18 * 777, sizeof chk, hipMemcpyDeviceToHost);
            printf("[%08x %08x] ", chk[0], chk[8]);
            const int prods = (v == 9 || v == 10) ? 3 : 1;
            const double chains = v == 11 ? threads / 4.0 : threads;   // V11: a quad of lanes per chain
            printf("waves/SIMD=%d variant=%d: %.2f ms  %.1f ns per dependent step per wave  %.2f Gmul/s (%d product%s per reduction%s)\n", waves / 1024, v, ms, ms * 1e6 / n,
                   chains * n * prods / ms / 1e6, prods, prods > 1 ? "s" : "", v == 11 ? ", one product per quad of lanes" : "");
        }
        hipFree(d);
    }
    // verify V2 and V3 against V0 on a few values is done in the unit tests of the adopted variant
    return 0;
}
