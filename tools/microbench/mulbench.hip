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
        for (int v : {3, 4, 5, 7}) {
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
            };
            launch();
            hipDeviceSynchronize();
            hipEventRecord(e0);
            launch();
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms;
            hipEventElapsedTime(&ms, e0, e1);
            uint32_t chk[9];
            hipMemcpy(chk, d + 18 * 777, sizeof chk, hipMemcpyDeviceToHost);
            printf("[%08x %08x] ", chk[0], chk[8]);
            printf("waves/SIMD=%d variant=%d: %.2f ms  %.1f ns per dependent mul per wave  %.2f Gmul/s\n", waves / 1024, v, ms, ms * 1e6 / n, (double)threads * n / ms / 1e6);
        }
        hipFree(d);
    }
    // verify V2 and V3 against V0 on a few values is done in the unit tests of the adopted variant
    return 0;
}
