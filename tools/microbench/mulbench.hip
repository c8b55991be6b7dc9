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

template <int V>
__global__ __launch_bounds__(64) void kbench(uint32_t* x, int n) {
    const int tid = blockIdx.x * blockDim.x + threadIdx.x;
    if (V == 3) {
        F29 a, b;
        for (int i = 0; i < 9; i++) { a.v[i] = x[tid * 18 + i] & M29; b.v[i] = x[tid * 18 + 9 + i] & M29; }
        a.v[8] &= 0xffff; b.v[8] &= 0xffff;
        for (int i = 0; i < n; i++) { a = mul_v3(a, b); b.v[0] ^= a.v[0] & 1; }
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
        for (int v : {0, 2, 3}) {
            hipMemcpy(d, h.data(), h.size() * 4, hipMemcpyHostToDevice);
            hipEvent_t e0, e1;
            hipEventCreate(&e0); hipEventCreate(&e1);
            auto launch = [&]() {
                if (v == 0) hipLaunchKernelGGL(kbench<0>, dim3(waves), dim3(64), 0, 0, d, n);
                if (v == 2) hipLaunchKernelGGL(kbench<2>, dim3(waves), dim3(64), 0, 0, d, n);
                if (v == 3) hipLaunchKernelGGL(kbench<3>, dim3(waves), dim3(64), 0, 0, d, n);
            };
            launch();
            hipDeviceSynchronize();
            hipEventRecord(e0);
            launch();
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms;
            hipEventElapsedTime(&ms, e0, e1);
            printf("waves/SIMD=%d variant=%d: %.2f ms  %.1f ns per dependent mul per wave  %.2f Gmul/s\n", waves / 1024, v, ms, ms * 1e6 / n, (double)threads * n / ms / 1e6);
        }
        hipFree(d);
    }
    // verify V2 and V3 against V0 on a few values is done in the unit tests of the adopted variant
    return 0;
}
