// Dependent-chain issue rate of scalar against vector integer instructions for ONE wavefront per SIMD (the regime of a single batch's
// SHA-256 chain: 766 dependent compressions on one lane): n dependent (rotate-xor-add) steps written with s_ and with v_ instructions.
// build: hipcc -O3 --offload-arch=gfx950 salubench.hip -o salubench
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

__global__ __launch_bounds__(64) void k_valu(uint32_t* out, uint32_t seed, int n) {
    uint32_t a = seed + threadIdx.x, b = seed * 3u + blockIdx.x;
    for (int i = 0; i < n; i++) {
#pragma unroll
        for (int k = 0; k < 8; k++) {
            asm volatile("v_alignbit_b32 %0, %1, %1, 7\n\tv_xor_b32 %0, %0, %2\n\tv_add_u32 %1, %0, %1\n\tv_alignbit_b32 %2, %1, %1, 13"
                         : "+v"(a), "+v"(b), "+v"(seed));
        }
    }
    out[blockIdx.x * 64 + threadIdx.x] = a ^ b ^ seed;
}
__global__ __launch_bounds__(64) void k_salu(uint32_t* out, uint32_t seed, int n) {
    // the same steps on the scalar unit (no scalar rotate: shift, shift, or); inline assembly, because the compiler moves a uniform
    // rotate-xor-add chain to the VECTOR unit by itself (v_alignbit_b32), whatever the uniformity of its operands
    uint32_t a = __builtin_amdgcn_readfirstlane(seed + blockIdx.x), b = __builtin_amdgcn_readfirstlane(seed * 3u + blockIdx.x), c = __builtin_amdgcn_readfirstlane(seed);
    uint32_t t = 0, u = 0;
    for (int i = 0; i < n; i++) {
#pragma unroll
        for (int k = 0; k < 8; k++) {
            asm volatile("s_lshr_b32 %3, %1, 7\n\ts_lshl_b32 %4, %1, 25\n\ts_or_b32 %3, %3, %4\n\ts_xor_b32 %0, %3, %2\n\ts_add_u32 %1, %0, %1\n\t"
                         "s_lshr_b32 %3, %1, 13\n\ts_lshl_b32 %4, %1, 19\n\ts_or_b32 %2, %3, %4"
                         : "+s"(a), "+s"(b), "+s"(c), "+s"(t), "+s"(u) : : "scc");
        }
    }
    if (threadIdx.x == 0) out[blockIdx.x * 64] = a ^ b ^ c;
}
int main() {
    const int waves = 1024, n = 20000;
    uint32_t* d;
    hipMalloc(&d, waves * 64 * 4);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    float msv = 0, mss = 0;
    hipLaunchKernelGGL(k_valu, dim3(waves), dim3(64), 0, 0, d, 5u, 10);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k_valu, dim3(waves), dim3(64), 0, 0, d, 5u, n);
    hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&msv, e0, e1);
    hipLaunchKernelGGL(k_salu, dim3(waves), dim3(64), 0, 0, d, 5u, 10);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k_salu, dim3(waves), dim3(64), 0, 0, d, 5u, n);
    hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&mss, e0, e1);
    const double steps = (double)n * 8 * 4;   // dependent "operations" (rotate, xor, add, rotate)
    printf("one wavefront per SIMD, %d x 32 dependent operations: vector %.3f ms = %.2f ns per operation (4 instructions per 4 operations), scalar %.3f ms = %.2f ns per operation (8 instructions per 4 operations: a rotate is shift, shift, or)\n",
           n, msv, msv * 1e6 / steps, mss, mss * 1e6 / steps);
    printf("per instruction: vector %.2f ns (%.1f cycles at 2.4 GHz), scalar %.2f ns (%.1f cycles)\n", msv * 1e6 / steps, msv * 1e6 / steps * 2.4,
           mss * 1e6 / (steps * 8 / 4), mss * 1e6 / (steps * 8 / 4) * 2.4);
    return 0;
}
