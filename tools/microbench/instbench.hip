// Issue-rate micro-benchmark of the integer VALU instructions the field arithmetic is made of (gfx950).
// Each lane runs NCH independent dependency chains of one instruction; with 8 waves per SIMD the
// result is the throughput-bound cycles per wave-instruction (2.4 GHz nominal clock assumed).
// build: hipcc -O3 --offload-arch=gfx950 instbench.hip -o instbench
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#define NCH 8
#define ITER 4096

#define BENCH_KERNEL(NAME, DECL, BODY)                                                     \
    __global__ __launch_bounds__(256) void NAME(uint32_t* out, uint32_t seed) {            \
        DECL;                                                                              \
        for (int it = 0; it < ITER; it++) {                                                \
            _Pragma("unroll") for (int c = 0; c < NCH; c++) { BODY; }                      \
        }                                                                                  \
        uint32_t acc = 0;                                                                  \
        _Pragma("unroll") for (int c = 0; c < NCH; c++) acc ^= (uint32_t)x[c];             \
        out[blockIdx.x * blockDim.x + threadIdx.x] = acc;                                  \
    }

#define DECL64 uint64_t x[NCH]; uint32_t a = seed | 1u, b = threadIdx.x | 3u; for (int c = 0; c < NCH; c++) x[c] = seed + c
#define DECL32 uint32_t x[NCH]; uint32_t a = seed | 1u, b = threadIdx.x | 3u; for (int c = 0; c < NCH; c++) x[c] = seed + c

BENCH_KERNEL(k_mad_u64_u32, DECL64, asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(x[c]) : "v"(a), "v"(b) : "vcc"))
BENCH_KERNEL(k_mad_u64_u32_s, DECL64, asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(x[c]) : "s"(seed), "v"(b) : "vcc"))
BENCH_KERNEL(k_mad_i64_i32, DECL64, asm volatile("v_mad_i64_i32 %0, vcc, %1, %2, %0" : "+v"(x[c]) : "v"(a), "v"(b) : "vcc"))
BENCH_KERNEL(k_mul_lo_u32, DECL32, asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(x[c]) : "v"(a)))
BENCH_KERNEL(k_mul_hi_u32, DECL32, asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(x[c]) : "v"(a)))
BENCH_KERNEL(k_mul_u32_u24, DECL32, asm volatile("v_mul_u32_u24 %0, %0, %1" : "+v"(x[c]) : "v"(a)))
BENCH_KERNEL(k_mad_u32_u24, DECL32, asm volatile("v_mad_u32_u24 %0, %0, %1, %2" : "+v"(x[c]) : "v"(a), "v"(b)))
BENCH_KERNEL(k_add_u32, DECL32, asm volatile("v_add_u32 %0, %0, %1" : "+v"(x[c]) : "v"(a)))
BENCH_KERNEL(k_and_b32, DECL32, asm volatile("v_and_b32 %0, %0, %1" : "+v"(x[c]) : "v"(a)))
BENCH_KERNEL(k_add3_u32, DECL32, asm volatile("v_add3_u32 %0, %0, %1, %2" : "+v"(x[c]) : "v"(a), "v"(b)))
BENCH_KERNEL(k_lshl_add_u32, DECL32, asm volatile("v_lshl_add_u32 %0, %0, 3, %1" : "+v"(x[c]) : "v"(a)))
BENCH_KERNEL(k_lshrrev_b64, DECL64, asm volatile("v_lshrrev_b64 %0, 1, %0" : "+v"(x[c])))
BENCH_KERNEL(k_lshl_add_u64, DECL64, asm volatile("v_lshl_add_u64 %0, %0, 0, %1" : "+v"(x[c]) : "v"(x[(c + 1) % NCH])))
BENCH_KERNEL(k_alignbit_b32, DECL32, asm volatile("v_alignbit_b32 %0, %0, %1, 29" : "+v"(x[c]) : "v"(a)))
BENCH_KERNEL(k_cndmask, DECL32, asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(x[c]) : "v"(a) : "vcc"))
BENCH_KERNEL(k_addc, DECL32, asm volatile("v_addc_co_u32 %0, vcc, %0, %1, vcc" : "+v"(x[c]) : "v"(a) : "vcc"))
BENCH_KERNEL(k_fma_f64, DECL64, asm volatile("v_fma_f64 %0, %0, %1, %0" : "+v"(x[c]) : "v"(x[(c + 1) % NCH])))
BENCH_KERNEL(k_dot4_u32_u8, DECL32, asm volatile("v_dot4_u32_u8 %0, %1, %2, %0" : "+v"(x[c]) : "v"(a), "v"(b)))
BENCH_KERNEL(k_pk_mul_lo_u16, DECL32, asm volatile("v_pk_mul_lo_u16 %0, %0, %1" : "+v"(x[c]) : "v"(a)))
BENCH_KERNEL(k_pk_mad_u16, DECL32, asm volatile("v_pk_mad_u16 %0, %0, %1, %2" : "+v"(x[c]) : "v"(a), "v"(b)))
// a dependent chain of ONE accumulator (latency-bound): cycles between dependent mads
__global__ __launch_bounds__(64) void k_mad_dep(uint32_t* out, uint32_t seed) {
    uint64_t x = seed; uint32_t a = seed | 1u, b = threadIdx.x | 3u;
    for (int it = 0; it < ITER * NCH; it++) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(x) : "v"(a), "v"(b) : "vcc");
    out[blockIdx.x * blockDim.x + threadIdx.x] = (uint32_t)x;
}

template <class K>
static void run(const char* name, K kernel, int blocks, int threads, double per_wave_insts, uint32_t* d) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(kernel, dim3(blocks), dim3(threads), 0, 0, d, 12345u);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(kernel, dim3(blocks), dim3(threads), 0, 0, d, 12345u);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    const double waves = (double)blocks * threads / 64;
    const double waves_per_simd = waves / 1024.0;
    const double cyc = ms * 1e-3 * 2.4e9 / (per_wave_insts * waves_per_simd);
    printf("%-18s %8.3f ms  %6.2f cycles per wave-instruction per SIMD (%.1f waves/SIMD)\n", name, ms, cyc, waves_per_simd);
}

int main() {
    uint32_t* d;
    hipMalloc(&d, 256 * 32 * 256 * 4 * 4);
    const double n = (double)ITER * NCH;
#define RUN(K) run(#K, K, 256 * 8 * 4, 256, n, d)
    RUN(k_mad_u64_u32); RUN(k_mad_u64_u32_s); RUN(k_mad_i64_i32); RUN(k_mul_lo_u32); RUN(k_mul_hi_u32); RUN(k_mul_u32_u24); RUN(k_mad_u32_u24);
    RUN(k_add_u32); RUN(k_and_b32); RUN(k_add3_u32); RUN(k_lshl_add_u32); RUN(k_lshrrev_b64); RUN(k_lshl_add_u64); RUN(k_alignbit_b32);
    RUN(k_cndmask); RUN(k_addc); RUN(k_fma_f64); RUN(k_dot4_u32_u8); RUN(k_pk_mul_lo_u16); RUN(k_pk_mad_u16);
    run("k_mad_dep(1w/SIMD)", k_mad_dep, 1024, 64, n, d);
    run("k_mad_u64(1w/SIMD)", k_mad_u64_u32, 1024, 64, n, d);
    run("k_mad_u64(2w/SIMD)", k_mad_u64_u32, 2048, 64, n, d);
    return 0;
}
