// MFMA-int8 versus v_mad_u64_u32 as a source of byte products, in time AND in energy (gfx950). VERDICT r1 item 9: the constant-
// operand products of the field arithmetic (the Montgomery m*p, Poseidon's linear layers) are matrix products with an operand that
// every lane shares. This measures the raw rate of both units with the device's package power sampled by the caller
// (tools/microbench/run_mfmabench.py reads rocm-smi while each kernel loops for ~3 s); the padding a 261 x 261-bit product needs
// on the matrix unit is accounted for in profiles/r02_mfmabench.txt.
// build: hipcc -O3 --offload-arch=gfx950 mfmabench.hip -o mfmabench ; run: mfmabench <valu|mfma32|mfma16> <seconds>
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <chrono>

typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));
#define ITER 2048

__global__ __launch_bounds__(256) void k_valu(uint32_t* out, uint32_t seed) {
    uint64_t x[8];
    uint32_t a = seed | 1u, b = threadIdx.x | 3u;
    for (int c = 0; c < 8; c++) x[c] = seed + c;
    for (int it = 0; it < ITER; it++) {
#pragma unroll
        for (int c = 0; c < 8; c++) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(x[c]) : "v"(a), "v"(b) : "vcc");
    }
    uint32_t acc = 0;
    for (int c = 0; c < 8; c++) acc ^= (uint32_t)x[c];
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}
// 32x32x32 int8: 32768 multiply-accumulates per instruction and wavefront
__global__ __launch_bounds__(256) void k_mfma32(uint32_t* out, uint32_t seed) {
    v4i a = {(int)seed, (int)threadIdx.x, 3, 4}, b = {5, (int)seed, 7, (int)threadIdx.x};
    v16i c0 = {}, c1 = {}, c2 = {}, c3 = {};
    for (int it = 0; it < ITER; it++) {
        c0 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, c1, 0, 0, 0);
        c2 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, c2, 0, 0, 0);
        c3 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, c3, 0, 0, 0);
    }
    int acc = 0;
    for (int i = 0; i < 16; i++) acc ^= c0[i] ^ c1[i] ^ c2[i] ^ c3[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = (uint32_t)acc;
}
// 16x16x64 int8: 16384 multiply-accumulates per instruction and wavefront
__global__ __launch_bounds__(256) void k_mfma16(uint32_t* out, uint32_t seed) {
    v4i a = {(int)seed, (int)threadIdx.x, 3, 4}, b = {5, (int)seed, 7, (int)threadIdx.x};
    v4i c0 = {}, c1 = {}, c2 = {}, c3 = {};
    for (int it = 0; it < ITER; it++) {
        c0 = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, b, c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, b, c1, 0, 0, 0);
        c2 = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, b, c2, 0, 0, 0);
        c3 = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, b, c3, 0, 0, 0);
    }
    int acc = 0;
    for (int i = 0; i < 4; i++) acc ^= c0[i] ^ c1[i] ^ c2[i] ^ c3[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = (uint32_t)acc;
}

int main(int argc, char** argv) {
    const char* which = argc > 1 ? argv[1] : "valu";
    const double secs = argc > 2 ? atof(argv[2]) : 3.0;
    uint32_t* d;
    hipMalloc(&d, 256 * 32 * 256 * 4);
    const int blocks = 256 * 8, threads = 256;   // 8 wavefronts per SIMD
    double insts_per_wave, macs_per_inst;
    if (!strcmp(which, "valu")) { insts_per_wave = ITER * 8.0; macs_per_inst = 64 * 16; }          // 64 lanes x (4 x 4 byte products of a 32 x 32-bit multiply)
    else if (!strcmp(which, "mfma32")) { insts_per_wave = ITER * 4.0; macs_per_inst = 32768; }
    else { insts_per_wave = ITER * 4.0; macs_per_inst = 16384; }
    auto launch = [&]() {
        if (!strcmp(which, "valu")) hipLaunchKernelGGL(k_valu, dim3(blocks), dim3(threads), 0, 0, d, 12345u);
        else if (!strcmp(which, "mfma32")) hipLaunchKernelGGL(k_mfma32, dim3(blocks), dim3(threads), 0, 0, d, 12345u);
        else hipLaunchKernelGGL(k_mfma16, dim3(blocks), dim3(threads), 0, 0, d, 12345u);
    };
    launch();
    hipDeviceSynchronize();
    const auto t0 = std::chrono::steady_clock::now();
    long n = 0;
    double el = 0;
    while (el < secs) {
        for (int i = 0; i < 20; i++) launch();
        hipDeviceSynchronize();
        n += 20;
        el = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    }
    const double waves = (double)blocks * threads / 64;
    const double inst_s = n * waves * insts_per_wave / el;
    printf("%s: %.2f s, %.3e wave-instructions/s, %.2f cycles per wave-instruction per SIMD at 2.4 GHz, %.3e byte-MAC/s\n", which, el, inst_s,
           2.4e9 * 1024 / inst_s, inst_s * macs_per_inst);
    return 0;
}
