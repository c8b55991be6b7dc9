// How well do witness stores hide under integer work? Every wavefront (one per workgroup, like the witness kernels) owns 64 units
// and alternates K v_mad_u64_u32 (eight independent chains, the S-box's instruction) with the stores of `rows` 2 KiB signal rows
// (two dwordx4 per lane, signals N*32 bytes apart: store_fr). Sweeping K at a fixed byte count separates the two rooflines:
//   K = 0: the store stream alone; large K: the integer pipe alone; k_smt sits at ~93 VALU instructions per store instruction.
// Patterns of the stores: 0 = store_fr (32-byte lane stride), 1 = each instruction 1 KiB contiguous, 2 = store_fr with the
// nontemporal hint, 3 = burst (all `rows` rows of an iteration back to back, as the S-box sink does) instead of spread.
// build: hipcc -O3 --offload-arch=gfx950 mixbench.hip -o mixbench ; run: mixbench [units=262144] [signals=2304] [waves_per_simd_cap=2]
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

template <int P>
__device__ __forceinline__ void put(uint8_t* row, uint32_t lane, const uint32_t* v) {
    if constexpr (P == 1) {
        uint4* q = reinterpret_cast<uint4*>(row + lane * 16);
        q[0] = make_uint4(v[0], v[1], v[2], v[3]);
        q[64] = make_uint4(v[4], v[5], v[6], v[7]);
    } else if constexpr (P == 2) {
        typedef uint32_t u4 __attribute__((ext_vector_type(4)));
        u4* q = reinterpret_cast<u4*>(row + lane * 32);
        u4 a = {v[0], v[1], v[2], v[3]}, b = {v[4], v[5], v[6], v[7]};
        __builtin_nontemporal_store(a, q);
        __builtin_nontemporal_store(b, q + 1);
    } else {
        uint4* q = reinterpret_cast<uint4*>(row + lane * 32);
        q[0] = make_uint4(v[0], v[1], v[2], v[3]);
        q[1] = make_uint4(v[4], v[5], v[6], v[7]);
    }
}

// K8 = MAC groups of eight per store row. OCC = wavefronts per SIMD the kernel allows: its register count is raised to 512 / OCC by
// naming the top register of that budget in an empty asm (an LDS reservation does not do it: launches draw on a 64 KB pool).
template <int P, int OCC>
__global__ __launch_bounds__(64) void k_mix(uint8_t* base, uint32_t N, uint32_t S, int K8, int rows, uint32_t seed, uint64_t* sink) {
    if constexpr (OCC == 1) asm volatile("" ::: "v255", "a250");
    else if constexpr (OCC == 2) asm volatile("" ::: "v250");
    else if constexpr (OCC == 3) asm volatile("" ::: "v165");
    else if constexpr (OCC == 4) asm volatile("" ::: "v125");
    const uint32_t lane = threadIdx.x;
    uint64_t acc[8];
    uint32_t a = blockIdx.x * 64 + lane + seed, b = a * 2654435761u + 12345u;
#pragma unroll
    for (int i = 0; i < 8; i++) acc[i] = a + i;
    for (uint32_t s = 0; s < S; s += rows) {
        if (P == 3) {
            for (int k = 0; k < K8 * rows; k++) {
#pragma unroll
                for (int i = 0; i < 8; i++) acc[i] = (uint64_t)(uint32_t)acc[i] * b + acc[(i + 1) & 7];
            }
            for (int r = 0; r < rows; r++) {
                uint32_t v[8];
#pragma unroll
                for (int i = 0; i < 8; i++) v[i] = (uint32_t)acc[i] + r;
                put<0>(base + ((size_t)(s + r) * N + (size_t)blockIdx.x * 64) * 32, lane, v);
            }
        } else {
            for (int r = 0; r < rows; r++) {
                for (int k = 0; k < K8; k++) {
#pragma unroll
                    for (int i = 0; i < 8; i++) acc[i] = (uint64_t)(uint32_t)acc[i] * b + acc[(i + 1) & 7];
                }
                uint32_t v[8];
#pragma unroll
                for (int i = 0; i < 8; i++) v[i] = (uint32_t)acc[i];
                put<P>(base + ((size_t)(s + r) * N + (size_t)blockIdx.x * 64) * 32, lane, v);
            }
        }
    }
    uint64_t x = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) x ^= acc[i];
    if (x == 0x1234567u) sink[0] = x;
}

template <int P, int OCC>
static void run(uint8_t* d, uint64_t* sink, uint32_t N, uint32_t S, int K8, int rows, size_t lds, const char* what) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k_mix<P, OCC>), dim3(N / 64), dim3(64), lds, 0, d, N, S, K8, rows, 1u, sink);
    hipDeviceSynchronize();
    float best = 1e30f;
    for (int r = 0; r < 3; r++) {
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL((k_mix<P, OCC>), dim3(N / 64), dim3(64), lds, 0, d, N, S, K8, rows, (uint32_t)r, sink);
        hipEventRecord(e1, 0);
        hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    const double gb = (double)N * S * 32 / 1e9;
    const double macs = (double)(N / 64) * S * K8 * 8;              // wave-level v_mad_u64_u32
    const double cyc = macs > 0 ? best * 1e-3 * 2.4e9 * 1024 / macs : 0;   // SIMD cycles per wave-MAC (1024 SIMDs at 2.4 GHz)
    printf("%-28s K=%4d MACs/row-store-pair  %7.3f ms  %5.2f TB/s  %5.2f cycles/MAC\n", what, K8 * 8, best, gb / best, cyc);
}

template <int OCC>
static void sweep(uint8_t* d, uint64_t* sink, uint32_t N, uint32_t S) {
    printf("-- at most %d wavefront(s) per SIMD\n", OCC);
    const int ks[] = {0, 6, 12, 23, 35, 46, 70, 93};   // x8 MACs per row (a row = 2 store instructions): k_smt ~ 186 VALU per row
    for (int k : ks) run<0, OCC>(d, sink, N, S, k, 1, 0, "store_fr, spread");
    for (int k : {12, 23, 46}) run<1, OCC>(d, sink, N, S, k, 1, 0, "1 KiB contiguous, spread");
    for (int k : {23}) run<3, OCC>(d, sink, N, S, k, 12, 0, "store_fr, bursts of 12 rows");
}
int main(int argc, char** argv) {
    const uint32_t N = argc > 1 ? (uint32_t)atoi(argv[1]) : 262144u;
    const uint32_t S = argc > 2 ? (uint32_t)atoi(argv[2]) : 2304u;
    uint8_t* d;
    uint64_t* sink;
    if (hipMalloc(&d, (size_t)N * S * 32 + 4096) != hipSuccess) { printf("hipMalloc failed\n"); return 1; }
    hipMalloc(&sink, 64);
    hipMemset(d, 0, (size_t)N * S * 32);
    printf("units %u (%u wavefronts), signals %u, %.1f GB per launch\n", N, N / 64, S, (double)N * S * 32 / 1e9);
    sweep<1>(d, sink, N, S);
    sweep<2>(d, sink, N, S);
    sweep<3>(d, sink, N, S);
    sweep<4>(d, sink, N, S);
    hipFree(d);
    return 0;
}
