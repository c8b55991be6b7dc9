// Store-stream ceiling of the witness layout: a wavefront owns 64 consecutive units and writes one 32-byte element per lane and
// signal (2 KiB of contiguous HBM per signal, signals N*32 bytes apart) -- the pattern of every witness kernel (devcommon.h store_fr).
// Patterns:  0 = two dwordx4 per lane at a 32-byte lane stride (store_fr as it is: every instruction half-fills its 64-byte sectors)
//            1 = the same 2 KiB written as two fully contiguous 1 KiB instructions (lane stride 16 bytes)
//            2 = pattern 0, but lanes paired through DPP so that each instruction is contiguous (what a transposed store_fr would do)
//            4 = a bit signal (bit, 0, .., 0) through store_fr;  3 = the same, each instruction 1 KiB contiguous (bit of element L/2 in lane L)
//            5 = as 3 in half-wavefront groups: 512 contiguous bytes per group and instruction (sha_dev.h put_word_bits)
// build: hipcc -O3 --offload-arch=gfx950 storebench.hip -o storebench ; run: storebench [units=65536] [signals=2048] [reps=5]
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

template <int P>
__global__ __launch_bounds__(64) void k_store(uint8_t* base, uint32_t N, uint32_t S, uint32_t seed) {
    const uint32_t u = blockIdx.x * 64 + threadIdx.x, lane = threadIdx.x;
    uint32_t v = u * 2654435761u + seed;
#pragma unroll 4
    for (uint32_t s = 0; s < S; s++) {
        uint8_t* row = base + ((size_t)s * N + (size_t)blockIdx.x * 64) * 32;
        v = v * 1664525u + 1013904223u;
        if constexpr (P == 0) {
            uint4* q = reinterpret_cast<uint4*>(row + lane * 32);
            q[0] = make_uint4(v, v + 1, v + 2, v + 3);
            q[1] = make_uint4(v + 4, v + 5, v + 6, v + 7);
        } else if constexpr (P == 1) {
            uint4* q = reinterpret_cast<uint4*>(row + lane * 16);
            q[0] = make_uint4(v, v + 1, v + 2, v + 3);
            q[64] = make_uint4(v + 4, v + 5, v + 6, v + 7);
        } else if constexpr (P == 2) {
            // element of lane L = (lo, hi). Instruction A writes elements 0..31: lane 2j -> lo of element j, lane 2j+1 -> hi of j.
            // Instruction B writes elements 32..63 likewise. Data movement: 8 ds_bpermute per element (worst case, no DPP shortcut).
            const uint32_t srcA = lane >> 1, srcB = 32 + (lane >> 1);
            uint32_t e[8];
            for (int i = 0; i < 8; i++) e[i] = v + i;
            uint32_t a[4], b[4];
            for (int i = 0; i < 4; i++) {
                const uint32_t lo = __shfl(e[i], srcA), hi = __shfl(e[4 + i], srcA);
                a[i] = (lane & 1) ? hi : lo;
                const uint32_t lo2 = __shfl(e[i], srcB), hi2 = __shfl(e[4 + i], srcB);
                b[i] = (lane & 1) ? hi2 : lo2;
            }
            uint4* q = reinterpret_cast<uint4*>(row + lane * 16);
            q[0] = make_uint4(a[0], a[1], a[2], a[3]);
            q[64] = make_uint4(b[0], b[1], b[2], b[3]);
        } else if constexpr (P == 3) {
            // bit signal: element = (bit, 0, 0, 0 | 0, 0, 0, 0); contiguous form needs the bit of element L/2 in lane L
            const uint32_t bit = v & 1u;
            const uint32_t bA = __shfl(bit, lane >> 1), bB = __shfl(bit, 32 + (lane >> 1));
            uint4* q = reinterpret_cast<uint4*>(row + lane * 16);
            q[0] = make_uint4((lane & 1) ? 0u : bA, 0u, 0u, 0u);
            q[64] = make_uint4((lane & 1) ? 0u : bB, 0u, 0u, 0u);
        } else if constexpr (P == 5) {
            // bit signal, half-wavefront groups: each instruction writes 512 contiguous bytes per group (sha_dev.h put_word_bits)
            const uint32_t bit = v & 1u, gl = lane & 31u;
            const int src = (int)((lane & 32u) | (gl >> 1));
            const uint32_t bA = __shfl(bit, src), bB = __shfl(bit, src + 16);
            uint4* q = reinterpret_cast<uint4*>(row + (lane & 32u) * 32 + gl * 16);
            q[0] = make_uint4((gl & 1) ? 0u : bA, 0u, 0u, 0u);
            q[32] = make_uint4((gl & 1) ? 0u : bB, 0u, 0u, 0u);
        } else if constexpr (P == 4) {
            // bit signal, store_fr as it is
            const uint32_t bit = v & 1u;
            uint4* q = reinterpret_cast<uint4*>(row + lane * 32);
            q[0] = make_uint4(bit, 0u, 0u, 0u);
            q[1] = make_uint4(0u, 0u, 0u, 0u);
        }
    }
}

template <int P>
static void run(uint8_t* d, uint32_t N, uint32_t S, int reps, const char* what) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k_store<P>, dim3(N / 64), dim3(64), 0, 0, d, N, S, 1u);
    hipDeviceSynchronize();
    float best = 1e30f, sum = 0;
    for (int r = 0; r < reps; r++) {
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL(k_store<P>, dim3(N / 64), dim3(64), 0, 0, d, N, S, (uint32_t)r);
        hipEventRecord(e1, 0);
        hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        sum += ms; if (ms < best) best = ms;
    }
    const double gb = (double)N * S * 32 / 1e9;
    printf("pattern %d  %-58s %7.2f GB  mean %7.3f ms  best %7.3f ms  -> %5.2f TB/s (best %5.2f)\n", P, what, gb, sum / reps, best,
           gb / (sum / reps), gb / best);
}

int main(int argc, char** argv) {
    const uint32_t N = argc > 1 ? (uint32_t)atoi(argv[1]) : 65536u;
    const uint32_t S = argc > 2 ? (uint32_t)atoi(argv[2]) : 2048u;
    const int reps = argc > 3 ? atoi(argv[3]) : 5;
    uint8_t* d;
    if (hipMalloc(&d, (size_t)N * S * 32) != hipSuccess) { printf("hipMalloc failed\n"); return 1; }
    hipMemset(d, 0, (size_t)N * S * 32);
    printf("units %u, signals %u, one wavefront per workgroup, %u workgroups\n", N, S, N / 64);
    run<0>(d, N, S, reps, "2 x dwordx4 per lane, 32-byte lane stride (store_fr)");
    run<1>(d, N, S, reps, "2 x dwordx4, each instruction 1 KiB contiguous");
    run<2>(d, N, S, reps, "as 1, elements transposed across lanes with 16 bpermutes");
    run<4>(d, N, S, reps, "bit signal, store_fr");
    run<3>(d, N, S, reps, "bit signal, contiguous (2 bpermutes)");
    run<5>(d, N, S, reps, "bit signal, 512 B contiguous per half-wavefront");
    hipFree(d);
    return 0;
}
