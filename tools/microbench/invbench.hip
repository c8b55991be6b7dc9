// Throughput of the field primitives on gfx950: product, square, inversion (safegcd), canonicalisation.
// build: hipcc -O3 --offload-arch=gfx950 invbench.hip -o invbench
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <vector>
#define HZ_FR_INLINE 1
#include "../../circuits_amd/csrc/fr.h"
using namespace hz;

template <int V>
__global__ __launch_bounds__(64) void kbench(uint32_t* x, int n) {
    const int tid = blockIdx.x * blockDim.x + threadIdx.x;
    Fc a, b;
    for (int i = 0; i < 8; i++) { a.v[i] = x[tid * 16 + i]; b.v[i] = x[tid * 16 + 8 + i]; }
    a.v[7] &= 0x0fffffff; b.v[7] &= 0x0fffffff;
    Fr am = fr_unpack(a), bm = fr_unpack(b);
    for (int i = 0; i < n; i++) {
        if (V == 0) am = fr_mul(am, bm);
        if (V == 1) am = fr_add(fr_inv(am), bm);
        if (V == 2) am = fr_sqr(am);
        if (V == 3) { const Fc c = fr_to_canon(am); am = fr_add(fr_unpack(c), bm); }
        if (V == 4) { Fr z[2] = {am, bm}; am = fr_dot<2>(z, z); }
        bm.v[0] ^= am.v[0] & 1;
    }
    for (int i = 0; i < 8; i++) x[tid * 16 + i] = am.v[i];
}

int main() {
    const char* names[] = {"mul", "inv", "sqr", "to_canon", "dot2"};
    for (int waves : {256 * 4 / 2, 256 * 4, 256 * 4 * 2, 256 * 4 * 4}) {
        const int threads = waves * 64;
        std::vector<uint32_t> h(threads * 16);
        for (size_t i = 0; i < h.size(); i++) h[i] = (uint32_t)(i * 2654435761u + 12345);
        uint32_t* d;
        hipMalloc(&d, h.size() * 4);
        for (int v = 0; v < 5; v++) {
            const int n = v == 1 ? 200 : 10000;
            hipMemcpy(d, h.data(), h.size() * 4, hipMemcpyHostToDevice);
            hipEvent_t e0, e1;
            hipEventCreate(&e0); hipEventCreate(&e1);
            auto launch = [&]() {
                if (v == 0) hipLaunchKernelGGL(kbench<0>, dim3(waves), dim3(64), 0, 0, d, n);
                if (v == 1) hipLaunchKernelGGL(kbench<1>, dim3(waves), dim3(64), 0, 0, d, n);
                if (v == 2) hipLaunchKernelGGL(kbench<2>, dim3(waves), dim3(64), 0, 0, d, n);
                if (v == 3) hipLaunchKernelGGL(kbench<3>, dim3(waves), dim3(64), 0, 0, d, n);
                if (v == 4) hipLaunchKernelGGL(kbench<4>, dim3(waves), dim3(64), 0, 0, d, n);
            };
            launch();
            hipDeviceSynchronize();
            hipEventRecord(e0);
            launch();
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms;
            hipEventElapsedTime(&ms, e0, e1);
            printf("waves/SIMD %.1f %-9s %8.3f ms  %8.2f Gop/s  (%.1f ns per op per wave)\n", waves / 1024.0, names[v], ms, (double)threads * n / ms / 1e6,
                   ms * 1e6 / n);
        }
        hipFree(d);
    }
    return 0;
}
