// Shader clock actually delivered while something else loads the device: a one-wavefront kernel compares s_memtime (shader-engine
// clock ticks) with s_memrealtime (constant 100 MHz) over ~2 ms windows. Run it beside a benchmark (another process on the same GPU):
//   clockprobe <seconds> [period_ms]   -> one line per sample: time, MHz
// build: hipcc -O3 --offload-arch=gfx950 clockprobe.hip -o clockprobe
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <chrono>
#include <thread>

__global__ void k_probe(uint64_t* out, uint64_t real_ticks) {
    if (threadIdx.x != 0) return;
    const uint64_t r0 = __builtin_amdgcn_s_memrealtime(), c0 = __builtin_amdgcn_s_memtime();
    uint64_t r1 = r0;
    while (r1 - r0 < real_ticks) { __builtin_amdgcn_s_sleep(8); r1 = __builtin_amdgcn_s_memrealtime(); }
    const uint64_t c1 = __builtin_amdgcn_s_memtime();
    out[0] = r1 - r0;
    out[1] = c1 - c0;
}

int main(int argc, char** argv) {
    const double secs = argc > 1 ? atof(argv[1]) : 10.0;
    const int period_ms = argc > 2 ? atoi(argv[2]) : 250;
    uint64_t* d;
    uint64_t h[2];
    hipMalloc(&d, 16);
    const auto t0 = std::chrono::steady_clock::now();
    for (;;) {
        const double el = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        if (el > secs) break;
        hipLaunchKernelGGL(k_probe, dim3(1), dim3(64), 0, 0, d, (uint64_t)200000);   // 2 ms at 100 MHz
        hipMemcpy(h, d, 16, hipMemcpyDeviceToHost);
        printf("%8.2f s  shader clock %7.1f MHz\n", el, h[0] ? (double)h[1] / (double)h[0] * 100.0 : 0.0);
        fflush(stdout);
        std::this_thread::sleep_for(std::chrono::milliseconds(period_ms));
    }
    return 0;
}
