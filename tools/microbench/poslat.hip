// Latency of ONE Poseidon t = 3 permutation (digest only) at one wavefront per SIMD: the adopted one-lane form (poseidon.h, sparse partial
// rounds) against the state spread over the lanes of a quad -- lane j holds state element j, the dense round of circomlib's template:
// every lane adds its constant, lane 0 (or every lane in a full round) takes the S-box, then each lane computes ITS row of the mix as
// one three-product dot (the other two state elements arrive by DPP quad broadcasts). The single-batch regime (DESIGN 3 "One product
// spread over lanes") is bound by 33 dependent level hashes per SMT chain; this is the measurement of what parallelism INSIDE a hash buys.
// build: hipcc -O3 --offload-arch=gfx950 poslat.hip -o poslat     (tools/microbench/build.sh)
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <vector>
#define HZ_FR_INLINE 1
#include "../../circuits_amd/csrc/devcommon.h"
namespace hzd {
#include "../../circuits_amd/csrc/gen/poseidon_consts_host.inc"
}
using namespace hz;

// dependent chain: h <- Poseidon(h, c), n times, one chain per lane
__global__ __launch_bounds__(64) void k_one_lane(uint32_t* x, int n) {
    const int tid = blockIdx.x * blockDim.x + threadIdx.x;
    const Fr* K = poseidon_consts<3>();
    Fr in[2];
    for (int i = 0; i < 9; i++) { in[0].v[i] = x[tid * 18 + i] & HZ_M29; in[1].v[i] = x[tid * 18 + 9 + i] & HZ_M29; }
    in[0].v[8] &= 0xffff; in[1].v[8] &= 0xffff;
    NoSink sink;
    for (int i = 0; i < n; i++) in[0] = poseidon_hash<3>(in, K, sink);
    for (int i = 0; i < 9; i++) x[tid * 18 + i] = in[0].v[i];
}

template <int CTRL>
__device__ __forceinline__ Fr dpp_fr(const Fr& a) {
    Fr r;
#pragma unroll
    for (int i = 0; i < 9; i++) r.v[i] = (uint32_t)__builtin_amdgcn_update_dpp((int)a.v[i], (int)a.v[i], CTRL, 0xF, 0xF, false);
    return r;
}
// one chain per QUAD: lane j < 3 holds state element j (lane 3 mirrors lane 0 and is ignored)
__global__ __launch_bounds__(64) void k_quad(uint32_t* x, int n, const Fr* __restrict__ C /*[65][3] Montgomery*/, const Fr* __restrict__ M /*[3][3] Montgomery*/, uint32_t* out9) {
    const int tid = blockIdx.x * blockDim.x + threadIdx.x;
    const int j = threadIdx.x & 3, quad = tid >> 2;
    const int jj = j < 3 ? j : 0;
    Fr in0, in1;
    for (int i = 0; i < 9; i++) { in0.v[i] = x[quad * 18 + i] & HZ_M29; in1.v[i] = x[quad * 18 + 9 + i] & HZ_M29; }
    in0.v[8] &= 0xffff; in1.v[8] &= 0xffff;
    const Fr m0 = M[jj * 3 + 0], m1 = M[jj * 3 + 1], m2 = M[jj * 3 + 2];   // this lane's row of the mix
    Fr h = in0;
    for (int it = 0; it < n; it++) {
        Fr st = jj == 0 ? fr_zero() : jj == 1 ? h : in1;   // (0, h, c)
#pragma unroll 1
        for (int r = 0; r < 65; r++) {
            st = fr_add(st, C[r * 3 + jj]);
            const bool full = r < 4 || r >= 61;
            if (full || jj == 0) {   // x^5 (lanes 1, 2 skip it in the partial rounds: the branch is the same for 57 rounds in a row)
                const Fr x2 = fr_sqr(st), x4 = fr_sqr(x2);
                st = fr_mul(x4, st);
            }
            const Fr v[3] = {dpp_fr<0x00>(st), dpp_fr<0x55>(st), dpp_fr<0xAA>(st)};
            const Fr row[3] = {m0, m1, m2};
            st = fr_dot<3>(row, v);
        }
        h = dpp_fr<0x00>(st);   // the digest = state[0] after the last mix
    }
    if (j == 0) for (int i = 0; i < 9; i++) out9[quad * 9 + i] = h.v[i];
}

static Fr to_mont(const uint64_t* w) {
    Fc c;
    for (int i = 0; i < 4; i++) { c.v[2 * i] = (uint32_t)w[i]; c.v[2 * i + 1] = (uint32_t)(w[i] >> 32); }
    return fr_from_canon(c);
}

int main() {
    const int n = 200;
    std::vector<Fr> C(65 * 3), M(9);
    for (int i = 0; i < 65 * 3; i++) C[i] = to_mont(hzd::HZ_POSEIDON_HC_T3[i]);
    for (int i = 0; i < 9; i++) M[i] = to_mont(hzd::HZ_POSEIDON_HM_T3[i]);
    Fr *dC, *dM;
    hipMalloc(&dC, C.size() * sizeof(Fr)); hipMalloc(&dM, M.size() * sizeof(Fr));
    hipMemcpy(dC, C.data(), C.size() * sizeof(Fr), hipMemcpyHostToDevice);
    hipMemcpy(dM, M.data(), M.size() * sizeof(Fr), hipMemcpyHostToDevice);
    for (int waves : {1024, 2048}) {
        const int threads = waves * 64;
        std::vector<uint32_t> h(threads * 18);
        for (size_t i = 0; i < h.size(); i++) h[i] = (uint32_t)(i * 2654435761u + 12345);
        uint32_t *d, *o9;
        hipMalloc(&d, h.size() * 4); hipMalloc(&o9, (size_t)(threads / 4) * 9 * 4);
        float ms1 = 0, ms4 = 0;
        hipEvent_t e0, e1;
        hipEventCreate(&e0); hipEventCreate(&e1);
        // one lane per chain (the first threads/4 chains are the quads' chains: same operands)
        hipMemcpy(d, h.data(), h.size() * 4, hipMemcpyHostToDevice);
        hipLaunchKernelGGL(k_one_lane, dim3(waves), dim3(64), 0, 0, d, 2);   // warm-up
        hipMemcpy(d, h.data(), h.size() * 4, hipMemcpyHostToDevice);
        hipEventRecord(e0);
        hipLaunchKernelGGL(k_one_lane, dim3(waves), dim3(64), 0, 0, d, n);
        hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms1, e0, e1);
        std::vector<uint32_t> r1(h.size());
        hipMemcpy(r1.data(), d, h.size() * 4, hipMemcpyDeviceToHost);
        hipMemcpy(d, h.data(), h.size() * 4, hipMemcpyHostToDevice);
        hipLaunchKernelGGL(k_quad, dim3(waves), dim3(64), 0, 0, d, 2, dC, dM, o9);
        hipEventRecord(e0);
        hipLaunchKernelGGL(k_quad, dim3(waves), dim3(64), 0, 0, d, n, dC, dM, o9);
        hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms4, e0, e1);
        std::vector<uint32_t> r4((size_t)(threads / 4) * 9);
        hipMemcpy(r4.data(), o9, r4.size() * 4, hipMemcpyDeviceToHost);
        // same field element? (both are Montgomery values below 2p with normalised limbs: compare modulo p through the difference)
        long bad = 0;
        for (int q = 0; q < threads / 4; q++) {
            Fr a, b;
            for (int i = 0; i < 9; i++) { a.v[i] = r1[(size_t)q * 18 + i]; b.v[i] = r4[(size_t)q * 9 + i]; }
            if (!fr_eq(a, b)) bad++;
        }
        printf("waves/SIMD=%d: one lane per hash %.1f us per dependent Poseidon(t=3), quad per hash %.1f us (%.2f x); %ld of %d chains differ after %d hashes\n", waves / 1024,
               ms1 * 1e3 / n, ms4 * 1e3 / n, ms4 / ms1, bad, threads / 4, n);
        hipFree(d); hipFree(o9);
    }
    return 0;
}
