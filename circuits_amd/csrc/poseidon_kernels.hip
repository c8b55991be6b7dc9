// K1 poseidon_batch<t>: one Poseidon permutation per lane (SURVEY 8a' K1; metric "Poseidon-BN254/sec").
// Replaces the witness code of circomlib 0.5.2 `Poseidon(nInputs)` (call sites
// reference src/lib/hash-state.circom:32, src/decode-tx.circom:275).
//
// HBM traffic per permutation: 32*(t-1) B in + 32 B out (digest mode), plus 96*(8t+R_P) B when the
// S-box witness is requested. Digest mode is integer-VALU bound; witness mode is the HBM-write
// bound regime of the rollup witness.
#define HZ_FR_INLINE 1  // throughput kernel: keep the product inline (register-allocated operands)
#include <hip/hip_runtime.h>
#include "../../include/hermez_witness.h"
#include "devcommon.h"
#include "hostutil.h"

namespace hz {

template <int T, bool WIT>
__global__ __launch_bounds__(256) void poseidon_batch_kernel(const uint8_t* __restrict__ in, uint8_t* __restrict__ out,
                                                              uint8_t* __restrict__ wit, size_t n) {
    extern __shared__ __attribute__((aligned(16))) uint32_t lds_raw[];
    uint32_t* lds = lds_raw;
    const Fr* K = poseidon_consts<T>(lds);
    __syncthreads();
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        Fr x[T - 1];
#pragma unroll
        for (int j = 0; j < T - 1; j++) x[j] = fr_from_canon(load_fr(in + (i * (T - 1) + j) * 32));
        Fr h;
        if (WIT) {
            WitSboxSink sink{WitOut{wit, (uint32_t)n, (uint32_t)i}, 0u};
            h = poseidon_hash<T>(x, K, sink);
        } else {
            NoSink sink;
            h = poseidon_hash<T>(x, K, sink);
        }
        store_fr(out + i * 32, fr_to_canon(h));
    }
}

template <int T>
static hipError_t launch_poseidon(size_t n, const void* d_in, void* d_out, void* d_wit, hipStream_t s) {
    if (n == 0) return hipSuccess;
    const int block = 256;
    size_t blocks = (n + block - 1) / block;
    if (blocks > 256 * 8) blocks = 256 * 8;  // grid-stride beyond 8 blocks per CU
    const size_t lds = poseidon_lds_bytes<T>();
    if (d_wit)
        hipLaunchKernelGGL((poseidon_batch_kernel<T, true>), dim3((unsigned)blocks), dim3(block), lds, s,
                           (const uint8_t*)d_in, (uint8_t*)d_out, (uint8_t*)d_wit, n);
    else
        hipLaunchKernelGGL((poseidon_batch_kernel<T, false>), dim3((unsigned)blocks), dim3(block), lds, s,
                           (const uint8_t*)d_in, (uint8_t*)d_out, (uint8_t*)nullptr, n);
    return hipGetLastError();
}

// K0 fr_ops: one field operation per lane on canonical operands (SURVEY 8a' K0), the self test of fr.h on the device
__global__ __launch_bounds__(256) void fr_ops_kernel(int op, const uint8_t* __restrict__ a, const uint8_t* __restrict__ b, uint8_t* __restrict__ out, size_t n) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const Fr x = fr_from_canon(load_fr(a + i * 32));
        const Fr y = b ? fr_from_canon(load_fr(b + i * 32)) : fr_zero();
        Fr r;
        switch (op) {
            case HZ_FR_ADD: r = fr_add(x, y); break;
            case HZ_FR_SUB: r = fr_sub(x, y); break;
            case HZ_FR_MUL: r = fr_mul(x, y); break;
            case HZ_FR_SQR: r = fr_sqr(x); break;
            case HZ_FR_INV: r = fr_inv(x); break;
            case HZ_FR_MULADD: r = fr_cond_sub_4p(fr_muladd(x, y, fr_add(x, y))); break;   // x*y + (x + y)
            default: r = fr_mul(fr_dbl(x), fr_neg(y)); break;                          // HZ_FR_MIX: 2x * (-y)
        }
        store_fr(out + i * 32, fr_to_canon(r));
    }
}

hipError_t poseidon_batch_launch(int t, size_t n, const void* d_in, void* d_out, void* d_wit, hipStream_t s) {
    switch (t) {
        case 2: return launch_poseidon<2>(n, d_in, d_out, d_wit, s);
        case 3: return launch_poseidon<3>(n, d_in, d_out, d_wit, s);
        case 4: return launch_poseidon<4>(n, d_in, d_out, d_wit, s);
        case 5: return launch_poseidon<5>(n, d_in, d_out, d_wit, s);
        case 6: return launch_poseidon<6>(n, d_in, d_out, d_wit, s);
        case 7: return launch_poseidon<7>(n, d_in, d_out, d_wit, s);
    }
    return hipErrorInvalidValue;
}

}  // namespace hz

using namespace hz;

extern "C" hz_status hz_poseidon_batch_dev(int32_t t, size_t n, const void* d_in, void* d_out, void* d_sbox_witness, void* stream) {
    if (t < 2 || t > 7 || (n && (!d_in || !d_out))) return set_err(HZ_ERR_ARG, "hz_poseidon_batch_dev: bad argument");
    if (hz_device_count() <= 0) return set_err(HZ_ERR_NODEVICE, "no usable gfx950 device");
    HZ_HIP(poseidon_batch_launch(t, n, d_in, d_out, d_sbox_witness, (hipStream_t)stream));
    return HZ_OK;
}

extern "C" hz_status hz_fr_ops(int32_t device, int32_t op, size_t n, const uint8_t* a, const uint8_t* b, uint8_t* out) {
    if (op < 0 || op > HZ_FR_MIX || (n && (!a || !out))) return set_err(HZ_ERR_ARG, "hz_fr_ops: bad argument");
    if (hz_device_count() <= 0) return set_err(HZ_ERR_NODEVICE, "no usable gfx950 device");
    if (n == 0) return HZ_OK;
    HZ_HIP(hipSetDevice(device));
    for (size_t i = 0; i < n; i++)
        if (!canon_lt_p(a + i * 32) || (b && !canon_lt_p(b + i * 32))) return set_err(HZ_ERR_INPUT, "hz_fr_ops: operand >= r");
    DevBuf d_a, d_b, d_o;
    HZ_HIP(d_a.alloc(n * 32));
    HZ_HIP(d_o.alloc(n * 32));
    HZ_HIP(hipMemcpy(d_a.p, a, n * 32, hipMemcpyHostToDevice));
    if (b) {
        HZ_HIP(d_b.alloc(n * 32));
        HZ_HIP(hipMemcpy(d_b.p, b, n * 32, hipMemcpyHostToDevice));
    }
    size_t blocks = (n + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(fr_ops_kernel, dim3((unsigned)blocks), dim3(256), 0, 0, op, (const uint8_t*)d_a.p, (const uint8_t*)d_b.p, (uint8_t*)d_o.p, n);
    HZ_HIP(hipGetLastError());
    HZ_HIP(hipDeviceSynchronize());
    HZ_HIP(hipMemcpy(out, d_o.p, n * 32, hipMemcpyDeviceToHost));
    return HZ_OK;
}

extern "C" hz_status hz_poseidon_batch(int32_t device, int32_t t, size_t n, const uint8_t* in, uint8_t* out, uint8_t* sbox_witness) {
    if (t < 2 || t > 7 || (n && (!in || !out))) return set_err(HZ_ERR_ARG, "hz_poseidon_batch: bad argument");
    if (hz_device_count() <= 0) return set_err(HZ_ERR_NODEVICE, "no usable gfx950 device");
    if (n == 0) return HZ_OK;
    HZ_HIP(hipSetDevice(device));
    // reject non-canonical inputs like the reference's input parser does (values are reduced
    // mod r there; at this ABI they must already be < r)
    for (size_t i = 0; i < n * (size_t)(t - 1); i++)
        if (!canon_lt_p(in + i * 32)) return set_err(HZ_ERR_INPUT, "hz_poseidon_batch: input element >= r");
    DevBuf d_in, d_out, d_wit;
    const size_t wit_bytes = (size_t)3 * poseidon_nsbox_rt(t) * n * 32;
    HZ_HIP(d_in.alloc(n * (size_t)(t - 1) * 32));
    HZ_HIP(d_out.alloc(n * 32));
    if (sbox_witness) HZ_HIP(d_wit.alloc(wit_bytes));
    HZ_HIP(hipMemcpy(d_in.p, in, n * (size_t)(t - 1) * 32, hipMemcpyHostToDevice));
    HZ_HIP(poseidon_batch_launch(t, n, d_in.p, d_out.p, sbox_witness ? d_wit.p : nullptr, 0));
    HZ_HIP(hipDeviceSynchronize());
    HZ_HIP(hipMemcpy(out, d_out.p, n * 32, hipMemcpyDeviceToHost));
    if (sbox_witness) HZ_HIP(hipMemcpy(sbox_witness, d_wit.p, wit_bytes, hipMemcpyDeviceToHost));
    return HZ_OK;
}
