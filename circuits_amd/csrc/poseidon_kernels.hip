// K1 poseidon_batch<t>: one Poseidon permutation per lane (SURVEY 8a' K1; metric "Poseidon-BN254/sec").
// Replaces the witness code of circomlib 0.5.2 `Poseidon(nInputs)` (call sites
// reference src/lib/hash-state.circom:32, src/decode-tx.circom:275).
//
// HBM traffic per permutation: 32*(t-1) B in + 32 B out (digest mode), plus 96*(8t+R_P) B when the
// S-box witness is requested. Digest mode is integer-VALU bound; witness mode is the HBM-write
// bound regime of the rollup witness.
#define HZ_FR_INLINE 1  // throughput kernel: keep the product inline (register-allocated operands)
#include <hip/hip_runtime.h>
#include <mutex>
#include "../../include/hermez_witness.h"
#include "devcommon.h"
#include "hostutil.h"

namespace hz {
hipError_t launch_fr_sqrt(const void* d_a, void* d_out, size_t n, hipStream_t s);   // eddsa_kernels.hip (beside the curve code that uses it)

template <int T, bool WIT>
__global__ __launch_bounds__(256) void poseidon_batch_kernel(const uint8_t* __restrict__ in, uint8_t* __restrict__ out,
                                                              uint8_t* __restrict__ wit, size_t n) {
    const Fr* K = poseidon_consts<T, WIT>();   // the witness sink takes canonical S-box outputs: its own constant block
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
    if (d_wit)
        hipLaunchKernelGGL((poseidon_batch_kernel<T, true>), dim3((unsigned)blocks), dim3(block), 0, s,
                           (const uint8_t*)d_in, (uint8_t*)d_out, (uint8_t*)d_wit, n);
    else
        hipLaunchKernelGGL((poseidon_batch_kernel<T, false>), dim3((unsigned)blocks), dim3(block), 0, s,
                           (const uint8_t*)d_in, (uint8_t*)d_out, (uint8_t*)nullptr, n);
    return hipGetLastError();
}

// Poseidon DAG (the batch builder's Merkle work): lane = one hash job whose inputs are gathered from a table of
// field elements and whose digest goes back into the table. The host orders the jobs so that a segment only reads values
// written by earlier segments: all node versions of one tree level are one segment (level-parallel path recomputation).
template <int T>
__global__ __launch_bounds__(256) void poseidon_dag_kernel(uint8_t* __restrict__ vals, const uint32_t* __restrict__ job_in,
                                                            const uint32_t* __restrict__ job_out, size_t first, size_t count) {
    const Fr* K = poseidon_consts<T>();
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    const size_t job = first + i;
    Fr x[T - 1];
#pragma unroll
    for (int j = 0; j < T - 1; j++) x[j] = fr_from_canon(load_fr(vals + (size_t)job_in[job * HZ_DAG_MAX_IN + j] * 32));
    NoSink sink;
    const Fr h = poseidon_hash<T>(x, K, sink);
    store_fr(vals + (size_t)job_out[job] * 32, fr_to_canon(h));
}

template <int T>
static hipError_t launch_poseidon_dag(uint8_t* vals, const uint32_t* job_in, const uint32_t* job_out, size_t first, size_t count, hipStream_t s) {
    if (count == 0) return hipSuccess;
    // one wavefront per workgroup: a level of a 2048-transaction batch is a few thousand jobs, which should spread over all CUs
    hipLaunchKernelGGL(poseidon_dag_kernel<T>, dim3((unsigned)((count + 63) / 64)), dim3(64), 0, s, vals, job_in, job_out, first, count);
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
    if (op < 0 || op > HZ_FR_SQRT || (n && (!a || !out))) return set_err(HZ_ERR_ARG, "hz_fr_ops: bad argument");
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
    if (op == HZ_FR_SQRT) HZ_HIP(launch_fr_sqrt(d_a.p, d_o.p, n, 0));
    else hipLaunchKernelGGL(fr_ops_kernel, dim3((unsigned)blocks), dim3(256), 0, 0, op, (const uint8_t*)d_a.p, (const uint8_t*)d_b.p, (uint8_t*)d_o.p, n);
    HZ_HIP(hipGetLastError());
    HZ_HIP(hipDeviceSynchronize());
    HZ_HIP(hipMemcpy(out, d_o.p, n * 32, hipMemcpyDeviceToHost));
    return HZ_OK;
}

extern "C" hz_status hz_poseidon_dag(int32_t device, uint8_t* vals, uint64_t n_vals, const uint32_t* job_in, const uint32_t* job_out, uint64_t n_jobs,
                                     const uint32_t* seg_t, const uint64_t* seg_first, const uint64_t* seg_count, uint32_t n_segs, double* device_ms) {
    if ((n_vals && !vals) || (n_jobs && (!job_in || !job_out)) || (n_segs && (!seg_t || !seg_first || !seg_count)))
        return set_err(HZ_ERR_ARG, "hz_poseidon_dag: null argument");
    if (n_vals >= 0xFFFFFFFFull) return set_err(HZ_ERR_ARG, "hz_poseidon_dag: the value table is indexed with 32 bits");
    if (hz_device_count() <= 0) return set_err(HZ_ERR_NODEVICE, "no usable gfx950 device");
    for (uint32_t g = 0; g < n_segs; g++) {
        if (seg_t[g] < 2 || seg_t[g] > 7) return set_err(HZ_ERR_ARG, "hz_poseidon_dag: segment %u has width %u (2..7)", g, seg_t[g]);
        if (seg_first[g] + seg_count[g] > n_jobs) return set_err(HZ_ERR_ARG, "hz_poseidon_dag: segment %u exceeds the job list", g);
        for (uint64_t j = seg_first[g]; j < seg_first[g] + seg_count[g]; j++) {
            if (job_out[j] >= n_vals) return set_err(HZ_ERR_ARG, "hz_poseidon_dag: job %llu writes outside the table", (unsigned long long)j);
            for (uint32_t k = 0; k + 1 < seg_t[g]; k++)
                if (job_in[j * HZ_DAG_MAX_IN + k] >= n_vals) return set_err(HZ_ERR_ARG, "hz_poseidon_dag: job %llu reads outside the table", (unsigned long long)j);
        }
    }
    if (n_segs == 0 || n_jobs == 0) return HZ_OK;
    HZ_HIP(hipSetDevice(device));
    // The evaluator stays resident on the device between calls (a batch builder calls twice per batch: the messages of the signatures,
    // then every Merkle hash): buffers that only grow, a stream and two events of its own per device -- a first version allocated,
    // freed and synchronised the whole device on every call (a third of the evaluator's 6.4 ms per 2048-transaction batch).
    struct Resident { DevBuf vals, in, out; hipStream_t s = nullptr; hipEvent_t e0 = nullptr, e1 = nullptr; std::mutex mu; };
    // Two sets per device: a batch builder's worker thread evaluates one batch's Merkle hashes while the thread that walks the next
    // batch evaluates its signatures' messages (hzb_batch_build_begin) -- neither waits for the other's whole evaluation.
    static Resident resident[16][2];
    if (device < 0 || device >= 16) return set_err(HZ_ERR_ARG, "hz_poseidon_dag: device %d", device);
    std::unique_lock<std::mutex> lock(resident[device][0].mu, std::try_to_lock);
    int slot = 0;
    if (!lock.owns_lock()) {
        lock = std::unique_lock<std::mutex>(resident[device][1].mu);
        slot = 1;
    }
    Resident& R = resident[device][slot];
    if (!R.s) {
        HZ_HIP(hipStreamCreateWithFlags(&R.s, hipStreamNonBlocking));
        HZ_HIP(hipEventCreate(&R.e0));
        HZ_HIP(hipEventCreate(&R.e1));
    }
    auto grow = [](DevBuf& b, size_t bytes) -> hipError_t { return b.bytes >= bytes ? hipSuccess : b.alloc(bytes + bytes / 2); };
    HZ_HIP(grow(R.vals, n_vals * 32));
    HZ_HIP(grow(R.in, n_jobs * HZ_DAG_MAX_IN * sizeof(uint32_t)));
    HZ_HIP(grow(R.out, n_jobs * sizeof(uint32_t)));
    HZ_HIP(hipMemcpyAsync(R.vals.p, vals, n_vals * 32, hipMemcpyHostToDevice, R.s));
    HZ_HIP(hipMemcpyAsync(R.in.p, job_in, n_jobs * HZ_DAG_MAX_IN * sizeof(uint32_t), hipMemcpyHostToDevice, R.s));
    HZ_HIP(hipMemcpyAsync(R.out.p, job_out, n_jobs * sizeof(uint32_t), hipMemcpyHostToDevice, R.s));
    if (device_ms) HZ_HIP(hipEventRecord(R.e0, R.s));
    hipError_t e = hipSuccess;
    for (uint32_t g = 0; g < n_segs && e == hipSuccess; g++) {
        uint8_t* v = (uint8_t*)R.vals.p;
        const uint32_t* ji = (const uint32_t*)R.in.p;
        const uint32_t* jo = (const uint32_t*)R.out.p;
        switch (seg_t[g]) {
            case 2: e = launch_poseidon_dag<2>(v, ji, jo, seg_first[g], seg_count[g], R.s); break;
            case 3: e = launch_poseidon_dag<3>(v, ji, jo, seg_first[g], seg_count[g], R.s); break;
            case 4: e = launch_poseidon_dag<4>(v, ji, jo, seg_first[g], seg_count[g], R.s); break;
            case 5: e = launch_poseidon_dag<5>(v, ji, jo, seg_first[g], seg_count[g], R.s); break;
            case 6: e = launch_poseidon_dag<6>(v, ji, jo, seg_first[g], seg_count[g], R.s); break;
            default: e = launch_poseidon_dag<7>(v, ji, jo, seg_first[g], seg_count[g], R.s); break;
        }
    }
    HZ_HIP(e);
    if (device_ms) HZ_HIP(hipEventRecord(R.e1, R.s));
    HZ_HIP(hipMemcpyAsync(vals, R.vals.p, n_vals * 32, hipMemcpyDeviceToHost, R.s));
    HZ_HIP(hipStreamSynchronize(R.s));
    if (device_ms) {
        float ms = 0;
        HZ_HIP(hipEventElapsedTime(&ms, R.e0, R.e1));
        *device_ms = ms;
    }
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
