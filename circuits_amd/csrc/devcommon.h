// Device-side helpers shared by all kernels: constant staging into LDS, 32-byte element
// loads/stores, the witness writer and the constraint-failure record.
#pragma once
#include <hip/hip_runtime.h>
#include "fr.h"
#include "poseidon.h"

namespace hz {

// ---- constant tables in device memory (Montgomery form); one private copy per translation unit,
// unused widths are dropped by the compiler -----------------------------------------------------
#define HZ_CONST_ARR static __device__ const
#include "gen/poseidon_consts.inc"
#include "gen/fee_table.inc"
#undef HZ_CONST_ARR

template <int T> __device__ __forceinline__ const Fr* poseidon_k_global();
#define HZ_PC(T) \
    template <> __device__ __forceinline__ const Fr* poseidon_k_global<T>() { return reinterpret_cast<const Fr*>(&HZ_POSEIDON_K_T##T[0][0]); }
HZ_PC(2) HZ_PC(3) HZ_PC(4) HZ_PC(5) HZ_PC(6) HZ_PC(7)
#undef HZ_PC
// the block for sinks that store the S-box witness (canonical S-box outputs, poseidon.h)
template <int T> __device__ __forceinline__ const Fr* poseidon_k_global_w();
#if HZ_POSEIDON_CANON_SBOX
#define HZ_PC(T) \
    template <> __device__ __forceinline__ const Fr* poseidon_k_global_w<T>() { return reinterpret_cast<const Fr*>(&HZ_POSEIDON_KW_T##T[0][0]); }
#else
#define HZ_PC(T) \
    template <> __device__ __forceinline__ const Fr* poseidon_k_global_w<T>() { return reinterpret_cast<const Fr*>(&HZ_POSEIDON_K_T##T[0][0]); }
#endif
HZ_PC(2) HZ_PC(3) HZ_PC(4) HZ_PC(5) HZ_PC(6) HZ_PC(7)
#undef HZ_PC

// Where the kernels read the Poseidon constants from. Every lane of a wavefront needs the same
// constant at the same time, so the block is either read straight from device memory through the
// scalar cache (s_load into SGPRs, which v_mad_u64_u32 takes as an operand: no VGPRs, no LDS), or
// staged once per workgroup into LDS and read as broadcasts (HZ_POSEIDON_LDS = 1).
#ifndef HZ_POSEIDON_LDS
#define HZ_POSEIDON_LDS 0
#endif
template <int T> constexpr size_t poseidon_lds_bytes() { return HZ_POSEIDON_LDS ? (size_t)poseidon_const_frs<T>() * sizeof(Fr) : 0; }

// Returns the width-T constant block; in LDS mode the whole block cooperates in copying it to
// `lds` (advanced past the block) and the caller must __syncthreads() before the first use.
template <int T, bool W = false>
__device__ __forceinline__ const Fr* poseidon_consts(uint32_t*& lds) {
#if HZ_POSEIDON_LDS
    constexpr int NW = poseidon_const_frs<T>() * 9;  // 32-bit words (an Fr is 9 limbs)
    const uint32_t* g = reinterpret_cast<const uint32_t*>(W ? poseidon_k_global_w<T>() : poseidon_k_global<T>());
    uint32_t* d = lds;
    for (int i = threadIdx.x; i < NW; i += blockDim.x) d[i] = g[i];
    lds += NW;
    return reinterpret_cast<const Fr*>(d);
#else
    return W ? poseidon_k_global_w<T>() : poseidon_k_global<T>();
#endif
}
// the block for a sink that stores the S-box witness (every witness kernel)
template <int T>
__device__ __forceinline__ const Fr* poseidon_consts_w(uint32_t*& lds) { return poseidon_consts<T, true>(lds); }

// ---- 32-byte element I/O (canonical form) ------------------------------------------------------
__device__ __forceinline__ Fc load_fr(const void* p) {
    const uint4* q = reinterpret_cast<const uint4*>(p);
    const uint4 a = q[0], b = q[1];
    Fc r;
    r.v[0] = a.x; r.v[1] = a.y; r.v[2] = a.z; r.v[3] = a.w;
    r.v[4] = b.x; r.v[5] = b.y; r.v[6] = b.z; r.v[7] = b.w;
    return r;
}
// witness elements are written once and (with few exceptions) never read again by the device. Marking the stores non-temporal
// (HZ_NT_STORES=1) was measured and is much SLOWER (k_smt 24 -> 37 ms, withdraw 1.90 -> 0.78 M/s): a lane writes 32 bytes, the
// wavefront's 2 KB per signal are combined in L2, which streaming stores bypass.
#ifndef HZ_NT_STORES
#define HZ_NT_STORES 0
#endif
typedef uint32_t hz_u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void store_fr(void* p, const Fc& r) {
#ifdef HZ_EXPERIMENT_SPLIT_STORES   // timing experiment only (WRONG placement): each store instruction of a wavefront writes 1 KB of full lines
    {
        const uint32_t lane = threadIdx.x & 63u;
        uint8_t* row = reinterpret_cast<uint8_t*>(p) - (size_t)lane * 32;
        *reinterpret_cast<uint4*>(row + lane * 16) = make_uint4(r.v[0], r.v[1], r.v[2], r.v[3]);
        *reinterpret_cast<uint4*>(row + 1024 + lane * 16) = make_uint4(r.v[4], r.v[5], r.v[6], r.v[7]);
        return;
    }
#endif
#if HZ_NT_STORES
    hz_u32x4* q = reinterpret_cast<hz_u32x4*>(p);
    hz_u32x4 a = {r.v[0], r.v[1], r.v[2], r.v[3]}, b = {r.v[4], r.v[5], r.v[6], r.v[7]};
    __builtin_nontemporal_store(a, q);
    __builtin_nontemporal_store(b, q + 1);
#else
    uint4* q = reinterpret_cast<uint4*>(p);
    q[0] = make_uint4(r.v[0], r.v[1], r.v[2], r.v[3]);
    q[1] = make_uint4(r.v[4], r.v[5], r.v[6], r.v[7]);
#endif
}

// ---- witness writer ---------------------------------------------------------------------------
// Signals of a section are stored signal-major: element (sig, unit) of instance `inst` lives at
// base + ((sig * n_units) + unit) * 32, so the 64 lanes of a wavefront (consecutive units) write
// 2 KiB of contiguous HBM per signal.
struct WitOut {
    uint8_t* base;     // first byte of this section for this instance
    uint32_t n_units;  // units (transactions, fee txs, witnesses) in the section
    uint32_t unit;     // this lane's unit
    __device__ __forceinline__ uint8_t* addr(uint32_t sig) const {
        return base + ((size_t)sig * n_units + unit) * 32;
    }
    __device__ __forceinline__ void put_mont(uint32_t sig, const Fr& m) const { store_fr(addr(sig), fr_to_canon(m)); }
    __device__ __forceinline__ void put_canon(uint32_t sig, const Fc& c) const { store_fr(addr(sig), c); }
    __device__ __forceinline__ void put_u64(uint32_t sig, uint64_t x) const {
        Fc c;
        c.v[0] = (uint32_t)x; c.v[1] = (uint32_t)(x >> 32);
        c.v[2] = c.v[3] = c.v[4] = c.v[5] = c.v[6] = c.v[7] = 0u;
        store_fr(addr(sig), c);
    }
    __device__ __forceinline__ void put_bit(uint32_t sig, uint32_t b) const { put_u64(sig, b & 1u); }
};

// Poseidon S-box sink that stores the three product signals of S-box k at sig0 + 3k + {0,1,2}.
#ifndef HZ_SINK_EARLY
#define HZ_SINK_EARLY 0
#endif
struct WitSboxSink {
    static constexpr bool kCanon = HZ_POSEIDON_CANON_SBOX != 0;   // the S-box hands over canonical values (29-bit limbs, < p)
    static constexpr bool kEarly = kCanon && HZ_SINK_EARLY != 0;   // one signal at a time, right after its product (poseidon_sbox)
    WitOut w;
    uint32_t sig0;
    __device__ __forceinline__ void put(int k, int j, const Fr& v) const {
#ifndef HZ_EXPERIMENT_NO_SINK_STORES
        w.put_canon(sig0 + 3 * k + j, fr_pack_canon(v));
#endif
    }
    __device__ __forceinline__ void operator()(int k, const Fr& x2, const Fr& x4, const Fr& x5) const {
#ifdef HZ_EXPERIMENT_NO_SINK_STORES   // timing experiment only (wrong witness): the arithmetic without its stores
        asm volatile("" :: "v"(x2.v[0]), "v"(x4.v[0]), "v"(x5.v[0]));
        return;
#endif
        if constexpr (kCanon) {
            w.put_canon(sig0 + 3 * k + 0, fr_pack_canon(x2));
            w.put_canon(sig0 + 3 * k + 1, fr_pack_canon(x4));
            w.put_canon(sig0 + 3 * k + 2, fr_pack_canon(x5));
        } else {
            w.put_mont(sig0 + 3 * k + 0, x2);
            w.put_mont(sig0 + 3 * k + 1, x4);
            w.put_mont(sig0 + 3 * k + 2, x5);
        }
    }
};

// ---- constraint failure record ------------------------------------------------------------------
// The reference stops at the first violated `===` (circom_runtime throws "Constraint doesn't
// match lhs != rhs"). Lanes evaluate everything; a failing lane (1) lowers `minkey`
// with atomicMin -- key = (instance, unit, constraint id), so the minimum is the first failure in
// evaluation order -- and (2) appends its operands to a bounded list. If the list overflowed and
// lost the minimum, the host re-enqueues with `filter` = minkey so only that lane appends.
#define HZ_ERR_CAP 1024
struct ErrRec {
    unsigned long long key;
    uint32_t lhs[8];
    uint32_t rhs[8];
};
struct ErrBuf {
    unsigned long long minkey;  // ~0ull = no failure
    unsigned long long filter;  // ~0ull = record everything
    unsigned int count;
    unsigned int pad;
    ErrRec rec[HZ_ERR_CAP];
};

__device__ __forceinline__ unsigned long long err_key(uint32_t inst, uint32_t unit, uint32_t cid) {
    return ((unsigned long long)inst << 40) | ((unsigned long long)unit << 16) | cid;
}

__device__ __noinline__ void report_fail(ErrBuf* e, uint32_t inst, uint32_t unit, uint32_t cid, const Fr& lhs_m, const Fr& rhs_m) {
    const unsigned long long key = err_key(inst, unit, cid);
    atomicMin(&e->minkey, key);
    if (e->filter != ~0ull && e->filter != key) return;
    const unsigned int slot = atomicAdd(&e->count, 1u);
    if (slot >= HZ_ERR_CAP) return;
    const Fc l = fr_to_canon(lhs_m), r = fr_to_canon(rhs_m);
    e->rec[slot].key = key;
    for (int i = 0; i < 8; i++) {
        e->rec[slot].lhs[i] = l.v[i];
        e->rec[slot].rhs[i] = r.v[i];
    }
}

}  // namespace hz
