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
#define HZ_PC(T) \
    template <> __device__ __forceinline__ const Fr* poseidon_k_global_w<T>() { return reinterpret_cast<const Fr*>(&HZ_POSEIDON_KW_T##T[0][0]); }
HZ_PC(2) HZ_PC(3) HZ_PC(4) HZ_PC(5) HZ_PC(6) HZ_PC(7)
#undef HZ_PC

// Where the kernels read the Poseidon constants from: every lane of a wavefront needs the same constant at the same time, so the
// block is read straight from device memory through the scalar cache (s_load into SGPRs, which v_mad_u64_u32 takes as an
// operand: no VGPRs, no LDS). Staging the block in LDS (north star) was measured in rounds 1-2 and was slower: LDS reads share the
// lgkmcnt counter with the scalar loads, and the reservation held k_hash4 to three workgroups per CU.
template <int T, bool W = false>
__device__ __forceinline__ const Fr* poseidon_consts() { return W ? poseidon_k_global_w<T>() : poseidon_k_global<T>(); }
// the block for a sink that stores the S-box witness (every witness kernel)
template <int T>
__device__ __forceinline__ const Fr* poseidon_consts_w() { return poseidon_consts<T, true>(); }

// ---- 32-byte element I/O (canonical form) ------------------------------------------------------
// Witness elements always live in device memory: the pointer is cast to the global address space explicitly, so that the access is a
// global_load / global_store even where the pointer reached the code through a by-reference argument of an out-of-line function
// (a generic pointer compiles to flat_*: same memory, but flat operations also count against lgkmcnt, the counter LDS reads and
// the scalar constant loads wait on).
typedef uint32_t hz_u32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(1))) hz_u32x4 hz_g_u32x4;
__device__ __forceinline__ Fc load_fr(const void* p) {
    const hz_g_u32x4* q = (const hz_g_u32x4*)p;
    const hz_u32x4 a = q[0], b = q[1];
    Fc r;
    r.v[0] = a.x; r.v[1] = a.y; r.v[2] = a.z; r.v[3] = a.w;
    r.v[4] = b.x; r.v[5] = b.y; r.v[6] = b.z; r.v[7] = b.w;
    return r;
}
// witness elements are written once and (with few exceptions) never read again by the device. Non-temporal stores were measured and
// are much SLOWER (k_smt 24 -> 37 ms, withdraw 1.90 -> 0.78 M/s; tools/microbench/mixbench.hip: 1.9 TB/s whatever the mix): a lane
// writes 32 bytes, the wavefront's 2 KB per signal are combined in L2, which streaming stores bypass.
__device__ __forceinline__ void store_fr(void* p, const Fc& r) {
    hz_g_u32x4* q = (hz_g_u32x4*)p;
    const hz_u32x4 a = {r.v[0], r.v[1], r.v[2], r.v[3]}, b = {r.v[4], r.v[5], r.v[6], r.v[7]};
    q[0] = a;
    q[1] = b;
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
struct WitSboxSink {
    static constexpr bool kCanon = true;   // the S-box hands over canonical values (29-bit limbs, < p)
    WitOut w;
    uint32_t sig0;
    __device__ __forceinline__ void operator()(int k, const Fr& x2, const Fr& x4, const Fr& x5) const {
        w.put_canon(sig0 + 3 * k + 0, fr_pack_canon(x2));
        w.put_canon(sig0 + 3 * k + 1, fr_pack_canon(x4));
        w.put_canon(sig0 + 3 * k + 2, fr_pack_canon(x5));
    }
};

// ---- constraint failure record ------------------------------------------------------------------
// The reference stops at the first violated `===` (circom_runtime throws "Constraint doesn't
// match lhs != rhs"). Lanes evaluate everything; a failing lane (1) lowers `minkey`
// with atomicMin -- key = (instance, unit, constraint id), so the minimum is the first failure in
// evaluation order -- and (2) appends its operands to a bounded list. If the list overflowed and
// lost the minimum, the host re-enqueues with `filter` = minkey so only that lane appends.
#define HZ_ERR_CAP 1024
// `filter` value of the second pass of hz_witness_failures: every instance's first failure, found by the first pass
// (inst_min), gets its operands written to inst_rec[instance]; nothing else is recorded.
#define HZ_FILTER_PER_INST (~1ull)
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
    unsigned long long* inst_min;  // [n_instances] lowest failing key of each instance (~0ull: none); every pass lowers it
    ErrRec* inst_rec;              // [n_instances] operands of those failures, written by the HZ_FILTER_PER_INST pass
    ErrRec rec[HZ_ERR_CAP];
};

__device__ __forceinline__ unsigned long long err_key(uint32_t inst, uint32_t unit, uint32_t cid) {
    return ((unsigned long long)inst << 40) | ((unsigned long long)unit << 16) | cid;
}

// Out of line (the failing path is cold), in two parts so that nothing is passed by reference: a reference -- or a second 36-byte
// struct, which the calling convention passes `byref` -- to a non-inlined function makes the caller keep the operand in scratch memory for
// the whole kernel (round 4: six such slots in k_smt). report_slot claims the record, report_operand converts and stores ONE operand that
// arrives in nine registers.
__device__ __noinline__ ErrRec* report_slot(ErrBuf* e, uint32_t inst, uint32_t unit, uint32_t cid) {
    const unsigned long long key = err_key(inst, unit, cid);
    atomicMin(&e->minkey, key);
    const unsigned long long filter = e->filter;
    ErrRec* dst;
    if (filter == HZ_FILTER_PER_INST) {
        // the lane that reports an instance's lowest key takes that instance's slot; a lane that reports the same key several
        // times (a loop over the elements of one `===` array) keeps the first, as the shared list does
        if (e->inst_min[inst] != key || atomicCAS(&e->inst_rec[inst].key, ~0ull, key) != ~0ull) return nullptr;
        dst = &e->inst_rec[inst];
    } else {
        atomicMin(&e->inst_min[inst], key);
        if (filter != ~0ull && filter != key) return nullptr;
        const unsigned int slot = atomicAdd(&e->count, 1u);
        if (slot >= HZ_ERR_CAP) return nullptr;
        dst = &e->rec[slot];
        dst->key = key;
    }
    return dst;
}
__device__ __noinline__ void report_operand(uint32_t* dst, const Fr v_m) {
    const Fc c = fr_to_canon(v_m);
    for (int i = 0; i < 8; i++) dst[i] = c.v[i];
}
__device__ __forceinline__ void report_fail(ErrBuf* e, uint32_t inst, uint32_t unit, uint32_t cid, const Fr& lhs_m, const Fr& rhs_m) {
    ErrRec* dst = report_slot(e, inst, unit, cid);
    if (dst) {
        report_operand(dst->lhs, lhs_m);
        report_operand(dst->rhs, rhs_m);
    }
}

}  // namespace hz
