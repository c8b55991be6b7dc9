// The witness in the COMPILER's variable order, produced on the device (SURVEY 8a' K8 `witness_layout`; the reference's
// calculateWitness returns w[] in circom's numbering, test/helpers/helpers.js:142,149, and the prove step reads it beside the
// .r1cs / zkey, tools/helpers/actions.js:132-170).
//
// The kernels of a step write a signal-major physical buffer (include/hz_layout.h: element (sig, unit) of a section at
// base + sig * n_units + unit, so that the 64 lanes of a wavefront store 2 KB of contiguous HBM per signal). A prover wants
// w[var] for var = 0..nVars-1 of ONE instance, contiguous. hz_symmap (formats.hip) knows, per variable, which stored signal it is
// or how it follows from stored signals. This file takes that map to the device ONCE (DevPlan) and turns an instance of the
// physical buffer into the variable-ordered vector with a handful of launches:
//
//   k_export_quads     stored variables of sections whose units-per-instance is a multiple of 4 (the transaction and fee sections):
//                      one lane reads one 128-byte LINE of the physical buffer -- four consecutive units of one signal -- and
//                      writes its four 32-byte elements to their four variables. Work items are ordered by their first variable,
//                      and circom numbers a component's signals consecutively, so consecutive lanes write consecutive variables:
//                      every store instruction of a wavefront covers 2 KB of contiguous output, every load a full line.
//   k_export_singles   stored variables of the other sections (global signals; HashInputs, whose unit is the instance): 32 bytes each;
//                      when ALL instances are exported together (instance = -1) a lane takes four instances of one variable, which
//                      are again one line of the physical buffer.
//   k_derive_poseidon  every signal INSIDE a Poseidon component (ark / mix / S-box inputs of circomlib 0.5.2 poseidon.circom, kept as
//                      variables by a compile without constraint reduction, test/rollup-main.test.js:52): one lane per component walks
//                      the dense permutation forward from the component's stored S-box products and drops the values the map asks for
//   k_derive_forms     linear forms / products / quotients / IsZero inputs over stored and earlier derived values (CSR tables), one lane
//                      per derived variable, one launch per dependency level
//   k_export_dvars     derived values -> their variables
//
// Nothing here computes a witness signal the kernels of the step did not already pin down: derived values are the linear (or
// constant-factor) consequences the unreduced R1CS states, evaluated in the same field arithmetic as everything else (fr.h).
#include <hip/hip_runtime.h>
#include <stdlib.h>
#include <stdio.h>
#include <string.h>
#include <algorithm>
#include <array>
#include <map>
#include <unordered_map>
#include <vector>
#include "hostutil.h"
#include "ctx_internal.h"
#include "symmap.h"
#include "devcommon.h"

using namespace hz;

namespace hzexp {

static const uint32_t NONE32 = 0xFFFFFFFFu;
static const uint32_t SRC_DERIVED = 0x80000000u;
enum { COEF_PLUS1 = 0, COEF_MINUS1 = 1 };

struct SecTab { SecMap sec[4]; uint32_t nsec; };
// the sections that have lines (units per instance a multiple of four), ascending: lines >= first_line[i]; a section's first element
// need not be line-aligned: element = 4 * line + rem
struct LineTab { uint32_t first_line[4], upi[4], rem[4]; uint32_t nsec; };

struct PosGroup {
    uint64_t n = 0;
    DevBuf first, loff;   // per component: virtual index of sigmaF[0][0].in2; [n + 1] offsets into pl_slot / pl_k
    DevBuf C, M;          // plain round constants / MDS matrix of this width, Montgomery Fr
};
struct DevPlan {
    // what the plan was made for
    int device = -1;
    SecTab st{};
    uint64_t per_instance = 0, total = 0;
    uint32_t n_inst = 0;
    uint64_t nvars = 0;
    // stored variables
    uint64_t n_quads = 0, n_s1 = 0, n_sx = 0;
    DevBuf q_src, q_dst;          // u32[n_quads] physical line of instance 0 (in units of 4 elements), uint4[n_quads] the four variables
    LineTab lt{};
    DevBuf s1_src, s1_dst;        // singles of sections with one unit per instance
    DevBuf sx_src, sx_dst;        // every other single
    // derived variables
    uint64_t D = 0;               // slots of the derived-value buffer (= derived variables of the map)
    uint64_t n_dvars = 0;
    DevBuf dv_v, dv_k;            // out[dv_v[i]] = dval[dv_k[i]]
    DevBuf kind, arg;             // per derived variable: DV_*, form index (or the stored index of IsZero.inv)
    DevBuf form_off, form_c0, term_coef, term_src, pool;
    std::vector<std::pair<uint64_t, uint64_t>> levels;   // (first, count) into lvl_k, one launch each
    DevBuf lvl_k;
    PosGroup pg[6];
    DevBuf pl_slot, pl_k;
    DevBuf phys0, istride;        // hz_symmap_dev_index
    uint64_t bytes = 0;           // device bytes of the tables
};
void devplan_free(DevPlan* p) { delete p; }

// ---- device side ------------------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ uint64_t v2p(const SecTab& t, uint32_t v, uint32_t inst, uint32_t* n_units = nullptr) {
    uint64_t vbase = t.sec[0].vbase, base = t.sec[0].base;
    uint32_t upi = t.sec[0].upi, nu = t.sec[0].n_units;
#pragma unroll
    for (int i = 1; i < 4; i++)
        if ((uint32_t)i < t.nsec && (uint64_t)v >= t.sec[i].vbase) { vbase = t.sec[i].vbase; base = t.sec[i].base; upi = t.sec[i].upi; nu = t.sec[i].n_units; }
    const uint32_t rel = v - (uint32_t)vbase;
    const uint32_t s = rel / upi, u = rel - s * upi;
    if (n_units) *n_units = nu;
    return base + (uint64_t)s * nu + (uint64_t)inst * upi + u;
}

struct ExpArgs {
    const hz_u32x4* wit;
    hz_u32x4* out;          // instance blockIdx.y of the launch goes to out + blockIdx.y * out_stride elements
    uint64_t out_stride;    // elements
    uint32_t inst0;
    SecTab st;
};
typedef __attribute__((address_space(1))) const hz_u32x4 g_cu4;
typedef __attribute__((address_space(1))) hz_u32x4 g_u4;

// Eight lanes per line: a wavefront's load instruction reads eight whole 128-byte lines (16 bytes per lane, the pattern of a copy
// kernel), its store instruction writes, for each of the four units of those lines, eight consecutive variables = 256 contiguous
// bytes. (A first version gave each lane a line of its own: eight load instructions per wavefront all touching the same 64 lines,
// 2.25 ms for the headline batch's transaction section against 1.x ms.) `line0` = the line's physical position for instance 0 in
// units of four elements: no division on the way to the address.
__device__ __forceinline__ void export_quads(const ExpArgs& a, const LineTab& lt, const uint32_t* __restrict__ line0, const uint32_t* __restrict__ dst, uint64_t n, uint32_t bx, uint32_t nbx) {
    const uint32_t inst = a.inst0 + blockIdx.y;
    g_u4* out = (g_u4*)(a.out + 2 * (uint64_t)blockIdx.y * a.out_stride);
    g_cu4* wit = (g_cu4*)a.wit;
    const uint32_t lane = threadIdx.x & 63u, part = lane & 7u, sub = lane >> 3;
    const uint64_t wave = ((uint64_t)bx * blockDim.x + threadIdx.x) >> 6, nwaves = ((uint64_t)nbx * blockDim.x) >> 6;
    for (uint64_t q0 = wave * 64; q0 < n; q0 += nwaves * 64) {
        hz_u32x4 x[8];
        uint32_t d[8];
#pragma unroll
        for (int st = 0; st < 8; st++) {
            const uint64_t q = q0 + (uint64_t)st * 8 + sub;
            d[st] = NONE32;
            if (q < n) {
                const uint32_t l0 = line0[q];
                uint32_t upi = lt.upi[0], rem = lt.rem[0];
#pragma unroll
                for (int i = 1; i < 4; i++)
                    if ((uint32_t)i < lt.nsec && l0 >= lt.first_line[i]) { upi = lt.upi[i]; rem = lt.rem[i]; }
                d[st] = dst[4 * q + (part >> 1)];
                x[st] = wit[8 * (uint64_t)l0 + 2 * ((uint64_t)inst * upi + rem) + part];
            }
        }
#pragma unroll
        for (int st = 0; st < 8; st++)
            if (d[st] != NONE32) out[2 * (uint64_t)d[st] + (part & 1u)] = x[st];
    }
}
// two lanes per element
template <bool UPI1>
__device__ __forceinline__ void export_singles(const ExpArgs& a, const uint32_t* __restrict__ src, const uint32_t* __restrict__ dst, uint64_t n, uint32_t bx, uint32_t nbx) {
    const uint32_t inst = a.inst0 + blockIdx.y;
    g_u4* out = (g_u4*)(a.out + 2 * (uint64_t)blockIdx.y * a.out_stride);
    g_cu4* wit = (g_cu4*)a.wit;
    const uint32_t half = threadIdx.x & 1u;
    const uint64_t t0 = ((uint64_t)bx * blockDim.x + threadIdx.x) >> 1, nt = ((uint64_t)nbx * blockDim.x) >> 1;
    for (uint64_t i0 = t0; i0 < n; i0 += 4 * nt) {
        hz_u32x4 x[4];
        uint32_t d[4];
#pragma unroll
        for (int st = 0; st < 4; st++) {
            const uint64_t i = i0 + (uint64_t)st * nt;
            d[st] = NONE32;
            if (i < n) {
                d[st] = dst[i];
                uint64_t p;
                if (UPI1) {   // the section with one unit per instance (the last one of the layout that has it): no division
                    const uint32_t v = src[i];
                    uint64_t vbase = a.st.sec[0].vbase, base = a.st.sec[0].base;
                    uint32_t nu = a.st.sec[0].n_units;
#pragma unroll
                    for (int k = 1; k < 4; k++)
                        if ((uint32_t)k < a.st.nsec && (uint64_t)v >= a.st.sec[k].vbase) { vbase = a.st.sec[k].vbase; base = a.st.sec[k].base; nu = a.st.sec[k].n_units; }
                    p = base + (uint64_t)(v - (uint32_t)vbase) * nu + inst;
                } else p = v2p(a.st, src[i], inst);
                x[st] = wit[2 * p + half];
            }
        }
#pragma unroll
        for (int st = 0; st < 4; st++)
            if (d[st] != NONE32) out[2 * (uint64_t)d[st] + half] = x[st];
    }
}
// ONE launch for every stored variable of an instance: the first bq blocks take the lines, the next b1 the elements of the sections
// with one unit per instance (scattered 32-byte reads: latency, which the streaming blocks beside them hide), the last bx the rest
struct StoredLists { const uint32_t* q_line0; const uint32_t* q_dst; uint64_t nq; const uint32_t* s1_src; const uint32_t* s1_dst; uint64_t n1; const uint32_t* sx_src; const uint32_t* sx_dst; uint64_t nx;
                     uint32_t bq, b1, bx; };
__global__ void __launch_bounds__(256) k_export_stored(const ExpArgs a, const LineTab lt, const StoredLists l) {
    const uint32_t b = blockIdx.x;
    if (b < l.bq) export_quads(a, lt, l.q_line0, l.q_dst, l.nq, b, l.bq);
    else if (b < l.bq + l.b1) export_singles<true>(a, l.s1_src, l.s1_dst, l.n1, b - l.bq, l.b1);
    else export_singles<false>(a, l.sx_src, l.sx_dst, l.nx, b - l.bq - l.b1, l.bx);
}
__global__ void __launch_bounds__(256) k_export_singles_only(const ExpArgs a, const uint32_t* __restrict__ src, const uint32_t* __restrict__ dst, uint64_t n) {
    export_singles<false>(a, src, dst, n, blockIdx.x, gridDim.x);
}
// sections with ONE unit per instance, four instances per lane: instance inst0 + 4 * blockIdx.y + j of variable dst goes to
// out + (4 * blockIdx.y + j) * out_stride; the four sources are one 128-byte line
__global__ void __launch_bounds__(256) k_export_singles_x4(const ExpArgs a, const uint32_t* __restrict__ src, const uint32_t* __restrict__ dst, uint64_t n) {
    const uint32_t inst = a.inst0 + 4 * blockIdx.y;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
        g_cu4* p = (g_cu4*)(a.wit + 2 * v2p(a.st, src[i], inst));
        const hz_u32x4 x0 = p[0], x1 = p[1], x2 = p[2], x3 = p[3], x4 = p[4], x5 = p[5], x6 = p[6], x7 = p[7];
        const uint64_t d = dst[i];
        g_u4* o = (g_u4*)(a.out + 2 * ((uint64_t)(4 * blockIdx.y) * a.out_stride + d));
        o[0] = x0; o[1] = x1;
        o += 2 * a.out_stride; o[0] = x2; o[1] = x3;
        o += 2 * a.out_stride; o[0] = x4; o[1] = x5;
        o += 2 * a.out_stride; o[0] = x6; o[1] = x7;
    }
}

// derived values: canonical 32-byte elements, slot k of instance blockIdx.y of the launch at dval + (blockIdx.y * D + k)
struct DrvArgs {
    const hz_u32x4* wit;
    hz_u32x4* dval;
    uint64_t D;
    uint32_t inst0;
    SecTab st;
    const uint8_t* kind; const uint32_t* arg;
    const uint32_t* form_off; const uint32_t* form_c0; const uint32_t* term_coef; const uint32_t* term_src;
    const Fr* pool;           // per coefficient: [2 * id] Montgomery form, [2 * id + 1] the plain number in limbs
    const uint32_t* lvl_k;
};
__device__ __forceinline__ Fr r2_const() {
    Fr r;
#pragma unroll
    for (int i = 0; i < 9; i++) r.v[i] = fr_r2(i);
    return r;
}
// a stored signal or an earlier derived value as the plain number in 29-bit limbs (< p)
__device__ __forceinline__ Fr drv_operand(const DrvArgs& a, uint32_t src, uint32_t inst) {
    const hz_u32x4* p = (src & SRC_DERIVED) ? a.dval + 2 * ((uint64_t)blockIdx.y * a.D + (src & ~SRC_DERIVED)) : a.wit + 2 * v2p(a.st, src, inst);
    return fr_unpack(load_fr(p));
}
__device__ __noinline__ Fr drv_form(const DrvArgs& a, uint32_t f, uint32_t inst) {
    Fr acc = fr_zero();
    const uint32_t c0 = a.form_c0[f];
    if (c0 != NONE32) acc = a.pool[2 * (uint64_t)c0 + 1];
    for (uint32_t t = a.form_off[f]; t < a.form_off[f + 1]; t++) {
        const Fr x = drv_operand(a, a.term_src[t], inst);
        const uint32_t c = a.term_coef[t];
        if (c == COEF_PLUS1) acc = fr_add(acc, x);
        else if (c == COEF_MINUS1) acc = fr_sub(acc, x);
        else acc = fr_add(acc, fr_mul(a.pool[2 * (uint64_t)c], x));   // Montgomery coefficient times a plain number: a plain number
    }
    return acc;   // [0, 2p)
}
__global__ void __launch_bounds__(64) k_derive_forms(const DrvArgs a, uint64_t first, uint64_t count) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    const uint32_t inst = a.inst0 + blockIdx.y;
    const uint32_t k = a.lvl_k[first + i];
    const uint32_t kind = a.kind[k], arg = a.arg[k];
    Fr r;
    if (kind == DV_ISZERO_IN) {   // inv <-- in != 0 ? 1 / in : 0  =>  in = inv != 0 ? 1 / inv : 0
        const Fr inv = fr_unpack(load_fr(a.wit + 2 * v2p(a.st, arg, inst)));
        r = fr_canon_limbs(fr_inv(fr_mul(inv, r2_const())));
    } else if (kind == DV_LINEAR) {
        r = drv_form(a, arg, inst);
    } else {
        const Fr A = drv_form(a, arg, inst), B = drv_form(a, arg + 1, inst), C = drv_form(a, arg + 2, inst);
        if (kind == DV_PRODUCT) r = fr_add(fr_mul(fr_mul(A, r2_const()), B), C);
        else r = fr_add(fr_mul(A, fr_inv(fr_mul(B, r2_const()))), C);   // 0 / 0 = 0: inverse(0) = 0
    }
    store_fr(a.dval + 2 * ((uint64_t)blockIdx.y * a.D + k), fr_pack_canon(fr_cond_sub_p(r)));
}

struct PosArgs {
    const hz_u32x4* wit;
    hz_u32x4* dval;
    uint64_t D, n;
    uint32_t inst0;
    SecTab st;
    const uint32_t* first; const uint32_t* loff;
    const uint16_t* pl_slot; const uint32_t* pl_k;
    const Fr* C; const Fr* M;
};
struct PosEmit {
    hz_u32x4* dval;       // this instance's slots
    const uint16_t* slot; const uint32_t* k;
    uint32_t cur, end;
};
// the value of trace slot `s` (Montgomery form) goes to every derived variable that names it; slots arrive in ascending order
__device__ __noinline__ void pos_emit(PosEmit& e, uint32_t s, const Fr& v) {
    if (e.cur >= e.end || e.slot[e.cur] != s) return;
    const Fc c = fr_to_canon(v);
    do { store_fr(e.dval + 2 * (uint64_t)e.k[e.cur], c); e.cur++; } while (e.cur < e.end && e.slot[e.cur] == s);
}
__device__ __forceinline__ int pos_sbox_dev(int t, int rp, int i, int j) {
    if (i < 4) return i * t + j;
    if (i < 4 + rp) return j == 0 ? 4 * t + (i - 4) : -1;
    return 4 * t + rp + (i - 4 - rp) * t + j;
}
// One lane per Poseidon component: the dense trace of circomlib's template, forward from the stored S-box products (derived.h is the
// host statement of the same walk). Slot of (what, round i, lane j) = (4 i + what) T + j, what = PW_ARK_IN .. PW_MIX_OUT.
template <int T>
__global__ void __launch_bounds__(64) k_derive_poseidon(const PosArgs a) {
    const uint64_t c = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= a.n) return;
    constexpr int RP = PoseidonCfg<T>::RP, R = 8 + RP;
    const uint32_t inst = a.inst0 + blockIdx.y;
    uint32_t nu = 0;
    const uint64_t p0 = v2p(a.st, a.first[c], inst, &nu);
    PosEmit e{a.dval + 2 * (uint64_t)blockIdx.y * a.D, a.pl_slot, a.pl_k, a.loff[c], a.loff[c + 1]};
    auto sig = [&](int k, int which) { return fr_from_canon(load_fr(a.wit + 2 * (p0 + (uint64_t)(3 * k + which) * nu))); };
    Fr st[T], mi[T];
#pragma unroll 1
    for (int j = 0; j < T; j++) {   // the S-box inputs of round 0: x = x^5 / x^4 (0 when x^4 = 0)
        const Fr x4 = sig(j, 1);
        const Fr x = fr_mul(sig(j, 2), fr_inv(x4));
        st[j] = x;
    }
#pragma unroll 1
    for (int j = 0; j < T; j++) pos_emit(e, (uint32_t)(PW_ARK_IN * T + j), fr_sub(st[j], a.C[j]));
#pragma unroll 1
    for (int i = 0; i < R; i++) {
        if (i > 0) {
#pragma unroll 1
            for (int j = 0; j < T; j++) pos_emit(e, (uint32_t)((4 * i + PW_ARK_IN) * T + j), st[j]);
#pragma unroll 1
            for (int j = 0; j < T; j++) st[j] = fr_add(st[j], a.C[(size_t)T * i + j]);
        }
#pragma unroll 1
        for (int j = 0; j < T; j++) pos_emit(e, (uint32_t)((4 * i + PW_ARK_OUT) * T + j), st[j]);
#pragma unroll 1
        for (int j = 0; j < T; j++) {
            const int k = pos_sbox_dev(T, RP, i, j);
            mi[j] = k >= 0 ? sig(k, 2) : st[j];
            pos_emit(e, (uint32_t)((4 * i + PW_MIX_IN) * T + j), mi[j]);
        }
#pragma unroll 1
        for (int r = 0; r < T; r++) {
            Fr acc = fr_mul(a.M[r * T], mi[0]);
#pragma unroll 1
            for (int j = 1; j < T; j++) acc = fr_add(acc, fr_mul(a.M[r * T + j], mi[j]));
            st[r] = acc;
        }
#pragma unroll 1
        for (int j = 0; j < T; j++) pos_emit(e, (uint32_t)((4 * i + PW_MIX_OUT) * T + j), st[j]);
    }
}
__global__ void __launch_bounds__(256) k_export_dvars(const hz_u32x4* dval, uint64_t D, hz_u32x4* out, uint64_t out_stride, const uint32_t* __restrict__ v, const uint32_t* __restrict__ k, uint64_t n) {
    const hz_u32x4* dv = dval + 2 * (uint64_t)blockIdx.y * D;
    hz_u32x4* o = out + 2 * (uint64_t)blockIdx.y * out_stride;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
        const uint64_t s = k[i], d = v[i];
        const hz_u32x4 x0 = dv[2 * s], x1 = dv[2 * s + 1];
        o[2 * d] = x0; o[2 * d + 1] = x1;
    }
}
// physical element of every stored variable for instance 0 and what an instance adds to it (hz_symmap_dev_index)
__global__ void k_phys_index(SecTab st, const uint64_t* __restrict__ index, uint64_t n, uint64_t* phys0, uint32_t* istride) {
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
        const uint64_t v = index[i];
        if (v & DERIVED_FLAG) { phys0[i] = v; istride[i] = 0; continue; }
        phys0[i] = v2p(st, (uint32_t)v, 0);
        istride[i] = (uint32_t)(v2p(st, (uint32_t)v, 1) - phys0[i]);
    }
}

// ---- host side: the plan ------------------------------------------------------------------------------------------------------------------
template <class T>
static hipError_t upload(DevBuf& b, const std::vector<T>& v, uint64_t& bytes) {
    hipError_t e = b.alloc(std::max<size_t>(v.size(), 1) * sizeof(T));
    if (e == hipSuccess && !v.empty()) e = hipMemcpy(b.p, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice);
    bytes += b.bytes;
    return e;
}
static Fr fr_from_hostf(const hzh::F& f, bool montgomery) {
    Fc c;
    hzh::f_to_canon(f, (uint8_t*)c.v);
    return montgomery ? fr_from_canon(c) : fr_unpack(c);
}
struct FKeyHash { size_t operator()(const std::array<uint64_t, 4>& k) const { return (size_t)(k[0] * 0x9E3779B97F4A7C15ull ^ k[1] ^ (k[2] << 1) ^ (k[3] * 31)); } };

static hz_status build_plan(const CtxGeom& g, const hz_symmap* m, DevPlan** out) {
    std::unique_ptr<DevPlan> P(new DevPlan());
    P->device = g.device; P->per_instance = g.per_instance; P->total = g.total; P->n_inst = g.n_inst; P->nvars = m->index.size();
    P->st.nsec = g.nsec;
    for (uint32_t i = 0; i < g.nsec; i++) P->st.sec[i] = g.sec[i];
    for (uint32_t i = 0; i < g.nsec; i++)
        if (g.sec[i].upi % 4 == 0) {
            LineTab& lt = P->lt;
            lt.first_line[lt.nsec] = (uint32_t)(g.sec[i].base / 4); lt.upi[lt.nsec] = g.sec[i].upi; lt.rem[lt.nsec] = (uint32_t)(g.sec[i].base % 4);
            lt.nsec++;
        }
    if (g.total / 4 >= NONE32) return set_err(HZ_ERR_ARG, "witness export: a physical buffer of %llu elements does not fit the 32-bit line numbers", (unsigned long long)g.total);
    const uint64_t nv = m->index.size(), D = m->derived.size();
    if (g.per_instance >= SRC_DERIVED || nv >= NONE32 || D >= SRC_DERIVED)
        return set_err(HZ_ERR_ARG, "witness export: %llu signals / %llu variables / %llu derived do not fit the 32-bit tables", (unsigned long long)g.per_instance, (unsigned long long)nv, (unsigned long long)D);
    auto section_of = [&](uint64_t v) { uint32_t si = g.nsec - 1; while (si > 0 && g.sec[si].vbase > v) si--; return si; };
    // -- stored variables: lines of four units where the section allows it, single elements elsewhere; in ascending order of the
    //    (first) variable, so that the lanes of a wavefront write consecutive variables
    std::vector<uint32_t> head((size_t)g.per_instance, NONE32);   // stored index -> its first variable
    for (uint64_t v = 0; v < nv; v++) {
        const uint64_t idx = m->index[v];
        if (idx == ~0ull) return set_err(HZ_ERR_INPUT, "witness export: variable %llu of the map is not resolved", (unsigned long long)v);
        if (idx & DERIVED_FLAG) { if ((idx & ~DERIVED_FLAG) >= D) return set_err(HZ_ERR_INPUT, "witness export: damaged map"); continue; }
        if (idx >= g.per_instance) return set_err(HZ_ERR_INPUT, "witness export: variable %llu names signal %llu of %llu", (unsigned long long)v, (unsigned long long)idx, (unsigned long long)g.per_instance);
        if (head[(size_t)idx] == NONE32) head[(size_t)idx] = (uint32_t)v;
    }
    std::vector<uint32_t> q_src, s1_src, s1_dst, sx_src, sx_dst, dv_v, dv_k;
    std::vector<uint4> q_dst;
    std::vector<uint8_t> quad_done((size_t)(g.per_instance / 4 + g.nsec + 1), 0);   // per (section-relative) line
    std::vector<uint64_t> line0(g.nsec, 0);   // first line number of each section (a section's lines never straddle into the next one)
    for (uint32_t si = 1; si < g.nsec; si++) line0[si] = line0[si - 1] + (g.sec[si].vbase - g.sec[si - 1].vbase) / 4 + 1;
    for (uint64_t v = 0; v < nv; v++) {
        const uint64_t idx = m->index[v];
        if (idx & DERIVED_FLAG) { dv_v.push_back((uint32_t)v); dv_k.push_back((uint32_t)(idx & ~DERIVED_FLAG)); continue; }
        const uint32_t si = section_of(idx);
        const SecMap& s = g.sec[si];
        const bool first_of_idx = head[(size_t)idx] == (uint32_t)v;
        if (first_of_idx && s.upi % 4 == 0) {
            const uint64_t rel = idx - s.vbase;
            const uint64_t line = line0[si] + rel / 4;
            if (quad_done[(size_t)line]) continue;   // written with the line's first variable
            quad_done[(size_t)line] = 1;
            const uint64_t q0 = idx - (rel & 3);
            const uint64_t p0 = s.base + ((q0 - s.vbase) / s.upi) * (uint64_t)s.n_units + (q0 - s.vbase) % s.upi;   // instance 0
            q_src.push_back((uint32_t)((p0 - s.base % 4) / 4));
            q_dst.push_back(uint4{head[(size_t)q0], head[(size_t)q0 + 1], head[(size_t)q0 + 2], head[(size_t)q0 + 3]});
        } else if (s.upi == 1) { s1_src.push_back((uint32_t)idx); s1_dst.push_back((uint32_t)v);
        } else { sx_src.push_back((uint32_t)idx); sx_dst.push_back((uint32_t)v); }
    }
    std::vector<uint32_t>().swap(head);
    std::vector<uint8_t>().swap(quad_done);
    P->n_quads = q_src.size(); P->n_s1 = s1_src.size(); P->n_sx = sx_src.size(); P->n_dvars = dv_v.size(); P->D = D;
    HZ_HIP(hipSetDevice(g.device));
    HZ_HIP(upload(P->q_src, q_src, P->bytes)); HZ_HIP(upload(P->q_dst, q_dst, P->bytes));
    HZ_HIP(upload(P->s1_src, s1_src, P->bytes)); HZ_HIP(upload(P->s1_dst, s1_dst, P->bytes));
    HZ_HIP(upload(P->sx_src, sx_src, P->bytes)); HZ_HIP(upload(P->sx_dst, sx_dst, P->bytes));
    HZ_HIP(upload(P->dv_v, dv_v, P->bytes)); HZ_HIP(upload(P->dv_k, dv_k, P->bytes));
    if (D == 0) { *out = P.release(); return HZ_OK; }

    // -- derived variables: which are needed, at which dependency level, and their tables
    std::vector<uint8_t> used((size_t)D, 0);
    {
        std::vector<uint32_t> stack(dv_k.begin(), dv_k.end());
        while (!stack.empty()) {
            const uint32_t k = stack.back();
            stack.pop_back();
            if (used[k]) continue;
            used[k] = 1;
            const DerivedVar& d = m->derived[k];
            if (d.kind == DV_POSEIDON || d.kind == DV_ISZERO_IN) continue;
            for (uint32_t f = d.lin; f < d.lin + (d.kind == DV_LINEAR ? 1u : 3u); f++) {
                if (f >= m->lins.size()) return set_err(HZ_ERR_INPUT, "witness export: damaged map (form %u)", f);
                for (const auto& tm : m->lins[f].terms)
                    if (tm.second & DERIVED_FLAG) {
                        const uint64_t j = tm.second & ~DERIVED_FLAG;
                        if (j >= k) return set_err(HZ_ERR_INPUT, "witness export: derived variable %u refers to a later one", k);
                        stack.push_back((uint32_t)j);
                    } else if (tm.second >= g.per_instance) return set_err(HZ_ERR_INPUT, "witness export: damaged map (term)");
            }
        }
    }
    std::vector<uint8_t> kind((size_t)D, 0);
    std::vector<uint32_t> arg((size_t)D, 0), level((size_t)D, 0);
    std::vector<uint32_t> form_off(1, 0), form_c0, term_coef, term_src;
    std::vector<Fr> pool;
    std::unordered_map<std::array<uint64_t, 4>, uint32_t, FKeyHash> pool_id;
    auto coef_id = [&](const hzh::F& f) -> uint32_t {
        std::array<uint64_t, 4> key;
        memcpy(key.data(), f.v, 32);
        auto it = pool_id.find(key);
        if (it != pool_id.end()) return it->second;
        const uint32_t id = (uint32_t)(pool.size() / 2);
        pool.push_back(fr_from_hostf(f, true));
        pool.push_back(fr_from_hostf(f, false));
        pool_id.emplace(key, id);
        return id;
    };
    coef_id(hzh::f_one());                                  // COEF_PLUS1
    coef_id(hzh::f_sub(hzh::f_zero(), hzh::f_one()));       // COEF_MINUS1
    // Poseidon components: (first stored signal) -> the trace slots the map names
    struct PosReq { uint64_t first; uint16_t slot; uint32_t k; uint8_t t; };
    std::vector<PosReq> preq;
    uint32_t max_level = 0;
    for (uint64_t k = 0; k < D; k++) {
        if (!used[k]) continue;
        const DerivedVar& d = m->derived[k];
        kind[k] = d.kind;
        if (d.kind == DV_POSEIDON) {
            if (d.t < 2 || d.t > 7 || d.what > PW_MIX_OUT || d.round >= hzderived::pos_rounds(d.t) || d.lane >= d.t || d.first >= g.per_instance)
                return set_err(HZ_ERR_INPUT, "witness export: damaged map (Poseidon record %llu)", (unsigned long long)k);
            const uint32_t si = section_of(d.first);
            const uint64_t last = d.first + (uint64_t)(3 * hzderived::pos_nsbox(d.t) - 1) * d.stride;
            if (d.stride != g.sec[si].upi || last >= g.per_instance || section_of(last) != si)
                return set_err(HZ_ERR_INPUT, "witness export: Poseidon block at %llu does not lie in one section of this layout", (unsigned long long)d.first);
            preq.push_back(PosReq{d.first, (uint16_t)((4 * d.round + d.what) * d.t + d.lane), (uint32_t)k, d.t});
            continue;
        }
        if (d.kind == DV_ISZERO_IN) {
            if (d.first >= g.per_instance) return set_err(HZ_ERR_INPUT, "witness export: damaged map (IsZero record)");
            arg[k] = (uint32_t)d.first;
            continue;
        }
        arg[k] = (uint32_t)form_c0.size();
        uint32_t lv = 0;
        for (uint32_t f = d.lin; f < d.lin + (d.kind == DV_LINEAR ? 1u : 3u); f++) {
            const LinForm& lf = m->lins[f];
            form_c0.push_back(hzh::f_is_zero(lf.c0) ? NONE32 : coef_id(lf.c0));
            for (const auto& tm : lf.terms) {
                term_coef.push_back(coef_id(tm.first));
                if (tm.second & DERIVED_FLAG) {
                    const uint32_t j = (uint32_t)(tm.second & ~DERIVED_FLAG);
                    term_src.push_back(SRC_DERIVED | j);
                    lv = std::max(lv, level[j] + 1);
                } else term_src.push_back((uint32_t)tm.second);
            }
            if (term_src.size() >= NONE32) return set_err(HZ_ERR_ARG, "witness export: more than 2^32 terms");
            form_off.push_back((uint32_t)term_src.size());
        }
        level[k] = lv;
        max_level = std::max(max_level, lv);
    }
    // level lists (Poseidon components run before level 0 in their own kernels; IsZero inputs and forms over stored signals are level 0)
    {
        std::vector<uint64_t> cnt(max_level + 2, 0);
        for (uint64_t k = 0; k < D; k++)
            if (used[k] && kind[k] != DV_POSEIDON) cnt[level[k] + 1]++;
        for (uint32_t l = 0; l <= max_level; l++) cnt[l + 1] += cnt[l];
        std::vector<uint32_t> lvl_k((size_t)cnt[max_level + 1]);
        for (uint32_t l = 0; l <= max_level; l++)
            if (cnt[l + 1] > cnt[l]) P->levels.push_back({cnt[l], cnt[l + 1] - cnt[l]});
        std::vector<uint64_t> fill(cnt.begin(), cnt.end() - 1);
        for (uint64_t k = 0; k < D; k++)
            if (used[k] && kind[k] != DV_POSEIDON) lvl_k[(size_t)fill[level[k]]++] = (uint32_t)k;
        HZ_HIP(upload(P->lvl_k, lvl_k, P->bytes));
    }
    HZ_HIP(upload(P->kind, kind, P->bytes)); HZ_HIP(upload(P->arg, arg, P->bytes));
    HZ_HIP(upload(P->form_off, form_off, P->bytes)); HZ_HIP(upload(P->form_c0, form_c0, P->bytes));
    HZ_HIP(upload(P->term_coef, term_coef, P->bytes)); HZ_HIP(upload(P->term_src, term_src, P->bytes));
    HZ_HIP(upload(P->pool, pool, P->bytes));
    // Poseidon components by width, ordered by their place in the buffer (consecutive lanes: consecutive units of one block)
    std::sort(preq.begin(), preq.end(), [](const PosReq& x, const PosReq& y) { return x.t != y.t ? x.t < y.t : x.first != y.first ? x.first < y.first : x.slot != y.slot ? x.slot < y.slot : x.k < y.k; });
    std::vector<uint16_t> pl_slot(preq.size());
    std::vector<uint32_t> pl_k(preq.size());
    for (size_t i = 0; i < preq.size(); i++) { pl_slot[i] = preq[i].slot; pl_k[i] = preq[i].k; }
    HZ_HIP(upload(P->pl_slot, pl_slot, P->bytes)); HZ_HIP(upload(P->pl_k, pl_k, P->bytes));
    for (size_t i = 0; i < preq.size();) {
        const int t = preq[i].t;
        std::vector<uint32_t> first, loff;
        size_t j = i;
        for (; j < preq.size() && preq[j].t == t; j++)
            if (j == i || preq[j].first != preq[j - 1].first) { first.push_back((uint32_t)preq[j].first); loff.push_back((uint32_t)j); }
        loff.push_back((uint32_t)j);
        PosGroup& G = P->pg[t - 2];
        G.n = first.size();
        HZ_HIP(upload(G.first, first, P->bytes)); HZ_HIP(upload(G.loff, loff, P->bytes));
        const hzderived::PosTab& tb = hzderived::pos_tab(t);
        std::vector<Fr> C(tb.C.size()), M(tb.M.size());
        for (size_t q = 0; q < C.size(); q++) C[q] = fr_from_hostf(tb.C[q], true);
        for (size_t q = 0; q < M.size(); q++) M[q] = fr_from_hostf(tb.M[q], true);
        HZ_HIP(upload(G.C, C, P->bytes)); HZ_HIP(upload(G.M, M, P->bytes));
        i = j;
    }
    *out = P.release();
    return HZ_OK;
}

static hz_status get_plan(hz_ctx* ctx, const hz_symmap* m, CtxGeom& g, DevPlan** plan) {
    if (!ctx || !m) return set_err(HZ_ERR_ARG, "witness export: null argument");
    if (!m->unresolved.empty())
        return set_err(HZ_ERR_INPUT, "witness export: %zu of %zu variables of the .sym are not stored by this layout (first: variable %llu, %s)", m->unresolved.size(), m->index.size(),
                       (unsigned long long)m->unresolved[0], m->first_label[0].c_str());
    ctx_geometry(ctx, g);
    std::lock_guard<std::mutex> lk(m->dev_mu);
    DevPlan* P = nullptr;
    for (DevPlan* Q : m->devs) {
        bool same = Q->device == g.device && Q->per_instance == g.per_instance && Q->total == g.total && Q->n_inst == g.n_inst && Q->st.nsec == g.nsec && Q->nvars == m->index.size();
        for (uint32_t i = 0; same && i < g.nsec; i++) same = !memcmp(&Q->st.sec[i], &g.sec[i], sizeof(SecMap));
        if (same) { P = Q; break; }
    }
    if (!P) {
        try {
            const hz_status st = build_plan(g, m, &P);
            if (st != HZ_OK) return st;
        } catch (const std::bad_alloc&) {
            return set_err(HZ_ERR_INPUT, "witness export: out of host memory while building the device plan");
        }
        m->devs.push_back(P);
    }
    *plan = P;
    return HZ_OK;
}

// HZ_EXPORT_BLOCKS: workgroups per launch (experiments: a copy that runs BESIDE a step should not take every wavefront slot of the
// device; the kernels loop over their lists whatever the grid)
static uint64_t block_cap() {
    static const uint64_t cap = getenv("HZ_EXPORT_BLOCKS") ? (uint64_t)std::max(1, atoi(getenv("HZ_EXPORT_BLOCKS"))) : (1u << 16);
    return cap;
}
static unsigned grid_for(uint64_t n, unsigned block) { return (unsigned)std::min<uint64_t>((n + block - 1) / block, block_cap()); }

// derived values of `ninst` instances from inst0 on into P->dval
static hz_status derive(DevPlan* P, ExportScratch* X, const CtxGeom& g, uint32_t inst0, uint32_t ninst, hipStream_t s) {
    if (P->D == 0 || P->n_dvars == 0) return HZ_OK;
    if (X->dval_elems < (uint64_t)ninst * P->D) {   // (hipFree waits for the device: nothing in flight reads the old one)
        HZ_HIP(X->dval.alloc((size_t)ninst * P->D * 32));
        X->dval_elems = (uint64_t)ninst * P->D;
    }
    for (int t = 2; t <= 7; t++) {
        const PosGroup& G = P->pg[t - 2];
        if (!G.n) continue;
        PosArgs a;
        memset(&a, 0, sizeof a);
        a.wit = (const hz_u32x4*)g.wit; a.dval = (hz_u32x4*)X->dval.p; a.D = P->D; a.n = G.n; a.inst0 = inst0; a.st = P->st;
        a.first = (const uint32_t*)G.first.p; a.loff = (const uint32_t*)G.loff.p; a.pl_slot = (const uint16_t*)P->pl_slot.p; a.pl_k = (const uint32_t*)P->pl_k.p;
        a.C = (const Fr*)G.C.p; a.M = (const Fr*)G.M.p;
        const dim3 grid((unsigned)((G.n + 63) / 64), ninst), block(64);
        switch (t) {
            case 2: hipLaunchKernelGGL(k_derive_poseidon<2>, grid, block, 0, s, a); break;
            case 3: hipLaunchKernelGGL(k_derive_poseidon<3>, grid, block, 0, s, a); break;
            case 4: hipLaunchKernelGGL(k_derive_poseidon<4>, grid, block, 0, s, a); break;
            case 5: hipLaunchKernelGGL(k_derive_poseidon<5>, grid, block, 0, s, a); break;
            case 6: hipLaunchKernelGGL(k_derive_poseidon<6>, grid, block, 0, s, a); break;
            default: hipLaunchKernelGGL(k_derive_poseidon<7>, grid, block, 0, s, a); break;
        }
        HZ_HIP(hipGetLastError());
    }
    DrvArgs a;
    memset(&a, 0, sizeof a);
    a.wit = (const hz_u32x4*)g.wit; a.dval = (hz_u32x4*)X->dval.p; a.D = P->D; a.inst0 = inst0; a.st = P->st;
    a.kind = (const uint8_t*)P->kind.p; a.arg = (const uint32_t*)P->arg.p;
    a.form_off = (const uint32_t*)P->form_off.p; a.form_c0 = (const uint32_t*)P->form_c0.p; a.term_coef = (const uint32_t*)P->term_coef.p; a.term_src = (const uint32_t*)P->term_src.p;
    a.pool = (const Fr*)P->pool.p; a.lvl_k = (const uint32_t*)P->lvl_k.p;
    for (const auto& lv : P->levels) {
        hipLaunchKernelGGL(k_derive_forms, dim3((unsigned)((lv.second + 63) / 64), ninst), dim3(64), 0, s, a, lv.first, lv.second);
        HZ_HIP(hipGetLastError());
    }
    return HZ_OK;
}

}  // namespace hzexp
using namespace hzexp;

extern "C" hz_status hz_symmap_upload(hz_ctx* ctx, const hz_symmap* m, uint64_t* device_bytes) {
    CtxGeom g;
    DevPlan* P = nullptr;
    const hz_status st = get_plan(ctx, m, g, &P);
    if (st != HZ_OK) return st;
    if (device_bytes) *device_bytes = P->bytes;
    return HZ_OK;
}

namespace hzexp {
// instances [inst0, inst0 + ninst) into out[j][var], j = 0..ninst-1; ninst <= 32768
static hz_status export_chunk(DevPlan* P, ExportScratch* X, const CtxGeom& g, uint32_t inst0, uint32_t ninst, void* d_out, hipStream_t s) {
    ExpArgs a;
    memset(&a, 0, sizeof a);
    a.wit = (const hz_u32x4*)g.wit; a.out = (hz_u32x4*)d_out; a.out_stride = P->nvars; a.inst0 = inst0; a.st = P->st;
    StoredLists l;
    memset(&l, 0, sizeof l);
    l.q_line0 = (const uint32_t*)P->q_src.p; l.q_dst = (const uint32_t*)P->q_dst.p; l.nq = P->n_quads;
    l.sx_src = (const uint32_t*)P->sx_src.p; l.sx_dst = (const uint32_t*)P->sx_dst.p; l.nx = P->n_sx;
    const bool x4 = P->n_s1 && ninst % 4 == 0;   // several instances together: the one-unit-per-instance sections four instances per lane
    if (!x4) { l.s1_src = (const uint32_t*)P->s1_src.p; l.s1_dst = (const uint32_t*)P->s1_dst.p; l.n1 = P->n_s1; }
    // blocks in proportion to the bytes each list moves, 2^16 in all at most
    const uint64_t wq = (l.nq + 255) / 256, w1 = (l.n1 + 511) / 512, wx = (l.nx + 511) / 512, wsum = wq + w1 + wx;
    if (wsum) {
        const uint64_t cap = std::max<uint64_t>(3, block_cap() == (1u << 16) ? block_cap() : block_cap() / std::max<uint32_t>(1, ninst));
        auto share = [&](uint64_t w) { return (uint32_t)(w == 0 ? 0 : wsum <= cap ? w : std::max<uint64_t>(1, w * cap / wsum)); };
        l.bq = share(wq); l.b1 = share(w1); l.bx = share(wx);
        hipLaunchKernelGGL(k_export_stored, dim3(l.bq + l.b1 + l.bx, ninst), dim3(256), 0, s, a, P->lt, l);
        HZ_HIP(hipGetLastError());
    }
    if (x4) {
        hipLaunchKernelGGL(k_export_singles_x4, dim3(grid_for(P->n_s1, 256), ninst / 4), dim3(256), 0, s, a, (const uint32_t*)P->s1_src.p, (const uint32_t*)P->s1_dst.p, P->n_s1);
        HZ_HIP(hipGetLastError());
    }
    if (P->n_dvars) {
        const hz_status sd = derive(P, X, g, inst0, ninst, s);
        if (sd != HZ_OK) return sd;
        hipLaunchKernelGGL(k_export_dvars, dim3(grid_for(P->n_dvars, 256), ninst), dim3(256), 0, s, (const hz_u32x4*)X->dval.p, P->D, (hz_u32x4*)d_out, P->nvars, (const uint32_t*)P->dv_v.p,
                           (const uint32_t*)P->dv_k.p, P->n_dvars);
        HZ_HIP(hipGetLastError());
    }
    return HZ_OK;
}
}  // namespace hzexp

extern "C" hz_status hz_witness_export_range_dev(hz_ctx* ctx, const hz_symmap* m, int32_t first_instance, int32_t count, void* d_out, void* stream) {
    if (!d_out) return set_err(HZ_ERR_ARG, "hz_witness_export_dev: null output buffer");
    CtxGeom g;
    DevPlan* P = nullptr;
    const hz_status st = get_plan(ctx, m, g, &P);
    if (st != HZ_OK) return st;
    if (first_instance < 0 || count < 0 || (uint64_t)first_instance + (uint64_t)count > g.n_inst) return set_err(HZ_ERR_ARG, "hz_witness_export_dev: instances %d..%d of %u", first_instance, first_instance + count - 1, g.n_inst);
    HZ_HIP(hipSetDevice(g.device));
    hipStream_t s = stream ? (hipStream_t)stream : g.s_main;
    const uint32_t CH = 32768;   // (grid.y is 16 bits; the derived-value buffer is sized for one chunk)
    for (uint32_t done = 0; done < (uint32_t)count; done += CH) {
        const uint32_t n = std::min<uint32_t>(CH, (uint32_t)count - done);
        const hz_status sc = export_chunk(P, ctx_export_scratch(ctx), g, (uint32_t)first_instance + done, n, (uint8_t*)d_out + (uint64_t)done * P->nvars * 32, s);
        if (sc != HZ_OK) return sc;
    }
    return HZ_OK;
}
extern "C" hz_status hz_witness_export_dev(hz_ctx* ctx, const hz_symmap* m, int32_t instance, void* d_out, void* stream) {
    if (instance >= 0) return hz_witness_export_range_dev(ctx, m, instance, 1, d_out, stream);
    if (instance != -1 || !ctx) return set_err(HZ_ERR_ARG, "hz_witness_export_dev: bad instance %d", instance);
    CtxGeom g;
    ctx_geometry(ctx, g);
    return hz_witness_export_range_dev(ctx, m, 0, (int32_t)g.n_inst, d_out, stream);
}

// For a consumer that prefers indirection over a copy: where every variable lives. phys0[v] = element of hz_witness_dev_ptr() that
// holds variable v of instance 0, inst_stride[v] = elements to add per instance; a DERIVED variable has phys0[v] = 2^63 | k and its
// value is element instance_slot * n_derived_slots + k of the buffer hz_witness_derive_dev fills.
extern "C" hz_status hz_symmap_dev_index(hz_ctx* ctx, const hz_symmap* m, const uint64_t** d_phys0, const uint32_t** d_inst_stride, uint64_t* n_derived_slots) {
    CtxGeom g;
    DevPlan* P = nullptr;
    const hz_status st = get_plan(ctx, m, g, &P);
    if (st != HZ_OK) return st;
    HZ_HIP(hipSetDevice(g.device));
    if (!P->phys0.p && P->nvars) {
        DevBuf idx;
        HZ_HIP(idx.alloc(P->nvars * 8));
        HZ_HIP(hipMemcpy(idx.p, m->index.data(), P->nvars * 8, hipMemcpyHostToDevice));
        HZ_HIP(P->phys0.alloc(P->nvars * 8));
        HZ_HIP(P->istride.alloc(P->nvars * 4));
        hipLaunchKernelGGL(k_phys_index, dim3(grid_for(P->nvars, 256)), dim3(256), 0, g.s_main, P->st, (const uint64_t*)idx.p, P->nvars, (uint64_t*)P->phys0.p, (uint32_t*)P->istride.p);
        HZ_HIP(hipGetLastError());
        HZ_HIP(hipStreamSynchronize(g.s_main));
        P->bytes += P->phys0.bytes + P->istride.bytes;
    }
    if (d_phys0) *d_phys0 = (const uint64_t*)P->phys0.p;
    if (d_inst_stride) *d_inst_stride = (const uint32_t*)P->istride.p;
    if (n_derived_slots) *n_derived_slots = P->D;
    return HZ_OK;
}
extern "C" hz_status hz_witness_derive_dev(hz_ctx* ctx, const hz_symmap* m, int32_t instance, const void** d_derived, void* stream) {
    CtxGeom g;
    DevPlan* P = nullptr;
    const hz_status st = get_plan(ctx, m, g, &P);
    if (st != HZ_OK) return st;
    if (instance < -1 || instance >= (int32_t)g.n_inst) return set_err(HZ_ERR_ARG, "hz_witness_derive_dev: bad instance %d", instance);
    HZ_HIP(hipSetDevice(g.device));
    if (instance < 0 && g.n_inst > 32768) return set_err(HZ_ERR_ARG, "hz_witness_derive_dev: at most 32768 instances at once");
    ExportScratch* X = ctx_export_scratch(ctx);
    const hz_status sd = derive(P, X, g, instance < 0 ? 0 : (uint32_t)instance, instance < 0 ? g.n_inst : 1, stream ? (hipStream_t)stream : g.s_main);
    if (sd != HZ_OK) return sd;
    if (d_derived) *d_derived = X->dval.p;
    return HZ_OK;
}

// Host delivery: the same device pass into a device buffer of the context's device, then asynchronous copies of `chunk`-sized pieces
// through a ring of two pinned buffers; `sink` consumes piece i while piece i + 1 crosses PCIe. sink(bytes, n) returns false to stop.
namespace hzexp {
struct PinnedRing {
    void* b[2] = {nullptr, nullptr};
    hipEvent_t ev[2] = {nullptr, nullptr};
    size_t bytes = 0;
    ~PinnedRing() {
        for (int i = 0; i < 2; i++) { if (b[i]) (void)hipHostFree(b[i]); if (ev[i]) (void)hipEventDestroy(ev[i]); }
    }
};
template <class Sink>
static hz_status export_through_ring(hz_ctx* ctx, const hz_symmap* m, int32_t instance, uint64_t first, uint64_t count, Sink sink, uint8_t* direct_out = nullptr) {
    CtxGeom g;
    DevPlan* P = nullptr;
    hz_status st = get_plan(ctx, m, g, &P);
    if (st != HZ_OK) return st;
    if (instance < 0 || instance >= (int32_t)g.n_inst) return set_err(HZ_ERR_ARG, "witness export: bad instance %d", instance);
    if (first > P->nvars || count > P->nvars - first) return set_err(HZ_ERR_ARG, "witness export: range beyond the %llu variables", (unsigned long long)P->nvars);
    HZ_HIP(hipSetDevice(g.device));
    ExportScratch* X = ctx_export_scratch(ctx);
    if (X->xbuf.bytes < std::max<uint64_t>(P->nvars, 1) * 32) HZ_HIP(X->xbuf.alloc(std::max<uint64_t>(P->nvars, 1) * 32));
    DevBuf& dout = X->xbuf;
    struct Release {   // a whole-witness staging vector (3.86 GB at the headline shape) does not stay behind a host delivery
        DevBuf& b;
        ~Release() { if (b.bytes > (64ull << 20)) b.release(); }
    } release_after{dout};
    st = hz_witness_export_dev(ctx, m, instance, dout.p, g.s_main);
    if (st != HZ_OK) return st;
    if (direct_out) {   // the caller's buffer is pinned: one copy at PCIe speed, no staging
        HZ_HIP(hipMemcpyAsync(direct_out, (const uint8_t*)dout.p + first * 32, count * 32, hipMemcpyDeviceToHost, g.s_main));
        HZ_HIP(hipStreamSynchronize(g.s_main));
        return HZ_OK;
    }
    PinnedRing ring;
    ring.bytes = (size_t)std::min<uint64_t>(std::max<uint64_t>(count, 1) * 32, 64ull << 20);
    for (int i = 0; i < 2; i++) { HZ_HIP(hipHostMalloc(&ring.b[i], ring.bytes, hipHostMallocDefault)); HZ_HIP(hipEventCreateWithFlags(&ring.ev[i], hipEventDisableTiming)); }
    const uint64_t per = ring.bytes / 32;
    const uint64_t nchunks = (count + per - 1) / per;
    auto issue = [&](uint64_t c) -> hipError_t {
        const uint64_t n = std::min<uint64_t>(per, count - c * per);
        hipError_t e = hipMemcpyAsync(ring.b[c & 1], (const uint8_t*)dout.p + (first + c * per) * 32, n * 32, hipMemcpyDeviceToHost, g.s_main);
        if (e == hipSuccess) e = hipEventRecord(ring.ev[c & 1], g.s_main);
        return e;
    };
    if (nchunks) HZ_HIP(issue(0));
    for (uint64_t c = 0; c < nchunks; c++) {
        HZ_HIP(hipEventSynchronize(ring.ev[c & 1]));
        if (c + 1 < nchunks) HZ_HIP(issue(c + 1));
        if (!sink((const uint8_t*)ring.b[c & 1], std::min<uint64_t>(per, count - c * per))) return set_err(HZ_ERR_ARG, "witness export: the consumer stopped");
    }
    HZ_HIP(hipStreamSynchronize(g.s_main));
    return HZ_OK;
}
}  // namespace hzexp

extern "C" hz_status hz_witness_export_host(hz_ctx* ctx, const hz_symmap* m, int32_t instance, uint64_t first_var, uint64_t count, uint8_t* out) {
    if (!out) return set_err(HZ_ERR_ARG, "hz_witness_export_host: null output buffer");
    uint8_t* p = out;
    hipPointerAttribute_t at;
    bool pinned = hipPointerGetAttributes(&at, out) == hipSuccess && at.type == hipMemoryTypeHost;
    if (!pinned) (void)hipGetLastError();   // (an ordinary host pointer is an error to the query, not to us)
    return export_through_ring(ctx, m, instance, first_var, count, [&](const uint8_t* b, uint64_t n) { memcpy(p, b, n * 32); p += n * 32; return true; }, pinned ? out : nullptr);
}

namespace hz {
// the .wtns of formats.hip, fed from the ring (formats.hip: hz_witness_write_wtns_sym)
hz_status export_to_file(hz_ctx* ctx, const hz_symmap* m, int32_t instance, FILE* f, const char* path) {
    bool wrote = true;
    const hz_status st = export_through_ring(ctx, m, instance, 0, m->index.size(), [&](const uint8_t* b, uint64_t n) { return wrote = (fwrite(b, 32, n, f) == n); });
    return wrote ? st : set_err(HZ_ERR_ARG, "short write to %s", path);
}
}  // namespace hz
