// K5/K6 eddsa: AySign2Ax (reference src/lib/utils-bjj.circom:37-58 + circomlib pointbits.circom
// Bits2Point_Strict) and EdDSAPoseidonVerifier (circomlib eddsaposeidon.circom with babyjub,
// escalarmulany, escalarmulfix, montgomery, compconstant, aliascheck) -- call site
// reference src/rollup-tx.circom:467-482. One transaction per lane.
//
// Every intermediate point of the two scalar multiplications is a witness signal, so the ladders
// are evaluated in the circuit's own affine Montgomery-curve formulas, one division per curve
// operation. A field inversion costs ~85 products (tools/microbench/invbench.hip) and there are
// ~340 of them per signature, so each lane walks HZ_ED_G signatures in lockstep and shares ONE
// inversion per ladder step / window among them (Montgomery's trick): 3 extra products per shared
// element instead of an inversion. EscalarMulFix's window tables are compile-time constants
// (bjj_consts.inc).
//
// The field routines are inlined here too (HZ_FR_INLINE): the call overhead of the out-of-line product (operands copied in and
// out of the argument registers) was 8 % of the ladder -- 27.7 -> 25.4 ms per 65 536 signatures, +2.5 % rollup-main throughput.
#ifndef HZ_FR_INLINE
#define HZ_FR_INLINE 1
#endif
#include <hip/hip_runtime.h>
#include "gadgets_dev.h"
#include "tx_dev.h"
#include "kernels.h"

namespace hz {

#define HZ_CONST_ARR static __device__ const
#include "gen/bjj_consts.inc"
#undef HZ_CONST_ARR

struct PtA { Fr x, y; };
// The helpers below (point conversions, BabyAdd, the segment heads and tails, the prologue) were separate functions until round 6: a
// call passes its structs BY REFERENCE (more than 16 dwords of aggregates never travel in registers), so every caller kept the curve
// constants, the witness cursor, the kernel's argument block and each point in private memory -- 1.6 KB per lane in k_eddsa_seg<4>, none
// of it inside the ladder loop. Inlined, those objects are registers.
#ifndef HZ_ED_CALL
#define HZ_ED_CALL __forceinline__
#endif
__device__ __forceinline__ Fr ld_const(const uint32_t* p) {
    Fr r;
    for (int i = 0; i < 9; i++) r.v[i] = p[i];
    return r;
}

// a / b with the 0-divisor convention (-> 0)
__device__ __forceinline__ Fr fr_div(const Fr& a, const Fr& b) { return fr_mul(a, fr_inv(b)); }

// pointbits.circom sqrt(): root <= (r-1)/2, or 0 when n is a non-residue. One exponentiation
// z = n^((q-1)/2) gives r = z*n = n^((q+1)/2) and t = r*z = n^q; the Tonelli-Shanks loop then
// detects non-residues itself (t^(2^27) == -1).
__device__ Fr fr_sqrt_circom_dev(const Fr& n) {
    if (fr_is_zero(n)) return fr_zero();
    const Fr one = fr_one();
    uint32_t e[8];
    for (int i = 0; i < 8; i++) e[i] = HZ_SQRT_ZEXP[i];
    const Fr z = fr_pow(n, e);
    Fr r = fr_mul(z, n), t = fr_mul(r, z), c = ld_const(HZ_SQRT_ROOT);
    int m = 28;
    while (!fr_eq(t, one)) {
        Fr sq = fr_sqr(t);
        int i = 1;
        while (i < m && !fr_eq(sq, one)) { sq = fr_sqr(sq); i++; }
        if (i >= m) return fr_zero();  // non-residue
        Fr b = c;
        for (int j = 0; j < m - i - 1; j++) b = fr_sqr(b);
        m = i;
        c = fr_sqr(b);
        t = fr_mul(t, c);
        r = fr_mul(r, b);
    }
    // normalise: canonical value > (r-1)/2 -> negate
    const Fc rc = fr_to_canon(r);
    bool gt = false;
    for (int i = 7; i >= 0; i--) {
        if (rc.v[i] > CT_HALF_D[i]) { gt = true; break; }
        if (rc.v[i] < CT_HALF_D[i]) break;
    }
    return gt ? fr_neg(r) : r;
}

#ifndef HZ_ED_G
#define HZ_ED_G 4   // signatures per segment lane (launches above HZ_ED_SPLIT_MAX signatures: k_eddsa_seg)
#endif
#ifndef HZ_ED_SPLIT_MAX
// launches up to this many signatures: one segment of one signature per lane, no inversion per step (seg_any_proj). Measured with
// k_eddsa_seg<4> as the alternative (two contexts in flight): 8 192 signatures 10.2 vs 12.7 ms per step, 16 384: 13.3-13.5 vs 14.6,
// 32 768: 21.4 vs 20.7 -- the crossover lies between 8 and 16 batches of 2048 (profiles/r03_eddsa_seg_ab.txt)
#define HZ_ED_SPLIT_MAX 16384
#endif

struct EdCtx {
    UnitIO io;
    Fr a, d, A;   // 168700, 168696, 168698 (Montgomery-curve A; B = 1)
    Fr one;
};
// the curve constants once per lane, the witness cursor per signature
struct EdK {
    Fr a, d, A, one;
    __device__ __forceinline__ EdCtx with(const UnitIO& io) const { return EdCtx{io, a, d, A, one}; }
};
// 168700, 168696, 168698 in Montgomery form as literals (c * 2^261 mod p): a constant the compiler can re-materialise costs no register
// across the ladder and no spill slot; fr_from_u64 computed them with a product per kernel and kept 27 registers alive
__device__ __forceinline__ EdK ed_k() {
    constexpr uint32_t ka[9] = {0x1e4c3bf6u, 0x1632a3c4u, 0x0f025983u, 0x1533bdfbu, 0x1e960fcfu, 0x1e77617fu, 0x136b897du, 0x18c1d173u, 0x0002d053u};
    constexpr uint32_t kd[9] = {0x1e4c3e9cu, 0x19b5d231u, 0x170a87f5u, 0x0793a2bdu, 0x1f019767u, 0x0f24dfc7u, 0x0d641be0u, 0x00ceff39u, 0x002c7818u};
    constexpr uint32_t kA[9] = {0x0e4c3d49u, 0x17f43afbu, 0x130670bcu, 0x0e63b05cu, 0x1ecbd39bu, 0x06ce20a3u, 0x1067d2afu, 0x1cc86856u, 0x0017a435u};
    EdK K;
#pragma unroll
    for (int i = 0; i < 9; i++) { K.a.v[i] = ka[i]; K.d.v[i] = kd[i]; K.A.v[i] = kA[i]; }
    K.one = fr_one();
    return K;
}

// BabyAdd: stores beta,gamma,delta,tau,xout,yout; checks the two division constraints
__device__ HZ_ED_CALL PtA baby_add_dev(const EdCtx& c, BabyAddOff off, const PtA& p, const PtA& q) {
    const Fr beta = fr_mul(p.x, q.y), gamma = fr_mul(p.y, q.x);
    const Fr delta = fr_mul(fr_sub(p.y, fr_mul(c.a, p.x)), fr_add(q.x, q.y));
    const Fr tau = fr_mul(beta, gamma);
    const Fr dt = fr_mul(c.d, tau);
    Fr den[2] = {fr_add(c.one, dt), fr_sub(c.one, dt)};
    Fr inv[2] = {den[0], den[1]};
    inv_pair(inv[0], inv[1]);
    const Fr numx = fr_add(beta, gamma), numy = fr_sub(fr_add(delta, fr_mul(c.a, beta)), gamma);
    PtA r;
    r.x = fr_mul(numx, inv[0]);
    r.y = fr_mul(numy, inv[1]);
    c.io.put_m(off + BA_BETA, beta); c.io.put_m(off + BA_GAMMA, gamma); c.io.put_m(off + BA_DELTA, delta); c.io.put_m(off + BA_TAU, tau);
    c.io.put_m(off + BA_XOUT, r.x); c.io.put_m(off + BA_YOUT, r.y);
    if (fr_is_zero(den[0])) c.io.chk(C_RTX_SIG_EC, fr_zero(), numx);
    if (fr_is_zero(den[1])) c.io.chk(C_RTX_SIG_EC, fr_zero(), numy);
    return r;
}
struct MDbl { Fr x1_2, lamda; PtA out; };
__device__ HZ_ED_CALL MDbl mont_dbl_dev(const EdCtx& c, const PtA& p) {
    MDbl r;
    r.x1_2 = fr_sqr(p.x);
    const Fr num = fr_add(fr_add(fr_add(fr_dbl(r.x1_2), r.x1_2), fr_mul(fr_dbl(c.A), p.x)), c.one);
    const Fr den = fr_dbl(p.y);
    r.lamda = fr_div(num, den);
    if (fr_is_zero(den)) c.io.chk(C_RTX_SIG_EC, fr_zero(), num);
    r.out.x = fr_sub(fr_sub(fr_sqr(r.lamda), c.A), fr_dbl(p.x));
    r.out.y = fr_sub(fr_mul(r.lamda, fr_sub(p.x, r.out.x)), p.y);
    return r;
}
__device__ HZ_ED_CALL PtA e2m_dev(const EdCtx& c, const PtA& p) {
    Fr den[2] = {fr_sub(c.one, p.y), p.x};
    Fr inv[2] = {den[0], den[1]};
    inv_pair(inv[0], inv[1]);
    PtA o;
    o.x = fr_mul(fr_add(c.one, p.y), inv[0]);
    o.y = fr_mul(o.x, inv[1]);
    if (fr_is_zero(den[0])) c.io.chk(C_RTX_SIG_EC, fr_zero(), fr_add(c.one, p.y));
    if (fr_is_zero(den[1])) c.io.chk(C_RTX_SIG_EC, fr_zero(), o.x);
    return o;
}
__device__ HZ_ED_CALL PtA m2e_dev(const EdCtx& c, const PtA& p) {
    Fr den[2] = {p.y, fr_add(p.x, c.one)};
    Fr inv[2] = {den[0], den[1]};
    inv_pair(inv[0], inv[1]);
    PtA o;
    o.x = fr_mul(p.x, inv[0]);
    o.y = fr_mul(fr_sub(p.x, c.one), inv[1]);
    if (fr_is_zero(den[0])) c.io.chk(C_RTX_SIG_EC, fr_zero(), p.x);
    if (fr_is_zero(den[1])) c.io.chk(C_RTX_SIG_EC, fr_zero(), fr_sub(p.x, c.one));
    return o;
}

// SegmentMulAny(n) for G signatures in lockstep: bits e[g][e0 .. e0+n) of the canonical integers.
// Step i is doubler_i (D_{i+1} = 2 D_i) followed by adder_i (D_{i+1} + acc_i). adder_i and
// doubler_{i+1} both depend only on D_{i+1} and acc_i, so one step needs 2 divisions per signature:
// all 2G of them share one inversion.
// in: p[g] (Edwards). out: p[g] <- segment output (Edwards), dbl[g] <- last doubler output (Montgomery).
// Scales (HZ_ED_SCALES, default): a field element v is carried as the integer v * R^k mod p, k = 1 the Montgomery form, k = 0 the
// number itself -- which is what the witness stores. The product routine maps scales (j, k) to j + k - 1, sums need equal scales.
// Every point of the ladder is a stored signal, and in the all-Montgomery form each store pays a reduction of its own (nine per
// step and signature, against eight products). Here the points live in scale 0 and only what is squared keeps a scale-1 twin:
//   inverses: the scale-0 divisors go through the Montgomery batch inversion as they are, which returns 1/v in scale 2;
//   lamda[1] = num[0] * inv[2], lamda[0] = one reduction; x'[0] = lamda[0] * lamda[1] - A - ..., y'[0] = lamda[1] * (x - x')[0] - y[0]
//   (scale 0 straight out of the product); the doubler's x1_2[0] = x[0] * x[1] needs x[1]: its x' is computed in scale 1 and reduced.
// Three reductions per step instead of nine; the selector outputs are copies of scale-0 values. Same field elements, same signals.
// scale-0 value in [0, 2p), normalised limbs -> the witness
__device__ __forceinline__ Fr ed_put0(const UnitIO& w, uint32_t sig, const Fr& x) {
    const Fr c = fr_cond_sub_p(x);
    w.put_c(sig, fr_pack_canon(c));
    return c;
}
__device__ __forceinline__ Fr fr_scale_up(const Fr& x) {   // scale k -> k + 1
    Fr r2;
#pragma unroll
    for (int i = 0; i < 9; i++) r2.v[i] = fr_r2(i);
    return fr_mul(x, r2);
}
__device__ __forceinline__ Fr fr_limbs_u64(uint64_t x) {   // small integer in scale 0
    Fr r = fr_zero();
    r.v[0] = (uint32_t)x & HZ_M29; r.v[1] = (uint32_t)(x >> 29) & HZ_M29; r.v[2] = (uint32_t)(x >> 58);
    return r;
}
template <int G>
__device__ HZ_ED_CALL void seg_any_lock(const EdK& K, const UnitIO* io, const SegAnyOff& o, const Fc* e, int e0, int n, PtA* p, PtA* dbl) {
    Fr dx0[G], dx1[G], dy0[G];   // doubler output D_{i+1}: x in scales 0 and 1, y in scale 0
    PtA addIn[G];                // the accumulator, scale 0, canonical
    Fr nx1_2[G], d_num[G];       // scale 0
    const int steps = n - 1;
    const Fr A0 = fr_limbs_u64(168698), one0 = fr_limbs_u64(1);
#pragma unroll 1
    for (int g = 0; g < G; g++) {
        const EdCtx c = K.with(io[g]);
        const PtA m = e2m_dev(c, p[g]);
        addIn[g].x = fr_canon_limbs(m.x); addIn[g].y = fr_canon_limbs(m.y);
        c.io.put_c(o.e2m, fr_pack_canon(addIn[g].x)); c.io.put_c(o.e2m + 1, fr_pack_canon(addIn[g].y));
        const MDbl d = mont_dbl_dev(c, m);   // doubler_0
        c.io.put_m(o.bits + BIT_DBL_X1_2, d.x1_2); c.io.put_m(o.bits + BIT_DBL_LAMDA, d.lamda);
        dx1[g] = d.out.x;
        dx0[g] = fr_canon_limbs(d.out.x); dy0[g] = fr_canon_limbs(d.out.y);
        c.io.put_c(o.bits + BIT_DBL_OUT0, fr_pack_canon(dx0[g])); c.io.put_c(o.bits + BIT_DBL_OUT1, fr_pack_canon(dy0[g]));
    }
    const Fr A2 = fr_dbl(K.A);
#pragma unroll 1
    for (int i = 0; i < steps; i++) {
        const bool more = i + 1 < steps;
        const uint32_t b = o.bits + BIT_N * i;
        Fr inv[2 * G];
        uint32_t zmask = 0;
#pragma unroll 1
        for (int g = 0; g < G; g++) {
            // adder_i: in1 = D_{i+1}, in2 = accumulator ; doubler_{i+1}: in = D_{i+1}
            const Fr a_den = fr_sub(addIn[g].x, dx0[g]);
            Fr dd = fr_zero();
            if (more) {
                nx1_2[g] = fr_mul(dx0[g], dx1[g]);
                d_num[g] = fr_add(fr_add(fr_add(fr_dbl(nx1_2[g]), nx1_2[g]), fr_mul(A2, dx0[g])), one0);
                dd = fr_dbl(dy0[g]);
            }
            inv[2 * g] = a_den;
            inv[2 * g + 1] = dd;
            if (fr_is_zero(a_den)) zmask |= 1u << (2 * g);
            if (more && fr_is_zero(dd)) zmask |= 1u << (2 * g + 1);
        }
        if constexpr (G == 1) inv_pair(inv[0], inv[1]);   // scale-0 divisors in, scale-2 inverses out
        else batch_inv<2 * G>(inv, 2 * G);
#pragma unroll 1
        for (int g = 0; g < G; g++) {
            const UnitIO& w = io[g];
            const Fr a_num = fr_sub(addIn[g].y, dy0[g]);
            const Fr a_l1 = fr_mul(a_num, inv[2 * g]);
            const Fr a_l0 = fr_canon_limbs(a_l1);
            if ((zmask >> (2 * g)) & 1) w.chk(C_RTX_SIG_EC, fr_zero(), fr_scale_up(a_num));
            PtA ao;
            ao.x = fr_sub(fr_sub(fr_sub(fr_mul(a_l0, a_l1), A0), dx0[g]), addIn[g].x);
            ao.y = fr_sub(fr_mul(a_l1, fr_sub(dx0[g], ao.x)), dy0[g]);
            w.put_c(b + BIT_ADD_LAMDA, fr_pack_canon(a_l0));
            ao.x = ed_put0(w, b + BIT_ADD_OUT0, ao.x); ao.y = ed_put0(w, b + BIT_ADD_OUT1, ao.y);
            const uint32_t sel = c_bit(e[g], e0 + i + 1);
            const PtA so = sel ? ao : addIn[g];
            w.put_c(b + BIT_SEL_OUT0, fr_pack_canon(so.x)); w.put_c(b + BIT_SEL_OUT1, fr_pack_canon(so.y));
            addIn[g] = so;
            if (more) {
                const Fr l1 = fr_mul(d_num[g], inv[2 * g + 1]);
                const Fr l0 = fr_canon_limbs(l1);
                if ((zmask >> (2 * g + 1)) & 1) w.chk(C_RTX_SIG_EC, fr_zero(), fr_scale_up(d_num[g]));
                const Fr nx1 = fr_sub(fr_sub(fr_sqr(l1), K.A), fr_dbl(dx1[g]));
                const Fr nx0 = fr_canon_limbs(nx1);
                const Fr ny0 = fr_sub(fr_mul(l1, fr_sub(dx0[g], nx0)), dy0[g]);
                const uint32_t bn = b + BIT_N;
                (void)ed_put0(w, bn + BIT_DBL_X1_2, nx1_2[g]);
                w.put_c(bn + BIT_DBL_LAMDA, fr_pack_canon(l0)); w.put_c(bn + BIT_DBL_OUT0, fr_pack_canon(nx0));
                dy0[g] = ed_put0(w, bn + BIT_DBL_OUT1, ny0);
                dx1[g] = nx1; dx0[g] = nx0;
            }
        }
    }
#pragma unroll 1
    for (int g = 0; g < G; g++) {
        const EdCtx c = K.with(io[g]);
        dbl[g].x = dx1[g];
        dbl[g].y = fr_scale_up(dy0[g]);
        PtA acc;
        acc.x = fr_scale_up(addIn[g].x); acc.y = fr_scale_up(addIn[g].y);
        const PtA me = m2e_dev(c, acc);
        c.io.put_m(o.m2e, me.x); c.io.put_m(o.m2e + 1, me.y);
        PtA negp;
        negp.x = fr_neg(p[g].x);
        negp.y = p[g].y;
        const PtA ea = baby_add_dev(c, o.eadder, me, negp);
        const PtA r = c_bit(e[g], e0) ? me : ea;
        c.io.put_m(o.lastSel, r.x); c.io.put_m(o.lastSel + 1, r.y);
        p[g] = r;
    }
}

// ---------------------------------------------------------------------------------------------------------------------------
// The same ladder with the per-signature state parked in LDS between a signature's turns (throughput launches, k_eddsa_seg<G>).
// seg_any_lock keeps its state in arrays indexed by g: with G = 2 the compiler puts them in scratch memory, and every turn of a
// signature loads and stores its nine field elements through the vector-memory path -- round 2's counters: 23 GB of traffic for
// 9.4 GB of signals, and a full memory latency in front of every turn of a kernel that is one long dependent chain on 512
// wavefronts. One wavefront per workgroup, slot s limb l of lane L at st[(s * 9 + l) * 64 + L] (conflict-free 256-byte rows);
// nothing else of a step uses LDS.
typedef __attribute__((address_space(3))) uint32_t lds_u32;   // an LDS pointer the compiler can prove is one: ds_read / ds_write, not flat_*
struct LaneLds {
    lds_u32* p;   // st + lane
    __device__ __forceinline__ Fr get(int slot) const {
        Fr r;
#pragma unroll
        for (int l = 0; l < 9; l++) r.v[l] = p[(slot * 9 + l) * 64];
        return r;
    }
    __device__ __forceinline__ void put(int slot, const Fr& v) const {
#pragma unroll
        for (int l = 0; l < 9; l++) p[(slot * 9 + l) * 64] = v.v[l];
    }
};
// Slots: six per signature (the doubling chain's point in scales 0 / 1, the accumulator, the next doubler's numerator), the 2 G - 1
// prefix products of the step's shared inversion, and the ladder bits of the G signatures (8 words each): 17 slots = 38 KB at G = 2
// (four such wavefronts per CU), 31 slots = 70 KB at G = 4 (two per CU: see k_eddsa_seg for what that means for two contexts in
// flight). With whole signatures as lanes (the 254-step chain of round 3) a footprint above a quarter of a CU made the step longer;
// with the segments as lanes the chain is short enough for the contexts' ladders to alternate.
enum { LS_DX1 = 0, LS_DY0, LS_AX, LS_AY, LS_DNUM, LS_DX0, LS_PER_G_MAX };
// G <= 3 parks the doubling chain's x in both scales; G = 4 recomputes the scale-0 twin (one reduction, three times a step) to
// stay within two wavefronts per CU (31 slots = 70 KB)
template <int G> constexpr bool ls_keep_dx0() { return G <= 3; }
template <int G> constexpr int ls_per_g() { return ls_keep_dx0<G>() ? LS_PER_G_MAX : LS_PER_G_MAX - 1; }
template <int G> constexpr int ls_pre0() { return G * ls_per_g<G>(); }                 // pre[1 .. 2G-1] (pre[0] = 1 is not stored)
template <int G> constexpr int ls_h0() { return G * ls_per_g<G>() + 2 * G - 1; }       // word w of signature g's bits at dword g * 8 + w of this region
template <int G> constexpr int ls_slots() { return ls_h0<G>() + (8 * G + 8) / 9; }
template <int G> __device__ __forceinline__ Fr ls_dx0(const LaneLds& L, int s0) {
    if constexpr (ls_keep_dx0<G>()) return L.get(s0 + LS_DX0);
    else return fr_canon_limbs(L.get(s0 + LS_DX1));
}
__device__ __forceinline__ uint32_t ls_hbit(const LaneLds& L, int h0, int g, int bit) {
    const int d = g * 8 + (bit >> 5);
    return (L.p[(h0 * 9 + d) * 64] >> (bit & 31)) & 1u;
}
__device__ __forceinline__ void ls_hput(const LaneLds& L, int h0, int g, const Fc& h) {
#pragma unroll
    for (int q = 0; q < 8; q++) L.p[(h0 * 9 + g * 8 + q) * 64] = h.v[q];
}

// start of a segment for signature g: e2m, doubler_0 (as the head of seg_any_lock)
template <int G>
__device__ HZ_ED_CALL void seg_lds_init(const EdK& K, const UnitIO& io, const SegAnyOff& o, const PtA& p, const LaneLds& L, int g) {
    constexpr int LS_PER_G = ls_per_g<G>();
    const EdCtx c = K.with(io);
    const PtA m = e2m_dev(c, p);
    const Fr ax = fr_canon_limbs(m.x), ay = fr_canon_limbs(m.y);
    L.put(g * LS_PER_G + LS_AX, ax); L.put(g * LS_PER_G + LS_AY, ay);
    c.io.put_c(o.e2m, fr_pack_canon(ax)); c.io.put_c(o.e2m + 1, fr_pack_canon(ay));
    const MDbl d = mont_dbl_dev(c, m);   // doubler_0
    c.io.put_m(o.bits + BIT_DBL_X1_2, d.x1_2); c.io.put_m(o.bits + BIT_DBL_LAMDA, d.lamda);
    const Fr dx0 = fr_canon_limbs(d.out.x), dy0 = fr_canon_limbs(d.out.y);
    L.put(g * LS_PER_G + LS_DX1, d.out.x); L.put(g * LS_PER_G + LS_DY0, dy0);
    if constexpr (ls_keep_dx0<G>()) L.put(g * LS_PER_G + LS_DX0, dx0);
    c.io.put_c(o.bits + BIT_DBL_OUT0, fr_pack_canon(dx0)); c.io.put_c(o.bits + BIT_DBL_OUT1, fr_pack_canon(dy0));
}
// the steps of a segment for the G signatures of a lane in lockstep (the loop of seg_any_lock, state in LDS); `mk_io(g)` gives
// signature g's witness cursor. One inversion per step for the 2 G divisors (adder and next doubler of every signature): forward
// pass = prefix products (parked), backward pass = each signature's two inverses peeled off and USED at once -- the divisors are
// recomputed from the state (a subtraction, a doubling) instead of being parked, the inverses never leave the registers.
template <int G, class MkIo>
__device__ __forceinline__ void seg_lds_steps(const EdK& K, const MkIo& mk_io, const SegAnyOff& o, int e0, int n, const LaneLds& L) {
    constexpr int LS_PER_G = ls_per_g<G>();
    const int steps = n - 1;
    const Fr A0 = fr_limbs_u64(168698), one0 = fr_limbs_u64(1);
    const Fr A2 = fr_dbl(K.A);
#pragma unroll 1
    for (int i = 0; i < steps; i++) {
        const bool more = i + 1 < steps;
        const uint32_t b = o.bits + BIT_N * i;
        Fr acc = fr_one();
#pragma unroll 1
        for (int g = 0; g < G; g++) {
            const int s0 = g * LS_PER_G;
            const Fr dx0 = ls_dx0<G>(L, s0);
            const Fr a_den = fr_sub(L.get(s0 + LS_AX), dx0);
            if (g > 0) L.put(ls_pre0<G>() + 2 * g - 1, acc);          // pre[2g]
            if (!fr_is_zero(a_den)) acc = fr_mul(acc, a_den);
            L.put(ls_pre0<G>() + 2 * g, acc);                          // pre[2g + 1]
            if (more) {
                const UnitIO w = mk_io(g);
                const Fr nx1_2 = fr_mul(dx0, L.get(s0 + LS_DX1));
                (void)ed_put0(w, b + BIT_N + BIT_DBL_X1_2, nx1_2);     // doubler_{i+1}.x1_2
                // 3 x1_2 + 2A x + 1 below 4.2 p, not reduced: it only feeds the product with the inverse (fr.h "lazily reduced sums")
                L.put(s0 + LS_DNUM, fr_3a_b_c_lazy(nx1_2, fr_mul(A2, dx0), one0));
                const Fr dd = fr_dbl_lazy(L.get(s0 + LS_DY0));   // y is stored canonical: 2y < 2p as it is
                if (!fr_is_zero(dd)) acc = fr_mul(acc, dd);
            }
        }
        Fr inv = fr_inv(acc);   // scale-0 divisors in, scale-2 inverses out
#pragma unroll 1
        for (int g = G - 1; g >= 0; g--) {
            const int s0 = g * LS_PER_G;
            const UnitIO w = mk_io(g);
            const Fr dx0 = ls_dx0<G>(L, s0), dy0 = L.get(s0 + LS_DY0);
            PtA addIn;
            addIn.x = L.get(s0 + LS_AX); addIn.y = L.get(s0 + LS_AY);
            // divisor 2g + 1 (the next doubler's 2y), then divisor 2g (the adder's x2 - x1): batch_inv's backward pass
            Fr inv_dd = fr_zero();
            bool dd_zero = true;
            if (more) {
                const Fr dd = fr_dbl_lazy(dy0);
                dd_zero = fr_is_zero(dd);
                if (!dd_zero) {
                    inv_dd = fr_mul(inv, L.get(ls_pre0<G>() + 2 * g));
                    inv = fr_mul(inv, dd);
                }
            }
            const Fr a_den = fr_sub(addIn.x, dx0);
            const bool a_zero = fr_is_zero(a_den);
            Fr inv_a = fr_zero();
            if (!a_zero) {
                inv_a = g > 0 ? fr_mul(inv, L.get(ls_pre0<G>() + 2 * g - 1)) : inv;
                if (g > 0) inv = fr_mul(inv, a_den);
            }
            const Fr a_num = fr_sub_lazy(addIn.y, dy0);   // in (p, 3p): a multiplicand only
            const Fr a_l1 = fr_mul(a_num, inv_a);
            const Fr a_l0 = fr_canon_limbs(a_l1);
            if (a_zero) w.chk(C_RTX_SIG_EC, fr_zero(), fr_scale_up(a_num));
            PtA ao;
            ao.x = ed_put0(w, b + BIT_ADD_OUT0, fr_sub3(fr_mul(a_l0, a_l1), A0, dx0, addIn.x));   // one reduction for the three differences
            ao.y = fr_sub(fr_mul(a_l1, fr_sub_lazy(dx0, ao.x)), dy0);
            w.put_c(b + BIT_ADD_LAMDA, fr_pack_canon(a_l0));
            ao.y = ed_put0(w, b + BIT_ADD_OUT1, ao.y);
            const uint32_t sel = ls_hbit(L, ls_h0<G>(), g, e0 + i + 1);
            const PtA so = sel ? ao : addIn;
            w.put_c(b + BIT_SEL_OUT0, fr_pack_canon(so.x)); w.put_c(b + BIT_SEL_OUT1, fr_pack_canon(so.y));
            L.put(s0 + LS_AX, so.x); L.put(s0 + LS_AY, so.y);
            if (more) {
                const Fr d_num = L.get(s0 + LS_DNUM);
                const Fr l1 = fr_mul(d_num, inv_dd);
                const Fr l0 = fr_canon_limbs(l1);
                if (dd_zero) w.chk(C_RTX_SIG_EC, fr_zero(), fr_scale_up(d_num));
                const Fr nx1 = fr_sub_a_2x(fr_sqr(l1), K.A, L.get(s0 + LS_DX1));   // l^2 - A - 2x in [0, 2p), one reduction
                const Fr nx0 = fr_canon_limbs(nx1);
                const Fr ny0 = fr_sub(fr_mul(l1, fr_sub_lazy(dx0, nx0)), dy0);
                const uint32_t bn = b + BIT_N;
                w.put_c(bn + BIT_DBL_LAMDA, fr_pack_canon(l0)); w.put_c(bn + BIT_DBL_OUT0, fr_pack_canon(nx0));
                L.put(s0 + LS_DY0, ed_put0(w, bn + BIT_DBL_OUT1, ny0));
                L.put(s0 + LS_DX1, nx1);
                if constexpr (ls_keep_dx0<G>()) L.put(s0 + LS_DX0, nx0);
            }
        }
    }
}
// end of a segment for signature g (the tail of seg_any_lock): p = the segment's start point (Edwards); returns its output and
// the last doubler output (Montgomery form)
template <int G>
__device__ HZ_ED_CALL PtA seg_lds_fin(const EdK& K, const UnitIO& io, const SegAnyOff& o, uint32_t bit0, const PtA& p, const LaneLds& L, int g, PtA* dbl) {
    const EdCtx c = K.with(io);
    const int s0 = g * ls_per_g<G>();
    dbl->x = L.get(s0 + LS_DX1);
    dbl->y = fr_scale_up(L.get(s0 + LS_DY0));
    PtA acc;
    acc.x = fr_scale_up(L.get(s0 + LS_AX)); acc.y = fr_scale_up(L.get(s0 + LS_AY));
    const PtA me = m2e_dev(c, acc);
    c.io.put_m(o.m2e, me.x); c.io.put_m(o.m2e + 1, me.y);
    PtA negp;
    negp.x = fr_neg(p.x);
    negp.y = p.y;
    const PtA ea = baby_add_dev(c, o.eadder, me, negp);
    const PtA r = bit0 ? me : ea;
    c.io.put_m(o.lastSel, r.x); c.io.put_m(o.lastSel + 1, r.y);
    return r;
}

// ---------------------------------------------------------------------------------------------------------------------------
// SegmentMulAny(n) for ONE signature without an inversion per step (launches the device does not fill: a step of seg_any_lock is
// 60 us of one wavefront, 42 of them the inversion). The circuit's recurrences are rational maps, so they are walked with
// denominators carried along -- x = X / Z^2, y = Y / Z^3 for the doubling chain D_i and for the accumulator -- and every signal is
// a numerator times a power of an inverse:
//   doubler (lamda = (3x^2 + 2Ax + 1) / 2y):  Z' = 2YZ, N = 3X^2 + 2A X Z^2 + Z^4, lamda = N / Z', X' = N^2 - A Z'^2 - 8 X Y^2,
//                                              Y' = N (4 X Y^2 - X') - 8 Y^4
//   adder (lamda = (y2 - y1) / (x2 - x1)):     U1 = X1 Z2^2, U2 = X2 Z1^2, S1 = Y1 Z2^3, S2 = Y2 Z1^3, H = U2 - U1, R = S2 - S1,
//                                              Z3 = H Z1 Z2, lamda = R / Z3, X3 = R^2 - A Z3^2 - (U1 + U2) H^2, Y3 = R (U1 H^2 - X3) - S1 H^3
// Pass 1 walks the steps forward and parks the numerators, every adder's Z3 and the running product of the Z3 in `side` (8 field
// elements per step, [step][field][limb][lane]); ONE inversion of (product of all Z3) * (last doubler Z) follows; pass 2 walks
// backward, peels the inverses off (Montgomery's trick for the Z3; 1 / Z_i = (1 / Z_{i+1}) * 2 Y_i for the doubling chain) and
// stores the signals; pass 3 copies the selector outputs forward (the selected adder outputs, already in the witness).
// Any zero divisor of the circuit (x2 = x1, y = 0) makes the product zero: that signature is walked again by seg_any_lock<1>, with
// the circuit's own 0-divisor convention and its constraint checks. Values are field elements either way: identical signals.
enum { SD_R = 0, SD_X3, SD_Y3, SD_Z3, SD_P, SD_N, SD_XN, SD_YN, SD_FIELDS };
struct SideBuf {
    uint32_t* p;
    uint32_t stride, lane;
    __device__ __forceinline__ void put(int step, int f, const Fr& v) const {
        uint32_t* q = p + (size_t)((step * SD_FIELDS + f) * 9) * stride + lane;
#pragma unroll
        for (int l = 0; l < 9; l++) q[(size_t)l * stride] = v.v[l];
    }
    __device__ __forceinline__ Fr get(int step, int f) const {
        const uint32_t* q = p + (size_t)((step * SD_FIELDS + f) * 9) * stride + lane;
        Fr r;
#pragma unroll
        for (int l = 0; l < 9; l++) r.v[l] = q[(size_t)l * stride];
        return r;
    }
};


__device__ HZ_ED_CALL void seg_any_proj(const EdK& K, const UnitIO& io, const SegAnyOff& o, const Fc& e, int e0, int n, PtA* p, PtA* dbl, const SideBuf& sd) {
    const EdCtx c = K.with(io);
    const int steps = n - 1;
    const PtA m = e2m_dev(c, *p);
    io.put_m(o.e2m, m.x); io.put_m(o.e2m + 1, m.y);
    const MDbl d0 = mont_dbl_dev(c, m);   // doubler_0
    io.put_m(o.bits + BIT_DBL_X1_2, d0.x1_2); io.put_m(o.bits + BIT_DBL_LAMDA, d0.lamda);
    io.put_m(o.bits + BIT_DBL_OUT0, d0.out.x); io.put_m(o.bits + BIT_DBL_OUT1, d0.out.y);
    const Fr A2 = fr_dbl(K.A);
    // ---- pass 1
    Fr X1 = d0.out.x, Y1 = d0.out.y, Z1 = K.one;
    Fr X2 = m.x, Y2 = m.y, Z2 = K.one, Z2s = K.one, Z2c = K.one;
    Fr P = K.one;
#pragma unroll 1
    for (int i = 0; i < steps; i++) {
        const Fr Z1s = fr_sqr(Z1), Z1c = fr_mul(Z1s, Z1);
        const Fr U1 = fr_mul(X1, Z2s), U2 = fr_mul(X2, Z1s), S1 = fr_mul(Y1, Z2c), S2 = fr_mul(Y2, Z1c);
        const Fr H = fr_sub(U2, U1), R = fr_sub(S2, S1);
        const Fr Z3 = fr_mul(H, fr_mul(Z1, Z2));
        const Fr Hs = fr_sqr(H), Hc = fr_mul(Hs, H), Z3s = fr_sqr(Z3), U1Hs = fr_mul(U1, Hs);
        const Fr X3 = fr_sub(fr_sub(fr_sqr(R), fr_mul(K.A, Z3s)), fr_mul(fr_add(U1, U2), Hs));
        const Fr Y3 = fr_sub(fr_mul(R, fr_sub(U1Hs, X3)), fr_mul(S1, Hc));
        P = fr_mul(P, Z3);
        sd.put(i, SD_R, R); sd.put(i, SD_X3, X3); sd.put(i, SD_Y3, Y3); sd.put(i, SD_Z3, Z3); sd.put(i, SD_P, P);
        if (i + 1 < steps) {
            const Fr Ys = fr_sqr(Y1), XYs = fr_mul(X1, Ys), Y4 = fr_sqr(Ys), XX = fr_sqr(X1);
            const Fr N = fr_add(fr_add(fr_add(fr_dbl(XX), XX), fr_mul(A2, fr_mul(X1, Z1s))), fr_sqr(Z1s));
            const Fr Zn = fr_dbl(fr_mul(Y1, Z1));
            const Fr XYs4 = fr_dbl(fr_dbl(XYs));
            const Fr Xn = fr_sub(fr_sub(fr_sqr(N), fr_mul(K.A, fr_sqr(Zn))), fr_dbl(XYs4));
            const Fr Yn = fr_sub(fr_mul(N, fr_sub(XYs4, Xn)), fr_dbl(fr_dbl(fr_dbl(Y4))));
            sd.put(i, SD_N, N); sd.put(i, SD_XN, Xn); sd.put(i, SD_YN, Yn);
            X1 = Xn; Y1 = Yn; Z1 = Zn;
        }
        if (c_bit(e, e0 + i + 1)) { X2 = X3; Y2 = Y3; Z2 = Z3; Z2s = Z3s; Z2c = fr_mul(Z3s, Z3); }
    }
    // ---- the one inversion: every adder's Z3 and the Z of the last doubler output (Z1 now)
    const Fr total = fr_mul(P, Z1);
    if (fr_is_zero(total)) {
        seg_any_lock<1>(K, &io, o, &e, e0, n, p, dbl);
        return;
    }
    const Fr tinv = fr_inv(total);
    Fr iZ = fr_mul(tinv, P);      // 1 / Z of dout_{steps-1}
    Fr Rr = fr_mul(tinv, Z1);     // 1 / (Z3_0 ... Z3_{steps-1})
    // ---- pass 2
    PtA fin = m, dlast = d0.out;
    bool have_fin = false;
#pragma unroll 1
    for (int i = steps - 1; i >= 0; i--) {
        const uint32_t b = o.bits + BIT_N * i;
        {
            const Fr iZ3 = i > 0 ? fr_mul(Rr, sd.get(i - 1, SD_P)) : Rr;
            Rr = fr_mul(Rr, sd.get(i, SD_Z3));
            const Fr t = fr_sqr(iZ3);
            PtA ao;
            ao.x = fr_mul(sd.get(i, SD_X3), t);
            ao.y = fr_mul(sd.get(i, SD_Y3), fr_mul(t, iZ3));
            io.put_m(b + BIT_ADD_LAMDA, fr_mul(sd.get(i, SD_R), iZ3)); io.put_m(b + BIT_ADD_OUT0, ao.x); io.put_m(b + BIT_ADD_OUT1, ao.y);
            if (!have_fin && c_bit(e, e0 + i + 1)) { fin = ao; have_fin = true; }
        }
        if (i + 1 < steps) {   // doubler_{i+1}: iZ = 1 / Z of its output
            const uint32_t bn = b + BIT_N;
            const Fr t = fr_sqr(iZ);
            PtA no;
            no.x = fr_mul(sd.get(i, SD_XN), t);
            no.y = fr_mul(sd.get(i, SD_YN), fr_mul(t, iZ));
            io.put_m(bn + BIT_DBL_LAMDA, fr_mul(sd.get(i, SD_N), iZ)); io.put_m(bn + BIT_DBL_OUT0, no.x); io.put_m(bn + BIT_DBL_OUT1, no.y);
            if (i + 2 < steps) io.put_m(bn + BIT_N + BIT_DBL_X1_2, fr_sqr(no.x));   // doubler_{i+2}.x1_2 = x_{i+1}^2
            if (i + 2 == steps) dlast = no;
            iZ = fr_mul(iZ, fr_dbl(i > 0 ? sd.get(i - 1, SD_YN) : d0.out.y));       // Z_{i+1} = 2 Y_i Z_i
        }
    }
    if (steps > 1) io.put_m(o.bits + BIT_N + BIT_DBL_X1_2, fr_sqr(d0.out.x));
    // ---- pass 3: selector outputs = the accumulator after each step
    {
        Fc cx = fr_to_canon(m.x), cy = fr_to_canon(m.y);
#pragma unroll 1
        for (int i = 0; i < steps; i++) {
            const uint32_t b = o.bits + BIT_N * i;
            if (c_bit(e, e0 + i + 1)) { cx = io.in_c(b + BIT_ADD_OUT0); cy = io.in_c(b + BIT_ADD_OUT1); }
            io.put_c(b + BIT_SEL_OUT0, cx); io.put_c(b + BIT_SEL_OUT1, cy);
        }
    }
    *dbl = dlast;
    const PtA me = m2e_dev(c, fin);
    io.put_m(o.m2e, me.x); io.put_m(o.m2e + 1, me.y);
    PtA negp;
    negp.x = fr_neg(p->x);
    negp.y = p->y;
    const PtA ea = baby_add_dev(c, o.eadder, me, negp);
    const PtA r = c_bit(e, e0) ? me : ea;
    io.put_m(o.lastSel, r.x); io.put_m(o.lastSel + 1, r.y);
    *p = r;
}

// SegmentMulFix on the constant base for G signatures in lockstep: window tables from
// HZ_BJJ_FIX_WIN; the G additions of one window share one inversion.
__device__ __forceinline__ uint32_t fix_window_bits(const Fc& e, int e0, int nbits, int i) {
    uint32_t k = 0;
    for (int j = 0; j < 3; j++) {
        const uint32_t bit = (3 * i + j < nbits) ? c_bit(e, e0 + 3 * i + j) : 0u;
        k |= bit << j;
    }
    return k;
}
template <int G>
__device__ HZ_ED_CALL void seg_fix_lock(const EdK& K, const UnitIO* io, const SegFixOff& o, const Fc* e, int e0, int nbits, int win0, int seg, PtA* out) {
    // the chain in scale 0 (see seg_any_lock): window points from the plain table, one reduction per window (lamda) instead of five
    PtA acc[G];
    const Fr A0 = fr_limbs_u64(168698);
#pragma unroll 1
    for (int g = 0; g < G; g++) {
        acc[g].x = ld_const(HZ_BJJ_FIX_DBLLAST0[2 * seg]);
        acc[g].y = ld_const(HZ_BJJ_FIX_DBLLAST0[2 * seg + 1]);
    }
#pragma unroll 1
    for (int i = 0; i < (int)o.nwin; i++) {
        const uint32_t wb = o.windows + WIN_N * i;
        Fr inv[G];
        uint32_t zmask = 0;
#pragma unroll 1
        for (int g = 0; g < G; g++) {
            const uint32_t k = fix_window_bits(e[g], e0, nbits, i);
            const Fr mx = ld_const(HZ_BJJ_FIX_WIN0[((win0 + i) * 8 + k) * 2]);
            inv[g] = fr_sub(mx, acc[g].x);
            if (fr_is_zero(inv[g])) zmask |= 1u << g;
        }
        batch_inv<G>(inv, G);   // scale-0 divisors in, scale-2 inverses out
#pragma unroll 1
        for (int g = 0; g < G; g++) {
            const UnitIO& w = io[g];
            const uint32_t k = fix_window_bits(e[g], e0, nbits, i);
            PtA mo;
            mo.x = ld_const(HZ_BJJ_FIX_WIN0[((win0 + i) * 8 + k) * 2]);
            mo.y = ld_const(HZ_BJJ_FIX_WIN0[((win0 + i) * 8 + k) * 2 + 1]);
            w.put_bit(wb + WIN_S10, (k & 1) & ((k >> 1) & 1));
            w.put_c(wb + WIN_MUX0, fr_pack_canon(mo.x)); w.put_c(wb + WIN_MUX1, fr_pack_canon(mo.y));
            const Fr num = fr_sub_lazy(mo.y, acc[g].y);   // a multiplicand only (fr.h "lazily reduced sums")
            const Fr l1 = fr_mul(num, inv[g]);
            const Fr l0 = fr_canon_limbs(l1);
            if ((zmask >> g) & 1) w.chk(C_RTX_SIG_EC, fr_zero(), fr_scale_up(num));
            PtA ao;
            ao.x = ed_put0(w, wb + WIN_ADD_OUT0, fr_sub3(fr_mul(l0, l1), A0, acc[g].x, mo.x));
            ao.y = fr_sub(fr_mul(l1, fr_sub_lazy(acc[g].x, ao.x)), acc[g].y);
            w.put_c(wb + WIN_ADD_LAMDA, fr_pack_canon(l0));
            acc[g].x = ao.x; acc[g].y = ed_put0(w, wb + WIN_ADD_OUT1, ao.y);
        }
    }
#pragma unroll 1
    for (int g = 0; g < G; g++) { acc[g].x = fr_scale_up(acc[g].x); acc[g].y = fr_scale_up(acc[g].y); }
#pragma unroll 1
    for (int g = 0; g < G; g++) {
        const EdCtx c = K.with(io[g]);
        const PtA me = m2e_dev(c, acc[g]);
        c.io.put_m(o.m2e, me.x); c.io.put_m(o.m2e + 1, me.y);
        PtA cn;
        cn.x = ld_const(HZ_BJJ_FIX_CNEG[2 * seg]);
        cn.y = ld_const(HZ_BJJ_FIX_CNEG[2 * seg + 1]);
        out[g] = baby_add_dev(c, o.cAdd, me, cn);
    }
}

// ---------------------------------------------------------------------------------------------------------------------------
// The fixed-base half with G > 1 signatures per lane in lockstep (throughput launches, k_eddsa_fix<8>): what a signature carries from
// one turn to the next -- its accumulator, the prefix product of the window's shared inversion, its scalar bits, the first segment's
// output -- is 25 field elements for eight signatures and cannot live in registers. seg_fix_lock keeps it in arrays indexed by g,
// which the compiler puts in PRIVATE memory (2.6 KB per lane, the largest scratch frame of the step until round 6); the LDS is taken
// (two ladder wavefronts of 70 KB per CU). Here the same state sits in a GLOBAL buffer of the context (EddsaArgs::side, unused by the
// throughput ladder), slot s limb l of lane L at p[(s * 9 + l) * stride + L]: the same coalesced traffic, addressed explicitly, and no
// scratch reservation on the queue. Values and their order of evaluation are seg_fix_lock's.
struct LaneMem {
    uint32_t* p;       // buffer + lane
    uint32_t stride;   // lanes of the launch, rounded up to a wavefront
    __device__ __forceinline__ Fr get(int slot) const {
        Fr r;
#pragma unroll
        for (int l = 0; l < 9; l++) r.v[l] = p[(size_t)(slot * 9 + l) * stride];
        return r;
    }
    __device__ __forceinline__ void put(int slot, const Fr& v) const {
#pragma unroll
        for (int l = 0; l < 9; l++) p[(size_t)(slot * 9 + l) * stride] = v.v[l];
    }
    __device__ __forceinline__ uint32_t getw(int word) const { return p[(size_t)word * stride]; }
    __device__ __forceinline__ void putw(int word, uint32_t v) const { p[(size_t)word * stride] = v; }
};
template <int G> struct FixMem {   // slots, in field elements: accumulator (x, y) per signature, prefix products, first-segment outputs, scalar bits
    static constexpr int ACC = 0, PRE = 2 * G, Q = 3 * G, BITS = 5 * G, SLOTS = 5 * G + (8 * G + 8) / 9;
};
// window i of signature g's scalar: bits e0 + 3 i .. + 2 (those at or above nbits read as zero), from the words parked at FixMem::BITS
template <int G>
__device__ __forceinline__ uint32_t lm_window_bits(const LaneMem& M, int g, int e0, int nbits, int i) {
    const int pos = e0 + 3 * i, w = pos >> 5, sh = pos & 31;
    const int w0 = FixMem<G>::BITS * 9 + g * 8;
    uint64_t v = M.getw(w0 + w);
    if (sh > 29 && w + 1 < 8) v |= (uint64_t)M.getw(w0 + w + 1) << 32;
    uint32_t k = (uint32_t)(v >> sh) & 7u;
    const int left = nbits - 3 * i;   // bits of this window that belong to the segment
    if (left < 3) k &= (1u << (left > 0 ? left : 0)) - 1u;
    return k;
}
template <int G, class MkIo, class Sink>
__device__ __forceinline__ void seg_fix_mem(const EdK& K, const MkIo& mk_io, const SegFixOff& o, const LaneMem& M, int e0, int nbits, int win0, int seg, const Sink& sink) {
    using FM = FixMem<G>;
    const Fr A0 = fr_limbs_u64(168698);
    {
        const Fr x0 = ld_const(HZ_BJJ_FIX_DBLLAST0[2 * seg]), y0 = ld_const(HZ_BJJ_FIX_DBLLAST0[2 * seg + 1]);
#pragma unroll 1
        for (int g = 0; g < G; g++) { M.put(FM::ACC + 2 * g, x0); M.put(FM::ACC + 2 * g + 1, y0); }
    }
#pragma unroll 1
    for (int i = 0; i < (int)o.nwin; i++) {
        const uint32_t wb = o.windows + WIN_N * i;
        // one inversion for the G divisors of the window (batch_inv's two passes): forward = prefix products, parked; backward = each
        // signature's inverse peeled off and used at once, its divisor evaluated again (a table load and a subtraction)
        Fr prod = fr_one();
#pragma unroll 1
        for (int g = 0; g < G; g++) {
            const uint32_t k = lm_window_bits<G>(M, g, e0, nbits, i);
            const Fr d = fr_sub(ld_const(HZ_BJJ_FIX_WIN0[((win0 + i) * 8 + k) * 2]), M.get(FM::ACC + 2 * g));
            M.put(FM::PRE + g, prod);
            if (!fr_is_zero(d)) prod = fr_mul(prod, d);
        }
        Fr inv = fr_inv(prod);   // scale-0 divisors in, scale-2 inverses out
#pragma unroll 1
        for (int g = G - 1; g >= 0; g--) {
            const UnitIO w = mk_io(g);
            const uint32_t k = lm_window_bits<G>(M, g, e0, nbits, i);
            PtA mo;
            mo.x = ld_const(HZ_BJJ_FIX_WIN0[((win0 + i) * 8 + k) * 2]);
            mo.y = ld_const(HZ_BJJ_FIX_WIN0[((win0 + i) * 8 + k) * 2 + 1]);
            const Fr ax = M.get(FM::ACC + 2 * g), ay = M.get(FM::ACC + 2 * g + 1);
            const Fr d = fr_sub(mo.x, ax);
            const bool dz = fr_is_zero(d);
            Fr inv_g = fr_zero();
            if (!dz) {
                inv_g = fr_mul(inv, M.get(FM::PRE + g));
                inv = fr_mul(inv, d);
            }
            w.put_bit(wb + WIN_S10, (k & 1) & ((k >> 1) & 1));
            w.put_c(wb + WIN_MUX0, fr_pack_canon(mo.x)); w.put_c(wb + WIN_MUX1, fr_pack_canon(mo.y));
            const Fr num = fr_sub_lazy(mo.y, ay);   // a multiplicand only (fr.h "lazily reduced sums")
            const Fr l1 = fr_mul(num, inv_g);
            const Fr l0 = fr_canon_limbs(l1);
            if (dz) w.chk(C_RTX_SIG_EC, fr_zero(), fr_scale_up(num));
            const Fr nx = ed_put0(w, wb + WIN_ADD_OUT0, fr_sub3(fr_mul(l0, l1), A0, ax, mo.x));
            const Fr ny = fr_sub(fr_mul(l1, fr_sub_lazy(ax, nx)), ay);
            w.put_c(wb + WIN_ADD_LAMDA, fr_pack_canon(l0));
            M.put(FM::ACC + 2 * g, nx); M.put(FM::ACC + 2 * g + 1, ed_put0(w, wb + WIN_ADD_OUT1, ny));
        }
    }
#pragma unroll 1
    for (int g = 0; g < G; g++) {
        const UnitIO io = mk_io(g);
        const EdCtx c = K.with(io);
        PtA acc;
        acc.x = fr_scale_up(M.get(FM::ACC + 2 * g)); acc.y = fr_scale_up(M.get(FM::ACC + 2 * g + 1));
        const PtA me = m2e_dev(c, acc);
        c.io.put_m(o.m2e, me.x); c.io.put_m(o.m2e + 1, me.y);
        PtA cn;
        cn.x = ld_const(HZ_BJJ_FIX_CNEG[2 * seg]);
        cn.y = ld_const(HZ_BJJ_FIX_CNEG[2 * seg + 1]);
        sink(g, c, baby_add_dev(c, o.cAdd, me, cn));
    }
}

// Everything before the scalar multiplications of one signature: AySign2Ax, the S range check,
// the message hash and its bits, 8*A and the zero-point substitution.
struct EdSig {
    Fc h_c;
    Fr enabled, zp;
    PtA R8, p0;
};
// AySign2Ax (src/lib/utils-bjj.circom:37-58) with circomlib's Bits2Point_Strict: x from y and the sign, both alias checks
__device__ __forceinline__ Fr ay_sign_2_ax_dev(const EdCtx& c, const UnitIO& io, const EddsaOff& o, const Fr& aySig, const Fr& signSig) {
    const Fc ay_c = fr_to_canon(aySig);
    for (int k = 0; k < 254; k++) io.put_bit(o.ax_n2bAy + k, c_bit(ay_c, k));
    if (comp_constant_dev(io, o.ax_aliasY, ay_c, CT_MINUS1_D)) report_fail(io.err, io.inst, io.err_unit, C_RTX_AX_ALIAS_Y, c.one, fr_zero());
    const Fr y = aySig;
    const Fr y2 = fr_sqr(y);
    Fr x = fr_sqrt_circom_dev(fr_div(fr_sub(c.one, y2), fr_sub(c.a, fr_mul(c.d, y2))));
    if (fr_eq(signSig, c.one)) x = fr_neg(x);
    const Fc x_c = fr_to_canon(x);
    io.put_c(o.ax_x, x_c);
    const Fr x2 = fr_sqr(x);
    io.put_m(o.ax_x2, x2); io.put_m(o.ax_y2, y2);
    io.chk(C_RTX_AX_BABYCHECK, fr_add(fr_mul(c.a, x2), y2), fr_add(c.one, fr_mul(fr_mul(c.d, x2), y2)));
    for (int k = 0; k < 254; k++) io.put_bit(o.ax_n2bX + k, c_bit(x_c, k));
    if (comp_constant_dev(io, o.ax_aliasX, x_c, CT_MINUS1_D)) report_fail(io.err, io.inst, io.err_unit, C_RTX_AX_ALIAS_X, c.one, fr_zero());
    const uint32_t sg = comp_constant_dev(io, o.ax_signCalc, x_c, CT_HALF_D);
    io.chk(C_RTX_AX_SIGN, fr_from_bit(sg), signSig);
    return x;
}

// the prologue without the message hash: AySign2Ax, 8A, the zero checks and the point the ladder starts from; returns Ax
__device__ __forceinline__ Fr ed_prologue_point(const EdK& K, const UnitIO& io, const EddsaOff& o, const Fr& enabled, const Fr& signSig, const Fr& aySig, const Fr& Ay,
                                                EdSig& out, bool* on_curve) {
    const EdCtx c = K.with(io);
    const Fr x = ay_sign_2_ax_dev(c, io, o, aySig, signSig);
    PtA A;
    A.x = x; A.y = Ay;
    {
        const Fr x2 = fr_sqr(x), y2 = fr_sqr(Ay);   // is the point the ladder starts from on the curve? (the fast doubling chain)
        *on_curve = fr_eq(fr_add(fr_mul(c.a, x2), y2), fr_add(c.one, fr_mul(fr_mul(c.d, x2), y2)));
    }
    const PtA d1 = baby_add_dev(c, o.dbl1, A, A);
    const PtA d2 = baby_add_dev(c, o.dbl2, d1, d1);
    const PtA d3 = baby_add_dev(c, o.dbl3, d2, d2);
    // isZero.in <== dbl3.x (the input x of the third doubling) ; zeropoint.in <== dbl3.xout
    Fr z[2] = {d2.x, d3.x};
    Fr zi[2] = {z[0], z[1]};
    inv_pair(zi[0], zi[1]);
    const Fr az = is_zero_dev(io, o.isZero, z[0], zi[0]);
    io.chk_zero(C_RTX_SIG_A_NONZERO, fr_mul(az, enabled));
    out.zp = is_zero_dev(io, o.zeropoint, z[1], zi[1]);
    const bool zpb = fr_is_zero(z[1]);
    out.p0.x = zpb ? ld_const(HZ_BJJ_BASE8_X) : d3.x;
    out.p0.y = zpb ? ld_const(HZ_BJJ_BASE8_Y) : d3.y;
    io.put_m(o.seg0p, out.p0.x); io.put_m(o.seg0p + 1, out.p0.y);
    out.enabled = enabled;
    return x;
}
// the message hash of EdDSAPoseidonVerifier (the S decomposition and range check belong to k_eddsa_fix) and its bits
__device__ __forceinline__ void ed_prologue_hash(const UnitIO& io, const EddsaOff& o, const Fr* K6, const Fr& R8x, const Fr& R8y, const Fr& x, const Fr& Ay, const Fr& M, EdSig& out) {
    Fr hin[5] = {R8x, R8y, x, Ay, M};
    WitSboxSink s6 = io.sbox_sink(o.hash);
    const Fr h = poseidon_hash<6>(hin, K6, s6);
    out.h_c = fr_to_canon(h);
    num2bits_strict_dev(io, o.h2bits, out.h_c, C_RTX_SIG_H_ALIAS);
    out.R8.x = R8x; out.R8.y = R8y;
}

// ---------------------------------------------------------------------------------------------------------------------------
// h * 8A = the two SegmentMulAny of circomlib's EscalarMulAny(254): bits 0..147 from 8A, bits 148..253 from 2^148 * 8A. The second
// segment's start is the END of the first one's doubling chain, so a signature used to be one 254-step dependent chain -- the longest
// chain of a step and of a single batch's latency. But the doublings do not depend on the additions: k_eddsa_pre walks them alone,
// 147 inversion-free projective doublings (8 products each) and one inversion, and the two segments then run as independent lanes of
// k_eddsa_ladder: 148 and 106 steps instead of 254, twice the wavefronts. Values are field elements, so the affine point the
// circuit reaches by 147 affine doublings is the same number however it is computed -- as long as the chain is regular: 8A lies on
// the curve (projective twisted-Edwards doubling is then exception free) and is not the identity (replaced by Base8 by the circuit
// itself). Off-curve keys (a witness that fails BabyCheck anyway) take the circuit's own affine doubling chain instead (ed_dbl_chain_affine).
struct PtP { Fr X, Y, Z; };
// dbl-2008-hwcd for a*x^2 + y^2 = 1 + d*x^2*y^2 without the T coordinate (doubling only): 3M + 4S + 1 small-constant product
// Every sum here only feeds products (operands below 2^257 > 10 p): none is reduced (fr.h "lazily reduced sums"). With the
// coordinates below 2p: A, B, D and the squares are below 1.1 p; E = (X + Y)^2 - A - B + 4p < 5.2 p, G = D + B < 2.2 p,
// F = G - 2 Z^2 + 4p < 6.2 p, H = D - B + 2p < 3.1 p; the three products come out below 1.2 p again.
__device__ __forceinline__ PtP ed_dbl_proj(const EdK& K, const PtP& p) {
    const Fr A = fr_sqr(p.X), B = fr_sqr(p.Y), C = fr_dbl_lazy(fr_sqr(p.Z)), D = fr_mul(K.a, A);
    const Fr E = fr_sub2_lazy(fr_sqr(fr_add_lazy(p.X, p.Y)), A, B), G = fr_add_lazy(D, B);
    const Fr zero = fr_zero();
    const Fr F = fr_sub2_lazy(G, C, zero), H = fr_sub_lazy(D, B);
    PtP r;
    r.X = fr_mul(E, F); r.Y = fr_mul(G, H); r.Z = fr_mul(F, G);
    return r;
}
// the circuit's own chain: e2m, then `count` Montgomery doublings with the 0-divisor convention (no signals, no checks: the
// segment lanes store and check every one of these steps again)
__device__ HZ_ED_CALL PtA ed_dbl_chain_affine(const EdK& K, const PtA& p0, int count) {
    Fr den[2] = {fr_sub(K.one, p0.y), p0.x};
    inv_pair(den[0], den[1]);
    PtA m;
    m.x = fr_mul(fr_add(K.one, p0.y), den[0]);
    m.y = fr_mul(m.x, den[1]);
    const Fr A2 = fr_dbl(K.A);
    for (int i = 0; i < count; i++) {
        if (fr_is_zero(m.y)) {
            // 2y = 0: the quotient is 0 by the 0-divisor convention, so x' = -A - 2x and y' = -y = 0 -- and it stays that way. This is
            // the chain of every lane without a signature (L1 and padding transactions: all-zero key), no inversion needed.
            m.x = fr_sub(fr_neg(K.A), fr_dbl(m.x));
            m.y = fr_zero();
            continue;
        }
        const Fr x2 = fr_sqr(m.x);
        const Fr num = fr_add(fr_add(fr_add(fr_dbl(x2), x2), fr_mul(A2, m.x)), K.one);
        const Fr lamda = fr_div(num, fr_dbl(m.y));
        PtA o;
        o.x = fr_sub(fr_sub(fr_sqr(lamda), K.A), fr_dbl(m.x));
        o.y = fr_sub(fr_mul(lamda, fr_sub(m.x, o.x)), m.y);
        m = o;
    }
    return m;
}
#ifndef HZ_ED_FORCE_AFFINE_PRE
#define HZ_ED_FORCE_AFFINE_PRE 0   // test builds: every lane takes the affine chain (the path off-curve keys take)
#endif
// Off-curve start (the key of a lane without a signature: ay forced to 0 for AySign2Ax, the leaf's own ay for the verifier): the
// curve equation cannot be used, but the circuit's affine doubling x' = l^2 - A - 2x, y' = l (x - x') - y, l = (3x^2 + 2Ax + 1) / (2y)
// is a rational map: carried as (X : Y : Z) with x = X/Z, y = Y/Z it needs no inversion either (4S + 12M per step). A zero divisor
// anywhere (the circuit's quotient is then 0 by convention, not a rational value) makes Z = 0 for good: then, and only then, the
// affine chain is walked (its y = 0 case is a linear recurrence).
__device__ HZ_ED_CALL PtA ed_dbl_chain_generic(const EdK& K, const PtA& p0, int count) {
    Fr den[2] = {fr_sub(K.one, p0.y), p0.x};
    inv_pair(den[0], den[1]);
    PtA m;
    m.x = fr_mul(fr_add(K.one, p0.y), den[0]);
    m.y = fr_mul(m.x, den[1]);
    Fr X = m.x, Y = m.y, Z = K.one;
    const Fr A2 = fr_dbl(K.A);
#pragma unroll 1
    for (int i = 0; i < count; i++) {
        const Fr XX = fr_sqr(X), XZ = fr_mul(X, Z), ZZ = fr_sqr(Z);
        const Fr N = fr_add(fr_add(fr_add(fr_dbl(XX), XX), fr_mul(A2, XZ)), ZZ);
        const Fr D = fr_dbl(fr_mul(Y, Z));
        const Fr D2 = fr_sqr(D), D3 = fr_mul(D2, D);
        const Fr X3n = fr_sub(fr_mul(fr_sqr(N), Z), fr_mul(fr_add(fr_mul(K.A, Z), fr_dbl(X)), D2));
        const Fr Yn = fr_sub(fr_mul(N, fr_sub(fr_mul(X, D2), X3n)), fr_mul(Y, D3));
        X = fr_mul(X3n, D);
        Y = Yn;
        Z = fr_mul(D3, Z);
    }
    if (fr_is_zero(Z)) return ed_dbl_chain_affine(K, p0, count);
    const Fr zi = fr_inv(Z);
    PtA r;
    r.x = fr_mul(X, zi);
    r.y = fr_mul(Y, zi);
    return r;
}
// 2^count * p0 in Montgomery affine coordinates (p0 Edwards affine)
__device__ __forceinline__ PtA ed_dbl_chain(const EdK& K, const PtA& p0, int count, bool regular) {
    if (HZ_ED_FORCE_AFFINE_PRE) return ed_dbl_chain_affine(K, p0, count);
    if (!__all(regular)) {
        if (!regular) return ed_dbl_chain_generic(K, p0, count);
    }
    PtP q{p0.x, p0.y, K.one};
#pragma unroll 1
    for (int i = 0; i < count; i++) q = ed_dbl_proj(K, q);
    // Edwards (X/Z, Y/Z) -> Montgomery u = (Z + Y) / (Z - Y), v = u * Z / X
    Fr den[2] = {fr_sub(q.Z, q.Y), q.X};
    Fr inv[2] = {den[0], den[1]};
    inv_pair(inv[0], inv[1]);
    PtA m;
    m.x = fr_mul(fr_add(q.Z, q.Y), inv[0]);
    m.y = fr_mul(fr_mul(m.x, q.Z), inv[1]);
    if (fr_is_zero(den[0]) || fr_is_zero(den[1])) return ed_dbl_chain_affine(K, p0, count);   // cannot happen for a regular chain
    return m;
}

#ifndef HZ_ED_WAVES
#define HZ_ED_WAVES 2
#endif
#ifndef HZ_ED_FIX_G
#define HZ_ED_FIX_G 8   // signatures per lane of the fixed-base half in throughput launches
#endif
// the kernels of the SPLIT form (launches of at most HZ_ED_SPLIT_MAX signatures: a few dozen wavefronts, one long dependent chain each).
// At ONE wavefront per SIMD (512 registers: what would spill goes to the accumulation registers) they use no private memory at all and
// one batch's ladder takes 5.70 instead of 5.84 ms -- but two flagged contexts in flight lose 3 % (403 k against 415-419 k tx/s,
// profiles/r06_eddsa_inline_ab.txt): the default stays two, -DHZ_ED_WAVES_LAT=1 is the scratch-free build.
#ifndef HZ_ED_WAVES_LAT
#define HZ_ED_WAVES_LAT 2
#endif
// lane = signature: everything before the scalar multiplication, and the start of its second segment
__global__ __launch_bounds__(HZ_BLOCK) __attribute__((amdgpu_waves_per_eu(HZ_ED_WAVES))) void k_eddsa_pre(const EddsaArgs a) {
    const Fr* K6 = poseidon_consts_w<6>();
    const uint32_t n = a.ucnt ? a.ucnt : a.n_units;
    const uint32_t li = blockIdx.x * blockDim.x + threadIdx.x;
    if (li >= n) return;
    const uint32_t i = a.u0 + li;
    const EdK K = ed_k();
    const UnitIO io{a.base, a.n_units, i, i / a.upi, i % a.upi, a.err};
    const Scratch sc{a.scratch, a.n_units, i};
    // Three phases that hand over through the inter-kernel scratch instead of through registers (what one phase leaves for the kernels
    // behind it is stored when it is known, what the next phase needs is loaded where it is used): the point half, the message hash --
    // only Ax crosses from the first into the second --, the doubling chain from the stored 8A. Held in one EdSig across the width-6
    // Poseidon, the prologue's outputs were 62 registers of live values and 1.9 KB of private memory (round 5).
    bool on_curve = false;
    Fr x;
    {
        EdSig sg;
        x = ed_prologue_point(K, io, a.ed, sc.get(SC_ED_ENABLED), sc.get(SC_ED_SIGN), sc.get(SC_ED_AYSIG), sc.get(SC_ED_AY), sg, &on_curve);
        sc.set(SC_ED_ZP, sg.zp);
        sc.set(SC_ED_P0X, sg.p0.x); sc.set(SC_ED_P0Y, sg.p0.y);
    }
    {
        EdSig sg;
        ed_prologue_hash(io, a.ed, K6, sc.get(SC_ED_R8X), sc.get(SC_ED_R8Y), x, sc.get(SC_ED_AY), sc.get(SC_SIGL2HASH), sg);
        sc.set(SC_ED_H, fr_from_canon(sg.h_c));
    }
    if (a.chain_in_ladder) {   // small launches: the second segment's lane walks the doubling chain itself (k_eddsa_ladder)
        sc.set(SC_ED_DBLX, on_curve ? K.one : fr_zero());
        return;
    }
    // 8A of an on-curve A is on the curve; when it is the identity the circuit substitutes Base8 (zp = 1): regular either way
    PtA p0;
    p0.x = sc.get(SC_ED_P0X); p0.y = sc.get(SC_ED_P0Y);
    const PtA d147 = ed_dbl_chain(K, p0, 147, on_curve);
    sc.set(SC_ED_DBLX, d147.x); sc.set(SC_ED_DBLY, d147.y);
}

// RollupMain, small launches: the prologue as two kernels (EddsaArgs.in_*). The signature's point half needs four values of the front
// step -- verifySignEnabled and the sender's sign / ay behind the new-account multiplexers -- and they are a handful of products of
// INPUTS (src/rollup-tx-states.circom:99-130, src/rollup-tx.circom:318-335: finalFromIdx = Mux1(fromIdx, auxFromIdx, onChain * newAccount),
// verifySignEnabled = (1 - onChain) * (1 - IsZero(finalFromIdx)), s1Sign / s1Ay = Mux1(sign1 / ay1, the key's, isP1Insert)): recomputed
// here (values only, the front kernel stores the signals) this kernel starts WITH the front kernel instead of after it, and the
// chain front -> prologue -> ladder of a single batch loses a millisecond. The key's bits are read only when isP1Insert is not zero
// (a product with zero is zero whatever the other operand).
__global__ __launch_bounds__(HZ_BLOCK) __attribute__((amdgpu_waves_per_eu(HZ_ED_WAVES_LAT))) void k_eddsa_pre_a(const EddsaArgs a) {
    const uint32_t n = a.ucnt ? a.ucnt : a.n_units;
    const uint32_t li = blockIdx.x * blockDim.x + threadIdx.x;
    if (li >= n) return;
    const uint32_t i = a.u0 + li;
    const EdK K = ed_k();
    const UnitIO io{a.base, a.n_units, i, i / a.upi, i % a.upi, a.err};
    const Scratch sc{a.scratch, a.n_units, i};
    const Fr one = K.one;
    const Fr onChain = io.in_m(a.in_onChain), newAccount = io.in_m(a.in_newAccount);
    const Fr fromIdx = fr_from_u64(c_bits64(io.in_c(a.in_txCompressedData), 48, 48));
    const Fr sel = fr_mul(onChain, newAccount);   // selFromIdx.s = isP1Insert
    const Fr finalFromIdx = mux1_dev(fromIdx, io.in_m(a.in_auxFromIdx), sel);
    const Fr enabled = fr_is_zero(finalFromIdx) ? fr_zero() : fr_sub(one, onChain);   // (1 - onChain) * (1 - isZero)
    Fr bjjAy = fr_zero(), bjjSign = fr_zero();
    if (!fr_is_zero(sel)) {
        // BitsCompressed2AySign as the front kernel evaluates it (tx_dev.h rollup_tx_front_dev, phase E): bits 0..253, bit 255
        Fc packed = fc_zero();
        bool all_bool = true;
        for (int k = 0; k < 254; k++) {
            const Fc b = io.in_c(a.in_fromBjjCompressed + k);
            bool is1 = b.v[0] == 1u, is0 = b.v[0] == 0u;
            for (int q = 1; q < 8; q++) { is1 = is1 && b.v[q] == 0u; is0 = is0 && b.v[q] == 0u; }
            if (!(is0 || is1)) all_bool = false;
            if (is1) packed.v[k >> 5] |= 1u << (k & 31);
        }
        if (all_bool) {
            fc_cond_sub_p(packed.v);
            bjjAy = fr_from_canon(packed);
        } else {
            Fr acc = fr_zero();
            for (int k = 253; k >= 0; k--) acc = fr_add(fr_dbl(acc), io.in_m(a.in_fromBjjCompressed + k));
            bjjAy = acc;
        }
        bjjSign = io.in_m(a.in_fromBjjCompressed + 255);
    }
    const Fr s1Sign = mux1_dev(io.in_m(a.in_sign1), bjjSign, sel), s1Ay = mux1_dev(io.in_m(a.in_ay1), bjjAy, sel);
    EdSig sg;
    bool on_curve = false;
    const Fr x = ed_prologue_point(K, io, a.ed, enabled, fr_mul(s1Sign, enabled), fr_mul(s1Ay, enabled), s1Ay, sg, &on_curve);
    sc.set(SC_ED_ZP, sg.zp);
    sc.set(SC_ED_P0X, sg.p0.x); sc.set(SC_ED_P0Y, sg.p0.y);
    sc.set(SC_ED_DBLX, on_curve ? K.one : fr_zero());   // (chain_in_ladder: the second segment's lane walks the doubling chain)
    sc.set(SC_ED_DBLY, x);                              // Ax for k_eddsa_pre_b (the slot is free in this form)
}
// lane = signature: the message hash (needs the front step: R8, Ay, the message) and its bits
__global__ __launch_bounds__(HZ_BLOCK) __attribute__((amdgpu_waves_per_eu(HZ_ED_WAVES))) void k_eddsa_pre_b(const EddsaArgs a) {
    const Fr* K6 = poseidon_consts_w<6>();
    const uint32_t n = a.ucnt ? a.ucnt : a.n_units;
    const uint32_t li = blockIdx.x * blockDim.x + threadIdx.x;
    if (li >= n) return;
    const uint32_t i = a.u0 + li;
    const UnitIO io{a.base, a.n_units, i, i / a.upi, i % a.upi, a.err};
    const Scratch sc{a.scratch, a.n_units, i};
    EdSig sg;
    ed_prologue_hash(io, a.ed, K6, sc.get(SC_ED_R8X), sc.get(SC_ED_R8Y), sc.get(SC_ED_DBLY), sc.get(SC_ED_AY), sc.get(SC_SIGL2HASH), sg);
    sc.set(SC_ED_H, fr_from_canon(sg.h_c));
}

// lane = (segment, G signatures in lockstep). Lane li of a segment evaluates the signatures of units li, li + nl, li + 2 nl, ...
// (nl lanes): consecutive lanes keep writing consecutive units. A slot past the end repeats the lane's first unit (same values to
// the same addresses).
// field by field: `c ? a : b` on two structs selects an address, and the kernel's argument block would follow it into private memory
__device__ __forceinline__ SegAnyOff seg_off_select(bool first, const SegAnyOff& a, const SegAnyOff& b) {
    SegAnyOff r;
    r.e2m = first ? a.e2m : b.e2m; r.bits = first ? a.bits : b.bits; r.m2e = first ? a.m2e : b.m2e;
    r.eadder = first ? a.eadder : b.eadder;
    r.lastSel = first ? a.lastSel : b.lastSel; r.nbits = first ? a.nbits : b.nbits;
    return r;
}
template <int G>
__global__ __launch_bounds__(HZ_BLOCK) __attribute__((amdgpu_waves_per_eu(HZ_ED_WAVES_LAT))) void k_eddsa_ladder(const EddsaArgs a) {
    const uint32_t n = a.ucnt ? a.ucnt : a.n_units;
    const uint32_t nl = (n + G - 1) / G;
    const uint32_t li = blockIdx.x * blockDim.x + threadIdx.x;
    if (li >= nl) return;
    const uint32_t seg = blockIdx.y;
    const EdK K = ed_k();
    const EddsaOff& o = a.ed;
    UnitIO io[G];
    Fc h_c[G];
    PtA p[G], dbl[G];
#pragma unroll 1
    for (int g = 0; g < G; g++) {
        uint32_t ui = li + (uint32_t)g * nl;
        if (ui >= n) ui = li;
        const uint32_t i = a.u0 + ui;
        io[g] = UnitIO{a.base, a.n_units, i, i / a.upi, i % a.upi, a.err};
        const Scratch sc{a.scratch, a.n_units, i};
        h_c[g] = fr_to_canon(sc.get(SC_ED_H));
        if (seg == 0) {
            p[g].x = sc.get(SC_ED_P0X); p[g].y = sc.get(SC_ED_P0Y);
        } else {
            // the doubling between the segments and the second segment's base point (escalarmulany.circom: doublers / m2e)
            const EdCtx c = K.with(io[g]);
            PtA d147;
            if (a.chain_in_ladder) {
                // The 147 doublings between the segments' base points feed THIS lane only, and its segment is 42 steps shorter than
                // the first one's: walked here (0.6 ms of one wavefront) they are off the launch's critical path, which is the first
                // segment's lane, instead of in front of both (k_eddsa_pre). SC_ED_DBLX carries the prologue's on-curve flag.
                PtA p0;
                p0.x = sc.get(SC_ED_P0X); p0.y = sc.get(SC_ED_P0Y);
                d147 = ed_dbl_chain(K, p0, 147, !fr_is_zero(sc.get(SC_ED_DBLX)));
            } else {
                d147.x = sc.get(SC_ED_DBLX); d147.y = sc.get(SC_ED_DBLY);
            }
            const MDbl dd = mont_dbl_dev(c, d147);
            c.io.put_m(o.dblr, dd.x1_2); c.io.put_m(o.dblr + 1, dd.lamda); c.io.put_m(o.dblr + 2, dd.out.x); c.io.put_m(o.dblr + 3, dd.out.y);
            p[g] = m2e_dev(c, dd.out);
            c.io.put_m(o.m2e0, p[g].x); c.io.put_m(o.m2e0 + 1, p[g].y);
        }
    }
    // ONE call site per form (the bodies are inlined: two segments times two forms were four copies of the ladder in this kernel)
    const SegAnyOff so = seg_off_select(seg == 0, o.seg[0], o.seg[1]);
    const int e0 = seg == 0 ? 0 : 148, nb = seg == 0 ? 148 : 106;
    if (G == 1 && a.side) {
        const SideBuf sd{a.side, 2 * nl, seg * nl + li};
        seg_any_proj(K, io[0], so, h_c[0], e0, nb, p, dbl, sd);
    } else {
        seg_any_lock<G>(K, io, so, h_c, e0, nb, p, dbl);
    }
#pragma unroll 1
    for (int g = 0; g < G; g++) {
        const Scratch sc{a.scratch, a.n_units, io[g].unit};
        sc.set(seg == 0 ? SC_ED_S0X : SC_ED_S1X, p[g].x);
        sc.set(seg == 0 ? SC_ED_S0Y : SC_ED_S1Y, p[g].y);
    }
}

// The throughput form (launches above HZ_ED_SPLIT_MAX signatures): after k_eddsa_pre, lane = (segment, G signatures in lockstep). The
// G signatures of a lane take turns; whatever a signature carries from one turn to the next lives in LDS (LaneLds above), its start
// points and segment outputs in the inter-kernel scratch. 148 / 106 dependent steps instead of 254, so FOUR signatures can share an
// inversion (a quarter of one per ladder step and signature) and the chain is still shorter than one lane walking both segments of
// two signatures (round 3's k_eddsa_chain<2>: 4.03 G wave-instructions and 21.1 ms alone per 65 536 signatures; this form 2.59 G +
// 0.89 G for k_eddsa_pre -- whose prologue the chain contained -- and 17.3 ms; step 38.0 -> 36.6-37.1 ms on one box,
// profiles/r03_eddsa_seg_ab.txt). G = 4 needs 31 LDS slots = 70 KB per wavefront: two wavefronts per CU, i.e. ONE context's 512
// ladder wavefronts are resident at a time and the other context's follow -- which is how two contexts in flight alternate anyway.
// G = 3 (26 slots) and G = 2 (17) measured: 36.9-37.4 / 37.4-37.8 ms.
template <int G>
__global__ __launch_bounds__(HZ_BLOCK) __attribute__((amdgpu_waves_per_eu(HZ_ED_WAVES))) void k_eddsa_seg(const EddsaArgs a) {
    extern __shared__ __attribute__((aligned(16))) uint32_t st[];   // ls_slots<G>() * 9 * 64 words
    const uint32_t n = a.ucnt ? a.ucnt : a.n_units;
    const uint32_t nl = (n + G - 1) / G;
    const uint32_t li = blockIdx.x * blockDim.x + threadIdx.x;
    if (li >= nl) return;
    const uint32_t seg = blockIdx.y;
    const LaneLds L{(lds_u32*)(st + threadIdx.x)};
    const EdK K = ed_k();
    const EddsaOff& o = a.ed;
    typedef __attribute__((address_space(1))) uint8_t gl_u8;
    uint8_t* const wbase = (uint8_t*)(gl_u8*)a.base;
    const uint32_t n_units = a.n_units, upi = a.upi, u0 = a.u0;
    ErrBuf* const errp = a.err;
    auto mk_io = [=](int g) {
        uint32_t ui = li + (uint32_t)g * nl;
        if (ui >= n) ui = li;
        const uint32_t i = u0 + ui;
        return UnitIO{wbase, n_units, i, i / upi, i % upi, errp};
    };
    const SegAnyOff& so = o.seg[seg];
    const int e0 = seg ? 148 : 0, nb = seg ? 106 : 148;
    const int sx = seg ? SC_ED_S1X : SC_ED_S0X, sy = seg ? SC_ED_S1Y : SC_ED_S0Y;
#pragma unroll 1
    for (int g = 0; g < G; g++) {
        const UnitIO io = mk_io(g);
        const Scratch sc{a.scratch, a.n_units, io.unit};
        ls_hput(L, ls_h0<G>(), g, fr_to_canon(sc.get(SC_ED_H)));
        PtA p;
        if (seg == 0) {
            p.x = sc.get(SC_ED_P0X); p.y = sc.get(SC_ED_P0Y);
        } else {
            // the doubling between the segments and the second segment's base point (escalarmulany.circom: doublers / m2e)
            const EdCtx c = K.with(io);
            PtA d147;
            d147.x = sc.get(SC_ED_DBLX); d147.y = sc.get(SC_ED_DBLY);
            const MDbl dd = mont_dbl_dev(c, d147);
            c.io.put_m(o.dblr, dd.x1_2); c.io.put_m(o.dblr + 1, dd.lamda); c.io.put_m(o.dblr + 2, dd.out.x); c.io.put_m(o.dblr + 3, dd.out.y);
            p = m2e_dev(c, dd.out);
            c.io.put_m(o.m2e0, p.x); c.io.put_m(o.m2e0 + 1, p.y);
            sc.set(SC_ED_Q1X, p.x); sc.set(SC_ED_Q1Y, p.y);   // kept for the end of the segment
        }
        seg_lds_init<G>(K, io, so, p, L, g);
    }
    seg_lds_steps<G>(K, mk_io, so, e0, nb, L);
#pragma unroll 1
    for (int g = 0; g < G; g++) {
        const UnitIO io = mk_io(g);
        const Scratch sc{a.scratch, a.n_units, io.unit};
        PtA p, dbl;
        p.x = sc.get(seg ? SC_ED_Q1X : SC_ED_P0X); p.y = sc.get(seg ? SC_ED_Q1Y : SC_ED_P0Y);
        const PtA r = seg_lds_fin<G>(K, io, so, ls_hbit(L, ls_h0<G>(), g, e0), p, L, g, &dbl);
        sc.set(sx, r.x); sc.set(sy, r.y);
    }
}

// mulFix = S * B8 (two SegmentMulFix: 82 + 3 windows of the constant base) with the S decomposition and range check: it depends on
// nothing but S, so it runs beside the variable-base ladder in its own kernel with its own signatures-per-lane (85 inversions
// per lane: more signatures share each of them than in the 254-step ladder kernel).
template <int G>
__global__ __launch_bounds__(HZ_BLOCK) __attribute__((amdgpu_waves_per_eu(G == 1 ? HZ_ED_WAVES_LAT : HZ_ED_WAVES))) void k_eddsa_fix(const EddsaArgs a) {
    const uint32_t n = a.ucnt ? a.ucnt : a.n_units;
    const uint32_t nl = (n + G - 1) / G;
    const uint32_t li = blockIdx.x * blockDim.x + threadIdx.x;
    if (li >= nl) return;
    const EdK K = ed_k();
    const EddsaOff& o = a.ed;
    if constexpr (G > 1) {
        // several signatures per lane: their state in the context's lane buffer (seg_fix_mem above), nothing indexed by g in this kernel
        using FM = FixMem<G>;
        const LaneMem M{a.side + li, (nl + 63u) & ~63u};
        auto mk_io = [&](int g) __attribute__((always_inline)) {
            uint32_t ui = li + (uint32_t)g * nl;
            if (ui >= n) ui = li;   // a padding slot repeats the lane's first signature
            const uint32_t i = a.u0 + ui;
            return UnitIO{a.base, a.n_units, i, i / a.upi, i % a.upi, a.err};
        };
#pragma unroll 1
        for (int g = 0; g < G; g++) {
            const UnitIO io = mk_io(g);
            const Scratch sc{a.scratch, a.n_units, io.unit};
            const Fc S_c = fr_to_canon(sc.get(SC_ED_S));
            num2bits_dev(io, o.snum2bits, S_c, 253, C_RTX_SIG_N2B_S);
            const Fc S253 = c_extract(S_c, 0, 253);
            if (comp_constant_dev(io, o.sCmp, S253, CT_SUBORDER_M1_D)) io.chk_zero(C_RTX_SIG_S_RANGE, sc.get(SC_ED_ENABLED));
#pragma unroll
            for (int q = 0; q < 8; q++) M.putw(FM::BITS * 9 + g * 8 + q, S253.v[q]);
        }
        seg_fix_mem<G>(K, mk_io, o.fseg[0], M, 0, 246, 0, 0, [&](int g, const EdCtx&, const PtA& q) __attribute__((always_inline)) {
            M.put(FM::Q + 2 * g, q.x); M.put(FM::Q + 2 * g + 1, q.y);
        });
        seg_fix_mem<G>(K, mk_io, o.fseg[1], M, 246, 7, 82, 1, [&](int g, const EdCtx& c, const PtA& r) __attribute__((always_inline)) {
            PtA q;
            q.x = M.get(FM::Q + 2 * g); q.y = M.get(FM::Q + 2 * g + 1);
            const PtA left = baby_add_dev(c, o.fadders0, q, r);
            const Scratch sc{a.scratch, a.n_units, c.io.unit};
            sc.set(SC_ED_LEFTX, left.x); sc.set(SC_ED_LEFTY, left.y);
        });
        return;
    }
    UnitIO io[G];
    Fc S253[G];
    PtA q[G], r[G];
#pragma unroll 1
    for (int g = 0; g < G; g++) {
        uint32_t ui = li + (uint32_t)g * nl;
        if (ui >= n) ui = li;
        const uint32_t i = a.u0 + ui;
        io[g] = UnitIO{a.base, a.n_units, i, i / a.upi, i % a.upi, a.err};
        const Scratch sc{a.scratch, a.n_units, i};
        const Fc S_c = fr_to_canon(sc.get(SC_ED_S));
        num2bits_dev(io[g], o.snum2bits, S_c, 253, C_RTX_SIG_N2B_S);
        S253[g] = c_extract(S_c, 0, 253);
        if (comp_constant_dev(io[g], o.sCmp, S253[g], CT_SUBORDER_M1_D)) io[g].chk_zero(C_RTX_SIG_S_RANGE, sc.get(SC_ED_ENABLED));
    }
    seg_fix_lock<G>(K, io, o.fseg[0], S253, 0, 246, 0, 0, q);
    seg_fix_lock<G>(K, io, o.fseg[1], S253, 246, 7, 82, 1, r);
#pragma unroll 1
    for (int g = 0; g < G; g++) {
        const EdCtx c = K.with(io[g]);
        const PtA left = baby_add_dev(c, o.fadders0, q[g], r[g]);
        const Scratch sc{a.scratch, a.n_units, io[g].unit};
        sc.set(SC_ED_LEFTX, left.x); sc.set(SC_ED_LEFTY, left.y);
    }
}

// eqCheck: in[0] = mulFix.out (left), in[1] = addRight; both gated by `enabled`
__global__ __launch_bounds__(HZ_BLOCK) void k_eddsa_final(const EddsaArgs a) {
    const uint32_t n = a.ucnt ? a.ucnt : a.n_units;
    const uint32_t li = blockIdx.x * blockDim.x + threadIdx.x;
    if (li >= n) return;
    const uint32_t i = a.u0 + li;
    const UnitIO io{a.base, a.n_units, i, i / a.upi, i % a.upi, a.err};
    const Scratch sc{a.scratch, a.n_units, i};
    const EddsaOff& o = a.ed;
    const Fr one = fr_one(), enabled = sc.get(SC_ED_ENABLED);
    PtA right;
    {
        // mulAny.out = segment 0 + segment 1, the zero-point substitution undone, R8 + h*8A (eddsaposeidon.circom)
        const EdK K = ed_k();
        const EdCtx c = K.with(io);
        PtA s0, s1, R8;
        s0.x = sc.get(SC_ED_S0X); s0.y = sc.get(SC_ED_S0Y); s1.x = sc.get(SC_ED_S1X); s1.y = sc.get(SC_ED_S1Y);
        R8.x = sc.get(SC_ED_R8X); R8.y = sc.get(SC_ED_R8Y);
        const Fr zp = sc.get(SC_ED_ZP);
        const PtA sum = baby_add_dev(c, o.adders0, s0, s1);
        PtA any;
        any.x = fr_mul(sum.x, fr_sub(one, zp));
        any.y = fr_add(sum.y, fr_mul(fr_sub(one, sum.y), zp));
        io.put_m(o.anyOut, any.x); io.put_m(o.anyOut + 1, any.y);
        right = baby_add_dev(c, o.addRight, R8, any);
    }
    Fr d2[2] = {fr_sub(right.x, sc.get(SC_ED_LEFTX)), fr_sub(right.y, sc.get(SC_ED_LEFTY))};
    Fr di[2] = {d2[0], d2[1]};
    inv_pair(di[0], di[1]);
    const Fr ex = is_zero_dev(io, o.eqCheckX, d2[0], di[0]);
    io.chk_zero(C_RTX_SIG_EQX, fr_mul(fr_sub(one, ex), enabled));
    const Fr ey = is_zero_dev(io, o.eqCheckY, d2[1], di[1]);
    io.chk_zero(C_RTX_SIG_EQY, fr_mul(fr_sub(one, ey), enabled));
}

// Signatures per lane: sharing an inversion among G signatures cuts the instruction count (throughput) but leaves 1/G of the
// wavefronts, each G times as long (latency). Few units (a single batch or a handful): the device is far from full and
// latency is what counts; many units per launch: the integer pipe is the limit.
template <int G>
static hipError_t launch_eddsa_split(const EddsaArgs& a0, uint32_t n, hipStream_t s, hipEvent_t front_done, hipEvent_t hash_done) {
    EddsaArgs a = a0;
    a.chain_in_ladder = 1;
    if (a.in_onChain != ~0u && front_done) {   // RollupMain: the point half of the prologue beside the front kernel, the hash half after it
        hipLaunchKernelGGL(k_eddsa_pre_a, dim3((n + HZ_BLOCK - 1) / HZ_BLOCK), dim3(HZ_BLOCK), 0, s, a);
        hipError_t e = hipStreamWaitEvent(s, front_done, 0);
        if (e == hipSuccess && hash_done) e = hipStreamWaitEvent(s, hash_done, 0);   // sigL2Hash (k_main_sighash, on another stream)
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL(k_eddsa_pre_b, dim3((n + HZ_BLOCK - 1) / HZ_BLOCK), dim3(HZ_BLOCK), 0, s, a);
    } else {
        if (front_done) { const hipError_t e = hipStreamWaitEvent(s, front_done, 0); if (e != hipSuccess) return e; }
        if (hash_done) { const hipError_t e = hipStreamWaitEvent(s, hash_done, 0); if (e != hipSuccess) return e; }
        hipLaunchKernelGGL(k_eddsa_pre, dim3((n + HZ_BLOCK - 1) / HZ_BLOCK), dim3(HZ_BLOCK), 0, s, a);
    }
    const uint32_t nl = (n + G - 1) / G;
    hipLaunchKernelGGL(k_eddsa_ladder<G>, dim3((nl + HZ_BLOCK - 1) / HZ_BLOCK, 2), dim3(HZ_BLOCK), 0, s, a);
    return hipGetLastError();
}
template <int G>
static hipError_t launch_eddsa_seg(const EddsaArgs& a, uint32_t n, hipStream_t s) {
    hipLaunchKernelGGL(k_eddsa_pre, dim3((n + HZ_BLOCK - 1) / HZ_BLOCK), dim3(HZ_BLOCK), 0, s, a);
    const uint32_t nl = (n + G - 1) / G;
    constexpr size_t lds = (size_t)ls_slots<G>() * 9 * HZ_BLOCK * sizeof(uint32_t);
    if (lds > 64 * 1024) {   // beyond the default limit of dynamic LDS per workgroup (gfx950 has 160 KB per CU); the attribute is per device
        static bool done[64] = {};
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return hipErrorInvalidDevice;
        if (!done[dev]) {
            const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_eddsa_seg<G>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            if (e != hipSuccess) return e;
            done[dev] = true;
        }
    }
    hipLaunchKernelGGL(k_eddsa_seg<G>, dim3((nl + HZ_BLOCK - 1) / HZ_BLOCK, 2), dim3(HZ_BLOCK), lds, s, a);
    return hipGetLastError();
}
template <int G>
static hipError_t launch_eddsa_fix_g(const EddsaArgs& a, uint32_t n, hipStream_t s) {
    const uint32_t nl = (n + G - 1) / G;
    hipLaunchKernelGGL(k_eddsa_fix<G>, dim3((nl + HZ_BLOCK - 1) / HZ_BLOCK), dim3(HZ_BLOCK), 0, s, a);
    return hipGetLastError();
}
size_t eddsa_side_bytes(uint32_t n) {
    if (n <= HZ_ED_SPLIT_MAX) return (size_t)2 * n * 147 * SD_FIELDS * 9 * sizeof(uint32_t);   // seg_any_proj's numerators
    // the throughput form: the lane state of k_eddsa_fix<HZ_ED_FIX_G> (1.7 KB per lane of eight signatures)
    const size_t lanes = (((size_t)n + HZ_ED_FIX_G - 1) / HZ_ED_FIX_G + 63) & ~(size_t)63;
    return lanes * FixMem<HZ_ED_FIX_G>::SLOTS * 9 * sizeof(uint32_t);
}
hipError_t launch_eddsa(const EddsaArgs& a, hipStream_t s, hipEvent_t front_done, hipEvent_t hash_done) {
    const uint32_t n = a.ucnt ? a.ucnt : a.n_units;
    // a launch the device does not fill is latency bound: the two segments of every signature as independent lanes (148 / 106
    // dependent steps instead of 254), one signature per lane
    if (n <= HZ_ED_SPLIT_MAX) return launch_eddsa_split<1>(a, n, s, front_done, hash_done);
    if (front_done) { const hipError_t e = hipStreamWaitEvent(s, front_done, 0); if (e != hipSuccess) return e; }
    if (hash_done) { const hipError_t e = hipStreamWaitEvent(s, hash_done, 0); if (e != hipSuccess) return e; }
    return launch_eddsa_seg<HZ_ED_G>(a, n, s);
}
hipError_t launch_eddsa_fix(const EddsaArgs& a, hipStream_t s) {
    const uint32_t n = a.ucnt ? a.ucnt : a.n_units;
    if (n <= HZ_ED_SPLIT_MAX) return launch_eddsa_fix_g<1>(a, n, s);   // the same switch as the ladder (16 384 signatures: 12.9 ms per step against 13.3 with eight per lane)
    // Eight signatures per lane: 85 windows x (8 turns + one shared inversion) is still a shorter chain than the variable-base ladder
    // beside it (11.0 ms alone against 17.4), and an eighth of an inversion per window instead of a quarter is 0.2 G fewer
    // wave-instructions per 65 536 signatures (four per lane: 7.5 ms alone, step +0.3..0.6 ms; profiles/r03_eddsa_seg_ab.txt).
    if (!a.side) return hipErrorInvalidValue;   // the lane state lives in the context's side buffer (eddsa_side_bytes)
    return launch_eddsa_fix_g<HZ_ED_FIX_G>(a, n, s);
}
hipError_t launch_eddsa_final(const EddsaArgs& a, hipStream_t s) {
    const uint32_t n = a.ucnt ? a.ucnt : a.n_units;
    hipLaunchKernelGGL(k_eddsa_final, dim3((n + HZ_BLOCK - 1) / HZ_BLOCK), dim3(HZ_BLOCK), 0, s, a);
    return hipGetLastError();
}

// the field self test's square root (hz_fr_ops, HZ_FR_SQRT): circomlib pointbits.circom sqrt() -- the root <= (r-1)/2, 0 for a
// non-residue -- one operand per lane, canonical in and out
__global__ __launch_bounds__(256) void k_fr_sqrt(const uint8_t* __restrict__ a, uint8_t* __restrict__ out, size_t n) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride)
        store_fr(out + i * 32, fr_to_canon(fr_sqrt_circom_dev(fr_from_canon(load_fr(a + i * 32)))));
}
hipError_t launch_fr_sqrt(const void* d_a, void* d_out, size_t n, hipStream_t s) {
    size_t blocks = (n + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(k_fr_sqrt, dim3((unsigned)blocks), dim3(256), 0, s, (const uint8_t*)d_a, (uint8_t*)d_out, n);
    return hipGetLastError();
}

// `component main = AySign2Ax()` (test/lib/utils-bjj.test.js:104-150): one lane per instance
__global__ __launch_bounds__(HZ_BLOCK) void k_ay_sign_2_ax_main(const GadgetArgs a, const EddsaOff o) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= a.N) return;
    const UnitIO io{a.base, a.N, i, i, 0, a.err};
    io.put_u64(0, 1);
    const EdK K = ed_k();
    io.put_m(a.io.out[0], ay_sign_2_ax_dev(K.with(io), io, o, io.in_m(a.io.in[0]), io.in_m(a.io.in[1])));
}
hipError_t launch_ay_sign_2_ax_main(const GadgetArgs& a, const EddsaOff& o, hipStream_t s) {
    hipLaunchKernelGGL(k_ay_sign_2_ax_main, dim3((a.N + HZ_BLOCK - 1) / HZ_BLOCK), dim3(HZ_BLOCK), 0, s, a, o);
    return hipGetLastError();
}

}  // namespace hz
