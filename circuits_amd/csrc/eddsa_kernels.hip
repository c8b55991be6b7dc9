// K5/K6 eddsa: AySign2Ax (reference src/lib/utils-bjj.circom:37-58 + circomlib pointbits.circom
// Bits2Point_Strict) and EdDSAPoseidonVerifier (circomlib eddsaposeidon.circom with babyjub,
// escalarmulany, escalarmulfix, montgomery, compconstant, aliascheck) -- call site
// reference src/rollup-tx.circom:467-482. One transaction per lane.
//
// Every intermediate point of the two scalar multiplications is a witness signal, so the ladder
// is evaluated in the circuit's own affine Montgomery-curve formulas; each division is one field
// inversion. EscalarMulFix's window tables are compile-time constants (bjj_consts.inc).
#include <hip/hip_runtime.h>
#include "gadgets_dev.h"
#include "tx_dev.h"
#include "kernels.h"

namespace hz {

#define HZ_CONST_ARR static __device__ const
#include "gen/bjj_consts.inc"
#undef HZ_CONST_ARR

struct PtA { Fr x, y; };

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

struct EdCtx {
    UnitIO io;
    Fr a, d, A;   // 168700, 168696, 168698 (Montgomery-curve A; B = 1)
    Fr one;
};

// BabyAdd: stores beta,gamma,delta,tau,xout,yout; checks the two division constraints
__device__ __forceinline__ PtA baby_add_dev(const EdCtx& c, BabyAddOff off, const PtA& p, const PtA& q) {
    const Fr beta = fr_mul(p.x, q.y), gamma = fr_mul(p.y, q.x);
    const Fr delta = fr_mul(fr_sub(p.y, fr_mul(c.a, p.x)), fr_add(q.x, q.y));
    const Fr tau = fr_mul(beta, gamma);
    const Fr dt = fr_mul(c.d, tau);
    Fr den[2] = {fr_add(c.one, dt), fr_sub(c.one, dt)};
    Fr inv[2] = {den[0], den[1]};
    batch_inv<2>(inv, 2);
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
__device__ __forceinline__ MDbl mont_dbl_dev(const EdCtx& c, const PtA& p) {
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
struct MAdd { Fr lamda; PtA out; };
__device__ __forceinline__ MAdd mont_add_dev(const EdCtx& c, const PtA& p1, const PtA& p2) {
    MAdd r;
    const Fr num = fr_sub(p2.y, p1.y), den = fr_sub(p2.x, p1.x);
    r.lamda = fr_div(num, den);
    if (fr_is_zero(den)) c.io.chk(C_RTX_SIG_EC, fr_zero(), num);
    r.out.x = fr_sub(fr_sub(fr_sub(fr_sqr(r.lamda), c.A), p1.x), p2.x);
    r.out.y = fr_sub(fr_mul(r.lamda, fr_sub(p1.x, r.out.x)), p1.y);
    return r;
}
__device__ __forceinline__ PtA e2m_dev(const EdCtx& c, const PtA& p) {
    Fr den[2] = {fr_sub(c.one, p.y), p.x};
    Fr inv[2] = {den[0], den[1]};
    batch_inv<2>(inv, 2);
    PtA o;
    o.x = fr_mul(fr_add(c.one, p.y), inv[0]);
    o.y = fr_mul(o.x, inv[1]);
    if (fr_is_zero(den[0])) c.io.chk(C_RTX_SIG_EC, fr_zero(), fr_add(c.one, p.y));
    if (fr_is_zero(den[1])) c.io.chk(C_RTX_SIG_EC, fr_zero(), o.x);
    return o;
}
__device__ __forceinline__ PtA m2e_dev(const EdCtx& c, const PtA& p) {
    Fr den[2] = {p.y, fr_add(p.x, c.one)};
    Fr inv[2] = {den[0], den[1]};
    batch_inv<2>(inv, 2);
    PtA o;
    o.x = fr_mul(p.x, inv[0]);
    o.y = fr_mul(fr_sub(p.x, c.one), inv[1]);
    if (fr_is_zero(den[0])) c.io.chk(C_RTX_SIG_EC, fr_zero(), p.x);
    if (fr_is_zero(den[1])) c.io.chk(C_RTX_SIG_EC, fr_zero(), fr_sub(p.x, c.one));
    return o;
}

// SegmentMulAny(n): bits e[e0 .. e0+n) of the canonical integer `e`.
// Step i is doubler_i (D_{i+1} = 2 D_i) followed by adder_i (D_{i+1} + acc_i). adder_i and
// doubler_{i+1} both depend only on D_{i+1} and acc_i, so their two divisions share ONE field
// inversion (Montgomery's trick): the ladder costs one inversion per scalar bit instead of two.
struct SegAnyRes { PtA out, dbl; };
__device__ SegAnyRes seg_any_dev(const EdCtx& c, const SegAnyOff& o, const Fc& e, int e0, int n, const PtA& p) {
    const PtA m = e2m_dev(c, p);
    c.io.put_m(o.e2m, m.x); c.io.put_m(o.e2m + 1, m.y);
    const int steps = n - 1;
    // doubler_0 alone
    MDbl d = mont_dbl_dev(c, m);
    PtA addIn = m;
#pragma unroll 1
    for (int i = 0; i < steps; i++) {
        const uint32_t b = o.bits + BIT_N * i;
        // adder_i: in1 = d.out, in2 = addIn ; doubler_{i+1}: in = d.out
        const Fr a_num = fr_sub(addIn.y, d.out.y), a_den = fr_sub(addIn.x, d.out.x);
        const bool more = i + 1 < steps;
        Fr nx1_2 = fr_zero(), d_num = fr_zero(), d_den = fr_zero();
        if (more) {
            nx1_2 = fr_sqr(d.out.x);
            d_num = fr_add(fr_add(fr_add(fr_dbl(nx1_2), nx1_2), fr_mul(fr_dbl(c.A), d.out.x)), c.one);
            d_den = fr_dbl(d.out.y);
        }
        Fr den[2] = {a_den, d_den};
        Fr inv[2] = {a_den, d_den};
        batch_inv<2>(inv, more ? 2 : 1);
        MAdd a;
        a.lamda = fr_mul(a_num, inv[0]);
        if (fr_is_zero(den[0])) c.io.chk(C_RTX_SIG_EC, fr_zero(), a_num);
        a.out.x = fr_sub(fr_sub(fr_sub(fr_sqr(a.lamda), c.A), d.out.x), addIn.x);
        a.out.y = fr_sub(fr_mul(a.lamda, fr_sub(d.out.x, a.out.x)), d.out.y);
        const uint32_t sel = c_bit(e, e0 + i + 1);
        const PtA so = sel ? a.out : addIn;
        c.io.put_m(b + BIT_DBL_X1_2, d.x1_2); c.io.put_m(b + BIT_DBL_LAMDA, d.lamda); c.io.put_m(b + BIT_DBL_OUT0, d.out.x); c.io.put_m(b + BIT_DBL_OUT1, d.out.y);
        c.io.put_m(b + BIT_ADD_LAMDA, a.lamda); c.io.put_m(b + BIT_ADD_OUT0, a.out.x); c.io.put_m(b + BIT_ADD_OUT1, a.out.y);
        c.io.put_m(b + BIT_SEL_OUT0, so.x); c.io.put_m(b + BIT_SEL_OUT1, so.y);
        addIn = so;
        if (more) {
            MDbl nd;
            nd.x1_2 = nx1_2;
            nd.lamda = fr_mul(d_num, inv[1]);
            if (fr_is_zero(den[1])) c.io.chk(C_RTX_SIG_EC, fr_zero(), d_num);
            nd.out.x = fr_sub(fr_sub(fr_sqr(nd.lamda), c.A), fr_dbl(d.out.x));
            nd.out.y = fr_sub(fr_mul(nd.lamda, fr_sub(d.out.x, nd.out.x)), d.out.y);
            d = nd;
        }
    }
    SegAnyRes r;
    r.dbl = d.out;
    const PtA me = m2e_dev(c, addIn);
    c.io.put_m(o.m2e, me.x); c.io.put_m(o.m2e + 1, me.y);
    PtA negp;
    negp.x = fr_neg(p.x);
    negp.y = p.y;
    const PtA ea = baby_add_dev(c, o.eadder, me, negp);
    r.out = c_bit(e, e0) ? me : ea;
    c.io.put_m(o.lastSel, r.out.x); c.io.put_m(o.lastSel + 1, r.out.y);
    return r;
}

// SegmentMulFix on the constant base: window tables from HZ_BJJ_FIX_WIN
__device__ PtA seg_fix_dev(const EdCtx& c, const SegFixOff& o, const Fc& e, int e0, int nbits, int win0, int seg) {
    PtA acc;
    acc.x = ld_const(HZ_BJJ_FIX_DBLLAST[2 * seg]);
    acc.y = ld_const(HZ_BJJ_FIX_DBLLAST[2 * seg + 1]);
#pragma unroll 1
    for (int i = 0; i < o.nwin; i++) {
        uint32_t k = 0, b0 = 0, b1 = 0;
        for (int j = 0; j < 3; j++) {
            const uint32_t bit = (3 * i + j < nbits) ? c_bit(e, e0 + 3 * i + j) : 0u;
            k |= bit << j;
            if (j == 0) b0 = bit;
            if (j == 1) b1 = bit;
        }
        PtA mo;
        mo.x = ld_const(HZ_BJJ_FIX_WIN[((win0 + i) * 8 + k) * 2]);
        mo.y = ld_const(HZ_BJJ_FIX_WIN[((win0 + i) * 8 + k) * 2 + 1]);
        const uint32_t wb = o.windows + WIN_N * i;
        c.io.put_bit(wb + WIN_S10, b1 & b0);
        c.io.put_m(wb + WIN_MUX0, mo.x); c.io.put_m(wb + WIN_MUX1, mo.y);
        const MAdd a = mont_add_dev(c, acc, mo);
        c.io.put_m(wb + WIN_ADD_LAMDA, a.lamda); c.io.put_m(wb + WIN_ADD_OUT0, a.out.x); c.io.put_m(wb + WIN_ADD_OUT1, a.out.y);
        acc = a.out;
    }
    const PtA me = m2e_dev(c, acc);
    c.io.put_m(o.m2e, me.x); c.io.put_m(o.m2e + 1, me.y);
    PtA cn;
    cn.x = ld_const(HZ_BJJ_FIX_CNEG[2 * seg]);
    cn.y = ld_const(HZ_BJJ_FIX_CNEG[2 * seg + 1]);
    return baby_add_dev(c, o.cAdd, me, cn);
}

__global__ __launch_bounds__(HZ_BLOCK) void k_eddsa(const EddsaArgs a) {
    extern __shared__ __attribute__((aligned(16))) uint32_t lds_raw[];
    Fr* C6 = reinterpret_cast<Fr*>(lds_raw);
    Fr* M6 = C6 + poseidon_nconst<6>();
    stage_poseidon_consts<6>(C6);
    __syncthreads();
    const uint32_t li = blockIdx.x * blockDim.x + threadIdx.x;
    if (li >= (a.ucnt ? a.ucnt : a.n_units)) return;
    const uint32_t i = a.u0 + li;
    EdCtx c;
    c.io = UnitIO{a.base, a.n_units, i, i / a.upi, i % a.upi, a.err};
    c.one = fr_one();
    c.a = fr_from_u64(168700); c.d = fr_from_u64(168696); c.A = fr_from_u64(168698);
    const UnitIO& io = c.io;
    const Scratch sc{a.scratch, a.n_units, i};
    const EddsaOff& o = a.ed;
    const Fr enabled = sc.get(SC_ED_ENABLED), signSig = sc.get(SC_ED_SIGN), aySig = sc.get(SC_ED_AYSIG), Ay = sc.get(SC_ED_AY);
    const Fr S = sc.get(SC_ED_S), R8x = sc.get(SC_ED_R8X), R8y = sc.get(SC_ED_R8Y), M = sc.get(SC_SIGL2HASH);
    // ---- AySign2Ax
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
    {
        const uint32_t sg = comp_constant_dev(io, o.ax_signCalc, x_c, CT_HALF_D);
        io.chk(C_RTX_AX_SIGN, fr_from_bit(sg), signSig);
    }
    // ---- EdDSAPoseidonVerifier
    const Fc S_c = fr_to_canon(S);
    num2bits_dev(io, o.snum2bits, S_c, 253, C_RTX_SIG_N2B_S);
    const Fc S253 = c_extract(S_c, 0, 253);
    {
        const uint32_t gt = comp_constant_dev(io, o.sCmp, S253, CT_SUBORDER_M1_D);
        if (gt) io.chk_zero(C_RTX_SIG_S_RANGE, enabled);
    }
    Fr hin[5] = {R8x, R8y, x, Ay, M};
    WitSboxSink s6 = io.sbox_sink(o.hash);
    const Fr h = poseidon_hash<6>(hin, C6, M6, s6);
    const Fc h_c = fr_to_canon(h);
    num2bits_strict_dev(io, o.h2bits, h_c, C_RTX_SIG_H_ALIAS);
    PtA A;
    A.x = x; A.y = Ay;
    const PtA d1 = baby_add_dev(c, o.dbl1, A, A);
    const PtA d2 = baby_add_dev(c, o.dbl2, d1, d1);
    const PtA d3 = baby_add_dev(c, o.dbl3, d2, d2);
    {
        // isZero.in <== dbl3.x (the input x of the third doubling) ; zeropoint.in <== dbl3.xout
        Fr z[2] = {d2.x, d3.x};
        Fr zi[2] = {z[0], z[1]};
        batch_inv<2>(zi, 2);
        const Fr az = is_zero_dev(io, o.isZero, z[0], zi[0]);
        io.chk_zero(C_RTX_SIG_A_NONZERO, fr_mul(az, enabled));
        const Fr zp = is_zero_dev(io, o.zeropoint, z[1], zi[1]);
        const bool zpb = fr_is_zero(z[1]);
        PtA p0;
        p0.x = zpb ? ld_const(HZ_BJJ_BASE8_X) : d3.x;
        p0.y = zpb ? ld_const(HZ_BJJ_BASE8_Y) : d3.y;
        io.put_m(o.seg0p, p0.x); io.put_m(o.seg0p + 1, p0.y);
        const SegAnyRes s0 = seg_any_dev(c, o.seg[0], h_c, 0, 148, p0);
        const MDbl dd = mont_dbl_dev(c, s0.dbl);
        io.put_m(o.dblr, dd.x1_2); io.put_m(o.dblr + 1, dd.lamda); io.put_m(o.dblr + 2, dd.out.x); io.put_m(o.dblr + 3, dd.out.y);
        const PtA p1 = m2e_dev(c, dd.out);
        io.put_m(o.m2e0, p1.x); io.put_m(o.m2e0 + 1, p1.y);
        const SegAnyRes s1 = seg_any_dev(c, o.seg[1], h_c, 148, 106, p1);
        const PtA sum = baby_add_dev(c, o.adders0, s0.out, s1.out);
        PtA any;
        any.x = fr_mul(sum.x, fr_sub(c.one, zp));
        any.y = fr_add(sum.y, fr_mul(fr_sub(c.one, sum.y), zp));
        io.put_m(o.anyOut, any.x); io.put_m(o.anyOut + 1, any.y);
        PtA R8;
        R8.x = R8x; R8.y = R8y;
        const PtA right = baby_add_dev(c, o.addRight, R8, any);
        const PtA f0 = seg_fix_dev(c, o.fseg[0], S253, 0, 246, 0, 0);
        const PtA f1 = seg_fix_dev(c, o.fseg[1], S253, 246, 7, 82, 1);
        const PtA left = baby_add_dev(c, o.fadders0, f0, f1);
        Fr q[2] = {fr_sub(right.x, left.x), fr_sub(right.y, left.y)};   // eqCheck: in[0] = mulFix.out, in[1] = addRight
        Fr qi[2] = {q[0], q[1]};
        batch_inv<2>(qi, 2);
        const Fr ex = is_zero_dev(io, o.eqCheckX, q[0], qi[0]);
        io.chk_zero(C_RTX_SIG_EQX, fr_mul(fr_sub(c.one, ex), enabled));
        const Fr ey = is_zero_dev(io, o.eqCheckY, q[1], qi[1]);
        io.chk_zero(C_RTX_SIG_EQY, fr_mul(fr_sub(c.one, ey), enabled));
    }
}

hipError_t launch_eddsa(const EddsaArgs& a, hipStream_t s) {
    hipLaunchKernelGGL(k_eddsa, dim3(((a.ucnt ? a.ucnt : a.n_units) + HZ_BLOCK - 1) / HZ_BLOCK), dim3(HZ_BLOCK), (size_t)poseidon_const_frs<6>() * sizeof(Fr), s, a);
    return hipGetLastError();
}

}  // namespace hz
