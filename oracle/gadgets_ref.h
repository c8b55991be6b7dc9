// TEST INFRASTRUCTURE -- CPU oracle. Restatement of the circomlib 0.5.2 gadgets the Hermez circuits
// include (`include "../node_modules/circomlib/circuits/..."`, reference src/rollup-tx.circom:1-4,
// src/decode-tx.circom:1-3, src/withdraw.circom:1-4 ...). circomlib is pinned in the reference's
// package-lock.json:861-862 but absent from /root/reference: each gadget follows the published
// template (SURVEY Appendix A is the working spec) and names the template it restates.
// Parity status: Poseidon and EdDSA-Poseidon pinned on upstream circomlib known answers
// (tests/golden/poseidon_kat.json, eddsa_poseidon_kat.json); SMT pinned on self-checks (independent tree
// rebuild, membership proofs) -- "parity unpinned" against the reference repository itself, whose tests hold
// no literal values for hashes, roots or signatures.
#pragma once
#include <functional>
#include <string>
#include <vector>
#include "../include/hz_layout.h"
#include "fr64.h"
#include <map>
#include "poseidon_ref.h"

namespace orc {

// ---- witness writer + first-failure record ---------------------------------------------------
struct FailRec {
    bool failed = false;
    uint64_t key = ~0ull;
    int inst = 0, unit = 0, cid = 0;
    F lhs, rhs;
    void take(uint64_t k, int i, int u, int c, const F& l, const F& r) {
        if (k >= key) return;   // lowest key; among equal keys the first evaluated
        failed = true; key = k; inst = i; unit = u; cid = c; lhs = l; rhs = r;
    }
};
// the first failure of the whole run and of every instance (orc_failure_of: what the product's hz_witness_failures reports)
struct Fail : FailRec {
    std::vector<FailRec> per_inst;
};

struct W {
    const hzl::Layout* lo;
    std::vector<F>* vals;   // physical buffer
    std::vector<uint8_t>* written;
    Fail* fail;
    int sec = 0;
    uint32_t unit = 0;
    uint32_t inst = 0;      // instance (for the failure key)
    int err_unit = 0;       // unit reported in failures
    void set(uint32_t sig, const F& v) const {
        const uint64_t p = lo->phys(sec, sig, unit);
        (*vals)[p] = v;
        if (written) (*written)[p] = 1;
    }
    F get(uint32_t sig) const { return (*vals)[lo->phys(sec, sig, unit)]; }
    F get_unit(uint32_t sig, uint32_t u) const { return (*vals)[lo->phys(sec, sig, u)]; }
    // `lhs === rhs`
    void chk(int cid, const F& lhs, const F& rhs) const {
        if (lhs == rhs) return;
        const uint64_t key = ((uint64_t)inst << 40) | ((uint64_t)(uint32_t)err_unit << 16) | (uint32_t)cid;
        fail->take(key, (int)inst, err_unit, cid, lhs, rhs);
        if (inst < fail->per_inst.size()) fail->per_inst[inst].take(key, (int)inst, err_unit, cid, lhs, rhs);
    }
};

// ---- bitify.circom ---------------------------------------------------------------------------
// Num2Bits(n): out[i] <-- (in >> i) & 1; booleanity; sum(out[i]*2^i) === in
inline std::vector<int> num2bits(const W& w, uint32_t off, const F& in, int n, int cid) {
    std::vector<int> b(n);
    F lc(0);
    for (int i = 0; i < n; i++) {
        b[i] = in.bit(i);
        w.set(off + i, F(b[i]));
        if (b[i]) lc += pow2(i);
    }
    w.chk(cid, lc, in);
    return b;
}
inline F bits2num(const std::vector<int>& b, int from, int n) {
    F r(0);
    for (int i = 0; i < n; i++)
        if (b[from + i]) r += pow2(i);
    return r;
}

// ---- comparators.circom ----------------------------------------------------------------------
// IsZero: inv <-- in!=0 ? 1/in : 0; out <== -in*inv + 1; in*out === 0
inline F is_zero(const W& w, hzl::IsZOff off, const F& in) {
    F inv = in.inv();
    F out = F(1) - in * inv;
    w.set(off, inv);
    w.set(off + 1, out);
    return out;
}
// IsEqual: IsZero(in[1] - in[0])
inline F is_equal(const W& w, hzl::IsZOff off, const F& in0, const F& in1) { return is_zero(w, off, in1 - in0); }
// ForceEqualIfEnabled: isz.in = in[1]-in[0]; (1 - isz.out)*enabled === 0
inline void force_equal_if_enabled(const W& w, hzl::IsZOff off, const F& enabled, const F& in0, const F& in1, int cid) {
    F out = is_zero(w, off, in1 - in0);
    w.chk(cid, (F(1) - out) * enabled, F(0));
}

// ---- mux1.circom: out = (c1 - c0)*s + c0 -----------------------------------------------------
inline F mux1(const F& c0, const F& c1, const F& s) { return (c1 - c0) * s + c0; }

// ---- compconstant.circom: CompConstant(ct) over 254 input bits ---------------------------------
// returns out (1 if in > ct)
inline int comp_constant(const W& w, const hzl::CompConstOff& off, const std::vector<int>& in, const uint64_t* ct /*256-bit LE*/) {
    // b = 2^128 - 1, a = 1, e = 1
    F b = pow2(128) - F(1), a(1), e(1), sum(0);
    for (int i = 0; i < 127; i++) {
        const int clsb = (int)((ct[(2 * i) >> 6] >> ((2 * i) & 63)) & 1);
        const int cmsb = (int)((ct[(2 * i + 1) >> 6] >> ((2 * i + 1) & 63)) & 1);
        const F slsb(in[2 * i]), smsb(in[2 * i + 1]);
        F part;
        if (cmsb == 0 && clsb == 0) part = -(b * smsb * slsb) + b * smsb + b * slsb;
        else if (cmsb == 0 && clsb == 1) part = a * smsb * slsb - a * slsb + b * smsb - a * smsb + a;
        else if (cmsb == 1 && clsb == 0) part = b * smsb * slsb - a * smsb + a;
        else part = -(a * smsb * slsb) + a;
        w.set(off.parts + i, part);
        sum += part;
        b = b - e;
        a = a + e;
        e = e + e;
    }
    // sout <== sum; Num2Bits(135); out = bit 127
    int out = 0;
    for (int i = 0; i < 135; i++) {
        const int bit = sum.bit(i);
        w.set(off.bits + i, F(bit));
        if (i == 127) out = bit;
    }
    return out;
}
extern const uint64_t CT_MINUS1[4];      // r - 1   (AliasCheck)
extern const uint64_t CT_HALF[4];        // (r-1)/2 (sign of x, pointbits.circom / sign.circom)
extern const uint64_t CT_SUBORDER_M1[4]; // subgroup order - 1 (eddsaposeidon.circom)

// Num2Bits_strict = Num2Bits(254) + AliasCheck (CompConstant(-1).out === 0)
inline std::vector<int> num2bits_strict(const W& w, const hzl::N2BStrictOff& off, const F& in, int cid_n2b, int cid_alias) {
    std::vector<int> b = num2bits(w, off.bits, in, 254, cid_n2b);
    const int o = comp_constant(w, off.cc, b, CT_MINUS1);
    w.chk(cid_alias, F(o), F(0));
    return b;
}

// ---- babyjub.circom / montgomery.circom ------------------------------------------------------
struct Pt { F x, y; };
extern const F& BJ_A();  // 168700
extern const F& BJ_D();  // 168696
// BabyAdd: beta, gamma, delta, tau, xout, yout
inline Pt baby_add(const W& w, hzl::BabyAddOff off, const Pt& p1, const Pt& p2, int cid_ec) {
    const F a = BJ_A(), d = BJ_D();
    F beta = p1.x * p2.y, gamma = p1.y * p2.x;
    F delta = (-(a * p1.x) + p1.y) * (p2.x + p2.y);
    F tau = beta * gamma;
    F xout = (beta + gamma) / (F(1) + d * tau);
    F yout = (delta + a * beta - gamma) / (F(1) - d * tau);
    w.set(off + hzl::BA_BETA, beta); w.set(off + hzl::BA_GAMMA, gamma); w.set(off + hzl::BA_DELTA, delta);
    w.set(off + hzl::BA_TAU, tau); w.set(off + hzl::BA_XOUT, xout); w.set(off + hzl::BA_YOUT, yout);
    w.chk(cid_ec, (F(1) + d * tau) * xout, beta + gamma);
    w.chk(cid_ec, (F(1) - d * tau) * yout, delta + a * beta - gamma);
    return Pt{xout, yout};
}
// MontgomeryAdd / MontgomeryDouble with A = 168698, B = 1
struct MAddOut { F lamda; Pt out; };
inline MAddOut mont_add(const W& w, const Pt& p1, const Pt& p2, int cid_ec) {
    const F A(168698);
    MAddOut r;
    r.lamda = (p2.y - p1.y) / (p2.x - p1.x);
    w.chk(cid_ec, r.lamda * (p2.x - p1.x), p2.y - p1.y);
    r.out.x = r.lamda * r.lamda - A - p1.x - p2.x;
    r.out.y = r.lamda * (p1.x - r.out.x) - p1.y;
    return r;
}
struct MDblOut { F x1_2, lamda; Pt out; };
inline MDblOut mont_dbl(const W& w, const Pt& p, int cid_ec) {
    const F A(168698);
    MDblOut r;
    r.x1_2 = p.x * p.x;
    const F num = F(3) * r.x1_2 + F(2) * A * p.x + F(1), den = F(2) * p.y;
    r.lamda = num / den;
    w.chk(cid_ec, r.lamda * den, num);
    r.out.x = r.lamda * r.lamda - A - F(2) * p.x;
    r.out.y = r.lamda * (p.x - r.out.x) - p.y;
    return r;
}
inline Pt edwards2montgomery(const W& w, const Pt& p, int cid_ec) {
    Pt o;
    o.x = (F(1) + p.y) / (F(1) - p.y);
    o.y = o.x / p.x;
    w.chk(cid_ec, o.x * (F(1) - p.y), F(1) + p.y);
    w.chk(cid_ec, o.y * p.x, o.x);
    return o;
}
inline Pt montgomery2edwards(const W& w, const Pt& p, int cid_ec) {
    Pt o;
    o.x = p.x / p.y;
    o.y = (p.x - F(1)) / (p.x + F(1));
    w.chk(cid_ec, o.x * p.y, p.x);
    w.chk(cid_ec, o.y * (p.x + F(1)), p.x - F(1));
    return o;
}

// pointbits.circom sqrt(): Tonelli-Shanks, returns 0 when no root exists, root <= (r-1)/2
F fr_sqrt_circom(const F& n);

// ---- plain (signal-free) helpers used by the batch-independent self checks -------------------
Pt bj_add_plain(const Pt& p, const Pt& q);
Pt bj_mul_plain(const Pt& p, const uint64_t* k256);
extern const Pt& BJ_BASE8();
extern const uint64_t BJ_SUBORDER[4];

// ---- smt ---------------------------------------------------------------------------------------
F smt_hash1(const W& w, hzl::PoseidonOff off, const F& key, const F& value);   // Poseidon(3)(key,value,1)
F smt_hash2(const W& w, hzl::PoseidonOff off, const F& l, const F& r);         // Poseidon(2)(l,r)
struct SmtCids { int n2b_old, alias_old, n2b_new, alias_new, levins, sm_final, oldroot, keys; };
// SMTProcessor(n): returns newRoot
F smt_processor(const W& w, const hzl::SmtProcOff& o, int n, const F& oldRoot, const F* siblings, const F& oldKey, const F& oldValue,
                const F& isOld0, const F& newKey, const F& newValue, const F& fnc0, const F& fnc1, const SmtCids& c);

// SMTVerifier(n) (circomlib smt/smtverifier.circom + smtverifierlevel, smtverifiersm), withdraw_ref.cpp
struct SmtVerCids { int n2b_old, alias_old, n2b_new, alias_new, levins, sm_final, keys, root; };
void smt_verifier(const W& w, const hzl::SmtVerOff& o, int n, const F& enabled, const F& root, const F* siblings, const F& oldKey,
                  const F& oldValue, const F& isOld0, const F& key, const F& value, const F& fnc, const SmtVerCids& c);

// writes the S-box signals of a Poseidon and returns the digest
extern thread_local std::map<uint64_t, std::vector<F>>* g_poseidon_log;
F poseidon_w(const W& w, hzl::PoseidonOff off, const F* in, int n_in);

}  // namespace orc
