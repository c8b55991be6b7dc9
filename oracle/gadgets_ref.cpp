// TEST INFRASTRUCTURE -- CPU oracle (see gadgets_ref.h).
#include "gadgets_ref.h"
#include <map>

namespace orc {

// r - 1
const uint64_t CT_MINUS1[4] = {0x43e1f593f0000000ull, 0x2833e84879b97091ull, 0xb85045b68181585dull, 0x30644e72e131a029ull};
// (r - 1) / 2 = 10944121435919637611123202872628637544274182200208017171849102093287904247808
const uint64_t CT_HALF[4] = {0xa1f0fac9f8000000ull, 0x9419f4243cdcb848ull, 0xdc2822db40c0ac2eull, 0x183227397098d014ull};
// subgroup order 2736030358979909402780800718157159386076813972158567259200215660948447373041
const uint64_t BJ_SUBORDER[4] = {0x677297dc392126f1ull, 0xab3eedb83920ee0aull, 0x370a08b6d0302b0bull, 0x060c89ce5c263405ull};
const uint64_t CT_SUBORDER_M1[4] = {0x677297dc392126f0ull, 0xab3eedb83920ee0aull, 0x370a08b6d0302b0bull, 0x060c89ce5c263405ull};

const F& BJ_A() { static const F a(168700); return a; }
const F& BJ_D() { static const F d(168696); return d; }

static F from_dec(const char* s) {
    F r(0), ten(10);
    for (; *s; s++) r = r * ten + F((int)(*s - '0'));
    return r;
}
const Pt& BJ_BASE8() {
    static const Pt b{from_dec("5299619240641551281634865583518297030282874472190772894086521144482721001553"),
                      from_dec("16950150798460657717958625567821834550301663161624707787222815936182638968203")};
    return b;
}

Pt bj_add_plain(const Pt& p, const Pt& q) {
    const F a = BJ_A(), d = BJ_D();
    F x1y2 = p.x * q.y, y1x2 = p.y * q.x, t = d * x1y2 * y1x2;
    Pt r;
    r.x = (x1y2 + y1x2) / (F(1) + t);
    r.y = (p.y * q.y - a * p.x * q.x) / (F(1) - t);
    return r;
}
Pt bj_mul_plain(const Pt& p, const uint64_t* k) {
    Pt acc{F(0), F(1)};
    for (int i = 255; i >= 0; i--) {
        acc = bj_add_plain(acc, acc);
        if ((k[i >> 6] >> (i & 63)) & 1) acc = bj_add_plain(acc, p);
    }
    return acc;
}

// pointbits.circom `function sqrt(n)`: Tonelli-Shanks with s = 28; result normalised to the root
// that is "non-negative" in circom's signed view, i.e. <= (r-1)/2; 0 if n is a non-residue.
F fr_sqrt_circom(const F& n) {
    if (n.is_zero()) return F(0);
    // (r-1)/2
    if (n.pow(CT_HALF) != F(1)) return F(0);
    // r - 1 = 2^28 * q
    static const uint64_t Q[4] = {0x9b9709143e1f593full, 0x181585d2833e8487ull, 0x131a029b85045b68ull, 0x000000030644e72eull};
    static const uint64_t Q1H[4] = {0xcdcb848a1f0faca0ull, 0x0c0ac2e9419f4243ull, 0x098d014dc2822db4ull, 0x0000000183227397ull};  // (q+1)/2
    static F c_root;
    static bool init = false;
    if (!init) {
        // any quadratic non-residue g gives a primitive 2^28-th root g^q
        for (int g = 2;; g++) {
            F gg(g);
            if (gg.pow(CT_HALF) != F(1)) { c_root = gg.pow(Q); break; }
        }
        init = true;
    }
    int m = 28;
    F c = c_root, t = n.pow(Q), r = n.pow(Q1H);
    while (!r.is_zero() && t != F(1)) {
        F sq = t * t;
        int i = 1;
        while (sq != F(1)) { i++; sq = sq * sq; }
        F b = c;
        for (int j = 0; j < m - i - 1; j++) b = b * b;
        m = i;
        c = b * b;
        t = t * c;
        r = r * b;
    }
    // if (r < 0) r = -r   (signed view: r > (p-1)/2)
    uint64_t rc[4];
    r.to_canon(rc);
    bool gt = false;
    for (int i = 3; i >= 0; i--) {
        if (rc[i] > CT_HALF[i]) { gt = true; break; }
        if (rc[i] < CT_HALF[i]) break;
    }
    return gt ? -r : r;
}

// test aid (orc_poseidon_inputs): the inputs every Poseidon component was evaluated on, by the physical index of its first stored signal
thread_local std::map<uint64_t, std::vector<F>>* g_poseidon_log = nullptr;
F poseidon_w(const W& w, hzl::PoseidonOff off, const F* in, int n_in) {
    std::vector<F> sb;
    F h = poseidon(in, n_in, &sb);
    if (g_poseidon_log) (*g_poseidon_log)[w.lo->phys(w.sec, off, w.unit)] = std::vector<F>(in, in + n_in);
    for (size_t k = 0; k < sb.size(); k++) w.set(off + (uint32_t)k, sb[k]);
    return h;
}
F smt_hash1(const W& w, hzl::PoseidonOff off, const F& key, const F& value) {
    F in[3] = {key, value, F(1)};
    return poseidon_w(w, off, in, 3);
}
F smt_hash2(const W& w, hzl::PoseidonOff off, const F& l, const F& r) {
    F in[2] = {l, r};
    return poseidon_w(w, off, in, 2);
}

// circomlib smt/smtprocessor.circom (+ smtlevins, smtprocessorsm, smtprocessorlevel, switcher)
F smt_processor(const W& w, const hzl::SmtProcOff& o, int n, const F& oldRoot, const F* siblings, const F& oldKey, const F& oldValue,
                const F& isOld0, const F& newKey, const F& newValue, const F& fnc0, const F& fnc1, const SmtCids& c) {
    using namespace hzl;
    if (o.fnc != ~0u) { w.set(o.fnc, fnc0); w.set(o.fnc + 1, fnc1); }
    const F enabled = fnc0 + fnc1 - fnc0 * fnc1;
    w.set(o.enabled, enabled);
    const F h1old = smt_hash1(w, o.hash1Old, oldKey, oldValue);
    const F h1new = smt_hash1(w, o.hash1New, newKey, newValue);
    const std::vector<int> bOld = num2bits_strict(w, o.n2bOld, oldKey, c.n2b_old, c.alias_old);
    const std::vector<int> bNew = num2bits_strict(w, o.n2bNew, newKey, c.n2b_new, c.alias_new);
    // SMTLevIns
    std::vector<F> isz(n), levIns(n), done(n);
    for (int i = 0; i < n; i++) isz[i] = is_zero(w, o.isz + 2 * i, siblings[i]);
    w.chk(c.levins, (isz[n - 1] - F(1)) * enabled, F(0));
    levIns[n - 1] = F(1) - isz[n - 2];
    done[n - 2] = levIns[n - 1];
    for (int i = n - 2; i > 0; i--) {
        levIns[i] = (F(1) - done[i]) * (F(1) - isz[i - 1]);
        w.set(o.levIns + (i - 1), levIns[i]);
        done[i - 1] = levIns[i] + done[i];
    }
    levIns[0] = F(1) - done[0];
    // xors + state machine
    std::vector<F> st_top(n), st_old0(n), st_bot(n), st_new1(n), st_na(n), st_upd(n);
    F p_top = enabled, p_old0(0), p_bot(0), p_new1(0), p_na = F(1) - enabled, p_upd(0);
    for (int i = 0; i < n; i++) {
        const F a(bOld[i]), b(bNew[i]);
        const F x = a + b - F(2) * a * b;  // XOR
        w.set(o.xors + i, x);
        const F aux1 = p_top * levIns[i];
        const F aux2 = aux1 * fnc0;
        st_top[i] = p_top - aux1;
        st_old0[i] = aux2 * isOld0;
        st_new1[i] = (aux2 - st_old0[i] + p_bot) * x;
        st_bot[i] = (F(1) - x) * (aux2 - st_old0[i] + p_bot);
        st_upd[i] = aux1 - aux2;
        st_na[i] = p_new1 + p_old0 + p_na + p_upd;
        w.set(o.sm + SM_N * i + SM_AUX1, aux1); w.set(o.sm + SM_N * i + SM_AUX2, aux2);
        w.set(o.sm + SM_N * i + SM_OLD0, st_old0[i]); w.set(o.sm + SM_N * i + SM_NEW1, st_new1[i]);
        w.set(o.sm + SM_N * i + SM_BOT, st_bot[i]);
        p_top = st_top[i]; p_old0 = st_old0[i]; p_bot = st_bot[i]; p_new1 = st_new1[i]; p_na = st_na[i]; p_upd = st_upd[i];
    }
    w.chk(c.sm_final, st_na[n - 1] + st_new1[n - 1] + st_old0[n - 1] + st_upd[n - 1], F(1));
    // levels, bottom-up
    F oldChild(0), newChild(0);
    for (int i = n - 1; i >= 0; i--) {
        const uint32_t lv = o.levels + LV_SIZE * i;
        const F sel(bNew[i]);
        // old side
        const F oaux = (siblings[i] - oldChild) * sel;  // Switcher: aux = (R-L)*sel
        const F oL = oaux + oldChild, oR = -oaux + siblings[i];
        w.set(lv + LV_OLDSW_AUX, oaux);
        const F oh = smt_hash2(w, lv + LV_OLDHASH, oL, oR);
        const F aux0 = h1old * (st_bot[i] + st_new1[i] + st_upd[i]);
        const F oRoot = aux0 + oh * st_top[i];
        w.set(lv + LV_AUX0, aux0); w.set(lv + LV_OLDROOT, oRoot);
        // new side
        const F aux1 = newChild * (st_top[i] + st_bot[i]);
        const F nswL = aux1 + h1new * st_new1[i];
        const F aux2 = siblings[i] * st_top[i];
        const F nswR = aux2 + h1old * st_new1[i];
        const F naux = (nswR - nswL) * sel;
        const F nL = naux + nswL, nR = -naux + nswR;
        const F nh = smt_hash2(w, lv + LV_NEWHASH, nL, nR);
        const F aux3 = nh * (st_top[i] + st_bot[i] + st_new1[i]);
        const F nRoot = aux3 + h1new * (st_old0[i] + st_upd[i]);
        w.set(lv + LV_NEWSW_AUX, naux); w.set(lv + LV_AUX1, aux1); w.set(lv + LV_AUX2, aux2); w.set(lv + LV_AUX3, aux3);
        w.set(lv + LV_NEWSW_L, nswL); w.set(lv + LV_NEWSW_R, nswR); w.set(lv + LV_NEWROOT, nRoot);
        oldChild = oRoot;
        newChild = nRoot;
    }
    // top
    const F topSel = fnc0 * fnc1;
    const F topAux = (newChild - oldChild) * topSel;
    const F outL = topAux + oldChild, outR = -topAux + newChild;
    w.set(o.topSel, topSel); w.set(o.topAux, topAux);
    force_equal_if_enabled(w, o.checkOld, enabled, oldRoot, outL, c.oldroot);
    const F newRoot = enabled * (outR - oldRoot) + oldRoot;
    w.set(o.newRoot, newRoot);
    const F keq = is_equal(w, o.keyEq, oldKey, newKey);
    // keysOk = MultiAND(3)(1-fnc0, fnc1, 1-keq): ands[1] = AND(in1,in2); and2 = AND(in0, ands[1])
    const F and1 = fnc1 * (F(1) - keq);
    const F and2 = (F(1) - fnc0) * and1;
    w.set(o.and1, and1); w.set(o.and2, and2);
    w.chk(c.keys, and2, F(0));
    return newRoot;
}

}  // namespace orc

#include "gen/fee_table_canon.inc"
namespace orc {
const uint64_t* fee_table() { return ORC_FEE_TABLE; }
}
