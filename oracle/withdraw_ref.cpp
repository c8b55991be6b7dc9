// TEST INFRASTRUCTURE -- CPU oracle. Withdraw(nLevels) (reference src/withdraw.circom:21-176) with
// circomlib smt/smtverifier.circom (+ smtverifierlevel, smtverifiersm) restated from the published
// templates.
#include "templates_ref.h"

namespace orc {
using namespace hzl;

void smt_verifier(const W& w, const SmtVerOff& o, int n, const F& enabled, const F& root, const F* siblings, const F& oldKey,
                  const F& oldValue, const F& isOld0, const F& key, const F& value, const F& fnc, const SmtVerCids& c) {
    const F h1old = smt_hash1(w, o.hash1Old, oldKey, oldValue);
    const F h1new = smt_hash1(w, o.hash1New, key, value);
    num2bits_strict(w, o.n2bOld, oldKey, c.n2b_old, c.alias_old);
    const std::vector<int> bNew = num2bits_strict(w, o.n2bNew, key, c.n2b_new, c.alias_new);
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
    std::vector<F> st_top(n), st_i0(n), st_iold(n), st_inew(n), st_na(n);
    F p_top = enabled, p_i0(0), p_iold(0), p_inew(0), p_na = F(1) - enabled;
    for (int i = 0; i < n; i++) {
        const F ptli = p_top * levIns[i];
        const F ptlif = ptli * fnc;
        st_top[i] = p_top - ptli;
        st_inew[i] = ptli - ptlif;
        st_iold[i] = ptlif * (F(1) - isOld0);
        st_i0[i] = ptli * isOld0;
        st_na[i] = p_na + p_inew + p_iold + p_i0;
        w.set(o.sm + VSM_N * i + VSM_PTLI, ptli); w.set(o.sm + VSM_N * i + VSM_PTLIF, ptlif);
        w.set(o.sm + VSM_N * i + VSM_IOLD, st_iold[i]); w.set(o.sm + VSM_N * i + VSM_I0, st_i0[i]);
        p_top = st_top[i]; p_i0 = st_i0[i]; p_iold = st_iold[i]; p_inew = st_inew[i]; p_na = st_na[i];
    }
    w.chk(c.sm_final, st_na[n - 1] + st_iold[n - 1] + st_inew[n - 1] + st_i0[n - 1], F(1));
    F child(0);
    for (int i = n - 1; i >= 0; i--) {
        const uint32_t lv = o.levels + VL_SIZE * i;
        const F sel(bNew[i]);
        const F aux = (siblings[i] - child) * sel;
        const F Lh = aux + child, Rh = -aux + siblings[i];
        w.set(lv + VL_SW_AUX, aux);
        const F ph = smt_hash2(w, lv + VL_HASH, Lh, Rh);
        const F a0 = ph * st_top[i], a1 = h1old * st_iold[i];
        const F rt = a0 + a1 + h1new * st_inew[i];
        w.set(lv + VL_AUX0, a0); w.set(lv + VL_AUX1, a1); w.set(lv + VL_ROOT, rt);
        child = rt;
    }
    const F keq = is_equal(w, o.keyEq, oldKey, key);
    // MultiAND(4)(fnc, 1-isOld0, keq, enabled): ands[0]=AND(in0,in1), ands[1]=AND(in2,in3), and2
    const F aa = fnc * (F(1) - isOld0), ab = keq * enabled, ac = aa * ab;
    w.set(o.and_a, aa); w.set(o.and_b, ab); w.set(o.and_c, ac);
    w.chk(c.keys, ac, F(0));
    force_equal_if_enabled(w, o.checkRoot, enabled, child, root, c.root);
}

F withdraw_main(const W& w, const WithdrawOff& o, int L) {
    const F rootExit = w.get(o.rootExit), ethAddr = w.get(o.ethAddr), tokenID = w.get(o.tokenID), balance = w.get(o.balance), idx = w.get(o.idx),
            sign = w.get(o.sign), ay = w.get(o.ay);
    std::vector<F> sib(L + 1);
    for (int i = 0; i <= L; i++) sib[i] = w.get(o.siblingsState + i);
    // HashState with nonce 0 (:37-43)
    const F e0 = tokenID + sign * pow2(72);
    F hin[4] = {e0, balance, ay, ethAddr};
    const F st = poseidon_w(w, o.accountState, hin, 4);
    static const SmtVerCids cids{C_WD_N2B_OLD, C_WD_N2B_OLD, C_WD_N2B_NEW, C_WD_ALIAS_NEW, C_WD_LEVINS, C_WD_SM_FINAL, C_WD_KEYS, C_WD_ROOT};
    smt_verifier(w, o.ver, L + 1, F(1), rootExit, sib.data(), F(0), F(0), F(0), idx, st, F(0), cids);
    // HashInputsWithdrawal (:84-176)
    const std::vector<int> bR = num2bits(w, o.n2bRootExit, rootExit, 256, C_WD_HI_N2B);
    const std::vector<int> bE = num2bits(w, o.n2bEthAddr, ethAddr, 160, C_WD_HI_N2B);
    const std::vector<int> bT = num2bits(w, o.n2bTokenID, tokenID, 32, C_WD_HI_N2B);
    const std::vector<int> bB = num2bits(w, o.n2bBalance, balance, 192, C_WD_HI_N2B);
    const std::vector<int> bI = num2bits(w, o.n2bIdx, idx, 48, C_WD_HI_N2B);
    F pad(0);
    for (int j = L; j < 48; j++) pad += F(bI[j]);
    w.chk(C_WD_HI_PAD, pad, F(0));
    std::vector<int> msg;
    auto be = [&](const std::vector<int>& b, int n) { for (int i = n - 1; i >= 0; i--) msg.push_back(b[i]); };
    be(bR, 256); be(bE, 160); be(bT, 32); be(bB, 192); be(bI, 48);
    const std::vector<int> dg = sha256_bits(w, o.sha, msg);
    F out(0);
    for (int i = 0; i < 256; i++)
        if (dg[255 - i]) out += pow2(i);
    w.set(o.hashGlobalInputs, out);
    return out;
}

}  // namespace orc
