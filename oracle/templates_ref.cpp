// TEST INFRASTRUCTURE -- CPU oracle. Restatement of the reference's own templates, one function per
// template, statement order following the .circom source (cited per function). Values are
// written into the named witness table of include/hz_layout.h.
#include "templates_ref.h"

namespace orc {
using namespace hzl;

// src/lib/decode-float.circom:12-44 DecodeFloatBin
static F decode_float_bin(const W& w, const DecodeFloatOff& o, const std::vector<int>& bits /*40*/) {
    F pe = F(9) * F(bits[35]) + F(1);
    F ten(10), p10 = ten;  // 10^(2^i)
    for (int i = 1; i < 5; i++) {
        p10 = p10 * p10;  // 10^(2^i)
        pe = (pe * p10 - pe) * F(bits[35 + i]) + pe;
        w.set(o.pe + (i - 1), pe);
    }
    const F lcm = bits2num(bits, 0, 35);
    const F out = lcm * pe;
    w.set(o.out, out);
    return out;
}

// src/lib/hash-state.circom:18-40
static F hash_state(const W& w, PoseidonOff off, const F& tokenID, const F& nonce, const F& sign, const F& balance, const F& ay, const F& ethAddr) {
    const F e0 = tokenID + nonce * pow2(32) + sign * pow2(72);
    F in[4] = {e0, balance, ay, ethAddr};
    return poseidon_w(w, off, in, 4);
}

// src/decode-tx.circom:44-369
DecOut decode_tx(const W& w, const DecOff& o, int L, const DecIn& in) {
    DecOut r;
    const F notOn = F(1) - in.onChain;
    const std::vector<int> d = num2bits(w, o.n2bData, in.txCompressedData, 225, C_DEC_N2B_DATA);
    const F constSig = bits2num(d, 0, 32), chainID = bits2num(d, 32, 16);
    r.fromIdx = bits2num(d, 48, 48);
    F padFrom(0), padTo(0);
    for (int i = L; i < 48; i++) { padFrom += F(d[48 + i]); padTo += F(d[96 + i]); }
    w.chk(C_DEC_PAD_FROM, padFrom, F(0));
    r.toIdx = bits2num(d, 96, 48);
    w.chk(C_DEC_PAD_TO, padTo, F(0));
    r.tokenID = bits2num(d, 144, 32);
    r.nonce = bits2num(d, 176, 40);
    r.userFee = bits2num(d, 216, 8);
    r.toBjjSign = F(d[224]);
    const std::vector<int> am = num2bits(w, o.n2bAmount, in.amountF, 40, C_DEC_N2B_AMOUNT);
    r.amount = decode_float_bin(w, o.dfAmount, am);
    // txCompressedDataV2 (:174-212)
    {
        F v2(0);
        int k = 0;
        auto put = [&](int bit) { const F x = F(bit) * notOn; w.set(o.v2in + k, x); v2 += x * pow2(k); k++; };
        for (int i = 0; i < 48; i++) put(d[48 + i]);
        for (int i = 0; i < 48; i++) put(d[96 + i]);
        for (int i = 0; i < 40; i++) put(am[i]);
        for (int i = 0; i < 32; i++) put(d[144 + i]);
        for (int i = 0; i < 40; i++) put(d[176 + i]);
        for (int i = 0; i < 8; i++) put(d[216 + i]);
        v2 += F(d[224]) * pow2(216);
        r.txCompressedDataV2 = v2;
    }
    // L1L2TxData (:214-247)
    const F tz = is_zero(w, o.toIdxIsZero, r.toIdx);
    const F sel_s = notOn * tz;
    const F finalTo = mux1(r.toIdx, in.auxToIdx, sel_s);
    w.set(o.selToIdx_s, sel_s);
    w.set(o.selToIdx_out, finalTo);
    const std::vector<int> ft = num2bits(w, o.n2bFinalToIdx, finalTo, L, C_DEC_N2B_FINALTOIDX);
    r.L1L2TxData.assign(2 * L + 48, F(0));
    for (int i = 0; i < L; i++) r.L1L2TxData[L - 1 - i] = F(d[48 + i]);
    for (int i = 0; i < L; i++) r.L1L2TxData[2 * L - 1 - i] = F(ft[i]);
    for (int i = 0; i < 40; i++) r.L1L2TxData[2 * L + 40 - 1 - i] = F(am[i]);
    for (int i = 0; i < 8; i++) {
        const F x = F(d[216 + i]) * notOn;
        r.L1L2TxData[2 * L + 48 - 1 - i] = x;
        w.set(o.l1l2Fee + (7 - i), x);  // stored in output order: index 0 = L1L2TxData[2L+40]
    }
    // sigL2Hash (:249-283)
    const std::vector<int> te = num2bits(w, o.n2bToEthAddr, in.toEthAddr, 160, C_DEC_N2B_TOETHADDR);
    const std::vector<int> mb = num2bits(w, o.n2bMaxNumBatch, in.maxNumBatch, 32, C_DEC_N2B_MAXNUMBATCH);
    const F e1 = bits2num(te, 0, 160) + bits2num(am, 0, 40) * pow2(160) + bits2num(mb, 0, 32) * pow2(200);
    {
        F hin[6] = {in.txCompressedData, e1, in.toBjjAy, in.rqTxCompressedDataV2, in.rqToEthAddr, in.rqToBjjAy};
        r.sigL2Hash = poseidon_w(w, o.hashSig, hin, 6);
    }
    // L1TxFullData (:285-324)
    const std::vector<int> fe = num2bits(w, o.n2bFromEthAddr, in.fromEthAddr, 160, C_DEC_N2B_FROMETHADDR);
    const std::vector<int> la = num2bits(w, o.n2bLoadAmountF, in.loadAmountF, 40, C_DEC_N2B_LOADAMOUNTF);
    r.L1TxFullData.assign(hzl::L1FULL_BITS, F(0));
    {
        auto put = [&](int pos, const F& bit) { const F x = bit * in.onChain; r.L1TxFullData[pos] = x; w.set(o.l1full + pos, x); };
        for (int i = 0; i < 160; i++) put(160 - 1 - i, F(fe[i]));
        for (int i = 0; i < 256; i++) put(160 + 256 - 1 - i, in.fromBjjCompressed[i]);
        for (int i = 0; i < 48; i++) put(160 + 256 + 48 - 1 - i, F(d[48 + i]));
        for (int i = 0; i < 40; i++) put(160 + 256 + 48 + 40 - 1 - i, F(la[i]));
        for (int i = 0; i < 40; i++) put(160 + 256 + 48 + 40 + 40 - 1 - i, F(am[i]));
        for (int i = 0; i < 32; i++) put(160 + 256 + 48 + 40 + 40 + 32 - 1 - i, F(d[144 + i]));
        for (int i = 0; i < 48; i++) put(160 + 256 + 48 + 40 + 40 + 32 + 48 - 1 - i, F(d[96 + i]));
    }
    // checks (:326-368)
    const F fz = is_zero(w, o.fromIdxIsZero, r.fromIdx);
    w.chk(C_DEC_NEWACCOUNT, in.onChain * fz, in.newAccount);
    const F onNew = in.onChain * in.newAccount;
    r.outIdx = in.inIdx + onNew;
    w.set(o.outIdx, r.outIdx);
    w.set(o.idxChecker_en, onNew);
    force_equal_if_enabled(w, o.idxChecker, onNew, in.auxFromIdx, r.outIdx, C_DEC_IDXCHECKER);
    w.chk(C_DEC_L1_BEFORE_L2, (F(1) - in.previousOnChain) * in.onChain, F(0));
    force_equal_if_enabled(w, o.chainIDChecker, notOn, in.globalChainID, chainID, C_DEC_CHAINID);
    force_equal_if_enabled(w, o.constSigChecker, notOn, constSig, F((uint64_t)3322668559ull), C_DEC_CONSTSIG);
    const F mz = is_zero(w, o.maxNumBatchIsZero, in.maxNumBatch);
    // GreaterEqThan(32)(maxNumBatch, currentNumBatch) = LessThan(32)(currentNumBatch, maxNumBatch+1)
    //   LessThan: n2b(33)(in0 + 2^32 - in1); out = 1 - bit32
    const std::vector<int> lt = num2bits(w, o.maxNumBatchLt, in.currentNumBatch + pow2(32) - (in.maxNumBatch + F(1)), 33, C_DEC_N2B_MAXNUMBATCH_LT);
    const F ok = F(1) - F(lt[32]);
    w.chk(C_DEC_MAXNUMBATCH, (F(1) - ok) * (F(1) - mz), F(0));
    r.onChain = in.onChain;
    if (o.o_fromIdx != ~0u) {
        w.set(o.o_fromIdx, r.fromIdx); w.set(o.o_toIdx, r.toIdx); w.set(o.o_tokenID, r.tokenID); w.set(o.o_nonce, r.nonce);
        w.set(o.o_userFee, r.userFee); w.set(o.o_toBjjSign, r.toBjjSign); w.set(o.o_amount, r.amount);
        w.set(o.o_sigL2Hash, r.sigL2Hash); w.set(o.o_v2, r.txCompressedDataV2);
        for (int i = 0; i < 2 * L + 48; i++) w.set(o.o_l1l2 + i, r.L1L2TxData[i]);
    }
    return r;
}

// src/rollup-tx-states.circom:39-314
struct StatesOut {
    F isP1Insert, isP2Insert, key1, key2, P1_fnc0, P1_fnc1, P2_fnc0, P2_fnc1, isExit, verifySignEnabled, nop, checkToEthAddr,
        checkToBjj, nullifyLoadAmount, nullifyAmount;
};
static StatesOut rollup_tx_states(const W& w, const StatesOff& o, const RtxIn& in, const F& loadAmount) {
    StatesOut r;
    const F one(1);
    const F selFrom_s = in.onChain * in.newAccount;
    const F finalFromIdx = mux1(in.fromIdx, in.auxFromIdx, selFrom_s);
    w.set(o.selFromIdx_s, selFrom_s); w.set(o.selFromIdx_out, finalFromIdx);
    const F tz = is_zero(w, o.toIdxIsZero, in.toIdx);
    const F selectAuxToIdx = (one - in.onChain) * tz;
    w.set(o.selectAuxToIdx, selectAuxToIdx);
    const F finalToIdx = mux1(in.toIdx, in.auxToIdx, selectAuxToIdx);
    w.set(o.selToIdx_out, finalToIdx);
    const F isAny = is_equal(w, o.isToEthAddrAny, pow2(160) - one, in.toEthAddr);
    r.isExit = is_equal(w, o.checkIsExit, one, finalToIdx);
    const F ffz = is_zero(w, o.finalFromIdxIsZero, finalFromIdx);
    const F isFinalFromIdx = one - ffz;
    const F isLoadAmount = one - is_zero(w, o.loadAmountIsZero, loadAmount);
    const F isAmount = one - is_zero(w, o.amountIsZero, in.amount);
    w.chk(C_RTX_ST_L2_LOADAMOUNT, (one - in.onChain) * isLoadAmount, F(0));
    w.chk(C_RTX_ST_L2_NEWACCOUNT, (one - in.onChain) * in.newAccount, F(0));
    r.isP1Insert = in.onChain * in.newAccount;
    r.P1_fnc0 = r.isP1Insert * isFinalFromIdx;
    r.P1_fnc1 = (one - r.isP1Insert) * isFinalFromIdx;
    w.set(o.isP1Insert, r.isP1Insert); w.set(o.P1_fnc0, r.P1_fnc0); w.set(o.P1_fnc1, r.P1_fnc1);
    {   // Mux2: c = [0, f, f, f], s = [P1_fnc0, P1_fnc1]
        const F c0(0), c1 = finalFromIdx, c2 = finalFromIdx, c3 = finalFromIdx;
        const F s10 = r.P1_fnc1 * r.P1_fnc0;
        const F a10 = (c3 - c2 - c1 + c0) * s10, a1 = (c2 - c0) * r.P1_fnc1, a0 = (c1 - c0) * r.P1_fnc0;
        w.set(o.mux1 + M2_S10, s10); w.set(o.mux1 + M2_A10, a10); w.set(o.mux1 + M2_A1, a1); w.set(o.mux1 + M2_A0, a0);
        r.key1 = a10 + a1 + a0 + c0;
    }
    r.isP2Insert = r.isExit * in.newExit;
    r.P2_fnc0 = r.isP2Insert * isFinalFromIdx;
    r.P2_fnc1 = (one - r.isP2Insert) * isFinalFromIdx;
    w.set(o.isP2Insert, r.isP2Insert); w.set(o.P2_fnc0, r.P2_fnc0); w.set(o.P2_fnc1, r.P2_fnc1);
    {   // Mux2: c = [0, finalToIdx, 0, finalFromIdx], s = [isAmount, isExit]
        const F c0(0), c1 = finalToIdx, c2(0), c3 = finalFromIdx;
        const F s10 = r.isExit * isAmount;
        const F a10 = (c3 - c2 - c1 + c0) * s10, a1 = (c2 - c0) * r.isExit, a0 = (c1 - c0) * isAmount;
        w.set(o.mux2 + M2_S10, s10); w.set(o.mux2 + M2_A10, a10); w.set(o.mux2 + M2_A1, a1); w.set(o.mux2 + M2_A0, a0);
        r.key2 = a10 + a1 + a0 + c0;
    }
    r.verifySignEnabled = (one - in.onChain) * isFinalFromIdx;
    w.set(o.verifySignEnabled, r.verifySignEnabled);
    r.nop = ffz;
    const F tmpE = (one - isAny) * selectAuxToIdx, tmpB = isAny * selectAuxToIdx;
    r.checkToEthAddr = tmpE * (one - r.nop);
    r.checkToBjj = tmpB * (one - r.nop);
    w.set(o.tmpCheckToEthAddr, tmpE); w.set(o.tmpCheckToBjj, tmpB); w.set(o.checkToEthAddr, r.checkToEthAddr); w.set(o.checkToBjj, r.checkToBjj);
    const F onNotCreate = (one - in.newAccount) * in.onChain;
    const F shouldEth = onNotCreate * isAmount;
    w.set(o.onChainNotCreateAccount, onNotCreate); w.set(o.shouldCheckEthAddr, shouldEth);
    const F eqEth = is_equal(w, o.checkFromEthAddr, in.fromEthAddr, in.ethAddr1);
    const F nullEth = shouldEth * (one - eqEth);
    w.set(o.applyNullifierEthAddr, nullEth);
    const F eqT1 = is_equal(w, o.checkTokenID1, in.tokenID, in.tokenID1);
    const F nullT1 = onNotCreate * (one - eqT1);
    w.set(o.applyNullifierTokenID1, nullT1);
    const F sc20 = in.onChain * isAmount;
    const F sc21 = sc20 * (one - r.isP2Insert);
    w.set(o.shouldCheckTokenID2_0, sc20); w.set(o.shouldCheckTokenID2_1, sc21);
    const F eqT2 = is_equal(w, o.checkTokenID2, in.tokenID, in.tokenID2);
    const F nullT2 = sc21 * (one - eqT2);
    w.set(o.applyNullifierTokenID2, nullT2);
    r.nullifyLoadAmount = nullT1 * isLoadAmount;
    w.set(o.nullifyLoadAmount, r.nullifyLoadAmount);
    const F applyT1Amt = nullT1 * isAmount;
    w.set(o.applyCheckTokenID1ToAmount, applyT1Amt);
    const F na0 = one - (one - nullEth) * (one - nullT2);
    r.nullifyAmount = one - (one - na0) * (one - applyT1Amt);
    w.set(o.nullifyAmount_0, na0); w.set(o.nullifyAmount, r.nullifyAmount);
    return r;
}

// circomlib mux3.circom MultiMux3(1); c[8], s[3]
static F mux3(const W& w, const Mux3Off& o, const F* c, const F* s) {
    const F s10 = s[1] * s[0];
    const F a210 = (c[7] - c[6] - c[5] + c[4] - c[3] + c[2] + c[1] - c[0]) * s10;
    const F a21 = (c[6] - c[4] - c[2] + c[0]) * s[1];
    const F a20 = (c[5] - c[4] - c[1] + c[0]) * s[0];
    const F a2 = c[4] - c[0];
    const F a10 = (c[3] - c[2] - c[1] + c[0]) * s10;
    const F a1 = (c[2] - c[0]) * s[1];
    const F a0 = (c[1] - c[0]) * s[0];
    const F a = c[0];
    const F out = (a210 + a21 + a20 + a2) * s[2] + (a10 + a1 + a0 + a);
    w.set(o.base + M3_S10, s10); w.set(o.base + M3_A210, a210); w.set(o.base + M3_A21, a21); w.set(o.base + M3_A20, a20);
    w.set(o.base + M3_A10, a10); w.set(o.base + M3_A1, a1); w.set(o.base + M3_A0, a0); w.set(o.base + M3_OUT, out);
    return out;
}

// src/rq-tx-verifier.circom:19-94
static void rq_tx_verifier(const W& w, const RtxOff& o, const RtxIn& in) {
    const std::vector<int> b = num2bits(w, o.rq_n2b, in.rqOffset, 3, C_RTX_RQ_N2B);
    const F s[3] = {F(b[0]), F(b[1]), F(b[2])};
    const F* fut[3] = {in.futureV2, in.futureToEthAddr, in.futureToBjjAy};
    const F* pst[3] = {in.pastV2, in.pastToEthAddr, in.pastToBjjAy};
    const F rq[3] = {in.rqTxCompressedDataV2, in.rqToEthAddr, in.rqToBjjAy};
    const int cid[3] = {C_RTX_RQ_V2, C_RTX_RQ_ETHADDR, C_RTX_RQ_BJJAY};
    for (int m = 0; m < 3; m++) {
        const F c[8] = {F(0), fut[m][0], fut[m][1], fut[m][2], pst[m][3], pst[m][2], pst[m][1], pst[m][0]};
        const F out = mux3(w, o.rq_mux[m], c, s);
        w.chk(cid[m], out, rq[m]);
    }
}

extern const uint64_t* fee_table();

// circomlib mux4.circom MultiMux4(1) with constant inputs: only s10,s20,s21,s210,out are variables
static F mux4_const(const W& w, uint32_t off, const F* c, const F* s) {
    const F s10 = s[1] * s[0], s20 = s[2] * s[0], s21 = s[2] * s[1], s210 = s21 * s[0];
    const F hi = (c[15] - c[14] - c[13] + c[12] - c[11] + c[10] + c[9] - c[8] - c[7] + c[6] + c[5] - c[4] + c[3] - c[2] - c[1] + c[0]) * s210 +
                 (c[14] - c[12] - c[10] + c[8] - c[6] + c[4] + c[2] - c[0]) * s21 + (c[13] - c[12] - c[9] + c[8] - c[5] + c[4] + c[1] - c[0]) * s20 +
                 (c[11] - c[10] - c[9] + c[8] - c[3] + c[2] + c[1] - c[0]) * s10 + (c[12] - c[8] - c[4] + c[0]) * s[2] +
                 (c[10] - c[8] - c[2] + c[0]) * s[1] + (c[9] - c[8] - c[1] + c[0]) * s[0] + (c[8] - c[0]);
    const F lo = (c[7] - c[6] - c[5] + c[4] - c[3] + c[2] + c[1] - c[0]) * s210 + (c[6] - c[4] - c[2] + c[0]) * s21 +
                 (c[5] - c[4] - c[1] + c[0]) * s20 + (c[3] - c[2] - c[1] + c[0]) * s10 + (c[4] - c[0]) * s[2] + (c[2] - c[0]) * s[1] +
                 (c[1] - c[0]) * s[0] + c[0];
    const F out = hi * s[3] + lo;
    w.set(off + MX4_S10, s10); w.set(off + MX4_S20, s20); w.set(off + MX4_S21, s21); w.set(off + MX4_S210, s210); w.set(off + MX4_OUT_C, out);
    return out;
}
// MultiMux4(1) with signal inputs: all product terms are variables
static F mux4_var(const W& w, uint32_t off, const F* c, const F* s) {
    const F s10 = s[1] * s[0], s20 = s[2] * s[0], s21 = s[2] * s[1], s210 = s21 * s[0];
    const F a3210 = (c[15] - c[14] - c[13] + c[12] - c[11] + c[10] + c[9] - c[8] - c[7] + c[6] + c[5] - c[4] + c[3] - c[2] - c[1] + c[0]) * s210;
    const F a321 = (c[14] - c[12] - c[10] + c[8] - c[6] + c[4] + c[2] - c[0]) * s21;
    const F a320 = (c[13] - c[12] - c[9] + c[8] - c[5] + c[4] + c[1] - c[0]) * s20;
    const F a310 = (c[11] - c[10] - c[9] + c[8] - c[3] + c[2] + c[1] - c[0]) * s10;
    const F a32 = (c[12] - c[8] - c[4] + c[0]) * s[2];
    const F a31 = (c[10] - c[8] - c[2] + c[0]) * s[1];
    const F a30 = (c[9] - c[8] - c[1] + c[0]) * s[0];
    const F a3 = c[8] - c[0];
    const F a210 = (c[7] - c[6] - c[5] + c[4] - c[3] + c[2] + c[1] - c[0]) * s210;
    const F a21 = (c[6] - c[4] - c[2] + c[0]) * s21;
    const F a20 = (c[5] - c[4] - c[1] + c[0]) * s20;
    const F a10 = (c[3] - c[2] - c[1] + c[0]) * s10;
    const F a2 = (c[4] - c[0]) * s[2];
    const F a1 = (c[2] - c[0]) * s[1];
    const F a0 = (c[1] - c[0]) * s[0];
    const F a = c[0];
    const F out = (a3210 + a321 + a320 + a310 + a32 + a31 + a30 + a3) * s[3] + (a210 + a21 + a20 + a10 + a2 + a1 + a0 + a);
    w.set(off + MX4_S10, s10); w.set(off + MX4_S20, s20); w.set(off + MX4_S21, s21); w.set(off + MX4_S210, s210);
    w.set(off + MX4V_A3210, a3210); w.set(off + MX4V_A321, a321); w.set(off + MX4V_A320, a320); w.set(off + MX4V_A310, a310);
    w.set(off + MX4V_A32, a32); w.set(off + MX4V_A31, a31); w.set(off + MX4V_A30, a30);
    w.set(off + MX4V_A210, a210); w.set(off + MX4V_A21, a21); w.set(off + MX4V_A20, a20); w.set(off + MX4V_A10, a10);
    w.set(off + MX4V_A2, a2); w.set(off + MX4V_A1, a1); w.set(off + MX4V_A0, a0); w.set(off + MX4V_OUT, out);
    return out;
}

// src/compute-fee.circom:12-94 (+ src/lib/mux256.circom:10-52)
static F compute_fee(const W& w, const ComputeFeeOff& o, const F& feeSel, const F& amount, const F& applyFee) {
    w.set(o.applyFee, applyFee);
    const std::vector<int> sb = num2bits(w, o.n2bFeeSel, feeSel, 8, C_RTX_FEE_N2B_SEL);
    F s[8];
    for (int i = 0; i < 8; i++) { s[i] = F(sb[i]) * applyFee; w.set(o.muxS + i, s[i]); }
    const uint64_t* tab = fee_table();
    F lvl1[16];
    for (int m = 0; m < 16; m++) {
        F c[16];
        for (int k = 0; k < 16; k++) c[k] = F(tab[16 * m + k]);
        lvl1[m] = mux4_const(w, o.mux1 + MX4C_N * m, c, s);
    }
    const F factor = mux4_var(w, o.mux2, lvl1, s + 4);
    const F notShifted = factor * amount;
    w.set(o.feeOutNotShifted, notShifted);
    const F applyShift = F(1) - F(sb[6]) * F(sb[7]);
    w.set(o.applyShift, applyShift);
    F lcIn(0), lcShifted(0), lcNot(0), ovS(0), ovN(0);
    for (int i = 0; i < 253; i++) {
        const int b = notShifted.bit(i);
        w.set(o.bits + i, F(b));
        if (!b) continue;
        lcIn += pow2(i);
        if (i >= 60) { if (i < 188) lcShifted += pow2(i - 60); else ovS += F(1); }
        if (i < 128) lcNot += pow2(i); else ovN += F(1);
    }
    w.chk(C_RTX_FEE_BITS, lcIn, notShifted);
    w.chk(C_RTX_FEE_OVF_SHIFTED, applyShift * ovS, F(0));
    w.chk(C_RTX_FEE_OVF_NOTSHIFTED, (F(1) - applyShift) * ovN, F(0));
    const F feeOut = applyShift * (lcShifted - lcNot) + lcNot;
    w.set(o.feeOut, feeOut);
    return feeOut;
}

// src/balance-updater.circom:24-105
struct BalOut { F newSender, newReceiver, isP2Nop, fee2Charge, isAmountNullified; };
static BalOut balance_updater(const W& w, const BalUpdOff& o, const F& oldSender, const F& oldReceiver, const F& amount, const F& loadAmount,
                              const F& feeSel, const F& onChain, const F& nop, const F& nullifyLoad, const F& nullifyAmt) {
    BalOut r;
    const F one(1);
    r.fee2Charge = compute_fee(w, o.fee, feeSel, amount, (one - onChain) * (one - nop));
    const F el1 = loadAmount * onChain, el2 = el1 * (one - nullifyLoad);
    const F ea1 = amount * (one - nop), ea2 = ea1 * (one - nullifyAmt);
    w.set(o.effLoad1, el1); w.set(o.effLoad2, el2); w.set(o.effAmt1, ea1); w.set(o.effAmt2, ea2);
    const std::vector<int> sb = num2bits(w, o.n2bSender, pow2(192) + oldSender + el2 - ea2 - r.fee2Charge, 193, C_RTX_BU_N2B_SENDER);
    const F underflowOk(sb[192]);
    w.chk(C_RTX_BU_UNDERFLOW, (one - underflowOk) * (one - onChain), F(0));
    const F ea3 = underflowOk * ea2;
    w.set(o.effAmt3, ea3);
    r.newSender = oldSender + el2 - ea3 - r.fee2Charge;
    r.newReceiver = oldReceiver + ea3;
    const F ez = is_zero(w, o.effAmtIsZero, ea1);
    r.isAmountNullified = one - (one - nullifyAmt) * underflowOk;
    w.set(o.isAmountNullified, r.isAmountNullified);
    r.isP2Nop = one - ez;
    return r;
}

// src/fee-accumulator.circom:17-91
static void fee_accumulator(const W& w, uint32_t off, int Fn, const F& tokenID, const F& fee2Charge, const F* plan, const F* accIn, F* accOut) {
    F selIn(0);
    for (int i = 0; i < Fn; i++) {
        const uint32_t b = off + FA_N * i;
        const F eq = is_equal(w, b + FA_ISZ_INV, tokenID, plan[i]);
        const F selOut = F(1) - (F(1) - eq) * (F(1) - selIn);
        const F ms = eq * (F(1) - selIn);
        const F out = mux1(accIn[i], accIn[i] + fee2Charge, ms);
        w.set(b + FA_SELOUT, selOut); w.set(b + FA_MUX_S, ms); w.set(b + FA_MUX_OUT, out);
        accOut[i] = out;
        selIn = selOut;
    }
}

// src/lib/utils-bjj.circom:37-58 AySign2Ax + circomlib pointbits.circom Bits2Point_Strict
static F ay_sign_2_ax(const W& w, const EddsaOff& o, const F& ay, const F& sign) {
    const std::vector<int> yb = num2bits(w, o.ax_n2bAy, ay, 254, C_RTX_AX_N2B_AY);
    const int ay_alias = comp_constant(w, o.ax_aliasY, yb, CT_MINUS1);
    w.chk(C_RTX_AX_ALIAS_Y, F(ay_alias), F(0));
    const F y = bits2num(yb, 0, 254);
    const F y2v = y * y;
    F x = fr_sqrt_circom((F(1) - y2v) / (BJ_A() - BJ_D() * y2v));
    if (sign == F(1)) x = -x;
    w.set(o.ax_x, x);
    const F x2 = x * x, y2 = y * y;
    w.set(o.ax_x2, x2); w.set(o.ax_y2, y2);
    w.chk(C_RTX_AX_BABYCHECK, BJ_A() * x2 + y2, F(1) + BJ_D() * x2 * y2);
    const std::vector<int> xb = num2bits(w, o.ax_n2bX, x, 254, C_RTX_AX_N2B_X);
    const int x_alias = comp_constant(w, o.ax_aliasX, xb, CT_MINUS1);
    w.chk(C_RTX_AX_ALIAS_X, F(x_alias), F(0));
    const int sg = comp_constant(w, o.ax_signCalc, xb, CT_HALF);
    w.chk(C_RTX_AX_SIGN, F(sg), sign);
    return x;
}

// circomlib escalarmulany.circom SegmentMulAny(n)
struct SegAnyOut { Pt out, dbl; };
static SegAnyOut segment_mul_any(const W& w, const SegAnyOff& o, const std::vector<int>& e, int e0, int n, const Pt& p) {
    const int cid = C_RTX_SIG_EC;
    const Pt m = edwards2montgomery(w, p, cid);
    w.set(o.e2m, m.x); w.set(o.e2m + 1, m.y);
    Pt dblIn = m, addIn = m;
    for (int i = 0; i < n - 1; i++) {
        const uint32_t b = o.bits + BIT_N * i;
        const MDblOut d = mont_dbl(w, dblIn, cid);
        const MAddOut a = mont_add(w, d.out, addIn, cid);
        const F sel(e[e0 + i + 1]);
        Pt so;
        so.x = (a.out.x - addIn.x) * sel + addIn.x;
        so.y = (a.out.y - addIn.y) * sel + addIn.y;
        w.set(b + BIT_DBL_X1_2, d.x1_2); w.set(b + BIT_DBL_LAMDA, d.lamda); w.set(b + BIT_DBL_OUT0, d.out.x); w.set(b + BIT_DBL_OUT1, d.out.y);
        w.set(b + BIT_ADD_LAMDA, a.lamda); w.set(b + BIT_ADD_OUT0, a.out.x); w.set(b + BIT_ADD_OUT1, a.out.y);
        w.set(b + BIT_SEL_OUT0, so.x); w.set(b + BIT_SEL_OUT1, so.y);
        dblIn = d.out;
        addIn = so;
    }
    SegAnyOut r;
    r.dbl = dblIn;
    const Pt me = montgomery2edwards(w, addIn, cid);
    w.set(o.m2e, me.x); w.set(o.m2e + 1, me.y);
    const Pt ea = baby_add(w, o.eadder, me, Pt{-p.x, p.y}, cid);
    const F sel(e[e0]);
    r.out.x = (me.x - ea.x) * sel + ea.x;
    r.out.y = (me.y - ea.y) * sel + ea.y;
    w.set(o.lastSel, r.out.x); w.set(o.lastSel + 1, r.out.y);
    return r;
}

// circomlib escalarmulfix.circom SegmentMulFix(nWindows) on a constant base (window tables are
// compile-time constants and not witness variables)
struct SegFixOut { Pt out, dbl; };
static SegFixOut segment_mul_fix(const W& w, const SegFixOff& o, const std::vector<int>& e, int e0, int nbits, const Pt& base) {
    const int cid = C_RTX_SIG_EC;
    W nw = w;  // constant parts: computed, not stored
    std::vector<F> scratch(1);
    const int nwin = o.nwin;
    Pt wbase = edwards2montgomery(nw, base, cid);
    Pt cacc = wbase;  // cadders chain
    std::vector<Pt> wout(nwin);
    Pt dblLast{F(0), F(0)};
    for (int i = 0; i < nwin; i++) {
        // WindowMulFix table: 1..8 times the window base
        Pt tab[8];
        tab[0] = wbase;
        tab[1] = mont_dbl(nw, wbase, cid).out;
        for (int k = 2; k < 8; k++) tab[k] = mont_add(nw, wbase, tab[k - 1], cid).out;
        int b[3];
        for (int j = 0; j < 3; j++) b[j] = (3 * i + j < nbits) ? e[e0 + 3 * i + j] : 0;
        const F s0(b[0]), s1(b[1]), s2(b[2]);
        const F s10 = s1 * s0;
        Pt mo;
        for (int c = 0; c < 2; c++) {
            F cc[8];
            for (int k = 0; k < 8; k++) cc[k] = c == 0 ? tab[k].x : tab[k].y;
            const F a210 = (cc[7] - cc[6] - cc[5] + cc[4] - cc[3] + cc[2] + cc[1] - cc[0]) * s10;
            const F a21 = (cc[6] - cc[4] - cc[2] + cc[0]) * s1, a20 = (cc[5] - cc[4] - cc[1] + cc[0]) * s0, a2 = cc[4] - cc[0];
            const F a10 = (cc[3] - cc[2] - cc[1] + cc[0]) * s10, a1 = (cc[2] - cc[0]) * s1, a0 = (cc[1] - cc[0]) * s0, a = cc[0];
            const F out = (a210 + a21 + a20 + a2) * s2 + (a10 + a1 + a0 + a);
            (c == 0 ? mo.x : mo.y) = out;
        }
        const uint32_t wb = o.windows + WIN_N * i;
        w.set(wb + WIN_S10, s10); w.set(wb + WIN_MUX0, mo.x); w.set(wb + WIN_MUX1, mo.y);
        wout[i] = mo;
        const Pt out8 = tab[7];
        if (i < nwin - 1) {
            if (i == 0) cacc = mont_add(nw, wbase, out8, cid).out;   // cadders[0]: in1 = e2m.out, in2 = windows[0].out8
            else cacc = mont_add(nw, cacc, out8, cid).out;
        } else {
            dblLast = mont_dbl(nw, out8, cid).out;
            if (i == 0) cacc = mont_add(nw, wbase, dblLast, cid).out;
            else cacc = mont_add(nw, cacc, dblLast, cid).out;
        }
        wbase = out8;
    }
    Pt acc = dblLast;
    for (int i = 0; i < nwin; i++) {
        const MAddOut a = mont_add(w, acc, wout[i], cid);
        const uint32_t wb = o.windows + WIN_N * i;
        w.set(wb + WIN_ADD_LAMDA, a.lamda); w.set(wb + WIN_ADD_OUT0, a.out.x); w.set(wb + WIN_ADD_OUT1, a.out.y);
        acc = a.out;
    }
    const Pt me = montgomery2edwards(w, acc, cid);
    w.set(o.m2e, me.x); w.set(o.m2e + 1, me.y);
    const Pt cme = montgomery2edwards(nw, cacc, cid);
    SegFixOut r;
    r.out = baby_add(w, o.cAdd, me, Pt{-cme.x, cme.y}, cid);
    r.dbl = wbase;  // windows[nWindows-1].out8
    return r;
}

// circomlib eddsaposeidon.circom EdDSAPoseidonVerifier
static void eddsa_poseidon_verifier(const W& w, const EddsaOff& o, const F& enabled, const F& Ax, const F& Ay, const F& S, const F& R8x,
                                    const F& R8y, const F& M) {
    const int cid = C_RTX_SIG_EC;
    std::vector<int> sb = num2bits(w, o.snum2bits, S, 253, C_RTX_SIG_N2B_S);
    sb.push_back(0);
    const int sgt = comp_constant(w, o.sCmp, sb, CT_SUBORDER_M1);
    w.chk(C_RTX_SIG_S_RANGE, F(sgt) * enabled, F(0));
    F hin[5] = {R8x, R8y, Ax, Ay, M};
    const F h = poseidon_w(w, o.hash, hin, 5);
    const std::vector<int> hb = num2bits_strict(w, o.h2bits, h, C_RTX_SIG_H_N2B, C_RTX_SIG_H_ALIAS);
    const Pt A{Ax, Ay};
    const Pt d1 = baby_add(w, o.dbl1, A, A, cid);
    const Pt d2 = baby_add(w, o.dbl2, d1, d1, cid);
    const Pt d3 = baby_add(w, o.dbl3, d2, d2, cid);
    const F az = is_zero(w, o.isZero, d2.x);  // isZero.in <== dbl3.x  (the INPUT x of dbl3)
    w.chk(C_RTX_SIG_A_NONZERO, az * enabled, F(0));
    // EscalarMulAny(254)
    const F zp = is_zero(w, o.zeropoint, d3.x);
    Pt p0;
    p0.x = d3.x + (BJ_BASE8().x - d3.x) * zp;
    p0.y = d3.y + (BJ_BASE8().y - d3.y) * zp;
    w.set(o.seg0p, p0.x); w.set(o.seg0p + 1, p0.y);
    const SegAnyOut s0 = segment_mul_any(w, o.seg[0], hb, 0, 148, p0);
    const MDblOut dd = mont_dbl(w, s0.dbl, cid);
    w.set(o.dblr, dd.x1_2); w.set(o.dblr + 1, dd.lamda); w.set(o.dblr + 2, dd.out.x); w.set(o.dblr + 3, dd.out.y);
    const Pt p1 = montgomery2edwards(w, dd.out, cid);
    w.set(o.m2e0, p1.x); w.set(o.m2e0 + 1, p1.y);
    const SegAnyOut s1 = segment_mul_any(w, o.seg[1], hb, 148, 106, p1);
    const Pt sum = baby_add(w, o.adders0, s0.out, s1.out, cid);
    Pt any;
    any.x = sum.x * (F(1) - zp);
    any.y = sum.y + (F(1) - sum.y) * zp;
    w.set(o.anyOut, any.x); w.set(o.anyOut + 1, any.y);
    const Pt right = baby_add(w, o.addRight, Pt{R8x, R8y}, any, cid);
    // EscalarMulFix(253, BASE8): segments of 246 bits
    const SegFixOut f0 = segment_mul_fix(w, o.fseg[0], sb, 0, 246, BJ_BASE8());
    W nw = w;
    const Pt b1 = montgomery2edwards(nw, f0.dbl, cid);  // constant
    const SegFixOut f1 = segment_mul_fix(w, o.fseg[1], sb, 246, 7, b1);
    const Pt left = baby_add(w, o.fadders0, f0.out, f1.out, cid);
    force_equal_if_enabled(w, o.eqCheckX, enabled, left.x, right.x, C_RTX_SIG_EQX);
    force_equal_if_enabled(w, o.eqCheckY, enabled, left.y, right.y, C_RTX_SIG_EQY);
}

// src/rollup-tx.circom:78-591
RtxOut rollup_tx(const W& w, const RtxOff& o, int L, int Fn, const RtxIn& in) {
    RtxOut r;
    const F one(1);
    // A
    const std::vector<int> lab = num2bits(w, o.n2bLoadAmountF, in.loadAmountF, 40, C_RTX_N2B_LOADAMOUNTF);
    const F loadAmount = decode_float_bin(w, o.dfLoadAmount, lab);
    const StatesOut st = rollup_tx_states(w, o.st, in, loadAmount);
    // B
    rq_tx_verifier(w, o, in);
    // C
    force_equal_if_enabled(w, o.nonceChecker, one - in.onChain, in.nonce, in.nonce1, C_RTX_NONCE);
    const F en_toEth = one - (one - st.checkToEthAddr) * (one - st.checkToBjj);
    w.set(o.checkToEthAddr_en, en_toEth);
    force_equal_if_enabled(w, o.checkToEthAddr, en_toEth, in.toEthAddr, in.ethAddr2, C_RTX_TOETHADDR);
    force_equal_if_enabled(w, o.toBjjAyChecker, st.checkToBjj, in.ay2, in.toBjjAy, C_RTX_TOBJJAY);
    force_equal_if_enabled(w, o.toBjjSignChecker, st.checkToBjj, in.sign2, in.toBjjSign, C_RTX_TOBJJSIGN);
    force_equal_if_enabled(w, o.checkTokenID1, one - in.onChain, in.tokenID, in.tokenID1, C_RTX_TOKENID1);
    const F en_t2 = (one - in.onChain) * (one - st.isP2Insert);
    w.set(o.checkTokenID2_en, en_t2);
    force_equal_if_enabled(w, o.checkTokenID2, en_t2, in.tokenID, in.tokenID2, C_RTX_TOKENID2);
    force_equal_if_enabled(w, o.checkTokenID1L1, st.isP1Insert, in.tokenID, in.tokenID1, C_RTX_TOKENID1_L1);
    force_equal_if_enabled(w, o.fromEthAddrChecker, st.isP1Insert, in.fromEthAddr, in.ethAddr1, C_RTX_FROMETHADDR);
    // D
    const F oldSt1 = hash_state(w, o.oldSt1Hash, in.tokenID1, in.nonce1, in.sign1, in.balance1, in.ay1, in.ethAddr1);
    const F oldSt2 = hash_state(w, o.oldSt2Hash, in.tokenID2, in.nonce2, in.sign2, in.balance2, in.ay2, in.ethAddr2);
    // E: BitsCompressed2AySign (src/lib/utils-bjj.circom:12-28)
    F bjjAy(0);
    for (int i = 0; i < 254; i++) bjjAy += in.fromBjjCompressed[i] * pow2(i);
    const F bjjSign = in.fromBjjCompressed[255];
    F mx[MX_N];
    mx[MX_S1BALANCE] = mux1(in.balance1, F(0), st.isP1Insert);
    mx[MX_S1SIGN] = mux1(in.sign1, bjjSign, st.isP1Insert);
    mx[MX_S1AY] = mux1(in.ay1, bjjAy, st.isP1Insert);
    mx[MX_S1NONCE] = mux1(in.nonce1, F(0), st.isP1Insert);
    mx[MX_S1ETHADDR] = mux1(in.ethAddr1, in.fromEthAddr, st.isP1Insert);
    mx[MX_S1TOKENID] = mux1(in.tokenID1, in.tokenID, st.isP1Insert);
    mx[MX_S1OLDKEY] = mux1(st.key1, in.oldKey1, st.isP1Insert);
    mx[MX_S1OLDVALUE] = mux1(oldSt1, in.oldValue1, st.isP1Insert);
    mx[MX_S2BALANCE] = mux1(in.balance2, F(0), st.isP2Insert);
    mx[MX_S2SIGN] = mux1(in.sign2, mx[MX_S1SIGN], st.isP2Insert);
    mx[MX_S2AY] = mux1(in.ay2, mx[MX_S1AY], st.isP2Insert);
    mx[MX_S2NONCE] = mux1(in.nonce2, F(0), st.isP2Insert);
    mx[MX_S2ETHADDR] = mux1(in.ethAddr2, mx[MX_S1ETHADDR], st.isP2Insert);
    mx[MX_S2TOKENID] = mux1(in.tokenID2, mx[MX_S1TOKENID], st.isP2Insert);
    mx[MX_S2OLDKEY] = mux1(st.key2, in.oldKey2, st.isP2Insert);
    mx[MX_S2OLDVALUE] = mux1(oldSt2, in.oldValue2, st.isP2Insert);
    for (int i = 0; i < MX_N; i++) w.set(o.mux16 + i, mx[i]);
    // F
    const F signSig = mux1(F(0), mx[MX_S1SIGN], st.verifySignEnabled);
    const F aySig = mux1(F(0), mx[MX_S1AY], st.verifySignEnabled);
    w.set(o.ed.signSignature, signSig); w.set(o.ed.aySignature, aySig);
    const F ax = ay_sign_2_ax(w, o.ed, aySig, signSig);
    eddsa_poseidon_verifier(w, o.ed, st.verifySignEnabled, ax, mx[MX_S1AY], in.s, in.r8x, in.r8y, in.sigL2Hash);
    // G
    const BalOut bu = balance_updater(w, o.bu, mx[MX_S1BALANCE], mx[MX_S2BALANCE], in.amount, loadAmount, in.userFee, in.onChain, st.nop,
                                      st.nullifyLoadAmount, st.nullifyAmount);
    r.isAmountNullified = bu.isAmountNullified;
    // H
    r.accFeeOut.resize(Fn);
    fee_accumulator(w, o.feeAcc, Fn, in.tokenID, bu.fee2Charge, in.feePlanTokens.data(), in.accFeeIn.data(), r.accFeeOut.data());
    // I
    const F newSt1 = hash_state(w, o.newSt1Hash, mx[MX_S1TOKENID], mx[MX_S1NONCE] + (one - in.onChain), mx[MX_S1SIGN], bu.newSender, mx[MX_S1AY], mx[MX_S1ETHADDR]);
    const F newSt2 = hash_state(w, o.newSt2Hash, mx[MX_S2TOKENID], mx[MX_S2NONCE], mx[MX_S2SIGN], bu.newReceiver, mx[MX_S2AY], mx[MX_S2ETHADDR]);
    // J
    static const SmtCids c1 = {C_RTX_P1_N2B_OLD, C_RTX_P1_ALIAS_OLD, C_RTX_P1_N2B_NEW, C_RTX_P1_ALIAS_NEW, C_RTX_P1_LEVINS, C_RTX_P1_SM_FINAL, C_RTX_P1_OLDROOT, C_RTX_P1_KEYS};
    static const SmtCids c2 = {C_RTX_P2_N2B_OLD, C_RTX_P2_ALIAS_OLD, C_RTX_P2_N2B_NEW, C_RTX_P2_ALIAS_NEW, C_RTX_P2_LEVINS, C_RTX_P2_SM_FINAL, C_RTX_P2_OLDROOT, C_RTX_P2_KEYS};
    const F p1root = smt_processor(w, o.p1, L + 1, in.oldStateRoot, in.siblings1.data(), mx[MX_S1OLDKEY], mx[MX_S1OLDVALUE], in.isOld0_1, st.key1,
                                   newSt1, st.P1_fnc0, st.P1_fnc1, c1);
    const F s3 = mux1(p1root, in.oldExitRoot, st.isExit);
    w.set(o.s3, s3);
    const F p2root = smt_processor(w, o.p2, L + 1, s3, in.siblings2.data(), mx[MX_S2OLDKEY], mx[MX_S2OLDVALUE], in.isOld0_2, st.key2, newSt2,
                                   st.P2_fnc0 * bu.isP2Nop, st.P2_fnc1 * bu.isP2Nop, c2);
    // K
    r.newStateRoot = mux1(p2root, p1root, st.isExit);
    r.newExitRoot = mux1(in.oldExitRoot, p2root, st.isExit);
    w.set(o.s4, r.newStateRoot); w.set(o.s5, r.newExitRoot);
    return r;
}

// src/fee-tx.circom:26-112
F fee_tx(const W& w, const FeeTxOff& o, int L, const FeeIn& in) {
    const F fz = is_zero(w, o.feeIdxIsZero, in.feeIdx);
    force_equal_if_enabled(w, o.tokenIDChecker, F(1) - fz, in.feePlanToken, in.tokenID, C_FEE_TOKENID);
    const F fnc0(0), fnc1 = F(1) - fz;
    const F oldH = hash_state(w, o.oldHash, in.tokenID, in.nonce, in.sign, in.balance, in.ay, in.ethAddr);
    const F newH = hash_state(w, o.newHash, in.tokenID, in.nonce, in.sign, in.accFee + in.balance, in.ay, in.ethAddr);
    static const SmtCids c = {C_FEE_P_N2B_OLD, C_FEE_P_ALIAS_OLD, C_FEE_P_N2B_NEW, C_FEE_P_ALIAS_NEW, C_FEE_P_LEVINS, C_FEE_P_SM_FINAL, C_FEE_P_OLDROOT, C_FEE_P_KEYS};
    const F root = smt_processor(w, o.p, L + 1, in.oldStateRoot, in.siblings.data(), in.feeIdx, oldH, F(0), in.feeIdx, newH, fnc0, fnc1, c);
    if (o.o_newStateRoot != ~0u) w.set(o.o_newStateRoot, root);
    return root;
}

F hash_state_main(const W& w, const HashStateOff& o, const F& tokenID, const F& nonce, const F& sign, const F& balance, const F& ay, const F& ethAddr) {
    const F out = hash_state(w, o.hash, tokenID, nonce, sign, balance, ay, ethAddr);
    w.set(o.out, out);
    return out;
}

// The gadget templates instantiated as `component main` by the reference's unit suites (test/lib/decode-float.test.js,
// test/compute-fee.test.js, test/fee-accumulator.test.js, test/balance-updater.test.js, test/rollup-tx-states.test.js,
// test/rq-tx-verifier.test.js): inputs/outputs per GadIO, internals through the same functions RollupTx uses.
void gadget_main(const W& w, const Layout& lo) {
    const GadIO& g = lo.gad;
    auto in = [&](int k, int j = 0) { return w.get(g.in[k] + j); };
    switch (lo.p.tmpl) {
        case T_DECODE_FLOAT: {   // src/lib/decode-float.circom:50-64
            const std::vector<int> b = num2bits(w, lo.rtx.n2bLoadAmountF, in(0), 40, C_RTX_N2B_LOADAMOUNTF);
            w.set(g.out[0], decode_float_bin(w, lo.rtx.dfLoadAmount, b));
            break;
        }
        case T_COMPUTE_FEE:      // src/compute-fee.circom:12-109 (feeOut and applyFee are the gadget's own signals)
            compute_fee(w, lo.rtx.bu.fee, in(0), in(1), in(2));
            break;
        case T_FEE_ACCUMULATOR: {
            const int Fn = lo.p.F;
            std::vector<F> plan(Fn), acc(Fn), out(Fn);
            for (int i = 0; i < Fn; i++) { plan[i] = in(2, i); acc[i] = in(3, i); }
            fee_accumulator(w, lo.rtx.feeAcc, Fn, in(0), in(1), plan.data(), acc.data(), out.data());
            for (int i = 0; i < Fn; i++) w.set(g.out[0] + i, out[i]);
            break;
        }
        case T_BALANCE_UPDATER: {
            const BalOut r = balance_updater(w, lo.rtx.bu, in(0), in(1), in(2), in(3), in(4), in(5), in(6), in(7), in(8));
            w.set(g.out[0], r.newSender); w.set(g.out[1], r.newReceiver); w.set(g.out[2], r.isP2Nop); w.set(g.out[3], r.fee2Charge);
            break;
        }
        case T_ROLLUP_TX_STATES: {
            RtxIn ri;
            ri.fromIdx = in(0); ri.toIdx = in(1); ri.toEthAddr = in(2); ri.auxFromIdx = in(3); ri.auxToIdx = in(4); ri.amount = in(5);
            ri.newExit = in(6); ri.newAccount = in(8); ri.onChain = in(9); ri.fromEthAddr = in(10); ri.ethAddr1 = in(11);
            ri.tokenID = in(12); ri.tokenID1 = in(13); ri.tokenID2 = in(14);
            const StatesOut r = rollup_tx_states(w, lo.rtx.st, ri, in(7));
            w.set(g.out[2], r.key1); w.set(g.out[3], r.key2); w.set(g.out[8], r.isExit); w.set(g.out[10], r.nop);
            break;
        }
        case T_RQ_TX_VERIFIER: {
            RtxIn ri;
            for (int j = 0; j < 3; j++) { ri.futureV2[j] = in(0, j); ri.futureToEthAddr[j] = in(2, j); ri.futureToBjjAy[j] = in(4, j); }
            for (int j = 0; j < 4; j++) { ri.pastV2[j] = in(1, j); ri.pastToEthAddr[j] = in(3, j); ri.pastToBjjAy[j] = in(5, j); }
            ri.rqTxCompressedDataV2 = in(6); ri.rqToEthAddr = in(7); ri.rqToBjjAy = in(8); ri.rqOffset = in(9);
            rq_tx_verifier(w, lo.rtx, ri);
            break;
        }
        case T_MUX256: {         // src/lib/mux256.circom:10-52: 16 Mux4 on s[0..3], one Mux4 on s[4..7]
            F sel[8], lvl1[16];
            for (int i = 0; i < 8; i++) sel[i] = in(0, i);
            for (int m = 0; m < 16; m++) {
                F c[16];
                for (int k = 0; k < 16; k++) c[k] = in(1, 16 * m + k);
                lvl1[m] = mux4_var(w, g.mux + MX4V_N * m, c, sel);
            }
            w.set(g.out[0], mux4_var(w, g.mux + MX4V_N * 16, lvl1, sel + 4));
            break;
        }
        case T_BITS2AYSIGN: {    // src/lib/utils-bjj.circom:12-28
            F ay(0);
            for (int i = 0; i < 254; i++) ay += in(0, i) * pow2(i);
            w.set(g.out[0], ay); w.set(g.out[1], in(0, 255));
            break;
        }
        case T_AYSIGN2AX:        // src/lib/utils-bjj.circom:37-58
            w.set(g.out[0], ay_sign_2_ax(w, lo.rtx.ed, in(0), in(1)));
            break;
        default: break;
    }
}

}  // namespace orc
