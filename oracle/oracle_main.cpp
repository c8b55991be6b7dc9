// TEST INFRASTRUCTURE -- CPU oracle: the main components (RollupMain src/rollup-main.circom:82-475,
// Withdraw src/withdraw.circom:21-72, and the sub-templates the reference's suites instantiate as
// `component main`) and the C entry points mirroring include/hermez_witness.h.
#include <memory>
#include "oracle_api.h"
#include "templates_ref.h"

using namespace orc;
using namespace hzl;

namespace orc {
F withdraw_main(const W& w, const WithdrawOff& o, int L);  // withdraw_ref.cpp
}

struct OrcCtx {
    Layout lo;
    std::vector<F> vals;
    std::vector<uint8_t> written;
    std::vector<uint8_t> input_set;
    Fail fail;
    bool log_poseidon = false;                          // orc_log_poseidon: keep every Poseidon component's inputs of the next runs
    std::map<uint64_t, std::vector<F>> poseidon_in;     // physical index of the component's first stored signal -> its inputs
};

static bool lt_p(const uint8_t* b) {
    uint64_t c[4];
    memcpy(c, b, 32);
    return !F::geq_p(c);
}

extern "C" void* orc_ctx_create(int tmpl, int nTx, int nLevels, int maxL1Tx, int maxFeeTx, int n_instances) {
    auto* c = new OrcCtx();
    Params p;
    p.tmpl = tmpl; p.nTx = nTx; p.L = nLevels; p.maxL1 = maxL1Tx; p.F = maxFeeTx; p.n_inst = n_instances > 0 ? n_instances : 1;
    build_layout(p, c->lo);
    c->vals.assign(c->lo.total, F(0));
    c->written.assign(c->lo.total, 0);
    c->input_set.assign(c->lo.inputs.size(), 0);
    return c;
}
extern "C" void orc_ctx_destroy(void* h) { delete (OrcCtx*)h; }
extern "C" uint64_t orc_witness_len(void* h) { return ((OrcCtx*)h)->lo.per_instance; }
extern "C" uint64_t orc_witness_total(void* h) { return ((OrcCtx*)h)->lo.total; }

extern "C" int orc_set_input(void* h, int instance, const char* name, const uint8_t* v, size_t count) {
    OrcCtx* c = (OrcCtx*)h;
    const InputDesc* d = c->lo.find_input(name);
    if (!d) return 4;
    const Section& s = c->lo.sections[d->section];
    const uint32_t B = c->lo.n_inst;
    for (size_t i = 0; i < count; i++)
        if (!lt_p(v + 32 * i)) return 4;
    const size_t per = (size_t)d->inner * d->outer;
    uint32_t b0, b1;
    if (instance >= 0) {
        if ((uint32_t)instance >= B || count != per) return 4;
        b0 = (uint32_t)instance; b1 = b0 + 1;
    } else {
        if (count != per * B) return 4;
        b0 = 0; b1 = B;
    }
    for (uint32_t b = b0; b < b1; b++)
        for (uint32_t u = 0; u < d->outer; u++)
            for (uint32_t k = 0; k < d->inner; k++) {
                const uint64_t p = c->lo.phys(d->section, d->off + k, b * s.upi + u);
                c->vals[p] = F::from_bytes(v + 32 * (((size_t)(b - b0) * d->outer + u) * d->inner + k));
                c->written[p] = 1;
            }
    c->input_set[d - &c->lo.inputs[0]] = 1;
    return 0;
}

static void run_rollup_main(OrcCtx* c, uint32_t b) {
    const Layout& lo = c->lo;
    const int L = lo.p.L, Fn = lo.p.F, nTx = lo.p.nTx, maxL1 = lo.p.maxL1;
    const uint32_t tx0 = b * (uint32_t)lo.p.nTx, fee0 = b * (uint32_t)lo.p.F;
    W g{&lo, &c->vals, &c->written, &c->fail, lo.sec_glob, b, b, 0};
    g.set(lo.g.one, F(1));
    const MainTxInOff& m = lo.mi;
    std::vector<DecOut> dec(nTx);
    std::vector<RtxOut> rtx(nTx);
    HashInputsIn hin;
    hin.L1L2TxsData.resize((size_t)nTx * (2 * L + 48));
    hin.L1TxsFullData.resize((size_t)maxL1 * hzl::L1FULL_BITS);
    for (int i = 0; i < nTx; i++) {
        W w{&lo, &c->vals, &c->written, &c->fail, lo.sec_tx, tx0 + (uint32_t)i, b, i};
        auto in = [&](uint32_t off) { return w.get(off); };
        auto inu = [&](uint32_t off, int u) { return w.get_unit(off, tx0 + (uint32_t)u); };
        // A (:207-219)
        if (i < nTx - 1) w.chk(C_MAIN_IMONCHAIN_BOOL, in(m.imOnChain) * (in(m.imOnChain) - F(1)), F(0));
        w.chk(C_MAIN_ONCHAIN_BOOL, in(m.onChain) * (in(m.onChain) - F(1)), F(0));
        w.chk(C_MAIN_NEWACCOUNT_BOOL, in(m.newAccount) * (in(m.newAccount) - F(1)), F(0));
        for (int j = 0; j < 256; j++) w.chk(C_MAIN_BJJ_BOOL, in(m.fromBjjCompressed + j) * (in(m.fromBjjCompressed + j) - F(1)), F(0));
        w.chk(C_MAIN_ISOLD0_1_BOOL, in(m.isOld0_1) * (in(m.isOld0_1) - F(1)), F(0));
        w.chk(C_MAIN_ISOLD0_2_BOOL, in(m.isOld0_2) * (in(m.isOld0_2) - F(1)), F(0));
        // B (:223-254)
        DecIn di;
        di.previousOnChain = i == 0 ? F(1) : inu(m.imOnChain, i - 1);
        di.inIdx = i == 0 ? g.get(lo.g.oldLastIdx) : inu(m.imOutIdx, i - 1);
        di.txCompressedData = in(m.txCompressedData); di.amountF = in(m.amountF); di.toEthAddr = in(m.toEthAddr); di.toBjjAy = in(m.toBjjAy);
        di.rqTxCompressedDataV2 = in(m.rqTxCompressedDataV2); di.rqToEthAddr = in(m.rqToEthAddr); di.rqToBjjAy = in(m.rqToBjjAy);
        di.fromEthAddr = in(m.fromEthAddr); di.loadAmountF = in(m.loadAmountF);
        di.fromBjjCompressed.resize(256);
        for (int j = 0; j < 256; j++) di.fromBjjCompressed[j] = in(m.fromBjjCompressed + j);
        di.globalChainID = g.get(lo.g.globalChainID); di.currentNumBatch = g.get(lo.g.currentNumBatch);
        di.maxNumBatch = in(m.maxNumBatch); di.onChain = in(m.onChain); di.newAccount = in(m.newAccount);
        di.auxFromIdx = in(m.auxFromIdx); di.auxToIdx = in(m.auxToIdx);
        dec[i] = decode_tx(w, lo.dec, L, di);
        // C (:258-265)
        w.chk(C_MAIN_IM_V2, dec[i].txCompressedDataV2, in(m.txCompressedDataV2));
        if (i < nTx - 1) {
            w.chk(C_MAIN_IM_ONCHAIN, dec[i].onChain, in(m.imOnChain));
            w.chk(C_MAIN_IM_OUTIDX, dec[i].outIdx, in(m.imOutIdx));
        }
        // D (:269-379)
        RtxIn ri;
        ri.feePlanTokens.resize(Fn); ri.accFeeIn.resize(Fn);
        W fw{&lo, &c->vals, nullptr, &c->fail, lo.sec_fee, fee0, b, 0};
        for (int j = 0; j < Fn; j++) {
            ri.feePlanTokens[j] = fw.get_unit(lo.fi.feePlanTokens, fee0 + (uint32_t)j);
            ri.accFeeIn[j] = i == 0 ? F(0) : inu(m.imAccFeeOut + j, i - 1);
        }
        for (int j = 0; j < 3; j++) {
            const bool ok = i + j + 1 < nTx;
            ri.futureV2[j] = ok ? inu(m.txCompressedDataV2, i + j + 1) : F(0);
            ri.futureToEthAddr[j] = ok ? inu(m.toEthAddr, i + j + 1) : F(0);
            ri.futureToBjjAy[j] = ok ? inu(m.toBjjAy, i + j + 1) : F(0);
        }
        for (int j = 0; j < 4; j++) {
            const bool ok = i - j - 1 >= 0;
            ri.pastV2[j] = ok ? inu(m.txCompressedDataV2, i - j - 1) : F(0);
            ri.pastToEthAddr[j] = ok ? inu(m.toEthAddr, i - j - 1) : F(0);
            ri.pastToBjjAy[j] = ok ? inu(m.toBjjAy, i - j - 1) : F(0);
        }
        ri.fromIdx = dec[i].fromIdx; ri.auxFromIdx = in(m.auxFromIdx); ri.toIdx = dec[i].toIdx; ri.auxToIdx = in(m.auxToIdx);
        ri.toBjjAy = in(m.toBjjAy); ri.toBjjSign = dec[i].toBjjSign; ri.toEthAddr = in(m.toEthAddr); ri.amount = dec[i].amount;
        ri.tokenID = dec[i].tokenID; ri.nonce = dec[i].nonce; ri.userFee = dec[i].userFee; ri.rqOffset = in(m.rqOffset);
        ri.onChain = in(m.onChain); ri.newAccount = in(m.newAccount); ri.rqTxCompressedDataV2 = in(m.rqTxCompressedDataV2);
        ri.rqToEthAddr = in(m.rqToEthAddr); ri.rqToBjjAy = in(m.rqToBjjAy); ri.sigL2Hash = dec[i].sigL2Hash; ri.s = in(m.s);
        ri.r8x = in(m.r8x); ri.r8y = in(m.r8y); ri.fromEthAddr = in(m.fromEthAddr); ri.loadAmountF = in(m.loadAmountF);
        ri.fromBjjCompressed = di.fromBjjCompressed;
        ri.tokenID1 = in(m.tokenID1); ri.nonce1 = in(m.nonce1); ri.sign1 = in(m.sign1); ri.balance1 = in(m.balance1); ri.ay1 = in(m.ay1);
        ri.ethAddr1 = in(m.ethAddr1); ri.isOld0_1 = in(m.isOld0_1); ri.oldKey1 = in(m.oldKey1); ri.oldValue1 = in(m.oldValue1);
        ri.tokenID2 = in(m.tokenID2); ri.nonce2 = in(m.nonce2); ri.sign2 = in(m.sign2); ri.balance2 = in(m.balance2); ri.newExit = in(m.newExit);
        ri.ay2 = in(m.ay2); ri.ethAddr2 = in(m.ethAddr2); ri.isOld0_2 = in(m.isOld0_2); ri.oldKey2 = in(m.oldKey2); ri.oldValue2 = in(m.oldValue2);
        ri.siblings1.resize(L + 1); ri.siblings2.resize(L + 1);
        for (int j = 0; j <= L; j++) { ri.siblings1[j] = in(m.siblings1 + j); ri.siblings2[j] = in(m.siblings2 + j); }
        ri.oldStateRoot = i == 0 ? g.get(lo.g.oldStateRoot) : inu(m.imStateRoot, i - 1);
        ri.oldExitRoot = i == 0 ? F(0) : inu(m.imExitRoot, i - 1);
        rtx[i] = rollup_tx(w, lo.rtx, L, Fn, ri);
        // E (:383-389)
        if (i < nTx - 1) {
            w.chk(C_MAIN_IM_STATEROOT, rtx[i].newStateRoot, in(m.imStateRoot));
            w.chk(C_MAIN_IM_EXITROOT, rtx[i].newExitRoot, in(m.imExitRoot));
            for (int j = 0; j < Fn; j++) w.chk(C_MAIN_IM_ACCFEE, rtx[i].accFeeOut[j], in(m.imAccFeeOut + j));
        } else {
            // G (:427-431)
            w.chk(C_MAIN_IM_INITFEEROOT, rtx[i].newStateRoot, g.get(lo.g.imInitStateRootFee));
            for (int j = 0; j < Fn; j++) w.chk(C_MAIN_IM_FINALACCFEE, rtx[i].accFeeOut[j], fw.get_unit(lo.fi.imFinalAccFee, fee0 + (uint32_t)j));
        }
        // H, data availability (:443-464)
        const int W2 = 2 * L + 48;
        for (int j = 0; j < W2; j++) {
            F v = dec[i].L1L2TxData[j];
            if (j >= 2 * L && j < 2 * L + 40) {
                v = v * (F(1) - rtx[i].isAmountNullified);
                w.set(lo.rtx.main_l1l2amt + (j - 2 * L), v);
            }
            hin.L1L2TxsData[(size_t)i * W2 + j] = v;
        }
        if (i < maxL1)
            for (int j = 0; j < hzl::L1FULL_BITS; j++) hin.L1TxsFullData[(size_t)i * hzl::L1FULL_BITS + j] = dec[i].L1TxFullData[j];
    }
    // F (:393-417), G (:422-424)
    F feeRoot(0);
    hin.feeTxsData.resize(Fn);
    for (int j = 0; j < Fn; j++) {
        W w{&lo, &c->vals, &c->written, &c->fail, lo.sec_fee, fee0 + (uint32_t)j, b, j};
        const MainFeeInOff& f = lo.fi;
        FeeIn fi;
        fi.oldStateRoot = j == 0 ? g.get(lo.g.imInitStateRootFee) : w.get_unit(f.imStateRootFee, fee0 + (uint32_t)(j - 1));
        fi.feePlanToken = w.get(f.feePlanTokens); fi.feeIdx = w.get(f.feeIdxs); fi.accFee = w.get(f.imFinalAccFee);
        fi.tokenID = w.get(f.tokenID3); fi.nonce = w.get(f.nonce3); fi.sign = w.get(f.sign3); fi.balance = w.get(f.balance3);
        fi.ay = w.get(f.ay3); fi.ethAddr = w.get(f.ethAddr3);
        fi.siblings.resize(L + 1);
        for (int k = 0; k <= L; k++) fi.siblings[k] = w.get(f.siblings3 + k);
        feeRoot = fee_tx(w, lo.fee, L, fi);
        if (j < Fn - 1) w.chk(C_MAIN_IM_FEEROOT, feeRoot, w.get(f.imStateRootFee));
        hin.feeTxsData[j] = fi.feeIdx;
    }
    // H (:435-474)
    hin.oldLastIdx = g.get(lo.g.oldLastIdx); hin.newLastIdx = dec[nTx - 1].outIdx; hin.oldStateRoot = g.get(lo.g.oldStateRoot);
    hin.newStateRoot = feeRoot; hin.newExitRoot = rtx[nTx - 1].newExitRoot;
    hin.globalChainID = g.get(lo.g.globalChainID); hin.currentNumBatch = g.get(lo.g.currentNumBatch);
    W hw{&lo, &c->vals, &c->written, &c->fail, lo.sec_hi, b, b, 0};
    const F h = hash_inputs(hw, lo.hi, L, nTx, maxL1, Fn, hin);
    g.set(lo.g.hashGlobalInputs, h);
}

static void run_instanced(OrcCtx* c) {
    const Layout& lo = c->lo;
    const int L = lo.p.L, Fn = lo.p.F;
    const uint32_t N = lo.sections[0].n_units;
    for (uint32_t u = 0; u < N; u++) {
        W w{&lo, &c->vals, &c->written, &c->fail, 0, u, u, 0};
        auto in = [&](uint32_t off) { return w.get(off); };
        w.set(0, F(1));  // main.one
        switch (lo.p.tmpl) {
            case T_DECODE_FLOAT: case T_COMPUTE_FEE: case T_FEE_ACCUMULATOR: case T_BALANCE_UPDATER: case T_ROLLUP_TX_STATES: case T_RQ_TX_VERIFIER:
            case T_MUX256: case T_BITS2AYSIGN: case T_AYSIGN2AX:
                gadget_main(w, lo);
                break;
            case T_HASH_STATE: {
                const HashStateOff& h = lo.hs;
                hash_state_main(w, h, in(h.tokenID), in(h.nonce), in(h.sign), in(h.balance), in(h.ay), in(h.ethAddr));
                break;
            }
            case T_DECODE_TX: {
                const DecInOff& d = lo.deci;
                DecIn di;
                di.previousOnChain = in(d.previousOnChain); di.txCompressedData = in(d.txCompressedData); di.maxNumBatch = in(d.maxNumBatch);
                di.amountF = in(d.amountF); di.toEthAddr = in(d.toEthAddr); di.toBjjAy = in(d.toBjjAy);
                di.rqTxCompressedDataV2 = in(d.rqTxCompressedDataV2); di.rqToEthAddr = in(d.rqToEthAddr); di.rqToBjjAy = in(d.rqToBjjAy);
                di.fromEthAddr = in(d.fromEthAddr); di.loadAmountF = in(d.loadAmountF); di.globalChainID = in(d.globalChainID);
                di.currentNumBatch = in(d.currentNumBatch); di.onChain = in(d.onChain); di.newAccount = in(d.newAccount);
                di.auxFromIdx = in(d.auxFromIdx); di.auxToIdx = in(d.auxToIdx); di.inIdx = in(d.inIdx);
                di.fromBjjCompressed.resize(256);
                for (int j = 0; j < 256; j++) di.fromBjjCompressed[j] = in(d.fromBjjCompressed + j);
                decode_tx(w, lo.dec, L, di);
                break;
            }
            case T_ROLLUP_TX: {
                const RtxInOff& r = lo.rtxi;
                RtxIn ri;
                ri.feePlanTokens.resize(Fn); ri.accFeeIn.resize(Fn);
                for (int j = 0; j < Fn; j++) { ri.feePlanTokens[j] = in(r.feePlanTokens + j); ri.accFeeIn[j] = in(r.accFeeIn + j); }
                for (int j = 0; j < 3; j++) { ri.futureV2[j] = in(r.futureV2 + j); ri.futureToEthAddr[j] = in(r.futureToEthAddr + j); ri.futureToBjjAy[j] = in(r.futureToBjjAy + j); }
                for (int j = 0; j < 4; j++) { ri.pastV2[j] = in(r.pastV2 + j); ri.pastToEthAddr[j] = in(r.pastToEthAddr + j); ri.pastToBjjAy[j] = in(r.pastToBjjAy + j); }
                ri.fromIdx = in(r.fromIdx); ri.auxFromIdx = in(r.auxFromIdx); ri.toIdx = in(r.toIdx); ri.auxToIdx = in(r.auxToIdx);
                ri.toBjjAy = in(r.toBjjAy); ri.toBjjSign = in(r.toBjjSign); ri.toEthAddr = in(r.toEthAddr); ri.amount = in(r.amount);
                ri.tokenID = in(r.tokenID); ri.nonce = in(r.nonce); ri.userFee = in(r.userFee); ri.rqOffset = in(r.rqOffset);
                ri.onChain = in(r.onChain); ri.newAccount = in(r.newAccount); ri.rqTxCompressedDataV2 = in(r.rqTxCompressedDataV2);
                ri.rqToEthAddr = in(r.rqToEthAddr); ri.rqToBjjAy = in(r.rqToBjjAy); ri.sigL2Hash = in(r.sigL2Hash); ri.s = in(r.s);
                ri.r8x = in(r.r8x); ri.r8y = in(r.r8y); ri.fromEthAddr = in(r.fromEthAddr); ri.loadAmountF = in(r.loadAmountF);
                ri.fromBjjCompressed.resize(256);
                for (int j = 0; j < 256; j++) ri.fromBjjCompressed[j] = in(r.fromBjjCompressed + j);
                ri.tokenID1 = in(r.tokenID1); ri.nonce1 = in(r.nonce1); ri.sign1 = in(r.sign1); ri.balance1 = in(r.balance1); ri.ay1 = in(r.ay1);
                ri.ethAddr1 = in(r.ethAddr1); ri.isOld0_1 = in(r.isOld0_1); ri.oldKey1 = in(r.oldKey1); ri.oldValue1 = in(r.oldValue1);
                ri.tokenID2 = in(r.tokenID2); ri.nonce2 = in(r.nonce2); ri.sign2 = in(r.sign2); ri.balance2 = in(r.balance2);
                ri.newExit = in(r.newExit); ri.ay2 = in(r.ay2); ri.ethAddr2 = in(r.ethAddr2); ri.isOld0_2 = in(r.isOld0_2);
                ri.oldKey2 = in(r.oldKey2); ri.oldValue2 = in(r.oldValue2);
                ri.siblings1.resize(L + 1); ri.siblings2.resize(L + 1);
                for (int j = 0; j <= L; j++) { ri.siblings1[j] = in(r.siblings1 + j); ri.siblings2[j] = in(r.siblings2 + j); }
                ri.oldStateRoot = in(r.oldStateRoot); ri.oldExitRoot = in(r.oldExitRoot);
                const RtxOut ro = rollup_tx(w, lo.rtx, L, Fn, ri);
                w.set(r.o_isAmountNullified, ro.isAmountNullified); w.set(r.o_newStateRoot, ro.newStateRoot); w.set(r.o_newExitRoot, ro.newExitRoot);
                for (int j = 0; j < Fn; j++) w.set(lo.rtx.o_accFeeOut + j, ro.accFeeOut[j]);
                break;
            }
            case T_FEE_TX: {
                const FeeTxInOff& f = lo.feei;
                FeeIn fi;
                fi.oldStateRoot = in(f.oldStateRoot); fi.feePlanToken = in(f.feePlanToken); fi.feeIdx = in(f.feeIdx); fi.accFee = in(f.accFee);
                fi.tokenID = in(f.tokenID); fi.nonce = in(f.nonce); fi.sign = in(f.sign); fi.balance = in(f.balance); fi.ay = in(f.ay);
                fi.ethAddr = in(f.ethAddr);
                fi.siblings.resize(L + 1);
                for (int k = 0; k <= L; k++) fi.siblings[k] = in(f.siblings + k);
                fee_tx(w, lo.fee, L, fi);
                break;
            }
            case T_WITHDRAW: {
                withdraw_main(w, lo.wd, L);
                break;
            }
            case T_SMT_PROCESSOR: {   // circomlib SMTProcessor(nLevels) as main: n = L levels
                const SmtProcInOff& d = lo.smtpi;
                std::vector<F> sib(L);
                for (int k = 0; k < L; k++) sib[k] = in(d.siblings + k);
                static const SmtCids cids{C_SMTP_N2B_OLD, C_SMTP_ALIAS_OLD, C_SMTP_N2B_NEW, C_SMTP_ALIAS_NEW, C_SMTP_LEVINS, C_SMTP_SM_FINAL, C_SMTP_OLDROOT, C_SMTP_KEYS};
                smt_processor(w, lo.smtp, L, in(d.oldRoot), sib.data(), in(d.oldKey), in(d.oldValue), in(d.isOld0), in(d.newKey), in(d.newValue),
                              in(d.fnc), in(d.fnc + 1), cids);
                break;
            }
            case T_SMT_VERIFIER: {    // circomlib SMTVerifier(nLevels) as main
                const SmtVerInOff& d = lo.smtvi;
                std::vector<F> sib(L);
                for (int k = 0; k < L; k++) sib[k] = in(d.siblings + k);
                static const SmtVerCids cids{C_SMTV_N2B_OLD, C_SMTV_ALIAS_OLD, C_SMTV_N2B_NEW, C_SMTV_ALIAS_NEW, C_SMTV_LEVINS, C_SMTV_SM_FINAL, C_SMTV_KEYS, C_SMTV_ROOT};
                smt_verifier(w, lo.smtv, L, in(d.enabled), in(d.root), sib.data(), in(d.oldKey), in(d.oldValue), in(d.isOld0), in(d.key), in(d.value),
                             in(d.fnc), cids);
                break;
            }
        }
    }
}

extern "C" int orc_run(void* h, int32_t* err_inst, int32_t* err_unit, int32_t* err_cid, uint8_t* lhs, uint8_t* rhs) {
    OrcCtx* c = (OrcCtx*)h;
    for (size_t i = 0; i < c->input_set.size(); i++)
        if (!c->input_set[i]) return 4;
    c->fail = Fail();
    c->fail.per_inst.assign(c->lo.n_inst, FailRec());
    c->poseidon_in.clear();
    struct LogScope { LogScope(OrcCtx* c) { g_poseidon_log = c->log_poseidon ? &c->poseidon_in : nullptr; } ~LogScope() { g_poseidon_log = nullptr; } } log_scope(c);
    if (c->lo.p.tmpl == T_ROLLUP_MAIN) {
        for (uint32_t b = 0; b < c->lo.n_inst; b++) run_rollup_main(c, b);
    }
    else if (c->lo.p.tmpl == T_HASH_INPUTS) {
        const Layout& lo = c->lo;
        W w{&lo, &c->vals, &c->written, &c->fail, 0, 0, 0, 0};
        const HashInputsOff& o = lo.hi;
        HashInputsIn in;
        in.oldLastIdx = w.get(o.i_oldLastIdx); in.newLastIdx = w.get(o.i_newLastIdx); in.oldStateRoot = w.get(o.i_oldStateRoot);
        in.newStateRoot = w.get(o.i_newStateRoot); in.newExitRoot = w.get(o.i_newExitRoot); in.globalChainID = w.get(o.i_globalChainID);
        in.currentNumBatch = w.get(o.i_currentNumBatch);
        for (int i = 0; i < lo.p.maxL1 * hzl::L1FULL_BITS; i++) in.L1TxsFullData.push_back(w.get(o.i_L1TxsFullData + i));
        for (int i = 0; i < lo.p.nTx * (2 * lo.p.L + 48); i++) in.L1L2TxsData.push_back(w.get(o.i_L1L2TxsData + i));
        for (int i = 0; i < lo.p.F; i++) in.feeTxsData.push_back(w.get(o.i_feeTxsData + i));
        w.set(o.one, F(1));
        hash_inputs(w, o, lo.p.L, lo.p.nTx, lo.p.maxL1, lo.p.F, in);
    } else run_instanced(c);
    if (c->fail.failed) {
        if (err_inst) *err_inst = c->fail.inst;
        if (err_unit) *err_unit = c->fail.unit;
        if (err_cid) *err_cid = c->fail.cid;
        if (lhs) c->fail.lhs.to_bytes(lhs);
        if (rhs) c->fail.rhs.to_bytes(rhs);
        return 3;
    }
    return 0;
}

// the first failure of one instance in the last run: 0 = none, 3 = filled (the per-instance twin of orc_run's report)
extern "C" int orc_failure_of(void* h, int instance, int32_t* err_unit, int32_t* err_cid, uint8_t* lhs, uint8_t* rhs) {
    OrcCtx* c = (OrcCtx*)h;
    if (instance < 0 || (size_t)instance >= c->fail.per_inst.size()) return 1;
    const FailRec& f = c->fail.per_inst[instance];
    if (!f.failed) return 0;
    if (err_unit) *err_unit = f.unit;
    if (err_cid) *err_cid = f.cid;
    if (lhs) f.lhs.to_bytes(lhs);
    if (rhs) f.rhs.to_bytes(rhs);
    return 3;
}
// test aid for the derived (linear) signals: what each Poseidon component was evaluated on. orc_log_poseidon(h, 1) before orc_run;
// orc_poseidon_inputs(h, instance, virtual index of the component's first stored signal, out, cap) -> number of inputs (0: none there)
extern "C" void orc_log_poseidon(void* h, int on) { ((OrcCtx*)h)->log_poseidon = on != 0; }
extern "C" int orc_poseidon_inputs(void* h, int instance, uint64_t virt_first, uint8_t* out, int cap) {
    OrcCtx* c = (OrcCtx*)h;
    auto it = c->poseidon_in.find(c->lo.virt_to_phys(virt_first, (uint32_t)instance));
    if (it == c->poseidon_in.end()) return 0;
    const int n = (int)it->second.size();
    for (int i = 0; i < n && i < cap; i++) it->second[i].to_bytes(out + 32 * i);
    return n;
}
extern "C" int orc_read(void* h, int instance, uint64_t first, uint64_t count, uint8_t* out) {
    OrcCtx* c = (OrcCtx*)h;
    if (first + count > c->lo.per_instance) return 1;
    for (uint64_t i = 0; i < count; i++) c->vals[c->lo.virt_to_phys(first + i, (uint32_t)instance)].to_bytes(out + 32 * i);
    return 0;
}
extern "C" int orc_read_raw(void* h, uint64_t first, uint64_t count, uint8_t* out) {
    OrcCtx* c = (OrcCtx*)h;
    if (first + count > c->lo.total) return 1;
    for (uint64_t i = 0; i < count; i++) c->vals[first + i].to_bytes(out + 32 * i);
    return 0;
}
// number of stored signals the last run did not write (layout coverage check); name of the first
extern "C" uint64_t orc_unwritten(void* h, char* name_out, size_t cap) {
    OrcCtx* c = (OrcCtx*)h;
    uint64_t n = 0;
    bool have = false;
    c->lo.for_each_symbol([&](const std::string& nm, int sec, uint32_t sig, uint32_t unit) {
        const Section& s = c->lo.sections[sec];
        for (uint32_t u = 0; u < c->lo.n_inst; u++) {
            const uint64_t p = c->lo.phys(sec, sig, u * s.upi + unit);
            if (!c->written[p]) {
                n++;
                if (!have && name_out) { snprintf(name_out, cap, "%s", nm.c_str()); have = true; }
            }
        }
    });
    return n;
}
extern "C" int orc_symbol_lookup(void* h, const char* name, uint64_t* idx) { return ((OrcCtx*)h)->lo.lookup(name, idx) ? 1 : 0; }
extern "C" uint64_t orc_symbol_count(void* h) {
    uint64_t n = 0;
    ((OrcCtx*)h)->lo.for_each_symbol([&](const std::string&, int, uint32_t, uint32_t) { n++; });
    return n;
}
// every stored signal's name, newline separated (Poseidon blocks as one "<component>.sigma*" line); returns the bytes needed
extern "C" uint64_t orc_symbol_names(void* h, char* out, uint64_t cap) {
    uint64_t n = 0;
    ((OrcCtx*)h)->lo.for_each_symbol([&](const std::string& nm, int, uint32_t, uint32_t) {
        if (out && n + nm.size() + 1 <= cap) { memcpy(out + n, nm.data(), nm.size()); out[n + nm.size()] = '\n'; }
        n += nm.size() + 1;
    }, false);
    return n;
}
extern "C" const char* orc_constraint_name(int id) { return constraint_name(id); }
