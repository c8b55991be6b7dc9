// Device functions for the per-transaction templates of the reference (one transaction per lane):
// DecodeTx (src/decode-tx.circom:44-369), RollupTx phases A-C,E,G,H (src/rollup-tx.circom:178-512)
// with RollupTxStates (src/rollup-tx-states.circom), RqTxVerifier (src/rq-tx-verifier.circom),
// BalanceUpdater/ComputeFee/Mux256 (src/balance-updater.circom, src/compute-fee.circom,
// src/lib/mux256.circom) and FeeAccumulator (src/fee-accumulator.circom).
// The hash chains, the SMT processors and the EdDSA verifier run in their own kernels
// (hash/smt/eddsa) so that the independent chains of one transaction occupy separate lanes.
#pragma once
#include "gadgets_dev.h"

namespace hz {

// ---- inter-kernel scratch fields (Montgomery), per transaction ----------------------------------
enum ScratchField {
    // written by the decode step (or copied from inputs for a standalone RollupTx)
    SC_FROMIDX = 0, SC_TOIDX, SC_TOBJJSIGN, SC_AMOUNT, SC_TOKENID, SC_NONCE, SC_USERFEE, SC_SIGL2HASH, SC_OUTIDX,
    // written by the front step
    SC_HS_IN,                       // 16: [old1, old2, new1, new2] x [e0, balance, ay, ethAddr]
    SC_ISP1INSERT = SC_HS_IN + 16, SC_ISP2INSERT, SC_OLDVALUE1, SC_OLDVALUE2,
    SC_KEY_S1OLD, SC_KEY_1, SC_KEY_S2OLD, SC_KEY_2,
    SC_P1_FNC0, SC_P1_FNC1, SC_P2_FNC0, SC_P2_FNC1, SC_ISOLD0_1, SC_ISOLD0_2,
    SC_ISEXIT, SC_OLDSTATEROOT, SC_OLDEXITROOT,
    SC_ED_ENABLED, SC_ED_SIGN, SC_ED_AYSIG, SC_ED_AY, SC_ED_S, SC_ED_R8X, SC_ED_R8Y,
    SC_ED_LEFTX, SC_ED_LEFTY, SC_ED_RIGHTX, SC_ED_RIGHTY,   // S*B8 (k_eddsa_fix) and R8 + h*8A (k_eddsa) for k_eddsa_final
    // k_eddsa_pre -> the two segment lanes of k_eddsa_ladder -> k_eddsa_final: message hash, zero-point flag, 8A (or Base8), 2^147 * 8A
    // (Montgomery form), the two segment outputs
    SC_ED_H, SC_ED_ZP, SC_ED_P0X, SC_ED_P0Y, SC_ED_DBLX, SC_ED_DBLY, SC_ED_S0X, SC_ED_S0Y, SC_ED_S1X, SC_ED_S1Y,
    SC_ED_Q1X, SC_ED_Q1Y,   // k_eddsa_seg: the second segment's start point (DBL* keeps 2^147 * 8A: a lane's padding slot repeats a unit)
    SC_ISAMTNULL,
    SC_FEE2CHARGE, SC_FA_TOKEN,   // RollupMain: the front kernel's hand-off to k_main_feeacc (the FeeAccumulator as a kernel of its own)
    // written by the hash step
    SC_LEAF_P1OLD, SC_LEAF_P1NEW, SC_LEAF_P2OLD, SC_LEAF_P2NEW,
    // written by the smt step: levels[0].oldRoot / newRoot per processor
    SC_ROOT_P1OLD, SC_ROOT_P1NEW, SC_ROOT_P2OLD, SC_ROOT_P2NEW,
    // the same for the early evaluation of the last transaction of a batch (ctx.hip early tail): the main chain may be launched in
    // pieces that keep their running roots in the slots above
    SC_EROOT_P1OLD, SC_EROOT_P1NEW, SC_EROOT_P2OLD, SC_EROOT_P2NEW,
    SC_COUNT
};

__device__ __forceinline__ Fr fr_iszero_bit(const Fr& v) { return fr_from_bit(fr_is_zero(v) ? 1u : 0u); }   // an IsZero's OUTPUT (its `inv` needs the inversion)


// L1TxFullData[160 + 255 - i] = fromBjjCompressed[i] * onChain (src/decode-tx.circom:300-303). The bit is an input signal (checked
// boolean by RollupMain phase A): for 0 and 1 the product is 0 or onChain itself -- a copy, whatever onChain is; anything else takes
// the field product.
__device__ __forceinline__ void l1full_bjj_bit_dev(const UnitIO& io, uint32_t l1full, int i, const Fc& b, const Fr& onChain, const Fc& on_c) {
    uint32_t hi = 0;
#pragma unroll
    for (int k = 1; k < 8; k++) hi |= b.v[k];
    const uint32_t sig = l1full + (160 + 256 - 1 - i);
    if (hi == 0 && b.v[0] <= 1u) io.put_c(sig, b.v[0] ? on_c : fc_zero());
    else io.put_m(sig, fr_mul(fr_from_canon(b), onChain));
}

// DecodeTx's sigL2Hash (src/decode-tx.circom:249-283) from the inputs alone: Poseidon(txCompressedData, toEthAddr[0..159] | amountF << 160 |
// maxNumBatch << 200, toBjjAy, rqTxCompressedDataV2, rqToEthAddr, rqToBjjAy) -- e1 from the same bits the Num2Bits of the template decompose
template <class IN>
__device__ __forceinline__ Fr decode_sig_hash_dev(const UnitIO& io, const DecOff& o, const IN& in, const Fr* K7) {
    Fc e1 = c_extract(io.in_c(in.toEthAddr), 0, 160);
    c_or_bits64(e1, 160, c_bits64(io.in_c(in.amountF), 0, 40));
    c_or_bits64(e1, 200, c_bits64(io.in_c(in.maxNumBatch), 0, 32));
    Fr hin[6];
    hin[0] = io.in_m(in.txCompressedData); hin[1] = fr_from_canon(e1); hin[2] = io.in_m(in.toBjjAy); hin[3] = io.in_m(in.rqTxCompressedDataV2);
    hin[4] = io.in_m(in.rqToEthAddr); hin[5] = io.in_m(in.rqToBjjAy);
    WitSboxSink sink = io.sbox_sink(o.hashSig);
    return poseidon_hash<7>(hin, K7, sink);
}

// DecodeTx (src/decode-tx.circom:44-369). `IN` provides the signal offsets of the inputs inside the lane's section (MainTxInOff or DecInOff
// share the member names used here); `EXT` the four values that come from outside the transaction -- previousOnChain(), inIdx(),
// globalChainID(), currentNumBatch() -- LOADED WHERE THEY ARE USED; K7 = Poseidon t = 7 constant block.
// OUT: the template's output signals are stored (the standalone main). with_bjj = false: the 256 fromBjjCompressed rows of L1TxFullData are
// somebody else's (k_main_front: the mux lane reads those bits anyway). with_hash = false: sigL2Hash is k_main_sighash's.
// Written so that little lives long (rounds 1-5: 2 KB of spills per lane): the fields of txCompressedData are kept as 64-bit integers
// and converted where a field element is needed, every bit decomposition is stored next to the load of its input, a value that is
// only needed for a late check is loaded at the check.
struct DecResult {
    Fr v2, outIdx, sigL2Hash;
};
template <bool OUT, class IN, class EXT>
__device__ __forceinline__ DecResult decode_tx_dev(const UnitIO& io, const DecOff& o, const IN& in, int L, const EXT& ext, const Fr* K7,
                                                   bool with_bjj = true, bool with_hash = true) {
    DecResult r;
    const Fr one = fr_one();
    const Fr onChain = io.in_m(in.onChain);
    const Fr notOn = fr_sub(one, onChain);
    const Fc notOn_c = fr_to_canon(notOn), on_c = fr_to_canon(onChain);
    auto gate_rows = [&](uint32_t sig0, int step, uint64_t field, int n, const Fc& g) __attribute__((always_inline)) {   // rows sig0, sig0 + step, ..: bit i of field times the gate
#pragma unroll 1
        for (int i = 0; i < n; i++) io.put_c((uint32_t)((int)sig0 + step * i), ((field >> i) & 1ull) ? g : fc_zero());
    };
    // ---- txCompressedData: bits, fields, padding
    uint64_t constSig, chainID, fromIdx, toIdx, tokenID, nonce, userFee;
    uint32_t toBjjSign;
    {
        const Fc d = io.in_c(in.txCompressedData);
        num2bits_dev(io, o.n2bData, d, 225, C_DEC_N2B_DATA);
        constSig = c_bits64(d, 0, 32); chainID = c_bits64(d, 32, 16); fromIdx = c_bits64(d, 48, 48); toIdx = c_bits64(d, 96, 48);
        tokenID = c_bits64(d, 144, 32); nonce = c_bits64(d, 176, 40); userFee = c_bits64(d, 216, 8);
        toBjjSign = c_bit(d, 224);
        const uint64_t hi_mask = L < 48 ? ~((1ull << L) - 1ull) & ((1ull << 48) - 1ull) : 0ull;   // the index bits above nLevels must be 0
        const uint32_t pf = (uint32_t)__popcll(fromIdx & hi_mask), pt = (uint32_t)__popcll(toIdx & hi_mask);
        if (pf) report_fail(io.err, io.inst, io.err_unit, C_DEC_PAD_FROM, fr_from_u64(pf), fr_zero());
        if (pt) report_fail(io.err, io.inst, io.err_unit, C_DEC_PAD_TO, fr_from_u64(pt), fr_zero());
    }
    // ---- amountF
    uint64_t amountF;
    {
        const Fc am = io.in_c(in.amountF);
        num2bits_dev(io, o.n2bAmount, am, 40, C_DEC_N2B_AMOUNT);
        amountF = c_bits64(am, 0, 40);
        const Fr amount = decode_float_dev(io, o.dfAmount, amountF);
        if constexpr (OUT) io.put_m(o.o_amount, amount);
    }
    // ---- the other bit decompositions, each next to its load; L1TxFullData's rows (:285-324, every bit times onChain) with them
    uint64_t loadAmountF;
    {
        const Fc te = io.in_c(in.toEthAddr);
        num2bits_dev(io, o.n2bToEthAddr, te, 160, C_DEC_N2B_TOETHADDR);
    }
    {
        const Fc fe = io.in_c(in.fromEthAddr);
        num2bits_dev(io, o.n2bFromEthAddr, fe, 160, C_DEC_N2B_FROMETHADDR);
        gate_rows(o.l1full + 159, -1, c_bits64(fe, 0, 64), 64, on_c);
        gate_rows(o.l1full + 159 - 64, -1, c_bits64(fe, 64, 64), 64, on_c);
        gate_rows(o.l1full + 159 - 128, -1, c_bits64(fe, 128, 32), 32, on_c);
    }
    {
        const Fc la = io.in_c(in.loadAmountF);
        num2bits_dev(io, o.n2bLoadAmountF, la, 40, C_DEC_N2B_LOADAMOUNTF);
        loadAmountF = c_bits64(la, 0, 40);
    }
    if (with_bjj) {
#pragma unroll 1
        for (int i = 0; i < 256; i++) l1full_bjj_bit_dev(io, o.l1full, i, io.in_c(in.fromBjjCompressed + i), onChain, on_c);
    }
    gate_rows(o.l1full + 160 + 256 + 48 - 1, -1, fromIdx, 48, on_c);
    gate_rows(o.l1full + 160 + 256 + 48 + 40 - 1, -1, loadAmountF, 40, on_c);
    gate_rows(o.l1full + 160 + 256 + 48 + 40 + 40 - 1, -1, amountF, 40, on_c);
    gate_rows(o.l1full + 160 + 256 + 48 + 40 + 40 + 32 - 1, -1, tokenID, 32, on_c);
    gate_rows(o.l1full + 160 + 256 + 48 + 40 + 40 + 32 + 48 - 1, -1, toIdx, 48, on_c);
    // ---- txCompressedDataV2 (:174-212): every field bit times (1 - onChain)
    gate_rows(o.v2in + 0, 1, fromIdx, 48, notOn_c); gate_rows(o.v2in + 48, 1, toIdx, 48, notOn_c); gate_rows(o.v2in + 96, 1, amountF, 40, notOn_c);
    gate_rows(o.v2in + 136, 1, tokenID, 32, notOn_c); gate_rows(o.v2in + 168, 1, nonce, 40, notOn_c); gate_rows(o.v2in + 208, 1, userFee, 8, notOn_c);
    gate_rows(o.l1l2Fee + 7, -1, userFee, 8, notOn_c);
    {
        // the plain integer of the 216 gated bits, field by field (constant positions: no word of it is indexed at run time)
        Fc v2bits = fc_zero();
        c_or_bits64(v2bits, 0, fromIdx); c_or_bits64(v2bits, 48, toIdx); c_or_bits64(v2bits, 96, amountF);
        c_or_bits64(v2bits, 136, tokenID); c_or_bits64(v2bits, 168, nonce); c_or_bits64(v2bits, 208, userFee);
        Fr v2 = fr_mul(fr_from_canon(v2bits), notOn);
        if (toBjjSign) v2 = fr_add(v2, m_pow2(216));
        r.v2 = v2;
        if constexpr (OUT) io.put_m(o.o_v2, v2);
    }
    // ---- L1L2TxData (:214-247)
    {
        const Fr sel_s = fr_select(toIdx == 0, notOn, fr_zero());   // (1 - onChain) * toIdxIsZero.out
        const Fr finalTo = mux1_dev(fr_from_u64(toIdx), io.in_m(in.auxToIdx), sel_s);
        io.put_m(o.selToIdx_s, sel_s);
        const Fc finalTo_c = fr_to_canon(finalTo);
        io.put_c(o.selToIdx_out, finalTo_c);
        num2bits_dev(io, o.n2bFinalToIdx, finalTo_c, L, C_DEC_N2B_FINALTOIDX);
        if constexpr (OUT) {
            for (int i = 0; i < L; i++) io.put_bit(o.o_l1l2 + (L - 1 - i), (uint32_t)((fromIdx >> i) & 1ull));
            for (int i = 0; i < L; i++) io.put_bit(o.o_l1l2 + (2 * L - 1 - i), c_bit(finalTo_c, i));
            for (int i = 0; i < 40; i++) io.put_bit(o.o_l1l2 + (2 * L + 40 - 1 - i), (uint32_t)((amountF >> i) & 1ull));
            gate_rows(o.o_l1l2 + 2 * L + 48 - 1, -1, userFee, 8, notOn_c);
        }
    }
    // ---- maxNumBatch: bits, LessThan(32)(currentNumBatch, maxNumBatch + 1) = Num2Bits(33)(in0 + 2^32 - in1), the check (:352-368)
    {
        const Fc maxNumBatch_c = io.in_c(in.maxNumBatch);
        num2bits_dev(io, o.n2bMaxNumBatch, maxNumBatch_c, 32, C_DEC_N2B_MAXNUMBATCH);
        const Fr maxNumBatch = fr_from_canon(maxNumBatch_c);
        const Fr v = fr_sub(fr_add(ext.currentNumBatch(), m_pow2(32)), fr_add(maxNumBatch, one));
        const Fc vc = fr_to_canon(v);
        num2bits_dev(io, o.maxNumBatchLt, vc, 33, C_DEC_N2B_MAXNUMBATCH_LT);
        const Fr ok = fr_from_bit(1u - c_bit(vc, 32));
        io.chk_zero(C_DEC_MAXNUMBATCH, fr_mul(fr_sub(one, ok), fr_sub(one, fr_iszero_bit(maxNumBatch))));
    }
    // ---- the other checks (:326-351), each with the loads it needs
    const Fr newAccount = io.in_m(in.newAccount);
    const Fr onNew = fr_mul(onChain, newAccount);
    r.outIdx = fr_add(ext.inIdx(), onNew);
    io.chk(C_DEC_NEWACCOUNT, fr_select(fromIdx == 0, onChain, fr_zero()), newAccount);   // onChain * fromIdxIsZero.out
    io.put_m(o.outIdx, r.outIdx);
    io.put_m(o.idxChecker_en, onNew);
    io.chk_zero(C_DEC_IDXCHECKER, fr_mul(fr_sub(one, fr_iszero_bit(fr_sub(r.outIdx, io.in_m(in.auxFromIdx)))), onNew));
    io.chk_zero(C_DEC_L1_BEFORE_L2, fr_mul(fr_sub(one, ext.previousOnChain()), onChain));
    io.chk_zero(C_DEC_CHAINID, fr_mul(fr_sub(one, fr_iszero_bit(fr_sub(fr_from_u64(chainID), ext.globalChainID()))), notOn));
    io.chk_zero(C_DEC_CONSTSIG, fr_select(constSig == 3322668559ull, fr_zero(), notOn));
    // ---- the six IsZero of this template (inv, out): one inversion, the prefix products in a rotating register window
    {
        const Fr outIdx = r.outIdx;
        auto operand = [&](int k) __attribute__((always_inline)) -> Fr {
            switch (k) {
                case 0: return fr_from_u64(toIdx);
                case 1: return fr_from_u64(fromIdx);
                case 2: return fr_sub(outIdx, io.in_m(in.auxFromIdx));
                case 3: return fr_sub(fr_from_u64(chainID), ext.globalChainID());
                case 4: return fr_sub(fr_from_u64(3322668559ull), fr_from_u64(constSig));
                default: return io.in_m(in.maxNumBatch);
            }
        };
        auto store = [&](int k, const Fr& v, const Fr& vi) __attribute__((always_inline)) {
            const IsZOff off = k == 0 ? o.toIdxIsZero : k == 1 ? o.fromIdxIsZero : k == 2 ? o.idxChecker : k == 3 ? o.chainIDChecker : k == 4 ? o.constSigChecker : o.maxNumBatchIsZero;
            (void)is_zero_dev(io, off, v, vi);
        };
        (void)is_zero_run_store_dev<6>(6, operand, store);
    }
    // ---- sigL2Hash (:249-283) (with_hash = false: k_main_sighash has it -- a permutation of width 7 keeps 63 registers of state and as many
    // of temporaries)
    r.sigL2Hash = with_hash ? decode_sig_hash_dev(io, o, in, K7) : fr_zero();
    if constexpr (OUT) {
        io.put_m(o.o_fromIdx, fr_from_u64(fromIdx)); io.put_m(o.o_toIdx, fr_from_u64(toIdx)); io.put_m(o.o_tokenID, fr_from_u64(tokenID));
        io.put_m(o.o_nonce, fr_from_u64(nonce)); io.put_m(o.o_userFee, fr_from_u64(userFee)); io.put_bit(o.o_toBjjSign, toBjjSign);
        io.put_m(o.o_sigL2Hash, r.sigL2Hash);
    }
    return r;
}

// Values a RollupTx lane needs that do not sit at a fixed offset of its own section. (The neighbours' fields of RqTxVerifier are loaded
// where they are used, through the kernel's `NB` source: 21 field elements held from the top of the function were 189 registers.)
struct RtxExt {
    Fr fromIdx, toIdx, toBjjSign, amount, tokenID, nonce, userFee, sigL2Hash;   // from DecodeTx (or inputs)
    Fr oldStateRoot, oldExitRoot;
};

// The DecodeTx outputs RollupTx consumes, from the same bits decode_tx_dev reads, without its signals or checks: the front kernel
// evaluates DecodeTx and the RollupTx front logic of one transaction in two lanes (k_main_front).
template <class IN>
__device__ __forceinline__ void decode_fields_dev(const UnitIO& io, const IN& in, RtxExt& x) {
    const Fc d = io.in_c(in.txCompressedData);
    x.fromIdx = fr_from_u64(c_bits64(d, 48, 48)); x.toIdx = fr_from_u64(c_bits64(d, 96, 48)); x.tokenID = fr_from_u64(c_bits64(d, 144, 32));
    x.nonce = fr_from_u64(c_bits64(d, 176, 40)); x.userFee = fr_from_u64(c_bits64(d, 216, 8)); x.toBjjSign = fr_from_bit(c_bit(d, 224));
    const uint64_t f40 = c_bits64(io.in_c(in.amountF), 0, 40);
    Fr pe = fr_from_u64(((f40 >> 35) & 1) ? 10 : 1), p10 = fr_from_u64(10);   // decode_float_dev without its signals
    for (int i = 1; i < 5; i++) {
        p10 = fr_sqr(p10);
        if ((f40 >> (35 + i)) & 1) pe = fr_mul(pe, p10);
    }
    x.amount = fr_mul(fr_from_u64(f40 & ((1ull << 35) - 1)), pe);
    x.sigL2Hash = fr_zero();   // stays with the DecodeTx lane
}

// MultiMux3(1): stores s10,a210,a21,a20,a10,a1,a0,out
__device__ __forceinline__ Fr mux3_dev(const UnitIO& io, const Mux3Off& o, const Fr* c, const Fr* s) {
    const Fr s10 = fr_mul(s[1], s[0]);
    const Fr a210 = fr_mul(fr_sub(fr_add(fr_add(fr_sub(fr_add(fr_sub(fr_sub(c[7], c[6]), c[5]), c[4]), c[3]), c[2]), c[1]), c[0]), s10);
    const Fr a21 = fr_mul(fr_add(fr_sub(fr_sub(c[6], c[4]), c[2]), c[0]), s[1]);
    const Fr a20 = fr_mul(fr_add(fr_sub(fr_sub(c[5], c[4]), c[1]), c[0]), s[0]);
    const Fr a2 = fr_sub(c[4], c[0]);
    const Fr a10 = fr_mul(fr_add(fr_sub(fr_sub(c[3], c[2]), c[1]), c[0]), s10);
    const Fr a1 = fr_mul(fr_sub(c[2], c[0]), s[1]);
    const Fr a0 = fr_mul(fr_sub(c[1], c[0]), s[0]);
    const Fr out = fr_add(fr_mul(fr_add(fr_add(fr_add(a210, a21), a20), a2), s[2]), fr_add(fr_add(fr_add(a10, a1), a0), c[0]));
    io.put_m(o.base + M3_S10, s10); io.put_m(o.base + M3_A210, a210); io.put_m(o.base + M3_A21, a21); io.put_m(o.base + M3_A20, a20);
    io.put_m(o.base + M3_A10, a10); io.put_m(o.base + M3_A1, a1); io.put_m(o.base + M3_A0, a0); io.put_m(o.base + M3_OUT, out);
    return out;
}

// sum_k coef_k * c[k] for the multilinear Mux4 coefficient of selector subset `mask`
// (coefficient of prod_{b in mask} s_b in the interpolation of c over {0,1}^4)
__device__ __forceinline__ Fr mux4_coef(const Fr* c, int mask) {
    Fr acc = fr_zero();
    for (int k = 0; k < 16; k++) {
        if ((k & ~mask) != 0) continue;                  // only sub-masks of `mask`
        const int diff = __popc(mask ^ k);
        acc = (diff & 1) ? fr_sub(acc, c[k]) : fr_add(acc, c[k]);
    }
    return acc;
}

// MultiMux4(1) with signal inputs: stores s10,s20,s21,s210 and the 14 product terms + out (MX4V_* order)
__device__ __forceinline__ Fr mux4_var_dev(const UnitIO& io, uint32_t b, const Fr* c, const Fr* t) {
    const Fr t10 = fr_mul(t[1], t[0]), t20 = fr_mul(t[2], t[0]), t21 = fr_mul(t[2], t[1]), t210 = fr_mul(t21, t[0]);
    const Fr a3210 = fr_mul(mux4_coef(c, 15), t210), a321 = fr_mul(mux4_coef(c, 14), t21), a320 = fr_mul(mux4_coef(c, 13), t20);
    const Fr a310 = fr_mul(mux4_coef(c, 11), t10), a32 = fr_mul(mux4_coef(c, 12), t[2]), a31 = fr_mul(mux4_coef(c, 10), t[1]);
    const Fr a30 = fr_mul(mux4_coef(c, 9), t[0]), a3 = mux4_coef(c, 8);
    const Fr a210 = fr_mul(mux4_coef(c, 7), t210), a21 = fr_mul(mux4_coef(c, 6), t21), a20 = fr_mul(mux4_coef(c, 5), t20);
    const Fr a10 = fr_mul(mux4_coef(c, 3), t10), a2 = fr_mul(mux4_coef(c, 4), t[2]), a1 = fr_mul(mux4_coef(c, 2), t[1]);
    const Fr a0 = fr_mul(mux4_coef(c, 1), t[0]);
    const Fr hi = fr_add(fr_add(fr_add(fr_add(fr_add(fr_add(fr_add(a3210, a321), a320), a310), a32), a31), a30), a3);
    const Fr lo = fr_add(fr_add(fr_add(fr_add(fr_add(fr_add(fr_add(a210, a21), a20), a10), a2), a1), a0), c[0]);
    const Fr out = fr_add(fr_mul(hi, t[3]), lo);
    io.put_m(b + MX4_S10, t10); io.put_m(b + MX4_S20, t20); io.put_m(b + MX4_S21, t21); io.put_m(b + MX4_S210, t210);
    io.put_m(b + MX4V_A3210, a3210); io.put_m(b + MX4V_A321, a321); io.put_m(b + MX4V_A320, a320); io.put_m(b + MX4V_A310, a310);
    io.put_m(b + MX4V_A32, a32); io.put_m(b + MX4V_A31, a31); io.put_m(b + MX4V_A30, a30);
    io.put_m(b + MX4V_A210, a210); io.put_m(b + MX4V_A21, a21); io.put_m(b + MX4V_A20, a20); io.put_m(b + MX4V_A10, a10);
    io.put_m(b + MX4V_A2, a2); io.put_m(b + MX4V_A1, a1); io.put_m(b + MX4V_A0, a0); io.put_m(b + MX4V_OUT, out);
    return out;
}

__device__ __forceinline__ Fr compute_fee_tail_dev(const UnitIO& io, const ComputeFeeOff& o, const Fc& feeSel_c, const Fr& amount, const Fr& factor);
// value 1 in either representative of [0, 2p) (Montgomery form)
__device__ __forceinline__ bool fr_is_one_m(const Fr& a) {
    uint32_t d0 = 0, d1 = 0;
    int32_t c = 0;
#pragma unroll
    for (int i = 0; i < 9; i++) {
        d0 |= a.v[i] ^ fr_r1(i);
        const int32_t x = (int32_t)fr_r1(i) + (int32_t)fr_p29(i) + c;   // R mod p + p, limb by limb
        d1 |= a.v[i] ^ (i < 8 ? (uint32_t)(x & (int32_t)HZ_M29) : (uint32_t)x);
        c = x >> 29;
    }
    return d0 == 0 || d1 == 0;
}
// The 15 product terms of a MultiMux4 whose selectors are bits (s_i = b_i * a with a = 1: t[i] in {0, 1}): a product of selectors is
// 1 exactly when all its bits are set, so every term is either its coefficient or 0 -- additions only (mux4_var_dev multiplies).
// Same signals, same values. `sel` = the four selector bits.
__device__ __forceinline__ Fr mux4_bits_dev(const UnitIO& io, uint32_t b, const Fr* c, uint32_t sel) {
    const Fr zero = fr_zero(), one = fr_one();
    // out = (a3210 + ... + a30 + a3) * s3 + (a210 + ... + a0 + c0); a term = its coefficient times the product of the selectors
    // BELOW bit 3 (a3210 = coef * s2 s1 s0, a32 = coef * s2, ...), so the low three bits decide it
    auto term = [&](int mask) __attribute__((always_inline)) { return ((sel & (mask & 7)) == (uint32_t)(mask & 7)) ? mux4_coef(c, mask) : zero; };
    auto prod = [&](int mask) __attribute__((always_inline)) { return ((sel & mask) == (uint32_t)mask) ? one : zero; };
    const Fr h3210 = term(15), h321 = term(14), h320 = term(13), h310 = term(11), h32 = term(12), h31 = term(10), h30 = term(9);
    const Fr a210 = term(7), a21 = term(6), a20 = term(5), a10 = term(3), a2 = term(4), a1 = term(2), a0 = term(1);
    const Fr a3 = mux4_coef(c, 8);
    const Fr hi = fr_add(fr_add(fr_add(fr_add(fr_add(fr_add(fr_add(h3210, h321), h320), h310), h32), h31), h30), a3);
    const Fr lo = fr_add(fr_add(fr_add(fr_add(fr_add(fr_add(fr_add(a210, a21), a20), a10), a2), a1), a0), c[0]);
    const Fr out = (sel & 8u) ? fr_add(hi, lo) : lo;
    io.put_m(b + MX4_S10, prod(3)); io.put_m(b + MX4_S20, prod(5)); io.put_m(b + MX4_S21, prod(6)); io.put_m(b + MX4_S210, prod(7));
    io.put_m(b + MX4V_A3210, h3210); io.put_m(b + MX4V_A321, h321); io.put_m(b + MX4V_A320, h320); io.put_m(b + MX4V_A310, h310);
    io.put_m(b + MX4V_A32, h32); io.put_m(b + MX4V_A31, h31); io.put_m(b + MX4V_A30, h30);
    io.put_m(b + MX4V_A210, a210); io.put_m(b + MX4V_A21, a21); io.put_m(b + MX4V_A20, a20); io.put_m(b + MX4V_A10, a10);
    io.put_m(b + MX4V_A2, a2); io.put_m(b + MX4V_A1, a1); io.put_m(b + MX4V_A0, a0); io.put_m(b + MX4V_OUT, out);
    return out;
}

// ---- ComputeFee (src/compute-fee.circom:12-94) incl. Mux256 (src/lib/mux256.circom), without an array in sight ------------------------
// Mux256 = 16 MultiMux4 over the constant fee table (selectors s0..s3) feeding one MultiMux4 with signal inputs (s4..s7). A MultiMux4
// is the multilinear interpolation of its 16 inputs: out = sum over masks of coef(mask) * prod_{b in mask} s_b with
// coef(mask) = sum_{k subset of mask} (-1)^{|mask| - |k|} c[k]. Rounds 1-5 kept the 16 inputs (and the first level's 16 outputs) in
// arrays written by rolled loops -- 1.7 KB of scratch memory per lane. Here a coefficient is summed where it is needed from a LOADER
// of the inputs: the fee table's entries are 64-bit integers (their signed sums fit 70 bits: one conversion per coefficient), the first
// level's outputs are signals this lane has just stored (MX4_OUT_C: read back from the witness buffer -- same lane, same address).
typedef __int128 hz_i128;
__device__ __forceinline__ Fr fr_from_i128(hz_i128 x) {   // |x| < 2^127
    const bool neg = x < 0;
    const unsigned __int128 a = neg ? (unsigned __int128)(-x) : (unsigned __int128)x;
    Fc c = fc_zero();
    c.v[0] = (uint32_t)a; c.v[1] = (uint32_t)(a >> 32); c.v[2] = (uint32_t)(a >> 64); c.v[3] = (uint32_t)(a >> 96);
    const Fr m = fr_from_canon(c);
    return neg ? fr_neg(m) : m;
}
// coef(mask) of the table block m as an integer
__device__ __forceinline__ hz_i128 fee_coef_int(int m, int mask) {
    hz_i128 acc = 0;
#pragma unroll 1
    for (int k = 0; k < 16; k++) {
        if ((k & ~mask) != 0) continue;
        const hz_i128 e = (hz_i128)HZ_FEE_TABLE[16 * m + k];
        acc = (__popc(mask ^ k) & 1) ? acc - e : acc + e;
    }
    return acc;
}
// coef(mask) over inputs given by a loader (Montgomery values)
// (a ROLLED loop: unrolled, the sixteen inputs are common subexpressions of the fifteen coefficients and the compiler keeps all of them
//  -- 144 registers -- alive across the whole multiplexer)
template <class LOAD>
__device__ __forceinline__ Fr mux4_coef_ld(LOAD c, int mask) {
    Fr acc = fr_zero();
#pragma unroll 1
    for (int k = 0; k < 16; k++) {
        if ((k & ~mask) != 0) continue;
        const Fr v = c(k);
        acc = (__popc(mask ^ k) & 1) ? fr_sub(acc, v) : fr_add(acc, v);
    }
    return acc;
}
// ... and for inputs that are 64-bit integers (the fee table): the coefficient as a signed integer, one conversion
template <class LOAD64>
__device__ __forceinline__ Fr mux4_coef_int(LOAD64 c, int mask) {
    hz_i128 acc = 0;
#pragma unroll 1
    for (int k = 0; k < 16; k++) {
        if ((k & ~mask) != 0) continue;
        const hz_i128 e = (hz_i128)c(k);
        acc = (__popc(mask ^ k) & 1) ? acc - e : acc + e;
    }
    return fr_from_i128(acc);
}
// MultiMux4(1) with signal inputs from a loader: the body of mux4_var_dev (same signals, same values). `tprod(m)` = the product of the
// selectors in the three-bit mask m (1: t0, 2: t1, 3: t1 t0, 4: t2, .. 7: t2 t1 t0), `t3` the fourth selector -- recomputed per term: this
// is the path of inputs no batch builder produces, and four products held across fifteen terms are 36 registers of a kernel's budget.
template <class LOAD, class TPROD>
__device__ __forceinline__ Fr mux4_var_ld_dev(const UnitIO& io, uint32_t b, LOAD c, TPROD tprod, const Fr& t3) {
    io.put_m(b + MX4_S10, tprod(3)); io.put_m(b + MX4_S20, tprod(5)); io.put_m(b + MX4_S21, tprod(6)); io.put_m(b + MX4_S210, tprod(7));
    Fr hi = fr_zero(), lo = fr_zero();
    auto term = [&](int mask, uint32_t sig, bool high) __attribute__((always_inline)) {   // one product term, stored and added to its half
        const Fr a = fr_mul(mux4_coef_ld(c, mask), tprod(mask & 7));
        io.put_m(b + sig, a);
        if (high) hi = fr_add(hi, a); else lo = fr_add(lo, a);
    };
    term(15, MX4V_A3210, true); term(14, MX4V_A321, true); term(13, MX4V_A320, true); term(11, MX4V_A310, true);
    term(12, MX4V_A32, true); term(10, MX4V_A31, true); term(9, MX4V_A30, true);
    hi = fr_add(hi, mux4_coef_ld(c, 8));   // a3: no product, not stored
    term(7, MX4V_A210, false); term(6, MX4V_A21, false); term(5, MX4V_A20, false); term(3, MX4V_A10, false);
    term(4, MX4V_A2, false); term(2, MX4V_A1, false); term(1, MX4V_A0, false);
    lo = fr_add(lo, c(0));
    const Fr out = fr_add(fr_mul(hi, t3), lo);
    io.put_m(b + MX4V_OUT, out);
    return out;
}
// ... and with selectors that are bits (s_i = b_i * a with a = 1): a product of selectors is 1 exactly when all its bits are set, so every
// term is either its coefficient or 0 -- additions only. Same signals, same values. `sel` = the four selector bits.
template <class COEF, class LOAD>
__device__ __forceinline__ Fr mux4_bits_ld_dev(const UnitIO& io, uint32_t b, COEF coef, LOAD c, uint32_t sel) {
    const Fr zero = fr_zero(), one = fr_one();
    auto prod = [&](int mask) __attribute__((always_inline)) { return ((sel & mask) == (uint32_t)mask) ? one : zero; };
    io.put_m(b + MX4_S10, prod(3)); io.put_m(b + MX4_S20, prod(5)); io.put_m(b + MX4_S21, prod(6)); io.put_m(b + MX4_S210, prod(7));
    Fr hi = zero, lo = zero;
    // a term = its coefficient times the product of the selectors BELOW bit 3 (a3210 = coef * s2 s1 s0, a32 = coef * s2, ...)
    auto term = [&](int mask, uint32_t sig, bool high) __attribute__((always_inline)) {
        const bool on = (sel & (uint32_t)(mask & 7)) == (uint32_t)(mask & 7);
        const Fr a = fr_select(on, coef(mask), zero);
        io.put_m(b + sig, a);
        if (high) hi = fr_add(hi, a); else lo = fr_add(lo, a);
    };
    term(15, MX4V_A3210, true); term(14, MX4V_A321, true); term(13, MX4V_A320, true); term(11, MX4V_A310, true);
    term(12, MX4V_A32, true); term(10, MX4V_A31, true); term(9, MX4V_A30, true);
    hi = fr_add(hi, coef(8));
    term(7, MX4V_A210, false); term(6, MX4V_A21, false); term(5, MX4V_A20, false); term(3, MX4V_A10, false);
    term(4, MX4V_A2, false); term(2, MX4V_A1, false); term(1, MX4V_A0, false);
    lo = fr_add(lo, c(0));
    const Fr out = fr_select((sel & 8u) != 0, fr_add(hi, lo), lo);
    io.put_m(b + MX4V_OUT, out);
    return out;
}

// `amount()` is evaluated where the tail multiplies by it (a loader: nothing of the caller's is kept alive across the multiplexers)
template <class AMT>
__device__ __forceinline__ Fr compute_fee_ld_dev(const UnitIO& io, const ComputeFeeOff& o, const Fc& feeSel_c, AMT amount, const Fr& applyFee) {
    const Fr one = fr_one(), zero = fr_zero();
    io.put_m(o.applyFee, applyFee);
    num2bits_dev(io, o.n2bFeeSel, feeSel_c, 8, C_RTX_FEE_N2B_SEL);
    const uint32_t selbyte = (uint32_t)c_bits64(feeSel_c, 0, 8);
    {
        const Fc applyFee_c = fr_to_canon(applyFee);
#pragma unroll 1
        for (int i = 0; i < 8; i++) io.put_c(o.muxS + i, ((selbyte >> i) & 1u) ? applyFee_c : fc_zero());
    }
    // applyFee is 0 or 1 in every witness RollupTx produces ((1 - onChain) * (1 - nop)); as a main component it is an input and
    // may be anything. When it is a bit on every lane of the wavefront the selectors are bits and the 16 + 1 multiplexers need no
    // field product at all: the selected table entry IS the first level's output.
    const bool ap1 = fr_is_one_m(applyFee);
    if (__all(ap1 || fr_is_zero(applyFee))) {
        const uint32_t selbits = ap1 ? selbyte : 0u;
        const uint32_t lo4 = selbits & 15u;
#pragma unroll 1
        for (int m = 0; m < 16; m++) {
            const uint32_t b = o.mux1 + MX4C_N * m;
            io.put_bit(b + MX4_S10, (lo4 & 3u) == 3u); io.put_bit(b + MX4_S20, (lo4 & 5u) == 5u); io.put_bit(b + MX4_S21, (lo4 & 6u) == 6u); io.put_bit(b + MX4_S210, (lo4 & 7u) == 7u);
            io.put_m(b + MX4_OUT_C, fr_from_u64(HZ_FEE_TABLE[16 * m + lo4]));   // Mux4 with constant inputs and bit selectors: the selected entry
        }
        // the second level's inputs ARE table entries here: its coefficients are integer sums (no field element is loaded back)
        auto tab = [&](int k) __attribute__((always_inline)) { return HZ_FEE_TABLE[16 * k + lo4]; };
        const Fr factor = mux4_bits_ld_dev(io, o.mux2, [&](int mask) __attribute__((always_inline)) { return mux4_coef_int(tab, mask); },
                                           [&](int k) __attribute__((always_inline)) { return fr_from_u64(tab(k)); }, selbits >> 4);
        return compute_fee_tail_dev(io, o, feeSel_c, amount(), factor);
    }
    // The general path (selectors that are field elements): every selector and selector product is recomputed where it is used --
    // slower, and nothing but applyFee stays alive across the seventeen multiplexers.
    auto sel = [&](int bit) __attribute__((always_inline)) { return fr_select(((selbyte >> bit) & 1u) != 0, applyFee, zero); };
    auto sprod = [&](int m3, int base) __attribute__((always_inline)) -> Fr {   // product of the selectors base + {the bits of m3}, m3 = 1..7
        switch (m3) {
            case 1: return sel(base);
            case 2: return sel(base + 1);
            case 3: return fr_mul(sel(base + 1), sel(base));
            case 4: return sel(base + 2);
            case 5: return fr_mul(sel(base + 2), sel(base));
            case 6: return fr_mul(sel(base + 2), sel(base + 1));
            default: return fr_mul(fr_mul(sel(base + 2), sel(base + 1)), sel(base));
        }
    };
#pragma unroll 1
    for (int m = 0; m < 16; m++) {
        // out = (sum over masks with bit 3) * s3 + (sum over masks without): constant inputs, the a-terms are linear in the selector
        // products and are not stored
        Fr hi = fr_from_i128(fee_coef_int(m, 8)), lo = fr_from_i128(fee_coef_int(m, 0));
#pragma unroll 1
        for (int mask = 1; mask < 8; mask++) {
            const Fr sp = sprod(mask, 0);
            lo = fr_add(lo, fr_mul(fr_from_i128(fee_coef_int(m, mask)), sp));
            hi = fr_add(hi, fr_mul(fr_from_i128(fee_coef_int(m, mask | 8)), sp));
        }
        const uint32_t b = o.mux1 + MX4C_N * m;
        io.put_m(b + MX4_S10, sprod(3, 0)); io.put_m(b + MX4_S20, sprod(5, 0)); io.put_m(b + MX4_S21, sprod(6, 0)); io.put_m(b + MX4_S210, sprod(7, 0));
        io.put_m(b + MX4_OUT_C, fr_add(fr_mul(hi, sel(3)), lo));
    }
    // second level: selectors s[4..7], signal inputs (the first level's outputs, read back from the witness buffer -- same lane, same
    // address): every product term is stored
    auto lvl1 = [&](int m) __attribute__((always_inline)) { return io.in_m(o.mux1 + MX4C_N * m + MX4_OUT_C); };
    const Fr factor = mux4_var_ld_dev(io, o.mux2, lvl1, [&](int m3) __attribute__((always_inline)) { return sprod(m3, 4); }, sel(7));
    return compute_fee_tail_dev(io, o, feeSel_c, amount(), factor);
}
__device__ __forceinline__ Fr compute_fee_dev(const UnitIO& io, const ComputeFeeOff& o, const Fc& feeSel_c, const Fr& amount, const Fr& applyFee) {
    return compute_fee_ld_dev(io, o, feeSel_c, [&]() __attribute__((always_inline)) { return amount; }, applyFee);
}
__device__ __forceinline__ Fr compute_fee_tail_dev(const UnitIO& io, const ComputeFeeOff& o, const Fc& feeSel_c, const Fr& amount, const Fr& factor) {
    const Fr notShifted = fr_mul(factor, amount);
    const Fc ns_c = fr_to_canon(notShifted);
    io.put_c(o.feeOutNotShifted, ns_c);
    const uint32_t shiftOff = c_bit(feeSel_c, 6) & c_bit(feeSel_c, 7);
    io.put_bit(o.applyShift, 1u - shiftOff);
#pragma unroll 1
    for (int i = 0; i < 253; i++) io.put_bit(o.bits + i, c_bit(ns_c, i));
    if (!c_fits(ns_c, 253)) report_fail(io.err, io.inst, io.err_unit, C_RTX_FEE_BITS, fr_from_canon(c_extract(ns_c, 0, 253)), notShifted);
    uint32_t ovS = 0, ovN = 0;
    for (int i = 188; i < 253; i++) ovS += c_bit(ns_c, i);
    for (int i = 128; i < 253; i++) ovN += c_bit(ns_c, i);
    if (!shiftOff && ovS) report_fail(io.err, io.inst, io.err_unit, C_RTX_FEE_OVF_SHIFTED, fr_from_u64(ovS), fr_zero());
    if (shiftOff && ovN) report_fail(io.err, io.inst, io.err_unit, C_RTX_FEE_OVF_NOTSHIFTED, fr_from_u64(ovN), fr_zero());
    const Fc feeOut_c = shiftOff ? c_extract(ns_c, 0, 128) : c_extract(ns_c, 60, 128);
    io.put_c(o.feeOut, feeOut_c);
    return fr_from_canon(feeOut_c);
}

// FeeAccumulator (src/fee-accumulator.circom:56-91; RollupTx phase H): IsZero(feePlanTokenID[i] - tokenID) for every fee slot, then the
// selection chain. The inverses come from is_zero_run_dev (gadgets_dev.h): Montgomery's trick over HZ_FA_WINDOW slots at a time in a
// rotating register window, the differences recomputed by the backward pass -- eight slots per inversion instead of sixteen (8
// inversions for 64 slots), no private memory (three arrays of 576 bytes per lane until round 6).
#ifndef HZ_FA_WINDOW
#define HZ_FA_WINDOW 8
#endif
template <class FEE>
__device__ __forceinline__ void fee_accumulator_dev(const UnitIO& io, const RtxOff& o, int Fn, const FEE& feeSrc, const Fr& fee2Charge, const Fr& tokenID) {
    (void)is_zero_run_dev<HZ_FA_WINDOW>(io, Fn, [&](int i) __attribute__((always_inline)) { return fr_sub(feeSrc.plan(i), tokenID); }, [&](int i) __attribute__((always_inline)) { return o.feeAcc + FA_N * i + FA_ISZ_INV; });
    // IsEqual's output and the running "already selected" flag are bits whatever the inputs are: the chain
    // selOut = 1 - (1 - eq)(1 - selIn), s = eq (1 - selIn), out = fee2Charge * s + accIn is logic plus one selection
    bool sel_in = false;
#pragma unroll 1
    for (int i = 0; i < Fn; i++) {
        const uint32_t b = o.feeAcc + FA_N * i;
        const bool eq = fr_is_zero(fr_sub(feeSrc.plan(i), tokenID));
        const bool ms = eq && !sel_in;
        const bool sel_out = eq || sel_in;
        const Fr accIn = feeSrc.acc(i);
        const Fr out = fr_select(ms, fr_add(fee2Charge, accIn), accIn);   // (accIn + fee - accIn)*s + accIn
        io.put_bit(b + FA_SELOUT, sel_out ? 1u : 0u); io.put_bit(b + FA_MUX_S, ms ? 1u : 0u); io.put_m(b + FA_MUX_OUT, out);
        feeSrc.out(io, i, out);
        sel_in = sel_out;
    }
}

struct FrontOut {
    Fr isAmountNullified;
};

// ---- RollupTx front: phases A, B, C, E, G, H and the preparation of D / I (hash-state inputs), J (keys, fnc), F (signature inputs) ----------
// (src/rollup-tx.circom:178-512 with RollupTxStates, RqTxVerifier, BalanceUpdater / ComputeFee, FeeAccumulator.)
// Rounds 1-5 evaluated all of it in one function: two dozen inputs converted at the top and used again at the end, every selector a full
// field element, a dozen IsZero operands and their prefix products in arrays -- 256 registers and 7.7 KB of scratch memory per lane.
// Round 6: THREE lanes per transaction, each holding only what its own signals need:
//   states lane   RollupTxStates' signals and checks, RqTxVerifier, the ForceEqualIfEnabled checks, every IsZero of the front (their
//                 inverses batched in a rotating register window: is_zero_run_dev), hand-off of the processor functions and keys
//   mux lane      BitsCompressed2AySign (the 256 key rows: read once, L1TxFullData rows copied), the 16 Mux1, signature inputs, hand-off
//                 of the hash-state inputs
//   balance lane  BalanceUpdater with ComputeFee / Mux256 (array-free: above), FeeAccumulator hand-off (or the accumulator itself)
// What the lanes share is RollupTxStates' LOGIC: rtx_states_dev computes it as values from a handful of inputs -- an IsZero's OUTPUT
// needs a comparison, only its `inv` signal needs the inversion -- and every lane evaluates the part its signals depend on (the
// compiler drops the rest): some thirty products per lane against the thousand of a ComputeFee, no lane waits for another, no signal
// is written twice, every constraint is checked by exactly one lane.
struct RtxStates {
    Fr onChain, newAccount, notOn;
    Fr loadAmount, isLoadAmount, isAmount;
    Fr isP1Insert, finalFromIdx, finalToIdx, isFinalFromIdx;
    bool ffz, eqEth, eqT1, eqT2;          // IsZero outputs are bits whatever the inputs are: kept as bits (a field element is nine registers)
    Fr key1, effAmt1, isP2Insert, key2;
    Fr verifySignEnabled, checkToEthAddr, checkToBjj, nullifyLoadAmount, nullifyAmount;
};
// DecodeFloatBin's value without its signals (decode_float_dev stores them)
__device__ __forceinline__ Fr decode_float_val(uint64_t f40) {
    Fr pe = fr_from_u64(((f40 >> 35) & 1) ? 10 : 1), p10 = fr_from_u64(10);
    for (int i = 1; i < 5; i++) {
        p10 = fr_sqr(p10);
        if ((f40 >> (35 + i)) & 1) pe = fr_mul(pe, p10);
    }
    return fr_mul(fr_from_u64(f40 & ((1ull << 35) - 1)), pe);
}
// STORE (the states lane): every signal of RollupTxStates is stored, every hand-off field written, WHERE it is computed -- a value that
// is only kept for a store at the end of the function is nine registers held across everything in between (the witness pointer serves
// loads and stores alike: the compiler does not move one across the other).
template <bool STORE, class IN>
__device__ __forceinline__ RtxStates rtx_states_dev(const UnitIO& io, const Scratch& sc, const StatesOff& so, const IN& in, const RtxExt& x) {
    RtxStates f;
    const Fr one = fr_one(), zero = fr_zero();
    auto put = [&](uint32_t sig, const Fr& v) __attribute__((always_inline)) { if constexpr (STORE) io.put_m(sig, v); };
    auto hand = [&](uint32_t field, const Fr& v) __attribute__((always_inline)) { if constexpr (STORE) sc.set(field, v); };
    auto bit = [&](bool b) __attribute__((always_inline)) { return fr_from_bit(b ? 1u : 0u); };
    f.onChain = io.in_m(in.onChain); f.newAccount = io.in_m(in.newAccount);
    f.notOn = fr_sub(one, f.onChain);
    f.loadAmount = decode_float_val(c_bits64(io.in_c(in.loadAmountF), 0, 40));
    f.isLoadAmount = bit(!fr_is_zero(f.loadAmount));
    f.isAmount = bit(!fr_is_zero(x.amount));
    if constexpr (STORE) {
        io.chk_zero(C_RTX_ST_L2_LOADAMOUNT, fr_mul(f.notOn, f.isLoadAmount));
        io.chk_zero(C_RTX_ST_L2_NEWACCOUNT, fr_mul(f.notOn, f.newAccount));
    }
    f.isP1Insert = fr_mul(f.onChain, f.newAccount);                                        // selFromIdx.s
    f.finalFromIdx = mux1_dev(x.fromIdx, io.in_m(in.auxFromIdx), f.isP1Insert);
    put(so.selFromIdx_s, f.isP1Insert); put(so.selFromIdx_out, f.finalFromIdx); put(so.isP1Insert, f.isP1Insert);
    hand(SC_ISP1INSERT, f.isP1Insert);
    const bool tz = fr_is_zero(x.toIdx);
    const Fr selectAuxToIdx = fr_mul(f.notOn, bit(tz));
    f.finalToIdx = mux1_dev(x.toIdx, io.in_m(in.auxToIdx), selectAuxToIdx);
    put(so.selectAuxToIdx, selectAuxToIdx); put(so.selToIdx_out, f.finalToIdx);
    f.ffz = fr_is_zero(f.finalFromIdx);
    f.isFinalFromIdx = bit(!f.ffz);
    const Fr notNop = f.isFinalFromIdx;                                                    // 1 - nop, nop = finalFromIdxIsZero.out
    {
        const Fr P1_fnc0 = fr_mul(f.isP1Insert, f.isFinalFromIdx), P1_fnc1 = fr_mul(fr_sub(one, f.isP1Insert), f.isFinalFromIdx);
        put(so.P1_fnc0, P1_fnc0); put(so.P1_fnc1, P1_fnc1);
        hand(SC_P1_FNC0, P1_fnc0); hand(SC_P1_FNC1, P1_fnc1);
        // Mux2 c = [0,f,f,f], s = [P1_fnc0, P1_fnc1]
        const Fr s10 = fr_mul(P1_fnc1, P1_fnc0);
        const Fr a10 = fr_mul(fr_neg(f.finalFromIdx), s10), a1 = fr_mul(f.finalFromIdx, P1_fnc1), a0 = fr_mul(f.finalFromIdx, P1_fnc0);
        put(so.mux1 + M2_S10, s10); put(so.mux1 + M2_A10, a10); put(so.mux1 + M2_A1, a1); put(so.mux1 + M2_A0, a0);
        f.key1 = fr_add(fr_add(a10, a1), a0);
        hand(SC_KEY_1, f.key1);
    }
    const bool isExit_b = fr_is_zero(fr_sub(f.finalToIdx, one));
    const Fr isExit = bit(isExit_b);
    hand(SC_ISEXIT, isExit);
    f.effAmt1 = fr_mul(x.amount, notNop);                                                  // amount * (1 - nop)
    f.isP2Insert = fr_mul(isExit, io.in_m(in.newExit));
    put(so.isP2Insert, f.isP2Insert);
    hand(SC_ISP2INSERT, f.isP2Insert);
    {
        const Fr P2_fnc0 = fr_mul(f.isP2Insert, f.isFinalFromIdx), P2_fnc1 = fr_mul(fr_sub(one, f.isP2Insert), f.isFinalFromIdx);
        put(so.P2_fnc0, P2_fnc0); put(so.P2_fnc1, P2_fnc1);
        if constexpr (STORE) {   // the processor sees a NOP when nothing is transferred: isP2Nop = 1 - IsZero(effectiveAmount)
            const bool moves = !fr_is_zero(f.effAmt1);
            sc.set(SC_P2_FNC0, fr_select(moves, P2_fnc0, zero)); sc.set(SC_P2_FNC1, fr_select(moves, P2_fnc1, zero));
        }
        // Mux2 c = [0, finalToIdx, 0, finalFromIdx], s = [isAmount, isExit]
        const Fr s10 = fr_mul(isExit, f.isAmount);
        const Fr a10 = fr_mul(fr_sub(f.finalFromIdx, f.finalToIdx), s10), a0 = fr_mul(f.finalToIdx, f.isAmount);
        put(so.mux2 + M2_S10, s10); put(so.mux2 + M2_A10, a10); put(so.mux2 + M2_A1, zero); put(so.mux2 + M2_A0, a0);
        f.key2 = fr_add(a10, a0);
        hand(SC_KEY_2, f.key2);
    }
    f.verifySignEnabled = fr_mul(f.notOn, f.isFinalFromIdx);
    put(so.verifySignEnabled, f.verifySignEnabled);
    hand(SC_ED_ENABLED, f.verifySignEnabled);
    {
        const Fr isAny = bit(fr_is_zero(fr_sub(io.in_m(in.toEthAddr), fr_sub(m_pow2(160), one))));
        const Fr tmpE = fr_mul(fr_sub(one, isAny), selectAuxToIdx), tmpB = fr_mul(isAny, selectAuxToIdx);
        f.checkToEthAddr = fr_mul(tmpE, notNop); f.checkToBjj = fr_mul(tmpB, notNop);
        put(so.tmpCheckToEthAddr, tmpE); put(so.tmpCheckToBjj, tmpB); put(so.checkToEthAddr, f.checkToEthAddr); put(so.checkToBjj, f.checkToBjj);
    }
    const Fr onNotCreate = fr_mul(fr_sub(one, f.newAccount), f.onChain);
    const Fr shouldEth = fr_mul(onNotCreate, f.isAmount);
    put(so.onChainNotCreateAccount, onNotCreate); put(so.shouldCheckEthAddr, shouldEth);
    f.eqEth = fr_is_zero(fr_sub(io.in_m(in.ethAddr1), io.in_m(in.fromEthAddr)));
    const Fr nullEth = fr_mul(shouldEth, bit(!f.eqEth));
    put(so.applyNullifierEthAddr, nullEth);
    f.eqT1 = fr_is_zero(fr_sub(io.in_m(in.tokenID1), x.tokenID));
    const Fr nullT1 = fr_mul(onNotCreate, bit(!f.eqT1));
    put(so.applyNullifierTokenID1, nullT1);
    const Fr sc20 = fr_mul(f.onChain, f.isAmount), sc21 = fr_mul(sc20, fr_sub(one, f.isP2Insert));
    put(so.shouldCheckTokenID2_0, sc20); put(so.shouldCheckTokenID2_1, sc21);
    f.eqT2 = fr_is_zero(fr_sub(io.in_m(in.tokenID2), x.tokenID));
    const Fr nullT2 = fr_mul(sc21, bit(!f.eqT2));
    put(so.applyNullifierTokenID2, nullT2);
    f.nullifyLoadAmount = fr_mul(nullT1, f.isLoadAmount);
    const Fr applyT1Amt = fr_mul(nullT1, f.isAmount);
    put(so.nullifyLoadAmount, f.nullifyLoadAmount); put(so.applyCheckTokenID1ToAmount, applyT1Amt);
    const Fr na0 = fr_sub(one, fr_mul(fr_sub(one, nullEth), fr_sub(one, nullT2)));
    f.nullifyAmount = fr_sub(one, fr_mul(fr_sub(one, na0), fr_sub(one, applyT1Amt)));
    put(so.nullifyAmount_0, na0); put(so.nullifyAmount, f.nullifyAmount);
    return f;
}

// ---- states lane. `NB` gives the neighbours' fields of RqTxVerifier: fut(m, j), past(m, j), m = 0 txCompressedDataV2, 1 toEthAddr, 2 toBjjAy
// `xs()` evaluates the transaction's decoded fields and old roots (RtxExt) where they are needed (rtx_balance_lane_dev does the same)
template <class IN, class NB, class XS>
__device__ __forceinline__ void rtx_states_lane_dev(const UnitIO& io, const Scratch& sc, const RtxOff& o, const IN& in, XS xs, const NB& nb, bool own_sig) {
    const Fr one = fr_one(), zero = fr_zero();
    const StatesOff& so = o.st;
    {   // ---- A: decode loadAmountF (its signals), RollupTxStates
        const Fc la_c = io.in_c(in.loadAmountF);
        num2bits_dev(io, o.n2bLoadAmountF, la_c, 40, C_RTX_N2B_LOADAMOUNTF);
        (void)decode_float_dev(io, o.dfLoadAmount, c_bits64(la_c, 0, 40));
    }
    {   // ---- B: RqTxVerifier
        const Fc rq_c = io.in_c(in.rqOffset);
        num2bits_dev(io, o.rq_n2b, rq_c, 3, C_RTX_RQ_N2B);
        const Fr s[3] = {fr_from_bit(c_bit(rq_c, 0)), fr_from_bit(c_bit(rq_c, 1)), fr_from_bit(c_bit(rq_c, 2))};
        auto one_mux = [&](const int m, uint32_t rq_sig, int cid) __attribute__((always_inline)) {
            const Fr c[8] = {zero, nb.fut(m, 0), nb.fut(m, 1), nb.fut(m, 2), nb.past(m, 3), nb.past(m, 2), nb.past(m, 1), nb.past(m, 0)};
            io.chk(cid, mux3_dev(io, o.rq_mux[m], c, s), io.in_m(rq_sig));
        };
        one_mux(0, in.rqTxCompressedDataV2, C_RTX_RQ_V2);
        one_mux(1, in.rqToEthAddr, C_RTX_RQ_ETHADDR);
        one_mux(2, in.rqToBjjAy, C_RTX_RQ_BJJAY);
    }
    const RtxStates f = rtx_states_dev<true>(io, sc, so, in, xs());   // RollupTxStates: signals, checks, hand-off of keys and functions
    // ---- C: ForceEqualIfEnabled x8 ((1 - isz.out) * enabled === 0): the outputs are the comparisons above, the IsZero signals follow
    const Fr eqNonce = fr_iszero_bit(fr_sub(io.in_m(in.nonce1), xs().nonce));
    const Fr eqToEth = fr_iszero_bit(fr_sub(io.in_m(in.ethAddr2), io.in_m(in.toEthAddr)));
    const Fr eqToAy = fr_iszero_bit(fr_sub(io.in_m(in.toBjjAy), io.in_m(in.ay2)));
    const Fr eqToSign = fr_iszero_bit(fr_sub(xs().toBjjSign, io.in_m(in.sign2)));
    auto force = [&](const Fr& e, const Fr& enabled, int cid) __attribute__((always_inline)) { io.chk_zero(cid, fr_mul(fr_sub(one, e), enabled)); };
    force(eqNonce, f.notOn, C_RTX_NONCE);
    const Fr en_toEth = fr_sub(one, fr_mul(fr_sub(one, f.checkToEthAddr), fr_sub(one, f.checkToBjj)));
    io.put_m(o.checkToEthAddr_en, en_toEth);
    force(eqToEth, en_toEth, C_RTX_TOETHADDR);
    force(eqToAy, f.checkToBjj, C_RTX_TOBJJAY);
    force(eqToSign, f.checkToBjj, C_RTX_TOBJJSIGN);
    force(fr_from_bit(f.eqT1 ? 1u : 0u), f.notOn, C_RTX_TOKENID1);
    const Fr en_t2 = fr_mul(f.notOn, fr_sub(one, f.isP2Insert));
    io.put_m(o.checkTokenID2_en, en_t2);
    force(fr_from_bit(f.eqT2 ? 1u : 0u), en_t2, C_RTX_TOKENID2);
    force(fr_from_bit(f.eqT1 ? 1u : 0u), f.isP1Insert, C_RTX_TOKENID1_L1);
    force(fr_from_bit(f.eqEth ? 1u : 0u), f.isP1Insert, C_RTX_FROMETHADDR);
    // ---- hand-off to the hash / smt / eddsa / back steps: what is not RollupTxStates' (those fields are written where they are computed)
    sc.set(SC_OLDVALUE1, io.in_m(in.oldValue1)); sc.set(SC_OLDVALUE2, io.in_m(in.oldValue2));
    sc.set(SC_ISOLD0_1, io.in_m(in.isOld0_1)); sc.set(SC_ISOLD0_2, io.in_m(in.isOld0_2));
    sc.set(SC_OLDSTATEROOT, xs().oldStateRoot); sc.set(SC_OLDEXITROOT, xs().oldExitRoot);
    sc.set(SC_ED_S, io.in_m(in.s)); sc.set(SC_ED_R8X, io.in_m(in.r8x)); sc.set(SC_ED_R8Y, io.in_m(in.r8y));
    if (own_sig) sc.set(SC_SIGL2HASH, xs().sigL2Hash);   // else: k_main_sighash stores it
    // ---- every IsZero of the front (states, phase C, BalanceUpdater's effectiveAmount): (inv, out) signals, two inversions for 14 slots.
    // Last: an inversion is the lane's peak of register use, and by now nothing but the run's own operands is alive.
    {
        const Fr finalFromIdx = f.finalFromIdx, finalToIdx = f.finalToIdx, loadAmount = f.loadAmount, effAmt1 = f.effAmt1;
        auto operand = [&](int k) __attribute__((always_inline)) -> Fr {
            switch (k) {
                case 0: return xs().toIdx;
                case 1: return fr_sub(io.in_m(in.toEthAddr), fr_sub(m_pow2(160), one));
                case 2: return finalFromIdx;
                case 3: return loadAmount;
                case 4: return xs().amount;
                case 5: return fr_sub(io.in_m(in.ethAddr1), io.in_m(in.fromEthAddr));
                case 6: return fr_sub(io.in_m(in.tokenID1), xs().tokenID);
                case 7: return fr_sub(io.in_m(in.tokenID2), xs().tokenID);
                case 8: return fr_sub(io.in_m(in.nonce1), xs().nonce);
                case 9: return fr_sub(io.in_m(in.ethAddr2), io.in_m(in.toEthAddr));
                case 10: return fr_sub(io.in_m(in.toBjjAy), io.in_m(in.ay2));
                case 11: return fr_sub(xs().toBjjSign, io.in_m(in.sign2));
                case 12: return fr_sub(finalToIdx, one);
                default: return effAmt1;
            }
        };
        auto store = [&](int k, const Fr& v, const Fr& vi) __attribute__((always_inline)) {
            auto put = [&](IsZOff off) __attribute__((always_inline)) { (void)is_zero_dev(io, off, v, vi); };
            switch (k) {
                case 0: put(so.toIdxIsZero); break;
                case 1: put(so.isToEthAddrAny); break;
                case 2: put(so.finalFromIdxIsZero); break;
                case 3: put(so.loadAmountIsZero); break;
                case 4: put(so.amountIsZero); break;
                case 5: put(so.checkFromEthAddr); put(o.fromEthAddrChecker); break;
                case 6: put(so.checkTokenID1); put(o.checkTokenID1); put(o.checkTokenID1L1); break;
                case 7: put(so.checkTokenID2); put(o.checkTokenID2); break;
                case 8: put(o.nonceChecker); break;
                case 9: put(o.checkToEthAddr); break;
                case 10: put(o.toBjjAyChecker); break;
                case 11: put(o.toBjjSignChecker); break;
                case 12: put(so.checkIsExit); break;
                default: put(o.bu.effAmtIsZero); break;
            }
        };
        (void)is_zero_run_store_dev<7>(14, operand, store);
    }
}

// ---- mux lane. `l1full` != ~0u (k_main_front): this lane also stores DecodeTx's L1TxFullData rows of the fromBjjCompressed bits (signal
// offset of L1TxFullData in the section) and makes RollupMain's boolean check of them (`bjj_bool_cid`), from the one read of those inputs.
template <class IN>
__device__ __forceinline__ void rtx_mux_lane_dev(const UnitIO& io, const Scratch& sc, const RtxOff& o, const IN& in, const RtxExt& x, uint32_t l1full, int bjj_bool_cid) {
    const Fr one = fr_one(), zero = fr_zero();
    // ---- E: BitsCompressed2AySign
    Fr bjjAy, bjjSign;
    {
        // fromBjjCompressed are boolean inputs (RollupMain phase A); pack bits 0..253 into an integer, word by word (constant indices)
        uint32_t w0 = 0, w1 = 0, w2 = 0, w3 = 0, w4 = 0, w5 = 0, w6 = 0, w7 = 0;
        bool all_bool = true;
        const Fr onChain = io.in_m(in.onChain);
        const Fc on_c = l1full != ~0u ? fr_to_canon(onChain) : fc_zero();
        const int rows = l1full != ~0u ? 256 : 254;
#pragma unroll 1
        for (int i = 0; i < rows; i++) {
            const Fc b = io.in_c(in.fromBjjCompressed + i);
            uint32_t hi = 0;
#pragma unroll
            for (int k = 1; k < 8; k++) hi |= b.v[k];
            const bool is1 = hi == 0 && b.v[0] == 1u, is0 = hi == 0 && b.v[0] == 0u;
            if (l1full != ~0u) {
                l1full_bjj_bit_dev(io, l1full, i, b, onChain, on_c);
                if (!(is0 || is1) && bjj_bool_cid >= 0) {
                    const Fr v = fr_from_canon(b);
                    io.chk_zero(bjj_bool_cid, fr_mul(v, fr_sub(v, one)));
                }
            }
            if (i >= 254) continue;
            if (!(is0 || is1)) all_bool = false;
            const uint32_t bit = is1 ? (1u << (i & 31)) : 0u;
            const int w = i >> 5;
            w0 |= w == 0 ? bit : 0u; w1 |= w == 1 ? bit : 0u; w2 |= w == 2 ? bit : 0u; w3 |= w == 3 ? bit : 0u;
            w4 |= w == 4 ? bit : 0u; w5 |= w == 5 ? bit : 0u; w6 |= w == 6 ? bit : 0u; w7 |= w == 7 ? bit : 0u;
        }
        Fr acc;
        if (all_bool) {
            // the packed integer may exceed r (2^254 > r); values < 2^254 < 2r: one conditional subtraction makes them canonical
            Fc lo;
            lo.v[0] = w0; lo.v[1] = w1; lo.v[2] = w2; lo.v[3] = w3; lo.v[4] = w4; lo.v[5] = w5; lo.v[6] = w6; lo.v[7] = w7;
            fc_cond_sub_p(lo.v);
            acc = fr_from_canon(lo);
        } else {
            acc = zero;
#pragma unroll 1
            for (int i = 253; i >= 0; i--) acc = fr_add(fr_dbl(acc), io.in_m(in.fromBjjCompressed + i));
        }
        bjjAy = acc;
        bjjSign = io.in_m(in.fromBjjCompressed + 255);
    }
    const Fr p32 = m_pow2(32), p72 = m_pow2(72);
    auto e0 = [&](const Fr& tok, const Fr& non, const Fr& sg) __attribute__((always_inline)) { return fr_add(fr_add(tok, fr_mul(non, p32)), fr_mul(sg, p72)); };
    // hand-off of the OLD states' hash inputs first (nothing of them stays alive), then the multiplexers one account field at a time
    sc.set(SC_HS_IN + 0, e0(io.in_m(in.tokenID1), io.in_m(in.nonce1), io.in_m(in.sign1))); sc.set(SC_HS_IN + 1, io.in_m(in.balance1));
    sc.set(SC_HS_IN + 2, io.in_m(in.ay1)); sc.set(SC_HS_IN + 3, io.in_m(in.ethAddr1));
    sc.set(SC_HS_IN + 4, e0(io.in_m(in.tokenID2), io.in_m(in.nonce2), io.in_m(in.sign2))); sc.set(SC_HS_IN + 5, io.in_m(in.balance2));
    sc.set(SC_HS_IN + 6, io.in_m(in.ay2)); sc.set(SC_HS_IN + 7, io.in_m(in.ethAddr2));
    const RtxStates f = rtx_states_dev<false>(io, sc, o.st, in, x);
    const Fr isP1 = f.isP1Insert, isP2 = f.isP2Insert, notOn = f.notOn, vse = f.verifySignEnabled;
    // the 16 Mux1 (s1OldValue / s2OldValue need the old hashes: hash step), each stored where it is computed
    io.put_m(o.mux16 + MX_S1BALANCE, mux1_dev(io.in_m(in.balance1), zero, isP1));
    io.put_m(o.mux16 + MX_S2BALANCE, mux1_dev(io.in_m(in.balance2), zero, isP2));
    {
        const Fr s1OldKey = mux1_dev(f.key1, io.in_m(in.oldKey1), isP1), s2OldKey = mux1_dev(f.key2, io.in_m(in.oldKey2), isP2);
        io.put_m(o.mux16 + MX_S1OLDKEY, s1OldKey); io.put_m(o.mux16 + MX_S2OLDKEY, s2OldKey);
        sc.set(SC_KEY_S1OLD, s1OldKey); sc.set(SC_KEY_S2OLD, s2OldKey);
    }
    {
        const Fr s1EthAddr = mux1_dev(io.in_m(in.ethAddr1), io.in_m(in.fromEthAddr), isP1), s2EthAddr = mux1_dev(io.in_m(in.ethAddr2), s1EthAddr, isP2);
        io.put_m(o.mux16 + MX_S1ETHADDR, s1EthAddr); io.put_m(o.mux16 + MX_S2ETHADDR, s2EthAddr);
        sc.set(SC_HS_IN + 11, s1EthAddr); sc.set(SC_HS_IN + 15, s2EthAddr);
    }
    const Fr s1Ay = mux1_dev(io.in_m(in.ay1), bjjAy, isP1);
    {
        const Fr s2Ay = mux1_dev(io.in_m(in.ay2), s1Ay, isP2);
        io.put_m(o.mux16 + MX_S1AY, s1Ay); io.put_m(o.mux16 + MX_S2AY, s2Ay);
        sc.set(SC_HS_IN + 10, s1Ay); sc.set(SC_HS_IN + 14, s2Ay); sc.set(SC_ED_AY, s1Ay);
    }
    const Fr s1Sign = mux1_dev(io.in_m(in.sign1), bjjSign, isP1), s2Sign = mux1_dev(io.in_m(in.sign2), s1Sign, isP2);
    io.put_m(o.mux16 + MX_S1SIGN, s1Sign); io.put_m(o.mux16 + MX_S2SIGN, s2Sign);
    {   // ---- F (inputs only): signSignature / aySignature
        const Fr signSig = fr_mul(s1Sign, vse), aySig = fr_mul(s1Ay, vse);
        io.put_m(o.ed.signSignature, signSig); io.put_m(o.ed.aySignature, aySig);
        sc.set(SC_ED_SIGN, signSig); sc.set(SC_ED_AYSIG, aySig);
    }
    const Fr s1TokenID = mux1_dev(io.in_m(in.tokenID1), x.tokenID, isP1), s2TokenID = mux1_dev(io.in_m(in.tokenID2), s1TokenID, isP2);
    io.put_m(o.mux16 + MX_S1TOKENID, s1TokenID); io.put_m(o.mux16 + MX_S2TOKENID, s2TokenID);
    const Fr s1Nonce = mux1_dev(io.in_m(in.nonce1), zero, isP1), s2Nonce = mux1_dev(io.in_m(in.nonce2), zero, isP2);
    io.put_m(o.mux16 + MX_S1NONCE, s1Nonce); io.put_m(o.mux16 + MX_S2NONCE, s2Nonce);
    // the new states' e0 (tokenID + nonce * 2^32 + sign * 2^72); the two new balances are the balance lane's
    sc.set(SC_HS_IN + 8, e0(s1TokenID, fr_add(s1Nonce, notOn), s1Sign));
    sc.set(SC_HS_IN + 12, e0(s2TokenID, s2Nonce, s2Sign));
}

// ---- balance lane: G (BalanceUpdater) and H -- the FeeAccumulator here (standalone RollupTx) or as a kernel of its own beside the chains the
// front kernel feeds (RollupMain: it is half of a transaction's front arithmetic and feeds none of them)
// `xs()` evaluates the transaction's decoded fields (RtxExt): called where they are needed instead of held across ComputeFee
template <class IN, class FEE, bool FEEACC, class XS>
__device__ __forceinline__ FrontOut rtx_balance_lane_dev(const UnitIO& io, const Scratch& sc, const RtxOff& o, const IN& in, XS xs, int Fn, const FEE& feeSrc) {
    const Fr one = fr_one(), zero = fr_zero();
    const BalUpdOff& bo = o.bu;
    Fr fee2Charge;
    {   // ComputeFee needs applyFee = (1 - onChain) * (1 - nop) of the states and nothing else: evaluated for it alone, so that none of the
        // balance arithmetic's operands is alive across the 17 multiplexers
        const RtxExt x0 = xs();
        const RtxStates f0 = rtx_states_dev<false>(io, sc, o.st, in, x0);
        fee2Charge = compute_fee_ld_dev(io, bo.fee, fr_to_canon(x0.userFee), [&]() __attribute__((always_inline)) { return xs().amount; }, fr_mul(f0.notOn, f0.isFinalFromIdx));
    }
    const RtxExt x = xs();
    const RtxStates f = rtx_states_dev<false>(io, sc, o.st, in, x);
    const Fr el1 = fr_mul(f.loadAmount, f.onChain), el2 = fr_mul(el1, fr_sub(one, f.nullifyLoadAmount));
    const Fr ea2 = fr_mul(f.effAmt1, fr_sub(one, f.nullifyAmount));
    io.put_m(bo.effLoad1, el1); io.put_m(bo.effLoad2, el2); io.put_m(bo.effAmt1, f.effAmt1); io.put_m(bo.effAmt2, ea2);
    const Fr s1Balance = mux1_dev(io.in_m(in.balance1), zero, f.isP1Insert), s2Balance = mux1_dev(io.in_m(in.balance2), zero, f.isP2Insert);
    const Fr sb = fr_sub(fr_sub(fr_add(fr_add(m_pow2(192), s1Balance), el2), ea2), fee2Charge);
    const Fc sb_c = fr_to_canon(sb);
    num2bits_dev(io, bo.n2bSender, sb_c, 193, C_RTX_BU_N2B_SENDER);
    const uint32_t ufOk = c_bit(sb_c, 192);
    const Fr underflowOk = fr_from_bit(ufOk);
    io.chk_zero(C_RTX_BU_UNDERFLOW, fr_mul(fr_sub(one, underflowOk), f.notOn));
    const Fr ea3 = fr_select(ufOk != 0, ea2, zero);
    io.put_m(bo.effAmt3, ea3);
    const Fr newSender = fr_sub(fr_sub(fr_add(s1Balance, el2), ea3), fee2Charge);
    const Fr newReceiver = fr_add(s2Balance, ea3);
    const Fr isAmountNullified = fr_sub(one, fr_mul(fr_sub(one, f.nullifyAmount), underflowOk));
    io.put_m(bo.isAmountNullified, isAmountNullified);
    if constexpr (FEEACC) fee_accumulator_dev(io, o, Fn, feeSrc, fee2Charge, x.tokenID);
    else { sc.set(SC_FEE2CHARGE, fee2Charge); sc.set(SC_FA_TOKEN, x.tokenID); }
    sc.set(SC_HS_IN + 9, newSender); sc.set(SC_HS_IN + 13, newReceiver);
    sc.set(SC_ISAMTNULL, isAmountNullified);
    FrontOut r;
    r.isAmountNullified = isAmountNullified;
    return r;
}

}  // namespace hz
