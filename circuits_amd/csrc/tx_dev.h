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

struct DecResult {
    Fr fromIdx, toIdx, toBjjSign, amount, tokenID, nonce, userFee, sigL2Hash, outIdx, v2;
};

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

// DecodeTx. `IN` provides the signal offsets of the inputs inside the lane's section (MainTxInOff
// or DecInOff share the member names used here). K7 = Poseidon t=7 constant block.
// `with_bjj` = false: the 256 fromBjjCompressed rows of L1TxFullData are somebody else's (k_main_front: the RollupTx-front lane reads
// those bits anyway -- l1full_bjj_bit_dev -- so that 8 KB of input per transaction are read once instead of three times).
template <class IN>
__device__ __forceinline__ DecResult decode_tx_dev(const UnitIO& io, const DecOff& o, const IN& in, int L, const Fr& previousOnChain,
                                                   const Fr& inIdx, const Fr& globalChainID, const Fr& currentNumBatch, const Fr* K7,
                                                   bool with_bjj = true) {
    DecResult r;
    const Fr one = fr_one();
    const Fr onChain = io.in_m(in.onChain), newAccount = io.in_m(in.newAccount);
    const Fr notOn = fr_sub(one, onChain);
    const Fc notOn_c = fr_to_canon(notOn), on_c = fr_to_canon(onChain);
    const Fc d = io.in_c(in.txCompressedData);
    num2bits_dev(io, o.n2bData, d, 225, C_DEC_N2B_DATA);
    const uint64_t constSig = c_bits64(d, 0, 32), chainID = c_bits64(d, 32, 16), fromIdx = c_bits64(d, 48, 48), toIdx = c_bits64(d, 96, 48);
    const uint64_t tokenID = c_bits64(d, 144, 32), nonce = c_bits64(d, 176, 40), userFee = c_bits64(d, 216, 8);
    const uint32_t toBjjSign = c_bit(d, 224);
    {
        uint32_t pf = 0, pt = 0;
        for (int i = L; i < 48; i++) { pf += c_bit(d, 48 + i); pt += c_bit(d, 96 + i); }
        if (pf) report_fail(io.err, io.inst, io.err_unit, C_DEC_PAD_FROM, fr_from_u64(pf), fr_zero());
        if (pt) report_fail(io.err, io.inst, io.err_unit, C_DEC_PAD_TO, fr_from_u64(pt), fr_zero());
    }
    r.fromIdx = fr_from_u64(fromIdx); r.toIdx = fr_from_u64(toIdx); r.tokenID = fr_from_u64(tokenID); r.nonce = fr_from_u64(nonce);
    r.userFee = fr_from_u64(userFee); r.toBjjSign = fr_from_bit(toBjjSign);
    const Fc am = io.in_c(in.amountF);
    num2bits_dev(io, o.n2bAmount, am, 40, C_DEC_N2B_AMOUNT);
    const uint64_t amountF = c_bits64(am, 0, 40);
    r.amount = decode_float_dev(io, o.dfAmount, amountF);
    // txCompressedDataV2 (:174-212): every field bit times (1-onChain)
    {
        Fc v2bits = fc_zero();  // plain integer of the 216 gated bits
        int k = 0;
        auto put = [&](uint32_t bit) {
            io.put_c(o.v2in + k, bit ? notOn_c : fc_zero());
            v2bits.v[k >> 5] |= bit << (k & 31);
            k++;
        };
        for (int i = 0; i < 48; i++) put(c_bit(d, 48 + i));
        for (int i = 0; i < 48; i++) put(c_bit(d, 96 + i));
        for (int i = 0; i < 40; i++) put(c_bit(am, i));
        for (int i = 0; i < 32; i++) put(c_bit(d, 144 + i));
        for (int i = 0; i < 40; i++) put(c_bit(d, 176 + i));
        for (int i = 0; i < 8; i++) put(c_bit(d, 216 + i));
        Fr v2 = fr_mul(fr_from_canon(v2bits), notOn);
        if (toBjjSign) v2 = fr_add(v2, m_pow2(216));
        r.v2 = v2;
    }
    // batched inverses of the six IsZero inputs of this template
    const Fr auxFromIdx = io.in_m(in.auxFromIdx), auxToIdx = io.in_m(in.auxToIdx);
    const Fc maxNumBatch_c = io.in_c(in.maxNumBatch);
    const Fr maxNumBatch = fr_from_canon(maxNumBatch_c);
    const Fr onNew = fr_mul(onChain, newAccount);
    r.outIdx = fr_add(inIdx, onNew);
    Fr z[6];
    z[0] = r.toIdx; z[1] = r.fromIdx; z[2] = fr_sub(r.outIdx, auxFromIdx); z[3] = fr_sub(fr_from_u64(chainID), globalChainID);
    z[4] = fr_sub(fr_from_u64(3322668559ull), fr_from_u64(constSig)); z[5] = maxNumBatch;
    Fr zi[6];
    for (int i = 0; i < 6; i++) zi[i] = z[i];
    batch_inv<6>(zi, 6);
    // L1L2TxData (:214-247)
    const Fr tz = is_zero_dev(io, o.toIdxIsZero, z[0], zi[0]);
    const Fr sel_s = fr_mul(notOn, tz);
    const Fr finalTo = mux1_dev(r.toIdx, auxToIdx, sel_s);
    io.put_m(o.selToIdx_s, sel_s);
    const Fc finalTo_c = fr_to_canon(finalTo);
    io.put_c(o.selToIdx_out, finalTo_c);
    num2bits_dev(io, o.n2bFinalToIdx, finalTo_c, L, C_DEC_N2B_FINALTOIDX);
    for (int i = 0; i < 8; i++) io.put_c(o.l1l2Fee + (7 - i), c_bit(d, 216 + i) ? notOn_c : fc_zero());
    // sigL2Hash (:249-283)
    const Fc te = io.in_c(in.toEthAddr);
    num2bits_dev(io, o.n2bToEthAddr, te, 160, C_DEC_N2B_TOETHADDR);
    num2bits_dev(io, o.n2bMaxNumBatch, maxNumBatch_c, 32, C_DEC_N2B_MAXNUMBATCH);
    {
        // e1 = toEthAddr[0..159] | amountF << 160 | maxNumBatch << 200 (from the bit decompositions)
        Fc e1 = c_extract(te, 0, 160);
        for (int i = 0; i < 40; i++) e1.v[(160 + i) >> 5] |= c_bit(am, i) << ((160 + i) & 31);
        for (int i = 0; i < 32; i++) e1.v[(200 + i) >> 5] |= c_bit(maxNumBatch_c, i) << ((200 + i) & 31);
        Fr hin[6];
        hin[0] = fr_from_canon(d); hin[1] = fr_from_canon(e1); hin[2] = io.in_m(in.toBjjAy); hin[3] = io.in_m(in.rqTxCompressedDataV2);
        hin[4] = io.in_m(in.rqToEthAddr); hin[5] = io.in_m(in.rqToBjjAy);
        WitSboxSink sink = io.sbox_sink(o.hashSig);
        r.sigL2Hash = poseidon_hash<7>(hin, K7, sink);
    }
    // L1TxFullData (:285-324): every bit times onChain
    const Fc fe = io.in_c(in.fromEthAddr), la = io.in_c(in.loadAmountF);
    num2bits_dev(io, o.n2bFromEthAddr, fe, 160, C_DEC_N2B_FROMETHADDR);
    num2bits_dev(io, o.n2bLoadAmountF, la, 40, C_DEC_N2B_LOADAMOUNTF);
    {
        auto put = [&](int pos, uint32_t bit) { io.put_c(o.l1full + pos, bit ? on_c : fc_zero()); };
        for (int i = 0; i < 160; i++) put(160 - 1 - i, c_bit(fe, i));
        if (with_bjj) {
            for (int i = 0; i < 256; i++) l1full_bjj_bit_dev(io, o.l1full, i, io.in_c(in.fromBjjCompressed + i), onChain, on_c);
        }
        for (int i = 0; i < 48; i++) put(160 + 256 + 48 - 1 - i, c_bit(d, 48 + i));
        for (int i = 0; i < 40; i++) put(160 + 256 + 48 + 40 - 1 - i, c_bit(la, i));
        for (int i = 0; i < 40; i++) put(160 + 256 + 48 + 40 + 40 - 1 - i, c_bit(am, i));
        for (int i = 0; i < 32; i++) put(160 + 256 + 48 + 40 + 40 + 32 - 1 - i, c_bit(d, 144 + i));
        for (int i = 0; i < 48; i++) put(160 + 256 + 48 + 40 + 40 + 32 + 48 - 1 - i, c_bit(d, 96 + i));
    }
    // checks (:326-368)
    const Fr fz = is_zero_dev(io, o.fromIdxIsZero, z[1], zi[1]);
    io.chk(C_DEC_NEWACCOUNT, fr_mul(onChain, fz), newAccount);
    io.put_m(o.outIdx, r.outIdx);
    io.put_m(o.idxChecker_en, onNew);
    {
        const Fr e = is_zero_dev(io, o.idxChecker, z[2], zi[2]);
        io.chk_zero(C_DEC_IDXCHECKER, fr_mul(fr_sub(one, e), onNew));
    }
    io.chk_zero(C_DEC_L1_BEFORE_L2, fr_mul(fr_sub(one, previousOnChain), onChain));
    {
        const Fr e = is_zero_dev(io, o.chainIDChecker, z[3], zi[3]);
        io.chk_zero(C_DEC_CHAINID, fr_mul(fr_sub(one, e), notOn));
    }
    {
        const Fr e = is_zero_dev(io, o.constSigChecker, z[4], zi[4]);
        io.chk_zero(C_DEC_CONSTSIG, fr_mul(fr_sub(one, e), notOn));
    }
    const Fr mz = is_zero_dev(io, o.maxNumBatchIsZero, z[5], zi[5]);
    {
        // LessThan(32)(currentNumBatch, maxNumBatch + 1): Num2Bits(33)(in0 + 2^32 - in1)
        const Fr v = fr_sub(fr_add(currentNumBatch, m_pow2(32)), fr_add(maxNumBatch, one));
        const Fc vc = fr_to_canon(v);
        num2bits_dev(io, o.maxNumBatchLt, vc, 33, C_DEC_N2B_MAXNUMBATCH_LT);
        const Fr ok = fr_from_bit(1u - c_bit(vc, 32));
        io.chk_zero(C_DEC_MAXNUMBATCH, fr_mul(fr_sub(one, ok), fr_sub(one, mz)));
    }
    if (o.o_fromIdx != ~0u) {
        io.put_m(o.o_fromIdx, r.fromIdx); io.put_m(o.o_toIdx, r.toIdx); io.put_m(o.o_tokenID, r.tokenID); io.put_m(o.o_nonce, r.nonce);
        io.put_m(o.o_userFee, r.userFee); io.put_bit(o.o_toBjjSign, toBjjSign); io.put_m(o.o_amount, r.amount);
        io.put_m(o.o_sigL2Hash, r.sigL2Hash); io.put_m(o.o_v2, r.v2);
        for (int i = 0; i < L; i++) io.put_bit(o.o_l1l2 + (L - 1 - i), c_bit(d, 48 + i));
        for (int i = 0; i < L; i++) io.put_bit(o.o_l1l2 + (2 * L - 1 - i), c_bit(finalTo_c, i));
        for (int i = 0; i < 40; i++) io.put_bit(o.o_l1l2 + (2 * L + 40 - 1 - i), c_bit(am, i));
        for (int i = 0; i < 8; i++) io.put_c(o.o_l1l2 + (2 * L + 48 - 1 - i), c_bit(d, 216 + i) ? notOn_c : fc_zero());
    }
    return r;
}

// Values a RollupTx lane needs that do not sit at a fixed offset of its own section.
struct RtxExt {
    Fr fromIdx, toIdx, toBjjSign, amount, tokenID, nonce, userFee, sigL2Hash;   // from DecodeTx (or inputs)
    Fr oldStateRoot, oldExitRoot;
    Fr futV2[3], pastV2[4], futEth[3], pastEth[4], futAy[3], pastAy[4];
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
    auto term = [&](int mask) { return ((sel & (mask & 7)) == (uint32_t)(mask & 7)) ? mux4_coef(c, mask) : zero; };
    auto prod = [&](int mask) { return ((sel & mask) == (uint32_t)mask) ? one : zero; };
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

// ComputeFee (src/compute-fee.circom:12-94) incl. Mux256 (src/lib/mux256.circom)
__device__ __forceinline__ Fr compute_fee_dev(const UnitIO& io, const ComputeFeeOff& o, const Fc& feeSel_c, const Fr& amount, const Fr& applyFee) {
    const Fr one = fr_one();
    io.put_m(o.applyFee, applyFee);
    num2bits_dev(io, o.n2bFeeSel, feeSel_c, 8, C_RTX_FEE_N2B_SEL);
    Fr s[8];
    const Fc applyFee_c = fr_to_canon(applyFee);
    for (int i = 0; i < 8; i++) {
        s[i] = c_bit(feeSel_c, i) ? applyFee : fr_zero();
        io.put_c(o.muxS + i, c_bit(feeSel_c, i) ? applyFee_c : fc_zero());
    }
    // applyFee is 0 or 1 in every witness RollupTx produces ((1 - onChain) * (1 - nop)); as a main component it is an input and
    // may be anything. When it is a bit on every lane of the wavefront the selectors are bits and the 16 + 1 multiplexers need no
    // field product at all: the selected table entry IS the first level's output (256 + 256 products and as many conversions of
    // table entries otherwise: a quarter of the front kernel's instructions).
    const bool ap1 = fr_is_one_m(applyFee);
    if (__all(ap1 || fr_is_zero(applyFee))) {
        const uint32_t selbits = ap1 ? (uint32_t)c_bits64(feeSel_c, 0, 8) : 0u;
        const uint32_t lo4 = selbits & 15u;
        const Fr s10 = ((lo4 & 3u) == 3u) ? one : fr_zero(), s20 = ((lo4 & 5u) == 5u) ? one : fr_zero(), s21 = ((lo4 & 6u) == 6u) ? one : fr_zero(),
                 s210 = ((lo4 & 7u) == 7u) ? one : fr_zero();
        Fr lvl1[16];
#pragma unroll 1
        for (int m = 0; m < 16; m++) {
            lvl1[m] = fr_from_u64(HZ_FEE_TABLE[16 * m + lo4]);   // Mux4 with constant inputs and bit selectors: the selected entry
            const uint32_t b = o.mux1 + MX4C_N * m;
            io.put_m(b + MX4_S10, s10); io.put_m(b + MX4_S20, s20); io.put_m(b + MX4_S21, s21); io.put_m(b + MX4_S210, s210);
            io.put_m(b + MX4_OUT_C, lvl1[m]);
        }
        const Fr factor = mux4_bits_dev(io, o.mux2, lvl1, selbits >> 4);
        return compute_fee_tail_dev(io, o, feeSel_c, amount, factor);
    }
    const Fr s10 = fr_mul(s[1], s[0]), s20 = fr_mul(s[2], s[0]), s21 = fr_mul(s[2], s[1]), s210 = fr_mul(s21, s[0]);
    const Fr sp[16] = {one, s[0], s[1], s10, s[2], s20, s21, s210, s[3], fr_zero(), fr_zero(), fr_zero(), fr_zero(), fr_zero(), fr_zero(), fr_zero()};
    Fr lvl1[16];
    for (int m = 0; m < 16; m++) {
        Fr c[16];
        for (int k = 0; k < 16; k++) c[k] = fr_from_u64(HZ_FEE_TABLE[16 * m + k]);
        // out = (sum over masks with bit3) * s3 + (sum over masks without bit3); constant inputs:
        // the a-terms are linear in the selector products and are not stored
        Fr hi = fr_zero(), lo = fr_zero();
        for (int mask = 0; mask < 8; mask++) {
            lo = fr_add(lo, fr_mul(mux4_coef(c, mask), sp[mask]));
            hi = fr_add(hi, fr_mul(mux4_coef(c, mask | 8), sp[mask]));
        }
        lvl1[m] = fr_add(fr_mul(hi, s[3]), lo);
        const uint32_t b = o.mux1 + MX4C_N * m;
        io.put_m(b + MX4_S10, s10); io.put_m(b + MX4_S20, s20); io.put_m(b + MX4_S21, s21); io.put_m(b + MX4_S210, s210);
        io.put_m(b + MX4_OUT_C, lvl1[m]);
    }
    // second level: selectors s[4..7], signal inputs: every product term is stored
    const Fr factor = mux4_var_dev(io, o.mux2, lvl1, s + 4);
    return compute_fee_tail_dev(io, o, feeSel_c, amount, factor);
}
__device__ __forceinline__ Fr compute_fee_tail_dev(const UnitIO& io, const ComputeFeeOff& o, const Fc& feeSel_c, const Fr& amount, const Fr& factor) {
    const Fr notShifted = fr_mul(factor, amount);
    const Fc ns_c = fr_to_canon(notShifted);
    io.put_c(o.feeOutNotShifted, ns_c);
    const uint32_t shiftOff = c_bit(feeSel_c, 6) & c_bit(feeSel_c, 7);
    io.put_bit(o.applyShift, 1u - shiftOff);
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

// FeeAccumulator (src/fee-accumulator.circom; RollupTx phase H, src/rollup-tx.circom): batched inverses of tokenID - feePlanTokenID[i]
template <class FEE>
__device__ __forceinline__ void fee_accumulator_dev(const UnitIO& io, const RtxOff& o, int Fn, const FEE& feeSrc, const Fr& fee2Charge, const Fr& tokenID) {
    {
        bool sel_in = false;
        for (int base = 0; base < Fn; base += 16) {
            const int n = (Fn - base) < 16 ? (Fn - base) : 16;
            Fr dz[16], dzi[16];
            for (int i = 0; i < n; i++) { dz[i] = fr_sub(feeSrc.plan(base + i), tokenID); dzi[i] = dz[i]; }
            batch_inv<16>(dzi, n);
            for (int i = 0; i < n; i++) {
                const uint32_t b = o.feeAcc + FA_N * (base + i);
                // IsEqual's output and the running "already selected" flag are bits whatever the inputs are: the chain
                // selOut = 1 - (1 - eq)(1 - selIn), s = eq (1 - selIn), out = fee2Charge * s + accIn is logic plus one selection
                const bool eq = fr_is_zero(dz[i]);
                (void)is_zero_dev(io, b + FA_ISZ_INV, dz[i], dzi[i]);
                const bool ms = eq && !sel_in;
                const bool sel_out = eq || sel_in;
                const Fr accIn = feeSrc.acc(base + i);
                const Fr out = ms ? fr_add(fee2Charge, accIn) : accIn;   // (accIn + fee - accIn)*s + accIn
                io.put_bit(b + FA_SELOUT, sel_out ? 1u : 0u); io.put_bit(b + FA_MUX_S, ms ? 1u : 0u); io.put_m(b + FA_MUX_OUT, out);
                feeSrc.out(io, base + i, out);
                sel_in = sel_out;
            }
        }
    }
}

struct FrontOut {
    Fr isAmountNullified;
};

// RollupTx phases A, B, C, E, G, H and the preparation of D/I (hash-state inputs), J (keys, fnc)
// and F (signature inputs). `IN` gives the offsets of the per-unit inputs (MainTxInOff / RtxInOff).
// accFeeIn / feePlanTokens are read through the pointers (Montgomery conversion on load).
// `l1full` != ~0u (k_main_front): this lane also stores DecodeTx's L1TxFullData rows of the fromBjjCompressed bits (signal offset of
// L1TxFullData in the section) and makes RollupMain's boolean check of them (`bjj_bool_cid`), from the one read of those inputs.
template <class IN, class FEE, bool FEEACC = true>
__device__ __forceinline__ FrontOut rollup_tx_front_dev(const UnitIO& io, const Scratch& sc, const RtxOff& o, const IN& in, const RtxExt& x,
                                                       int Fn, const FEE& feeSrc, bool own_sig = true, uint32_t l1full = ~0u, int bjj_bool_cid = -1) {
    const Fr one = fr_one(), zero = fr_zero();
    const Fr onChain = io.in_m(in.onChain), newAccount = io.in_m(in.newAccount);
    const Fr notOn = fr_sub(one, onChain);
    // ---- A: decode loadAmountF, states
    const Fc la_c = io.in_c(in.loadAmountF);
    num2bits_dev(io, o.n2bLoadAmountF, la_c, 40, C_RTX_N2B_LOADAMOUNTF);
    const Fr loadAmount = decode_float_dev(io, o.dfLoadAmount, c_bits64(la_c, 0, 40));
    const Fr auxFromIdx = io.in_m(in.auxFromIdx), auxToIdx = io.in_m(in.auxToIdx), toEthAddr = io.in_m(in.toEthAddr);
    const Fr fromEthAddr = io.in_m(in.fromEthAddr), ethAddr1 = io.in_m(in.ethAddr1), ethAddr2 = io.in_m(in.ethAddr2);
    const Fr tokenID1 = io.in_m(in.tokenID1), tokenID2 = io.in_m(in.tokenID2), nonce1 = io.in_m(in.nonce1), nonce2 = io.in_m(in.nonce2);
    const Fr sign1 = io.in_m(in.sign1), sign2 = io.in_m(in.sign2), ay1 = io.in_m(in.ay1), ay2 = io.in_m(in.ay2);
    const Fr balance1 = io.in_m(in.balance1), balance2 = io.in_m(in.balance2), toBjjAy = io.in_m(in.toBjjAy), newExit = io.in_m(in.newExit);
    const StatesOff& so = o.st;
    const Fr selFrom_s = fr_mul(onChain, newAccount);
    const Fr finalFromIdx = mux1_dev(x.fromIdx, auxFromIdx, selFrom_s);
    io.put_m(so.selFromIdx_s, selFrom_s); io.put_m(so.selFromIdx_out, finalFromIdx);
    // batch 1: every IsZero input that is available up front (states + phase C)
    enum { Z_TOIDX = 0, Z_ANY, Z_FFROM, Z_LOAD, Z_AMT, Z_FETH, Z_T1, Z_T2, Z_NONCE, Z_TOETH, Z_TOAY, Z_TOSIGN, Z_N };
    Fr z[Z_N], zi[Z_N];
    z[Z_TOIDX] = x.toIdx;
    z[Z_ANY] = fr_sub(toEthAddr, fr_sub(m_pow2(160), one));
    z[Z_FFROM] = finalFromIdx;
    z[Z_LOAD] = loadAmount;
    z[Z_AMT] = x.amount;
    z[Z_FETH] = fr_sub(ethAddr1, fromEthAddr);
    z[Z_T1] = fr_sub(tokenID1, x.tokenID);
    z[Z_T2] = fr_sub(tokenID2, x.tokenID);
    z[Z_NONCE] = fr_sub(nonce1, x.nonce);
    z[Z_TOETH] = fr_sub(ethAddr2, toEthAddr);
    z[Z_TOAY] = fr_sub(toBjjAy, ay2);
    z[Z_TOSIGN] = fr_sub(x.toBjjSign, sign2);
    for (int i = 0; i < Z_N; i++) zi[i] = z[i];
    batch_inv<Z_N>(zi, Z_N);
    const Fr tz = is_zero_dev(io, so.toIdxIsZero, z[Z_TOIDX], zi[Z_TOIDX]);
    const Fr selectAuxToIdx = fr_mul(notOn, tz);
    io.put_m(so.selectAuxToIdx, selectAuxToIdx);
    const Fr finalToIdx = mux1_dev(x.toIdx, auxToIdx, selectAuxToIdx);
    io.put_m(so.selToIdx_out, finalToIdx);
    const Fr isAny = is_zero_dev(io, so.isToEthAddrAny, z[Z_ANY], zi[Z_ANY]);
    const Fr ffz = is_zero_dev(io, so.finalFromIdxIsZero, z[Z_FFROM], zi[Z_FFROM]);
    const Fr isFinalFromIdx = fr_sub(one, ffz);
    const Fr isLoadAmount = fr_sub(one, is_zero_dev(io, so.loadAmountIsZero, z[Z_LOAD], zi[Z_LOAD]));
    const Fr isAmount = fr_sub(one, is_zero_dev(io, so.amountIsZero, z[Z_AMT], zi[Z_AMT]));
    io.chk_zero(C_RTX_ST_L2_LOADAMOUNT, fr_mul(notOn, isLoadAmount));
    io.chk_zero(C_RTX_ST_L2_NEWACCOUNT, fr_mul(notOn, newAccount));
    const Fr isP1Insert = selFrom_s;
    const Fr P1_fnc0 = fr_mul(isP1Insert, isFinalFromIdx), P1_fnc1 = fr_mul(fr_sub(one, isP1Insert), isFinalFromIdx);
    io.put_m(so.isP1Insert, isP1Insert); io.put_m(so.P1_fnc0, P1_fnc0); io.put_m(so.P1_fnc1, P1_fnc1);
    Fr key1;
    {   // Mux2 c = [0,f,f,f], s = [P1_fnc0, P1_fnc1]
        const Fr s10 = fr_mul(P1_fnc1, P1_fnc0);
        const Fr a10 = fr_mul(fr_neg(finalFromIdx), s10), a1 = fr_mul(finalFromIdx, P1_fnc1), a0 = fr_mul(finalFromIdx, P1_fnc0);
        io.put_m(so.mux1 + M2_S10, s10); io.put_m(so.mux1 + M2_A10, a10); io.put_m(so.mux1 + M2_A1, a1); io.put_m(so.mux1 + M2_A0, a0);
        key1 = fr_add(fr_add(a10, a1), a0);
    }
    // batch 2: checkIsExit depends on finalToIdx
    Fr zb[2], zbi[2];
    zb[0] = fr_sub(finalToIdx, one);
    Fr isExit;
    // effectiveAmount1 = amount*(1-nop) needed for BalanceUpdater's IsZero
    const Fr nop = ffz;
    const Fr effAmt1 = fr_mul(x.amount, fr_sub(one, nop));
    zb[1] = effAmt1;
    zbi[0] = zb[0]; zbi[1] = zb[1];
    batch_inv<2>(zbi, 2);
    isExit = is_zero_dev(io, so.checkIsExit, zb[0], zbi[0]);
    const Fr isP2Insert = fr_mul(isExit, newExit);
    const Fr P2_fnc0 = fr_mul(isP2Insert, isFinalFromIdx), P2_fnc1 = fr_mul(fr_sub(one, isP2Insert), isFinalFromIdx);
    io.put_m(so.isP2Insert, isP2Insert); io.put_m(so.P2_fnc0, P2_fnc0); io.put_m(so.P2_fnc1, P2_fnc1);
    Fr key2;
    {   // Mux2 c = [0, finalToIdx, 0, finalFromIdx], s = [isAmount, isExit]
        const Fr s10 = fr_mul(isExit, isAmount);
        const Fr a10 = fr_mul(fr_sub(finalFromIdx, finalToIdx), s10), a1 = zero, a0 = fr_mul(finalToIdx, isAmount);
        io.put_m(so.mux2 + M2_S10, s10); io.put_m(so.mux2 + M2_A10, a10); io.put_m(so.mux2 + M2_A1, a1); io.put_m(so.mux2 + M2_A0, a0);
        key2 = fr_add(fr_add(a10, a1), a0);
    }
    const Fr verifySignEnabled = fr_mul(notOn, isFinalFromIdx);
    io.put_m(so.verifySignEnabled, verifySignEnabled);
    const Fr tmpE = fr_mul(fr_sub(one, isAny), selectAuxToIdx), tmpB = fr_mul(isAny, selectAuxToIdx);
    const Fr checkToEthAddr = fr_mul(tmpE, fr_sub(one, nop)), checkToBjj = fr_mul(tmpB, fr_sub(one, nop));
    io.put_m(so.tmpCheckToEthAddr, tmpE); io.put_m(so.tmpCheckToBjj, tmpB); io.put_m(so.checkToEthAddr, checkToEthAddr); io.put_m(so.checkToBjj, checkToBjj);
    const Fr onNotCreate = fr_mul(fr_sub(one, newAccount), onChain);
    const Fr shouldEth = fr_mul(onNotCreate, isAmount);
    io.put_m(so.onChainNotCreateAccount, onNotCreate); io.put_m(so.shouldCheckEthAddr, shouldEth);
    const Fr eqEth = is_zero_dev(io, so.checkFromEthAddr, z[Z_FETH], zi[Z_FETH]);
    const Fr nullEth = fr_mul(shouldEth, fr_sub(one, eqEth));
    io.put_m(so.applyNullifierEthAddr, nullEth);
    const Fr eqT1 = is_zero_dev(io, so.checkTokenID1, z[Z_T1], zi[Z_T1]);
    const Fr nullT1 = fr_mul(onNotCreate, fr_sub(one, eqT1));
    io.put_m(so.applyNullifierTokenID1, nullT1);
    const Fr sc20 = fr_mul(onChain, isAmount), sc21 = fr_mul(sc20, fr_sub(one, isP2Insert));
    io.put_m(so.shouldCheckTokenID2_0, sc20); io.put_m(so.shouldCheckTokenID2_1, sc21);
    const Fr eqT2 = is_zero_dev(io, so.checkTokenID2, z[Z_T2], zi[Z_T2]);
    const Fr nullT2 = fr_mul(sc21, fr_sub(one, eqT2));
    io.put_m(so.applyNullifierTokenID2, nullT2);
    const Fr nullifyLoadAmount = fr_mul(nullT1, isLoadAmount);
    io.put_m(so.nullifyLoadAmount, nullifyLoadAmount);
    const Fr applyT1Amt = fr_mul(nullT1, isAmount);
    io.put_m(so.applyCheckTokenID1ToAmount, applyT1Amt);
    const Fr na0 = fr_sub(one, fr_mul(fr_sub(one, nullEth), fr_sub(one, nullT2)));
    const Fr nullifyAmount = fr_sub(one, fr_mul(fr_sub(one, na0), fr_sub(one, applyT1Amt)));
    io.put_m(so.nullifyAmount_0, na0); io.put_m(so.nullifyAmount, nullifyAmount);
    // ---- B: RqTxVerifier
    {
        const Fc rq_c = io.in_c(in.rqOffset);
        num2bits_dev(io, o.rq_n2b, rq_c, 3, C_RTX_RQ_N2B);
        const Fr s[3] = {fr_from_bit(c_bit(rq_c, 0)), fr_from_bit(c_bit(rq_c, 1)), fr_from_bit(c_bit(rq_c, 2))};
        const Fr* fut[3] = {x.futV2, x.futEth, x.futAy};
        const Fr* pst[3] = {x.pastV2, x.pastEth, x.pastAy};
        const Fr rq[3] = {io.in_m(in.rqTxCompressedDataV2), io.in_m(in.rqToEthAddr), io.in_m(in.rqToBjjAy)};
        const int cid[3] = {C_RTX_RQ_V2, C_RTX_RQ_ETHADDR, C_RTX_RQ_BJJAY};
        for (int m = 0; m < 3; m++) {
            const Fr c[8] = {zero, fut[m][0], fut[m][1], fut[m][2], pst[m][3], pst[m][2], pst[m][1], pst[m][0]};
            io.chk(cid[m], mux3_dev(io, o.rq_mux[m], c, s), rq[m]);
        }
    }
    // ---- C: ForceEqualIfEnabled x8  ((1 - isz.out) * enabled === 0)
    auto force_eq = [&](IsZOff off, int zidx, const Fr& enabled, int cid) {
        const Fr e = is_zero_dev(io, off, z[zidx], zi[zidx]);
        io.chk_zero(cid, fr_mul(fr_sub(one, e), enabled));
    };
    force_eq(o.nonceChecker, Z_NONCE, notOn, C_RTX_NONCE);
    const Fr en_toEth = fr_sub(one, fr_mul(fr_sub(one, checkToEthAddr), fr_sub(one, checkToBjj)));
    io.put_m(o.checkToEthAddr_en, en_toEth);
    force_eq(o.checkToEthAddr, Z_TOETH, en_toEth, C_RTX_TOETHADDR);
    force_eq(o.toBjjAyChecker, Z_TOAY, checkToBjj, C_RTX_TOBJJAY);
    force_eq(o.toBjjSignChecker, Z_TOSIGN, checkToBjj, C_RTX_TOBJJSIGN);
    force_eq(o.checkTokenID1, Z_T1, notOn, C_RTX_TOKENID1);
    const Fr en_t2 = fr_mul(notOn, fr_sub(one, isP2Insert));
    io.put_m(o.checkTokenID2_en, en_t2);
    force_eq(o.checkTokenID2, Z_T2, en_t2, C_RTX_TOKENID2);
    force_eq(o.checkTokenID1L1, Z_T1, isP1Insert, C_RTX_TOKENID1_L1);
    force_eq(o.fromEthAddrChecker, Z_FETH, isP1Insert, C_RTX_FROMETHADDR);
    // ---- E: BitsCompressed2AySign + 16 Mux1 (s1OldValue / s2OldValue need the old hashes: hash step)
    Fc bjjAy_c = fc_zero();
    Fr bjjSign;
    {
        // fromBjjCompressed are boolean inputs (RollupMain phase A); pack bits 0..253 into an integer
        Fr acc = fr_zero();
        bool all_bool = true;
        const Fc on_c = l1full != ~0u ? fr_to_canon(onChain) : fc_zero();
        for (int i = 0; i < (l1full != ~0u ? 256 : 254); i++) {
            const Fc b = io.in_c(in.fromBjjCompressed + i);
            bool is1 = b.v[0] == 1u, is0 = b.v[0] == 0u;
            for (int k = 1; k < 8; k++) { is1 = is1 && b.v[k] == 0u; is0 = is0 && b.v[k] == 0u; }
            if (l1full != ~0u) {
                l1full_bjj_bit_dev(io, l1full, i, b, onChain, on_c);
                if (!(is0 || is1) && bjj_bool_cid >= 0) {
                    const Fr v = fr_from_canon(b);
                    io.chk_zero(bjj_bool_cid, fr_mul(v, fr_sub(v, one)));
                }
            }
            if (i >= 254) continue;
            if (!(is0 || is1)) all_bool = false;
            if (is1) bjjAy_c.v[i >> 5] |= 1u << (i & 31);
        }
        if (all_bool) {
            // the packed integer may exceed r (2^254 > r): reduce with a field conversion
            Fc lo = bjjAy_c;
            // values < 2^254 < 2r: one conditional subtraction makes them canonical
            fc_cond_sub_p(lo.v);
            acc = fr_from_canon(lo);
        } else {
            for (int i = 253; i >= 0; i--) acc = fr_add(fr_dbl(acc), io.in_m(in.fromBjjCompressed + i));
        }
        bjjAy_c = fr_to_canon(acc);
        bjjSign = io.in_m(in.fromBjjCompressed + 255);
    }
    const Fr bjjAy = fr_from_canon(bjjAy_c);
    Fr mx[MX_N];
    mx[MX_S1BALANCE] = mux1_dev(balance1, zero, isP1Insert);
    mx[MX_S1SIGN] = mux1_dev(sign1, bjjSign, isP1Insert);
    mx[MX_S1AY] = mux1_dev(ay1, bjjAy, isP1Insert);
    mx[MX_S1NONCE] = mux1_dev(nonce1, zero, isP1Insert);
    mx[MX_S1ETHADDR] = mux1_dev(ethAddr1, fromEthAddr, isP1Insert);
    mx[MX_S1TOKENID] = mux1_dev(tokenID1, x.tokenID, isP1Insert);
    mx[MX_S1OLDKEY] = mux1_dev(key1, io.in_m(in.oldKey1), isP1Insert);
    mx[MX_S2BALANCE] = mux1_dev(balance2, zero, isP2Insert);
    mx[MX_S2SIGN] = mux1_dev(sign2, mx[MX_S1SIGN], isP2Insert);
    mx[MX_S2AY] = mux1_dev(ay2, mx[MX_S1AY], isP2Insert);
    mx[MX_S2NONCE] = mux1_dev(nonce2, zero, isP2Insert);
    mx[MX_S2ETHADDR] = mux1_dev(ethAddr2, mx[MX_S1ETHADDR], isP2Insert);
    mx[MX_S2TOKENID] = mux1_dev(tokenID2, mx[MX_S1TOKENID], isP2Insert);
    mx[MX_S2OLDKEY] = mux1_dev(key2, io.in_m(in.oldKey2), isP2Insert);
    for (int i = 0; i < MX_N; i++)
        if (i != MX_S1OLDVALUE && i != MX_S2OLDVALUE) io.put_m(o.mux16 + i, mx[i]);
    // ---- F (inputs only): signSignature / aySignature
    const Fr signSig = fr_mul(mx[MX_S1SIGN], verifySignEnabled), aySig = fr_mul(mx[MX_S1AY], verifySignEnabled);
    io.put_m(o.ed.signSignature, signSig); io.put_m(o.ed.aySignature, aySig);
    // ---- G: BalanceUpdater
    const BalUpdOff& bo = o.bu;
    const Fc userFee_c = fr_to_canon(x.userFee);
    const Fr fee2Charge = compute_fee_dev(io, bo.fee, userFee_c, x.amount, fr_mul(notOn, fr_sub(one, nop)));
    const Fr el1 = fr_mul(loadAmount, onChain), el2 = fr_mul(el1, fr_sub(one, nullifyLoadAmount));
    const Fr ea2 = fr_mul(effAmt1, fr_sub(one, nullifyAmount));
    io.put_m(bo.effLoad1, el1); io.put_m(bo.effLoad2, el2); io.put_m(bo.effAmt1, effAmt1); io.put_m(bo.effAmt2, ea2);
    const Fr sb = fr_sub(fr_sub(fr_add(fr_add(m_pow2(192), mx[MX_S1BALANCE]), el2), ea2), fee2Charge);
    const Fc sb_c = fr_to_canon(sb);
    num2bits_dev(io, bo.n2bSender, sb_c, 193, C_RTX_BU_N2B_SENDER);
    const uint32_t ufOk = c_bit(sb_c, 192);
    const Fr underflowOk = fr_from_bit(ufOk);
    io.chk_zero(C_RTX_BU_UNDERFLOW, fr_mul(fr_sub(one, underflowOk), notOn));
    const Fr ea3 = ufOk ? ea2 : zero;
    io.put_m(bo.effAmt3, ea3);
    const Fr newSender = fr_sub(fr_sub(fr_add(mx[MX_S1BALANCE], el2), ea3), fee2Charge);
    const Fr newReceiver = fr_add(mx[MX_S2BALANCE], ea3);
    const Fr ez = is_zero_dev(io, bo.effAmtIsZero, zb[1], zbi[1]);
    const Fr isAmountNullified = fr_sub(one, fr_mul(fr_sub(one, nullifyAmount), underflowOk));
    io.put_m(bo.isAmountNullified, isAmountNullified);
    const Fr isP2Nop = fr_sub(one, ez);
    // ---- H: FeeAccumulator -- here (standalone RollupTx) or as a kernel of its own beside the chains this kernel feeds (RollupMain):
    // it is half of this function's arithmetic (64 IsZero with four inversions, 128 input conversions) and feeds none of them
    if constexpr (FEEACC) fee_accumulator_dev(io, o, Fn, feeSrc, fee2Charge, x.tokenID);
    else { sc.set(SC_FEE2CHARGE, fee2Charge); sc.set(SC_FA_TOKEN, x.tokenID); }
    // ---- hand-off to the hash / smt / eddsa / back steps
    const Fr p32 = m_pow2(32), p72 = m_pow2(72);
    auto e0 = [&](const Fr& tok, const Fr& non, const Fr& sg) { return fr_add(fr_add(tok, fr_mul(non, p32)), fr_mul(sg, p72)); };
    sc.set(SC_HS_IN + 0, e0(tokenID1, nonce1, sign1)); sc.set(SC_HS_IN + 1, balance1); sc.set(SC_HS_IN + 2, ay1); sc.set(SC_HS_IN + 3, ethAddr1);
    sc.set(SC_HS_IN + 4, e0(tokenID2, nonce2, sign2)); sc.set(SC_HS_IN + 5, balance2); sc.set(SC_HS_IN + 6, ay2); sc.set(SC_HS_IN + 7, ethAddr2);
    sc.set(SC_HS_IN + 8, e0(mx[MX_S1TOKENID], fr_add(mx[MX_S1NONCE], notOn), mx[MX_S1SIGN])); sc.set(SC_HS_IN + 9, newSender);
    sc.set(SC_HS_IN + 10, mx[MX_S1AY]); sc.set(SC_HS_IN + 11, mx[MX_S1ETHADDR]);
    sc.set(SC_HS_IN + 12, e0(mx[MX_S2TOKENID], mx[MX_S2NONCE], mx[MX_S2SIGN])); sc.set(SC_HS_IN + 13, newReceiver);
    sc.set(SC_HS_IN + 14, mx[MX_S2AY]); sc.set(SC_HS_IN + 15, mx[MX_S2ETHADDR]);
    sc.set(SC_ISP1INSERT, isP1Insert); sc.set(SC_ISP2INSERT, isP2Insert);
    sc.set(SC_OLDVALUE1, io.in_m(in.oldValue1)); sc.set(SC_OLDVALUE2, io.in_m(in.oldValue2));
    sc.set(SC_KEY_S1OLD, mx[MX_S1OLDKEY]); sc.set(SC_KEY_1, key1); sc.set(SC_KEY_S2OLD, mx[MX_S2OLDKEY]); sc.set(SC_KEY_2, key2);
    sc.set(SC_P1_FNC0, P1_fnc0); sc.set(SC_P1_FNC1, P1_fnc1);
    sc.set(SC_P2_FNC0, fr_mul(P2_fnc0, isP2Nop)); sc.set(SC_P2_FNC1, fr_mul(P2_fnc1, isP2Nop));
    sc.set(SC_ISOLD0_1, io.in_m(in.isOld0_1)); sc.set(SC_ISOLD0_2, io.in_m(in.isOld0_2));
    sc.set(SC_ISEXIT, isExit); sc.set(SC_OLDSTATEROOT, x.oldStateRoot); sc.set(SC_OLDEXITROOT, x.oldExitRoot);
    sc.set(SC_ED_ENABLED, verifySignEnabled); sc.set(SC_ED_SIGN, signSig); sc.set(SC_ED_AYSIG, aySig); sc.set(SC_ED_AY, mx[MX_S1AY]);
    sc.set(SC_ED_S, io.in_m(in.s)); sc.set(SC_ED_R8X, io.in_m(in.r8x)); sc.set(SC_ED_R8Y, io.in_m(in.r8y));
    if (own_sig) sc.set(SC_SIGL2HASH, x.sigL2Hash);   // else: the DecodeTx lane stores it
    sc.set(SC_ISAMTNULL, isAmountNullified);
    FrontOut r;
    r.isAmountNullified = isAmountNullified;
    return r;
}

}  // namespace hz
