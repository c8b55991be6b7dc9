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
#pragma unroll
    for (int k = 0; k < 16; k++) {
        if ((k & ~mask) != 0) continue;
        const hz_i128 e = (hz_i128)HZ_FEE_TABLE[16 * m + k];
        acc = (__popc(mask ^ k) & 1) ? acc - e : acc + e;
    }
    return acc;
}
// coef(mask) over inputs given by a loader (Montgomery values)
template <class LOAD>
__device__ __forceinline__ Fr mux4_coef_ld(LOAD c, int mask) {
    Fr acc = fr_zero();
#pragma unroll
    for (int k = 0; k < 16; k++) {
        if ((k & ~mask) != 0) continue;
        acc = (__popc(mask ^ k) & 1) ? fr_sub(acc, c(k)) : fr_add(acc, c(k));
    }
    return acc;
}
// MultiMux4(1) with signal inputs from a loader: the body of mux4_var_dev (same signals, same values)
template <class LOAD>
__device__ __forceinline__ Fr mux4_var_ld_dev(const UnitIO& io, uint32_t b, LOAD c, const Fr& t0, const Fr& t1, const Fr& t2, const Fr& t3) {
    const Fr t10 = fr_mul(t1, t0), t20 = fr_mul(t2, t0), t21 = fr_mul(t2, t1), t210 = fr_mul(t21, t0);
    io.put_m(b + MX4_S10, t10); io.put_m(b + MX4_S20, t20); io.put_m(b + MX4_S21, t21); io.put_m(b + MX4_S210, t210);
    Fr hi = fr_zero(), lo = fr_zero();
    auto term = [&](int mask, const Fr& sel, uint32_t sig, bool high) {   // one product term, stored and added to its half
        const Fr a = fr_mul(mux4_coef_ld(c, mask), sel);
        io.put_m(b + sig, a);
        if (high) hi = fr_add(hi, a); else lo = fr_add(lo, a);
    };
    term(15, t210, MX4V_A3210, true); term(14, t21, MX4V_A321, true); term(13, t20, MX4V_A320, true); term(11, t10, MX4V_A310, true);
    term(12, t2, MX4V_A32, true); term(10, t1, MX4V_A31, true); term(9, t0, MX4V_A30, true);
    hi = fr_add(hi, mux4_coef_ld(c, 8));   // a3: no product, not stored
    term(7, t210, MX4V_A210, false); term(6, t21, MX4V_A21, false); term(5, t20, MX4V_A20, false); term(3, t10, MX4V_A10, false);
    term(4, t2, MX4V_A2, false); term(2, t1, MX4V_A1, false); term(1, t0, MX4V_A0, false);
    lo = fr_add(lo, c(0));
    const Fr out = fr_add(fr_mul(hi, t3), lo);
    io.put_m(b + MX4V_OUT, out);
    return out;
}
// ... and with selectors that are bits (s_i = b_i * a with a = 1): a product of selectors is 1 exactly when all its bits are set, so every
// term is either its coefficient or 0 -- additions only. Same signals, same values. `sel` = the four selector bits.
template <class LOAD>
__device__ __forceinline__ Fr mux4_bits_ld_dev(const UnitIO& io, uint32_t b, LOAD c, uint32_t sel) {
    const Fr zero = fr_zero(), one = fr_one();
    auto prod = [&](int mask) { return ((sel & mask) == (uint32_t)mask) ? one : zero; };
    io.put_m(b + MX4_S10, prod(3)); io.put_m(b + MX4_S20, prod(5)); io.put_m(b + MX4_S21, prod(6)); io.put_m(b + MX4_S210, prod(7));
    Fr hi = zero, lo = zero;
    // a term = its coefficient times the product of the selectors BELOW bit 3 (a3210 = coef * s2 s1 s0, a32 = coef * s2, ...)
    auto term = [&](int mask, uint32_t sig, bool high) {
        const bool on = (sel & (uint32_t)(mask & 7)) == (uint32_t)(mask & 7);
        const Fr a = fr_select(on, mux4_coef_ld(c, mask), zero);
        io.put_m(b + sig, a);
        if (high) hi = fr_add(hi, a); else lo = fr_add(lo, a);
    };
    term(15, MX4V_A3210, true); term(14, MX4V_A321, true); term(13, MX4V_A320, true); term(11, MX4V_A310, true);
    term(12, MX4V_A32, true); term(10, MX4V_A31, true); term(9, MX4V_A30, true);
    hi = fr_add(hi, mux4_coef_ld(c, 8));
    term(7, MX4V_A210, false); term(6, MX4V_A21, false); term(5, MX4V_A20, false); term(3, MX4V_A10, false);
    term(4, MX4V_A2, false); term(2, MX4V_A1, false); term(1, MX4V_A0, false);
    lo = fr_add(lo, c(0));
    const Fr out = fr_select((sel & 8u) != 0, fr_add(hi, lo), lo);
    io.put_m(b + MX4V_OUT, out);
    return out;
}

__device__ __forceinline__ Fr compute_fee_dev(const UnitIO& io, const ComputeFeeOff& o, const Fc& feeSel_c, const Fr& amount, const Fr& applyFee) {
    const Fr one = fr_one(), zero = fr_zero();
    io.put_m(o.applyFee, applyFee);
    num2bits_dev(io, o.n2bFeeSel, feeSel_c, 8, C_RTX_FEE_N2B_SEL);
    const uint32_t selbyte = (uint32_t)c_bits64(feeSel_c, 0, 8);
    const Fc applyFee_c = fr_to_canon(applyFee);
#pragma unroll
    for (int i = 0; i < 8; i++) io.put_c(o.muxS + i, ((selbyte >> i) & 1u) ? applyFee_c : fc_zero());
    // the first level's outputs, read back where the second level needs them
    auto lvl1 = [&](int m) { return io.in_m(o.mux1 + MX4C_N * m + MX4_OUT_C); };
    // applyFee is 0 or 1 in every witness RollupTx produces ((1 - onChain) * (1 - nop)); as a main component it is an input and
    // may be anything. When it is a bit on every lane of the wavefront the selectors are bits and the 16 + 1 multiplexers need no
    // field product at all: the selected table entry IS the first level's output.
    const bool ap1 = fr_is_one_m(applyFee);
    if (__all(ap1 || fr_is_zero(applyFee))) {
        const uint32_t selbits = ap1 ? selbyte : 0u;
        const uint32_t lo4 = selbits & 15u;
        const Fr s10 = ((lo4 & 3u) == 3u) ? one : zero, s20 = ((lo4 & 5u) == 5u) ? one : zero, s21 = ((lo4 & 6u) == 6u) ? one : zero,
                 s210 = ((lo4 & 7u) == 7u) ? one : zero;
#pragma unroll 1
        for (int m = 0; m < 16; m++) {
            const uint32_t b = o.mux1 + MX4C_N * m;
            io.put_m(b + MX4_S10, s10); io.put_m(b + MX4_S20, s20); io.put_m(b + MX4_S21, s21); io.put_m(b + MX4_S210, s210);
            io.put_m(b + MX4_OUT_C, fr_from_u64(HZ_FEE_TABLE[16 * m + lo4]));   // Mux4 with constant inputs and bit selectors: the selected entry
        }
        const Fr factor = mux4_bits_ld_dev(io, o.mux2, lvl1, selbits >> 4);
        return compute_fee_tail_dev(io, o, feeSel_c, amount, factor);
    }
    const Fr s0 = fr_select((selbyte & 1u) != 0, applyFee, zero), s1 = fr_select((selbyte & 2u) != 0, applyFee, zero),
             s2 = fr_select((selbyte & 4u) != 0, applyFee, zero), s3 = fr_select((selbyte & 8u) != 0, applyFee, zero);
    const Fr s10 = fr_mul(s1, s0), s20 = fr_mul(s2, s0), s21 = fr_mul(s2, s1), s210 = fr_mul(s21, s0);
#pragma unroll 1
    for (int m = 0; m < 16; m++) {
        // out = (sum over masks with bit 3) * s3 + (sum over masks without): constant inputs, the a-terms are linear in the selector
        // products and are not stored
        Fr hi = fr_from_i128(fee_coef_int(m, 8)), lo = fr_from_i128(fee_coef_int(m, 0));
        auto both = [&](int mask, const Fr& sp) {
            lo = fr_add(lo, fr_mul(fr_from_i128(fee_coef_int(m, mask)), sp));
            hi = fr_add(hi, fr_mul(fr_from_i128(fee_coef_int(m, mask | 8)), sp));
        };
        both(1, s0); both(2, s1); both(3, s10); both(4, s2); both(5, s20); both(6, s21); both(7, s210);
        const uint32_t b = o.mux1 + MX4C_N * m;
        io.put_m(b + MX4_S10, s10); io.put_m(b + MX4_S20, s20); io.put_m(b + MX4_S21, s21); io.put_m(b + MX4_S210, s210);
        io.put_m(b + MX4_OUT_C, fr_add(fr_mul(hi, s3), lo));
    }
    // second level: selectors s[4..7], signal inputs: every product term is stored
    const Fr t0 = fr_select((selbyte & 16u) != 0, applyFee, zero), t1 = fr_select((selbyte & 32u) != 0, applyFee, zero),
             t2 = fr_select((selbyte & 64u) != 0, applyFee, zero), t3 = fr_select((selbyte & 128u) != 0, applyFee, zero);
    const Fr factor = mux4_var_ld_dev(io, o.mux2, lvl1, t0, t1, t2, t3);
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

// FeeAccumulator (src/fee-accumulator.circom:56-91; RollupTx phase H): IsZero(feePlanTokenID[i] - tokenID) for every fee slot, then the
// selection chain. The inverses come from is_zero_run_dev (gadgets_dev.h): Montgomery's trick over HZ_FA_WINDOW slots at a time in a
// rotating register window, the differences recomputed by the backward pass -- eight slots per inversion instead of sixteen (8
// inversions for 64 slots), no private memory (three arrays of 576 bytes per lane until round 6).
#ifndef HZ_FA_WINDOW
#define HZ_FA_WINDOW 8
#endif
template <class FEE>
__device__ __forceinline__ void fee_accumulator_dev(const UnitIO& io, const RtxOff& o, int Fn, const FEE& feeSrc, const Fr& fee2Charge, const Fr& tokenID) {
    (void)is_zero_run_dev<HZ_FA_WINDOW>(io, Fn, [&](int i) { return fr_sub(feeSrc.plan(i), tokenID); }, [&](int i) { return o.feeAcc + FA_N * i + FA_ISZ_INV; });
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
    Fr onChain, newAccount, notOn, newExit;
    Fr loadAmount, isLoadAmount, isAmount;
    Fr isP1Insert, finalFromIdx, tz, selectAuxToIdx, finalToIdx, isAny, ffz, isFinalFromIdx;
    Fr P1_fnc0, P1_fnc1, m1_s10, m1_a10, m1_a1, m1_a0, key1;
    Fr isExit, effAmt1, isP2Insert, P2_fnc0, P2_fnc1, m2_s10, m2_a10, m2_a0, key2;
    Fr verifySignEnabled, tmpE, tmpB, checkToEthAddr, checkToBjj, onNotCreate, shouldEth, eqEth, nullEth, eqT1, nullT1, sc20, sc21, eqT2, nullT2;
    Fr nullifyLoadAmount, applyT1Amt, na0, nullifyAmount;
};
__device__ __forceinline__ Fr fr_iszero_bit(const Fr& v) { return fr_from_bit(fr_is_zero(v) ? 1u : 0u); }
// DecodeFloatBin's value without its signals (decode_float_dev stores them)
__device__ __forceinline__ Fr decode_float_val(uint64_t f40) {
    Fr pe = fr_from_u64(((f40 >> 35) & 1) ? 10 : 1), p10 = fr_from_u64(10);
    for (int i = 1; i < 5; i++) {
        p10 = fr_sqr(p10);
        if ((f40 >> (35 + i)) & 1) pe = fr_mul(pe, p10);
    }
    return fr_mul(fr_from_u64(f40 & ((1ull << 35) - 1)), pe);
}
template <class IN>
__device__ __forceinline__ RtxStates rtx_states_dev(const UnitIO& io, const IN& in, const RtxExt& x) {
    RtxStates f;
    const Fr one = fr_one();
    f.onChain = io.in_m(in.onChain); f.newAccount = io.in_m(in.newAccount); f.newExit = io.in_m(in.newExit);
    f.notOn = fr_sub(one, f.onChain);
    f.loadAmount = decode_float_val(c_bits64(io.in_c(in.loadAmountF), 0, 40));
    f.isLoadAmount = fr_sub(one, fr_iszero_bit(f.loadAmount));
    f.isAmount = fr_sub(one, fr_iszero_bit(x.amount));
    f.isP1Insert = fr_mul(f.onChain, f.newAccount);                                        // selFromIdx.s
    f.finalFromIdx = mux1_dev(x.fromIdx, io.in_m(in.auxFromIdx), f.isP1Insert);
    f.tz = fr_iszero_bit(x.toIdx);
    f.selectAuxToIdx = fr_mul(f.notOn, f.tz);
    f.finalToIdx = mux1_dev(x.toIdx, io.in_m(in.auxToIdx), f.selectAuxToIdx);
    const Fr toEthAddr = io.in_m(in.toEthAddr);
    f.isAny = fr_iszero_bit(fr_sub(toEthAddr, fr_sub(m_pow2(160), one)));
    f.ffz = fr_iszero_bit(f.finalFromIdx);
    f.isFinalFromIdx = fr_sub(one, f.ffz);
    f.P1_fnc0 = fr_mul(f.isP1Insert, f.isFinalFromIdx); f.P1_fnc1 = fr_mul(fr_sub(one, f.isP1Insert), f.isFinalFromIdx);
    // Mux2 c = [0,f,f,f], s = [P1_fnc0, P1_fnc1]
    f.m1_s10 = fr_mul(f.P1_fnc1, f.P1_fnc0);
    f.m1_a10 = fr_mul(fr_neg(f.finalFromIdx), f.m1_s10); f.m1_a1 = fr_mul(f.finalFromIdx, f.P1_fnc1); f.m1_a0 = fr_mul(f.finalFromIdx, f.P1_fnc0);
    f.key1 = fr_add(fr_add(f.m1_a10, f.m1_a1), f.m1_a0);
    f.isExit = fr_iszero_bit(fr_sub(f.finalToIdx, one));
    f.effAmt1 = fr_mul(x.amount, fr_sub(one, f.ffz));                                      // amount * (1 - nop)
    f.isP2Insert = fr_mul(f.isExit, f.newExit);
    f.P2_fnc0 = fr_mul(f.isP2Insert, f.isFinalFromIdx); f.P2_fnc1 = fr_mul(fr_sub(one, f.isP2Insert), f.isFinalFromIdx);
    // Mux2 c = [0, finalToIdx, 0, finalFromIdx], s = [isAmount, isExit]
    f.m2_s10 = fr_mul(f.isExit, f.isAmount);
    f.m2_a10 = fr_mul(fr_sub(f.finalFromIdx, f.finalToIdx), f.m2_s10); f.m2_a0 = fr_mul(f.finalToIdx, f.isAmount);
    f.key2 = fr_add(f.m2_a10, f.m2_a0);
    f.verifySignEnabled = fr_mul(f.notOn, f.isFinalFromIdx);
    f.tmpE = fr_mul(fr_sub(one, f.isAny), f.selectAuxToIdx); f.tmpB = fr_mul(f.isAny, f.selectAuxToIdx);
    f.checkToEthAddr = fr_mul(f.tmpE, fr_sub(one, f.ffz)); f.checkToBjj = fr_mul(f.tmpB, fr_sub(one, f.ffz));
    f.onNotCreate = fr_mul(fr_sub(one, f.newAccount), f.onChain);
    f.shouldEth = fr_mul(f.onNotCreate, f.isAmount);
    f.eqEth = fr_iszero_bit(fr_sub(io.in_m(in.ethAddr1), io.in_m(in.fromEthAddr)));
    f.nullEth = fr_mul(f.shouldEth, fr_sub(one, f.eqEth));
    f.eqT1 = fr_iszero_bit(fr_sub(io.in_m(in.tokenID1), x.tokenID));
    f.nullT1 = fr_mul(f.onNotCreate, fr_sub(one, f.eqT1));
    f.sc20 = fr_mul(f.onChain, f.isAmount); f.sc21 = fr_mul(f.sc20, fr_sub(one, f.isP2Insert));
    f.eqT2 = fr_iszero_bit(fr_sub(io.in_m(in.tokenID2), x.tokenID));
    f.nullT2 = fr_mul(f.sc21, fr_sub(one, f.eqT2));
    f.nullifyLoadAmount = fr_mul(f.nullT1, f.isLoadAmount);
    f.applyT1Amt = fr_mul(f.nullT1, f.isAmount);
    f.na0 = fr_sub(one, fr_mul(fr_sub(one, f.nullEth), fr_sub(one, f.nullT2)));
    f.nullifyAmount = fr_sub(one, fr_mul(fr_sub(one, f.na0), fr_sub(one, f.applyT1Amt)));
    return f;
}

// ---- states lane. `NB` gives the neighbours' fields of RqTxVerifier: fut(m, j), past(m, j), m = 0 txCompressedDataV2, 1 toEthAddr, 2 toBjjAy
template <class IN, class NB>
__device__ __forceinline__ void rtx_states_lane_dev(const UnitIO& io, const Scratch& sc, const RtxOff& o, const IN& in, const RtxExt& x, const NB& nb, bool own_sig) {
    const Fr one = fr_one(), zero = fr_zero();
    const StatesOff& so = o.st;
    {   // ---- A: decode loadAmountF (its signals), RollupTxStates
        const Fc la_c = io.in_c(in.loadAmountF);
        num2bits_dev(io, o.n2bLoadAmountF, la_c, 40, C_RTX_N2B_LOADAMOUNTF);
        (void)decode_float_dev(io, o.dfLoadAmount, c_bits64(la_c, 0, 40));
    }
    const RtxStates f = rtx_states_dev(io, in, x);
    io.put_m(so.selFromIdx_s, f.isP1Insert); io.put_m(so.selFromIdx_out, f.finalFromIdx);
    io.put_m(so.selectAuxToIdx, f.selectAuxToIdx); io.put_m(so.selToIdx_out, f.finalToIdx);
    io.chk_zero(C_RTX_ST_L2_LOADAMOUNT, fr_mul(f.notOn, f.isLoadAmount));
    io.chk_zero(C_RTX_ST_L2_NEWACCOUNT, fr_mul(f.notOn, f.newAccount));
    io.put_m(so.isP1Insert, f.isP1Insert); io.put_m(so.P1_fnc0, f.P1_fnc0); io.put_m(so.P1_fnc1, f.P1_fnc1);
    io.put_m(so.mux1 + M2_S10, f.m1_s10); io.put_m(so.mux1 + M2_A10, f.m1_a10); io.put_m(so.mux1 + M2_A1, f.m1_a1); io.put_m(so.mux1 + M2_A0, f.m1_a0);
    io.put_m(so.isP2Insert, f.isP2Insert); io.put_m(so.P2_fnc0, f.P2_fnc0); io.put_m(so.P2_fnc1, f.P2_fnc1);
    io.put_m(so.mux2 + M2_S10, f.m2_s10); io.put_m(so.mux2 + M2_A10, f.m2_a10); io.put_m(so.mux2 + M2_A1, zero); io.put_m(so.mux2 + M2_A0, f.m2_a0);
    io.put_m(so.verifySignEnabled, f.verifySignEnabled);
    io.put_m(so.tmpCheckToEthAddr, f.tmpE); io.put_m(so.tmpCheckToBjj, f.tmpB); io.put_m(so.checkToEthAddr, f.checkToEthAddr); io.put_m(so.checkToBjj, f.checkToBjj);
    io.put_m(so.onChainNotCreateAccount, f.onNotCreate); io.put_m(so.shouldCheckEthAddr, f.shouldEth);
    io.put_m(so.applyNullifierEthAddr, f.nullEth); io.put_m(so.applyNullifierTokenID1, f.nullT1);
    io.put_m(so.shouldCheckTokenID2_0, f.sc20); io.put_m(so.shouldCheckTokenID2_1, f.sc21);
    io.put_m(so.applyNullifierTokenID2, f.nullT2);
    io.put_m(so.nullifyLoadAmount, f.nullifyLoadAmount); io.put_m(so.applyCheckTokenID1ToAmount, f.applyT1Amt);
    io.put_m(so.nullifyAmount_0, f.na0); io.put_m(so.nullifyAmount, f.nullifyAmount);
    {   // ---- B: RqTxVerifier
        const Fc rq_c = io.in_c(in.rqOffset);
        num2bits_dev(io, o.rq_n2b, rq_c, 3, C_RTX_RQ_N2B);
        const Fr s[3] = {fr_from_bit(c_bit(rq_c, 0)), fr_from_bit(c_bit(rq_c, 1)), fr_from_bit(c_bit(rq_c, 2))};
        auto one_mux = [&](const int m, uint32_t rq_sig, int cid) {
            const Fr c[8] = {zero, nb.fut(m, 0), nb.fut(m, 1), nb.fut(m, 2), nb.past(m, 3), nb.past(m, 2), nb.past(m, 1), nb.past(m, 0)};
            io.chk(cid, mux3_dev(io, o.rq_mux[m], c, s), io.in_m(rq_sig));
        };
        one_mux(0, in.rqTxCompressedDataV2, C_RTX_RQ_V2);
        one_mux(1, in.rqToEthAddr, C_RTX_RQ_ETHADDR);
        one_mux(2, in.rqToBjjAy, C_RTX_RQ_BJJAY);
    }
    // ---- C: ForceEqualIfEnabled x8 ((1 - isz.out) * enabled === 0): the outputs are the comparisons above, the IsZero signals follow
    const Fr eqNonce = fr_iszero_bit(fr_sub(io.in_m(in.nonce1), x.nonce));
    const Fr eqToEth = fr_iszero_bit(fr_sub(io.in_m(in.ethAddr2), io.in_m(in.toEthAddr)));
    const Fr eqToAy = fr_iszero_bit(fr_sub(io.in_m(in.toBjjAy), io.in_m(in.ay2)));
    const Fr eqToSign = fr_iszero_bit(fr_sub(x.toBjjSign, io.in_m(in.sign2)));
    auto force = [&](const Fr& e, const Fr& enabled, int cid) { io.chk_zero(cid, fr_mul(fr_sub(one, e), enabled)); };
    force(eqNonce, f.notOn, C_RTX_NONCE);
    const Fr en_toEth = fr_sub(one, fr_mul(fr_sub(one, f.checkToEthAddr), fr_sub(one, f.checkToBjj)));
    io.put_m(o.checkToEthAddr_en, en_toEth);
    force(eqToEth, en_toEth, C_RTX_TOETHADDR);
    force(eqToAy, f.checkToBjj, C_RTX_TOBJJAY);
    force(eqToSign, f.checkToBjj, C_RTX_TOBJJSIGN);
    force(f.eqT1, f.notOn, C_RTX_TOKENID1);
    const Fr en_t2 = fr_mul(f.notOn, fr_sub(one, f.isP2Insert));
    io.put_m(o.checkTokenID2_en, en_t2);
    force(f.eqT2, en_t2, C_RTX_TOKENID2);
    force(f.eqT1, f.isP1Insert, C_RTX_TOKENID1_L1);
    force(f.eqEth, f.isP1Insert, C_RTX_FROMETHADDR);
    // ---- every IsZero of the front (states, phase C, BalanceUpdater's effectiveAmount): (inv, out) signals, two inversions for 14 slots
    {
        const Fr finalFromIdx = f.finalFromIdx, finalToIdx = f.finalToIdx, loadAmount = f.loadAmount, effAmt1 = f.effAmt1;
        auto operand = [&](int k) -> Fr {
            switch (k) {
                case 0: return x.toIdx;
                case 1: return fr_sub(io.in_m(in.toEthAddr), fr_sub(m_pow2(160), one));
                case 2: return finalFromIdx;
                case 3: return loadAmount;
                case 4: return x.amount;
                case 5: return fr_sub(io.in_m(in.ethAddr1), io.in_m(in.fromEthAddr));
                case 6: return fr_sub(io.in_m(in.tokenID1), x.tokenID);
                case 7: return fr_sub(io.in_m(in.tokenID2), x.tokenID);
                case 8: return fr_sub(io.in_m(in.nonce1), x.nonce);
                case 9: return fr_sub(io.in_m(in.ethAddr2), io.in_m(in.toEthAddr));
                case 10: return fr_sub(io.in_m(in.toBjjAy), io.in_m(in.ay2));
                case 11: return fr_sub(x.toBjjSign, io.in_m(in.sign2));
                case 12: return fr_sub(finalToIdx, one);
                default: return effAmt1;
            }
        };
        auto store = [&](int k, const Fr& v, const Fr& vi) {
            auto put = [&](IsZOff off) { (void)is_zero_dev(io, off, v, vi); };
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
    // ---- hand-off to the hash / smt / eddsa / back steps (this lane's share)
    const Fr isP2Nop = fr_sub(one, fr_iszero_bit(f.effAmt1));
    sc.set(SC_ISP1INSERT, f.isP1Insert); sc.set(SC_ISP2INSERT, f.isP2Insert);
    sc.set(SC_OLDVALUE1, io.in_m(in.oldValue1)); sc.set(SC_OLDVALUE2, io.in_m(in.oldValue2));
    sc.set(SC_KEY_1, f.key1); sc.set(SC_KEY_2, f.key2);
    sc.set(SC_P1_FNC0, f.P1_fnc0); sc.set(SC_P1_FNC1, f.P1_fnc1);
    sc.set(SC_P2_FNC0, fr_mul(f.P2_fnc0, isP2Nop)); sc.set(SC_P2_FNC1, fr_mul(f.P2_fnc1, isP2Nop));
    sc.set(SC_ISOLD0_1, io.in_m(in.isOld0_1)); sc.set(SC_ISOLD0_2, io.in_m(in.isOld0_2));
    sc.set(SC_ISEXIT, f.isExit); sc.set(SC_OLDSTATEROOT, x.oldStateRoot); sc.set(SC_OLDEXITROOT, x.oldExitRoot);
    sc.set(SC_ED_ENABLED, f.verifySignEnabled);
    sc.set(SC_ED_S, io.in_m(in.s)); sc.set(SC_ED_R8X, io.in_m(in.r8x)); sc.set(SC_ED_R8Y, io.in_m(in.r8y));
    if (own_sig) sc.set(SC_SIGL2HASH, x.sigL2Hash);   // else: the DecodeTx lane stores it
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
    const RtxStates f = rtx_states_dev(io, in, x);
    const Fr tokenID1 = io.in_m(in.tokenID1), tokenID2 = io.in_m(in.tokenID2), nonce1 = io.in_m(in.nonce1), nonce2 = io.in_m(in.nonce2);
    const Fr sign1 = io.in_m(in.sign1), sign2 = io.in_m(in.sign2), ay1 = io.in_m(in.ay1), ay2 = io.in_m(in.ay2);
    const Fr ethAddr1 = io.in_m(in.ethAddr1), ethAddr2 = io.in_m(in.ethAddr2);
    const Fr p32 = m_pow2(32), p72 = m_pow2(72);
    auto e0 = [&](const Fr& tok, const Fr& non, const Fr& sg) { return fr_add(fr_add(tok, fr_mul(non, p32)), fr_mul(sg, p72)); };
    // the 16 Mux1 (s1OldValue / s2OldValue need the old hashes: hash step), each stored where it is computed
    const Fr s1Balance = mux1_dev(io.in_m(in.balance1), zero, f.isP1Insert);
    const Fr s1Sign = mux1_dev(sign1, bjjSign, f.isP1Insert);
    const Fr s1Ay = mux1_dev(ay1, bjjAy, f.isP1Insert);
    const Fr s1Nonce = mux1_dev(nonce1, zero, f.isP1Insert);
    const Fr s1EthAddr = mux1_dev(ethAddr1, io.in_m(in.fromEthAddr), f.isP1Insert);
    const Fr s1TokenID = mux1_dev(tokenID1, x.tokenID, f.isP1Insert);
    const Fr s1OldKey = mux1_dev(f.key1, io.in_m(in.oldKey1), f.isP1Insert);
    io.put_m(o.mux16 + MX_S1BALANCE, s1Balance); io.put_m(o.mux16 + MX_S1SIGN, s1Sign); io.put_m(o.mux16 + MX_S1AY, s1Ay); io.put_m(o.mux16 + MX_S1NONCE, s1Nonce);
    io.put_m(o.mux16 + MX_S1ETHADDR, s1EthAddr); io.put_m(o.mux16 + MX_S1TOKENID, s1TokenID); io.put_m(o.mux16 + MX_S1OLDKEY, s1OldKey);
    const Fr s2Balance = mux1_dev(io.in_m(in.balance2), zero, f.isP2Insert);
    const Fr s2Sign = mux1_dev(sign2, s1Sign, f.isP2Insert);
    const Fr s2Ay = mux1_dev(ay2, s1Ay, f.isP2Insert);
    const Fr s2Nonce = mux1_dev(nonce2, zero, f.isP2Insert);
    const Fr s2EthAddr = mux1_dev(ethAddr2, s1EthAddr, f.isP2Insert);
    const Fr s2TokenID = mux1_dev(tokenID2, s1TokenID, f.isP2Insert);
    const Fr s2OldKey = mux1_dev(f.key2, io.in_m(in.oldKey2), f.isP2Insert);
    io.put_m(o.mux16 + MX_S2BALANCE, s2Balance); io.put_m(o.mux16 + MX_S2SIGN, s2Sign); io.put_m(o.mux16 + MX_S2AY, s2Ay); io.put_m(o.mux16 + MX_S2NONCE, s2Nonce);
    io.put_m(o.mux16 + MX_S2ETHADDR, s2EthAddr); io.put_m(o.mux16 + MX_S2TOKENID, s2TokenID); io.put_m(o.mux16 + MX_S2OLDKEY, s2OldKey);
    // ---- F (inputs only): signSignature / aySignature
    const Fr signSig = fr_mul(s1Sign, f.verifySignEnabled), aySig = fr_mul(s1Ay, f.verifySignEnabled);
    io.put_m(o.ed.signSignature, signSig); io.put_m(o.ed.aySignature, aySig);
    // ---- hand-off: the hash-state inputs but the two new balances (balance lane), the old keys, the signature's key
    sc.set(SC_HS_IN + 0, e0(tokenID1, nonce1, sign1)); sc.set(SC_HS_IN + 1, io.in_m(in.balance1)); sc.set(SC_HS_IN + 2, ay1); sc.set(SC_HS_IN + 3, ethAddr1);
    sc.set(SC_HS_IN + 4, e0(tokenID2, nonce2, sign2)); sc.set(SC_HS_IN + 5, io.in_m(in.balance2)); sc.set(SC_HS_IN + 6, ay2); sc.set(SC_HS_IN + 7, ethAddr2);
    sc.set(SC_HS_IN + 8, e0(s1TokenID, fr_add(s1Nonce, f.notOn), s1Sign));
    sc.set(SC_HS_IN + 10, s1Ay); sc.set(SC_HS_IN + 11, s1EthAddr);
    sc.set(SC_HS_IN + 12, e0(s2TokenID, s2Nonce, s2Sign));
    sc.set(SC_HS_IN + 14, s2Ay); sc.set(SC_HS_IN + 15, s2EthAddr);
    sc.set(SC_KEY_S1OLD, s1OldKey); sc.set(SC_KEY_S2OLD, s2OldKey);
    sc.set(SC_ED_SIGN, signSig); sc.set(SC_ED_AYSIG, aySig); sc.set(SC_ED_AY, s1Ay);
}

// ---- balance lane: G (BalanceUpdater) and H -- the FeeAccumulator here (standalone RollupTx) or as a kernel of its own beside the chains the
// front kernel feeds (RollupMain: it is half of a transaction's front arithmetic and feeds none of them)
template <class IN, class FEE, bool FEEACC>
__device__ __forceinline__ FrontOut rtx_balance_lane_dev(const UnitIO& io, const Scratch& sc, const RtxOff& o, const IN& in, const RtxExt& x, int Fn, const FEE& feeSrc) {
    const Fr one = fr_one(), zero = fr_zero();
    const RtxStates f = rtx_states_dev(io, in, x);
    const BalUpdOff& bo = o.bu;
    const Fc userFee_c = fr_to_canon(x.userFee);
    const Fr fee2Charge = compute_fee_dev(io, bo.fee, userFee_c, x.amount, fr_mul(f.notOn, fr_sub(one, f.ffz)));
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
