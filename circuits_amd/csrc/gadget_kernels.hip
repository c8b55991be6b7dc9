// The gadget templates the reference's unit suites instantiate as `component main`: one lane evaluates one
// instance and writes its witness signal-major, like every other kernel of the library.
//   DecodeFloat      src/lib/decode-float.circom:51     test/lib/decode-float.test.js
//   ComputeFee       src/compute-fee.circom:12          test/compute-fee.test.js
//   FeeAccumulator   src/fee-accumulator.circom:56      test/fee-accumulator.test.js
//   BalanceUpdater   src/balance-updater.circom:24      test/balance-updater.test.js
//   RollupTxStates   src/rollup-tx-states.circom:39     test/rollup-tx-states.test.js
//   RqTxVerifier     src/rq-tx-verifier.circom:19       test/rq-tx-verifier.test.js
//   Mux256           src/lib/mux256.circom:10           test/lib/mux256.test.js
//   BitsCompressed2AySign  src/lib/utils-bjj.circom:12  test/lib/utils-bjj.test.js   (AySign2Ax: eddsa_kernels.hip, beside the curve code)
// Inside RollupTx the same statements are fused into rollup_tx_front_dev (tx_dev.h) with their inversions batched across the
// gadgets; here each gadget stands alone and shares the leaf helpers (compute_fee_dev, mux3_dev, decode_float_dev, is_zero_dev).
#include <hip/hip_runtime.h>
#include "tx_dev.h"
#include "kernels.h"

namespace hz {

// IsZero / IsEqual of up to N operands with one inversion
template <int N>
struct IszBatch {
    Fr z[N], zi[N];
    __device__ __forceinline__ void invert() {
        for (int i = 0; i < N; i++) zi[i] = z[i];
        batch_inv<N>(zi, N);
    }
    __device__ __forceinline__ Fr out(const UnitIO& io, IsZOff off, int i) const { return is_zero_dev(io, off, z[i], zi[i]); }
};

static __device__ void decode_float_main(const UnitIO& io, const GadgetArgs& a) {
    const Fc in = io.in_c(a.io.in[0]);
    num2bits_dev(io, a.n2b40, in, 40, C_RTX_N2B_LOADAMOUNTF);
    io.put_m(a.io.out[0], decode_float_dev(io, a.df, c_bits64(in, 0, 40)));
}

static __device__ void compute_fee_main(const UnitIO& io, const GadgetArgs& a) {
    compute_fee_dev(io, a.bu.fee, io.in_c(a.io.in[0]), io.in_m(a.io.in[1]), io.in_m(a.io.in[2]));
}

static __device__ void fee_accumulator_main(const UnitIO& io, const GadgetArgs& a) {
    const Fr one = fr_one();
    const Fr tokenID = io.in_m(a.io.in[0]), fee2Charge = io.in_m(a.io.in[1]);
    Fr selIn = fr_zero();
    for (uint32_t base = 0; base < a.F; base += 16) {
        const int n = (int)(a.F - base < 16 ? a.F - base : 16);
        Fr dz[16], dzi[16];
        for (int i = 0; i < n; i++) { dz[i] = fr_sub(io.in_m(a.io.in[2] + base + i), tokenID); dzi[i] = dz[i]; }
        batch_inv<16>(dzi, n);
        for (int i = 0; i < n; i++) {
            const uint32_t b = a.feeAcc + FA_N * (base + i);
            const Fr eq = is_zero_dev(io, b + FA_ISZ_INV, dz[i], dzi[i]);
            const Fr selOut = fr_sub(one, fr_mul(fr_sub(one, eq), fr_sub(one, selIn)));
            const Fr ms = fr_mul(eq, fr_sub(one, selIn));
            const Fr out = fr_add(fr_mul(fee2Charge, ms), io.in_m(a.io.in[3] + base + i));
            io.put_m(b + FA_SELOUT, selOut); io.put_m(b + FA_MUX_S, ms); io.put_m(b + FA_MUX_OUT, out);
            io.put_m(a.io.out[0] + base + i, out);
            selIn = selOut;
        }
    }
}

static __device__ void balance_updater_main(const UnitIO& io, const GadgetArgs& a) {
    const Fr one = fr_one(), zero = fr_zero();
    const BalUpdOff& bo = a.bu;
    const Fr oldSender = io.in_m(a.io.in[0]), oldReceiver = io.in_m(a.io.in[1]), amount = io.in_m(a.io.in[2]), loadAmount = io.in_m(a.io.in[3]);
    const Fc feeSel = io.in_c(a.io.in[4]);
    const Fr onChain = io.in_m(a.io.in[5]), nop = io.in_m(a.io.in[6]), nullifyLoad = io.in_m(a.io.in[7]), nullifyAmt = io.in_m(a.io.in[8]);
    const Fr notOn = fr_sub(one, onChain);
    const Fr fee2Charge = compute_fee_dev(io, bo.fee, feeSel, amount, fr_mul(notOn, fr_sub(one, nop)));
    const Fr el1 = fr_mul(loadAmount, onChain), el2 = fr_mul(el1, fr_sub(one, nullifyLoad));
    const Fr ea1 = fr_mul(amount, fr_sub(one, nop)), ea2 = fr_mul(ea1, fr_sub(one, nullifyAmt));
    io.put_m(bo.effLoad1, el1); io.put_m(bo.effLoad2, el2); io.put_m(bo.effAmt1, ea1); io.put_m(bo.effAmt2, ea2);
    const Fc sb_c = fr_to_canon(fr_sub(fr_sub(fr_add(fr_add(m_pow2(192), oldSender), el2), ea2), fee2Charge));
    num2bits_dev(io, bo.n2bSender, sb_c, 193, C_RTX_BU_N2B_SENDER);
    const uint32_t ufOk = c_bit(sb_c, 192);
    const Fr underflowOk = fr_from_bit(ufOk);
    io.chk_zero(C_RTX_BU_UNDERFLOW, fr_mul(fr_sub(one, underflowOk), notOn));
    const Fr ea3 = ufOk ? ea2 : zero;
    io.put_m(bo.effAmt3, ea3);
    IszBatch<1> zb;
    zb.z[0] = ea1;
    zb.invert();
    const Fr ez = zb.out(io, bo.effAmtIsZero, 0);
    io.put_m(bo.isAmountNullified, fr_sub(one, fr_mul(fr_sub(one, nullifyAmt), underflowOk)));
    io.put_m(a.io.out[0], fr_sub(fr_sub(fr_add(oldSender, el2), ea3), fee2Charge));
    io.put_m(a.io.out[1], fr_add(oldReceiver, ea3));
    io.put_m(a.io.out[2], fr_sub(one, ez));
    io.put_m(a.io.out[3], fee2Charge);
}

static __device__ void rollup_tx_states_main(const UnitIO& io, const GadgetArgs& a) {
    const Fr one = fr_one(), zero = fr_zero();
    const StatesOff& so = a.st;
    const Fr fromIdx = io.in_m(a.io.in[0]), toIdx = io.in_m(a.io.in[1]), toEthAddr = io.in_m(a.io.in[2]), auxFromIdx = io.in_m(a.io.in[3]);
    const Fr auxToIdx = io.in_m(a.io.in[4]), amount = io.in_m(a.io.in[5]), newExit = io.in_m(a.io.in[6]), loadAmount = io.in_m(a.io.in[7]);
    const Fr newAccount = io.in_m(a.io.in[8]), onChain = io.in_m(a.io.in[9]), fromEthAddr = io.in_m(a.io.in[10]), ethAddr1 = io.in_m(a.io.in[11]);
    const Fr tokenID = io.in_m(a.io.in[12]), tokenID1 = io.in_m(a.io.in[13]), tokenID2 = io.in_m(a.io.in[14]);
    const Fr notOn = fr_sub(one, onChain);
    const Fr selFrom_s = fr_mul(onChain, newAccount);
    const Fr finalFromIdx = mux1_dev(fromIdx, auxFromIdx, selFrom_s);
    io.put_m(so.selFromIdx_s, selFrom_s); io.put_m(so.selFromIdx_out, finalFromIdx);
    enum { Z_TOIDX = 0, Z_ANY, Z_FFROM, Z_LOAD, Z_AMT, Z_FETH, Z_T1, Z_T2, Z_N };
    IszBatch<Z_N> zb;
    zb.z[Z_TOIDX] = toIdx;
    zb.z[Z_ANY] = fr_sub(toEthAddr, fr_sub(m_pow2(160), one));
    zb.z[Z_FFROM] = finalFromIdx;
    zb.z[Z_LOAD] = loadAmount;
    zb.z[Z_AMT] = amount;
    zb.z[Z_FETH] = fr_sub(ethAddr1, fromEthAddr);
    zb.z[Z_T1] = fr_sub(tokenID1, tokenID);
    zb.z[Z_T2] = fr_sub(tokenID2, tokenID);
    zb.invert();
    const Fr tz = zb.out(io, so.toIdxIsZero, Z_TOIDX);
    const Fr selectAuxToIdx = fr_mul(notOn, tz);
    io.put_m(so.selectAuxToIdx, selectAuxToIdx);
    const Fr finalToIdx = mux1_dev(toIdx, auxToIdx, selectAuxToIdx);
    io.put_m(so.selToIdx_out, finalToIdx);
    const Fr isAny = zb.out(io, so.isToEthAddrAny, Z_ANY);
    IszBatch<1> ze;
    ze.z[0] = fr_sub(finalToIdx, one);
    ze.invert();
    const Fr isExit = ze.out(io, so.checkIsExit, 0);
    const Fr ffz = zb.out(io, so.finalFromIdxIsZero, Z_FFROM);
    const Fr isFinalFromIdx = fr_sub(one, ffz);
    const Fr isLoadAmount = fr_sub(one, zb.out(io, so.loadAmountIsZero, Z_LOAD));
    const Fr isAmount = fr_sub(one, zb.out(io, so.amountIsZero, Z_AMT));
    io.chk_zero(C_RTX_ST_L2_LOADAMOUNT, fr_mul(notOn, isLoadAmount));
    io.chk_zero(C_RTX_ST_L2_NEWACCOUNT, fr_mul(notOn, newAccount));
    const Fr isP1Insert = selFrom_s;
    const Fr P1_fnc0 = fr_mul(isP1Insert, isFinalFromIdx), P1_fnc1 = fr_mul(fr_sub(one, isP1Insert), isFinalFromIdx);
    io.put_m(so.isP1Insert, isP1Insert); io.put_m(so.P1_fnc0, P1_fnc0); io.put_m(so.P1_fnc1, P1_fnc1);
    {   // Mux2 c = [0, f, f, f], s = [P1_fnc0, P1_fnc1]
        const Fr s10 = fr_mul(P1_fnc1, P1_fnc0);
        const Fr a10 = fr_mul(fr_neg(finalFromIdx), s10), a1 = fr_mul(finalFromIdx, P1_fnc1), a0 = fr_mul(finalFromIdx, P1_fnc0);
        io.put_m(so.mux1 + M2_S10, s10); io.put_m(so.mux1 + M2_A10, a10); io.put_m(so.mux1 + M2_A1, a1); io.put_m(so.mux1 + M2_A0, a0);
        io.put_m(a.io.out[2], fr_add(fr_add(a10, a1), a0));   // key1
    }
    const Fr isP2Insert = fr_mul(isExit, newExit);
    const Fr P2_fnc0 = fr_mul(isP2Insert, isFinalFromIdx), P2_fnc1 = fr_mul(fr_sub(one, isP2Insert), isFinalFromIdx);
    io.put_m(so.isP2Insert, isP2Insert); io.put_m(so.P2_fnc0, P2_fnc0); io.put_m(so.P2_fnc1, P2_fnc1);
    {   // Mux2 c = [0, finalToIdx, 0, finalFromIdx], s = [isAmount, isExit]
        const Fr s10 = fr_mul(isExit, isAmount);
        const Fr a10 = fr_mul(fr_sub(finalFromIdx, finalToIdx), s10), a0 = fr_mul(finalToIdx, isAmount);
        io.put_m(so.mux2 + M2_S10, s10); io.put_m(so.mux2 + M2_A10, a10); io.put_m(so.mux2 + M2_A1, zero); io.put_m(so.mux2 + M2_A0, a0);
        io.put_m(a.io.out[3], fr_add(a10, a0));               // key2
    }
    io.put_m(a.io.out[8], isExit);
    io.put_m(so.verifySignEnabled, fr_mul(notOn, isFinalFromIdx));
    const Fr nop = ffz;
    io.put_m(a.io.out[10], nop);
    const Fr tmpE = fr_mul(fr_sub(one, isAny), selectAuxToIdx), tmpB = fr_mul(isAny, selectAuxToIdx);
    io.put_m(so.tmpCheckToEthAddr, tmpE); io.put_m(so.tmpCheckToBjj, tmpB);
    io.put_m(so.checkToEthAddr, fr_mul(tmpE, fr_sub(one, nop))); io.put_m(so.checkToBjj, fr_mul(tmpB, fr_sub(one, nop)));
    const Fr onNotCreate = fr_mul(fr_sub(one, newAccount), onChain);
    const Fr shouldEth = fr_mul(onNotCreate, isAmount);
    io.put_m(so.onChainNotCreateAccount, onNotCreate); io.put_m(so.shouldCheckEthAddr, shouldEth);
    const Fr nullEth = fr_mul(shouldEth, fr_sub(one, zb.out(io, so.checkFromEthAddr, Z_FETH)));
    io.put_m(so.applyNullifierEthAddr, nullEth);
    const Fr nullT1 = fr_mul(onNotCreate, fr_sub(one, zb.out(io, so.checkTokenID1, Z_T1)));
    io.put_m(so.applyNullifierTokenID1, nullT1);
    const Fr sc20 = fr_mul(onChain, isAmount), sc21 = fr_mul(sc20, fr_sub(one, isP2Insert));
    io.put_m(so.shouldCheckTokenID2_0, sc20); io.put_m(so.shouldCheckTokenID2_1, sc21);
    const Fr nullT2 = fr_mul(sc21, fr_sub(one, zb.out(io, so.checkTokenID2, Z_T2)));
    io.put_m(so.applyNullifierTokenID2, nullT2);
    io.put_m(so.nullifyLoadAmount, fr_mul(nullT1, isLoadAmount));
    const Fr applyT1Amt = fr_mul(nullT1, isAmount);
    io.put_m(so.applyCheckTokenID1ToAmount, applyT1Amt);
    const Fr na0 = fr_sub(one, fr_mul(fr_sub(one, nullEth), fr_sub(one, nullT2)));
    io.put_m(so.nullifyAmount_0, na0);
    io.put_m(so.nullifyAmount, fr_sub(one, fr_mul(fr_sub(one, na0), fr_sub(one, applyT1Amt))));
}

static __device__ void rq_tx_verifier_main(const UnitIO& io, const GadgetArgs& a) {
    const Fc rq_c = io.in_c(a.io.in[9]);
    num2bits_dev(io, a.rq_n2b, rq_c, 3, C_RTX_RQ_N2B);
    const Fr s[3] = {fr_from_bit(c_bit(rq_c, 0)), fr_from_bit(c_bit(rq_c, 1)), fr_from_bit(c_bit(rq_c, 2))};
    const int cid[3] = {C_RTX_RQ_V2, C_RTX_RQ_ETHADDR, C_RTX_RQ_BJJAY};
    for (int m = 0; m < 3; m++) {
        const uint32_t fut = a.io.in[2 * m], pst = a.io.in[2 * m + 1];
        const Fr c[8] = {fr_zero(), io.in_m(fut), io.in_m(fut + 1), io.in_m(fut + 2), io.in_m(pst + 3), io.in_m(pst + 2), io.in_m(pst + 1), io.in_m(pst)};
        io.chk(cid[m], mux3_dev(io, a.rq_mux[m], c, s), io.in_m(a.io.in[6 + m]));
    }
}

static __device__ void mux256_main(const UnitIO& io, const GadgetArgs& a) {
    Fr sel[8], lvl1[16];
    for (int i = 0; i < 8; i++) sel[i] = io.in_m(a.io.in[0] + i);
#pragma unroll 1
    for (int m = 0; m < 16; m++) {
        Fr c[16];
        for (int k = 0; k < 16; k++) c[k] = io.in_m(a.io.in[1] + 16 * m + k);
        lvl1[m] = mux4_var_dev(io, a.io.mux + MX4V_N * m, c, sel);
    }
    io.put_m(a.io.out[0], mux4_var_dev(io, a.io.mux + MX4V_N * 16, lvl1, sel + 4));
}

static __device__ void bits_compressed_2_ay_sign_main(const UnitIO& io, const GadgetArgs& a) {
    Fr acc = fr_zero();
#pragma unroll 1
    for (int i = 253; i >= 0; i--) acc = fr_add(fr_dbl(acc), io.in_m(a.io.in[0] + i));   // Bits2Num(254): inputs need not be boolean
    io.put_m(a.io.out[0], acc);
    io.put_c(a.io.out[1], io.in_c(a.io.in[0] + 255));
}

template <int TMPL>
__global__ __launch_bounds__(HZ_BLOCK) void k_gadget(const GadgetArgs a) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= a.N) return;
    const UnitIO io{a.base, a.N, i, i, 0, a.err};
    io.put_u64(0, 1);
    if (TMPL == T_DECODE_FLOAT) decode_float_main(io, a);
    else if (TMPL == T_COMPUTE_FEE) compute_fee_main(io, a);
    else if (TMPL == T_FEE_ACCUMULATOR) fee_accumulator_main(io, a);
    else if (TMPL == T_BALANCE_UPDATER) balance_updater_main(io, a);
    else if (TMPL == T_ROLLUP_TX_STATES) rollup_tx_states_main(io, a);
    else if (TMPL == T_MUX256) mux256_main(io, a);
    else if (TMPL == T_BITS2AYSIGN) bits_compressed_2_ay_sign_main(io, a);
    else rq_tx_verifier_main(io, a);
}

hipError_t launch_gadget(int tmpl, const GadgetArgs& a, hipStream_t s) {
    const dim3 g((a.N + HZ_BLOCK - 1) / HZ_BLOCK), b(HZ_BLOCK);
    switch (tmpl) {
        case T_DECODE_FLOAT: hipLaunchKernelGGL(k_gadget<T_DECODE_FLOAT>, g, b, 0, s, a); break;
        case T_COMPUTE_FEE: hipLaunchKernelGGL(k_gadget<T_COMPUTE_FEE>, g, b, 0, s, a); break;
        case T_FEE_ACCUMULATOR: hipLaunchKernelGGL(k_gadget<T_FEE_ACCUMULATOR>, g, b, 0, s, a); break;
        case T_BALANCE_UPDATER: hipLaunchKernelGGL(k_gadget<T_BALANCE_UPDATER>, g, b, 0, s, a); break;
        case T_ROLLUP_TX_STATES: hipLaunchKernelGGL(k_gadget<T_ROLLUP_TX_STATES>, g, b, 0, s, a); break;
        case T_RQ_TX_VERIFIER: hipLaunchKernelGGL(k_gadget<T_RQ_TX_VERIFIER>, g, b, 0, s, a); break;
        case T_MUX256: hipLaunchKernelGGL(k_gadget<T_MUX256>, g, b, 0, s, a); break;
        case T_BITS2AYSIGN: hipLaunchKernelGGL(k_gadget<T_BITS2AYSIGN>, g, b, 0, s, a); break;
        default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

}  // namespace hz
