// hz_layout.h -- the witness FORMAT of the MI355X-native Hermez witness generator.
//
// circom numbers witness variables in compiler order, which cannot be reproduced without the
// compiler (absent here, SURVEY 7 "Witness layout"). This header defines our own deterministic
// numbering plus the symbol table that maps circom-style dotted names to it, the role the
// reference's generated `.sym` file plays for circuit.assertOut / getSignal
// (reference test/helpers/helpers.js:143,154,170).
//
// Which signals are stored: the ones a constraint-reducing R1CS build keeps as variables --
// main inputs/outputs, every signal assigned with `<--`, and every signal defined by a `<==`
// whose right-hand side is a product of signals. Signals that are linear combinations of stored
// ones (Ark/Mix outputs of Poseidon, Bits2Num outputs, wire-through component inputs ...) are
// not stored; neither are signals that are compile-time constants (the EscalarMulFix window
// tables of BASE8, the Mux256 fee table). Per transaction this is ~46.5 k elements at
// nLevels=32,maxFeeTx=64, next to the reference's own size model of 47 641
// (reference tools/circuit-constraints.js:31-44).
//
// Physical order: a witness is a list of SECTIONS. A section has `n_units` units (transactions,
// fee transactions, or independent instances) and `n_sigs` signals per unit, stored
// signal-major: element (sig, unit) sits at  base + sig * n_units + unit  (x 32 bytes). The 64
// lanes of a wavefront work on consecutive units, so every store instruction of a lane group
// covers contiguous HBM.
//
// Header-only, plain C++17, no device code: shared by the product (circuits_amd/csrc), the CPU
// oracle (oracle/) and the Node addon. It contains no arithmetic.
#pragma once
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <string>
#include <unordered_map>
#include <vector>

namespace hzl {

// ------------------------------------------------------------------------------------------------
// template ids (== hz_template in hermez_witness.h)
enum { T_ROLLUP_MAIN = 0, T_ROLLUP_TX = 1, T_DECODE_TX = 2, T_FEE_TX = 3, T_HASH_STATE = 4, T_WITHDRAW = 5, T_HASH_INPUTS = 6,
       // the gadget templates the reference's remaining suites instantiate as `component main`
       T_DECODE_FLOAT = 7, T_COMPUTE_FEE = 8, T_FEE_ACCUMULATOR = 9, T_BALANCE_UPDATER = 10, T_ROLLUP_TX_STATES = 11, T_RQ_TX_VERIFIER = 12,
       T_MUX256 = 13, T_BITS2AYSIGN = 14, T_AYSIGN2AX = 15,
       // circomlib's own SMT templates as `component main` (SMTProcessor(nLevels), SMTVerifier(nLevels)): the route through which
       // the published sparse-Merkle-tree known answers (tests/golden/smt_kat.json) reach the level-hash kernels
       T_SMT_PROCESSOR = 16, T_SMT_VERIFIER = 17,
       T_COUNT = 18 };

struct Params {
    int tmpl = 0, nTx = 0, L = 0, maxL1 = 0, F = 0, n_inst = 1;
};

// L1TxFullData bits: fromEthAddr 160 | fromBjjCompressed 256 | fromIdx 48 | loadAmountF 40 | amountF 40 | tokenID 32 | toIdx 48
// (reference src/decode-tx.circom:73, src/rollup-main.circom:443)
enum { L1FULL_BITS = 160 + 256 + 48 + 40 + 40 + 32 + 48 };

inline int poseidon_rp(int t) { static const int rp[6] = {56, 57, 56, 60, 60, 63}; return rp[t - 2]; }
inline int poseidon_nsbox(int t) { return 8 * t + poseidon_rp(t); }
inline int poseidon_nsig(int t) { return 3 * poseidon_nsbox(t); }

// ------------------------------------------------------------------------------------------------
// constraint ids: stable numbers for every `===` (and every `<==`-implied check that can fail on
// bad inputs). Lower id = earlier in evaluation order inside one unit.
#define HZL_CONSTRAINTS(X)                                                                         \
    /* RollupMain phase A, src/rollup-main.circom:207-219 */                                       \
    X(C_MAIN_IMONCHAIN_BOOL, "main: imOnChain boolean")                                             \
    X(C_MAIN_ONCHAIN_BOOL, "main: onChain boolean")                                                 \
    X(C_MAIN_NEWACCOUNT_BOOL, "main: newAccount boolean")                                           \
    X(C_MAIN_BJJ_BOOL, "main: fromBjjCompressed boolean")                                           \
    X(C_MAIN_ISOLD0_1_BOOL, "main: isOld0_1 boolean")                                               \
    X(C_MAIN_ISOLD0_2_BOOL, "main: isOld0_2 boolean")                                               \
    /* DecodeTx, src/decode-tx.circom */                                                            \
    X(C_DEC_N2B_DATA, "decodeTx.n2bData: Num2Bits(225) sum")                                        \
    X(C_DEC_PAD_FROM, "decodeTx: paddingFrom === 0")                                                \
    X(C_DEC_PAD_TO, "decodeTx: paddingTo === 0")                                                    \
    X(C_DEC_N2B_AMOUNT, "decodeTx.n2bAmount: Num2Bits(40) sum")                                     \
    X(C_DEC_N2B_FINALTOIDX, "decodeTx.n2bFinalToIdx: Num2Bits(nLevels) sum")                        \
    X(C_DEC_N2B_TOETHADDR, "decodeTx.n2bToEthAddr: Num2Bits(160) sum")                              \
    X(C_DEC_N2B_MAXNUMBATCH, "decodeTx.n2bMaxNumBatch: Num2Bits(32) sum")                           \
    X(C_DEC_N2B_FROMETHADDR, "decodeTx.n2bFromEthAddr: Num2Bits(160) sum")                          \
    X(C_DEC_N2B_LOADAMOUNTF, "decodeTx.n2bLoadAmountF: Num2Bits(40) sum")                           \
    X(C_DEC_NEWACCOUNT, "decodeTx: onChain*fromIdxIsZero === newAccount")                           \
    X(C_DEC_IDXCHECKER, "decodeTx.idxChecker")                                                      \
    X(C_DEC_L1_BEFORE_L2, "decodeTx: (1-previousOnChain)*onChain === 0")                            \
    X(C_DEC_CHAINID, "decodeTx.chainIDChecker")                                                     \
    X(C_DEC_CONSTSIG, "decodeTx.constSigChecker")                                                   \
    X(C_DEC_N2B_MAXNUMBATCH_LT, "decodeTx.isMaxNumBatchOk: Num2Bits(33) sum")                       \
    X(C_DEC_MAXNUMBATCH, "decodeTx: maxNumBatch check")                                             \
    /* RollupMain phase C, src/rollup-main.circom:258-265 */                                       \
    X(C_MAIN_IM_V2, "main: decodeTx.txCompressedDataV2 === txCompressedDataV2")                     \
    X(C_MAIN_IM_ONCHAIN, "main: decodeTx.onChain === imOnChain")                                    \
    X(C_MAIN_IM_OUTIDX, "main: decodeTx.outIdx === imOutIdx")                                       \
    /* RollupTx, src/rollup-tx.circom */                                                            \
    X(C_RTX_N2B_LOADAMOUNTF, "rollupTx.n2bloadAmountF: Num2Bits(40) sum")                           \
    X(C_RTX_ST_L2_LOADAMOUNT, "rollupTx.states: (1-onChain)*isLoadAmount === 0")                    \
    X(C_RTX_ST_L2_NEWACCOUNT, "rollupTx.states: (1-onChain)*newAccount === 0")                      \
    X(C_RTX_RQ_N2B, "rollupTx.rqTxVerifier.n2b: Num2Bits(3) sum")                                   \
    X(C_RTX_RQ_V2, "rollupTx.rqTxVerifier: txCompressedDataV2")                                     \
    X(C_RTX_RQ_ETHADDR, "rollupTx.rqTxVerifier: toEthAddr")                                         \
    X(C_RTX_RQ_BJJAY, "rollupTx.rqTxVerifier: toBjjAy")                                             \
    X(C_RTX_NONCE, "rollupTx.nonceChecker")                                                         \
    X(C_RTX_TOETHADDR, "rollupTx.checkToEthAddr")                                                   \
    X(C_RTX_TOBJJAY, "rollupTx.toBjjAyChecker")                                                     \
    X(C_RTX_TOBJJSIGN, "rollupTx.toBjjSignChecker")                                                 \
    X(C_RTX_TOKENID1, "rollupTx.checkTokenID1")                                                     \
    X(C_RTX_TOKENID2, "rollupTx.checkTokenID2")                                                     \
    X(C_RTX_TOKENID1_L1, "rollupTx.checkTokenID1L1")                                                \
    X(C_RTX_FROMETHADDR, "rollupTx.fromEthAddrChecker")                                             \
    X(C_RTX_FEE_N2B_SEL, "rollupTx.balanceUpdater.computeFee.n2bFeeSel: Num2Bits(8) sum")           \
    X(C_RTX_FEE_BITS, "rollupTx.balanceUpdater.computeFee: lcIn === feeOutNotShifted")              \
    X(C_RTX_FEE_OVF_SHIFTED, "rollupTx.balanceUpdater.computeFee: applyShift*lcOverflowShifted === 0") \
    X(C_RTX_FEE_OVF_NOTSHIFTED, "rollupTx.balanceUpdater.computeFee: (1-applyShift)*lcOverflowNotShifted === 0") \
    X(C_RTX_BU_N2B_SENDER, "rollupTx.balanceUpdater.n2bSender: Num2Bits(193) sum")                  \
    X(C_RTX_BU_UNDERFLOW, "rollupTx.balanceUpdater: (1-underflowOk)*(1-onChain) === 0")             \
    /* EdDSA + AySign2Ax */                                                                         \
    X(C_RTX_AX_N2B_AY, "rollupTx.getAx.n2bAy: Num2Bits(254) sum")                                   \
    X(C_RTX_AX_ALIAS_Y, "rollupTx.getAx.b2Point.aliasCheckY")                                       \
    X(C_RTX_AX_BABYCHECK, "rollupTx.getAx.b2Point.babyCheck")                                       \
    X(C_RTX_AX_N2B_X, "rollupTx.getAx.b2Point.n2bX: Num2Bits(254) sum")                             \
    X(C_RTX_AX_ALIAS_X, "rollupTx.getAx.b2Point.aliasCheckX")                                       \
    X(C_RTX_AX_SIGN, "rollupTx.getAx.b2Point: signCalc.out === in[255]")                            \
    X(C_RTX_SIG_N2B_S, "rollupTx.sigVerifier.snum2bits: Num2Bits(253) sum")                         \
    X(C_RTX_SIG_S_RANGE, "rollupTx.sigVerifier: compConstant.out*enabled === 0")                    \
    X(C_RTX_SIG_H_N2B, "rollupTx.sigVerifier.h2bits: Num2Bits(254) sum")                            \
    X(C_RTX_SIG_H_ALIAS, "rollupTx.sigVerifier.h2bits.aliasCheck")                                  \
    X(C_RTX_SIG_A_NONZERO, "rollupTx.sigVerifier: isZero.out*enabled === 0")                        \
    X(C_RTX_SIG_EC, "rollupTx.sigVerifier: elliptic-curve division check")                          \
    X(C_RTX_SIG_EQX, "rollupTx.sigVerifier.eqCheckX")                                               \
    X(C_RTX_SIG_EQY, "rollupTx.sigVerifier.eqCheckY")                                               \
    /* SMT processors */                                                                            \
    X(C_RTX_P1_N2B_OLD, "rollupTx.processor1.n2bOld")                                               \
    X(C_RTX_P1_ALIAS_OLD, "rollupTx.processor1.n2bOld.aliasCheck")                                  \
    X(C_RTX_P1_N2B_NEW, "rollupTx.processor1.n2bNew")                                               \
    X(C_RTX_P1_ALIAS_NEW, "rollupTx.processor1.n2bNew.aliasCheck")                                  \
    X(C_RTX_P1_LEVINS, "rollupTx.processor1.smtLevIns: last sibling must be zero")                  \
    X(C_RTX_P1_SM_FINAL, "rollupTx.processor1: final state")                                        \
    X(C_RTX_P1_OLDROOT, "rollupTx.processor1.checkOldInput")                                        \
    X(C_RTX_P1_KEYS, "rollupTx.processor1: keysOk.out === 0")                                       \
    X(C_RTX_P2_N2B_OLD, "rollupTx.processor2.n2bOld")                                               \
    X(C_RTX_P2_ALIAS_OLD, "rollupTx.processor2.n2bOld.aliasCheck")                                  \
    X(C_RTX_P2_N2B_NEW, "rollupTx.processor2.n2bNew")                                               \
    X(C_RTX_P2_ALIAS_NEW, "rollupTx.processor2.n2bNew.aliasCheck")                                  \
    X(C_RTX_P2_LEVINS, "rollupTx.processor2.smtLevIns: last sibling must be zero")                  \
    X(C_RTX_P2_SM_FINAL, "rollupTx.processor2: final state")                                        \
    X(C_RTX_P2_OLDROOT, "rollupTx.processor2.checkOldInput")                                        \
    X(C_RTX_P2_KEYS, "rollupTx.processor2: keysOk.out === 0")                                       \
    /* RollupMain phase E / G, src/rollup-main.circom:383-389,422-431 */                            \
    X(C_MAIN_IM_STATEROOT, "main: rollupTx.newStateRoot === imStateRoot")                           \
    X(C_MAIN_IM_EXITROOT, "main: rollupTx.newExitRoot === imExitRoot")                              \
    X(C_MAIN_IM_ACCFEE, "main: rollupTx.accFeeOut === imAccFeeOut")                                 \
    X(C_MAIN_IM_INITFEEROOT, "main: rollupTx[nTx-1].newStateRoot === imInitStateRootFee")           \
    X(C_MAIN_IM_FINALACCFEE, "main: rollupTx[nTx-1].accFeeOut === imFinalAccFee")                   \
    /* FeeTx, src/fee-tx.circom */                                                                  \
    X(C_FEE_TOKENID, "feeTx.tokenIDChecker")                                                        \
    X(C_FEE_P_N2B_OLD, "feeTx.processor.n2bOld")                                                    \
    X(C_FEE_P_ALIAS_OLD, "feeTx.processor.n2bOld.aliasCheck")                                       \
    X(C_FEE_P_N2B_NEW, "feeTx.processor.n2bNew")                                                    \
    X(C_FEE_P_ALIAS_NEW, "feeTx.processor.n2bNew.aliasCheck")                                       \
    X(C_FEE_P_LEVINS, "feeTx.processor.smtLevIns: last sibling must be zero")                       \
    X(C_FEE_P_SM_FINAL, "feeTx.processor: final state")                                             \
    X(C_FEE_P_OLDROOT, "feeTx.processor.checkOldInput")                                             \
    X(C_FEE_P_KEYS, "feeTx.processor: keysOk.out === 0")                                            \
    X(C_MAIN_IM_FEEROOT, "main: feeTx.newStateRoot === imStateRootFee")                             \
    /* HashInputs, src/hash-inputs.circom */                                                        \
    X(C_HI_N2B, "hashInputs: Num2Bits sum")                                                         \
    X(C_HI_PAD, "hashInputs: index padding === 0")                                                  \
    /* Withdraw, src/withdraw.circom */                                                             \
    X(C_WD_N2B_OLD, "withdraw.smtVerify.n2bOld")                                                    \
    X(C_WD_N2B_NEW, "withdraw.smtVerify.n2bNew")                                                    \
    X(C_WD_ALIAS_NEW, "withdraw.smtVerify.n2bNew.aliasCheck")                                       \
    X(C_WD_LEVINS, "withdraw.smtVerify.smtLevIns: last sibling must be zero")                       \
    X(C_WD_SM_FINAL, "withdraw.smtVerify: final state")                                             \
    X(C_WD_KEYS, "withdraw.smtVerify: keysOk.out === 0")                                            \
    X(C_WD_ROOT, "withdraw.smtVerify.checkRoot")                                                    \
    X(C_WD_HI_N2B, "withdraw.hasherInputs: Num2Bits sum")                                           \
    X(C_WD_HI_PAD, "withdraw.hasherInputs: paddingIdx === 0")                                       \
    /* SMTProcessor(n) / SMTVerifier(n) as main (circomlib smt/smtprocessor.circom, smt/smtverifier.circom) */ \
    X(C_SMTP_N2B_OLD, "smtProcessor.n2bOld")                                                        \
    X(C_SMTP_ALIAS_OLD, "smtProcessor.n2bOld.aliasCheck")                                           \
    X(C_SMTP_N2B_NEW, "smtProcessor.n2bNew")                                                        \
    X(C_SMTP_ALIAS_NEW, "smtProcessor.n2bNew.aliasCheck")                                           \
    X(C_SMTP_LEVINS, "smtProcessor.smtLevIns: last sibling must be zero")                           \
    X(C_SMTP_SM_FINAL, "smtProcessor: final state")                                                 \
    X(C_SMTP_OLDROOT, "smtProcessor.checkOldInput")                                                 \
    X(C_SMTP_KEYS, "smtProcessor: keysOk.out === 0")                                                \
    X(C_SMTV_N2B_OLD, "smtVerifier.n2bOld")                                                         \
    X(C_SMTV_ALIAS_OLD, "smtVerifier.n2bOld.aliasCheck")                                            \
    X(C_SMTV_N2B_NEW, "smtVerifier.n2bNew")                                                         \
    X(C_SMTV_ALIAS_NEW, "smtVerifier.n2bNew.aliasCheck")                                            \
    X(C_SMTV_LEVINS, "smtVerifier.smtLevIns: last sibling must be zero")                            \
    X(C_SMTV_SM_FINAL, "smtVerifier: final state")                                                  \
    X(C_SMTV_KEYS, "smtVerifier: keysOk.out === 0")                                                 \
    X(C_SMTV_ROOT, "smtVerifier.checkRoot")

enum ConstraintId {
#define X(id, text) id,
    HZL_CONSTRAINTS(X)
#undef X
        C_COUNT
};
inline const char* constraint_name(int id) {
    static const char* names[] = {
#define X(id, text) text,
        HZL_CONSTRAINTS(X)
#undef X
    };
    return (id >= 0 && id < C_COUNT) ? names[id] : "unknown constraint";
}

// ------------------------------------------------------------------------------------------------
// offsets of gadget signal groups inside a section (units of signals)

// CompConstant: parts[127], num2bits.out[135]
struct CompConstOff { uint32_t parts, bits; };
// Num2Bits_strict: n2b.out[254] + AliasCheck.compConstant
struct N2BStrictOff { uint32_t bits; CompConstOff cc; };
// IsZero: inv, out (consecutive)
typedef uint32_t IsZOff;
// BabyAdd: beta, gamma, delta, tau, xout, yout (consecutive)
typedef uint32_t BabyAddOff;
enum { BA_BETA = 0, BA_GAMMA, BA_DELTA, BA_TAU, BA_XOUT, BA_YOUT, BA_N };
// Poseidon: 3 signals per S-box (in2,in4,out), S-boxes in evaluation order
typedef uint32_t PoseidonOff;

// one level of SMTProcessor: fixed internal order
enum {
    LV_OLDSW_AUX = 0, LV_AUX0, LV_OLDROOT, LV_OLDHASH,            // old side (written by the old-chain lane)
    LV_NEWSW_AUX = LV_OLDHASH + 243, LV_AUX1, LV_AUX2, LV_AUX3, LV_NEWSW_L, LV_NEWSW_R, LV_NEWROOT, LV_NEWHASH,
    LV_SIZE = LV_NEWHASH + 243
};
// SMTProcessorSM stored signals per level
enum { SM_AUX1 = 0, SM_AUX2, SM_OLD0, SM_NEW1, SM_BOT, SM_N };

struct SmtProcOff {
    uint32_t fnc;       // fnc[0],fnc[1] -- only when they are products (processor2), else ~0u
    uint32_t enabled;
    PoseidonOff hash1Old, hash1New;  // t = 4
    N2BStrictOff n2bOld, n2bNew;
    uint32_t isz;      // isZero[i]: inv,out pairs, 2*n
    uint32_t levIns;   // levIns[1 .. n-2]   (n-2 signals, index i-1)
    uint32_t xors;     // n
    uint32_t sm;       // SM_N * n
    uint32_t levels;   // LV_SIZE * n
    uint32_t topSel, topAux;
    IsZOff checkOld;
    uint32_t newRoot;
    IsZOff keyEq;
    uint32_t and1, and2;  // keysOk: MultiAND(3) internals
};

// one level of SMTVerifier: switcher.aux, aux[0], aux[1], root, proofHash
enum { VL_SW_AUX = 0, VL_AUX0, VL_AUX1, VL_ROOT, VL_HASH, VL_SIZE = VL_HASH + 243 };
enum { VSM_PTLI = 0, VSM_PTLIF, VSM_IOLD, VSM_I0, VSM_N };  // prev_top_lev_ins, .._fnc, st_iold, st_i0
struct SmtVerOff {
    PoseidonOff hash1Old, hash1New;
    N2BStrictOff n2bOld, n2bNew;
    uint32_t isz, levIns, sm, levels;
    IsZOff keyEq;
    uint32_t and_a, and_b, and_c;  // MultiAND(4): ands[0].and1, ands[1].and1, and2
    IsZOff checkRoot;
};

struct DecodeFloatOff { uint32_t pe; uint32_t out; };  // pe[1..4] (4 signals), out

struct DecOff {  // DecodeTx(nLevels)
    uint32_t n2bData, n2bAmount;
    DecodeFloatOff dfAmount;
    uint32_t v2in;        // b2nTxCompressedDataV2.in[0..215] (products with 1-onChain)
    IsZOff toIdxIsZero;
    uint32_t selToIdx_s, selToIdx_out;
    uint32_t n2bFinalToIdx;   // L
    uint32_t l1l2Fee;         // L1L2TxData fee bits (8 products)
    uint32_t n2bToEthAddr, n2bMaxNumBatch;
    PoseidonOff hashSig;      // t = 7
    uint32_t n2bFromEthAddr, n2bLoadAmountF;
    uint32_t l1full;          // L1TxFullData[624] products with onChain
    IsZOff fromIdxIsZero;
    uint32_t outIdx;
    uint32_t idxChecker_en;
    IsZOff idxChecker, chainIDChecker, constSigChecker, maxNumBatchIsZero;
    uint32_t maxNumBatchLt;   // Num2Bits(33)
    // outputs that are linear (stored only when DecodeTx is the main component)
    uint32_t o_fromIdx, o_toIdx, o_tokenID, o_nonce, o_userFee, o_toBjjSign, o_amount, o_sigL2Hash, o_v2, o_l1l2;
};

struct Mux3Off { uint32_t base; };  // s10,a210,a21,a20,a10,a1,a0,out
enum { M3_S10 = 0, M3_A210, M3_A21, M3_A20, M3_A10, M3_A1, M3_A0, M3_OUT, M3_N };
enum { M2_S10 = 0, M2_A10, M2_A1, M2_A0, M2_N };

struct StatesOff {  // RollupTxStates
    uint32_t selFromIdx_s, selFromIdx_out;
    IsZOff toIdxIsZero;
    uint32_t selectAuxToIdx, selToIdx_out;
    IsZOff isToEthAddrAny, checkIsExit, finalFromIdxIsZero, loadAmountIsZero, amountIsZero;
    uint32_t isP1Insert, P1_fnc0, P1_fnc1, mux1;   // mux1: M2_N
    uint32_t isP2Insert, P2_fnc0, P2_fnc1, mux2;
    uint32_t verifySignEnabled, tmpCheckToEthAddr, tmpCheckToBjj, checkToEthAddr, checkToBjj;
    uint32_t onChainNotCreateAccount, shouldCheckEthAddr;
    IsZOff checkFromEthAddr;
    uint32_t applyNullifierEthAddr;
    IsZOff checkTokenID1;
    uint32_t applyNullifierTokenID1, shouldCheckTokenID2_0, shouldCheckTokenID2_1;
    IsZOff checkTokenID2;
    uint32_t applyNullifierTokenID2, nullifyLoadAmount, applyCheckTokenID1ToAmount, nullifyAmount_0, nullifyAmount;
};

struct ComputeFeeOff {
    uint32_t applyFee, n2bFeeSel, muxS;   // muxS: 8 products
    uint32_t mux1;                        // 16 x (s10,s20,s21,s210,out)
    uint32_t mux2;                        // mux[16]: s10,s20,s21,s210, 14 a-terms, out  (19)
    uint32_t feeOutNotShifted, applyShift, bits /*253*/, feeOut;
};
enum { MX4_S10 = 0, MX4_S20, MX4_S21, MX4_S210, MX4_OUT_C, MX4C_N };  // constant-input Mux4
enum { MX4V_A3210 = 4, MX4V_A321, MX4V_A320, MX4V_A310, MX4V_A32, MX4V_A31, MX4V_A30, MX4V_A210, MX4V_A21, MX4V_A20, MX4V_A10, MX4V_A2, MX4V_A1, MX4V_A0, MX4V_OUT, MX4V_N };

struct BalUpdOff {
    ComputeFeeOff fee;
    uint32_t effLoad1, effLoad2, effAmt1, effAmt2, n2bSender /*193*/, effAmt3;
    IsZOff effAmtIsZero;
    uint32_t isAmountNullified;
};

// segment of EscalarMulAny: e2m.out[2], bits[nbits] x 10, m2e.out[2], eadder(6), lastSel.out[2]
enum { BIT_DBL_X1_2 = 0, BIT_DBL_LAMDA, BIT_DBL_OUT0, BIT_DBL_OUT1, BIT_ADD_LAMDA, BIT_ADD_OUT0, BIT_ADD_OUT1, BIT_SEL_OUT0, BIT_SEL_OUT1, BIT_N };
struct SegAnyOff { uint32_t e2m, bits, m2e; BabyAddOff eadder; uint32_t lastSel; int nbits; };
// window of EscalarMulFix: mux.s10, mux.out[2], adder.lamda, adder.out[2]
enum { WIN_S10 = 0, WIN_MUX0, WIN_MUX1, WIN_ADD_LAMDA, WIN_ADD_OUT0, WIN_ADD_OUT1, WIN_N };
struct SegFixOff { uint32_t windows, m2e; BabyAddOff cAdd; int nwin; };

struct EddsaOff {
    uint32_t signSignature, aySignature;
    // AySign2Ax / Bits2Point_Strict
    uint32_t ax_n2bAy; CompConstOff ax_aliasY; uint32_t ax_x, ax_x2, ax_y2, ax_n2bX; CompConstOff ax_aliasX, ax_signCalc;
    // EdDSAPoseidonVerifier
    uint32_t snum2bits; CompConstOff sCmp;
    PoseidonOff hash;  // t = 6
    N2BStrictOff h2bits;
    BabyAddOff dbl1, dbl2, dbl3;
    IsZOff isZero;
    IsZOff zeropoint;
    uint32_t seg0p;    // segments[0].p[0..1]
    SegAnyOff seg[2];
    uint32_t dblr;     // doublers[0]: x1_2, lamda, out0, out1
    uint32_t m2e0;     // m2e[0].out[0..1]
    BabyAddOff adders0;
    uint32_t anyOut;   // mulAny.out[0..1]
    BabyAddOff addRight;
    SegFixOff fseg[2];
    BabyAddOff fadders0;
    IsZOff eqCheckX, eqCheckY;
};

struct RtxOff {  // RollupTx(nLevels, maxFeeTx)
    uint32_t n2bLoadAmountF; DecodeFloatOff dfLoadAmount;
    StatesOff st;
    uint32_t rq_n2b; Mux3Off rq_mux[3];
    IsZOff nonceChecker, checkToEthAddr; uint32_t checkToEthAddr_en;
    IsZOff toBjjAyChecker, toBjjSignChecker, checkTokenID1, checkTokenID2; uint32_t checkTokenID2_en;
    IsZOff checkTokenID1L1, fromEthAddrChecker;
    PoseidonOff oldSt1Hash, oldSt2Hash;     // t = 5
    uint32_t mux16;                          // s1Balance,s1Sign,s1Ay,s1Nonce,s1EthAddr,s1TokenID,s1OldKey,s1OldValue, s2...(8)
    EddsaOff ed;
    BalUpdOff bu;
    uint32_t feeAcc;                         // 5 per step: isz.inv, isz.out, isSelectedOut, mux.s, mux.out
    PoseidonOff newSt1Hash, newSt2Hash;
    SmtProcOff p1, p2;
    uint32_t s3, s4, s5;
    // main-level extras when embedded in RollupMain: L1L2 amount bits masked by isAmountNullified
    uint32_t main_l1l2amt;                   // 40 products (RollupMain only, else ~0u)
    // outputs stored only when RollupTx is the main component
    uint32_t o_accFeeOut;                    // F (linear aliases of feeAcc mux outs) -- main only
};
enum { MX_S1BALANCE = 0, MX_S1SIGN, MX_S1AY, MX_S1NONCE, MX_S1ETHADDR, MX_S1TOKENID, MX_S1OLDKEY, MX_S1OLDVALUE,
       MX_S2BALANCE, MX_S2SIGN, MX_S2AY, MX_S2NONCE, MX_S2ETHADDR, MX_S2TOKENID, MX_S2OLDKEY, MX_S2OLDVALUE, MX_N };
enum { FA_ISZ_INV = 0, FA_ISZ_OUT, FA_SELOUT, FA_MUX_S, FA_MUX_OUT, FA_N };

struct FeeTxOff {
    IsZOff feeIdxIsZero, tokenIDChecker;
    PoseidonOff oldHash, newHash;
    SmtProcOff p;
    uint32_t o_newStateRoot;   // main only
};

// per-unit inputs of RollupTx (standalone) / per-tx inputs of RollupMain live in the same
// section as ordinary signals; these structs give their signal offsets.
struct RtxInOff {
    uint32_t feePlanTokens, accFeeIn;  // F each
    uint32_t futureV2, pastV2, futureToEthAddr, pastToEthAddr, futureToBjjAy, pastToBjjAy;  // 3/4
    uint32_t fromIdx, auxFromIdx, toIdx, auxToIdx, toBjjAy, toBjjSign, toEthAddr, amount, tokenID, nonce, userFee, rqOffset,
        onChain, newAccount, rqTxCompressedDataV2, rqToEthAddr, rqToBjjAy, sigL2Hash, s, r8x, r8y, fromEthAddr,
        fromBjjCompressed /*256*/, loadAmountF, tokenID1, nonce1, sign1, balance1, ay1, ethAddr1, siblings1 /*L+1*/, isOld0_1,
        oldKey1, oldValue1, tokenID2, nonce2, sign2, balance2, newExit, ay2, ethAddr2, siblings2, isOld0_2, oldKey2, oldValue2,
        oldStateRoot, oldExitRoot;
    uint32_t o_isAmountNullified, o_newStateRoot, o_newExitRoot;  // outputs (standalone)
};
struct MainTxInOff {  // per-tx inputs of RollupMain (unit = tx)
    uint32_t imOnChain, imOutIdx, imStateRoot, imExitRoot, imAccFeeOut /*F*/;   // valid for units < nTx-1
    uint32_t txCompressedData, amountF, txCompressedDataV2, fromIdx, auxFromIdx, toIdx, auxToIdx, toBjjAy, toEthAddr,
        maxNumBatch, onChain, newAccount, rqOffset, rqTxCompressedDataV2, rqToEthAddr, rqToBjjAy, s, r8x, r8y, loadAmountF,
        fromEthAddr, fromBjjCompressed, tokenID1, nonce1, sign1, balance1, ay1, ethAddr1, siblings1, isOld0_1, oldKey1,
        oldValue1, tokenID2, nonce2, sign2, balance2, ay2, ethAddr2, siblings2, newExit, isOld0_2, oldKey2, oldValue2;
};
struct MainFeeInOff {  // per-fee-tx inputs of RollupMain (unit = fee index)
    uint32_t feeIdxs, feePlanTokens, imStateRootFee /* units < F-1 */, imFinalAccFee, tokenID3, nonce3, sign3, balance3, ay3,
        ethAddr3, siblings3;
};
struct MainGlobOff {
    uint32_t one, hashGlobalInputs, oldLastIdx, oldStateRoot, globalChainID, currentNumBatch, imInitStateRootFee;
};
struct DecInOff {
    uint32_t previousOnChain, txCompressedData, maxNumBatch, amountF, toEthAddr, toBjjAy, rqTxCompressedDataV2, rqToEthAddr,
        rqToBjjAy, fromEthAddr, fromBjjCompressed, loadAmountF, globalChainID, currentNumBatch, onChain, newAccount, auxFromIdx,
        auxToIdx, inIdx;
};
struct FeeTxInOff { uint32_t oldStateRoot, feePlanToken, feeIdx, accFee, tokenID, nonce, sign, balance, ay, ethAddr, siblings; };
// SMTProcessor(n) / SMTVerifier(n) as main: input offsets (the gadgets' own signals are SmtProcOff / SmtVerOff)
struct SmtProcInOff { uint32_t one, oldRoot, siblings, oldKey, oldValue, isOld0, newKey, newValue, fnc; };
struct SmtVerInOff { uint32_t one, enabled, root, siblings, oldKey, oldValue, isOld0, key, value, fnc; };
struct HashStateOff { uint32_t one, out, tokenID, nonce, sign, balance, ay, ethAddr; PoseidonOff hash; };

// SHA-256 bit-level witness (circomlib sha256): per block, see sha256 section in DESIGN.md
struct Sha256Off { uint32_t blocks; uint32_t block_size; int nblocks; };

struct HashInputsOff {
    uint32_t n2bOldLastIdx, n2bNewLastIdx, n2bOldStateRoot, n2bNewStateRoot, n2bNewExitRoot, n2bFee /*F x 48*/, n2bChainID,
        n2bCurrentNumBatch;
    Sha256Off sha;
    uint32_t out;  // hashInputsOut (main output when HashInputs is main)
    // inputs when HashInputs is the main component
    uint32_t i_oldLastIdx, i_newLastIdx, i_oldStateRoot, i_newStateRoot, i_newExitRoot, i_L1TxsFullData, i_L1L2TxsData,
        i_feeTxsData, i_globalChainID, i_currentNumBatch, one;
    uint64_t totalBits;
};

struct WithdrawOff {
    uint32_t one, hashGlobalInputs, rootExit, ethAddr, tokenID, balance, idx, sign, ay, siblingsState;
    PoseidonOff accountState;
    SmtVerOff ver;
    uint32_t n2bRootExit, n2bEthAddr, n2bTokenID, n2bBalance, n2bIdx;
    Sha256Off sha;
};

// ------------------------------------------------------------------------------------------------
// blocks, sections, symbols

enum BlockKind : uint8_t { BK_PLAIN = 0, BK_POSEIDON = 1, BK_SHA = 2 };

struct Block {
    std::string name;   // may contain "{u}" (unit index); "[k]" is appended when count > 1
    uint32_t off = 0;   // first signal of the block inside the section
    uint32_t count = 1;
    uint8_t kind = BK_PLAIN;
    uint8_t t = 0;      // Poseidon width for BK_POSEIDON
    int32_t max_units = -1;  // units for which the block is defined (-1 = all); e.g. im* arrays have nTx-1
    bool scalar_array = false;  // name gets "[k]" even if count == 1
    // A block that holds a SLICE of one of the reference's arrays keeps circom's own index: element k of unit u is
    // name[idx0 + u * ustride + k] (ustride != 0: the array belongs to a component above the units and the name has no "{u}",
    // e.g. hasherInputs.L1L2TxsData[i * bitsL1L2TxData + j] of reference src/rollup-main.circom:451-462).
    uint32_t idx0 = 0, ustride = 0;
};

struct InputDesc {
    std::string name;   // main input signal name, e.g. "siblings1"
    int section = 0;
    uint32_t off = 0;   // signal offset of element [.][0]
    uint32_t inner = 1; // elements per unit
    uint32_t outer = 1; // units per instance that carry it (nTx, nTx-1, maxFeeTx, 1)
    uint32_t ebytes = 32; // bytes per element in the packed bulk-upload format (hz_inputs_upload): 32, or 1 for bit-valued signals
};

struct Section {
    std::string tag;
    uint64_t base = 0;      // first element in the physical buffer
    uint64_t vbase = 0;     // first element in the per-instance (virtual) witness
    uint32_t upi = 1;       // units per instance (transactions, fee slots, 1)
    uint32_t n_units = 1;   // upi * n_instances: unit index = instance * upi + local unit
    uint32_t n_sigs = 0;
    std::vector<Block> blocks;

    uint32_t add(const std::string& name, uint32_t count = 1, int32_t max_units = -1, bool scalar_array = false) {
        Block b;
        b.name = name;
        b.off = n_sigs;
        b.count = count;
        b.max_units = max_units;
        b.scalar_array = scalar_array;
        blocks.push_back(b);
        n_sigs += count;
        return b.off;
    }
    // `count` elements that are name[idx0 ...] of the reference (see Block::idx0)
    uint32_t add_slice(const std::string& name, uint32_t count, uint32_t idx0, uint32_t ustride = 0) {
        const uint32_t off = add(name, count, -1, true);
        blocks.back().idx0 = idx0;
        blocks.back().ustride = ustride;
        return off;
    }
    PoseidonOff add_poseidon(const std::string& name, int t) {
        Block b;
        b.name = name;
        b.off = n_sigs;
        b.count = (uint32_t)poseidon_nsig(t);
        b.kind = BK_POSEIDON;
        b.t = (uint8_t)t;
        blocks.push_back(b);
        n_sigs += b.count;
        return b.off;
    }
    // `nblocks` SHA-256 compression blocks: name = "<...>.sha256compression", symbols name[b] + sha_signame(j)
    uint32_t add_sha(const std::string& name, uint32_t nblocks, uint32_t block_sigs) {
        const uint32_t off = add(name, nblocks * block_sigs);
        blocks.back().kind = BK_SHA;
        return off;
    }
    IsZOff add_isz(const std::string& name) {
        uint32_t o = add(name + ".inv");
        add(name + ".out");
        return o;
    }
    CompConstOff add_cc(const std::string& name) {
        CompConstOff c;
        c.parts = add(name + ".parts", 127);
        c.bits = add(name + ".num2bits.out", 135);
        return c;
    }
    N2BStrictOff add_n2bs(const std::string& name) {
        N2BStrictOff o;
        o.bits = add(name + ".n2b.out", 254);
        o.cc = add_cc(name + ".aliasCheck.compConstant");
        return o;
    }
    BabyAddOff add_babyadd(const std::string& name) {
        uint32_t o = add(name + ".beta");
        add(name + ".gamma");
        add(name + ".delta");
        add(name + ".tau");
        add(name + ".xout");
        add(name + ".yout");
        return o;
    }
    uint64_t size() const { return (uint64_t)n_sigs * n_units; }
};

inline std::string ssub(const std::string& pat, const std::string& key, const std::string& val) {
    std::string r = pat;
    size_t p;
    while ((p = r.find(key)) != std::string::npos) r.replace(p, key.size(), val);
    return r;
}
inline std::string istr(long long v) { char b[32]; snprintf(b, sizeof b, "%lld", v); return b; }

// name of Poseidon signal j (0..3*nsbox) relative to the Poseidon component
inline std::string poseidon_signame(int t, int j) {
    const int k = j / 3, s = j % 3;
    static const char* sn[3] = {"in2", "in4", "out"};
    const int rp = poseidon_rp(t);
    char b[64];
    if (k < 4 * t) snprintf(b, sizeof b, ".sigmaF[%d][%d].%s", k / t, k % t, sn[s]);
    else if (k < 4 * t + rp) snprintf(b, sizeof b, ".sigmaP[%d].%s", k - 4 * t, sn[s]);
    else snprintf(b, sizeof b, ".sigmaF[%d][%d].%s", 4 + (k - 4 * t - rp) / t, (k - 4 * t - rp) % t, sn[s]);
    return b;
}

// Names of the stored signals of one SHA-256 block, relative to `sha256compression[b]` (circomlib 0.5.2 sha256/sha256compression.circom
// and the templates it instantiates: SigmaPlus = {sigma0, sigma1: SmallSigma -> xor3: Xor3(32) with mid / out, sum: BinSum(32, 4)},
// T1 = {bigsigma1: BigSigma -> xor3, ch: Ch_t(32), sum: BinSum(32, 5)}, T2 = {bigsigma0: BigSigma -> xor3, maj: Maj_t(32) with mid / out,
// sum: BinSum(32, 2)}, suma / sume / fsum: BinSum(32, 2)). The products and the sums' hinted bits are what a block stores (in the order
// of SHA_SCHED_W / SHA_ROUND_W below); everything else inside the component is a wire of one of them.
struct ShaPart { const char* name; int width; };
inline const ShaPart* sha_sched_parts() {
    static const ShaPart p[] = {{"sigma0.xor3.mid", 32}, {"sigma0.xor3.out", 32}, {"sigma1.xor3.mid", 32}, {"sigma1.xor3.out", 32}, {"sum.out", 34}, {nullptr, 0}};
    return p;
}
struct ShaRoundPart { const char* comp; const char* name; int width; };
inline const ShaRoundPart* sha_round_parts() {
    static const ShaRoundPart p[] = {{"t1", "bigsigma1.xor3.mid", 32}, {"t1", "bigsigma1.xor3.out", 32}, {"t1", "ch.out", 32}, {"t1", "sum.out", 35},
                                     {"t2", "bigsigma0.xor3.mid", 32}, {"t2", "bigsigma0.xor3.out", 32}, {"t2", "maj.mid", 32}, {"t2", "maj.out", 32}, {"t2", "sum.out", 33},
                                     {"sume", "out", 33}, {"suma", "out", 33}, {nullptr, nullptr, 0}};
    return p;
}
inline std::string sha_signame(uint32_t j) {
    char b[96];
    if (j < 48u * 162u) {
        uint32_t t = j / 162u, r = j % 162u;
        for (const ShaPart* p = sha_sched_parts(); p->name; p++) {
            if (r < (uint32_t)p->width) { snprintf(b, sizeof b, ".sigmaPlus[%u].%s[%u]", t, p->name, r); return b; }
            r -= (uint32_t)p->width;
        }
    }
    j -= 48u * 162u;
    if (j < 64u * 358u) {
        uint32_t t = j / 358u, r = j % 358u;
        for (const ShaRoundPart* p = sha_round_parts(); p->name; p++) {
            if (r < (uint32_t)p->width) { snprintf(b, sizeof b, ".%s[%u].%s[%u]", p->comp, t, p->name, r); return b; }
            r -= (uint32_t)p->width;
        }
    }
    j -= 64u * 358u;
    snprintf(b, sizeof b, ".fsum[%u].out[%u]", j / 33u, j % 33u);
    return b;
}
// the inverse: ".sigmaPlus[3].sigma1.xor3.out[7]" -> index inside the block, or -1
inline long sha_sigindex(const std::string& rest) {
    unsigned t = 0, k = 0;
    char comp[16], name[40];
    int used = 0;
    if (sscanf(rest.c_str(), ".%15[a-zA-Z0-9][%u].%39[a-zA-Z0-9.][%u]%n", comp, &t, name, &k, &used) != 4 || (size_t)used != rest.size()) return -1;
    const std::string c = comp, n = name;
    if (c == "sigmaPlus" && t < 48) {
        uint32_t r = 0;
        for (const ShaPart* p = sha_sched_parts(); p->name; p++) {
            if (n == p->name) return k < (unsigned)p->width ? (long)(t * 162u + r + k) : -1;
            r += (uint32_t)p->width;
        }
        return -1;
    }
    if (c == "fsum" && t < 8 && n == "out") return k < 33 ? (long)(48u * 162u + 64u * 358u + t * 33u + k) : -1;
    if (t >= 64) return -1;
    uint32_t r = 0;
    for (const ShaRoundPart* p = sha_round_parts(); p->name; p++) {
        if (c == p->comp && n == p->name) return k < (unsigned)p->width ? (long)(48u * 162u + t * 358u + r + k) : -1;
        r += (uint32_t)p->width;
    }
    return -1;
}

// the inverse of poseidon_signame: ".sigmaF[r][j].in2" / ".sigmaP[k].out" -> index of the signal inside the component, or -1
inline int poseidon_sigindex(int t, const std::string& suffix) {
    const int rp = poseidon_rp(t);
    const char* p = suffix.c_str();
    const bool full = suffix.compare(0, 8, ".sigmaF[") == 0;
    if (!full && suffix.compare(0, 8, ".sigmaP[") != 0) return -1;
    p += 8;
    char* e = nullptr;
    const long a = strtol(p, &e, 10);
    if (e == p || *e != ']' || a < 0) return -1;
    p = e + 1;
    long k;
    if (full) {
        if (*p != '[') return -1;
        const long j = strtol(p + 1, &e, 10);
        if (e == p + 1 || *e != ']' || j < 0 || j >= t || a >= 8) return -1;
        p = e + 1;
        k = a < 4 ? a * t + j : 4 * t + rp + (a - 4) * t + j;
    } else {
        if (a >= rp) return -1;
        k = 4 * t + a;
    }
    const int sidx = !strcmp(p, ".in2") ? 0 : !strcmp(p, ".in4") ? 1 : !strcmp(p, ".out") ? 2 : -1;
    return sidx < 0 ? -1 : (int)(3 * k + sidx);
}

struct Layout;
inline void build_layout(const Params& p, Layout& out);  // defined below

// inputs / outputs of a gadget main, in the declaration order of its template (the gadget's internal signals reuse the offset
// structs of RtxOff: StatesOff, BalUpdOff, ComputeFeeOff, DecodeFloatOff, the rq muxes, the fee-accumulator chain)
struct GadIO { uint32_t in[24]; uint32_t out[16]; uint32_t mux; /* Mux256: first of 17 variable-input Mux4 blocks */ };

struct Layout {
    Params p;
    std::vector<Section> sections;
    std::vector<InputDesc> inputs;
    std::vector<std::pair<std::string, uint64_t>> outputs;  // main outputs: name, virtual index (first element)
    uint64_t total = 0;        // elements in the physical buffer
    uint64_t per_instance = 0; // virtual witness length of one instance
    uint32_t n_inst = 1;       // instances evaluated together (independent batches / witnesses)

    // offsets
    MainGlobOff g{};
    MainTxInOff mi{};
    MainFeeInOff fi{};
    DecOff dec{};
    DecInOff deci{};
    RtxOff rtx{};
    RtxInOff rtxi{};
    FeeTxOff fee{};
    FeeTxInOff feei{};
    HashStateOff hs{};
    HashInputsOff hi{};
    WithdrawOff wd{};
    GadIO gad{};
    SmtProcOff smtp{};
    SmtProcInOff smtpi{};
    SmtVerOff smtv{};
    SmtVerInOff smtvi{};
    int sec_tx = -1, sec_fee = -1, sec_glob = -1, sec_hi = -1;  // section indices

    // physical index of (section, sig, global unit = instance * upi + local unit)
    uint64_t phys(int sec, uint32_t sig, uint32_t unit) const {
        const Section& s = sections[sec];
        return s.base + (uint64_t)sig * s.n_units + unit;
    }

    // ---- symbol enumeration: calls f(name, section, sig, unit) for every stored signal ----------
    template <class Fn>
    void for_each_symbol(Fn f, bool expand_poseidon = true) const {
        for (size_t si = 0; si < sections.size(); si++) {
            const Section& s = sections[si];
            const uint32_t nu = s.upi;  // names describe ONE instance; the instance is not part of a name
            for (const Block& b : s.blocks) {
                const uint32_t units = (b.max_units >= 0) ? (uint32_t)b.max_units : nu;
                for (uint32_t u = 0; u < units; u++) {
                    const std::string base = ssub(b.name, "{u}", istr(u));
                    if (b.kind == BK_POSEIDON) {
                        if (!expand_poseidon) { f(base + ".sigma*", (int)si, b.off, u); continue; }
                        for (uint32_t j = 0; j < b.count; j++) f(base + poseidon_signame(b.t, (int)j), (int)si, b.off + j, u);
                    } else if (b.kind == BK_SHA) {
                        const uint32_t per = 48u * 162u + 64u * 358u + 8u * 33u;
                        for (uint32_t k = 0; k < b.count; k++) f(base + "[" + istr(k / per) + "]" + sha_signame(k % per), (int)si, b.off + k, u);
                    } else if (b.count == 1 && !b.scalar_array) {
                        f(base, (int)si, b.off, u);
                    } else {
                        for (uint32_t k = 0; k < b.count; k++) f(base + "[" + istr(b.idx0 + u * b.ustride + k) + "]", (int)si, b.off + k, u);
                    }
                }
            }
        }
    }
    // virtual (per-instance) index of (section, sig, local unit)
    uint64_t virt(int sec, uint32_t sig, uint32_t unit) const {
        const Section& s = sections[sec];
        return s.vbase + (uint64_t)sig * s.upi + unit;
    }
    // physical index of virtual index v of instance inst
    uint64_t virt_to_phys(uint64_t v, uint32_t inst) const {
        size_t si = sections.size() - 1;
        while (si > 0 && sections[si].vbase > v) si--;
        const Section& s = sections[si];
        const uint64_t rel = v - s.vbase;
        return s.base + (rel / s.upi) * s.n_units + (uint64_t)inst * s.upi + rel % s.upi;
    }

    // name -> virtual index. Accepts names with or without "main.".
    bool lookup(const std::string& name_in, uint64_t* out) const {
        std::string name = name_in;
        if (name.compare(0, 5, "main.") != 0) name = "main." + name;
        if (index_.empty()) const_cast<Layout*>(this)->build_index();
        {   // a signal inside a SHA-256 block: <component>.sha256compression[b].<rest>
            static const char mark[] = ".sha256compression[";
            const size_t sp = name.find(mark);
            if (sp != std::string::npos) {
                auto it = index_.find(name.substr(0, sp + sizeof mark - 2));
                const size_t rb = name.find(']', sp);
                if (it == index_.end() || rb == std::string::npos) return false;
                const Block& b = sections[it->second.first].blocks[it->second.second];
                const uint32_t per = 48u * 162u + 64u * 358u + 8u * 33u;
                const long long blk = atoll(name.c_str() + sp + sizeof mark - 1);
                const long j = sha_sigindex(name.substr(rb + 1));
                if (b.kind != BK_SHA || j < 0 || blk < 0 || (uint64_t)blk * per + (uint64_t)j >= b.count) return false;
                *out = virt(it->second.first, b.off + (uint32_t)blk * per + (uint32_t)j, 0);
                return true;
            }
        }
        // split trailing Poseidon suffix
        std::string key = name;
        long long unit = 0, k = 0;
        int pj = -1;
        size_t sp = key.find(".sigmaF[");
        if (sp == std::string::npos) sp = key.find(".sigmaP[");
        std::string suffix;
        if (sp != std::string::npos) { suffix = key.substr(sp); key = key.substr(0, sp); }
        // try: as is; strip trailing [k]; replace first [n] by [{u}] (both variants)
        for (int variant = 0; variant < 4; variant++) {
            std::string cand = key;
            unit = 0; k = 0;
            bool has_k = false;
            if (variant & 1) {
                if (cand.empty() || cand.back() != ']') continue;
                size_t lb = cand.rfind('[');
                if (lb == std::string::npos) continue;
                k = atoll(cand.c_str() + lb + 1);
                cand = cand.substr(0, lb);
                has_k = true;
            }
            if (variant & 2) {
                size_t lb = cand.find('[');
                if (lb == std::string::npos) continue;
                size_t rb = cand.find(']', lb);
                if (rb == std::string::npos) continue;
                unit = atoll(cand.c_str() + lb + 1);
                cand = cand.substr(0, lb + 1) + "{u}" + cand.substr(rb);
            }
            auto it = index_.find(cand);
            if (it == index_.end()) continue;
            const Section& s = sections[it->second.first];
            const Block& b = s.blocks[it->second.second];
            const uint32_t units = (b.max_units >= 0) ? (uint32_t)b.max_units : s.upi;
            if (unit < 0 || (uint64_t)unit >= units) return false;
            if (b.kind == BK_POSEIDON) {
                if (suffix.empty() || has_k) return false;
                pj = poseidon_sigindex(b.t, suffix);
                if (pj < 0 || (uint32_t)pj >= b.count) return false;
                *out = virt(it->second.first, b.off + (uint32_t)pj, (uint32_t)unit);
                return true;
            }
            if (!suffix.empty()) return false;
            const bool is_array = b.count > 1 || b.scalar_array;
            if (is_array != has_k) continue;
            if (b.ustride) { unit = k / b.ustride; k %= b.ustride; if ((uint64_t)unit >= units) return false; }
            k -= b.idx0;
            if (k < 0 || (uint64_t)k >= b.count) return false;
            *out = virt(it->second.first, b.off + (uint32_t)k, (uint32_t)unit);
            return true;
        }
        return false;
    }

    const InputDesc* find_input(const std::string& name) const {
        for (const InputDesc& d : inputs)
            if (d.name == name) return &d;
        return nullptr;
    }

   private:
    std::unordered_map<std::string, std::pair<int, int>> index_;
    void build_index() {
        for (size_t si = 0; si < sections.size(); si++)
            for (size_t bi = 0; bi < sections[si].blocks.size(); bi++) index_[sections[si].blocks[bi].name] = {(int)si, (int)bi};
    }
};

// ------------------------------------------------------------------------------------------------
// layout builders. `pre` is the circom component prefix, e.g. "main.rollupTx[{u}]." or "main.".

inline void lay_smtproc(Section& s, const std::string& pre, int n, bool fnc_products, SmtProcOff& o) {
    o.fnc = fnc_products ? s.add(pre + "fnc", 2) : ~0u;
    o.enabled = s.add(pre + "enabled");
    o.hash1Old = s.add_poseidon(pre + "hash1Old.h", 4);
    o.hash1New = s.add_poseidon(pre + "hash1New.h", 4);
    o.n2bOld = s.add_n2bs(pre + "n2bOld");
    o.n2bNew = s.add_n2bs(pre + "n2bNew");
    o.isz = s.n_sigs;
    for (int i = 0; i < n; i++) s.add_isz(pre + "smtLevIns.isZero[" + istr(i) + "]");
    o.levIns = s.n_sigs;
    for (int i = 1; i <= n - 2; i++) s.add(pre + "smtLevIns.levIns[" + istr(i) + "]");
    o.xors = s.n_sigs;
    for (int i = 0; i < n; i++) s.add(pre + "xors[" + istr(i) + "].out");
    o.sm = s.n_sigs;
    for (int i = 0; i < n; i++) {
        const std::string q = pre + "sm[" + istr(i) + "].";
        s.add(q + "aux1"); s.add(q + "aux2"); s.add(q + "st_old0"); s.add(q + "st_new1"); s.add(q + "st_bot");
    }
    o.levels = s.n_sigs;
    for (int i = 0; i < n; i++) {
        const std::string q = pre + "levels[" + istr(i) + "].";
        s.add(q + "oldSwitcher.aux"); s.add(q + "aux[0]"); s.add(q + "oldRoot");
        s.add_poseidon(q + "oldProofHash.h", 3);
        s.add(q + "newSwitcher.aux"); s.add(q + "aux[1]"); s.add(q + "aux[2]"); s.add(q + "aux[3]");
        s.add(q + "newSwitcher.L"); s.add(q + "newSwitcher.R"); s.add(q + "newRoot");
        s.add_poseidon(q + "newProofHash.h", 3);
    }
    o.topSel = s.add(pre + "topSwitcher.sel");
    o.topAux = s.add(pre + "topSwitcher.aux");
    o.checkOld = s.add_isz(pre + "checkOldInput.isz");
    o.newRoot = s.add(pre + "newRoot");
    o.keyEq = s.add_isz(pre + "areKeyEquals.isz");
    o.and1 = s.add(pre + "keysOk.ands[1].and1.out");
    o.and2 = s.add(pre + "keysOk.and2.out");
}

inline void lay_smtver(Section& s, const std::string& pre, int n, SmtVerOff& o) {
    o.hash1Old = s.add_poseidon(pre + "hash1Old.h", 4);
    o.hash1New = s.add_poseidon(pre + "hash1New.h", 4);
    o.n2bOld = s.add_n2bs(pre + "n2bOld");
    o.n2bNew = s.add_n2bs(pre + "n2bNew");
    o.isz = s.n_sigs;
    for (int i = 0; i < n; i++) s.add_isz(pre + "smtLevIns.isZero[" + istr(i) + "]");
    o.levIns = s.n_sigs;
    for (int i = 1; i <= n - 2; i++) s.add(pre + "smtLevIns.levIns[" + istr(i) + "]");
    o.sm = s.n_sigs;
    for (int i = 0; i < n; i++) {
        const std::string q = pre + "sm[" + istr(i) + "].";
        s.add(q + "prev_top_lev_ins"); s.add(q + "prev_top_lev_ins_fnc"); s.add(q + "st_iold"); s.add(q + "st_i0");
    }
    o.levels = s.n_sigs;
    for (int i = 0; i < n; i++) {
        const std::string q = pre + "levels[" + istr(i) + "].";
        s.add(q + "switcher.aux"); s.add(q + "aux[0]"); s.add(q + "aux[1]"); s.add(q + "root");
        s.add_poseidon(q + "proofHash.h", 3);
    }
    o.keyEq = s.add_isz(pre + "areKeyEquals.isz");
    o.and_a = s.add(pre + "keysOk.ands[0].and1.out");
    o.and_b = s.add(pre + "keysOk.ands[1].and1.out");
    o.and_c = s.add(pre + "keysOk.and2.out");
    o.checkRoot = s.add_isz(pre + "checkRoot.isz");
}

inline DecodeFloatOff lay_decodefloat(Section& s, const std::string& pre) {
    DecodeFloatOff o;
    o.pe = s.n_sigs;
    for (int i = 1; i <= 4; i++) s.add(pre + "pe[" + istr(i) + "]");
    o.out = s.add(pre + "out");
    return o;
}

inline void lay_decode(Section& s, const std::string& pre, int L, bool is_main, DecOff& o) {
    o.n2bData = s.add(pre + "n2bData.out", 225);
    o.n2bAmount = s.add(pre + "n2bAmount.out", 40);
    o.dfAmount = lay_decodefloat(s, pre + "dfAmount.");
    o.v2in = s.add(pre + "b2nTxCompressedDataV2.in", 216);
    o.toIdxIsZero = s.add_isz(pre + "toIdxIsZero");
    o.selToIdx_s = s.add(pre + "selectToIdx.s");
    o.selToIdx_out = s.add(pre + "selectToIdx.out");
    o.n2bFinalToIdx = s.add(pre + "n2bFinalToIdx.out", (uint32_t)L);
    // the fee bits of the data-availability output are products, n2bData.out[216 + i] * (1 - onChain) (src/decode-tx.circom:246):
    // L1L2TxData[2L+40 .. 2L+47]. As `component main` the whole output array is stored as well (below) and owns the name; '#' marks
    // a name of this layout that is not a circom label.
    o.l1l2Fee = is_main ? s.add(pre + "L1L2TxData#fee", 8) : s.add_slice(pre + "L1L2TxData", 8, (uint32_t)(2 * L + 40));
    o.n2bToEthAddr = s.add(pre + "n2bToEthAddr.out", 160);
    o.n2bMaxNumBatch = s.add(pre + "n2bMaxNumBatch.out", 32);
    o.hashSig = s.add_poseidon(pre + "hashSig", 7);
    o.n2bFromEthAddr = s.add(pre + "n2bFromEthAddr.out", 160);
    o.n2bLoadAmountF = s.add(pre + "n2bLoadAmountF.out", 40);
    o.l1full = s.add(pre + "L1TxFullData", L1FULL_BITS);
    o.fromIdxIsZero = s.add_isz(pre + "fromIdxIsZero");
    o.outIdx = s.add(pre + "outIdx");
    o.idxChecker_en = s.add(pre + "idxChecker.enabled");
    o.idxChecker = s.add_isz(pre + "idxChecker.isz");
    o.chainIDChecker = s.add_isz(pre + "chainIDChecker.isz");
    o.constSigChecker = s.add_isz(pre + "constSigChecker.isz");
    o.maxNumBatchIsZero = s.add_isz(pre + "maxNumBatchIsZero");
    o.maxNumBatchLt = s.add(pre + "isMaxNumBatchOk.lt.n2b.out", 33);
    if (is_main) {
        o.o_fromIdx = s.add(pre + "fromIdx"); o.o_toIdx = s.add(pre + "toIdx"); o.o_tokenID = s.add(pre + "tokenID");
        o.o_nonce = s.add(pre + "nonce"); o.o_userFee = s.add(pre + "userFee"); o.o_toBjjSign = s.add(pre + "toBjjSign");
        o.o_amount = s.add(pre + "amount"); o.o_sigL2Hash = s.add(pre + "sigL2Hash"); o.o_v2 = s.add(pre + "txCompressedDataV2");
        o.o_l1l2 = s.add(pre + "L1L2TxData", (uint32_t)(2 * L + 48));
    } else {
        o.o_fromIdx = o.o_toIdx = o.o_tokenID = o.o_nonce = o.o_userFee = o.o_toBjjSign = o.o_amount = o.o_sigL2Hash = o.o_v2 = o.o_l1l2 = ~0u;
    }
}

inline void lay_states(Section& s, const std::string& pre, StatesOff& o) {
    o.selFromIdx_s = s.add(pre + "selectFromIdx.s");
    o.selFromIdx_out = s.add(pre + "selectFromIdx.out");
    o.toIdxIsZero = s.add_isz(pre + "toIdxIsZero");
    o.selectAuxToIdx = s.add(pre + "selectAuxToIdx");
    o.selToIdx_out = s.add(pre + "selectToIdx.out");
    o.isToEthAddrAny = s.add_isz(pre + "isToEthAddrAny.isz");
    o.checkIsExit = s.add_isz(pre + "checkIsExit.isz");
    o.finalFromIdxIsZero = s.add_isz(pre + "finalFromIdxIsZero");
    o.loadAmountIsZero = s.add_isz(pre + "loadAmountIsZero");
    o.amountIsZero = s.add_isz(pre + "amountIsZero");
    o.isP1Insert = s.add(pre + "isP1Insert");
    o.P1_fnc0 = s.add(pre + "P1_fnc0");
    o.P1_fnc1 = s.add(pre + "P1_fnc1");
    // Mux2 wraps `component mux = MultiMux2(1)` (circomlib mux2.circom): the products are signals of that inner component
    o.mux1 = s.add(pre + "mux1.mux.s10"); s.add(pre + "mux1.mux.a10[0]"); s.add(pre + "mux1.mux.a1[0]"); s.add(pre + "mux1.mux.a0[0]");
    o.isP2Insert = s.add(pre + "isP2Insert");
    o.P2_fnc0 = s.add(pre + "P2_fnc0");
    o.P2_fnc1 = s.add(pre + "P2_fnc1");
    o.mux2 = s.add(pre + "mux2.mux.s10"); s.add(pre + "mux2.mux.a10[0]"); s.add(pre + "mux2.mux.a1[0]"); s.add(pre + "mux2.mux.a0[0]");
    o.verifySignEnabled = s.add(pre + "verifySignEnabled");
    o.tmpCheckToEthAddr = s.add(pre + "tmpCheckToEthAddr");
    o.tmpCheckToBjj = s.add(pre + "tmpCheckToBjj");
    o.checkToEthAddr = s.add(pre + "checkToEthAddr");
    o.checkToBjj = s.add(pre + "checkToBjj");
    o.onChainNotCreateAccount = s.add(pre + "onChainNotCreateAccount");
    o.shouldCheckEthAddr = s.add(pre + "shouldCheckEthAddr");
    o.checkFromEthAddr = s.add_isz(pre + "checkFromEthAddr.isz");
    o.applyNullifierEthAddr = s.add(pre + "applyNullifierEthAddr");
    o.checkTokenID1 = s.add_isz(pre + "checkTokenID1.isz");
    o.applyNullifierTokenID1 = s.add(pre + "applyNullifierTokenID1");
    o.shouldCheckTokenID2_0 = s.add(pre + "shouldCheckTokenID2_0");
    o.shouldCheckTokenID2_1 = s.add(pre + "shouldCheckTokenID2_1");
    o.checkTokenID2 = s.add_isz(pre + "checkTokenID2.isz");
    o.applyNullifierTokenID2 = s.add(pre + "applyNullifierTokenID2");
    o.nullifyLoadAmount = s.add(pre + "nullifyLoadAmount");
    o.applyCheckTokenID1ToAmount = s.add(pre + "applyCheckTokenID1ToAmount");
    o.nullifyAmount_0 = s.add(pre + "nullifyAmount_0");
    o.nullifyAmount = s.add(pre + "nullifyAmount");
}

// circomlib mux4.circom MultiMux4(1) whose inputs are signals: every product term is a variable (MX4V_* order)
inline uint32_t lay_mux4v(Section& s, const std::string& q) {
    // Mux4 wraps `component mux = MultiMux4(1)` (circomlib mux4.circom): its products are `<q>mux.s10` ... `<q>mux.a0[0]`; the output
    // keeps the outer label (`mux.out[0] ==> out`: one variable in a reducing compile, an alias otherwise -- formats.hip)
    const uint32_t base = s.add(q + "mux.s10"); s.add(q + "mux.s20"); s.add(q + "mux.s21"); s.add(q + "mux.s210");
    static const char* an[14] = {"a3210", "a321", "a320", "a310", "a32", "a31", "a30", "a210", "a21", "a20", "a10", "a2", "a1", "a0"};
    for (int i = 0; i < 14; i++) s.add(q + "mux." + an[i] + "[0]");
    s.add(q + "out");
    return base;
}

inline void lay_computefee(Section& s, const std::string& pre, ComputeFeeOff& o) {
    o.applyFee = s.add(pre + "applyFee");
    o.n2bFeeSel = s.add(pre + "n2bFeeSel.out", 8);
    o.muxS = s.add(pre + "mux256.s", 8);
    o.mux1 = s.n_sigs;
    for (int i = 0; i < 16; i++) {
        const std::string q = pre + "mux256.mux[" + istr(i) + "].";
        s.add(q + "mux.s10"); s.add(q + "mux.s20"); s.add(q + "mux.s21"); s.add(q + "mux.s210"); s.add(q + "out");
    }
    o.mux2 = lay_mux4v(s, pre + "mux256.mux[16].");
    o.feeOutNotShifted = s.add(pre + "feeOutNotShifted");
    o.applyShift = s.add(pre + "applyShift");
    o.bits = s.add(pre + "bitsFeeOut", 253);
    o.feeOut = s.add(pre + "feeOut");
}

inline void lay_seg_any(Section& s, const std::string& pre, int n, SegAnyOff& o) {
    o.nbits = n - 1;
    o.e2m = s.add(pre + "e2m.out", 2);
    o.bits = s.n_sigs;
    for (int i = 0; i < n - 1; i++) {
        const std::string q = pre + "bits[" + istr(i) + "].";
        s.add(q + "doubler.x1_2"); s.add(q + "doubler.lamda"); s.add(q + "doubler.out", 2);
        s.add(q + "adder.lamda"); s.add(q + "adder.out", 2); s.add(q + "selector.out", 2);
    }
    o.m2e = s.add(pre + "m2e.out", 2);
    o.eadder = s.add_babyadd(pre + "eadder");
    o.lastSel = s.add(pre + "lastSel.out", 2);
}

inline void lay_seg_fix(Section& s, const std::string& pre, int nwin, SegFixOff& o) {
    o.nwin = nwin;
    o.windows = s.n_sigs;
    for (int i = 0; i < nwin; i++) {
        s.add(pre + "windows[" + istr(i) + "].mux.s10");
        s.add(pre + "windows[" + istr(i) + "].mux.out", 2);
        s.add(pre + "adders[" + istr(i) + "].lamda");
        s.add(pre + "adders[" + istr(i) + "].out", 2);
    }
    o.m2e = s.add(pre + "m2e.out", 2);
    o.cAdd = s.add_babyadd(pre + "cAdd");
}

// AySign2Ax (src/lib/utils-bjj.circom:37-58) incl. circomlib pointbits.circom Bits2Point_Strict
inline void lay_ax(Section& s, const std::string& g, EddsaOff& o) {
    o.ax_n2bAy = s.add(g + "n2bAy.out", 254);
    o.ax_aliasY = s.add_cc(g + "b2Point.aliasCheckY.compConstant");
    o.ax_x = s.add(g + "b2Point.out[0]");
    o.ax_x2 = s.add(g + "b2Point.babyCheck.x2");
    o.ax_y2 = s.add(g + "b2Point.babyCheck.y2");
    o.ax_n2bX = s.add(g + "b2Point.n2bX.out", 254);
    o.ax_aliasX = s.add_cc(g + "b2Point.aliasCheckX.compConstant");
    o.ax_signCalc = s.add_cc(g + "b2Point.signCalc");
}

inline void lay_eddsa(Section& s, const std::string& pre, EddsaOff& o) {
    o.signSignature = s.add(pre + "signSignature.out");
    o.aySignature = s.add(pre + "aySignature.out");
    lay_ax(s, pre + "getAx.", o);
    const std::string v = pre + "sigVerifier.";
    o.snum2bits = s.add(v + "snum2bits.out", 253);
    o.sCmp = s.add_cc(v + "compConstant");
    o.hash = s.add_poseidon(v + "hash", 6);
    o.h2bits = s.add_n2bs(v + "h2bits");
    o.dbl1 = s.add_babyadd(v + "dbl1.adder");
    o.dbl2 = s.add_babyadd(v + "dbl2.adder");
    o.dbl3 = s.add_babyadd(v + "dbl3.adder");
    o.isZero = s.add_isz(v + "isZero");
    o.zeropoint = s.add_isz(v + "mulAny.zeropoint");
    o.seg0p = s.add(v + "mulAny.segments[0].p", 2);
    lay_seg_any(s, v + "mulAny.segments[0].", 148, o.seg[0]);
    lay_seg_any(s, v + "mulAny.segments[1].", 106, o.seg[1]);
    o.dblr = s.add(v + "mulAny.doublers[0].x1_2"); s.add(v + "mulAny.doublers[0].lamda"); s.add(v + "mulAny.doublers[0].out", 2);
    o.m2e0 = s.add(v + "mulAny.m2e[0].out", 2);
    o.adders0 = s.add_babyadd(v + "mulAny.adders[0]");
    o.anyOut = s.add(v + "mulAny.out", 2);
    o.addRight = s.add_babyadd(v + "addRight");
    lay_seg_fix(s, v + "mulFix.segments[0].", 82, o.fseg[0]);
    lay_seg_fix(s, v + "mulFix.segments[1].", 3, o.fseg[1]);
    o.fadders0 = s.add_babyadd(v + "mulFix.adders[0]");
    o.eqCheckX = s.add_isz(v + "eqCheckX.isz");
    o.eqCheckY = s.add_isz(v + "eqCheckY.isz");
}

inline void lay_balupd(Section& s, const std::string& b, BalUpdOff& o) {
    lay_computefee(s, b + "computeFee.", o.fee);
    o.effLoad1 = s.add(b + "effectiveLoadAmount1");
    o.effLoad2 = s.add(b + "effectiveLoadAmount2");
    o.effAmt1 = s.add(b + "effectiveAmount1");
    o.effAmt2 = s.add(b + "effectiveAmount2");
    o.n2bSender = s.add(b + "n2bSender.out", 193);
    o.effAmt3 = s.add(b + "effectiveAmount3");
    o.effAmtIsZero = s.add_isz(b + "effectiveAmountIsZero");
    o.isAmountNullified = s.add(b + "isAmountNullified");
}
inline uint32_t lay_feeacc(Section& s, const std::string& pre, int F) {
    const uint32_t first = s.n_sigs;
    for (int i = 0; i < F; i++) {
        const std::string q = pre + "chain[" + istr(i) + "].";
        s.add(q + "isEqual.isz.inv"); s.add(q + "isEqual.isz.out"); s.add(q + "isSelectedOut"); s.add(q + "mux.s"); s.add(q + "mux.out");
    }
    return first;
}
template <class RTX>
inline void lay_rq(Section& s, const std::string& pre, RTX& o) {
    o.rq_n2b = s.add(pre + "n2b.out", 3);
    static const char* mn[3] = {"muxTxCompressedDataV2", "muxToEthAddr", "muxToBjjAy"};
    for (int m = 0; m < 3; m++) {
        const std::string q = pre + mn[m] + ".mux.";
        o.rq_mux[m].base = s.add(q + "s10");
        s.add(q + "a210[0]"); s.add(q + "a21[0]"); s.add(q + "a20[0]"); s.add(q + "a10[0]"); s.add(q + "a1[0]"); s.add(q + "a0[0]"); s.add(q + "out[0]");
    }
}
inline void lay_rtx(Section& s, const std::string& pre, int L, int F, bool in_main, RtxOff& o) {
    o.n2bLoadAmountF = s.add(pre + "n2bloadAmountF.out", 40);
    o.dfLoadAmount = lay_decodefloat(s, pre + "dfLoadAmount.");
    lay_states(s, pre + "states.", o.st);
    lay_rq(s, pre + "rqTxVerifier.", o);
    o.nonceChecker = s.add_isz(pre + "nonceChecker.isz");
    o.checkToEthAddr = s.add_isz(pre + "checkToEthAddr.isz");
    o.checkToEthAddr_en = s.add(pre + "checkToEthAddr.enabled");
    o.toBjjAyChecker = s.add_isz(pre + "toBjjAyChecker.isz");
    o.toBjjSignChecker = s.add_isz(pre + "toBjjSignChecker.isz");
    o.checkTokenID1 = s.add_isz(pre + "checkTokenID1.isz");
    o.checkTokenID2 = s.add_isz(pre + "checkTokenID2.isz");
    o.checkTokenID2_en = s.add(pre + "checkTokenID2.enabled");
    o.checkTokenID1L1 = s.add_isz(pre + "checkTokenID1L1.isz");
    o.fromEthAddrChecker = s.add_isz(pre + "fromEthAddrChecker.isz");
    o.oldSt1Hash = s.add_poseidon(pre + "oldSt1Hash.hash", 5);
    o.oldSt2Hash = s.add_poseidon(pre + "oldSt2Hash.hash", 5);
    static const char* mx[MX_N] = {"s1Balance", "s1Sign", "s1Ay", "s1Nonce", "s1EthAddr", "s1TokenID", "s1OldKey", "s1OldValue",
                                   "s2Balance", "s2Sign", "s2Ay", "s2Nonce", "s2EthAddr", "s2TokenID", "s2OldKey", "s2OldValue"};
    o.mux16 = s.n_sigs;
    for (int i = 0; i < MX_N; i++) s.add(pre + mx[i] + ".out");
    lay_eddsa(s, pre, o.ed);
    lay_balupd(s, pre + "balanceUpdater.", o.bu);
    o.feeAcc = lay_feeacc(s, pre + "feeAccumulator.", F);
    o.newSt1Hash = s.add_poseidon(pre + "newSt1Hash.hash", 5);
    o.newSt2Hash = s.add_poseidon(pre + "newSt2Hash.hash", 5);
    lay_smtproc(s, pre + "processor1.", L + 1, false, o.p1);
    lay_smtproc(s, pre + "processor2.", L + 1, true, o.p2);
    o.s3 = s.add(pre + "s3.out");
    o.s4 = s.add(pre + "s4.out");
    o.s5 = s.add(pre + "s5.out");
    o.main_l1l2amt = ~0u;
    o.o_accFeeOut = ~0u;
    (void)in_main;
}

inline void lay_feetx(Section& s, const std::string& pre, int L, FeeTxOff& o) {
    o.feeIdxIsZero = s.add_isz(pre + "feeIdxIsZero");
    o.tokenIDChecker = s.add_isz(pre + "tokenIDChecker.isz");
    o.oldHash = s.add_poseidon(pre + "oldStFeePck.hash", 5);
    o.newHash = s.add_poseidon(pre + "newStFeePck.hash", 5);
    lay_smtproc(s, pre + "processor.", L + 1, false, o.p);
    o.o_newStateRoot = ~0u;
}

// SHA-256 per-block signal count (see DESIGN.md "SHA-256 witness"):
//   schedule 48 x [ sigma0: mid 32 + out 32, sigma1: mid 32 + out 32, sum out 32 + carry 2 ]
//   rounds   64 x [ bigsigma1: 64, ch: 32, t1 sum: 32+3, bigsigma0: 64, maj: 64, t2 sum: 32+1, sume 32+1, suma 32+1 ]
//   final     8 x [ 32 + 1 ]
enum {
    SHA_SCHED_W = 32 + 32 + 32 + 32 + 34,                        // 162
    SHA_ROUND_W = 64 + 32 + 35 + 64 + 64 + 33 + 33 + 33,         // 358
    SHA_BLOCK_SIGS = 48 * SHA_SCHED_W + 64 * SHA_ROUND_W + 8 * 33  // 30952
};

inline uint64_t hash_inputs_bits(int L, int nTx, int maxL1, int F) {
    return 2 * 48 + 3 * 256 + 16 + 32 + (uint64_t)maxL1 * L1FULL_BITS + (uint64_t)nTx * (2 * L + 48) + (uint64_t)F * L;
}

inline void lay_hashinputs(Section& s, const std::string& pre, int L, int nTx, int maxL1, int F, bool is_main, HashInputsOff& o) {
    o.totalBits = hash_inputs_bits(L, nTx, maxL1, F);
    if (is_main) {
        o.one = s.add("main.one");
        o.out = s.add(pre + "hashInputsOut");
        o.i_oldLastIdx = s.add(pre + "oldLastIdx"); o.i_newLastIdx = s.add(pre + "newLastIdx");
        o.i_oldStateRoot = s.add(pre + "oldStateRoot"); o.i_newStateRoot = s.add(pre + "newStateRoot");
        o.i_newExitRoot = s.add(pre + "newExitRoot");
        o.i_L1TxsFullData = s.add(pre + "L1TxsFullData", (uint32_t)(maxL1 * L1FULL_BITS));
        o.i_L1L2TxsData = s.add(pre + "L1L2TxsData", (uint32_t)(nTx * (2 * L + 48)));
        o.i_feeTxsData = s.add(pre + "feeTxsData", (uint32_t)F, -1, true);
        o.i_globalChainID = s.add(pre + "globalChainID"); o.i_currentNumBatch = s.add(pre + "currentNumBatch");
    } else {
        o.one = o.out = o.i_oldLastIdx = o.i_newLastIdx = o.i_oldStateRoot = o.i_newStateRoot = o.i_newExitRoot = o.i_L1TxsFullData =
            o.i_L1L2TxsData = o.i_feeTxsData = o.i_globalChainID = o.i_currentNumBatch = ~0u;
    }
    o.n2bOldLastIdx = s.add(pre + "n2bOldLastIdx.out", 48);
    o.n2bNewLastIdx = s.add(pre + "n2bNewLastIdx.out", 48);
    o.n2bOldStateRoot = s.add(pre + "n2bOldStateRoot.out", 256);
    o.n2bNewStateRoot = s.add(pre + "n2bNewStateRoot.out", 256);
    o.n2bNewExitRoot = s.add(pre + "n2bNewExitRoot.out", 256);
    o.n2bFee = s.n_sigs;
    for (int i = 0; i < F; i++) s.add(pre + "n2bFeeTxsData[" + istr(i) + "].out", 48);
    o.n2bChainID = s.add(pre + "n2bChainID.out", 16);
    o.n2bCurrentNumBatch = s.add(pre + "n2bCurrentNumBatch.out", 32);
    o.sha.nblocks = (int)((o.totalBits + 64) / 512 + 1);
    o.sha.block_size = SHA_BLOCK_SIGS;
    o.sha.blocks = s.add_sha(pre + "inputsHasher.sha256compression", (uint32_t)o.sha.nblocks, SHA_BLOCK_SIGS);
}

inline void add_input(Layout& lo, const std::string& name, int sec, uint32_t off, uint32_t inner, uint32_t outer, bool /*unused*/) {
    InputDesc d;
    d.name = name; d.section = sec; d.off = off; d.inner = inner; d.outer = outer;
    lo.inputs.push_back(d);
}

inline void build_layout(const Params& p, Layout& lo) {
    lo = Layout();
    lo.p = p;
    const int L = p.L, F = p.F, nTx = p.nTx;
    const uint32_t N = (uint32_t)(p.n_inst > 0 ? p.n_inst : 1);
    lo.n_inst = N;
    // perinst = the signal is a scalar/array of the main component itself (no per-unit index in its name)
    auto in1 = [&](Section& s, int sec, const std::string& nm, uint32_t cnt, uint32_t outer, int32_t maxu, bool perinst) {
        uint32_t o = s.add("main." + nm + (perinst ? "" : "[{u}]"), cnt, maxu);
        add_input(lo, nm, sec, o, cnt, perinst ? 1u : outer, perinst);
        return o;
    };
    switch (p.tmpl) {
        case T_ROLLUP_MAIN: {
            lo.sections.resize(4);
            lo.sec_glob = 0; lo.sec_tx = 1; lo.sec_fee = 2; lo.sec_hi = 3;
            Section& G = lo.sections[0]; G.tag = "global"; G.upi = 1;
            lo.g.one = G.add("main.one");
            lo.g.hashGlobalInputs = G.add("main.hashGlobalInputs");
            lo.outputs.push_back({"hashGlobalInputs", 0});
            auto gin = [&](const char* nm) { uint32_t o = G.add(std::string("main.") + nm); add_input(lo, nm, 0, o, 1, 1, false); return o; };
            lo.g.oldLastIdx = gin("oldLastIdx"); lo.g.oldStateRoot = gin("oldStateRoot");
            lo.g.globalChainID = gin("globalChainID"); lo.g.currentNumBatch = gin("currentNumBatch");
            lo.g.imInitStateRootFee = gin("imInitStateRootFee");
            Section& T = lo.sections[1]; T.tag = "tx"; T.upi = (uint32_t)nTx;
            MainTxInOff& m = lo.mi;
            const int32_t nm1 = nTx - 1;
            m.imOnChain = in1(T, 1, "imOnChain", 1, nm1, nm1, false);
            m.imOutIdx = in1(T, 1, "imOutIdx", 1, nm1, nm1, false);
            m.imStateRoot = in1(T, 1, "imStateRoot", 1, nm1, nm1, false);
            m.imExitRoot = in1(T, 1, "imExitRoot", 1, nm1, nm1, false);
            m.imAccFeeOut = in1(T, 1, "imAccFeeOut", F, nm1, nm1, false);
#define HZL_TXIN(f, cnt) m.f = in1(T, 1, #f, cnt, nTx, -1, false)
            HZL_TXIN(txCompressedData, 1); HZL_TXIN(amountF, 1); HZL_TXIN(txCompressedDataV2, 1); HZL_TXIN(fromIdx, 1);
            HZL_TXIN(auxFromIdx, 1); HZL_TXIN(toIdx, 1); HZL_TXIN(auxToIdx, 1); HZL_TXIN(toBjjAy, 1); HZL_TXIN(toEthAddr, 1);
            HZL_TXIN(maxNumBatch, 1); HZL_TXIN(onChain, 1); HZL_TXIN(newAccount, 1); HZL_TXIN(rqOffset, 1);
            HZL_TXIN(rqTxCompressedDataV2, 1); HZL_TXIN(rqToEthAddr, 1); HZL_TXIN(rqToBjjAy, 1); HZL_TXIN(s, 1); HZL_TXIN(r8x, 1);
            HZL_TXIN(r8y, 1); HZL_TXIN(loadAmountF, 1); HZL_TXIN(fromEthAddr, 1); HZL_TXIN(fromBjjCompressed, 256);
            HZL_TXIN(tokenID1, 1); HZL_TXIN(nonce1, 1); HZL_TXIN(sign1, 1); HZL_TXIN(balance1, 1); HZL_TXIN(ay1, 1);
            HZL_TXIN(ethAddr1, 1); HZL_TXIN(siblings1, L + 1); HZL_TXIN(isOld0_1, 1); HZL_TXIN(oldKey1, 1); HZL_TXIN(oldValue1, 1);
            HZL_TXIN(tokenID2, 1); HZL_TXIN(nonce2, 1); HZL_TXIN(sign2, 1); HZL_TXIN(balance2, 1); HZL_TXIN(ay2, 1);
            HZL_TXIN(ethAddr2, 1); HZL_TXIN(siblings2, L + 1); HZL_TXIN(newExit, 1); HZL_TXIN(isOld0_2, 1); HZL_TXIN(oldKey2, 1);
            HZL_TXIN(oldValue2, 1);
#undef HZL_TXIN
            lay_decode(T, "main.decodeTx[{u}].", L, false, lo.dec);
            lay_rtx(T, "main.rollupTx[{u}].", L, F, true, lo.rtx);
            // hasherInputs.L1L2TxsData[i * bits + j] <== decodeTx[i].L1L2TxData[j] * (1 - rollupTx[i].isAmountNullified), j in [2L, 2L + 40)
            // (src/rollup-main.circom:456-458): the only elements of that array that are products
            lo.rtx.main_l1l2amt = T.add_slice("main.hasherInputs.L1L2TxsData", 40, (uint32_t)(2 * L), (uint32_t)(2 * L + 48));
            Section& Fs = lo.sections[2]; Fs.tag = "fee"; Fs.upi = (uint32_t)F;
            MainFeeInOff& f = lo.fi;
            f.feeIdxs = in1(Fs, 2, "feeIdxs", 1, F, -1, false);
            f.feePlanTokens = in1(Fs, 2, "feePlanTokens", 1, F, -1, false);
            f.imStateRootFee = in1(Fs, 2, "imStateRootFee", 1, F - 1, F - 1, false);
            f.imFinalAccFee = in1(Fs, 2, "imFinalAccFee", 1, F, -1, false);
            f.tokenID3 = in1(Fs, 2, "tokenID3", 1, F, -1, false); f.nonce3 = in1(Fs, 2, "nonce3", 1, F, -1, false);
            f.sign3 = in1(Fs, 2, "sign3", 1, F, -1, false); f.balance3 = in1(Fs, 2, "balance3", 1, F, -1, false);
            f.ay3 = in1(Fs, 2, "ay3", 1, F, -1, false); f.ethAddr3 = in1(Fs, 2, "ethAddr3", 1, F, -1, false);
            f.siblings3 = in1(Fs, 2, "siblings3", L + 1, F, -1, false);
            lay_feetx(Fs, "main.feeTx[{u}].", L, lo.fee);
            Section& H = lo.sections[3]; H.tag = "hashinputs"; H.upi = 1;
            lay_hashinputs(H, "main.hasherInputs.", L, nTx, p.maxL1, F, false, lo.hi);
            break;
        }
        case T_ROLLUP_TX: {
            lo.sections.resize(1);
            lo.sec_tx = 0;
            Section& T = lo.sections[0]; T.tag = "rollup-tx"; T.upi = 1;
            RtxInOff& r = lo.rtxi;
            T.add("main.one");
            r.o_isAmountNullified = T.add("main.isAmountNullified");
            const uint32_t o_accFeeOut = T.add("main.accFeeOut", F, -1, true);
            r.o_newStateRoot = T.add("main.newStateRoot");
            r.o_newExitRoot = T.add("main.newExitRoot");
            lo.outputs = {{"isAmountNullified", r.o_isAmountNullified}, {"accFeeOut", o_accFeeOut}, {"newStateRoot", r.o_newStateRoot}, {"newExitRoot", r.o_newExitRoot}};
#define HZL_RIN(f, cnt) r.f = in1(T, 0, #f, cnt, N, -1, true)
            HZL_RIN(feePlanTokens, F); HZL_RIN(accFeeIn, F);
            r.futureV2 = in1(T, 0, "futureTxCompressedDataV2", 3, N, -1, true);
            r.pastV2 = in1(T, 0, "pastTxCompressedDataV2", 4, N, -1, true);
            HZL_RIN(futureToEthAddr, 3); HZL_RIN(pastToEthAddr, 4); HZL_RIN(futureToBjjAy, 3); HZL_RIN(pastToBjjAy, 4);
            HZL_RIN(fromIdx, 1); HZL_RIN(auxFromIdx, 1); HZL_RIN(toIdx, 1); HZL_RIN(auxToIdx, 1); HZL_RIN(toBjjAy, 1);
            HZL_RIN(toBjjSign, 1); HZL_RIN(toEthAddr, 1); HZL_RIN(amount, 1); HZL_RIN(tokenID, 1); HZL_RIN(nonce, 1);
            HZL_RIN(userFee, 1); HZL_RIN(rqOffset, 1); HZL_RIN(onChain, 1); HZL_RIN(newAccount, 1); HZL_RIN(rqTxCompressedDataV2, 1);
            HZL_RIN(rqToEthAddr, 1); HZL_RIN(rqToBjjAy, 1); HZL_RIN(sigL2Hash, 1); HZL_RIN(s, 1); HZL_RIN(r8x, 1); HZL_RIN(r8y, 1);
            HZL_RIN(fromEthAddr, 1); HZL_RIN(fromBjjCompressed, 256); HZL_RIN(loadAmountF, 1); HZL_RIN(tokenID1, 1);
            HZL_RIN(nonce1, 1); HZL_RIN(sign1, 1); HZL_RIN(balance1, 1); HZL_RIN(ay1, 1); HZL_RIN(ethAddr1, 1);
            HZL_RIN(siblings1, L + 1); HZL_RIN(isOld0_1, 1); HZL_RIN(oldKey1, 1); HZL_RIN(oldValue1, 1); HZL_RIN(tokenID2, 1);
            HZL_RIN(nonce2, 1); HZL_RIN(sign2, 1); HZL_RIN(balance2, 1); HZL_RIN(newExit, 1); HZL_RIN(ay2, 1); HZL_RIN(ethAddr2, 1);
            HZL_RIN(siblings2, L + 1); HZL_RIN(isOld0_2, 1); HZL_RIN(oldKey2, 1); HZL_RIN(oldValue2, 1); HZL_RIN(oldStateRoot, 1);
            HZL_RIN(oldExitRoot, 1);
#undef HZL_RIN
            lay_rtx(T, "main.", L, F, false, lo.rtx);
            lo.rtx.o_accFeeOut = o_accFeeOut;
            break;
        }
        case T_DECODE_TX: {
            lo.sections.resize(1);
            lo.sec_tx = 0;
            Section& T = lo.sections[0]; T.tag = "decode-tx"; T.upi = 1;
            T.add("main.one");
            DecInOff& d = lo.deci;
#define HZL_DIN(f, cnt) d.f = in1(T, 0, #f, cnt, N, -1, true)
            HZL_DIN(previousOnChain, 1); HZL_DIN(txCompressedData, 1); HZL_DIN(maxNumBatch, 1); HZL_DIN(amountF, 1);
            HZL_DIN(toEthAddr, 1); HZL_DIN(toBjjAy, 1); HZL_DIN(rqTxCompressedDataV2, 1); HZL_DIN(rqToEthAddr, 1);
            HZL_DIN(rqToBjjAy, 1); HZL_DIN(fromEthAddr, 1); HZL_DIN(fromBjjCompressed, 256); HZL_DIN(loadAmountF, 1);
            HZL_DIN(globalChainID, 1); HZL_DIN(currentNumBatch, 1); HZL_DIN(onChain, 1); HZL_DIN(newAccount, 1);
            HZL_DIN(auxFromIdx, 1); HZL_DIN(auxToIdx, 1); HZL_DIN(inIdx, 1);
#undef HZL_DIN
            lay_decode(T, "main.", L, true, lo.dec);
            lo.outputs = {{"L1L2TxData", lo.dec.o_l1l2}, {"txCompressedDataV2", lo.dec.o_v2}, {"L1TxFullData", lo.dec.l1full},
                          {"outIdx", lo.dec.outIdx}, {"fromIdx", lo.dec.o_fromIdx}, {"toIdx", lo.dec.o_toIdx}, {"tokenID", lo.dec.o_tokenID},
                          {"nonce", lo.dec.o_nonce}, {"userFee", lo.dec.o_userFee}, {"toBjjSign", lo.dec.o_toBjjSign},
                          {"amount", lo.dec.o_amount}, {"sigL2Hash", lo.dec.o_sigL2Hash}};
            break;
        }
        case T_FEE_TX: {
            lo.sections.resize(1);
            lo.sec_fee = 0;
            Section& T = lo.sections[0]; T.tag = "fee-tx"; T.upi = 1;
            T.add("main.one");
            lo.fee.o_newStateRoot = 0;
            uint32_t o_root = T.add("main.newStateRoot");
            FeeTxInOff& d = lo.feei;
#define HZL_FIN(f, cnt) d.f = in1(T, 0, #f, cnt, N, -1, true)
            HZL_FIN(oldStateRoot, 1); HZL_FIN(feePlanToken, 1); HZL_FIN(feeIdx, 1); HZL_FIN(accFee, 1); HZL_FIN(tokenID, 1);
            HZL_FIN(nonce, 1); HZL_FIN(sign, 1); HZL_FIN(balance, 1); HZL_FIN(ay, 1); HZL_FIN(ethAddr, 1); HZL_FIN(siblings, L + 1);
#undef HZL_FIN
            lay_feetx(T, "main.", L, lo.fee);
            lo.fee.o_newStateRoot = o_root;
            lo.outputs = {{"newStateRoot", o_root}};
            break;
        }
        case T_DECODE_FLOAT: case T_COMPUTE_FEE: case T_FEE_ACCUMULATOR: case T_BALANCE_UPDATER: case T_ROLLUP_TX_STATES: case T_RQ_TX_VERIFIER:
        case T_MUX256: case T_BITS2AYSIGN: case T_AYSIGN2AX: {
            // gadget mains (reference test/lib/decode-float.test.js, test/compute-fee.test.js, test/fee-accumulator.test.js,
            // test/balance-updater.test.js, test/rollup-tx-states.test.js, test/rq-tx-verifier.test.js): one instance per unit
            lo.sections.resize(1);
            Section& T = lo.sections[0]; T.tag = "gadget"; T.upi = 1;
            T.add("main.one");
            GadIO& g = lo.gad;
            int ni = 0, no = 0;
            auto gin = [&](const char* nm, uint32_t cnt = 1) { g.in[ni++] = in1(T, 0, nm, cnt, N, -1, true); };
            auto gin_at = [&](const char* nm, uint32_t off) { g.in[ni++] = off; add_input(lo, nm, 0, off, 1, 1, true); };
            auto gout_new = [&](const char* nm, uint32_t cnt = 1) { g.out[no++] = T.add(std::string("main.") + nm, cnt); lo.outputs.push_back({nm, g.out[no - 1]}); };
            auto gout_at = [&](const char* nm, uint32_t off) { g.out[no++] = off; lo.outputs.push_back({nm, off}); };
            if (p.tmpl == T_DECODE_FLOAT) {            // src/lib/decode-float.circom:50-64
                gin("in");
                gout_new("out");
                lo.rtx.n2bLoadAmountF = T.add("main.n2b.out", 40);
                lo.rtx.dfLoadAmount = lay_decodefloat(T, "main.decoder.");
            } else if (p.tmpl == T_COMPUTE_FEE) {      // src/compute-fee.circom:12-109
                gin("feeSel"); gin("amount");
                lay_computefee(T, "main.", lo.rtx.bu.fee);
                gin_at("applyFee", lo.rtx.bu.fee.applyFee);
                gout_at("feeOut", lo.rtx.bu.fee.feeOut);
            } else if (p.tmpl == T_FEE_ACCUMULATOR) {  // src/fee-accumulator.circom:56-91
                gin("tokenID"); gin("fee2Charge"); gin("feePlanTokenID", (uint32_t)F); gin("accFeeIn", (uint32_t)F);
                gout_new("accFeeOut", (uint32_t)F);
                lo.rtx.feeAcc = lay_feeacc(T, "main.", F);
            } else if (p.tmpl == T_BALANCE_UPDATER) {  // src/balance-updater.circom:24-105
                for (const char* nm : {"oldStBalanceSender", "oldStBalanceReceiver", "amount", "loadAmount", "feeSelector", "onChain", "nop", "nullifyLoadAmount", "nullifyAmount"}) gin(nm);
                gout_new("newStBalanceSender"); gout_new("newStBalanceReceiver"); gout_new("isP2Nop"); gout_new("fee2Charge");
                lay_balupd(T, "main.", lo.rtx.bu);
                gout_at("isAmountNullified", lo.rtx.bu.isAmountNullified);
            } else if (p.tmpl == T_ROLLUP_TX_STATES) { // src/rollup-tx-states.circom:39-314
                for (const char* nm : {"fromIdx", "toIdx", "toEthAddr", "auxFromIdx", "auxToIdx", "amount", "newExit", "loadAmount", "newAccount", "onChain",
                                       "fromEthAddr", "ethAddr1", "tokenID", "tokenID1", "tokenID2"}) gin(nm);
                StatesOff& st = lo.rtx.st;
                lay_states(T, "main.", st);
                gout_at("isP1Insert", st.isP1Insert); gout_at("isP2Insert", st.isP2Insert);
                gout_new("key1"); gout_new("key2");
                gout_at("P1_fnc0", st.P1_fnc0); gout_at("P1_fnc1", st.P1_fnc1); gout_at("P2_fnc0", st.P2_fnc0); gout_at("P2_fnc1", st.P2_fnc1);
                gout_new("isExit");
                gout_at("verifySignEnabled", st.verifySignEnabled);
                gout_new("nop");
                gout_at("checkToEthAddr", st.checkToEthAddr); gout_at("checkToBjj", st.checkToBjj);
                gout_at("nullifyLoadAmount", st.nullifyLoadAmount); gout_at("nullifyAmount", st.nullifyAmount);
            } else if (p.tmpl == T_MUX256) {           // src/lib/mux256.circom:10-52 (test/lib/mux256.test.js)
                gin("s", 8); gin("in", 256);
                gout_new("out");
                g.mux = T.n_sigs;
                for (int i = 0; i < 17; i++) lay_mux4v(T, "main.mux[" + istr(i) + "].");
            } else if (p.tmpl == T_BITS2AYSIGN) {      // src/lib/utils-bjj.circom:12-28 (test/lib/utils-bjj.test.js)
                gin("bjjCompressed", 256);
                gout_new("ay"); gout_new("sign");
            } else if (p.tmpl == T_AYSIGN2AX) {        // src/lib/utils-bjj.circom:37-58
                gin("ay"); gin("sign");
                gout_new("ax");
                lay_ax(T, "main.", lo.rtx.ed);
            } else {                                   // src/rq-tx-verifier.circom:19-94
                gin("futureTxCompressedDataV2", 3); gin("pastTxCompressedDataV2", 4); gin("futureToEthAddr", 3); gin("pastToEthAddr", 4);
                gin("futureToBjjAy", 3); gin("pastToBjjAy", 4); gin("rqTxCompressedDataV2"); gin("rqToEthAddr"); gin("rqToBjjAy"); gin("rqTxOffset");
                lay_rq(T, "main.", lo.rtx);
            }
            break;
        }
        case T_HASH_STATE: {
            lo.sections.resize(1);
            Section& T = lo.sections[0]; T.tag = "hash-state"; T.upi = 1;
            HashStateOff& h = lo.hs;
            h.one = T.add("main.one");
            h.out = T.add("main.out");
            h.tokenID = in1(T, 0, "tokenID", 1, N, -1, true); h.nonce = in1(T, 0, "nonce", 1, N, -1, true);
            h.sign = in1(T, 0, "sign", 1, N, -1, true); h.balance = in1(T, 0, "balance", 1, N, -1, true);
            h.ay = in1(T, 0, "ay", 1, N, -1, true); h.ethAddr = in1(T, 0, "ethAddr", 1, N, -1, true);
            h.hash = T.add_poseidon("main.hash", 5);
            lo.outputs = {{"out", h.out}};
            break;
        }
        case T_WITHDRAW: {
            lo.sections.resize(1);
            Section& T = lo.sections[0]; T.tag = "withdraw"; T.upi = 1;
            WithdrawOff& w = lo.wd;
            w.one = T.add("main.one");
            w.hashGlobalInputs = T.add("main.hashGlobalInputs");
            w.rootExit = in1(T, 0, "rootExit", 1, N, -1, true); w.ethAddr = in1(T, 0, "ethAddr", 1, N, -1, true);
            w.tokenID = in1(T, 0, "tokenID", 1, N, -1, true); w.balance = in1(T, 0, "balance", 1, N, -1, true);
            w.idx = in1(T, 0, "idx", 1, N, -1, true); w.sign = in1(T, 0, "sign", 1, N, -1, true);
            w.ay = in1(T, 0, "ay", 1, N, -1, true); w.siblingsState = in1(T, 0, "siblingsState", L + 1, N, -1, true);
            w.accountState = T.add_poseidon("main.accountState.hash", 5);
            lay_smtver(T, "main.smtVerify.", L + 1, w.ver);
            const std::string h = "main.hasherInputs.";
            w.n2bRootExit = T.add(h + "n2bRootExit.out", 256); w.n2bEthAddr = T.add(h + "n2bEthAddr.out", 160);
            w.n2bTokenID = T.add(h + "n2bTokenID.out", 32); w.n2bBalance = T.add(h + "n2bBalance.out", 192);
            w.n2bIdx = T.add(h + "n2bIdx.out", 48);
            w.sha.nblocks = 2; w.sha.block_size = SHA_BLOCK_SIGS;
            w.sha.blocks = T.add_sha(h + "inputsHasher.sha256compression", 2, SHA_BLOCK_SIGS);
            lo.outputs = {{"hashGlobalInputs", w.hashGlobalInputs}};
            break;
        }
        case T_SMT_PROCESSOR: {   // circomlib smt/smtprocessor.circom: SMTProcessor(nLevels), n = nLevels levels
            lo.sections.resize(1);
            lo.sec_fee = 0;   // shares the fee-transaction scratch allocation (one SMTProcessor per unit)
            Section& T = lo.sections[0]; T.tag = "smt-processor"; T.upi = 1;
            SmtProcInOff& d = lo.smtpi;
            d.one = T.add("main.one");
            d.oldRoot = in1(T, 0, "oldRoot", 1, N, -1, true); d.siblings = in1(T, 0, "siblings", (uint32_t)L, N, -1, true);
            d.oldKey = in1(T, 0, "oldKey", 1, N, -1, true); d.oldValue = in1(T, 0, "oldValue", 1, N, -1, true);
            d.isOld0 = in1(T, 0, "isOld0", 1, N, -1, true); d.newKey = in1(T, 0, "newKey", 1, N, -1, true);
            d.newValue = in1(T, 0, "newValue", 1, N, -1, true); d.fnc = in1(T, 0, "fnc", 2, N, -1, true);
            lay_smtproc(T, "main.", L, false, lo.smtp);
            lo.outputs = {{"newRoot", lo.smtp.newRoot}};
            break;
        }
        case T_SMT_VERIFIER: {    // circomlib smt/smtverifier.circom: SMTVerifier(nLevels)
            lo.sections.resize(1);
            Section& T = lo.sections[0]; T.tag = "smt-verifier"; T.upi = 1;
            SmtVerInOff& d = lo.smtvi;
            d.one = T.add("main.one");
            d.enabled = in1(T, 0, "enabled", 1, N, -1, true); d.root = in1(T, 0, "root", 1, N, -1, true);
            d.siblings = in1(T, 0, "siblings", (uint32_t)L, N, -1, true); d.oldKey = in1(T, 0, "oldKey", 1, N, -1, true);
            d.oldValue = in1(T, 0, "oldValue", 1, N, -1, true); d.isOld0 = in1(T, 0, "isOld0", 1, N, -1, true);
            d.key = in1(T, 0, "key", 1, N, -1, true); d.value = in1(T, 0, "value", 1, N, -1, true); d.fnc = in1(T, 0, "fnc", 1, N, -1, true);
            lay_smtver(T, "main.", L, lo.smtv);
            break;
        }
        case T_HASH_INPUTS: {
            lo.sections.resize(1);
            lo.sec_hi = 0;
            Section& H = lo.sections[0]; H.tag = "hashinputs"; H.upi = 1;
            lay_hashinputs(H, "main.", L, nTx, p.maxL1, F, true, lo.hi);
            HashInputsOff& o = lo.hi;
            add_input(lo, "oldLastIdx", 0, o.i_oldLastIdx, 1, 1, false); add_input(lo, "newLastIdx", 0, o.i_newLastIdx, 1, 1, false);
            add_input(lo, "oldStateRoot", 0, o.i_oldStateRoot, 1, 1, false); add_input(lo, "newStateRoot", 0, o.i_newStateRoot, 1, 1, false);
            add_input(lo, "newExitRoot", 0, o.i_newExitRoot, 1, 1, false);
            add_input(lo, "L1TxsFullData", 0, o.i_L1TxsFullData, (uint32_t)(p.maxL1 * L1FULL_BITS), 1, false);
            add_input(lo, "L1L2TxsData", 0, o.i_L1L2TxsData, (uint32_t)(nTx * (2 * L + 48)), 1, false);
            add_input(lo, "feeTxsData", 0, o.i_feeTxsData, (uint32_t)F, 1, false);
            add_input(lo, "globalChainID", 0, o.i_globalChainID, 1, 1, false); add_input(lo, "currentNumBatch", 0, o.i_currentNumBatch, 1, 1, false);
            lo.outputs = {{"hashInputsOut", o.out}};
            break;
        }
    }
    uint64_t base = 0, vbase = 0;
    for (Section& s : lo.sections) {
        s.n_units = s.upi * N;
        s.base = base;
        s.vbase = vbase;
        base += s.size();
        vbase += (uint64_t)s.n_sigs * s.upi;
    }
    lo.total = base;
    lo.per_instance = vbase;
    // packed bulk-upload format: the 256 bits of fromBjjCompressed travel as one byte each (16.7 MB -> 0.5 MB per 2048-tx batch)
    for (InputDesc& d : lo.inputs)
        if (d.name == "fromBjjCompressed") d.ebytes = 1;
}

// closed-form constraint estimate of the reference (tools/circuit-constraints.js:31-75)
inline uint64_t constraint_estimate(const Params& p) {
    const uint64_t L = p.L, F = p.F, nTx = p.nTx, m1 = p.maxL1;
    const uint64_t dec = 4 * L + 1473, fee = 483 * L + 2592, rtx = 974 * L + 14552 + 5 * F;
    switch (p.tmpl) {
        case T_ROLLUP_MAIN: {
            const uint64_t bitsL1 = m1 * (2 * L + 528), bitsL2 = nTx * (2 * L + 48), bitsFee = F * L;
            const uint64_t bitsSha = 2 * L + 3 * 256 + 16 + bitsL1 + bitsL2 + bitsFee;
            const uint64_t sha = 28953 + 29305 * ((bitsSha + 64) / 512);
            const uint64_t hi = sha + 2 * bitsL1 + 2 * bitsL2 + (48 + 2 * L) * F;
            const uint64_t im = 2 * 3 * nTx + (2 + F) * 2 * nTx + 2 * (1 + 2 * F);
            return dec * nTx + fee * F + rtx * nTx + hi + im;
        }
        case T_ROLLUP_TX: return rtx;
        case T_DECODE_TX: return dec;
        case T_FEE_TX: return fee;
        default: return 0;
    }
}

}  // namespace hzl
