// Kernel argument blocks and host launchers of the witness kernels (tx_kernels.hip,
// eddsa_kernels.hip, fee_kernels.hip, sha_kernels.hip). Argument blocks are passed by value.
#pragma once
#include <hip/hip_runtime.h>
#include "../../include/hz_layout.h"
#include "devcommon.h"

namespace hz {
using namespace hzl;

// One wavefront per workgroup: at nTx = 2048 a batch is only 32 wavefronts per lane mapping, so
// small workgroups spread them over as many CUs (and all 8 XCDs) as possible.
#define HZ_BLOCK 64
#define HZ_MAX_SMT_LEVELS 49

struct MainFrontArgs {
    uint32_t u0, ucnt;   // unit range evaluated by this launch (multi-GPU shard); ucnt = 0 means all
    uint8_t* tx_base;
    uint8_t* fee_base;
    uint8_t* glob_base;
    Fr* scratch;
    ErrBuf* err;
    uint32_t nTx, L, F, B;   // B = batches (instances) evaluated by this launch; units = B * nTx
    MainGlobOff g;
    MainTxInOff mi;
    MainFeeInOff fi;
    DecOff dec;
    RtxOff rtx;
};

struct RtxFrontArgs {
    uint8_t* base;
    Fr* scratch;
    ErrBuf* err;
    uint32_t N, L, F;
    RtxInOff in;
    RtxOff rtx;
};

struct DecMainArgs {
    uint8_t* base;
    ErrBuf* err;
    uint32_t N, L;
    DecInOff in;
    DecOff dec;
};

// a gadget main (gadget_kernels.hip): the offsets of the one gadget the template instantiates are set, the rest unused
struct GadgetArgs {
    uint8_t* base;
    ErrBuf* err;
    uint32_t N, F;
    GadIO io;
    uint32_t n2b40;          // DecodeFloat: n2b.out[40]
    DecodeFloatOff df;
    BalUpdOff bu;            // ComputeFee uses bu.fee
    uint32_t feeAcc;
    StatesOff st;
    uint32_t rq_n2b;
    Mux3Off rq_mux[3];
};

// one hash-state job of k_hash4 (blockIdx.y selects the job)
struct HashJob {
    PoseidonOff hs;        // HashState Poseidon (t = 5) signals
    PoseidonOff h1;        // SMTHash1 (t = 4) signals, ~0u = none
    uint32_t sc_in;        // scratch: 4 Poseidon inputs
    uint32_t sc_key;       // scratch: key for SMTHash1
    uint32_t sc_leaf;      // scratch: destination of the leaf hash
    uint32_t mux_off;      // s1OldValue/s2OldValue signal or ~0u
    uint32_t sc_ins, sc_oldvalue;
    uint32_t out_sig;      // signal receiving the hash-state output (~0u = none)
};
struct Hash4Args {
    uint32_t u0, ucnt;   // unit range evaluated by this launch (multi-GPU shard); ucnt = 0 means all
    uint8_t* base;
    Fr* scratch;
    uint32_t n_units, n_jobs;
    HashJob job[4];
};

struct SmtProcDesc {
    SmtProcOff o;
    uint32_t siblings;            // input offset of siblings[0]
    int sc_oldkey, sc_newkey, sc_fnc0, sc_fnc1, sc_isold0, sc_leaf_old, sc_leaf_new, sc_root_old, sc_root_new;
    int cid_alias_old, cid_alias_new, cid_levins, cid_sm_final;
};

struct SmtArgs {
    uint32_t u0, ucnt;   // unit range evaluated by this launch (multi-GPU shard); ucnt = 0 means all
    uint32_t ustride;    // 0 / 1: units u0 .. u0+ucnt-1; k > 1: units u0, u0+k, ... (the last transaction of every batch: early HashInputs)
    uint8_t* base;
    Fr* scratch;
    ErrBuf* err;
    uint32_t n_units, n_levels;   // n_levels = L + 1
    uint32_t n_proc;
    uint32_t upi;
    uint32_t skip_mod;            // != 0: units u with u % skip_mod == skip_mod - 1 belong to another launch of the step (early tail)
    // Constant marks (ctx.hip "constant marks"): two bytes per (chain, unit) that say from which level up the persistent witness buffer
    // ALREADY holds what an empty level gets -- low byte: the constant S-box block of the level hash (HZ_POSEIDON3_ZERO_WIT), high byte: zeros
    // in the per-level switcher / state-machine signals. k_smt stores such a level only below the mark and leaves the new mark behind.
    // NULL: every level is stored (HZ_NO_ZMARK, experiments).
    uint16_t* zmark;              // [2 * n_proc][n_units]
    unsigned long long* skipped;  // profiling: elements (32 B each) this launch did NOT store because of the marks; NULL: not counted
    uint32_t chain_order;         // blockIdx.y -> chain, four bits each (0: the identity). Workgroups are dispatched in blockIdx order and a
                                  // launch of the headline size is exactly two rounds of the device's wavefront slots: which chains
                                  // share the first round decides how long the last one runs alone (smt_kernels.hip launch_smt)
    unsigned long long* trace;    // experiments (HZ_SMT_TRACE): four words per wavefront of the transaction launch -- start, end (100 MHz
                                  // wall clock), HW_ID | XCC_ID << 32, blockIdx.x | blockIdx.y << 24 | wave << 28; NULL: off
    const Fr* pos3_dense;         // the dense constants of poseidon_quad.h (C[195], M1[9], M2[9]); NULL: no latency form for this launch
    SmtProcDesc p[2];
};

struct RtxBackArgs {
    uint32_t u0, ucnt;   // unit range evaluated by this launch (multi-GPU shard); ucnt = 0 means all
    uint32_t ustride;    // as in SmtArgs
    uint8_t* base;
    uint8_t* glob_base;
    Fr* scratch;
    ErrBuf* err;
    uint32_t n_units, L, is_main, upi, B;
    uint32_t skip_mod;   // as in SmtArgs
    uint32_t skip_h;     // != 0: phase H (data-availability amount masking) is k_da_mask's this step
    SmtProcDesc p[2];
    uint32_t s3, s4, s5;
    // main: integrity checks + data-availability masking
    uint32_t im_stateroot, im_exitroot, g_initfeeroot, main_l1l2amt, n2bAmount;
    // standalone: outputs
    uint32_t o_newStateRoot, o_newExitRoot;
};

struct EddsaArgs {
    uint32_t u0, ucnt;   // unit range evaluated by this launch (multi-GPU shard); ucnt = 0 means all
    uint8_t* base;
    Fr* scratch;
    ErrBuf* err;
    uint32_t n_units, upi;
    EddsaOff ed;
    uint32_t* side;      // seg_any_proj's parked numerators (eddsa_side_bytes(n)); nullptr: the inversion-per-step ladder
    uint32_t chain_in_ladder;   // set by launch_eddsa for small launches: the doubling chain between the segments is the second segment lane's
    // RollupMain, small launches (set by ctx.hip; in_onChain == ~0u: off): the prologue in two kernels. k_eddsa_pre_a -- AySign2Ax, 8A,
    // the zero checks -- takes what it needs of the front step straight from the INPUTS (five signals and, for a new account, the key
    // bits) and so runs beside the front kernel instead of behind it; k_eddsa_pre_b -- the message hash and its bits -- follows both.
    uint32_t in_onChain, in_newAccount, in_auxFromIdx, in_sign1, in_ay1, in_txCompressedData, in_fromBjjCompressed;
};
size_t eddsa_side_bytes(uint32_t n_signatures);   // 0 when a launch of that size does not use the buffer

// FeeTx (reference src/fee-tx.circom:26-112): front = IsZero/checker + hash-state inputs,
// back = processor top + newStateRoot (+ RollupMain phase G check)
struct FeeFrontArgs {
    uint8_t* base;
    uint8_t* glob_base;
    Fr* scratch;
    ErrBuf* err;
    uint32_t n_units, is_main, upi, B;
    FeeTxOff fee;
    // input offsets (MainFeeInOff names for main, FeeTxInOff for standalone)
    uint32_t in_feePlanToken, in_feeIdx, in_accFee, in_tokenID, in_nonce, in_sign, in_balance, in_ay, in_ethAddr, in_oldStateRoot;
    uint32_t im_stateRootFee, g_initfeeroot;
};
struct FeeBackArgs {
    uint8_t* base;
    Fr* scratch;
    ErrBuf* err;
    uint32_t n_units, is_main, upi;
    SmtProcDesc p;
    uint32_t im_stateRootFee, o_newStateRoot;
};

struct HashInputsArgs {
    uint8_t* hi_base;    // hash-inputs section (1 unit)
    uint8_t* glob_base;
    uint8_t* tx_base;
    uint8_t* fee_base;
    Fr* tx_scratch;
    Fr* fee_scratch;
    ErrBuf* err;
    uint32_t nTx, L, maxL1, F, is_main, B;
    HashInputsOff hi;
    MainGlobOff g;
    uint32_t mi_onChain, fi_feeIdxs;
    DecOff dec;
    uint32_t rtx_s5, rtx_main_l1l2amt;
    uint32_t fee_newRoot_sc;   // scratch field holding feeTx newStateRoot
    uint8_t* msg;              // device byte buffer for the padded message (nblocks * 64)
    uint32_t* chain;           // device buffer: (nblocks + 1) * 8 chaining words
    uint32_t blk0, blk1;       // SHA-256 blocks handled by one k_sha_chain / k_sha_expand launch (set by launch_hash_inputs)
    uint32_t prep_part;        // k_hi_prep: 0 = the whole message, 1 = everything but the header lane, 2 = the header lane alone
};

struct WithdrawArgs {
    uint8_t* base;
    Fr* scratch;
    ErrBuf* err;
    uint32_t N, L;
    WithdrawOff wd;
};

// SMTProcessor(n) / SMTVerifier(n) as main (smt_main_kernels.hip)
struct SmtMainArgs {
    uint8_t* base;
    Fr* scratch;
    ErrBuf* err;
    uint32_t N, n_levels;
    SmtProcInOff pin;
    SmtProcDesc proc;
    SmtVerInOff vin;
    SmtVerOff ver;
};
hipError_t launch_smtproc_front(const SmtMainArgs& a, hipStream_t s);
hipError_t launch_smtproc_back(const SmtMainArgs& a, hipStream_t s);
hipError_t launch_smtver_main(const SmtMainArgs& a, hipStream_t s);

hipError_t launch_withdraw_sha(const WithdrawArgs& a, hipStream_t s);
hipError_t launch_main_front(const MainFrontArgs& a, hipStream_t s);
hipError_t launch_main_sighash(const MainFrontArgs& a, hipStream_t s);  // DecodeTx's sigL2Hash from the inputs alone (SC_SIGL2HASH for the signature prologue)
hipError_t launch_main_feeacc(const MainFrontArgs& a, hipStream_t s);   // RollupTx's FeeAccumulator, after the front kernel (reads its scratch hand-off)
hipError_t launch_rtx_front(const RtxFrontArgs& a, hipStream_t s);
hipError_t launch_dec_main(const DecMainArgs& a, hipStream_t s);
hipError_t launch_gadget(int tmpl, const GadgetArgs& a, hipStream_t s);
hipError_t launch_fr_sqrt(const void* d_a, void* d_out, size_t n, hipStream_t s);   // eddsa_kernels.hip (hz_fr_ops HZ_FR_SQRT)
hipError_t launch_ay_sign_2_ax_main(const GadgetArgs& a, const EddsaOff& o, hipStream_t s);   // eddsa_kernels.hip (shares the curve code)
hipError_t launch_hash4(const Hash4Args& a, hipStream_t s);
#ifndef HZ_SMT_LAT_MAX
#define HZ_SMT_LAT_MAX 2048u   // units up to which k_smt may run in its latency form (a quad of lanes per chain): one wavefront per SIMD of half the
                               // device holds 2048 units x 4 chains x 4 lanes; two batches alone: 11.7 ms with it, 9.7 without (one: 7.9 / 9.4)
#endif
hipError_t launch_smt(const SmtArgs& a, hipStream_t s);
size_t pos3_dense_bytes();
hipError_t upload_pos3_dense(Fr* dst);   // synchronous; dst holds pos3_dense_bytes()
hipError_t launch_rtx_back(const RtxBackArgs& a, hipStream_t s);
hipError_t launch_da_mask(const RtxBackArgs& a, hipStream_t s);   // RollupMain phase H alone (amount bits of L1L2TxData times 1 - isAmountNullified), every unit
hipError_t launch_eddsa(const EddsaArgs& a, hipStream_t s, hipEvent_t front_done = nullptr, hipEvent_t hash_done = nullptr);   // hash_done: SC_SIGL2HASH is written (RollupMain: k_main_sighash on the fee stream)   // front_done: waited for where the front step's scratch is first read (NULL: the caller has ordered the stream already)         // AySign2Ax, message hash, variable-base ladder, R8 + h*8A
hipError_t launch_eddsa_fix(const EddsaArgs& a, hipStream_t s);     // S bits / range, S*B8 (independent of the above)
hipError_t launch_eddsa_final(const EddsaArgs& a, hipStream_t s);   // the equality of the two sides
hipError_t launch_fee_front(const FeeFrontArgs& a, hipStream_t s);
hipError_t launch_fee_back(const FeeBackArgs& a, hipStream_t s);
hipError_t launch_hash_inputs(const HashInputsArgs& a, hipStream_t s, hipStream_t side = nullptr, hipEvent_t* ev = nullptr, int n_ev = 0, bool body_done = false);
// the part of the message that does not wait for the roots (the data-availability bits: front kernel + k_da_mask); launch_hash_inputs(body_done = true) follows
hipError_t launch_hi_prep_body(const HashInputsArgs& a, hipStream_t s);
// tx-sharded batches (multi-GPU): the message and the sequential chain alone (rank 0), then the bit-level witness of a range of
// blocks on whichever rank holds the message blocks and chaining values (hz_sha_export / hz_sha_expand)
hipError_t launch_hi_chain_only(const HashInputsArgs& a, hipStream_t s);
hipError_t launch_sha_expand_range(const HashInputsArgs& a, uint32_t first, uint32_t count, hipStream_t s);
hipError_t launch_hash_state_main(uint8_t* base, uint32_t N, const HashStateOff& hs, hipStream_t s);
hipError_t launch_withdraw(const WithdrawArgs& a, hipStream_t s);

// data-availability record of one transaction (multi-GPU shard exchange), see ctx.hip hz_da_export
#define HZ_DA_RECORD_BYTES 160
struct DaArgs {
    uint8_t* tx_base;
    Fr* tx_scratch;
    uint8_t* buf;          // records of units [u0, u0 + ucnt)
    uint32_t nTx, L, u0, ucnt;
    uint32_t l1full, n2bData, n2bFinalToIdx, l1l2amt, l1l2Fee, s5;
};
hipError_t launch_da_export(const DaArgs& a, hipStream_t s);
hipError_t launch_da_import(const DaArgs& a, hipStream_t s);

}  // namespace hz
