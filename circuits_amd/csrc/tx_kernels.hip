// Kernels of the per-transaction path (SURVEY 8a' K2-K4 and the back/top step):
//   k_main_front  lane = (tx, role) RollupMain phase A/C checks + DecodeTx | RollupTx front: states | mux | balance
//   k_rtx_front   lane = (instance, role) standalone RollupTx front
//   k_dec_main    lane = instance   standalone DecodeTx
//   k_hash4       lane = (tx, j)    HashState j in {old1, old2, new1, new2} + its SMTHash1
//   k_smt         lane = (tx, c)    one of the four independent level-hash chains of the two
//                                   SMTProcessors (c = processor*2 + side)
//   k_rtx_back    lane = tx         processor tops, root selection, im* integrity checks
// Every lane writes its signals signal-major, so a wavefront's stores are contiguous in HBM.
// field routines inlined (HZ_FR_INLINE): k_main_front 6.5 -> 5.8 ms per 65 536 transactions against the out-of-line product
#ifndef HZ_FR_INLINE
#define HZ_FR_INLINE 1
#endif
#include <hip/hip_runtime.h>
#include "tx_dev.h"
#include "kernels.h"
#include "smt_dev.h"

namespace hz {

// ---------------------------------------------------------------------------------------------------
struct FeeSrcMain {
    const UnitIO* io;         // tx section
    const uint8_t* fee_base;  // fee section base
    uint32_t n_fee_units;     // B * F
    uint32_t fee0;            // b * F: first fee unit of this batch
    uint32_t plan_off;        // fi.feePlanTokens
    uint32_t final_off;       // fi.imFinalAccFee
    uint32_t im_off;          // mi.imAccFeeOut
    uint32_t nTx, i;          // transactions per batch, index inside the batch
    __device__ __forceinline__ Fr plan(int j) const { return fr_from_canon(load_fr(fee_base + ((size_t)plan_off * n_fee_units + fee0 + j) * 32)); }
    __device__ __forceinline__ Fr acc(int j) const { return i == 0 ? fr_zero() : io->in_m_u(im_off + j, io->unit - 1); }
    __device__ __forceinline__ void out(const UnitIO& w, int j, const Fr& v) const {
        // phase E / G of RollupMain (src/rollup-main.circom:386-388,429-431)
        if (i + 1 < nTx) w.chk(C_MAIN_IM_ACCFEE, v, w.in_m(im_off + j));
        else w.chk(C_MAIN_IM_FINALACCFEE, v, fr_from_canon(load_fr(fee_base + ((size_t)final_off * n_fee_units + fee0 + j) * 32)));
    }
};
struct FeeSrcRtx {
    const UnitIO* io;
    uint32_t plan_off, acc_off, out_off;
    __device__ __forceinline__ Fr plan(int j) const { return io->in_m(plan_off + j); }
    __device__ __forceinline__ Fr acc(int j) const { return io->in_m(acc_off + j); }
    __device__ __forceinline__ void out(const UnitIO& w, int j, const Fr& v) const { w.put_m(out_off + j, v); }
};


// what DecodeTx takes from outside its transaction (decode_tx_dev's EXT): RollupMain wires it from the transaction before / the batch's
// globals (src/rollup-main.circom:221-256), the standalone main has inputs
struct DecExtMain {
    const UnitIO* io;
    const uint8_t* glob_base;
    uint32_t B, b, i, u, imOnChain, imOutIdx, g_oldLastIdx, g_chainID, g_numBatch;
    __device__ __forceinline__ Fr glob(uint32_t sig) const { return fr_from_canon(load_fr(glob_base + ((size_t)sig * B + b) * 32)); }
    __device__ __forceinline__ Fr previousOnChain() const { return i == 0 ? fr_one() : io->in_m_u(imOnChain, u - 1); }
    __device__ __forceinline__ Fr inIdx() const { return i == 0 ? glob(g_oldLastIdx) : io->in_m_u(imOutIdx, u - 1); }
    __device__ __forceinline__ Fr globalChainID() const { return glob(g_chainID); }
    __device__ __forceinline__ Fr currentNumBatch() const { return glob(g_numBatch); }
};
struct DecExtIn {
    const UnitIO* io;
    uint32_t s_prev, s_inIdx, s_chainID, s_numBatch;
    __device__ __forceinline__ Fr previousOnChain() const { return io->in_m(s_prev); }
    __device__ __forceinline__ Fr inIdx() const { return io->in_m(s_inIdx); }
    __device__ __forceinline__ Fr globalChainID() const { return io->in_m(s_chainID); }
    __device__ __forceinline__ Fr currentNumBatch() const { return io->in_m(s_numBatch); }
};

// the neighbours' fields of RqTxVerifier (rtx_states_lane_dev): RollupMain wires them from the transactions around (src/rollup-main.circom:269-379)
struct NbMain {
    const UnitIO* io;
    uint32_t off[3];       // txCompressedDataV2, toEthAddr, toBjjAy
    uint32_t u, i, nTx;
    __device__ __forceinline__ Fr fut(int m, int j) const { return i + j + 1 < nTx ? io->in_m_u(off[m], u + j + 1) : fr_zero(); }
    __device__ __forceinline__ Fr past(int m, int j) const { return (int)i - j - 1 >= 0 ? io->in_m_u(off[m], u - j - 1) : fr_zero(); }
};
struct NbRtx {             // standalone RollupTx: they are inputs
    const UnitIO* io;
    uint32_t f[3], p[3];
    __device__ __forceinline__ Fr fut(int m, int j) const { return io->in_m(f[m] + j); }
    __device__ __forceinline__ Fr past(int m, int j) const { return io->in_m(p[m] + j); }
};

// Four lanes per transaction (blockIdx.y): 0 = RollupMain's boolean checks, DecodeTx and the im* checks on its outputs; 1..3 = the three
// lanes of the RollupTx front (tx_dev.h: states, mux, balance), which take the few DecodeTx outputs they consume straight from the input
// bits (decode_fields_dev). The lanes share no signal; a single batch has 32 wavefronts per lane and the kernel is the head of both of
// its critical paths. The 256 fromBjjCompressed input rows (8 KB per transaction) are read by the mux lane only: it packs them into the
// key (its own job), checks them boolean (RollupMain phase A) and stores DecodeTx's L1TxFullData rows bit * onChain as copies.
#ifndef HZ_FRONT_WAVES
#define HZ_FRONT_WAVES 2
#endif
#ifndef HZ_SIGHASH_WAVES
#define HZ_SIGHASH_WAVES 2
#endif
#ifndef HZ_FRONT_BLOCK
#define HZ_FRONT_BLOCK HZ_BLOCK
#endif
__global__ __launch_bounds__(HZ_FRONT_BLOCK) __attribute__((amdgpu_waves_per_eu(HZ_FRONT_WAVES))) void k_main_front(const MainFrontArgs a) {
    const Fr* K7 = poseidon_consts_w<7>();
    const uint32_t li = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t n_units = a.B * a.nTx;
    if (li >= (a.ucnt ? a.ucnt : n_units)) return;
    const uint32_t u = a.u0 + li;                 // global unit = batch * nTx + transaction
    const uint32_t b = u / a.nTx, i = u % a.nTx;
    const UnitIO io{a.tx_base, n_units, u, b, i, a.err};
    const Scratch sc{a.scratch, n_units, u};
    const MainTxInOff& m = a.mi;
    auto glob = [&](uint32_t sig) __attribute__((always_inline)) { return fr_from_canon(load_fr(a.glob_base + ((size_t)sig * a.B + b) * 32)); };
    const Fr one = fr_one();
    if (blockIdx.y == 0) {
        // A (src/rollup-main.circom:207-219)
        auto bool_chk = [&](int cid, const Fr& v) __attribute__((always_inline)) { io.chk_zero(cid, fr_mul(v, fr_sub(v, one))); };
        if (i + 1 < a.nTx) bool_chk(C_MAIN_IMONCHAIN_BOOL, io.in_m(m.imOnChain));
        bool_chk(C_MAIN_ONCHAIN_BOOL, io.in_m(m.onChain));
        bool_chk(C_MAIN_NEWACCOUNT_BOOL, io.in_m(m.newAccount));
        // (the boolean check of the 256 fromBjjCompressed bits and their L1TxFullData rows: the mux lane, which reads them anyway)
        bool_chk(C_MAIN_ISOLD0_1_BOOL, io.in_m(m.isOld0_1));
        bool_chk(C_MAIN_ISOLD0_2_BOOL, io.in_m(m.isOld0_2));
        // B
        const DecExtMain ext{&io, a.glob_base, a.B, b, i, u, m.imOnChain, m.imOutIdx, a.g.oldLastIdx, a.g.globalChainID, a.g.currentNumBatch};
        const DecResult d = decode_tx_dev<false>(io, a.dec, m, (int)a.L, ext, K7, false, false);
        // C (:258-265)
        io.chk(C_MAIN_IM_V2, d.v2, io.in_m(m.txCompressedDataV2));
        if (i + 1 < a.nTx) {
            io.chk(C_MAIN_IM_ONCHAIN, io.in_m(m.onChain), io.in_m(m.imOnChain));
            io.chk(C_MAIN_IM_OUTIDX, d.outIdx, io.in_m(m.imOutIdx));
        }
        sc.set(SC_OUTIDX, d.outIdx);   // (sigL2Hash: k_main_sighash)
        return;
    }
    // D: wiring (:269-379)
    RtxExt x;
    decode_fields_dev(io, m, x);
    if (blockIdx.y == 1) {
        const NbMain nb{&io, {m.txCompressedDataV2, m.toEthAddr, m.toBjjAy}, u, i, a.nTx};
        auto xs = [&]() __attribute__((always_inline)) {
            RtxExt y;
            decode_fields_dev(io, m, y);
            y.oldStateRoot = i == 0 ? glob(a.g.oldStateRoot) : io.in_m_u(m.imStateRoot, u - 1);
            y.oldExitRoot = i == 0 ? fr_zero() : io.in_m_u(m.imExitRoot, u - 1);
            return y;
        };
        rtx_states_lane_dev(io, sc, a.rtx, m, xs, nb, false);
    } else if (blockIdx.y == 2) {
        rtx_mux_lane_dev(io, sc, a.rtx, m, x, a.dec.l1full, C_MAIN_BJJ_BOOL);
    } else {
        const FeeSrcMain fs{&io, a.fee_base, a.B * a.F, b * a.F, a.fi.feePlanTokens, a.fi.imFinalAccFee, m.imAccFeeOut, a.nTx, i};
        (void)rtx_balance_lane_dev<MainTxInOff, FeeSrcMain, false>(io, sc, a.rtx, m, [&]() __attribute__((always_inline)) { RtxExt y; decode_fields_dev(io, m, y); return y; }, (int)a.F, fs);
    }
}

// DecodeTx's sigL2Hash for RollupMain, lane = transaction: one Poseidon of width 7 over six INPUTS (decode_sig_hash_dev) -- nothing of the
// front step feeds it and only the signature prologue reads it, so it runs at the head of the signature stream, beside the front kernel
// instead of inside its DecodeTx lane (where its 63 registers of state sat on top of everything DecodeTx keeps alive).
__global__ __launch_bounds__(HZ_BLOCK) __attribute__((amdgpu_waves_per_eu(HZ_SIGHASH_WAVES))) void k_main_sighash(const MainFrontArgs a) {
    const Fr* K7 = poseidon_consts_w<7>();
    const uint32_t li = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t n_units = a.B * a.nTx;
    if (li >= (a.ucnt ? a.ucnt : n_units)) return;
    const uint32_t u = a.u0 + li;
    const UnitIO io{a.tx_base, n_units, u, u / a.nTx, u % a.nTx, a.err};
    const Scratch sc{a.scratch, n_units, u};
    sc.set(SC_SIGL2HASH, decode_sig_hash_dev(io, a.dec, a.mi, K7));
}

// RollupTx's FeeAccumulator for RollupMain, lane = transaction: needs the fee the transaction pays and its token (scratch, from the
// front kernel), the fee plan and the accumulated fees before it (inputs); feeds only its own signals and RollupMain's im* checks.
// A kernel of its own on the fixed-base signature stream: half of what k_main_front used to do, off the two chains that kernel heads.
__global__ __launch_bounds__(HZ_BLOCK) __attribute__((amdgpu_waves_per_eu(2))) void k_main_feeacc(const MainFrontArgs a) {
    const uint32_t li = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t n_units = a.B * a.nTx;
    if (li >= (a.ucnt ? a.ucnt : n_units)) return;
    const uint32_t u = a.u0 + li;
    const uint32_t b = u / a.nTx, i = u % a.nTx;
    const UnitIO io{a.tx_base, n_units, u, b, i, a.err};
    const Scratch sc{a.scratch, n_units, u};
    const FeeSrcMain fs{&io, a.fee_base, a.B * a.F, b * a.F, a.fi.feePlanTokens, a.fi.imFinalAccFee, a.mi.imAccFeeOut, a.nTx, i};
    fee_accumulator_dev(io, a.rtx, (int)a.F, fs, sc.get(SC_FEE2CHARGE), sc.get(SC_FA_TOKEN));
}


__global__ __launch_bounds__(HZ_BLOCK) __attribute__((amdgpu_waves_per_eu(2))) void k_rtx_front(const RtxFrontArgs a) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= a.N) return;
    const UnitIO io{a.base, a.N, i, i, 0, a.err};
    const Scratch sc{a.scratch, a.N, i};
    const RtxInOff& r = a.in;
    RtxExt x;
    x.fromIdx = io.in_m(r.fromIdx); x.toIdx = io.in_m(r.toIdx); x.toBjjSign = io.in_m(r.toBjjSign); x.amount = io.in_m(r.amount);
    x.tokenID = io.in_m(r.tokenID); x.nonce = io.in_m(r.nonce); x.userFee = io.in_m(r.userFee); x.sigL2Hash = io.in_m(r.sigL2Hash);
    x.oldStateRoot = io.in_m(r.oldStateRoot); x.oldExitRoot = io.in_m(r.oldExitRoot);
    if (blockIdx.y == 0) {   // the same three lanes as k_main_front's (tx_dev.h)
        io.put_u64(0, 1);  // main.one
        const NbRtx nb{&io, {r.futureV2, r.futureToEthAddr, r.futureToBjjAy}, {r.pastV2, r.pastToEthAddr, r.pastToBjjAy}};
        rtx_states_lane_dev(io, sc, a.rtx, r, [&]() __attribute__((always_inline)) { return x; }, nb, true);
    } else if (blockIdx.y == 1) {
        rtx_mux_lane_dev(io, sc, a.rtx, r, x, ~0u, -1);
    } else {
        const FeeSrcRtx fs{&io, r.feePlanTokens, r.accFeeIn, a.rtx.o_accFeeOut};
        const FrontOut fo = rtx_balance_lane_dev<RtxInOff, FeeSrcRtx, true>(io, sc, a.rtx, r, [&]() __attribute__((always_inline)) { return x; }, (int)a.F, fs);
        io.put_m(r.o_isAmountNullified, fo.isAmountNullified);
    }
}

__global__ __launch_bounds__(HZ_BLOCK) __attribute__((amdgpu_waves_per_eu(2))) void k_dec_main(const DecMainArgs a) {
    const Fr* K7 = poseidon_consts_w<7>();
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= a.N) return;
    const UnitIO io{a.base, a.N, i, i, 0, a.err};
    io.put_u64(0, 1);
    const DecExtIn ext{&io, a.in.previousOnChain, a.in.inIdx, a.in.globalChainID, a.in.currentNumBatch};
    (void)decode_tx_dev<true>(io, a.dec, a.in, (int)a.L, ext, K7);
}

__global__ __launch_bounds__(HZ_BLOCK) void k_rtx_back(const RtxBackArgs a) {
    const uint32_t li = blockIdx.x * blockDim.x + threadIdx.x;
    if (li >= (a.ucnt ? a.ucnt : a.n_units)) return;
    const uint32_t u = a.u0 + li * (a.ustride > 1 ? a.ustride : 1u);
    if (a.skip_mod && u % a.skip_mod == a.skip_mod - 1) return;   // evaluated by the early-tail launch of this step (ctx.hip)
    const uint32_t b = u / a.upi, i = u % a.upi;
    const UnitIO io{a.base, a.n_units, u, b, i, a.err};
    const Scratch sc{a.scratch, a.n_units, u};
    const Fr isExit = sc.get(SC_ISEXIT), oldExitRoot = sc.get(SC_OLDEXITROOT);
    const Fr p1root = smt_top_dev(io, sc, a.p[0], sc.get(SC_OLDSTATEROOT), C_RTX_P1_OLDROOT, C_RTX_P1_KEYS);
    const Fr s3 = mux1_dev(p1root, oldExitRoot, isExit);
    io.put_m(a.s3, s3);
    const Fr p2root = smt_top_dev(io, sc, a.p[1], s3, C_RTX_P2_OLDROOT, C_RTX_P2_KEYS);
    const Fr newStateRoot = mux1_dev(p2root, p1root, isExit), newExitRoot = mux1_dev(oldExitRoot, p2root, isExit);
    io.put_m(a.s4, newStateRoot); io.put_m(a.s5, newExitRoot);
    if (a.is_main) {
        // E / G (src/rollup-main.circom:383-389,427)
        if (i + 1 < a.upi) {
            io.chk(C_MAIN_IM_STATEROOT, newStateRoot, io.in_m(a.im_stateroot));
            io.chk(C_MAIN_IM_EXITROOT, newExitRoot, io.in_m(a.im_exitroot));
        } else {
            io.chk(C_MAIN_IM_INITFEEROOT, newStateRoot, fr_from_canon(load_fr(a.glob_base + ((size_t)a.g_initfeeroot * a.B + b) * 32)));
        }
        // H (:456-459): amountF bits of L1L2TxData times (1 - isAmountNullified) -- unless k_da_mask has it this step
        if (a.skip_h) return;
        const Fc keep_c = fr_to_canon(fr_sub(fr_one(), sc.get(SC_ISAMTNULL)));
        for (int j = 0; j < 40; j++) {
            // L1L2TxData[2L + 40 - 1 - k] = n2bAmount.out[k]
            const Fc b = io.in_c(a.n2bAmount + (39 - j));
            io.put_c(a.main_l1l2amt + j, b.v[0] ? keep_c : fc_zero());
        }
    } else {
        io.put_m(a.o_newStateRoot, newStateRoot); io.put_m(a.o_newExitRoot, newExitRoot);
    }
}

// RollupMain phase H alone (src/rollup-main.circom:456-459): the amount bits of L1L2TxData times (1 - isAmountNullified). Everything
// it needs exists after the front kernel, and HashInputs needs it from every transaction: launched early so that the SHA-256 chain
// does not have to wait for the SMT chains of 2048 transactions when it only depends on the last one's exit root (ctx.hip).
__global__ __launch_bounds__(HZ_BLOCK) void k_da_mask(const RtxBackArgs a) {
    const uint32_t u = blockIdx.x * blockDim.x + threadIdx.x;
    if (u >= a.n_units) return;
    const UnitIO io{a.base, a.n_units, u, u / a.upi, u % a.upi, a.err};
    const Scratch sc{a.scratch, a.n_units, u};
    const Fc keep_c = fr_to_canon(fr_sub(fr_one(), sc.get(SC_ISAMTNULL)));
    for (int j = 0; j < 40; j++) {
        const Fc b = io.in_c(a.n2bAmount + (39 - j));
        io.put_c(a.main_l1l2amt + j, b.v[0] ? keep_c : fc_zero());
    }
}

// ---------------------------------------------------------------------------------------------------
// host launchers
static inline dim3 grid1(uint32_t n) { return dim3((n + HZ_BLOCK - 1) / HZ_BLOCK); }

hipError_t launch_main_front(const MainFrontArgs& a, hipStream_t s) {
    const uint32_t nl = a.ucnt ? a.ucnt : a.B * a.nTx;
    dim3 g((nl + HZ_FRONT_BLOCK - 1) / HZ_FRONT_BLOCK);
    g.y = 4;   // DecodeTx lane, the three RollupTx-front lanes
    hipLaunchKernelGGL(k_main_front, g, dim3(HZ_FRONT_BLOCK), 0, s, a);
    return hipGetLastError();
}
hipError_t launch_main_sighash(const MainFrontArgs& a, hipStream_t s) {
    const uint32_t nl = a.ucnt ? a.ucnt : a.B * a.nTx;
    hipLaunchKernelGGL(k_main_sighash, grid1(nl), dim3(HZ_BLOCK), 0, s, a);
    return hipGetLastError();
}
hipError_t launch_main_feeacc(const MainFrontArgs& a, hipStream_t s) {
    const uint32_t nl = a.ucnt ? a.ucnt : a.B * a.nTx;
    hipLaunchKernelGGL(k_main_feeacc, grid1(nl), dim3(HZ_BLOCK), 0, s, a);
    return hipGetLastError();
}
hipError_t launch_rtx_front(const RtxFrontArgs& a, hipStream_t s) {
    dim3 g = grid1(a.N);
    g.y = 3;   // states, mux, balance
    hipLaunchKernelGGL(k_rtx_front, g, dim3(HZ_BLOCK), 0, s, a);
    return hipGetLastError();
}
hipError_t launch_dec_main(const DecMainArgs& a, hipStream_t s) {
    hipLaunchKernelGGL(k_dec_main, grid1(a.N), dim3(HZ_BLOCK), 0, s, a);
    return hipGetLastError();
}
hipError_t launch_rtx_back(const RtxBackArgs& a, hipStream_t s) {
    hipLaunchKernelGGL(k_rtx_back, grid1(a.ucnt ? a.ucnt : a.n_units), dim3(HZ_BLOCK), 0, s, a);
    return hipGetLastError();
}
hipError_t launch_da_mask(const RtxBackArgs& a, hipStream_t s) {
    hipLaunchKernelGGL(k_da_mask, dim3((a.n_units + HZ_BLOCK - 1) / HZ_BLOCK), dim3(HZ_BLOCK), 0, s, a);
    return hipGetLastError();
}

}  // namespace hz
