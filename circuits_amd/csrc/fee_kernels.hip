// FeeTx (reference src/fee-tx.circom:26-112), HashState as main (src/lib/hash-state.circom:18-40),
// HashInputs (src/hash-inputs.circom:23-185) and Withdraw (src/withdraw.circom:21-176).
// (the field routines inlined here too since round 6: an out-of-line fr_mul takes its operands by reference, 36 bytes of private memory
// each -- 270-340 B per lane in k_fee_front / k_fee_back / k_withdraw; the kernels are latency / store bound either way)
#define HZ_FR_INLINE
#include <hip/hip_runtime.h>
#include "kernels.h"
#include "sha_dev.h"
#include <algorithm>
#include "smt_dev.h"

#ifndef HZ_SHA_CHAIN_PRIO
#define HZ_SHA_CHAIN_PRIO 0
#endif
namespace hz {

// ---- FeeTx front: lane = fee tx ---------------------------------------------------------------
__global__ __launch_bounds__(HZ_BLOCK) void k_fee_front(const FeeFrontArgs a) {
    const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= a.n_units) return;
    const uint32_t b = j / a.upi, jj = j % a.upi;   // batch, fee slot inside the batch
    const UnitIO io{a.base, a.n_units, j, b, jj, a.err};
    const Scratch sc{a.scratch, a.n_units, j};
    const Fr one = fr_one(), zero = fr_zero();
    if (!a.is_main) io.put_u64(0, 1);
    const Fr feeIdx = io.in_m(a.in_feeIdx), feePlanToken = io.in_m(a.in_feePlanToken), tokenID = io.in_m(a.in_tokenID);
    Fr z[2] = {feeIdx, fr_sub(tokenID, feePlanToken)};   // tokenIDChecker: in[0] = feePlanToken, in[1] = tokenID
    Fr zi[2] = {z[0], z[1]};
    inv_pair(zi[0], zi[1]);
    const Fr fz = is_zero_dev(io, a.fee.feeIdxIsZero, z[0], zi[0]);
    const Fr e = is_zero_dev(io, a.fee.tokenIDChecker, z[1], zi[1]);
    io.chk_zero(C_FEE_TOKENID, fr_mul(fr_sub(one, e), fr_sub(one, fz)));
    const Fr nonce = io.in_m(a.in_nonce), sign = io.in_m(a.in_sign), balance = io.in_m(a.in_balance), ay = io.in_m(a.in_ay),
             ethAddr = io.in_m(a.in_ethAddr), accFee = io.in_m(a.in_accFee);
    const Fr e0 = fr_add(fr_add(tokenID, fr_mul(nonce, m_pow2(32))), fr_mul(sign, m_pow2(72)));
    sc.set(SC_HS_IN + 0, e0); sc.set(SC_HS_IN + 1, balance); sc.set(SC_HS_IN + 2, ay); sc.set(SC_HS_IN + 3, ethAddr);
    sc.set(SC_HS_IN + 8, e0); sc.set(SC_HS_IN + 9, fr_add(accFee, balance)); sc.set(SC_HS_IN + 10, ay); sc.set(SC_HS_IN + 11, ethAddr);
    sc.set(SC_KEY_S1OLD, feeIdx); sc.set(SC_KEY_1, feeIdx);
    sc.set(SC_P1_FNC0, zero); sc.set(SC_P1_FNC1, fr_sub(one, fz)); sc.set(SC_ISOLD0_1, zero);
    Fr oldRoot;
    if (a.is_main) {
        oldRoot = jj == 0 ? fr_from_canon(load_fr(a.glob_base + ((size_t)a.g_initfeeroot * a.B + b) * 32)) : io.in_m_u(a.im_stateRootFee, j - 1);
    } else {
        oldRoot = io.in_m(a.in_oldStateRoot);
    }
    sc.set(SC_OLDSTATEROOT, oldRoot);
}

__global__ __launch_bounds__(HZ_BLOCK) void k_fee_back(const FeeBackArgs a) {
    const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= a.n_units) return;
    const UnitIO io{a.base, a.n_units, j, j / a.upi, j % a.upi, a.err};
    const Scratch sc{a.scratch, a.n_units, j};
    const Fr root = smt_top_dev(io, sc, a.p, sc.get(SC_OLDSTATEROOT), C_FEE_P_OLDROOT, C_FEE_P_KEYS);
    sc.set(SC_ROOT_P2NEW, root);  // feeTx.newStateRoot for HashInputs
    if (a.is_main) {
        if (j % a.upi + 1 < a.upi) io.chk(C_MAIN_IM_FEEROOT, root, io.in_m(a.im_stateRootFee));
    } else {
        io.put_m(a.o_newStateRoot, root);
    }
}

// ---- HashState as main component ------------------------------------------------------------------
struct HsMainArgs { uint8_t* base; uint32_t N; HashStateOff hs; };
__global__ __launch_bounds__(HZ_BLOCK) void k_hash_state_main(const HsMainArgs a) {
    const Fr* K5 = poseidon_consts_w<5>();
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= a.N) return;
    const UnitIO io{a.base, a.N, i, i, 0, nullptr};
    const HashStateOff& h = a.hs;
    io.put_u64(h.one, 1);
    Fr hin[4];
    hin[0] = fr_add(fr_add(io.in_m(h.tokenID), fr_mul(io.in_m(h.nonce), m_pow2(32))), fr_mul(io.in_m(h.sign), m_pow2(72)));
    hin[1] = io.in_m(h.balance); hin[2] = io.in_m(h.ay); hin[3] = io.in_m(h.ethAddr);
    WitSboxSink s5 = io.sbox_sink(h.hash);
    io.put_m(h.out, poseidon_hash<5>(hin, K5, s5));
}

// ---- HashInputs --------------------------------------------------------------------------------------
// k_hi_prep: lane 0 = header/tail fields (+ their Num2Bits signals), then one lane per L1 slot,
// per transaction and per fee transaction; every lane ORs its bits into the zeroed message buffer.
__device__ __forceinline__ void msg_put_be(uint32_t* msgw, uint64_t pos, const Fc& canon, int n) {
    for (int k = 0; k < n; k++) msg_set_bit(msgw, pos + k, c_bit(canon, n - 1 - k));
}

__global__ __launch_bounds__(HZ_BLOCK) void k_hi_prep(const HashInputsArgs a) {
    const uint32_t gt = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t L = a.L, nTx = a.nTx, maxL1 = a.maxL1, Fn = a.F, B = a.B;
    const uint32_t n_items = 1 + maxL1 + nTx + Fn;
    if (gt >= (a.prep_part == 2 ? 1u : n_items) * B) return;
    const uint32_t bt = a.prep_part == 2 ? gt : gt / n_items, t = a.prep_part == 2 ? 0u : gt % n_items;   // batch, item
    if (a.prep_part == 1 && t == 0) return;
    const uint32_t txU = B * nTx, feeU = B * Fn, tx0 = bt * nTx, fee0 = bt * Fn;
    uint32_t* msgw = reinterpret_cast<uint32_t*>(a.msg) + (size_t)bt * a.hi.sha.nblocks * 16;
    const HashInputsOff& o = a.hi;
    const uint64_t offL1 = 2 * 48 + 3 * 256, offL2 = offL1 + (uint64_t)maxL1 * L1FULL_BITS, offFee = offL2 + (uint64_t)nTx * (2 * L + 48);
    const uint64_t offTail = offFee + (uint64_t)Fn * L;
    const UnitIO hio{a.hi_base, B, bt, bt, 0, a.err};
    if (t == 0) {
        Fc oldLastIdx, newLastIdx, oldStateRoot, newStateRoot, newExitRoot, chainID, batch;
        if (a.is_main) {
            auto glob = [&](uint32_t sig) __attribute__((always_inline)) { return load_fr(a.glob_base + ((size_t)sig * B + bt) * 32); };
            oldLastIdx = glob(a.g.oldLastIdx); oldStateRoot = glob(a.g.oldStateRoot); chainID = glob(a.g.globalChainID); batch = glob(a.g.currentNumBatch);
            newLastIdx = fr_to_canon(a.tx_scratch[(size_t)SC_OUTIDX * txU + tx0 + (nTx - 1)]);
            newStateRoot = fr_to_canon(a.fee_scratch[(size_t)SC_ROOT_P2NEW * feeU + fee0 + (Fn - 1)]);
            newExitRoot = load_fr(a.tx_base + ((size_t)a.rtx_s5 * txU + tx0 + (nTx - 1)) * 32);
        } else {
            hio.put_u64(o.one, 1);
            oldLastIdx = hio.in_c(o.i_oldLastIdx); newLastIdx = hio.in_c(o.i_newLastIdx); oldStateRoot = hio.in_c(o.i_oldStateRoot);
            newStateRoot = hio.in_c(o.i_newStateRoot); newExitRoot = hio.in_c(o.i_newExitRoot); chainID = hio.in_c(o.i_globalChainID);
            batch = hio.in_c(o.i_currentNumBatch);
        }
        auto idx48 = [&](uint32_t off, const Fc& v) __attribute__((always_inline)) {
            num2bits_dev(hio, off, v, 48, C_HI_N2B);
            uint32_t pad = 0;
            for (uint32_t i = L; i < 48; i++) pad += c_bit(v, i);
            if (pad) report_fail(hio.err, bt, 0, C_HI_PAD, fr_from_u64(pad), fr_zero());
        };
        idx48(o.n2bOldLastIdx, oldLastIdx);
        idx48(o.n2bNewLastIdx, newLastIdx);
        // Num2Bits(256) of a field element never fails (value < r < 2^254)
        for (int k = 0; k < 256; k++) { hio.put_bit(o.n2bOldStateRoot + k, c_bit(oldStateRoot, k)); hio.put_bit(o.n2bNewStateRoot + k, c_bit(newStateRoot, k));
                                        hio.put_bit(o.n2bNewExitRoot + k, c_bit(newExitRoot, k)); }
        num2bits_dev(hio, o.n2bChainID, chainID, 16, C_HI_N2B);
        num2bits_dev(hio, o.n2bCurrentNumBatch, batch, 32, C_HI_N2B);
        msg_put_be(msgw, 0, oldLastIdx, 48); msg_put_be(msgw, 48, newLastIdx, 48); msg_put_be(msgw, 96, oldStateRoot, 256);
        msg_put_be(msgw, 352, newStateRoot, 256); msg_put_be(msgw, 608, newExitRoot, 256);
        msg_put_be(msgw, offTail, chainID, 16); msg_put_be(msgw, offTail + 16, batch, 32);
        // padding: 1 bit, zeros, 64-bit length
        const uint64_t nbits = o.totalBits;
        msg_set_bit(msgw, nbits, 1);
        const uint64_t total = (uint64_t)o.sha.nblocks * 512;
        for (int k = 0; k < 64; k++) msg_set_bit(msgw, total - 1 - k, (uint32_t)((nbits >> k) & 1));
        return;
    }
    uint32_t u = t - 1;
    if (u < maxL1) {   // L1TxsFullData slot u: the stored products are bit*onChain
        for (uint32_t k = 0; k < L1FULL_BITS; k++) {
            uint32_t bit;
            if (a.is_main) bit = (u < nTx) ? (load_fr(a.tx_base + ((size_t)(a.dec.l1full + k) * txU + tx0 + u) * 32).v[0] & 1u) : 0u;
            else bit = hio.in_c(o.i_L1TxsFullData + u * L1FULL_BITS + k).v[0] & 1u;
            msg_set_bit(msgw, offL1 + (uint64_t)u * L1FULL_BITS + k, bit);
        }
        return;
    }
    u -= maxL1;
    if (u < nTx) {
        const uint64_t pos = offL2 + (uint64_t)u * (2 * L + 48);
        if (a.is_main) {
            auto txs = [&](uint32_t sig) __attribute__((always_inline)) { return load_fr(a.tx_base + ((size_t)sig * txU + tx0 + u) * 32).v[0] & 1u; };
            for (uint32_t k = 0; k < L; k++) msg_set_bit(msgw, pos + (L - 1 - k), txs(a.dec.n2bData + 48 + k));
            for (uint32_t k = 0; k < L; k++) msg_set_bit(msgw, pos + (2 * L - 1 - k), txs(a.dec.n2bFinalToIdx + k));
            for (uint32_t k = 0; k < 40; k++) msg_set_bit(msgw, pos + 2 * L + k, txs(a.rtx_main_l1l2amt + k));
            for (uint32_t k = 0; k < 8; k++) msg_set_bit(msgw, pos + 2 * L + 40 + k, txs(a.dec.l1l2Fee + k));
        } else {
            for (uint32_t k = 0; k < 2 * L + 48; k++) msg_set_bit(msgw, pos + k, hio.in_c(o.i_L1L2TxsData + u * (2 * L + 48) + k).v[0] & 1u);
        }
        return;
    }
    u -= nTx;
    {
        const Fc v = a.is_main ? load_fr(a.fee_base + ((size_t)a.fi_feeIdxs * feeU + fee0 + u) * 32) : hio.in_c(o.i_feeTxsData + u);
        num2bits_dev(hio, o.n2bFee + 48 * u, v, 48, C_HI_N2B);
        uint32_t pad = 0;
        for (uint32_t i = L; i < 48; i++) pad += c_bit(v, i);
        if (pad) report_fail(hio.err, bt, 0, C_HI_PAD, fr_from_u64(pad), fr_zero());
        msg_put_be(msgw, offFee + (uint64_t)u * L, v, (int)L);
    }
}

// Sequential chaining values of blocks [blk0, blk1): chain[b] = state before block b; after the last block the digest -> output signal.
// One lane per batch (the chains of different batches are independent), 766 dependent compressions at nTx = 2048: the critical
// path of a step's tail. Alone the lane is bound by its own instruction stream; beside the integer-bound kernels of the other
// context it used to wait on every block's message load (HBM latency under a store stream: 3.7 ms alone became 16 ms). The
// workgroup is therefore four wavefronts: all of them prefetch the next HZ_SHA_CHUNK blocks of every batch into LDS (global ->
// registers before the compute, registers -> LDS after it) while wavefront 0 walks the current chunk out of LDS.
#define HZ_SHA_CHUNK 8
#define HZ_SHA_ROW (HZ_SHA_CHUNK * 16 + 1)   // words per batch and chunk, odd: lanes reading the same word of their rows hit different banks
__global__ __launch_bounds__(256) void k_sha_chain(const HashInputsArgs a) {
    __shared__ uint32_t buf[2][64 * HZ_SHA_ROW];
    const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const uint32_t bt0 = blockIdx.x * 64;              // first batch of this workgroup
    const uint32_t nbt = min(64u, a.B - bt0);          // batches of this workgroup
    const int nb = a.hi.sha.nblocks;
    const int blk0 = (int)a.blk0, blk1 = (int)a.blk1;
    const uint4* msg4 = reinterpret_cast<const uint4*>(a.msg);
    if (wave == 0) __builtin_amdgcn_s_setprio(3);
    // prefetch: item q = (batch l, uint4 w of the chunk's 8 x 4): the 512 bytes of a batch's chunk are contiguous in the message
    uint4 pre[8];
    auto fetch = [&](int c0) __attribute__((always_inline)) {
#pragma unroll
        for (int r = 0; r < 8; r++) {
            const uint32_t q = tid + 256u * r, l = q >> 5, w = q & 31;
            const int b = c0 + (int)(w >> 2);
            pre[r] = (l < nbt && b < blk1) ? msg4[((size_t)(bt0 + l) * nb + b) * 4 + (w & 3)] : make_uint4(0u, 0u, 0u, 0u);
        }
    };
    auto stash = [&](int which) __attribute__((always_inline)) {
#pragma unroll
        for (int r = 0; r < 8; r++) {
            const uint32_t q = tid + 256u * r, l = q >> 5, w = q & 31;
            uint32_t* d = &buf[which][l * HZ_SHA_ROW + w * 4];
            d[0] = pre[r].x; d[1] = pre[r].y; d[2] = pre[r].z; d[3] = pre[r].w;
        }
    };
    fetch(blk0);
    stash(0);
    __syncthreads();
    const bool active = wave == 0 && lane < nbt;
    const uint32_t bt = bt0 + lane;
    uint32_t* chain = a.chain + (size_t)bt * (nb + 1) * 8;
    uint32_t hv[8];
    if (active)
        for (int i = 0; i < 8; i++) hv[i] = blk0 == 0 ? SHA_H0[i] : chain[8 * blk0 + i];
    int which = 0;
    for (int c0 = blk0; c0 < blk1; c0 += HZ_SHA_CHUNK) {
        const bool more = c0 + HZ_SHA_CHUNK < blk1;
        if (more) fetch(c0 + HZ_SHA_CHUNK);
        if (active) {
            const int cnt = min(HZ_SHA_CHUNK, blk1 - c0);
            for (int j = 0; j < cnt; j++) {
                for (int i = 0; i < 8; i++) chain[8 * (c0 + j) + i] = hv[i];
                uint32_t w16[16];
                for (int i = 0; i < 16; i++) w16[i] = buf[which][lane * HZ_SHA_ROW + j * 16 + i];
                sha256_compress(hv, w16);
            }
        }
        if (more) stash(which ^ 1);
        __syncthreads();
        which ^= 1;
    }
    if (!active) return;
    if (blk1 < nb) {   // the next group's launch continues from here
        for (int i = 0; i < 8; i++) chain[8 * blk1 + i] = hv[i];
        return;
    }
    const Fc out = sha_digest_to_fr(hv);
    if (a.is_main) store_fr(a.glob_base + ((size_t)a.g.hashGlobalInputs * a.B + bt) * 32, out);
    else store_fr(a.hi_base + ((size_t)a.hi.out * a.B + bt) * 32, out);
    if (a.is_main) {
        uint4* q = reinterpret_cast<uint4*>(a.glob_base + ((size_t)a.g.one * a.B + bt) * 32);
        q[0] = make_uint4(1u, 0u, 0u, 0u);
        q[1] = make_uint4(0u, 0u, 0u, 0u);
    }
}

// The same chain for launches of a few batches (a single batch: the chain is 766 dependent compressions of ONE lane and the tail of
// the step's critical path). The message schedule does not depend on the chaining value: wavefronts 1-3 expand the next chunk's
// blocks (one block per lane: W[0..63] + K) into LDS while lane l of wavefront 0 walks batch l's current chunk with the round
// function alone -- 4.2 -> 2.6 ms per 766 blocks.
#define HZ_SHAW_BATCH 8
#define HZ_SHAW_ROW 65   // words per expanded block, odd: the lanes of wavefront 0 read the same word of different rows
__global__ __launch_bounds__(256) void k_sha_chain_w(const HashInputsArgs a) {
    __shared__ uint32_t wk[2][HZ_SHAW_BATCH * HZ_SHA_CHUNK * HZ_SHAW_ROW];
    const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const uint32_t bt0 = blockIdx.x * HZ_SHAW_BATCH;
    const uint32_t nbt = min((uint32_t)HZ_SHAW_BATCH, a.B - bt0);
    const int nb = a.hi.sha.nblocks;
    const int blk0 = (int)a.blk0, blk1 = (int)a.blk1;
    const uint32_t* msgw = reinterpret_cast<const uint32_t*>(a.msg);
    if (wave == 0) __builtin_amdgcn_s_setprio(3);
    auto expand = [&](int c0, int which) __attribute__((always_inline)) {
        if (wave == 0) return;
        for (uint32_t q = tid - 64; q < nbt * HZ_SHA_CHUNK; q += 192) {
            const uint32_t l = q / HZ_SHA_CHUNK, j = q % HZ_SHA_CHUNK;
            const int b = c0 + (int)j;
            if (b >= blk1) continue;
            const uint32_t* src = msgw + ((size_t)(bt0 + l) * nb + b) * 16;
            uint32_t* dst = &wk[which][(l * HZ_SHA_CHUNK + j) * HZ_SHAW_ROW];
            uint32_t w[64];
            for (int t = 0; t < 16; t++) w[t] = src[t];
            for (int t = 16; t < 64; t++) {
                const uint32_t s0 = rotr32(w[t - 15], 7) ^ rotr32(w[t - 15], 18) ^ (w[t - 15] >> 3);
                const uint32_t s1 = rotr32(w[t - 2], 17) ^ rotr32(w[t - 2], 19) ^ (w[t - 2] >> 10);
                w[t] = s1 + w[t - 7] + s0 + w[t - 16];
            }
            for (int t = 0; t < 64; t++) dst[t] = w[t] + SHA_K[t];
        }
    };
    if (wave == 0) {   // nothing to walk yet: wavefront 0 helps with the first chunk
        for (uint32_t q = tid; q < nbt * HZ_SHA_CHUNK; q += 64) {
            const uint32_t l = q / HZ_SHA_CHUNK, j = q % HZ_SHA_CHUNK;
            const int b = blk0 + (int)j;
            if (b >= blk1) continue;
            const uint32_t* src = msgw + ((size_t)(bt0 + l) * nb + b) * 16;
            uint32_t* dst = &wk[0][(l * HZ_SHA_CHUNK + j) * HZ_SHAW_ROW];
            uint32_t w[64];
            for (int t = 0; t < 16; t++) w[t] = src[t];
            for (int t = 16; t < 64; t++) {
                const uint32_t s0 = rotr32(w[t - 15], 7) ^ rotr32(w[t - 15], 18) ^ (w[t - 15] >> 3);
                const uint32_t s1 = rotr32(w[t - 2], 17) ^ rotr32(w[t - 2], 19) ^ (w[t - 2] >> 10);
                w[t] = s1 + w[t - 7] + s0 + w[t - 16];
            }
            for (int t = 0; t < 64; t++) dst[t] = w[t] + SHA_K[t];
        }
    }
    __syncthreads();
    const bool active = wave == 0 && lane < nbt;
    const uint32_t bt = bt0 + lane;
    uint32_t* chain = a.chain + (size_t)bt * (nb + 1) * 8;
    uint32_t hv[8];
    if (active)
        for (int i = 0; i < 8; i++) hv[i] = blk0 == 0 ? SHA_H0[i] : chain[8 * blk0 + i];
    int which = 0;
    for (int c0 = blk0; c0 < blk1; c0 += HZ_SHA_CHUNK) {
        if (c0 + HZ_SHA_CHUNK < blk1) expand(c0 + HZ_SHA_CHUNK, which ^ 1);
        if (active) {
            const int cnt = min(HZ_SHA_CHUNK, blk1 - c0);
            for (int j = 0; j < cnt; j++) {
                for (int i = 0; i < 8; i++) chain[8 * (c0 + j) + i] = hv[i];
                const uint32_t* w = &wk[which][(lane * HZ_SHA_CHUNK + j) * HZ_SHAW_ROW];
                uint32_t x0 = hv[0], x1 = hv[1], x2 = hv[2], x3 = hv[3], x4 = hv[4], x5 = hv[5], x6 = hv[6], x7 = hv[7];
#pragma unroll 8
                for (int t = 0; t < 64; t++) {
                    const uint32_t t1 = x7 + (rotr32(x4, 6) ^ rotr32(x4, 11) ^ rotr32(x4, 25)) + ((x4 & x5) ^ (~x4 & x6)) + w[t];
                    const uint32_t t2 = (rotr32(x0, 2) ^ rotr32(x0, 13) ^ rotr32(x0, 22)) + ((x0 & x1) ^ (x0 & x2) ^ (x1 & x2));
                    x7 = x6; x6 = x5; x5 = x4; x4 = x3 + t1; x3 = x2; x2 = x1; x1 = x0; x0 = t1 + t2;
                }
                hv[0] += x0; hv[1] += x1; hv[2] += x2; hv[3] += x3; hv[4] += x4; hv[5] += x5; hv[6] += x6; hv[7] += x7;
            }
        }
        __syncthreads();
        which ^= 1;
    }
    if (!active) return;
    if (blk1 < nb) {
        for (int i = 0; i < 8; i++) chain[8 * blk1 + i] = hv[i];
        return;
    }
    const Fc out = sha_digest_to_fr(hv);
    if (a.is_main) store_fr(a.glob_base + ((size_t)a.g.hashGlobalInputs * a.B + bt) * 32, out);
    else store_fr(a.hi_base + ((size_t)a.hi.out * a.B + bt) * 32, out);
    if (a.is_main) {
        uint4* q = reinterpret_cast<uint4*>(a.glob_base + ((size_t)a.g.one * a.B + bt) * 32);
        q[0] = make_uint4(1u, 0u, 0u, 0u);
        q[1] = make_uint4(0u, 0u, 0u, 0u);
    }
}

// per-block bit-level witness of blocks [blk0, blk1): HZ_SHA_PARTS lanes per (block, batch), each storing one slice of the block's
// rounds (sha_dev.h). Lane order (part, block, batch): consecutive lanes = consecutive batches (coalesced stores) and a wavefront
// works on one part. A batch has only 766 blocks: one lane per block left this 0.7 GB-per-batch store stream on 12 wavefronts per
// batch, each 29 k signals long; eight lanes per block fill the device and shorten the step's tail accordingly.
#ifndef HZ_SHA_PARTS
#define HZ_SHA_PARTS 8
#endif
#ifndef HZ_SHA_COOP
#define HZ_SHA_COOP 1   // bit rows of the SHA-256 witness stored cooperatively by half-wavefronts (sha_dev.h put_word_bits)
#endif
__global__ __launch_bounds__(HZ_BLOCK) void k_sha_expand(const HashInputsArgs a) {
    const uint32_t gt = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t nb = (uint32_t)a.hi.sha.nblocks, nblk = a.blk1 - a.blk0;
    if (gt >= nblk * a.B * HZ_SHA_PARTS) return;
    const uint32_t bt = gt % a.B, rest = gt / a.B;
    const uint32_t b = a.blk0 + rest % nblk, part = rest / nblk;
    const uint32_t* msgw = reinterpret_cast<const uint32_t*>(a.msg) + (size_t)bt * nb * 16;
    UnitIO hio{a.hi_base, a.B, bt, bt, 0, a.err};
    hio.coop32 = HZ_SHA_COOP && (a.B % 32u == 0u);   // a half-wavefront = 32 consecutive batches of one (block, part); the grid is a multiple of 32
    uint32_t hv[8], w16[16];
    for (int i = 0; i < 8; i++) hv[i] = a.chain[((size_t)bt * (nb + 1) + b) * 8 + i];
    for (int i = 0; i < 16; i++) w16[i] = msgw[16 * b + i];
    sha256_block_witness_part(hio, a.hi.sha.blocks + b * a.hi.sha.block_size, hv, w16, part, HZ_SHA_PARTS);
}

// ---- Withdraw: lane = instance -----------------------------------------------------------------------
// Two wavefronts per workgroup for the Poseidon-bound half (like k_smt, smt_kernels.hip): beside the store-bound k_withdraw_sha a
// launch of 2^16 witnesses takes 24.2-24.6 ms instead of 26.2-27.1 (2.67-2.71 M witnesses/s against 2.42-2.50 M on one box; 256:
// 26.0-26.7 ms; the SHA half at 128: 24.8-25.6 ms; profiles/r03_workgroup_ab.txt).
#ifndef HZ_WD_BLOCK
#define HZ_WD_BLOCK 128
#endif
#ifndef HZ_WDSHA_BLOCK
#define HZ_WDSHA_BLOCK HZ_BLOCK
#endif
__global__ __launch_bounds__(HZ_WD_BLOCK) void k_withdraw(const WithdrawArgs a) {
    const Fr* K5 = poseidon_consts_w<5>();
    const Fr* K4 = poseidon_consts_w<4>();
    const Fr* K3 = poseidon_consts_w<3>();
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= a.N) return;
    const UnitIO io{a.base, a.N, i, i, 0, a.err};
    const WithdrawOff& o = a.wd;
    const int n = (int)a.L + 1;
    const Fr one = fr_one(), zero = fr_zero();
    io.put_u64(o.one, 1);
    const Fc rootExit_c = io.in_c(o.rootExit), ethAddr_c = io.in_c(o.ethAddr), tokenID_c = io.in_c(o.tokenID), balance_c = io.in_c(o.balance),
             idx_c = io.in_c(o.idx);
    const Fr rootExit = fr_from_canon(rootExit_c), idx = fr_from_canon(idx_c);
    // accountState = HashState(tokenID, nonce 0, sign, balance, ay, ethAddr)
    Fr hin[4];
    hin[0] = fr_add(fr_from_canon(tokenID_c), fr_mul(io.in_m(o.sign), m_pow2(72)));
    hin[1] = fr_from_canon(balance_c); hin[2] = io.in_m(o.ay); hin[3] = fr_from_canon(ethAddr_c);
    WitSboxSink s5 = io.sbox_sink(o.accountState);
    const Fr st = poseidon_hash<5>(hin, K5, s5);
    // SMTVerifier(n): enabled = 1, fnc = 0, oldKey = oldValue = isOld0 = 0
    const SmtVerOff& v = o.ver;
    Fr h1in[3] = {zero, zero, one};
    WitSboxSink so = io.sbox_sink(v.hash1Old);
    const Fr h1old = poseidon_hash<4>(h1in, K4, so);
    (void)h1old;
    h1in[0] = idx; h1in[1] = st;
    WitSboxSink sn = io.sbox_sink(v.hash1New);
    const Fr h1new = poseidon_hash<4>(h1in, K4, sn);
    num2bits_strict_dev(io, v.n2bOld, fc_zero(), C_WD_N2B_OLD);
    num2bits_strict_dev(io, v.n2bNew, idx_c, C_WD_ALIAS_NEW);
    // SMTLevIns
    const uint64_t zmask = is_zero_run_dev<8>(io, n, [&](int k) __attribute__((always_inline)) { return io.in_m(o.siblingsState + k); }, [&](int k) __attribute__((always_inline)) { return v.isz + 2 * k; });
    if (!((zmask >> (n - 1)) & 1)) io.chk_zero(C_WD_LEVINS, fr_neg(one));
    uint64_t levmask = 0;
    {
        uint32_t done = 0;
        uint32_t li = 1u - (uint32_t)((zmask >> (n - 2)) & 1);
        if (li) levmask |= 1ull << (n - 1);
        done = li;
        for (int k = n - 2; k > 0; k--) {
            li = (1u - done) * (1u - (uint32_t)((zmask >> (k - 1)) & 1));
            if (li) levmask |= 1ull << k;
            done += li;
        }
        if (!done) levmask |= 1ull;
    }
    for (int k = 1; k <= n - 2; k++) io.put_bit(v.levIns + (k - 1), (uint32_t)((levmask >> k) & 1));
    // SMTVerifierSM with enabled = 1, fnc = 0, is0 = 0: st_top stays 1 until the insertion level,
    // where st_inew becomes 1; afterwards st_na. All values are bits.
    uint64_t topmask = 0, inewmask = 0;
    {
        uint32_t p_top = 1, p_inew = 0, p_na = 0;
        uint32_t last = 0;
        for (int k = 0; k < n; k++) {
            const uint32_t lev = (uint32_t)((levmask >> k) & 1);
            const uint32_t ptli = p_top & lev;
            const uint32_t t_top = p_top - ptli, t_inew = ptli, t_na = p_na + p_inew;
            io.put_bit(v.sm + VSM_N * k + VSM_PTLI, ptli); io.put_bit(v.sm + VSM_N * k + VSM_PTLIF, 0);
            io.put_bit(v.sm + VSM_N * k + VSM_IOLD, 0); io.put_bit(v.sm + VSM_N * k + VSM_I0, 0);
            if (t_top) topmask |= 1ull << k;
            if (t_inew) inewmask |= 1ull << k;
            if (k == n - 1) last = t_na + t_inew;
            p_top = t_top; p_inew = t_inew; p_na = t_na;
        }
        if (last != 1) io.chk(C_WD_SM_FINAL, fr_from_u64(last), one);
    }
    Fr child = zero;
    for (int k = n - 1; k >= 0; k--) {
        const uint32_t lv = v.levels + VL_SIZE * k;
        const uint32_t sel = c_bit(idx_c, k);
        const Fr sib = io.in_m(o.siblingsState + k);
        io.put_m(lv + VL_SW_AUX, sel ? fr_sub(sib, child) : zero);
        Fr h2[2];
        h2[0] = sel ? sib : child;
        h2[1] = sel ? child : sib;
        WitSboxSink sk = io.sbox_sink(lv + VL_HASH);
        Fr ph;
        if (__all(fr_is_zero(h2[0]) && fr_is_zero(h2[1]))) ph = poseidon3_zero_level(io, lv + VL_HASH);   // empty subtree (smt_dev.h)
        else ph = poseidon_hash<3>(h2, K3, sk);
        const Fr a0 = ((topmask >> k) & 1) ? ph : zero;
        const Fr root = ((inewmask >> k) & 1) ? fr_add(a0, h1new) : a0;
        io.put_m(lv + VL_AUX0, a0); io.put_u64(lv + VL_AUX1, 0); io.put_m(lv + VL_ROOT, root);
        child = root;
    }
    {
        // areKeyEquals(oldKey = 0, key); keysOk = MultiAND(4)(fnc=0, 1-isOld0=1, keq, enabled=1)
        Fr z[2] = {idx, fr_sub(rootExit, child)};   // checkRoot: in[0] = levels[0].root, in[1] = root
        Fr zi[2] = {z[0], z[1]};
        inv_pair(zi[0], zi[1]);
        const Fr keq = is_zero_dev(io, v.keyEq, z[0], zi[0]);
        io.put_u64(v.and_a, 0); io.put_m(v.and_b, keq); io.put_u64(v.and_c, 0);
        const Fr e = is_zero_dev(io, v.checkRoot, z[1], zi[1]);
        io.chk_zero(C_WD_ROOT, fr_sub(one, e));
    }
}

// HashInputsWithdrawal (reference src/withdraw.circom:73-176): bit decompositions, the 688-bit message and the bit-level
// witness of its two SHA-256 blocks. Lane = (instance, block): 86 % of a Withdraw witness is this store stream, which runs
// beside the Poseidon-bound k_withdraw on a second stream instead of behind it in the same lane.
// Lane granularity measured on 2^16 instances per launch: one lane per instance (both blocks) 1.915 M witnesses/s, per block 1.910 M,
// two / four lanes per block (unstored rounds recomputed) 1.76-1.83 M: more concurrent store streams are slower, the ~4.5 TB/s
// are the store rate of this access pattern (hipMemset of the same buffer: 6.0 TB/s).
__global__ __launch_bounds__(HZ_WDSHA_BLOCK) void k_withdraw_sha(const WithdrawArgs a) {
    const uint32_t gt = blockIdx.x * blockDim.x + threadIdx.x;
    if (gt >= 2 * a.N) return;
    const uint32_t i = gt % a.N, blk = gt / a.N;   // consecutive lanes = consecutive instances: coalesced stores
    UnitIO io{a.base, a.N, i, i, 0, a.err};
    io.coop32 = HZ_SHA_COOP && (a.N % 32u == 0u);   // then every aligned half-wavefront is 32 consecutive instances of one block, all active
    const WithdrawOff& o = a.wd;
    const Fr zero = fr_zero();
    const Fc rootExit_c = io.in_c(o.rootExit), ethAddr_c = io.in_c(o.ethAddr), tokenID_c = io.in_c(o.tokenID), balance_c = io.in_c(o.balance),
             idx_c = io.in_c(o.idx);
    if (blk == 0) {
        for (int k = 0; k < 256; k++) io.put_bit(o.n2bRootExit + k, c_bit(rootExit_c, k));
        num2bits_dev(io, o.n2bEthAddr, ethAddr_c, 160, C_WD_HI_N2B);
        num2bits_dev(io, o.n2bTokenID, tokenID_c, 32, C_WD_HI_N2B);
        num2bits_dev(io, o.n2bBalance, balance_c, 192, C_WD_HI_N2B);
        num2bits_dev(io, o.n2bIdx, idx_c, 48, C_WD_HI_N2B);
        uint32_t pad = 0;
        for (int j = (int)a.L; j < 48; j++) pad += c_bit(idx_c, j);
        if (pad) report_fail(io.err, io.inst, 0, C_WD_HI_PAD, fr_from_u64(pad), zero);
    }
    uint32_t msg[32];
    for (int k = 0; k < 32; k++) msg[k] = 0;
    auto put_be = [&](int pos, const Fc& c, int nb) __attribute__((always_inline)) {
        for (int k = 0; k < nb; k++)
            if (c_bit(c, nb - 1 - k)) msg[(pos + k) >> 5] |= 1u << (31 - ((pos + k) & 31));
    };
    put_be(0, rootExit_c, 256); put_be(256, ethAddr_c, 160); put_be(416, tokenID_c, 32); put_be(448, balance_c, 192); put_be(640, idx_c, 48);
    msg[688 >> 5] |= 1u << (31 - (688 & 31));
    msg[31] = 688;
    uint32_t hv[8];
    for (int k = 0; k < 8; k++) hv[k] = SHA_H0[k];
    if (blk == 0) {
        sha256_block_witness(io, o.sha.blocks, hv, msg);
    } else {
        sha256_compress(hv, msg);   // chaining value of block 0, recomputed (64 rounds) rather than exchanged
        sha256_block_witness(io, o.sha.blocks + o.sha.block_size, hv, msg + 16);
        io.put_c(o.hashGlobalInputs, sha_digest_to_fr(hv));
    }
}

// ---- multi-GPU shard exchange: per-transaction data-availability records --------------------------------
// record (HZ_DA_RECORD_BYTES = 160): [0,78) L1TxFullData bits | [78,84) fromIdx bits | [84,90) finalToIdx bits |
// [90,95) amountF bits (already masked by isAmountNullified) | [95] fee bits | [96,128) decodeTx.outIdx | [128,160) newExitRoot
__global__ __launch_bounds__(HZ_BLOCK) void k_da_export(const DaArgs a) {
    const uint32_t li = blockIdx.x * blockDim.x + threadIdx.x;
    if (li >= a.ucnt) return;
    const uint32_t u = a.u0 + li;
    uint8_t* r = a.buf + (size_t)li * HZ_DA_RECORD_BYTES;
    auto bit = [&](uint32_t sig) __attribute__((always_inline)) { return load_fr(a.tx_base + ((size_t)sig * a.nTx + u) * 32).v[0] & 1u; };
    auto pack = [&](uint32_t sig0, uint32_t n, uint8_t* dst, uint32_t nbytes) __attribute__((always_inline)) {
        for (uint32_t b = 0; b < nbytes; b++) {
            uint32_t v = 0;
            for (uint32_t k = 0; k < 8 && 8 * b + k < n; k++) v |= bit(sig0 + 8 * b + k) << k;
            dst[b] = (uint8_t)v;
        }
    };
    pack(a.l1full, L1FULL_BITS, r, 78);
    pack(a.n2bData + 48, a.L, r + 78, 6);
    pack(a.n2bFinalToIdx, a.L, r + 84, 6);
    pack(a.l1l2amt, 40, r + 90, 5);
    pack(a.l1l2Fee, 8, r + 95, 1);
    const Fc outIdx = fr_to_canon(a.tx_scratch[(size_t)SC_OUTIDX * a.nTx + u]);
    const Fc exitRoot = load_fr(a.tx_base + ((size_t)a.s5 * a.nTx + u) * 32);
    for (int k = 0; k < 8; k++) { reinterpret_cast<uint32_t*>(r + 96)[k] = outIdx.v[k]; reinterpret_cast<uint32_t*>(r + 128)[k] = exitRoot.v[k]; }
}
__global__ __launch_bounds__(HZ_BLOCK) void k_da_import(const DaArgs a) {
    const uint32_t li = blockIdx.x * blockDim.x + threadIdx.x;
    if (li >= a.ucnt) return;
    const uint32_t u = a.u0 + li;
    const uint8_t* r = a.buf + (size_t)li * HZ_DA_RECORD_BYTES;
    const UnitIO io{a.tx_base, a.nTx, u, 0, u, nullptr};
    auto unpack = [&](uint32_t sig0, uint32_t n, const uint8_t* src) __attribute__((always_inline)) {
        for (uint32_t k = 0; k < n; k++) io.put_bit(sig0 + k, (src[k >> 3] >> (k & 7)) & 1u);
    };
    unpack(a.l1full, L1FULL_BITS, r);
    unpack(a.n2bData + 48, a.L, r + 78);
    unpack(a.n2bFinalToIdx, a.L, r + 84);
    unpack(a.l1l2amt, 40, r + 90);
    unpack(a.l1l2Fee, 8, r + 95);
    Fc outIdx, exitRoot;
    for (int k = 0; k < 8; k++) { outIdx.v[k] = reinterpret_cast<const uint32_t*>(r + 96)[k]; exitRoot.v[k] = reinterpret_cast<const uint32_t*>(r + 128)[k]; }
    a.tx_scratch[(size_t)SC_OUTIDX * a.nTx + u] = fr_from_canon(outIdx);
    io.put_c(a.s5, exitRoot);
}

// ---- launchers ---------------------------------------------------------------------------------------
static inline dim3 grid1(uint32_t n) { return dim3((n + HZ_BLOCK - 1) / HZ_BLOCK); }
hipError_t launch_fee_front(const FeeFrontArgs& a, hipStream_t s) {
    hipLaunchKernelGGL(k_fee_front, grid1(a.n_units), dim3(HZ_BLOCK), 0, s, a);
    return hipGetLastError();
}
hipError_t launch_fee_back(const FeeBackArgs& a, hipStream_t s) {
    hipLaunchKernelGGL(k_fee_back, grid1(a.n_units), dim3(HZ_BLOCK), 0, s, a);
    return hipGetLastError();
}
hipError_t launch_hash_state_main(uint8_t* base, uint32_t N, const HashStateOff& hs, hipStream_t s) {
    HsMainArgs a{base, N, hs};
    hipLaunchKernelGGL(k_hash_state_main, grid1(N), dim3(HZ_BLOCK), 0, s, a);
    return hipGetLastError();
}
// The chain is sequential, the expansion (a 0.7 GB store stream per batch) is not: the blocks are split into groups, group g's
// expansion runs on the side stream while the main stream chains group g + 1, and the main stream joins at the end. The tail
// of a step is then about chain + expansion / groups instead of chain + expansion. side == s (or no events): one after the other.
#ifndef HZ_SHA_GROUPS
#define HZ_SHA_GROUPS 8
#endif
hipError_t launch_hi_prep_body(const HashInputsArgs& a0, hipStream_t s) {
    HashInputsArgs a = a0;
    const hipError_t e = hipMemsetAsync(a.msg, 0, (size_t)a.hi.sha.nblocks * 64 * a.B, s);
    if (e != hipSuccess) return e;
    a.prep_part = 1;
    hipLaunchKernelGGL(k_hi_prep, grid1((1 + a.maxL1 + a.nTx + a.F) * a.B), dim3(HZ_BLOCK), 0, s, a);
    return hipGetLastError();
}
hipError_t launch_hash_inputs(const HashInputsArgs& a0, hipStream_t s, hipStream_t side, hipEvent_t* ev, int n_ev, bool body_done) {
    HashInputsArgs a = a0;
    hipError_t e = hipSuccess;
    if (body_done) {
        a.prep_part = 2;
        hipLaunchKernelGGL(k_hi_prep, grid1(a.B), dim3(HZ_BLOCK), 0, s, a);
    } else {
        const size_t msg_bytes = (size_t)a.hi.sha.nblocks * 64 * a.B;
        e = hipMemsetAsync(a.msg, 0, msg_bytes, s);
        if (e != hipSuccess) return e;
        a.prep_part = 0;
        hipLaunchKernelGGL(k_hi_prep, grid1((1 + a.maxL1 + a.nTx + a.F) * a.B), dim3(HZ_BLOCK), 0, s, a);
    }
    const uint32_t nb = (uint32_t)a.hi.sha.nblocks;
    const bool piped = side && side != s && ev && n_ev >= 2 && nb >= 64;
    const uint32_t groups = piped ? (uint32_t)std::min<int>(HZ_SHA_GROUPS, n_ev - 1) : 1u;
    const uint32_t per = (nb + groups - 1) / groups;
    for (uint32_t g = 0, b0 = 0; b0 < nb; g++, b0 += per) {
        a.blk0 = b0;
        a.blk1 = std::min(nb, b0 + per);
        if (a.B <= 2 * HZ_SHAW_BATCH) hipLaunchKernelGGL(k_sha_chain_w, dim3((a.B + HZ_SHAW_BATCH - 1) / HZ_SHAW_BATCH), dim3(256), 0, s, a);
        else hipLaunchKernelGGL(k_sha_chain, dim3((a.B + 63) / 64), dim3(256), 0, s, a);
        hipStream_t sx = s;
        if (piped) {
            if ((e = hipEventRecord(ev[g], s)) != hipSuccess) return e;
            if ((e = hipStreamWaitEvent(side, ev[g], 0)) != hipSuccess) return e;
            sx = side;
        }
        hipLaunchKernelGGL(k_sha_expand, grid1((a.blk1 - a.blk0) * a.B * HZ_SHA_PARTS), dim3(HZ_BLOCK), 0, sx, a);
    }
    if (piped) {
        if ((e = hipEventRecord(ev[n_ev - 1], side)) != hipSuccess) return e;
        if ((e = hipStreamWaitEvent(s, ev[n_ev - 1], 0)) != hipSuccess) return e;
    }
    return hipGetLastError();
}
hipError_t launch_hi_chain_only(const HashInputsArgs& a0, hipStream_t s) {
    HashInputsArgs a = a0;
    hipError_t e = hipMemsetAsync(a.msg, 0, (size_t)a.hi.sha.nblocks * 64 * a.B, s);
    if (e != hipSuccess) return e;
    a.prep_part = 0;
    hipLaunchKernelGGL(k_hi_prep, grid1((1 + a.maxL1 + a.nTx + a.F) * a.B), dim3(HZ_BLOCK), 0, s, a);
    a.blk0 = 0;
    a.blk1 = (uint32_t)a.hi.sha.nblocks;
    if (a.B <= 2 * HZ_SHAW_BATCH) hipLaunchKernelGGL(k_sha_chain_w, dim3((a.B + HZ_SHAW_BATCH - 1) / HZ_SHAW_BATCH), dim3(256), 0, s, a);
    else hipLaunchKernelGGL(k_sha_chain, dim3((a.B + 63) / 64), dim3(256), 0, s, a);
    return hipGetLastError();
}
hipError_t launch_sha_expand_range(const HashInputsArgs& a0, uint32_t first, uint32_t count, hipStream_t s) {
    if (count == 0) return hipSuccess;
    HashInputsArgs a = a0;
    a.blk0 = first;
    a.blk1 = first + count;
    hipLaunchKernelGGL(k_sha_expand, grid1(count * a.B * HZ_SHA_PARTS), dim3(HZ_BLOCK), 0, s, a);
    return hipGetLastError();
}
hipError_t launch_da_export(const DaArgs& a, hipStream_t s) {
    if (a.ucnt) hipLaunchKernelGGL(k_da_export, grid1(a.ucnt), dim3(HZ_BLOCK), 0, s, a);
    return hipGetLastError();
}
hipError_t launch_da_import(const DaArgs& a, hipStream_t s) {
    if (a.ucnt) hipLaunchKernelGGL(k_da_import, grid1(a.ucnt), dim3(HZ_BLOCK), 0, s, a);
    return hipGetLastError();
}
hipError_t launch_withdraw(const WithdrawArgs& a, hipStream_t s) {
    hipLaunchKernelGGL(k_withdraw, dim3((a.N + HZ_WD_BLOCK - 1) / HZ_WD_BLOCK), dim3(HZ_WD_BLOCK), 0, s, a);
    return hipGetLastError();
}
hipError_t launch_withdraw_sha(const WithdrawArgs& a, hipStream_t s) {
    hipLaunchKernelGGL(k_withdraw_sha, dim3((2 * a.N + HZ_WDSHA_BLOCK - 1) / HZ_WDSHA_BLOCK), dim3(HZ_WDSHA_BLOCK), 0, s, a);
    return hipGetLastError();
}

}  // namespace hz
