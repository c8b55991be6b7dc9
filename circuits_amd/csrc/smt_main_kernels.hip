// circomlib's sparse-Merkle-tree templates as `component main`: SMTProcessor(nLevels) (smt/smtprocessor.circom, the template behind
// reference src/rollup-tx.circom:537-570 and src/fee-tx.circom:97) and SMTVerifier(nLevels) (smt/smtverifier.circom, behind
// src/withdraw.circom:47-58). They exist so that the published known answers of the tree (tests/golden/smt_kat.json: the roots of
// iden3's go-merkletree, circomlib's Go twin) reach the same level-hash kernels the rollup circuits use:
//   SMTProcessor main = k_smtproc_front (leaf hashes, scratch) -> k_smt (the chain kernel of smt_kernels.hip, unchanged) -> k_smtproc_back
//   SMTVerifier main  = k_smtver_main, the general form of the verifier k_withdraw specialises (enabled = 1, fnc = 0, oldKey = 0)
#define HZ_FR_INLINE   // (an out-of-line product takes its operands through private memory: fee_kernels.hip)
#include <hip/hip_runtime.h>
#include "kernels.h"
#include "smt_dev.h"

namespace hz {

__global__ __launch_bounds__(HZ_BLOCK) void k_smtproc_front(const SmtMainArgs a) {
    const Fr* K4 = poseidon_consts_w<4>();
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= a.N) return;
    const UnitIO io{a.base, a.N, i, i, 0, a.err};
    const Scratch sc{a.scratch, a.N, i};
    const SmtProcInOff& d = a.pin;
    io.put_u64(d.one, 1);
    const Fr oldKey = io.in_m(d.oldKey), newKey = io.in_m(d.newKey);
    sc.set(SC_KEY_S1OLD, oldKey); sc.set(SC_KEY_1, newKey);
    sc.set(SC_P1_FNC0, io.in_m(d.fnc)); sc.set(SC_P1_FNC1, io.in_m(d.fnc + 1)); sc.set(SC_ISOLD0_1, io.in_m(d.isOld0));
    sc.set(SC_OLDSTATEROOT, io.in_m(d.oldRoot));
    Fr h1in[3] = {oldKey, io.in_m(d.oldValue), fr_one()};
    WitSboxSink so = io.sbox_sink(a.proc.o.hash1Old);
    sc.set(SC_LEAF_P1OLD, poseidon_hash<4>(h1in, K4, so));
    h1in[0] = newKey; h1in[1] = io.in_m(d.newValue);
    WitSboxSink sn = io.sbox_sink(a.proc.o.hash1New);
    sc.set(SC_LEAF_P1NEW, poseidon_hash<4>(h1in, K4, sn));
}

__global__ __launch_bounds__(HZ_BLOCK) void k_smtproc_back(const SmtMainArgs a) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= a.N) return;
    const UnitIO io{a.base, a.N, i, i, 0, a.err};
    const Scratch sc{a.scratch, a.N, i};
    (void)smt_top_dev(io, sc, a.proc, sc.get(SC_OLDSTATEROOT), C_SMTP_OLDROOT, C_SMTP_KEYS);   // stores main.newRoot
}

// SMTVerifier(n) with arbitrary field inputs: the state machine is evaluated in the field (smtverifiersm.circom), levIns from the
// zero pattern of the siblings (IsZero outputs are bits).
__global__ __launch_bounds__(HZ_BLOCK) void k_smtver_main(const SmtMainArgs a) {
    const Fr* K4 = poseidon_consts_w<4>();
    const Fr* K3 = poseidon_consts_w<3>();
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= a.N) return;
    const UnitIO io{a.base, a.N, i, i, 0, a.err};
    const SmtVerInOff& d = a.vin;
    const SmtVerOff& v = a.ver;
    const int n = (int)a.n_levels;
    const Fr one = fr_one(), zero = fr_zero();
    io.put_u64(d.one, 1);
    const Fr enabled = io.in_m(d.enabled), root = io.in_m(d.root), isOld0 = io.in_m(d.isOld0), fnc = io.in_m(d.fnc);
    const Fc oldKey_c = io.in_c(d.oldKey), key_c = io.in_c(d.key);
    const Fr oldKey = fr_from_canon(oldKey_c), key = fr_from_canon(key_c);
    Fr h1in[3] = {oldKey, io.in_m(d.oldValue), one};
    WitSboxSink so = io.sbox_sink(v.hash1Old);
    const Fr h1old = poseidon_hash<4>(h1in, K4, so);
    h1in[0] = key; h1in[1] = io.in_m(d.value);
    WitSboxSink sn = io.sbox_sink(v.hash1New);
    const Fr h1new = poseidon_hash<4>(h1in, K4, sn);
    num2bits_strict_dev(io, v.n2bOld, oldKey_c, C_SMTV_ALIAS_OLD);
    num2bits_strict_dev(io, v.n2bNew, key_c, C_SMTV_ALIAS_NEW);
    // SMTLevIns
    uint64_t zmask = 0;
    for (int base = 0; base < n; base += 8) {
        const int cnt = (n - base) < 8 ? (n - base) : 8;
        Fr z[8], zi[8];
        for (int k = 0; k < cnt; k++) { z[k] = io.in_m(d.siblings + base + k); zi[k] = z[k]; if (fr_is_zero(z[k])) zmask |= 1ull << (base + k); }
        batch_inv<8>(zi, cnt);
        for (int k = 0; k < cnt; k++) is_zero_dev(io, v.isz + 2 * (base + k), z[k], zi[k]);
    }
    if (!((zmask >> (n - 1)) & 1)) io.chk_zero(C_SMTV_LEVINS, fr_neg(enabled));
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
    // SMTVerifierSM, level 0 .. n-1. levIns is one-hot at kl: top = enabled above it, the three insertion states at kl, na below.
    const int kl = __builtin_ctzll(levmask);
    const Fr ptlif_kl = fr_mul(enabled, fnc);
    const Fr inew_kl = fr_sub(enabled, ptlif_kl);
    const Fr iold_kl = fr_mul(ptlif_kl, fr_sub(one, isOld0));
    const Fr i0_kl = fr_mul(enabled, isOld0);
    {
        Fr p_na = fr_sub(one, enabled), p_inew = zero, p_iold = zero, p_i0 = zero, last = zero;
        for (int k = 0; k < n; k++) {
            const bool at = k == kl;
            const Fr t_na = fr_add(fr_add(fr_add(p_na, p_inew), p_iold), p_i0);
            const Fr t_inew = at ? inew_kl : zero, t_iold = at ? iold_kl : zero, t_i0 = at ? i0_kl : zero;
            io.put_m(v.sm + VSM_N * k + VSM_PTLI, at ? enabled : zero); io.put_m(v.sm + VSM_N * k + VSM_PTLIF, at ? ptlif_kl : zero);
            io.put_m(v.sm + VSM_N * k + VSM_IOLD, t_iold); io.put_m(v.sm + VSM_N * k + VSM_I0, t_i0);
            if (k == n - 1) last = fr_add(fr_add(fr_add(t_na, t_iold), t_inew), t_i0);
            p_na = t_na; p_inew = t_inew; p_iold = t_iold; p_i0 = t_i0;
        }
        io.chk(C_SMTV_SM_FINAL, last, one);
    }
    Fr child = zero;
    for (int k = n - 1; k >= 0; k--) {
        const uint32_t lv = v.levels + VL_SIZE * k;
        const uint32_t sel = c_bit(key_c, k);
        const Fr sib = io.in_m(d.siblings + k);
        io.put_m(lv + VL_SW_AUX, sel ? fr_sub(sib, child) : zero);
        Fr h2[2];
        h2[0] = sel ? sib : child;
        h2[1] = sel ? child : sib;
        WitSboxSink sk = io.sbox_sink(lv + VL_HASH);
        const Fr ph = poseidon_hash<3>(h2, K3, sk);
        const Fr a0 = k < kl ? fr_mul(ph, enabled) : zero;
        const Fr a1 = k == kl ? fr_mul(h1old, iold_kl) : zero;
        const Fr rt = k == kl ? fr_add(fr_add(a0, a1), fr_mul(h1new, inew_kl)) : fr_add(a0, a1);
        io.put_m(lv + VL_AUX0, a0); io.put_m(lv + VL_AUX1, a1); io.put_m(lv + VL_ROOT, rt);
        child = rt;
    }
    {
        Fr z[2] = {fr_sub(key, oldKey), fr_sub(root, child)};   // areKeyEquals: in[0] = oldKey, in[1] = key; checkRoot: in[0] = levels[0].root, in[1] = root
        Fr zi[2] = {z[0], z[1]};
        inv_pair(zi[0], zi[1]);
        const Fr keq = is_zero_dev(io, v.keyEq, z[0], zi[0]);
        // keysOk = MultiAND(4)(fnc, 1 - isOld0, keq, enabled)
        const Fr aa = fr_mul(fnc, fr_sub(one, isOld0)), ab = fr_mul(keq, enabled), ac = fr_mul(aa, ab);
        io.put_m(v.and_a, aa); io.put_m(v.and_b, ab); io.put_m(v.and_c, ac);
        io.chk_zero(C_SMTV_KEYS, ac);
        const Fr e = is_zero_dev(io, v.checkRoot, z[1], zi[1]);
        io.chk_zero(C_SMTV_ROOT, fr_mul(fr_sub(one, e), enabled));
    }
}

static inline dim3 grid1(uint32_t n) { return dim3((n + HZ_BLOCK - 1) / HZ_BLOCK); }
hipError_t launch_smtproc_front(const SmtMainArgs& a, hipStream_t s) {
    hipLaunchKernelGGL(k_smtproc_front, grid1(a.N), dim3(HZ_BLOCK), 0, s, a);
    return hipGetLastError();
}
hipError_t launch_smtproc_back(const SmtMainArgs& a, hipStream_t s) {
    hipLaunchKernelGGL(k_smtproc_back, grid1(a.N), dim3(HZ_BLOCK), 0, s, a);
    return hipGetLastError();
}
hipError_t launch_smtver_main(const SmtMainArgs& a, hipStream_t s) {
    hipLaunchKernelGGL(k_smtver_main, grid1(a.N), dim3(HZ_BLOCK), 0, s, a);
    return hipGetLastError();
}

}  // namespace hz
