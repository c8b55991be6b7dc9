// Poseidon-dominated kernels of the per-transaction path, compiled with the field product inlined
// (HZ_FR_INLINE): in the dependent-chain regime of the SMT level hashes an out-of-line product
// costs about 2x (tools/microbench/mulbench.hip), and the rolled round loops keep the code small.
//   k_hash4  lane = (unit, job)    HashState + its SMTHash1
//   k_smt    lane = (unit, chain)  one of the independent level-hash chains of the SMTProcessors
#define HZ_FR_INLINE 1
#include <hip/hip_runtime.h>
#include <stdlib.h>
#include <string.h>
#include "tx_dev.h"
#include "kernels.h"
#include <vector>
#include "smt_dev.h"
#include "poseidon_quad.h"

namespace hz {

// ---------------------------------------------------------------------------------------------------
// HashState x4 + SMTHash1 x4 (reference src/rollup-tx.circom:297-312,517-532 and the hash1Old /
// hash1New components of circomlib's SMTProcessor). blockIdx.y = j.

#ifndef HZ_HASH4_BLOCK
#define HZ_HASH4_BLOCK HZ_BLOCK
#endif
__global__ __launch_bounds__(HZ_HASH4_BLOCK) __attribute__((amdgpu_waves_per_eu(2))) void k_hash4(const Hash4Args a) {
    const Fr* K5 = poseidon_consts_w<5>();
    const Fr* K4 = poseidon_consts_w<4>();
    const uint32_t li = blockIdx.x * blockDim.x + threadIdx.x;
    if (li >= (a.ucnt ? a.ucnt : a.n_units)) return;
    const uint32_t i = a.u0 + li;
    const HashJob& J = a.job[blockIdx.y];
    const UnitIO io{a.base, a.n_units, i, 0, i, nullptr};
    const Scratch sc{a.scratch, a.n_units, i};
    Fr hin[4];
    for (int k = 0; k < 4; k++) hin[k] = sc.get(J.sc_in + k);
    WitSboxSink s5 = io.sbox_sink(J.hs);
    const Fr h = poseidon_hash<5>(hin, K5, s5);
    Fr value = h;
    if (J.mux_off != ~0u) {
        // s1OldValue / s2OldValue: Mux1(c0 = old state hash, c1 = oldValue input, s = isInsert)
        value = mux1_dev(h, sc.get(J.sc_oldvalue), sc.get(J.sc_ins));
        io.put_m(J.mux_off, value);
    }
    if (J.out_sig != ~0u) io.put_m(J.out_sig, h);
    if (J.h1 == ~0u) return;
    Fr h1in[3] = {sc.get(J.sc_key), value, fr_one()};
    WitSboxSink s4 = io.sbox_sink(J.h1);
    sc.set(J.sc_leaf, poseidon_hash<4>(h1in, K4, s4));
}

// ---------------------------------------------------------------------------------------------------
// SMTProcessor level chains (circomlib smt/smtprocessor.circom + smtlevins, smtprocessorsm,
// smtprocessorlevel, switcher). One lane per (unit, processor, side). blockIdx.y = chain.
// The old-side lane also writes enabled, n2bOld, SMTLevIns; the new-side lane n2bNew, xors, sm.


// The empty-subtree levels of a proof (smt_dev.h) are pure stores of a constant block; the levels that hash data are pure integer
// arithmetic. Run one after the other -- or in different kernels: a store-bound grid gets few dispatch slots beside a resident
// integer-bound one (tools/experiments/overlap_probe.py: fills beside Poseidon keep 10 % of their rate) -- they cost the sum of
// their times. Inside ONE instruction stream the vector-memory port and the integer pipe do overlap: every S-box of a level that
// hashes also stores a few signals of an empty level's block (wave-uniform cursor, scalar registers), so the constant blocks leave
// in the shadow of the arithmetic. (Round 3 attribution, timing-only builds: the chain without any store 15.7 ms, with the S-box
// stores 18.1, with these background stores 21.5 -- and 21.5 whether their constants are loaded or not, whether they leave spread
// after each product or in one burst, at two or three wavefronts per SIMD: tools/microbench/mixbench.hip shows the same
// max(integer, store) x 1.2 for any kernel that mixes v_mad_u64_u32 and stores at this ratio.)
struct BgZero {
    uint8_t* base;            // section base
    uint32_t n_units, unit;
    uint32_t off0;            // signal offset of level 0's hash block of this chain (o.levels + LV_OLDHASH / LV_NEWHASH)
    uint32_t j, j_end;        // empty levels still to store: [j, j_end), wave-uniform
    uint32_t s;               // next signal of level j's block
    uint32_t per;             // signals per S-box
    // The constants come through the scalar cache like Poseidon's own (s_load + v_mov). Staging the block in LDS was measured and is
    // slower (k_smt 22.4 -> 27.2 ms): LDS reads share the lgkmcnt counter with the scalar loads that stream the round constants.
    __device__ __forceinline__ void one() {
        Fc c;
#pragma unroll
        for (int q = 0; q < 8; q++) c.v[q] = HZ_POSEIDON3_ZERO_WIT[s][q];
        store_fr(base + ((size_t)(off0 + LV_SIZE * j + s) * n_units + unit) * 32, c);
        if (++s == 243) { s = 0; j++; }
    }
    __device__ __forceinline__ void emit() {
        for (uint32_t q = 0; q < per && j < j_end; q++) one();
    }
    __device__ __forceinline__ void flush() {
        while (j < j_end) one();
    }
};
struct SmtSboxSink {
    static constexpr bool kCanon = WitSboxSink::kCanon;
    WitSboxSink w;
    BgZero* bg;
    __device__ __forceinline__ void operator()(int k, const Fr& x2, const Fr& x4, const Fr& x5) const {
        w(k, x2, x4, x5);
        bg->emit();
    }
};
// max over the wavefront of a small non-negative integer (< 64), as a scalar
__device__ __forceinline__ uint32_t wave_max_u6(uint32_t v) {
    uint32_t r = 0;
#pragma unroll
    for (int b = 5; b >= 0; b--) {
        const uint32_t cand = r | (1u << b);
        if (__any(v >= cand)) r = cand;
    }
    return r;
}
// The shape of one SMTProcessor chain of one unit -- the zero pattern of its siblings, the level SMTLevIns selects, where the keys
// part -- and from it the first level from which the chain is STRUCTURALLY empty.
struct SmtShape {
    uint64_t zmask;    // bit i = siblings[i] == 0
    uint64_t levmask;  // levIns, one-hot
    int kl, kx;        // the level with levIns = 1; the first level >= kl whose key bits differ (n: none)
};
__device__ __forceinline__ SmtShape smt_shape(const UnitIO& io, uint32_t siblings, int n, uint64_t keylo_old, uint64_t keylo_new) {
    SmtShape sh;
    // SMTLevIns: levIns[i] from the zero pattern of the siblings; isz[i] in {0,1}; done/levIns are 0/1 as well -> integer logic, exact for any input
    uint64_t zmask = 0;
    for (int k = 0; k < n; k++) {
        const Fc s = io.in_c(siblings + k);
        uint32_t any = 0;
        for (int q = 0; q < 8; q++) any |= s.v[q];
        if (!any) zmask |= 1ull << k;
    }
    uint64_t levmask = 0;
    {
        // levIns[n-1] = 1 - isz[n-2]; done[n-2] = levIns[n-1]; levIns[i] = (1-done[i])*(1-isz[i-1]); done[i-1] = levIns[i]+done[i]
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
    sh.zmask = zmask; sh.levmask = levmask;
    const uint64_t xmask = (keylo_old ^ keylo_new) & ((1ull << n) - 1);   // n <= HZ_MAX_SMT_LEVELS < 64
    sh.kl = __builtin_ctzll(levmask);
    const uint64_t xabove = xmask >> sh.kl;
    sh.kx = xabove ? sh.kl + __builtin_ctzll(xabove) : n;   // n: the keys agree on every level >= kl
    return sh;
}
// Levels that are empty for STRUCTURAL reasons (child and sibling are zero whatever the hashes are): from the level returned on.
// Old side: the child of level k is root(k+1) = h1old * s_a(k+1), zero from k = kx on (k = kl on when m = 0: update / nop / insert
// into an empty slot); new side: both switcher inputs vanish above kx (from kl on when m = 0). The siblings above the highest
// non-zero one are zero by inspection.
__device__ __forceinline__ uint32_t smt_thr_lane(const SmtShape& sh, bool m_zero, bool new_side, int n) {
    const uint64_t nz = ~sh.zmask & ((n < 64 ? (1ull << n) : 0ull) - 1ull);
    const int hi_nz = nz ? 63 - __builtin_clzll(nz) : -1;
    int thr_lane = m_zero ? sh.kl : (new_side ? sh.kx + 1 : sh.kx);
    if (thr_lane < hi_nz + 1) thr_lane = hi_nz + 1;
    if (thr_lane > n) thr_lane = n;
    return (uint32_t)thr_lane;
}
// Two wavefronts of this kernel per SIMD saturate the integer pipe (three, with the register budget that implies: no faster).
// Two wavefronts per workgroup: the kernel alone does not care (20.7 ms either way), the STEP does -- with the signature ladders, the
// fee chain and the SHA-256 tail of two contexts beside it, pairs of k_smt wavefronts placed together measured 35.7-35.9 ms per step
// against 36.7-36.9 for single-wavefront workgroups and 38.7 for four per workgroup on one box (profiles/r03_workgroup_ab.txt); 128
// for k_hash4 / k_main_front as well: within noise.
#ifndef HZ_SMT_BLOCK
#define HZ_SMT_BLOCK 128
#endif
// LAT (round 5): the latency form for small launches -- a QUAD of lanes per (unit, chain). All four lanes walk the chain's logic with the
// same values (the three that are not the leader store the same bytes to the same places and report nothing); the level hash itself is
// poseidon3_quad (poseidon_quad.h): the state over the lanes of the quad, 0.63 x the time of a dependent hash. The constant blocks of
// empty levels are stored where the chain passes them, a quarter per lane (no BgZero: a launch this small is nowhere near the store roofline).
template <bool LAT>
__global__ __launch_bounds__(HZ_SMT_BLOCK) __attribute__((amdgpu_waves_per_eu(LAT ? 1 : 2))) void k_smt(const SmtArgs a) {
    const Fr* K3 = poseidon_consts_w<3>();
    const uint32_t tidx = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t li = LAT ? tidx >> 2 : tidx, qj = LAT ? (threadIdx.x & 3u) : 0u;
    if (li >= (a.ucnt ? a.ucnt : a.n_units)) return;
    const uint32_t i = a.u0 + li * (a.ustride > 1 ? a.ustride : 1u);
    // the units another launch of this step evaluates (the last transaction of every batch: the early HashInputs chain, ctx.hip)
    if (a.skip_mod && i % a.skip_mod == a.skip_mod - 1) return;
    const uint32_t chain = a.chain_order ? ((a.chain_order >> (4u * blockIdx.y)) & 15u) : blockIdx.y, pi = chain >> 1;
    const bool new_side = chain & 1;
    const unsigned long long t_start = a.trace ? wall_clock64() : 0ull;
    const SmtProcDesc& P = a.p[pi];
    const SmtProcOff& o = P.o;
    const int n = (int)a.n_levels;
    UnitIO io{a.base, a.n_units, i, i / a.upi, i % a.upi, a.err};
    io.mute = qj != 0;
    const Scratch sc{a.scratch, a.n_units, i};
    const Fr one = fr_one(), zero = fr_zero();
    const Fr fnc0 = sc.get(P.sc_fnc0), fnc1 = sc.get(P.sc_fnc1), isOld0 = sc.get(P.sc_isold0);
    const Fr enabled = fr_sub(fr_add(fnc0, fnc1), fr_mul(fnc0, fnc1));
    const Fc oldKey_c = fr_to_canon(sc.get(P.sc_oldkey)), newKey_c = fr_to_canon(sc.get(P.sc_newkey));
    // the low 64 bits of the keys: the path bits of every level (n <= HZ_MAX_SMT_LEVELS < 64), read by shifts, never by indexing the key
    const uint64_t keylo_old = (uint64_t)oldKey_c.v[0] | ((uint64_t)oldKey_c.v[1] << 32), keylo_new = (uint64_t)newKey_c.v[0] | ((uint64_t)newKey_c.v[1] << 32);
    const Fr h1old = sc.get(P.sc_leaf_old), h1new = sc.get(P.sc_leaf_new);
    const SmtShape shape = smt_shape(io, P.siblings, n, keylo_old, keylo_new);
    const uint64_t zmask = shape.zmask, levmask = shape.levmask;
    if (!new_side) {
        if (o.fnc != ~0u) { io.put_m(o.fnc, fnc0); io.put_m(o.fnc + 1, fnc1); }
        io.put_m(o.enabled, enabled);
        num2bits_strict_dev(io, o.n2bOld, oldKey_c, P.cid_alias_old);
        // isZero[i]: batched inverses, 8 at a time. A group whose siblings are zero on every lane of the wavefront (the levels above
        // the leaf: 20 of 33 in a tree of 2^13 accounts) needs no inversion: IsZero(0) = (inv 0, out 1).
        for (int base = 0; base < n; base += 8) {
            const int cnt = (n - base) < 8 ? (n - base) : 8;
            const uint64_t grp = ((1ull << cnt) - 1ull) << base;
            if (__all((zmask & grp) == grp)) {
                for (int k = 0; k < cnt; k++) { io.put_c(o.isz + 2 * (base + k), fc_zero()); io.put_bit(o.isz + 2 * (base + k) + 1, 1u); }
                continue;
            }
            // Montgomery's trick on the group without an array indexed at run time (such arrays live in scratch memory): the prefix
            // products are pushed into a register array that ROTATES (constant indices only) and popped from the other end by the
            // backward pass, which reads the siblings again (they are in the cache) rather than keeping eight of them beside the
            // inversion's own state. One product body per loop: the loops stay rolled.
            Fr pre[8];
#pragma unroll
            for (int q = 0; q < 8; q++) pre[q] = zero;
            Fr acc = one;
#pragma unroll 1
            for (int k = 0; k < 8; k++) {
                const Fr v = k < cnt ? io.in_m(P.siblings + base + (k < cnt ? k : 0)) : zero;
#pragma unroll
                for (int q = 0; q < 7; q++) pre[q] = pre[q + 1];
                pre[7] = acc;
                if (!fr_is_zero(v)) acc = fr_mul(acc, v);
            }
            Fr inv = fr_inv(acc);
#pragma unroll 1
            for (int k = 7; k >= 0; k--) {
                if (k < cnt) {
                    const Fr v = io.in_m(P.siblings + base + k);
                    Fr vi = zero;
                    if (!fr_is_zero(v)) { vi = fr_mul(inv, pre[7]); inv = fr_mul(inv, v); }
                    is_zero_dev(io, o.isz + 2 * (base + k), v, vi);
                }
#pragma unroll
                for (int q = 7; q > 0; q--) pre[q] = pre[q - 1];
            }
        }
        // (isZero[n-1].out - 1) * enabled === 0
        if (!((zmask >> (n - 1)) & 1)) io.chk_zero(P.cid_levins, fr_neg(enabled));
        for (int k = 1; k <= n - 2; k++) io.put_bit(o.levIns + (k - 1), (uint32_t)((levmask >> k) & 1));
    } else {
        num2bits_strict_dev(io, o.n2bNew, newKey_c, P.cid_alias_new);
        for (int k = 0; k < n; k++) io.put_bit(o.xors + k, (uint32_t)(((keylo_old ^ keylo_new) >> k) & 1));
    }
    // State machine (smtprocessorsm.circom). levIns is one-hot (integer logic above) and the xor bits
    // are integers, so every per-level state is one of a few field values selected by the level's
    // position relative to kl (the level with levIns = 1) and kx (the first level >= kl whose key
    // bits differ): no per-level arrays are kept for the bottom-up pass.
    //   k <  kl : top = E, everything else 0
    //   k == kl : aux1 = E, aux2 = E*fnc0, old0 = aux2*isOld0, upd = E - aux2, m = aux2 - old0 -> new1 or bot
    //   k >  kl : bot = m until kx, new1 = m at kx, 0 afterwards
    // fnc0 / fnc1 / isOld0 may be arbitrary field elements in a standalone RollupTx: field arithmetic.
    const int kl = shape.kl, kx = shape.kx;
    const Fr A2 = fr_mul(enabled, fnc0);
    const Fr O = fr_mul(A2, isOld0);
    const Fr U = fr_sub(enabled, A2);
    const Fr m = fr_sub(A2, O);
    // thr_wave = the first level that is structurally empty (smt_thr_lane) for every lane of the wavefront.
    // ABOVE thr_wave (k > thr_wave) every per-level signal of every lane is zero as well -- the state-machine states (k > kl, and
    // k > kx or m = 0), the switcher auxiliaries and the roots (root(k) = h1 * s(k) with s(k) = 0): they are stored as zeros without
    // the products and conversions (`dead`, wave-uniform; 19 of 33 levels in a tree of 2^13 accounts).
    const uint32_t thr_wave = wave_max_u6(smt_thr_lane(shape, fr_is_zero(m), new_side, n));
    // Constant marks (SmtArgs::zmark): the witness buffer outlives the step, and what an empty level gets does not depend on the inputs --
    // the 243-signal block of Poseidon(0, 0) in its hash slots (from level markH up this unit's chain holds it already), zeros in the
    // switcher / state-machine signals of a dead level (from markD up). Such a level is stored only BELOW the wavefront's mark (the
    // maximum over its lanes: every lane holds the constant from there on); the lane leaves the marks of what the buffer holds now.
    uint32_t mkH = (uint32_t)n, mkD = (uint32_t)n;
    if (a.zmark) {
        const uint32_t v = *(const __attribute__((address_space(1))) uint16_t*)(a.zmark + (size_t)chain * a.n_units + i);
        mkH = (v & 0xffu) < (uint32_t)n ? (v & 0xffu) : (uint32_t)n;
        mkD = (v >> 8) < (uint32_t)n ? (v >> 8) : (uint32_t)n;
    }
    const uint32_t markH = wave_max_u6(mkH), markD = wave_max_u6(mkD);
    uint32_t newH = thr_wave > 0 ? thr_wave : (uint32_t)n;   // lowered below by hashing levels whose inputs vanish on every lane
    uint32_t n_skipped = 0;                                   // signals of this lane left as they were (wave-uniform)
    if (new_side) {
        Fr p_na = fr_sub(one, enabled), p_new1 = zero, p_old0 = zero, p_upd = zero;
        Fr last_sum = zero;
        for (int k = 0; k < n; k++) {
            const Fr aux1 = fr_select(k == kl, enabled, zero);   // (selects of values: a ?: on two lvalues selects an ADDRESS and sends both to scratch)
            const Fr aux2 = fr_select(k == kl, A2, zero);
            const Fr t_old0 = fr_select(k == kl, O, zero);
            const Fr t_upd = fr_select(k == kl, U, zero);
            const Fr t_new1 = fr_select(k == kx, m, zero);
            const Fr t_bot = fr_select(k >= kl && k < kx, m, zero);
            const Fr t_na = fr_add(fr_add(fr_add(p_new1, p_old0), p_na), p_upd);
            const uint32_t b = o.sm + SM_N * k;
            if (thr_wave > 0 && (uint32_t)k > thr_wave) {
                if ((uint32_t)k < markD) {
                    const Fc z0 = fc_zero();
                    io.put_c(b + SM_AUX1, z0); io.put_c(b + SM_AUX2, z0); io.put_c(b + SM_OLD0, z0); io.put_c(b + SM_NEW1, z0); io.put_c(b + SM_BOT, z0);
                } else n_skipped += 5;
            } else {
                io.put_m(b + SM_AUX1, aux1); io.put_m(b + SM_AUX2, aux2); io.put_m(b + SM_OLD0, t_old0); io.put_m(b + SM_NEW1, t_new1);
                io.put_m(b + SM_BOT, t_bot);
            }
            if (k == n - 1) last_sum = fr_add(fr_add(fr_add(t_na, t_new1), t_old0), t_upd);
            p_old0 = t_old0; p_new1 = t_new1; p_na = t_na; p_upd = t_upd;
        }
        io.chk(P.cid_sm_final, last_sum, one);
    }
    const Fr mU = fr_add(m, U), OU = fr_add(O, U);
    // level chain, bottom-up. Both sides run the level hash through ONE inlined copy of the permutation (the kernel's
    // hot code): wavefronts of the old and the new side that share a CU then share its instruction-cache lines.
    const int root_slot = new_side ? P.sc_root_new : P.sc_root_old;
    Fr child = zero;
    for (int k = n - 1; k >= 0; k--) {
        const uint32_t lv = o.levels + LV_SIZE * k;
        if (thr_wave > 0 && (uint32_t)k > thr_wave) {
            // dead level: nothing but zeros beside its (constant) hash block, which a hashing level stores (the latency form: here)
            const Fc z0 = fc_zero();
            if ((uint32_t)k >= markH) n_skipped += 243;
            else if (LAT) poseidon3_zero_level_quad(io, lv + (new_side ? LV_NEWHASH : LV_OLDHASH), qj);
            if ((uint32_t)k >= markD) {
                n_skipped += new_side ? 7 : 3;
            } else if (!new_side) {
                io.put_c(lv + LV_OLDSW_AUX, z0); io.put_c(lv + LV_AUX0, z0); io.put_c(lv + LV_OLDROOT, z0);
            } else {
                io.put_c(lv + LV_NEWSW_AUX, z0); io.put_c(lv + LV_AUX1, z0); io.put_c(lv + LV_AUX2, z0);
                io.put_c(lv + LV_NEWSW_L, z0); io.put_c(lv + LV_NEWSW_R, z0); io.put_c(lv + LV_AUX3, z0); io.put_c(lv + LV_NEWROOT, z0);
            }
            continue;   // child stays zero
        }
        const uint32_t sel = (uint32_t)((keylo_new >> k) & 1);
        const Fr sib = io.in_m(P.siblings + k);
        Fr hin[2];
        Fr s_tb = zero, s_n1 = zero;
        if (!new_side) {
            // oldSwitcher(L = oldChild, R = sibling, sel); aux = (R-L)*sel
            const Fr aux = fr_select(sel != 0, fr_sub(sib, child), zero);
            io.put_m(lv + LV_OLDSW_AUX, aux);
            hin[0] = fr_select(sel != 0, sib, child);
            hin[1] = fr_select(sel != 0, child, sib);
        } else {
            // st_top + st_bot ; st_new1 ; st_old0 + st_upd ; st_top
            s_tb = fr_select(k < kl, enabled, fr_select(k < kx, m, zero));
            s_n1 = fr_select(k == kx, m, zero);
            const Fr aux1 = fr_mul(child, s_tb);
            const Fr swL = fr_add(aux1, fr_mul(h1new, s_n1));
            Fr aux2 = zero;
            if (k < kl) aux2 = fr_mul(sib, enabled);
            const Fr swR = fr_add(aux2, fr_mul(h1old, s_n1));
            const Fr aux = fr_select(sel != 0, fr_sub(swR, swL), zero);
            hin[0] = fr_select(sel != 0, swR, swL);
            hin[1] = fr_select(sel != 0, swL, swR);
            io.put_m(lv + LV_NEWSW_AUX, aux); io.put_m(lv + LV_AUX1, aux1); io.put_m(lv + LV_AUX2, aux2);
            io.put_m(lv + LV_NEWSW_L, swL); io.put_m(lv + LV_NEWSW_R, swR);
        }
        Fr h;
        if (thr_wave > 0 && (uint32_t)k >= thr_wave) {
            // structurally empty for the whole wavefront: its block is stored by one of the hashing levels (below), only the digest here
            if ((uint32_t)k >= markH) n_skipped += 243;   // (== thr_wave: the dead levels above counted themselves)
            else if (LAT) poseidon3_zero_level_quad(io, lv + (new_side ? LV_NEWHASH : LV_OLDHASH), qj);
#pragma unroll
            for (int q = 0; q < 9; q++) h.v[q] = HZ_POSEIDON3_ZERO_HASH[q];
        } else if (LAT) {
            const uint32_t hs = lv + (new_side ? LV_NEWHASH : LV_OLDHASH);
            if (__all(fr_is_zero(hin[0]) && fr_is_zero(hin[1]))) {
                if ((uint32_t)k >= markH) n_skipped += 243;
                else poseidon3_zero_level_quad(io, hs, qj);
                if (newH == (uint32_t)k + 1) newH = (uint32_t)k;
#pragma unroll
                for (int q = 0; q < 9; q++) h.v[q] = HZ_POSEIDON3_ZERO_HASH[q];
            } else {
                const Pos3Dense KD{a.pos3_dense, a.pos3_dense + 195, a.pos3_dense + 204};
                h = poseidon3_quad(hin[0], hin[1], KD, WitOut{a.base, a.n_units, i}, hs, qj);
            }
        } else {
            // hashing level k (< thr_wave) also stores the blocks of the empty levels that do not hold theirs yet:
            // thr + [k E / H, (k+1) E / H), E = min(markH, n) - thr (often 0: the previous step left them), H = thr
            BgZero bg{a.base, a.n_units, i, o.levels + (new_side ? LV_NEWHASH : LV_OLDHASH), 0, 0, 0, 0};
            if (thr_wave > 0 && markH > thr_wave && !LAT) {
                const uint32_t E = markH - thr_wave;
                bg.j = thr_wave + (uint32_t)k * E / thr_wave;
                bg.j_end = thr_wave + ((uint32_t)k + 1) * E / thr_wave;
                bg.per = ((bg.j_end - bg.j) * 243 + 80) / 81;
            }
            SmtSboxSink sk{io.sbox_sink(lv + (new_side ? LV_NEWHASH : LV_OLDHASH)), &bg};
            if (__all(fr_is_zero(hin[0]) && fr_is_zero(hin[1]))) {
                if ((uint32_t)k >= markH) {
                    n_skipped += 243;
#pragma unroll
                    for (int q = 0; q < 9; q++) h.v[q] = HZ_POSEIDON3_ZERO_HASH[q];
                } else h = poseidon3_zero_level(io, lv + (new_side ? LV_NEWHASH : LV_OLDHASH));
                if (newH == (uint32_t)k + 1) newH = (uint32_t)k;
            } else h = poseidon_hash<3>(hin, K3, sk);
            bg.flush();
        }
        if (!new_side) {
            // st_bot + st_new1 + st_upd ; st_top
            const Fr s_a = fr_select(k < kl, zero, fr_select(k == kl, mU, fr_select(k <= kx, m, zero)));
            const Fr aux0 = fr_mul(h1old, s_a);
            const Fr root = k < kl ? fr_add(aux0, fr_mul(h, enabled)) : aux0;
            io.put_m(lv + LV_AUX0, aux0); io.put_m(lv + LV_OLDROOT, root);
            child = root;
        } else {
            const Fr aux3 = fr_mul(h, fr_add(s_tb, s_n1));
            const Fr root = k == kl ? fr_add(aux3, fr_mul(h1new, OU)) : aux3;
            io.put_m(lv + LV_AUX3, aux3); io.put_m(lv + LV_NEWROOT, root);
            child = root;
        }
    }
    sc.set(root_slot, child);
    if (a.zmark && qj == 0) {
        const uint32_t newD = thr_wave > 0 ? (thr_wave + 1 < (uint32_t)n ? thr_wave + 1 : (uint32_t)n) : (uint32_t)n;
        *(__attribute__((address_space(1))) uint16_t*)(a.zmark + (size_t)chain * a.n_units + i) = (uint16_t)(newH | (newD << 8));
    }
    if (a.trace && (threadIdx.x & 63u) == 0) {
        const uint32_t w = (blockIdx.y * gridDim.x + blockIdx.x) * (blockDim.x >> 6) + (threadIdx.x >> 6);
        unsigned long long* t = a.trace + 4ull * w;
        t[0] = t_start; t[1] = wall_clock64();
        t[2] = (unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 4) | ((unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 20) << 32);
        t[3] = blockIdx.x | (chain << 24) | ((threadIdx.x >> 6) << 28) | ((unsigned long long)thr_wave << 32);
    }
    if (a.skipped) {   // (profiling only) n_skipped is wave-uniform: one atomic per wavefront
        const unsigned long long lead = __ballot(qj == 0);
        if (n_skipped && lead && (threadIdx.x & 63u) == (uint32_t)(__ffsll((long long)lead) - 1))
            atomicAdd(a.skipped, (unsigned long long)n_skipped * (unsigned long long)__popcll(lead));
    }
}

static inline dim3 grid1(uint32_t n) { return dim3((n + HZ_BLOCK - 1) / HZ_BLOCK); }
hipError_t launch_hash4(const Hash4Args& a, hipStream_t s) {
    const uint32_t nl = a.ucnt ? a.ucnt : a.n_units;
    dim3 g((nl + HZ_HASH4_BLOCK - 1) / HZ_HASH4_BLOCK);
    g.y = a.n_jobs;
    hipLaunchKernelGGL(k_hash4, g, dim3(HZ_HASH4_BLOCK), 0, s, a);
    return hipGetLastError();
}
// One launch for the whole chain (round 1 launched it in chunks of 11 levels so that workgroup slots turned over during its store
// phase; with the empty-level blocks stored in the shadow of the hashing levels one launch is as good for the step and better for
// the kernel itself: no per-chunk prologue, no tail of a chunk waiting for its slowest wavefront).
namespace pos3host {
#include "gen/poseidon_consts_host.inc"
}
size_t pos3_dense_bytes() { return (size_t)HZ_POS3_DENSE_FRS * sizeof(Fr); }
hipError_t upload_pos3_dense(Fr* dst) {
    static const std::vector<Fr> tab = [] {
        auto mont = [](const uint64_t* w) {
            Fc c;
            for (int i = 0; i < 4; i++) { c.v[2 * i] = (uint32_t)w[i]; c.v[2 * i + 1] = (uint32_t)(w[i] >> 32); }
            return fr_from_canon(c);
        };
        Fr r2;
        for (int i = 0; i < 9; i++) r2.v[i] = fr_r2(i);
        std::vector<Fr> t(HZ_POS3_DENSE_FRS);
        for (int i = 0; i < 195; i++) t[i] = mont(pos3host::HZ_POSEIDON_HC_T3[i]);
        for (int i = 0; i < 9; i++) {
            t[195 + i] = mont(pos3host::HZ_POSEIDON_HM_T3[i]);   // M R
            t[204 + i] = fr_mul(t[195 + i], r2);                 // M R * R^2 / R = M R^2: times a canonical a, over R: (M a) R
        }
        return t;
    }();
    return hipMemcpy(dst, tab.data(), tab.size() * sizeof(Fr), hipMemcpyHostToDevice);
}
hipError_t launch_smt(const SmtArgs& a, hipStream_t s) {
    const uint32_t nl = a.ucnt ? a.ucnt : a.n_units;
    if (a.pos3_dense && nl <= HZ_SMT_LAT_MAX) {   // the latency form: a quad of lanes per (unit, chain)
        dim3 g(((size_t)nl * 4 + HZ_SMT_BLOCK - 1) / HZ_SMT_BLOCK);
        g.y = 2 * a.n_proc;
        hipLaunchKernelGGL(k_smt<true>, g, dim3(HZ_SMT_BLOCK), 0, s, a);
        return hipGetLastError();
    }
    dim3 g((nl + HZ_SMT_BLOCK - 1) / HZ_SMT_BLOCK);
    g.y = 2 * a.n_proc;
    // (Which chains share the first of a headline launch's two rounds of wavefront slots -- SmtArgs::chain_order -- was measured with the
    //  per-wavefront trace, profiles/r06_smt_wave_trace.txt: five orders, the kernel alone 16.34-16.50 ms in every one. Identity.)
    hipLaunchKernelGGL(k_smt<false>, g, dim3(HZ_SMT_BLOCK), 0, s, a);
    return hipGetLastError();
}

}  // namespace hz
