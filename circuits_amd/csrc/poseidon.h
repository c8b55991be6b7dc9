// Poseidon permutation over BN254 Fr as used by circomlib 0.5.2's `Poseidon(nInputs)` template
// (absent from /root/reference; call sites: src/lib/hash-state.circom:32 (n=4),
// src/decode-tx.circom:275 (n=6), SMTHash1 n=3 / SMTHash2 n=2 inside SMTProcessor,
// EdDSAPoseidonVerifier n=5). x^5 S-box, R_F = 8, R_P(t) = {56,57,56,60,60,63}[t-2],
// state[0] = 0 (capacity first), state[1..] = inputs, out = state[0] after the last Mix.
//
// One permutation per lane. Round constants and the MDS matrix live in LDS (Montgomery form);
// every lane reads the same address in the same cycle, i.e. an LDS broadcast, no bank conflicts.
#pragma once
#include "fr.h"

namespace hz {

template <int T>
struct PoseidonCfg;
template <> struct PoseidonCfg<2> { static constexpr int RP = 56; };
template <> struct PoseidonCfg<3> { static constexpr int RP = 57; };
template <> struct PoseidonCfg<4> { static constexpr int RP = 56; };
template <> struct PoseidonCfg<5> { static constexpr int RP = 60; };
template <> struct PoseidonCfg<6> { static constexpr int RP = 60; };
template <> struct PoseidonCfg<7> { static constexpr int RP = 63; };

template <int T> constexpr int poseidon_rounds() { return 8 + PoseidonCfg<T>::RP; }
template <int T> constexpr int poseidon_nconst() { return T * poseidon_rounds<T>(); }
template <int T> constexpr int poseidon_nsbox() { return 8 * T + PoseidonCfg<T>::RP; }
// number of Fr the constant block of width T occupies: C then M
template <int T> constexpr int poseidon_const_frs() { return poseidon_nconst<T>() + T * T; }

HZ_HD constexpr int poseidon_nsbox_rt(int t) {
    return 8 * t + (t == 2 ? 56 : t == 3 ? 57 : t == 4 ? 56 : t == 5 ? 60 : t == 6 ? 60 : 63);
}

struct NoSink {
    HZ_HD void operator()(int, const Fr&, const Fr&, const Fr&) const {}
};

// x -> x^5, reporting the three product signals of circomlib's Sigma(): in2, in4, out.
template <class Sink>
HZ_HD Fr poseidon_sbox(const Fr& x, int k, Sink& sink) {
    const Fr x2 = fr_sqr(x);
    const Fr x4 = fr_sqr(x2);
    const Fr x5 = fr_mul(x4, x);
    sink(k, x2, x4, x5);
    return x5;
}

// Mix followed by the next round's AddRoundConstants: out[i] = sum_j M[i][j] * st[j] + c[i], one lazily
// reduced dot product per output (rows of more than 6 terms are split in two). `Cn` = the next round's
// constants in c*R^2 form (nullptr after the last round): added to the column sums before the division
// by R they cost 9 integer additions instead of a modular addition.
template <int T>
HZ_HD void poseidon_mix_ark(Fr (&st)[T], const Fr* M, const Fr* Cn) {
    Fr o[T];
#pragma unroll
    for (int i = 0; i < T; i++) {
        if constexpr (T <= 6) {
            o[i] = fr_dot<T>(M + i * T, st, Cn ? Cn + i : nullptr);
        } else {
            o[i] = fr_add(fr_dot<4>(M + i * T, st, Cn ? Cn + i : nullptr), fr_dot<T - 4>(M + i * T + 4, st + 4));
        }
    }
#pragma unroll
    for (int i = 0; i < T; i++) st[i] = o[i];
}

// Full permutation; `in` are the T-1 inputs (Montgomery). `C`/`M` point at the staged constants
// (gen/poseidon_consts.inc: round 0 in Montgomery form, later rounds in c*R^2 form).
// S-box k is numbered in evaluation order: 4 full rounds (T each), R_P partial, 4 full rounds.
template <int T, class Sink>
HZ_HD Fr poseidon_hash(const Fr* in, const Fr* C, const Fr* M, Sink& sink) {
    constexpr int RP = PoseidonCfg<T>::RP;
    Fr st[T];
    st[0] = C[0];
#pragma unroll
    for (int j = 1; j < T; j++) st[j] = fr_add(in[j - 1], C[j]);
    int k = 0;
    int c = T;   // constants of the NEXT round
#pragma unroll 1
    for (int r = 0; r < 4; r++) {
#pragma unroll
        for (int j = 0; j < T; j++) st[j] = poseidon_sbox(st[j], k + j, sink);
        k += T;
        poseidon_mix_ark<T>(st, M, C + c);
        c += T;
    }
#pragma unroll 1
    for (int r = 0; r < RP; r++) {
        st[0] = poseidon_sbox(st[0], k, sink);
        k += 1;
        poseidon_mix_ark<T>(st, M, C + c);
        c += T;
    }
#pragma unroll 1
    for (int r = 0; r < 3; r++) {
#pragma unroll
        for (int j = 0; j < T; j++) st[j] = poseidon_sbox(st[j], k + j, sink);
        k += T;
        poseidon_mix_ark<T>(st, M, C + c);
        c += T;
    }
#pragma unroll
    for (int j = 0; j < T; j++) st[j] = poseidon_sbox(st[j], k + j, sink);
    // only state[0] of the last Mix is the digest
    if constexpr (T <= 6) {
        return fr_dot<T>(M, st);
    } else {
        return fr_add(fr_dot<4>(M, st), fr_dot<T - 4>(M + 4, st + 4));
    }
}

}  // namespace hz
