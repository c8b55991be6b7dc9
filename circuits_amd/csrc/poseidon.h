// Poseidon permutation over BN254 Fr as used by circomlib 0.5.2's `Poseidon(nInputs)` template
// (absent from /root/reference; call sites: src/lib/hash-state.circom:32 (n=4),
// src/decode-tx.circom:275 (n=6), SMTHash1 n=3 / SMTHash2 n=2 inside SMTProcessor,
// EdDSAPoseidonVerifier n=5). x^5 S-box, R_F = 8, R_P(t) = {56,57,56,60,60,63}[t-2],
// state[0] = 0 (capacity first), state[1..] = inputs, out = state[0] after the last Mix.
//
// One permutation per lane. The partial rounds are evaluated in the sparse form derived in
// tools/poseidon_sparse.py: only lane 0 receives a constant and passes the S-box, lanes 1..T-1 are
// updated with one product each, and lane 0's S-box inputs and outputs -- the only partial-round
// values the witness holds -- are bit-identical to the circuit's dense evaluation (2T-1 constant
// products per round instead of T^2). The oracle keeps the dense form.
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
template <int T> constexpr int poseidon_nsbox() { return 8 * T + PoseidonCfg<T>::RP; }

// Layout of the constant block of width T (gen/poseidon_consts.inc, HZ_POSEIDON_K_T<T>, written by
// tools/gen_constants.py from tools/poseidon_sparse.py), in Fr units:
//   HEAD   4T        Ark constants of full rounds 0..3 (round 0 in Montgomery form, the others c*R^2)
//   E0     1         lane-0 constant of partial round 0 (c*R^2)
//   PART   per PAIR of partial rounds (a, b), 4T+1 Fr: rowA[T], eB, rowB[T], beta, eNext, colA[T-1], colB[T-1];
//          an odd last round, 2T Fr: row[T], col[T-1], eNext. rows/cols/beta in Montgomery form; e* = lane-0
//          constant of the following round as c*R^2 (after the last partial round: lane 0 of the next full
//          round, with the pushed-forward constants folded in)
//   DENSE  (T-1)^2   the pending lanes-1.. matrix applied once after the partial rounds
//   CF     T-1       lanes 1.. of the full round that follows (c*R^2, pushed constants folded in)
//   TAIL   3T        Ark constants of the last three full rounds (c*R^2)
//   M      T*T       MDS matrix, row-major (Montgomery form)
template <int T> constexpr int poseidon_k_e0() { return 4 * T; }
template <int T> constexpr int poseidon_k_part() { return 4 * T + 1; }
template <int T> constexpr int poseidon_k_dense() {
    return poseidon_k_part<T>() + (PoseidonCfg<T>::RP / 2) * (4 * T + 1) + (PoseidonCfg<T>::RP % 2) * 2 * T;
}
template <int T> constexpr int poseidon_k_cf() { return poseidon_k_dense<T>() + (T - 1) * (T - 1); }
template <int T> constexpr int poseidon_k_tail() { return poseidon_k_cf<T>() + (T - 1); }
template <int T> constexpr int poseidon_k_m() { return poseidon_k_tail<T>() + 3 * T; }
// number of Fr the constant block of width T occupies
template <int T> constexpr int poseidon_const_frs() { return poseidon_k_m<T>() + T * T; }

HZ_HD constexpr int poseidon_nsbox_rt(int t) {
    return 8 * t + (t == 2 ? 56 : t == 3 ? 57 : t == 4 ? 56 : t == 5 ? 60 : t == 6 ? 60 : 63);
}

struct NoSink {
    static constexpr bool kCanon = false;   // digest only: Montgomery form throughout, constant block HZ_POSEIDON_K_T*
    HZ_HD void operator()(int, const Fr&, const Fr&, const Fr&) const {}
};

// x -> x^5, reporting the three product signals of circomlib's Sigma(): in2, in4, out.
// A sink that stores them needs them in canonical form. Converting each Montgomery product costs a reduction of its own (three
// per S-box). Instead (Sink::kCanon): with x = aR,
//     x2  = x*x/R            = a^2 R          (Montgomery, not stored)
//     in2 = x2/R             = a^2            (one reduction, canonical)
//     in4 = x2*in2/R         = a^4            (canonical straight out of the product: Montgomery times canonical)
//     out = in4*x/R          = a^5            (the same)
// i.e. three products and four reductions instead of three and six (-20 % of the multiply-accumulates of an S-box). The S-box
// then returns a CANONICAL value; the constants that multiply S-box outputs carry the missing factor R (block HZ_POSEIDON_KW_T*,
// tools/gen_constants.py), so every linear layer lands in Montgomery form again and nothing else changes.
template <class Sink>
HZ_HD Fr poseidon_sbox(const Fr& x, int k, Sink& sink) {
    if constexpr (Sink::kCanon) {
        const Fr x2 = fr_sqr(x);
        const Fr in2 = fr_canon_limbs(x2);
        const Fr in4 = fr_cond_sub_p_rare(fr_mul(x2, in2));
        const Fr out = fr_cond_sub_p_rare(fr_mul(in4, x));
        sink(k, in2, in4, out);
        return out;
    } else {
        const Fr x2 = fr_sqr(x);
        const Fr x4 = fr_sqr(x2);
        const Fr x5 = fr_mul(x4, x);
        sink(k, x2, x4, x5);
        return x5;
    }
}

// one row of a constant matrix times the state, plus an optional c*R^2 addend, one reduction
// (rows of more than 6 terms are split in two)
template <int N>
HZ_HD Fr poseidon_row(const Fr* row, const Fr* st, const Fr* addend) {
    if constexpr (N <= 6) {
        return fr_dot<N>(row, st, addend);
    } else {
        return fr_add(fr_dot<4>(row, st, addend), fr_dot<N - 4>(row + 4, st + 4));   // N <= 8
    }
}

// Mix followed by the next round's AddRoundConstants: out[i] = sum_j M[i][j] * st[j] + c[i], one lazily
// reduced dot product per output. `Cn` = the next round's constants in c*R^2 form: added to the column
// sums before the division by R they cost 9 integer additions instead of a modular addition.
// NC = how many lanes receive a constant (T, or 1 before the first partial round).
//
// The loops over the T lanes stay ROLLED (one S-box / one row body per loop: the code of a permutation has to stay small, eight
// wavefronts of a CU stream through it at different places) and must not index the state with the loop counter: an array indexed
// at run time lives in scratch memory, and every round then loads and stores its lanes through it (k_hash4 moved 2.6x its
// witness bytes that way). The state ROTATES instead -- lane 0 is processed, everything moves down one place, the result is
// appended -- so that every access has a constant index and the state never leaves the registers; T-1 register moves per step.
template <int T>
HZ_HD void poseidon_rotate_in(Fr (&a)[T], const Fr& v) {
#pragma unroll
    for (int i = 0; i + 1 < T; i++) a[i] = a[i + 1];
    a[T - 1] = v;
}
template <int T, int NC>
HZ_HD void poseidon_mix_ark(Fr (&st)[T], const Fr* M, const Fr* Cn) {
    static_assert(NC == T || NC == 1, "constants for every lane or for lane 0 only");
    Fr o[T];
#pragma unroll
    for (int i = 0; i < T; i++) o[i] = st[i];
    if constexpr (NC == T) {
#pragma unroll 1
        for (int i = 0; i < T; i++) poseidon_rotate_in<T>(o, poseidon_row<T>(M + i * T, st, Cn + i));
    } else {
        poseidon_rotate_in<T>(o, poseidon_row<T>(M, st, Cn));
#pragma unroll 1
        for (int i = 1; i < T; i++) poseidon_rotate_in<T>(o, poseidon_row<T>(M + i * T, st, nullptr));
    }
#pragma unroll
    for (int i = 0; i < T; i++) st[i] = o[i];
}
// the S-box on every lane (a full round), S-boxes numbered k .. k+T-1
template <int T, class Sink>
HZ_HD void poseidon_sbox_layer(Fr (&st)[T], int k, Sink& sink) {
#pragma unroll 1
    for (int j = 0; j < T; j++) poseidon_rotate_in<T>(st, poseidon_sbox(st[0], k + j, sink));
}

// Full permutation; `in` are the T-1 inputs (Montgomery), K the constant block (layout above) in the form the sink asks for:
// poseidon_consts<T>() for a digest-only sink, poseidon_consts_w<T>() for Sink::kCanon. The digest is in Montgomery form.
// S-box k is numbered in evaluation order: 4 full rounds (T each), R_P partial, 4 full rounds.
// (A rolled form with one S-box body for the full rounds and one for the partial rounds -- half the code, 48 KB instead of 98 KB for
// t = 3 with the witness sink -- was measured in round 3: the instruction cache was never the limit (0.1 % misses), k_smt unchanged,
// Poseidon t = 3 with witness 250 instead of 256 M/s. Not kept.)
template <int T, class Sink>
HZ_HD Fr poseidon_hash(const Fr* in, const Fr* K, Sink& sink) {
    constexpr int RP = PoseidonCfg<T>::RP;
    const Fr* M = K + poseidon_k_m<T>();
    Fr st[T];
    st[0] = K[0];
#pragma unroll
    for (int j = 1; j < T; j++) st[j] = fr_add(in[j - 1], K[j]);
    int k = 0;
#pragma unroll 1
    for (int r = 0; r < 3; r++) {
        poseidon_sbox_layer<T>(st, k, sink);
        k += T;
        poseidon_mix_ark<T, T>(st, M, K + T * (r + 1));
    }
    poseidon_sbox_layer<T>(st, k, sink);
    k += T;
    poseidon_mix_ark<T, 1>(st, M, K + poseidon_k_e0<T>());
    // Partial rounds, sparse form: lane 0 <- row . (y, lanes) + next constant ; lane i <- lane i + col[i] * y. Two rounds
    // per iteration so that lanes 1.. are reduced once per pair: round b's row is applied to the lanes as they were before
    // round a (its constants absorb colA through beta * yA), then lane i <- lane i + colA[i]*yA + colB[i]*yB in one
    // reduction. Lanes 1.. grow by at most ~1.02 p per pair (two products of values below p, the reduction's multiple of p) and are
    // brought back below 4p after every THIRD pair: fr_muladd2 takes an addend below 8p, and 4p + 3 x 1.02 p stays under it (the
    // conditional subtraction is 45 instructions per lane: after every pair it was 2 % of a permutation).
    const Fr* S = K + poseidon_k_part<T>();
    int lazy = 0;
#pragma unroll 1
    for (int r = 0; r + 1 < RP; r += 2) {
        Fr v[T + 1];
        v[0] = poseidon_sbox(st[0], k, sink);
#pragma unroll
        for (int j = 1; j < T; j++) v[j] = st[j];
        const Fr sa = poseidon_row<T>(S, v, S + T);
        v[T] = v[0];                                   // yA, multiplied by beta
        v[0] = poseidon_sbox(sa, k + 1, sink);         // yB
        k += 2;
        st[0] = poseidon_row<T + 1>(S + T + 1, v, S + 2 * T + 2);
        const Fr* CA = S + 2 * T + 3;
#pragma unroll
        for (int j = 1; j < T; j++) st[j] = fr_muladd2(CA[j - 1], v[T], CA[T - 1 + j - 1], v[0], st[j]);
        if (++lazy == 3) {   // (a branch around the subtraction only: two copies of the products cost the batch kernel twice its registers)
            lazy = 0;
#pragma unroll
            for (int j = 1; j < T; j++) st[j] = fr_cond_sub_4p(st[j]);
        }
        S += 4 * T + 1;
    }
    if constexpr (RP % 2 == 1) {
        st[0] = poseidon_sbox(st[0], k, sink);
        k += 1;
        const Fr s0 = poseidon_row<T>(S, st, S + 2 * T - 1);
#pragma unroll
        for (int j = 1; j < T; j++) st[j] = fr_muladd(S[T + j - 1], st[0], st[j]);
        st[0] = s0;
    }
    {
        // pending lanes-1.. matrix, with the constants of the following full round (rolled over the rows, rotating like the mix)
        Fr o[T - 1];
#pragma unroll
        for (int i = 0; i + 1 < T; i++) o[i] = st[i + 1];
#pragma unroll 1
        for (int i = 0; i + 1 < T; i++)
            poseidon_rotate_in<T - 1>(o, poseidon_row<T - 1>(K + poseidon_k_dense<T>() + i * (T - 1), st + 1, K + poseidon_k_cf<T>() + i));
#pragma unroll
        for (int i = 1; i < T; i++) st[i] = o[i - 1];
    }
#pragma unroll 1
    for (int r = 0; r < 3; r++) {
        poseidon_sbox_layer<T>(st, k, sink);
        k += T;
        poseidon_mix_ark<T, T>(st, M, K + poseidon_k_tail<T>() + T * r);
    }
    poseidon_sbox_layer<T>(st, k, sink);
    // only state[0] of the last Mix is the digest
    return poseidon_row<T>(M, st, nullptr);
}

}  // namespace hz
