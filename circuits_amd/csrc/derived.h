// Dense trace of circomlib's Poseidon(n) from the S-box signals this layout stores (host code; formats.hip "Derived signals").
// circomlib 0.5.2 poseidon.circom declares, per component: inputs[n], out, ark[i].in/out[t], mix[i].in/out[t], sigmaF[k][j].{in,in2,in4,out},
// sigmaP[k].{in,in2,in4,out}; the witness stores in2 / in4 / out of every S-box (the products). Everything else is linear in those:
// the S-box inputs of round 0 are out / in4 (x^5 / x^4), every later Ark / Mix value follows forward from the stored S-box outputs.
// Pure C++ over the host field (no HIP): also compiled by tests/native/derived_check.cpp.
#pragma once
#include <stdint.h>
#include <vector>
#include "host/hostfield.h"
namespace hzd {   // Poseidon's plain parameters (round constants, MDS matrix), 4 x u64 little-endian
#include "gen/poseidon_consts_host.inc"
}

namespace hzderived {
enum { PW_ARK_IN = 0, PW_ARK_OUT = 1, PW_MIX_IN = 2, PW_MIX_OUT = 3 };
using hzh::F;
static const int POS_RP[6] = {56, 57, 56, 60, 60, 63};
struct PosTab { int t = 0, rp = 0; std::vector<F> C, M; };
static inline const PosTab& pos_tab(int t) {
    static const PosTab* tabs = [] {
        PosTab* tb = new PosTab[6];
        const uint64_t (*c[6])[4] = {hzd::HZ_POSEIDON_HC_T2, hzd::HZ_POSEIDON_HC_T3, hzd::HZ_POSEIDON_HC_T4, hzd::HZ_POSEIDON_HC_T5, hzd::HZ_POSEIDON_HC_T6, hzd::HZ_POSEIDON_HC_T7};
        const uint64_t (*m[6])[4] = {hzd::HZ_POSEIDON_HM_T2, hzd::HZ_POSEIDON_HM_T3, hzd::HZ_POSEIDON_HM_T4, hzd::HZ_POSEIDON_HM_T5, hzd::HZ_POSEIDON_HM_T6, hzd::HZ_POSEIDON_HM_T7};
        for (int k = 0; k < 6; k++) {
            const int t = k + 2;
            tb[k].t = t; tb[k].rp = POS_RP[k];
            for (int i = 0; i < t * (8 + POS_RP[k]); i++) tb[k].C.push_back(hzh::f_from_words(c[k][i]));
            for (int i = 0; i < t * t; i++) tb[k].M.push_back(hzh::f_from_words(m[k][i]));
        }
        return tb;
    }();
    return tabs[t - 2];
}
inline int pos_rounds(int t) { return 8 + POS_RP[t - 2]; }
inline int pos_nsbox(int t) { return 8 * t + POS_RP[t - 2]; }
// S-box number (evaluation order, the order the signals are stored in) of lane j of round i; -1: that lane passes no S-box
inline int pos_sbox(int t, int i, int j) {
    const int rp = POS_RP[t - 2];
    if (i < 4) return i * t + j;
    if (i < 4 + rp) return j == 0 ? 4 * t + (i - 4) : -1;
    return 4 * t + rp + (i - 4 - rp) * t + j;
}
// the dense trace of one permutation from its stored S-box signals S[3k + {0: in2, 1: in4, 2: out}] (canonical 32-byte elements):
// tr[(what * R + i) * t + j], what = PW_*
static inline void pos_trace(int t, const uint8_t* S, std::vector<F>& tr) {
    const PosTab& tb = pos_tab(t);
    const int R = pos_rounds(t);
    tr.assign((size_t)4 * R * t, hzh::f_zero());
    auto at = [&](int what, int i, int j) -> F& { return tr[((size_t)what * R + i) * t + j]; };
    auto sig = [&](int k, int which) { return hzh::f_from_canon(S + 32 * (3 * (size_t)k + which)); };
    for (int j = 0; j < t; j++) {   // the S-box inputs of round 0: x = x^5 / x^4 (0 when x^4 = 0)
        const F x4 = sig(j, 1);
        at(PW_ARK_OUT, 0, j) = hzh::f_is_zero(x4) ? hzh::f_zero() : hzh::f_mul(sig(j, 2), hzh::f_inv(x4));
        at(PW_ARK_IN, 0, j) = hzh::f_sub(at(PW_ARK_OUT, 0, j), tb.C[j]);
    }
    for (int i = 0; i < R; i++) {
        if (i > 0)
            for (int j = 0; j < t; j++) {
                at(PW_ARK_IN, i, j) = at(PW_MIX_OUT, i - 1, j);
                at(PW_ARK_OUT, i, j) = hzh::f_add(at(PW_ARK_IN, i, j), tb.C[(size_t)t * i + j]);
            }
        for (int j = 0; j < t; j++) {
            const int k = pos_sbox(t, i, j);
            at(PW_MIX_IN, i, j) = k >= 0 ? sig(k, 2) : at(PW_ARK_OUT, i, j);
        }
        for (int r = 0; r < t; r++) {
            F acc = hzh::f_mul(tb.M[(size_t)r * t], at(PW_MIX_IN, i, 0));
            for (int j = 1; j < t; j++) acc = hzh::f_add(acc, hzh::f_mul(tb.M[(size_t)r * t + j], at(PW_MIX_IN, i, j)));
            at(PW_MIX_OUT, i, r) = acc;
        }
    }
}

}  // namespace hzderived
