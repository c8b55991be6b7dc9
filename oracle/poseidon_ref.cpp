// TEST INFRASTRUCTURE -- CPU oracle (see poseidon_ref.h).
#include "poseidon_ref.h"
#include "gen/poseidon_consts_canon.inc"

namespace orc {

static PoseidonTab make_tab(int t, int rp, const uint64_t (*c)[4], const uint64_t (*m)[4]) {
    PoseidonTab tab;
    tab.t = t;
    tab.rp = rp;
    for (int i = 0; i < t * (8 + rp); i++) tab.C.push_back(F::from_canon(c[i]));
    for (int i = 0; i < t * t; i++) tab.M.push_back(F::from_canon(m[i]));
    return tab;
}

const PoseidonTab& poseidon_tab(int t) {
    static const PoseidonTab tabs[6] = {
        make_tab(2, ORC_POSEIDON_RP_T2, ORC_POSEIDON_C_T2, ORC_POSEIDON_M_T2),
        make_tab(3, ORC_POSEIDON_RP_T3, ORC_POSEIDON_C_T3, ORC_POSEIDON_M_T3),
        make_tab(4, ORC_POSEIDON_RP_T4, ORC_POSEIDON_C_T4, ORC_POSEIDON_M_T4),
        make_tab(5, ORC_POSEIDON_RP_T5, ORC_POSEIDON_C_T5, ORC_POSEIDON_M_T5),
        make_tab(6, ORC_POSEIDON_RP_T6, ORC_POSEIDON_C_T6, ORC_POSEIDON_M_T6),
        make_tab(7, ORC_POSEIDON_RP_T7, ORC_POSEIDON_C_T7, ORC_POSEIDON_M_T7),
    };
    return tabs[t - 2];
}

F poseidon(const F* inputs, int n_inputs, std::vector<F>* sbox_out) {
    const int t = n_inputs + 1;
    const PoseidonTab& tab = poseidon_tab(t);
    const int nr = 8 + tab.rp;
    std::vector<F> st(t), nx(t);
    st[0] = F(0);
    for (int j = 1; j < t; j++) st[j] = inputs[j - 1];
    for (int r = 0; r < nr; r++) {
        for (int j = 0; j < t; j++) st[j] = st[j] + tab.C[t * r + j];  // Ark
        const bool full = (r < 4) || (r >= 4 + tab.rp);
        for (int j = 0; j < (full ? t : 1); j++) {  // Sigma
            F in2 = st[j] * st[j];
            F in4 = in2 * in2;
            F out = in4 * st[j];
            if (sbox_out) {
                sbox_out->push_back(in2);
                sbox_out->push_back(in4);
                sbox_out->push_back(out);
            }
            st[j] = out;
        }
        for (int i = 0; i < t; i++) {  // Mix
            F lc(0);
            for (int j = 0; j < t; j++) lc += tab.M[i * t + j] * st[j];
            nx[i] = lc;
        }
        st = nx;
    }
    return st[0];
}

}  // namespace orc
