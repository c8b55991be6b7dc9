// TEST INFRASTRUCTURE -- CPU oracle. Poseidon as in circomlib 0.5.2 poseidon.circom (not on disk;
// pinned dependency, reference package-lock.json:861-862). Restates the template literally:
// Ark (add C[t*r+j]) -> Sigma on all lanes (full rounds) or lane 0 (partial) -> Mix (dense M).
// Call sites in the reference: src/lib/hash-state.circom:32, src/decode-tx.circom:275.
#pragma once
#include <vector>
#include "fr64.h"

namespace orc {

struct PoseidonTab {
    int t, rp;
    std::vector<F> C, M;
};
const PoseidonTab& poseidon_tab(int t);

// sbox_out (optional): 3 entries per S-box in evaluation order: in2, in4, out
F poseidon(const F* inputs, int n_inputs, std::vector<F>* sbox_out = nullptr);

}  // namespace orc
