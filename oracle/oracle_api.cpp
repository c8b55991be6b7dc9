// TEST INFRASTRUCTURE -- CPU oracle, C entry points.
#include "oracle_api.h"
#include "poseidon_ref.h"

using namespace orc;

extern "C" int orc_poseidon_batch(int t, size_t n, const uint8_t* in, uint8_t* out, uint8_t* sbox_witness) {
    if (t < 2 || t > 7) return 1;
    const int ni = t - 1;
    for (size_t i = 0; i < n; i++) {
        F x[6];
        for (int j = 0; j < ni; j++) x[j] = F::from_bytes(in + (i * ni + j) * 32);
        std::vector<F> sb;
        F h = poseidon(x, ni, sbox_witness ? &sb : nullptr);
        h.to_bytes(out + i * 32);
        if (sbox_witness)
            for (size_t k = 0; k < sb.size(); k++) sb[k].to_bytes(sbox_witness + (k * n + i) * 32);  // [signal][instance]
    }
    return 0;
}
