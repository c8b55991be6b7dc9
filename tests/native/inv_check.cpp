// host check of the field inversion (circuits_amd/csrc/fr.h fr_inv, both division-step forms): a * inv(a) = 1 for random and edge
// operands, inv(0) = 0, and the two forms return the same limbs. Built twice by tests/test_poseidon.py (HZ_INV_VAR = 0 / 1); the
// program prints a checksum of all inverses, which must agree between the builds.
#include <stdint.h>
#include <stdio.h>
#include "../../circuits_amd/csrc/fr.h"
using namespace hz;
int main() {
    uint64_t seed = 0x9e3779b97f4a7c15ull, sum = 1469598103934665603ull;
    int bad = 0;
    const Fr one = fr_one();
    for (int it = 0; it < 20000; it++) {
        Fc c;
        for (int q = 0; q < 8; q++) { seed = seed * 6364136223846793005ull + 1442695040888963407ull; c.v[q] = (uint32_t)(seed >> 32); }
        c.v[7] &= 0x0fffffff;
        if (it < 64) { for (int q = 0; q < 8; q++) c.v[q] = 0; c.v[it >> 5 ? 7 : 0] = 1u << (it & 31 & (it >> 5 ? 27 : 31)); }   // powers of two
        if (it == 64) for (int q = 0; q < 8; q++) c.v[q] = 0;                                                                        // zero
        if (it == 65) { for (int q = 0; q < 8; q++) c.v[q] = fc_p(q); c.v[0] -= 1; }                                                  // p - 1
        if (it == 66) { for (int q = 0; q < 8; q++) c.v[q] = 0; c.v[0] = 2; }
        const Fr a = fr_from_canon(c);
        const Fr i = fr_inv(a);
        const Fr prod = fr_mul(a, i);
        const bool zero = fr_is_zero(a);
        if (zero ? !fr_is_zero(i) : !fr_eq(prod, one)) bad++;
        const Fc ic = fr_to_canon(i);
        for (int q = 0; q < 8; q++) sum = (sum ^ ic.v[q]) * 1099511628211ull;
    }
    printf("mismatches=%d checksum=%016llx\n", bad, (unsigned long long)sum);
    return bad != 0;
}
