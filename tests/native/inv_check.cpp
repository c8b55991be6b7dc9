// host check of the field inversion (circuits_amd/csrc/fr.h fr_inv, Bernstein-Yang division steps): a * inv(a) = 1 for random and
// edge operands, inv(0) = 0, and the same canonical value as the Fermat inverse a^(p-2) (every 16th operand: it is 4x slower).
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
        if (!zero && (it & 15) == 0) {
            const Fc fc = fr_to_canon(fr_inv_fermat(a));
            for (int q = 0; q < 8; q++) if (fc.v[q] != ic.v[q]) { bad++; break; }
        }
    }
    printf("mismatches=%d checksum=%016llx\n", bad, (unsigned long long)sum);
    return bad != 0;
}
