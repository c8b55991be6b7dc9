// Host check of the lazily reduced sums of fr.h (the signature ladder's fr_sub_lazy / fr_dbl_lazy / fr_sub3 / fr_sub_a_2x /
// fr_3a_b_c_lazy) against the plain routines: same field elements, results in the stated ranges, on random operands and on the
// edges of the operand ranges (0, 1, p - 1, p, p + 1, 2p - 1).
#include "../../circuits_amd/csrc/fr.h"
#include <stdio.h>
#include <random>
#include <vector>
using namespace hz;
static bool same(const Fr& a, const Fr& b) {
    const Fc x = fr_to_canon(a), y = fr_to_canon(b);
    for (int i = 0; i < 8; i++) if (x.v[i] != y.v[i]) return false;
    return true;
}
static bool below_2p_normalised(const Fr& a) {
    for (int i = 0; i < 8; i++) if (a.v[i] >> 29) return false;
    Fr t = a;
    fr_cond_sub_2p(t.v);
    for (int i = 0; i < 9; i++) if (t.v[i] != a.v[i]) return false;
    return true;
}
int main() {
    std::mt19937_64 rng(7);
    auto limbs = [](const uint32_t (&k)[9]) { Fr r; for (int i = 0; i < 9; i++) r.v[i] = k[i]; return r; };
    Fr P, zero = fr_zero(), one = fr_zero();
    one.v[0] = 1;
    for (int i = 0; i < 9; i++) P.v[i] = fr_p29(i);
    auto plus = [](const Fr& a, const Fr& b) { Fr r; for (int i = 0; i < 9; i++) r.v[i] = a.v[i] + b.v[i]; fr_norm(r.v); return r; };   // plain integers
    auto minus1 = [](Fr a) { int i = 0; while (a.v[i] == 0) { a.v[i] = HZ_M29; i++; } a.v[i]--; return a; };
    (void)limbs;
    // plain integers in [0, 2p): the edges, then random ones
    std::vector<Fr> pool = {zero, one, minus1(P), P, plus(P, one), minus1(plus(P, P))};
    auto rnd_p = [&]() { Fc c; for (int i = 0; i < 8; i++) c.v[i] = (uint32_t)rng(); c.v[7] &= 0x0fffffff; return fr_unpack(c); };   // < 2^252 < p
    for (int i = 0; i < 40; i++) { Fr a = rnd_p(); pool.push_back(a); pool.push_back(plus(a, P)); }
    std::vector<Fr> canon;   // operands that must be below p
    for (const Fr& a : pool) if (!same(a, zero) || true) { Fr t = a; Fr c = fr_cond_sub_p(t); canon.push_back(c); }
    Fr small = zero;
    small.v[0] = 168698;
    long bad = 0, n = 0;
    const Fr mul_by = rnd_p();
    for (const Fr& m : pool)
        for (const Fr& a : canon)
            for (const Fr& b : canon) {
                n++;
                const Fr r1 = fr_sub3(m, small, a, b), r2 = fr_sub(fr_sub(fr_sub(m, small), a), b);
                if (!same(r1, r2) || !below_2p_normalised(r1)) bad++;
                const Fr r3 = fr_sub_a_2x(m, a, pool[(n * 7) % pool.size()]), r4 = fr_sub(fr_sub(m, a), fr_dbl(pool[(n * 7) % pool.size()]));
                if (!same(r3, r4) || !below_2p_normalised(r3)) bad++;
                if (!same(fr_mul(fr_sub_lazy(a, b), mul_by), fr_mul(fr_sub(a, b), mul_by))) bad++;
                if (!same(fr_mul(fr_sub_lazy(m, pool[(n * 3) % pool.size()]), mul_by), fr_mul(fr_sub(m, pool[(n * 3) % pool.size()]), mul_by))) bad++;
                if (!same(fr_dbl_lazy(a), fr_dbl(a)) || !below_2p_normalised(fr_dbl_lazy(a))) bad++;
                if (!same(fr_mul(fr_add_lazy(m, a), mul_by), fr_mul(fr_add(m, a), mul_by))) bad++;
                if (!same(fr_mul(fr_sub2_lazy(m, a, b), mul_by), fr_mul(fr_sub(fr_sub(m, a), b), mul_by))) bad++;
                const Fr q = fr_3a_b_c_lazy(m, b, small);
                if (!same(fr_mul(q, mul_by), fr_mul(fr_add(fr_add(fr_add(fr_dbl(m), m), b), small), mul_by))) bad++;
            }
    printf("cases=%ld mismatches=%ld\n", n, bad);
    return bad != 0;
}
