// BabyJubjub (twisted Edwards a=168700, d=168696 over BN254 Fr) helpers shared by the EdDSA
// kernels and the host batch builder. Circuit-exact affine formulas (the ones whose intermediate
// values are witness signals) live in the kernels; this header has the signal-free arithmetic:
// extended-coordinate add/double used for scalar multiplication.
#pragma once
#include "fr.h"

namespace hz {

struct PtE {  // extended twisted Edwards: x = X/Z, y = Y/Z, T = XY/Z
    Fr X, Y, Z, T;
};

HZ_HD Fr bj_a() { return fr_from_u64(168700); }
HZ_HD Fr bj_d() { return fr_from_u64(168696); }

HZ_HD PtE pte_identity() {
    PtE r;
    r.X = fr_zero(); r.Y = fr_one(); r.Z = fr_one(); r.T = fr_zero();
    return r;
}
HZ_HD PtE pte_from_affine(const Fr& x, const Fr& y) {
    PtE r;
    r.X = x; r.Y = y; r.Z = fr_one(); r.T = fr_mul(x, y);
    return r;
}
// unified addition (add-2008-hwcd for general a), complete on the prime-order subgroup
HZ_HD PtE pte_add(const PtE& p, const PtE& q, const Fr& a, const Fr& d) {
    const Fr A = fr_mul(p.X, q.X), B = fr_mul(p.Y, q.Y), C = fr_mul(fr_mul(p.T, q.T), d), D = fr_mul(p.Z, q.Z);
    const Fr E = fr_sub(fr_sub(fr_mul(fr_add(p.X, p.Y), fr_add(q.X, q.Y)), A), B);
    const Fr F = fr_sub(D, C), G = fr_add(D, C), H = fr_sub(B, fr_mul(a, A));
    PtE r;
    r.X = fr_mul(E, F); r.Y = fr_mul(G, H); r.T = fr_mul(E, H); r.Z = fr_mul(F, G);
    return r;
}
HZ_HD void pte_to_affine(const PtE& p, Fr& x, Fr& y) {
    const Fr zi = fr_inv(p.Z);
    x = fr_mul(p.X, zi);
    y = fr_mul(p.Y, zi);
}
// k * P, k given as 8 LE u32 limbs (plain integer)
HZ_HD PtE pte_mul(const PtE& p, const uint32_t* k) {
    const Fr a = bj_a(), d = bj_d();
    PtE acc = pte_identity();
    for (int i = 255; i >= 0; i--) {
        acc = pte_add(acc, acc, a, d);
        if ((k[i >> 5] >> (i & 31)) & 1u) acc = pte_add(acc, p, a, d);
    }
    return acc;
}

}  // namespace hz
