// libhz_host.so -- host-side arithmetic for the batch builder (the counterpart of the JS
// @hermeznetwork/commonjs BatchBuilder/RollupDB the reference's tests and tools call at
// test/helpers/helpers.js:46,148 and tools/generate-input.js:70-107). It is caller-side code: it
// prepares circuit INPUTS (state tree, signatures); it never computes a witness. It reuses the
// product's own field/Poseidon headers compiled for the host, and nothing from oracle/.
#include <stdint.h>
#include <string.h>
#include "../babyjub.h"
#include "../poseidon.h"

namespace hz {
#define HZ_CONST_ARR static const
#include "../gen/poseidon_consts.inc"
#undef HZ_CONST_ARR
}  // namespace hz
using namespace hz;

static Fr load(const uint8_t* b) {
    Fc c;
    memcpy(c.v, b, 32);
    return fr_from_canon(c);
}
static void store(uint8_t* b, const Fr& m) {
    const Fc c = fr_to_canon(m);
    memcpy(b, c.v, 32);
}

template <int T>
static Fr hash_t(const Fr* in, const uint32_t (*K)[9]) {
    NoSink s;
    return poseidon_hash<T>(in, reinterpret_cast<const Fr*>(K), s);
}

extern "C" int hzb_poseidon(int n_in, const uint8_t* in, uint8_t* out) {
    Fr x[6];
    if (n_in < 1 || n_in > 6) return 1;
    for (int i = 0; i < n_in; i++) x[i] = load(in + 32 * i);
    Fr h;
    switch (n_in + 1) {
        case 2: h = hash_t<2>(x, HZ_POSEIDON_K_T2); break;
        case 3: h = hash_t<3>(x, HZ_POSEIDON_K_T3); break;
        case 4: h = hash_t<4>(x, HZ_POSEIDON_K_T4); break;
        case 5: h = hash_t<5>(x, HZ_POSEIDON_K_T5); break;
        case 6: h = hash_t<6>(x, HZ_POSEIDON_K_T6); break;
        default: h = hash_t<7>(x, HZ_POSEIDON_K_T7); break;
    }
    store(out, h);
    return 0;
}

// (ox,oy) = k * (px,py) on BabyJubjub, affine Edwards coordinates, k a 256-bit LE integer
extern "C" int hzb_bjj_mul(const uint8_t* px, const uint8_t* py, const uint8_t* k, uint8_t* ox, uint8_t* oy) {
    uint32_t kk[8];
    memcpy(kk, k, 32);
    const PtE r = pte_mul(pte_from_affine(load(px), load(py)), kk);
    Fr x, y;
    pte_to_affine(r, x, y);
    store(ox, x);
    store(oy, y);
    return 0;
}
// self test hooks: both inverses of a canonical element (Montgomery domain inside)
extern "C" int hzb_fr_inv(const uint8_t* x, uint8_t* safegcd_out, uint8_t* fermat_out) {
    const Fr a = load(x);
    store(safegcd_out, fr_inv(a));
    store(fermat_out, fr_inv_fermat(a));
    return 0;
}
extern "C" int hzb_bjj_add(const uint8_t* px, const uint8_t* py, const uint8_t* qx, const uint8_t* qy, uint8_t* ox, uint8_t* oy) {
    const PtE r = pte_add(pte_from_affine(load(px), load(py)), pte_from_affine(load(qx), load(qy)), bj_a(), bj_d());
    Fr x, y;
    pte_to_affine(r, x, y);
    store(ox, x);
    store(oy, y);
    return 0;
}
