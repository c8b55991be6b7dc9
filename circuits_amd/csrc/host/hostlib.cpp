// libhz_host.so -- host-side arithmetic for the batch builder (the counterpart of the JS
// @hermeznetwork/commonjs BatchBuilder/RollupDB the reference's tests and tools call at
// test/helpers/helpers.js:46,148 and tools/generate-input.js:70-107). It is caller-side code: it
// prepares circuit INPUTS (state tree, signatures); it never computes a witness, and it uses nothing from oracle/.
//
// Hashing and curve arithmetic run on a 4 x 64-bit Montgomery field (hostfield.h: what an x86-64 core multiplies natively); the
// permutation is the textbook one
// (Ark, S-box, Mix per round, circomlib 0.5.2 poseidon.circom) on the plain parameters of gen/poseidon_consts_host.inc. The
// device-form routines stay reachable for the self tests (hzb_fr_inv, hzb_poseidon_dev9: same digests).
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include <stdlib.h>
#include <algorithm>
#include <thread>
#include <vector>
#include "../babyjub.h"
#include "../poseidon.h"
#include "hostfield.h"
#include "../../../include/hz_host.h"

namespace hz {
#define HZ_CONST_ARR static const
#include "../gen/poseidon_consts.inc"
#undef HZ_CONST_ARR
}  // namespace hz
#include "../gen/poseidon_consts_host.inc"
using namespace hz;
using hzh::F;

// ---- Poseidon on the host field -----------------------------------------------------------------------------------------------
struct HostTab {
    int t, rp;
    std::vector<F> C, M;
};
static HostTab make_tab(int t, int rp, const uint64_t (*c)[4], const uint64_t (*m)[4]) {
    HostTab tab;
    tab.t = t; tab.rp = rp;
    for (int i = 0; i < t * (8 + rp); i++) tab.C.push_back(hzh::f_from_words(c[i]));
    for (int i = 0; i < t * t; i++) tab.M.push_back(hzh::f_from_words(m[i]));
    return tab;
}
static const HostTab& host_tab(int t) {
    static const HostTab tabs[6] = {
        make_tab(2, HZ_POSEIDON_RP_T2, HZ_POSEIDON_HC_T2, HZ_POSEIDON_HM_T2), make_tab(3, HZ_POSEIDON_RP_T3, HZ_POSEIDON_HC_T3, HZ_POSEIDON_HM_T3),
        make_tab(4, HZ_POSEIDON_RP_T4, HZ_POSEIDON_HC_T4, HZ_POSEIDON_HM_T4), make_tab(5, HZ_POSEIDON_RP_T5, HZ_POSEIDON_HC_T5, HZ_POSEIDON_HM_T5),
        make_tab(6, HZ_POSEIDON_RP_T6, HZ_POSEIDON_HC_T6, HZ_POSEIDON_HM_T6), make_tab(7, HZ_POSEIDON_RP_T7, HZ_POSEIDON_HC_T7, HZ_POSEIDON_HM_T7)};
    return tabs[t - 2];
}
static F host_poseidon(const F* in, int n_in) {
    const int t = n_in + 1;
    const HostTab& tab = host_tab(t);
    F st[7], nx[7];
    st[0] = hzh::f_zero();
    for (int j = 1; j < t; j++) st[j] = in[j - 1];
    const int nr = 8 + tab.rp;
    for (int r = 0; r < nr; r++) {
        for (int j = 0; j < t; j++) st[j] = hzh::f_add(st[j], tab.C[(size_t)t * r + j]);
        const int ns = (r < 4 || r >= 4 + tab.rp) ? t : 1;
        for (int j = 0; j < ns; j++) {
            const F x2 = hzh::f_sqr(st[j]), x4 = hzh::f_sqr(x2);
            st[j] = hzh::f_mul(x4, st[j]);
        }
        for (int i = 0; i < t; i++) {
            F acc = hzh::f_mul(tab.M[(size_t)i * t], st[0]);
            for (int j = 1; j < t; j++) acc = hzh::f_add(acc, hzh::f_mul(tab.M[(size_t)i * t + j], st[j]));
            nx[i] = acc;
        }
        for (int i = 0; i < t; i++) st[i] = nx[i];
    }
    return st[0];
}
extern "C" int hzb_poseidon(int n_in, const uint8_t* in, uint8_t* out) {
    if (n_in < 1 || n_in > 6) return 1;
    F x[6];
    for (int i = 0; i < n_in; i++) x[i] = hzh::f_from_canon(in + 32 * i);
    hzh::f_to_canon(host_poseidon(x, n_in), out);
    return 0;
}
// `count` independent hashes of n_in inputs each: in = [count][n_in] elements, out = [count] digests
extern "C" int hzb_poseidon_many(int n_in, uint64_t count, const uint8_t* in, uint8_t* out) {
    if (n_in < 1 || n_in > 6) return 1;
    for (uint64_t k = 0; k < count; k++) {
        F x[6];
        for (int i = 0; i < n_in; i++) x[i] = hzh::f_from_canon(in + 32 * (k * n_in + i));
        hzh::f_to_canon(host_poseidon(x, n_in), out + 32 * k);
    }
    return 0;
}

// ---- BabyJubjub (twisted Edwards a = 168700, d = 168696), extended coordinates, on the host field -------------------------------
struct HPt { F X, Y, Z, T; };
static HPt hpt_from_affine(const F& x, const F& y) { return HPt{x, y, hzh::f_one(), hzh::f_mul(x, y)}; }
static HPt hpt_add(const HPt& p, const HPt& q, const F& a, const F& d) {   // add-2008-hwcd, unified
    using namespace hzh;
    const F A = f_mul(p.X, q.X), B = f_mul(p.Y, q.Y), C = f_mul(f_mul(p.T, q.T), d), D = f_mul(p.Z, q.Z);
    const F E = f_sub(f_sub(f_mul(f_add(p.X, p.Y), f_add(q.X, q.Y)), A), B);
    const F Fv = f_sub(D, C), G = f_add(D, C), H = f_sub(B, f_mul(a, A));
    return HPt{f_mul(E, Fv), f_mul(G, H), f_mul(Fv, G), f_mul(E, H)};
}
static void hpt_to_affine(const HPt& p, uint8_t* ox, uint8_t* oy) {
    const F zi = hzh::f_inv(p.Z);
    hzh::f_to_canon(hzh::f_mul(p.X, zi), ox);
    hzh::f_to_canon(hzh::f_mul(p.Y, zi), oy);
}
// (ox,oy) = k * (px,py), affine Edwards coordinates, k a 256-bit LE integer
extern "C" int hzb_bjj_mul(const uint8_t* px, const uint8_t* py, const uint8_t* k, uint8_t* ox, uint8_t* oy) {
    const F a = hzh::f_from_u64(168700), d = hzh::f_from_u64(168696);
    const HPt p = hpt_from_affine(hzh::f_from_canon(px), hzh::f_from_canon(py));
    HPt acc{hzh::f_zero(), hzh::f_one(), hzh::f_one(), hzh::f_zero()};
    for (int i = 255; i >= 0; i--) {
        acc = hpt_add(acc, acc, a, d);
        if ((k[i >> 3] >> (i & 7)) & 1) acc = hpt_add(acc, p, a, d);
    }
    hpt_to_affine(acc, ox, oy);
    return 0;
}
extern "C" int hzb_bjj_add(const uint8_t* px, const uint8_t* py, const uint8_t* qx, const uint8_t* qy, uint8_t* ox, uint8_t* oy) {
    const F a = hzh::f_from_u64(168700), d = hzh::f_from_u64(168696);
    hpt_to_affine(hpt_add(hpt_from_affine(hzh::f_from_canon(px), hzh::f_from_canon(py)), hpt_from_affine(hzh::f_from_canon(qx), hzh::f_from_canon(qy)), a, d), ox, oy);
    return 0;
}

// k * Base8 (circomlib babyjub.js Base8 = 8 * Generator): 32 windows of 8 bits over a table of AFFINE points j * 256^w * Base8
// (X, Y, T = XY; Z = 1 saves a product per addition), built at first use and normalised with one inversion; the result stays
// projective so that callers with many scalars pay one inversion for all of them (hzb_bjj_mul_base8_many)
struct APt { F X, Y, T; };
static HPt hpt_add_affine(const HPt& p, const APt& q, const F& a, const F& d) {   // madd-2008-hwcd
    using namespace hzh;
    const F A = f_mul(p.X, q.X), B = f_mul(p.Y, q.Y), C = f_mul(f_mul(p.T, q.T), d), D = p.Z;
    const F E = f_sub(f_sub(f_mul(f_add(p.X, p.Y), f_add(q.X, q.Y)), A), B);
    const F Fv = f_sub(D, C), G = f_add(D, C), H = f_sub(B, f_mul(a, A));
    return HPt{f_mul(E, Fv), f_mul(G, H), f_mul(Fv, G), f_mul(E, H)};
}
// out[i] = 1 / in[i] for every i with one inversion (Montgomery's trick); zeros stay zero
static void batch_inverse(const F* in, F* out, size_t n) {
    using namespace hzh;
    std::vector<F> pre(n);
    F acc = f_one();
    for (size_t i = 0; i < n; i++) {
        pre[i] = acc;
        if (!f_is_zero(in[i])) acc = f_mul(acc, in[i]);
    }
    F inv = f_inv(acc);
    for (size_t i = n; i-- > 0;) {
        if (f_is_zero(in[i])) { out[i] = f_zero(); continue; }
        out[i] = f_mul(inv, pre[i]);
        inv = f_mul(inv, in[i]);
    }
}
static const F& bjj_a() { static const F v = hzh::f_from_u64(168700); return v; }
static const F& bjj_d() { static const F v = hzh::f_from_u64(168696); return v; }
static const std::vector<APt>& base8_table() {
    static const std::vector<APt> table = [] {
        static const uint64_t BX[4] = {0x2893f3f6bb957051ull, 0x2ab8d8010534e0b6ull, 0x4eacb2e09d6277c1ull, 0x0bb77a6ad63e739bull};
        static const uint64_t BY[4] = {0x4b3c257a872d7d8bull, 0xfce0051fb9e13377ull, 0x25572e1cd16bf9edull, 0x25797203f7a0b249ull};
        const F &a = bjj_a(), &d = bjj_d();
        std::vector<HPt> t(32 * 256);
        HPt base = hpt_from_affine(hzh::f_from_words(BX), hzh::f_from_words(BY));
        for (int w = 0; w < 32; w++) {
            t[(size_t)w * 256] = HPt{hzh::f_zero(), hzh::f_one(), hzh::f_one(), hzh::f_zero()};
            for (int j = 1; j < 256; j++) t[(size_t)w * 256 + j] = hpt_add(t[(size_t)w * 256 + j - 1], base, a, d);
            base = hpt_add(t[(size_t)w * 256 + 255], base, a, d);
        }
        std::vector<F> z(t.size()), zi(t.size());
        for (size_t i = 0; i < t.size(); i++) z[i] = t[i].Z;
        batch_inverse(z.data(), zi.data(), z.size());
        std::vector<APt> out(t.size());
        for (size_t i = 0; i < t.size(); i++) {
            out[i].X = hzh::f_mul(t[i].X, zi[i]);
            out[i].Y = hzh::f_mul(t[i].Y, zi[i]);
            out[i].T = hzh::f_mul(out[i].X, out[i].Y);
        }
        return out;
    }();
    return table;
}
static HPt base8_mul_proj(const uint8_t* k) {
    const std::vector<APt>& table = base8_table();
    const F &a = bjj_a(), &d = bjj_d();
    HPt acc{hzh::f_zero(), hzh::f_one(), hzh::f_one(), hzh::f_zero()};
    for (int w = 0; w < 32; w++)
        if (k[w]) acc = hpt_add_affine(acc, table[(size_t)w * 256 + k[w]], a, d);
    return acc;
}
extern "C" int hzb_bjj_mul_base8(const uint8_t* k, uint8_t* ox, uint8_t* oy) {
    hpt_to_affine(base8_mul_proj(k), ox, oy);
    return 0;
}
// CPUs this process may really use: the cgroup's quota where there is one (the GPU boxes show 256 logical CPUs and grant 16), the
// affinity mask otherwise
static unsigned host_cpus(unsigned cap) {
    unsigned n = std::max(1u, std::thread::hardware_concurrency());
    if (FILE* f = fopen("/sys/fs/cgroup/cpu.max", "r")) {
        char q[32] = {0};
        unsigned long long period = 0;
        if (fscanf(f, "%31s %llu", q, &period) == 2 && period && strcmp(q, "max")) {
            const unsigned long long quota = strtoull(q, nullptr, 10);
            if (quota) n = (unsigned)std::max<unsigned long long>(1, std::min<unsigned long long>(n, (quota + period - 1) / period));
        }
        fclose(f);
    }
    return std::min(cap, n);
}
// count scalars (32 bytes each, little endian) -> count affine points: the additions spread over `threads` host threads (0: as many
// as the host grants, at most 32; a signature batch is a few thousand independent scalars), ONE inversion for all of them
extern "C" int hzb_bjj_mul_base8_many(uint64_t count, const uint8_t* k, uint8_t* ox, uint8_t* oy, int32_t threads) {
    if (!count) return 0;
    base8_table();
    std::vector<HPt> pts((size_t)count);
    unsigned nt = threads > 0 ? (unsigned)threads : host_cpus(32u);
    if (const char* e = getenv("HZB_THREADS")) nt = (unsigned)std::max(1, atoi(e));
    nt = (unsigned)std::min<uint64_t>(nt, (count + 63) / 64);
    auto work = [&](uint64_t lo, uint64_t hi) { for (uint64_t i = lo; i < hi; i++) pts[(size_t)i] = base8_mul_proj(k + 32 * i); };
    if (nt <= 1) work(0, count);
    else {
        std::vector<std::thread> th;
        const uint64_t per = (count + nt - 1) / nt;
        for (unsigned t = 0; t < nt; t++) {
            const uint64_t lo = std::min<uint64_t>(count, per * t), hi = std::min<uint64_t>(count, lo + per);
            if (lo < hi) th.emplace_back(work, lo, hi);
        }
        for (auto& x : th) x.join();
    }
    std::vector<F> z((size_t)count), zi((size_t)count);
    for (size_t i = 0; i < (size_t)count; i++) z[i] = pts[i].Z;
    batch_inverse(z.data(), zi.data(), (size_t)count);
    for (size_t i = 0; i < (size_t)count; i++) {
        hzh::f_to_canon(hzh::f_mul(pts[i].X, zi[i]), ox + 32 * i);
        hzh::f_to_canon(hzh::f_mul(pts[i].Y, zi[i]), oy + 32 * i);
    }
    return 0;
}

// ---- self test hooks: the device-form (9 x 29-bit) routines compiled for the host ----------------------------------------------
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
// the digest through the device's sparse-form permutation (poseidon.h): must equal hzb_poseidon
extern "C" int hzb_poseidon_dev9(int n_in, const uint8_t* in, uint8_t* out) {
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
// both inverses of a canonical element (Montgomery domain inside)
extern "C" int hzb_fr_inv(const uint8_t* x, uint8_t* safegcd_out, uint8_t* fermat_out) {
    const Fr a = load(x);
    store(safegcd_out, fr_inv(a));
    store(fermat_out, fr_inv_fermat(a));
    return 0;
}
