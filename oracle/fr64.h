// TEST INFRASTRUCTURE -- CPU oracle, not part of the product path (only tests/, smoke() and
// bench.py's cpu_baseline leg may load this).
//
// BN254 scalar field, 4 x 64-bit limbs, Montgomery form R = 2^256: the same shape as the field
// the reference's production witness binary links (ffiasm buildZqField(p,"Fr") -> fr.asm,
// reference tools/helpers/actions.js:207-215), restated in portable C++ with unsigned __int128.
// Deliberately independent of circuits_amd/csrc/fr.h (different limb width, different reduction
// schedule) so that a GPU/oracle agreement is meaningful.
#pragma once
#include <stdint.h>
#include <string.h>
#include <string>

namespace orc {

typedef unsigned __int128 u128;

static const uint64_t P64[4] = {0x43e1f593f0000001ull, 0x2833e84879b97091ull, 0xb85045b68181585dull, 0x30644e72e131a029ull};
static const uint64_t R1_64[4] = {0xac96341c4ffffffbull, 0x36fc76959f60cd29ull, 0x666ea36f7879462eull, 0x0e0a77c19a07df2full};
static const uint64_t R2_64[4] = {0x1bb8e645ae216da7ull, 0x53fe3ab1e35c59e3ull, 0x8c49833d53bb8085ull, 0x0216d0b17f4e44a5ull};
static const uint64_t INV64 = 0xc2e1f593efffffffull;

struct F {
    uint64_t l[4];  // Montgomery form

    F() { l[0] = l[1] = l[2] = l[3] = 0; }
    static F raw(const uint64_t* x) { F r; memcpy(r.l, x, 32); return r; }
    F(uint64_t x) { uint64_t c[4] = {x, 0, 0, 0}; *this = from_canon(c); }
    F(int x) { if (x >= 0) { uint64_t c[4] = {(uint64_t)x, 0, 0, 0}; *this = from_canon(c); } else { uint64_t c[4] = {(uint64_t)(-(int64_t)x), 0, 0, 0}; *this = from_canon(c).neg(); } }

    static bool geq_p(const uint64_t* a) {
        for (int i = 3; i >= 0; i--) {
            if (a[i] > P64[i]) return true;
            if (a[i] < P64[i]) return false;
        }
        return true;
    }
    static void sub_p(uint64_t* a) {
        u128 br = 0;
        for (int i = 0; i < 4; i++) {
            u128 d = (u128)a[i] - P64[i] - br;
            a[i] = (uint64_t)d;
            br = (d >> 64) & 1;
        }
    }
    static F mont_mul(const F& a, const F& b) {
        // separated operand scanning: full 512-bit product, then word-by-word reduction
        uint64_t t[9] = {0};
        uint64_t prod[8] = {0};
        for (int i = 0; i < 4; i++) {
            u128 c = 0;
            for (int j = 0; j < 4; j++) {
                c += (u128)a.l[j] * b.l[i] + prod[i + j];
                prod[i + j] = (uint64_t)c;
                c >>= 64;
            }
            prod[i + 4] = (uint64_t)c;
        }
        memcpy(t, prod, 64);
        t[8] = 0;
        for (int i = 0; i < 4; i++) {
            uint64_t m = t[i] * INV64;
            u128 c = 0;
            for (int j = 0; j < 4; j++) {
                c += (u128)m * P64[j] + t[i + j];
                t[i + j] = (uint64_t)c;
                c >>= 64;
            }
            for (int k = i + 4; k < 9 && c; k++) {
                c += t[k];
                t[k] = (uint64_t)c;
                c >>= 64;
            }
        }
        F r;
        memcpy(r.l, t + 4, 32);
        if (t[8] || geq_p(r.l)) sub_p(r.l);
        return r;
    }
    static F from_canon(const uint64_t* c) { return mont_mul(raw(c), raw(R2_64)); }
    void to_canon(uint64_t* out) const {
        uint64_t one[4] = {1, 0, 0, 0};
        F r = mont_mul(*this, raw(one));
        memcpy(out, r.l, 32);
    }
    static F from_bytes(const uint8_t* b) { uint64_t c[4]; memcpy(c, b, 32); return from_canon(c); }
    void to_bytes(uint8_t* b) const { uint64_t c[4]; to_canon(c); memcpy(b, c, 32); }

    F operator+(const F& o) const {
        F r;
        u128 c = 0;
        for (int i = 0; i < 4; i++) {
            c += (u128)l[i] + o.l[i];
            r.l[i] = (uint64_t)c;
            c >>= 64;
        }
        if (geq_p(r.l)) sub_p(r.l);
        return r;
    }
    F neg() const {
        if (is_zero()) return *this;
        F r;
        u128 br = 0;
        for (int i = 0; i < 4; i++) {
            u128 d = (u128)P64[i] - l[i] - br;
            r.l[i] = (uint64_t)d;
            br = (d >> 64) & 1;
        }
        return r;
    }
    F operator-() const { return neg(); }
    F operator-(const F& o) const { return *this + o.neg(); }
    F operator*(const F& o) const { return mont_mul(*this, o); }
    F& operator+=(const F& o) { *this = *this + o; return *this; }
    F& operator-=(const F& o) { *this = *this - o; return *this; }
    F& operator*=(const F& o) { *this = *this * o; return *this; }
    bool operator==(const F& o) const { return memcmp(l, o.l, 32) == 0; }
    bool operator!=(const F& o) const { return !(*this == o); }
    bool is_zero() const { return (l[0] | l[1] | l[2] | l[3]) == 0; }

    F pow(const uint64_t* e) const {  // 256-bit exponent
        F r(1);
        for (int i = 255; i >= 0; i--) {
            r = r * r;
            if ((e[i >> 6] >> (i & 63)) & 1) r = r * *this;
        }
        return r;
    }
    // field inverse; 0 -> 0 (circom `<--` division-by-zero convention)
    F inv() const {
        uint64_t e[4] = {P64[0] - 2, P64[1], P64[2], P64[3]};
        return pow(e);
    }
    F operator/(const F& o) const { return *this * o.inv(); }

    // bit i of the canonical integer
    int bit(int i) const {
        uint64_t c[4];
        to_canon(c);
        return i < 256 ? (int)((c[i >> 6] >> (i & 63)) & 1) : 0;
    }
    std::string dec() const {
        uint64_t c[4];
        to_canon(c);
        // repeated division by 10^19
        std::string s;
        uint64_t x[4] = {c[0], c[1], c[2], c[3]};
        while (x[0] | x[1] | x[2] | x[3]) {
            u128 rem = 0;
            for (int i = 3; i >= 0; i--) {
                u128 cur = (rem << 64) | x[i];
                x[i] = (uint64_t)(cur / 10);
                rem = cur % 10;
            }
            s.insert(s.begin(), (char)('0' + (int)rem));
        }
        return s.empty() ? "0" : s;
    }
};

// 2^k as field element
inline F pow2(int k) {
    uint64_t c[4] = {0, 0, 0, 0};
    c[k >> 6] = 1ull << (k & 63);
    if (F::geq_p(c)) {  // k = 254, 255
        F r(1);
        for (int i = 0; i < k; i++) r = r + r;
        return r;
    }
    return F::from_canon(c);
}

}  // namespace orc
