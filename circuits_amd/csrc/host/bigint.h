// 256-bit unsigned integers, SHA-256 over bit strings and SHA-512 for the host-side batch builder (batchbuilder.cpp): balances, float40
// amounts, fee arithmetic (reference src/compute-fee.circom:105-143, src/lib/decode-float.circom:14-60), the data-availability hash of
// src/hash-inputs.circom:117-184 and the deterministic EdDSA nonce of circomlib's eddsa.js (a wallet's job; the synthetic generator
// signs its own transactions). Caller-side code: it prepares circuit INPUTS, never a witness, and shares nothing with oracle/.
#pragma once
#include <stdint.h>
#include <string.h>
#include <vector>

namespace hzh {

typedef unsigned __int128 u128;
struct U256 { uint64_t w[4]; };

static inline U256 u_zero() { U256 r; memset(r.w, 0, 32); return r; }
static inline U256 u_from64(uint64_t x) { U256 r = u_zero(); r.w[0] = x; return r; }
static inline U256 u_from_bytes(const uint8_t* b) { U256 r; memcpy(r.w, b, 32); return r; }
static inline void u_to_bytes(const U256& a, uint8_t* b) { memcpy(b, a.w, 32); }
static inline bool u_is_zero(const U256& a) { return (a.w[0] | a.w[1] | a.w[2] | a.w[3]) == 0; }
static inline bool u_eq(const U256& a, const U256& b) { return memcmp(a.w, b.w, 32) == 0; }
static inline int u_cmp(const U256& a, const U256& b) {
    for (int i = 3; i >= 0; i--) {
        if (a.w[i] != b.w[i]) return a.w[i] > b.w[i] ? 1 : -1;
    }
    return 0;
}
static inline U256 u_add(const U256& a, const U256& b) {
    U256 r;
    u128 c = 0;
    for (int i = 0; i < 4; i++) { c += (u128)a.w[i] + b.w[i]; r.w[i] = (uint64_t)c; c >>= 64; }
    return r;
}
static inline U256 u_sub(const U256& a, const U256& b) {
    U256 r;
    u128 br = 0;
    for (int i = 0; i < 4; i++) { const u128 d = (u128)a.w[i] - b.w[i] - br; r.w[i] = (uint64_t)d; br = (d >> 64) & 1; }
    return r;
}
static inline U256 u_mul(const U256& a, const U256& b) {   // low 256 bits
    U256 r = u_zero();
    for (int i = 0; i < 4; i++) {
        u128 c = 0;
        for (int j = 0; i + j < 4; j++) { c += (u128)a.w[j] * b.w[i] + r.w[i + j]; r.w[i + j] = (uint64_t)c; c >>= 64; }
    }
    return r;
}
static inline U256 u_shr(const U256& a, unsigned n) {
    U256 r = u_zero();
    const unsigned q = n >> 6, s = n & 63;
    for (unsigned i = 0; i + q < 4; i++) {
        r.w[i] = a.w[i + q] >> s;
        if (s && i + q + 1 < 4) r.w[i] |= a.w[i + q + 1] << (64 - s);
    }
    return r;
}
static inline U256 u_shl(const U256& a, unsigned n) {
    U256 r = u_zero();
    const unsigned q = n >> 6, s = n & 63;
    for (unsigned i = q; i < 4; i++) {
        r.w[i] = a.w[i - q] << s;
        if (s && i > q) r.w[i] |= a.w[i - q - 1] >> (64 - s);
    }
    return r;
}
static inline U256 u_or(const U256& a, const U256& b) { U256 r; for (int i = 0; i < 4; i++) r.w[i] = a.w[i] | b.w[i]; return r; }
static inline unsigned u_bit(const U256& a, unsigned i) { return (unsigned)((a.w[i >> 6] >> (i & 63)) & 1); }
static inline U256 u_low_bits(const U256& a, unsigned n) {   // a mod 2^n
    if (n >= 256) return a;
    return u_shr(u_shl(a, 256 - n), 256 - n);
}
static inline U256 u_pow10(unsigned e) {
    U256 r = u_from64(1);
    const U256 ten = u_from64(10);
    for (unsigned i = 0; i < e; i++) r = u_mul(r, ten);
    return r;
}
// float40: 35-bit mantissa, 5-bit decimal exponent (reference src/lib/decode-float.circom:14-60)
static inline U256 float40_to_fix(uint64_t f) { return u_mul(u_from64(f & ((1ull << 35) - 1)), u_pow10((unsigned)(f >> 35) & 31)); }

// x (n64 <= 10 words, little endian) mod m (m != 0): Knuth's algorithm D on 64-bit digits (a bit-serial version of this took 10 us per
// 512-bit operand: two of them per signature were a quarter of a batch's build)
static inline U256 u_mod_wide(const uint64_t* x, int n64, const U256& m) {
    typedef unsigned __int128 u128w;
    int n = 4;
    while (n > 1 && m.w[n - 1] == 0) n--;
    const int s = __builtin_clzll(m.w[n - 1]);
    uint64_t v[4] = {0, 0, 0, 0}, u[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    for (int i = n - 1; i >= 0; i--) v[i] = s ? ((m.w[i] << s) | (i ? (m.w[i - 1] >> (64 - s)) : 0)) : m.w[i];
    u[n64] = s ? (x[n64 - 1] >> (64 - s)) : 0;
    for (int i = n64 - 1; i >= 0; i--) u[i] = s ? ((x[i] << s) | (i ? (x[i - 1] >> (64 - s)) : 0)) : x[i];
    for (int j = n64 - n; j >= 0; j--) {
        u128w num = ((u128w)u[j + n] << 64) | u[j + n - 1];
        u128w qhat = num / v[n - 1], rhat = num % v[n - 1];
        while (qhat >> 64 || (n > 1 && qhat * v[n - 2] > ((rhat << 64) | u[j + n - 2]))) {
            qhat--; rhat += v[n - 1];
            if (rhat >> 64) break;
        }
        // u[j .. j+n] -= qhat * v
        u128w borrow = 0, carry = 0;
        for (int i = 0; i < n; i++) {
            const u128w p = qhat * v[i] + carry;
            carry = p >> 64;
            const u128w d = (u128w)u[i + j] - (uint64_t)p - borrow;
            u[i + j] = (uint64_t)d;
            borrow = (d >> 64) & 1;
        }
        const u128w d = (u128w)u[j + n] - carry - borrow;
        u[j + n] = (uint64_t)d;
        if ((d >> 64) & 1) {   // qhat was one too large: add the divisor back
            u128w c = 0;
            for (int i = 0; i < n; i++) { c += (u128w)u[i + j] + v[i]; u[i + j] = (uint64_t)c; c >>= 64; }
            u[j + n] += (uint64_t)c;
        }
    }
    U256 r = u_zero();
    for (int i = 0; i < n; i++) r.w[i] = s ? ((u[i] >> s) | ((i + 1 < n ? u[i + 1] : 0) << (64 - s))) : u[i];   // the remainder, shifted back
    return r;
}
static inline void u_mul_wide(const U256& a, const U256& b, uint64_t* out8) {   // full 512-bit product
    memset(out8, 0, 64);
    for (int i = 0; i < 4; i++) {
        u128 c = 0;
        for (int j = 0; j < 4; j++) { c += (u128)a.w[j] * b.w[i] + out8[i + j]; out8[i + j] = (uint64_t)c; c >>= 64; }
        out8[i + 4] = (uint64_t)c;
    }
}

// ---- SHA-256 (FIPS 180-4) over a BIT string: the circuit's Sha256(nBits) hashes strings whose length need not be a multiple of 8 ----------
struct BitString {
    std::vector<uint8_t> bytes;
    uint64_t nbits = 0;
    void put(unsigned b) {
        if ((nbits & 7) == 0) bytes.push_back(0);
        if (b) bytes.back() |= (uint8_t)(0x80u >> (nbits & 7));
        nbits++;
    }
    void be(const U256& v, unsigned n) {   // n bits of v, most significant first
        for (unsigned k = 0; k < n; k++) put(u_bit(v, n - 1 - k));
    }
    void be64(uint64_t v, unsigned n) { be(u_from64(v), n); }
    void zeros(unsigned n) { for (unsigned k = 0; k < n; k++) put(0); }
};
static inline uint32_t rotr32(uint32_t x, unsigned r) { return (x >> r) | (x << (32 - r)); }
static inline void sha256_bits(const BitString& in, uint8_t out[32]) {
    static const uint32_t K[64] = {
        0x428a2f98, 0x71374491, 0xb5c0fbcf, 0xe9b5dba5, 0x3956c25b, 0x59f111f1, 0x923f82a4, 0xab1c5ed5, 0xd807aa98, 0x12835b01, 0x243185be, 0x550c7dc3,
        0x72be5d74, 0x80deb1fe, 0x9bdc06a7, 0xc19bf174, 0xe49b69c1, 0xefbe4786, 0x0fc19dc6, 0x240ca1cc, 0x2de92c6f, 0x4a7484aa, 0x5cb0a9dc, 0x76f988da,
        0x983e5152, 0xa831c66d, 0xb00327c8, 0xbf597fc7, 0xc6e00bf3, 0xd5a79147, 0x06ca6351, 0x14292967, 0x27b70a85, 0x2e1b2138, 0x4d2c6dfc, 0x53380d13,
        0x650a7354, 0x766a0abb, 0x81c2c92e, 0x92722c85, 0xa2bfe8a1, 0xa81a664b, 0xc24b8b70, 0xc76c51a3, 0xd192e819, 0xd6990624, 0xf40e3585, 0x106aa070,
        0x19a4c116, 0x1e376c08, 0x2748774c, 0x34b0bcb5, 0x391c0cb3, 0x4ed8aa4a, 0x5b9cca4f, 0x682e6ff3, 0x748f82ee, 0x78a5636f, 0x84c87814, 0x8cc70208,
        0x90befffa, 0xa4506ceb, 0xbef9a3f7, 0xc67178f2};
    BitString m = in;
    m.put(1);
    while (m.nbits % 512 != 448) m.put(0);
    m.be64(in.nbits, 64);
    uint32_t h[8] = {0x6a09e667, 0xbb67ae85, 0x3c6ef372, 0xa54ff53a, 0x510e527f, 0x9b05688c, 0x1f83d9ab, 0x5be0cd19};
    for (size_t b = 0; b < m.bytes.size(); b += 64) {
        uint32_t w[64];
        for (int i = 0; i < 16; i++) w[i] = ((uint32_t)m.bytes[b + 4 * i] << 24) | ((uint32_t)m.bytes[b + 4 * i + 1] << 16) | ((uint32_t)m.bytes[b + 4 * i + 2] << 8) | m.bytes[b + 4 * i + 3];
        for (int i = 16; i < 64; i++) {
            const uint32_t s0 = rotr32(w[i - 15], 7) ^ rotr32(w[i - 15], 18) ^ (w[i - 15] >> 3);
            const uint32_t s1 = rotr32(w[i - 2], 17) ^ rotr32(w[i - 2], 19) ^ (w[i - 2] >> 10);
            w[i] = w[i - 16] + s0 + w[i - 7] + s1;
        }
        uint32_t a = h[0], bb = h[1], c = h[2], d = h[3], e = h[4], f = h[5], g = h[6], hh = h[7];
        for (int i = 0; i < 64; i++) {
            const uint32_t t1 = hh + (rotr32(e, 6) ^ rotr32(e, 11) ^ rotr32(e, 25)) + ((e & f) ^ (~e & g)) + K[i] + w[i];
            const uint32_t t2 = (rotr32(a, 2) ^ rotr32(a, 13) ^ rotr32(a, 22)) + ((a & bb) ^ (a & c) ^ (bb & c));
            hh = g; g = f; f = e; e = d + t1; d = c; c = bb; bb = a; a = t1 + t2;
        }
        h[0] += a; h[1] += bb; h[2] += c; h[3] += d; h[4] += e; h[5] += f; h[6] += g; h[7] += hh;
    }
    for (int i = 0; i < 8; i++) { out[4 * i] = (uint8_t)(h[i] >> 24); out[4 * i + 1] = (uint8_t)(h[i] >> 16); out[4 * i + 2] = (uint8_t)(h[i] >> 8); out[4 * i + 3] = (uint8_t)h[i]; }
}

// ---- SHA-512 (FIPS 180-4), byte strings -------------------------------------------------------------------------------------------
static inline uint64_t rotr64(uint64_t x, unsigned r) { return (x >> r) | (x << (64 - r)); }
static inline void sha512(const uint8_t* data, size_t len, uint8_t out[64]) {
    static const uint64_t K[80] = {
        0x428a2f98d728ae22ull, 0x7137449123ef65cdull, 0xb5c0fbcfec4d3b2full, 0xe9b5dba58189dbbcull, 0x3956c25bf348b538ull, 0x59f111f1b605d019ull,
        0x923f82a4af194f9bull, 0xab1c5ed5da6d8118ull, 0xd807aa98a3030242ull, 0x12835b0145706fbeull, 0x243185be4ee4b28cull, 0x550c7dc3d5ffb4e2ull,
        0x72be5d74f27b896full, 0x80deb1fe3b1696b1ull, 0x9bdc06a725c71235ull, 0xc19bf174cf692694ull, 0xe49b69c19ef14ad2ull, 0xefbe4786384f25e3ull,
        0x0fc19dc68b8cd5b5ull, 0x240ca1cc77ac9c65ull, 0x2de92c6f592b0275ull, 0x4a7484aa6ea6e483ull, 0x5cb0a9dcbd41fbd4ull, 0x76f988da831153b5ull,
        0x983e5152ee66dfabull, 0xa831c66d2db43210ull, 0xb00327c898fb213full, 0xbf597fc7beef0ee4ull, 0xc6e00bf33da88fc2ull, 0xd5a79147930aa725ull,
        0x06ca6351e003826full, 0x142929670a0e6e70ull, 0x27b70a8546d22ffcull, 0x2e1b21385c26c926ull, 0x4d2c6dfc5ac42aedull, 0x53380d139d95b3dfull,
        0x650a73548baf63deull, 0x766a0abb3c77b2a8ull, 0x81c2c92e47edaee6ull, 0x92722c851482353bull, 0xa2bfe8a14cf10364ull, 0xa81a664bbc423001ull,
        0xc24b8b70d0f89791ull, 0xc76c51a30654be30ull, 0xd192e819d6ef5218ull, 0xd69906245565a910ull, 0xf40e35855771202aull, 0x106aa07032bbd1b8ull,
        0x19a4c116b8d2d0c8ull, 0x1e376c085141ab53ull, 0x2748774cdf8eeb99ull, 0x34b0bcb5e19b48a8ull, 0x391c0cb3c5c95a63ull, 0x4ed8aa4ae3418acbull,
        0x5b9cca4f7763e373ull, 0x682e6ff3d6b2b8a3ull, 0x748f82ee5defb2fcull, 0x78a5636f43172f60ull, 0x84c87814a1f0ab72ull, 0x8cc702081a6439ecull,
        0x90befffa23631e28ull, 0xa4506cebde82bde9ull, 0xbef9a3f7b2c67915ull, 0xc67178f2e372532bull, 0xca273eceea26619cull, 0xd186b8c721c0c207ull,
        0xeada7dd6cde0eb1eull, 0xf57d4f7fee6ed178ull, 0x06f067aa72176fbaull, 0x0a637dc5a2c898a6ull, 0x113f9804bef90daeull, 0x1b710b35131c471bull,
        0x28db77f523047d84ull, 0x32caab7b40c72493ull, 0x3c9ebe0a15c9bebcull, 0x431d67c49c100d4cull, 0x4cc5d4becb3e42b6ull, 0x597f299cfc657e2aull,
        0x5fcb6fab3ad6faecull, 0x6c44198c4a475817ull};
    std::vector<uint8_t> m(data, data + len);
    m.push_back(0x80);
    while (m.size() % 128 != 112) m.push_back(0);
    for (int i = 0; i < 8; i++) m.push_back(0);
    const uint64_t bits = (uint64_t)len * 8;
    for (int i = 7; i >= 0; i--) m.push_back((uint8_t)(bits >> (8 * i)));
    uint64_t h[8] = {0x6a09e667f3bcc908ull, 0xbb67ae8584caa73bull, 0x3c6ef372fe94f82bull, 0xa54ff53a5f1d36f1ull,
                     0x510e527fade682d1ull, 0x9b05688c2b3e6c1full, 0x1f83d9abfb41bd6bull, 0x5be0cd19137e2179ull};
    for (size_t b = 0; b < m.size(); b += 128) {
        uint64_t w[80];
        for (int i = 0; i < 16; i++) {
            w[i] = 0;
            for (int k = 0; k < 8; k++) w[i] = (w[i] << 8) | m[b + 8 * i + k];
        }
        for (int i = 16; i < 80; i++) {
            const uint64_t s0 = rotr64(w[i - 15], 1) ^ rotr64(w[i - 15], 8) ^ (w[i - 15] >> 7);
            const uint64_t s1 = rotr64(w[i - 2], 19) ^ rotr64(w[i - 2], 61) ^ (w[i - 2] >> 6);
            w[i] = w[i - 16] + s0 + w[i - 7] + s1;
        }
        uint64_t a = h[0], bb = h[1], c = h[2], d = h[3], e = h[4], f = h[5], g = h[6], hh = h[7];
        for (int i = 0; i < 80; i++) {
            const uint64_t t1 = hh + (rotr64(e, 14) ^ rotr64(e, 18) ^ rotr64(e, 41)) + ((e & f) ^ (~e & g)) + K[i] + w[i];
            const uint64_t t2 = (rotr64(a, 28) ^ rotr64(a, 34) ^ rotr64(a, 39)) + ((a & bb) ^ (a & c) ^ (bb & c));
            hh = g; g = f; f = e; e = d + t1; d = c; c = bb; bb = a; a = t1 + t2;
        }
        h[0] += a; h[1] += bb; h[2] += c; h[3] += d; h[4] += e; h[5] += f; h[6] += g; h[7] += hh;
    }
    for (int i = 0; i < 8; i++)
        for (int k = 0; k < 8; k++) out[8 * i + k] = (uint8_t)(h[i] >> (56 - 8 * k));
}

}  // namespace hzh
