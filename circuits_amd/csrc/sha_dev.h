// Bit-level SHA-256 compression witness (circomlib 0.5.2 sha256/*.circom: sha256compression,
// sigmaplus, sigma, t1, t2, ch, maj, xor3, binsum; absent from /root/reference -- standard
// FIPS 180-4 arithmetic, so values are checkable against any SHA-256). Used by HashInputs
// (reference src/hash-inputs.circom:112-184) and HashInputsWithdrawal (src/withdraw.circom:84-176).
// Per-block signal order: see include/hz_layout.h (SHA_SCHED_W / SHA_ROUND_W).
#pragma once
#include "gadgets_dev.h"

namespace hz {

__constant__ const uint32_t SHA_K[64] = {
    0x428a2f98, 0x71374491, 0xb5c0fbcf, 0xe9b5dba5, 0x3956c25b, 0x59f111f1, 0x923f82a4, 0xab1c5ed5, 0xd807aa98, 0x12835b01, 0x243185be,
    0x550c7dc3, 0x72be5d74, 0x80deb1fe, 0x9bdc06a7, 0xc19bf174, 0xe49b69c1, 0xefbe4786, 0x0fc19dc6, 0x240ca1cc, 0x2de92c6f, 0x4a7484aa,
    0x5cb0a9dc, 0x76f988da, 0x983e5152, 0xa831c66d, 0xb00327c8, 0xbf597fc7, 0xc6e00bf3, 0xd5a79147, 0x06ca6351, 0x14292967, 0x27b70a85,
    0x2e1b2138, 0x4d2c6dfc, 0x53380d13, 0x650a7354, 0x766a0abb, 0x81c2c92e, 0x92722c85, 0xa2bfe8a1, 0xa81a664b, 0xc24b8b70, 0xc76c51a3,
    0xd192e819, 0xd6990624, 0xf40e3585, 0x106aa070, 0x19a4c116, 0x1e376c08, 0x2748774c, 0x34b0bcb5, 0x391c0cb3, 0x4ed8aa4a, 0x5b9cca4f,
    0x682e6ff3, 0x748f82ee, 0x78a5636f, 0x84c87814, 0x8cc70208, 0x90befffa, 0xa4506ceb, 0xbef9a3f7, 0xc67178f2};
__constant__ const uint32_t SHA_H0[8] = {0x6a09e667, 0xbb67ae85, 0x3c6ef372, 0xa54ff53a, 0x510e527f, 0x9b05688c, 0x1f83d9ab, 0x5be0cd19};

__device__ __forceinline__ uint32_t rotr32(uint32_t x, int r) { return (x >> r) | (x << (32 - r)); }

// plain compression (chaining value only)
// (both loops unrolled: a schedule array indexed by a loop counter lives in private memory, 256 bytes per lane)
__device__ __forceinline__ void sha256_compress(uint32_t* hv, const uint32_t* w16) {
    uint32_t w[64];
#pragma unroll
    for (int t = 0; t < 16; t++) w[t] = w16[t];
#pragma unroll
    for (int t = 16; t < 64; t++) {
        const uint32_t s0 = rotr32(w[t - 15], 7) ^ rotr32(w[t - 15], 18) ^ (w[t - 15] >> 3);
        const uint32_t s1 = rotr32(w[t - 2], 17) ^ rotr32(w[t - 2], 19) ^ (w[t - 2] >> 10);
        w[t] = s1 + w[t - 7] + s0 + w[t - 16];
    }
    uint32_t a = hv[0], b = hv[1], c = hv[2], d = hv[3], e = hv[4], f = hv[5], g = hv[6], h = hv[7];
#pragma unroll
    for (int t = 0; t < 64; t++) {
        const uint32_t t1 = h + (rotr32(e, 6) ^ rotr32(e, 11) ^ rotr32(e, 25)) + ((e & f) ^ (~e & g)) + SHA_K[t] + w[t];
        const uint32_t t2 = (rotr32(a, 2) ^ rotr32(a, 13) ^ rotr32(a, 22)) + ((a & b) ^ (a & c) ^ (b & c));
        h = g; g = f; f = e; e = d + t1; d = c; c = b; b = a; a = t1 + t2;
    }
    hv[0] += a; hv[1] += b; hv[2] += c; hv[3] += d; hv[4] += e; hv[5] += f; hv[6] += g; hv[7] += h;
}

// n consecutive bit signals off .. off+n-1 of this lane's unit = the low n bits of v.
// A bit is a 32-byte element (bit, 0, ..., 0); stored lane by lane (store_fr) every instruction writes 16 of every 32 bytes and the
// second one of a pair nothing but zeros -- measured at 3.5-4.6 TB/s (tools/microbench/storebench.hip, pattern 4). When the 32
// lanes of a half-wavefront own 32 consecutive units (io.coop32) the 1 KiB row of a signal is written as two fully contiguous
// 512-byte instructions instead: lane L of the group carries the first or second 16 bytes of unit L/2 (then 16 + L/2), the bits
// travel once per word through two lane permutes (pattern 5: 4.9-5.7 TB/s). Same bytes at the same addresses.
// INVARIANT of the cooperative path: all 32 lanes of the aligned half-wavefront are active, converged, and call with the same `off`
// and `n` (they store for each other). The SHA-256 kernels guarantee it by construction: unit counts that are multiples of 32,
// lane order (part, block, batch), HZ_BLOCK a multiple of 64, no per-lane branch around a bit store. A half-wavefront that is not
// whole (a future data-dependent branch, an early return) takes the lane-by-lane path instead of corrupting its neighbours' rows.
static_assert(HZ_BLOCK % 64 == 0, "put_word_bits: half-wavefront groups need whole wavefronts");
__device__ __forceinline__ void put_word_bits(const UnitIO& io, uint32_t off, uint64_t v, int n) {
    const unsigned long long act = __ballot(1);
    const bool whole = (__lane_id() & 32u) ? (act >> 32) == 0xffffffffull : (act & 0xffffffffull) == 0xffffffffull;
    if (io.coop32 && whole) {
        const uint32_t lane = __lane_id(), gl = lane & 31u;
        const int src = (int)((lane & 32u) | (gl >> 1));
        uint32_t loA = __shfl((uint32_t)v, src), loB = __shfl((uint32_t)v, src + 16), hiA = 0, hiB = 0;
        if (n > 32) { hiA = __shfl((uint32_t)(v >> 32), src); hiB = __shfl((uint32_t)(v >> 32), src + 16); }
        if (gl & 1u) loA = loB = hiA = hiB = 0u;   // odd lanes carry the upper half of an element: zeros
        const uint64_t vA = ((uint64_t)hiA << 32) | loA, vB = ((uint64_t)hiB << 32) | loB;
        uint8_t* p = io.addr(off) - (size_t)gl * 16;
        const size_t row = (size_t)io.n_units * 32;
        for (int k = 0; k < n; k++) {
            uint4* q = reinterpret_cast<uint4*>(p);
            q[0] = make_uint4((uint32_t)(vA >> k) & 1u, 0u, 0u, 0u);
            q[32] = make_uint4((uint32_t)(vB >> k) & 1u, 0u, 0u, 0u);
            p += row;
        }
        return;
    }
    for (int k = 0; k < n; k++) io.put_bit(off + k, (uint32_t)((v >> k) & 1));
}
// Xor3: mid = b & c, out = a ^ b ^ c  (64 signals)
__device__ __forceinline__ uint32_t xor3_dev(const UnitIO& io, uint32_t off, uint32_t a, uint32_t b, uint32_t c) {
    put_word_bits(io, off, b & c, 32);
    const uint32_t o = a ^ b ^ c;
    put_word_bits(io, off + 32, o, 32);
    return o;
}

// compression with the full bit-level witness written at signal offset `base`.
// The message schedule is a WINDOW of sixteen words that moves with the round -- r[0] = w[t] in round t, the step that makes w[t + 16]
// follows the round and the window shifts by one (fifteen register moves beside ~360 stored signals) -- instead of an array w[64] indexed
// by the loop counter, which is 256 bytes of private memory per lane. Same signals at the same offsets; only the order in which a lane
// issues its stores changes (schedule step t + 16 after round t).
__device__ void sha256_block_witness(const UnitIO& io, uint32_t base, uint32_t* hv, const uint32_t* w16) {
    uint32_t r[16];
#pragma unroll
    for (int t = 0; t < 16; t++) r[t] = w16[t];
    uint32_t a = hv[0], b = hv[1], c = hv[2], d = hv[3], e = hv[4], f = hv[5], g = hv[6], h = hv[7];
    const uint32_t rbase = base + 48 * SHA_SCHED_W;
#pragma unroll 1
    for (int t = 0; t < 64; t++) {
        const uint32_t o = rbase + (uint32_t)t * SHA_ROUND_W;
        const uint32_t S1 = xor3_dev(io, o, rotr32(e, 6), rotr32(e, 11), rotr32(e, 25));
        const uint32_t ch = (e & f) ^ (~e & g);
        put_word_bits(io, o + 64, ch, 32);
        const uint64_t t1 = (uint64_t)h + S1 + ch + SHA_K[t] + r[0];
        put_word_bits(io, o + 96, t1, 35);
        const uint32_t S0 = xor3_dev(io, o + 131, rotr32(a, 2), rotr32(a, 13), rotr32(a, 22));
        const uint32_t mid = b & c, maj = (a & b) ^ (a & c) ^ (b & c);
        put_word_bits(io, o + 195, mid, 32);
        put_word_bits(io, o + 227, maj, 32);
        const uint64_t t2 = (uint64_t)S0 + maj;
        put_word_bits(io, o + 259, t2, 33);
        const uint64_t se = (uint64_t)d + (uint32_t)t1, sa = (uint64_t)(uint32_t)t1 + (uint32_t)t2;
        put_word_bits(io, o + 292, se, 33);
        put_word_bits(io, o + 325, sa, 33);
        h = g; g = f; f = e; e = (uint32_t)se; d = c; c = b; b = a; a = (uint32_t)sa;
        uint32_t nw = 0;
        if (t < 48) {   // schedule step t + 16: w[t + 16] from w[t + 1], w[t + 14], w[t + 9], w[t]
            const uint32_t so = base + (uint32_t)t * SHA_SCHED_W;
            const uint32_t x15 = r[1], x2 = r[14];
            const uint32_t s0 = xor3_dev(io, so, rotr32(x15, 7), rotr32(x15, 18), x15 >> 3);
            const uint32_t s1 = xor3_dev(io, so + 64, rotr32(x2, 17), rotr32(x2, 19), x2 >> 10);
            const uint64_t sum = (uint64_t)s1 + r[9] + s0 + r[0];
            put_word_bits(io, so + 128, sum, 34);
            nw = (uint32_t)sum;
        }
#pragma unroll
        for (int i = 0; i < 15; i++) r[i] = r[i + 1];
        r[15] = nw;
    }
    const uint32_t fbase = rbase + 64 * SHA_ROUND_W;
    const uint32_t st[8] = {a, b, c, d, e, f, g, h};
#pragma unroll
    for (int i = 0; i < 8; i++) {
        const uint64_t s = (uint64_t)hv[i] + st[i];
        put_word_bits(io, fbase + 33 * i, s, 33);
        hv[i] = (uint32_t)s;
    }
}

// The same witness split over `nparts` lanes: lane `part` stores the schedule steps and the rounds of its slice (and the last lane
// the final additions); what precedes its slice is recomputed without stores (at most 48 + 64 cheap steps against ~3 600 stored
// signals per lane at nparts = 8). A block is then nparts independent store streams instead of one 29 k-signal stream.
__device__ void sha256_block_witness_part(const UnitIO& io, uint32_t base, const uint32_t* hv, const uint32_t* w16, uint32_t part, uint32_t nparts) {
    uint32_t r[16];   // the schedule window of sha256_block_witness
#pragma unroll
    for (int t = 0; t < 16; t++) r[t] = w16[t];
    const int s_lo = 16 + (int)(48u * part / nparts), s_hi = 16 + (int)(48u * (part + 1) / nparts);
    const int r_lo = (int)(64u * part / nparts), r_hi = (int)(64u * (part + 1) / nparts);
    uint32_t a = hv[0], b = hv[1], c = hv[2], d = hv[3], e = hv[4], f = hv[5], g = hv[6], h = hv[7];
    const uint32_t rbase = base + 48 * SHA_SCHED_W;
    // (the slice's last schedule step, s_hi - 1, is made in iteration s_hi - 17 < r_hi: 48 (part + 1) / nparts <= 64 (part + 1) / nparts)
#pragma unroll 1
    for (int t = 0; t < r_hi; t++) {
        const uint32_t S1 = rotr32(e, 6) ^ rotr32(e, 11) ^ rotr32(e, 25);
        const uint32_t ch = (e & f) ^ (~e & g);
        const uint64_t t1 = (uint64_t)h + S1 + ch + SHA_K[t] + r[0];
        const uint32_t S0 = rotr32(a, 2) ^ rotr32(a, 13) ^ rotr32(a, 22);
        const uint32_t mid = b & c, maj = (a & b) ^ (a & c) ^ (b & c);
        const uint64_t t2 = (uint64_t)S0 + maj;
        const uint64_t se = (uint64_t)d + (uint32_t)t1, sa = (uint64_t)(uint32_t)t1 + (uint32_t)t2;
        if (t >= r_lo) {
            const uint32_t o = rbase + (uint32_t)t * SHA_ROUND_W;
            (void)xor3_dev(io, o, rotr32(e, 6), rotr32(e, 11), rotr32(e, 25));
            put_word_bits(io, o + 64, ch, 32);
            put_word_bits(io, o + 96, t1, 35);
            (void)xor3_dev(io, o + 131, rotr32(a, 2), rotr32(a, 13), rotr32(a, 22));
            put_word_bits(io, o + 195, mid, 32);
            put_word_bits(io, o + 227, maj, 32);
            put_word_bits(io, o + 259, t2, 33);
            put_word_bits(io, o + 292, se, 33);
            put_word_bits(io, o + 325, sa, 33);
        }
        h = g; g = f; f = e; e = (uint32_t)se; d = c; c = b; b = a; a = (uint32_t)sa;
        uint32_t nw = 0;
        const int ts = t + 16;   // the schedule step that follows round t
        if (ts < 64) {
            const uint32_t x15 = r[1], x2 = r[14];
            if (ts >= s_lo && ts < s_hi) {
                const uint32_t so = base + (uint32_t)t * SHA_SCHED_W;
                const uint32_t s0 = xor3_dev(io, so, rotr32(x15, 7), rotr32(x15, 18), x15 >> 3);
                const uint32_t s1 = xor3_dev(io, so + 64, rotr32(x2, 17), rotr32(x2, 19), x2 >> 10);
                const uint64_t sum = (uint64_t)s1 + r[9] + s0 + r[0];
                put_word_bits(io, so + 128, sum, 34);
                nw = (uint32_t)sum;
            } else {
                const uint32_t s0 = rotr32(x15, 7) ^ rotr32(x15, 18) ^ (x15 >> 3);
                const uint32_t s1 = rotr32(x2, 17) ^ rotr32(x2, 19) ^ (x2 >> 10);
                nw = s1 + r[9] + s0 + r[0];
            }
        }
#pragma unroll
        for (int i = 0; i < 15; i++) r[i] = r[i + 1];
        r[15] = nw;
    }
    if (part + 1 == nparts) {
        const uint32_t fbase = rbase + 64 * SHA_ROUND_W;
        const uint32_t st[8] = {a, b, c, d, e, f, g, h};
#pragma unroll
        for (int i = 0; i < 8; i++) put_word_bits(io, fbase + 33 * i, (uint64_t)hv[i] + st[i], 33);
    }
}

// digest (8 big-endian words) as a 256-bit integer reduced mod r -> canonical Fr
__device__ __forceinline__ Fc sha_digest_to_fr(const uint32_t* hv) {
    Fc r;
    for (int i = 0; i < 8; i++) r.v[i] = hv[7 - i];
    // value < 2^256 < 6r: subtract r while >= r
    for (int k = 0; k < 5; k++) fc_cond_sub_p(r.v);
    return r;
}

// set message bit p (0 = first bit of the message) in a word buffer laid out as SHA words
__device__ __forceinline__ void msg_set_bit(uint32_t* msgw, uint64_t p, uint32_t bit) {
    if (bit) atomicOr(&msgw[p >> 5], 1u << (31 - (uint32_t)(p & 31)));
}

}  // namespace hz
