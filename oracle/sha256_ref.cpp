// TEST INFRASTRUCTURE -- CPU oracle. Bit-level SHA-256 witness following circomlib 0.5.2
// sha256/{sha256,sha256compression,sigmaplus,sigma,t1,t2,ch,maj,xor3,rotate,shift,binsum}.circom
// (not on disk; standard FIPS 180-4 arithmetic, so values are checkable against hashlib) and
// src/hash-inputs.circom:23-185 of the reference.
#include "templates_ref.h"

namespace orc {
using namespace hzl;

static const uint32_t K256[64] = {
    0x428a2f98, 0x71374491, 0xb5c0fbcf, 0xe9b5dba5, 0x3956c25b, 0x59f111f1, 0x923f82a4, 0xab1c5ed5, 0xd807aa98, 0x12835b01, 0x243185be,
    0x550c7dc3, 0x72be5d74, 0x80deb1fe, 0x9bdc06a7, 0xc19bf174, 0xe49b69c1, 0xefbe4786, 0x0fc19dc6, 0x240ca1cc, 0x2de92c6f, 0x4a7484aa,
    0x5cb0a9dc, 0x76f988da, 0x983e5152, 0xa831c66d, 0xb00327c8, 0xbf597fc7, 0xc6e00bf3, 0xd5a79147, 0x06ca6351, 0x14292967, 0x27b70a85,
    0x2e1b2138, 0x4d2c6dfc, 0x53380d13, 0x650a7354, 0x766a0abb, 0x81c2c92e, 0x92722c85, 0xa2bfe8a1, 0xa81a664b, 0xc24b8b70, 0xc76c51a3,
    0xd192e819, 0xd6990624, 0xf40e3585, 0x106aa070, 0x19a4c116, 0x1e376c08, 0x2748774c, 0x34b0bcb5, 0x391c0cb3, 0x4ed8aa4a, 0x5b9cca4f,
    0x682e6ff3, 0x748f82ee, 0x78a5636f, 0x84c87814, 0x8cc70208, 0x90befffa, 0xa4506ceb, 0xbef9a3f7, 0xc67178f2};
static const uint32_t H256[8] = {0x6a09e667, 0xbb67ae85, 0x3c6ef372, 0xa54ff53a, 0x510e527f, 0x9b05688c, 0x1f83d9ab, 0x5be0cd19};

static inline uint32_t rotr(uint32_t x, int r) { return (x >> r) | (x << (32 - r)); }

// Xor3 on words a,b,c: mid = b&c (per-bit product), out = a^b^c; writes 64 signals
static uint32_t xor3_w(const W& w, uint32_t off, uint32_t a, uint32_t b, uint32_t c) {
    const uint32_t mid = b & c, out = a ^ b ^ c;
    for (int k = 0; k < 32; k++) { w.set(off + k, F((int)((mid >> k) & 1))); w.set(off + 32 + k, F((int)((out >> k) & 1))); }
    return out;
}
static void put_bits(const W& w, uint32_t off, uint64_t v, int n) {
    for (int k = 0; k < n; k++) w.set(off + k, F((int)((v >> k) & 1)));
}

std::vector<int> sha256_bits(const W& w, const Sha256Off& so, const std::vector<int>& bits) {
    const size_t nBits = bits.size();
    const size_t nBlocks = (nBits + 64) / 512 + 1;
    std::vector<int> padded(nBlocks * 512, 0);
    for (size_t k = 0; k < nBits; k++) padded[k] = bits[k];
    padded[nBits] = 1;
    for (int k = 0; k < 64; k++) padded[nBlocks * 512 - k - 1] = (int)(((uint64_t)nBits >> k) & 1);
    uint32_t hv[8];
    memcpy(hv, H256, sizeof hv);
    for (size_t blk = 0; blk < nBlocks; blk++) {
        const uint32_t base = so.blocks + (uint32_t)(blk * so.block_size);
        uint32_t wv[64];
        for (int t = 0; t < 16; t++) {
            uint32_t x = 0;
            for (int k = 0; k < 32; k++) x = (x << 1) | (uint32_t)padded[blk * 512 + t * 32 + k];
            wv[t] = x;
        }
        for (int t = 16; t < 64; t++) {
            const uint32_t o = base + (uint32_t)(t - 16) * SHA_SCHED_W;
            const uint32_t x15 = wv[t - 15], x2 = wv[t - 2];
            const uint32_t s0 = xor3_w(w, o, rotr(x15, 7), rotr(x15, 18), x15 >> 3);
            const uint32_t s1 = xor3_w(w, o + 64, rotr(x2, 17), rotr(x2, 19), x2 >> 10);
            const uint64_t sum = (uint64_t)s1 + wv[t - 7] + s0 + wv[t - 16];
            put_bits(w, o + 128, sum, 34);
            wv[t] = (uint32_t)sum;
        }
        uint32_t a = hv[0], b = hv[1], c = hv[2], d = hv[3], e = hv[4], f = hv[5], g = hv[6], h = hv[7];
        const uint32_t rbase = base + 48 * SHA_SCHED_W;
        for (int t = 0; t < 64; t++) {
            const uint32_t o = rbase + (uint32_t)t * SHA_ROUND_W;
            const uint32_t S1 = xor3_w(w, o, rotr(e, 6), rotr(e, 11), rotr(e, 25));
            const uint32_t ch = (e & f) ^ (~e & g);
            put_bits(w, o + 64, ch, 32);
            const uint64_t t1 = (uint64_t)h + S1 + ch + K256[t] + wv[t];
            put_bits(w, o + 96, t1, 35);
            const uint32_t S0 = xor3_w(w, o + 131, rotr(a, 2), rotr(a, 13), rotr(a, 22));
            const uint32_t mid = b & c, maj = (a & b) ^ (a & c) ^ (b & c);
            put_bits(w, o + 195, mid, 32);
            put_bits(w, o + 227, maj, 32);
            const uint64_t t2 = (uint64_t)S0 + maj;
            put_bits(w, o + 259, t2, 33);
            const uint64_t se = (uint64_t)d + (uint32_t)t1;
            const uint64_t sa = (uint64_t)(uint32_t)t1 + (uint32_t)t2;
            put_bits(w, o + 292, se, 33);
            put_bits(w, o + 325, sa, 33);
            h = g; g = f; f = e; e = (uint32_t)se; d = c; c = b; b = a; a = (uint32_t)sa;
        }
        const uint32_t fbase = rbase + 64 * SHA_ROUND_W;
        const uint32_t st[8] = {a, b, c, d, e, f, g, h};
        for (int i = 0; i < 8; i++) {
            const uint64_t s = (uint64_t)hv[i] + st[i];
            put_bits(w, fbase + 33 * i, s, 33);
            hv[i] = (uint32_t)s;
        }
    }
    std::vector<int> out(256);
    for (int i = 0; i < 8; i++)
        for (int k = 0; k < 32; k++) out[32 * i + k] = (int)((hv[i] >> (31 - k)) & 1);
    return out;
}

F hash_inputs(const W& w, const HashInputsOff& o, int L, int nTx, int maxL1, int Fn, const HashInputsIn& in) {
    std::vector<int> msg;
    msg.reserve(o.totalBits);
    auto idx48 = [&](uint32_t off, const F& v) {
        const std::vector<int> b = num2bits(w, off, v, 48, C_HI_N2B);
        F pad(0);
        for (int i = L; i < 48; i++) pad += F(b[i]);
        w.chk(C_HI_PAD, pad, F(0));
        return b;
    };
    const std::vector<int> b_old = idx48(o.n2bOldLastIdx, in.oldLastIdx);
    const std::vector<int> b_new = idx48(o.n2bNewLastIdx, in.newLastIdx);
    const std::vector<int> b_osr = num2bits(w, o.n2bOldStateRoot, in.oldStateRoot, 256, C_HI_N2B);
    const std::vector<int> b_nsr = num2bits(w, o.n2bNewStateRoot, in.newStateRoot, 256, C_HI_N2B);
    const std::vector<int> b_ner = num2bits(w, o.n2bNewExitRoot, in.newExitRoot, 256, C_HI_N2B);
    std::vector<std::vector<int>> b_fee(Fn);
    for (int i = 0; i < Fn; i++) b_fee[i] = idx48(o.n2bFee + 48 * i, in.feeTxsData[i]);
    const std::vector<int> b_chain = num2bits(w, o.n2bChainID, in.globalChainID, 16, C_HI_N2B);
    const std::vector<int> b_batch = num2bits(w, o.n2bCurrentNumBatch, in.currentNumBatch, 32, C_HI_N2B);
    auto be = [&](const std::vector<int>& b, int n) { for (int i = n - 1; i >= 0; i--) msg.push_back(b[i]); };
    be(b_old, 48); be(b_new, 48); be(b_osr, 256); be(b_nsr, 256); be(b_ner, 256);
    // the data-availability inputs are "already in bits": the bit used by the hash is the value's LSB
    for (int i = 0; i < maxL1 * hzl::L1FULL_BITS; i++) msg.push_back(in.L1TxsFullData[i].bit(0));
    for (int i = 0; i < nTx * (2 * L + 48); i++) msg.push_back(in.L1L2TxsData[i].bit(0));
    for (int i = 0; i < Fn; i++) be(b_fee[i], L);
    be(b_chain, 16); be(b_batch, 32);
    const std::vector<int> dg = sha256_bits(w, o.sha, msg);
    // Bits2Num(256): in[i] = digest bit 255-i  -> big-endian integer, reduced mod r by field arithmetic
    F out(0);
    for (int i = 0; i < 256; i++)
        if (dg[255 - i]) out += pow2(i);
    if (o.out != ~0u) w.set(o.out, out);
    return out;
}

}  // namespace orc
