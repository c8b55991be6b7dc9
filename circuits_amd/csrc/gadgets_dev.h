// Device-side circuit gadgets: one lane evaluates the gadget for its unit and writes the gadget's
// stored signals (see include/hz_layout.h for which signals are stored and where).
// Restates circomlib 0.5.2 bitify / comparators / compconstant / mux (absent from
// /root/reference) for CDNA4 lanes.
#pragma once
#include "../../include/hz_layout.h"
#include "devcommon.h"

namespace hz {
using namespace hzl;

// One lane's view of a witness section: loads inputs, stores signals, reports failed constraints.
struct UnitIO {
    uint8_t* base;      // section base (bytes)
    uint32_t n_units;
    uint32_t unit;
    uint32_t inst;      // instance id for the failure key
    uint32_t err_unit;  // unit id for the failure key
    ErrBuf* err;
    // 1: the 32 lanes of this lane's aligned half-wavefront own 32 consecutive units and run in lockstep, so runs of bit signals may
    // be stored cooperatively (put_word_bits in sha_dev.h). Set by the kernels that guarantee it, wave-uniform.
    uint32_t coop32 = 0;
    // 1: this lane repeats another lane's work (the non-leading lanes of a quad in the latency form of k_smt): it reports no failed
    // constraint (its stores are the leader's, byte for byte)
    uint32_t mute = 0;

    __device__ __forceinline__ uint8_t* addr(uint32_t sig) const { return base + ((size_t)sig * n_units + unit) * 32; }
    __device__ __forceinline__ uint8_t* addr_u(uint32_t sig, uint32_t u) const { return base + ((size_t)sig * n_units + u) * 32; }
    __device__ __forceinline__ Fc in_c(uint32_t sig) const { return load_fr(addr(sig)); }                    // canonical
    __device__ __forceinline__ Fr in_m(uint32_t sig) const { return fr_from_canon(load_fr(addr(sig))); }      // Montgomery
    __device__ __forceinline__ Fc in_c_u(uint32_t sig, uint32_t u) const { return load_fr(addr_u(sig, u)); }
    __device__ __forceinline__ Fr in_m_u(uint32_t sig, uint32_t u) const { return fr_from_canon(load_fr(addr_u(sig, u))); }
    __device__ __forceinline__ void put_m(uint32_t sig, const Fr& m) const { store_fr(addr(sig), fr_to_canon(m)); }
    __device__ __forceinline__ void put_c(uint32_t sig, const Fc& c) const { store_fr(addr(sig), c); }
    __device__ __forceinline__ void put_u64(uint32_t sig, uint64_t x) const {
        Fc c;
        c.v[0] = (uint32_t)x; c.v[1] = (uint32_t)(x >> 32);
        c.v[2] = c.v[3] = c.v[4] = c.v[5] = c.v[6] = c.v[7] = 0u;
        store_fr(addr(sig), c);
    }
    __device__ __forceinline__ void put_bit(uint32_t sig, uint32_t b) const { put_u64(sig, b & 1u); }
    // `lhs === rhs` (Montgomery operands)
    __device__ __forceinline__ void chk(int cid, const Fr& lhs, const Fr& rhs) const {
        if (!mute && !fr_eq(lhs, rhs)) report_fail(err, inst, err_unit, (uint32_t)cid, lhs, rhs);
    }
    __device__ __forceinline__ void chk_zero(int cid, const Fr& lhs) const {
        if (!mute && !fr_is_zero(lhs)) report_fail(err, inst, err_unit, (uint32_t)cid, lhs, fr_zero());
    }
    __device__ __forceinline__ WitSboxSink sbox_sink(uint32_t sig0) const { return WitSboxSink{WitOut{base, n_units, unit}, sig0}; }
};

// inter-kernel scratch: Montgomery elements, field-major [field][unit]
struct Scratch {
    Fr* p;
    uint32_t n_units;
    uint32_t unit;
    // explicit global address space, like load_fr / store_fr (devcommon.h)
    typedef __attribute__((address_space(1))) uint32_t g_u32;
    __device__ __forceinline__ Fr get(uint32_t f) const {
        const g_u32* q = (const g_u32*)(p + ((size_t)f * n_units + unit));
        Fr r;
#pragma unroll
        for (int l = 0; l < 9; l++) r.v[l] = q[l];
        return r;
    }
    __device__ __forceinline__ void set(uint32_t f, const Fr& v) const {
        g_u32* q = (g_u32*)(p + ((size_t)f * n_units + unit));
#pragma unroll
        for (int l = 0; l < 9; l++) q[l] = v.v[l];
    }
};

// ---- canonical-integer helpers ---------------------------------------------------------------------
// None of these indexes a register array with a run-time value: an array indexed that way lives in scratch memory (round 4 counted
// 1 040 bytes per lane in k_smt, half of it the two keys whose bits the level loop reads). A word is picked by a chain of selects;
// with a constant position (most call sites, after inlining) the chain folds away.
__device__ __forceinline__ uint32_t c_word(const Fc& c, int w) {   // word w, 0 beyond the integer
    uint32_t r = c.v[0];
#pragma unroll
    for (int j = 1; j < 8; j++) r = (w == j) ? c.v[j] : r;
    return ((unsigned)w < 8u) ? r : 0u;
}
__device__ __forceinline__ uint32_t c_bit(const Fc& c, int i) { return (c_word(c, i >> 5) >> (i & 31)) & 1u; }
// bits [from, from+n) of a canonical integer as u64 (n <= 64)
__device__ __forceinline__ uint64_t c_bits64(const Fc& c, int from, int n) {
    const int w = from >> 5, sh = from & 31;
    const uint64_t lo = (uint64_t)c_word(c, w) | ((uint64_t)c_word(c, w + 1) << 32);
    uint64_t r = lo >> sh;
    if (sh) r |= (uint64_t)c_word(c, w + 2) << (64 - sh);
    return n >= 64 ? r : r & ((1ull << n) - 1ull);
}
// canonical integer with bits [from, from+n) of c moved to position 0 (n <= 256)
__device__ __forceinline__ Fc c_extract(const Fc& c, int from, int n) {
    const int w = from >> 5, sh = from & 31;
    Fc r;
#pragma unroll
    for (int j = 0; j < 8; j++) {
        const uint32_t lo = c_word(c, w + j), hi = c_word(c, w + j + 1);
        const uint32_t v = sh ? ((lo >> sh) | (hi << (32 - sh))) : lo;
        const int left = n - 32 * j;   // bits of this word that belong to the field
        r.v[j] = left >= 32 ? v : left <= 0 ? 0u : (v & ((1u << left) - 1u));
    }
    return r;
}
// is the canonical integer < 2^n ?
__device__ __forceinline__ bool c_fits(const Fc& c, int n) {
    uint32_t any = 0;
#pragma unroll
    for (int j = 0; j < 8; j++) {
        const int lo = n - 32 * j;     // bits of word j below 2^n
        any |= lo >= 32 ? 0u : lo <= 0 ? c.v[j] : (c.v[j] >> lo);
    }
    return any == 0;
}
// c |= v << pos for a value of at most 64 bits (pos + 64 may run past the top word: those bits are dropped). `pos` is a constant at
// every call site, so the three word indices are constants after inlining: no run-time indexed register array (those live in scratch).
__device__ __forceinline__ void c_or_bits64(Fc& c, const int pos, uint64_t v) {
    const int w = pos >> 5, sh = pos & 31;
    const uint32_t lo = (uint32_t)v, hi = (uint32_t)(v >> 32);
    const uint32_t p0 = lo << sh, p1 = sh ? ((lo >> (32 - sh)) | (hi << sh)) : hi, p2 = sh ? (hi >> (32 - sh)) : 0u;
#pragma unroll
    for (int j = 0; j < 8; j++) c.v[j] |= (j == w) ? p0 : (j == w + 1) ? p1 : (j == w + 2) ? p2 : 0u;
}
__device__ __forceinline__ Fc c_pow2(int k) {  // 2^k as a plain integer, k < 256
    Fc r;
#pragma unroll
    for (int j = 0; j < 8; j++) r.v[j] = ((k >> 5) == j) ? (1u << (k & 31)) : 0u;
    return r;
}
// Montgomery 2^k for k < 253
__device__ __forceinline__ Fr m_pow2(int k) { return fr_from_canon(c_pow2(k)); }
// stores the bits [first, first + count) of `word` (count <= 32) as signals off + first .. : the inner loop of every bit decomposition
__device__ __forceinline__ void put_word_bits_plain(const UnitIO& io, uint32_t off, uint32_t word, int count) {
    for (int b = 0; b < count; b++) io.put_bit(off + b, (word >> b) & 1u);
}
// bits [0, n) of a canonical integer as signals off .. off + n, word by word (constant word indices)
__device__ __forceinline__ void put_bits(const UnitIO& io, uint32_t off, const Fc& c, int n) {
#define HZ_PB(j) { const int cnt = n - 32 * (j); if (cnt > 0) put_word_bits_plain(io, off + 32 * (j), c.v[j], cnt < 32 ? cnt : 32); }
    HZ_PB(0) HZ_PB(1) HZ_PB(2) HZ_PB(3) HZ_PB(4) HZ_PB(5) HZ_PB(6) HZ_PB(7)
#undef HZ_PB
}

// Num2Bits(n): stores out[0..n) from the canonical value; reports `sum === in` when it cannot hold
__device__ __forceinline__ void num2bits_dev(const UnitIO& io, uint32_t off, const Fc& canon, int n, int cid) {
    put_bits(io, off, canon, n);
    if (n < 254 && !io.mute && !c_fits(canon, n)) {
        // lc1 = value mod 2^n
        const Fc lc = c_extract(canon, 0, n);
        report_fail(io.err, io.inst, io.err_unit, (uint32_t)cid, fr_from_canon(lc), fr_from_canon(canon));
    }
}

// IsZero with a precomputed inverse: stores inv,out; returns out (Montgomery 0/1)
__device__ __forceinline__ Fr is_zero_dev(const UnitIO& io, IsZOff off, const Fr& in_m, const Fr& inv_m) {
    const bool z = fr_is_zero(in_m);
    io.put_m(off, inv_m);
    io.put_bit(off + 1, z ? 1u : 0u);
    return fr_from_bit(z ? 1u : 0u);
}

// Montgomery batch inversion of n values in place (zeros stay zero). 3(n-1) products + 1 inversion.
template <int N>
__device__ __forceinline__ void batch_inv(Fr (&x)[N], int n) {
    Fr pre[N];
    Fr acc = fr_one();
    for (int i = 0; i < n; i++) {
        pre[i] = acc;
        if (!fr_is_zero(x[i])) acc = fr_mul(acc, x[i]);
    }
    Fr inv = fr_inv(acc);
    for (int i = n - 1; i >= 0; i--) {
        if (fr_is_zero(x[i])) continue;
        const Fr xi = x[i];
        x[i] = fr_mul(inv, pre[i]);
        inv = fr_mul(inv, xi);
    }
}

// two inverses with one inversion, zeros stay zero: batch_inv<2> without its arrays (a loop over two field products does not unroll, so
// `pre[]` and the operands would sit in private memory)
__device__ __forceinline__ void inv_pair(Fr& a, Fr& b) {
    const bool za = fr_is_zero(a), zb = fr_is_zero(b);
    const Fr one = fr_one();
    const Fr a1 = fr_select(za, one, a), b1 = fr_select(zb, one, b);   // (limb selects: `c ? x : y` on two structs selects an ADDRESS)
    const Fr inv = fr_inv(fr_mul(a1, b1));
    const Fr ia = fr_mul(inv, b1), ib = fr_mul(inv, a1);
    a = fr_select(za, a, ia);
    b = fr_select(zb, b, ib);
}

// A RUN of n IsZero gadgets (slot k: input load(k); store(k, in, inv) writes its signals -- is_zero_dev at one or several offsets) with
// batched inverses and no private memory. batch_inv above keeps its operands and prefix products in arrays indexed by a loop counter; a
// loop this size does not unroll, so those arrays live in scratch memory (three of 576 bytes per lane in k_withdraw and the
// FeeAccumulator, 1.5 KB in the RollupTx front until round 6). Here the prefix products of W slots are pushed into a register window
// that ROTATES (constant indices only) and popped by the backward pass, which evaluates its operand again (`load` is a cached input load
// and a conversion) instead of keeping it. One inversion per W slots. Returns the mask of the zero inputs (slots below 64).
template <int W, class LOAD, class STORE>
__device__ __forceinline__ uint64_t is_zero_run_store_dev(int n, LOAD load, STORE store) {
    const Fr one = fr_one(), zero = fr_zero();
    uint64_t zmask = 0;
    for (int base = 0; base < n; base += W) {
        const int cnt = (n - base) < W ? (n - base) : W;
        Fr pre[W];
#pragma unroll
        for (int q = 0; q < W; q++) pre[q] = zero;
        Fr acc = one;
#pragma unroll 1
        for (int k = 0; k < W; k++) {
            const Fr v = k < cnt ? load(base + (k < cnt ? k : 0)) : zero;
#pragma unroll
            for (int q = 0; q < W - 1; q++) pre[q] = pre[q + 1];
            pre[W - 1] = acc;
            if (!fr_is_zero(v)) acc = fr_mul(acc, v);
        }
        Fr inv = fr_inv(acc);
#pragma unroll 1
        for (int k = W - 1; k >= 0; k--) {
            if (k < cnt) {
                const Fr v = load(base + k);
                Fr vi = zero;
                if (!fr_is_zero(v)) { vi = fr_mul(inv, pre[W - 1]); inv = fr_mul(inv, v); }
                else if (base + k < 64) zmask |= 1ull << (base + k);
                store(base + k, v, vi);
            }
#pragma unroll
            for (int q = W - 1; q > 0; q--) pre[q] = pre[q - 1];
        }
    }
    return zmask;
}
template <int W, class LOAD, class OFF>
__device__ __forceinline__ uint64_t is_zero_run_dev(const UnitIO& io, int n, LOAD load, OFF off) {
    return is_zero_run_store_dev<W>(n, load, [&](int k, const Fr& v, const Fr& vi) __attribute__((always_inline)) { (void)is_zero_dev(io, off(k), v, vi); });
}

__device__ __forceinline__ Fr mux1_dev(const Fr& c0, const Fr& c1, const Fr& s) { return fr_add(fr_mul(fr_sub(c1, c0), s), c0); }

// CompConstant(ct) over 254 bits given as canonical integer `bits` (bit i = in[i]); `nbits_valid`
// lets the caller force upper inputs to 0 (EdDSA passes 253 bits + a zero). Stores parts[127] and
// num2bits.out[135]; returns out (1 if in > ct).
// parts are small: each is  +-b_i, +-a_i, ... with a_i = 2^i, b_i = 2^128 - 2^i, so the sum fits
// in 135 bits of a plain integer; evaluated with 192-bit integer arithmetic, no field products.
__device__ __forceinline__ uint32_t comp_constant_dev(const UnitIO& io, const CompConstOff& off, const Fc& bits, const uint32_t* ct /*8 LE limbs*/) {
    // sum accumulates in 5 x 32-bit limbs (160 bits); a = 2^i walks up as a 128-bit integer, b = 2^128 - 2^i is its negative mod 2^128.
    // Pairs of bits are taken word by word (constant word indices), 16 pairs per word, 15 from the last one: no array is indexed at run time.
    uint32_t s0 = 0, s1 = 0, s2 = 0, s3 = 0, s4 = 0;
    uint32_t a0 = 1, a1 = 0, a2 = 0, a3 = 0;
    auto word = [&](uint32_t sw, uint32_t cw, const int w, const int pairs) __attribute__((always_inline)) {
        for (int q = 0; q < pairs; q++) {
            const int i = 16 * w + q;
            const uint32_t clsb = cw & 1u, cmsb = (cw >> 1) & 1u, slsb = sw & 1u, smsb = (sw >> 1) & 1u;
            cw >>= 2; sw >>= 2;
            // part in {0, a, b}:
            //  c=00: -b*m*l + b*m + b*l  -> b if (m|l) else 0
            //  c=01: a*m*l - a*l + b*m - a*m + a -> m ? b : (l ? 0 : a)
            //  c=10: b*m*l - a*m + a -> m ? (l ? b : 0) : a
            //  c=11: -a*m*l + a -> (m&l) ? 0 : a
            int sel;  // 0: zero, 1: a, 2: b
            if (!cmsb && !clsb) sel = (smsb | slsb) ? 2 : 0;
            else if (!cmsb && clsb) sel = smsb ? 2 : (slsb ? 0 : 1);
            else if (cmsb && !clsb) sel = smsb ? (slsb ? 2 : 0) : 1;
            else sel = (smsb & slsb) ? 0 : 1;
            // b = 2^128 - a: the 128-bit negative of a (a > 0, so the fifth limb is 0)
            const uint32_t n0 = 0u - a0, n1 = ~a1 + (a0 == 0), n2 = ~a2 + ((a0 | a1) == 0), n3 = ~a3 + ((a0 | a1 | a2) == 0);
            Fc part = fc_zero();
            part.v[0] = sel == 1 ? a0 : sel == 2 ? n0 : 0u;
            part.v[1] = sel == 1 ? a1 : sel == 2 ? n1 : 0u;
            part.v[2] = sel == 1 ? a2 : sel == 2 ? n2 : 0u;
            part.v[3] = sel == 1 ? a3 : sel == 2 ? n3 : 0u;
            io.put_c(off.parts + i, part);
            uint64_t c = (uint64_t)s0 + part.v[0];
            s0 = (uint32_t)c; c >>= 32;
            c += (uint64_t)s1 + part.v[1]; s1 = (uint32_t)c; c >>= 32;
            c += (uint64_t)s2 + part.v[2]; s2 = (uint32_t)c; c >>= 32;
            c += (uint64_t)s3 + part.v[3]; s3 = (uint32_t)c; c >>= 32;
            s4 += (uint32_t)c;
            a3 = (a3 << 1) | (a2 >> 31); a2 = (a2 << 1) | (a1 >> 31); a1 = (a1 << 1) | (a0 >> 31); a0 <<= 1;
        }
    };
    word(bits.v[0], ct[0], 0, 16); word(bits.v[1], ct[1], 1, 16); word(bits.v[2], ct[2], 2, 16); word(bits.v[3], ct[3], 3, 16);
    word(bits.v[4], ct[4], 4, 16); word(bits.v[5], ct[5], 5, 16); word(bits.v[6], ct[6], 6, 16); word(bits.v[7], ct[7], 7, 15);
    put_word_bits_plain(io, off.bits, s0, 32);
    put_word_bits_plain(io, off.bits + 32, s1, 32);
    put_word_bits_plain(io, off.bits + 64, s2, 32);
    put_word_bits_plain(io, off.bits + 96, s3, 32);
    put_word_bits_plain(io, off.bits + 128, s4, 7);
    return (s3 >> 31) & 1u;
}

__constant__ const uint32_t CT_MINUS1_D[8] = {0xf0000000u, 0x43e1f593u, 0x79b97091u, 0x2833e848u, 0x8181585du, 0xb85045b6u, 0xe131a029u, 0x30644e72u};
__constant__ const uint32_t CT_HALF_D[8] = {0xf8000000u, 0xa1f0fac9u, 0x3cdcb848u, 0x9419f424u, 0x40c0ac2eu, 0xdc2822dbu, 0x7098d014u, 0x18322739u};
__constant__ const uint32_t CT_SUBORDER_M1_D[8] = {0x392126f0u, 0x677297dcu, 0x3920ee0au, 0xab3eedb8u, 0xd0302b0bu, 0x370a08b6u, 0x5c263405u, 0x060c89ceu};

// Num2Bits_strict: bits + AliasCheck
__device__ __forceinline__ void num2bits_strict_dev(const UnitIO& io, const N2BStrictOff& off, const Fc& canon, int cid_alias) {
    put_bits(io, off.bits, canon, 254);
    const uint32_t o = comp_constant_dev(io, off.cc, canon, CT_MINUS1_D);
    if (o && !io.mute) report_fail(io.err, io.inst, io.err_unit, (uint32_t)cid_alias, fr_one(), fr_zero());
}

// DecodeFloatBin (reference src/lib/decode-float.circom:12-44) on 40 bits given as u64
__device__ __forceinline__ Fr decode_float_dev(const UnitIO& io, const DecodeFloatOff& o, uint64_t f40) {
    // pe[0] = 9*e0 + 1 ; pe[i] = (pe[i-1]*10^(2^i) - pe[i-1])*e[i] + pe[i-1]
    Fr pe = fr_from_u64(((f40 >> 35) & 1) ? 10 : 1);
    Fr p10 = fr_from_u64(10);
    for (int i = 1; i < 5; i++) {
        p10 = fr_sqr(p10);
        if ((f40 >> (35 + i)) & 1) pe = fr_mul(pe, p10);
        io.put_m(o.pe + (i - 1), pe);
    }
    const Fr out = fr_mul(fr_from_u64(f40 & ((1ull << 35) - 1)), pe);
    io.put_m(o.out, out);
    return out;
}

}  // namespace hz
