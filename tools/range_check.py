#!/usr/bin/env python3
"""Worst-case magnitudes through the lazily reduced field arithmetic of the device code.

circuits_amd/csrc/fr.h keeps values in [0, 2p) only where it has to: products take operands below 2^257, sums that only feed
products are not reduced, the lanes beside Poseidon's S-box lane are trimmed every third pair of partial rounds. Every routine
states the range of its operands and of its result in a comment; random operands never reach the worst case of a 28-pair chain
with the real constants. This script PROVES the bounds by interval arithmetic on exact integers:

  * every fr.h routine is modelled by its precondition (asserted) and the tightest upper bound its definition gives -- a Montgomery
    reduction of column sums worth T returns (T + M p) / R with M < R, i.e. less than T / R + p; a constant operand enters with its
    exact value (the real constant blocks of tools/gen_constants.py, both the digest-only and the witness form);
  * 64-bit accumulator safety of the column sums is checked from the limb bounds (limbs 0..7 below 2^29, the top limb from the value);
  * the callers are transcribed statement by statement: poseidon_hash<T> for T = 2..7 (poseidon.h), the ladder step of the throughput
    signature kernel seg_lds_steps<G> and of seg_any_lock<G>, the fixed-base window step seg_fix_lock<G>, batch_inv / fr_inv, and the
    projective doubling chain of k_eddsa_pre (ed_dbl_proj) iterated to its fixed point (eddsa_kernels.hip).

The transcription is tied to the source: every modelled statement quotes the source text it stands for, and `check_sources()` fails
when that text is no longer in the file. Run: python tools/range_check.py (also run by tests/test_range_check.py in the CPU suite).
"""
import os
import sys
from fractions import Fraction

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))

P = 21888242871839275222246405745257275088548364400416034343698204186575808495617
R = 1 << 261
LIMIT = 1 << 257          # operands of a product (fr.h fr_mul: "inputs normalised with value < 2^257")
QUOTES = []               # (file, source text) pairs the model stands for


class RangeError(AssertionError):
    pass


def need(cond, what):
    if not cond:
        raise RangeError(what)


def src(path, text):
    QUOTES.append((path, text))


class V:
    """an integer in [0, ub) held in normalised 29-bit limbs (limbs 0..7 < 2^29, limb 8 = the rest); `exact` for constants"""

    def __init__(self, ub, exact=None, name=""):
        self.ub = int(ub) if exact is None else exact + 1
        self.exact = exact
        self.name = name

    def top(self):   # bound of limb 8
        return ((self.ub - 1) >> 232) + 1

    def hi(self):    # the value a product sees: exact for constants, ub - 1 otherwise
        return self.exact if self.exact is not None else self.ub - 1

    def __repr__(self):
        return "<%s %.3f p>" % (self.name, self.ub / P)


def const(x, name="c"):
    return V(0, exact=x, name=name)


def ceil_frac(f):
    return -((-f.numerator) // f.denominator)


def _col_check(pairs, addend_limb=0, what=""):
    """64-bit accumulator of fr_reduce_cols and of the column sums: products a_i * b_j of every pair, 9 reduction terms m_i * p_j,
    the carry of the previous column and one addend limb"""
    worst = 0
    for k in range(17):
        s = 0
        for a, b in pairs:
            for i in range(9):
                j = k - i
                if 0 <= j <= 8:
                    la = a.top() if i == 8 else (1 << 29)
                    lb = b.top() if j == 8 else (1 << 29)
                    s += (la - 1) * (lb - 1)
        worst = max(worst, s)
    total = worst + 9 * ((1 << 29) - 1) ** 2 + (1 << 36) + addend_limb
    need(total < (1 << 64), "%s: column accumulator may reach 2^%.2f" % (what, total.bit_length()))


def mont(pairs, addend_low=None, addend_high=None, what="product"):
    """(sum a*b + addend_low + addend_high * R) / R reduced once: fr_mul / fr_sqr / fr_dot<N> / fr_muladd / fr_muladd2"""
    for a, b in pairs:
        need(a.ub <= LIMIT and b.ub <= LIMIT, "%s: operand %r or %r not below 2^257" % (what, a, b))
    need(len(pairs) <= 6, "%s: more than 6 products per reduction" % what)
    lim = (1 << 29) if addend_low is None and addend_high is None else max((addend_low.top() if addend_low else 0), (addend_high.top() if addend_high else 0), 1 << 29)
    _col_check(pairs, lim, what)
    t = sum(a.hi() * b.hi() for a, b in pairs)
    if addend_low is not None:
        t += addend_low.hi()
    if addend_high is not None:
        t += addend_high.hi() * R
    # (T + M p) / R with M <= R - 1
    ub = ceil_frac(Fraction(t + (R - 1) * P, R)) + 1
    need(ub < (1 << 262), "%s: result does not fit the limbs" % what)
    return V(ub, name=what)


src("circuits_amd/csrc/fr.h", "HZ_HD_HEAVY Fr fr_mul(HZ_HEAVY_ARG(Fr) a, HZ_HEAVY_ARG(Fr) b) {")
def fr_mul(a, b, what="fr_mul"): return mont([(a, b)], what=what)                                    # noqa: E704
def fr_sqr(a, what="fr_sqr"): return mont([(a, a)], what=what)                                       # noqa: E704
src("circuits_amd/csrc/fr.h", "HZ_HD Fr fr_dot(const Fr* a, const Fr* b, const Fr* addend = nullptr) {")
def fr_dot(a, b, addend=None, what="fr_dot"): return mont(list(zip(a, b)), addend_low=addend, what=what)   # noqa: E704


src("circuits_amd/csrc/fr.h", "// normalised with b < 2^257, s < 8p; the result is normalised and < s + 1.01 p.")
def fr_muladd(a, b, s, what="fr_muladd"):                                                            # noqa: E302
    need(s.ub <= 8 * P, "%s: addend %r not below 8p" % (what, s))
    need(a.ub <= P, "%s: constant operand %r not below p" % (what, a))
    return mont([(a, b)], addend_high=s, what=what)


src("circuits_amd/csrc/fr.h", "// (a0*b0 + a1*b1 + s*R) / R with one reduction; a0, a1 < p, b0, b1 normalised < 2^257, s < 8p.")
def fr_muladd2(a0, b0, a1, b1, s, what="fr_muladd2"):                                                # noqa: E302
    need(s.ub <= 8 * P, "%s: addend %r not below 8p" % (what, s))
    need(a0.ub <= P and a1.ub <= P, "%s: constant operands not below p" % what)
    return mont([(a0, b0), (a1, b1)], addend_high=s, what=what)


def cond_sub(x, k, what):
    need(x.ub <= 2 * k * P, "%s: operand %r not below %dp" % (what, x, 2 * k))
    return V(k * P, name=what)


src("circuits_amd/csrc/fr.h", "// a (normalised, value < 8p) -> a - 4p if a >= 4p; branch-free (selects), by value")
def fr_cond_sub_4p(x): return cond_sub(x, 4, "fr_cond_sub_4p")                                       # noqa: E704
src("circuits_amd/csrc/fr.h", "// t (normalised, value < 4p) -> t - 2p if t >= 2p")
def fr_cond_sub_2p(x): return cond_sub(x, 2, "fr_cond_sub_2p")                                       # noqa: E704
src("circuits_amd/csrc/fr.h", "// t (normalised limbs, value < 2p) -> t - p if t >= p: the unique representative in [0, p)")
def fr_cond_sub_p(x): return cond_sub(x, 1, "fr_cond_sub_p")                                         # noqa: E704
fr_cond_sub_p_rare = fr_cond_sub_p   # the wave-uniform skip returns the operand only when no lane can be >= p (top limb below p's)


def fr_add(a, b):
    need(a.ub + b.ub - 1 <= 4 * P, "fr_add: %r + %r not below 4p" % (a, b))
    return V(2 * P, name="fr_add")


def fr_sub(a, b):
    need(a.ub <= 2 * P and b.ub <= 2 * P, "fr_sub: %r - %r operands not below 2p" % (a, b))
    return V(2 * P, name="fr_sub")


def fr_dbl(a): return fr_add(a, a)                                                                   # noqa: E704


src("circuits_amd/csrc/fr.h", "// a - b + 2p in (0, A + 2p) for a in [0, A), b in [0, 2p): not reduced")
def fr_sub_lazy(a, b):                                                                               # noqa: E302
    need(b.ub <= 2 * P, "fr_sub_lazy: subtrahend %r not below 2p" % b)
    return V(a.ub + 2 * P, name="fr_sub_lazy")


src("circuits_amd/csrc/fr.h", "// 2a in [0, 2A) for a in [0, A): not reduced (for a below p this IS fr_dbl(a))")
def fr_dbl_lazy(a): return V(2 * a.ub, name="fr_dbl_lazy")                                           # noqa: E704
src("circuits_amd/csrc/fr.h", "// a + b, not reduced: below A + B")
def fr_add_lazy(a, b): return V(a.ub + b.ub, name="fr_add_lazy")                                     # noqa: E704


src("circuits_amd/csrc/fr.h", "// a - b - c + 4p in (0, A + 4p) for a in [0, A), b + c < 4p: not reduced")
def fr_sub2_lazy(a, b, c):                                                                           # noqa: E302
    need(b.ub + c.ub - 1 <= 4 * P, "fr_sub2_lazy: %r + %r not below 4p" % (b, c))
    return V(a.ub + 4 * P, name="fr_sub2_lazy")


src("circuits_amd/csrc/fr.h", "// m - a - b - c reduced to [0, 2p); m in [0, 2p), a, b, c in [0, 2p) with a + b + c < 4p")
def fr_sub3(m, a, b, c):                                                                             # noqa: E302
    need(m.ub <= 2 * P and max(a.ub, b.ub, c.ub) <= 2 * P and a.ub + b.ub + c.ub - 2 <= 4 * P, "fr_sub3: operands %r %r %r %r" % (m, a, b, c))
    return V(2 * P, name="fr_sub3")


src("circuits_amd/csrc/fr.h", "// s - a - 2x reduced to [0, 2p); s, a, x in [0, 2p)")
def fr_sub_a_2x(s, a, x):                                                                            # noqa: E302
    need(max(s.ub, a.ub, x.ub) <= 2 * P, "fr_sub_a_2x: operands %r %r %r" % (s, a, x))
    return V(2 * P, name="fr_sub_a_2x")


src("circuits_amd/csrc/fr.h", "// 3a + b + c, not reduced: below 3A + B + C for a in [0, A), b in [0, B), c in [0, C)")
def fr_3a_b_c_lazy(a, b, c):                                                                         # noqa: E302
    need(3 * a.top() + b.top() + c.top() < (1 << 32), "fr_3a_b_c_lazy: top limb overflows")
    return V(3 * a.ub + b.ub + c.ub, name="fr_3a_b_c_lazy")


src("circuits_amd/csrc/fr.h", "    return fr_cond_sub_p_rare(fr_reduce_cols(t));   // (a + m p)/R <= p")
def fr_canon_limbs(a):                                                                               # noqa: E302
    need(a.ub <= R, "fr_canon_limbs: %r not below R" % a)      # (a + M p) / R <= p needs a <= R
    return V(P, name="fr_canon_limbs")


def fr_is_zero(a):
    need(a.ub <= 2 * P, "fr_is_zero: %r not below 2p (only 0 and p are recognised)" % a)


src("circuits_amd/csrc/fr.h", "// value of a (< 2p, normalised) reduced to [0, p) and re-cut into 30-bit limbs")
def fr_inv(a):                                                                                       # noqa: E302
    need(a.ub <= 2 * P, "fr_inv: %r not below 2p" % a)
    return fr_mul(V(P, name="inverse limbs"), const(pow(R, 3, P), "R^3"), what="fr_inv")


def batch_inv(xs):
    src("circuits_amd/csrc/gadgets_dev.h", "        if (!fr_is_zero(x[i])) acc = fr_mul(acc, x[i]);")
    acc = const(R % P, "one")
    pre = []
    for x in xs:
        pre.append(acc)
        fr_is_zero(x)
        acc = fr_mul(acc, x, "batch_inv prefix")
    inv = fr_inv(acc)
    out = [None] * len(xs)
    for i in range(len(xs) - 1, -1, -1):
        out[i] = fr_mul(inv, pre[i], "batch_inv peel")
        inv = fr_mul(inv, xs[i], "batch_inv carry")
    return out


# ---- Poseidon (poseidon.h) -------------------------------------------------------------------------------------------------------
def poseidon_sbox(x, witness):
    if witness:
        src("circuits_amd/csrc/poseidon.h", "        const Fr in4 = fr_cond_sub_p_rare(fr_mul(x2, in2));")
        src("circuits_amd/csrc/poseidon.h", "        const Fr out = fr_cond_sub_p_rare(fr_mul(in4, x));")
        x2 = fr_sqr(x, "sbox x2")
        in2 = fr_canon_limbs(x2)
        in4 = fr_cond_sub_p_rare(fr_mul(x2, in2, "sbox in4"))
        return fr_cond_sub_p_rare(fr_mul(in4, x, "sbox out"))
    src("circuits_amd/csrc/poseidon.h", "        const Fr x5 = fr_mul(x4, x);")
    x2 = fr_sqr(x)
    return fr_mul(fr_sqr(x2), x, "sbox x5")


def poseidon_row(row, st, addend):
    src("circuits_amd/csrc/poseidon.h", "        return fr_add(fr_dot<4>(row, st, addend), fr_dot<N - 4>(row + 4, st + 4));   // N <= 8")
    n = len(st)
    if n <= 6:
        return fr_dot(row[:n], st, addend, "poseidon_row<%d>" % n)
    return fr_add(fr_dot(row[:4], st[:4], addend, "poseidon_row lo"), fr_dot(row[4:n], st[4:], None, "poseidon_row hi"))


def poseidon_hash(T, witness, in_ub=2 * P, report=None, trim_every=3):
    from gen_constants import poseidon_block
    from poseidon_params import N_ROUNDS_P
    RP = N_ROUNDS_P[T - 2]
    K = [const(x, "K[%d]" % i) for i, x in enumerate(poseidon_block(T, witness))]
    e0 = 4 * T
    part = 4 * T + 1
    dense = part + (RP // 2) * (4 * T + 1) + (RP % 2) * 2 * T
    cf = dense + (T - 1) * (T - 1)
    tail = cf + (T - 1)
    mo = tail + 3 * T
    need(len(K) == mo + T * T, "constant block layout")
    M = K[mo:]
    src("circuits_amd/csrc/poseidon.h", "    for (int j = 1; j < T; j++) st[j] = fr_add(in[j - 1], K[j]);")
    st = [K[0]] + [fr_add(V(in_ub, name="input"), K[j]) for j in range(1, T)]

    def sbox_layer(st):
        return [poseidon_sbox(x, witness) for x in st]

    def mix_ark(st, Cn, nc):
        return [poseidon_row(M[i * T:(i + 1) * T], st, Cn[i] if i < nc else None) for i in range(T)]
    for r in range(3):
        st = mix_ark(sbox_layer(st), K[T * (r + 1):T * (r + 2)], T)
    st = mix_ark(sbox_layer(st), K[e0:e0 + 1], 1)
    src("circuits_amd/csrc/poseidon.h", "        for (int j = 1; j < T; j++) st[j] = fr_muladd2(CA[j - 1], v[T], CA[T - 1 + j - 1], v[0], st[j]);")
    src("circuits_amd/csrc/poseidon.h", "        if (++lazy == 3) {")
    S = part
    lazy = 0
    worst_lane = 0
    for r in range(0, RP - 1, 2):
        v = [poseidon_sbox(st[0], witness)] + st[1:]
        sa = poseidon_row(K[S:S + T], v, K[S + T])
        v.append(v[0])
        v[0] = poseidon_sbox(sa, witness)
        st0 = poseidon_row(K[S + T + 1:S + 2 * T + 2], v, K[S + 2 * T + 2])
        CA = K[S + 2 * T + 3:]
        lanes = [fr_muladd2(CA[j - 1], v[T], CA[T - 1 + j - 1], v[0], st[j], "partial-round lane") for j in range(1, T)]
        st = [st0] + lanes
        worst_lane = max([worst_lane] + [x.ub for x in lanes])
        lazy += 1
        if lazy == trim_every:
            lazy = 0
            st = [st[0]] + [fr_cond_sub_4p(x) for x in st[1:]]
        S += 4 * T + 1
    if RP % 2:
        src("circuits_amd/csrc/poseidon.h", "        for (int j = 1; j < T; j++) st[j] = fr_muladd(S[T + j - 1], st[0], st[j]);")
        y = poseidon_sbox(st[0], witness)
        st = [y] + st[1:]
        s0 = poseidon_row(K[S:S + T], st, K[S + 2 * T - 1])
        st = [s0] + [fr_muladd(K[S + T + j - 1], st[0], st[j], "odd partial-round lane") for j in range(1, T)]
    src("circuits_amd/csrc/poseidon.h", "poseidon_row<T - 1>(K + poseidon_k_dense<T>() + i * (T - 1), st + 1, K + poseidon_k_cf<T>() + i));")
    st = [st[0]] + [poseidon_row(K[dense + i * (T - 1):dense + (i + 1) * (T - 1)], st[1:], K[cf + i]) for i in range(T - 1)]
    for r in range(3):
        st = mix_ark(sbox_layer(st), K[tail + T * r:tail + T * (r + 1)], T)
    st = sbox_layer(st)
    out = poseidon_row(M[:T], st, None)
    if report is not None:
        report.append("poseidon_hash<%d> %s: lanes beside the S-box lane peak at %.3f p (addend limit 8 p), digest below %.3f p" % (
            T, "witness" if witness else "digest ", worst_lane / P, out.ub / P))
    need(out.ub <= 2 * P, "poseidon_hash<%d>: digest %r not below 2p" % (T, out))
    return out


# ---- signature ladders (eddsa_kernels.hip) -----------------------------------------------------------------------------------------
A_SMALL = 168698


def seg_any_step(G, report=None):
    """one step of seg_lds_steps<G> on the state a previous step leaves: DX1 < 2p (scale 1), DY0 / AX / AY canonical"""
    f = "circuits_amd/csrc/eddsa_kernels.hip"
    one0, A0 = const(1, "one0"), const(A_SMALL, "A0")
    KA = V(2 * P, name="K.A")          # fr_from_u64: a product's output
    A2 = fr_dbl(KA)
    st = [dict(DX1=V(2 * P, name="DX1"), DY0=V(P, name="DY0"), AX=V(P, name="AX"), AY=V(P, name="AY")) for _ in range(G)]
    acc = const(R % P, "one")
    pre = []
    for g in range(G):
        s = st[g]
        dx0 = fr_canon_limbs(s["DX1"])
        src(f, "            const Fr a_den = fr_sub(L.get(s0 + LS_AX), dx0);")
        a_den = fr_sub(s["AX"], dx0)
        pre.append(acc)
        fr_is_zero(a_den)
        src(f, "            if (!fr_is_zero(a_den)) acc = fr_mul(acc, a_den);")
        acc = fr_mul(acc, a_den, "prefix product")
        pre.append(acc)
        src(f, "                const Fr nx1_2 = fr_mul(dx0, L.get(s0 + LS_DX1));")
        nx1_2 = fr_mul(dx0, s["DX1"], "x1_2")
        fr_cond_sub_p(nx1_2)   # ed_put0
        src(f, "                L.put(s0 + LS_DNUM, fr_3a_b_c_lazy(nx1_2, fr_mul(A2, dx0), one0));")
        s["DNUM"] = fr_3a_b_c_lazy(nx1_2, fr_mul(A2, dx0, "2A x"), one0)
        src(f, "                const Fr dd = fr_dbl_lazy(L.get(s0 + LS_DY0));   // y is stored canonical: 2y < 2p as it is")
        dd = fr_dbl_lazy(s["DY0"])
        fr_is_zero(dd)
        acc = fr_mul(acc, dd, "prefix product")
    src(f, "        Fr inv = fr_inv(acc);   // scale-0 divisors in, scale-2 inverses out")
    inv = fr_inv(acc)
    worst = 0
    for g in range(G - 1, -1, -1):
        s = st[g]
        dx0, dy0 = fr_canon_limbs(s["DX1"]), s["DY0"]
        dd = fr_dbl_lazy(dy0)
        src(f, "                    inv_dd = fr_mul(inv, L.get(ls_pre0<G>() + 2 * g));")
        inv_dd = fr_mul(inv, pre[2 * g + 1], "inv_dd")
        inv = fr_mul(inv, dd, "inverse carry")
        a_den = fr_sub(s["AX"], dx0)
        inv_a = fr_mul(inv, pre[2 * g], "inv_a") if g > 0 else inv
        if g > 0:
            inv = fr_mul(inv, a_den, "inverse carry")
        src(f, "            const Fr a_num = fr_sub_lazy(addIn.y, dy0);   // in (p, 3p): a multiplicand only")
        a_num = fr_sub_lazy(s["AY"], dy0)
        a_l1 = fr_mul(a_num, inv_a, "adder lamda")
        a_l0 = fr_canon_limbs(a_l1)
        src(f, "fr_sub3(fr_mul(a_l0, a_l1), A0, dx0, addIn.x));   // one reduction for the three differences")
        aox = fr_cond_sub_p(fr_sub3(fr_mul(a_l0, a_l1, "lamda^2"), A0, dx0, s["AX"]))
        src(f, "            ao.y = fr_sub(fr_mul(a_l1, fr_sub_lazy(dx0, ao.x)), dy0);")
        aoy = fr_cond_sub_p(fr_sub(fr_mul(a_l1, fr_sub_lazy(dx0, aox), "lamda (x - x')"), dy0))
        src(f, "                const Fr l1 = fr_mul(d_num, inv_dd);")
        l1 = fr_mul(s["DNUM"], inv_dd, "doubler lamda")
        worst = max(worst, s["DNUM"].ub, a_num.ub)
        fr_canon_limbs(l1)
        src(f, "                const Fr nx1 = fr_sub_a_2x(fr_sqr(l1), K.A, L.get(s0 + LS_DX1));   // l^2 - A - 2x in [0, 2p), one reduction")
        nx1 = fr_sub_a_2x(fr_sqr(l1), KA, s["DX1"])
        nx0 = fr_canon_limbs(nx1)
        src(f, "                const Fr ny0 = fr_sub(fr_mul(l1, fr_sub_lazy(dx0, nx0)), dy0);")
        ny0 = fr_cond_sub_p(fr_sub(fr_mul(l1, fr_sub_lazy(dx0, nx0), "lamda (x - x')"), dy0))
        new = dict(DX1=nx1, DY0=ny0, AX=aox, AY=aoy)
        for k, b in dict(DX1=2 * P, DY0=P, AX=P, AY=P).items():   # the step reproduces the state it assumed
            need(new[k].ub <= b, "seg_lds_steps<%d>: %s leaves the step as %r" % (G, k, new[k]))
    if report is not None:
        report.append("seg_lds_steps<%d>: largest unreduced multiplicand %.3f p (limit 2^257 = %.1f p)" % (G, worst / P, LIMIT / P))


def seg_fix_step(G, report=None):
    f = "circuits_amd/csrc/eddsa_kernels.hip"
    A0 = const(A_SMALL, "A0")
    accx, accy = V(P, name="acc.x"), V(P, name="acc.y")
    mx, my = V(P, name="table x"), V(P, name="table y")
    src(f, "            inv[g] = fr_sub(mx, acc[g].x);")
    inv = batch_inv([fr_sub(mx, accx) for _ in range(G)])
    for g in range(G):
        src(f, "            const Fr num = fr_sub_lazy(mo.y, acc[g].y);   // a multiplicand only (fr.h \"lazily reduced sums\")")
        num = fr_sub_lazy(my, accy)
        l1 = fr_mul(num, inv[g], "window lamda")
        l0 = fr_canon_limbs(l1)
        src(f, "            ao.x = ed_put0(w, wb + WIN_ADD_OUT0, fr_sub3(fr_mul(l0, l1), A0, acc[g].x, mo.x));")
        aox = fr_cond_sub_p(fr_sub3(fr_mul(l0, l1, "lamda^2"), A0, accx, mx))
        src(f, "            ao.y = fr_sub(fr_mul(l1, fr_sub_lazy(acc[g].x, ao.x)), acc[g].y);")
        aoy = fr_cond_sub_p(fr_sub(fr_mul(l1, fr_sub_lazy(accx, aox), "lamda (x - x')"), accy))
        need(aox.ub <= P and aoy.ub <= P, "seg_fix_lock: accumulator leaves the window as %r %r" % (aox, aoy))
    if report is not None:
        report.append("seg_fix_lock<%d>: window step closed on canonical accumulators" % G)


def ed_dbl_proj_chain(count=147, report=None):
    """k_eddsa_pre's inversion-free doubling chain: none of its sums is reduced; the coordinates must reach a fixed point below 2^257"""
    f = "circuits_amd/csrc/eddsa_kernels.hip"
    src(f, "    const Fr A = fr_sqr(p.X), B = fr_sqr(p.Y), C = fr_dbl_lazy(fr_sqr(p.Z)), D = fr_mul(K.a, A);")
    src(f, "    const Fr E = fr_sub2_lazy(fr_sqr(fr_add_lazy(p.X, p.Y)), A, B), G = fr_add_lazy(D, B);")
    src(f, "    const Fr F = fr_sub2_lazy(G, C, zero), H = fr_sub_lazy(D, B);")
    src(f, "    r.X = fr_mul(E, F); r.Y = fr_mul(G, H); r.Z = fr_mul(F, G);")
    Ka = V(2 * P, name="K.a")
    zero = const(0, "zero")
    X, Y, Z = V(2 * P, name="X"), V(2 * P, name="Y"), V(2 * P, name="Z")
    peak = 0
    for i in range(count):
        A, B, C, D = fr_sqr(X), fr_sqr(Y), fr_dbl_lazy(fr_sqr(Z)), fr_mul(Ka, fr_sqr(X))
        E = fr_sub2_lazy(fr_sqr(fr_add_lazy(X, Y)), A, B)
        G = fr_add_lazy(D, B)
        F = fr_sub2_lazy(G, C, zero)
        H = fr_sub_lazy(D, B)
        peak = max(peak, E.ub, F.ub, G.ub, H.ub)
        X, Y, Z = fr_mul(E, F, "X3"), fr_mul(G, H, "Y3"), fr_mul(F, G, "Z3")
        need(max(X.ub, Y.ub, Z.ub) <= 2 * P, "ed_dbl_proj: coordinates grow to %r %r %r at step %d" % (X, Y, Z, i))
    if report is not None:
        report.append("ed_dbl_proj x %d: largest unreduced multiplicand %.3f p, coordinates stay below %.3f p" % (count, peak / P, max(X.ub, Y.ub, Z.ub) / P))
    # what follows the chain: (Z - Y), X through batch_inv, products with Z + Y
    src(f, "    Fr den[2] = {fr_sub(q.Z, q.Y), q.X};")
    inv = batch_inv([fr_sub(Z, Y), X])
    fr_mul(fr_mul(fr_mul(fr_add(Z, Y), inv[0]), Z), inv[1])


def check_sources():
    """every quoted statement is still in the source it was transcribed from"""
    missing = []
    cache = {}
    for path, text in QUOTES:
        if path not in cache:
            cache[path] = open(os.path.join(ROOT, path)).read()
        if text not in cache[path]:
            missing.append((path, text))
    return missing


def run(report=None):
    for T in range(2, 8):
        for witness in (False, True):
            poseidon_hash(T, witness, report=report)
    for G in (1, 2, 3, 4):
        seg_any_step(G, report)
    for G in (1, 8):
        seg_fix_step(G, report)
    ed_dbl_proj_chain(147, report)
    return check_sources()


if __name__ == "__main__":
    rep = []
    missing = run(rep)
    print("\n".join(rep))
    for path, text in missing:
        print("STALE TRANSCRIPTION: %s no longer contains: %s" % (path, text))
    print("range check:", "FAILED (source drifted)" if missing else "all bounds hold")
    sys.exit(1 if missing else 0)
