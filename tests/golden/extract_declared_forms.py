#!/usr/bin/env python3
"""Records, from the reference's own circuit sources, HOW every signal that is not a product or a hint is wired: a symbolic run of
the templates under /root/reference/src (a small interpreter for the circom subset they use: templates, functions, var, for, if,
signal / component arrays, <== ==> <-- --> ===) in which every signal is its own symbol, as in an unreduced compile. Output:
tests/golden/declared_forms.json.gz -- per main template and shape

    forms   {signal name: [constant, [[coefficient, signal name], ...]]}   (field elements as decimal strings, those above r / 2 as negative
            numbers) every `x <== linear expression` of the sources, one level at a time (the right-hand names are signals again: of the same template, inputs of a sub-component, outputs of one)
    quads   [[A, B, C], ...]   every other constraint of the sources as A * B = C with A, B, C linear forms as above: the `x <== a * b + c`
            lines and the `===` lines (A = B = 0 for a linear `===`)
    bases   [signal name, ...]   signals defined by a product of signals or a `<--` hint, inputs of the main component, outputs of
            circomlib components: values a witness has to hold (or, for a few circomlib outputs, derive: see MODELS)
    declared [signal name, ...]  every signal the reference's templates declare below this main, all indices

-- data, no source text. The sources of circomlib 0.5.2 are not in the reference repository; its templates are black boxes here:
their inputs receive forms from the reference's wiring, their outputs are symbols. The few whose outputs are LINEAR in their inputs
or wrap another component (Bits2Num, LessThan and its family, IsEqual, ForceEqualIfEnabled, NOT, Switcher, Mux1..Mux4) are stated in
`Run.model` from circomlib's published sources, and so are the constraints of its small gadgets the reference's logic is made of:
Num2Bits (bits are bits, their sum is the input), IsZero, ForceEqualIfEnabled, MultiMux1..4 (every product term). Poseidon, SHA-256,
the sparse Merkle tree and the EdDSA verifier stay opaque (they are pinned by upstream known answers).

tests/test_declared_signals.py imports these names as a .sym, asks the library for every value and compares it with the form
evaluated on the oracle's witness.

    python tests/golden/extract_declared_forms.py [/root/reference]        (run in the build container; the reference does not travel)
"""
import gzip
import json
import os
import re
import sys

P = 21888242871839275222246405745257275088548364400416034343698204186575808495617

TOKEN = re.compile(r"\s*(?:(0x[0-9a-fA-F]+|\d+)|([A-Za-z_$][\w$]*)|(<==|==>|<--|-->|===|\*\*|<<|>>|<=|>=|==|!=|&&|\|\||\+\+|--|\+=|-=|\*=|/=|[-+*/\\%<>=!&|^~?:;,.(){}\[\]]))")


def tokenize(src):
    src = re.sub(r"/\*.*?\*/", " ", src, flags=re.S)
    src = re.sub(r"//[^\n]*", " ", src)
    src = re.sub(r'include\s+"[^"]*"\s*;?', " ", src)
    out, i = [], 0
    while i < len(src):
        m = TOKEN.match(src, i)
        if not m:
            if src[i:].strip() == "":
                break
            raise SyntaxError("cannot tokenize at %r" % src[i:i + 40])
        i = m.end()
        if m.group(1) is not None:
            out.append(("num", int(m.group(1), 0)))
        elif m.group(2) is not None:
            out.append(("id", m.group(2)))
        else:
            out.append(("op", m.group(3)))
    return out


# ---- parser: statements and expressions as nested tuples -------------------------------------------------------------------------
class Parser:
    def __init__(self, toks):
        self.t, self.i = toks, 0

    def peek(self, k=0):
        return self.t[self.i + k] if self.i + k < len(self.t) else ("eof", None)

    def next(self):
        tok = self.peek()
        self.i += 1
        return tok

    def accept(self, kind, val=None):
        tok = self.peek()
        if tok[0] == kind and (val is None or tok[1] == val):
            self.i += 1
            return tok
        return None

    def expect(self, kind, val=None):
        tok = self.accept(kind, val)
        if tok is None:
            raise SyntaxError("expected %s %r, found %r (token %d)" % (kind, val, self.peek(), self.i))
        return tok

    def program(self):
        defs = {}
        while self.peek()[0] != "eof":
            kw = self.expect("id")[1]
            if kw in ("template", "function"):
                name = self.expect("id")[1]
                self.expect("op", "(")
                params = []
                while not self.accept("op", ")"):
                    params.append(self.expect("id")[1])
                    self.accept("op", ",")
                defs[name] = (kw, params, self.block())
            elif kw == "component":   # component main = ...;
                while not self.accept("op", ";"):
                    self.next()
            else:
                raise SyntaxError("top level: %r" % kw)
        return defs

    def block(self):
        self.expect("op", "{")
        out = []
        while not self.accept("op", "}"):
            out.append(self.statement())
        return ("block", out)

    def body(self):
        return self.block() if self.peek() == ("op", "{") else self.statement()

    def dims(self):
        d = []
        while self.accept("op", "["):
            d.append(self.expr())
            self.expect("op", "]")
        return d

    def statement(self):
        tok = self.peek()
        if tok == ("op", "{"):
            return self.block()
        if tok[0] == "id" and tok[1] == "signal":
            self.next()
            kind = "intermediate"
            if self.accept("id", "private"):
                pass
            if self.accept("id", "input"):
                kind = "input"
            elif self.accept("id", "output"):
                kind = "output"
            decls = []
            while True:
                decls.append((self.expect("id")[1], self.dims()))
                if not self.accept("op", ","):
                    break
            self.expect("op", ";")
            return ("signal", kind, decls)
        if tok[0] == "id" and tok[1] == "component":
            self.next()
            name = self.expect("id")[1]
            d = self.dims()
            init = None
            if self.accept("op", "="):
                init = self.expr()
            self.expect("op", ";")
            return ("component", name, d, init)
        if tok[0] == "id" and tok[1] == "var":
            self.next()
            decls = []
            while True:
                name = self.expect("id")[1]
                d = self.dims()
                init = self.expr() if self.accept("op", "=") else None
                decls.append((name, d, init))
                if not self.accept("op", ","):
                    break
            self.expect("op", ";")
            return ("var", decls)
        if tok[0] == "id" and tok[1] == "for":
            self.next()
            self.expect("op", "(")
            init = self.statement()          # consumes its ';'
            cond = self.expr()
            self.expect("op", ";")
            step = self.simple()
            self.expect("op", ")")
            return ("for", init, cond, step, self.body())
        if tok[0] == "id" and tok[1] == "while":
            self.next()
            self.expect("op", "(")
            cond = self.expr()
            self.expect("op", ")")
            return ("for", ("block", []), cond, ("block", []), self.body())
        if tok[0] == "id" and tok[1] == "if":
            self.next()
            self.expect("op", "(")
            cond = self.expr()
            self.expect("op", ")")
            then = self.body()
            other = self.body() if self.accept("id", "else") else None
            return ("if", cond, then, other)
        if tok[0] == "id" and tok[1] == "return":
            self.next()
            e = self.expr()
            self.expect("op", ";")
            return ("return", e)
        s = self.simple()
        if self.peek() != ("op", "}"):   # (one statement of the sources ends its block without a semicolon)
            self.expect("op", ";")
        return s

    def simple(self):
        """assignment-like statement without its ';'"""
        lhs = self.expr()
        tok = self.peek()
        if tok[0] == "op" and tok[1] in ("=", "+=", "-=", "*=", "/=", "<==", "<--", "==>", "-->", "==="):
            self.next()
            rhs = self.expr()
            return ("assign", tok[1], lhs, rhs)
        if tok[0] == "op" and tok[1] in ("++", "--"):
            self.next()
            return ("assign", "+=" if tok[1] == "++" else "-=", lhs, ("num", 1))
        return ("expr", lhs)

    LEVELS = [["||"], ["&&"], ["|"], ["^"], ["&"], ["==", "!="], ["<", ">", "<=", ">="], ["<<", ">>"], ["+", "-"], ["*", "/", "\\", "%"]]

    def expr(self):
        c = self.binary(0)
        if self.accept("op", "?"):
            a = self.expr()
            self.expect("op", ":")
            b = self.expr()
            return ("cond", c, a, b)
        return c

    def binary(self, lv):
        if lv == len(self.LEVELS):
            return self.power()
        left = self.binary(lv + 1)
        while self.peek()[0] == "op" and self.peek()[1] in self.LEVELS[lv]:
            op = self.next()[1]
            left = ("bin", op, left, self.binary(lv + 1))
        return left

    def power(self):
        base = self.unary()
        if self.accept("op", "**"):
            return ("bin", "**", base, self.power())
        return base

    def unary(self):
        if self.accept("op", "-"):
            return ("neg", self.unary())
        if self.accept("op", "!"):
            return ("not", self.unary())
        if self.accept("op", "~"):
            return ("inv", self.unary())
        return self.postfix()

    def postfix(self):
        tok = self.next()
        if tok[0] == "num":
            node = ("num", tok[1])
        elif tok == ("op", "("):
            node = self.expr()
            self.expect("op", ")")
        elif tok == ("op", "["):
            items = []
            while not self.accept("op", "]"):
                items.append(self.expr())
                self.accept("op", ",")
            node = ("list", items)
        elif tok[0] == "id":
            node = ("name", tok[1])
            if self.accept("op", "("):
                args = []
                while not self.accept("op", ")"):
                    args.append(self.expr())
                    self.accept("op", ",")
                node = ("call", tok[1], args)
        else:
            raise SyntaxError("unexpected %r" % (tok,))
        while True:
            if self.accept("op", "["):
                node = ("index", node, self.expr())
                self.expect("op", "]")
            elif self.accept("op", "."):
                node = ("member", node, self.expect("id")[1])
            else:
                return node


# ---- symbolic values -----------------------------------------------------------------------------------------------------------------
class Lin:
    """constant + sum of coefficient * signal name, over the field"""
    __slots__ = ("c", "t")

    def __init__(self, c=0, t=None):
        self.c, self.t = c % P, t or {}

    @staticmethod
    def of(v):
        return v if isinstance(v, Lin) else Lin(int(v))

    def is_const(self):
        return not self.t

    def add(self, o, sign=1):
        t = dict(self.t)
        for k, v in o.t.items():
            nv = (t.get(k, 0) + sign * v) % P
            if nv:
                t[k] = nv
            else:
                t.pop(k, None)
        return Lin(self.c + sign * o.c, t)

    def scale(self, k):
        k %= P
        return Lin(self.c * k, {n: v * k % P for n, v in self.t.items() if v * k % P})


class Quad:
    """A*B + C with A, B, C linear: what one R1CS constraint can hold. a is None for anything else non-linear (shifts, masks and
    comparisons of signals: they only occur on the right of `<--`)."""
    __slots__ = ("a", "b", "c")

    def __init__(self, a=None, b=None, c=None):
        self.a, self.b, self.c = a, b, c or Lin()


QUAD = Quad()


class Comp:   # a component instance
    def __init__(self, tmpl, path, args, known):
        self.tmpl, self.path, self.args, self.known = tmpl, path, args, known
        self.signals = {}   # reference templates: name -> dims (list of ints)


class Return(Exception):
    def __init__(self, v):
        self.v = v


def arr(dims, fill):
    return fill() if not dims else [arr(dims[1:], fill) for _ in range(dims[0])]


class Run:
    def __init__(self, defs, sha_model=False, pos_model=False, smt_model=False, eddsa_model=False):
        self.defs = defs
        self.eddsa_model = eddsa_model
        self.smt_model = smt_model
        self.pos_model, self._pos = pos_model, {}   # Poseidon with every round signal (a few mains only: ~1 500 entries per component)
        self.sha_model = sha_model   # Sha256 with every wire and constraint (one main only: 40 k signals per block), else a black box
        self.forms, self.bases, self.declared, self.quads, self.models_used = {}, set(), [], [], set()

    # -- black boxes: circomlib templates whose outputs are linear in their inputs or that wrap another component ----------------
    def model(self, tmpl, path, args):
        f = self.forms
        one = lambda n: Lin(0, {n: 1})   # noqa: E731
        q = lambda a, b, c: self.quads.append((Lin.of(a), Lin.of(b), Lin.of(c)))   # noqa: E731   a * b = c
        self.models_used.add(tmpl)
        if tmpl == "Sha256" and self.sha_model:
            self.sha256(path, args[0])
            return
        if tmpl == "Poseidon" and self.pos_model:
            self.poseidon(path, args[0])
            return
        if tmpl == "SMTProcessor" and self.smt_model:
            self.smt_processor(path, args[0])
            return
        if tmpl == "Bits2Point_Strict" and self.eddsa_model:
            self.bits2point_strict(path)
            return
        if tmpl == "SMTVerifier" and self.smt_model:
            self.smt_verifier(path, args[0])
            return
        if tmpl == "EdDSAPoseidonVerifier" and self.eddsa_model:
            self.eddsa(path)
            return
        if tmpl == "Num2Bits":           # bitify.circom: out[i] * (out[i] - 1) === 0; sum of 2^i out[i] === in
            acc = Lin()
            for i in range(args[0]):
                b = one("%s.out[%d]" % (path, i))
                q(b, b.add(Lin(1), -1), Lin())
                acc = acc.add(b.scale(1 << i))
            q(Lin(), Lin(), acc.add(one(path + ".in"), -1))
        elif tmpl == "IsZero":           # comparators.circom: out <== -in * inv + 1; in * out === 0
            q(one(path + ".in").scale(-1), one(path + ".inv"), one(path + ".out").add(Lin(1), -1))
            q(one(path + ".in"), one(path + ".out"), Lin())
        if tmpl == "Bits2Num":
            acc = Lin()
            for i in range(args[0]):
                acc = acc.add(one("%s.in[%d]" % (path, i)).scale(1 << i))
            f[path + ".out"] = acc
        elif tmpl == "LessThan":
            n = args[0]
            f[path + ".n2b.in"] = one(path + ".in[0]").add(Lin(1 << n)).add(one(path + ".in[1]"), -1)
            f[path + ".out"] = Lin(1).add(one("%s.n2b.out[%d]" % (path, n)), -1)
            self.model("Num2Bits", path + ".n2b", [n + 1])
        elif tmpl in ("GreaterThan", "LessEqThan", "GreaterEqThan"):
            a, b = (1, 0) if tmpl != "LessEqThan" else (0, 1)
            f[path + ".lt.in[0]"] = one("%s.in[%d]" % (path, a))
            f[path + ".lt.in[1]"] = one("%s.in[%d]" % (path, b)).add(Lin(0 if tmpl == "GreaterThan" else 1))
            f[path + ".out"] = one(path + ".lt.out")
            self.model("LessThan", path + ".lt", args)
        elif tmpl in ("IsEqual", "ForceEqualIfEnabled"):
            f[path + ".isz.in"] = one(path + ".in[1]").add(one(path + ".in[0]"), -1)
            self.model("IsZero", path + ".isz", [])
            if tmpl == "IsEqual":
                f[path + ".out"] = one(path + ".isz.out")
            else:                        # (1 - isz.out) * enabled === 0
                q(Lin(1).add(one(path + ".isz.out"), -1), one(path + ".enabled"), Lin())
        elif tmpl == "NOT":
            f[path + ".out"] = Lin(1).add(one(path + ".in"), -1)
        elif tmpl == "Switcher":
            f[path + ".outL"] = one(path + ".aux").add(one(path + ".L"))
            f[path + ".outR"] = one(path + ".R").add(one(path + ".aux"), -1)
        elif tmpl in ("Mux1", "Mux2", "Mux3", "Mux4"):
            k = int(tmpl[3])
            for i in range(1 << k):
                f["%s.mux.c[0][%d]" % (path, i)] = one("%s.c[%d]" % (path, i))
            if k == 1:
                f[path + ".mux.s"] = one(path + ".s")
            else:
                for i in range(k):
                    f["%s.mux.s[%d]" % (path, i)] = one("%s.s[%d]" % (path, i))
            f[path + ".out"] = one(path + ".mux.out[0]")
            # MultiMux<k>(1): the term without a selector is a signal of its own, so is the difference that only the top selector
            # multiplies; MultiMux2's output is the plain sum of its terms (mux2.circom / mux3.circom / mux4.circom)
            m = path + ".mux."
            c = lambda i: one("%sc[0][%d]" % (m, i))   # noqa: E731
            sel = lambda i: one(m + "s") if k == 1 else one("%ss[%d]" % (m, i))   # noqa: E731
            if k == 1:                   # mux1.circom: out[i] <== (c[i][1] - c[i][0]) * s + c[i][0]
                q(c(1).add(c(0), -1), sel(0), one(m + "out[0]").add(c(0), -1))
                return
            # MultiMux2 / 3 / 4 (n = 1): the multilinear expansion over the LOW selector bits, one signal per term -- a<bits>[0] =
            # (alternating sum of the inputs below that bit set) * (product of those selector bits); the products of two and three
            # selector bits are signals themselves (s10, s20, s21, s210 <== s21 * s[0]); Mux3 / Mux4 keep the top bit apart:
            # out <== (sum of the terms with the top bit) * s[top] + (sum of the terms without)
            low = 2 if k <= 3 else 3
            top = low if k >= 3 else None
            digits = lambda T: "".join(str(b) for b in sorted(T, reverse=True))   # noqa: E731
            subsets = [[b for b in range(low) if (x >> b) & 1] for x in range(1 << low)]
            for T in subsets:
                if len(T) == 2:
                    q(sel(T[1]), sel(T[0]), one("%ss%s" % (m, digits(T))))
                elif len(T) == 3:
                    q(one(m + "s21"), sel(0), one(m + "s210"))

            def prod_of(T):
                return None if not T else sel(T[0]) if len(T) == 1 else one("%ss%s" % (m, digits(T)))
            sums = {False: Lin(), True: Lin()}
            for with_top in ([False, True] if top is not None else [False]):
                for T in subsets:
                    alt = Lin()
                    for x in range(1 << len(T)):
                        U = [T[j] for j in range(len(T)) if (x >> j) & 1]
                        idx = sum(1 << b for b in U)
                        sign = -1 if (len(T) - len(U)) % 2 else 1
                        term = c(idx + (1 << top)).add(c(idx), -1) if with_top else c(idx)
                        alt = alt.add(term, sign)
                    name = "%sa%s%s[0]" % (m, str(top) if with_top else "", digits(T))
                    if T:
                        q(alt, prod_of(T), one(name))
                    else:
                        f[name] = alt
                    sums[with_top] = sums[with_top].add(one(name))
            if top is None:
                f[m + "out[0]"] = sums[False]
            else:
                q(sums[True], sel(top), one(m + "out[0]").add(sums[False], -1))

    # -- circomlib 0.5.2 poseidon.circom as published: Ark (constants), Sigma (in2 = in * in, in4 = in2 * in2, out = in4 * in), Mix (the MDS
    # matrix), 4 full rounds, R_P partial rounds on lane 0, 4 full rounds; state[0] = 0, out = mix[last].out[0] --------------------------
    def poseidon(self, P_, n_in):
        sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tools"))
        from poseidon_params import N_ROUNDS_P, generate
        f = self.forms
        one = lambda n: Lin(0, {n: 1})   # noqa: E731
        t = n_in + 1
        if t not in self._pos:
            self._pos[t] = generate(t)
        C, M = self._pos[t]
        rp = N_ROUNDS_P[t - 2]

        def sigma(S):
            x, x2, x4 = one(S + ".in"), one(S + ".in2"), one(S + ".in4")
            self.quads.append((x, x, x2))
            self.quads.append((x2, x2, x4))
            self.quads.append((x4, x, one(S + ".out")))
        for i in range(8 + rp):
            A, X = "%s.ark[%d]" % (P_, i), "%s.mix[%d]" % (P_, i)
            for j in range(t):
                if i == 0:
                    f["%s.in[%d]" % (A, j)] = one("%s.inputs[%d]" % (P_, j - 1)) if j else Lin()
                else:
                    f["%s.in[%d]" % (A, j)] = one("%s.mix[%d].out[%d]" % (P_, i - 1, j))
                f["%s.out[%d]" % (A, j)] = one("%s.in[%d]" % (A, j)).add(Lin(C[t * i + j]))
            if i < 4 or i >= 4 + rp:
                k = i if i < 4 else i - rp
                for j in range(t):
                    S = "%s.sigmaF[%d][%d]" % (P_, k, j)
                    f[S + ".in"] = one("%s.out[%d]" % (A, j))
                    sigma(S)
                    f["%s.in[%d]" % (X, j)] = one(S + ".out")
            else:
                S = "%s.sigmaP[%d]" % (P_, i - 4)
                f[S + ".in"] = one(A + ".out[0]")
                sigma(S)
                f[X + ".in[0]"] = one(S + ".out")
                for j in range(1, t):
                    f["%s.in[%d]" % (X, j)] = one("%s.out[%d]" % (A, j))
            for r in range(t):
                acc = Lin()
                for j in range(t):
                    acc = acc.add(one("%s.in[%d]" % (X, j)).scale(M[r][j]))
                f["%s.out[%d]" % (X, r)] = acc
        f[P_ + ".out"] = one("%s.mix[%d].out[0]" % (P_, 8 + rp - 1))

    # -- circomlib 0.5.2 smt/smtprocessor.circom as published, with smtlevins, smtprocessorsm, smtprocessorlevel, smthash_poseidon, switcher,
    # gates (XOR, AND, MultiAND), bitify (Num2Bits_strict), aliascheck, compconstant ----------------------------------------------------
    # -- circomlib 0.5.2 pointbits.circom Bits2Point_Strict as published (AliasCheck, Bits2Num, BabyCheck, Num2Bits, CompConstant) -----------
    def bits2point_strict(self, B):
        f = self.forms
        one = lambda x: Lin(0, {x: 1})   # noqa: E731
        q = lambda a, b, c: self.quads.append((Lin.of(a), Lin.of(b), Lin.of(c)))   # noqa: E731

        def comp_constant(CC, ct):
            b, a, e = (1 << 128) - 1, 1, 1
            total = Lin()
            for i in range(127):
                clsb, cmsb = (ct >> (2 * i)) & 1, (ct >> (2 * i + 1)) & 1
                slsb, smsb, part = one("%s.in[%d]" % (CC, 2 * i)), one("%s.in[%d]" % (CC, 2 * i + 1)), one("%s.parts[%d]" % (CC, i))
                if not cmsb and not clsb:
                    q(smsb.scale(-b), slsb, part.add(smsb.scale(b), -1).add(slsb.scale(b), -1))
                elif not cmsb and clsb:
                    q(smsb.scale(a), slsb, part.add(slsb.scale(a)).add(smsb.scale(b), -1).add(smsb.scale(a)).add(Lin(a), -1))
                elif cmsb and not clsb:
                    q(smsb.scale(b), slsb, part.add(smsb.scale(a)).add(Lin(a), -1))
                else:
                    q(smsb.scale(-a), slsb, part.add(Lin(a), -1))
                total = total.add(part)
                b, a, e = b - e, a + e, e * 2
            f[CC + ".sout"] = total
            f[CC + ".num2bits.in"] = one(CC + ".sout")
            self.model("Num2Bits", CC + ".num2bits", [135])
            f[CC + ".out"] = one(CC + ".num2bits.out[127]")
        acc = Lin()
        for i in range(254):
            f["%s.aliasCheckY.in[%d]" % (B, i)] = one("%s.in[%d]" % (B, i))
            f["%s.aliasCheckY.compConstant.in[%d]" % (B, i)] = one("%s.aliasCheckY.in[%d]" % (B, i))
            f["%s.b2nY.in[%d]" % (B, i)] = one("%s.in[%d]" % (B, i))
            acc = acc.add(one("%s.b2nY.in[%d]" % (B, i)).scale(1 << i))
        comp_constant(B + ".aliasCheckY.compConstant", P - 1)
        q(Lin(), Lin(), one(B + ".aliasCheckY.compConstant.out"))
        q(Lin(), Lin(), one(B + ".in[254]"))                                  # in[254] === 0
        f[B + ".b2nY.out"] = acc
        f[B + ".out[1]"] = one(B + ".b2nY.out")
        self.bases.add(B + ".out[0]")                                        # out[0] <-- the square root with the sign of in[255]
        f[B + ".babyCheck.x"] = one(B + ".out[0]")
        f[B + ".babyCheck.y"] = one(B + ".out[1]")
        x, y, x2, y2 = (one("%s.babyCheck.%s" % (B, n)) for n in ("x", "y", "x2", "y2"))
        q(x, x, x2); q(y, y, y2)
        q(x2.scale(168696), y2, x2.scale(168700).add(y2).add(Lin(1), -1))    # a x2 + y2 === 1 + d x2 y2
        f[B + ".n2bX.in"] = one(B + ".out[0]")
        self.model("Num2Bits", B + ".n2bX", [254])
        for i in range(254):
            f["%s.aliasCheckX.in[%d]" % (B, i)] = one("%s.n2bX.out[%d]" % (B, i))
            f["%s.aliasCheckX.compConstant.in[%d]" % (B, i)] = one("%s.aliasCheckX.in[%d]" % (B, i))
            f["%s.signCalc.in[%d]" % (B, i)] = one("%s.n2bX.out[%d]" % (B, i))
        comp_constant(B + ".aliasCheckX.compConstant", P - 1)
        q(Lin(), Lin(), one(B + ".aliasCheckX.compConstant.out"))
        comp_constant(B + ".signCalc", 10944121435919637611123202872628637544274182200208017171849102093287904247808)
        q(Lin(), Lin(), one(B + ".signCalc.out").add(one(B + ".in[255]"), -1))

    def smt_parts(self):
        """the pieces SMTProcessor and SMTVerifier share: wire, Num2Bits_strict, Switcher, SMTHash1 / 2, SMTLevIns, MultiAND"""
        f = self.forms
        one = lambda x: Lin(0, {x: 1})   # noqa: E731
        q = lambda a, b, c: self.quads.append((Lin.of(a), Lin.of(b), Lin.of(c)))   # noqa: E731
        K1 = Lin(1)

        def wire(dst, src):
            f[dst] = src if isinstance(src, Lin) else one(src)

        def num2bits_strict(N):
            wire(N + ".n2b.in", N + ".in")
            self.model("Num2Bits", N + ".n2b", [254])
            for i in range(254):
                wire("%s.out[%d]" % (N, i), "%s.n2b.out[%d]" % (N, i))
                wire("%s.aliasCheck.in[%d]" % (N, i), "%s.n2b.out[%d]" % (N, i))
                wire("%s.aliasCheck.compConstant.in[%d]" % (N, i), "%s.aliasCheck.in[%d]" % (N, i))
            # compconstant.circom against ct = -1 = r - 1: parts of two bits each, their sum through Num2Bits(135), out = bit 127
            CC = N + ".aliasCheck.compConstant"
            ct = P - 1
            b, a, e = (1 << 128) - 1, 1, 1
            total = Lin()
            for i in range(127):
                clsb, cmsb = (ct >> (2 * i)) & 1, (ct >> (2 * i + 1)) & 1
                slsb, smsb, part = one("%s.in[%d]" % (CC, 2 * i)), one("%s.in[%d]" % (CC, 2 * i + 1)), one("%s.parts[%d]" % (CC, i))
                if not cmsb and not clsb:      # parts = -b smsb slsb + b smsb + b slsb
                    q(smsb.scale(-b), slsb, part.add(smsb.scale(b), -1).add(slsb.scale(b), -1))
                elif not cmsb and clsb:        # a smsb slsb - a slsb + b smsb - a smsb + a
                    q(smsb.scale(a), slsb, part.add(slsb.scale(a)).add(smsb.scale(b), -1).add(smsb.scale(a)).add(Lin(a), -1))
                elif cmsb and not clsb:        # b smsb slsb - a smsb + a
                    q(smsb.scale(b), slsb, part.add(smsb.scale(a)).add(Lin(a), -1))
                else:                          # -a smsb slsb + a
                    q(smsb.scale(-a), slsb, part.add(Lin(a), -1))
                total = total.add(part)
                b, a, e = b - e, a + e, e * 2
            wire(CC + ".sout", total)
            wire(CC + ".num2bits.in", CC + ".sout")
            self.model("Num2Bits", CC + ".num2bits", [135])
            wire(CC + ".out", CC + ".num2bits.out[127]")
            q(Lin(), Lin(), one(CC + ".out"))                                   # aliascheck.circom: compConstant.out === 0

        def switcher(W):               # switcher.circom
            q(one(W + ".R").add(one(W + ".L"), -1), one(W + ".sel"), one(W + ".aux"))
            wire(W + ".outL", one(W + ".aux").add(one(W + ".L")))
            wire(W + ".outR", one(W + ".R").add(one(W + ".aux"), -1))

        def hash_(Hc, ins):            # smthash_poseidon.circom: SMTHash1 = Poseidon(3) of (key, value, 1), SMTHash2 = Poseidon(2) of (L, R)
            for j, src in enumerate(ins):
                wire("%s.h.inputs[%d]" % (Hc, j), src)
            if self.pos_model:
                self.poseidon(Hc + ".h", len(ins))
            wire(Hc + ".out", Hc + ".h.out")


        def levins(S, n):              # smtlevins.circom
            LI = S + ".smtLevIns"
            wire(LI + ".enabled", S + ".enabled")
            for i in range(n):
                wire("%s.siblings[%d]" % (LI, i), "%s.siblings[%d]" % (S, i))
                wire("%s.isZero[%d].in" % (LI, i), "%s.siblings[%d]" % (LI, i))
                self.model("IsZero", "%s.isZero[%d]" % (LI, i), [])
            q(one("%s.isZero[%d].out" % (LI, n - 1)).add(K1, -1), one(LI + ".enabled"), Lin())
            wire("%s.levIns[%d]" % (LI, n - 1), K1.add(one("%s.isZero[%d].out" % (LI, n - 2)), -1))
            wire("%s.done[%d]" % (LI, n - 2), "%s.levIns[%d]" % (LI, n - 1))
            for i in range(n - 2, 0, -1):
                q(K1.add(one("%s.done[%d]" % (LI, i)), -1), K1.add(one("%s.isZero[%d].out" % (LI, i - 1)), -1), one("%s.levIns[%d]" % (LI, i)))
                wire("%s.done[%d]" % (LI, i - 1), one("%s.levIns[%d]" % (LI, i)).add(one("%s.done[%d]" % (LI, i))))
            wire(LI + ".levIns[0]", K1.add(one(LI + ".done[0]"), -1))

        def multi_and(Kk, n):          # gates.circom: MultiAND(1) = wire, MultiAND(2) = AND, else AND(MultiAND(n \ 2), MultiAND(n - n \ 2))
            if n == 1:
                wire(Kk + ".out", Kk + ".in[0]")
            elif n == 2:
                wire(Kk + ".and1.a", Kk + ".in[0]"); wire(Kk + ".and1.b", Kk + ".in[1]")
                q(one(Kk + ".and1.a"), one(Kk + ".and1.b"), one(Kk + ".and1.out"))
                wire(Kk + ".out", Kk + ".and1.out")
            else:
                n1 = n // 2
                for i in range(n1):
                    wire("%s.ands[0].in[%d]" % (Kk, i), "%s.in[%d]" % (Kk, i))
                for i in range(n - n1):
                    wire("%s.ands[1].in[%d]" % (Kk, i), "%s.in[%d]" % (Kk, n1 + i))
                multi_and(Kk + ".ands[0]", n1); multi_and(Kk + ".ands[1]", n - n1)
                wire(Kk + ".and2.a", Kk + ".ands[0].out"); wire(Kk + ".and2.b", Kk + ".ands[1].out")
                q(one(Kk + ".and2.a"), one(Kk + ".and2.b"), one(Kk + ".and2.out"))
                wire(Kk + ".out", Kk + ".and2.out")
        return f, one, q, K1, wire, num2bits_strict, switcher, hash_, levins, multi_and

    # -- circomlib 0.5.2 smt/smtverifier.circom as published, with smtverifiersm, smtverifierlevel ------------------------------------------
    def smt_verifier(self, S, n):
        f, one, q, K1, wire, num2bits_strict, switcher, hash_, levins, multi_and = self.smt_parts()
        for nm, key, value in (("hash1Old", "oldKey", "oldValue"), ("hash1New", "key", "value")):
            wire("%s.%s.key" % (S, nm), "%s.%s" % (S, key))
            wire("%s.%s.value" % (S, nm), "%s.%s" % (S, value))
            hash_("%s.%s" % (S, nm), ["%s.%s.key" % (S, nm), "%s.%s.value" % (S, nm), K1])
        wire(S + ".n2bOld.in", S + ".oldKey"); wire(S + ".n2bNew.in", S + ".key")
        num2bits_strict(S + ".n2bOld"); num2bits_strict(S + ".n2bNew")
        levins(S, n)
        st = ("top", "i0", "iold", "inew", "na")
        for i in range(n):
            M = "%s.sm[%d]" % (S, i)
            for nm in st:
                if i == 0:
                    wire("%s.prev_%s" % (M, nm), one(S + ".enabled") if nm == "top" else K1.add(one(S + ".enabled"), -1) if nm == "na" else Lin())
                else:
                    wire("%s.prev_%s" % (M, nm), "%s.sm[%d].st_%s" % (S, i - 1, nm))
            wire(M + ".is0", S + ".isOld0"); wire(M + ".fnc", S + ".fnc"); wire(M + ".levIns", "%s.smtLevIns.levIns[%d]" % (S, i))
            g = lambda x, M=M: one("%s.%s" % (M, x))   # noqa: E731
            q(g("prev_top"), g("levIns"), g("prev_top_lev_ins"))
            q(g("prev_top_lev_ins"), g("fnc"), g("prev_top_lev_ins_fnc"))
            wire(M + ".st_top", g("prev_top").add(g("prev_top_lev_ins"), -1))
            wire(M + ".st_inew", g("prev_top_lev_ins").add(g("prev_top_lev_ins_fnc"), -1))
            q(g("prev_top_lev_ins_fnc"), K1.add(g("is0"), -1), g("st_iold"))
            q(g("prev_top_lev_ins"), g("is0"), g("st_i0"))
            wire(M + ".st_na", g("prev_na").add(g("prev_inew")).add(g("prev_iold")).add(g("prev_i0")))
        last = "%s.sm[%d]" % (S, n - 1)
        q(Lin(), Lin(), one(last + ".st_na").add(one(last + ".st_iold")).add(one(last + ".st_inew")).add(one(last + ".st_i0")).add(K1, -1))
        for i in range(n - 1, -1, -1):
            Lv = "%s.levels[%d]" % (S, i)
            for nm in st:
                wire("%s.st_%s" % (Lv, nm), "%s.sm[%d].st_%s" % (S, i, nm))
            wire(Lv + ".sibling", "%s.siblings[%d]" % (S, i))
            wire(Lv + ".old1leaf", S + ".hash1Old.out"); wire(Lv + ".new1leaf", S + ".hash1New.out")
            wire(Lv + ".lrbit", "%s.n2bNew.out[%d]" % (S, i))
            wire(Lv + ".child", Lin() if i == n - 1 else one("%s.levels[%d].root" % (S, i + 1)))
            g = lambda x, Lv=Lv: one("%s.%s" % (Lv, x))   # noqa: E731
            wire(Lv + ".switcher.L", g("child")); wire(Lv + ".switcher.R", g("sibling")); wire(Lv + ".switcher.sel", g("lrbit"))
            switcher(Lv + ".switcher")
            hash_(Lv + ".proofHash", [Lv + ".proofHash.L", Lv + ".proofHash.R"])
            wire(Lv + ".proofHash.L", Lv + ".switcher.outL"); wire(Lv + ".proofHash.R", Lv + ".switcher.outR")
            q(g("proofHash.out"), g("st_top"), g("aux[0]"))
            q(g("old1leaf"), g("st_iold"), g("aux[1]"))
            q(g("new1leaf"), g("st_inew"), g("root").add(g("aux[0]"), -1).add(g("aux[1]"), -1))
        E = S + ".areKeyEquals"
        wire(E + ".in[0]", S + ".oldKey"); wire(E + ".in[1]", S + ".key")
        self.model("IsEqual", E, [])
        Kk = S + ".keysOk"
        wire(Kk + ".in[0]", S + ".fnc"); wire(Kk + ".in[1]", K1.add(one(S + ".isOld0"), -1)); wire(Kk + ".in[2]", E + ".out"); wire(Kk + ".in[3]", S + ".enabled")
        multi_and(Kk, 4)
        q(Lin(), Lin(), one(Kk + ".out"))
        Ck = S + ".checkRoot"
        wire(Ck + ".enabled", S + ".enabled"); wire(Ck + ".in[0]", S + ".levels[0].root"); wire(Ck + ".in[1]", S + ".root")
        self.model("ForceEqualIfEnabled", Ck, [])

    def smt_processor(self, S, n):
        f, one, q, K1, wire, num2bits_strict, switcher, hash_, levins, multi_and = self.smt_parts()
        fnc0, fnc1 = one(S + ".fnc[0]"), one(S + ".fnc[1]")
        q(fnc0, fnc1, fnc0.add(fnc1).add(one(S + ".enabled"), -1))             # enabled <== fnc[0] + fnc[1] - fnc[0]*fnc[1]
        for nm, key, value in (("hash1Old", "oldKey", "oldValue"), ("hash1New", "newKey", "newValue")):
            wire("%s.%s.key" % (S, nm), "%s.%s" % (S, key))
            wire("%s.%s.value" % (S, nm), "%s.%s" % (S, value))
            hash_("%s.%s" % (S, nm), ["%s.%s.key" % (S, nm), "%s.%s.value" % (S, nm), K1])
        wire(S + ".n2bOld.in", S + ".oldKey")
        wire(S + ".n2bNew.in", S + ".newKey")
        num2bits_strict(S + ".n2bOld")
        num2bits_strict(S + ".n2bNew")
        levins(S, n)
        LI = S + ".smtLevIns"
        for i in range(n):             # gates.circom XOR: out <== a + b - 2*a*b
            X = "%s.xors[%d]" % (S, i)
            wire(X + ".a", "%s.n2bOld.out[%d]" % (S, i))
            wire(X + ".b", "%s.n2bNew.out[%d]" % (S, i))
            q(one(X + ".a").scale(2), one(X + ".b"), one(X + ".a").add(one(X + ".b")).add(one(X + ".out"), -1))
        st = ("top", "old0", "bot", "new1", "na", "upd")
        for i in range(n):             # smtprocessorsm.circom
            M = "%s.sm[%d]" % (S, i)
            for nm in st:
                if i == 0:
                    wire("%s.prev_%s" % (M, nm), one(S + ".enabled") if nm == "top" else K1.add(one(S + ".enabled"), -1) if nm == "na" else Lin())
                else:
                    wire("%s.prev_%s" % (M, nm), "%s.sm[%d].st_%s" % (S, i - 1, nm))
            wire(M + ".is0", S + ".isOld0")
            wire(M + ".xor", "%s.xors[%d].out" % (S, i))
            wire(M + ".fnc[0]", S + ".fnc[0]")
            wire(M + ".fnc[1]", S + ".fnc[1]")
            wire(M + ".levIns", "%s.levIns[%d]" % (LI, i))
            g = lambda x, M=M: one("%s.%s" % (M, x))   # noqa: E731
            q(g("prev_top"), g("levIns"), g("aux1"))
            q(g("aux1"), g("fnc[0]"), g("aux2"))
            wire(M + ".st_top", g("prev_top").add(g("aux1"), -1))
            q(g("aux2"), g("is0"), g("st_old0"))
            mid = g("aux2").add(g("st_old0"), -1).add(g("prev_bot"))
            q(mid, g("xor"), g("st_new1"))
            q(K1.add(g("xor"), -1), mid, g("st_bot"))
            wire(M + ".st_upd", g("aux1").add(g("aux2"), -1))
            wire(M + ".st_na", g("prev_new1").add(g("prev_old0")).add(g("prev_na")).add(g("prev_upd")))
        last = "%s.sm[%d]" % (S, n - 1)
        q(Lin(), Lin(), one(last + ".st_na").add(one(last + ".st_new1")).add(one(last + ".st_old0")).add(one(last + ".st_upd")).add(K1, -1))
        for i in range(n - 1, -1, -1):  # smtprocessorlevel.circom
            Lv = "%s.levels[%d]" % (S, i)
            for nm in st:
                wire("%s.st_%s" % (Lv, nm), "%s.sm[%d].st_%s" % (S, i, nm))
            wire(Lv + ".sibling", "%s.siblings[%d]" % (S, i))
            wire(Lv + ".old1leaf", S + ".hash1Old.out")
            wire(Lv + ".new1leaf", S + ".hash1New.out")
            wire(Lv + ".newlrbit", "%s.n2bNew.out[%d]" % (S, i))
            wire(Lv + ".oldChild", Lin() if i == n - 1 else one("%s.levels[%d].oldRoot" % (S, i + 1)))
            wire(Lv + ".newChild", Lin() if i == n - 1 else one("%s.levels[%d].newRoot" % (S, i + 1)))
            g = lambda x, Lv=Lv: one("%s.%s" % (Lv, x))   # noqa: E731
            wire(Lv + ".oldSwitcher.L", g("oldChild")); wire(Lv + ".oldSwitcher.R", g("sibling")); wire(Lv + ".oldSwitcher.sel", g("newlrbit"))
            switcher(Lv + ".oldSwitcher")
            hash_(Lv + ".oldProofHash", [Lv + ".oldProofHash.L", Lv + ".oldProofHash.R"])
            wire(Lv + ".oldProofHash.L", Lv + ".oldSwitcher.outL"); wire(Lv + ".oldProofHash.R", Lv + ".oldSwitcher.outR")
            q(g("old1leaf"), g("st_bot").add(g("st_new1")).add(g("st_upd")), g("aux[0]"))
            q(g("oldProofHash.out"), g("st_top"), g("oldRoot").add(g("aux[0]"), -1))
            q(g("newChild"), g("st_top").add(g("st_bot")), g("aux[1]"))
            q(g("new1leaf"), g("st_new1"), g("newSwitcher.L").add(g("aux[1]"), -1))
            q(g("sibling"), g("st_top"), g("aux[2]"))
            q(g("old1leaf"), g("st_new1"), g("newSwitcher.R").add(g("aux[2]"), -1))
            wire(Lv + ".newSwitcher.sel", g("newlrbit"))
            switcher(Lv + ".newSwitcher")
            hash_(Lv + ".newProofHash", [Lv + ".newProofHash.L", Lv + ".newProofHash.R"])
            wire(Lv + ".newProofHash.L", Lv + ".newSwitcher.outL"); wire(Lv + ".newProofHash.R", Lv + ".newSwitcher.outR")
            q(g("newProofHash.out"), g("st_top").add(g("st_bot")).add(g("st_new1")), g("aux[3]"))
            q(g("new1leaf"), g("st_old0").add(g("st_upd")), g("newRoot").add(g("aux[3]"), -1))
        T = S + ".topSwitcher"
        q(fnc0, fnc1, one(T + ".sel"))
        wire(T + ".L", S + ".levels[0].oldRoot")
        wire(T + ".R", S + ".levels[0].newRoot")
        switcher(T)
        Ck = S + ".checkOldInput"
        wire(Ck + ".enabled", S + ".enabled"); wire(Ck + ".in[0]", S + ".oldRoot"); wire(Ck + ".in[1]", T + ".outL")
        self.model("ForceEqualIfEnabled", Ck, [])
        q(one(S + ".enabled"), one(T + ".outR").add(one(S + ".oldRoot"), -1), one(S + ".newRoot").add(one(S + ".oldRoot"), -1))
        E = S + ".areKeyEquals"
        wire(E + ".in[0]", S + ".oldKey"); wire(E + ".in[1]", S + ".newKey")
        self.model("IsEqual", E, [])
        Kk = S + ".keysOk"             # gates.circom MultiAND(3) = AND(MultiAND(1), MultiAND(2))
        wire(Kk + ".in[0]", K1.add(fnc0, -1)); wire(Kk + ".in[1]", fnc1); wire(Kk + ".in[2]", K1.add(one(E + ".out"), -1))
        wire(Kk + ".ands[0].in[0]", Kk + ".in[0]"); wire(Kk + ".ands[0].out", Kk + ".ands[0].in[0]")
        wire(Kk + ".ands[1].in[0]", Kk + ".in[1]"); wire(Kk + ".ands[1].in[1]", Kk + ".in[2]")
        wire(Kk + ".ands[1].and1.a", Kk + ".ands[1].in[0]"); wire(Kk + ".ands[1].and1.b", Kk + ".ands[1].in[1]")
        q(one(Kk + ".ands[1].and1.a"), one(Kk + ".ands[1].and1.b"), one(Kk + ".ands[1].and1.out"))
        wire(Kk + ".ands[1].out", Kk + ".ands[1].and1.out")
        wire(Kk + ".and2.a", Kk + ".ands[0].out"); wire(Kk + ".and2.b", Kk + ".ands[1].out")
        q(one(Kk + ".and2.a"), one(Kk + ".and2.b"), one(Kk + ".and2.out"))
        wire(Kk + ".out", Kk + ".and2.out")
        q(Lin(), Lin(), one(Kk + ".out"))                                       # keysOk.out === 0

    # -- circomlib 0.5.2 eddsaposeidon.circom as published, with babyjub (BabyAdd, BabyDbl), montgomery (Edwards2Montgomery,
    # Montgomery2Edwards, MontgomeryAdd, MontgomeryDouble), escalarmulany (Multiplexor2, BitElementMulAny, SegmentMulAny, EscalarMulAny),
    # escalarmulfix (WindowMulFix, SegmentMulFix, EscalarMulFix over BASE8), compconstant, bitify -------------------------------------------
    def eddsa(self, V):
        f = self.forms
        one = lambda x: Lin(0, {x: 1})   # noqa: E731
        q = lambda a, b, c: self.quads.append((Lin.of(a), Lin.of(b), Lin.of(c)))   # noqa: E731
        K1 = Lin(1)
        a_, d_ = 168700, 168696
        A_, B_ = 168698, 1                                   # (2 (a + d)) / (a - d), 4 / (a - d)
        BASE8 = (5299619240641551281634865583518297030282874472190772894086521144482721001553, 16950150798460657717958625567821834550301663161624707787222815936182638968203)

        def wire(dst, src):
            f[dst] = src if isinstance(src, Lin) else one(src)

        def pair(dst, src):                                  # dst[0], dst[1] <== src[0], src[1]
            for k in range(2):
                wire("%s[%d]" % (dst, k), "%s[%d]" % (src, k))

        def baby_add(C):
            x1, y1, x2, y2 = (one("%s.%s" % (C, n)) for n in ("x1", "y1", "x2", "y2"))
            beta, gamma, delta, tau = (one("%s.%s" % (C, n)) for n in ("beta", "gamma", "delta", "tau"))
            q(x1, y2, beta); q(y1, x2, gamma)
            q(x1.scale(-a_).add(y1), x2.add(y2), delta)
            q(beta, gamma, tau)
            q(K1.add(tau.scale(d_)), one(C + ".xout"), beta.add(gamma))
            q(K1.add(tau.scale(d_), -1), one(C + ".yout"), delta.add(beta.scale(a_)).add(gamma, -1))

        def baby_dbl(D):
            for k, n in (("x1", "x"), ("y1", "y"), ("x2", "x"), ("y2", "y")):
                wire("%s.adder.%s" % (D, k), "%s.%s" % (D, n))
            baby_add(D + ".adder")
            wire(D + ".xout", D + ".adder.xout"); wire(D + ".yout", D + ".adder.yout")

        def e2m(E):                                          # out[0] * (1 - in[1]) === 1 + in[1]; out[1] * in[0] === out[0]
            i0, i1, o0, o1 = one(E + ".in[0]"), one(E + ".in[1]"), one(E + ".out[0]"), one(E + ".out[1]")
            q(o0, K1.add(i1, -1), K1.add(i1)); q(o1, i0, o0)

        def m2e(E):                                          # out[0] * in[1] === in[0]; out[1] * (in[0] + 1) === in[0] - 1
            i0, i1, o0, o1 = one(E + ".in[0]"), one(E + ".in[1]"), one(E + ".out[0]"), one(E + ".out[1]")
            q(o0, i1, i0); q(o1, i0.add(K1), i0.add(K1, -1))

        def mont_add(M):
            a0, a1, b0, b1 = (one("%s.%s" % (M, n)) for n in ("in1[0]", "in1[1]", "in2[0]", "in2[1]"))
            lam, o0, o1 = one(M + ".lamda"), one(M + ".out[0]"), one(M + ".out[1]")
            q(lam, b0.add(a0, -1), b1.add(a1, -1))
            q(lam.scale(B_), lam, o0.add(Lin(A_)).add(a0).add(b0))          # out[0] <== B lamda^2 - A - in1[0] - in2[0]
            q(lam, a0.add(o0, -1), o1.add(a1))                                # out[1] <== lamda (in1[0] - out[0]) - in1[1]

        def mont_dbl(M):
            i0, i1, lam, x2, o0, o1 = (one("%s.%s" % (M, n)) for n in ("in[0]", "in[1]", "lamda", "x1_2", "out[0]", "out[1]"))
            q(i0, i0, x2)
            q(lam, i1.scale(2 * B_), x2.scale(3).add(i0.scale(2 * A_)).add(K1))
            q(lam.scale(B_), lam, o0.add(Lin(A_)).add(i0.scale(2)))
            q(lam, i0.add(o0, -1), o1.add(i1))

        def mux2(X):                                         # escalarmulany.circom Multiplexor2
            for k in range(2):
                lo, hi = one("%s.in[0][%d]" % (X, k)), one("%s.in[1][%d]" % (X, k))
                q(hi.add(lo, -1), one(X + ".sel"), one("%s.out[%d]" % (X, k)).add(lo, -1))

        def segment_any(G, n):
            pair(G + ".e2m.in", G + ".p")
            e2m(G + ".e2m")
            for i in range(n - 1):
                Bt = "%s.bits[%d]" % (G, i)
                if i == 0:
                    pair(Bt + ".dblIn", G + ".e2m.out"); pair(Bt + ".addIn", G + ".e2m.out")
                else:
                    pair(Bt + ".dblIn", "%s.bits[%d].dblOut" % (G, i - 1)); pair(Bt + ".addIn", "%s.bits[%d].addOut" % (G, i - 1))
                wire(Bt + ".sel", "%s.e[%d]" % (G, i + 1))
                wire(Bt + ".selector.sel", Bt + ".sel")
                pair(Bt + ".doubler.in", Bt + ".dblIn")
                pair(Bt + ".adder.in1", Bt + ".doubler.out"); pair(Bt + ".adder.in2", Bt + ".addIn")
                pair(Bt + ".selector.in[0]", Bt + ".addIn"); pair(Bt + ".selector.in[1]", Bt + ".adder.out")
                pair(Bt + ".dblOut", Bt + ".doubler.out"); pair(Bt + ".addOut", Bt + ".selector.out")
                mont_dbl(Bt + ".doubler"); mont_add(Bt + ".adder"); mux2(Bt + ".selector")
            pair(G + ".dbl", "%s.bits[%d].dblOut" % (G, n - 2))
            pair(G + ".m2e.in", "%s.bits[%d].addOut" % (G, n - 2))
            m2e(G + ".m2e")
            wire(G + ".eadder.x1", G + ".m2e.out[0]"); wire(G + ".eadder.y1", G + ".m2e.out[1]")
            wire(G + ".eadder.x2", one(G + ".p[0]").scale(-1)); wire(G + ".eadder.y2", G + ".p[1]")
            baby_add(G + ".eadder")
            wire(G + ".lastSel.sel", G + ".e[0]")
            wire(G + ".lastSel.in[0][0]", G + ".eadder.xout"); wire(G + ".lastSel.in[0][1]", G + ".eadder.yout")
            pair(G + ".lastSel.in[1]", G + ".m2e.out")
            mux2(G + ".lastSel")
            pair(G + ".out", G + ".lastSel.out")

        def escalar_any(Mx, n):
            nseg_all = (n - 1) // 148 + 1
            nlast = n - (nseg_all - 1) * 148
            wire(Mx + ".zeropoint.in", Mx + ".p[0]")
            self.model("IsZero", Mx + ".zeropoint", [])
            zp = one(Mx + ".zeropoint.out")
            for s_ in range(nseg_all):
                nseg = 148 if s_ < nseg_all - 1 else nlast
                G = "%s.segments[%d]" % (Mx, s_)
                for i in range(nseg):
                    wire("%s.e[%d]" % (G, i), "%s.e[%d]" % (Mx, s_ * 148 + i))
                if s_ == 0:
                    for k in range(2):                       # the G8 point instead of a zero input point
                        p = one("%s.p[%d]" % (Mx, k))
                        q(Lin(BASE8[k]).add(p, -1), zp, one("%s.p[%d]" % (G, k)).add(p, -1))
                else:
                    D, E, Ad = "%s.doublers[%d]" % (Mx, s_ - 1), "%s.m2e[%d]" % (Mx, s_ - 1), "%s.adders[%d]" % (Mx, s_ - 1)
                    pair(D + ".in", "%s.segments[%d].dbl" % (Mx, s_ - 1))
                    mont_dbl(D)
                    pair(E + ".in", D + ".out")
                    m2e(E)
                    pair(G + ".p", E + ".out")
                    if s_ == 1:
                        wire(Ad + ".x1", "%s.segments[0].out[0]" % Mx); wire(Ad + ".y1", "%s.segments[0].out[1]" % Mx)
                    else:
                        wire(Ad + ".x1", "%s.adders[%d].xout" % (Mx, s_ - 2)); wire(Ad + ".y1", "%s.adders[%d].yout" % (Mx, s_ - 2))
                    wire(Ad + ".x2", G + ".out[0]"); wire(Ad + ".y2", G + ".out[1]")
                    baby_add(Ad)
                segment_any(G, nseg)
            fx, fy = ("%s.segments[0].out[0]" % Mx, "%s.segments[0].out[1]" % Mx) if nseg_all == 1 else ("%s.adders[%d].xout" % (Mx, nseg_all - 2), "%s.adders[%d].yout" % (Mx, nseg_all - 2))
            q(one(fx), K1.add(zp, -1), one(Mx + ".out[0]"))
            q(K1.add(one(fy), -1), zp, one(Mx + ".out[1]").add(one(fy), -1))

        def window_fix(W):
            for j in range(3):
                wire("%s.mux.s[%d]" % (W, j), "%s.in[%d]" % (W, j))
            pair(W + ".dbl2.in", W + ".base")
            mont_dbl(W + ".dbl2")
            prev = W + ".dbl2.out"
            for k in range(3, 9):
                Ad = "%s.adr%d" % (W, k)
                pair(Ad + ".in1", W + ".base"); pair(Ad + ".in2", prev)
                mont_add(Ad)
                prev = Ad + ".out"
            srcs = [W + ".base", W + ".dbl2.out"] + ["%s.adr%d.out" % (W, k) for k in range(3, 9)]
            for j, src in enumerate(srcs):
                for k in range(2):
                    wire("%s.mux.c[%d][%d]" % (W, k, j), "%s[%d]" % (src, k))
            pair(W + ".out8", W + ".adr8.out")
            pair(W + ".out", W + ".mux.out")
            # MultiMux3(2): s10 <== s[1] * s[0]; a210 / a21 / a20 / a10 / a1 / a0 products, a2 and a linear; out = (..) * s[2] + (..)
            m = W + ".mux."
            sel = lambda i: one("%ss[%d]" % (m, i))   # noqa: E731
            q(sel(1), sel(0), one(m + "s10"))
            for k in range(2):
                c = lambda i, k=k: one("%sc[%d][%d]" % (m, k, i))   # noqa: E731
                nm = lambda t, k=k: one("%s%s[%d]" % (m, t, k))   # noqa: E731
                q(c(7).add(c(6), -1).add(c(5), -1).add(c(4)).add(c(3), -1).add(c(2)).add(c(1)).add(c(0), -1), one(m + "s10"), nm("a210"))
                q(c(6).add(c(4), -1).add(c(2), -1).add(c(0)), sel(1), nm("a21"))
                q(c(5).add(c(4), -1).add(c(1), -1).add(c(0)), sel(0), nm("a20"))
                wire("%sa2[%d]" % (m, k), c(4).add(c(0), -1))
                q(c(3).add(c(2), -1).add(c(1), -1).add(c(0)), one(m + "s10"), nm("a10"))
                q(c(2).add(c(0), -1), sel(1), nm("a1"))
                q(c(1).add(c(0), -1), sel(0), nm("a0"))
                wire("%sa[%d]" % (m, k), c(0))
                q(nm("a210").add(nm("a21")).add(nm("a20")).add(nm("a2")), sel(2), nm("out").add(nm("a10"), -1).add(nm("a1"), -1).add(nm("a0"), -1).add(nm("a"), -1))

        def segment_fix(G, nw):
            pair(G + ".e2m.in", G + ".base")
            e2m(G + ".e2m")
            for i in range(nw):
                W, Cd = "%s.windows[%d]" % (G, i), "%s.cadders[%d]" % (G, i)
                if i == 0:
                    pair(W + ".base", G + ".e2m.out"); pair(Cd + ".in1", G + ".e2m.out")
                else:
                    pair(W + ".base", "%s.windows[%d].out8" % (G, i - 1)); pair(Cd + ".in1", "%s.cadders[%d].out" % (G, i - 1))
                for j in range(3):
                    wire("%s.in[%d]" % (W, j), "%s.e[%d]" % (G, 3 * i + j))
                if i < nw - 1:
                    pair(Cd + ".in2", W + ".out8")
                else:
                    pair(G + ".dblLast.in", W + ".out8")
                    mont_dbl(G + ".dblLast")
                    pair(Cd + ".in2", G + ".dblLast.out")
                window_fix(W)
                mont_add(Cd)
            for i in range(nw):
                Ad = "%s.adders[%d]" % (G, i)
                pair(Ad + ".in1", G + ".dblLast.out" if i == 0 else "%s.adders[%d].out" % (G, i - 1))
                pair(Ad + ".in2", "%s.windows[%d].out" % (G, i))
                mont_add(Ad)
            pair(G + ".m2e.in", "%s.adders[%d].out" % (G, nw - 1)); m2e(G + ".m2e")
            pair(G + ".cm2e.in", "%s.cadders[%d].out" % (G, nw - 1)); m2e(G + ".cm2e")
            wire(G + ".cAdd.x1", G + ".m2e.out[0]"); wire(G + ".cAdd.y1", G + ".m2e.out[1]")
            wire(G + ".cAdd.x2", one(G + ".cm2e.out[0]").scale(-1)); wire(G + ".cAdd.y2", G + ".cm2e.out[1]")
            baby_add(G + ".cAdd")
            wire(G + ".out[0]", G + ".cAdd.xout"); wire(G + ".out[1]", G + ".cAdd.yout")
            pair(G + ".dbl", "%s.windows[%d].out8" % (G, nw - 1))

        def escalar_fix(Mx, n):
            nseg_all = (n - 1) // 246 + 1                # 246 bits = 82 windows of three per segment
            nlast = n - (nseg_all - 1) * 246
            for s_ in range(nseg_all):
                nseg = 246 if s_ < nseg_all - 1 else nlast
                nw = (nseg - 1) // 3 + 1
                G = "%s.segments[%d]" % (Mx, s_)
                for i in range(nw * 3):
                    wire("%s.e[%d]" % (G, i), one("%s.e[%d]" % (Mx, s_ * 246 + i)) if i < nseg else Lin())
                if s_ == 0:
                    wire(G + ".base[0]", Lin(BASE8[0])); wire(G + ".base[1]", Lin(BASE8[1]))
                else:
                    E, Ad = "%s.m2e[%d]" % (Mx, s_ - 1), "%s.adders[%d]" % (Mx, s_ - 1)
                    pair(E + ".in", "%s.segments[%d].dbl" % (Mx, s_ - 1))
                    m2e(E)
                    pair(G + ".base", E + ".out")
                    if s_ == 1:
                        wire(Ad + ".x1", "%s.segments[0].out[0]" % Mx); wire(Ad + ".y1", "%s.segments[0].out[1]" % Mx)
                    else:
                        wire(Ad + ".x1", "%s.adders[%d].xout" % (Mx, s_ - 2)); wire(Ad + ".y1", "%s.adders[%d].yout" % (Mx, s_ - 2))
                    wire(Ad + ".x2", G + ".out[0]"); wire(Ad + ".y2", G + ".out[1]")
                    baby_add(Ad)
                segment_fix(G, nw)
            if nseg_all == 1:
                pair(Mx + ".out", Mx + ".segments[0].out")
            else:
                wire(Mx + ".out[0]", "%s.adders[%d].xout" % (Mx, nseg_all - 2)); wire(Mx + ".out[1]", "%s.adders[%d].yout" % (Mx, nseg_all - 2))

        def comp_constant(CC, ct):     # compconstant.circom
            b, a, e = (1 << 128) - 1, 1, 1
            total = Lin()
            for i in range(127):
                clsb, cmsb = (ct >> (2 * i)) & 1, (ct >> (2 * i + 1)) & 1
                slsb, smsb, part = one("%s.in[%d]" % (CC, 2 * i)), one("%s.in[%d]" % (CC, 2 * i + 1)), one("%s.parts[%d]" % (CC, i))
                if not cmsb and not clsb:
                    q(smsb.scale(-b), slsb, part.add(smsb.scale(b), -1).add(slsb.scale(b), -1))
                elif not cmsb and clsb:
                    q(smsb.scale(a), slsb, part.add(slsb.scale(a)).add(smsb.scale(b), -1).add(smsb.scale(a)).add(Lin(a), -1))
                elif cmsb and not clsb:
                    q(smsb.scale(b), slsb, part.add(smsb.scale(a)).add(Lin(a), -1))
                else:
                    q(smsb.scale(-a), slsb, part.add(Lin(a), -1))
                total = total.add(part)
                b, a, e = b - e, a + e, e * 2
            wire(CC + ".sout", total)
            wire(CC + ".num2bits.in", CC + ".sout")
            self.model("Num2Bits", CC + ".num2bits", [135])
            wire(CC + ".out", CC + ".num2bits.out[127]")

        en = one(V + ".enabled")
        wire(V + ".snum2bits.in", V + ".S")
        self.model("Num2Bits", V + ".snum2bits", [253])
        for i in range(253):
            wire("%s.compConstant.in[%d]" % (V, i), "%s.snum2bits.out[%d]" % (V, i))
        wire(V + ".compConstant.in[253]", Lin())
        comp_constant(V + ".compConstant", 2736030358979909402780800718157159386076813972158567259200215660948447373040)
        q(one(V + ".compConstant.out"), en, Lin())
        for j, nm in enumerate(("R8x", "R8y", "Ax", "Ay", "M")):
            wire("%s.hash.inputs[%d]" % (V, j), "%s.%s" % (V, nm))
        if self.pos_model:
            self.poseidon(V + ".hash", 5)
        # Num2Bits_strict on the hash
        H = V + ".h2bits"
        wire(H + ".in", V + ".hash.out")
        wire(H + ".n2b.in", H + ".in")
        self.model("Num2Bits", H + ".n2b", [254])
        for i in range(254):
            wire("%s.out[%d]" % (H, i), "%s.n2b.out[%d]" % (H, i))
            wire("%s.aliasCheck.in[%d]" % (H, i), "%s.n2b.out[%d]" % (H, i))
            wire("%s.aliasCheck.compConstant.in[%d]" % (H, i), "%s.aliasCheck.in[%d]" % (H, i))
        comp_constant(H + ".aliasCheck.compConstant", P - 1)
        q(Lin(), Lin(), one(H + ".aliasCheck.compConstant.out"))
        wire(V + ".dbl1.x", V + ".Ax"); wire(V + ".dbl1.y", V + ".Ay")
        wire(V + ".dbl2.x", V + ".dbl1.xout"); wire(V + ".dbl2.y", V + ".dbl1.yout")
        wire(V + ".dbl3.x", V + ".dbl2.xout"); wire(V + ".dbl3.y", V + ".dbl2.yout")
        for nm in ("dbl1", "dbl2", "dbl3"):
            baby_dbl("%s.%s" % (V, nm))
        wire(V + ".isZero.in", V + ".dbl3.x")
        self.model("IsZero", V + ".isZero", [])
        q(one(V + ".isZero.out"), en, Lin())
        for i in range(254):
            wire("%s.mulAny.e[%d]" % (V, i), "%s.out[%d]" % (H, i))
        wire(V + ".mulAny.p[0]", V + ".dbl3.xout"); wire(V + ".mulAny.p[1]", V + ".dbl3.yout")
        escalar_any(V + ".mulAny", 254)
        wire(V + ".addRight.x1", V + ".R8x"); wire(V + ".addRight.y1", V + ".R8y")
        wire(V + ".addRight.x2", V + ".mulAny.out[0]"); wire(V + ".addRight.y2", V + ".mulAny.out[1]")
        baby_add(V + ".addRight")
        for i in range(253):
            wire("%s.mulFix.e[%d]" % (V, i), "%s.snum2bits.out[%d]" % (V, i))
        escalar_fix(V + ".mulFix", 253)
        for nm, a0, a1 in (("eqCheckX", "mulFix.out[0]", "addRight.xout"), ("eqCheckY", "mulFix.out[1]", "addRight.yout")):
            Cq = "%s.%s" % (V, nm)
            wire(Cq + ".enabled", V + ".enabled"); wire(Cq + ".in[0]", "%s.%s" % (V, a0)); wire(Cq + ".in[1]", "%s.%s" % (V, a1))
            self.model("ForceEqualIfEnabled", Cq, [])

    # -- circomlib 0.5.2 sha256/*.circom as published: every wire and every constraint of Sha256(nBits) -----------------------------------
    def sha256(self, P, n_bits):
        f = self.forms
        one = lambda n: Lin(0, {n: 1})   # noqa: E731
        q = lambda a, b, c: self.quads.append((Lin.of(a), Lin.of(b), Lin.of(c)))   # noqa: E731
        H0 = [0x6a09e667, 0xbb67ae85, 0x3c6ef372, 0xa54ff53a, 0x510e527f, 0x9b05688c, 0x1f83d9ab, 0x5be0cd19]
        K = [0x428a2f98, 0x71374491, 0xb5c0fbcf, 0xe9b5dba5, 0x3956c25b, 0x59f111f1, 0x923f82a4, 0xab1c5ed5, 0xd807aa98, 0x12835b01, 0x243185be, 0x550c7dc3,
             0x72be5d74, 0x80deb1fe, 0x9bdc06a7, 0xc19bf174, 0xe49b69c1, 0xefbe4786, 0x0fc19dc6, 0x240ca1cc, 0x2de92c6f, 0x4a7484aa, 0x5cb0a9dc, 0x76f988da,
             0x983e5152, 0xa831c66d, 0xb00327c8, 0xbf597fc7, 0xc6e00bf3, 0xd5a79147, 0x06ca6351, 0x14292967, 0x27b70a85, 0x2e1b2138, 0x4d2c6dfc, 0x53380d13,
             0x650a7354, 0x766a0abb, 0x81c2c92e, 0x92722c85, 0xa2bfe8a1, 0xa81a664b, 0xc24b8b70, 0xc76c51a3, 0xd192e819, 0xd6990624, 0xf40e3585, 0x106aa070,
             0x19a4c116, 0x1e376c08, 0x2748774c, 0x34b0bcb5, 0x391c0cb3, 0x4ed8aa4a, 0x5b9cca4f, 0x682e6ff3, 0x748f82ee, 0x78a5636f, 0x84c87814, 0x8cc70208,
             0x90befffa, 0xa4506ceb, 0xbef9a3f7, 0xc67178f2]
        nb = (n_bits + 64) // 512 + 1

        def wire(dst, src):
            f[dst] = src if isinstance(src, Lin) else one(src)

        def binsum(B, ops):            # binsum.circom: out bits are bits; sum of 2^k in[j][k] === sum of 2^k out[k]
            nout = ((2 ** 32 - 1) * ops).bit_length()
            lin, lout = Lin(), Lin()
            for k in range(32):
                for j in range(ops):
                    lin = lin.add(one("%s.in[%d][%d]" % (B, j, k)).scale(1 << k))
            for k in range(nout):
                b = one("%s.out[%d]" % (B, k))
                q(b, b.add(Lin(1), -1), Lin())
                lout = lout.add(b.scale(1 << k))
            q(Lin(), Lin(), lin.add(lout, -1))

        def xor3(X):                   # xor3.circom: mid = b * c; out = a * (1 - 2b - 2c + 4 mid) + b + c - 2 mid
            for k in range(32):
                a, b, c, mid = (one("%s.%s[%d]" % (X, n, k)) for n in ("a", "b", "c", "mid"))
                q(b, c, mid)
                q(a, Lin(1).add(b.scale(2), -1).add(c.scale(2), -1).add(mid.scale(4)), one("%s.out[%d]" % (X, k)).add(b, -1).add(c, -1).add(mid.scale(2)))

        def sigma(G, ra, rb, rc, big):  # sigma.circom: SmallSigma = RotR, RotR, ShR; BigSigma = three RotR; then Xor3
            for nm, r, shift in (("rota", ra, False), ("rotb", rb, False), ("rotc" if big else "shrc", rc, not big)):
                for k in range(32):
                    wire("%s.%s.in[%d]" % (G, nm, k), "%s.in[%d]" % (G, k))
                    if shift:
                        wire("%s.%s.out[%d]" % (G, nm, k), Lin() if k + r >= 32 else one("%s.%s.in[%d]" % (G, nm, k + r)))
                    else:
                        wire("%s.%s.out[%d]" % (G, nm, k), "%s.%s.in[%d]" % (G, nm, (k + r) % 32))
            for k in range(32):
                wire("%s.xor3.a[%d]" % (G, k), "%s.rota.out[%d]" % (G, k))
                wire("%s.xor3.b[%d]" % (G, k), "%s.rotb.out[%d]" % (G, k))
                wire("%s.xor3.c[%d]" % (G, k), "%s.%s.out[%d]" % (G, "rotc" if big else "shrc", k))
                wire("%s.out[%d]" % (G, k), "%s.xor3.out[%d]" % (G, k))
            xor3(G + ".xor3")

        for k in range(nb * 512):
            if k < n_bits:
                src = one("%s.in[%d]" % (P, k))
            elif k == n_bits:
                src = Lin(1)
            elif k < nb * 512 - 64:
                src = Lin()
            else:
                src = Lin((n_bits >> (nb * 512 - 1 - k)) & 1)
            wire("%s.paddedIn[%d]" % (P, k), src)
        for j in range(8):
            for k in range(32):
                wire("%s.ha%d.out[%d]" % (P, j, k), Lin((H0[j] >> k) & 1))
        regs = "abcdefgh"
        for i in range(nb):
            C = "%s.sha256compression[%d]" % (P, i)
            for j in range(8):
                for k in range(32):
                    wire("%s.hin[%d]" % (C, 32 * j + k), "%s.ha%d.out[%d]" % (P, j, k) if i == 0 else "%s.sha256compression[%d].out[%d]" % (P, i - 1, 32 * j + 31 - k))
            for k in range(512):
                wire("%s.inp[%d]" % (C, k), "%s.paddedIn[%d]" % (P, i * 512 + k))
            for t in range(64):
                for k in range(32):
                    wire("%s.ct_k[%d].out[%d]" % (C, t, k), Lin((K[t] >> k) & 1))
                if t >= 16:
                    S = "%s.sigmaPlus[%d]" % (C, t - 16)
                    for k in range(32):
                        wire("%s.in2[%d]" % (S, k), "%s.w[%d][%d]" % (C, t - 2, k))
                        wire("%s.in7[%d]" % (S, k), "%s.w[%d][%d]" % (C, t - 7, k))
                        wire("%s.in15[%d]" % (S, k), "%s.w[%d][%d]" % (C, t - 15, k))
                        wire("%s.in16[%d]" % (S, k), "%s.w[%d][%d]" % (C, t - 16, k))
                        wire("%s.sigma1.in[%d]" % (S, k), "%s.in2[%d]" % (S, k))
                        wire("%s.sigma0.in[%d]" % (S, k), "%s.in15[%d]" % (S, k))
                        wire("%s.sum.in[0][%d]" % (S, k), "%s.sigma1.out[%d]" % (S, k))
                        wire("%s.sum.in[1][%d]" % (S, k), "%s.in7[%d]" % (S, k))
                        wire("%s.sum.in[2][%d]" % (S, k), "%s.sigma0.out[%d]" % (S, k))
                        wire("%s.sum.in[3][%d]" % (S, k), "%s.in16[%d]" % (S, k))
                        wire("%s.out[%d]" % (S, k), "%s.sum.out[%d]" % (S, k))
                    sigma(S + ".sigma1", 17, 19, 10, False)
                    sigma(S + ".sigma0", 7, 18, 3, False)
                    binsum(S + ".sum", 4)
                for k in range(32):
                    wire("%s.w[%d][%d]" % (C, t, k), "%s.inp[%d]" % (C, t * 32 + 31 - k) if t < 16 else "%s.sigmaPlus[%d].out[%d]" % (C, t - 16, k))
            for j, rname in enumerate(regs):
                for k in range(32):
                    wire("%s.%s[0][%d]" % (C, rname, k), "%s.hin[%d]" % (C, 32 * j + k))
            for t in range(64):
                T1, T2 = "%s.t1[%d]" % (C, t), "%s.t2[%d]" % (C, t)
                for k in range(32):
                    for nm in "hefg":
                        wire("%s.%s[%d]" % (T1, nm, k), "%s.%s[%d][%d]" % (C, nm, t, k))
                    wire("%s.k[%d]" % (T1, k), "%s.ct_k[%d].out[%d]" % (C, t, k))
                    wire("%s.w[%d]" % (T1, k), "%s.w[%d][%d]" % (C, t, k))
                    for nm in "abc":
                        wire("%s.%s[%d]" % (T2, nm, k), "%s.%s[%d][%d]" % (C, nm, t, k))
                    # T1: ch = Ch_t(32) on (e, f, g), bigsigma1 = BigSigma(6, 11, 25) on e, sum = BinSum(32, 5) of h, bigsigma1, ch, k, w
                    wire("%s.bigsigma1.in[%d]" % (T1, k), "%s.e[%d]" % (T1, k))
                    wire("%s.ch.a[%d]" % (T1, k), "%s.e[%d]" % (T1, k))
                    wire("%s.ch.b[%d]" % (T1, k), "%s.f[%d]" % (T1, k))
                    wire("%s.ch.c[%d]" % (T1, k), "%s.g[%d]" % (T1, k))
                    a, b, c = (one("%s.ch.%s[%d]" % (T1, n, k)) for n in "abc")
                    q(a, b.add(c, -1), one("%s.ch.out[%d]" % (T1, k)).add(c, -1))          # ch.circom: out = a * (b - c) + c
                    for j, src in enumerate(("h[%d]" % k, "bigsigma1.out[%d]" % k, "ch.out[%d]" % k, "k[%d]" % k, "w[%d]" % k)):
                        wire("%s.sum.in[%d][%d]" % (T1, j, k), "%s.%s" % (T1, src))
                    wire("%s.out[%d]" % (T1, k), "%s.sum.out[%d]" % (T1, k))
                    # T2: bigsigma0 = BigSigma(2, 13, 22) on a, maj = Maj_t(32) on (a, b, c), sum = BinSum(32, 2)
                    wire("%s.bigsigma0.in[%d]" % (T2, k), "%s.a[%d]" % (T2, k))
                    for nm in "abc":
                        wire("%s.maj.%s[%d]" % (T2, nm, k), "%s.%s[%d]" % (T2, nm, k))
                    a, b, c, mid = (one("%s.maj.%s[%d]" % (T2, n, k)) for n in ("a", "b", "c", "mid"))
                    q(b, c, mid)                                                              # maj.circom: mid = b * c; out = a * (b + c - 2 mid) + mid
                    q(a, b.add(c).add(mid.scale(2), -1), one("%s.maj.out[%d]" % (T2, k)).add(mid, -1))
                    wire("%s.sum.in[0][%d]" % (T2, k), "%s.bigsigma0.out[%d]" % (T2, k))
                    wire("%s.sum.in[1][%d]" % (T2, k), "%s.maj.out[%d]" % (T2, k))
                    wire("%s.out[%d]" % (T2, k), "%s.sum.out[%d]" % (T2, k))
                    wire("%s.sume[%d].in[0][%d]" % (C, t, k), "%s.d[%d][%d]" % (C, t, k))
                    wire("%s.sume[%d].in[1][%d]" % (C, t, k), "%s.out[%d]" % (T1, k))
                    wire("%s.suma[%d].in[0][%d]" % (C, t, k), "%s.out[%d]" % (T1, k))
                    wire("%s.suma[%d].in[1][%d]" % (C, t, k), "%s.out[%d]" % (T2, k))
                    for dst, src in (("h", "g"), ("g", "f"), ("f", "e"), ("d", "c"), ("c", "b"), ("b", "a")):
                        wire("%s.%s[%d][%d]" % (C, dst, t + 1, k), "%s.%s[%d][%d]" % (C, src, t, k))
                    wire("%s.e[%d][%d]" % (C, t + 1, k), "%s.sume[%d].out[%d]" % (C, t, k))
                    wire("%s.a[%d][%d]" % (C, t + 1, k), "%s.suma[%d].out[%d]" % (C, t, k))
                sigma(T1 + ".bigsigma1", 6, 11, 25, True)
                sigma(T2 + ".bigsigma0", 2, 13, 22, True)
                binsum(T1 + ".sum", 5)
                binsum(T2 + ".sum", 2)
                binsum("%s.sume[%d]" % (C, t), 2)
                binsum("%s.suma[%d]" % (C, t), 2)
            for j, rname in enumerate(regs):
                for k in range(32):
                    wire("%s.fsum[%d].in[0][%d]" % (C, j, k), "%s.hin[%d]" % (C, 32 * j + k))
                    wire("%s.fsum[%d].in[1][%d]" % (C, j, k), "%s.%s[64][%d]" % (C, rname, k))
                    wire("%s.out[%d]" % (C, 32 * j + 31 - k), "%s.fsum[%d].out[%d]" % (C, j, k))
                binsum("%s.fsum[%d]" % (C, j), 2)
        for k in range(256):
            wire("%s.out[%d]" % (P, k), "%s.sha256compression[%d].out[%d]" % (P, nb - 1, k))

    # -- templates ---------------------------------------------------------------------------------------------------------------------
    def instantiate(self, tmpl, args, path):
        if tmpl not in self.defs or self.defs[tmpl][0] != "template":
            self.model(tmpl, path, args)
            return Comp(tmpl, path, args, False)
        _, params, body = self.defs[tmpl]
        c = Comp(tmpl, path, args, True)
        env = {"vars": dict(zip(params, args)), "comp": c, "components": {}, "is_main": path == "main"}
        self.exec(body, env)
        return c

    def sig_names(self, base, dims):
        if not dims:
            return [base]
        return [n for i in range(dims[0]) for n in self.sig_names("%s[%d]" % (base, i), dims[1:])]

    def exec(self, node, env):
        k = node[0]
        if k == "block":
            for s in node[1]:
                self.exec(s, env)
        elif k == "signal":
            c = env["comp"]
            for name, dims in node[2]:
                d = [self.const(self.eval(x, env)) for x in dims]
                c.signals[name] = d
                names = self.sig_names("%s.%s" % (c.path, name), d)
                self.declared += names
                if node[1] == "input" and env["is_main"]:
                    self.bases.update(names)
        elif k == "component":
            _, name, dims, init = node
            d = [self.const(self.eval(x, env)) for x in dims]
            env["components"][name] = arr(d, lambda: None) if d else None
            if init is not None:
                env["components"][name] = self.make(init, env, "%s.%s" % (env["comp"].path, name))
        elif k == "var":
            for name, dims, init in node[1]:
                d = [self.const(self.eval(x, env)) for x in dims]
                v = self.eval(init, env) if init is not None else (arr(d, lambda: 0) if d else 0)
                env["vars"][name] = v
        elif k == "for":
            _, init, cond, step, body = node
            self.exec(init, env)
            while self.const(self.eval(cond, env)):
                self.exec(body, env)
                self.exec(step, env)
        elif k == "if":
            if self.const(self.eval(node[1], env)):
                self.exec(node[2], env)
            elif node[3] is not None:
                self.exec(node[3], env)
        elif k == "return":
            raise Return(self.eval(node[1], env))
        elif k == "expr":
            self.eval(node[1], env)
        elif k == "assign":
            self.assign(node, env)
        else:
            raise NotImplementedError(k)

    def make(self, init, env, path):
        assert init[0] == "call", init
        return self.instantiate(init[1], [self.const(self.eval(a, env)) for a in init[2]], path)

    def const(self, v):
        if isinstance(v, Lin):
            assert v.is_const(), "a compile-time value depends on a signal"
            v = v.c
        return int(v)

    # lvalues: ("var", name, idx) | ("sig", full name) | ("comp", name, idx)
    def lvalue(self, node, env):
        idx = []
        while node[0] == "index":
            idx.insert(0, self.const(self.eval(node[2], env)))
            node = node[1]
        if node[0] == "name":
            nm = node[1]
            if nm in env["comp"].signals:
                return ("sig", "%s.%s%s" % (env["comp"].path, nm, "".join("[%d]" % i for i in idx)))
            if nm in env["components"]:
                return ("comp", nm, idx)
            return ("var", nm, idx)
        if node[0] == "member":
            comp = self.eval_comp(node[1], env)
            return ("sig", "%s.%s%s" % (comp.path, node[2], "".join("[%d]" % i for i in idx)))
        raise NotImplementedError(node)

    def eval_comp(self, node, env):
        idx = []
        while node[0] == "index":
            idx.insert(0, self.const(self.eval(node[2], env)))
            node = node[1]
        assert node[0] == "name" and node[1] in env["components"], node
        c = env["components"][node[1]]
        for i in idx:
            c = c[i]
        assert isinstance(c, Comp), "component %s%s used before it is assigned" % (node[1], idx)
        return c

    def assign(self, node, env):
        _, op, lhs, rhs = node
        if op in ("==>", "-->"):
            lhs, rhs, op = rhs, lhs, {"==>": "<==", "-->": "<--"}[op]
        if op == "===":
            d = self.binop("-", self.eval(lhs, env), self.eval(rhs, env))
            self.constrain(d, "=== in %s" % env["comp"].path)
            return
        lv = self.lvalue(lhs, env)
        if lv[0] == "comp":
            _, name, idx = lv
            path = "%s.%s%s" % (env["comp"].path, name, "".join("[%d]" % i for i in idx))
            c = self.make(rhs, env, path)
            if idx:
                a = env["components"][name]
                for i in idx[:-1]:
                    a = a[i]
                a[idx[-1]] = c
            else:
                env["components"][name] = c
            return
        if lv[0] == "sig":
            name = lv[1]
            if op == "<--":
                self.bases.add(name)
                return
            assert op == "<==", (op, name)
            v = self.eval(rhs, env)
            if isinstance(v, Quad):
                self.bases.add(name)
                self.constrain(self.binop("-", v, Lin(0, {name: 1})), name)
            else:
                assert name not in self.forms, "%s assigned twice" % name
                self.forms[name] = Lin.of(v)
            return
        _, name, idx = lv
        v = self.eval(rhs, env)
        if op != "=":
            cur = env["vars"][name]
            for i in idx:
                cur = cur[i]
            v = self.binop(op[0], cur, v)
        if idx:
            a = env["vars"][name]
            for i in idx[:-1]:
                a = a[i]
            a[idx[-1]] = v
        else:
            env["vars"][name] = v

    def constrain(self, d, where):
        """d == 0: A*B + C = 0 is stored as A*B = -C"""
        if isinstance(d, Quad):
            assert d.a is not None, "a constraint that is not quadratic: %s" % where
            self.quads.append((d.a, d.b, d.c.scale(-1)))
        else:
            d = Lin.of(d)
            assert not d.is_const() or d.c == 0, "a constant constraint that does not hold: %s" % where
            if not d.is_const():
                self.quads.append((Lin(), Lin(), d))

    def binop(self, op, a, b):
        if isinstance(a, Quad) or isinstance(b, Quad):
            qa, qb = isinstance(a, Quad), isinstance(b, Quad)
            if (qa and a.a is None) or (qb and b.a is None) or (qa and qb):
                return QUAD
            q, o, left = (a, b, True) if qa else (b, a, False)
            o = Lin.of(o)
            if op == "+":
                return Quad(q.a, q.b, q.c.add(o))
            if op == "-":
                return Quad(q.a, q.b, q.c.add(o, -1)) if left else Quad(q.a.scale(-1), q.b, o.add(q.c, -1))
            if op == "*" and o.is_const():
                return Quad(q.a.scale(o.c), q.b, q.c.scale(o.c))
            return QUAD
        if isinstance(a, Lin) or isinstance(b, Lin):
            a, b = Lin.of(a), Lin.of(b)
            if op == "+":
                return a.add(b)
            if op == "-":
                return a.add(b, -1)
            if op == "*":
                if a.is_const():
                    return b.scale(a.c)
                if b.is_const():
                    return a.scale(b.c)
                return Quad(a, b, Lin())
            if a.is_const() and b.is_const():
                return self.binop(op, a.c, b.c)
            return QUAD   # shifts / masks / comparisons of signals: only inside <-- hints
        if op == "+":
            return a + b
        if op == "-":
            return a - b
        if op == "*":
            return a * b
        if op == "**":
            return a ** b
        if op == "/":
            return a * pow(b, P - 2, P) % P
        if op == "\\":
            return a // b
        if op == "%":
            return a % b
        if op == "<<":
            return a << b
        if op == ">>":
            return a >> b
        if op == "&":
            return a & b
        if op == "|":
            return a | b
        if op == "^":
            return a ^ b
        if op == "&&":
            return int(bool(a) and bool(b))
        if op == "||":
            return int(bool(a) or bool(b))
        return int({"==": a == b, "!=": a != b, "<": a < b, ">": a > b, "<=": a <= b, ">=": a >= b}[op])

    def eval(self, node, env):
        k = node[0]
        if k == "num":
            return node[1]
        if k == "name":
            nm = node[1]
            if nm in env["vars"]:
                return env["vars"][nm]
            if nm in env["comp"].signals:
                assert not env["comp"].signals[nm], "array signal %s used whole" % nm
                return Lin(0, {"%s.%s" % (env["comp"].path, nm): 1})
            raise NameError(nm)
        if k == "index" or k == "member":
            idx, base = [], node
            while base[0] == "index":
                idx.insert(0, self.const(self.eval(base[2], env)))
                base = base[1]
            if base[0] == "name" and base[1] in env["vars"]:
                v = env["vars"][base[1]]
                for i in idx:
                    v = v[i]
                return v
            if base[0] == "call":
                v = self.eval(base, env)
                for i in idx:
                    v = v[i]
                return v
            lv = self.lvalue(node, env)
            assert lv[0] == "sig", lv
            return Lin(0, {lv[1]: 1})
        if k == "bin":
            return self.binop(node[1], self.eval(node[2], env), self.eval(node[3], env))
        if k == "neg":
            v = self.eval(node[1], env)
            if isinstance(v, Quad):
                return QUAD if v.a is None else Quad(v.a.scale(-1), v.b, v.c.scale(-1))
            return Lin.of(v).scale(-1) if isinstance(v, Lin) else -v
        if k == "not":
            return int(not self.const(self.eval(node[1], env)))
        if k == "cond":
            return self.eval(node[2] if self.const(self.eval(node[1], env)) else node[3], env)
        if k == "list":
            return [self.eval(x, env) for x in node[1]]
        if k == "call":
            kind, params, body = self.defs[node[1]]
            assert kind == "function", node[1]
            fenv = {"vars": dict(zip(params, [self.eval(a, env) for a in node[2]])), "comp": Comp("", "", [], True), "components": {}, "is_main": False}
            try:
                self.exec(body, fenv)
            except Return as r:
                return r.v
            return 0
        raise NotImplementedError(k)


MAINS = [
    ("rollup-main", "RollupMain", [6, 16, 3, 2]),
    ("rollup-tx", "RollupTx", [16, 2]),
    ("decode-tx", "DecodeTx", [16]),
    ("fee-tx", "FeeTx", [16]),
    ("hash-inputs", "HashInputs", [16, 6, 3, 2]),
    ("withdraw", "Withdraw", [16]),
    ("hash-state", "HashState", []),
    ("decode-float", "DecodeFloat", []),
    ("compute-fee", "ComputeFee", []),
    ("fee-accumulator", "FeeAccumulator", [16]),
    ("balance-updater", "BalanceUpdater", []),
    ("rollup-tx-states", "RollupTxStates", []),
    ("rq-tx-verifier", "RqTxVerifier", []),
    ("mux256", "Mux256", []),
    ("bits-compressed-2-ay-sign", "BitsCompressed2AySign", []),
    ("ay-sign-2-ax", "AySign2Ax", []),
]


def load_defs(ref):
    defs = {}
    for root, _, files in os.walk(os.path.join(ref, "src")):
        for f in sorted(files):
            if f.endswith(".circom"):
                defs.update(Parser(tokenize(open(os.path.join(root, f)).read())).program())
    return defs


def build(defs, tmpl, args, **models):
    """the recorded system of `tmpl(args)` as `component main` (the object stored per main in declared_forms.json.gz)"""
    r = Run(defs, **models)
    r.instantiate(tmpl, args, "main")
    # outputs of black boxes and everything else a form refers to without defining it
    used = {n for f in r.forms.values() for n in f.t} | {n for q in r.quads for f in q for n in f.t}
    bases = sorted((r.bases | used) - set(r.forms))
    sgn = lambda c: str(c if c <= P // 2 else c - P)   # noqa: E731   (-1 instead of 77 digits: a third of the file)
    lin = lambda f: [sgn(f.c), [[sgn(c), m] for m, c in sorted(f.t.items())]]   # noqa: E731
    return {"template": tmpl, "args": args, "forms": {n: lin(f) for n, f in sorted(r.forms.items())},
            "quads": [[lin(a), lin(b), lin(c)] for a, b, c in r.quads], "bases": bases, "declared": r.declared}


def main():
    ref = sys.argv[1] if len(sys.argv) > 1 else "/root/reference"
    defs = load_defs(ref)
    out = {}
    for key, tmpl, args in MAINS:
        out[key] = m = build(defs, tmpl, args, sha_model=key == "withdraw", pos_model=key in ("hash-state", "decode-tx", "fee-tx", "rollup-tx", "withdraw"),
                             smt_model=key in ("fee-tx", "rollup-tx", "withdraw"), eddsa_model=key in ("rollup-tx", "ay-sign-2-ax"))
        print("%-28s %6d forms, %6d product / === constraints, %6d bases, %6d declared" % ("%s(%s)" % (tmpl, ",".join(map(str, args))), len(m["forms"]), len(m["quads"]),
                                                                                             len(m["bases"]), len(m["declared"])))
    dst = os.path.join(os.path.dirname(os.path.abspath(__file__)), "declared_forms.json.gz")
    with gzip.GzipFile(dst, "wb", mtime=0) as g:
        g.write(json.dumps(out, sort_keys=True, separators=(",", ":")).encode())
    print("-> %s (%d bytes)" % (dst, os.path.getsize(dst)))


if __name__ == "__main__":
    main()
