"""tests/golden/declared_forms.json.gz (the reference's own wiring and constraints, recorded by extract_declared_forms.py) as a checker:
values for every signal name from a witness that holds only SOME of them, the constraint check, and the same system written as the
files a circom compile would hand over (.sym + .r1cs, unreduced: every signal a variable of its own)."""
import gzip
import json
import os
import struct

P = 21888242871839275222246405745257275088548364400416034343698204186575808495617
_HERE = os.path.dirname(os.path.abspath(__file__))
_DATA = None


def load(key):
    global _DATA
    if _DATA is None:
        _DATA = json.loads(gzip.open(os.path.join(_HERE, "golden", "declared_forms.json.gz")).read())
    return _DATA[key]


def load_file(path):
    """one main's system from a file of its own (tests/_generated/, written by __graft_entry__.build() where the reference is present)"""
    return json.loads(gzip.open(path).read())


def _lin(f):
    return int(f[0]) % P, [(int(c) % P, n) for c, n in f[1]]


def all_names(m):
    """every signal name of the system, in a fixed order"""
    if "_names" in m:
        return list(m["_names"])   # (a copy: callers shuffle it)
    names, seen = [], set()

    def add(n):
        if n not in seen:
            seen.add(n)
            names.append(n)
    for n in m["declared"]:
        add(n)
    for n, f in m["forms"].items():
        add(n)
        for _, t in f[1]:
            add(t)
    for q in m["quads"]:
        for f in q:
            for _, t in f[1]:
                add(t)
    for n in m["bases"]:
        add(n)
    m["_names"] = names
    return list(names)


def linear_constraints(m):
    """[(constant, [(coefficient, name)])] == 0: every form as `form - name`, and the linear === lines"""
    out = []
    for n, f in m["forms"].items():
        c, t = _lin(f)
        out.append((c, t + [(P - 1, n)]))
    for a, b, c in m["quads"]:
        if not a[1] and not b[1] and int(a[0]) * int(b[0]) % P == 0:
            cc, t = _lin(c)
            out.append((cc, t))
    return out


def _system(m):
    """the constraints in the solver's form, cached on the fixture object: (linear [(constant, [(coefficient, name)])],
    products [(A, B, C)], names of each, for products the names of A and B)"""
    if "_sys" not in m:
        lin = linear_constraints(m)
        quads = [(_lin(a), _lin(b), _lin(c)) for a, b, c in m["quads"] if a[1] and b[1]]
        names_of = [sorted({n for _, n in t}) for _, t in lin] + [sorted({n for f in q for _, n in f[1]}) for q in quads]
        ab_of = [frozenset(n for f in q[:2] for _, n in f[1]) for q in quads]
        where = {}
        for i, ns in enumerate(names_of):
            for n in ns:
                where.setdefault(n, []).append(i)
        m["_sys"] = (lin, quads, names_of, ab_of, where)
    return m["_sys"]


def solve(m, known):
    """values for every name from `known` {name: value} by propagation: a linear constraint with exactly one unknown name defines
    it; so does a product constraint A * B = C whose A and B are known and whose C holds exactly one unknown name (a product
    signal the witness does not store under that name), and one whose single unknown sits in one factor while the other factor and C are
    known (a quotient: the `x <-- a / b; x * b === a` hints of values that are constants of the circuit). Returns (values, names that
    stayed unknown)."""
    val = dict(known)
    lin, quads, names_of, ab_of, where = _system(m)
    n_lin = len(lin)
    unknown = [sum(1 for n in ns if n not in val) for ns in names_of]
    work = [i for i, u in enumerate(unknown) if u == 1]
    later = []   # lines that could define their unknown only as a quotient: tried when nothing else is left (the same signal often has a
    #              plain definition that just is not ready yet, and a quotient by a factor that happens to be 0 defines nothing)
    while work or later:
        if work:
            i, quotient_ok = work.pop(), False
        else:
            i, quotient_ok = later.pop(), True
        if unknown[i] != 1:
            continue
        u = next(n for n in names_of[i] if n not in val)
        if i < n_lin:
            c, t = lin[i]
            rest, coef = c, 0
            for k, n in t:
                if n == u:
                    coef += k
                else:
                    rest += k * val[n]
        else:
            a, b, c = quads[i - n_lin]
            in_a, in_b, in_c = (any(n == u for _, n in f[1]) for f in (a, b, c))
            if in_a + in_b + in_c != 1 and not (in_c and not in_a and not in_b):
                continue   # the unknown in more than one of A, B, C: this line cannot define it
            if in_c:
                ea = a[0] + sum(k * val[n] for k, n in a[1])
                eb = b[0] + sum(k * val[n] for k, n in b[1])
                rest, coef = c[0] - ea * eb, 0
                for k, n in c[1]:
                    if n == u:
                        coef += k
                    else:
                        rest += k * val[n]
            else:
                # a quotient: (k u + rest) * other = C with `other` and C known -- the `x <-- a / b; x * b === a` hints
                if not quotient_ok:
                    later.append(i)
                    continue
                mine, other = (a, b) if in_a else (b, a)
                eo = (other[0] + sum(k * val[n] for k, n in other[1])) % P
                if eo == 0:
                    continue   # 0 * u = C says nothing about u
                ec = (c[0] + sum(k * val[n] for k, n in c[1])) % P
                rest, coef = (mine[0] + sum(k * val[n] for k, n in mine[1] if n != u)), sum(k for k, n in mine[1] if n == u)
                rest -= ec * pow(eo, P - 2, P)
        coef %= P
        if coef == 0:
            continue
        val[u] = (-rest) * pow(coef, P - 2, P) % P
        for j in where[u]:
            unknown[j] -= 1
            if unknown[j] == 1:
                work.append(j)
    return val, [n for n in m.get("_names") or all_names(m) if n not in val]


def solve_with_hashes(m, known, poseidon):
    """solve(), then the outputs of the black boxes that are neither linear nor stored under their circom name: a Poseidon component
    (`<c>.inputs[j]` all known, `<c>.out` not) through `poseidon(list) -> int` (an implementation that is neither the oracle nor the
    device code), the Sha256 component (`<c>.in[k]` known, `<c>.out[k]` not) through hashlib; repeated until nothing changes."""
    import hashlib
    import re
    val, unk = solve(m, known)
    while unk:
        new = {}
        for n in unk:
            if n.endswith(".out") and (n[:-4] + ".inputs[0]") in val:
                c, ins, j = n[:-4], [], 0
                while "%s.inputs[%d]" % (c, j) in val:
                    ins.append(val["%s.inputs[%d]" % (c, j)])
                    j += 1
                if all("%s.inputs[%d]" % (c, k) not in unk for k in range(j)):
                    new[n] = poseidon(ins)
            mm = re.match(r"(.*)\.out\[0\]$", n)
            if mm and (mm.group(1) + ".in[0]") in val and (mm.group(1) + ".out[255]") in unk:
                c, bits, j = mm.group(1), [], 0
                while "%s.in[%d]" % (c, j) in val:
                    bits.append(val["%s.in[%d]" % (c, j)])
                    j += 1
                assert j % 8 == 0 and all(b in (0, 1) for b in bits)
                data = bytes(int("".join(map(str, bits[8 * k:8 * k + 8])), 2) for k in range(j // 8))
                d = hashlib.sha256(data).digest()
                for k in range(256):
                    new["%s.out[%d]" % (c, k)] = (d[k >> 3] >> (7 - (k & 7))) & 1
        if not new:
            break
        known = dict(val)
        known.update(new)
        val, unk = solve(m, known)
    return val, unk


def violated(m, val):
    """indices of the constraints (forms first, then quads) that `val` does not satisfy"""
    ev = lambda f: (int(f[0]) + sum(int(c) * val[n] for c, n in f[1])) % P   # noqa: E731
    bad = []
    for i, (n, f) in enumerate(m["forms"].items()):
        if ev(f) != val[n] % P:
            bad.append(("form", n))
    for i, (a, b, c) in enumerate(m["quads"]):
        if ev(a) * ev(b) % P != ev(c):
            bad.append(("quad", i))
    return bad


# ---- the same system as a compiler's files ------------------------------------------------------------------------------------------
def sym_and_r1cs(m, order=None):
    """(.sym text, .r1cs bytes, names in variable order): variable v = names[v - 1], wire 0 = the constant one. .r1cs: iden3 binary
    format version 1 (header, constraints, wire-to-label map), one constraint per form (as 0 * 0 = form - name, circom's shape for a
    linear constraint) and per product / === line."""
    names = list(order) if order is not None else all_names(m)
    var = {n: i + 1 for i, n in enumerate(names)}
    sym = "".join("%d,%d,0,%s\n" % (v, v, n) for n, v in var.items())

    def lc(c, terms):
        t = ([(0, c % P)] if c % P else []) + [(var[n], k % P) for k, n in terms if k % P]
        return struct.pack("<I", len(t)) + b"".join(struct.pack("<I", w) + k.to_bytes(32, "little") for w, k in sorted(t))
    cons = []
    for n, f in m["forms"].items():
        c, t = _lin(f)
        cons.append(lc(0, []) + lc(0, []) + lc(c, t + [(P - 1, n)]))
    for a, b, c in m["quads"]:
        cons.append(lc(*_lin(a)) + lc(*_lin(b)) + lc(*_lin(c)))
    nw = len(names) + 1
    header = struct.pack("<I", 32) + P.to_bytes(32, "little") + struct.pack("<IIIIQI", nw, 0, 0, 0, nw, len(cons))
    body = b"".join(cons)
    w2l = b"".join(struct.pack("<Q", i) for i in range(nw))
    sections = [(1, header), (2, body), (3, w2l)]
    r1cs = b"r1cs" + struct.pack("<II", 1, len(sections)) + b"".join(struct.pack("<IQ", t, len(d)) + d for t, d in sections)
    return sym, r1cs, names
