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


def _lin(f):
    return int(f[0]), [(int(c), n) for c, n in f[1]]


def all_names(m):
    """every signal name of the system, in a fixed order"""
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
    return names


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


def solve(m, known):
    """values for every name from `known` {name: value} by propagation: a linear constraint with exactly one unknown name defines
    it; so does a product constraint A * B = C whose A and B are known and whose C holds exactly one unknown name (a product
    signal the witness does not store under that name). Returns (values, names that stayed unknown)."""
    val = dict(known)
    lin = linear_constraints(m)
    n_lin = len(lin)
    # constraint i < n_lin: linear (constant, terms) == 0; otherwise the product line quads[i - n_lin]
    quads = [(_lin(a), _lin(b), _lin(c)) for a, b, c in m["quads"] if a[1] and b[1]]
    names_of = [[n for _, n in t] for _, t in lin] + [[n for f in q for _, n in f[1]] for q in quads]
    where = {}
    for i, ns in enumerate(names_of):
        for n in ns:
            where.setdefault(n, []).append(i)

    def ready(i):
        """the single unknown name constraint i can define, or None"""
        unk = {n for n in names_of[i] if n not in val}
        if len(unk) != 1:
            return None
        u = next(iter(unk))
        if i >= n_lin:
            a, b, _ = quads[i - n_lin]
            if any(n == u for _, n in a[1]) or any(n == u for _, n in b[1]):
                return None
        return u
    work = [i for i in range(len(names_of)) if ready(i) is not None]
    ev = lambda f: (f[0] + sum(k * val[n] for k, n in f[1])) % P   # noqa: E731
    while work:
        i = work.pop()
        u = ready(i)
        if u is None:
            continue
        if i < n_lin:
            c, t = lin[i]
            rest = (c + sum(k * val[n] for k, n in t if n != u)) % P
            coef = sum(k for k, n in t if n == u) % P
        else:
            a, b, c = quads[i - n_lin]
            rest = (c[0] + sum(k * val[n] for k, n in c[1] if n != u) - ev(a) * ev(b)) % P
            coef = sum(k for k, n in c[1] if n == u) % P
        if coef == 0:
            continue
        val[u] = (-rest) * pow(coef, P - 2, P) % P
        for j in where[u]:
            if ready(j) is not None:
                work.append(j)
    return val, [n for n in all_names(m) if n not in val]


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
