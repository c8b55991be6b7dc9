#!/usr/bin/env python3
"""Sparse form of the Poseidon partial rounds (product-side constants only).

The circuit (circomlib 0.5.2 poseidon.circom) evaluates every partial round as Ark (t constants),
one S-box on state[0], Mix (dense t x t MDS): t^2 constant products per round. The witness only
holds the S-box signals (in2, in4, out), so any evaluation order that reproduces the S-box INPUTS
is admissible. Two exact rewrites (both classic, cf. the Poseidon paper's appendix on optimised
implementations) bring a partial round down to 2t - 1 constant products:

1. constants: only state[0] passes through the S-box, so the constants of the other t-1 lanes are
   pushed through the linear layer into the next round: e_r = c_r + carry_r, only e_r[0] is added,
   carry_{r+1} = M * (0, e_r[1..]); what is left after the last partial round is added to the
   constants of the following full round.
2. linear layer: with N_0 = I, round r applies M * diag(1, N_r) = diag(1, Mh N_r) * S_r where
   S_r = [[m00, v^T N_r], [(Mh N_r)^-1 w, I]] is sparse (M = [[m00, v^T], [w, Mh]]), and
   N_{r+1} = Mh N_r stays pending on lanes 1..t-1 (it commutes with the S-box, which only touches
   lane 0). After the last partial round the pending N_RP is applied once (dense, (t-1)^2 products).

state[0] is never transformed, so the S-box inputs and outputs are bit-identical to the naive
evaluation; `check()` verifies this against tools/poseidon_params.py (the textbook permutation) for random inputs.
"""
import os
import random
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
from poseidon_params import P, N_ROUNDS_P, generate  # noqa: E402


def mat_mul(A, B):
    n, m, k = len(A), len(B[0]), len(B)
    return [[sum(A[i][x] * B[x][j] for x in range(k)) % P for j in range(m)] for i in range(n)]


def mat_vec(A, v):
    return [sum(a * b for a, b in zip(row, v)) % P for row in A]


def mat_inv(A):
    n = len(A)
    a = [list(r) + [1 if i == j else 0 for j in range(n)] for i, r in enumerate(A)]
    for c in range(n):
        piv = next(r for r in range(c, n) if a[r][c] % P)
        a[c], a[piv] = a[piv], a[c]
        inv = pow(a[c][c], P - 2, P)
        a[c] = [x * inv % P for x in a[c]]
        for r in range(n):
            if r != c and a[r][c]:
                f = a[r][c]
                a[r] = [(x - f * y) % P for x, y in zip(a[r], a[c])]
    return [r[n:] for r in a]


def sparse_params(t):
    """Returns dict: rp, e0[rp] (constant added to lane 0 before the S-box of partial round r),
    row[rp][t] (lane-0 row of S_r), col[rp][t-1], dense[(t-1)][(t-1)] = N_RP, and cfull: the constants of
    the first full round after the partial rounds with the leftover carry folded in."""
    C, M = generate(t)
    rp = N_ROUNDS_P[t - 2]
    m00, v, w = M[0][0], M[0][1:], [M[i][0] for i in range(1, t)]
    Mh = [M[i][1:] for i in range(1, t)]
    carry = [0] * t
    e0, row, col = [], [], []
    N = [[1 if i == j else 0 for j in range(t - 1)] for i in range(t - 1)]
    for r in range(rp):
        c = C[t * (4 + r): t * (5 + r)]
        e = [(c[j] + carry[j]) % P for j in range(t)]
        e0.append(e[0])
        carry = mat_vec(M, [0] + e[1:])
        vN = [sum(v[k] * N[k][j] for k in range(t - 1)) % P for j in range(t - 1)]
        MhN = mat_mul(Mh, N)
        col.append(mat_vec(mat_inv(MhN), w))
        row.append([m00] + vN)
        N = MhN
    cf = C[t * (4 + rp): t * (5 + rp)]
    cfull = [(cf[j] + carry[j]) % P for j in range(t)]
    return {"t": t, "rp": rp, "C": C, "M": M, "e0": e0, "row": row, "col": col, "dense": N, "cfull": cfull}


def poseidon_sparse(inputs, sp, sboxes=None):
    """Evaluation with the sparse partial rounds; appends every S-box input to `sboxes`."""
    t, rp, C, M = sp["t"], sp["rp"], sp["C"], sp["M"]
    st = [0] + [x % P for x in inputs]

    def full(st, c):
        st = [(a + b) % P for a, b in zip(st, c)]
        if sboxes is not None:
            sboxes.extend(st)
        st = [pow(x, 5, P) for x in st]
        return mat_vec(M, st)

    for r in range(4):
        st = full(st, C[t * r: t * (r + 1)])
    for r in range(rp):
        x = (st[0] + sp["e0"][r]) % P
        if sboxes is not None:
            sboxes.append(x)
        y = pow(x, 5, P)
        s0 = (sp["row"][r][0] * y + sum(a * b for a, b in zip(sp["row"][r][1:], st[1:]))) % P
        st = [s0] + [(st[1 + i] + sp["col"][r][i] * y) % P for i in range(t - 1)]
    st = [st[0]] + mat_vec(sp["dense"], st[1:])
    st = full(st, sp["cfull"])
    for r in range(5 + rp, 8 + rp):
        st = full(st, C[t * r: t * (r + 1)])
    return st[0]


def poseidon_naive(inputs, t, sboxes):
    C, M = generate(t)
    rp = N_ROUNDS_P[t - 2]
    st = [0] + [x % P for x in inputs]
    for r in range(8 + rp):
        st = [(st[j] + C[t * r + j]) % P for j in range(t)]
        if r < 4 or r >= 4 + rp:
            sboxes.extend(st)
            st = [pow(x, 5, P) for x in st]
        else:
            sboxes.append(st[0])
            st[0] = pow(st[0], 5, P)
        st = mat_vec(M, st)
    return st[0]


def check(t, n=5, seed=1):
    rng = random.Random(seed + t)
    sp = sparse_params(t)
    for _ in range(n):
        x = [rng.randrange(P) for _ in range(t - 1)]
        a, b = [], []
        assert poseidon_naive(x, t, a) == poseidon_sparse(x, sp, b)
        assert a == b, "S-box inputs differ"
    return sp


if __name__ == "__main__":
    for t in range(2, 8):
        check(t)
        print("t=%d sparse == naive (digest and every S-box input)" % t)
