"""Poseidon parameter generation for BN254 Fr (x^5, R_F=8), test infrastructure.

circomlib 0.5.2 (package-lock.json:861-862 of the reference; NOT on disk) ships
poseidon_constants.circom, produced by the Poseidon paper's public
`generate_parameters_grain` procedure with (field=1, sbox=0, n=254, t, R_F=8, R_P(t)).
This file restates that published procedure (Grain LFSR self-shrinking generator,
rejection sampling of round constants, Cauchy MDS) in plain Python big-int
arithmetic. It is pinned by known answers of the upstream hash (see
tests/golden/gen_golden.py); nothing here is shipped in the product path.
"""
P = 21888242871839275222246405745257275088548364400416034343698204186575808495617
N_ROUNDS_F = 8
N_ROUNDS_P = [56, 57, 56, 60, 60, 63, 64, 63]  # index t-2


class Grain:
    def __init__(self, field, sbox, n, t, r_f, r_p):
        bits = []
        for val, width in ((field, 2), (sbox, 4), (n, 12), (t, 12), (r_f, 10), (r_p, 10)):
            bits += [int(c) for c in bin(val)[2:].zfill(width)]
        bits += [1] * 30
        assert len(bits) == 80
        self.s = bits
        for _ in range(160):
            self._step()

    def _step(self):
        s = self.s
        nb = s[62] ^ s[51] ^ s[38] ^ s[23] ^ s[13] ^ s[0]
        s.pop(0)
        s.append(nb)
        return nb

    def bit(self):
        # self-shrinking: take pairs, emit second bit only when first bit is 1
        while True:
            b1 = self._step()
            b2 = self._step()
            if b1 == 1:
                return b2

    def bits(self, n):
        v = 0
        for _ in range(n):
            v = (v << 1) | self.bit()
        return v


def generate(t):
    r_p = N_ROUNDS_P[t - 2]
    g = Grain(1, 0, 254, t, N_ROUNDS_F, r_p)
    C = []
    while len(C) < t * (N_ROUNDS_F + r_p):
        v = g.bits(254)
        if v < P:
            C.append(v)
    while True:
        xs_ys = [g.bits(254) % P for _ in range(2 * t)]
        if len(set(xs_ys)) != 2 * t:
            continue
        xs, ys = xs_ys[:t], xs_ys[t:]
        if any((x + y) % P == 0 for x in xs for y in ys):
            continue
        M = [[pow((xs[i] + ys[j]) % P, P - 2, P) for j in range(t)] for i in range(t)]
        return C, M


def poseidon(inputs, _cache={}):
    """circomlib 0.5.x Poseidon(nInputs): capacity element first, out = state[0]."""
    t = len(inputs) + 1
    if t not in _cache:
        _cache[t] = generate(t)
    C, M = _cache[t]
    r_p = N_ROUNDS_P[t - 2]
    st = [0] + [x % P for x in inputs]
    for r in range(N_ROUNDS_F + r_p):
        st = [(st[j] + C[t * r + j]) % P for j in range(t)]
        if r < N_ROUNDS_F // 2 or r >= N_ROUNDS_F // 2 + r_p:
            st = [pow(x, 5, P) for x in st]
        else:
            st[0] = pow(st[0], 5, P)
        st = [sum(M[i][j] * st[j] for j in range(t)) % P for i in range(t)]
    return st[0]


if __name__ == "__main__":
    C, M = generate(3)
    print(hex(C[0]), hex(M[0][0]))
    print(poseidon([1, 2]))
    print(poseidon([1]))
    print(poseidon([1, 2, 3, 4, 5, 6]))
