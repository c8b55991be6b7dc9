"""Poseidon (SURVEY 8 row a1 / kernel K1): oracle vs golden vectors (CPU), HIP vs oracle (GPU)."""
import json
import os
import random

import pytest

P = 21888242871839275222246405745257275088548364400416034343698204186575808495617
GOLD = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "poseidon_kat.json")))
NSBOX = {t: 8 * t + [56, 57, 56, 60, 60, 63][t - 2] for t in range(2, 8)}


def test_oracle_matches_upstream_known_answers(oracle):
    for v in GOLD["upstream_kat"]:
        inp = [int(x) for x in v["in"]]
        out, _ = oracle.poseidon_batch(len(inp) + 1, [inp])
        assert str(out[0]) == v["out"], inp


def test_oracle_matches_pyref_vectors(oracle):
    for v in GOLD["pyref_vectors"]:
        inp = [int(x) for x in v["in"]]
        out, _ = oracle.poseidon_batch(len(inp) + 1, [inp])
        assert str(out[0]) == v["out"]


def test_oracle_sbox_witness_is_consistent(oracle):
    # every S-box triple must satisfy in4 == in2^2 and the last S-box/out relation of Sigma()
    rng = random.Random(7)
    for t in (3, 5):
        inp = [[rng.randrange(P) for _ in range(t - 1)] for _ in range(3)]
        out, wit = oracle.poseidon_batch(t, inp, witness=True)
        n = len(inp)
        sig = [int.from_bytes(wit[i * 32:(i + 1) * 32], "little") for i in range(3 * NSBOX[t] * n)]
        for k in range(NSBOX[t]):
            for i in range(n):
                in2, in4, o = sig[(3 * k) * n + i], sig[(3 * k + 1) * n + i], sig[(3 * k + 2) * n + i]
                assert in4 == in2 * in2 % P
                # o == in4 * x with x^2 == in2
                x = o * pow(in4, P - 2, P) % P if in4 else 0
                assert x * x % P == in2


def _cases(t, n, seed):
    rng = random.Random(seed)
    rows = [[rng.randrange(P) for _ in range(t - 1)] for _ in range(n)]
    if n >= 4:
        rows[0] = [0] * (t - 1)
        rows[1] = [P - 1] * (t - 1)
        rows[2] = [1] * (t - 1)
        rows[3] = [(1 << 253) + 5] * (t - 1)
    return rows


@pytest.mark.gpu
@pytest.mark.parametrize("t", [2, 3, 4, 5, 6, 7])
def test_hip_poseidon_digest_bit_exact(hz, oracle, t):
    for n in (1, 63, 64, 65, 1000):  # ragged wavefront tails included
        rows = _cases(t, n, 100 * t + n)
        got, _ = hz.poseidon_batch(t, rows)
        exp, _ = oracle.poseidon_batch(t, rows)
        assert got == exp, (t, n)


@pytest.mark.gpu
def test_hip_poseidon_golden(hz):
    for v in GOLD["upstream_kat"] + GOLD["pyref_vectors"]:
        inp = [int(x) for x in v["in"]]
        got, _ = hz.poseidon_batch(len(inp) + 1, [inp])
        assert str(got[0]) == v["out"]


@pytest.mark.gpu
@pytest.mark.parametrize("t", [2, 3, 4, 5, 6, 7])
def test_hip_poseidon_sbox_witness_bit_exact(hz, oracle, t):
    rows = _cases(t, 130, 31 * t)
    got, gw = hz.poseidon_batch(t, rows, witness=True)
    exp, ew = oracle.poseidon_batch(t, rows, witness=True)
    assert got == exp
    assert gw == ew


@pytest.mark.gpu
@pytest.mark.parametrize("t", [2, 3, 4, 5, 6, 7])
def test_hip_poseidon_at_the_top_of_the_lazy_ranges(hz, oracle, t):
    """tools/range_check.py proves the bounds of the lazily reduced lanes (they peak at 7.03 p against the 8 p the product takes);
    this feeds the batch kernel the inputs that sit highest in those ranges -- every input p - 1, p - 2, (p - 1) / 2 +- 1, 2^253 and
    friends, on whole wavefronts so that the wave-uniform rare-subtraction branch (fr_cond_sub_p_rare) is taken and skipped -- with
    and without the S-box witness, all widths, against the oracle's dense evaluation."""
    tops = [P - 1, P - 2, P - 3, (P - 1) // 2, (P + 1) // 2, (1 << 253) - 1, 1 << 253, (1 << 253) + 1, P - (1 << 29), P - (1 << 232), (1 << 232) - 1, 1, 0]
    rows = [[v] * (t - 1) for v in tops for _ in range(64)]                      # whole wavefronts of one extreme value
    rows += [[tops[(i + j) % len(tops)] for j in range(t - 1)] for i in range(128)]   # mixed lanes
    got, gw = hz.poseidon_batch(t, rows, witness=True)
    exp, ew = oracle.poseidon_batch(t, rows, witness=True)
    assert got == exp and gw == ew
    assert hz.poseidon_batch(t, rows)[0] == exp


@pytest.mark.gpu
def test_hip_poseidon_empty_and_bad_input(hz):
    from circuits_amd import HzError
    assert hz.poseidon_batch(3, [])[0] == []
    import ctypes
    bad = (P).to_bytes(32, "little") + (1).to_bytes(32, "little")  # element == r is not canonical
    out = ctypes.create_string_buffer(32)
    st = hz.c.hz_poseidon_batch(0, 3, 1, bad, out, None)
    assert st == 4
    with pytest.raises(HzError):
        hz._check(hz.c.hz_poseidon_batch(0, 9, 1, bad, out, None))


@pytest.mark.gpu
def test_hip_field_ops_against_python_bigints(hz):
    """K0 (SURVEY 8a'): fr.h on the device, 2^16 random and edge operands per operation, against Python integers."""
    import random
    rng = random.Random(5)
    Pm = 21888242871839275222246405745257275088548364400416034343698204186575808495617
    edge = [0, 1, 2, Pm - 1, Pm - 2, (Pm - 1) // 2, (Pm + 1) // 2, 1 << 253, (1 << 253) - 1, (1 << 29) - 1, 1 << 29, (1 << 232) - 1]
    n = 1 << 16
    a = [edge[i % len(edge)] if i < 200 else rng.randrange(Pm) for i in range(n)]
    b = [edge[(i // len(edge)) % len(edge)] if i < 200 else rng.randrange(Pm) for i in range(n)]
    ref = {
        0: lambda x, y: (x + y) % Pm, 1: lambda x, y: (x - y) % Pm, 2: lambda x, y: x * y % Pm, 3: lambda x, y: x * x % Pm,
        4: lambda x, y: pow(x, Pm - 2, Pm), 5: lambda x, y: (x * y + x + y) % Pm, 6: lambda x, y: (2 * x * (Pm - y)) % Pm,
    }
    for op, f in ref.items():
        m = n if op != 4 else 4096
        got = hz.fr_ops(op, a[:m], b[:m])
        assert got == [f(x, y) for x, y in zip(a[:m], b[:m])], "field op %d" % op
    # op 7, the square root of circomlib's pointbits.circom (AySign2Ax): the root <= (r-1)/2, 0 for a non-residue
    m = 2048
    sq = [x * x % Pm for x in a[:m // 2]] + a[m // 2:m]
    got = hz.fr_ops(7, sq, None)
    for x, r in zip(sq, got):
        if pow(x, (Pm - 1) // 2, Pm) in (0, 1):
            assert r * r % Pm == x and r <= (Pm - 1) // 2, x
        else:
            assert r == 0, x


def test_canonical_sbox_form_matches_the_montgomery_form(tmp_path):
    """poseidon.h has two evaluations: digest only (Montgomery S-box, constant block K) and witness (S-box products straight in
    canonical form, constant block KW). Host build of both over the product headers: same digests, same S-box signals, t = 2..7."""
    import subprocess
    src = os.path.join(os.path.dirname(__file__), "native", "poseidon_canon_check.cpp")
    exe = str(tmp_path / "canon_check")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-Wno-unknown-pragmas", src, "-o", exe])
    r = subprocess.run([exe], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    assert r.stdout.count("mismatches=0") == 6


def test_inversion_on_the_host(tmp_path):
    """fr_inv (division steps, several per iteration) over the product header on the host: a * inv(a) = 1 on 20 000 operands (random,
    powers of two, p - 1), inv(0) = 0, and the Fermat inverse gives the same canonical value."""
    import subprocess
    src = os.path.join(os.path.dirname(__file__), "native", "inv_check.cpp")
    exe = str(tmp_path / "inv_check")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-Wno-unknown-pragmas", src, "-o", exe])
    r = subprocess.run([exe], capture_output=True, text=True)
    assert r.returncode == 0 and "mismatches=0" in r.stdout, r.stdout + r.stderr


def test_lazily_reduced_sums_on_the_host(tmp_path):
    """fr.h's lazily reduced sums (the signature ladder: differences that only feed products skip the conditional subtraction, chains
    of differences that are stored take one reduction) against the plain routines over the product header on the host: same field
    elements, results normalised and below 2p, on the edges of the operand ranges and on random operands."""
    import subprocess
    src = os.path.join(os.path.dirname(__file__), "native", "lazy_sums_check.cpp")
    exe = str(tmp_path / "lazy_sums_check")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-Wno-unknown-pragmas", src, "-o", exe])
    r = subprocess.run([exe], capture_output=True, text=True)
    assert r.returncode == 0 and "mismatches=0" in r.stdout, r.stdout + r.stderr
