"""Checks the V8CHECK line of tools/microbench/mulbench (double-FMA limbs, variant 8): a0, b and the result of n dependent products of
one lane, as 52-bit limbs in hex. Expected: a0 * b^n / R^n mod p with R = 2^260. Usage: ./mulbench | python check_v8.py"""
import sys

P = 21888242871839275222246405745257275088548364400416034343698204186575808495617
ok = None
for line in sys.stdin:
    sys.stdout.write(line)
    if not line.startswith("V8CHECK"):
        continue
    f = line.split()
    n = int(f[1].split("=")[1])
    w = [int(x, 16) for x in f[2:]]
    val = lambda l: sum(v << (52 * i) for i, v in enumerate(l))   # noqa: E731
    a0, b, r = val(w[0:5]), val(w[5:10]), val(w[10:15])
    exp = a0 * pow(b, n, P) * pow(pow(2, -260, P), n, P) % P
    ok = (r % P == exp) and r < 2 * P and all(v < (1 << 52) for v in w[10:15])
    print("V8CHECK %s: result %s a0 * b^n / R^n mod p (n = %d), result < 2p: %s" % ("ok" if ok else "FAILED", "==" if r % P == exp else "!=", n, r < 2 * P))
if ok is False:
    sys.exit(1)
