"""One worker of bench.py's cpu_baseline leg: builds the sample batch, waits for the common start time, runs the CPU oracle
once and prints the seconds it took (test infrastructure; nothing here is part of the product path)."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    n_tx, L, m1, F = (int(x) for x in sys.argv[1:5])
    start = float(sys.argv[5])
    from oracle_binding import OracleCtx
    from circuits_amd import builder as B
    bb = B.synthetic_batch(n_tx, L, m1, F, n_accounts=2 * n_tx, seed=7)
    o = OracleCtx("rollup-main", n_tx, L, m1, F)
    o.set_inputs(bb.get_input())
    late = time.time() - start
    if late < 0:
        time.sleep(-late)
    t = time.perf_counter()
    r = o.run()
    dt = time.perf_counter() - t
    assert r is None, r
    print("%.4f %.2f" % (dt, max(late, 0.0)))


if __name__ == "__main__":
    main()
