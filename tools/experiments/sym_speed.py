"""How fast is the .sym import? RollupMain(nTx,16,3,4): the library's own .sym (every stored signal) through hz_symmap_create, and the
read of the whole witness through the map."""
import os, sys, time, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from circuits_amd import lib, builder as B
nTx = int(sys.argv[1]) if len(sys.argv) > 1 else 64
L = lib()
g = L.ctx("rollup-main", nTx=nTx, nLevels=16, maxL1Tx=3, maxFeeTx=4)
g.set_inputs(B.synthetic_batch(nTx, 16, 3, 4, n_accounts=16, exits=1).get_input())
g.run()
p = os.path.join(tempfile.mkdtemp(), "own.sym")
t = time.time(); L._check(L.c.hz_symbols_write_sym(g.h, p.encode())); t_w = time.time() - t
lines = [ln.split(b",", 3) for ln in open(p, "rb").read().splitlines() if ln]
text = b"".join(b"%d,%d,%s,%s\n" % (i, i, f[2], f[3]) for i, f in enumerate(lines))   # consecutive variables, as a compiler numbers them
n = len(lines)
t = time.time(); m = g.import_sym(text); t_i = time.time() - t
t = time.time(); w = m.read_bytes() if hasattr(m, "read_bytes") else m.read(); t_r = time.time() - t
print("nTx %d: %d symbols, .sym %.1f MB written in %.2f s; import %.2f s (%.2f us per line); read of all variables %.2f s" % (nTx, n, len(text) / 1e6, t_w, t_i, t_i / n * 1e6, t_r))
