"""How fast is the .sym + .r1cs import and the constraint check? The recorded system of Withdraw(16) (complete circuit: ~440 k variables,
~437 k constraints) through hz_symmap_create_r1cs / hz_symmap_check_r1cs."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import declared_forms as DF
import test_declared_signals as T
from circuits_amd import lib
key = sys.argv[1] if len(sys.argv) > 1 else "withdraw"
m = DF.load(key)
kw = dict(zip(T.KEYS.get(key, ()), m["args"]))
g = lib().ctx(key, **kw)
g.set_inputs(T.inputs_of(key)[-1]); g.run()
sym, r1cs, names = DF.sym_and_r1cs(m)
t = time.time(); mp = g.import_sym(sym, r1cs); t_i = time.time() - t
t = time.time(); w = mp.read(); t_r = time.time() - t
t = time.time(); bad = mp.check_r1cs(); t_c = time.time() - t
nc = len(m["forms"]) + len(m["quads"])
print("%s: %d variables, %d constraints, .r1cs %.1f MB: import %.2f s (%.2f us per constraint), %d solved, %d unresolved; read all %.2f s; check %.2f s -> %s"
      % (key, len(names) + 1, nc, len(r1cs) / 1e6, t_i, t_i / nc * 1e6, mp.solved(), len(mp.unresolved()), t_r, t_c, bad))
