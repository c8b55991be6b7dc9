"""Which signals that the reference's templates DECLARE does hz_symmap not resolve? Walks tests/golden/circom_names.json from each
main template through every component that instantiates a reference template, probes index 0 of every array, imports the list as
a .sym and prints the unresolved names grouped by (template, signal). Needs a GPU (a context)."""
import json, os, sys, collections
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
NAMES = json.load(open(os.path.join(ROOT, "tests", "golden", "circom_names.json")))


def declared_names(top, prefix="main"):
    """[(name, template, signal, kind)]: index 0 of every dimension, every reference-template instance below `top`"""
    out = []
    d = NAMES[top]
    for sig, kind in d["signals"].items():
        out.append((prefix + "." + sig + "[0]" * d["signal_dims"][sig], top, sig, kind))
    for comp, tmpl in d["components"].items():
        if tmpl in NAMES:
            out += declared_names(tmpl, prefix + "." + comp + "[0]" * d["component_dims"][comp])
    return out


if __name__ == "__main__":
    from circuits_amd import lib
    L = lib()
    mains = [("rollup-main", "RollupMain", dict(nTx=3, nLevels=8, maxL1Tx=2, maxFeeTx=2)), ("withdraw", "Withdraw", dict(nLevels=8)),
             ("rollup-tx", "RollupTx", dict(nLevels=8, maxFeeTx=2)), ("decode-tx", "DecodeTx", dict(nLevels=8)), ("fee-tx", "FeeTx", dict(nLevels=8)),
             ("hash-inputs", "HashInputs", dict(nTx=3, nLevels=8, maxL1Tx=2, maxFeeTx=2)), ("hash-state", "HashState", {})]
    for tmpl, circom, shape in mains:
        g = L.ctx(tmpl, **shape)
        names = declared_names(circom)
        lines = ["0,0,0,one"] + ["%d,%d,1,%s" % (i, i, n[0]) for i, n in enumerate(names, 1)]
        m = g.import_sym("\n".join(lines) + "\n")
        un = m.unresolved()
        by = collections.OrderedDict()
        for v, nm in un:
            _, t, s, k = names[v - 1]
            by.setdefault((t, s, k), []).append(nm)
        print("== %s: %d declared names, %d unresolved (%d distinct template signals), %d derived" % (circom, len(names), len(un), len(by), m.derived()))
        for (t, s, k), v in by.items():
            print("   %-22s %-28s %-12s e.g. %s" % (t, s, k, v[0]))
