"""The witness layout's names against the reference's own declarations.

include/hz_layout.h (names + offsets) is shared by the product and the oracle, so the whole-buffer parity tests cannot see a
mis-named signal or component. tests/golden/circom_names.json records what every template of /root/reference/src declares
(tests/golden/extract_circom_names.py: signal names with their kind, component names with the template they instantiate); here
every stored signal of the layout is walked down those declarations, path element by path element, until it reaches a signal of a
reference template or leaves the reference (a circomlib component, whose sources are not in the reference repository)."""
import json
import os
import re

import pytest

from oracle_binding import OracleCtx

NAMES = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "circom_names.json")))

# circomlib templates (absent from the reference): the walk stops when it enters one of them
MAINS = [
    ("rollup-main", "RollupMain", dict(nTx=3, nLevels=8, maxL1Tx=2, maxFeeTx=2)),
    ("rollup-tx", "RollupTx", dict(nLevels=8, maxFeeTx=2)),
    ("decode-tx", "DecodeTx", dict(nLevels=8)),
    ("fee-tx", "FeeTx", dict(nLevels=8)),
    ("hash-inputs", "HashInputs", dict(nTx=3, nLevels=8, maxL1Tx=2, maxFeeTx=2)),
    ("withdraw", "Withdraw", dict(nLevels=8)),
    ("hash-state", "HashState", {}),
    # the gadget templates the reference's unit suites instantiate as `component main`
    ("decode-float", "DecodeFloat", {}),
    ("compute-fee", "ComputeFee", {}),
    ("fee-accumulator", "FeeAccumulator", dict(maxFeeTx=4)),
    ("balance-updater", "BalanceUpdater", {}),
    ("rollup-tx-states", "RollupTxStates", {}),
    ("rq-tx-verifier", "RqTxVerifier", {}),
    ("mux256", "Mux256", {}),
    ("bits-compressed-2-ay-sign", "BitsCompressed2AySign", {}),
    ("ay-sign-2-ax", "AySign2Ax", {}),
]


def walk(template, parts):
    """returns (how the walk ended, depth reached). Ends: 'signal' (a declared signal of a reference template), 'left' (entered a
    template that is not in the reference), or an error string."""
    t = template
    for d, p in enumerate(parts):
        base = re.sub(r"\[\d+\]", "", p)
        decl = NAMES[t]
        if base in decl["signals"]:
            if d != len(parts) - 1:
                return "signal %s of %s has sub-names %s" % (base, t, parts[d + 1:]), d
            return "signal", d
        if base in decl["components"]:
            sub = decl["components"][base]
            if sub not in NAMES:
                return "left", d
            t = sub
            continue
        return "%s declares neither a signal nor a component named %r" % (t, base), d
    return "path ends on component %s" % parts[-1], len(parts)


@pytest.mark.parametrize("tmpl,circom,shape", MAINS, ids=[m[0] for m in MAINS])
def test_every_stored_name_is_declared_by_the_reference(tmpl, circom, shape):
    try:
        o = OracleCtx(tmpl, **shape)
    except KeyError:
        pytest.skip("template %s is not a main of the oracle binding" % tmpl)
    names = o.symbol_names()
    assert names
    bad, ends = [], {"signal": 0, "left": 0}
    seen = set()
    for nm in names:
        assert nm.startswith("main."), nm
        key = re.sub(r"\[\d+\]", "[]", nm)
        if key in seen:
            continue
        seen.add(key)
        parts = nm[len("main."):].split(".")
        if parts == ["one"]:
            continue   # circom's constant signal 0
        if "#" in nm:
            # '#' marks a name of the layout that is not a circom label: a second copy of signals that are also stored under their
            # own name (the fee bits of DecodeTx's L1L2TxData when DecodeTx is `component main` and its whole output array is stored)
            assert nm.startswith("main.L1L2TxData#fee[") and circom == "DecodeTx", nm
            continue
        how, _ = walk(circom, parts)
        if how in ends:
            ends[how] += 1
        else:
            bad.append("%s: %s" % (nm, how))
    assert not bad, "\n".join(bad[:40])
    assert ends["signal"] > 0
    # every input and output the reference template declares is stored under its own name
    for sig, kind in NAMES[circom]["signals"].items():
        if kind in ("input", "output"):
            assert any(re.sub(r"\[\d+\]", "", n) == "main." + sig for n in names), "main.%s (%s of %s) is not in the layout" % (sig, kind, circom)


def test_intermediate_signals_of_reference_templates_are_stored_or_linear():
    """Every `signal x;` a reference template declares (an intermediate) is stored by the layout under that name wherever the
    template is instantiated inside RollupMain / Withdraw -- unless the reference's own source defines it as a wire-through or a
    linear combination with constant coefficients, which this layout leaves out (DESIGN.md 1 "Which signals are in the witness")."""
    o = OracleCtx("rollup-main", nTx=3, nLevels=8, maxL1Tx=2, maxFeeTx=2)
    w = OracleCtx("withdraw", nLevels=8)
    stored = set()
    for ctx, top in ((o, "RollupMain"), (w, "Withdraw")):
        for nm in ctx.symbol_names():
            parts = nm[len("main."):].split(".")
            t = top
            for p in parts:
                base = re.sub(r"\[\d+\]", "", p)
                if base in NAMES[t]["signals"]:
                    stored.add((t, base))
                    break
                sub = NAMES[t]["components"].get(base)
                if sub not in NAMES:
                    break
                t = sub
    reachable = set()
    def reach(t):
        if t in reachable or t not in NAMES:
            return
        reachable.add(t)
        for sub in NAMES[t]["components"].values():
            reach(sub)
    reach("RollupMain"); reach("Withdraw")
    # what the reference's own source says about each intermediate (extract_circom_names.py classify): a product of two signals or a
    # `<--` hint has to be stored under its name; a wire-through / linear combination with constant coefficients need not be
    kinds = {(t, s): k for t in reachable for s, k in NAMES[t]["intermediates"].items()}
    assert kinds and "unknown" not in kinds.values(), [k for k, v in kinds.items() if v == "unknown"]
    missing = sorted(k for k, v in kinds.items() if v in ("product", "hint") and k not in stored)
    assert not missing, missing
    # and the layout stores no reference intermediate that the source defines as linear, except as noted: none
    assert sum(1 for v in kinds.values() if v == "linear") == 16 and sum(1 for v in kinds.values() if v != "linear") == 21


def test_poseidon_signal_names_and_lookup_are_inverse():
    """every stored signal of a Poseidon component by its circom name (sigmaF[r][j] / sigmaP[k] . in2 / in4 / out) resolves to its
    position, in order; names outside the component's ranges do not resolve (the lookup computes the index from the name)"""
    o = OracleCtx("hash-state")
    t, rp = 5, 60
    names = []
    for k in range(8 * t + rp):
        for s in ("in2", "in4", "out"):
            if k < 4 * t:
                nm = ".sigmaF[%d][%d].%s" % (k // t, k % t, s)
            elif k < 4 * t + rp:
                nm = ".sigmaP[%d].%s" % (k - 4 * t, s)
            else:
                nm = ".sigmaF[%d][%d].%s" % (4 + (k - 4 * t - rp) // t, (k - 4 * t - rp) % t, s)
            names.append("main.hash" + nm)
    idx = [o.lookup(n) for n in names]
    assert idx == list(range(idx[0], idx[0] + len(names)))
    for bad in ("main.hash.sigmaF[8][0].in2", "main.hash.sigmaF[0][5].in2", "main.hash.sigmaP[60].out", "main.hash.sigmaP[1].in3", "main.hash.sigmaF[0][0].in2x",
                "main.hash.sigmaF[-1][0].in2", "main.hash.sigmaF[0].in2"):
        with pytest.raises(KeyError):
            o.lookup(bad)
