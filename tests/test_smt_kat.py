"""The sparse-Merkle-tree sub-path pinned on published known answers (VERDICT r1 "What's missing" 1).

tests/golden/smt_kat.json holds the three roots iden3's go-merkletree asserts in its own TestNewTree (Poseidon build): circomlib's
tree, whose JS twin the reference calls at run time (reference test/helpers/helpers.js:128-135,151-153, test/withdraw.test.js:150).
They are driven through
  * the batch builder's tree (circuits_amd/builder.py SMT),
  * the oracle's restatement of circomlib SMTProcessor(n) / SMTVerifier(n) as `component main` (INSERT chain with oldRoot
    chaining; DELETE walking the same roots backwards; UPDATE; NOP; inclusion and both kinds of exclusion proofs),
  * the HIP path through the C ABI (-m gpu): the same mains, whose SMTProcessor runs the unchanged k_smt level-chain kernel of
    RollupTx / FeeTx (call sites reference src/rollup-tx.circom:537-570), whole witness buffer compared with the oracle.
"""
import json
import os

import pytest

from oracle_binding import OracleCtx

KAT = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "smt_kat.json")))
N = KAT["nLevels"]
INS = [(int(r["key"]), int(r["value"]), int(r["root"])) for r in KAT["inserts"]]


def _pad(sib, n=N):
    return list(sib) + [0] * (n - len(sib))


def _builder_steps():
    """the three insertions on the builder's tree: SMTProcessor inputs + the root the tree itself reaches"""
    from circuits_amd import builder as B
    t = B.SMT()
    steps = []
    for k, v, _ in INS:
        r = t.insert(k, v)
        steps.append({"oldRoot": r["oldRoot"], "siblings": _pad(r["siblings"]), "oldKey": 0 if r["isOld0"] else r["oldKey"],
                      "oldValue": 0 if r["isOld0"] else r["oldValue"], "isOld0": 1 if r["isOld0"] else 0, "newKey": k, "newValue": v,
                      "newRoot": r["newRoot"]})
    return t, steps


def processor_cases():
    """(inputs, expected newRoot) of SMTProcessor(N): every function of the processor on the published roots"""
    _, steps = _builder_steps()
    cases = []
    roots = [0] + [r for _, _, r in INS]
    for i, s in enumerate(steps):
        ins = {k: s[k] for k in ("siblings", "oldKey", "oldValue", "isOld0", "newKey", "newValue")}
        cases.append((dict(ins, oldRoot=roots[i], fnc=[1, 0]), roots[i + 1]))       # INSERT: published root -> next published root
        cases.append((dict(ins, oldRoot=roots[i + 1], fnc=[1, 1]), roots[i]))      # DELETE walks the same chain backwards
        cases.append((dict(ins, oldRoot=roots[i], fnc=[0, 0]), roots[i]))          # NOP
    return cases


def verifier_cases():
    t, _ = _builder_steps()
    root3 = INS[2][2]
    cases = []
    for k, v, _ in INS:   # inclusion of every pair under the third published root
        f = t.find(k)
        assert f["found"] and f["foundValue"] == v
        cases.append({"enabled": 1, "root": root3, "siblings": _pad(f["siblings"]), "oldKey": 0, "oldValue": 0, "isOld0": 0, "key": k, "value": v, "fnc": 0})
    f = t.find(65)        # exclusion: the path ends in another leaf (65 = 0b1000001 shares its low bits with 1 and 33)
    assert not f["found"] and not f["isOld0"]
    cases.append({"enabled": 1, "root": root3, "siblings": _pad(f["siblings"]), "oldKey": f["notFoundKey"], "oldValue": f["notFoundValue"], "isOld0": 0,
                  "key": 65, "value": 0, "fnc": 1})
    f = t.find(3)         # exclusion: the path ends in an empty subtree
    assert not f["found"] and f["isOld0"]
    cases.append({"enabled": 1, "root": root3, "siblings": _pad(f["siblings"]), "oldKey": 0, "oldValue": 0, "isOld0": 1, "key": 3, "value": 0, "fnc": 1})
    cases.append(dict(cases[0], enabled=0, root=12345))   # disabled: nothing is enforced
    return cases


def test_builder_tree_reproduces_published_roots():
    t, steps = _builder_steps()
    assert [s["newRoot"] for s in steps] == [r for _, _, r in INS]
    assert t.root == INS[2][2]


def _run_processor(make_ctx, cases):
    c = make_ctx("smt-processor", nLevels=N, n_instances=len(cases))
    for i, (inp, _) in enumerate(cases):
        c.set_inputs(inp, instance=i)
    return c


def test_oracle_smt_processor_on_published_roots():
    cases = processor_cases()
    c = _run_processor(lambda t, **kw: OracleCtx(t, **kw), cases)
    assert c.run() is None
    for i, (_, exp) in enumerate(cases):
        assert c.get("main.newRoot", i) == exp, "case %d" % i
    assert c.unwritten()[0] == 0


def test_oracle_smt_processor_update_and_failures():
    from circuits_amd import builder as B
    t, _ = _builder_steps()
    u = t.update(33, 45)
    base = {"oldRoot": INS[2][2], "siblings": _pad(u["siblings"]), "oldKey": 33, "oldValue": 44, "isOld0": 0, "newKey": 33, "newValue": 45, "fnc": [0, 1]}
    c = OracleCtx("smt-processor", nLevels=N)
    c.set_inputs(base)
    assert c.run() is None
    assert c.get("main.newRoot") == u["newRoot"] == t.root
    # a wrong old root / old value is caught at checkOldInput with "1 != 0" (the message the reference's suites match)
    for bad in (dict(base, oldRoot=INS[1][2]), dict(base, oldValue=43)):
        c = OracleCtx("smt-processor", nLevels=N)
        c.set_inputs(bad)
        f = c.run()
        assert f is not None and f[3] == "smtProcessor.checkOldInput" and (f[4], f[5]) == (1, 0)
    assert isinstance(B.SMT(), B.SMT)


def test_oracle_smt_verifier_on_published_roots():
    cases = verifier_cases()
    c = OracleCtx("smt-verifier", nLevels=N, n_instances=len(cases))
    for i, inp in enumerate(cases):
        c.set_inputs(inp, instance=i)
    assert c.run() is None
    assert c.unwritten()[0] == 0
    # inclusion of (33, 44) under the SECOND root must fail at checkRoot; claiming exclusion of a present key at keysOk
    bad = dict(cases[1], root=INS[0][2])
    c = OracleCtx("smt-verifier", nLevels=N)
    c.set_inputs(bad)
    f = c.run()
    assert f is not None and f[3] == "smtVerifier.checkRoot" and (f[4], f[5]) == (1, 0)
    bad = dict(cases[1], fnc=1, oldKey=33, oldValue=44)
    c = OracleCtx("smt-verifier", nLevels=N)
    c.set_inputs(bad)
    f = c.run()
    assert f is not None and f[3] == "smtVerifier: keysOk.out === 0"


# ---- HIP path --------------------------------------------------------------------------------------------------------------------
@pytest.mark.gpu
def test_hip_smt_processor_on_published_roots(hz):
    cases = processor_cases()
    g = _run_processor(lambda t, **kw: hz.ctx(t, **kw), cases)
    o = _run_processor(lambda t, **kw: OracleCtx(t, **kw), cases)
    g.run()
    assert o.run() is None
    for i, (_, exp) in enumerate(cases):
        assert g.get("main.newRoot", i) == exp, "case %d" % i
    assert g.read_raw_bytes() == o.read_raw_bytes()


@pytest.mark.gpu
@pytest.mark.parametrize("n_levels,n_inst", [(10, 1), (33, 200)])
def test_hip_smt_processor_random_trees(hz, n_levels, n_inst):
    """more than one wavefront of independent processors on a growing tree: insert / update, whole buffer vs the oracle"""
    import random
    from circuits_amd import builder as B
    rng = random.Random(n_levels)
    t = B.SMT()
    cases = []
    keys = []
    for i in range(n_inst):
        if keys and rng.random() < 0.3:
            k = rng.choice(keys)
            v = rng.randrange(1, 1 << 200)
            u = t.update(k, v)
            cases.append({"oldRoot": u["oldRoot"], "siblings": _pad(u["siblings"], n_levels), "oldKey": k, "oldValue": u["oldValue"], "isOld0": 0, "newKey": k,
                          "newValue": v, "fnc": [0, 1]})
        else:
            k = rng.randrange(1 << (n_levels - 1))
            while k in keys:
                k = rng.randrange(1 << (n_levels - 1))
            keys.append(k)
            v = rng.randrange(1, 1 << 200)
            r = t.insert(k, v)
            cases.append({"oldRoot": r["oldRoot"], "siblings": _pad(r["siblings"], n_levels), "oldKey": 0 if r["isOld0"] else r["oldKey"],
                          "oldValue": 0 if r["isOld0"] else r["oldValue"], "isOld0": 1 if r["isOld0"] else 0, "newKey": k, "newValue": v, "fnc": [1, 0]})
        cases[-1]["_root"] = t.root
    g = hz.ctx("smt-processor", nLevels=n_levels, n_instances=n_inst)
    o = OracleCtx("smt-processor", nLevels=n_levels, n_instances=n_inst)
    for i, cs in enumerate(cases):
        inp = {k: v for k, v in cs.items() if k != "_root"}
        g.set_inputs(inp, instance=i)
        o.set_inputs(inp, instance=i)
    g.run()
    assert o.run() is None
    for i, cs in enumerate(cases):
        assert g.get("main.newRoot", i) == cs["_root"]
    assert g.read_raw_bytes() == o.read_raw_bytes()


@pytest.mark.gpu
def test_hip_smt_verifier_on_published_roots(hz):
    cases = verifier_cases()
    g = hz.ctx("smt-verifier", nLevels=N, n_instances=len(cases))
    o = OracleCtx("smt-verifier", nLevels=N, n_instances=len(cases))
    for i, inp in enumerate(cases):
        g.set_inputs(inp, instance=i)
        o.set_inputs(inp, instance=i)
    g.run()
    assert o.run() is None
    assert g.read_raw_bytes() == o.read_raw_bytes()
    from circuits_amd.capi import ConstraintError
    for bad, name in ((dict(cases[1], root=INS[0][2]), "smtVerifier.checkRoot"), (dict(cases[1], fnc=1, oldKey=33, oldValue=44), "smtVerifier: keysOk.out === 0")):
        g = hz.ctx("smt-verifier", nLevels=N)
        o = OracleCtx("smt-verifier", nLevels=N)
        g.set_inputs(bad)
        o.set_inputs(bad)
        f = o.run()
        with pytest.raises(ConstraintError) as e:
            g.run()
        assert e.value.name == name == f[3] and (e.value.lhs, e.value.rhs) == (f[4], f[5])


@pytest.mark.gpu
def test_hip_smt_processor_failure_reports(hz):
    from circuits_amd.capi import ConstraintError
    t, _ = _builder_steps()
    u = t.update(33, 45)
    base = {"oldRoot": INS[2][2], "siblings": _pad(u["siblings"]), "oldKey": 33, "oldValue": 44, "isOld0": 0, "newKey": 33, "newValue": 45, "fnc": [0, 1]}
    for bad in (dict(base, oldRoot=INS[1][2]), dict(base, oldValue=43)):
        g = hz.ctx("smt-processor", nLevels=N)
        g.set_inputs(bad)
        with pytest.raises(ConstraintError) as e:
            g.run()
        assert e.value.name == "smtProcessor.checkOldInput" and (e.value.lhs, e.value.rhs) == (1, 0)
