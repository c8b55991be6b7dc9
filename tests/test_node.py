"""Node.js facade (circuits_amd/node): the tester()/calculateWitness()/assertOut() surface of the
reference's suites over the N-API addon. The JS test itself is tests/node/run_facade.js."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
JS = os.path.join(ROOT, "tests", "node", "run_facade.js")
FX = os.path.join(ROOT, "tests", "golden", "node_fixture.json")


def _node(*args):
    if shutil.which("node") is None:
        pytest.skip("node is not installed")
    addon = os.path.join(ROOT, "circuits_amd", "node", "hermez_addon.node")
    if not os.path.exists(addon):
        subprocess.check_call(["make", "-C", os.path.dirname(addon)])
    return subprocess.run(["node", JS, FX] + list(args), capture_output=True, text=True, timeout=600)


def test_node_facade_loads_and_fails_loudly_without_gpu():
    r = _node("cpu")
    assert r.returncode == 0, r.stdout + r.stderr
    assert "ok" in r.stdout or "skipped" in r.stdout


@pytest.mark.gpu
def test_node_facade_rollup_main_and_hash_state():
    r = _node()
    assert r.returncode == 0, r.stdout + r.stderr
    assert "node facade: ok" in r.stdout


@pytest.mark.gpu
def test_node_facade_writes_the_wtns_of_an_unreduced_compile(tmp_path):
    """circuit.writeWtns(file, 0, symText, r1csBuffer, true): the .sym + .r1cs of the recorded RollupTx(16, 2) system
    (tests/declared_forms.py) through the N-API addon -- every variable of the compile in its order, equal to the values that follow
    from the oracle's witness; an .r1cs the witness violates is refused."""
    import copy
    import json
    import sys
    if shutil.which("node") is None:
        pytest.skip("node is not installed")
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import declared_forms as DF
    import test_declared_signals as T
    from test_witness_gpu import _parse_wtns
    from circuits_amd import builder as B
    m = DF.load("rollup-tx")
    inp = T.inputs_of("rollup-tx")[1]
    _, known = T.oracle_known("rollup-tx", m, inp)
    val, _ = DF.solve_with_hashes(m, known, lambda xs: B.host().poseidon(xs))
    sym, r1cs, names = DF.sym_and_r1cs(m)
    m2 = {k: copy.deepcopy(v) for k, v in m.items() if not k.startswith("_")}   # (without the solver's caches)
    q = [i for i, (a, b, c) in enumerate(m2["quads"]) if a[1] and b[1] and all(n in known for f in (a, b, c) for _, n in f[1])][3]
    m2["quads"][q][2][0] = str((int(m2["quads"][q][2][0]) + 1) % DF.P)
    files = {n: str(tmp_path / n) for n in ("input.json", "c.sym", "c.r1cs", "out.wtns", "bad.r1cs")}
    json.dump({k: ([str(x) for x in v] if isinstance(v, list) else str(v)) for k, v in inp.items()}, open(files["input.json"], "w"))
    open(files["c.sym"], "w").write(sym)
    open(files["c.r1cs"], "wb").write(r1cs)
    open(files["bad.r1cs"], "wb").write(DF.sym_and_r1cs(m2)[1])
    r = subprocess.run(["node", os.path.join(ROOT, "tests", "node", "wtns_r1cs.js"), "component main = RollupTx(16,2);", files["input.json"], files["c.sym"], files["c.r1cs"],
                        files["out.wtns"], files["bad.r1cs"]], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "wtns_r1cs: ok" in r.stdout, r.stdout + r.stderr
    got = _parse_wtns(files["out.wtns"])
    assert got[0] == 1 and got[1:] == [val[n] for n in names]


@pytest.mark.gpu
@pytest.mark.parametrize("transport", ["socket", "rccl"])
def test_node_sharded_batch_with_the_collectives_in_the_library(transport):
    """tests/node/shard_two.js: one RollupMain batch over two Node processes, hz_shard_step's all_gather and broadcast staged over a Unix
    socket (two ranks on this one GPU) -- and the same pass through RCCL loaded with dlopen, one rank (RCCL refuses two ranks on one
    device; the calls on the pass's stream are the ones eight ranks make). Reference: tools/helpers/actions.js:39-45."""
    if shutil.which("node") is None:
        pytest.skip("node is not installed")
    r = subprocess.run(["node", os.path.join(ROOT, "tests", "node", "shard_two.js"), FX, transport], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "rank 0: ok" in r.stdout, r.stdout + r.stderr
    if transport == "socket":
        assert "rank 1: ok" in r.stdout
