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
