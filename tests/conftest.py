import os
import sys

import pytest

try:  # load torch's bundled ROCm runtime first: a process must not end up with two HIP runtimes
    import torch  # noqa: F401
except Exception:  # pragma: no cover
    torch = None

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    from oracle_binding import Oracle
    return Oracle()


@pytest.fixture(scope="session")
def hz():
    """The product library. GPU tests must run the HIP path: fail loudly when it is missing."""
    from circuits_amd import lib
    L = lib()
    if L.device_count() <= 0:
        pytest.fail("libhermez_witness.so loaded but no gfx950 device is usable")
    return L


@pytest.fixture(scope="session")
def config4():
    """BASELINE config 4's batch -- RollupMain(2048, 32, 256, 64), the benchmark's own synthetic recipe -- with the oracle's complete
    witness, computed once for the GPU tests that compare against it (33 s of one core, 3.9 GB); and a SECOND batch on a state of other
    depth ("batch2" / "input2" / "oracle2": test_headline_launch_whole_buffer), whose oracle runs beside the first one's on another core."""
    import threading
    from circuits_amd import builder as B
    from oracle_binding import OracleCtx
    shape = (2048, 32, 256, 64)
    out = {"shape": shape}

    # the batches one after the other (the Python builder shares its hashing state between instances: not for two threads), the two oracle
    # runs beside each other (the library call releases the interpreter lock; the oracle keeps no state between contexts)
    for suffix, n_accounts, exits, seed in (("", 2048, 32, 0x48455A31), ("2", 4096, 7, 0x48455A32)):
        bb = B.synthetic_batch(*shape, n_accounts=n_accounts, exits=exits, seed=seed)
        inp = bb.get_input()
        o = OracleCtx("rollup-main", *shape)
        o.set_inputs(inp)
        out["batch" + suffix], out["input" + suffix], out["oracle" + suffix] = bb, inp, o

    def run(suffix):
        out["error" + suffix] = out["oracle" + suffix].run()
    t = threading.Thread(target=run, args=("2",))
    t.start()
    run("")
    t.join()
    assert out["error"] is None and out["error2"] is None
    return out
