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
