import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    """CPU oracle (test infrastructure, oracle/ising_oracle.c)."""
    from oracle import oracle as O
    O.lib()
    return O


@pytest.fixture(scope="session")
def exact():
    from oracle import exact as X
    return X


@pytest.fixture(scope="session")
def capi():
    """C-ABI binding; the GPU tests call the HIP library through it."""
    from pyisingmontecarlo_amd import _capi
    _capi.lib()
    return _capi
