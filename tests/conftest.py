import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def built():
    """Build (or reuse) libfsm_hip.so + the oracle; hipcc cross-compiles without a GPU."""
    import __graft_entry__ as ge
    ge.build()
    ge.build_checker()   # the oracle and (here) oracle/_ref: test infrastructure, not part of build()
    return True
