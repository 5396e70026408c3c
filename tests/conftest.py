import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "robust-tracking-mpc-over-lossy-networks_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # Set-up LPs in the tests go through scipy's HiGHS -- the solver the reference calls (utils_polytope.py:19) and the
    # oracle of the batched LP kernel; tests/test_lp_kernel.py switches to the kernel explicitly.
    from LinearMPCOverNetworks import polytope_lite
    polytope_lite.set_lp_backend("scipy")


@pytest.fixture(scope="session")
def oracle_lib():
    """Builds (if needed) and loads the CPU oracle -- the checker, never the thing under test."""
    from oracle import oracle
    oracle.lib()
    return oracle


@pytest.fixture(scope="session")
def hip_lib():
    """The product library.  Missing library = hard failure, never a skip or a fall-back."""
    from LinearMPCOverNetworks import _native
    _native.lib()
    return _native
