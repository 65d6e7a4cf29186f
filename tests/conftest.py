import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs an MI355X (run with `pytest -m gpu` on the GPU box)")
    config.addinivalue_line("markers", "slow: exhaustive analytic tests of the oracle (minutes)")


def pytest_addoption(parser):
    parser.addoption("--run-slow", action="store_true", help="run the exhaustive (slow) analytic tests too")


def pytest_collection_modifyitems(config, items):
    if config.getoption("--run-slow") or "slow" in (config.getoption("-m") or ""):
        return
    skip = pytest.mark.skip(reason="exhaustive variant: needs --run-slow")
    for item in items:
        if "slow" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(scope="session")
def ctx():
    """One engine context on GPU 0 for the whole session (GPU tests only)."""
    from scri_amd import _lib

    c = _lib.Context(0)
    yield c
    c.close()


@pytest.fixture(autouse=True)
def _oracle_pole_convention(request):
    """GPU parity tests compare with the oracle's well-conditioned pole-angle variant (oracle/quat.py ROBUST_POLES);
    everything else (known-answer tests mirroring the reference) uses the reference's literal formula."""
    from oracle import quat

    old = quat.ROBUST_POLES
    quat.ROBUST_POLES = request.node.get_closest_marker("gpu") is not None
    yield
    quat.ROBUST_POLES = old
