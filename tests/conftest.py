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


@pytest.fixture
def route(monkeypatch):
    """route(name, value="1"): a route option (scri_amd/csrc/env.h) for the duration of one test -- set on every live context of
    the process through bms_ctx_set_option (contexts read the environment only when they are created) AND exported, so that a
    context the test creates afterwards starts with it.  Names that are not route options (SCRI_AMD_NO_PIPELINE and the other
    Python-level switches) are plain environment variables."""
    from scri_amd import _lib

    changed = []

    def set_(name, value="1"):
        """value None: back to the built-in default (the variable removed, the option cleared)"""
        if value is None:
            monkeypatch.delenv(name, raising=False)
            value = -1 if name.endswith("AXIS_BOOST_MIN_WORK") else 0  # (its 0 means "always": the unset state is -1)
        else:
            monkeypatch.setenv(name, str(value))
        for c in _lib.live_contexts():
            try:
                changed.append((c, name, c.option(name, int(value))))
            except ValueError:
                return  # not a route option of the library: the environment variable is all there is

    yield set_
    for c, name, old in reversed(changed):
        if getattr(c, "_h", None):
            c.option(name, old)
