"""A few seeds of tools/consistency_sweep.py inside the suite: medium-size random series (4e3 .. 6e4 steps, l <= 16, three kinds of time
axis, boosts up to 0.3 c) run whole, in chunks of a small work space, in the shards of sharding.plan (or as grid-column parts where the
halos call for them) and, on random windows, against the oracle.  The sweep itself has been run on 1 000 seeds of each kind
(profiles/r05_k_consistency_sweep_*.txt)."""
import importlib.util
import os

import pytest

pytestmark = pytest.mark.gpu


def _sweep():
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "consistency_sweep.py")
    spec = importlib.util.spec_from_file_location("consistency_sweep", path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


@pytest.mark.parametrize("seed", [0, 1, 3, 8, 12, 15, 90, 171])  # (types h / sigma / news / psi4, all three axes, weak and strong boosts)
def test_waveform_modes_whole_chunks_shards_oracle(ctx, seed):
    bad, what = _sweep().one(seed, ctx)
    assert not bad, (what, bad)


@pytest.mark.parametrize("seed", [0, 1, 2, 5])
def test_asymptotic_bondi_data_whole_chunks_shards_oracle(ctx, seed):
    bad, what = _sweep().one_abd(seed, ctx)
    assert not bad, (what, bad)
