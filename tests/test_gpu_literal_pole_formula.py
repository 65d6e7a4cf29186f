"""The GPU parity tests run the oracle with the well-conditioned form of the pole angle, 2 atan2(|(x, y)|, |(w, z)|)
(`oracle.quat.ROBUST_POLES`, tests/conftest.py), because numpy-quaternion's literal 2 acos(sqrt((w^2 + z^2)/n)) -- what the
reference's boosted grid goes through, scri/waveform_grid.py:141-161 -- loses half its digits at the grid's pole pixels
(|v| 3e-8 rad in the rotor there).  This file bounds what that choice hides: boosted transformations against the oracle with the
LITERAL formula.  Measured (tools/pole_formula_probe.py, beta = 3.7e-4, 1e-2, 0.1, with and without a frame rotation, which decides
whether the grid's poles are the rotors' poles): 0.9e-15 .. 2.1e-15 of the data's scale with either form -- the pole pixels' rotor
noise is invisible in the transformed modes.  The bar below is 2e-14."""
import numpy as np
import pytest

from oracle import abd_ref, quat, waveform_grid_ref as grid_ref
from oracle.containers import WM, h
from tests.test_gpu_transform_abd import smooth_abd

pytestmark = pytest.mark.gpu

DIRECTION = np.array([1.0, 2.0, 3.0]) / np.sqrt(14.0)
ROTATIONS = {"rotated": np.array([1.0, 2, 3, 4]) / np.sqrt(30), "unrotated": np.array([1.0, 0, 0, 0])}
BAR = 2e-14


@pytest.mark.parametrize("rot", sorted(ROTATIONS))
@pytest.mark.parametrize("beta", [3.7417e-4, 1e-2, 0.1])
def test_boosted_transform_against_the_literal_pole_formula(ctx, beta, rot):
    import scri_amd
    from scri_amd import synthetic

    quat.ROBUST_POLES = False  # (the autouse fixture of conftest.py restores its own setting afterwards)
    v = beta * DIRECTION
    n, ell_max = 200, 8
    t = np.linspace(-40.0, 60.0, n)
    data = synthetic.chirp_modes(t, 2, ell_max, 5)
    kw = dict(boost_velocity=v, supertranslation=synthetic.real_supertranslation(0.1 * np.arange(1, 10) * (1 + 0.5j)), frame_rotation=ROTATIONS[rot])
    got = scri_amd.WaveformModes(t=t, data=data, ell_min=2, ell_max=ell_max, dataType=scri_amd.h, frameType=scri_amd.Inertial,
                                 r_is_scaled_out=True, m_is_scaled_out=True, ctx=ctx).transform(**kw)
    e = grid_ref.transform(WM(t=t, data=data, ell_min=2, ell_max=ell_max, dataType=h), **kw)
    assert e.t.size == got.n_times and np.abs(got.t - e.t).max() < 1e-13
    assert np.abs(got.data - e.data).max() < BAR * max(1.0, np.abs(e.data).max())

    o = smooth_abd(160, 4, 9)
    kw["supertranslation"] = 0.05 * np.arange(1, 10) * (1 + 0.5j)  # (this flavour imposes reality itself)
    g = scri_amd.AsymptoticBondiData(o.u, o.ell_max, ctx=ctx)
    g._raw_data[:] = o.raw
    got = g.transform(**kw)
    e = abd_ref.transform(o, **kw)
    assert e.n_times == got.n_times
    assert np.abs(got._raw_data - e.raw).max() < BAR * max(1.0, np.abs(e.raw).max())
