"""Fluxes of energy, momentum, angular momentum and boost (scri_amd/flux.py: expectation values as l = 1 modes of grid products on
the GPU) against the oracle's literal grid integrals (oracle/flux_ref.py: the reference's own "silly" test implementations,
tests/test_flux.py:14-115, and the boost-flux formula of scri/flux.py:444-470 by quadrature), and the reference's rotation test
(tests/test_flux.py:133-143) on data that actually radiates."""
import numpy as np
import pytest

from oracle import flux_ref, quat

pytestmark = pytest.mark.gpu


def _waveform(ctx, ell_max=6, n=50, seed=4):
    import scri_amd
    from scri_amd import synthetic

    t = np.linspace(2.0, 30.0, n) + 0.03 * np.cos(np.arange(n))
    data = synthetic.chirp_modes(t, 2, ell_max, seed) * (1 + 0.02 * t[:, None])
    return scri_amd.WaveformModes(t=t, data=data, ell_min=2, ell_max=ell_max, dataType=scri_amd.h, frameType=scri_amd.Inertial,
                                  r_is_scaled_out=True, m_is_scaled_out=True, ctx=ctx)


@pytest.mark.parametrize("ell_max,n", [(6, 50), (3, 20), (9, 12)])
def test_fluxes_match_the_literal_grid_integrals(ctx, ell_max, n):
    import scri_amd

    h = _waveform(ctx, ell_max, n)
    hdot = flux_ref.data_dot(h.t, h.data)
    rel = lambda a, b: np.abs(a - b).max() / np.abs(b).max()  # noqa: E731
    assert rel(h.energy_flux(), flux_ref.energy_flux(hdot)) < 1e-12
    assert rel(h.momentum_flux(), flux_ref.silly_momentum_flux(hdot, 2, ell_max)) < 1e-12
    assert rel(h.angular_momentum_flux(), flux_ref.silly_angular_momentum_flux(h.data, hdot, 2, ell_max)) < 1e-12
    assert rel(h.boost_flux(), flux_ref.boost_flux(h.t, h.data, hdot, 2, ell_max)) < 1e-11
    e, p, j, b = h.poincare_fluxes()
    assert np.array_equal(e, h.energy_flux()) and np.array_equal(p, h.momentum_flux())


def test_fluxes_rotate_as_vectors(ctx):
    """tests/test_flux.py:133-143: the boost flux of the rotated waveform is the rotated boost flux (and so for the momentum and the
    angular-momentum flux); the energy flux is a scalar.  (The reference takes a constant single mode there, whose fluxes vanish.)"""
    from scri_amd import quaternions

    h = _waveform(ctx)
    R = np.array([1.0, 4.0, 3.0, 2.0]) / np.sqrt(30.0)
    Rinv = quat.qconj(R)

    def rotated(v):  # quaternion.rotate_vectors(R, v)
        q = np.concatenate([np.zeros((v.shape[0], 1)), v], axis=1)
        return quaternions.multiply(quaternions.multiply(R, q), Rinv)[:, 1:]

    before = [h.momentum_flux(), h.angular_momentum_flux(), h.boost_flux()]
    e_before = h.energy_flux()
    g = h.copy()
    g.rotate_decomposition_basis(Rinv)
    for a, b in zip(before, (g.momentum_flux(), g.angular_momentum_flux(), g.boost_flux())):
        assert np.allclose(rotated(a), b, rtol=1e-12, atol=1e-12 * np.abs(a).max())
    assert np.allclose(e_before, g.energy_flux(), rtol=1e-13, atol=0)
    import scri_amd

    single = scri_amd.WaveformModes(t=h.t, data=np.zeros_like(h.data), ell_min=2, ell_max=6, dataType=scri_amd.h, frameType=scri_amd.Inertial, ctx=ctx)
    single.data[:, single.index(5, 3)] = 1.0
    assert np.array_equal(single.boost_flux(), np.zeros((h.n_times, 3)))
