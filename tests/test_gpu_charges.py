"""GPU parity of the BMS charges (scri_amd/bms_charges.py on ModesTimeSeries.multiply / .dot -> bms_grid_multiply,
bms_spline_derivative) against the oracle, and the reference's analytic tests for boosted Schwarzschild / Kerr data
(tests/test_asymptoticbondidata.py:92-162), here with the transformation itself on the GPU as well."""
import numpy as np
import pytest

from oracle import bms_charges_ref as cref
from tests.test_oracle_charges import kerr_schild_abd

pytestmark = pytest.mark.gpu


def _abd(u, raw, ell_max, ctx):
    import scri_amd

    a = scri_amd.AsymptoticBondiData(u, ell_max, ctx=ctx)
    a._raw_data[:] = raw
    return a


def _random_abd(n, ell_max, seed):
    rng = np.random.default_rng(seed)
    u = np.linspace(-5.0, 20.0, n)
    nm = (ell_max + 1) ** 2
    m = np.concatenate([np.arange(-l, l + 1) for l in range(ell_max + 1)])
    raw = np.zeros((6, n, nm), dtype=complex)
    ph = 0.11 * u + 2e-3 * u**2
    for f, s in enumerate((2, 1, 0, -1, -2, 2)):
        a = (rng.normal(size=nm) + 1j * rng.normal(size=nm)) * 0.3
        a[: s * s] = 0
        raw[f] = a[None, :] * np.exp(1j * m[None, :] * ph[:, None]) * (1 + 0.02 * u[:, None])
    raw[2, :, 0] -= 5.0 * np.sqrt(4 * np.pi)  # a dominant mass monopole keeps the four-momentum timelike
    return u, raw


def test_charges_match_oracle_on_generic_data(ctx):
    u, raw = _random_abd(160, 4, 17)
    a = _abd(u, raw, 4, ctx)
    psi1, psi2, sigma = raw[1], raw[2], raw[5]
    tol = 5e-12
    assert np.abs(a.mass_aspect(1).ndarray - cref.mass_aspect(u, psi2, sigma, 1)).max() < tol
    assert np.abs(a.mass_aspect().ndarray - cref.mass_aspect(u, psi2, sigma, 4)).max() < tol  # default truncator: max
    # a false truncate_ell leaves the band limit to the series' own multiplication_truncator (the reference's plain
    # `sigma * sigma.bar.dot`, bms_charges.py:46): `sum` for a bare AsymptoticBondiData, `max` for those the file readers,
    # from_initial_values and map_to_superrest_frame build -- there the result keeps ell_max = 4, not 8
    full = a.mass_aspect(None)
    psi2_wide = np.pad(psi2, ((0, 0), (0, 81 - psi2.shape[1])))  # (psi2 carries no modes beyond l = 4)
    assert full.ell_max == 8 and np.abs(full.ndarray - cref.mass_aspect(u, psi2_wide, sigma, 8)).max() < tol
    import scri_amd

    a_max = scri_amd.AsymptoticBondiData(u, 4, multiplication_truncator=max, ctx=ctx)
    a_max._raw_data[:] = raw
    for falsy in (None, 0, False):
        kept = a_max.mass_aspect(falsy)
        assert kept.ell_max == 4 and np.abs(kept.ndarray - cref.mass_aspect(u, psi2, sigma, 4)).max() < tol, falsy
    assert np.abs(a.bondi_four_momentum() - cref.four_momentum(u, psi2, sigma)).max() < tol
    assert np.abs(a.bondi_angular_momentum() - cref.angular_momentum(psi1, sigma)).max() < tol
    assert np.abs(a.bondi_CoM_charge() - cref.com_charge(psi1, sigma)).max() < tol
    assert np.abs(a.bondi_boost_charge() - cref.boost_charge(u, psi1, psi2, sigma)).max() < 20 * tol  # x |u| <= 20
    assert np.abs(a.bondi_dimensionless_spin() - cref.dimensionless_spin(u, psi1, psi2, sigma)).max() < 1e-9
    assert np.abs(a.CWWY_angular_momentum() - cref.cwwy_angular_momentum(u, psi1, psi2, sigma)).max() < tol
    P = a.bondi_four_momentum()
    assert np.allclose(a.bondi_rest_mass() ** 2, P[:, 0] ** 2 - (P[:, 1:] ** 2).sum(axis=1), rtol=1e-14)
    for name in ("Bondi-Sachs", "M", "geroch", "GW"):
        got = a.supermomentum(name)
        assert (got.spin_weight, got.ell_max) == (0, 4)
        assert np.abs(got.ndarray - cref.supermomentum(u, psi2, sigma, name)).max() < tol, name
    gi = a.supermomentum("Moreschi", integrated=True, working_ell_max=6)
    assert np.abs(gi.ndarray - cref.supermomentum(u, psi2, sigma, "m", working_ell_max=6, integrated=True)).max() < tol
    with pytest.raises(ValueError, match="not recognized"):
        a.supermomentum("Bondi")


def test_schwarzschild_and_its_boosts(ctx):
    # reference tests test_abd_schwarzschild (:15-30) and test_abd_schwarzschild_transform (:92-116)
    mass, ell_max = 1.0, 8
    u = np.linspace(0, 100, 600)
    a = _abd(u, kerr_schild_abd(mass, 0.0, ell_max, u), ell_max, ctx)
    assert np.allclose(a.bondi_four_momentum(), [mass, 0, 0, 0], atol=1e-14, rtol=1e-14)
    assert np.allclose(a.bondi_angular_momentum(), 0, atol=1e-14)
    rest_mass = a.bondi_rest_mass()
    for v in (np.array([0.1, 0.0, 0.0]), np.array([0.0, 0.1, 0.0]), np.array([0.0, 0.0, 0.1])):
        gamma = 1 / np.sqrt(1 - v @ v)
        ap = a.transform(boost_velocity=v)
        assert np.allclose(ap.bondi_four_momentum(), mass * gamma * np.array([1, *-v]), atol=1e-14, rtol=1e-14)
        assert np.allclose(ap.bondi_rest_mass(), rest_mass[0], atol=1e-14, rtol=1e-14)


def test_kerr_angular_momentum_under_boosts(ctx):
    # reference test_abd_bondi_angular_momentum (:119-137) and test_abd_kerr (:139-162)
    mass, spin, ell_max = 1.0, 0.456, 8
    u = np.linspace(0, 100, 600)
    a = _abd(u, kerr_schild_abd(mass, spin, ell_max, u), ell_max, ctx)
    J = a.bondi_angular_momentum()[0]
    for v in (np.array([0.1, 0.0, 0.0]), np.array([0.0, 0.1, 0.0]), np.array([0.1, 0.1, 0.1]), np.array([0.0, 0.0, 0.1])):
        beta = np.linalg.norm(v)
        gamma = 1 / np.sqrt(1 - beta**2)
        ap = a.transform(boost_velocity=v)
        expect = gamma * J + (1 - gamma) * np.dot(J, v / beta) * (v / beta)
        assert np.allclose(ap.bondi_angular_momentum(), expect, atol=2e-14, rtol=2e-14)
    a2 = _abd(u, kerr_schild_abd(2.0, spin, ell_max, u), ell_max, ctx)
    S = a2.bondi_dimensionless_spin()
    assert np.allclose(S * 2.0**2, a2.bondi_angular_momentum(), atol=1e-14, rtol=1e-14)
    ap = a2.transform(boost_velocity=np.array([0.085, -0.034, 0.1]))
    assert np.allclose(ap.bondi_dimensionless_spin()[-1], S[-1], atol=3e-14, rtol=3e-14)
    N, G, P = ap.bondi_boost_charge(), ap.bondi_CoM_charge(), ap.bondi_four_momentum()
    assert np.allclose(N, G - ap.t[:, np.newaxis] * P[:, 1:], atol=1e-12, rtol=1e-13)
