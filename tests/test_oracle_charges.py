"""Pins of the oracle's BMS-charge restatement (oracle/bms_charges_ref.py) against the reference's analytic tests for
stationary data (tests/test_asymptoticbondidata.py:15-30 Schwarzschild; :139-162 Kerr in its rest frame)."""
import numpy as np

from oracle import bms_charges_ref as cref
from oracle import wigner


def kerr_schild_abd(mass, spin, ell_max, u):
    """tests/conftest.py:37-47 (Moreschi-Boyle convention) evolved with sigma = 0
    (from_initial_values.py: psi2, psi1 constant, psi0 = psi0(0) + u eth psi1)."""
    nm = (ell_max + 1) ** 2
    raw = np.zeros((6, u.size, nm), dtype=complex)
    raw[2, :, 0] = -wigner.constant_as_ell_0_mode(mass)
    raw[1, :, 2] = -np.sqrt(2) * (3j * spin / 2) * np.sqrt((8 / 3) * np.pi)
    psi0_0 = np.zeros(nm, dtype=complex)
    psi0_0[6] = 2 * (3 * spin**2 / mass / 2) * np.sqrt((32 / 15) * np.pi)
    raw[0] = psi0_0[None, :] + u[:, None] * wigner.eth_GHP(raw[1, 0], 1)[None, :]
    return raw


def test_schwarzschild_charges():
    u = np.linspace(0, 100, 50)
    raw = kerr_schild_abd(0.789, 0.0, 4, u)
    P = cref.four_momentum(u, raw[2], raw[5])
    assert np.allclose(P, [0.789, 0, 0, 0], atol=1e-14, rtol=1e-14)
    assert np.allclose(cref.angular_momentum(raw[1], raw[5]), 0, atol=1e-14)


def test_kerr_rest_frame_charges():
    mass, spin = 2.0, 0.456
    u = np.linspace(0, 100, 40)
    raw = kerr_schild_abd(mass, spin, 4, u)
    J = cref.angular_momentum(raw[1], raw[5])
    assert np.allclose(J, [0, 0, spin], atol=1e-14)  # the `spin` of conftest.kerr_schild is the angular momentum itself
    S = cref.dimensionless_spin(u, raw[1], raw[2], raw[5])
    assert np.allclose(S * mass**2, J, atol=1e-14, rtol=1e-14)  # centre-of-momentum frame (reference :147-153)
    N = cref.boost_charge(u, raw[1], raw[2], raw[5])
    G = cref.com_charge(raw[1], raw[5])
    P = cref.four_momentum(u, raw[2], raw[5])
    assert np.allclose(N, G - u[:, None] * P[:, 1:], atol=1e-14)


def test_supermomentum_definitions_agree_without_shear():
    u = np.linspace(0, 10, 12)
    raw = kerr_schild_abd(1.3, 0.2, 3, u)
    base = cref.supermomentum(u, raw[2], raw[5], "BS")
    for d in ("Moreschi", "G", "gw"):
        assert np.array_equal(cref.supermomentum(u, raw[2], raw[5], d), base)
    integ = cref.supermomentum(u, raw[2], raw[5], "M", integrated=True)
    assert np.allclose(integ[:, 0], 1.3, atol=1e-15)  # P_00 = M for Schwarzschild-like psi2
