"""ORACLE (test infrastructure only -- never imported by the product): CPU restatement of
scri/asymptotic_bondi_data/bms_charges.py:14-286 on plain arrays.  Products use the oracle's grid product on a grid
that is exact for the sum of the band limits (== spherical_functions.Modes.multiply up to rounding); time derivatives
are scipy CubicSpline derivatives (modes_time_series.py:72-98).  PARITY UNPINNED at bit level; pinned by the
reference's analytic tests (tests/test_asymptoticbondidata.py:15-162): Schwarzschild, boosted Schwarzschild,
Kerr angular momentum under boosts, N = G - t P."""
import math

import numpy as np

from . import modes_time_series_ref as mref
from . import wigner

SPINS = dict(psi0=2, psi1=1, psi2=0, psi3=-1, psi4=-2, sigma=2)


def bar(a, s):
    """modes of the conjugate function: (-1)^(s+m) conj(a_{l,-m}); spin -s"""
    a = np.asarray(a, dtype=complex)
    lmax = int(round(math.sqrt(a.shape[-1]))) - 1
    out = np.empty_like(a)
    for ell in range(lmax + 1):
        for m in range(-ell, ell + 1):
            out[..., wigner.LM_index(ell, m, 0)] = (-1.0) ** (s + m) * np.conj(a[..., wigner.LM_index(ell, -m, 0)])
    return out


def real(a):
    return 0.5 * (a + bar(a, 0))


def multiply(a, sa, b, sb, out_ell_max):
    la = int(round(math.sqrt(a.shape[-1]))) - 1
    lb = int(round(math.sqrt(b.shape[-1]))) - 1
    return mref.grid_multiply(a, sa, la, b, sb, lb, la + lb, out_ell_max)


def trunc(a, ell_max):
    return a[..., : (ell_max + 1) ** 2]


def charge_vector_from_aspect(charge):
    """bms_charges.py:50-69"""
    v = np.empty(charge.shape[:-1] + (4,))
    v[..., 0] = charge[..., 0].real
    v[..., 1] = (charge[..., 1] - charge[..., 3]).real / math.sqrt(6)
    v[..., 2] = (charge[..., 1] + charge[..., 3]).imag / math.sqrt(6)
    v[..., 3] = charge[..., 2].real / math.sqrt(3)
    return v / math.sqrt(4 * math.pi)


def mass_aspect(u, psi2, sigma, ell_max_out):
    """bms_charges.py:14-47 with an integer truncation"""
    sbdot = mref.interpolate(u, bar(sigma, 2), u, 1)
    return -real(trunc(psi2, ell_max_out) + multiply(sigma, 2, sbdot, -2, ell_max_out))


def four_momentum(u, psi2, sigma):
    return charge_vector_from_aspect(mass_aspect(u, psi2, sigma, 1))


def _psi1_sigma(psi1, sigma):
    return trunc(psi1, 1) + multiply(sigma, 2, wigner.eth_GHP(bar(sigma, 2), -2), -1, 1)


def angular_momentum(psi1, sigma):
    """bms_charges.py:91-106"""
    return charge_vector_from_aspect(1j * _psi1_sigma(psi1, sigma))[:, 1:]


def com_charge(psi1, sigma):
    """bms_charges.py:185-200"""
    a = -(_psi1_sigma(psi1, sigma) + 0.5 * wigner.eth_GHP(multiply(sigma, 2, bar(sigma, 2), -2, 1), 0))
    return charge_vector_from_aspect(a)[:, 1:]


def boost_charge(u, psi1, psi2, sigma):
    """bms_charges.py:163-182"""
    sbdot = mref.interpolate(u, bar(sigma, 2), u, 1)
    mass_term = wigner.eth_GHP(real(trunc(psi2, 1) + multiply(sigma, 2, sbdot, -2, 1)), 0)
    a = -(
        _psi1_sigma(psi1, sigma)
        + 0.5 * wigner.eth_GHP(multiply(sigma, 2, bar(sigma, 2), -2, 1), 0)
        - u[:, None] * mass_term
    )
    return charge_vector_from_aspect(a)[:, 1:]


def dimensionless_spin(u, psi1, psi2, sigma):
    """bms_charges.py:139-160"""
    N = boost_charge(u, psi1, psi2, sigma)
    J = angular_momentum(psi1, sigma)
    P = four_momentum(u, psi2, sigma)
    M_sqr = (P[:, 0] ** 2 - np.sum(P[:, 1:] ** 2, axis=1))[:, None]
    v = P[:, 1:] / P[:, 0][:, None]
    vn = np.linalg.norm(v, axis=1)
    vhat = v.copy()
    idx = vn != 0
    vhat[idx] = v[idx] / vn[idx, None]
    gamma = (1 / np.sqrt(1 - vn**2))[:, None]
    Jv = np.einsum("ij,ij->i", J, vhat)[:, None]
    return (gamma * (J + np.cross(v, N)) - (gamma - 1) * Jv * vhat) / M_sqr


def supermomentum(u, psi2, sigma, definition, working_ell_max=None, integrated=False):
    """bms_charges.py:203-286"""
    lmax = int(round(math.sqrt(psi2.shape[-1]))) - 1
    W = 2 * lmax if working_ell_max is None else working_ell_max
    sb = bar(sigma, 2)
    sbdot = mref.interpolate(u, sb, u, 1)
    base = psi2 + mref.grid_multiply(sigma, 2, lmax, sbdot, -2, lmax, W, lmax)
    e2sb = wigner.eth_GHP(wigner.eth_GHP(sb, -2), -1)
    eb2s = wigner.ethbar_GHP(wigner.ethbar_GHP(sigma, 2), 1)
    d = definition.lower()
    if d in ("bondi-sachs", "bs"):
        res = base
    elif d in ("moreschi", "m"):
        res = base + e2sb
    elif d in ("geroch", "g"):
        res = base + 0.5 * (e2sb - eb2s)
    elif d in ("geroch-winicour", "gw"):
        res = base - eb2s
    else:
        raise ValueError(definition)
    return -0.5 * bar(res, 0) / math.sqrt(math.pi) if integrated else res


def cwwy_angular_momentum(u, psi1, psi2, sigma):
    """bms_charges.py:109-136"""
    lmax = int(round(math.sqrt(sigma.shape[-1]))) - 1
    fac = np.concatenate([np.full(2 * l + 1, 0.0 if l < 2 else 4.0 / ((l + 2) * (l + 1) * l * (l - 1))) for l in range(lmax + 1)])
    pot = fac * (wigner.ethbar_GHP(wigner.ethbar_GHP(sigma, 2), 1) + wigner.eth_GHP(wigner.eth_GHP(bar(sigma, 2), -2), -1))
    m_eth = wigner.eth_GHP(mass_aspect(u, psi2, sigma, lmax), 0)
    a = 1j * (_psi1_sigma(psi1, sigma) + multiply(pot, 0, m_eth, 1, 1))
    return charge_vector_from_aspect(a)[:, 1:]
