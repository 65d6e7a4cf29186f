"""Bondi-gauge initial-value construction and constraint checks of AsymptoticBondiData
(scri/asymptotic_bondi_data/from_initial_values.py:2-154, constraints.py:9-278):

    psi0' = eth psi1 + 3 sigma psi2      psi1' = eth psi2 + 2 sigma psi3      psi2' = eth psi3 + sigma psi4
    psi3 = -eth d_u sigma-bar            psi4 = -d_u^2 sigma-bar              Im psi2 = -Im(eth^2 sigma-bar + sigma d_u sigma-bar)

Products of fields run on the GPU (bms_grid_multiply on a grid that is exact for the sum of the band limits, truncated to
ell_max as the reference's multiplication_truncator=max does), time derivatives and integrals through
bms_spline_derivative; the mode-space operators are diagonal maps applied with numpy.
"""
import math

import numpy as np

from . import engine
from .mode_algebra import LM_index


# ------------------------------------------------------------------------------------------------- helpers on [n, (L+1)^2] arrays
def _lfac(L, f):
    return np.concatenate([np.full(2 * l + 1, f(l)) for l in range(L + 1)])


def _eth(a, s, L):
    """GHP eth: x sqrt((l-s)(l+s+1)/2), spin s -> s+1"""
    return a * _lfac(L, lambda l: math.sqrt((l - s) * (l + s + 1) / 2.0) if l >= abs(s) and l >= abs(s + 1) else 0.0)


def _bar(a, s, L):
    out = np.empty_like(a)
    for l in range(L + 1):
        for m in range(-l, l + 1):
            out[..., LM_index(l, m, 0)] = (-1.0) ** (s + m) * np.conj(a[..., LM_index(l, -m, 0)])
    return out


def _real(a, L):
    return 0.5 * (a + _bar(a, 0, L))


def _imag(a, L):
    return -0.5j * (a - _bar(a, 0, L))


def _mul(a, sa, b, sb, L, ctx):
    """product of two fields, truncated to ell_max (rows broadcast: [1, nm] x [n, nm] is allowed)"""
    a, b = np.atleast_2d(a), np.atleast_2d(b)
    n = max(a.shape[0], b.shape[0])
    a = np.ascontiguousarray(np.broadcast_to(a, (n, a.shape[1])))
    b = np.ascontiguousarray(np.broadcast_to(b, (n, b.shape[1])))
    return engine.grid_multiply(a, sa, L, b, sb, L, 2 * L, L, ctx=ctx)


def _zero_below_spin(a, s):
    a = np.array(a, dtype=complex)
    a[..., : s * s] = 0
    return a


def _as_row(x, nm):
    x = np.asarray(x) + 0j
    if x.ndim == 0:
        return np.full((1, nm), complex(x)) if x != 0 else np.zeros((1, nm), dtype=complex)
    if x.ndim == 1:
        return x[np.newaxis, :].astype(complex)
    return x.astype(complex)


def from_initial_values(cls, time, ell_max=8, sigma0=0.0, sigmadot0=0.0, sigmaddot0=0.0, psi2=0.0, psi1=0.0, psi0=0.0, ctx=None):
    """Bondi data from the shear and initial values of psi2, psi1, psi0.

    sigma0 of dimension 0 or 1: sigma = sigma0 + u sigmadot0 + u^2 sigmaddot0 / 2 and everything is integrated exactly as
    polynomials in u (values given at u = 0).  sigma0 of dimension 2 ([n_times, modes]): sigma as a function of time; the
    rest is integrated with cubic splines from the values at time[0]."""
    L = int(ell_max)
    nm = (L + 1) ** 2
    abd = cls(time, L, multiplication_truncator=max, ctx=ctx)
    u = abd.time
    psi2 = _zero_below_spin(_as_row(psi2, nm), 0)
    psi1 = _zero_below_spin(_as_row(psi1, nm), 1)
    psi0 = _zero_below_spin(_as_row(psi0, nm), 2)
    eth = lambda x, s: _eth(x, s, L)  # noqa: E731
    mul = lambda a, sa, b, sb: _mul(a, sa, b, sb, L, ctx)  # noqa: E731
    if np.ndim(sigma0) in (0, 1):
        s0 = _zero_below_spin(_as_row(sigma0, nm), 2)
        s1 = _zero_below_spin(_as_row(sigmadot0, nm), 2)
        s2 = _zero_below_spin(_as_row(np.asarray(sigmaddot0) / 2, nm), 2)
        poly = lambda *c: sum(cn * u[:, None] ** n for n, cn in enumerate(c))  # noqa: E731
        abd._raw_data[5] = poly(s0, s1, s2)
        p4_0 = -2 * _bar(s2, 2, L)  # psi4 = -d_u^2 sigma-bar
        abd._raw_data[4] = poly(p4_0)
        p3_0 = -eth(_bar(s1, 2, L), -2)  # psi3 = -eth d_u sigma-bar
        p3_1 = -2 * eth(_bar(s2, 2, L), -2)
        abd._raw_data[3] = poly(p3_0, p3_1)
        # psi2 = int (eth psi3 + sigma psi4) du; its imaginary part is fixed by the mass-aspect condition
        p2_0 = _real(psi2, L) - 1j * _imag(eth(eth(_bar(s0, 2, L), -2), -1) + mul(s0, 2, _bar(s1, 2, L), -2), L)
        p2_1 = mul(s0, 2, p4_0, -2) + eth(p3_0, -1)
        p2_2 = (mul(s1, 2, p4_0, -2) + eth(p3_1, -1)) / 2
        p2_3 = (1 / 3) * mul(s2, 2, p4_0, -2)
        abd._raw_data[2] = poly(p2_0, p2_1, p2_2, p2_3)
        # psi1 = int (eth psi2 + 2 sigma psi3) du
        p1_0 = psi1
        p1_1 = 2 * mul(s0, 2, p3_0, -1) + eth(p2_0, 0)
        p1_2 = mul(s0, 2, p3_1, -1) + mul(s1, 2, p3_0, -1) + eth(p2_1, 0) / 2
        p1_3 = (2 * mul(s1, 2, p3_1, -1) + 2 * mul(s2, 2, p3_0, -1) + eth(p2_2, 0)) / 3
        p1_4 = (2 * mul(s2, 2, p3_1, -1) + eth(p2_3, 0)) / 4
        abd._raw_data[1] = poly(p1_0, p1_1, p1_2, p1_3, p1_4)
        # psi0 = int (eth psi1 + 3 sigma psi2) du
        p0_0 = psi0
        p0_1 = 3 * mul(s0, 2, p2_0, 0) + eth(p1_0, 1)
        p0_2 = (3 * mul(s0, 2, p2_1, 0) + 3 * mul(s1, 2, p2_0, 0) + eth(p1_1, 1)) / 2
        p0_3 = mul(s0, 2, p2_2, 0) + mul(s1, 2, p2_1, 0) + mul(s2, 2, p2_0, 0) + eth(p1_2, 1) / 3
        p0_4 = (3 * mul(s0, 2, p2_3, 0) + 3 * mul(s1, 2, p2_2, 0) + 3 * mul(s2, 2, p2_1, 0) + eth(p1_3, 1)) / 4
        p0_5 = (3 * mul(s1, 2, p2_3, 0) + 3 * mul(s2, 2, p2_2, 0) + eth(p1_4, 1)) / 5
        p0_6 = mul(s2, 2, p2_3, 0) / 2
        abd._raw_data[0] = poly(p0_0, p0_1, p0_2, p0_3, p0_4, p0_5, p0_6)
    elif np.ndim(sigma0) == 2:
        sig = _zero_below_spin(sigma0, 2)
        if sig.shape != (u.size, nm):
            raise ValueError(f"Input `sigma0` must have shape {(u.size, nm)}; it has {sig.shape}")
        d = lambda x, k: engine.spline_derivative(u, x, u, k, ctx=ctx)  # noqa: E731
        abd._raw_data[5] = sig
        sb = _bar(sig, 2, L)
        p4 = -d(sb, 2)
        sbdot = d(sb, 1)
        p3 = -eth(sbdot, -2)
        adjust = _real(psi2, L) - 1j * _imag(eth(eth(sb[:1], -2), -1) + mul(sig[:1], 2, sbdot[:1], -2), L)
        p2 = d(eth(p3, -1) + mul(sig, 2, p4, -2), -1) + adjust
        p1 = d(eth(p2, 0) + 2 * mul(sig, 2, p3, -1), -1) + psi1
        p0 = d(eth(p1, 1) + 3 * mul(sig, 2, p2, 0), -1) + psi0
        abd._raw_data[4], abd._raw_data[3], abd._raw_data[2], abd._raw_data[1], abd._raw_data[0] = p4, p3, p2, p1, p0
    else:
        raise ValueError(f"Input `sigma0` must have 1 or 2 dimensions; it has {np.ndim(sigma0)}")
    return abd


# ------------------------------------------------------------------------------------------------- constraints (constraints.py)
def _sides(lhs, rhs, lhs_value, rhs_value):
    if lhs and rhs:
        return (lhs_value(), rhs_value())
    if lhs:
        return lhs_value()
    if rhs:
        return rhs_value()


def bianchi_0(self, lhs=True, rhs=True):
    """psi0' = eth psi1 + 3 sigma psi2"""
    return _sides(lhs, rhs, lambda: self.psi0.dot, lambda: self.psi1.eth_GHP + 3 * (self.sigma * self.psi2))


def bianchi_1(self, lhs=True, rhs=True):
    """psi1' = eth psi2 + 2 sigma psi3"""
    return _sides(lhs, rhs, lambda: self.psi1.dot, lambda: self.psi2.eth_GHP + 2 * (self.sigma * self.psi3))


def bianchi_2(self, lhs=True, rhs=True):
    """psi2' = eth psi3 + sigma psi4"""
    return _sides(lhs, rhs, lambda: self.psi2.dot, lambda: self.psi3.eth_GHP + self.sigma * self.psi4)


def constraint_3(self, lhs=True, rhs=True):
    """psi3 = -eth d_u sigma-bar"""
    return _sides(lhs, rhs, lambda: self.psi3, lambda: -self.sigma.bar.dot.eth_GHP)


def constraint_4(self, lhs=True, rhs=True):
    """psi4 = -d_u^2 sigma-bar"""
    return _sides(lhs, rhs, lambda: self.psi4, lambda: -self.sigma.bar.ddot)


def constraint_mass_aspect(self, lhs=True, rhs=True):
    """Im psi2 = -Im(eth^2 sigma-bar + sigma d_u sigma-bar)"""
    return _sides(lhs, rhs, lambda: self.psi2.imag, lambda: -(self.sigma.bar.eth_GHP.eth_GHP + self.sigma * self.sigma.bar.dot).imag)


def bondi_constraints(self, lhs=True, rhs=True):
    """The six Bondi-gauge relations as (lhs, rhs) pairs (constraints.py:9-39)"""
    return (
        self.bianchi_0(lhs, rhs), self.bianchi_1(lhs, rhs), self.bianchi_2(lhs, rhs), self.constraint_3(lhs, rhs),
        self.constraint_4(lhs, rhs), self.constraint_mass_aspect(lhs, rhs),
    )


def bondi_violations(self):
    """lhs - rhs of the six relations (constraints.py:42-66)"""
    return [lhs - rhs for (lhs, rhs) in self.bondi_constraints(True, True)]


def bondi_violation_norms(self):
    """L2 norm over the sphere of each violation, at every time (constraints.py:69-94)"""
    return [v.norm() for v in self.bondi_violations]


METHODS = (bianchi_0, bianchi_1, bianchi_2, constraint_3, constraint_4, constraint_mass_aspect, bondi_constraints)
