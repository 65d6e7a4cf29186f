"""Fluxes of energy, momentum, angular momentum and boost carried by a waveform h (scri/flux.py:182-798; Ruiz et al. 0707.4654
eqs. (2.8), (2.11), (2.24); Flanagan & Nichols 1510.03386 eq. (C.1) for the boost flux).

The reference writes every term as an expectation value <a|M|b> = sum conj(a_jn) M_jnlm b_lm with sparse matrices of Clebsch-Gordan
coefficients.  Every one of those operators is multiplication by an l = 1 function -- chi = cos(theta), sin(theta) e^{+-i phi}, or
eth chi / ethbar chi -- so <a|chi|b> is the integral of conj(a) b against that function, i.e. a combination of the l = 1 modes of
the product conj(a) b: ONE `bms_grid_multiply` with output_ell_max = 1 per pair (a, b) (synthesis of both factors on a grid that
resolves the product, pointwise product, analysis), no matrix-element tables; the angular-momentum operators act along the mode axis
(`bms_mode_map`) and leave an l = 0 mode to take.  Same numbers as the sparse sums to rounding (tests/test_golden.py, vectors made
by the reference's own flux.py)."""
import math

import numpy as np

from . import engine
from . import h as htype, hdot as hdottype
from .mode_algebra import LM_range

SQRT_8PI_3 = math.sqrt(8.0 * math.pi / 3.0)
SQRT_4PI_3 = math.sqrt(4.0 * math.pi / 3.0)


class _Field:
    """modes from l = 0 of a function of spin weight s at every time step (host array)"""

    def __init__(self, data, s, ell_min, ell_max, ctx):
        full = np.zeros((data.shape[0], (ell_max + 1) ** 2), dtype=complex)
        full[:, ell_min**2:] = data
        self.data, self.s, self.ell_max, self.ctx = full, s, ell_max, ctx

    def bar(self):
        """modes of the conjugate function: (-1)^(s+m) conj(f_l,-m), spin -s (one launch of bms_mode_map)"""
        LM = LM_range(0, self.ell_max)
        ell, m = LM[:, 0], LM[:, 1]
        partner = (ell * (ell + 1) - m).astype(np.int32)
        sign = np.where((self.s + m) % 2 == 0, 1.0, -1.0).astype(complex)
        out = _Field.__new__(_Field)
        out.data = engine.mode_map(self.data, partner, sign, True, ctx=self.ctx)
        out.s, out.ell_max, out.ctx = -self.s, self.ell_max, self.ctx
        return out

    def laddered(self, step):
        """eth (step = +1) or ethbar (-1), Newman-Penrose convention (scri/waveform_modes.py:478-572)"""
        LM = LM_range(0, self.ell_max)
        ell = LM[:, 0].astype(float)
        s = self.s
        f = np.where(ell >= abs(s), np.sqrt(np.maximum((ell - s * step) * (ell + s * step + 1.0), 0.0)), 0.0) * step
        out = _Field.__new__(_Field)
        out.data = engine.mode_map(self.data, np.arange(ell.size, dtype=np.int32), f.astype(complex), ctx=self.ctx)
        out.s, out.ell_max, out.ctx = s + step, self.ell_max, self.ctx
        return out


def _ell1_of_product(a, b):
    """modes (0,0), (1,-1), (1,0), (1,1) of conj(a) b, a function of spin b.s - a.s, at every time step"""
    abar = a.bar()
    L = max(a.ell_max, b.ell_max)
    return engine.grid_multiply(abar.data, abar.s, a.ell_max, b.data, b.s, b.ell_max, 2 * L, 1, ctx=a.ctx)


def _chi(a, b):
    """<a|chi|b> for chi = sin(theta) e^{+i phi}, sin(theta) e^{-i phi}, cos(theta) (the reference's p_plus, p_minus, p_z,
    scri/flux.py:213-301): integral of P Y_1m = (-1)^m P_1,-m with P = conj(a) b (spin 0)"""
    P = _ell1_of_product(a, b)
    return SQRT_8PI_3 * P[:, 1], -SQRT_8PI_3 * P[:, 3], SQRT_4PI_3 * P[:, 2]


def _eth_chi(a, b):
    """<a|eth chi|b>, a of spin s + 1, b of spin s (scri/flux.py:471-560): eth Y_1m = sqrt2 1Y_1m, and the integral of
    Q 1Y_1m = (-1)^(1+m) Q_1,-m with Q = conj(a) b (spin -1)"""
    Q = _ell1_of_product(a, b)
    r2 = math.sqrt(2.0)
    # chi_+ = -sqrt(8pi/3) Y_11, chi_- = +sqrt(8pi/3) Y_1-1, chi_z = sqrt(4pi/3) Y_10
    plus = -SQRT_8PI_3 * r2 * (+1.0) * Q[:, 1]   # m = +1: (-1)^(1+1) Q_{1,-1}
    minus = SQRT_8PI_3 * r2 * (+1.0) * Q[:, 3]   # m = -1: (-1)^(1-1) Q_{1,+1}
    z = SQRT_4PI_3 * r2 * (-1.0) * Q[:, 2]       # m = 0: (-1)^1 Q_{1,0}
    return plus, minus, z


def _ethbar_chi(a, b):
    """<a|ethbar chi|b>, a of spin s, b of spin s + 1: ethbar Y_1m = -sqrt2 -1Y_1m, and the integral of
    R -1Y_1m = (-1)^(1+m) R_1,-m with R = conj(a) b (spin +1).  (The reference's matrix element of this name, scri/flux.py:487-600,
    leaves the minus sign of ethbar Y out and subtracts the term instead: the same sum.)"""
    R = _ell1_of_product(a, b)
    r2 = -math.sqrt(2.0)
    plus = -SQRT_8PI_3 * r2 * (+1.0) * R[:, 1]
    minus = SQRT_8PI_3 * r2 * (+1.0) * R[:, 3]
    z = SQRT_4PI_3 * r2 * (-1.0) * R[:, 2]
    return plus, minus, z


def _xyz(plus, minus, z):
    """(plus, minus, z) components -> (x, y, z) as the reference does (scri/flux.py:337-341)"""
    return np.stack([0.5 * (plus.real + minus.real), 0.5 * (plus.imag - minus.imag), z.real], axis=1)


def _check(h, what):
    from .waveform_modes import WaveformModes

    if not isinstance(h, WaveformModes):
        raise ValueError(f"{what} can only be calculated from a `WaveformModes` object; this object is of type `{type(h)}`.")


def _hdot_data(h):
    if h.dataType == hdottype:
        return h.data
    if h.dataType == htype:
        return h.data_dot
    raise ValueError(f"Input argument is expected to have data of type `h` or `hdot`; this waveform data has type `{h.data_type_string}`")


def _h_and_hdot(h, hdot, what):
    from .waveform_modes import WaveformModes

    _check(h, what)
    if (hdot is not None) and (not isinstance(hdot, WaveformModes)):
        raise ValueError(f"{what} can only be calculated from a `WaveformModes` object; `hdot` is of type `{type(hdot)}`.")
    if h.dataType != htype:
        raise ValueError(f"Input argument `h` is expected to have data of type `h`; this `h` waveform data has type `{h.data_type_string}`")
    if hdot is None:
        return h.data, h.data_dot
    if hdot.dataType != hdottype:
        raise ValueError(f"Input argument `hdot` is expected to have data of type `hdot`; this `hdot` waveform data has type `{h.data_type_string}`")
    return h.data, hdot.data


def energy_flux(h):
    """Energy flux, eq. (2.8) of Ruiz et al.: sum_lm |hdot_lm|^2 / 16 pi"""
    _check(h, "Energy flux")
    hdot = _hdot_data(h)
    return np.einsum("ij, ij -> i", hdot.conjugate(), hdot).real / (16.0 * np.pi)


def momentum_flux(h):
    """Momentum flux, eq. (2.11) of Ruiz et al.: the integral of |hdot|^2 n / 16 pi"""
    _check(h, "Momentum flux")
    f = _Field(_hdot_data(h), -2, h.ell_min, h.ell_max, h._ctx)
    return _xyz(*_chi(f, f)) / (16.0 * np.pi)


def angular_momentum_flux(h, hdot=None):
    """Angular-momentum flux, eq. (2.24) of Ruiz et al.: -<hdot| J |h> / 16 pi with <j,n|J_z|l,m> = i m,
    <l,m+-1|J_+-|l,m> = i sqrt((l -+ m)(l +- m + 1))"""
    hd, hdd = _h_and_hdot(h, hdot, "Angular momentum flux")
    LM = LM_range(h.ell_min, h.ell_max)
    ell, m = LM[:, 0], LM[:, 1]
    own = np.arange(ell.size, dtype=np.int32)
    # (J h) column by column: J_z keeps the column; J_+ fills (l, m) from (l, m - 1), J_- from (l, m + 1)
    jz = engine.mode_map(hd, own, 1j * m.astype(complex), ctx=h._ctx)
    up_src = np.where(m - 1 >= -ell, own - 1, -1).astype(np.int32)
    up = engine.mode_map(hd, up_src, 1j * np.sqrt(np.maximum((ell - (m - 1)) * (ell + (m - 1) + 1.0), 0.0)).astype(complex), ctx=h._ctx)
    dn_src = np.where(m + 1 <= ell, own + 1, -1).astype(np.int32)
    dn = engine.mode_map(hd, dn_src, 1j * np.sqrt(np.maximum((ell + (m + 1)) * (ell - (m + 1) + 1.0), 0.0)).astype(complex), ctx=h._ctx)
    dot = lambda x: np.einsum("ij, ij -> i", hdd.conjugate(), x)  # noqa: E731
    return _xyz(dot(up), dot(dn), dot(jz)) / (-16.0 * np.pi)


def boost_flux(h, hdot=None):
    """Boost flux, eq. (C.1) of Flanagan & Nichols in the Newman-Penrose form of scri/flux.py:444-748:
    (-1/32 pi) { (1/8) [<ebN|chi|ebh> - <eN|chi|eh> + 6 <N|chi|h> + <ebh|chi|ebN> - <eh|chi|eN> + 6 <h|chi|N>]
                 - (u/2) <N|chi|N> - (1/4) [<eN|e chi|h> + <h|eb chi|eN>] },   N = hdot, e = eth, eb = ethbar"""
    hd, hdd = _h_and_hdot(h, hdot, "Boost fluxes")
    H = _Field(hd, -2, h.ell_min, h.ell_max, h._ctx)
    N = _Field(hdd, -2, h.ell_min, h.ell_max, h._ctx)
    eH, ebH, eN, ebN = H.laddered(+1), H.laddered(-1), N.laddered(+1), N.laddered(-1)
    terms = [np.zeros(h.n_times, dtype=complex) for _ in range(3)]

    def add(factor, values):
        for k in range(3):
            terms[k] = terms[k] + factor * values[k]

    add(1 / 8, _chi(ebN, ebH))
    add(-1 / 8, _chi(eN, eH))
    add(6 / 8, _chi(N, H))
    add(1 / 8, _chi(ebH, ebN))
    add(-1 / 8, _chi(eH, eN))
    add(6 / 8, _chi(H, N))
    nn = _chi(N, N)
    add(-0.5, tuple(h.t * v for v in nn))
    add(-1 / 4, _eth_chi(eN, H))
    add(-1 / 4, _ethbar_chi(H, eN))
    return _xyz(*terms) / (-32.0 * np.pi)


def poincare_fluxes(h, hdot=None):
    """(energy, momentum, angular-momentum, boost) flux with one time derivative for all four (scri/flux.py:750-798)"""
    from .waveform_modes import WaveformModes

    _check(h, "Poincare fluxes")
    if (hdot is not None) and (not isinstance(hdot, WaveformModes)):
        raise ValueError(f"Poincare fluxes can only be calculated from a `WaveformModes` object; `hdot` is of type `{type(hdot)}`.")
    if h.dataType != htype:
        raise ValueError(f"Input argument `h` is expected to have data of type `h`; this `h` waveform data has type `{h.data_type_string}`")
    if hdot is None:
        hdot = h.copy()
        hdot.dataType = hdottype
        hdot.data = h.data_dot
    elif hdot.dataType != hdottype:
        raise ValueError(f"Input argument `hdot` is expected to have data of type `hdot`; this `hdot` waveform data has type `{h.data_type_string}`")
    return energy_flux(hdot), momentum_flux(hdot), angular_momentum_flux(h, hdot), boost_flux(h, hdot)
