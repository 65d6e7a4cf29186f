"""ORACLE (test infrastructure only -- never imported by the product): the fluxes of scri/flux.py computed the literal way, on a
grid: the "silly" momentum and angular-momentum fluxes of the reference's own tests/test_flux.py:14-115 restated on the oracle's
salm2map / map2salm, the energy flux (scri/flux.py:182-211), and the boost flux by brute-force quadrature of the formula in the
docstring of scri/flux.py:444-470 (every <a|chi|b> as a pointwise integral over the sphere).  Pinned by vectors computed with the
reference's own flux.py (tests/golden/g14_ref_fluxes.npz, tests/test_golden.py)."""
import numpy as np
from scipy.interpolate import CubicSpline

from . import spinsfast_ref as sfr
from . import wigner


def data_dot(t, data):
    return CubicSpline(t, data).derivative()(t)


def _full(data, ell_min, ell_max):
    out = np.zeros((data.shape[0], (ell_max + 1) ** 2), dtype=complex)
    out[:, ell_min**2 :] = data
    return out


def _sphere_integral(values):
    """integral over the sphere of spin-0 values on the grid: 2 sqrt(pi) times their (0, 0) mode (test_flux.py:47-52)"""
    return sfr.map2salm(values, 0, 0)[..., 0] * (2 * np.sqrt(np.pi))


def _axes(n):
    theta = np.linspace(0.0, np.pi, num=n, endpoint=True)
    phi = np.linspace(0.0, 2 * np.pi, num=n, endpoint=False)
    return np.outer(np.sin(theta), np.cos(phi)), np.outer(np.sin(theta), np.sin(phi)), np.outer(np.cos(theta), np.ones_like(phi))


def energy_flux(hdot):
    return np.einsum("ij, ij -> i", hdot.conjugate(), hdot).real / (16.0 * np.pi)


def silly_momentum_flux(hdot, ell_min, ell_max, s=-2):
    """tests/test_flux.py:14-55: |hdot|^2 n / 16 pi integrated over the sphere"""
    L = 2 * ell_max + 1
    n = 2 * L + 1
    m = sfr.salm2map(_full(hdot, ell_min, ell_max), s, ell_max, n, n)
    mag = m * m.conjugate()
    return np.array([_sphere_integral(mag * a / (16 * np.pi)).real for a in _axes(n)]).T


def silly_angular_momentum_flux(h, hdot, ell_min, ell_max, s=-2):
    """tests/test_flux.py:57-115: -Re integral of conj(J_i h) hdot / 16 pi, J_+- and J_z applied to the modes"""
    L = 2 * ell_max
    n = 2 * L + 1
    hdot_map = sfr.salm2map(_full(hdot, ell_min, ell_max), s, ell_max, n, n)
    idx = lambda ell, m: wigner.LM_index(ell, m, 0)  # noqa: E731
    H = _full(h, ell_min, ell_max)
    up, dn, jz = np.zeros_like(H), np.zeros_like(H), np.zeros_like(H)
    for ell in range(ell_min, ell_max + 1):
        for m in range(-ell, ell + 1):
            if m + 1 <= ell:
                up[:, idx(ell, m + 1)] = 1.0j * np.sqrt((ell - m) * (ell + m + 1)) * H[:, idx(ell, m)]
            if m - 1 >= -ell:
                dn[:, idx(ell, m - 1)] = 1.0j * np.sqrt((ell + m) * (ell - m + 1)) * H[:, idx(ell, m)]
            jz[:, idx(ell, m)] = 1.0j * m * H[:, idx(ell, m)]
    up_map, dn_map, jz_map = (sfr.salm2map(x, s, ell_max, n, n) for x in (up, dn, jz))
    jx, jy = 0.5 * (up_map + dn_map), -0.5j * (up_map - dn_map)
    return np.array([-_sphere_integral(j.conjugate() * hdot_map).real / (16 * np.pi) for j in (jx, jy, jz_map)]).T


def boost_flux(t, h, hdot, ell_min, ell_max):
    """The docstring formula of scri/flux.py:444-470, every bracket a pointwise integral:
    (-1/32 pi) { (1/8) [<ebN|chi|ebh> - <eN|chi|eh> + <ebh|chi|ebN> - <eh|chi|eN> + 6 <N|chi|h> + 6 <h|chi|N>]
                 - (1/4) [<eN|eth chi|h> + <h|ethbar chi|eN>] - (u/2) <N|chi|N> }"""
    L = ell_max
    n = 2 * (2 * L + 1) + 1  # resolves a product of three functions of band limits L, L, 1
    H, N = _full(h, ell_min, L), _full(hdot, ell_min, L)
    fields = {"h": (H, -2), "N": (N, -2), "eh": (wigner.eth_NP(H, -2), -1), "ebh": (wigner.ethbar_NP(H, -2), -3),
              "eN": (wigner.eth_NP(N, -2), -1), "ebN": (wigner.ethbar_NP(N, -2), -3)}
    maps = {k: sfr.salm2map(v, s, L, n, n) for k, (v, s) in fields.items()}
    out = np.zeros((h.shape[0], 3))
    for i, axis in enumerate(np.eye(3)):
        chi_modes = np.zeros(4, dtype=complex)
        chi_modes[1:] = wigner.vector_as_ell_1_modes(axis)
        chi = sfr.salm2map(chi_modes[None, :], 0, 1, n, n)[0]
        eth_chi = sfr.salm2map(wigner.eth_NP(chi_modes[None, :], 0), 1, 1, n, n)[0]
        ethbar_chi = sfr.salm2map(wigner.ethbar_NP(chi_modes[None, :], 0), -1, 1, n, n)[0]
        br = lambda a, op, b: _sphere_integral(maps[a].conjugate() * op * maps[b])  # noqa: E731
        total = (1 / 8) * (br("ebN", chi, "ebh") - br("eN", chi, "eh") + br("ebh", chi, "ebN") - br("eh", chi, "eN") + 6 * br("N", chi, "h")
                           + 6 * br("h", chi, "N"))
        total = total - (1 / 4) * (br("eN", eth_chi, "h") + br("h", ethbar_chi, "eN")) - (t / 2) * br("N", chi, "N")
        out[:, i] = total.real / (-32 * np.pi)
    return out
