"""Cross-checks of the third-party conventions the oracle restates (spherical_functions, spinsfast: un-vendored, absent from the
image) that do NOT go through oracle/wigner.py's arithmetic:

* Wigner D: `oracle.wigner.wigner_D_matrices` against sympy's `Rotation.D` (Varshalovich's convention, exact closed form) at
  rotors given by their Euler angles.  spherical_functions' D is the complex conjugate of that convention -- forced by its own
  sYlm(R) = (-1)^s sqrt((2l+1)/4pi) D^l_{m,-s}(R) with Y_lm ~ e^{+i m phi} -- so the expected relation is
  D_sf[l][m', m](R(alpha, beta, gamma)) = conj(Rotation.D(l, m', m, alpha, beta, gamma)), element by element: this pins the phase
  convention, the (m', m) index order and the (Ra, Rb) = (w + i z, y + i x) spinor ordering.
* map2salm: a second transcription of Huffenberger & Wandelt 2010 by their own route -- phi FFT, extension to the theta circle,
  REAL-SPACE multiplication by the quadrature weights w_r(theta_j), theta FFT, contraction with the Fourier coefficients of
  sympy's d^l_{m,-s}(theta) -- against `oracle.spinsfast_ref.map2salm` (which applies the same quadrature as a wrapped convolution
  assembled into matrices, with harmonics from oracle/wigner.py) on NON-band-limited maps, even and odd n_theta (Nyquist row of
  either parity), and exactness on band-limited ones.

This is the ceiling of what can be pinned here: the weights' frequency range (-M/2, M/2] on the M = 2 n_theta - 2 circle is the
restated reading of spinsfast's implementation; no stored output of the package itself exists in the reference to confirm it."""
import functools
import math

import numpy as np
import pytest

from oracle import quat, spinsfast_ref, wigner

sympy = pytest.importorskip("sympy")
from sympy.physics.quantum.spin import Rotation  # noqa: E402


def _rotor(alpha, beta, gamma):
    def ek(x):
        return np.array([math.cos(x / 2), 0.0, 0.0, math.sin(x / 2)])

    def ej(x):
        return np.array([math.cos(x / 2), 0.0, math.sin(x / 2), 0.0])

    return quat.qmul(quat.qmul(ek(alpha), ej(beta)), ek(gamma))


@pytest.mark.parametrize("angles", [(0.3, 1.1, -0.7), (2.9, 2.6, 0.4), (-1.2, 1e-3, 0.9), (0.5, math.pi - 2e-3, -2.2)])
def test_wigner_D_is_the_conjugate_of_sympys_rotation_D(angles):
    a, b, g = angles
    sp = quat.as_spinor_array(_rotor(a, b, g))
    ell_max = 3
    D = wigner.wigner_D_matrices(sp[0], sp[1], 0, ell_max)
    worst = 0.0
    for ell in range(ell_max + 1):
        for mp in range(-ell, ell + 1):
            for m in range(-ell, ell + 1):
                expect = np.conj(complex(Rotation.D(ell, mp, m, a, b, g).doit().evalf(30)))
                worst = max(worst, abs(D[wigner.LMpM_index(ell, mp, m, 0)] - expect))
    assert worst < 5e-15, worst


def _small_d(ell, mp, m, theta):
    """sympy's d^l_{m',m}(theta): a trigonometric polynomial of degree l, evaluated on the whole circle (beyond pi sympy's
    complex powers leave a rounding-size imaginary part)"""
    z = complex(sympy.N(Rotation.d(ell, mp, m, float(theta)).doit()))
    assert abs(z.imag) < 1e-14
    return z.real


@functools.lru_cache(maxsize=None)
def _d_fourier(ell, m, s):
    """Fourier coefficients Lam[m'] (m' = -l..l) of sympy's d^l_{m,-s}(theta) on the circle: d = sum_m' Lam[m'] exp(i m' theta)"""
    q = 2 * ell + 2
    th = 2 * np.pi * np.arange(q) / q
    d = np.array([_small_d(ell, m, -s, x) for x in th])
    c = np.fft.fft(d) / q
    return {mp: c[mp % q] for mp in range(-ell, ell + 1)}


def _w(p):
    if p == 1:
        return 1j * math.pi / 2
    if p == -1:
        return -1j * math.pi / 2
    return 2.0 / (1.0 - p * p) if p % 2 == 0 else 0.0


def map2salm_hw_fft(f, s, ell_max):
    """H&W 2010 sec. 2-3 with the quadrature as a real-space multiplication (their eq. for I_{m'm}), no oracle harmonics."""
    n_theta, n_phi = f.shape
    M = 2 * n_theta - 2
    fm = np.fft.fft(f, axis=1) / n_phi  # [j, m mod n_phi]
    theta = np.pi * np.arange(M) / (n_theta - 1)
    ps = np.arange(-M // 2 + 1, M // 2 + 1)  # (-M/2, M/2]
    wr = np.array([sum(_w(int(p)) * np.exp(-1j * p * th) for p in ps) for th in theta])
    out = np.zeros((ell_max + 1) ** 2, dtype=complex)
    for m in range(-ell_max, ell_max + 1):
        G = np.empty(M, dtype=complex)
        G[:n_theta] = fm[:, m % n_phi]
        for j in range(n_theta, M):
            G[j] = (-1.0) ** (m + s) * fm[M - j, m % n_phi]
        prod = G * wr
        J = {mp: np.sum(prod * np.exp(1j * mp * theta)) / M for mp in range(-ell_max, ell_max + 1)}
        for ell in range(max(abs(m), abs(s)), ell_max + 1):
            lam = _d_fourier(ell, m, s)
            norm = (-1.0) ** s * math.sqrt((2 * ell + 1) / (4 * math.pi))
            out[wigner.LM_index(ell, m, 0)] = 2 * math.pi * norm * sum(lam[mp] * J[mp] for mp in range(-ell, ell + 1))
    return out


@pytest.mark.parametrize("n_theta,n_phi", [(9, 9), (8, 7), (7, 10)])
@pytest.mark.parametrize("s", [-2, -1, 0, 1, 2])
def test_second_map2salm_agrees_on_non_band_limited_maps(s, n_theta, n_phi):
    ell_max = 3
    rng = np.random.default_rng(100 * n_theta + 10 * n_phi + s)
    f = rng.normal(size=(n_theta, n_phi)) + 1j * rng.normal(size=(n_theta, n_phi))  # white noise: every frequency up to Nyquist
    got = spinsfast_ref.map2salm(f, s, ell_max)
    expect = map2salm_hw_fft(f, s, ell_max)
    assert np.abs(got - expect).max() < 2e-13 * np.abs(expect).max()


@pytest.mark.parametrize("s", [-2, 0, 1])
def test_second_map2salm_is_exact_on_band_limited_maps(s):
    """a map synthesised from known mode weights with sympy's d functions (n_theta = 2 L + 1: no wrap) comes back exactly"""
    ell_max, n_theta, n_phi = 3, 9, 8
    rng = np.random.default_rng(s + 7)
    a = rng.normal(size=(ell_max + 1) ** 2) + 1j * rng.normal(size=(ell_max + 1) ** 2)
    a[: s * s] = 0
    theta = np.pi * np.arange(n_theta) / (n_theta - 1)
    phi = 2 * np.pi * np.arange(n_phi) / n_phi
    f = np.zeros((n_theta, n_phi), dtype=complex)
    for ell in range(abs(s), ell_max + 1):
        for m in range(-ell, ell + 1):
            d = np.array([_small_d(ell, m, -s, x) for x in theta])
            f += a[wigner.LM_index(ell, m, 0)] * (-1.0) ** s * math.sqrt((2 * ell + 1) / (4 * math.pi)) * np.outer(d, np.exp(1j * m * phi))
    assert np.abs(map2salm_hw_fft(f, s, ell_max) - a).max() < 1e-13
    assert np.abs(spinsfast_ref.map2salm(f, s, ell_max) - a).max() < 1e-13
    # and the oracle's synthesis is the same function
    assert np.abs(spinsfast_ref.salm2map(a, s, ell_max, n_theta, n_phi) - f).max() < 1e-13
