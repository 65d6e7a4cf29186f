"""Equiangular-grid SWSH analysis / synthesis (``spinsfast.map2salm`` / ``salm2map``).

``spinsfast`` (>=2022.4, C + FFTW; Huffenberger & Wandelt 2010, ApJS 189, 255) is un-vendored.
Call sites in the reference: ``map2salm`` scri/waveform_grid.py:305,
scri/asymptotic_bondi_data/transformations.py:419-429, scri/bms_transformations.py:178,
scri/modes_time_series.py:188; ``salm2map`` scri/modes_time_series.py:177,180.

Restated algorithm (H&W 2010 sec. 2-3), for a map f[j,k] on
theta_j = pi j/(n_theta-1) (both poles), phi_k = 2 pi k/n_phi:

  (1) f_m(theta_j) = (1/n_phi) sum_k f_jk exp(-i m phi_k)
  (2) extension to the theta circle, M = 2 n_theta - 2 samples:
        G_m(theta_j) = f_m(theta_j)                      j < n_theta
                     = (-1)^(m+s) f_m(theta_{M-j})       otherwise
  (3) c_{n,m} = (1/M) sum_j G_m(theta_j) exp(-i n theta_j),   n in (-M/2, M/2]
  (4) a_lm = 2 pi sum_n c_{n,m} int_0^pi exp(i n theta) sLambda_lm(theta) sin(theta) dtheta
      with sLambda_lm(theta) = sYlm(theta, 0) = sum_{m'} Lambda_{m'} exp(i m' theta) and
      w(p) = int_0^pi exp(i p theta) sin(theta) dtheta
           = 2/(1-p^2) (p even), +-i pi/2 (p = +-1), 0 (other odd p).
      The sum over n is the circular convolution on Z_M that H&W evaluate as a real-space
      multiplication by the quadrature weights w_r(theta_j) = sum_p w(p) exp(i p theta_j):
      frequencies n + m' are wrapped into (-M/2, M/2].  For input band-limited to
      n_theta >= 2 L + 1 no wrap occurs and the analysis is exact.

Output layout: [..., (ell_max+1)^2], l from 0, zeros for l < |s|  (as spinsfast).
"""
import functools
import math
import numpy as np

from . import quat
from .wigner import LM_index, swsh_grid


def _w(p):
    if p == 1:
        return 1j * math.pi / 2
    if p == -1:
        return -1j * math.pi / 2
    if p % 2 == 0:
        return 2.0 / (1.0 - p * p)
    return 0.0


def _wrap(p, M):
    """Representative of p mod M in (-M/2, M/2]."""
    q = ((p + M // 2 - 1) % M) - (M // 2 - 1)
    return q


@functools.lru_cache(maxsize=64)
def analysis_theta_matrix(s, ell_max, n_theta):
    """T[m + ell_max][l][j]: a_lm = sum_j T[m][l][j] f_m(theta_j)   (steps 2-4 as one matrix).

    Returned as complex array [2 ell_max+1, ell_max+1, n_theta]."""
    M = 2 * n_theta - 2
    # Fourier coefficients of sLambda_lm(theta) on the full circle
    Qn = 2 * ell_max + 2
    thq = 2 * np.pi * np.arange(Qn) / Qn
    lam = swsh_grid(quat.from_spherical_coords(thq, np.zeros(Qn)), s, ell_max)  # [Qn, (L+1)^2]
    Lam = np.fft.fft(lam, axis=0) / Qn  # coefficient of exp(i m' theta) at index m' mod Qn
    ns = np.arange(-M // 2 + 1, M // 2 + 1)
    theta = np.pi * np.arange(M) / (n_theta - 1)
    E = np.exp(-1j * np.outer(ns, theta)) / M  # c_n = sum_j E[n,j] G_j
    T = np.zeros((2 * ell_max + 1, ell_max + 1, n_theta), dtype=complex)
    for m in range(-ell_max, ell_max + 1):
        # extension operator X: G = X f  (M x n_theta)
        X = np.zeros((M, n_theta))
        for j in range(n_theta):
            X[j, j] = 1.0
        for j in range(n_theta, M):
            X[j, M - j] = (-1.0) ** (m + s)
        for ell in range(max(abs(m), abs(s)), ell_max + 1):
            # I[n] = sum_{m'} Lambda_{m'} w~(n + m')
            I = np.zeros(len(ns), dtype=complex)
            for mp in range(-ell, ell + 1):
                L = Lam[mp % Qn, LM_index(ell, m, 0)]
                if L == 0:
                    continue
                for i_n, n in enumerate(ns):
                    I[i_n] += L * _w(_wrap(n + mp, M))
            T[m + ell_max, ell] = 2 * np.pi * (I @ E @ X)
    return T


def map2salm(f, s, ell_max):
    """spinsfast.map2salm(f[..., n_theta, n_phi], s, ell_max) -> [..., (ell_max+1)^2]."""
    f = np.asarray(f, dtype=complex)
    n_theta, n_phi = f.shape[-2:]
    ms = np.arange(-ell_max, ell_max + 1)
    phi = 2 * np.pi * np.arange(n_phi) / n_phi
    F = np.exp(-1j * np.outer(phi, ms)) / n_phi  # f_m = f @ F
    fm = f @ F  # [..., n_theta, 2L+1]
    T = analysis_theta_matrix(s, ell_max, n_theta)
    out = np.zeros(f.shape[:-2] + ((ell_max + 1) ** 2,), dtype=complex)
    for m in ms:
        for ell in range(max(abs(m), abs(s)), ell_max + 1):
            out[..., LM_index(ell, m, 0)] = fm[..., :, m + ell_max] @ T[m + ell_max, ell]
    return out


def map2salm_matrix(s, ell_max, n_theta, n_phi):
    """Dense matrix A[(l,m), (j,k)] with map2salm(f) = A @ f.ravel()."""
    ms = np.arange(-ell_max, ell_max + 1)
    phi = 2 * np.pi * np.arange(n_phi) / n_phi
    F = np.exp(-1j * np.outer(phi, ms)) / n_phi
    T = analysis_theta_matrix(s, ell_max, n_theta)
    A = np.zeros(((ell_max + 1) ** 2, n_theta, n_phi), dtype=complex)
    for m in ms:
        for ell in range(max(abs(m), abs(s)), ell_max + 1):
            A[LM_index(ell, m, 0)] = np.outer(T[m + ell_max, ell], F[:, m + ell_max])
    return A.reshape((ell_max + 1) ** 2, n_theta * n_phi)


def salm2map(a, s, ell_max, n_theta, n_phi):
    """spinsfast.salm2map: synthesis on the same grid, f[..., n_theta, n_phi]."""
    a = np.asarray(a, dtype=complex)
    theta = np.pi * np.arange(n_theta) / (n_theta - 1)
    phi = 2 * np.pi * np.arange(n_phi) / n_phi
    th, ph = np.meshgrid(theta, phi, indexing="ij")
    Y = swsh_grid(quat.from_spherical_coords(th, ph), s, ell_max)  # [n_theta, n_phi, (L+1)^2]
    return np.tensordot(a, Y, axes=([-1], [-1]))
