"""Self-checks of the oracle's restated third-party pieces: these pin the
conventions (Wigner-D, SWSH, map2salm, spline) that the reference takes from un-vendored packages."""
import math

import numpy as np
import pytest
from scipy.interpolate import CubicSpline, InterpolatedUnivariateSpline
from scipy.special import sph_harm_y

from oracle import quat, wigner, spinsfast_ref, rotate_port


def _rand_rotor(rng):
    q = rng.normal(size=4)
    return q / np.linalg.norm(q)


def _Dmat(q, ell):
    Ra, Rb = quat.as_spinor_array(q)
    return wigner.wigner_D_matrices(Ra, Rb, ell, ell).reshape(2 * ell + 1, 2 * ell + 1)


def test_C1_C2_unitary_and_representation():
    rng = np.random.default_rng(0)
    q1, q2 = _rand_rotor(rng), _rand_rotor(rng)
    for ell in (1, 2, 5, 16):
        D1, D2, D12 = _Dmat(q1, ell), _Dmat(q2, ell), _Dmat(quat.qmul(q1, q2), ell)
        assert np.abs(D1 @ D1.conj().T - np.eye(2 * ell + 1)).max() < 2e-14
        assert np.abs(D12 - D1 @ D2).max() < 2e-14
        assert np.abs(D12 - D2 @ D1).max() > 0.1  # the other order is NOT the convention


def test_fast_D_matches_exact_sum():
    rng = np.random.default_rng(1)
    rotors = [_rand_rotor(rng) for _ in range(3)] + [np.array([1.0, 0, 0, 1e-9]), np.array([1e-9, 1.0, 0, 0]), np.array([1.0, 1, 0, 0]) / math.sqrt(2)]
    for q in rotors:
        q = q / np.linalg.norm(q)
        Ra, Rb = quat.as_spinor_array(q)
        fast = wigner.wigner_D_matrices(Ra, Rb, 0, 12)
        exact = wigner.wigner_D_matrices_exact(Ra, Rb, 0, 12, dps=40)
        assert np.abs(fast - exact).max() < 6e-15


def test_C_port_of_numba_kernels_matches_oracle_D():
    rng = np.random.default_rng(2)
    for q in [_rand_rotor(rng) for _ in range(5)] + [np.array([1.0, 0, 0, 0]), np.array([0, 1.0, 0, 0]), np.array([0, 0, 0, 1.0])]:
        Ra, Rb = quat.as_spinor_array(q)
        assert np.abs(rotate_port.wigner_D_matrices(Ra, Rb, 0, 8) - wigner.wigner_D_matrices(Ra, Rb, 0, 8)).max() < 2e-14


def test_C3_spin0_is_scipy_sph_harm():
    th, ph = 0.7, 1.9
    Y = wigner.swsh_grid(quat.from_spherical_coords(th, ph), 0, 5)
    ref = np.array([sph_harm_y(l, m, th, ph) for l, m in wigner.LM_range(0, 5)])
    assert np.abs(Y - ref).max() < 2e-15


def test_C4_spin_minus2_closed_form():
    th, ph = 1.1, -0.4
    Y = wigner.swsh_grid(quat.from_spherical_coords(th, ph), -2, 2)
    c = math.sqrt(5 / (64 * math.pi))
    assert abs(Y[wigner.LM_index(2, 2, 0)] - c * (1 + math.cos(th)) ** 2 * np.exp(2j * ph)) < 1e-15
    assert abs(Y[wigner.LM_index(2, -2, 0)] - c * (1 - math.cos(th)) ** 2 * np.exp(-2j * ph)) < 1e-15


@pytest.mark.parametrize("s", [-2, -1, 0, 1, 2])
def test_C5_analysis_inverts_synthesis(s):
    rng = np.random.default_rng(3)
    L = 4
    a = rng.normal(size=(L + 1) ** 2) + 1j * rng.normal(size=(L + 1) ** 2)
    a[: s * s] = 0
    for n_theta, n_phi in ((2 * L + 1, 2 * L + 1), (2 * L + 5, 2 * L + 3)):
        f = spinsfast_ref.salm2map(a, s, L, n_theta, n_phi)
        assert np.abs(spinsfast_ref.map2salm(f, s, L) - a).max() < 2e-14


def test_analysis_is_theta_quadrature_times_lambda():
    """steps (2)-(4) of the H&W analysis collapse to real quadrature weights q_j times sLambda_lm(theta_j)."""
    s, L, n_theta = -2, 6, 15
    T = spinsfast_ref.analysis_theta_matrix(s, L, n_theta)
    theta = np.pi * np.arange(n_theta) / (n_theta - 1)
    lam = wigner.swsh_grid(quat.from_spherical_coords(theta, np.zeros(n_theta)), s, L).real
    M = 2 * n_theta - 2
    E = sum(2.0 / (1 - p * p) * np.cos(p * theta) for p in range(-M // 2 + 1, M // 2 + 1) if p % 2 == 0)
    q = 2 * np.pi / M * E * 2
    q[0] /= 2
    q[-1] /= 2
    for m in range(-L, L + 1):
        for ell in range(max(abs(m), 2), L + 1):
            assert np.abs(T[m + L, ell] - q * lam[:, wigner.LM_index(ell, m, 0)]).max() < 1e-14


def test_C8_fitpack_is_not_a_knot_cubic_spline():
    rng = np.random.default_rng(4)
    x = np.cumsum(rng.uniform(0.5, 1.5, size=200))
    y = np.sin(0.3 * x)
    xn = np.linspace(x[0], x[-1], 777)
    assert np.abs(InterpolatedUnivariateSpline(x, y)(xn) - CubicSpline(x, y)(xn)).max() < 5e-15


def test_C9_affine_invariance_of_the_spline():
    x = np.linspace(0, 50, 501)
    y = np.sin(0.4 * x) * np.exp(-0.01 * x)
    k, alpha = 1.013, -0.37
    xp = np.linspace(k * (x[3] - alpha), k * (x[-4] - alpha), 400)
    a = CubicSpline(k * (x - alpha), y)(xp)
    b = CubicSpline(x, y)(xp / k + alpha)
    assert np.abs(a - b).max() < 5e-15


def test_C10_impulse_response_decays_like_0268_per_knot():
    n = 201
    x = np.arange(n, dtype=float)
    y = np.zeros(n)
    y[100] = 1.0
    cs = CubicSpline(x, y)
    d = np.abs(cs(x, 1))  # first derivatives at the knots = solution of the shared tridiagonal system
    ratios = d[102:125] / d[101:124]
    assert np.all(np.abs(ratios - (2 - math.sqrt(3))) < 1e-6)
    assert d[100 + 34] < 1e-18
