"""ORACLE (test infrastructure only -- never imported by the product): CPU restatement of the frame-construction
kernels that feed the rotation path, scri/mode_calculations.py:14-57 (_LdtVector), :209-313 (_LLMatrix), :403-432
(angular_velocity), vectorised over time.  data_dot is scipy CubicSpline(t, data).derivative()(t)
(scri/waveform_base.py:690-691).  PARITY UNPINNED at bit level; pinned by the reference's analytic tests
(tests/test_mode_calculations.py:75-110: zero, z and rotated angular velocity of a rigidly rotating waveform)."""
import math

import numpy as np
from scipy.interpolate import CubicSpline

from .wigner import LM_range


def ladder(l, m):
    """spherical_functions.ladder_operator_coefficient: sqrt(l(l+1) - m(m+1))"""
    return math.sqrt((l - m) * (l + m + 1))


def data_dot(t, data):
    return CubicSpline(t, data).derivative()(t)


def LdtVector(data, datadot, ell_min, ell_max):
    """mode_calculations.py:14-57"""
    lm = LM_range(ell_min, ell_max)
    Ldt = np.zeros((data.shape[0], 3))
    for i, (L, M) in enumerate(lm):
        L, M = int(L), int(M)
        Lp = np.conjugate(data[:, i + 1]) * datadot[:, i] * ladder(L, M) if M + 1 <= L else 0.0
        Lm = np.conjugate(data[:, i - 1]) * datadot[:, i] * ladder(L, -M) if M - 1 >= -L else 0.0
        Lz = np.conjugate(data[:, i]) * datadot[:, i] * M
        Ldt[:, 0] += 0.5 * (np.imag(Lp) + np.imag(Lm))
        Ldt[:, 1] += -0.5 * (np.real(Lp) - np.real(Lm))
        Ldt[:, 2] += np.imag(Lz)
    return Ldt


def LLMatrix(data, ell_min, ell_max):
    """mode_calculations.py:209-313"""
    lm = LM_range(ell_min, ell_max)
    LL = np.zeros((data.shape[0], 3, 3))
    z = np.zeros(data.shape[0], dtype=complex)
    for i, (L, M) in enumerate(lm):
        L, M = int(L), int(M)
        d = data[:, i]
        LpLp = np.conjugate(data[:, i + 2]) * d * (ladder(L, M + 1) * ladder(L, M)) if M + 2 <= L else z
        LpLm = np.conjugate(d) * d * (ladder(L, M - 1) * ladder(L, -M)) if M - 1 >= -L else z
        LmLp = np.conjugate(d) * d * (ladder(L, -(M + 1)) * ladder(L, M)) if M + 1 <= L else z
        LmLm = np.conjugate(data[:, i - 2]) * d * (ladder(L, -(M - 1)) * ladder(L, -M)) if M - 2 >= -L else z
        LpLz = np.conjugate(data[:, i + 1]) * d * (ladder(L, M) * M) if M + 1 <= L else z
        LzLp = np.conjugate(data[:, i + 1]) * d * ((M + 1) * ladder(L, M)) if M + 1 <= L else z
        LmLz = np.conjugate(data[:, i - 1]) * d * (ladder(L, -M) * M) if M - 1 >= -L else z
        LzLm = np.conjugate(data[:, i - 1]) * d * ((M - 1) * ladder(L, -M)) if M - 1 >= -L else z
        LzLz = np.conjugate(d) * d * M**2
        LxLx = 0.25 * (LpLp + LmLm + LmLp + LpLm)
        LxLy = -0.25j * (LpLp - LmLm + LmLp - LpLm)
        LxLz = 0.5 * (LpLz + LmLz)
        LyLx = -0.25j * (LpLp - LmLp + LpLm - LmLm)
        LyLy = -0.25 * (LpLp - LmLp - LpLm + LmLm)
        LyLz = -0.5j * (LpLz - LmLz)
        LzLx = 0.5 * (LzLp + LzLm)
        LzLy = -0.5j * (LzLp - LzLm)
        LL[:, 0, 0] += LxLx.real
        LL[:, 0, 1] += (LxLy + LyLx).real / 2.0
        LL[:, 0, 2] += (LxLz + LzLx).real / 2.0
        LL[:, 1, 0] += (LyLx + LxLy).real / 2.0
        LL[:, 1, 1] += LyLy.real
        LL[:, 1, 2] += (LyLz + LzLy).real / 2.0
        LL[:, 2, 0] += (LzLx + LxLz).real / 2.0
        LL[:, 2, 1] += (LzLy + LyLz).real / 2.0
        LL[:, 2, 2] += LzLz.real
    return LL


def angular_velocity(t, data, ell_min, ell_max):
    """mode_calculations.py:403-432 without the frame-velocity term"""
    l = LdtVector(data, data_dot(t, data), ell_min, ell_max)
    ll = LLMatrix(data, ell_min, ell_max)
    return -np.linalg.solve(ll, l[..., np.newaxis])[..., 0]


def integrate_angular_velocity(t, omega, R0, rtol=1e-13, atol=1e-13):
    """quaternion.integrate_angular_velocity as corotating_frame uses it (mode_calculations.py:470-471): the frame with
    R(t0) = R0 and dR/dt = (1/2) Omega R, Omega the cubic spline through the samples; here with scipy's DOP853."""
    from scipy.integrate import solve_ivp

    from . import quat

    sp = CubicSpline(t, omega)

    def rhs(tt, y):
        return 0.5 * quat.qmul(np.concatenate([[0.0], sp(tt)]), y)

    sol = solve_ivp(rhs, (t[0], t[-1]), np.asarray(R0, dtype=float), method="DOP853", t_eval=t, rtol=rtol, atol=atol)
    R = sol.y.T
    return R / np.linalg.norm(R, axis=1)[:, None]


def corotating_frame(t, data, ell_min, ell_max, R0):
    """mode_calculations.py:435-491 without z alignment"""
    return integrate_angular_velocity(t, angular_velocity(t, data, ell_min, ell_max), R0)
