"""ORACLE (test infrastructure only -- never imported by the product): CPU restatement of
scri/modes_time_series.py:72-202 (interpolate with derivative_order, grid_multiply) on scipy + the oracle's own
salm2map / map2salm.  PARITY UNPINNED at bit level (see oracle/__init__.py); pinned by analytic checks in
tests/test_oracle_series.py (polynomial calculus is exact for cubics; products of harmonics against Wigner-3j)."""
import numpy as np
from scipy.interpolate import CubicSpline

from . import spinsfast_ref as sfr


def interpolate(time, data, new_time, derivative_order=0):
    """scri/modes_time_series.py:72-98: CubicSpline(u, data, axis=-2), .antiderivative(-k) / .derivative(k), evaluated."""
    spline = CubicSpline(np.asarray(time, dtype=float), np.asarray(data, dtype=complex), axis=-2)
    if derivative_order < 0:
        spline = spline.antiderivative(-derivative_order)
    elif 0 < derivative_order <= 3:
        spline = spline.derivative(derivative_order)
    elif derivative_order > 3:
        raise ValueError("CubicSpline cannot take a derivative of that order")
    return spline(np.asarray(new_time, dtype=float))


def grid_multiply(a, spin_a, ell_max_a, b, spin_b, ell_max_b, working_ell_max=None, output_ell_max=None):
    """scri/modes_time_series.py:142-202: salm2map both (l_min = 0 layouts), multiply, map2salm, truncate."""
    if output_ell_max is None:
        output_ell_max = ell_max_a
    if working_ell_max is None:
        working_ell_max = ell_max_a + ell_max_b
    n = 2 * working_ell_max + 1
    ga = sfr.salm2map(np.asarray(a, dtype=complex), spin_a, ell_max_a, n, n)
    gb = sfr.salm2map(np.asarray(b, dtype=complex), spin_b, ell_max_b, n, n)
    prod = sfr.map2salm(ga * gb, spin_a + spin_b, working_ell_max)
    return prod[..., : (output_ell_max + 1) ** 2]
