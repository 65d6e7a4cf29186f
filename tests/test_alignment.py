"""scri_amd.alignment.align2d (the restatement of sxs.waveforms.alignment.align2d that the frame fixing of
scri/asymptotic_bondi_data/map_to_superrest_frame.py:979 and map_to_abd_frame.py:217,253 calls) on synthetic waveforms with a
known time offset and turn about z: host logic, no GPU."""
import numpy as np
import pytest

from scri_amd.alignment import align2d


class _Modes:
    def __init__(self, t, data, ell_min, ell_max):
        self.t, self.data, self.ell_min, self.ell_max = t, data, ell_min, ell_max

    def copy(self):
        return _Modes(self.t.copy(), self.data.copy(), self.ell_min, self.ell_max)


def _lm(ell_min, ell_max):
    return [(l, m) for l in range(ell_min, ell_max + 1) for m in range(-l, l + 1)]


def _chirp(t, ell_min, ell_max, seed=1):
    rng = np.random.default_rng(seed)
    LM = _lm(ell_min, ell_max)
    amp = rng.normal(size=len(LM)) + 1j * rng.normal(size=len(LM))
    phase = 0.07 * t + 2e-5 * t**2
    return np.stack([a * np.exp(-1j * m * phase) * (1 + 0.001 * t) for a, (l, m) in zip(amp, LM)], axis=1)


def _pair(dt, dphi, ell_min_a=2, ell_max_a=4):
    """wb = the chirp on its own times; wa = the waveform that the offset (dt, dphi) carries onto wb"""
    tb = np.linspace(-150.0, 150.0, 1501)
    ta = np.linspace(-160.0, 170.0, 1400)
    wb = _Modes(tb, _chirp(tb, 2, 4), 2, 4)
    m = np.array([m for _, m in _lm(2, 4)])
    full = _chirp(ta - dt, 2, 4) * np.exp(-1j * m * dphi)
    keep = [i for i, (l, _) in enumerate(_lm(2, 4)) if ell_min_a <= l <= ell_max_a]
    return _Modes(ta, full[:, keep], ell_min_a, ell_max_a), wb


@pytest.mark.parametrize("dt,dphi", [(3.217, 1.234), (-7.5, 5.9), (0.0, 0.0)])
def test_recovers_offset(dt, dphi):
    wa, wb = _pair(dt, dphi)
    err, wa_prime, res = align2d(wa, wb, -50.0, 50.0, n_brute_force_δt=200)
    assert abs(res.x[0] - dt) < 1e-5
    assert abs((res.x[1] - dphi + np.pi) % (2 * np.pi) - np.pi) < 1e-6
    assert err < 1e-12 and err == res.cost
    # wa_prime is wa moved by the optimum: it matches wb where they overlap
    from scipy.interpolate import CubicSpline

    t = np.linspace(-50, 50, 77)
    assert np.abs(CubicSpline(wa_prime.t, wa_prime.data)(t) - CubicSpline(wb.t, wb.data)(t)).max() < 1e-5


def test_include_modes_and_different_ell_ranges():
    wa, wb = _pair(2.5, 0.7, ell_min_a=2, ell_max_a=3)  # wa holds fewer modes than wb: the common ones are used
    err, _, res = align2d(wa, wb, -50.0, 50.0, n_brute_force_δt=100)
    assert abs(res.x[0] - 2.5) < 1e-5 and abs(res.x[1] - 0.7) < 1e-6
    # only m = +-2 modes: the turn is determined modulo pi
    err, _, res = align2d(wa, wb, -50.0, 50.0, n_brute_force_δt=100, include_modes=[(2, 2), (2, -2), (3, 2)])
    assert abs(res.x[0] - 2.5) < 1e-5
    assert abs((res.x[1] - 0.7 + np.pi / 2) % np.pi - np.pi / 2) < 1e-6
    with pytest.raises(ValueError, match="no common modes"):
        align2d(wa, wb, -50.0, 50.0, include_modes=[(7, 0)])


def test_cost_is_half_the_normalised_squared_distance():
    """a residual that cannot be removed: wb carries an extra m = 0 contribution that no time/phase offset produces"""
    wa, wb = _pair(1.0, 0.3)
    wb.data = wb.data.copy()
    wb.data[:, 2] += 0.5  # the (2, 0) mode
    err, wa_prime, res = align2d(wa, wb, -50.0, 50.0, n_brute_force_δt=100)
    rows = (wb.t >= -50) & (wb.t <= 50)
    t = wb.t[rows]
    from scipy.interpolate import CubicSpline

    diff = CubicSpline(wa_prime.t, wa_prime.data)(t) - wb.data[rows]
    trap = lambda y: 0.5 * np.sum((y[1:] + y[:-1]) * np.diff(t))  # noqa: E731
    expected = 0.5 * trap(np.sum(np.abs(diff) ** 2, axis=1)) / trap(np.sum(np.abs(wb.data[rows]) ** 2, axis=1))
    assert abs(err - expected) < 1e-9 * expected
    assert err > 1e-4


def test_window_checks():
    wa, wb = _pair(0.0, 0.0)
    with pytest.raises(ValueError, match="out of order"):
        align2d(wa, wb, 10.0, -10.0)
    with pytest.raises(ValueError, match="not contained in wb"):
        align2d(wa, wb, -200.0, 0.0)
    short = _Modes(wa.t[300:], wa.data[300:], wa.ell_min, wa.ell_max)
    with pytest.raises(ValueError, match="not contained in wa"):
        align2d(short, wb, -150.0, 150.0)
