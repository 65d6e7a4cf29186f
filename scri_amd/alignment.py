"""Time-and-phase alignment of two waveforms: what scri's frame fixing imports as `sxs.waveforms.alignment.align2d`
(scri/asymptotic_bondi_data/map_to_superrest_frame.py:9,979; map_to_abd_frame.py:15,217,253).

`sxs` (pinned ">=2022.4.0" by the pyproject.toml of scri) is a third-party package that is neither vendored by scri
nor in this image, so this is a restatement of its published algorithm, not a checked port -- parity with
the sxs implementation is UNPINNED; the tests pin the behaviour the reference relies on instead: the optimum (dt, dphi)
is the time translation and the turn about z whose BMSTransformation carries `wa` onto `wb`.

    wa'(t) = wa(t + dt) e^{i m dphi}      minimises      int_{t1}^{t2} sum_{lm} |wa'_{lm}(t) - wb_{lm}(t)|^2 dt
                                                          / int_{t1}^{t2} sum_{lm} |wb_{lm}(t)|^2 dt

A brute-force scan over (dt, dphi) seeds `scipy.optimize.least_squares`.  The scan costs one spline evaluation per dt:
the dphi dependence is  ||wa||^2 + ||wb||^2 - 2 Re sum_m e^{i m dphi} C_m(dt)  with  C_m = int sum_l wa_{lm}(t + dt) conj(wb_{lm}(t)) dt.

Host-side control logic (a two-parameter optimisation over a short window of a few modes): numpy/scipy, no GPU work.
"""
import numpy as np


def _window_rows(t, t1, t2):
    keep = (t >= t1) & (t <= t2)
    if keep.sum() < 4:
        raise ValueError(f"fewer than 4 samples of the fixed waveform lie in [{t1}, {t2}]")
    return keep


def _trapezoid_weights(t):
    w = np.zeros_like(t)
    w[:-1] += 0.5 * np.diff(t)
    w[1:] += 0.5 * np.diff(t)
    return w


def align2d(wa, wb, t1, t2, n_brute_force_δt=None, n_brute_force_δϕ=None, include_modes=None, nprocs=None):
    """Optimal time offset and turn about z to apply to `wa` so that it matches `wb` on [t1, t2].

    wa, wb: WaveformModes-like objects (.t, .data [n, modes], .ell_min, .ell_max, .LM); include_modes: optional list of
    (l, m) pairs the cost is restricted to; nprocs is accepted for call compatibility (the scan is vectorised instead).

    Returns (error, wa_prime, optimum): error = optimum.cost = half the normalised squared L2 distance at the optimum,
    wa_prime = `wa` on the times wa.t - dt with every mode multiplied by e^{i m dphi}, optimum = the
    scipy.optimize.OptimizeResult with optimum.x = [dt, dphi]."""
    from scipy.interpolate import CubicSpline
    from scipy.optimize import least_squares

    ell_min, ell_max = max(wa.ell_min, wb.ell_min), min(wa.ell_max, wb.ell_max)
    LM = [(l, m) for l in range(ell_min, ell_max + 1) for m in range(-l, l + 1)]
    if include_modes is not None:
        wanted = {tuple(x) for x in include_modes}
        LM = [lm for lm in LM if lm in wanted]
    if not LM:
        raise ValueError("no common modes to align")
    col = lambda w: np.array([l * (l + 1) - w.ell_min**2 + m for l, m in LM])  # noqa: E731
    m_of = np.array([m for _, m in LM], dtype=float)
    ta, tb = np.asarray(wa.t, dtype=float), np.asarray(wb.t, dtype=float)
    if not (t1 < t2):
        raise ValueError(f"(t1, t2) = ({t1}, {t2}) is out of order")
    if t1 < tb[0] or t2 > tb[-1]:
        raise ValueError(f"(t1, t2) = ({t1}, {t2}) is not contained in wb, which spans ({tb[0]}, {tb[-1]})")
    δt_lower = max(t1 - t2, ta[0] - t1)
    δt_upper = min(t2 - t1, ta[-1] - t2)
    if not (δt_lower <= 0.0 <= δt_upper):
        raise ValueError(f"(t1, t2) = ({t1}, {t2}) is not contained in wa, which spans ({ta[0]}, {ta[-1]})")

    rows = _window_rows(tb, t1, t2)
    t = tb[rows]
    B = np.asarray(wb.data)[rows][:, col(wb)]
    a_of = CubicSpline(ta, np.asarray(wa.data)[:, col(wa)])
    w = _trapezoid_weights(t)
    normalization = w @ np.sum(np.abs(B) ** 2, axis=1)
    if not normalization > 0.0:
        raise ValueError("wb vanishes on the window: nothing to align to")
    ms = np.unique(m_of)
    m_slot = np.searchsorted(ms, m_of)

    # residual vector whose squared length is the normalised squared distance (trapezoid weights folded in): the same
    # `cost` as a single scalar residual, but smooth at a perfect match
    root_w = np.sqrt(w / normalization)[:, None]

    def residual(x):
        A = a_of(t + x[0]) * np.exp(1j * m_of * x[1])
        return np.ascontiguousarray((A - B) * root_w).view(float).ravel()

    # ---- brute force: every dt costs one spline evaluation, every dphi a (2 l_max + 1)-term sum
    in_a = ((ta >= t1 + δt_lower) & (ta <= t2 + δt_upper)).sum()
    if n_brute_force_δt is None:
        n_brute_force_δt = int(max(in_a, rows.sum()))
    if n_brute_force_δϕ is None:
        n_brute_force_δϕ = 2 * ell_max + 1
    δts = np.linspace(δt_lower, δt_upper, max(int(n_brute_force_δt), 1)) if δt_upper > δt_lower else np.array([0.0])
    if not np.any(δts == 0.0):
        δts = np.sort(np.append(δts, 0.0))
    δϕs = np.linspace(0.0, 2 * np.pi, max(int(n_brute_force_δϕ), 1), endpoint=False)
    phases = np.exp(1j * np.outer(δϕs, ms))  # [n_dphi, n_m]
    best = (np.inf, 0.0, 0.0)
    for δt in δts:
        A = a_of(t + δt)
        norm_a = w @ np.sum(np.abs(A) ** 2, axis=1)
        cross = w @ (A * np.conj(B))  # per mode
        C = np.zeros(ms.size, dtype=complex)
        np.add.at(C, m_slot, cross)
        costs = norm_a + normalization - 2.0 * (phases @ C).real
        k = int(np.argmin(costs))
        if costs[k] < best[0]:
            best = (costs[k], δt, δϕs[k])

    # ---- refine.  dphi is periodic: let it run free around the seed and wrap afterwards
    x0 = np.array([best[1], best[2]])
    lo = np.array([δt_lower, x0[1] - np.pi])
    hi = np.array([δt_upper, x0[1] + np.pi])
    if hi[0] <= lo[0]:
        lo[0], hi[0] = lo[0] - 1e-12, hi[0] + 1e-12
    optimum = least_squares(residual, np.clip(x0, lo, hi), bounds=(lo, hi), xtol=1e-15, ftol=1e-15, gtol=1e-15, max_nfev=50000)
    optimum.x[1] = np.mod(optimum.x[1], 2 * np.pi)

    wa_prime = wa.copy()
    all_m = np.array([m for l in range(wa.ell_min, wa.ell_max + 1) for m in range(-l, l + 1)], dtype=float)
    wa_prime.data = np.asarray(wa.data) * np.exp(1j * all_m * optimum.x[1])[None, :]
    wa_prime.t = ta - optimum.x[0]
    return optimum.cost, wa_prime, optimum
