"""Seeded sweep of the time-series calculus and the grid product (bms_spline_derivative,
bms_cubic_spline, bms_grid_multiply) against the oracle at RANDOM sizes: series of 4 .. 40 000 samples on
uniform, jittered and graded axes (the spline kernels work in 320-knot tiles with a 32-knot run-in: the sizes in the suite are a few
fixed ones), derivative orders -5 .. 3, evaluation points inside, on the knots, outside and unordered; grid products of random spins,
l ranges, working and output l, 1 .. 3 000 rows.
Usage: python tools/series_sweep.py [last_seed [first_seed]]   (prints failures; exit code = their number)"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

import scri_amd
from oracle import modes_time_series_ref as mref
from scri_amd import engine

DONE = {"spline calculus cases": 0, "cubic spline cases": 0, "grid products": 0}


def _axis(rng, n):
    kind = int(rng.integers(4))
    if kind == 0:
        t = np.linspace(-3.0, 9.0, n)
    elif kind == 1:  # jittered
        t = np.linspace(-3.0, 9.0, n)
        if n > 1:
            t = t + rng.uniform(-0.3, 0.3, n) * (t[1] - t[0])
    elif kind == 2:  # random samples
        t = np.sort(rng.uniform(-3.0, 9.0, n)) + np.arange(n) * 1e-6
    else:  # steps shrinking geometrically by up to 30x
        r = rng.uniform(3.0, 30.0) ** (-1.0 / max(n - 1, 1))
        t = -3.0 + np.concatenate([[0.0], np.cumsum(r ** np.arange(n - 1))])
        t = -3.0 + (t + 3.0) * 12.0 / max(t[-1] + 3.0, 1e-300)
    return t, kind


def _signal(rng, t, ncols):
    w = rng.uniform(0.3, 2.0, ncols)
    a = rng.normal(size=ncols) + 1j * rng.normal(size=ncols)
    return a[None, :] * np.exp(1j * w[None, :] * t[:, None]) * (1 + 0.05 * t[:, None])


def one(seed, ctx):
    rng = np.random.default_rng(55_000 + seed)
    bad = []
    # ---- spline calculus
    n = int(10 ** rng.uniform(np.log10(4), np.log10(40_000)))
    ncols = int(rng.integers(1, 40))
    t, kind = _axis(rng, n)
    y = _signal(rng, t, ncols)
    order = int(rng.integers(-5, 4))
    n_new = int(rng.integers(1, 3_000))
    # (samples up to two steps outside the data: extrapolation, as scipy does it; farther out a cubic amplifies rounding by (distance / step)^3)
    tn = rng.uniform(t[0] - 2 * (t[1] - t[0]), t[-1] + 2 * (t[-1] - t[-2]), n_new)
    tn[: min(n_new, 20)] = t[rng.integers(0, n, min(n_new, 20))]  # on knots
    what = f"seed {seed}: n={n} cols={ncols} axis={kind} order={order} n_new={n_new}"
    h_min = np.diff(t).min() if n > 1 else 1.0
    # Neighbouring steps of very different length (random samples: ratios beyond 1e4) make the spline system ill conditioned -- scipy's own
    # result is then 1e-11 .. 1e-10 from the spline computed in long double (seed 1431: 7.6e-11 at a ratio of 12 365) -- so the bar widens with
    # the largest ratio of adjacent steps
    d = np.diff(t)
    mesh = max(1.0, float(np.max(np.maximum(d[1:] / d[:-1], d[:-1] / d[1:]))) / 30.0) if n > 2 else 1.0
    try:
        got = engine.spline_derivative(t, y, tn, order, ctx=ctx)
        ref = mref.interpolate(t, y, tn, order)
        DONE["spline calculus cases"] += 1
        scale = max(1.0, np.abs(ref).max())
        # a derivative of order k amplifies the rounding of the data by ~ 1 / h^k
        # (antiderivatives sum rounding over the whole series: the suite's long-series test allows 1e-11)
        tol = 5e-13 * mesh * max(1.0, (0.02 / h_min)) ** max(order, 0) * (40.0 if order > 0 else (4.0 if order < 0 else 1.0))
        err = np.abs(got - ref).max()
        if not err < tol * scale:
            bad.append(f"spline_derivative: {err / scale:.2e} (bar {tol:.1e})")
    except Exception as e:  # noqa: BLE001
        # the oracle (scipy) and the engine must agree on what they refuse: both raise, or neither
        try:
            mref.interpolate(t, y, tn, order)
            bad.append(f"spline_derivative raised {type(e).__name__}: {str(e)[:120]}")
        except Exception:  # noqa: BLE001
            pass
    # ---- cubic spline (interpolation only, its own entry point)
    if n >= 2:
        try:
            got = engine.cubic_spline(t, y, np.sort(tn), ctx=ctx)
            ref = mref.interpolate(t, y, np.sort(tn), 0)
            DONE["cubic spline cases"] += 1
            err = np.abs(got - ref).max()
            if not err < 5e-13 * mesh * max(1.0, np.abs(ref).max()):
                bad.append(f"cubic_spline: {err:.2e} (bar {5e-13 * mesh:.1e})")
        except Exception as e:  # noqa: BLE001
            try:
                mref.interpolate(t, y, np.sort(tn), 0)
                bad.append(f"cubic_spline raised {type(e).__name__}: {str(e)[:120]}")
            except Exception:  # noqa: BLE001
                pass
    # ---- grid product
    sa, sb = int(rng.integers(-2, 3)), int(rng.integers(-2, 3))
    la, lb = int(rng.integers(abs(sa), 11)), int(rng.integers(abs(sb), 11))
    rows = int(10 ** rng.uniform(0, np.log10(3_000)))
    a = rng.normal(size=(rows, (la + 1) ** 2)) + 1j * rng.normal(size=(rows, (la + 1) ** 2))
    b = rng.normal(size=(rows, (lb + 1) ** 2)) + 1j * rng.normal(size=(rows, (lb + 1) ** 2))
    a[:, : sa * sa] = 0
    b[:, : sb * sb] = 0
    W = max(1, int(rng.integers(max(la, lb), la + lb + 1)))  # (a working l of 0 is a one-point grid: refused by the engine)
    Lout = int(rng.integers(min(abs(sa + sb), W), W + 1))  # (0 <= output l <= working l is the engine's precondition)
    if rows * (2 * W + 1) ** 2 < 3e6:
        got = engine.grid_multiply(a, sa, la, b, sb, lb, W, Lout, ctx=ctx)
        ref = mref.grid_multiply(a, sa, la, b, sb, lb, W, Lout)
        DONE["grid products"] += 1
        err = np.abs(got - ref).max()
        if got.shape != ref.shape or not err < 3e-13 * max(1.0, np.abs(ref).max()):
            bad.append(f"grid_multiply s=({sa},{sb}) l=({la},{lb}) W={W} Lout={Lout} rows={rows}: {err:.2e}")
    return bad, what


def main():
    last = int(sys.argv[1]) if len(sys.argv) > 1 else 100
    first = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    ctx = scri_amd.Context(0)
    failures = 0
    for seed in range(first, last):
        try:
            bad, what = one(seed, ctx)
        except Exception as e:  # noqa: BLE001
            bad, what = [f"{type(e).__name__}: {str(e)[:300]}"], f"seed {seed}"
        if bad:
            failures += 1
            print("FAILED", what, "|", "; ".join(bad), flush=True)
    print("checked:", DONE)
    print("done, failures:", failures)
    return failures


if __name__ == "__main__":
    sys.exit(min(main(), 100))
