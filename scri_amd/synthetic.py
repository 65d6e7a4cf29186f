"""Synthetic workloads of BASELINE.json / SURVEY.md section 8(d): smooth-in-time mode data (so the per-pixel
splines are well conditioned) and the BMS transformations they are benchmarked with."""
import numpy as np

from .mode_algebra import LM_range, LM_index


def chirp_modes(t, ell_min, ell_max, seed):
    """data[t, (l,m)] = a_lm 10^(-l/2) exp(i m phi(t)), phi = 0.05 t + 2e-6 t^2, a ~ N(0,1) + i N(0,1)."""
    rng = np.random.default_rng(seed)
    LM = LM_range(ell_min, ell_max)
    a = rng.normal(size=LM.shape[0]) + 1j * rng.normal(size=LM.shape[0])
    amp = a * 10.0 ** (-LM[:, 0] / 2.0)
    phase = 0.05 * t + 2e-6 * t**2
    return amp[None, :] * np.exp(1j * LM[None, :, 1] * phase[:, None])


def real_supertranslation(alpha):
    """Project mode weights onto those of a real function: a_{l,m} = (-1)^m conj(a_{l,-m})."""
    a = np.array(alpha, dtype=complex)
    lmax = int(round(np.sqrt(a.size))) - 1
    for ell in range(lmax + 1):
        for m in range(ell + 1):
            ip, im = LM_index(ell, m, 0), LM_index(ell, -m, 0)
            a[ip] = (a[ip] + (-1.0) ** m * np.conj(a[im])) / 2
            a[im] = (-1.0) ** m * np.conj(a[ip])
    return a


# supertranslation of tests/test_bms_transformations.py:298 of the reference (made real), x 1e-3
S9 = real_supertranslation(np.array([1, 2 + 4j, 3, -2 + 4j, 7 - 5j, -3 - 2j, 4, 3 - 2j, 7 + 5j]) * 1e-3)

def rotor_series(t, seed, omega=8 * np.pi / 110.0, q0=None):
    """R(t) = exp(n omega t / 2) [* q0]: rotation about a fixed random axis n (seed) at angular velocity omega -- the
    `R_basis` series of the reference's rotation tests (tests/conftest.py:84-87) -- as a float array [N, 4] (w, x, y, z)."""
    rng = np.random.default_rng(seed)
    axis = rng.normal(size=3)
    axis /= np.linalg.norm(axis)
    half = 0.5 * omega * np.asarray(t, dtype=float)
    R = np.empty((half.shape[0], 4))
    R[:, 0] = np.cos(half)
    R[:, 1:] = np.sin(half)[:, None] * axis[None, :]
    if q0 is not None:  # quaternion product R(t) * q0
        a, b = R, np.asarray(q0, dtype=float)
        R = np.stack([
            a[:, 0] * b[0] - a[:, 1] * b[1] - a[:, 2] * b[2] - a[:, 3] * b[3],
            a[:, 0] * b[1] + a[:, 1] * b[0] + a[:, 2] * b[3] - a[:, 3] * b[2],
            a[:, 0] * b[2] - a[:, 1] * b[3] + a[:, 2] * b[0] + a[:, 3] * b[1],
            a[:, 0] * b[3] + a[:, 1] * b[2] - a[:, 2] * b[1] + a[:, 3] * b[0],
        ], axis=1)
    return R


Q1234 = np.array([1.0, 2, 3, 4]) / np.sqrt(30)


def cfg1():
    """BASELINE.json configs[0] / SURVEY 8(d) cfg1: h, l = 2..4, t = linspace(-10, 100, 2000),
    data[t, (l,m)] = a_lm exp(i m 0.3 t) (seed 1); rotations: (i) constant q = (1,2,3,4)/sqrt(30), (ii) the series
    exp(n omega t / 2), omega = 8 pi / 110 (seed 2).  Returns (t, data, dict(constant=q, series=R[N, 4]))."""
    t = np.linspace(-10.0, 100.0, 2000)
    rng = np.random.default_rng(1)
    LM = LM_range(2, 4)
    a = rng.normal(size=LM.shape[0]) + 1j * rng.normal(size=LM.shape[0])
    data = a[None, :] * np.exp(1j * LM[None, :, 1] * 0.3 * t[:, None])
    return t, data, dict(constant=Q1234.copy(), series=rotor_series(t, 2))


CONFIGS = {
    # name: (ell_max, n_times, dt, seed, transformation kwargs)
    # cfg2: rotation by `rotor_series(t, 4, omega = 8 pi / (t[-1] - t[0]))` first, then the supertranslation
    "cfg2": dict(ell_max=8, n_times=100_000, dt=0.1, seed=3, rotation_seed=4, kwargs=dict(supertranslation=S9)),
    "cfg3": dict(
        ell_max=16, n_times=100_000, dt=0.1, seed=5,
        kwargs=dict(supertranslation=S9, frame_rotation=np.array([1.0, 2, 3, 4]) / np.sqrt(30), boost_velocity=np.array([1.0, 2, 3]) * 1e-4),
    ),
    "cfg4": dict(
        ell_max=16, n_times=1_000_000, dt=0.1, seed=6,
        kwargs=dict(supertranslation=S9, frame_rotation=np.array([1.0, 2, 3, 4]) / np.sqrt(30), boost_velocity=np.array([1.0, 2, 3]) * 1e-4),
    ),
    # AsymptoticBondiData, psi0..psi4 + sigma; the grid follows working_ell_max (default 2 ell_max + 1 -> 99 x 99)
    "cfg5": dict(
        ell_max=24, n_times=200_000, dt=0.1, seed=7,
        kwargs=dict(supertranslation=S9, frame_rotation=np.array([1.0, 2, 3, 4]) / np.sqrt(30), boost_velocity=np.array([1.0, 2, 3]) * 1e-4),
    ),
}

ABD_SPINS = (2, 1, 0, -1, -2, 2)

TIME_AXES = ("uniform", "jitter", "sxs")


def time_axis(n, dt, kind="uniform"):
    """The time samples of a workload.  BASELINE.json's configurations are uniformly stepped (`uniform`: i dt); real scri inputs (SXS
    / CCE output) are not, so two more axes of the same span n dt carry the same smooth signals (the data are functions of t):
    `jitter` -- every sample moved by up to +-30 % of dt (seeded; steps between 0.4 and 1.6 dt);
    `sxs`    -- steps shrinking geometrically by 20x over the series, as an inspiral -> merger run's do."""
    i = np.arange(n, dtype=float)
    if kind == "uniform":
        return i * dt
    if kind == "jitter":
        return (i + 0.3 * np.random.default_rng(1234).uniform(-1.0, 1.0, size=n)) * dt
    if kind == "sxs":
        steps = 20.0 ** (-i / max(n - 1, 1))
        t = np.concatenate([[0.0], np.cumsum(steps[:-1])])
        return t * (n * dt / (t[-1] + steps[-1]))
    raise ValueError(f"time axis {kind!r}: one of {TIME_AXES}")


def abd_workload(name="cfg5", n_times=None, rows=None, ell_max=None, axis="uniform"):
    """(u_global, raw[6, rows, (ell_max+1)^2], spec): every field as `chirp_modes` (its own seed), zeros below |s|."""
    spec = dict(CONFIGS[name])
    n = int(n_times or spec["n_times"])
    spec["n_times"] = n
    if ell_max is not None:
        spec["ell_max"] = int(ell_max)
    u = time_axis(n, spec["dt"], axis)
    r0, r1 = rows if rows is not None else (0, n)
    nm = (spec["ell_max"] + 1) ** 2
    raw = np.zeros((6, r1 - r0, nm), dtype=complex)
    for f, s in enumerate(ABD_SPINS):
        raw[f] = chirp_modes(u[r0:r1], 0, spec["ell_max"], spec["seed"] + 10 * f)
        raw[f, :, : s * s] = 0
    return u, raw, spec


def workload(name, n_times=None, rows=None, axis="uniform"):
    """(t_global, data[rows], spec): `rows=(r0, r1)` generates only those rows of the global series; `axis`: time_axis()."""
    spec = dict(CONFIGS[name])
    n = int(n_times or spec["n_times"])
    spec["n_times"] = n
    t = time_axis(n, spec["dt"], axis)
    r0, r1 = rows if rows is not None else (0, n)
    data = chirp_modes(t[r0:r1], 2, spec["ell_max"], spec["seed"])
    return t, data, spec
