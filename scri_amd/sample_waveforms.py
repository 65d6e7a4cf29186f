"""Sample waveforms with known behaviour (scri/sample_waveforms.py:12-381): the objects the reference's own tests are built on --
constant and single-mode data, random data, and the single mode proportional to time with its analytically supertranslated
counterpart (the known answer of tests/test_waveform_grid.py).  Host-side generators; what is done WITH them runs on the GPU.
(`fake_precessing_waveform` and the finite-radius HDF5 writer of the reference are not restated here.)"""
import math
import warnings
from fractions import Fraction

import numpy as np

from . import DataType, Inertial, Corotating, SpinWeights, h
from .mode_algebra import LM_index, LM_range, LM_total_size, vector_as_ell_1_modes


def wigner_3j(j1, j2, j3, m1, m2, m3):
    """Wigner 3-j symbol for integer arguments by Racah's formula, the sum in exact rational arithmetic (sf.Wigner3j)"""
    j1, j2, j3, m1, m2, m3 = (int(x) for x in (j1, j2, j3, m1, m2, m3))
    if m1 + m2 + m3 != 0 or abs(m1) > j1 or abs(m2) > j2 or abs(m3) > j3 or j3 > j1 + j2 or j3 < abs(j1 - j2):
        return 0.0
    f = math.factorial
    delta = Fraction(f(j1 + j2 - j3) * f(j1 - j2 + j3) * f(-j1 + j2 + j3), f(j1 + j2 + j3 + 1))
    norm = delta * f(j1 + m1) * f(j1 - m1) * f(j2 + m2) * f(j2 - m2) * f(j3 + m3) * f(j3 - m3)
    total = Fraction(0)
    for k in range(max(0, j2 - j3 - m1, j1 - j3 + m2), min(j1 + j2 - j3, j1 - m1, j2 + m2) + 1):
        total += Fraction((-1) ** k, f(k) * f(j1 + j2 - j3 - k) * f(j1 - m1 - k) * f(j2 + m2 - k) * f(j3 - j2 + m1 + k) * f(j3 - j1 - m2 + k))
    return float((-1) ** (j1 - j2 - m3) * total) * math.sqrt(norm)


def _unused(kwargs):
    if kwargs:
        import pprint

        warnings.warn(f"\nUnused kwargs passed to this function:\n{pprint.pformat(kwargs, width=1)}")


def modes_constructor(constructor_statement, data_functor, **kwargs):
    """WaveformModes filled by `data_functor(t, LM)`; t (default 1101 samples on [-10, 100]), frame, frameType (Inertial), dataType
    (h), r_is_scaled_out / m_is_scaled_out (True), ell_min (|s| of the data type), ell_max (8) as keywords"""
    from .waveform_modes import WaveformModes

    t = np.array(kwargs.pop("t", np.linspace(-10.0, 100.0, num=1101)), dtype=float)
    frame = kwargs.pop("frame", None)
    frameType = int(kwargs.pop("frameType", Inertial))
    dataType = int(kwargs.pop("dataType", h))
    r_out, m_out = bool(kwargs.pop("r_is_scaled_out", True)), bool(kwargs.pop("m_is_scaled_out", True))
    ell_min = int(kwargs.pop("ell_min", abs(SpinWeights[dataType])))
    ell_max = int(kwargs.pop("ell_max", 8))
    ctx = kwargs.pop("ctx", None)
    _unused(kwargs)
    data = data_functor(t, LM_range(ell_min, ell_max))
    return WaveformModes(t=t, frame=frame, data=data, history=["# Called from constant_waveform"], frameType=frameType, dataType=dataType,
                         r_is_scaled_out=r_out, m_is_scaled_out=m_out, constructor_statement=constructor_statement, ell_min=ell_min,
                         ell_max=ell_max, ctx=ctx)


def constant_waveform(**kwargs):
    """every mode constant in time: (l, m) -> m - i m"""
    _unused({k: v for k, v in kwargs.items() if k not in ("t", "ell_min", "ell_max", "ctx")})
    keep = {k: v for k, v in kwargs.items() if k in ("t", "ell_min", "ell_max", "ctx")}
    return modes_constructor(f"constant_waveform(**{kwargs})", lambda t, LM: np.repeat((LM[:, 1] - 1j * LM[:, 1])[None, :].astype(complex), t.shape[0], axis=0), **keep)


def single_mode(ell, m, **kwargs):
    """1 in the (ell, m) slot and 0 elsewhere"""
    def functor(t, LM):
        data = np.zeros((t.shape[0], LM.shape[0]), dtype=complex)
        data[:, LM_index(ell, m, int(LM[:, 0].min()))] = 1.0
        return data

    keep = {k: kwargs.pop(k) for k in ("t", "ell_min", "ell_max", "ctx") if k in kwargs}
    _unused(kwargs)
    return modes_constructor(f"single_mode({ell}, {m}, **{kwargs})", functor, **keep)


def _random_setup(kwargs):
    begin, end, n_times = float(kwargs.pop("begin", -10.0)), float(kwargs.pop("end", 100.0)), int(kwargs.pop("n_times", 1101))
    rng = np.random.default_rng(kwargs.pop("seed", None))
    if kwargs.pop("uniform_time", False):
        t = np.linspace(begin, end, num=n_times)
    else:
        t = np.sort(rng.uniform(begin, end, size=n_times))
    rotating = kwargs.pop("rotating", True)
    frame = None
    if rotating:
        frame = rng.normal(size=(n_times, 4))
        frame /= np.linalg.norm(frame, axis=1)[:, None]
    return rng, t, frame, (Corotating if rotating else Inertial)


def random_waveform(**kwargs):
    """random data at each time step on (by default) random times, in a randomly oriented corotating frame (uniform_time, begin,
    end, n_times, rotating, seed as keywords)"""
    rng, t, frame, frameType = _random_setup(kwargs)
    return modes_constructor(f"random_waveform(**{kwargs})", lambda t_, LM: rng.normal(size=(t_.shape[0], LM.shape[0])) + 1j * rng.normal(size=(t_.shape[0], LM.shape[0])),
                             t=t, frame=frame, frameType=frameType, **kwargs)


def random_waveform_proportional_to_time(**kwargs):
    """every mode a random complex constant times the time"""
    rng, t, frame, frameType = _random_setup(kwargs)
    return modes_constructor(f"random_waveform_proportional_to_time(**{kwargs})",
                             lambda t_, LM: np.outer(t_, rng.normal(size=LM.shape[0]) + 1j * rng.normal(size=LM.shape[0])), t=t, frame=frame,
                             frameType=frameType, **kwargs)


def _single_mode_series(kwargs, values_of_t):
    from .waveform_modes import WaveformModes

    s = kwargs.pop("s", -2)
    ell = kwargs.pop("ell", abs(s))
    m = kwargs.pop("m", -ell)
    ell_min, ell_max = kwargs.pop("ell_min", abs(s)), kwargs.pop("ell_max", 8)
    data_type = kwargs.pop("data_type", DataType[SpinWeights.index(s)])
    t_0, t_1, dt = kwargs.pop("t_0", -20.0), kwargs.pop("t_1", 20.0), kwargs.pop("dt", 1.0 / 10.0)
    t = np.arange(t_0, t_1 + dt, dt)
    data = np.zeros((t.size, LM_total_size(ell_min, ell_max)), dtype=complex)
    data[:, LM_index(ell, m, ell_min)] = values_of_t(t)
    make = lambda d, ctx=None: WaveformModes(t=t, data=d, ell_min=ell_min, ell_max=ell_max, frameType=Inertial, dataType=data_type,  # noqa: E731
                                             r_is_scaled_out=True, m_is_scaled_out=True, ctx=ctx)
    return s, ell, m, ell_min, ell_max, t, data, make


def single_mode_constant_rotation(**kwargs):
    """one nonzero mode exp(i omega t) (omega may be complex: damping); s, ell, m, ell_min, ell_max, data_type, t_0, t_1, dt, omega"""
    omega = complex(kwargs.pop("omega", 0.5))
    ctx = kwargs.pop("ctx", None)
    *_, data, make = _single_mode_series(kwargs, lambda t: np.exp(1j * omega * t))
    _unused(kwargs)
    return make(data, ctx)


def single_mode_proportional_to_time(**kwargs):
    """one nonzero mode beta t"""
    beta = kwargs.pop("beta", 1.0)
    ctx = kwargs.pop("ctx", None)
    *_, data, make = _single_mode_series(kwargs, lambda t: beta * t)
    _unused(kwargs)
    return make(data, ctx)


def single_mode_proportional_to_time_supertranslated(**kwargs):
    """single_mode_proportional_to_time after an analytically applied supertranslation (`supertranslation` modes, or
    `space_translation`, default none): a mode beta t sY_lm seen at u - alpha picks up -beta alpha sY_lm, whose modes are the
    Gaunt coefficients of alpha_l''m'' against (l, m) -- sqrt((2l''+1)(2l+1)(2l'+1)/4pi) times two 3-j symbols and a sign."""
    beta = kwargs.pop("beta", 1.0)
    ctx = kwargs.pop("ctx", None)
    supertranslation = np.array(kwargs.pop("supertranslation", np.array([], dtype=complex)), dtype=complex)
    if "space_translation" in kwargs:
        if supertranslation.size < 4:
            supertranslation = np.concatenate([supertranslation, np.zeros(4 - supertranslation.size, dtype=complex)])
        supertranslation[1:4] = -vector_as_ell_1_modes(kwargs.pop("space_translation"))
    s, ell, m, ell_min, ell_max, t, data, make = _single_mode_series(kwargs, lambda t_: beta * t_)
    _unused(kwargs)
    lst = int(math.sqrt(supertranslation.size) - 1) if supertranslation.size else -1
    if supertranslation.size and lst * (lst + 2) + 1 != supertranslation.size:
        raise ValueError(f"Bad number of elements in supertranslation: {supertranslation.size}")
    for i, (ellpp, mpp) in enumerate(LM_range(0, lst) if lst >= 0 else []):
        if supertranslation[i] == 0.0:
            continue
        mp = m + mpp
        for ellp in range(ell_min, min(ell_max, ell + ellpp) + 1):
            if ellp < abs(mp):
                continue
            term = (beta * supertranslation[i] * math.sqrt(((2 * ellpp + 1) * (2 * ell + 1) * (2 * ellp + 1)) / (4 * math.pi))
                    * wigner_3j(ellpp, ell, ellp, 0, -s, s) * wigner_3j(ellpp, ell, ellp, mpp, m, -mp))
            data[:, LM_index(ellp, mp, ell_min)] += -term if (s + mp) % 2 == 1 else term
    return make(data, ctx)
