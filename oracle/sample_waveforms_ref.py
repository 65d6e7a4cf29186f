"""Synthetic inputs + analytic known answers, after scri/sample_waveforms.py:12-380 and
tests/conftest.py:55-192 of the reference.  Seeds are explicit integers (the reference seeds
with the per-process-salted ``hash(str)``).  Test infrastructure only.
"""
import math
import functools
import numpy as np

from .containers import WM, DataType, SpinWeights, Inertial, Corotating, h, psi4
from .wigner import LM_index, LM_total_size, LM_range, vector_as_ell_1_modes


@functools.lru_cache(maxsize=None)
def _w3j(j1, j2, j3, m1, m2, m3):
    from sympy.physics.wigner import wigner_3j

    return float(wigner_3j(j1, j2, j3, m1, m2, m3))


def _modes(t, data, ell_min, ell_max, dataType=h, frameType=Inertial, frame=None, r=True, m=True):
    return WM(
        t=np.asarray(t, dtype=float),
        data=data,
        ell_min=ell_min,
        ell_max=ell_max,
        dataType=dataType,
        frameType=frameType,
        r_is_scaled_out=r,
        m_is_scaled_out=m,
        frame=np.zeros((0, 4)) if frame is None else frame,
    )


def constant_waveform(t=None, ell_min=2, ell_max=8):
    """scri/sample_waveforms.py:60-77: data[:, (l,m)] = m - i m, Inertial, h."""
    t = np.linspace(-10.0, 100.0, num=1101) if t is None else t
    LM = LM_range(ell_min, ell_max)
    data = np.zeros((t.shape[0], LM.shape[0]), dtype=complex)
    for i, m in enumerate(LM[:, 1]):
        data[:, i] = m - 1j * m
    return _modes(t, data, ell_min, ell_max)


def linear_waveform(begin=-10.0, end=100.0, n_times=1000, ell_min=2, ell_max=8, seed=11):
    """tests/conftest.py:82-111."""
    rng = np.random.default_rng(seed)
    axis = rng.uniform(-1, 1, size=3)
    axis /= np.linalg.norm(axis)
    t = np.linspace(begin, end, num=n_times)
    omega = 2 * np.pi * 4 / (t[-1] - t[0])
    ang = omega * t / 2
    frame = np.concatenate([np.cos(ang)[:, None], np.sin(ang)[:, None] * axis[None, :]], axis=1)
    LM = LM_range(ell_min, ell_max)
    data = np.empty((t.shape[0], LM.shape[0]), dtype=complex)
    for i, m in enumerate(LM[:, 1]):
        data[:, i] = (m - 1j * m) * t
    return _modes(t, data, ell_min, ell_max, frameType=Corotating, frame=frame)


def random_waveform(begin=-10.0, end=100.0, n_times=1000, ell_min=2, ell_max=8, seed=12):
    """tests/conftest.py:114-140 (white noise in time: use for rotation tests only)."""
    rng = np.random.default_rng(seed)
    n_modes = LM_total_size(ell_min, ell_max)
    t = np.sort(rng.uniform(begin, end, size=n_times))
    frame = rng.uniform(-1, 1, size=(n_times, 4))
    frame /= np.linalg.norm(frame, axis=1)[:, None]
    data = rng.normal(size=(n_times, n_modes)) + 1j * rng.normal(size=(n_times, n_modes))
    return _modes(t, data, ell_min, ell_max, frameType=Corotating, frame=frame, m=False)


def delta_waveform(ell, m, begin=-10.0, end=100.0, n_times=1000, ell_min=2, ell_max=8):
    """tests/conftest.py:143-165."""
    t = np.linspace(begin, end, num=n_times)
    data = np.zeros((n_times, LM_total_size(ell_min, ell_max)), dtype=complex)
    data[:, LM_index(ell, m, ell_min)] = 1.0 + 0.0j
    return _modes(t, data, ell_min, ell_max, dataType=psi4, r=False, m=True)


def Rs(seed=13):
    """tests/conftest.py:173-179: 80 special + 20 random unit quaternions, [100, 4]."""
    ones = [0, -1.0, 1.0]
    rs = [np.array([w, x, y, z]) for w in ones for x in ones for y in ones for z in ones][1:]
    rs = [r / np.linalg.norm(r) for r in rs]
    rng = np.random.default_rng(seed)
    for _ in range(20):
        q = rng.uniform(-1, 1, 4)
        rs.append(q / np.linalg.norm(q))
    return np.array(rs)


def random_waveform_proportional_to_time(begin=-10.0, end=100.0, n_times=1101, ell_min=2, ell_max=8, seed=14):
    """scri/sample_waveforms.py:149-191 with rotating=False: data = outer(t, random modes), random times."""
    rng = np.random.default_rng(seed)
    t = np.sort(rng.uniform(begin, end, size=n_times))
    n_modes = LM_total_size(ell_min, ell_max)
    c = rng.normal(size=n_modes) + 1j * rng.normal(size=n_modes)
    return _modes(t, np.outer(t, c), ell_min, ell_max)


def _single_mode_setup(s, ell, m, ell_min, ell_max, data_type, t_0, t_1, dt):
    ell = abs(s) if ell is None else ell
    m = -ell if m is None else m
    ell_min = abs(s) if ell_min is None else ell_min
    data_type = DataType[SpinWeights.index(s)] if data_type is None else data_type
    t = np.arange(t_0, t_1 + dt, dt)
    return ell, m, ell_min, data_type, t


def single_mode_constant_rotation(s=-2, ell=None, m=None, ell_min=None, ell_max=8, data_type=None,
                                  t_0=-20.0, t_1=20.0, dt=0.1, omega=0.5):
    """scri/sample_waveforms.py:194-251: one mode = exp(i omega t)."""
    ell, m, ell_min, data_type, t = _single_mode_setup(s, ell, m, ell_min, ell_max, data_type, t_0, t_1, dt)
    data = np.zeros((t.size, LM_total_size(ell_min, ell_max)), dtype=complex)
    data[:, LM_index(ell, m, ell_min)] = np.exp(1j * complex(omega) * t)
    return _modes(t, data, ell_min, ell_max, dataType=data_type)


def single_mode_proportional_to_time(s=-2, ell=None, m=None, ell_min=None, ell_max=8, data_type=None,
                                     t_0=-20.0, t_1=20.0, dt=0.1, beta=1.0):
    """scri/sample_waveforms.py:254-309: one mode = beta t."""
    ell, m, ell_min, data_type, t = _single_mode_setup(s, ell, m, ell_min, ell_max, data_type, t_0, t_1, dt)
    data = np.zeros((t.size, LM_total_size(ell_min, ell_max)), dtype=complex)
    data[:, LM_index(ell, m, ell_min)] = beta * t
    return _modes(t, data, ell_min, ell_max, dataType=data_type)


def single_mode_proportional_to_time_supertranslated(s=-2, ell=None, m=None, ell_min=None, ell_max=8, data_type=None,
                                                     t_0=-20.0, t_1=20.0, dt=0.1, beta=1.0,
                                                     supertranslation=None, space_translation=None):
    """scri/sample_waveforms.py:312-380: analytic supertranslation via Wigner-3j (sympy replaces sf.Wigner3j)."""
    ell, m, ell_min, data_type, t = _single_mode_setup(s, ell, m, ell_min, ell_max, data_type, t_0, t_1, dt)
    data = np.zeros((t.size, LM_total_size(ell_min, ell_max)), dtype=complex)
    data[:, LM_index(ell, m, ell_min)] = beta * t
    st = np.array([] if supertranslation is None else supertranslation, dtype=complex)
    if space_translation is not None:
        if st.size < 4:
            st = np.concatenate([st, np.zeros(4 - st.size, dtype=complex)])
        st[1:4] = -vector_as_ell_1_modes(space_translation)
    st_ell_max = int(math.sqrt(st.size) - 1)
    if st_ell_max * (st_ell_max + 2) + 1 != st.size:
        raise ValueError(f"Bad number of elements in supertranslation: {st.size}")
    for i, (ellpp, mpp) in enumerate(LM_range(0, st_ell_max)):
        ellpp, mpp = int(ellpp), int(mpp)
        if st[i] != 0.0:
            mp = m + mpp
            for ellp in range(ell_min, min(ell_max, (ell + ellpp)) + 1):
                if ellp >= abs(mp):
                    addition = (
                        beta
                        * st[i]
                        * math.sqrt(((2 * ellpp + 1) * (2 * ell + 1) * (2 * ellp + 1)) / (4 * math.pi))
                        * _w3j(ellpp, ell, ellp, 0, -s, s)
                        * _w3j(ellpp, ell, ellp, mpp, m, -mp)
                    )
                    if (s + mp) % 2 == 1:
                        addition *= -1
                    data[:, LM_index(ellp, mp, ell_min)] += addition
    return _modes(t, data, ell_min, ell_max, dataType=data_type)
