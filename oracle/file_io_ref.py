"""TEST INFRASTRUCTURE ONLY (parity unpinned: the reference has no test or fixture for this path).
CPU restatement of the array-level part of scri/SpEC/file_io/__init__.py: the monotonic-time selection (:51-78), the
Christodoulou-mass scaling (:421-450) and the assembly of AsymptoticBondiData fields with the SpEC -> Moreschi-Boyle
convention factors (:733-811), written with plain loops / numpy on [6, N, (ell_max+1)^2] arrays."""
import numpy as np

# DataType indices of scri/__init__.py:82 : psi0..psi4 = 1..5, h = 7
DATATYPE = {"Psi0": 1, "Psi1": 2, "Psi2": 3, "Psi3": 4, "Psi4": 5, "h": 7, "Strain": 7}
FIELD = {"Psi0": 0, "Psi1": 1, "Psi2": 2, "Psi3": 3, "Psi4": 4, "h": 5, "Strain": 5}  # row of the [6, N, modes] storage


def index_is_monotonic(y):
    """scri/SpEC/file_io/__init__.py:51-69, literally"""
    length = y.size
    monotonic = np.ones_like(y, dtype=np.bool_)
    direction = y[-1] - y[0]
    if direction > 0.0:
        max_value = y[0]
        for i in range(1, length):
            if y[i] <= max_value:
                monotonic[i] = False
            else:
                max_value = y[i]
    else:
        min_value = y[0]
        for i in range(1, length):
            if y[i] >= min_value:
                monotonic[i] = False
            else:
                min_value = y[i]
    return monotonic


def assemble(t, fields, ell_mins, ell_max, convention="spec", time_shift=0.0, ch_mass=None, m_is_scaled_out=False):
    """fields: dict label -> c16[N, (ell_max+1)^2 - ell_min^2].  Returns (u, raw[6, N', (ell_max+1)^2])."""
    factor = {"moreschi-boyle": [1, 1, 1, 1, 1, 1], "spec": [2, -np.sqrt(2), 1, -1 / np.sqrt(2), 0.5, 0.5]}[convention]
    t = np.array(t, dtype=float)
    scale = {k: 1.0 for k in fields}
    if ch_mass is not None and not m_is_scaled_out:
        for k in fields:
            dt = DATATYPE[k]
            scale[k] = ch_mass ** (dt - 4) if dt <= 5 else 1 / ch_mass  # :439-442
        t = t / ch_mass
    t = t - time_shift
    keep = index_is_monotonic(t)
    u = t[keep]
    raw = np.zeros((6, u.size, (ell_max + 1) ** 2), dtype=complex)
    for k, data in fields.items():
        f = FIELD[k]
        raw[f][:, ell_mins[k] ** 2 :] = factor[f] * (data * scale[k])[keep]  # mass scaling first (:449), then the factor
        if f == 5:
            # sigma = conjugate of the rescaled strain as a FUNCTION: mode (l, m) -> (-1)^(s+m) conj(mode (l, -m)), s = -2
            out = np.zeros_like(raw[5])
            for l in range(ell_max + 1):
                for m in range(-l, l + 1):
                    out[:, l * l + l + m] = (-1) ** (m) * np.conj(raw[5][:, l * l + l - m])
            raw[5] = out
    return u, raw
