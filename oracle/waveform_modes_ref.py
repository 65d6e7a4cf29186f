"""CPU restatement (TEST INFRASTRUCTURE ONLY: imported by tests/, never by the product) of the mode-space operators of
scri.WaveformModes that sit either side of the transformation path: spin raising / lowering (scri/waveform_modes.py:478-572),
the parity conjugates and their (anti)symmetric parts and violation measures (:724-943), the conjugate-pair storage form
(:659-703), precision truncation (:458-476) and the all-angles inner product (:574-656).

Plain loops over (l, m) exactly as the reference writes them.  The operations on the `frame` quaternions are numpy-quaternion
ufuncs (x/y/z_parity_conjugate, ..._symmetric_part, ..._antisymmetric_part), a third-party package that is in neither tree nor
image: restated from its published definitions (reflection of a rotor along an axis keeps the scalar part and that axis'
component and flips the other two; the full parity leaves a rotor unchanged) -- parity with the package itself is unpinned; the
reference's tests/test_waveform.py:273-342 and tests/test_parity.py (involution, idempotence, null compositions, violation
measures) pin the structure and are mirrored in tests/test_oracle_mode_operators.py.
"""
from dataclasses import replace

import numpy as np
from scipy.interpolate import CubicSpline

from .containers import WM, UnknownDataType


def LM_index(ell, m, ell_min):
    return ell * (ell + 1) - ell_min**2 + m


# ------------------------------------------------------------------------------------------- eth / ethbar
def ladder_factor(operations, s, ell, eth_convention="NP"):
    """scri/waveform_modes.py:478-529.  `operations` is applied right to left."""
    op_dict = {"ð": +1, "ð̅": -1, "+": +1, "-": -1, +1: +1, -1: -1}
    conv_dict = {"NP": 1.0, "GHP": 0.5}
    if isinstance(operations, str):
        operations = operations.replace("ð̅", "-").replace("ð", "+")
    if not set(operations).issubset(op_dict.keys()):
        raise ValueError("operations must be a string composed of ...")
    if eth_convention not in conv_dict:
        raise ValueError("eth_convention must be one of {}".format(set(conv_dict.keys())))
    convention_factor = conv_dict[eth_convention]
    ladder = 1.0
    sign_factor = 1.0
    for op in reversed(operations):
        sign = op_dict[op]
        sign_factor *= sign
        ladder *= (ell - s * sign) * (ell + s * sign + 1.0) if (ell >= abs(s)) else 0.0
        ladder *= convention_factor
        s += sign
    return sign_factor * np.sqrt(ladder)


def apply_eth(w, operations, eth_convention="NP"):
    """scri/waveform_modes.py:531-562: an array shaped like w.data, same (l, m) layout."""
    s = w.spin_weight
    mode_data = w.data.copy()
    for ell in range(w.ell_min, w.ell_max + 1):
        f = ladder_factor(operations, s, ell, eth_convention=eth_convention)
        idx = [LM_index(ell, m, w.ell_min) for m in range(-ell, ell + 1)]
        mode_data[:, idx] *= f
    return mode_data


# ------------------------------------------------------------------------------------------- parity
def _frame_op(frame, keep_sign):
    """component-wise sign pattern on rotors [n, 4] (w, x, y, z); an empty frame stays empty"""
    frame = np.asarray(frame, dtype=float).reshape(-1, 4)
    return frame * np.asarray(keep_sign, dtype=float)[None, :]


FRAME_CONJUGATE = {"x": (1, 1, -1, -1), "y": (1, -1, 1, -1), "z": (1, -1, -1, 1), "": (1, 1, 1, 1)}
FRAME_SYMMETRIC = {"x": (1, 1, 0, 0), "y": (1, 0, 1, 0), "z": (1, 0, 0, 1), "": (1, 1, 1, 1)}
FRAME_ANTISYMMETRIC = {"x": (0, 0, 1, 1), "y": (0, 1, 0, 1), "z": (0, 1, 1, 0), "": (0, 0, 0, 0)}


def parity_conjugate(w, direction=""):
    """x: scri/waveform_modes.py:724-744; y: :780-795; z: :831-854; all axes: :890-911"""
    if w.dataType == UnknownDataType:
        raise ValueError("Cannot compute parity type for UnknownDataType.")
    out = np.empty_like(w.data)
    s = w.spin_weight
    for ell in range(w.ell_min, w.ell_max + 1):
        ms = list(range(-ell, ell + 1))
        idx = [LM_index(ell, m, w.ell_min) for m in ms]
        if direction == "x":
            for m, i in zip(ms, idx):
                out[:, i] = np.conjugate(w.data[:, i]) if m % 2 == 0 else -np.conjugate(w.data[:, i])
        elif direction == "y":
            out[:, idx] = np.conjugate(w.data[:, idx])
        elif direction == "z":
            rev = list(reversed(idx))
            out[:, idx] = np.conjugate(w.data[:, rev]) if (ell + s) % 2 == 0 else -np.conjugate(w.data[:, rev])
        elif direction == "":
            # the reference pairs the even (l + s + m) indices with their own reversed list, and the odd ones likewise: since l + s + m
            # and l + s - m have the same parity, element m takes +-conj of element -m
            for m, i in zip(ms, idx):
                j = LM_index(ell, -m, w.ell_min)
                out[:, i] = np.conjugate(w.data[:, j]) if (ell + s + m) % 2 == 0 else -np.conjugate(w.data[:, j])
        else:
            raise ValueError(direction)
    return replace(w, t=w.t.copy(), data=out, frame=_frame_op(w.frame, FRAME_CONJUGATE[direction]))


def parity_symmetric_part(w, direction=""):
    """scri/waveform_modes.py:748-757 (x), :799-808 (y), :858-867 (z), :915-924 (all)"""
    c = parity_conjugate(w, direction)
    return replace(c, data=0.5 * (w.data + c.data), frame=_frame_op(w.frame, FRAME_SYMMETRIC[direction]))


def parity_antisymmetric_part(w, direction=""):
    """scri/waveform_modes.py:759-767 (x), :810-818 (y), :869-877 (z), :926-934 (all)"""
    c = parity_conjugate(w, direction)
    return replace(c, data=0.5 * (w.data - c.data), frame=_frame_op(w.frame, FRAME_ANTISYMMETRIC[direction]))


def norm(w):
    """scri/waveform_base.py:19-26,535-551 (complex_array_norm): s[i] += c[i, j].real ** 2 + c[i, j].imag ** 2, j in order.
    The reference's loop is compiled by numba, which lowers `x ** 2` with the literal exponent to the product x * x (correctly
    rounded); CPython's float ** 2 goes through libm's pow and differs from it by an ulp in about 1 of 1200 values -- so the
    squares are written as products here (and the golden vectors, produced by running the reference's source WITHOUT numba,
    pin the norms to 1e-15 only)."""
    s = np.zeros(w.data.shape[0])
    for i in range(w.data.shape[0]):
        for j in range(w.data.shape[1]):
            re, im = float(w.data[i, j].real), float(w.data[i, j].imag)
            s[i] += re * re + im * im
    return s


def parity_violation_squared(w, direction=""):
    return norm(parity_antisymmetric_part(w, direction))


def parity_violation_normalized(w, direction=""):
    return np.sqrt(norm(parity_antisymmetric_part(w, direction)) / norm(w))


# ------------------------------------------------------------------------------------------- conjugate pairs
def convert_to_conjugate_pairs(w):
    """scri/waveform_modes.py:659-685 (returns a new container; the reference works in place)"""
    data = w.data.copy()
    for ell in range(w.ell_min, w.ell_max + 1):
        for m in range(1, ell + 1):
            ip, im = LM_index(ell, m, w.ell_min), LM_index(ell, -m, w.ell_min)
            plus, minus = w.data[..., ip].copy(), w.data[..., im].copy()
            data[..., ip] = (plus + np.conjugate(minus)) / np.sqrt(2)
            data[..., im] = (plus - np.conjugate(minus)) / np.sqrt(2)
    return replace(w, data=data)


def convert_from_conjugate_pairs(w):
    """scri/waveform_modes.py:688-703"""
    data = w.data.copy()
    for ell in range(w.ell_min, w.ell_max + 1):
        for m in range(1, ell + 1):
            ip, im = LM_index(ell, m, w.ell_min), LM_index(ell, -m, w.ell_min)
            plus, minus = w.data[..., ip].copy(), w.data[..., im].copy()
            data[..., ip] = (plus + minus) / np.sqrt(2)
            data[..., im] = np.conjugate(plus - minus) / np.sqrt(2)
    return replace(w, data=data)


# ------------------------------------------------------------------------------------------- truncate, inner product
def truncate(w, tol=1e-10):
    """scri/waveform_modes.py:458-476: bits below tol / sqrt(n_modes) of the norm at each instant are set to zero"""
    data = w.data.copy()
    if tol != 0.0:
        tol_per_mode = tol / np.sqrt(data.shape[1])
        absolute_tolerance = np.linalg.norm(data, axis=1) * tol_per_mode
        power_of_2 = (2.0 ** np.floor(-np.log2(absolute_tolerance)))[:, np.newaxis]
        data *= power_of_2
        np.round(data, out=data)
        data /= power_of_2
    return replace(w, data=data)


def inner_product(a, b, t1=None, t2=None):
    """scri/waveform_modes.py:574-656 for equal mode sets and times: the definite integral over [t1, t2] of
    sum_lm conj(a_lm) b_lm, by quaternion.calculus.spline_definite_integral = the integral of the cubic spline through the
    integrand's samples (third-party; restated with scipy's not-a-knot CubicSpline, which is what that routine wraps)."""
    if a.spin_weight != b.spin_weight:
        raise ValueError("Spin weights must match in inner_product")
    if a.ell_min != b.ell_min or a.ell_max != b.ell_max:
        raise ValueError("ell_min and ell_max must match in inner_product (use allow_LM_differ=True to override)")
    if not np.array_equal(a.t, b.t):
        raise ValueError("Time samples must match in inner_product (use allow_times_differ=True to override)")
    t1 = a.t[0] if t1 is None else t1
    t2 = a.t[-1] if t2 is None else t2
    integrand = np.sum(np.conj(a.data) * b.data, axis=1)
    return CubicSpline(a.t, integrand).integrate(t1, t2)


def intersection(t1, t2, min_step=None, min_time=None, max_time=None):
    """scri/extrapolation.py:47-122, the time axis inner_product(..., allow_times_differ=True) interpolates both series to.
    Kept in the reference's shape (reserved array, running indices I, I1, I2) as the check of the product's own loop."""
    t1, t2 = np.asarray(t1), np.asarray(t2)
    if t1.size == 0 or t2.size == 0:
        raise ValueError("empty time series")
    t = np.empty(t1.size + t2.size)
    mint = max(t1[0], t2[0]) if min_time is None else max(max(t1[0], t2[0]), min_time)
    maxt = min(t1[-1], t2[-1]) if max_time is None else min(min(t1[-1], t2[-1]), max_time)
    if mint > t1[-1] or mint > t2[-1] or maxt < t1[0] or maxt < t2[0]:
        raise ValueError("Empty intersection")
    if min_step is None:
        min_step = min(np.min(np.diff(t1)), np.min(np.diff(t2)))
    t[0] = mint
    I = I1 = I2 = 0
    while t[I] < maxt:
        if t[I] < t1[0] or t[I] > t1[-1]:
            I1 = 0
        else:
            I1 = max(I1, 1)
            while t[I] > t1[I1] and I1 < t1.size:
                I1 += 1
        if t[I] < t2[0] or t[I] > t2[-1]:
            I2 = 0
        else:
            I2 = max(I2, 1)
            while t[I] > t2[I2] and I2 < t2.size:
                I2 += 1
        t[I + 1] = t[I] + max(min(t1[I1] - t1[I1 - 1], t2[I2] - t2[I2 - 1]), min_step)
        I += 1
        if t[I] > maxt:
            break
    return t[:I]


def compare_data(w_b, w_a):
    """The data half of WaveformBase.compare (scri/waveform_base.py:577-687): A - B on intersection(B.t, A.t), each mode through its
    own real and imaginary CubicSpline (:675-683); returns (times, data)."""
    times = intersection(w_b.t, w_a.t)
    out = np.zeros((times.shape[0], w_a.data.shape[1]), dtype=complex)
    for a_mode in range(w_a.data.shape[1]):
        b_mode = a_mode  # same (l, m) order on both sides in the oracle's container
        re_a, im_a = CubicSpline(w_a.t, w_a.data[:, a_mode].real), CubicSpline(w_a.t, w_a.data[:, a_mode].imag)
        re_b, im_b = CubicSpline(w_b.t, w_b.data[:, b_mode].real), CubicSpline(w_b.t, w_b.data[:, b_mode].imag)
        out[:, a_mode] = (re_a(times) - re_b(times)) + 1j * (im_a(times) - im_b(times))
    return times, out
