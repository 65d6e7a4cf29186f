"""Mode-index algebra and the l <= 1 helpers of spherical_functions used while parsing
transformation parameters (sf.LM_index, sf.constant_as_ell_0_mode, sf.vector_as_ell_1_modes, ...;
call sites scri/waveform_grid.py:49-86, scri/asymptotic_bondi_data/transformations.py:39-73)."""
import math
import numpy as np


def LM_index(ell, m, ell_min):
    return ell * (ell + 1) - ell_min**2 + m


def LM_total_size(ell_min, ell_max):
    return (ell_max + 1) ** 2 - ell_min**2


def LM_range(ell_min, ell_max):
    return np.array([[ell, m] for ell in range(ell_min, ell_max + 1) for m in range(-ell, ell + 1)], dtype=int)


def constant_as_ell_0_mode(c):
    return c * math.sqrt(4 * math.pi)


def constant_from_ell_0_mode(mode):
    return mode / math.sqrt(4 * math.pi)


def vector_as_ell_1_modes(v):
    v = np.asarray(v, dtype=float)
    return np.array(
        [
            (v[0] + 1j * v[1]) * math.sqrt(2 * math.pi / 3.0),
            v[2] * math.sqrt(4 * math.pi / 3.0) + 0j,
            (-v[0] + 1j * v[1]) * math.sqrt(2 * math.pi / 3.0),
        ]
    )


def vector_from_ell_1_modes(modes):
    modes = np.asarray(modes, dtype=complex)
    return np.array(
        [
            (modes[0] - modes[2]) / (2 * math.sqrt(2 * math.pi / 3.0)),
            (modes[0] + modes[2]) / (2j * math.sqrt(2 * math.pi / 3.0)),
            modes[1] / math.sqrt(4 * math.pi / 3.0),
        ]
    )
