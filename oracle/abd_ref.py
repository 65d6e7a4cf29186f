"""Oracle for the AsymptoticBondiData BMS transform:
scri/asymptotic_bondi_data/transformations.py:8-97 (_process_transformation_kwargs),
:100-148 (boosted_grid), :151-196 (conformal_factors), :199-431 (transform), and
scri/bms_transformations.py:151-180 (transform_supertranslation).  numpy + live scipy
CubicSpline, statement-by-statement.  Test infrastructure only.
"""
import math
import numpy as np
from scipy.interpolate import CubicSpline

from . import quat
from .containers import ABD
from .wigner import (
    LM_index,
    swsh_grid,
    modes_evaluate,
    constant_as_ell_0_mode,
    constant_from_ell_0_mode,
    vector_as_ell_1_modes,
    vector_from_ell_1_modes,
    eth_NP,
)
from .spinsfast_ref import map2salm_matrix
from .waveform_grid_ref import rotor_grid


def process_transformation_kwargs(input_ell_max, **kwargs):
    """scri/asymptotic_bondi_data/transformations.py:8-97 (reality is IMPOSED, not checked)."""
    supertranslation = np.zeros((4,), dtype=complex)
    ell_max_supertranslation = 1
    if "supertranslation" in kwargs:
        supertranslation = np.array(kwargs.pop("supertranslation"), dtype=complex)
        if supertranslation.size <= 4:
            supertranslation = np.pad(supertranslation, (0, 4 - supertranslation.size), "constant", constant_values=(0.0,))
        ell_max_supertranslation = int(np.sqrt(len(supertranslation))) - 1
        if (ell_max_supertranslation + 1) ** 2 != len(supertranslation):
            raise ValueError("Input supertranslation parameter must contain modes from ell=0 up to some ell_max")
        for ell in range(ell_max_supertranslation + 1):
            for m in range(ell + 1):
                i_pos = LM_index(ell, m, 0)
                i_neg = LM_index(ell, -m, 0)
                a = supertranslation[i_pos]
                b = supertranslation[i_neg]
                supertranslation[i_pos] = (a + (-1.0) ** m * b.conjugate()) / 2.0
                supertranslation[i_neg] = (-1.0) ** m * supertranslation[i_pos].conjugate()
    spacetime_translation = np.zeros((4,), dtype=float)
    spacetime_translation[0] = constant_from_ell_0_mode(supertranslation[0]).real
    spacetime_translation[1:4] = -vector_from_ell_1_modes(supertranslation[1:4]).real
    if "spacetime_translation" in kwargs:
        st_trans = np.array(kwargs.pop("spacetime_translation"), dtype=float)
        if st_trans.shape != (4,):
            raise TypeError("Input argument `spacetime_translation` should be a float array of shape (4,).")
        spacetime_translation = st_trans[:]
        supertranslation[0] = constant_as_ell_0_mode(spacetime_translation[0])
        supertranslation[1:4] = vector_as_ell_1_modes(-spacetime_translation[1:4])
    if "space_translation" in kwargs:
        s_trans = np.array(kwargs.pop("space_translation"), dtype=float)
        if s_trans.shape != (3,):
            raise TypeError("Input argument `space_translation` should be an array of floats of shape (3,).")
        spacetime_translation[1:4] = s_trans[:]
        supertranslation[1:4] = vector_as_ell_1_modes(-spacetime_translation[1:4])
    if "time_translation" in kwargs:
        t_trans = kwargs.pop("time_translation")
        if not isinstance(t_trans, float):
            raise TypeError("Input argument `time_translation` should be a single float.")
        spacetime_translation[0] = t_trans
        supertranslation[0] = constant_as_ell_0_mode(spacetime_translation[0])

    output_ell_max = kwargs.pop("output_ell_max", input_ell_max)
    working_ell_max = kwargs.pop("working_ell_max", 2 * input_ell_max + ell_max_supertranslation)
    if working_ell_max < input_ell_max:
        raise ValueError(f"working_ell_max={working_ell_max} is too small; it must be at least ell_max={input_ell_max}")
    frame_rotation = np.array(kwargs.pop("frame_rotation", [1, 0, 0, 0]), dtype=float)
    if quat.qabs(frame_rotation) < 3e-16:
        raise ValueError(f"frame_rotation={frame_rotation} should be a single unit quaternion")
    frame_rotation = quat.qnormalized(frame_rotation)
    boost_velocity = np.array(kwargs.pop("boost_velocity", [0.0] * 3), dtype=float)
    beta = np.linalg.norm(boost_velocity)
    if boost_velocity.shape != (3,) or beta >= 1.0:
        raise ValueError("Input boost_velocity should be a 3-vector with magnitude strictly less than 1.0")
    return frame_rotation, boost_velocity, supertranslation, working_ell_max, output_ell_max


def boosted_grid(frame_rotation, boost_velocity, n_theta, n_phi):
    """scri/asymptotic_bondi_data/transformations.py:100-148."""
    return rotor_grid(frame_rotation, boost_velocity, n_theta, n_phi)


def conformal_factors(boost_velocity, distorted_grid_rotors):
    """scri/asymptotic_bondi_data/transformations.py:151-196 -> k, eth k / k, 1/k, 1/k^3 as [1, n_theta, n_phi]."""
    boost_velocity = np.asarray(boost_velocity, dtype=float)
    beta = np.linalg.norm(boost_velocity)
    gamma = 1 / math.sqrt(1 - beta**2)
    shape = distorted_grid_rotors.shape[:-1]
    v_dot_r = np.dot(quat.rotate_z(distorted_grid_rotors.reshape(-1, 4)), boost_velocity).reshape(shape)[np.newaxis]
    eth_v_dot_r = modes_evaluate(np.insert(vector_as_ell_1_modes(boost_velocity), 0, 0.0), distorted_grid_rotors, 1)[np.newaxis]
    one_over_k = gamma * (1 - v_dot_r)
    k = 1.0 / one_over_k
    ethk_over_k = eth_v_dot_r / (1 - v_dot_r)
    return k, ethk_over_k, one_over_k, one_over_k**3


def transform(abd, **kwargs):
    """scri/asymptotic_bondi_data/transformations.py:199-431."""
    frame_rotation, boost_velocity, supertranslation, working_ell_max, output_ell_max = process_transformation_kwargs(
        abd.ell_max, **kwargs
    )
    n_theta = 2 * working_ell_max + 1
    n_phi = n_theta
    beta = np.linalg.norm(boost_velocity)
    gamma = 1 / math.sqrt(1 - beta**2)

    rotors = boosted_grid(frame_rotation, boost_velocity, n_theta, n_phi)
    u = abd.u
    alpha = modes_evaluate(supertranslation, rotors, 0).real[np.newaxis]
    eth_alpha = (modes_evaluate(eth_NP(supertranslation, 0), rotors, 1) / np.sqrt(2))[np.newaxis]
    etheth_alpha = (0.5 * modes_evaluate(eth_NP(eth_NP(supertranslation, 0), 1), rotors, 2))[np.newaxis]
    k, ethk_over_k, one_over_k, one_over_k_cubed = conformal_factors(boost_velocity, rotors)

    X = ethk_over_k * (u[:, np.newaxis, np.newaxis] - alpha) - eth_alpha  # eth u' / k

    psi = [modes_evaluate(abd.raw[i], rotors, s) for i, s in enumerate(ABD.spins)]
    psi0, psi1, psi2, psi3, psi4, sigma = psi

    fprime = np.empty((6, abd.n_times, n_theta, n_phi), dtype=complex)
    tmp = psi4.copy()
    tmp *= X
    tmp += -4 * psi3
    tmp *= X
    tmp += 6 * psi2
    tmp *= X
    tmp += -4 * psi1
    tmp *= X
    tmp += psi0
    tmp *= one_over_k_cubed
    fprime[0] = tmp
    tmp = -psi4
    tmp *= X
    tmp += 3 * psi3
    tmp *= X
    tmp += -3 * psi2
    tmp *= X
    tmp += psi1
    tmp *= one_over_k_cubed
    fprime[1] = tmp
    tmp = psi4.copy()
    tmp *= X
    tmp += -2 * psi3
    tmp *= X
    tmp += psi2
    tmp *= one_over_k_cubed
    fprime[2] = tmp
    tmp = -psi4
    tmp *= X
    tmp += psi3
    tmp *= one_over_k_cubed
    fprime[3] = tmp
    tmp = psi4.copy()
    tmp *= one_over_k_cubed
    fprime[4] = tmp
    tmp = sigma.copy()
    tmp -= etheth_alpha
    tmp *= one_over_k
    fprime[5] = tmp

    timeprime = (u - constant_from_ell_0_mode(supertranslation[0]).real) / gamma
    earliest = np.max(k * (u[0] - alpha))
    latest = np.min(k * (u[-1] - alpha))
    timeprime = timeprime[(timeprime >= earliest) & (timeprime <= latest)]

    out = np.zeros((6, timeprime.size, n_theta, n_phi), dtype=complex)
    for i in range(n_theta):
        for j in range(n_phi):
            k_i_j = k[0, i, j]
            a_i_j = alpha[0, i, j]
            out[:, :, i, j] = CubicSpline(k_i_j * (u - a_i_j), fprime[:, :, i, j], axis=1)(timeprime)

    raw = np.zeros((6, timeprime.size, (output_ell_max + 1) ** 2), dtype=complex)
    for i, s in enumerate(ABD.spins):
        A = map2salm_matrix(s, output_ell_max, n_theta, n_phi)
        raw[i] = out[i].reshape(timeprime.size, -1) @ A.T
    return ABD(timeprime, raw, output_ell_max)


def transform_supertranslation(S, frame_rotation, boost_velocity, ell_max):
    """scri/bms_transformations.py:151-180 with the already-inverted Lorentz transformation
    (frame_rotation, boost_velocity in "frame_rotation then boost_velocity" order) passed in."""
    n_theta = 2 * ell_max + 1
    rotors = boosted_grid(frame_rotation, boost_velocity, n_theta, n_theta)
    k, _, _, _ = conformal_factors(boost_velocity, rotors)
    vals = (k[0] * modes_evaluate(S, rotors, 0)).real
    A = map2salm_matrix(0, ell_max, n_theta, n_theta)
    return A @ vals.ravel()
