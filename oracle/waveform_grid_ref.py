"""Oracle for the WaveformModes BMS transform: scri/waveform_grid.py:20-190 (kwargs -> rotor
grid), :331-613 (from_modes), :274-329 (to_modes), :615-630 (transform),
scri/waveform_modes.py:705-719 (delegate).  Statement-by-statement restatement in numpy with
the *live* scipy spline the reference itself calls.  Test infrastructure only.
"""
import math
import warnings
import numpy as np
from scipy import interpolate
from scipy.special import comb

from . import quat
from .containers import WM, DataNames, h, sigma, psi0, psi1, psi2, psi3, psi4, hdot, news, Inertial
from .wigner import (
    LM_index,
    LM_total_size,
    swsh_grid,
    constant_as_ell_0_mode,
    constant_from_ell_0_mode,
    vector_as_ell_1_modes,
    vector_from_ell_1_modes,
    eth_GHP,
    ethbar_GHP,
)
from .spinsfast_ref import map2salm_matrix


def boost_rotor_factory(boost_velocity):
    """Bprm_j_k of scri/waveform_grid.py:138-166 (== transformations.py:106-135)."""
    boost_velocity = np.asarray(boost_velocity, dtype=float)
    beta = np.linalg.norm(boost_velocity)
    varphi = math.atanh(beta)
    if beta > 3e-14:
        vhat = boost_velocity / beta

        def Bprm(thetaprm, phiprm):
            rprm = np.array(
                [math.cos(phiprm) * math.sin(thetaprm), math.sin(phiprm) * math.sin(thetaprm), math.cos(thetaprm)]
            )
            Thetaprm = math.acos(np.dot(vhat, rprm))
            Theta = 2 * math.atan(math.exp(-varphi) * math.tan(Thetaprm / 2.0))
            c = np.cross(rprm, vhat)
            cn = math.sqrt(np.dot(c, c))
            if cn > 1e-200:
                return quat.qexp(np.array([0.0, *(c / cn)]) * (Thetaprm - Theta) / 2)
            return np.array([1.0, 0.0, 0.0, 0.0])

    else:

        def Bprm(thetaprm, phiprm):
            return np.array([1.0, 0.0, 0.0, 0.0])

    return Bprm


def rotor_grid(frame_rotation, boost_velocity, n_theta, n_phi):
    """R_j_k of scri/waveform_grid.py:130-174 (== boosted_grid, transformations.py:100-148)."""
    Bprm = boost_rotor_factory(boost_velocity)
    thetas = np.linspace(0.0, np.pi, num=n_theta, endpoint=True)
    phis = np.linspace(0.0, 2 * np.pi, num=n_phi, endpoint=False)
    R = np.empty((n_theta, n_phi, 4))
    for j in range(n_theta):
        for k in range(n_phi):
            rq = quat.qmul(frame_rotation, quat.from_spherical_coords(thetas[j], phis[k]))
            th, ph = quat.as_spherical_coords(rq)
            R[j, k] = quat.qmul(Bprm(float(th), float(ph)), rq)
    return R


def process_transformation_kwargs(ell_max, **kwargs):
    """scri/waveform_grid.py:20-190."""
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
                a = supertranslation[LM_index(ell, m, 0)]
                b = supertranslation[LM_index(ell, -m, 0)]
                if abs(a - (-1.0) ** m * b.conjugate()) > 3e-16 + 1e-15 * abs(b):
                    raise ValueError("Will result in an imaginary supertranslation.")
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

    w_ell_max = ell_max
    ell_max = w_ell_max + ell_max_supertranslation
    n_theta = kwargs.pop("n_theta", 2 * ell_max + 1)
    n_phi = kwargs.pop("n_phi", 2 * ell_max + 1)
    if n_theta < 2 * ell_max + 1 and abs(supertranslation[1:]).max() > 0.0:
        warnings.warn(f"n_theta={n_theta} is small")
    if n_theta < 2 * w_ell_max + 1:
        raise ValueError(f"n_theta={n_theta} is too small")
    if n_phi < 2 * ell_max + 1 and abs(supertranslation[1:]).max() > 0.0:
        warnings.warn(f"n_phi={n_phi} is small")
    if n_phi < 2 * w_ell_max + 1:
        raise ValueError(f"n_phi={n_phi} is too small")

    frame_rotation = np.array(kwargs.pop("frame_rotation", [1, 0, 0, 0]), dtype=float)
    if quat.qabs(frame_rotation) < 3e-16:
        raise ValueError(f"frame_rotation={frame_rotation} should be a unit quaternion")
    frame_rotation = quat.qnormalized(frame_rotation)

    boost_velocity = np.array(kwargs.pop("boost_velocity", [0.0] * 3), dtype=float)
    beta = np.linalg.norm(boost_velocity)
    if boost_velocity.shape != (3,) or beta >= 1.0:
        raise ValueError("Input boost_velocity should be a 3-vector with magnitude strictly less than 1.0.")
    gamma = 1 / math.sqrt(1 - beta**2)
    varphi = math.atanh(beta)

    R_j_k = rotor_grid(frame_rotation, boost_velocity, n_theta, n_phi)
    return (supertranslation, ell_max_supertranslation, ell_max, n_theta, n_phi, boost_velocity, beta, gamma, varphi, R_j_k, kwargs)


def from_modes(w_modes, **kwargs):
    """scri/waveform_grid.py:331-613.  Returns (uprm_iprm, fprm_iprm_j_k[N', n_theta, n_phi], n_theta, n_phi)."""
    if w_modes.frameType != Inertial:
        raise ValueError("Input waveform object must be in an inertial frame")
    (supertranslation, ell_max_supertranslation, ell_max, n_theta, n_phi, boost_velocity, beta, gamma, varphi, R_j_k, kwargs) = (
        process_transformation_kwargs(w_modes.ell_max, **kwargs)
    )
    s = w_modes.spin_weight
    SWSH_j_k = swsh_grid(R_j_k, s, ell_max)
    SH_j_k = swsh_grid(R_j_k, 0, ell_max_supertranslation)
    r_j_k = quat.rotate_z(R_j_k.reshape(-1, 4)).T  # [3, n_pix]
    kconformal_j_k = 1.0 / (gamma * (1 - np.dot(boost_velocity, r_j_k).reshape(R_j_k.shape[:2])))
    alphasupertranslation_j_k = np.tensordot(supertranslation, SH_j_k, axes=([0], [2])).real
    fprm_i_j_k = np.tensordot(
        w_modes.data,
        SWSH_j_k[:, :, LM_index(w_modes.ell_min, -w_modes.ell_min, 0) : LM_index(w_modes.ell_max, w_modes.ell_max, 0) + 1],
        axes=([1], [2]),
    )
    if beta != 0 or (supertranslation[1:] != 0).any():
        if w_modes.dataType == h:
            supertranslation_deriv = 2 * ethbar_GHP(ethbar_GHP(supertranslation, 0, 0), -1, 0)
            vals = np.tensordot(
                supertranslation_deriv,
                SWSH_j_k[:, :, : LM_index(ell_max_supertranslation, ell_max_supertranslation, 0) + 1],
                axes=([0], [2]),
            )
            fprm_i_j_k -= vals[np.newaxis, :, :]
        elif w_modes.dataType == sigma:
            supertranslation_deriv = eth_GHP(eth_GHP(supertranslation, 0, 0), 1, 0)
            vals = np.tensordot(
                supertranslation_deriv,
                SWSH_j_k[:, :, : LM_index(ell_max_supertranslation, ell_max_supertranslation, 0) + 1],
                axes=([0], [2]),
            )
            fprm_i_j_k -= vals[np.newaxis, :, :]
        elif w_modes.dataType in [psi0, psi1, psi2, psi3]:
            eth_alpha_j_k = np.tensordot(
                1 / np.sqrt(2) * eth_GHP(supertranslation, 0),
                swsh_grid(R_j_k, 1, ell_max_supertranslation),
                axes=([0], [2]),
            )
            v_dot_rhat = np.insert(vector_as_ell_1_modes(boost_velocity), 0, 0.0)
            eth_v_dot_rhat_j_k = np.tensordot(1 / np.sqrt(2) * v_dot_rhat, swsh_grid(R_j_k, 1, 1), axes=([0], [2]))
            eth_uprm_over_k_i_j_k = (
                w_modes.t[:, np.newaxis, np.newaxis] - alphasupertranslation_j_k[np.newaxis, :, :]
            ) * gamma * kconformal_j_k[np.newaxis, :, :] * eth_v_dot_rhat_j_k[np.newaxis, :, :] - eth_alpha_j_k[np.newaxis, :, :]
            for DT in range(w_modes.dataType + 1, psi4 + 1):
                try:
                    w_tmp = kwargs.pop("psi{}_modes".format(DataNames[DT][-1]))
                except KeyError:
                    raise ValueError(
                        "\nA BMS transformation of {} requires information from {}, which "
                        "has not been supplied.".format(DataNames[w_modes.dataType], DataNames[DT])
                    )
                SW_tmp = swsh_grid(R_j_k, w_tmp.spin_weight, w_tmp.ell_max)
                f_i_j_k = np.tensordot(
                    w_tmp.data,
                    SW_tmp[:, :, LM_index(w_tmp.ell_min, -w_tmp.ell_min, 0) : LM_index(w_tmp.ell_max, w_tmp.ell_max, 0) + 1],
                    axes=([1], [2]),
                )
                fprm_i_j_k += comb(5 - w_modes.dataType, 5 - DT) * f_i_j_k * eth_uprm_over_k_i_j_k ** (DT - w_modes.dataType)
        elif w_modes.dataType not in [psi4, hdot, news]:
            warnings.warn("No BMS transformation is implemented for this dataType; proceeding as Psi4.")

    fprm_i_j_k *= (kconformal_j_k**w_modes.conformal_weight)[np.newaxis, :, :]

    time_translation = constant_from_ell_0_mode(supertranslation[0]).real
    uprm_i = (1 / gamma) * (w_modes.t - time_translation)
    uprm_min = (kconformal_j_k * (w_modes.t[0] - alphasupertranslation_j_k)).max()
    uprm_max = (kconformal_j_k * (w_modes.t[-1] - alphasupertranslation_j_k)).min()
    uprm_iprm = uprm_i[(uprm_i >= uprm_min) & (uprm_i <= uprm_max)]

    for j in range(n_theta):
        for k in range(n_phi):
            uprm_i_j_k = kconformal_j_k[j, k] * (w_modes.t - alphasupertranslation_j_k[j, k])
            re = interpolate.InterpolatedUnivariateSpline(uprm_i_j_k, fprm_i_j_k[:, j, k].real)
            im = interpolate.InterpolatedUnivariateSpline(uprm_i_j_k, fprm_i_j_k[:, j, k].imag)
            fprm_i_j_k[: len(uprm_iprm), j, k] = re(uprm_iprm) + 1j * im(uprm_iprm)
    fprm_iprm_j_k = fprm_i_j_k[: len(uprm_iprm)]
    if kwargs:
        warnings.warn("Unused kwargs passed to this function: {}".format(sorted(kwargs)))
    return uprm_iprm, fprm_iprm_j_k, n_theta, n_phi


def to_modes(t, grid, s, ell_max, ell_min=None):
    """scri/waveform_grid.py:274-329: per-time spinsfast.map2salm, drop l < ell_min."""
    if ell_min is None:
        ell_min = abs(s)
    n_times, n_theta, n_phi = grid.shape
    A = map2salm_matrix(s, ell_max, n_theta, n_phi)
    modes = grid.reshape(n_times, -1) @ A.T
    return modes[:, LM_index(ell_min, -ell_min, 0) :]


def transform(w_modes, **kwargs):
    """scri/waveform_grid.py:615-630 / scri/waveform_modes.py:705-719."""
    ell_max = kwargs.pop("ell_max", w_modes.ell_max)
    if np.ndim(w_modes.data) > 2:
        # Extra trailing data dimensions: `final_dim` = their product; the spline loop of from_modes (:574-588) and the map2salm
        # loop of to_modes (:299-308) walk `final_indices` one at a time, i.e. every trailing index is a series of its own under
        # the same transformation.  (Literally, :475-484 contracts with np.tensordot, which leaves the extra axes BEFORE the grid
        # axes, so that :581 indexes the wrong axis and raises IndexError -- the reference cannot run this case; what is restated
        # here is what its two loops spell out.)
        trailing = w_modes.data.shape[2:]
        flat = w_modes.data.reshape(w_modes.data.shape[:2] + (-1,))
        aux_keys = [k for k in kwargs if k.startswith("psi") and k.endswith("_modes")]
        outs = []
        for f in range(flat.shape[2]):
            kw_f = dict(kwargs, ell_max=ell_max)
            for k in aux_keys:
                a = kwargs[k]
                kw_f[k] = WM(t=a.t, data=np.ascontiguousarray(a.data.reshape(a.data.shape[:2] + (-1,))[:, :, f]), ell_min=a.ell_min, ell_max=a.ell_max,
                             dataType=a.dataType, frameType=a.frameType, r_is_scaled_out=a.r_is_scaled_out, m_is_scaled_out=a.m_is_scaled_out)
            w_f = WM(t=w_modes.t, data=np.ascontiguousarray(flat[:, :, f]), ell_min=w_modes.ell_min, ell_max=w_modes.ell_max, dataType=w_modes.dataType,
                     frameType=w_modes.frameType, r_is_scaled_out=w_modes.r_is_scaled_out, m_is_scaled_out=w_modes.m_is_scaled_out)
            outs.append(transform(w_f, **kw_f))
        data = np.stack([o.data for o in outs], axis=2).reshape(outs[0].data.shape + trailing)
        return WM(t=outs[0].t, data=data, ell_min=outs[0].ell_min, ell_max=outs[0].ell_max, dataType=w_modes.dataType, frameType=w_modes.frameType,
                  r_is_scaled_out=w_modes.r_is_scaled_out, m_is_scaled_out=w_modes.m_is_scaled_out)
    uprm, grid, n_theta, n_phi = from_modes(w_modes, **kwargs)
    s = w_modes.spin_weight
    data = to_modes(uprm, grid, s, ell_max)
    return WM(
        t=uprm,
        data=data,
        ell_min=abs(s),
        ell_max=ell_max,
        dataType=w_modes.dataType,
        frameType=w_modes.frameType,
        r_is_scaled_out=w_modes.r_is_scaled_out,
        m_is_scaled_out=w_modes.m_is_scaled_out,
    )
