"""Rotation of the decomposition basis of WaveformModes objects (scri/rotations.py:268-343).
The per-time-step Wigner-D evaluation and mode mixing run in the HIP kernel
``rotate_modes_kernel`` (scri_amd/csrc/kernels_rotate.hip) through ``bms_rotate_const`` /
``bms_rotate_series``."""
import numpy as np

from . import engine, quaternions


def rotate_physical_system(W, R_phys):
    """Rotate a Waveform in place: rotate the decomposition basis by the inverse rotor(s)
    (scri/rotations.py:268-281)."""
    Rf = quaternions.as_float_array(R_phys)
    W = rotate_decomposition_basis(W, quaternions.like_input(quaternions.conjugate(Rf), R_phys))
    W._append_history(f"{W}.rotate_physical_system({R_phys})")
    return W


def rotate_decomposition_basis(W, R_basis):
    """Rotate a Waveform in place (scri/rotations.py:284-343).

    `R_basis`: a quaternion, or a list/array of 1 or n_times quaternions (np.quaternion objects or
    float components [..., 4]).  The change of basis is recorded in `W.frame` by right-multiplication.
    """
    is_q_obj = quaternions.is_quaternion_object(R_basis)
    if isinstance(R_basis, (list, tuple)) and len(R_basis) and not np.isscalar(R_basis[0]):
        R = quaternions.as_float_array(list(R_basis))
    else:
        R = quaternions.as_float_array(R_basis)
    # a length-1 iterable is a single rotor (rotations.py:301-302)
    if R.ndim == 2 and R.shape[0] == 1:
        R = R[0]
    if R.ndim > 2:
        raise ValueError("Input dimension mismatch.  R_basis.shape={}".format(R.shape[:-1]))
    on_device = getattr(W, "is_device_resident", False)  # weights in HBM (WaveformModes.to_device): rotated there, in place
    if not on_device and not W.data.flags.c_contiguous:
        W.data = np.ascontiguousarray(W.data)
    if R.ndim == 2:
        if W.n_times != R.shape[0]:
            raise ValueError(
                "Input dimension mismatch.  (W.n_times={}) != (len(R_basis)={})".format(W.n_times, R.shape[0])
            )
        if on_device:
            from . import device_series

            sp = device_series.to_device(W._ctx, quaternions.as_spinor_array(R))
            engine.rotate_device(W._dev.data_ptr(), W.n_times, W.n_modes, W.ell_min, W.ell_max, spinors_ptr=sp.data_ptr(), ctx=W._ctx)
            W._ctx.synchronize()  # (the rotor tensor goes out of scope)
        else:
            engine.rotate_series(W.data, W.ell_min, W.ell_max, quaternions.as_spinor_array(R), ctx=W._ctx)
        # right-multiplication (rotations.py:313-321)
        if W.frame.size:
            W.frame = quaternions.multiply(W.frame, R)  # broadcasts a single frame element
        else:
            W.frame = np.copy(R)
    else:
        if on_device:
            engine.rotate_device(W._dev.data_ptr(), W.n_times, W.n_modes, W.ell_min, W.ell_max, quaternion=R, ctx=W._ctx)
        else:
            engine.rotate_const(W.data, W.ell_min, W.ell_max, R, ctx=W._ctx)
        if W.frame.size:
            W.frame = quaternions.multiply(W.frame, R)
        else:
            W.frame = np.array([R])
    opts = np.get_printoptions()
    np.set_printoptions(threshold=6)
    W._append_history(f"{W}.rotate_decomposition_basis({R_basis if is_q_obj else R})")
    np.set_printoptions(**opts)
    return W


def to_coprecessing_frame(W, RoughDirection=np.array([0.0, 0.0, 1.0]), RoughDirectionIndex=None, transition_times=None):
    """Transform a waveform (in place) to a coprecessing frame (scri/rotations.py:14-49): the dominant eigenvector of <LL>
    (GPU) becomes the z axis, and the remaining freedom about it is fixed by the minimal-rotation condition."""
    from . import Coprecessing
    from .mode_calculations import LLDominantEigenvector

    if RoughDirectionIndex is None:
        RoughDirectionIndex = W.n_times // 8
    dpa = LLDominantEigenvector(W, RoughDirection=RoughDirection, RoughDirectionIndex=RoughDirectionIndex)
    v = dpa / np.linalg.norm(dpa, axis=-1)[:, None]
    # sqrt(-v z): the rotor taking z to v
    minus_vz = np.concatenate([v[:, 2:3], -np.cross(v, np.array([0.0, 0.0, 1.0]))], axis=-1)
    R = quaternions.minimal_rotation(quaternions.sqrt(minus_vz), W.t, iterations=3)
    if transition_times is not None:
        from .utilities import transition_function

        i0, i1 = np.argmin(np.abs(W.t - transition_times[0])), np.argmin(np.abs(W.t - transition_times[1]))
        transition = transition_function(W.t[i0:], W.t[i0], W.t[i1], y0=1.0, y1=0.0)
        omega = quaternions.angular_velocity(R[i0:], W.t[i0:]) * transition[:, np.newaxis]
        slowing = engine.integrate_angular_velocity(W.t[i0:], omega, R[i0])
        R = np.concatenate((R[:i0], slowing))
    rotate_decomposition_basis(W, R)
    W._append_history(f"{W}.to_coprecessing_frame({RoughDirection}, {RoughDirectionIndex}, {transition_times})")
    W.frameType = Coprecessing
    return W


def to_inertial_frame(W):
    """Undo the rotations recorded in W.frame (scri/rotations.py:106-111)."""
    from . import Inertial

    if W.frame.size:
        W = rotate_decomposition_basis(W, quaternions.conjugate(W.frame))
    W.frameType = Inertial
    W._append_history(f"{W}.to_inertial_frame()")
    return W


def to_corotating_frame(W, R0=(1.0, 0.0, 0.0, 0.0), tolerance=1e-12, z_alignment_region=None, return_omega=False,
                        truncate_log_frame=False):
    """Transform the waveform (in place) to a corotating frame (scri/rotations.py:52-103)."""
    from . import Corotating
    from .mode_calculations import corotating_frame

    frame, omega = corotating_frame(W, R0=R0, tolerance=tolerance, z_alignment_region=z_alignment_region, return_omega=True)
    log_frame = None
    if truncate_log_frame:
        # keep only the bits of log(frame) above the tolerance: exp(truncated(log(frame))) rotates the waveform (:86-90)
        log_frame = quaternions.log(frame)
        power_of_2 = 2 ** int(-np.floor(np.log2(2 * tolerance)))
        log_frame = np.round(log_frame * power_of_2) / power_of_2
        frame = quaternions.exp(log_frame)
    W.rotate_decomposition_basis(frame)
    W._append_history(f"{W}.to_corotating_frame({R0}, {tolerance}, {z_alignment_region}, {return_omega}, {truncate_log_frame})")
    W.frameType = Corotating
    out = (W,) + ((omega,) if return_omega else ()) + ((log_frame,) if truncate_log_frame else ())
    return out if len(out) > 1 else W


def _qvec(v):
    return np.array([0.0, v[0], v[1], v[2]])


def _qinv(q):
    return quaternions.conjugate(q) / np.sum(np.asarray(q, dtype=float) ** 2)


def get_alignment_of_decomposition_frame_to_modes(w, t_fid, nHat_t_fid=(0.0, 1.0, 0.0, 0.0), ell_max=None):
    """The constant rotor R_eps that fixes the attitude of a corotating (coprecessing, coorbital) frame at the fiducial time
    (scri/rotations.py:114-225): the frame's z axis onto the dominant eigenvector of <LL> (on the side of the angular velocity),
    the phase of the (2, +-2) modes to zero, x nearer to `nHat_t_fid` than to its opposite.  Found, not applied."""
    from . import Coprecessing, Coorbital, Corotating
    from .mode_calculations import angular_velocity, LLDominantEigenvector

    if ell_max is None:
        ell_max = w.ell_max
    if w.frameType not in [Coprecessing, Coorbital, Corotating]:
        raise ValueError(
            "get_alignment_of_decomposition_frame_to_modes only takes Waveforms in the coprecessing, coorbital, or corotating frames.  "
            "This Waveform is in the '{0}' frame.".format(w.frame_type_string))
    if w.frame.shape[0] != w.n_times:
        raise ValueError(
            "get_alignment_of_decomposition_frame_to_modes requires full information about the Waveform's frame."
            "This Waveform has {0} time steps, but only {1} rotors in its frame.".format(w.n_times, w.frame.shape[0]))
    if t_fid < w.t[0] or t_fid > w.t[-1]:
        raise ValueError("The requested alignment time t_fid={0} is outside the range of times in this waveform ({1}, {2}).".format(t_fid, w.t[0], w.t[-1]))
    nHat = np.asarray(quaternions.as_float_array(nHat_t_fid), dtype=float).reshape(4)
    # direction of the angular velocity near t_fid: an 11-sample window of the l = 2 modes, back in the inertial frame
    i_fid = int(np.nonzero(w.t <= t_fid)[0][-1])
    if i_fid < w.t.size - 1:
        i_fid += 1
    i1 = max(i_fid - 5, 0)
    i2 = w.t.size if i1 + 11 > w.t.size else i1 + 11
    region = w[i1:i2, 2].copy().to_inertial_frame()
    omega = _qvec(angular_velocity(region)[i_fid - i1])
    omega /= np.linalg.norm(omega)
    # ... as seen from this waveform's frame
    R = w.frame[i_fid] if w.frame.shape[0] > 1 else w.frame[0]
    omega = quaternions.multiply(quaternions.multiply(_qinv(R), omega), R)
    instant = w[i1:i2].copy().interpolate(np.array([t_fid]))
    R_f0 = instant.frame[0]
    V_f = _qvec(LLDominantEigenvector(instant[:, : ell_max + 1])[0])
    V_f /= np.linalg.norm(V_f)
    if np.dot(omega[1:], V_f[1:]) < 0:
        V_f = -V_f
    z = np.array([0.0, 0.0, 0.0, 1.0])
    R_V_f = quaternions.sqrt(quaternions.multiply(-V_f, z))  # takes z onto V_f
    instant.rotate_decomposition_basis(R_V_f)
    d22, d2m2 = instant.data[0, instant.index(2, 2)], instant.data[0, instant.index(2, -2)]
    phase = np.arctan2(d22.imag, d22.real) - np.arctan2(d2m2.imag, d2m2.real)
    R_eps = quaternions.multiply(R_V_f, quaternions.exp(np.array([0.0, 0.0, 0.0, -phase / 8.0])))
    total = quaternions.multiply(R_f0, R_eps)
    x_axis = quaternions.multiply(quaternions.multiply(total, np.array([0.0, 1.0, 0.0, 0.0])), _qinv(total))
    if np.dot(nHat[1:], x_axis[1:]) < 0:
        R_eps = quaternions.multiply(R_eps, quaternions.exp(np.array([0.0, 0.0, 0.0, np.pi / 2.0])))
    return R_eps


def align_decomposition_frame_to_modes(w, t_fid, nHat_t_fid=(0.0, 1.0, 0.0, 0.0), ell_max=None):
    """Fix the attitude of the corotating frame (scri/rotations.py:228-265): a corotating frame is defined up to a constant rotor
    on the right; this one puts z along the dominant eigenvector of <LL> at t_fid and the (2, 2) phase to zero."""
    R_eps = get_alignment_of_decomposition_frame_to_modes(w, t_fid, nHat_t_fid, ell_max)
    w._append_history("{0}.align_decomposition_frame_to_modes({1}, {2}, {3})  # R_eps={4}".format(w, t_fid, nHat_t_fid, ell_max, R_eps))
    return w.rotate_decomposition_basis(R_eps)
