"""Frame-construction quantities that feed the rotation path (scri/mode_calculations.py:46-57 LdtVector,
:298-313 LLMatrix, :403-432 angular_velocity), SURVEY 8(f) rank 3.  One GPU pass (bms_angular_velocity): the cubic-spline
time derivative of the modes, the <Ldt> and <LL> sums over modes per time step, and the 3 x 3 solve.

corotating_frame (:435-491) integrates that angular velocity with the library's host integrator
(bms_integrate_angular_velocity: the frame is four numbers marching in time).

Not provided: the frame-velocity term (needs numpy-quaternion's `derivative`), z_alignment_region (needs
LLDominantEigenvector).
"""
import numpy as np

from . import engine, quaternions


def _parts(W):
    return engine.angular_velocity(W.t, W.data, W.ell_min, W.ell_max, ctx=getattr(W, "_ctx", None), parts=True)


def LdtVector(W):
    r"""<Ldt>^a = \sum \bar f^{l,m'} <l,m'|L_a|l,m> (df/dt)^{l,m}, in the mode frame; [n_times, 3]"""
    return _parts(W)[0]


def LLMatrix(W):
    r"""<LL>^{ab} = Re \sum \bar f^{l,m'} <l,m'|L_a L_b|l,m> f^{l,m}; [n_times, 3, 3]"""
    return _parts(W)[1]


def angular_velocity(W, include_frame_velocity=False):
    """Angular velocity of the waveform (arXiv:1302.2919 Sec. II): solve <Ldt> = -<LL> . omega at every time step."""
    if include_frame_velocity and len(W.frame) == W.n_times:
        raise NotImplementedError("include_frame_velocity needs quaternion.derivative, which is outside this build")
    return _parts(W)[2]


def corotating_frame(W, R0=(1.0, 0.0, 0.0, 0.0), tolerance=1e-12, z_alignment_region=None, return_omega=False):
    """Rotor taking the current mode frame into the corotating frame: the integral of the waveform's angular velocity
    starting from R0 (scri/mode_calculations.py:435-491)."""
    if z_alignment_region is not None:
        raise NotImplementedError("z_alignment_region needs LLDominantEigenvector, which is outside this build")
    omega = angular_velocity(W)
    R0 = np.asarray(getattr(R0, "components", R0), dtype=float)
    frame = engine.integrate_angular_velocity(W.t, omega, R0=R0, tolerance=tolerance)
    frame = frame / np.linalg.norm(frame, axis=1)[:, np.newaxis]
    return (frame, omega) if return_omega else frame
