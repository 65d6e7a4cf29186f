"""Frame-construction quantities that feed the rotation path (scri/mode_calculations.py:46-57 LdtVector,
:298-313 LLMatrix, :403-432 angular_velocity), SURVEY 8(f) rank 3.  One GPU pass (bms_angular_velocity): the cubic-spline
time derivative of the modes, the <Ldt> and <LL> sums over modes per time step, and the 3 x 3 solve.

Not provided: the frame-velocity term (needs numpy-quaternion's `derivative`), corotating_frame (numpy-quaternion's
adaptive integrate_angular_velocity) and the dominant-eigenvector routines.
"""
from . import engine


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
