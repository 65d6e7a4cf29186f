"""Frame-construction quantities that feed the rotation path (scri/mode_calculations.py:46-57 LdtVector,
:298-313 LLMatrix, :403-432 angular_velocity), SURVEY 8(f) rank 3.  One GPU pass (bms_angular_velocity): the cubic-spline
time derivative of the modes, the <Ldt> and <LL> sums over modes per time step, and the 3 x 3 solve.

corotating_frame (:435-491) integrates that angular velocity with the library's host integrator
(bms_integrate_angular_velocity: the frame is four numbers marching in time).

LLDominantEigenvector (:316-399) takes the <LL> matrices from the GPU and finishes on the host (3 x 3 symmetric
eigenproblems and the sequential sign choice that makes the axis continuous).

Not provided: the frame-velocity term (needs numpy-quaternion's `derivative`).
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
    n = W.n_times
    if 0 < n < 4:
        # the kernel also forms <Ldt>, whose spline needs four samples; <LL> needs none: pad the series with copies of its last row
        data = np.concatenate([W.data, np.repeat(W.data[-1:], 4 - n, axis=0)], axis=0)
        return engine.angular_velocity(np.arange(4.0), data, W.ell_min, W.ell_max, ctx=getattr(W, "_ctx", None), parts=True)[1][:n]
    return _parts(W)[1]


def angular_velocity(W, include_frame_velocity=False):
    """Angular velocity of the waveform (arXiv:1302.2919 Sec. II): solve <Ldt> = -<LL> . omega at every time step."""
    omega = _parts(W)[2]
    if include_frame_velocity and len(W.frame) == W.n_times:
        # + 2 Rdot R^-1 of the frame, its derivative from a cubic spline (scri/mode_calculations.py:426-430)
        from . import quaternions

        omega = omega + quaternions.angular_velocity(W.frame, W.t)
    return omega


def _make_continuous(dpa, rough, i_index):
    """Sign choice of scri/mode_calculations.py:316-363: the axis at i_index points along `rough` rather than against
    it, and going outwards from there a vector is flipped when it is further from its (already fixed) neighbour than its
    own length; every vector is normalised."""
    dpa = np.array(dpa, dtype=float)
    if np.dot(rough, dpa[i_index]) < 0.0:
        dpa[i_index] *= -1
    for rng, d in ((range(i_index - 1, -1, -1), -1), (range(i_index + 1, dpa.shape[0]), 1)):
        for i in rng:
            diff = dpa[i] - dpa[i - d]
            if diff @ diff > dpa[i] @ dpa[i]:
                dpa[i] *= -1
    norms = np.linalg.norm(dpa, axis=1)
    ok = norms != 0.0
    dpa[ok] /= norms[ok, np.newaxis]
    return dpa


def LLDominantEigenvector(W, RoughDirection=np.array([0.0, 0.0, 1.0]), RoughDirectionIndex=0):
    """Principal axis of the <LL> matrix at every time step, made continuous in time (mode frame)."""
    _, eigenvecs = np.linalg.eigh(LLMatrix(W))
    return _make_continuous(eigenvecs[:, :, 2], np.asarray(RoughDirection, dtype=float), RoughDirectionIndex)  # largest eigenvalue last


def _qsqrt(q):
    """square root of a unit quaternion"""
    p = np.array(q, dtype=float)
    p[0] += 1.0
    return p / np.linalg.norm(p)


def corotating_frame(W, R0=(1.0, 0.0, 0.0, 0.0), tolerance=1e-12, z_alignment_region=None, return_omega=False):
    """Rotor taking the current mode frame into the corotating frame: the integral of the waveform's angular velocity
    starting from R0 (scri/mode_calculations.py:435-491).  z_alignment_region = (f1, f2): additionally align the dominant
    eigenvector of <LL>, averaged over that fraction of the inspiral, with the z axis."""
    omega = angular_velocity(W)
    R0 = np.asarray(getattr(R0, "components", R0), dtype=float)
    frame = engine.integrate_angular_velocity(W.t, omega, R0=R0, tolerance=tolerance)
    if z_alignment_region is not None:
        initial_time = W.t[0]
        n4 = W.n_times // 4  # WaveformBase.max_norm_time: skips the first quarter (scri/waveform_base.py:553-575)
        inspiral_time = W.t[n4 + int(np.argmax((np.abs(W.data[n4:]) ** 2).sum(axis=1)))] - initial_time
        t1 = initial_time + z_alignment_region[0] * inspiral_time
        t2 = initial_time + z_alignment_region[1] * inspiral_time
        i1 = int(np.argmin(np.abs(W.t - t1)))
        i2 = int(np.argmin(np.abs(W.t - t2)))
        R = frame[i1:i2]
        i1m = max(0, i1 - 10)
        rough = omega[i1m + 10]
        _, vecs = np.linalg.eigh(LLMatrix(W)[i1:i2])
        Vhat = _make_continuous(vecs[:, :, 2], rough, 0)
        V = np.concatenate([np.zeros((Vhat.shape[0], 1)), Vhat], axis=1)
        Vhat_corot = quaternions.multiply(quaternions.multiply(quaternions.conjugate(R), V), R)[:, 1:]
        mean = np.concatenate([[0.0], np.mean(Vhat_corot, axis=0)])
        mean /= np.linalg.norm(mean)
        correction = quaternions.conjugate(_qsqrt(quaternions.multiply(np.array([0.0, 0.0, 0.0, -1.0]), mean)))  # sqrt(-z V)^-1
        frame = quaternions.multiply(frame, correction)
    frame = frame / np.linalg.norm(frame, axis=1)[:, np.newaxis]
    return (frame, omega) if return_omega else frame


# ---- two-waveform versions (scri/mode_calculations.py:55-260): <f| L_a |g> and <f| L_a L_b |g>
def _ladder_applied(W, which):
    """L_+ g, L_- g or L_z g along the mode axis (one bms_mode_map launch): (L_+ g)_{l,m} = sqrt((l - m + 1)(l + m)) g_{l,m-1}, ..."""
    LM = np.asarray(W.LM)
    ell, m = LM[:, 0], LM[:, 1]
    own = np.arange(ell.size, dtype=np.int32)
    if which == "z":
        src, coef = own, m.astype(float)
    elif which == "+":
        src = np.where(m - 1 >= -ell, own - 1, -1).astype(np.int32)
        coef = np.sqrt(np.maximum((ell - (m - 1)) * (ell + (m - 1) + 1.0), 0.0))
    else:
        src = np.where(m + 1 <= ell, own + 1, -1).astype(np.int32)
        coef = np.sqrt(np.maximum((ell + (m + 1)) * (ell - (m + 1) + 1.0), 0.0))
    out = W.copy()
    out.data = engine.mode_map(W.data, src, coef.astype(complex), ctx=getattr(W, "_ctx", None))
    return out


def _braket(W1, W2):
    return np.einsum("ij, ij -> i", np.conjugate(W1.data), W2.data)


def LVector(W1, W2):
    r"""<L>^a = \sum \bar f^{l,m'} <l,m'|L_a|l,m> g^{l,m} for two waveforms with the same modes; complex [n_times, 3], mode frame"""
    Lp, Lm, Lz = (_braket(W1, _ladder_applied(W2, w)) for w in "+-z")
    return np.stack([0.5 * (Lp + Lm), -0.5j * (Lp - Lm), Lz], axis=1)


def LLComparisonMatrix(W1, W2):
    r"""<LL>^{ab} = \sum \bar f^{l,m'} <l,m'|L_a L_b|l,m> g^{l,m}; complex [n_times, 3, 3].

    As in the reference (scri/mode_calculations.py:187), the <L_y L_z> term is added to the (y, y) element and the (y, z) element
    stays zero -- a slip of an index there, kept so that the numbers are the reference's."""
    T = {}
    for b in "+-z":
        right = _ladder_applied(W2, b)
        for a in "+-z":
            T[a + b] = _braket(W1, _ladder_applied(right, a))
    LL = np.empty((W1.n_times, 3, 3), dtype=complex)
    LL[:, 0, 0] = (T["++"] + T["+-"] + T["-+"] + T["--"]) / 4
    LL[:, 0, 1] = -1j * (T["++"] - T["+-"] + T["-+"] - T["--"]) / 4
    LL[:, 0, 2] = (T["+z"] + T["-z"]) / 2
    LL[:, 1, 0] = -1j * (T["++"] + T["+-"] - T["-+"] - T["--"]) / 4
    LL[:, 1, 1] = -(T["++"] - T["+-"] - T["-+"] + T["--"]) / 4 - 1j * (T["+z"] - T["-z"]) / 2
    LL[:, 1, 2] = 0.0
    LL[:, 2, 0] = (T["z+"] + T["z-"]) / 2
    LL[:, 2, 1] = -1j * (T["z+"] - T["z-"]) / 2
    LL[:, 2, 2] = T["zz"]
    return LL


def inner_product(t, abar, b, axis=None, apply_conjugate=False, ctx=None):
    """Time-domain complex inner product <a, b> of two arrays of samples (scri/mode_calculations.py:493-533): the definite integral
    over `t` of abar * b (of conj(abar) * b with apply_conjugate), through the not-a-knot cubic spline of the integrand -- the GPU
    spline antiderivative (bms_spline_derivative, order -1).  The time axis is `axis` (default: the first axis whose length is
    len(t), as quaternion.calculus.spline_definite_integral infers it)."""
    t = np.asarray(t, dtype=float)
    integrand = (np.conjugate(abar) if apply_conjugate else np.asarray(abar)) * np.asarray(b)
    if axis is None:
        matches = [i for i, n in enumerate(integrand.shape) if n == t.shape[0]]
        if not matches:
            raise ValueError(f"no axis of the integrand (shape {integrand.shape}) has the length of t ({t.shape[0]})")
        axis = matches[0]
    moved = np.moveaxis(np.asarray(integrand, dtype=complex), axis, 0)
    flat = np.ascontiguousarray(moved.reshape(moved.shape[0], -1))
    total = engine.spline_derivative(t, flat, t[-1:], order=-1, ctx=ctx)[0]
    out = total.reshape(moved.shape[1:])
    return out if np.iscomplexobj(integrand) else out.real
