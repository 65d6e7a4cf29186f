"""Minimal quaternion handling for the shim.  Rotors are float arrays [..., 4] = (w, x, y, z);
`numpy-quaternion` objects/arrays are accepted wherever the reference accepts them (converted with
``quaternion.as_float_array`` when that package is importable), so callers of scri are unchanged."""
import numpy as np

try:  # optional: only to recognise np.quaternion inputs
    import quaternion as _npq
except Exception:  # pragma: no cover
    _npq = None


def is_quaternion_object(x):
    return _npq is not None and (isinstance(x, _npq.quaternion) or (isinstance(x, np.ndarray) and x.dtype == _npq.quaternion))


def as_float_array(q):
    """-> float array [..., 4]."""
    if _npq is not None and (isinstance(q, _npq.quaternion) or (isinstance(q, np.ndarray) and q.dtype == _npq.quaternion)):
        return _npq.as_float_array(q)
    if isinstance(q, (list, tuple)) and len(q) and is_quaternion_object(q[0]):
        return np.array([_npq.as_float_array(x) for x in q])
    a = np.asarray(q, dtype=float)
    if a.shape[-1:] != (4,):
        raise ValueError(f"expected quaternion components [..., 4], got shape {a.shape}")
    return a


def like_input(q_float, template):
    """Return rotors in the flavour of `template` (np.quaternion array if the caller used those)."""
    if _npq is not None and template is not None and is_quaternion_object(template):
        return _npq.as_quat_array(np.ascontiguousarray(q_float))
    return q_float


def multiply(a, b):
    a = np.asarray(a, dtype=float)
    b = np.asarray(b, dtype=float)
    w1, x1, y1, z1 = np.moveaxis(a, -1, 0)
    w2, x2, y2, z2 = np.moveaxis(b, -1, 0)
    return np.stack(
        [
            w1 * w2 - x1 * x2 - y1 * y2 - z1 * z2,
            w1 * x2 + x1 * w2 + y1 * z2 - z1 * y2,
            w1 * y2 - x1 * z2 + y1 * w2 + z1 * x2,
            w1 * z2 + x1 * y2 - y1 * x2 + z1 * w2,
        ],
        axis=-1,
    )


def conjugate(a):
    return np.asarray(a, dtype=float) * np.array([1.0, -1.0, -1.0, -1.0])


def from_spherical_coords(theta, phi):
    """The rotor R(theta, phi) = exp(phi z/2) exp(theta y/2) taking z to the direction (theta, phi) and x, y to the tangent vectors there
    (quaternion.from_spherical_coords; scri/waveform_grid.py:138-140 builds its grid rotors from it): float array [..., 4]."""
    theta, phi = np.broadcast_arrays(np.asarray(theta, dtype=float), np.asarray(phi, dtype=float))
    ct, st, cp, sp = np.cos(theta / 2), np.sin(theta / 2), np.cos(phi / 2), np.sin(phi / 2)
    return np.stack([cp * ct, -sp * st, cp * st, sp * ct], axis=-1)


def as_spinor_array(q):
    """(w + i z, y + i x): quaternion.as_spinor_array (scri/rotations.py:311)."""
    q = np.asarray(q, dtype=float)
    return np.stack([q[..., 0] + 1j * q[..., 3], q[..., 2] + 1j * q[..., 1]], axis=-1)


def log(q):
    """Logarithm of unit(ish) quaternions [..., 4]: (ln |q|, v acos(w/|q|)/|v|)."""
    q = np.asarray(q, dtype=float)
    v = q[..., 1:]
    vn = np.linalg.norm(v, axis=-1)
    qn = np.linalg.norm(q, axis=-1)
    angle = np.arctan2(vn, q[..., 0])
    with np.errstate(invalid="ignore", divide="ignore"):
        scale = np.where(vn > 1e-300, angle / np.where(vn > 1e-300, vn, 1.0), 0.0)
    out = np.empty_like(q)
    out[..., 0] = np.log(np.where(qn > 0, qn, 1.0))
    out[..., 1:] = v * scale[..., None]
    return out


def exp(q):
    """Exponential of quaternions [..., 4]."""
    q = np.asarray(q, dtype=float)
    v = q[..., 1:]
    vn = np.linalg.norm(v, axis=-1)
    e = np.exp(q[..., 0])
    with np.errstate(invalid="ignore", divide="ignore"):
        s = np.where(vn > 1e-300, np.sin(vn) / np.where(vn > 1e-300, vn, 1.0), 1.0)
    out = np.empty_like(q)
    out[..., 0] = e * np.cos(vn)
    out[..., 1:] = (e * s)[..., None] * v
    return out


def slerp(q1, q2, tau):
    """q1 (q1^-1 q2)^tau for unit quaternions; tau broadcasts against the leading axes."""
    tau = np.asarray(tau, dtype=float)
    return multiply(q1, exp(log(multiply(conjugate(q1), q2)) * tau[..., None]))


def squad(R_in, t_in, t_out):
    """Spherical "quadrangular" interpolation of rotors (the C^1 analogue of a cubic spline on the rotation group), as
    `quaternion.squad` of the numpy-quaternion package, which scri/waveform_base.py:957 uses for the frame of an
    interpolated waveform: control points from the logarithms of neighbouring relative rotations, end values mirrored
    (R_{-1} = R_0 R_1^-1 R_0, R_n = R_{n-1} R_{n-2}^-1 R_{n-1}), and
        squad = slerp( slerp(R_i, R_{i+1}, tau), slerp(A_i, B_{i+1}, tau), 2 tau (1 - tau) ).
    R_in: float [n, 4] unit quaternions (n >= 2); t_in [n] increasing; t_out [m]."""
    R = np.asarray(R_in, dtype=float)
    t_in = np.asarray(t_in, dtype=float)
    t_out = np.asarray(t_out, dtype=float)
    if R.size == 0 or t_out.size == 0:
        return np.zeros((0, 4))
    if R.shape[0] < 2:
        return np.repeat(R[:1], t_out.size, axis=0)
    roll = lambda a, k: np.roll(a, k, axis=0)  # noqa: E731
    inv = conjugate
    i = np.clip(t_in.searchsorted(t_out, side="right") - 1, 0, t_in.size - 1)
    with np.errstate(invalid="ignore", divide="ignore"):
        r_next = ((roll(t_in, -1) - t_in) / (t_in - roll(t_in, 1)))[:, None]
        r_after = ((roll(t_in, -1) - t_in) / (roll(t_in, -2) - roll(t_in, -1)))[:, None]
    step = log(multiply(inv(R), roll(R, -1)))             # log(R_i^-1 R_{i+1})
    back = log(multiply(inv(roll(R, 1)), R))              # log(R_{i-1}^-1 R_i)
    ahead = log(multiply(inv(roll(R, -1)), roll(R, -2)))  # log(R_{i+1}^-1 R_{i+2})
    A = multiply(R, exp((-step + back * r_next) * 0.25))
    B = multiply(roll(R, -1), exp((ahead * r_after - step) * -0.25))
    last_next = multiply(multiply(R[-1], inv(R[-2])), R[-1])  # R_n, the mirrored value beyond the end
    A[0] = R[0]
    A[-1] = R[-1]
    B[-2] = R[-1]
    B[-1] = last_next
    R_ip1 = roll(R, -1).copy()
    R_ip1[-1] = last_next
    t_ip1 = roll(t_in, -1).copy()
    t_ip1[-1] = t_in[-1] + (t_in[-1] - t_in[-2])
    tau = (t_out - t_in[i]) / (t_ip1 - t_in)[i]
    return slerp(slerp(R[i], R_ip1[i], tau), slerp(A[i], B[i], tau), 2 * tau * (1 - tau))


def sqrt(q):
    """Square root of unit quaternions [..., 4] (the rotor of half the rotation); -1 maps to a rotation about x."""
    q = np.asarray(q, dtype=float)
    w = np.sqrt(np.maximum((1.0 + q[..., 0]) / 2.0, 0.0))
    out = np.empty_like(q)
    out[..., 0] = w
    with np.errstate(invalid="ignore", divide="ignore"):
        out[..., 1:] = np.where(w[..., None] > 1e-150, q[..., 1:] / (2.0 * np.where(w > 1e-150, w, 1.0))[..., None], np.array([1.0, 0.0, 0.0]))
    return out


def angular_velocity(R, t):
    """omega(t) = 2 Rdot R^-1 (vector part) of a rotor series, with the derivative of a cubic spline through the components
    (quaternion.angular_velocity)."""
    from scipy.interpolate import CubicSpline

    R = np.asarray(R, dtype=float)
    Rdot = CubicSpline(t, R).derivative()(t)
    return 2.0 * multiply(Rdot, conjugate(R))[..., 1:]


def minimal_rotation(R, t, iterations=2):
    """Adjust a frame R(t) by rotations about its own z axis so that the adjusted frame has no angular velocity along that
    axis (numpy-quaternion's `minimal_rotation`, which scri/rotations.py:38 applies to the coprecessing frame):
    R' = R exp(gamma z / 2) with  gamma-dot / 2 = Re[ Rdot z R^-1 ],  the time derivative and integral taken with cubic
    splines, iterated because the spline of R' differs from the rotated spline of R."""
    from scipy.interpolate import CubicSpline

    R = np.asarray(R, dtype=float)
    t = np.asarray(t, dtype=float)
    z = np.array([0.0, 0.0, 0.0, 1.0])
    for _ in range(int(iterations)):
        Rdot = CubicSpline(t, R).derivative()(t)
        halfgammadot = multiply(multiply(Rdot, z), conjugate(R))[..., 0]
        halfgamma = CubicSpline(t, halfgammadot).antiderivative()(t)
        Rgamma = np.zeros_like(R)
        Rgamma[:, 0], Rgamma[:, 3] = np.cos(halfgamma), np.sin(halfgamma)
        R = multiply(R, Rgamma)
    return R


def optimal_alignment_in_Euclidean_metric(avec, bvec, t=None):
    """Rotor R minimising  int |R a(t) R^-1 - b(t)|^2 dt  (a sum over the samples when t is None): numpy-quaternion's
    function of this name, used by scri's rotation_from_vectors (map_to_superrest_frame.py:510-526).  Horn's closed form:
    R is the eigenvector of the largest eigenvalue of the symmetric 4 x 4 matrix built from S_jk = int a_j b_k dt."""
    a = np.asarray(avec, dtype=float).reshape(-1, 3)
    b = np.asarray(bvec, dtype=float).reshape(-1, 3)
    outer = a[:, :, None] * b[:, None, :]
    if t is None:
        S = outer.sum(axis=0)
    else:
        from scipy.interpolate import CubicSpline

        t = np.asarray(t, dtype=float)
        S = CubicSpline(t, outer.reshape(-1, 9)).integrate(t[0], t[-1]).reshape(3, 3)
    N = np.array(
        [
            [S[0, 0] + S[1, 1] + S[2, 2], S[1, 2] - S[2, 1], S[2, 0] - S[0, 2], S[0, 1] - S[1, 0]],
            [S[1, 2] - S[2, 1], S[0, 0] - S[1, 1] - S[2, 2], S[0, 1] + S[1, 0], S[2, 0] + S[0, 2]],
            [S[2, 0] - S[0, 2], S[0, 1] + S[1, 0], -S[0, 0] + S[1, 1] - S[2, 2], S[1, 2] + S[2, 1]],
            [S[0, 1] - S[1, 0], S[2, 0] + S[0, 2], S[1, 2] + S[2, 1], -S[0, 0] - S[1, 1] + S[2, 2]],
        ]
    )
    vals, vecs = np.linalg.eigh(N)
    q = vecs[:, -1]
    return q if q[0] >= 0 else -q
