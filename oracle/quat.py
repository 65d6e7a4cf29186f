"""Quaternion helpers (numpy arrays [..., 4] ordered w, x, y, z).

Restates the few ``numpy-quaternion`` (>=2024.0.2, un-vendored) operations the hot path uses:
call sites ``scri/waveform_grid.py:113-116,141-174``, ``scri/rotations.py:311``,
``scri/asymptotic_bondi_data/transformations.py:100-148,183``.
"""
import numpy as np


def qmul(a, b):
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


def qconj(a):
    a = np.asarray(a, dtype=float)
    return a * np.array([1.0, -1.0, -1.0, -1.0])


def qabs(a):
    return np.sqrt(np.sum(np.asarray(a, dtype=float) ** 2, axis=-1))


def qnormalized(a):
    a = np.asarray(a, dtype=float)
    return a / qabs(a)[..., None]


def qinverse(a):
    a = np.asarray(a, dtype=float)
    return qconj(a) / np.sum(a * a, axis=-1)[..., None]


def qexp(a):
    """exp of a quaternion (numpy-quaternion ``quaternion.exp``)."""
    a = np.asarray(a, dtype=float)
    v = a[..., 1:]
    vn = np.sqrt(np.sum(v * v, axis=-1))
    e = np.exp(a[..., 0])
    out = np.empty_like(a)
    with np.errstate(invalid="ignore", divide="ignore"):
        s = np.where(vn > 1e-300, np.sin(vn) / np.where(vn > 1e-300, vn, 1.0), 1.0)
    out[..., 0] = e * np.cos(vn)
    out[..., 1:] = e[..., None] * s[..., None] * v
    return out


def from_spherical_coords(theta, phi):
    """R = exp(phi k/2) exp(theta j/2)  (quaternion.from_spherical_coords)."""
    theta = np.asarray(theta, dtype=float)
    phi = np.asarray(phi, dtype=float)
    ct, st = np.cos(theta / 2), np.sin(theta / 2)
    cp, sp = np.cos(phi / 2), np.sin(phi / 2)
    return np.stack([cp * ct, -sp * st, cp * st, sp * ct], axis=-1)


# numpy-quaternion extracts theta = 2 arccos(sqrt((w^2+z^2)/|q|^2)); near the poles that formula turns the 1-ulp
# rounding of w^2 + z^2 into +-3e-8 rad (arccos near 1), i.e. the reference's own boosted rotor grid is only defined
# to ~|v| 3e-8 at pole pixels.  ROBUST_POLES = True switches to the well-conditioned 2 arctan2(|(x,y)|, |(w,z)|) (the
# same angle), which is what the product uses; the GPU parity tests of boosted transforms run with it so that they can
# keep a 1e-12 bar, the known-answer tests run with the literal formula.
ROBUST_POLES = False


def as_spherical_coords(q):
    """(theta, phi) of q = (beta, alpha) of its z-y-z Euler angles
    (quaternion.as_spherical_coords = as_euler_angles(q)[..., 1::-1])."""
    q = np.asarray(q, dtype=float)
    n = np.sum(q * q, axis=-1)
    alpha = np.arctan2(q[..., 3], q[..., 0]) + np.arctan2(-q[..., 1], q[..., 2])
    if ROBUST_POLES:
        beta = 2 * np.arctan2(np.sqrt(q[..., 1] ** 2 + q[..., 2] ** 2), np.sqrt(q[..., 0] ** 2 + q[..., 3] ** 2))
    else:
        beta = 2 * np.arccos(np.sqrt(np.clip((q[..., 0] ** 2 + q[..., 3] ** 2) / n, 0.0, 1.0)))
    return beta, alpha


def rotate_z(q):
    """q z q^-1 as a 3-vector (``R * quaternion.z * R.inverse()``; waveform_grid.py:472)."""
    q = np.asarray(q, dtype=float)
    z = np.zeros(q.shape)
    z[..., 3] = 1.0
    return qmul(qmul(q, z), qinverse(q))[..., 1:]


def as_spinor_array(q):
    """(Ra, Rb) = (w + i z, y + i x)  (quaternion.as_spinor_array; rotations.py:311)."""
    q = np.asarray(q, dtype=float)
    return np.stack([q[..., 0] + 1j * q[..., 3], q[..., 2] + 1j * q[..., 1]], axis=-1)
