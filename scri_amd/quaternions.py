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


def as_spinor_array(q):
    """(w + i z, y + i x): quaternion.as_spinor_array (scri/rotations.py:311)."""
    q = np.asarray(q, dtype=float)
    return np.stack([q[..., 0] + 1j * q[..., 3], q[..., 2] + 1j * q[..., 1]], axis=-1)
