"""Oracle for scri/rotations.py:284-392 (rotate_decomposition_basis and its two numba kernels).

`data` is complex [N, n_modes] with modes (l,m), l = ell_min..ell_max; D is the packed
Wigner-D array, block l at ``linear_matrix_offset(l, ell_min)``, row-major (m', m).
"""
import numpy as np
from . import quat
from .wigner import wigner_D_matrices, linear_matrix_offset


def rotate_by_constant(data, ell_min, ell_max, D):
    """scri/rotations.py:346-367: data[t, l, m] <- sum_m' data[t, l, m'] D^l[m', m]."""
    out = np.array(data, dtype=complex, copy=True)
    for ell in range(ell_min, ell_max + 1):
        i0 = ell**2 - ell_min**2
        n = 2 * ell + 1
        iD = linear_matrix_offset(ell, ell_min)
        Dl = D[iD : iD + n * n].reshape(n, n)
        out[:, i0 : i0 + n] = data[:, i0 : i0 + n] @ Dl
    return out


def rotate_by_series(data, RaRb, ell_min, ell_max):
    """scri/rotations.py:370-392: one D per time step, from spinor pairs RaRb[N, 2]."""
    out = np.array(data, dtype=complex, copy=True)
    D = wigner_D_matrices(RaRb[:, 0], RaRb[:, 1], ell_min, ell_max)  # [N, size]
    for ell in range(ell_min, ell_max + 1):
        i0 = ell**2 - ell_min**2
        n = 2 * ell + 1
        iD = linear_matrix_offset(ell, ell_min)
        Dl = D[:, iD : iD + n * n].reshape(-1, n, n)
        out[:, i0 : i0 + n] = np.einsum("tp,tpm->tm", data[:, i0 : i0 + n], Dl)
    return out


def rotate_decomposition_basis(w, R_basis):
    """scri/rotations.py:284-343 on an oracle WM container (returns a new WM; the reference
    works in place).  R_basis: quaternion [4] or array [N,4] (or length-1 list)."""
    R = np.asarray(R_basis, dtype=float)
    if R.ndim == 2 and R.shape[0] == 1:
        R = R[0]
    out = w.copy()
    if R.ndim == 2:
        if R.shape[0] != w.n_times:
            raise ValueError(
                "Input dimension mismatch.  (W.n_times={}) != (len(R_basis)={})".format(w.n_times, R.shape[0])
            )
        out.data = rotate_by_series(w.data, quat.as_spinor_array(R), w.ell_min, w.ell_max)
        if w.frame.size:
            out.frame = quat.qmul(w.frame if w.frame.shape[0] != 1 else np.repeat(w.frame, R.shape[0], 0), R)
        else:
            out.frame = R.copy()
    else:
        Ra, Rb = quat.as_spinor_array(R)
        D = wigner_D_matrices(Ra, Rb, w.ell_min, w.ell_max)
        out.data = rotate_by_constant(w.data, w.ell_min, w.ell_max, D)
        out.frame = quat.qmul(w.frame, R) if w.frame.size else R[None, :].copy()
    return out
