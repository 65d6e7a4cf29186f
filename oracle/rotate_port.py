"""ctypes loader of oracle/librotate_port.so (C port of scri/rotations.py:346-392; test infrastructure only)."""
import ctypes
import os

import numpy as np

from .wigner import total_size_D_matrices

_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "librotate_port.so")
_lib = None


def load():
    global _lib
    if _lib is None:
        if not os.path.exists(_PATH):
            raise ImportError(f"{_PATH} missing: run `make -C oracle`")
        _lib = ctypes.CDLL(_PATH)
    return _lib


def wigner_D_matrices(Ra, Rb, ell_min, ell_max):
    D = np.zeros(total_size_D_matrices(ell_min, ell_max), dtype=np.complex128)
    f = load().wigner_D_matrices_p
    f.restype = None
    f.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
    sp = np.array([complex(Ra).real, complex(Ra).imag, complex(Rb).real, complex(Rb).imag])
    f(sp.ctypes.data, ell_min, ell_max, D.ctypes.data)
    return D


def rotate_by_series(data, RaRb, ell_min, ell_max):
    out = np.ascontiguousarray(data, dtype=np.complex128).copy()
    R = np.ascontiguousarray(RaRb, dtype=np.complex128)
    D = np.zeros(total_size_D_matrices(ell_min, ell_max), dtype=np.complex128)
    f = load().rotate_by_series
    f.restype = None
    f.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, ctypes.c_int64, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
    f(out.ctypes.data, R.ctypes.data, out.shape[0], out.shape[1], ell_min, ell_max, D.ctypes.data)
    return out


def rotate_by_series_omp(data, RaRb, ell_min, ell_max, n_threads=0):
    """OpenMP-over-time variant (all cores unless n_threads > 0): returns (rotated data, threads used)."""
    out = np.ascontiguousarray(data, dtype=np.complex128).copy()
    R = np.ascontiguousarray(RaRb, dtype=np.complex128)
    d_size = total_size_D_matrices(ell_min, ell_max)
    nt = n_threads if n_threads > 0 else (os.cpu_count() or 1)
    D = np.zeros(d_size * nt, dtype=np.complex128)
    f = load().rotate_by_series_omp
    f.restype = ctypes.c_int
    f.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, ctypes.c_int64, ctypes.c_int, ctypes.c_int, ctypes.c_void_p,
                  ctypes.c_int64, ctypes.c_int]
    used = f(out.ctypes.data, R.ctypes.data, out.shape[0], out.shape[1], ell_min, ell_max, D.ctypes.data, d_size, nt)
    return out, used
