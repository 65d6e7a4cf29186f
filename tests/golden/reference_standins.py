"""Stand-ins for the third-party packages the reference imports but this image lacks, so that the reference's OWN source
files (/root/reference/scri/*.py, imported unmodified) can be executed in the build container to generate golden vectors
(make_golden_from_reference.py).  Test infrastructure only; nothing under scri_amd/ imports this, and it never travels
anywhere the reference is needed: only the .npz files it produces are committed.

What is stood in, and with what:
  numba                -> `njit` = identity decorator (the two numba kernels of scri/rotations.py run as plain Python loops)
  quaternion           -> a small Python quaternion object type (`np.quaternion`) + the handful of module functions the hot
                          path calls, after numpy-quaternion's documented behaviour
  spherical_functions  -> index algebra, SWSH_grid, Wigner D, eth operators, Modes / Grid array types; arithmetic from
                          oracle/wigner.py (the restated published algorithms)
  spinsfast            -> map2salm / salm2map from oracle/spinsfast_ref.py
  sxs, h5py            -> permissive empty modules (imported by scri/__init__.py, never called on this path)

So the vectors pin the *scri layer* (kwarg handling, mixing signs and term order, trimming, frame bookkeeping, the numba
loops) with the reference's own statements; the sf / spinsfast / quaternion conventions underneath remain pinned by the
analytic known-answer tests (tests/test_oracle_known_answers.py), not by these vectors.
"""
import importlib.abc
import importlib.machinery
import importlib.metadata
import math
import os
import sys
import types

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from oracle import quat as oquat  # noqa: E402
from oracle import spinsfast_ref, wigner  # noqa: E402

REFERENCE = "/root/reference"


# ----------------------------------------------------------------------------------------------- permissive modules


class _Anything:
    def __init__(self, name="x"):
        self._n = name

    def __getattr__(self, k):
        if k.startswith("__") and k.endswith("__"):
            raise AttributeError(k)
        return _Anything(self._n + "." + k)

    def __call__(self, *a, **k):
        if len(a) == 1 and callable(a[0]) and not k:
            return a[0]
        return self

    def __mro_entries__(self, bases):
        return (object,)

    def __iter__(self):
        return iter(())


class _PermissiveModule(types.ModuleType):
    def __getattr__(self, k):
        if k == "__version__":
            return "stand-in"
        if k.startswith("__") and k.endswith("__"):
            raise AttributeError(k)
        return _Anything(self.__name__ + "." + k)


class _Finder(importlib.abc.MetaPathFinder, importlib.abc.Loader):
    roots = ("numba", "sxs", "h5py", "quaternionic", "quaternion", "spherical_functions", "spinsfast")

    def find_spec(self, fullname, path, target=None):
        if fullname.split(".")[0] in self.roots and fullname not in sys.modules:
            return importlib.machinery.ModuleSpec(fullname, self, is_package=True)

    def create_module(self, spec):
        m = _PermissiveModule(spec.name)
        m.__path__ = []
        return m

    def exec_module(self, module):
        pass


# ----------------------------------------------------------------------------------------------- quaternion


class quaternion:
    """numpy-quaternion's scalar type, as far as the hot path uses it: (w, x, y, z) with the Hamilton product."""

    __slots__ = ("w", "x", "y", "z")

    def __init__(self, *c):
        if len(c) == 4:
            self.w, self.x, self.y, self.z = (float(v) for v in c)
        elif len(c) == 3:
            self.w = 0.0
            self.x, self.y, self.z = (float(v) for v in c)
        elif len(c) == 1:
            self.w, self.x, self.y, self.z = float(c[0]), 0.0, 0.0, 0.0
        else:
            raise TypeError("quaternion takes 1, 3 or 4 components")

    # components
    @property
    def components(self):
        return np.array([self.w, self.x, self.y, self.z])

    @property
    def vec(self):
        return np.array([self.x, self.y, self.z])

    @property
    def real(self):
        return self.w

    @property
    def a(self):
        return complex(self.w, self.z)

    @property
    def b(self):
        return complex(self.y, self.x)

    # algebra
    def __mul__(self, o):
        if isinstance(o, quaternion):
            return quaternion(*oquat.qmul(self.components, o.components))
        if isinstance(o, np.ndarray):
            return NotImplemented
        return quaternion(*(self.components * float(o)))

    def __rmul__(self, o):
        if isinstance(o, np.ndarray):
            return NotImplemented
        return quaternion(*(self.components * float(o)))

    def __truediv__(self, o):
        if isinstance(o, quaternion):
            return self * o.inverse()
        return quaternion(*(self.components / float(o)))

    def __add__(self, o):
        return quaternion(*(self.components + (o.components if isinstance(o, quaternion) else np.array([float(o), 0, 0, 0]))))

    def __sub__(self, o):
        return quaternion(*(self.components - (o.components if isinstance(o, quaternion) else np.array([float(o), 0, 0, 0]))))

    def __radd__(self, o):  # scalar + q: the scalar is the real quaternion (numpy-quaternion's ufunc loops for mixed operands)
        return quaternion(*(np.array([float(o), 0, 0, 0]) + self.components))

    def __rsub__(self, o):
        return quaternion(*(np.array([float(o), 0, 0, 0]) - self.components))

    def __neg__(self):
        return quaternion(*(-self.components))

    def __eq__(self, o):
        return isinstance(o, quaternion) and bool(np.all(self.components == o.components))

    def __hash__(self):
        return hash(tuple(self.components))

    def __abs__(self):
        return self.abs()

    def abs(self):
        return math.sqrt(self.norm())

    def norm(self):  # numpy-quaternion: Cayley norm = sum of squares
        return self.w * self.w + self.x * self.x + self.y * self.y + self.z * self.z

    def conjugate(self):
        return quaternion(self.w, -self.x, -self.y, -self.z)

    conj = conjugate
    __invert__ = conjugate  # numpy-quaternion: ~q is the conjugate

    def inverse(self):
        n = self.norm()
        return quaternion(self.w / n, -self.x / n, -self.y / n, -self.z / n)

    def normalized(self):
        n = self.abs()
        return quaternion(self.w / n, self.x / n, self.y / n, self.z / n)

    def exp(self):
        v = math.sqrt(self.x * self.x + self.y * self.y + self.z * self.z)
        e = math.exp(self.w)
        if v > 1e-14:  # numpy-quaternion's _QUATERNION_EPS threshold
            s = e * math.sin(v) / v
            return quaternion(e * math.cos(v), s * self.x, s * self.y, s * self.z)
        return quaternion(e, 0.0, 0.0, 0.0)

    def log(self):
        b = math.sqrt(self.x * self.x + self.y * self.y + self.z * self.z)
        if b <= 1e-14 * abs(self.w):
            if self.w < 0.0:
                if abs(self.w + 1) > 1e-14:
                    return quaternion(math.log(-self.w), math.pi, 0.0, 0.0)
                return quaternion(0.0, math.pi, 0.0, 0.0)
            return quaternion(math.log(self.w), 0.0, 0.0, 0.0)
        v = math.atan2(b, self.w)
        f = v / b
        return quaternion(math.log(self.w * self.w + b * b) / 2.0, f * self.x, f * self.y, f * self.z)

    def sqrt(self):
        a = self.abs()
        if abs(a + self.w) < 1e-14 * a:  # -1: any pure unit vector
            return quaternion(0.0, math.sqrt(a), 0.0, 0.0)
        c = math.sqrt(a / (2 + 2 * self.w / a)) if a > 0 else 0.0
        return quaternion((1.0 + self.w / a) * c, self.x * c / a, self.y * c / a, self.z * c / a)

    def __repr__(self):
        return f"quaternion({self.w!r}, {self.x!r}, {self.y!r}, {self.z!r})"


def _q_float(a):
    a = np.asarray(a)
    if a.dtype == object:
        out = np.empty(a.shape + (4,))
        for idx in np.ndindex(a.shape):
            out[idx] = a[idx].components
        return out
    return np.asarray(a, dtype=float)


def as_float_array(a):
    if isinstance(a, quaternion):
        return a.components
    return _q_float(a)


def as_quat_array(a):
    a = np.asarray(a, dtype=float)
    out = np.empty(a.shape[:-1], dtype=object)
    for idx in np.ndindex(a.shape[:-1]):
        out[idx] = quaternion(*a[idx])
    return out


def as_spinor_array(a):
    return oquat.as_spinor_array(_q_float(a))


def from_spherical_coords(theta, phi=None):
    if phi is None:
        theta, phi = theta
    return quaternion(*oquat.from_spherical_coords(float(theta), float(phi)))


def as_spherical_coords(q):
    th, ph = oquat.as_spherical_coords(q.components)
    return np.array([float(th), float(ph)])


def rotate_vectors(R, v, axis=-1):
    """quaternion.rotate_vectors for a single vector v: result shape R.shape + (3,)."""
    Rf = _q_float(R)
    v = np.asarray(v, dtype=float)
    out = np.empty(Rf.shape[:-1] + (3,))
    for idx in np.ndindex(Rf.shape[:-1]):
        q = quaternion(*Rf[idx])
        out[idx] = (q * quaternion(0.0, *v) * q.inverse()).vec / 1.0
    return out


def make_quaternion_module():
    m = _PermissiveModule("quaternion")
    m.__path__ = []
    m.quaternion = quaternion
    m.one = quaternion(1, 0, 0, 0)
    m.x = quaternion(0, 1, 0, 0)
    m.y = quaternion(0, 0, 1, 0)
    m.z = quaternion(0, 0, 0, 1)
    m.as_float_array = as_float_array
    m.as_quat_array = as_quat_array
    m.as_spinor_array = as_spinor_array
    m.from_spherical_coords = from_spherical_coords
    m.as_spherical_coords = as_spherical_coords
    m.rotate_vectors = rotate_vectors
    m.from_rotation_vector = from_rotation_vector
    calculus = sys.modules.get("quaternion.calculus") or _PermissiveModule("quaternion.calculus")  # numpy-quaternion's calculus.py: FITPACK cubic splines through scipy
    calculus.__path__ = []
    calculus.indefinite_integral = calculus.spline_indefinite_integral = _spline_indefinite_integral
    m.calculus = calculus
    m.indefinite_integral = m.spline_indefinite_integral = _spline_indefinite_integral  # (numpy-quaternion re-exports them at the top level)
    sys.modules["quaternion.calculus"] = calculus
    m.as_vector_part = lambda q: np.array(q.vec) if isinstance(q, quaternion) else _q_float(q)[..., 1:]
    return m


def _spline_indefinite_integral(f, t, t_out=None, axis=0):
    """quaternion.calculus.spline_indefinite_integral: the antiderivative (zero at t[0]) of the degree-3 InterpolatedUnivariateSpline
    through f(t), evaluated at t_out (default t), along `axis`"""
    from scipy.interpolate import InterpolatedUnivariateSpline

    f = np.asarray(f)
    t = np.asarray(t, dtype=float)
    t_out = t if t_out is None else np.asarray(t_out, dtype=float)
    fm = np.moveaxis(f, axis, 0)
    flat = fm.reshape(fm.shape[0], -1)

    def one(col):
        return InterpolatedUnivariateSpline(t, col, k=3).antiderivative()(t_out)

    cols = [one(flat[:, i].real) + (1j * one(flat[:, i].imag) if np.iscomplexobj(flat) else 0) for i in range(flat.shape[1])]
    out = np.stack(cols, axis=1).reshape((t_out.size,) + fm.shape[1:])
    return np.moveaxis(out, 0, axis)


def from_rotation_vector(rot):
    """quaternion.from_rotation_vector: exp(v / 2) for the rotation vector v (axis x angle)"""
    rot = np.asarray(rot, dtype=float)
    return quaternion(0.0, *(rot / 2.0)).exp()


# ----------------------------------------------------------------------------------------------- spherical_functions


def _rotor_floats(R):
    return _q_float(R)


class Grid(np.ndarray):
    """sf.Grid: values on a grid with a spin weight `s` that follows products, quotients and powers."""

    def __new__(cls, arr, spin_weight=None, **kw):
        obj = np.asarray(arr).view(cls)
        obj._s = spin_weight if spin_weight is not None else getattr(arr, "_s", 0)
        return obj

    def __array_finalize__(self, obj):
        self._s = getattr(obj, "_s", 0)

    @property
    def s(self):
        return self._s

    spin_weight = s

    def __array_ufunc__(self, ufunc, method, *inputs, out=None, **kwargs):
        # sf.Grid: the last two axes are the grid; an ordinary array operand is broadcast against the leading axes
        raw = [
            np.asarray(i).view(np.ndarray) if isinstance(i, Grid)
            else (i[..., np.newaxis, np.newaxis] if isinstance(i, np.ndarray) and i.ndim >= 1 else i)
            for i in inputs
        ]
        spins = [i._s if isinstance(i, Grid) else 0 for i in inputs]
        if out is not None:
            kwargs["out"] = tuple(o.view(np.ndarray) if isinstance(o, Grid) else o for o in out)
        res = getattr(ufunc, method)(*raw, **kwargs)
        if ufunc in (np.multiply,):
            s = sum(spins)
        elif ufunc in (np.true_divide, np.divide):
            s = spins[0] - spins[1]
        elif ufunc in (np.power, np.float_power):
            s = spins[0] * int(np.asarray(raw[1]).item()) if np.ndim(raw[1]) == 0 else spins[0]
        elif ufunc in (np.add, np.subtract):
            both = [sp for sp, i in zip(spins, inputs) if isinstance(i, Grid)]
            if len(set(both)) > 1:
                raise ValueError(f"adding grids of spin weights {both}")
            s = both[0]
        elif ufunc in (np.conjugate,):
            s = -spins[0]
        else:
            s = spins[0]
        if out is not None:
            o = out[0]
            if isinstance(o, Grid):
                o._s = s
            return o
        if isinstance(res, np.ndarray):
            res = res.view(Grid)
            res._s = s
        return res

    @property
    def real(self):
        return Grid(np.asarray(self).real, spin_weight=self._s)


class Modes(np.ndarray):
    """sf.Modes: mode weights [..., (ell_max+1)^2 - ell_min^2] of a spin-weighted function, l from ell_min (0 here)."""

    def __new__(cls, input_array, **kwargs):
        metadata = dict(getattr(input_array, "_metadata", {}))
        metadata.update(kwargs)
        arr = np.asanyarray(input_array).view(np.ndarray)
        if arr.dtype != complex:
            arr = arr.astype(complex)
        obj = arr.view(cls)
        metadata.setdefault("ell_min", 0)
        if metadata.get("ell_max") is None:
            metadata["ell_max"] = int(round(math.sqrt(arr.shape[-1] + metadata["ell_min"] ** 2))) - 1
        if metadata.get("spin_weight") is None:
            raise ValueError("Spin weight must be specified")
        metadata.setdefault("multiplication_truncator", sum)
        obj._metadata = metadata
        return obj

    def __array_finalize__(self, obj):
        if obj is None:
            return
        self._metadata = dict(getattr(obj, "_metadata", {}))

    s = property(lambda self: self._metadata["spin_weight"])
    spin_weight = s
    ell_min = property(lambda self: self._metadata["ell_min"])
    ell_max = property(lambda self: self._metadata["ell_max"])
    multiplication_truncator = property(lambda self: self._metadata["multiplication_truncator"])

    @property
    def ndarray(self):
        return self.view(np.ndarray)

    def index(self, ell, m):
        return wigner.LM_index(ell, m, self.ell_min)

    def _like(self, arr, **changes):
        md = dict(self._metadata)
        md.update(changes)
        return type(self)(arr, **md) if type(self) is Modes else Modes(arr, **{k: v for k, v in md.items() if k != "time"})

    @property
    def eth(self):
        """Newman-Penrose eth: s -> s + 1, weights x sqrt((l - s)(l + s + 1))."""
        return self._with(wigner.eth_NP(self.view(np.ndarray), self.s, self.ell_min), self.s + 1)

    @property
    def ethbar(self):
        return self._with(wigner.ethbar_NP(self.view(np.ndarray), self.s, self.ell_min), self.s - 1)

    def _with(self, arr, s):
        md = dict(self._metadata)
        md["spin_weight"] = s
        out = np.asarray(arr).view(type(self))
        out._metadata = md
        return out

    @property
    def real(self):
        """Modes of the real part of the (spin-0) function: (f_lm + (-1)^m conj f_l,-m) / 2."""
        a = self.view(np.ndarray)
        out = np.empty_like(a)
        for ell in range(self.ell_min, self.ell_max + 1):
            for m in range(-ell, ell + 1):
                i, j = self.index(ell, m), self.index(ell, -m)
                out[..., i] = (a[..., i] + (-1.0) ** m * np.conj(a[..., j])) / 2
        return self._with(out, self.s)

    @property
    def imag(self):
        """Modes of the imaginary part of the (spin-0) function, itself a real function: (f_lm - (-1)^m conj f_l,-m) / 2i
        (scri/asymptotic_bondi_data/from_initial_values.py:106 subtracts 1j times it from psi2.real to set Im psi2)."""
        a = self.view(np.ndarray)
        out = np.empty_like(a)
        for ell in range(self.ell_min, self.ell_max + 1):
            for m in range(-ell, ell + 1):
                i, j = self.index(ell, m), self.index(ell, -m)
                out[..., i] = (a[..., i] - (-1.0) ** m * np.conj(a[..., j])) / 2j
        return self._with(out, self.s)

    @property
    def bar(self):
        """Modes of the conjugate function: spin -s, weights (-1)^(s+m) conj f_l,-m."""
        a = self.view(np.ndarray)
        out = np.empty_like(a)
        for ell in range(self.ell_min, self.ell_max + 1):
            for m in range(-ell, ell + 1):
                out[..., self.index(ell, m)] = (-1.0) ** (self.s + m) * np.conj(a[..., self.index(ell, -m)])
        return self._with(out, -self.s)

    # ---- what scri/asymptotic_bondi_data/bms_charges.py asks of sf.Modes beyond the above (third-party behaviour, restated)
    @property
    def eth_GHP(self):
        return self._with(wigner.eth_GHP(self.view(np.ndarray), self.s, self.ell_min), self.s + 1)

    @property
    def ethbar_GHP(self):
        return self._with(wigner.ethbar_GHP(self.view(np.ndarray), self.s, self.ell_min), self.s - 1)

    def truncate_ell(self, new_ell_max):
        """the modes l <= new_ell_max (a copy; more modes than present is an error in sf)"""
        if new_ell_max >= self.ell_max:
            return self
        md = dict(self._metadata)
        md["ell_max"] = int(new_ell_max)
        out = np.array(self.view(np.ndarray)[..., : (new_ell_max + 1) ** 2 - self.ell_min**2]).view(type(self))
        out._metadata = md
        return out

    def norm(self):
        return np.linalg.norm(self.view(np.ndarray), axis=-1)

    def grid(self, n_theta=None, n_phi=None, **kwargs):
        """sf.Modes.grid: the function on the equiangular grid spinsfast.salm2map gives, (2 ell_max + 1)^2 points unless told otherwise,
        as an sf.Grid carrying the spin weight"""
        n_theta = 2 * self.ell_max + 1 if n_theta is None else int(n_theta)
        n_phi = n_theta if n_phi is None else int(n_phi)
        if self.ell_min != 0:
            raise NotImplementedError
        return Grid(spinsfast_ref.salm2map(self.view(np.ndarray), self.s, self.ell_max, n_theta, n_phi), spin_weight=self.s)

    def multiply(self, other, truncator=None):
        """sf.Modes.multiply: the exact product of two band-limited functions (Wigner-3j sums there; here both are put on a grid
        that resolves the product, multiplied and analysed -- the same modes to rounding), truncated at truncator((l_a, l_b))."""
        from oracle import modes_time_series_ref as mts_ref

        if truncator is None:
            truncator = self.multiplication_truncator
        if self.ell_min != 0 or other.ell_min != 0:
            raise NotImplementedError
        out_ell = int(truncator((self.ell_max, other.ell_max)))
        prod = mts_ref.grid_multiply(self.view(np.ndarray), self.s, self.ell_max, np.asarray(other).view(np.ndarray), other.s, other.ell_max,
                                     working_ell_max=self.ell_max + other.ell_max, output_ell_max=min(out_ell, self.ell_max + other.ell_max))
        md = dict(self._metadata)
        md["spin_weight"] = self.s + other.s
        md["ell_max"] = min(out_ell, self.ell_max + other.ell_max)
        out = np.asarray(prod).view(type(self))
        out._metadata = md
        return out

    def __mul__(self, other):
        if isinstance(other, Modes):
            return self.multiply(other)
        return self._scaled_by(other)

    def __rmul__(self, other):
        return self._scaled_by(other)

    def _scaled_by(self, other):
        """sf.Modes times something that is not Modes: a scalar, or an array whose axes are the LEADING axes of the modes (the
        mode axis is appended to it: `abd.t * modes` scales every time step, bms_charges.py:166)"""
        o = np.asarray(other)
        if o.ndim >= 1:
            o = o[..., np.newaxis]
        return self._with(self.view(np.ndarray) * o, self.s)

    __array_priority__ = 100.0

    def evaluate(self, R):
        """Values at the rotors R (array of quaternions, any shape): shape self.shape[:-1] + R.shape."""
        a = self.view(np.ndarray)
        if self.ell_min != 0:
            raise NotImplementedError
        return wigner.modes_evaluate(a, _rotor_floats(R), self.s)


def _Wigner_D_matrices(Ra, Rb, ell_min, ell_max, D):
    D[:] = wigner.wigner_D_matrices(complex(Ra), complex(Rb), ell_min, ell_max)
    return D


class WignerD:
    _total_size_D_matrices = staticmethod(wigner.total_size_D_matrices)


def SWSH_grid(R, s, ell_max):
    return wigner.swsh_grid(_rotor_floats(R), s, ell_max)


def theta_phi(n_theta, n_phi):
    return np.array([[[th, ph] for ph in np.linspace(0.0, 2 * np.pi, num=n_phi, endpoint=False)]
                     for th in np.linspace(0.0, np.pi, num=n_theta, endpoint=True)])


_CG_CACHE = {}
_W3J_CACHE = {}


def _wigner_3j(j1, j2, j3, m1, m2, m3):
    """sf.Wigner3j(j_1, j_2, j_3, m_1, m_2, m_3), from sympy's exact value"""
    key = (int(j1), int(j2), int(j3), int(m1), int(m2), int(m3))
    if key not in _W3J_CACHE:
        from sympy.physics.wigner import wigner_3j

        _W3J_CACHE[key] = float(wigner_3j(*key))
    return _W3J_CACHE[key]


def _clebsch_gordan(j1, m1, j2, m2, j3, m3):
    """sf.clebsch_gordan(j_1, m_1, j_2, m_2, j_3, m_3) = <j1 m1 j2 m2 | j3 m3> (Condon-Shortley), here from sympy's exact value"""
    key = (j1, m1, j2, m2, j3, m3)
    if key not in _CG_CACHE:
        from sympy.physics.quantum.cg import CG

        _CG_CACHE[key] = float(CG(j1, m1, j2, m2, j3, m3).doit()) if abs(m1) <= j1 and abs(m2) <= j2 and abs(m3) <= j3 else 0.0
    return _CG_CACHE[key]


def make_sf_module():
    m = _PermissiveModule("spherical_functions")
    m.__path__ = []
    m.__version__ = "stand-in"
    for name in ("LM_index", "LM_total_size", "LM_range", "constant_as_ell_0_mode", "constant_from_ell_0_mode",
                 "vector_as_ell_1_modes", "vector_from_ell_1_modes"):
        setattr(m, name, getattr(wigner, name))
    m.eth_GHP = lambda modes, spin_weight=0, ell_min=0: wigner.eth_GHP(modes, spin_weight, ell_min)
    m.ethbar_GHP = lambda modes, spin_weight=0, ell_min=0: wigner.ethbar_GHP(modes, spin_weight, ell_min)
    m.SWSH_grid = SWSH_grid
    m.ladder_operator_coefficient = lambda ell, m_: math.sqrt(ell * (ell + 1) - m_ * (m_ + 1))  # <l, m+1| L+ |l, m>
    m.clebsch_gordan = _clebsch_gordan
    m.Wigner3j = _wigner_3j  # (scri/sample_waveforms.py:364: exact values from sympy here)
    m._Wigner_D_matrices = _Wigner_D_matrices
    m._linear_matrix_offset = wigner.linear_matrix_offset
    m.WignerD = WignerD
    m.theta_phi = theta_phi
    m.Modes = Modes
    m.Grid = Grid
    swsh_grids = types.ModuleType("spherical_functions.SWSH_grids")  # (map_to_superrest_frame.py:172 names Grid through its module)
    swsh_grids.Grid = Grid
    m.SWSH_grids = swsh_grids
    sys.modules["spherical_functions.SWSH_grids"] = swsh_grids
    swsh_modes = types.ModuleType("spherical_functions.SWSH_modes")  # (scri/modes_time_series.py:190 names the class through its module)
    swsh_modes.Modes = Modes
    m.SWSH_modes = swsh_modes
    sys.modules["spherical_functions.SWSH_modes"] = swsh_modes
    return m


def make_spinsfast_module():
    m = _PermissiveModule("spinsfast")
    m.__path__ = []
    m.map2salm = lambda f, s, lmax: spinsfast_ref.map2salm(np.asarray(f).view(np.ndarray), s, lmax)
    m.salm2map = lambda salm, s, lmax, Ntheta, Nphi: spinsfast_ref.salm2map(np.asarray(salm).view(np.ndarray), s, lmax, Ntheta, Nphi)
    m.N_lm = lambda lmax: (lmax + 1) ** 2
    return m


def install():
    """Put the stand-ins in sys.modules and the reference on sys.path; returns the imported reference package."""
    if not os.path.isdir(os.path.join(REFERENCE, "scri")):
        raise RuntimeError(f"{REFERENCE} is not here: golden vectors are generated in the build container only")
    sys.meta_path.insert(0, _Finder())
    numba = _PermissiveModule("numba")
    numba.__path__ = []

    def njit(*a, **k):
        if len(a) == 1 and callable(a[0]):
            return a[0]
        return lambda f: f

    numba.njit = numba.jit = njit
    numba.complex128, numba.float64, numba.int64 = np.complex128, np.float64, np.int64  # (dtype arguments of np.zeros in scri/flux.py)
    sys.modules["numba"] = numba
    q = make_quaternion_module()
    sys.modules["quaternion"] = q
    np.quaternion = quaternion
    _isfinite = np.isfinite

    def isfinite(x, *a, **k):  # numpy-quaternion gives its dtype an isfinite loop; object arrays have none
        if isinstance(x, np.ndarray) and x.dtype == object:
            return _isfinite(_q_float(x)).all(axis=-1) if x.size else np.ones(x.shape, dtype=bool)
        return _isfinite(x, *a, **k)

    np.isfinite = isfinite
    sys.modules["spherical_functions"] = make_sf_module()
    sys.modules["spinsfast"] = make_spinsfast_module()
    _version = importlib.metadata.version
    importlib.metadata.version = lambda name: "2024.0.13" if name == "scri" else _version(name)
    sys.path.insert(0, REFERENCE)
    import scri

    assert os.path.dirname(scri.__file__) == os.path.join(REFERENCE, "scri")
    return scri
