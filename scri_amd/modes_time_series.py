"""ModesTimeSeries: SWSH modes as functions of time, with the time calculus and the grid product that the BMS-charge
and super-rest-frame code wraps around the transformation path (scri/modes_time_series.py:7-202; the container
behaviour it inherits from spherical_functions.Modes: spin_weight / ell_min / ell_max metadata, eth, ethbar, bar).

The arithmetic on the series runs on the GPU through the C ABI: interpolate / derivative / antiderivative ->
bms_spline_derivative, grid_multiply -> bms_grid_multiply.  The mode-space operators (eth, ethbar, bar) are diagonal
or permutation maps on the mode axis, applied here with numpy.
"""
import copy
import functools
import math

import numpy as np

from . import engine
from .mode_algebra import LM_index, LM_range, LM_total_size


@functools.lru_cache(maxsize=64)
def _bar_tables(ell_min, ell_max, s):
    """(source index, sign) of every mode of the conjugate function: bar(a)_{l,m} = (-1)^(s+m) conj(a_{l,-m})"""
    perm = np.array([LM_index(ell, -m, ell_min) for ell in range(ell_min, ell_max + 1) for m in range(-ell, ell + 1)], dtype=np.intp)
    sign = np.array([(-1.0) ** (s + m) for ell in range(ell_min, ell_max + 1) for m in range(-ell, ell + 1)])
    return perm, sign


class Grid(np.ndarray):
    """Values of a spin-weighted function on an equiangular (theta, phi) grid -- the last two axes -- with its spin weight `s`: what
    `ModesTimeSeries.grid()` returns (the part of spherical_functions.Grid the charge and frame-fixing code relies on,
    scri/asymptotic_bondi_data/map_to_superrest_frame.py:173: `PsiM.grid().real`).  Products add the weights, quotients subtract them,
    conjugation flips the sign, sums want equal weights."""

    def __new__(cls, values, spin_weight=0):
        obj = np.asarray(values).view(cls)
        obj._s = int(spin_weight)
        return obj

    def __array_finalize__(self, obj):
        self._s = getattr(obj, "_s", 0)

    @property
    def s(self):
        return self._s

    spin_weight = s

    @property
    def n_theta(self):
        return self.shape[-2]

    @property
    def n_phi(self):
        return self.shape[-1]

    @property
    def ndarray(self):
        return self.view(np.ndarray)

    def __array_ufunc__(self, ufunc, method, *inputs, **kwargs):
        weights = [x._s for x in inputs if isinstance(x, Grid)]
        plain = [x.view(np.ndarray) if isinstance(x, Grid) else x for x in inputs]
        if "out" in kwargs:
            kwargs["out"] = tuple(o.view(np.ndarray) if isinstance(o, Grid) else o for o in kwargs["out"])
        result = getattr(ufunc, method)(*plain, **kwargs)
        if method != "__call__" or not isinstance(result, np.ndarray) or result.ndim < 2:
            return result
        if ufunc is np.multiply:
            s = sum(weights)
        elif ufunc in (np.divide, np.true_divide):
            s = (inputs[0]._s if isinstance(inputs[0], Grid) else 0) - (inputs[1]._s if isinstance(inputs[1], Grid) else 0)
        elif ufunc is np.conjugate and weights:
            s = -weights[0]
        elif ufunc is np.absolute:
            s = 0
        elif ufunc is np.square:  # (what numpy makes of x**2, x**-1)
            s = 2 * weights[0]
        elif ufunc is np.reciprocal:
            s = -weights[0]
        elif ufunc in (np.power, np.float_power) and isinstance(inputs[0], Grid) and np.ndim(inputs[1]) == 0:
            s = int(round(inputs[0]._s * float(np.real(inputs[1]))))
        elif ufunc in (np.add, np.subtract) and len(set(weights)) > 1:
            raise ValueError(f"The grids have spin weights {weights[0]} and {weights[1]}; their sum is not a spin-weighted function")
        else:
            s = weights[0] if weights else 0
        return Grid(result, spin_weight=s)

    @property
    def real(self):
        return Grid(self.view(np.ndarray).real, spin_weight=self._s)

    @property
    def imag(self):
        return Grid(self.view(np.ndarray).imag, spin_weight=self._s)


class ModesTimeSeries(np.ndarray):
    """complex ndarray [..., n_times, n_modes] + {time, spin_weight, ell_min, ell_max}."""

    def __new__(cls, input_array, *args, **kwargs):
        if len(args) > 2:
            raise ValueError("Only one positional argument may be passed")
        if len(args) == 1:
            kwargs["time"] = args[0]
        metadata = copy.copy(getattr(input_array, "_metadata", {}))
        metadata.update(**kwargs)
        arr = np.asanyarray(input_array).astype(complex, copy=False)
        time = metadata.get("time", None)
        if time is None:
            raise ValueError("Time data must be specified as part of input array or as constructor parameter")
        time = np.asarray(time, dtype=float)
        if time.ndim != 1:
            raise ValueError(f"Input time array must have exactly 1 dimension; it has {time.ndim}.")
        if arr.ndim == 0:
            arr = arr[np.newaxis, np.newaxis]
        elif arr.ndim == 1:
            arr = arr[np.newaxis, :]
        elif arr.shape[-2] != time.shape[0] and arr.shape[-2] != 1:
            raise ValueError(
                "Second-to-last axis of input array must have size 1 or same size as time array.\n            "
                f"Their shapes are {arr.shape} and {time.shape}, respectively."
            )
        spin_weight = metadata.get("spin_weight", None)
        if spin_weight is None:
            raise ValueError("spin_weight must be given")
        ell_min = int(metadata.get("ell_min", 0))
        ell_max = metadata.get("ell_max", None)
        if ell_max is None:
            ell_max = int(round(math.sqrt(arr.shape[-1] + ell_min**2))) - 1
        if arr.shape[-1] != LM_total_size(ell_min, int(ell_max)):
            raise ValueError(f"Last axis has size {arr.shape[-1]}, inconsistent with ell range [{ell_min}, {ell_max}]")
        obj = arr.view(cls)
        obj._metadata = dict(metadata)
        obj._metadata.update(time=time, spin_weight=int(spin_weight), ell_min=ell_min, ell_max=int(ell_max))
        return obj

    def __array_finalize__(self, obj):
        if obj is None:
            return
        self._metadata = copy.copy(getattr(obj, "_metadata", {}))

    # ------------------------------------------------------------------ metadata
    @property
    def time(self):
        return self._metadata["time"]

    u = time
    t = time

    @property
    def n_times(self):
        return self.time.size

    @property
    def spin_weight(self):
        return self._metadata["spin_weight"]

    s = spin_weight

    @property
    def ell_min(self):
        return self._metadata["ell_min"]

    @property
    def ell_max(self):
        return self._metadata["ell_max"]

    @property
    def ndarray(self):
        return self.view(np.ndarray)

    def index(self, ell, m):
        """Index of the (ell, m) mode along the last axis (spherical_functions.Modes.index; docs/tutorial_abd.rst:138)."""
        from .mode_algebra import LM_index

        if ell < self.ell_min or ell > self.ell_max or abs(m) > ell:
            raise ValueError(f"Requested (ell,m)=({ell},{m}) value is not found in this Modes object")
        return LM_index(ell, m, self.ell_min)

    @property
    def LM(self):
        return LM_range(self.ell_min, self.ell_max)

    def evaluate(self, *directions, ctx=None):
        """The function at the given directions or frames (spherical_functions.Modes.evaluate, the contraction of
        scri/asymptotic_bondi_data/transformations.py:312-334): one argument of rotors [..., 4] (or of [..., 2] = (theta, phi) pairs), or
        two arguments theta, phi.  Returns complex [..., n_times, *directions.shape]; on the GPU (bms_evaluate_modes)."""
        from . import quaternions

        if len(directions) == 1:
            R = np.asarray(directions[0], dtype=float)
            if R.shape[-1] == 2:
                R = quaternions.from_spherical_coords(R[..., 0], R[..., 1])
            elif R.shape[-1] != 4:
                raise ValueError(f"Directions must be rotors [..., 4] or (theta, phi) pairs [..., 2]; the last axis has size {R.shape[-1]}")
        elif len(directions) == 2:
            theta, phi = np.broadcast_arrays(np.asarray(directions[0], dtype=float), np.asarray(directions[1], dtype=float))
            R = quaternions.from_spherical_coords(theta, phi)
        else:
            raise ValueError(f"Can only evaluate one array of rotors or (theta, phi) pairs, or theta and phi; got {len(directions)} arguments")
        return engine.evaluate_modes(self.ndarray, self.s, self.ell_min, self.ell_max, R, ctx=ctx)

    def grid(self, n_theta=None, n_phi=None, ctx=None):
        """The function on the equiangular grid of n_theta x n_phi points (default 2 ell_max + 1 each): spherical_functions.Modes.grid,
        i.e. spinsfast.salm2map on every time slice, as `Grid` [..., n_times, n_theta, n_phi] with this spin weight."""
        n_theta = 2 * self.ell_max + 1 if n_theta is None else int(n_theta)
        n_phi = n_theta if n_phi is None else int(n_phi)
        a = self.ndarray
        if self.ell_min > 0:
            a = np.concatenate([np.zeros(a.shape[:-1] + (self.ell_min**2,), dtype=complex), a], axis=-1)
        return Grid(engine.salm2map(a, self.s, self.ell_max, n_theta, n_phi, ctx=ctx), spin_weight=self.s)

    def _like(self, data, **changes):
        md = dict(self._metadata)
        md.update(changes)
        return type(self)(data, **md)

    # ------------------------------------------------------------------ time calculus (modes_time_series.py:72-126)
    def interpolate(self, new_time, derivative_order=0, out=None):
        new_time = np.asarray(new_time, dtype=float)
        if new_time.ndim != 1:
            raise ValueError(f"New time array must have exactly 1 dimension; it has {new_time.ndim}.")
        new_shape = self.shape[:-2] + (new_time.size, self.shape[-1])
        if out is not None:
            out = np.asarray(out)
            if out.shape != new_shape:
                raise ValueError(
                    f"Output array should have shape {new_shape} for consistency with new time array and modes array"
                )
            if out.dtype != complex:
                raise ValueError(f"Output array should have dtype `complex`; it has dtype {out.dtype}")
        if derivative_order > 3:
            raise ValueError(
                f"{type(self)} interpolation uses CubicSpline, and cannot take a derivative of order {derivative_order}"
            )
        if derivative_order < -16:
            raise NotImplementedError("antiderivatives beyond the sixteenth are not provided")
        data = self.ndarray
        if data.shape[-2] != self.n_times:
            raise ValueError("cannot interpolate a time-independent series")
        # time is the second-to-last axis: move it first for the spline, restore afterwards
        moved = np.moveaxis(data, -2, 0)
        res = engine.spline_derivative(self.time, np.ascontiguousarray(moved), new_time, derivative_order)
        res = np.moveaxis(res, 0, -2)
        if out is not None:
            out[:] = res
            res = out
        return self._like(res, time=new_time)

    def antiderivative(self, antiderivative_order=1):
        """Integrate modes with respect to time"""
        return self.interpolate(self.time, derivative_order=-antiderivative_order)

    def derivative(self, derivative_order=1):
        """Differentiate modes with respect to time"""
        return self.interpolate(self.time, derivative_order=derivative_order)

    @property
    def dot(self):
        return self.derivative()

    @property
    def ddot(self):
        return self.derivative(2)

    @property
    def int(self):
        return self.antiderivative()

    @property
    def iint(self):
        return self.antiderivative(2)

    # ------------------------------------------------------------------ mode-space operators (sf.Modes)
    def _ell_factor(self, f):
        LM = self.LM
        return np.array([f(int(l)) for l in LM[:, 0]])

    @property
    def eth(self):
        """Newman-Penrose eth: x sqrt((l-s)(l+s+1)), spin s -> s+1"""
        s = self.spin_weight
        fac = self._ell_factor(lambda l: math.sqrt((l - s) * (l + s + 1)) if l >= abs(s) and l >= abs(s + 1) else 0.0)
        return self._like(self.ndarray * fac, spin_weight=s + 1)

    @property
    def ethbar(self):
        """Newman-Penrose ethbar: x -sqrt((l+s)(l-s+1)), spin s -> s-1"""
        s = self.spin_weight
        fac = self._ell_factor(lambda l: -math.sqrt((l + s) * (l - s + 1)) if l >= abs(s) and l >= abs(s - 1) else 0.0)
        return self._like(self.ndarray * fac, spin_weight=s - 1)

    @property
    def eth_GHP(self):
        """Raise spin-weight with GHP convention"""
        return self._like(self.eth.ndarray / math.sqrt(2), spin_weight=self.spin_weight + 1)

    @property
    def ethbar_GHP(self):
        """Lower spin-weight with GHP convention"""
        return self._like(self.ethbar.ndarray / math.sqrt(2), spin_weight=self.spin_weight - 1)

    @property
    def bar(self):
        """Modes of the complex-conjugate function: (-1)^(s+m) conj(a_{l,-m}), spin -s"""
        s = self.spin_weight
        perm, sign = _bar_tables(self.ell_min, self.ell_max, s)
        res = np.take(self.ndarray, perm, axis=-1)
        np.conjugate(res, out=res)
        res *= sign
        return self._like(res, spin_weight=-s)

    @property
    def real(self):
        """Modes of the real part of a spin-0 function: (a + bar(a)) / 2 (sf.Modes.real)"""
        if self.spin_weight != 0:
            raise ValueError("The real part of a function with non-zero spin weight is not a spin-weighted function")
        return self._like(0.5 * (self.ndarray + self.bar.ndarray))

    @property
    def imag(self):
        if self.spin_weight != 0:
            raise ValueError("The imaginary part of a function with non-zero spin weight is not a spin-weighted function")
        return self._like(-0.5j * (self.ndarray - self.bar.ndarray))

    def truncate_ell(self, new_ell_max):
        """Copy with modes above new_ell_max dropped (sf.Modes.truncate_ell)"""
        if new_ell_max >= self.ell_max:
            return self._like(self.ndarray.copy())
        if new_ell_max < self.ell_min:
            raise ValueError(f"new ell_max {new_ell_max} is below ell_min {self.ell_min}")
        return self._like(self.ndarray[..., : LM_total_size(self.ell_min, new_ell_max)].copy(), ell_max=new_ell_max)

    def _combine(self, other, op):
        """a +- b for two series of the same spin on the same times; the l ranges may differ (missing modes are zero)."""
        if not isinstance(other, ModesTimeSeries):
            return self._like(op(self.ndarray, other))
        if other.spin_weight != self.spin_weight:
            raise ValueError(f"Cannot add modes with different spin weights ({self.spin_weight} and {other.spin_weight})")
        lo, hi = min(self.ell_min, other.ell_min), max(self.ell_max, other.ell_max)
        a = np.zeros(self.shape[:-1] + (LM_total_size(lo, hi),), dtype=complex)
        b = np.zeros(other.shape[:-1] + (LM_total_size(lo, hi),), dtype=complex)
        a[..., LM_index(self.ell_min, -self.ell_min, lo) : LM_index(self.ell_max, self.ell_max, lo) + 1] = self.ndarray
        b[..., LM_index(other.ell_min, -other.ell_min, lo) : LM_index(other.ell_max, other.ell_max, lo) + 1] = other.ndarray
        return self._like(op(a, b), ell_min=lo, ell_max=hi)

    def __add__(self, other):
        return self._combine(other, np.add)

    def __sub__(self, other):
        return self._combine(other, np.subtract)

    def __neg__(self):
        return self._like(-self.ndarray)

    def __mul__(self, other):
        if isinstance(other, ModesTimeSeries):
            return self.multiply(other)
        return self._like(self.ndarray * other)

    def norm(self):
        """L2 norm over the sphere at every time: sqrt(sum |a_lm|^2) (sf.Modes.norm)"""
        return np.linalg.norm(self.ndarray, axis=-1)

    def __rmul__(self, other):
        return self._like(other * self.ndarray)

    def __truediv__(self, other):
        if isinstance(other, ModesTimeSeries):
            raise ValueError("Cannot divide by a mode series")
        return self._like(self.ndarray / other)

    def multiply(self, other, truncator=None):
        """Modes of the product (sf.Modes.multiply): exact, formed on a grid fine enough for l_a + l_b, truncated to
        `truncator((l_a, l_b))` (default: the sum)."""
        if truncator is None:
            truncator = self._metadata.get("multiplication_truncator", sum)
        return self.grid_multiply(
            other, working_ell_max=self.ell_max + other.ell_max, output_ell_max=int(truncator((self.ell_max, other.ell_max)))
        )

    # ------------------------------------------------------------------ grid product (modes_time_series.py:142-202)
    def grid_multiply(self, mts, **kwargs):
        """Mode weights of the product of two functions, formed on a (2 working_ell_max + 1)^2 grid."""
        output_ell_max = kwargs.pop("output_ell_max", self.ell_max)
        working_ell_max = kwargs.pop("working_ell_max", self.ell_max + mts.ell_max)
        if self.n_times != mts.n_times or not np.equal(self.t, mts.t).all():
            raise ValueError("The time series of objects to be multiplied must be the same.")
        a, b = self._from_ell_0(), mts._from_ell_0()
        prod = engine.grid_multiply(
            a, self.spin_weight, self.ell_max, b, mts.spin_weight, mts.ell_max, working_ell_max, output_ell_max
        )
        return type(self)(prod, time=self.t, spin_weight=self.spin_weight + mts.spin_weight, ell_min=0, ell_max=output_ell_max)

    def _from_ell_0(self):
        """[n_times, (ell_max+1)^2] copy with zeros below ell_min."""
        d = self.ndarray
        if d.ndim != 2 or d.shape[0] != self.n_times:
            raise ValueError("grid_multiply needs a [n_times, n_modes] series")
        if self.ell_min == 0:
            return np.ascontiguousarray(d)
        full = np.zeros((d.shape[0], (self.ell_max + 1) ** 2), dtype=complex)
        full[:, self.ell_min**2 :] = d
        return full
