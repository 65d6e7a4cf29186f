"""ModesTimeSeries resident in HBM ("BMS charges + map_to_superrest_frame loop kept device-resident").

`DeviceModesTimeSeries` has the operator interface of `scri_amd.ModesTimeSeries` (itself the mirror of
scri/modes_time_series.py:7-202 and of the spherical_functions.Modes algebra it inherits) but its mode weights live in a
device buffer, and every operation is a launch through the C ABI:

    eth, ethbar, bar, real, imag, +, -, scalar and per-row factors, truncate_ell    bms_mode_map     (kernels_modes.hip)
    multiply / grid_multiply                                                         bms_grid_multiply
    interpolate, dot, ddot, int, iint                                                bms_spline_derivative

so the BMS-charge formulas (scri_amd/bms_charges.py, written against that interface) and the frame-fixing iterations
(scri_amd/map_to_superrest_frame.py) run between transformations without the fields crossing PCIe; only what the control
loop reads (l <= 1 charge vectors, a window of rows around u = 0) comes to the host.

PyTorch provides the device allocations (plumbing); the context's stream is set to torch's current stream so that
allocation, the occasional device-to-device copy and the engine's kernels are ordered on one stream.
"""
import ctypes
import functools
import math

import numpy as np

from . import _lib
from ._lib import BMS_DEVICE, c_vp
from .mode_algebra import LM_index, LM_range, LM_total_size


def _torch():
    import torch

    return torch


def attach(ctx):
    """Order the context's work with torch's (one stream for allocations, copies and kernels); returns the torch device."""
    torch = _torch()
    dev = torch.device("cuda", ctx.device)
    if not getattr(ctx, "_torch_stream_attached", False):
        ctx.set_stream(torch.cuda.current_stream(dev).cuda_stream)
        ctx._torch_stream_attached = True
    return dev


def empty(ctx, shape):
    torch = _torch()
    return torch.empty(tuple(int(x) for x in shape), dtype=torch.complex128, device=attach(ctx))


def to_device(ctx, array):
    """numpy (complex) array -> device tensor"""
    torch = _torch()
    a = np.ascontiguousarray(array, dtype=np.complex128)
    return torch.from_numpy(a).to(attach(ctx))


def to_host(tensor):
    return tensor.cpu().numpy()


# ------------------------------------------------------------------------------------------------ per-column tables


@functools.lru_cache(maxsize=256)
def _ell_of_columns(ell_min, ell_max):
    return LM_range(ell_min, ell_max)[:, 0].astype(float)


@functools.lru_cache(maxsize=256)
def _identity(n):
    return np.arange(n, dtype=np.int32)


@functools.lru_cache(maxsize=256)
def _eth_factor(ell_min, ell_max, s, raising):
    ell = _ell_of_columns(ell_min, ell_max)
    if raising:  # Newman-Penrose eth: sqrt((l - s)(l + s + 1)), spin s -> s + 1
        f = np.where((ell >= abs(s)) & (ell >= abs(s + 1)), np.sqrt(np.maximum((ell - s) * (ell + s + 1), 0.0)), 0.0)
    else:  # ethbar: -sqrt((l + s)(l - s + 1)), spin s -> s - 1
        f = np.where((ell >= abs(s)) & (ell >= abs(s - 1)), -np.sqrt(np.maximum((ell + s) * (ell - s + 1), 0.0)), 0.0)
    return f.astype(np.complex128)


@functools.lru_cache(maxsize=256)
def _bar_tables(ell_min, ell_max, s):
    """bar(a)_{l,m} = (-1)^(s+m) conj(a_{l,-m}): (source column, sign) per column"""
    perm = np.array([LM_index(ell, -m, ell_min) for ell in range(ell_min, ell_max + 1) for m in range(-ell, ell + 1)], dtype=np.int32)
    sign = np.array([(-1.0) ** (s + m) for ell in range(ell_min, ell_max + 1) for m in range(-ell, ell + 1)], dtype=np.complex128)
    return perm, sign


@functools.lru_cache(maxsize=256)
def _embed(src_min, src_max, dst_min, dst_max):
    """source column of every column of the l range [dst_min, dst_max] in a series stored on [src_min, src_max]; -1 outside"""
    idx = np.full(LM_total_size(dst_min, dst_max), -1, dtype=np.int32)
    for ell in range(max(src_min, dst_min), min(src_max, dst_max) + 1):
        a = LM_index(ell, -ell, dst_min)
        b = LM_index(ell, -ell, src_min)
        idx[a : a + 2 * ell + 1] = np.arange(b, b + 2 * ell + 1)
    return idx


class DeviceModesTimeSeries:
    """complex128 device buffer [n_times, n_modes] + {time, spin_weight, ell_min, ell_max}."""

    __array_priority__ = 1000  # numpy_array * series -> series.__rmul__
    __array_ufunc__ = None

    def __init__(self, buf, time, spin_weight, ell_min=0, ell_max=None, ctx=None, multiplication_truncator=sum):
        self._ctx = ctx if ctx is not None else _lib.default_context()
        attach(self._ctx)
        self.buf = buf  # torch.complex128 [n_times, n_modes], unit stride along the modes
        self._time = np.asarray(time, dtype=float)
        if self.buf.ndim != 2 or self.buf.shape[0] != self._time.shape[0] or self.buf.stride(1) != 1:
            raise ValueError(f"device series needs a [n_times, n_modes] buffer with contiguous modes; got {tuple(self.buf.shape)}")
        self.spin_weight = int(spin_weight)
        self.ell_min = int(ell_min)
        if ell_max is None:
            ell_max = int(round(math.sqrt(self.buf.shape[1] + self.ell_min**2))) - 1
        self.ell_max = int(ell_max)
        if self.buf.shape[1] != LM_total_size(self.ell_min, self.ell_max):
            raise ValueError(f"last axis has size {self.buf.shape[1]}, inconsistent with the ell range [{self.ell_min}, {self.ell_max}]")
        self._truncator = multiplication_truncator

    # ------------------------------------------------------------------ construction / transfer
    @classmethod
    def from_host(cls, mts, ctx=None):
        """from a host ModesTimeSeries (or array + metadata attributes)"""
        ctx = ctx if ctx is not None else _lib.default_context()
        return cls(to_device(ctx, np.asarray(mts)), mts.t, mts.spin_weight, mts.ell_min, mts.ell_max, ctx=ctx,
                   multiplication_truncator=getattr(mts, "_metadata", {}).get("multiplication_truncator", sum))

    def to_host(self):
        from .modes_time_series import ModesTimeSeries

        return ModesTimeSeries(self.ndarray, self._time, spin_weight=self.spin_weight, ell_min=self.ell_min, ell_max=self.ell_max,
                               multiplication_truncator=self._truncator)

    @property
    def ndarray(self):
        """the mode weights as a numpy array (device -> host copy)"""
        return to_host(self.buf)

    def __array__(self, dtype=None, copy=None):
        a = self.ndarray
        return a if dtype is None else a.astype(dtype)

    def rows(self, lo, hi):
        """rows [lo, hi) on the host"""
        return to_host(self.buf[lo:hi])

    # ------------------------------------------------------------------ metadata
    time = property(lambda self: self._time)
    u = t = time
    s = property(lambda self: self.spin_weight)
    n_times = property(lambda self: self._time.size)
    shape = property(lambda self: tuple(self.buf.shape))
    LM = property(lambda self: LM_range(self.ell_min, self.ell_max))

    def _like(self, buf, **changes):
        md = dict(time=self._time, spin_weight=self.spin_weight, ell_min=self.ell_min, ell_max=self.ell_max, ctx=self._ctx,
                  multiplication_truncator=self._truncator)
        md.update(changes)
        return DeviceModesTimeSeries(buf, **md)

    # ------------------------------------------------------------------ the one kernel behind the mode-space operators
    def _map(self, n_cols, idx_a, coef_a, conj_a=False, other=None, idx_b=None, coef_b=None, conj_b=False, row_scale=None):
        out = empty(self._ctx, (self.n_times, n_cols))
        rs = None
        if row_scale is not None:
            torch = _torch()
            rs = torch.from_numpy(np.ascontiguousarray(row_scale, dtype=float)).to(self.buf.device)
        idx_a = np.ascontiguousarray(idx_a, dtype=np.int32)
        coef_a = np.ascontiguousarray(coef_a, dtype=np.complex128)
        args_b = (None, 0, None, None, 0)
        if other is not None:
            idx_b = np.ascontiguousarray(idx_b, dtype=np.int32)
            coef_b = np.ascontiguousarray(coef_b, dtype=np.complex128)
            args_b = (c_vp(other.buf.data_ptr()), other.buf.stride(0), idx_b.ctypes.data_as(ctypes.POINTER(ctypes.c_int32)),
                      c_vp(coef_b.ctypes.data), int(bool(conj_b)))
        rc = _lib.load().bms_mode_map(
            self._ctx.handle, c_vp(out.data_ptr()), out.stride(0), self.n_times, int(n_cols),
            c_vp(self.buf.data_ptr()), self.buf.stride(0), idx_a.ctypes.data_as(ctypes.POINTER(ctypes.c_int32)), c_vp(coef_a.ctypes.data),
            int(bool(conj_a)), *args_b, c_vp(rs.data_ptr()) if rs is not None else None, BMS_DEVICE,
        )
        self._ctx.check(rc, "bms_mode_map")
        return out

    def _scaled(self, factor):
        n = self.buf.shape[1]
        return self._like(self._map(n, _identity(n), np.full(n, complex(factor))))

    # ------------------------------------------------------------------ mode-space operators (sf.Modes)
    @property
    def eth(self):
        n = self.buf.shape[1]
        return self._like(self._map(n, _identity(n), _eth_factor(self.ell_min, self.ell_max, self.spin_weight, True)), spin_weight=self.spin_weight + 1)

    @property
    def ethbar(self):
        n = self.buf.shape[1]
        return self._like(self._map(n, _identity(n), _eth_factor(self.ell_min, self.ell_max, self.spin_weight, False)), spin_weight=self.spin_weight - 1)

    @property
    def eth_GHP(self):
        n = self.buf.shape[1]
        f = _eth_factor(self.ell_min, self.ell_max, self.spin_weight, True) / math.sqrt(2)
        return self._like(self._map(n, _identity(n), f), spin_weight=self.spin_weight + 1)

    @property
    def ethbar_GHP(self):
        n = self.buf.shape[1]
        f = _eth_factor(self.ell_min, self.ell_max, self.spin_weight, False) / math.sqrt(2)
        return self._like(self._map(n, _identity(n), f), spin_weight=self.spin_weight - 1)

    @property
    def bar(self):
        perm, sign = _bar_tables(self.ell_min, self.ell_max, self.spin_weight)
        return self._like(self._map(perm.size, perm, sign, conj_a=True), spin_weight=-self.spin_weight)

    def _re_im(self, ca, cb):
        if self.spin_weight != 0:
            raise ValueError("The real / imaginary part of a function with non-zero spin weight is not a spin-weighted function")
        perm, sign = _bar_tables(self.ell_min, self.ell_max, 0)
        n = perm.size
        return self._like(self._map(n, _identity(n), np.full(n, ca), other=self, idx_b=perm, coef_b=cb * sign, conj_b=True))

    @property
    def real(self):
        """(a + bar a) / 2"""
        return self._re_im(0.5 + 0j, 0.5 + 0j)

    @property
    def imag(self):
        """-i (a - bar a) / 2"""
        return self._re_im(-0.5j, 0.5j)

    def truncate_ell(self, new_ell_max):
        if new_ell_max < self.ell_min:
            raise ValueError(f"new ell_max {new_ell_max} is below ell_min {self.ell_min}")
        new_ell_max = min(int(new_ell_max), self.ell_max)
        n = LM_total_size(self.ell_min, new_ell_max)
        return self._like(self._map(n, _identity(n), np.ones(n, dtype=complex)), ell_max=new_ell_max)

    def _combine(self, other, sign):
        if not isinstance(other, DeviceModesTimeSeries):
            if hasattr(other, "spin_weight"):  # a host series: bring it over
                other = DeviceModesTimeSeries.from_host(other, ctx=self._ctx)
            else:
                raise TypeError("only mode series can be added to a device-resident series")
        if other.spin_weight != self.spin_weight:
            raise ValueError(f"Cannot add modes with different spin weights ({self.spin_weight} and {other.spin_weight})")
        if other.n_times != self.n_times:
            raise ValueError("The time series of objects to be added must be the same.")
        lo, hi = min(self.ell_min, other.ell_min), max(self.ell_max, other.ell_max)
        n = LM_total_size(lo, hi)
        out = self._map(n, _embed(self.ell_min, self.ell_max, lo, hi), np.ones(n, dtype=complex), other=other,
                        idx_b=_embed(other.ell_min, other.ell_max, lo, hi), coef_b=np.full(n, complex(sign)))
        return self._like(out, ell_min=lo, ell_max=hi)

    def __add__(self, other):
        return self._combine(other, 1.0)

    __radd__ = __add__

    def __sub__(self, other):
        return self._combine(other, -1.0)

    def __neg__(self):
        return self._scaled(-1.0)

    def __mul__(self, other):
        if isinstance(other, DeviceModesTimeSeries):
            return self.multiply(other)
        if np.ndim(other) == 0:
            return self._scaled(other)
        other = np.asarray(other)
        if other.shape in ((self.n_times, 1), (self.n_times,)) and not np.iscomplexobj(other):  # one real factor per time step
            n = self.buf.shape[1]
            return self._like(self._map(n, _identity(n), np.ones(n, dtype=complex), row_scale=other.reshape(-1)))
        raise TypeError(f"cannot multiply a device-resident series by an array of shape {other.shape}")

    __rmul__ = __mul__

    def __truediv__(self, other):
        if isinstance(other, DeviceModesTimeSeries) or np.ndim(other) != 0:
            raise ValueError("Cannot divide by a mode series")
        return self._scaled(1.0 / other)

    def scale_by_ell(self, f):
        """every mode (l, m) times f(l): the diagonal operators D, D^-1 of map_to_superrest_frame.py:76-105"""
        n = self.buf.shape[1]
        fac = np.array([f(int(l)) for l in _ell_of_columns(self.ell_min, self.ell_max)], dtype=np.complex128)
        return self._like(self._map(n, _identity(n), fac))

    def norm(self):
        """L2 norm over the sphere at every time (sf.Modes.norm), on the host"""
        return np.linalg.norm(self.ndarray, axis=-1)

    # ------------------------------------------------------------------ time calculus (modes_time_series.py:72-126)
    def interpolate(self, new_time, derivative_order=0, out=None):
        new_time = np.ascontiguousarray(new_time, dtype=float)
        if new_time.ndim != 1:
            raise ValueError(f"New time array must have exactly 1 dimension; it has {new_time.ndim}.")
        if derivative_order > 3:
            raise ValueError(f"{type(self)} interpolation uses CubicSpline, and cannot take a derivative of order {derivative_order}")
        if derivative_order < -16:
            raise NotImplementedError("antiderivatives beyond the sixteenth are not provided")
        n_cols = self.buf.shape[1]
        res = empty(self._ctx, (new_time.size, n_cols))
        x = np.ascontiguousarray(self._time)
        rc = _lib.load().bms_spline_derivative(
            self._ctx.handle, _lib.dptr(x), x.size, c_vp(self.buf.data_ptr()), self.buf.stride(0), n_cols, BMS_DEVICE, _lib.dptr(new_time),
            new_time.size, int(derivative_order), c_vp(res.data_ptr()),
        )
        self._ctx.check(rc, "bms_spline_derivative")
        return self._like(res, time=new_time)

    def antiderivative(self, antiderivative_order=1):
        return self.interpolate(self._time, derivative_order=-antiderivative_order)

    def derivative(self, derivative_order=1):
        return self.interpolate(self._time, derivative_order=derivative_order)

    dot = property(lambda self: self.derivative())
    ddot = property(lambda self: self.derivative(2))
    int = property(lambda self: self.antiderivative())
    iint = property(lambda self: self.antiderivative(2))

    # ------------------------------------------------------------------ products (modes_time_series.py:142-202)
    def multiply(self, other, truncator=None):
        if truncator is None:
            truncator = self._truncator
        return self.grid_multiply(other, working_ell_max=self.ell_max + other.ell_max,
                                  output_ell_max=int(truncator((self.ell_max, other.ell_max))))

    def _from_ell_0(self):
        if self.ell_min == 0 and self.buf.is_contiguous():
            return self.buf
        n = (self.ell_max + 1) ** 2
        return self._map(n, _embed(self.ell_min, self.ell_max, 0, self.ell_max), np.ones(n, dtype=complex))

    def grid_multiply(self, mts, **kwargs):
        output_ell_max = kwargs.pop("output_ell_max", self.ell_max)
        working_ell_max = kwargs.pop("working_ell_max", self.ell_max + mts.ell_max)
        if not isinstance(mts, DeviceModesTimeSeries):
            mts = DeviceModesTimeSeries.from_host(mts, ctx=self._ctx)
        if self.n_times != mts.n_times or not np.equal(self.t, mts.t).all():
            raise ValueError("The time series of objects to be multiplied must be the same.")
        a, b = self._from_ell_0(), mts._from_ell_0()
        out = empty(self._ctx, (self.n_times, (output_ell_max + 1) ** 2))
        rc = _lib.load().bms_grid_multiply(
            self._ctx.handle, c_vp(a.data_ptr()), self.spin_weight, self.ell_max, c_vp(b.data_ptr()), mts.spin_weight, mts.ell_max,
            BMS_DEVICE, self.n_times, int(working_ell_max), int(output_ell_max), c_vp(out.data_ptr()),
        )
        self._ctx.check(rc, "bms_grid_multiply")
        return self._like(out, spin_weight=self.spin_weight + mts.spin_weight, ell_min=0, ell_max=int(output_ell_max))
