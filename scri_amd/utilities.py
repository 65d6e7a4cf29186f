"""Bit transforms of the storage formats (scri/utilities.py:194-406): xor_timeseries, xor_timeseries_reverse,
fletcher32, multishuffle -- same names and calling conventions, executed on the GPU through the C ABI, bit-exact."""
import ctypes
import functools

import numpy as np

from . import _lib
from ._lib import BMS_HOST, default_context


def _ctx(ctx):
    return ctx if ctx is not None else default_context()


def _xor(c, reverse, ctx):
    ctx = _ctx(ctx)
    c = np.asarray(c)
    if not c.flags.c_contiguous or c.itemsize * int(np.prod(c.shape[1:], dtype=np.int64)) % 8:
        raise ValueError("xor_timeseries needs a C-contiguous array whose rows are a whole number of 64-bit words")
    n_rows = c.shape[0] if c.ndim else 0
    words = c.itemsize * int(np.prod(c.shape[1:], dtype=np.int64)) // 8
    rc = _lib.load().bms_xor_timeseries(ctx.handle, ctypes.c_void_p(c.ctypes.data), BMS_HOST, n_rows, words, int(reverse))
    ctx.check(rc, "bms_xor_timeseries")
    return c


def xor_timeseries(c, ctx=None):
    """XOR a time series in place (time along the first axis): each time step keeps only the bits that changed."""
    return _xor(c, False, ctx)


def xor_timeseries_reverse(c, ctx=None):
    """Undo xor_timeseries, bit for bit."""
    return _xor(c, True, ctx)


def fletcher32(data, ctx=None):
    """Fletcher-32 checksum of an array viewed as uint16 (blocks of 360, modulus 65535)."""
    ctx = _ctx(ctx)
    d = np.ascontiguousarray(data).reshape(-1).view(np.uint16)
    out = ctypes.c_uint32(0)
    rc = _lib.load().bms_fletcher32(ctx.handle, ctypes.c_void_p(d.ctypes.data), BMS_HOST, d.nbytes, ctypes.byref(out))
    ctx.check(rc, "bms_fletcher32")
    return np.uint32(out.value)


@functools.lru_cache()
def multishuffle(shuffle_widths, forward=True, ctx=None):
    """Function that "multi-shuffles" (forward) or un-shuffles a flat array: shuffle_widths lists the bits of each piece
    from the highest significance down and must sum to 8, 16, 32 or 64."""
    widths = [int(w) for w in shuffle_widths]
    bit_width = int(np.sum(widths, dtype=np.int64))
    if bit_width not in [8, 16, 32, 64]:
        raise ValueError(f"Total bit width must be one of [8, 16, 32, 64], not {bit_width}")
    dtype = np.dtype(f"u{bit_width // 8}")
    warr = (ctypes.c_int * len(widths))(*widths)

    def shuffle(a):
        a = np.ascontiguousarray(a).view(dtype)
        if a.ndim != 1:
            raise ValueError(
                "\nThis function only accepts flat arrays.  Make sure you flatten "
                "(using ravel, reshape, or flatten)\n in a way that keeps your data"
                "contiguous in the order you want."
            )
        b = np.zeros_like(a)
        c = _ctx(ctx)
        rc = _lib.load().bms_multishuffle(
            c.handle, ctypes.c_void_p(a.ctypes.data), ctypes.c_void_p(b.ctypes.data), BMS_HOST, a.size, warr, len(widths), int(forward)
        )
        c.check(rc, "bms_multishuffle")
        return b

    return shuffle


def transition_function(x, x0, x1, y0=0.0, y1=1.0, return_indices=False):
    """Smooth (C-infinity) step from y0 (x <= x0) to y1 (x >= x1): y0 + (y1 - y0) / (1 + exp(1/tau - 1/(1 - tau))),
    tau = (x - x0) / (x1 - x0)  (scri/utilities.py:12-58)."""
    x = np.asarray(x, dtype=float)
    out = np.empty_like(x)
    i0 = int(np.searchsorted(x, x0, side="right"))
    i1 = int(np.searchsorted(x, x1, side="left"))
    out[:i0] = y0
    out[i1:] = y1
    tau = (x[i0:i1] - x0) / (x1 - x0)
    with np.errstate(over="ignore", divide="ignore"):
        exponent = 1.0 / tau - 1.0 / (1.0 - tau)
        out[i0:i1] = np.where(exponent >= np.log(np.finfo(float).max), y0, y0 + (y1 - y0) / (1.0 + np.exp(exponent)))
    return (out, i0, i1) if return_indices else out


_MAXEXP = np.log(np.finfo(float).max)


def transition_function_derivative(x, x0, x1, y0=0.0, y1=1.0):
    """d/dx of transition_function with the same parameters (scri/utilities.py:60-102); zero outside (x0, x1)."""
    x = np.asarray(x, dtype=float)
    out = np.zeros_like(x)
    i0 = int(np.searchsorted(x, x0, side="right"))
    i1 = int(np.searchsorted(x, x1, side="left"))
    tau = (x[i0:i1] - x0) / (x1 - x0)
    with np.errstate(over="ignore", divide="ignore", invalid="ignore"):
        exponent = 1.0 / tau - 1.0 / (1.0 - tau)
        e = np.exp(exponent)
        slope = -(y1 - y0) * e * (-1.0 / tau**2 - 1.0 / (1.0 - tau) ** 2) * (1 / (x1 - x0)) / (1.0 + e) ** 2
        out[i0:i1] = np.where(exponent >= _MAXEXP, 0.0, slope)
    return out


def bump_function(x, x0, x1, x2, x3, y0=0.0, y12=1.0, y3=0.0):
    """Smooth bump: y0 up to x0, y12 on [x1, x2], y3 from x3, with the transition_function profile on either flank
    (scri/utilities.py:105-158)."""
    x = np.asarray(x, dtype=float)
    out = np.empty_like(x)
    i1 = int(np.searchsorted(x, x1, side="left"))
    i2 = max(i1, int(np.searchsorted(x, x2, side="right")))
    out[:i1] = transition_function(x[:i1], x0, x1, y0, y12)
    out[i1:i2] = y12
    out[i2:] = transition_function(x[i2:], x2, x3, y12, y3)
    return out


def transition_to_constant(f, t, t1, t2):
    """f up to t1, a constant from t2, smooth in between: f times the falling transition minus the running integral of f times the
    transition's derivative (scri/utilities.py:161-190; the integral is numpy-quaternion's `indefinite_integral`, the antiderivative of
    the not-a-knot cubic spline through the samples, here scipy's CubicSpline)."""
    from scipy.interpolate import CubicSpline

    f, t = np.asarray(f), np.asarray(t, dtype=float)
    transition, i1, i2 = transition_function(t, t1, t2, y0=1.0, y1=0.0, return_indices=True)
    transition_dot = transition_function_derivative(t, t1, t2, y0=1.0, y1=0.0)
    out = f * transition
    if i2 - i1 >= 4:
        spline = CubicSpline(t[i1:i2], f[i1:i2] * transition_dot[i1:i2])
        out[i1:i2] -= spline.antiderivative()(t[i1:i2])
    elif i2 > i1:
        raise ValueError(f"transition_to_constant needs at least four samples inside ({t1}, {t2}); there are {i2 - i1}")
    out[i2:] = out[i2 - 1]
    return out
