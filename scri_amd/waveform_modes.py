"""`WaveformModes`: the container the BMS hot path works on, mirroring the fields and the methods of
``scri.WaveformModes`` that the path touches (scri/waveform_base.py:222-231 fields, :299-367 validity
checks, :440-446 weights, :706-729 history, :950-967 interpolate; scri/waveform_modes.py:705-719
transform).  Everything numerical is delegated to the GPU engine."""
import numpy as np

from . import engine, quaternions
from . import UnknownFrameType, UnknownDataType, SpinWeights, ConformalWeights, RScaling, DataNames, FrameNames
from .mode_algebra import LM_range, LM_total_size
from .mode_operators import ModeOperators

_next_num = [0]


class WaveformModes(ModeOperators):
    """Object containing time, frame, and SWSH-mode data (subset of scri.WaveformModes).

    Parameters (keyword): t, data [n_times, n_modes] complex, ell_min, ell_max, frame (quaternion array or
    float [n,4]; default empty), frameType (default Inertial here only when given), dataType,
    r_is_scaled_out, m_is_scaled_out, history, ctx (engine context; default process-wide).
    """

    def __init__(self, *args, **kwargs):
        self._host = None
        self._dev = None  # device-resident weights (torch tensor used as memory only), see to_device()
        dev_copy = None
        if len(args) == 1 and isinstance(args[0], WaveformModes) and not kwargs:
            o = args[0]
            if o._dev is not None:  # a copy of a device-resident object stays on the device (and does not pull the source off it)
                dev_copy = o._dev.clone()
            kwargs = dict(
                t=o.t.copy(), data=(np.empty((0, 0)) if dev_copy is not None else o.data.copy()), ell_min=o.ell_min, ell_max=o.ell_max,
                frame=np.array(o.frame, copy=True),
                frameType=o.frameType, dataType=o.dataType, r_is_scaled_out=o.r_is_scaled_out,
                m_is_scaled_out=o.m_is_scaled_out, history=list(o.history), ctx=o._ctx,
            )
        elif args:
            raise ValueError("WaveformModes takes keyword arguments (or one WaveformModes to copy)")
        self._ctx = kwargs.pop("ctx", None)
        self.t = np.array(kwargs.pop("t", np.empty((0,))), dtype=float)
        self._frame_template = kwargs.get("frame", None)
        frame = kwargs.pop("frame", None)
        self.frame = np.zeros((0, 4)) if frame is None or np.size(frame) == 0 else np.atleast_2d(quaternions.as_float_array(frame)).copy()
        self.data = np.array(kwargs.pop("data", np.empty((0, 0))), dtype=complex)
        if dev_copy is not None:
            self._host, self._dev = None, dev_copy
        self.ell_min = int(kwargs.pop("ell_min", 0))
        self.ell_max = int(kwargs.pop("ell_max", -1))
        self.frameType = int(kwargs.pop("frameType", UnknownFrameType))
        self.dataType = int(kwargs.pop("dataType", UnknownDataType))
        self.r_is_scaled_out = bool(kwargs.pop("r_is_scaled_out", False))
        self.m_is_scaled_out = bool(kwargs.pop("m_is_scaled_out", False))
        self.history = list(kwargs.pop("history", []))
        constructor_statement = kwargs.pop("constructor_statement", None)
        if kwargs:
            import pprint, warnings

            warnings.warn("\nUnused kwargs passed to this function:\n{}".format(pprint.pformat(kwargs, width=1)))
        _next_num[0] += 1
        self.num = _next_num[0]
        self.history.append(f"{self} = {constructor_statement or 'WaveformModes(...)'}")
        shape = self._data_shape()
        if self.ell_max >= self.ell_min and len(shape) >= 2 and shape[1] != LM_total_size(self.ell_min, self.ell_max):
            raise ValueError(
                f"data.shape[1]={shape[1]} inconsistent with ell_min={self.ell_min}, ell_max={self.ell_max}"
            )

    # ---- where the mode weights live.  `data` is the host array of the reference's interface; an object moved to the GPU with
    # to_device() keeps its weights in HBM, and transform / rotate_decomposition_basis / copy work on them there (results
    # stay there too).  Reading `data` brings them back and makes the host array authoritative again (the caller may write to it).
    @property
    def data(self):
        if self._host is None and self._dev is not None:
            self._host, self._dev = self._dev.cpu().numpy(), None
        return self._host

    @data.setter
    def data(self, value):
        self._host, self._dev = value, None

    def _data_shape(self):
        return tuple(self._dev.shape) if self._dev is not None else np.shape(self._host)

    @property
    def is_device_resident(self):
        return self._dev is not None

    def to_device(self):
        """Move the mode weights to the GPU (in place; returns self).  Chains of rotations and BMS transformations then run
        without crossing PCIe; `to_host()` or reading `.data` brings the result back."""
        from . import device_series, _lib

        if self._dev is None:
            if self._host is None or self._host.ndim != 2:
                raise ValueError("to_device() takes data of shape [n_times, n_modes]")
            ctx = self._ctx if self._ctx is not None else _lib.default_context()
            self._ctx = ctx
            self._dev, self._host = device_series.to_device(ctx, self._host), None
        return self

    def to_host(self):
        _ = self.data
        return self

    def __str__(self):
        return f"{type(self).__name__}_{self.num}"

    # ---- sizes / weights (waveform_base.py:430-446)
    @property
    def n_times(self):
        return self.t.shape[0]

    @property
    def n_modes(self):
        return self._data_shape()[1]

    @property
    def LM(self):
        return LM_range(self.ell_min, self.ell_max)

    @property
    def spin_weight(self):
        return SpinWeights[self.dataType]

    @property
    def conformal_weight(self):
        return ConformalWeights[self.dataType] + (-RScaling[self.dataType] if self.r_is_scaled_out else 0)

    @property
    def data_type_string(self):
        return DataNames[self.dataType]

    @property
    def frame_type_string(self):
        return FrameNames[self.frameType]

    def _append_history(self, s):
        self.history.append(s if isinstance(s, str) else "\n".join(s))

    # ---- validity (core checks of waveform_base.py:299-367)
    def ensure_validity(self, alter=True, assertions=False):
        errors = []
        if not (self.t.ndim == 1 and self.t.dtype == np.dtype(float)):
            errors.append("t must be a 1-d float array")
        if self.t.size > 1 and not np.all(np.diff(self.t) > 0):
            errors.append("t must be strictly increasing")
        if not np.all(np.isfinite(self.t)):
            errors.append("t must be finite")
        if self.frame.size and self.frame.shape[0] not in (1, self.n_times):
            errors.append("frame must have 0, 1 or n_times elements")
        if not np.all(np.isfinite(self.frame)):
            errors.append("frame must be finite")
        if self.data.shape[0] != self.n_times:
            errors.append("data.shape[0] != n_times")
        if not np.all(np.isfinite(self.data.view(float))):
            errors.append("data must be finite")
        if self.data.dtype != np.dtype(complex):
            errors.append("data must be complex")
        if self.data.shape[1] != LM_total_size(self.ell_min, self.ell_max):
            errors.append("data.shape[1] inconsistent with ell range")
        if errors:
            if assertions:
                raise AssertionError("; ".join(errors))
            print("The following conditions were found to be incorrectly False:\n\t" + "\n\t".join(errors))
            return False
        return True

    def copy(self):
        return type(self)(self)

    # ---- container conveniences (waveform_modes.py:385-455, waveform_base.py:520-575, :917-948, waveform_modes.py:953-1021)
    @property
    def ells(self):
        """(ell_min, ell_max)"""
        return self.ell_min, self.ell_max

    @ells.setter
    def ells(self, new_ells):
        self.ell_min, self.ell_max = int(new_ells[0]), int(new_ells[1])
        if self.n_modes != LM_total_size(self.ell_min, self.ell_max):
            import warnings

            warnings.warn(
                f"\nWaveform's data.shape={self.data.shape} does not agree with "
                f"(ell_min,ell_max)=({self.ell_min},{self.ell_max}).\n"
                "Hopefully you are about to reset `data`.  To avoid this warning,\n"
                "reset `data` before resetting ell_min and/or ell_max."
            )

    def index(self, ell, m):
        """Index of the (ell, m) mode in the data"""
        return ell * (ell + 1) - self.ell_min**2 + m

    def indices(self, ell_m):
        """Indices of the given (ell, m) modes, an N x 2 integer array, in the data"""
        ell_m = np.asarray(ell_m)
        if not (ell_m.dtype.kind == "i" and ell_m.ndim == 2 and ell_m.shape[1] == 2):
            raise ValueError("Input `ell_m` should be an Nx2 sequence of integers")
        return ell_m[:, 0] * (ell_m[:, 0] + 1) - self.ell_min**2 + ell_m[:, 1]

    @property
    def n_data_sets(self):
        return int(np.prod(self.data.shape[1:]))

    @property
    def data_2d(self):
        return self.data.reshape((self.n_times, self.n_data_sets))

    @property
    def abs(self):
        return np.abs(self.data)

    @property
    def arg(self):
        return np.angle(self.data)

    @property
    def arg_unwrapped(self):
        return np.unwrap(np.angle(self.data), axis=0)

    def norm(self, take_sqrt=False, indices=slice(None, None, None)):
        """L2 norm of the waveform at each time: sum over modes of |data|^2 (its square root with take_sqrt), optionally on
        a slice of the times"""
        # the reference adds re^2 + im^2 column by column (complex_array_norm / complex_array_abs, waveform_base.py:19-35); the same
        # order here, so the sums agree to the bit.  Weights resident on the GPU are reduced there (bms_row_norm); a host array is
        # reduced where it is -- one pass over it, against an upload of the whole series.
        if self.is_device_resident and indices == slice(None, None, None) and len(self._data_shape()) == 2:
            return engine.row_norm(None, take_sqrt=take_sqrt, ctx=self._ctx, device_tensor=self._dev)
        d = self.data_2d[indices]
        n = np.zeros(d.shape[0], dtype=float)
        for j in range(d.shape[1]):
            n += d[:, j].real * d[:, j].real + d[:, j].imag * d[:, j].imag
        return np.sqrt(n) if take_sqrt else n

    def max_norm_index(self, skip_fraction_of_data=4):
        """Index of the time step with the largest norm, searching only the last (1 - 1/skip) of the data (0 or 1: all)"""
        if skip_fraction_of_data == 0 or skip_fraction_of_data == 1:
            return int(np.argmax(self.norm()))
        first = self.n_times // skip_fraction_of_data
        return int(np.argmax(self.norm(indices=slice(first, None, None)))) + first

    def max_norm_time(self, skip_fraction_of_data=4):
        return self.t[self.max_norm_index(skip_fraction_of_data=skip_fraction_of_data)]

    def __getitem__(self, key):
        """w[times], w[times, ells]: the second index selects a contiguous range of ell values (or one ell), not data
        columns -- scri/waveform_modes.py:953-1021."""
        if isinstance(key, tuple) and len(key) == 1:
            key = key[0]
        new_ells = self.ells
        if isinstance(key, tuple) and len(key) == 2:
            k1 = key[1]
            if isinstance(k1, (int, np.integer)):
                if k1 < self.ell_min or k1 > self.ell_max:
                    raise ValueError(
                        "Requested ell value {} lies outside ".format(k1)
                        + f"WaveformModes object's ell range ({self.ell_min},{self.ell_max})."
                    )
                new_ells = (int(k1), int(k1))
            elif isinstance(k1, slice):
                if k1.step and k1.step != 1:
                    raise ValueError("Can only slice WaveformModes over contiguous ell values (step={})".format(k1.step))
                if not k1.start and k1.stop == 0:
                    new_ells = (0, -1)
                else:
                    lo = self.ell_min if not k1.start else k1.start
                    hi = self.ell_max if not k1.stop else k1.stop - 1
                    if lo < self.ell_min or hi > self.ell_max:
                        raise ValueError(
                            f"Requested ell range [{lo},{hi}] lies outside "
                            + f"WaveformBase's ell range [{self.ell_min},{self.ell_max}]."
                        )
                    new_ells = (lo, hi)
            else:
                raise ValueError("Don't know what to do with slice of type `{}`".format(type(k1)))
            if new_ells == (0, -1):
                cols = slice(0)
            else:
                cols = slice(new_ells[0] ** 2 - self.ell_min**2, new_ells[1] * (new_ells[1] + 2) + 1 - self.ell_min**2)
            tkey = key[0]
        elif isinstance(key, (slice, int, np.integer)):
            cols, tkey = slice(None), key
        else:
            raise ValueError("Could not understand input `{}` (of type `{}`) ".format(key, type(key)))
        if isinstance(tkey, (int, np.integer)):
            tkey = slice(tkey, tkey + 1 if tkey != -1 else None)
        W = type(self)(
            t=self.t[tkey], data=self.data[tkey, cols], ell_min=new_ells[0], ell_max=new_ells[1],
            frame=self.frame[tkey] if self.frame.shape[0] == self.n_times else self.frame, frameType=self.frameType,
            dataType=self.dataType, r_is_scaled_out=self.r_is_scaled_out, m_is_scaled_out=self.m_is_scaled_out,
            history=list(self.history), ctx=self._ctx,
        )
        W.history.pop()  # the constructor's line, replaced by the slicing statement
        W._append_history(f"{W} = {self}[{key}]")
        return W

    def copy_without_data(self):
        W = type(self)(
            t=np.empty((0,)), data=np.empty((0, self.n_modes), dtype=complex), ell_min=self.ell_min, ell_max=self.ell_max,
            frameType=self.frameType, dataType=self.dataType, r_is_scaled_out=self.r_is_scaled_out,
            m_is_scaled_out=self.m_is_scaled_out, history=list(self.history), ctx=self._ctx,
        )
        return W

    # ---- interpolation in time (waveform_base.py:950-967): not-a-knot cubic spline on the GPU
    def interpolate(self, tprime):
        tprime = np.asarray(tprime, dtype=float)
        W = self.copy_without_data()
        W.t = np.copy(tprime)
        if self.frame.shape[0] > 1:  # waveform_base.py:957
            W.frame = quaternions.squad(self.frame, self.t, tprime)
        else:
            W.frame = np.array(self.frame, copy=True)
        W.data = engine.cubic_spline(self.t, self.data, tprime, ctx=self._ctx)
        W._append_history(f"{W} = {self}.interpolate({tprime})")
        return W

    # time calculus of the mode data (scri/waveform_base.py:689-703): cubic-spline derivatives / antiderivatives on the GPU
    @property
    def data_dot(self):
        return engine.spline_derivative(self.t, self.data, self.t, 1, ctx=self._ctx)

    @property
    def data_ddot(self):
        return engine.spline_derivative(self.t, self.data, self.t, 2, ctx=self._ctx)

    @property
    def data_int(self):
        return engine.spline_derivative(self.t, self.data, self.t, -1, ctx=self._ctx)

    @property
    def data_iint(self):
        return engine.spline_derivative(self.t, self.data, self.t, -2, ctx=self._ctx)

    def LdtVector(self):
        from . import mode_calculations

        return mode_calculations.LdtVector(self)

    def LLMatrix(self):
        from . import mode_calculations

        return mode_calculations.LLMatrix(self)

    def LLDominantEigenvector(self, RoughDirection=np.array([0.0, 0.0, 1.0]), RoughDirectionIndex=0):
        from . import mode_calculations

        return mode_calculations.LLDominantEigenvector(self, RoughDirection, RoughDirectionIndex)

    def angular_velocity(self, include_frame_velocity=False):
        from . import mode_calculations

        return mode_calculations.angular_velocity(self, include_frame_velocity)

    def _allclose(self, other, rtol=1e-05, atol=1e-08):
        return (
            np.allclose(self.t, other.t, rtol=rtol, atol=atol)
            and np.allclose(self.data, other.data, rtol=rtol, atol=atol)
            and self.ell_min == other.ell_min
            and self.ell_max == other.ell_max
            and self.dataType == other.dataType
            and self.frameType == other.frameType
        )

    # ---- BMS transformation (scri/waveform_modes.py:705-719 -> WaveformGrid.transform)
    def transform(self, **kwargs):
        """Transform modes by some BMS transformation (time/space/spacetime translation,
        supertranslation, frame_rotation, boost_velocity, n_theta, n_phi, ell_max, psi*_modes);
        see scri.WaveformGrid.from_modes for the meaning of the parameters."""
        from .waveform_grid import transform

        return transform(self, **kwargs)
