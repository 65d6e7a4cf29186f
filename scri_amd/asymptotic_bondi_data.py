"""`AsymptoticBondiData`: storage of psi0..psi4 and sigma as mode time series
(scri/asymptotic_bondi_data/__init__.py:9-263) and its BMS ``transform``
(scri/asymptotic_bondi_data/transformations.py:8-431), computed by ``bms_transform_abd`` on the GPU."""
import numpy as np

from . import engine
from .waveform_grid import _parse_translations, _parse_lorentz

_FIELDS = ("psi0", "psi1", "psi2", "psi3", "psi4", "sigma")
_SPINS = (2, 1, 0, -1, -2, 2)


def _process_transformation_kwargs(input_ell_max, **kwargs):
    """scri/asymptotic_bondi_data/transformations.py:8-97: reality of the supertranslation is imposed
    silently; unknown kwargs are ignored (the reference never warns here)."""
    supertranslation, ell_max_supertranslation = _parse_translations(kwargs, impose_reality=True)
    output_ell_max = kwargs.pop("output_ell_max", input_ell_max)
    working_ell_max = kwargs.pop("working_ell_max", 2 * input_ell_max + ell_max_supertranslation)
    if working_ell_max < input_ell_max:
        raise ValueError(f"working_ell_max={working_ell_max} is too small; it must be at least ell_max={input_ell_max}")
    try:
        frame_rotation, boost_velocity = _parse_lorentz(kwargs, single=True)
    except ValueError as e:
        if "boost_velocity" in str(e):
            raise ValueError(str(e).rstrip(".")) from None
        raise
    return frame_rotation, boost_velocity, supertranslation, working_ell_max, output_ell_max


class AsymptoticBondiData:
    """Asymptotic Bondi data on future null infinity: u [n_times] and the six fields
    psi0, psi1, psi2, psi3, psi4, sigma as complex mode weights [n_times, (ell_max+1)^2]
    (l from 0; modes with l < |s| are zero), stored contiguously as `_raw_data[6, n_times, n_modes]`."""

    def __init__(self, time, ell_max, multiplication_truncator=sum, frameType=None, ctx=None, _raw=None, device=False):
        """device=True: the six fields live in HBM (`_raw_dev`, a torch complex128 tensor [6, n_times, n_modes]); `transform`,
        the field operators (DeviceModesTimeSeries), the BMS charges and the frame-fixing iterations then run without the
        fields crossing PCIe.  `_raw_data` of such an object is a host COPY made on access (assign fields through the
        psi0 ... sigma setters or `to_device`)."""
        from . import Inertial

        self._time = np.array(time, dtype=float)
        if self._time.ndim != 1:
            raise ValueError(f"Input `time` parameter must be a 1-d array of floats; it has shape {self._time.shape}")
        self._ell_max = int(ell_max)
        self._truncator = multiplication_truncator
        self._ctx = ctx
        self._device = bool(device)
        shape = (6, self._time.size, (self._ell_max + 1) ** 2)
        if self._device:
            from . import _lib, device_series

            self._ctx = ctx if ctx is not None else _lib.default_context()
            if _raw is None:
                self._raw_dev = device_series.empty(self._ctx, shape).zero_()
            else:
                if tuple(_raw.shape) != shape or not _raw.is_contiguous():
                    raise ValueError(f"device storage of shape {tuple(_raw.shape)} does not fit {shape}")
                self._raw_dev = _raw
        elif _raw is None:
            self._raw_data = np.zeros(shape, dtype=complex)
        else:  # adopt freshly computed storage instead of zero-filling and copying (hundreds of MB per transformation)
            if _raw.shape != shape or _raw.dtype != np.complex128 or not _raw.flags.c_contiguous:
                raise ValueError(f"storage of shape {_raw.shape} does not fit {shape}")
            self._raw_data = _raw
        self.frameType = Inertial if frameType is None else frameType
        self.frame = np.zeros((0, 4))

    def __getattr__(self, name):
        # device-resident objects have no host storage: `_raw_data` is a copy made on demand
        if name == "_raw_data" and self.__dict__.get("_device"):
            from . import device_series

            return device_series.to_host(self._raw_dev)
        raise AttributeError(name)

    @property
    def is_device_resident(self):
        return self._device

    def to_device(self, ctx=None):
        """A device-resident copy of this object (itself if it already is one)."""
        if self._device:
            return self
        from . import _lib, device_series

        ctx = ctx if ctx is not None else (self._ctx if self._ctx is not None else _lib.default_context())
        new = type(self)(self._time, self._ell_max, multiplication_truncator=self._truncator, frameType=self.frameType, ctx=ctx,
                         _raw=device_series.to_device(ctx, self._raw_data), device=True)
        new.frame = self.frame
        return new

    def to_host(self):
        """A host-resident copy (itself if it already is one)."""
        if not self._device:
            return self
        new = type(self)(self._time, self._ell_max, multiplication_truncator=self._truncator, frameType=self.frameType, ctx=self._ctx,
                         _raw=np.ascontiguousarray(self._raw_data))
        new.frame = self.frame
        return new

    @property
    def time(self):
        return self._time

    @time.setter
    def time(self, new_time):
        self._time[:] = new_time

    u = t = time

    @property
    def LM(self):
        from .mode_algebra import LM_range

        return LM_range(0, self._ell_max)

    @property
    def h(self):
        """The strain h = 2 sigma-bar as a WaveformModes object, l >= 2 (scri/asymptotic_bondi_data/__init__.py:119-131)"""
        from . import Inertial, WaveformModes
        from . import h as h_DataType

        return WaveformModes(
            t=self._time.copy(), data=2.0 * self.sigma.bar.ndarray[:, 4:], ell_min=2, ell_max=self._ell_max, frameType=Inertial,
            dataType=h_DataType, r_is_scaled_out=True, m_is_scaled_out=True, ctx=self._ctx,
        )

    @property
    def n_times(self):
        return self._time.size

    @property
    def ell_min(self):
        return 0

    @property
    def ell_max(self):
        return self._ell_max

    @property
    def n_modes(self):
        return (self._ell_max + 1) ** 2

    def copy(self):
        if self._device:
            return type(self)(self._time.copy(), self._ell_max, multiplication_truncator=self._truncator, frameType=self.frameType,
                              ctx=self._ctx, _raw=self._raw_dev.clone(), device=True)
        new = type(self)(self._time.copy(), self._ell_max, multiplication_truncator=self._truncator, frameType=self.frameType, ctx=self._ctx)
        new._raw_data[:] = self._raw_data
        return new

    def __getitem__(self, key):
        """Slice along time (scri/asymptotic_bondi_data/__init__.py: abd[i0:i1])"""
        if isinstance(key, (int, np.integer)):
            key = slice(key, key + 1 if key != -1 else None)  # one time step, kept as a series of length 1
        if not isinstance(key, slice):
            raise ValueError(f"Invalid key `{key}` of type `{type(key)}`.")
        if self._device:
            return type(self)(self._time[key], self._ell_max, multiplication_truncator=self._truncator, frameType=self.frameType,
                              ctx=self._ctx, _raw=self._raw_dev[:, key].contiguous(), device=True)
        new = type(self)(self._time[key], self._ell_max, multiplication_truncator=self._truncator, frameType=self.frameType, ctx=self._ctx)
        new._raw_data[:] = self._raw_data[:, key]
        return new

    def interpolate(self, new_times):
        """scri/asymptotic_bondi_data/__init__.py:218-233: cubic-spline all six fields to `new_times`."""
        new_times = np.asarray(new_times, dtype=float)
        if self._device:
            new = type(self)(new_times, self._ell_max, multiplication_truncator=self._truncator, frameType=self.frameType, ctx=self._ctx,
                             device=True)
            for name in _FIELDS:
                getattr(new, name).buf.copy_(getattr(self, name).interpolate(new_times).buf)
            return new
        new = type(self)(new_times, self._ell_max, multiplication_truncator=self._truncator, frameType=self.frameType, ctx=self._ctx)
        y = np.ascontiguousarray(np.moveaxis(self._raw_data, 0, 1)).reshape(self.n_times, -1)
        out = engine.cubic_spline(self._time, y, new_times, ctx=self._ctx)
        new._raw_data[:] = np.moveaxis(out.reshape(new_times.size, 6, -1), 1, 0)
        return new

    def transform(self, **kwargs):
        """Apply a BMS transformation (scri/asymptotic_bondi_data/transformations.py:199-431).

        Keyword arguments: time_translation, space_translation, spacetime_translation,
        supertranslation, frame_rotation, boost_velocity, output_ell_max, working_ell_max."""
        # scri_amd extension (not in the reference): `group` = a torch.distributed process group over whose ranks the time axis is
        # split; this object then holds THIS rank's contiguous block of rows, and so does the result (scri_amd/sharding.py)
        group = kwargs.pop("group", None)
        devices = kwargs.pop("devices", None)  # the GPUs of this process a long host-memory series is dealt over (engine.transform_abd)
        frame_rotation, boost_velocity, supertranslation, working_ell_max, output_ell_max = _process_transformation_kwargs(
            self.ell_max, **kwargs
        )
        n_theta = 2 * working_ell_max + 1
        tr = engine.make_transformation(supertranslation, frame_rotation, boost_velocity, n_theta, n_theta, output_ell_max)
        if group is not None:
            from . import sharding

            u_global, have = sharding.gather_time_axis(self._time, group, ctx=self._ctx)
            u_new, raw_new, _ = sharding.transform_abd_sharded(self._raw_dev if self._device else self._raw_data, u_global, self.ell_max, tr,
                                                              group=group, have=have, ctx=self._ctx)
            if self._device:
                return type(self)(np.array(u_new), output_ell_max, ctx=self._ctx, _raw=raw_new.contiguous(), device=True)
            return type(self)(u_new, output_ell_max, ctx=self._ctx, _raw=np.ascontiguousarray(raw_new))
        if self._device:
            # HBM to HBM: the output window is known before anything moves, so the result is allocated at its exact size
            from . import device_series

            i_lo, i_hi = engine.output_window(self._time, tr, abd=True, ctx=self._ctx)
            out = device_series.empty(self._ctx, (6, max(i_hi - i_lo, 0), (output_ell_max + 1) ** 2))
            u_new, n_new, _ = engine.transform_abd(self._time, self._raw_dev.data_ptr(), self.ell_max, tr, ctx=self._ctx, device=True,
                                                   out_ptr=out.data_ptr(), shard=(0, self.n_times, i_lo, i_hi))
            if n_new != out.shape[1]:
                out = out[:, :n_new].contiguous()
            return type(self)(np.array(u_new), output_ell_max, ctx=self._ctx, _raw=out, device=True)
        u_new, raw_new = engine.transform_abd(self._time, self._raw_data, self.ell_max, tr, ctx=self._ctx, devices=devices)
        # `type(self)(timeprime, output_ell_max)` in the reference (transformations.py:417): the result starts from the
        # constructor's defaults (multiplication_truncator = sum, frameType = Inertial), whatever the input carried
        return type(self)(u_new, output_ell_max, ctx=self._ctx, _raw=raw_new)


def boosted_grid(frame_rotation, boost_velocity, n_theta, n_phi):
    """Rotors of the grid directions after the rotation and boost, float [n_theta, n_phi, 4]
    (scri/asymptotic_bondi_data/transformations.py:100-148)."""
    from . import quaternions

    return engine.rotor_grid(quaternions.as_float_array(frame_rotation), boost_velocity, n_theta, n_phi)


def conformal_factors(boost_velocity, distorted_grid_rotors):
    """k, eth k / k, 1/k, 1/k^3 on the distorted grid, each [1, n_theta, n_phi] so that they broadcast against time
    (scri/asymptotic_bondi_data/transformations.py:151-196; plain arrays in place of sf.Grid: spin weights 0, 1, 0, 0)."""
    from . import quaternions

    k, ethk_over_k, one_over_k, one_over_k_cubed = engine.conformal_factors(boost_velocity, quaternions.as_float_array(distorted_grid_rotors))
    return k[np.newaxis], ethk_over_k[np.newaxis], one_over_k[np.newaxis], one_over_k_cubed[np.newaxis]


_SPINS = (2, 1, 0, -1, -2, 2)  # psi0..psi4, sigma


def _field_property(i):
    def get(self):
        # a ModesTimeSeries VIEW of the field (scri/asymptotic_bondi_data/__init__.py:117-216): writes go to _raw_data
        if self._device:
            from .device_series import DeviceModesTimeSeries

            return DeviceModesTimeSeries(self._raw_dev[i], self._time, _SPINS[i], 0, self._ell_max, ctx=self._ctx,
                                         multiplication_truncator=self._truncator)
        from .modes_time_series import ModesTimeSeries

        return ModesTimeSeries(self._raw_data[i], self._time, spin_weight=_SPINS[i], ell_min=0, ell_max=self._ell_max,
                               multiplication_truncator=self._truncator)

    def set(self, value):
        if self._device:
            from . import device_series

            if isinstance(value, device_series.DeviceModesTimeSeries):
                self._raw_dev[i].copy_(value.buf)
            else:
                self._raw_dev[i].copy_(device_series.to_device(self._ctx, np.broadcast_to(np.asarray(value), self._raw_dev[i].shape)))
            return
        self._raw_data[i] = value

    return property(get, set)


for _i, _name in enumerate(_FIELDS):
    setattr(AsymptoticBondiData, _name, _field_property(_i))


# BMS charges as methods (scri/asymptotic_bondi_data/__init__.py:235-263 imports them into the class body)
from . import bms_charges as _bms_charges  # noqa: E402

for _f in _bms_charges.METHODS:
    setattr(AsymptoticBondiData, _f.__name__, _f)
AsymptoticBondiData.charge_vector_from_aspect = staticmethod(_bms_charges.charge_vector_from_aspect)

from . import map_to_superrest_frame as _superrest  # noqa: E402

AsymptoticBondiData.map_to_superrest_frame = _superrest.map_to_superrest_frame
from . import map_to_abd_frame as _abd_frame  # noqa: E402

AsymptoticBondiData.map_to_abd_frame = _abd_frame.map_to_abd_frame

from . import abd_ivp as _ivp  # noqa: E402

AsymptoticBondiData.from_initial_values = classmethod(_ivp.from_initial_values)
for _f in _ivp.METHODS:
    setattr(AsymptoticBondiData, _f.__name__, _f)
AsymptoticBondiData.bondi_violations = property(_ivp.bondi_violations)
AsymptoticBondiData.bondi_violation_norms = property(_ivp.bondi_violation_norms)
