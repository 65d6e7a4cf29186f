"""BMS transformation of WaveformModes: host-side mirror of ``process_transformation_kwargs``
(scri/waveform_grid.py:20-127) and of ``WaveformGrid.transform`` (scri/waveform_grid.py:615-630).

Only the parsing/validation of the keyword arguments happens here (same names, precedence, error
types and messages as the reference); the rotor grid, the SWSH synthesis, the conformal factor and
inhomogeneous terms, the per-pixel cubic-spline interpolation in retarded time and the SWSH analysis
(scri/waveform_grid.py:130-174, 470-613, 274-329) run on the GPU behind ``bms_transform_modes``.
"""
import math
import pprint
import warnings

import numpy as np
from scipy.special import comb

from . import engine, quaternions
from . import Inertial, DataNames, h, sigma, psi0, psi1, psi2, psi3, psi4, hdot, news
from .mode_algebra import (
    LM_index,
    constant_as_ell_0_mode,
    constant_from_ell_0_mode,
    vector_as_ell_1_modes,
    vector_from_ell_1_modes,
)


def _parse_translations(kwargs, impose_reality):
    """Supertranslation / translation arguments, identical in the two flavours except that the
    WaveformModes flavour *checks* reality (waveform_grid.py:46-58) while the ABD flavour *imposes* it
    (asymptotic_bondi_data/transformations.py:36-44) -- and that two of the AsymptoticBondiData flavour's messages are worded
    differently (`impose_reality` marks the flavour; tests/golden/g22 holds both sets word for word)."""
    supertranslation = np.zeros((4,), dtype=complex)
    ell_max_supertranslation = 1
    if "supertranslation" in kwargs:
        supertranslation = np.array(kwargs.pop("supertranslation"), dtype=complex)
        if supertranslation.dtype != "complex" and supertranslation.size > 0:
            raise TypeError(
                "\nInput argument `supertranslation` should be a complex array with size>0.\n"
                "Got a {} array of shape {}.".format(supertranslation.dtype, supertranslation.shape)
            )
        if supertranslation.size <= 4:
            supertranslation = np.pad(
                supertranslation, (0, 4 - supertranslation.size), "constant", constant_values=(0.0,)
            )
        ell_max_supertranslation = int(np.sqrt(len(supertranslation))) - 1
        if (ell_max_supertranslation + 1) ** 2 != len(supertranslation) and impose_reality:
            # (transformations.py:29-35: a plain string -- the length is not filled in -- and the source's indentation in it)
            pad = "\n" + " " * 11
            raise ValueError(
                "Input supertranslation parameter must contain modes from ell=0 up to some ell_max, including" + pad
                + "all relevant m modes in standard order (see `spherical_functions` documentation for details)." + pad
                + "Thus, it must be an array with length given by a perfect square; its length is {len(supertranslation)}"
            )
        if (ell_max_supertranslation + 1) ** 2 != len(supertranslation):
            raise ValueError(
                "\nInput supertranslation parameter must contain modes from ell=0 up to some ell_max, "
                "including\nall relevant m modes in standard order (see `spherical_functions` "
                "documentation for details).\nThus, it must be an array with length given by a "
                "perfect square; its length is {}".format(len(supertranslation))
            )
        for ell in range(ell_max_supertranslation + 1):
            for m in range(ell + 1):
                i_pos = LM_index(ell, m, 0)
                i_neg = LM_index(ell, -m, 0)
                a = supertranslation[i_pos]
                b = supertranslation[i_neg]
                if impose_reality:
                    supertranslation[i_pos] = (a + (-1.0) ** m * b.conjugate()) / 2.0
                    supertranslation[i_neg] = (-1.0) ** m * supertranslation[i_pos].conjugate()
                elif abs(a - (-1.0) ** m * b.conjugate()) > 3e-16 + 1e-15 * abs(b):
                    raise ValueError(
                        f"\nsupertranslation[{i_pos}]={a}  # (ell,m)=({ell},{m})\n"
                        + "supertranslation[{}]={}  # (ell,m)=({},{})\n".format(i_neg, b, ell, -m)
                        + "Will result in an imaginary supertranslation."
                    )
    spacetime_translation = np.zeros((4,), dtype=float)
    spacetime_translation[0] = constant_from_ell_0_mode(supertranslation[0]).real
    spacetime_translation[1:4] = -vector_from_ell_1_modes(supertranslation[1:4]).real
    if "spacetime_translation" in kwargs:
        st_trans = np.array(kwargs.pop("spacetime_translation"), dtype=float)
        if st_trans.shape != (4,) or st_trans.dtype != "float":
            raise TypeError(
                "\nInput argument `spacetime_translation` should be a float array of shape (4,).\n"
                "Got a {} array of shape {}.".format(st_trans.dtype, st_trans.shape)
            )
        spacetime_translation = st_trans[:]
        supertranslation[0] = constant_as_ell_0_mode(spacetime_translation[0])
        supertranslation[1:4] = vector_as_ell_1_modes(-spacetime_translation[1:4])
    if "space_translation" in kwargs:
        s_trans = np.array(kwargs.pop("space_translation"), dtype=float)
        if s_trans.shape != (3,) or s_trans.dtype != "float":
            raise TypeError(
                "\nInput argument `space_translation` should be an array of floats of shape (3,).\n"
                "Got a {} array of shape {}.".format(s_trans.dtype, s_trans.shape)
            )
        spacetime_translation[1:4] = s_trans[:]
        supertranslation[1:4] = vector_as_ell_1_modes(-spacetime_translation[1:4])
    if "time_translation" in kwargs:
        t_trans = kwargs.pop("time_translation")
        if not isinstance(t_trans, float) and impose_reality:  # (transformations.py:71)
            raise TypeError("Input argument `time_translation` should be a single float.  " f"Got {t_trans}")
        if not isinstance(t_trans, float):
            raise TypeError("\nInput argument `time_translation` should be a single float.\n" "Got {}.".format(t_trans))
        spacetime_translation[0] = t_trans
        supertranslation[0] = constant_as_ell_0_mode(spacetime_translation[0])
    return supertranslation, ell_max_supertranslation


def _parse_lorentz(kwargs, single=False):
    fr = kwargs.pop("frame_rotation", [1, 0, 0, 0])
    if not quaternions.is_quaternion_object(fr) and np.ndim(fr) == 1 and np.size(fr) in (1, 3):
        # np.quaternion(*components), as the reference builds it (waveform_grid.py:112): one number is the scalar part, three are the
        # vector part of a pure quaternion
        fr = [float(np.asarray(fr)[0]), 0.0, 0.0, 0.0] if np.size(fr) == 1 else [0.0, *np.asarray(fr, dtype=float)]
    frame_rotation = np.array(quaternions.as_float_array(fr), dtype=float).reshape(-1)[:4]
    norm = math.sqrt(float(np.sum(frame_rotation**2)))
    if norm < 3e-16:
        raise ValueError(
            f"frame_rotation={frame_rotation} should be a " + ("single unit quaternion" if single else "unit quaternion")
        )
    frame_rotation = frame_rotation / norm
    boost_velocity = np.array(kwargs.pop("boost_velocity", [0.0] * 3), dtype=float)
    beta = np.linalg.norm(boost_velocity)
    if boost_velocity.shape != (3,) or beta >= 1.0:
        raise ValueError(
            "Input boost_velocity=`{}` should be a 3-vector with "
            "magnitude strictly less than 1.0.".format(boost_velocity)
        )
    return frame_rotation, boost_velocity


def process_transformation_kwargs(ell_max, **kwargs):
    """scri/waveform_grid.py:20-127 (everything up to the rotor grid, which the engine builds).

    Returns (supertranslation, ell_max_supertranslation, ell_max, n_theta, n_phi, frame_rotation,
    boost_velocity, kwargs)."""
    supertranslation, ell_max_supertranslation = _parse_translations(kwargs, impose_reality=False)
    w_ell_max = ell_max
    ell_max = w_ell_max + ell_max_supertranslation
    n_theta = kwargs.pop("n_theta", 2 * ell_max + 1)
    n_phi = kwargs.pop("n_phi", 2 * ell_max + 1)
    if n_theta < 2 * ell_max + 1 and abs(supertranslation[1:]).max() > 0.0:
        warnings.warn(
            f"n_theta={n_theta} is small; because of the supertranslation, "
            + f"it will lose accuracy for anything less than 2*ell+1={ell_max}"
        )
    if n_theta < 2 * w_ell_max + 1:
        raise ValueError(f"n_theta={n_theta} is too small; " + "must be at least 2*ell+1={}".format(2 * w_ell_max + 1))
    if n_phi < 2 * ell_max + 1 and abs(supertranslation[1:]).max() > 0.0:
        warnings.warn(
            f"n_phi={n_phi} is small; because of the supertranslation, "
            + f"it will lose accuracy for anything less than 2*ell+1={ell_max}"
        )
    if n_phi < 2 * w_ell_max + 1:
        raise ValueError(f"n_phi={n_phi} is too small; " + "must be at least 2*ell+1={}".format(2 * w_ell_max + 1))
    frame_rotation, boost_velocity = _parse_lorentz(kwargs)
    return supertranslation, ell_max_supertranslation, ell_max, n_theta, n_phi, frame_rotation, boost_velocity, kwargs


def _prepare(w_modes, kwargs):
    """Everything from_modes does before touching the data (scri/waveform_grid.py:417-557): frame check, kwargs, the
    type-specific term and the auxiliary Weyl scalars it needs.  Consumes the keys it understands from `kwargs`."""
    if w_modes.frameType != Inertial:
        raise ValueError(
            "\nInput waveform object must be in an inertial frame; "
            "this is in a frame of type `{}`".format(w_modes.frame_type_string)
        )
    (supertranslation, ell_max_supertranslation, ell_max, n_theta, n_phi, frame_rotation, boost_velocity, rest) = (
        process_transformation_kwargs(w_modes.ell_max, **kwargs)
    )
    kwargs.clear()
    kwargs.update(rest)
    beta = np.linalg.norm(boost_velocity)
    dt = w_modes.dataType

    # type-specific term (waveform_grid.py:485-557)
    type_term = engine.BMS_TERM_NONE
    aux = []
    if beta != 0 or (supertranslation[1:] != 0).any():
        if dt == h:
            type_term = engine.BMS_TERM_H
        elif dt == sigma:
            type_term = engine.BMS_TERM_SIGMA
        elif dt in [psi0, psi1, psi2, psi3]:
            type_term = engine.BMS_TERM_PSI
            for DT in range(dt + 1, psi4 + 1):
                try:
                    w_tmp = kwargs.pop("psi{}_modes".format(DataNames[DT][-1]))
                except KeyError:
                    raise ValueError(
                        "\nA BMS transformation of {} requires information from {}, which "
                        "has not been supplied.".format(w_modes.data_type_string, DataNames[DT])
                    )
                if w_tmp.n_times != w_modes.n_times:
                    raise ValueError("auxiliary waveform has a different number of time steps")
                aux.append((w_tmp.data, w_tmp.ell_min, w_tmp.ell_max, w_tmp.spin_weight, comb(5 - dt, 5 - DT), DT - dt))
        elif dt not in [psi4, hdot, news]:
            warnings.warn(
                "\nNo BMS transformation is implemented for waveform objects "
                "of dataType '{}'. Proceeding with the transformation as if it "
                "were dataType 'Psi4'.".format(w_modes.data_type_string)
            )
    return supertranslation, frame_rotation, boost_velocity, n_theta, n_phi, type_term, aux


def transform(w_modes, **kwargs):
    """WaveformGrid.transform (scri/waveform_grid.py:615-630): from_modes(...).to_modes(ell_max), fused in one engine call.

    Returns a new WaveformModes on the retarded-time slices of the transformed frame; `ell_max` of the
    output defaults to that of the input, the output `ell_min` is |s| (to_modes default)."""
    from .waveform_modes import WaveformModes

    if not isinstance(w_modes, WaveformModes):
        raise TypeError(
            "Expected WaveformModes object in argument 1; " "got `{}` instead.".format(type(w_modes).__name__)
        )
    ell_max_out = kwargs.pop("ell_max", w_modes.ell_max)
    # scri_amd extension (not in the reference): `group` = a torch.distributed process group over whose ranks the time axis of the
    # series is split -- `w_modes` then holds THIS rank's contiguous block of rows, and so does the result (scri_amd/sharding.py)
    group = kwargs.pop("group", None)
    partition = kwargs.pop("partition", "auto")
    overlap_halo = kwargs.pop("overlap_halo", False)
    # ... and `devices` = the GPUs of THIS process a long host-memory series is dealt over (engine.transform_modes; default SCRI_AMD_DEVICES)
    devices = kwargs.pop("devices", None)
    original_kwargs = kwargs.copy()
    supertranslation, frame_rotation, boost_velocity, n_theta, n_phi, type_term, aux = _prepare(w_modes, kwargs)
    s = w_modes.spin_weight

    # to_modes argument checks (waveform_grid.py:291-297)
    import numbers

    if not isinstance(ell_max_out, numbers.Integral) or ell_max_out < 0:
        raise ValueError(f"Input `ell_max` should be a nonnegative integer; got `{ell_max_out}`.")
    ell_min_out = abs(s)
    if ell_min_out > ell_max_out:
        raise ValueError(f"Input `ell_min` should be an integer between 0 and {ell_max_out}; got `{ell_min_out}`.")

    tr = engine.make_transformation(supertranslation, frame_rotation, boost_velocity, n_theta, n_phi, ell_max_out)
    dev_out = None
    trailing = tuple(w_modes._data_shape()[2:])
    if group is not None:
        from . import sharding

        if trailing or aux:
            raise NotImplementedError("a series sharded over a process group carries neither trailing data dimensions nor psi companions")
        t_global, have = sharding.gather_time_axis(w_modes.t, group, ctx=w_modes._ctx)
        resident = getattr(w_modes, "is_device_resident", False)
        t_new, rows, _ = sharding.transform_modes_sharded(
            w_modes._dev if resident else w_modes.data, t_global, w_modes.ell_min, w_modes.ell_max, s, w_modes.conformal_weight, type_term, tr,
            group=group, have=have, partition=partition, overlap=overlap_halo, ctx=w_modes._ctx,
        )
        if resident:
            dev_out, data_new = rows, np.empty((0, 0))
        else:
            data_new = rows
    elif trailing:
        # Extra trailing data dimensions (scri/waveform_grid.py:299-308, 574-594: `final_dim`): every trailing index is an
        # independent series under the same transformation -- to_modes and the spline loop of from_modes walk them one by one.
        # Here they are the extra column blocks of ONE engine call (bms_transform_modes_series): transformation, time axis, tables and
        # synthesis matrix are shared.  (The reference's own from_modes cannot be run on such data: its tensordot puts the extra axes
        # BEFORE the grid axes and the per-pixel loop then indexes the wrong axis -- IndexError at :581.  What is built here is what
        # those loops say.)
        if aux and any(np.shape(a[0])[2:] != trailing for a in aux):
            raise ValueError("auxiliary waveforms must carry the same trailing data dimensions")
        flat = np.reshape(w_modes.data, w_modes.data.shape[:2] + (-1,))
        aux_flat = [(np.reshape(a[0], np.shape(a[0])[:2] + (-1,)),) + tuple(a[1:]) for a in aux]
        t_new, d = engine.transform_modes_series(w_modes.t, flat, w_modes.ell_min, w_modes.ell_max, s, w_modes.conformal_weight, type_term, tr,
                                                 aux=aux_flat, ctx=w_modes._ctx)
        data_new = d.reshape(d.shape[:2] + trailing)
    elif getattr(w_modes, "is_device_resident", False) and not aux:
        # weights in HBM (WaveformModes.to_device): the transformation reads and writes them there
        from . import device_series
        from .mode_algebra import LM_total_size

        n_out = LM_total_size(ell_min_out, int(ell_max_out))
        out = device_series.empty(w_modes._ctx, (w_modes.n_times, n_out))
        t_new, n_new = engine.transform_modes(
            w_modes.t, w_modes._dev.data_ptr(), w_modes.ell_min, w_modes.ell_max, s, w_modes.conformal_weight, type_term, tr,
            ctx=w_modes._ctx, device=True, ld=w_modes.n_modes, out_ptr=out.data_ptr(),
        )
        dev_out, data_new = out[:n_new], np.empty((0, 0))
    else:
        t_new, data_new = engine.transform_modes(
            w_modes.t, w_modes.data, w_modes.ell_min, w_modes.ell_max, s, w_modes.conformal_weight, type_term, tr,
            aux=aux, ctx=w_modes._ctx, devices=None if aux else devices,
        )
    if kwargs:
        warnings.warn("\nUnused kwargs passed to this function:\n{}".format(pprint.pformat(kwargs, width=1)))

    return _with_device_data(dev_out, WaveformModes, dict(
        t=t_new,
        data=data_new,
        history=w_modes.history,
        ell_min=ell_min_out,
        ell_max=int(ell_max_out),
        frameType=w_modes.frameType,
        dataType=w_modes.dataType,
        r_is_scaled_out=w_modes.r_is_scaled_out,
        m_is_scaled_out=w_modes.m_is_scaled_out,
        constructor_statement=f"WaveformGrid.from_modes({w_modes}, **{original_kwargs}).to_modes({ell_max_out})",
        ctx=w_modes._ctx,
    ))


def _with_device_data(dev, cls, kw):
    """cls(**kw), with the mode weights a device tensor when `dev` is given (the shape check of the constructor sees them)."""
    if dev is None:
        return cls(**kw)
    kw = dict(kw, data=np.empty((0, dev.shape[1]), dtype=complex))
    w = cls(**kw)
    w._host, w._dev = None, dev
    return w


class WaveformGrid:
    """A waveform on a (theta, phi) grid at each time (the part of scri.WaveformGrid the transformation path shows:
    scri/waveform_grid.py:193-630): `from_modes` puts a WaveformModes object on the -- optionally BMS-transformed -- grid,
    `to_modes` analyses it back, `transform` does both.  data: complex [n_times, n_theta * n_phi, ...], theta-major; extra
    trailing dimensions are independent series (scri/waveform_grid.py:299-308, 574-594)."""

    def __init__(self, t=None, data=None, n_theta=0, n_phi=0, frameType=None, dataType=None, r_is_scaled_out=False, m_is_scaled_out=False,
                 history=(), ctx=None, constructor_statement=None, frame=None):
        # Keywords with the reference's defaults (scri/waveform_base.py:220-231, waveform_grid.py:194-199), or one WaveformGrid to copy
        # (waveform_base.py:240-252); the positional order is this package's own.
        from . import UnknownDataType, UnknownFrameType

        if isinstance(t, WaveformGrid):
            if data is not None:
                raise ValueError(
                    "Did not understand input arguments to `WaveformGrid` constructor.\n"
                    "Note that explicit data values must be passed as keywords,\n"
                    "whereas objects to be copied must be passed as the sole argument."
                )
            o = t
            t, data, n_theta, n_phi, frame = o.t.copy(), o.data.copy(), o.n_theta, o.n_phi, np.array(o.frame, copy=True)
            frameType, dataType, r_is_scaled_out, m_is_scaled_out = o.frameType, o.dataType, o.r_is_scaled_out, o.m_is_scaled_out
            history, ctx, constructor_statement = o.history, o._ctx, f"WaveformGrid({o})"
        self.t = np.array(np.empty((0,)) if t is None else t, dtype=float)
        self.data = np.asarray(np.empty((0, 0)) if data is None else data, dtype=complex)
        self.n_theta, self.n_phi = int(n_theta), int(n_phi)
        self.frame = np.zeros((0, 4)) if frame is None or np.size(frame) == 0 else np.atleast_2d(quaternions.as_float_array(frame)).copy()
        self.frameType = UnknownFrameType if frameType is None else frameType
        self.dataType = UnknownDataType if dataType is None else dataType
        self.r_is_scaled_out, self.m_is_scaled_out = bool(r_is_scaled_out), bool(m_is_scaled_out)
        from .waveform_modes import _next_num  # (one counter for both containers, as the reference's WaveformBase keeps)

        _next_num[0] += 1
        self.num = _next_num[0]
        self.history = list(history) + [f"{self} = {constructor_statement or 'WaveformGrid(...)'}"]
        self._ctx = ctx
        if self.data.ndim < 2 or self.data.shape[:2] != (self.t.size, self.n_theta * self.n_phi):
            raise ValueError(f"data.shape={self.data.shape} does not agree with n_times={self.t.size}, n_theta*n_phi={self.n_theta * self.n_phi}")

    @property
    def n_times(self):
        return self.t.shape[0]

    @property
    def spin_weight(self):
        from . import SpinWeights

        return SpinWeights[self.dataType]

    @classmethod
    def from_modes(cls, w_modes, **kwargs):
        """The modes on the grid of the (rotated, boosted, supertranslated) frame at its retarded-time slices
        (scri/waveform_grid.py:331-613); keyword arguments as `transform`, without `ell_max`."""
        from .waveform_modes import WaveformModes

        if not isinstance(w_modes, WaveformModes):
            raise TypeError(  # (from_modes' own wording, scri/waveform_grid.py:417-421; `transform` has another, :626-629)
                "\nInput waveform object must be an instance of `WaveformModes`; " "this is of type `{}`".format(type(w_modes).__name__)
            )
        original_kwargs = kwargs.copy()
        supertranslation, frame_rotation, boost_velocity, n_theta, n_phi, type_term, aux = _prepare(w_modes, kwargs)
        tr = engine.make_transformation(supertranslation, frame_rotation, boost_velocity, n_theta, n_phi, w_modes.ell_max)
        trailing = tuple(np.shape(w_modes.data)[2:])
        if trailing:  # every trailing index is a series of its own: the column blocks of one engine call (see `transform`)
            if aux and any(np.shape(a[0])[2:] != trailing for a in aux):
                raise ValueError("auxiliary waveforms must carry the same trailing data dimensions")
            flat = np.reshape(w_modes.data, w_modes.data.shape[:2] + (-1,))
            aux_flat = [(np.reshape(a[0], np.shape(a[0])[:2] + (-1,)),) + tuple(a[1:]) for a in aux]
            t_new, g = engine.transform_modes_series(w_modes.t, flat, w_modes.ell_min, w_modes.ell_max, w_modes.spin_weight,
                                                     w_modes.conformal_weight, type_term, tr, aux=aux_flat, ctx=w_modes._ctx, grid=True)
            grid = g.reshape(g.shape[:2] + trailing)
        else:
            t_new, grid = engine.transform_modes(
                w_modes.t, w_modes.data, w_modes.ell_min, w_modes.ell_max, w_modes.spin_weight, w_modes.conformal_weight, type_term, tr,
                aux=aux, ctx=w_modes._ctx, grid=True,
            )
        if kwargs:
            warnings.warn("\nUnused kwargs passed to this function:\n{}".format(pprint.pformat(kwargs, width=1)))
        return cls(
            t_new, grid, n_theta, n_phi, w_modes.frameType, w_modes.dataType, w_modes.r_is_scaled_out, w_modes.m_is_scaled_out,
            history=w_modes.history, ctx=w_modes._ctx,
            constructor_statement=f"{cls.__name__}.from_modes({w_modes}, **{original_kwargs})",
        )

    def to_modes(self, ell_max=None, ell_min=None):
        """Spin-weighted spherical-harmonic modes of the grid data (scri/waveform_grid.py:274-329): map2salm on the GPU."""
        import numbers

        from .waveform_modes import WaveformModes

        s = self.spin_weight
        if ell_max is None:
            ell_max = int((max(self.n_theta, self.n_phi) - 1) // 2)
        if ell_min is None:
            ell_min = abs(s)
        if not isinstance(ell_max, numbers.Integral) or ell_max < 0:
            raise ValueError(f"Input `ell_max` should be a nonnegative integer; got `{ell_max}`.")
        if not isinstance(ell_min, numbers.Integral) or ell_min < 0 or ell_min > ell_max:
            raise ValueError(f"Input `ell_min` should be an integer between 0 and {ell_max}; got `{ell_min}`.")
        trailing = self.data.shape[2:]
        if trailing:
            # final_dim (scri/waveform_grid.py:299-308): the maps of all trailing indices of all times are one batch of map2salm
            final_dim = int(np.prod(trailing))
            grid = np.moveaxis(self.data.reshape(self.n_times, self.n_theta, self.n_phi, final_dim), 3, 1)
            modes = engine.map2salm(np.ascontiguousarray(grid).reshape(self.n_times * final_dim, self.n_theta, self.n_phi), s, int(ell_max),
                                    ell_min=int(ell_min), ctx=self._ctx)
            modes = np.moveaxis(modes.reshape(self.n_times, final_dim, -1), 1, 2).reshape((self.n_times, -1) + tuple(trailing))
        else:
            grid = self.data.reshape(self.n_times, self.n_theta, self.n_phi)
            modes = engine.map2salm(grid, s, int(ell_max), ell_min=int(ell_min), ctx=self._ctx)
        return WaveformModes(
            t=self.t, data=modes, history=self.history, ell_min=int(ell_min), ell_max=int(ell_max), frameType=self.frameType,
            dataType=self.dataType, r_is_scaled_out=self.r_is_scaled_out, m_is_scaled_out=self.m_is_scaled_out,
            constructor_statement=f"{self}.to_modes({ell_max})", ctx=self._ctx,
        )

    @classmethod
    def transform(cls, w_modes, **kwargs):
        """from_modes(...).to_modes(ell_max), `ell_max` defaulting to the input's (scri/waveform_grid.py:615-630); the fused
        engine call of the module-level `transform`."""
        return transform(w_modes, **kwargs)

    def __str__(self):
        return f"{type(self).__name__}_{self.num}"

    def __repr__(self):
        opts = np.get_printoptions()
        np.set_printoptions(threshold=6, linewidth=150, precision=6)
        try:
            rep = (f"{type(self).__name__}(\n    t={self.t},\n    data={str(self.data)},\n    frameType={self.frameType}, dataType={self.dataType},\n"
                   f"    r_is_scaled_out={self.r_is_scaled_out}, m_is_scaled_out={self.m_is_scaled_out})  # num = {self.num}")
        finally:
            np.set_printoptions(**opts)
        return rep + f"\n# n_theta={self.n_theta}, n_phi={self.n_phi}"
