"""Adapters between `scri` objects and the engine, used by `scri_amd.patch_scri()`.

The reference wires its hot path by attribute grafting (scri/__init__.py:125-150; class-body imports in
scri/asymptotic_bondi_data/__init__.py:235-263).  `install` grafts the GPU path the same way, at the three seams of
SURVEY section 8(b):

  * `scri.rotations._rotate_decomposition_basis_by_constant / _by_series` (the two numba kernels, rotations.py:346-392)
    are replaced by ctypes stubs with the kernels' own signatures -- so `rotate_decomposition_basis`,
    `rotate_physical_system`, `to_inertial_frame` and everything else that reaches them keep the reference's argument
    checks, frame bookkeeping and history, and only the arithmetic moves to the GPU;
  * `scri.WaveformModes.transform` (waveform_modes.py:705-719 -> waveform_grid.py:615-630);
  * `scri.AsymptoticBondiData.transform` (transformations.py:199-431).

The originals stay reachable as `<name>_reference`; `uninstall` puts them back.
"""
import numpy as np

from . import engine, waveform_grid
from .asymptotic_bondi_data import _process_transformation_kwargs as _abd_kwargs
from .waveform_modes import WaveformModes as _WM


def _quaternion_components(q):
    """np.quaternion (or anything with .components / w,x,y,z) -> float[4]; arrays pass through."""
    if hasattr(q, "components"):
        return np.asarray(q.components, dtype=float)
    if all(hasattr(q, a) for a in "wxyz"):
        return np.array([q.w, q.x, q.y, q.z], dtype=float)
    return np.asarray(q, dtype=float)


def install(scri, ctx=None):
    """Replace the hot-path entry points on scri's own modules and classes by GPU-backed versions.  Returns the list of
    patched attributes."""
    patched = []
    rot = scri.rotations

    # ---- the two numba kernels, at their own signatures (scri/rotations.py:346-392)
    def _rotate_decomposition_basis_by_constant(data, ell_min, ell_max, D, tmp):
        _in_place(data, lambda d: engine.rotate_const_D(d, ell_min, ell_max, D, ctx=ctx))

    def _rotate_decomposition_basis_by_series(data, R_basis, ell_min, ell_max, D):
        _in_place(data, lambda d: engine.rotate_series(d, ell_min, ell_max, R_basis, ctx=ctx))

    def _in_place(data, f):
        if data.dtype == np.complex128 and data.ndim == 2 and data.strides[1] == 16 and data.strides[0] % 16 == 0:
            f(data)  # row stride passed as `ld`: sliced views need no copy
        else:
            tmp = np.ascontiguousarray(data, dtype=np.complex128)
            f(tmp)
            data[...] = tmp

    for name, fn in (("_rotate_decomposition_basis_by_constant", _rotate_decomposition_basis_by_constant),
                     ("_rotate_decomposition_basis_by_series", _rotate_decomposition_basis_by_series)):
        if not hasattr(rot, name + "_reference"):
            setattr(rot, name + "_reference", getattr(rot, name))
        setattr(rot, name, fn)
        patched.append(f"rotations.{name}")
    # rotate_decomposition_basis & co. look the kernels up in scri.rotations at call time: nothing else to replace
    patched += ["WaveformModes.rotate_decomposition_basis", "WaveformModes.rotate_physical_system", "WaveformModes.to_inertial_frame"]

    # ---- WaveformModes.transform
    def transform(self, **kwargs):
        aux = {k: v for k, v in kwargs.items() if k.startswith("psi") and k.endswith("_modes")}
        for k, v in aux.items():
            kwargs[k] = _WM(t=v.t, data=v.data, ell_min=v.ell_min, ell_max=v.ell_max, dataType=v.dataType,
                            frameType=v.frameType, r_is_scaled_out=v.r_is_scaled_out, m_is_scaled_out=v.m_is_scaled_out, ctx=ctx)
        w = _WM(t=self.t, data=self.data, ell_min=self.ell_min, ell_max=self.ell_max, dataType=self.dataType,
                frameType=self.frameType, r_is_scaled_out=self.r_is_scaled_out, m_is_scaled_out=self.m_is_scaled_out, ctx=ctx)
        out = waveform_grid.transform(w, **kwargs)
        return scri.WaveformModes(
            t=out.t, data=out.data, history=self.history, ell_min=out.ell_min, ell_max=out.ell_max,
            frameType=self.frameType, dataType=self.dataType, r_is_scaled_out=self.r_is_scaled_out,
            m_is_scaled_out=self.m_is_scaled_out, constructor_statement=f"{self}.transform(...)  # scri_amd",
        )

    if not hasattr(scri.WaveformModes, "transform_reference"):
        scri.WaveformModes.transform_reference = scri.WaveformModes.transform
    scri.WaveformModes.transform = transform
    patched.append("WaveformModes.transform")

    # ---- AsymptoticBondiData.transform
    def abd_transform(self, **kwargs):
        for k in ("frame_rotation",):
            if k in kwargs:
                kwargs[k] = _quaternion_components(kwargs[k])
        # scri_amd's two multi-GPU keywords: the devices of this process, or the process group over whose ranks
        # the time axis is split (this object then holds the rank's block of rows)
        devices, group = kwargs.pop("devices", None), kwargs.pop("group", None)
        frame_rotation, boost_velocity, supertranslation, working_ell_max, output_ell_max = _abd_kwargs(self.ell_max, **kwargs)
        n_theta = 2 * working_ell_max + 1
        tr = engine.make_transformation(supertranslation, frame_rotation, boost_velocity, n_theta, n_theta, output_ell_max)
        raw = np.ascontiguousarray(np.asarray(self._raw_data).view(np.ndarray), dtype=np.complex128)
        if group is not None:
            from . import sharding

            u_global, have = sharding.gather_time_axis(np.asarray(self.u, dtype=float), group)
            u_new, raw_new, _ = sharding.transform_abd_sharded(raw, u_global, self.ell_max, tr, group=group, have=have, ctx=ctx)
        else:
            u_new, raw_new = engine.transform_abd(np.asarray(self.u, dtype=float), raw, self.ell_max, tr, ctx=ctx, devices=devices)
        abdprime = type(self)(u_new, output_ell_max)  # transformations.py:417
        abdprime.psi0, abdprime.psi1, abdprime.psi2, abdprime.psi3, abdprime.psi4, abdprime.sigma = raw_new
        return abdprime

    if not hasattr(scri.AsymptoticBondiData, "transform_reference"):
        scri.AsymptoticBondiData.transform_reference = scri.AsymptoticBondiData.transform
    scri.AsymptoticBondiData.transform = abd_transform
    patched.append("AsymptoticBondiData.transform")
    return patched


def uninstall(scri):
    """Put the reference implementations back."""
    rot = scri.rotations
    for name in ("_rotate_decomposition_basis_by_constant", "_rotate_decomposition_basis_by_series"):
        if hasattr(rot, name + "_reference"):
            setattr(rot, name, getattr(rot, name + "_reference"))
            delattr(rot, name + "_reference")
    for cls in (scri.WaveformModes, scri.AsymptoticBondiData):
        if hasattr(cls, "transform_reference"):
            cls.transform = cls.transform_reference
            del cls.transform_reference
