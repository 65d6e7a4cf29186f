"""Ingestion into AsymptoticBondiData: the array-level part of
`scri.SpEC.file_io.create_abd_from_h5` (scri/SpEC/file_io/__init__.py:575-830) -- monotonic time selection, scaling out
the Christodoulou mass, the SpEC -> Moreschi-Boyle convention factors (:772-811), strain -> shear, interpolation and the
optional map to the superrest frame -- on the GPU building blocks of this package, plus the SpECTRE CCE v1 reader itself
when `h5py` is importable (it is not part of this image; the RPXMB / RPDMB containers additionally need the third-party
`sxs` package and are not provided).
"""
import warnings

import numpy as np

from . import DataNames, Inertial, h, hdot, psi0, psi1, psi2, psi3, psi4
from .mode_algebra import LM_index

# input convention -> factors for [psi0, psi1, psi2, psi3, psi4, h]  (scri/SpEC/file_io/__init__.py:772-777)
CONVERSION_FACTORS = {
    "moreschi-boyle": [1, 1, 1, 1, 1, 1],
    "spec": [2, -np.sqrt(2), 1, -1 / np.sqrt(2), 0.5, 0.5],
}


def index_is_monotonic(y):
    """Mask of the samples that continue the strictly monotonic trend of y (scri/SpEC/file_io/__init__.py:51-69):
    a sample is kept when it lies beyond every earlier sample.  Vectorised with a running extremum."""
    y = np.asarray(y)
    keep = np.ones(y.shape, dtype=bool)
    if y.size < 2:
        return keep
    if y[-1] - y[0] > 0.0:
        keep[1:] = y[1:] > np.maximum.accumulate(y)[:-1]
    else:
        keep[1:] = y[1:] < np.minimum.accumulate(y)[:-1]
    return keep


def monotonic_indices(y):
    """scri/SpEC/file_io/__init__.py:72-74"""
    return np.arange(np.asarray(y).size)[index_is_monotonic(y)]


def monotonize(y):
    """scri/SpEC/file_io/__init__.py:77-78"""
    return np.asarray(y)[monotonic_indices(y)]


def _mass_power(data_type):
    """The power of the Christodoulou mass that a field of this type carries (psi_n: n - 4, strain: -1, its time derivative: 0)"""
    if data_type in (psi0, psi1, psi2, psi3, psi4):
        return data_type - 4
    if data_type == h:
        return -1
    if data_type == hdot:
        return 0
    raise ValueError("DataType not determined.")


def make_variable_dimensionless(WM, ch_mass=1.0):
    """Scale out the Christodoulou mass, in place (scri/SpEC/file_io/__init__.py:421-450): times in units of the mass, the field
    by the mass to the power its type carries."""
    if WM.m_is_scaled_out:
        print("Data is already dimensionless!")
        return
    power = _mass_power(WM.dataType)  # (raises before anything is touched)
    WM.t, WM.data = WM.t / ch_mass, WM.data * float(ch_mass) ** power
    WM.m_is_scaled_out = True


def create_abd_from_waveforms(
    WMs, convention="SpEC", time_shift=0.0, ch_mass=None, t_interpolate=None, t_0_superrest=None, padding_time=None, ctx=None
):
    """Everything `create_abd_from_h5` does after the files are read (scri/SpEC/file_io/__init__.py:733-829).

    WMs: dict label -> WaveformModes with labels among "Psi4", "Psi3", "Psi2", "Psi1", "Psi0" and "h" (or "Strain"), all
    on the same times.  time_shift is subtracted from the times (the worldtube radius of SpECTRE CCE output)."""
    from .asymptotic_bondi_data import AsymptoticBondiData

    convention = convention.lower()
    if convention not in CONVERSION_FACTORS:
        raise KeyError(convention)
    if not WMs:
        raise ValueError("No filenames have been provided. The data of at least one waveform quantity is required.")
    labels = list(WMs.keys())
    WM_ref = WMs[labels[0]]
    for label in labels:
        if WMs[label].t.shape != WM_ref.t.shape or not (WM_ref.t == WMs[label].t).all():
            raise ValueError(
                "All waveforms must share the same set of times. The data " f"for {label} has a different set of times."
            )
    for label in labels:
        w = WMs[label]
        if ch_mass is not None:
            make_variable_dimensionless(w, ch_mass)
        if time_shift:
            w.t = w.t - float(time_shift)
        keep = index_is_monotonic(w.t)
        if not keep.all():
            w.t = w.t[keep]
            w.data = w.data[keep]

    abd = AsymptoticBondiData(time=WM_ref.t, ell_max=WM_ref.ell_max, multiplication_truncator=max, ctx=ctx)
    factors = CONVERSION_FACTORS[convention]
    for i, (label, field) in enumerate(
        (("Psi0", abd.psi0), ("Psi1", abd.psi1), ("Psi2", abd.psi2), ("Psi3", abd.psi3), ("Psi4", abd.psi4))
    ):
        if label in WMs:
            w = WMs[label]
            field[:, LM_index(w.ell_min, -w.ell_min, 0) :] = factors[i] * w.data
    # the shear is the conjugate of the (rescaled) strain
    for label in ("h", "Strain"):
        if label in WMs:
            w = WMs[label]
            abd.sigma[:, LM_index(w.ell_min, -w.ell_min, 0) :] = factors[5] * w.data
            abd.sigma = abd.sigma.bar
            break

    if t_interpolate is not None:
        t_interpolate = np.asarray(t_interpolate, dtype=float)
        idx1 = np.argmin(abs(t_interpolate - abd.t[0])) + 1
        idx2 = np.argmin(abs(t_interpolate - abd.t[-1])) + 1 - 1
        abd = abd.interpolate(t_interpolate[idx1:idx2])
    if t_0_superrest is not None and padding_time is not None:
        abd, _, _ = abd.map_to_superrest_frame(t_0=t_0_superrest, padding_time=padding_time)
    return abd


def create_abd_from_h5(file_format, convention="SpEC", radius=None, ch_mass=None, t_interpolate=None, t_0_superrest=None,
                       padding_time=None, **kwargs):
    """scri.SpEC.file_io.create_abd_from_h5 for SpECTRE CCE v1 output (scri/SpEC/file_io/__init__.py:575-732): the six
    `[time, Re, Im, ...]` datasets of the `Cce` (or `SpectreR####`) group -> AsymptoticBondiData in the Moreschi-Boyle
    convention.  Needs `h5py`."""
    from .waveform_modes import WaveformModes

    file_format = file_format.lower()
    if file_format != "spectrecce_v1":
        raise NotImplementedError(
            f"file format '{file_format}': only 'SpECTRECCE_v1' is read here (the SXS/RPXMB/RPDMB containers are the "
            "`sxs` package's); build the WaveformModes objects and call create_abd_from_waveforms"
        )
    try:
        import h5py
    except ImportError as e:
        raise ImportError("create_abd_from_h5 needs h5py; with arrays at hand use create_abd_from_waveforms") from e
    try:
        file_name = kwargs.pop("file_name")
    except KeyError:
        raise ValueError('Need to specify "file_name" option!')
    WMs = {}
    with h5py.File(file_name, "r") as f:
        for x in f.keys():
            if "Spectre" in x:
                cce_key, suffix = x, ""
                radius = cce_key.split("R")[1][:4]
                break
        else:
            cce_key, suffix = "Cce", ".dat"
        cce = f[cce_key]
        time = cce[f"Strain{suffix}"][:, 0]
        indices = monotonic_indices(time)
        time = time[indices]
        ell_max = int(np.sqrt((cce[f"Strain{suffix}"].shape[1] - 1) / 2) - 1)
        for label in ["Psi4", "Psi3", "Psi2", "Psi1", "Psi0", "Strain"]:
            dataType = DataNames.index(label) if label != "Strain" else DataNames.index("h")
            WMs[label] = WaveformModes(
                t=time.copy(), data=np.ascontiguousarray(cce[f"{label}{suffix}"][indices, 1:]).view(np.complex128), ell_min=0,
                ell_max=ell_max, frameType=Inertial, dataType=dataType, r_is_scaled_out=True, m_is_scaled_out=False,
            )
    if kwargs:
        import pprint

        warnings.warn("\nUnused kwargs passed to this function:\n{}".format(pprint.pformat(kwargs, width=1)))
    return create_abd_from_waveforms(
        WMs, convention=convention, time_shift=float(radius) if radius is not None else 0.0, ch_mass=ch_mass,
        t_interpolate=t_interpolate, t_0_superrest=t_0_superrest, padding_time=padding_time,
    )
