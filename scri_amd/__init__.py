"""scri_amd -- MI355X-native BMS-transformation engine with the interface of moble/scri's hot path.

Mirrors (same names, argument meaning and error behaviour) the part of scri that lies on the
BMS-transformation path:

    WaveformModes.transform / rotate_decomposition_basis / rotate_physical_system / to_inertial_frame
    AsymptoticBondiData.transform
    bms_transformations.LorentzTransformation / BMSTransformation / transform_supertranslation
    ModesTimeSeries.interpolate / derivative / antiderivative / grid_multiply   (next row of the scope table)

All arithmetic on the data runs in hand-written HIP kernels for gfx950 behind the C ABI in
``include/scri_amd.h`` (``scri_amd/libscri_amd.so``).  There is no CPU fallback.
"""
import sys

# type tables: scri/__init__.py:78-86
FrameType = [UnknownFrameType, Inertial, Coprecessing, Coorbital, Corotating] = range(5)
FrameNames = ["UnknownFrameType", "Inertial", "Coprecessing", "Coorbital", "Corotating"]
DataType = [UnknownDataType, psi0, psi1, psi2, psi3, psi4, sigma, h, hdot, news, psin, psim] = range(12)
DataNames = ["UnknownDataType", "Psi0", "Psi1", "Psi2", "Psi3", "Psi4", "sigma", "h", "hdot", "news", "psin", "PsiM"]
SpinWeights = [sys.maxsize, 2, 1, 0, -1, -2, 2, -2, -2, -2, sys.maxsize, 0]
ConformalWeights = [sys.maxsize, 2, 1, 0, -1, -2, 1, 0, -1, -1, -3, 0]
RScaling = [sys.maxsize, 5, 4, 3, 2, 1, 2, 1, 1, 1, 0, 0]
MScaling = [sys.maxsize, 2, 2, 2, 2, 2, 0, 0, 1, 1, 2, 1]
DataNamesLaTeX = [r"\mathrm{unknown data type}", r"\psi_0", r"\psi_1", r"\psi_2", r"\psi_3", r"\psi_4", r"\sigma", r"h", r"\dot{h}", r"\mathrm{n}",
                  r"\psi_n", r"\psi_M"]
# units (scri/__init__.py:60-76): c exact; G M_sun / c^2 and / c^3 ("2015 Selected Astronomical Constants", TDB); the parsec of IAU 2012 B2
speed_of_light = 299792458.0  # m/s
m_sun_in_meters = 1476.62503851  # m
m_sun_in_seconds = 4.92549094916e-06  # s
parsec_in_meters = 3.0856775814913672789139379577965e16  # m

from . import _lib  # noqa: E402  (raises ImportError if libscri_amd.so has not been built)
from ._lib import Context, default_context, BMSError  # noqa: E402,F401
from . import engine  # noqa: E402,F401
from .waveform_modes import WaveformModes  # noqa: E402,F401
from .waveform_grid import WaveformGrid  # noqa: E402,F401

# scri/waveform_grid.py:640-641: the modes <-> grid conversions as capabilities of WaveformModes
WaveformModes.to_grid = lambda w_modes, **kwargs: WaveformGrid.from_modes(w_modes, **kwargs)
WaveformModes.from_grid = classmethod(lambda cls, w_grid, ell_max: WaveformGrid.to_modes(w_grid, ell_max))
WaveformModes.to_grid.__doc__ = WaveformGrid.from_modes.__doc__
WaveformModes.from_grid.__func__.__doc__ = WaveformGrid.to_modes.__doc__
from .rotations import rotate_decomposition_basis, rotate_physical_system, to_inertial_frame, to_corotating_frame, to_coprecessing_frame  # noqa: E402,F401
from .rotations import get_alignment_of_decomposition_frame_to_modes, align_decomposition_frame_to_modes  # noqa: E402,F401
from .asymptotic_bondi_data import AsymptoticBondiData  # noqa: E402,F401
from . import bms_transformations  # noqa: E402,F401
from . import mode_calculations  # noqa: E402,F401
from .modes_time_series import ModesTimeSeries  # noqa: E402,F401
from . import file_io  # noqa: E402,F401
from .file_io import create_abd_from_h5, create_abd_from_waveforms  # noqa: E402,F401
from .bms_transformations import LorentzTransformation, BMSTransformation  # noqa: E402,F401
from . import flux  # noqa: E402,F401
from . import utilities  # noqa: E402,F401
from . import sample_waveforms  # noqa: E402,F401
from .flux import energy_flux, momentum_flux, angular_momentum_flux, boost_flux, poincare_fluxes  # noqa: E402,F401
from .mode_calculations import (  # noqa: E402,F401
    LdtVector, LVector, LLComparisonMatrix, LLMatrix, LLDominantEigenvector, angular_velocity, corotating_frame, inner_product,
)

# Same grafting the reference performs at import (scri/__init__.py:140-142)
WaveformModes.rotate_decomposition_basis = rotate_decomposition_basis
WaveformModes.rotate_physical_system = rotate_physical_system
WaveformModes.to_inertial_frame = to_inertial_frame
WaveformModes.to_corotating_frame = to_corotating_frame
WaveformModes.to_coprecessing_frame = to_coprecessing_frame
WaveformModes.get_alignment_of_decomposition_frame_to_modes = get_alignment_of_decomposition_frame_to_modes
WaveformModes.LVector = mode_calculations.LVector
WaveformModes.LLComparisonMatrix = mode_calculations.LLComparisonMatrix
WaveformModes.energy_flux = energy_flux
WaveformModes.momentum_flux = momentum_flux
WaveformModes.angular_momentum_flux = angular_momentum_flux
WaveformModes.boost_flux = boost_flux
WaveformModes.poincare_fluxes = poincare_fluxes
WaveformModes.align_decomposition_frame_to_modes = align_decomposition_frame_to_modes


def patch_scri(scri=None, ctx=None):
    """Graft the GPU implementations onto an installed `scri` (opt-in drop-in): the two rotation
    kernels, `WaveformModes.transform` and `AsymptoticBondiData.transform`.  Returns the patched attribute names."""
    if scri is None:
        import scri  # only available where the reference is installed
    from . import adapters

    return adapters.install(scri, ctx=ctx)


def unpatch_scri(scri=None):
    """Undo `patch_scri`: put the reference's own functions back (kept as `_reference` aliases)."""
    if scri is None:
        import scri
    from . import adapters

    adapters.uninstall(scri)

# The reference's dotted paths.  There `scri.asymptotic_bondi_data` is a package (bms_charges, constraints, from_initial_values,
# map_to_abd_frame, map_to_superrest_frame, transformations: scri/asymptotic_bondi_data/__init__.py:235-263) and the readers live under
# `scri.SpEC`; user code and the tutorials spell those paths out (docs/tutorial_abd.rst:88,340), so they resolve here too.
from . import asymptotic_bondi_data, map_to_superrest_frame as _superrest, map_to_abd_frame as _abd_frame  # noqa: E402
from . import bms_charges as _charges, abd_ivp as _ivp  # noqa: E402

import sys as _sys  # noqa: E402
import types as _types  # noqa: E402

SpEC = _types.ModuleType(__name__ + ".SpEC")
SpEC.file_io, SpEC.create_abd_from_h5 = file_io, create_abd_from_h5
for _name, _module in (("map_to_superrest_frame", _superrest), ("map_to_abd_frame", _abd_frame), ("bms_charges", _charges),
                       ("from_initial_values", _ivp), ("constraints", _ivp), ("transformations", asymptotic_bondi_data)):
    setattr(asymptotic_bondi_data, _name, _module)
    _sys.modules[f"{__name__}.asymptotic_bondi_data.{_name}"] = _module  # (`from scri_amd.asymptotic_bondi_data.bms_charges import ...`)
_sys.modules[__name__ + ".SpEC"] = SpEC
_sys.modules[__name__ + ".SpEC.file_io"] = file_io
