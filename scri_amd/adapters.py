"""Adapters between `scri` objects and the engine, used by `scri_amd.patch_scri()`."""
from . import waveform_grid
from .waveform_modes import WaveformModes as _WM


def install(scri):
    """Replace the three hot-path methods on scri's own classes by GPU-backed versions.  Returns the list of
    patched attributes.  The originals stay available as ``<name>_reference``."""
    patched = []

    def transform(self, **kwargs):
        aux = {k: v for k, v in kwargs.items() if k.startswith("psi") and k.endswith("_modes")}
        for k, v in aux.items():
            kwargs[k] = _WM(t=v.t, data=v.data, ell_min=v.ell_min, ell_max=v.ell_max, dataType=v.dataType,
                            frameType=v.frameType, r_is_scaled_out=v.r_is_scaled_out, m_is_scaled_out=v.m_is_scaled_out)
        w = _WM(t=self.t, data=self.data, ell_min=self.ell_min, ell_max=self.ell_max, dataType=self.dataType,
                frameType=self.frameType, r_is_scaled_out=self.r_is_scaled_out, m_is_scaled_out=self.m_is_scaled_out)
        out = waveform_grid.transform(w, **kwargs)
        return scri.WaveformModes(
            t=out.t, data=out.data, history=self.history, ell_min=out.ell_min, ell_max=out.ell_max,
            frameType=self.frameType, dataType=self.dataType, r_is_scaled_out=self.r_is_scaled_out,
            m_is_scaled_out=self.m_is_scaled_out, constructor_statement=f"{self}.transform(...)  # scri_amd",
        )

    scri.WaveformModes.transform_reference = scri.WaveformModes.transform
    scri.WaveformModes.transform = transform
    patched.append("WaveformModes.transform")
    return patched
