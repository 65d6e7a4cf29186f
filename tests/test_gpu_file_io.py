"""create_abd_from_waveforms (the array-level part of scri.SpEC.file_io.create_abd_from_h5, :733-829) against the loop
restatement in oracle/file_io_ref.py; interpolation and the superrest step run on the GPU building blocks."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _waveforms(ctx, n=400, ell_max=5, labels=("Psi4", "Psi3", "Psi2", "Psi1", "Psi0", "h"), m_is_scaled_out=False):
    import scri_amd
    from oracle import file_io_ref

    rng = np.random.default_rng(11)
    t = np.cumsum(rng.uniform(0.05, 0.15, size=n))
    t[50:53] = t[49]  # a stalled clock and a step back, as worldtube dumps have after a restart
    t[200] = t[190]
    ell_mins = {"Psi4": 2, "Psi3": 1, "Psi2": 0, "Psi1": 1, "Psi0": 2, "h": 2, "Strain": 0}
    fields, WMs = {}, {}
    for k in labels:
        nm = (ell_max + 1) ** 2 - ell_mins[k] ** 2
        fields[k] = rng.normal(size=(n, nm)) + 1j * rng.normal(size=(n, nm))
        WMs[k] = scri_amd.WaveformModes(
            t=t.copy(), data=fields[k].copy(), ell_min=ell_mins[k], ell_max=ell_max, frameType=scri_amd.Inertial,
            dataType=file_io_ref.DATATYPE[k], r_is_scaled_out=True, m_is_scaled_out=m_is_scaled_out, ctx=ctx,
        )
    return t, fields, ell_mins, WMs


@pytest.mark.parametrize("convention", ["SpEC", "Moreschi-Boyle"])
def test_assembly_matches_the_restatement(ctx, convention):
    import scri_amd
    from oracle import file_io_ref

    t, fields, ell_mins, WMs = _waveforms(ctx)
    abd = scri_amd.create_abd_from_waveforms(WMs, convention=convention, time_shift=3.5, ch_mass=0.97, ctx=ctx)
    u, raw = file_io_ref.assemble(t, fields, ell_mins, 5, convention=convention.lower(), time_shift=3.5, ch_mass=0.97)
    assert np.all(np.diff(abd.t) > 0) and abd.t.size < t.size
    assert np.array_equal(abd.t, u)
    assert np.array_equal(np.asarray(abd._raw_data), raw)
    assert all(w.m_is_scaled_out for w in WMs.values())


def test_strain_label_partial_fields_and_errors(ctx):
    import scri_amd
    from oracle import file_io_ref

    t, fields, ell_mins, WMs = _waveforms(ctx, labels=("Psi2", "Strain"), m_is_scaled_out=True)
    abd = scri_amd.create_abd_from_waveforms(WMs, ctx=ctx)
    u, raw = file_io_ref.assemble(t, fields, ell_mins, 5, m_is_scaled_out=True)
    assert np.array_equal(abd.t, u) and np.array_equal(np.asarray(abd._raw_data), raw)
    assert not np.asarray(abd.psi4).any()
    with pytest.raises(ValueError, match="at least one waveform"):
        scri_amd.create_abd_from_waveforms({}, ctx=ctx)
    WMs["Psi2"].t = WMs["Psi2"].t + 1.0
    with pytest.raises(ValueError, match="same set of times"):
        scri_amd.create_abd_from_waveforms(WMs, ctx=ctx)
    with pytest.raises(NotImplementedError):
        scri_amd.create_abd_from_h5("RPXMB", h="nowhere.h5")


def test_interpolation_window(ctx):
    """t_interpolate is cut to the samples strictly inside the data (:813-816) and evaluated with the GPU spline."""
    import scri_amd
    from scipy.interpolate import CubicSpline

    t, fields, ell_mins, WMs = _waveforms(ctx, labels=("Psi4", "h"), m_is_scaled_out=True)
    plain = scri_amd.create_abd_from_waveforms({k: w.copy() for k, w in WMs.items()}, ctx=ctx)
    t_new = np.linspace(plain.t[0] - 1.0, plain.t[-1] + 1.0, 777)
    abd = scri_amd.create_abd_from_waveforms(WMs, t_interpolate=t_new, ctx=ctx)
    idx1 = np.argmin(abs(t_new - plain.t[0])) + 1
    idx2 = np.argmin(abs(t_new - plain.t[-1]))
    assert np.array_equal(abd.t, t_new[idx1:idx2])
    expect = CubicSpline(plain.t, np.asarray(plain.sigma), axis=0)(abd.t)
    assert np.abs(np.asarray(abd.sigma) - expect).max() < 1e-11 * np.abs(expect).max()
