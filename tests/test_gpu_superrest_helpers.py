"""The small public helpers of scri/asymptotic_bondi_data/map_to_superrest_frame.py that callers import by name
(map_to_abd_frame.py:19): MT_to_WM, WM_to_MT, the operator pair, time_translation and rotation."""
import numpy as np
import pytest

from oracle import bms_charges_ref as cref
from oracle import rotations_ref, quat
from oracle.containers import WM, h, psi2, SpinWeights

pytestmark = pytest.mark.gpu


def _abd(ctx, n=60, ell_max=4, seed=3):
    import scri_amd
    from tests.test_gpu_transform_abd import smooth_abd

    o = smooth_abd(n, ell_max, seed)
    a = scri_amd.AsymptoticBondiData(o.u, ell_max, ctx=ctx)
    a._raw_data[:] = o.raw
    return o, a


def test_rotation_is_the_reference_detour_and_a_frame_rotation(ctx):
    """rotation(abd, phi): the reference turns h = 2 sigma-bar and the rescaled Weyl scalars with rotate_physical_system and undoes
    the rescaling (map_to_superrest_frame.py:687-717).  Restated literally with the oracle (bar, rotate, bar) and compared; and it is
    the frame rotation by +phi about z that abd.transform applies."""
    from scri_amd import map_to_superrest_frame as m

    o, a = _abd(ctx)
    phi = 0.83
    got = m.rotation(a, phi)
    q = np.array([np.cos(-phi / 2), 0.0, 0.0, np.sin(-phi / 2)])  # from_rotation_vector(-phi z)
    q_basis = quat.qconj(q)  # rotate_physical_system(q) = rotate_decomposition_basis(q^-1)
    spins = (2, 1, 0, -1, -2, 2)
    for f, s in enumerate(spins):
        field = o.raw[f]
        if f == 5:  # sigma: through h = 2 sigma-bar (spin -2) and back
            hh = 2.0 * cref.bar(field, 2)
            w = WM(t=o.u, data=hh[:, 4:].copy(), ell_min=2, ell_max=o.ell_max, dataType=h)
            w = rotations_ref.rotate_decomposition_basis(w, q_basis)
            full = np.zeros_like(field)
            full[:, 4:] = w.data
            expect = 0.5 * cref.bar(full, -2)
        else:
            lmin = abs(s)
            fac = 0.5 * (-np.sqrt(2)) ** (4 - f)
            w = WM(t=o.u, data=(fac * field)[:, lmin**2 :].copy(), ell_min=lmin, ell_max=o.ell_max, dataType=psi2)
            w = rotations_ref.rotate_decomposition_basis(w, q_basis)
            expect = np.zeros_like(field)
            expect[:, lmin**2 :] = 2 * (-1.0 / np.sqrt(2)) ** (4 - f) * w.data
        assert np.abs(got._raw_data[f] - expect).max() < 1e-13 * max(1.0, np.abs(expect).max()), f
    again = a.transform(frame_rotation=[np.cos(phi / 2), 0.0, 0.0, np.sin(phi / 2)])
    assert again.n_times == got.n_times
    assert np.abs(again._raw_data - got._raw_data).max() < 1e-12 * np.abs(got._raw_data).max()
    assert np.array_equal(m.rotation(a, 0.0)._raw_data, a._raw_data)


def test_conversions_time_translation_and_operators(ctx):
    import scri_amd
    from scri_amd import map_to_superrest_frame as m

    o, a = _abd(ctx)
    w = m.MT_to_WM(2.0 * a.sigma.bar)
    assert (w.ell_min, w.ell_max, w.dataType, w.frameType) == (2, o.ell_max, scri_amd.h, scri_amd.Inertial) and w.r_is_scaled_out and w.m_is_scaled_out
    assert np.abs(w.data - 2.0 * cref.bar(o.raw[5], 2)[:, 4:]).max() < 1e-15 * np.abs(o.raw[5]).max()
    news = m.MT_to_WM(2.0 * a.sigma.bar.dot, dataType=scri_amd.hdot)
    assert news.dataType == scri_amd.hdot
    with pytest.raises(NotImplementedError):
        m.MT_to_WM(a.sigma, sxs_version=True)
    back = m.WM_to_MT(w)
    assert (back.spin_weight, back.ell_min, back.ell_max) == (-2, 2, o.ell_max) and np.array_equal(np.asarray(back), w.data) and np.array_equal(back.t, w.t)
    assert back.multiply(back.bar).ell_max == o.ell_max  # truncator max
    shifted = m.time_translation(a, 2.5)
    assert np.array_equal(shifted.t, a.t - 2.5) and np.array_equal(shifted.sigma.t, a.t - 2.5) and np.array_equal(shifted._raw_data, a._raw_data)
    assert np.array_equal(a.t, o.u)  # the source keeps its axis
    x = np.arange(25, dtype=complex)[None, :] + 1.0
    d = m.𝔇(x, 4)
    assert np.array_equal(d[0, :4], np.zeros(4)) and d[0, 4] == x[0, 4] * 6.0 and d[0, 9] == x[0, 9] * 30.0
    assert np.abs(m.𝔇inverse(d, 4)[0, 4:] - x[0, 4:]).max() < 1e-14 * 25 and np.array_equal(m.𝔇inverse(d, 4)[0, :4], np.zeros(4))
