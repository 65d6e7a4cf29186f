"""scri_amd.sample_waveforms (scri/sample_waveforms.py:12-381): the Racah 3-j symbols against the table computed with sympy
(tests/golden/g2_wigner_3j.npz), the deterministic generators against the oracle's restatement, the random ones for shape and
bookkeeping; on the GPU, the reference's own use of them: the analytically supertranslated single mode is what the transformation
produces (tests/test_waveform_grid.py of the reference)."""
import os

import numpy as np
import pytest

from oracle import sample_waveforms_ref as sref


def test_racah_3j_matches_the_golden_table():
    from scri_amd.sample_waveforms import wigner_3j

    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "g2_wigner_3j.npz"))
    for j1, j2, j3, m1, m2, m3, val in g["table"]:
        assert abs(wigner_3j(int(j1), int(j2), int(j3), int(m1), int(m2), int(m3)) - val) < 2e-15
    assert wigner_3j(2, 2, 5, 0, 0, 0) == 0.0 and wigner_3j(2, 2, 2, 1, 1, -1) == 0.0  # triangle rule, m sum
    for args in ((8, 6, 4, 3, -2, -1), (12, 12, 0, 5, -5, 0), (1, 8, 8, 0, -2, 2)):
        assert abs(wigner_3j(*args) - sref._w3j(*args)) < 1e-15


def test_deterministic_generators_match_the_oracle():
    import scri_amd
    from scri_amd import sample_waveforms as sw

    for kw in (dict(), dict(s=-1, ell=3, m=2), dict(s=0, ell=2, m=-1, ell_max=5, t_0=-3.0, t_1=4.0, dt=0.25)):
        a, b = sw.single_mode_constant_rotation(omega=0.3 + 0.02j, **kw), sref.single_mode_constant_rotation(omega=0.3 + 0.02j, **kw)
        assert np.array_equal(a.t, b.t) and np.array_equal(a.data, b.data) and (a.ell_min, a.ell_max, a.dataType) == (b.ell_min, b.ell_max, b.dataType)
        a, b = sw.single_mode_proportional_to_time(beta=2.0 - 1j, **kw), sref.single_mode_proportional_to_time(beta=2.0 - 1j, **kw)
        assert np.array_equal(a.data, b.data)
        st = np.zeros(16, dtype=complex)
        st[[0, 2, 5, 7, 12]] = [0.3, 0.1, 0.02 - 0.01j, 0.02 + 0.01j, 0.005]
        a = sw.single_mode_proportional_to_time_supertranslated(supertranslation=st, **kw)
        b = sref.single_mode_proportional_to_time_supertranslated(supertranslation=st, **kw)
        assert np.abs(a.data - b.data).max() < 1e-14 * np.abs(b.data).max()
        a = sw.single_mode_proportional_to_time_supertranslated(space_translation=[0.2, -0.1, 0.4], **kw)
        b = sref.single_mode_proportional_to_time_supertranslated(space_translation=[0.2, -0.1, 0.4], **kw)
        assert np.abs(a.data - b.data).max() < 1e-14 * np.abs(b.data).max()
    with pytest.raises(ValueError, match="Bad number of elements"):
        sw.single_mode_proportional_to_time_supertranslated(supertranslation=np.ones(5))
    c = sw.constant_waveform()
    assert c.data.shape == (1101, 77) and np.array_equal(c.data[0], c.data[-1]) and c.data[0, c.index(3, 2)] == 2 - 2j
    one = sw.single_mode(4, -3, ell_max=6)
    assert one.data.sum() == one.n_times and np.all(one.data[:, one.index(4, -3)] == 1.0)
    r = sw.random_waveform(n_times=50, seed=3)
    assert r.data.shape == (50, 77) and r.frame.shape == (50, 4) and r.frameType == scri_amd.Corotating and (np.diff(r.t) > 0).all()
    r2 = sw.random_waveform_proportional_to_time(n_times=40, rotating=False, uniform_time=True, seed=4)
    assert r2.frameType == scri_amd.Inertial and r2.frame.shape[0] == 0
    assert np.abs(r2.data[1:] / r2.t[1:, None] - r2.data[1] / r2.t[1]).max() < 1e-12
    with pytest.warns(UserWarning, match="Unused kwargs"):
        sw.single_mode(2, 2, nonsense=1)


@pytest.mark.gpu
def test_supertranslated_single_mode_is_what_the_transformation_gives(ctx):
    """the known answer of the reference's tests/test_waveform_grid.py::test_hyper_translation: psi3-type single mode beta t under a
    supertranslation, analytic against the transformation on the GPU"""
    import scri_amd
    from scri_amd import sample_waveforms as sw

    st = np.zeros(9, dtype=complex)  # l = 2 only: no time or space translation, so the time axis stays (as in the reference's test)
    st[6], st[7], st[5] = 0.03, 0.02 + 0.01j, -(0.02 - 0.01j)  # (2, 0) real; (2, -1) = -conj (2, 1): a real function
    for s, ell, m in ((-2, 2, 2), (-1, 3, -1), (0, 2, 0)):
        kw = dict(s=s, ell=ell, m=m, ell_max=8)
        w_in = sw.single_mode_proportional_to_time(ctx=ctx, **kw)
        aux = {}
        for i in range(s + 2):  # the higher Weyl scalars the mixing terms ask for, zeroed (tests/test_waveform_grid.py:117-120)
            a = sw.single_mode_proportional_to_time(s=i - 2, ctx=ctx)
            a.data *= 0
            aux[f"psi{4 - i}_modes"] = a
        expect = sw.single_mode_proportional_to_time_supertranslated(supertranslation=st, **kw)
        got = w_in.transform(supertranslation=st, **aux)
        i0 = int(np.argmin(np.abs(expect.t - got.t[0])))
        assert np.abs(expect.t[i0 : i0 + got.n_times] - got.t).max() < 1e-12
        assert np.abs(got.data - expect.data[i0 : i0 + got.n_times]).max() < 5e-14 * np.abs(expect.data).max(), (s, ell, m)


def test_g23_generators_vs_the_reference_itself():
    """the deterministic generators against scri/sample_waveforms.py run by the reference's own file (tests/golden/g23): the rotating
    and the linear-in-time single mode bit for bit, the analytically supertranslated one (the known answer of the reference's
    hyper-translation test, :312-381) to rounding -- the oracle's restatement and this package's both"""
    from scri_amd import sample_waveforms as sw

    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "g23_ref_sample_waveforms.npz"))
    st, space = g["supertranslation"], g["space_translation"]
    kws = (dict(), dict(s=-1, ell=3, m=2), dict(s=0, ell=2, m=-1, ell_max=5, t_0=-3.0, t_1=4.0, dt=0.25))
    for mod in (sw, sref):
        for i, kw in enumerate(kws):
            made = {
                "rot": mod.single_mode_constant_rotation(omega=0.3 + 0.02j, **kw),
                "prop": mod.single_mode_proportional_to_time(beta=2.0 - 1j, **kw),
                "super": mod.single_mode_proportional_to_time_supertranslated(supertranslation=st, **kw),
                "space": mod.single_mode_proportional_to_time_supertranslated(space_translation=list(space), **kw),
            }
            for tag, w in made.items():
                ref = g[f"{tag}_{i}_data"]
                assert np.array_equal(w.t, g[f"{tag}_{i}_t"]) and w.data.shape == ref.shape, (mod.__name__, tag, i)
                assert [w.ell_min, w.ell_max, int(w.dataType), int(w.frameType)] == list(g[f"{tag}_{i}_meta"]), (mod.__name__, tag, i)
                assert np.abs(w.data - ref).max() <= (0.0 if tag in ("rot", "prop") else 1e-14 * np.abs(ref).max()), (mod.__name__, tag, i)
    c = sw.constant_waveform()
    assert list(c.data.shape) == list(g["constant_shape"]) and np.array_equal(c.data[0], g["constant_row"])
