"""Container behaviour of WaveformModes that the path's callers rely on (slicing by times and ell ranges, mode indices,
norms): the reference's tests/test_waveform.py:76-181 on the same linear-in-time waveform (host logic only)."""
import numpy as np
import pytest


def linear_waveform(begin=-10.0, end=100.0, n_times=1000, ell_min=2, ell_max=8):
    """tests/conftest.py:82-106 of the reference: mode (l, m) = (m - i m) t, in a rotating frame"""
    import scri_amd

    rng = np.random.default_rng(1234)
    axis = rng.uniform(-1, 1, size=3)
    axis /= np.linalg.norm(axis)
    t = np.linspace(begin, end, num=n_times)
    omega = 2 * np.pi * 4 / (t[-1] - t[0])
    frame = np.array([[np.cos(omega * ti / 2), *(np.sin(omega * ti / 2) * axis)] for ti in t])
    lm = np.array([[ell, m] for ell in range(ell_min, ell_max + 1) for m in range(-ell, ell + 1)])
    data = np.empty((t.shape[0], lm.shape[0]), dtype=complex)
    for i, m in enumerate(lm[:, 1]):
        data[:, i] = (m - 1j * m) * t
    return scri_amd.WaveformModes(
        t=t, frame=frame, data=data, ell_min=ell_min, ell_max=ell_max, history=["# Called from linear_waveform"],
        frameType=scri_amd.Corotating, dataType=scri_amd.h, r_is_scaled_out=True, m_is_scaled_out=True,
    )


def test_indexing():
    w = linear_waveform()
    for i, (ell, m) in enumerate(w.LM):
        assert w.index(ell, m) == i
        assert np.allclose(w.data[:, w.index(ell, m)], (m - 1j * m) * w.t, rtol=0, atol=0)
    assert np.all(w.indices(w.LM) == range(w.n_modes))
    with pytest.raises(ValueError):
        w.indices([1.0, 2.0])


def test_empty_slice():
    import scri_amd

    w = linear_waveform()
    W = w[:0, :0]
    assert W.ensure_validity(alter=False)
    assert W.history[:1] == ["# Called from linear_waveform"]
    assert W.frameType == scri_amd.Corotating and W.dataType == scri_amd.h
    assert W.r_is_scaled_out and W.m_is_scaled_out
    assert W.num != w.num
    assert W.t.shape == (0,) and W.frame.shape[0] == 0 and W.data.shape == (0, 0)
    assert W.ell_min == 0 and W.ell_max == -1


def test_empty_mode_slice():
    w = linear_waveform()
    W = w[:, :0]
    assert W.ensure_validity(alter=False)
    assert np.all(W.t == w.t) and np.all(W.frame == w.frame)
    assert W.data.size == 0 and W.LM.size == 0
    assert W.ell_min == 0 and W.ell_max == -1
    assert W.history[:1] == ["# Called from linear_waveform"]
    assert (W.frameType, W.dataType, W.r_is_scaled_out, W.m_is_scaled_out) == (w.frameType, w.dataType, w.r_is_scaled_out, w.m_is_scaled_out)
    assert W.num != w.num


def test_time_slice():
    w = linear_waveform()
    W = w[10:50]
    assert W.ensure_validity(alter=False)
    assert np.all(W.t == w.t[10:50]) and np.all(W.frame == w.frame[10:50]) and np.all(W.data == w.data[10:50])
    assert np.all(W.LM == w.LM) and W.ells == w.ells
    assert W.history[:1] == ["# Called from linear_waveform"]
    assert isinstance(W.num, int) and W.num != w.num


def test_time_and_mode_slice():
    w = linear_waveform()
    W = w[10:50, :5]
    assert W.ensure_validity(alter=False)
    assert np.all(W.t == w.t[10:50]) and np.all(W.frame == w.frame[10:50])
    assert np.all(W.LM == np.array([[ell, m] for ell in range(w.ell_min, 5) for m in range(-ell, ell + 1)]))
    assert np.all(W.data == w.data[10:50, :21])
    one = w[:, 3]
    assert one.ells == (3, 3) and np.all(one.data == w.data[:, 5:12])
    with pytest.raises(ValueError, match="outside"):
        w[:, 1]
    with pytest.raises(ValueError, match="contiguous"):
        w[:, 2:8:2]


def test_norms():
    W = linear_waveform()
    W.data[0, :] = 6.0 * W.data[-1, :]
    W.data[10, :] = 5.0 * W.data[-1, :]
    assert W.ensure_validity(alter=False)
    assert np.allclose(W.norm(), np.sum(np.abs(W.data) ** 2, axis=-1), rtol=1.0e-15)
    assert np.allclose(W.norm(take_sqrt=True), np.sqrt(np.sum(np.abs(W.data) ** 2, axis=-1)), rtol=1.0e-15)
    q = W.data.shape[0] // 4
    assert np.allclose(W.norm(indices=slice(q, None, None)), np.sum(np.abs(W.data[q:]) ** 2, axis=-1), rtol=1.0e-15)
    assert W.max_norm_index() == W.data.shape[0] - 1
    assert W.max_norm_index(0) == 0 and W.max_norm_index(1) == 0
    assert W.max_norm_index(W.data.shape[0]) == 10
    assert W.max_norm_time() == W.t[-1] and W.max_norm_time(0) == W.t[0] and W.max_norm_time(W.data.shape[0]) == W.t[10]


def test_squad_rotor_interpolation():
    """quaternions.squad (the numpy-quaternion algorithm scri/waveform_base.py:957 calls for the frame of an interpolated
    waveform): identity at the knots, exact for a uniform rotation about a fixed axis (what the reference's
    test_linear_interpolation checks), unit norm, and convergent for a general smooth rotor."""
    from scri_amd import quaternions as Q

    w = linear_waveform()
    assert np.array_equal(Q.squad(w.frame, w.t, w.t), w.frame)
    t_out = (w.t[:-1] + w.t[1:]) / 2.0
    out = Q.squad(w.frame, w.t, t_out)
    axis = w.frame[1, 1:] / np.linalg.norm(w.frame[1, 1:]) * np.sign(np.sin(2 * np.pi * 4 / 110 * w.t[1] / 2))
    omega = 2 * np.pi * 4 / (w.t[-1] - w.t[0])
    exact = np.array([[np.cos(omega * ti / 2), *(np.sin(omega * ti / 2) * axis)] for ti in t_out])
    assert np.abs(out - exact).max() < 4e-15
    assert np.abs(np.linalg.norm(out, axis=-1) - 1).max() < 2e-15

    def rotor(tt):
        v = np.stack([0.3 * np.sin(0.2 * tt), 0.025 * tt, 0.2 * np.cos(0.13 * tt)], axis=-1)
        return Q.exp(np.concatenate([np.zeros(tt.shape + (1,)), v], axis=-1))

    errs = []
    for n in (200, 400):
        tt = np.linspace(0, 50, n)
        tm = (tt[:-1] + tt[1:]) / 2
        errs.append(np.abs(Q.squad(rotor(tt), tt, tm) - rotor(tm)).max())
    assert errs[0] < 1e-4 and errs[1] < errs[0] / 3.5
    # a single rotor or no output times
    assert Q.squad(w.frame[:1], w.t[:1], t_out[:5]).shape == (5, 4)
    assert Q.squad(w.frame, w.t, np.array([])).shape == (0, 4)


def test_SI_units():
    """the reference's tests/test_waveform.py:350-370 on the same linear waveform"""
    import scri_amd

    units_precision = 4.5e-16
    total_mass, distance = 1.1, 3.7
    mass_in_seconds = total_mass * 4.92549094916e-06
    distance_in_meters = distance * 3.0856775814913672789e22
    W_in = linear_waveform()
    W_out = W_in.SI_units(total_mass, distance)
    assert W_in.ensure_validity(alter=False) and W_out.ensure_validity(alter=False)
    assert W_out.history[:1] == ["# Called from linear_waveform"]
    assert W_out.frameType == scri_amd.Corotating and W_out.dataType == scri_amd.h
    assert not W_out.r_is_scaled_out and not W_out.m_is_scaled_out
    assert W_out.num != W_in.num
    assert np.allclose(W_out.t, W_in.t * mass_in_seconds, rtol=units_precision)
    assert np.allclose(W_out.data, W_in.data * mass_in_seconds * 299792458 / distance_in_meters, rtol=units_precision)
    with pytest.warns(UserWarning, match="radius is supposedly not scaled out"):
        W_out.SI_units(total_mass, distance)


def test_descriptors_and_weights():
    """waveform_base.py:440-516: the weights and the file-name descriptor for the combinations of what is scaled out"""
    import scri_amd

    def make(dataType, r, m):
        return scri_amd.WaveformModes(t=np.zeros(1), data=np.zeros((1, 5), dtype=complex), ell_min=2, ell_max=2, dataType=dataType,
                                      r_is_scaled_out=r, m_is_scaled_out=m)

    assert make(scri_amd.psi4, True, True).descriptor_string == "rMPsi4"
    assert make(scri_amd.h, True, True).descriptor_string == "rhOverM"
    assert make(scri_amd.h, False, True).descriptor_string == "h"
    assert make(scri_amd.h, True, False).descriptor_string == "rh"
    assert make(scri_amd.psi0, True, True).descriptor_string == "r5Psi0OverM3"
    assert make(scri_amd.psi2, False, True).descriptor_string == "M2Psi2"
    assert make(scri_amd.news, False, True).descriptor_string == "Mnews"
    assert scri_amd.WaveformModes(t=np.zeros(1), data=np.zeros((1, 5), dtype=complex), ell_min=2, ell_max=2).descriptor_string == "UnknownDataType"
    w = make(scri_amd.psi4, True, True)
    assert (w.r_scaling, w.m_scaling, w.gamma_weight, w.data_type_latex) == (1, 2, 1, r"\psi_4")
    assert make(scri_amd.psi4, True, False).gamma_weight == 0 and make(scri_amd.psi4, False, True).gamma_weight == 2
    assert w.is_valid
    c = w.deepcopy()
    assert c.num != w.num and np.array_equal(c.data, w.data) and c.data is not w.data and c.history[-1].endswith(".deepcopy()")


def test_pickling_and_repr():
    """tests/test_waveform.py:45-53 of the reference (an empty object) and the same for a waveform with data and a frame"""
    import copy
    import pickle

    import scri_amd

    W1 = scri_amd.WaveformModes()
    W2 = pickle.loads(pickle.dumps(W1))
    assert W1._allclose(W2, rtol=0, atol=0) and W2.num != W1.num
    W = linear_waveform(n_times=50)
    for clone in (pickle.loads(pickle.dumps(W)), copy.deepcopy(W), copy.copy(W)):
        assert clone.num != W.num and W._allclose(clone, rtol=0, atol=0)
        assert np.array_equal(clone.data, W.data) and np.array_equal(clone.frame, W.frame) and clone.data is not W.data
        assert (clone.ell_min, clone.ell_max, clone.dataType, clone.frameType) == (W.ell_min, W.ell_max, W.dataType, W.frameType)
        assert clone.history[: len(W.history)] == W.history and "unpickled as" in clone.history[len(W.history)]
    text = repr(W)
    assert text.strip().startswith("WaveformModes(") and f"# num = {W.num}" in text and "frameType=4, dataType=7" in text


def test_waveform_grid_constructor_forms():
    """scri.WaveformGrid takes keywords with defaults, or one object to copy (scri/waveform_base.py:220-258, waveform_grid.py:194-199);
    the tutorial's statement (docs/tutorial_waveformmodes.rst:74-83) word for word"""
    import scri_amd as scri

    my_strain_grid_data = np.zeros((100, 144), dtype=complex)
    h = scri.WaveformGrid(
        dataType=scri.h,
        t=np.linspace(0, 10, 100),
        data=my_strain_grid_data,
        n_theta=12,
        n_phi=12,
        frameType=scri.Inertial,
        r_is_scaled_out=True,
        m_is_scaled_out=True,
    )
    assert (h.n_theta, h.n_phi, h.n_times) == (12, 12, 100) and h.spin_weight == -2
    assert str(h).startswith("WaveformGrid_") and repr(h).endswith("# n_theta=12, n_phi=12")
    empty = scri.WaveformGrid()
    assert empty.n_times == 0 and empty.dataType == scri.UnknownDataType and empty.frameType == scri.UnknownFrameType
    assert not empty.r_is_scaled_out and not empty.m_is_scaled_out and (empty.n_theta, empty.n_phi) == (0, 0)
    copy = scri.WaveformGrid(h)
    assert copy.num != h.num and copy.history[-1] == f"{copy} = WaveformGrid({h})" and copy.history[:-1] == h.history
    with pytest.raises(ValueError, match="objects to be copied must be passed as the sole argument"):
        scri.WaveformGrid(h, my_strain_grid_data)
    with pytest.raises(ValueError, match="does not agree"):
        scri.WaveformGrid(t=np.linspace(0, 1, 5), data=np.zeros((5, 10), dtype=complex), n_theta=3, n_phi=3)
    assert callable(scri.WaveformModes.to_grid) and callable(scri.WaveformModes.from_grid)


def test_the_references_dotted_paths_resolve():
    """scri/asymptotic_bondi_data/ is a package in the reference and the readers live under scri.SpEC; code written against it spells
    the paths out (docs/tutorial_abd.rst:88,340; scri/asymptotic_bondi_data/__init__.py:235-263)"""
    import importlib

    import scri_amd as scri

    for path, names in (
        ("asymptotic_bondi_data", ["AsymptoticBondiData"]),
        ("asymptotic_bondi_data.map_to_superrest_frame", ["map_to_superrest_frame", "MT_to_WM", "WM_to_MT"]),
        ("asymptotic_bondi_data.map_to_abd_frame", ["map_to_abd_frame"]),
        ("asymptotic_bondi_data.bms_charges", ["mass_aspect", "bondi_four_momentum", "supermomentum"]),
        ("asymptotic_bondi_data.from_initial_values", ["from_initial_values"]),
        ("asymptotic_bondi_data.constraints", ["bondi_constraints", "bondi_violation_norms"]),
        ("asymptotic_bondi_data.transformations", ["AsymptoticBondiData"]),
        ("SpEC.file_io", ["create_abd_from_h5"]),
        ("bms_transformations", ["BMSTransformation", "LorentzTransformation"]),
        ("modes_time_series", ["ModesTimeSeries"]),
    ):
        module = importlib.import_module("scri_amd." + path)
        obj = scri
        for part in path.split("."):
            obj = getattr(obj, part)
        assert obj is module, path
        for n in names:
            assert hasattr(module, n), (path, n)
