"""Host-side logic of the shim (no GPU): kwargs parsing mirrors scri/waveform_grid.py:20-127 and
scri/asymptotic_bondi_data/transformations.py:8-97 (precedence, error types), rotor grid, shard planning."""
import math
import warnings

import numpy as np
import pytest

from oracle import waveform_grid_ref as grid_ref, abd_ref
from scri_amd import engine, synthetic, sharding
from scri_amd import waveform_grid as wg
from scri_amd import asymptotic_bondi_data as abd_mod


def test_kwargs_precedence_matches_oracle():
    st = synthetic.real_supertranslation(np.arange(9) * (0.1 + 0.05j))
    cases = [
        dict(supertranslation=st),
        dict(supertranslation=st, time_translation=0.7),
        dict(supertranslation=st, space_translation=[0.1, 0.2, -0.3]),
        dict(supertranslation=st, spacetime_translation=[0.5, 0.1, 0.2, 0.3], time_translation=-1.0),
        dict(space_translation=[1.0, 0.0, 0.0], frame_rotation=[1, 2, 3, 4], boost_velocity=[0.1, 0.0, -0.2]),
    ]
    for kw in cases:
        a = wg.process_transformation_kwargs(8, **{k: (np.array(v) if isinstance(v, list) else v) for k, v in kw.items()})
        b = grid_ref.process_transformation_kwargs(8, **{k: (np.array(v) if isinstance(v, list) else v) for k, v in kw.items()})
        assert np.array_equal(a[0], b[0])  # supertranslation modes
        assert a[1:5] == b[1:5]  # ell_max_supertranslation, ell_max, n_theta, n_phi
        assert np.allclose(a[6], b[5])  # boost velocity
        fr = np.array(kw.get("frame_rotation", [1, 0, 0, 0]), dtype=float)
        assert np.allclose(a[5], fr / np.linalg.norm(fr), atol=1e-16)


def test_kwargs_errors_like_reference():
    with pytest.raises(ValueError, match="perfect square"):
        wg.process_transformation_kwargs(8, supertranslation=np.zeros(7, dtype=complex))
    with pytest.raises(ValueError, match="imaginary supertranslation"):
        wg.process_transformation_kwargs(8, supertranslation=np.array([0, 1.0, 0, 0.5], dtype=complex))
    with pytest.raises(TypeError, match="time_translation"):
        wg.process_transformation_kwargs(8, time_translation=1)
    with pytest.raises(TypeError, match="space_translation"):
        wg.process_transformation_kwargs(8, space_translation=[1.0, 2.0])
    with pytest.raises(TypeError, match="spacetime_translation"):
        wg.process_transformation_kwargs(8, spacetime_translation=[1.0, 2.0, 3.0])
    with pytest.raises(ValueError, match="n_theta=5 is too small"):
        wg.process_transformation_kwargs(8, n_theta=5)
    with pytest.raises(ValueError, match="n_phi=5 is too small"):
        wg.process_transformation_kwargs(8, n_phi=5)
    with pytest.raises(ValueError, match="unit quaternion"):
        wg.process_transformation_kwargs(8, frame_rotation=[0, 0, 0, 0])
    with pytest.raises(ValueError, match="strictly less than 1.0"):
        wg.process_transformation_kwargs(8, boost_velocity=[1.0, 0.0, 0.0])
    with pytest.warns(UserWarning, match="n_theta=17 is small"):
        wg.process_transformation_kwargs(8, space_translation=[1.0, 0, 0], n_theta=17)


def test_abd_kwargs_impose_reality_and_defaults():
    st = np.array([1.0, 2 + 4j, 3, -2 + 4j, 7 - 5j, -3 - 2j, 4, 3 - 2j, 7 + 5j]) * 1e-3
    a = abd_mod._process_transformation_kwargs(8, supertranslation=st, boost_velocity=[0.01, 0, 0])
    b = abd_ref.process_transformation_kwargs(8, supertranslation=st, boost_velocity=np.array([0.01, 0, 0]))
    assert np.array_equal(a[2], b[2]) and a[3] == b[3] == 2 * 8 + 2 and a[4] == b[4] == 8
    with pytest.raises(ValueError, match="working_ell_max=3 is too small"):
        abd_mod._process_transformation_kwargs(8, working_ell_max=3)
    with pytest.raises(ValueError, match="strictly less than 1.0"):
        abd_mod._process_transformation_kwargs(8, boost_velocity=[0.8, 0.8, 0.0])


@pytest.mark.parametrize("fr,v", [([1, 0, 0, 0], [0, 0, 0]), ([1, 2, 3, 4], [0.01, -0.02, 0.03]), ([0.3, -1, 0.2, 0.5], [0, 0, 0.1]),
                                  ([1, 0, 0, 0], [0.3, 0, 0]), ([0, 1, 0, 0], [0, 0.05, 0])])
def test_rotor_grid_matches_oracle(fr, v):
    """bms_rotor_grid is host-only set-up code of the library: waveform_grid.py:130-174 / transformations.py:100-148."""
    fr = np.array(fr, dtype=float) / np.linalg.norm(fr)
    R = engine.rotor_grid(fr, v, 9, 11)
    Ro = grid_ref.rotor_grid(fr, np.array(v, dtype=float), 9, 11)
    assert np.abs(R - Ro).max() < 2e-15


def test_shard_plan_covers_skew_and_halo():
    t, _, spec = synthetic.workload("cfg3", n_times=20000)
    kw = spec["kwargs"]
    tr = engine.make_transformation(kw["supertranslation"], kw["frame_rotation"], kw["boost_velocity"], 37, 37, 16)
    have, need, window = sharding.plan(t, tr, 4)
    assert have[0][0] == 0 and have[-1][1] == 20000 and all(have[i][1] == have[i + 1][0] for i in range(3))
    assert 0 <= window[0] < 5 and 19990 < window[1] <= 20000
    for (h0, h1), (n0, n1) in zip(have, need):
        assert n0 <= h0 and n1 >= h1 or (n0 == 0 or n1 == 20000)
        assert h0 - n0 >= 32 or n0 == 0  # spline halo on the left
        assert n1 - h1 >= 32 or n1 == 20000
    # a large boost needs a wide halo (skew ~ beta * |u|max / dt)
    tr2 = engine.make_transformation(np.zeros(4, dtype=complex), [1, 0, 0, 0], [0.0, 0.0, 0.01], 35, 35, 16)
    (n0, n1), _ = engine.shard_plan(t, tr2, 10000, 15000)
    assert 10000 - n0 > 90 and n1 - 15000 > 140


def test_shard_bounds_partition():
    for n, w in ((10, 3), (100000, 8), (7, 8)):
        b = [sharding.shard_bounds(n, w, r) for r in range(w)]
        assert b[0][0] == 0 and b[-1][1] == n
        assert all(b[i][1] == b[i + 1][0] for i in range(w - 1))
        assert max(x[1] - x[0] for x in b) - min(x[1] - x[0] for x in b) <= 1


def test_waveform_modes_container_checks():
    import scri_amd

    t = np.linspace(0, 1, 10)
    w = scri_amd.WaveformModes(t=t, data=np.zeros((10, 21), dtype=complex), ell_min=2, ell_max=4, dataType=scri_amd.h,
                               frameType=scri_amd.Corotating, r_is_scaled_out=True)
    assert w.spin_weight == -2 and w.conformal_weight == -1 and w.n_times == 10 and w.ensure_validity()
    with pytest.raises(ValueError, match="inertial frame"):
        w.transform(time_translation=0.5)
    with pytest.raises(TypeError, match="Expected WaveformModes"):
        wg.transform(np.zeros(3))
    with pytest.raises(ValueError):
        scri_amd.WaveformModes(t=t, data=np.zeros((10, 20), dtype=complex), ell_min=2, ell_max=4)
    w2 = scri_amd.WaveformModes(t=t, data=np.zeros((10, 21), dtype=complex), ell_min=2, ell_max=4, dataType=scri_amd.psi2,
                                frameType=scri_amd.Inertial)
    with pytest.raises(ValueError, match="requires information from Psi3"):
        w2.transform(space_translation=[1.0, 0.0, 0.0])
    with pytest.raises(ValueError, match="Input dimension mismatch"):
        w.rotate_decomposition_basis(np.zeros((3, 4)) + [1, 0, 0, 0])


def test_boosted_grid_and_conformal_factors_match_the_oracle():
    """a7 / a8 of the scope table as Python-visible functions (transformations.py:100-196): host evaluation, no GPU."""
    from oracle import abd_ref
    from scri_amd import asymptotic_bondi_data as A

    v = np.array([0.03, -0.2, 0.1])
    q = np.array([1.0, 2, 3, 4]) / np.sqrt(30)
    R = A.boosted_grid(q, v, 11, 13)
    R_ref = abd_ref.boosted_grid(q, v, 11, 13)
    assert R.shape == (11, 13, 4) and np.abs(R - R_ref).max() < 1e-14
    for got, expect in zip(A.conformal_factors(v, R), abd_ref.conformal_factors(v, R_ref)):
        assert got.shape == (1, 11, 13) and np.abs(got - expect).max() < 1e-14


def test_register_if_reused_counts_objects_not_addresses(monkeypatch):
    """Sightings of an input array are counted per live OBJECT (scri_amd/_lib.py): a recycled address, a recycled id() or the
    same object with other memory do not add up to "seen twice"; the library calls are replaced by recorders here."""
    import gc

    from scri_amd import _lib

    calls = []

    class FakeLib:
        def bms_host_register(self, p, n):
            calls.append(("reg", p.value, n))
            return 0

        def bms_host_unregister(self, p):
            calls.append(("unreg", p.value))
            return 0

    monkeypatch.setattr(_lib, "load", lambda: FakeLib())
    monkeypatch.setattr(_lib, "REGISTER_MIN_BYTES", 1024)
    monkeypatch.delenv("SCRI_AMD_NO_REGISTER", raising=False)
    _lib._seen_inputs.clear()
    a = np.zeros(4096)
    assert _lib.register_if_reused(a) is False and calls == []
    assert _lib.register_if_reused(a) is True and calls == [("reg", a.ctypes.data, a.nbytes)]
    assert _lib.register_if_reused(a) is True and len(calls) == 1  # already page-locked
    assert _lib.register_if_reused(a[:2048]) is False  # a view with another extent of a registered owner: stale range released
    assert calls[-1] == ("unreg", a.ctypes.data)
    # fresh objects every time: never registered, whatever addresses they get
    n_before = len(calls)
    for _ in range(4):
        b = np.zeros(4096)
        assert _lib.register_if_reused(b) is False
        del b
    assert len(calls) == n_before
    # release with the owner
    c = np.ones(4096)
    _lib.register_if_reused(c)
    _lib.register_if_reused(c)
    addr = c.ctypes.data
    del c
    gc.collect()
    assert calls[-1] == ("unreg", addr) and not _lib._registered


def _zrot(a):
    return np.array([np.cos(a / 2), 0.0, 0.0, np.sin(a / 2)])


@pytest.mark.parametrize("fr,v,keeps", [
    ([1.0, 0, 0, 0], [0, 0, 0], True),
    ([0.5, -0.5, 0.5, 0.5], [0, 0, 0], True),
    ([1.0, 0, 0, 0], [0, 0, 0.3], True),
    ([1.0, 0, 0, 0], [0, 0, -0.45], True),
    (_zrot(0.7), [0, 0, 0.2], True),
    ([0.0, 1.0, 0, 0], [0, 0, -0.25], True),      # pi about x: the grid's axis is -z
    ([1.0, 0, 0, 0], [1e-9, 0, 0.3], False),      # a hair off the axis
    ([0.5, -0.5, 0.5, 0.5], [0, 0, 0.1], False),  # frame axis = x
    ([1.0, 0, 0, 0], [0.1, 0, 0], False),
])
def test_ring_colatitudes_against_the_oracle_rotor_grid(fr, v, keeps):
    """bms_ring_colatitudes (host): the rotor grid of scri/waveform_grid.py:130-174 keeps its rings without a boost and with one along
    the polar axis of the rotated grid; the colatitudes it returns rebuild the oracle's rotors as frame_rotation * R(Theta_j, phi_k)."""
    from oracle import quat
    from scri_amd import engine

    n_theta, n_phi = 11, 9
    fr = np.asarray(fr, dtype=float)
    th = engine.ring_colatitudes(fr, v, n_theta, n_phi)
    assert (th is not None) == keeps
    if not keeps:
        return
    R = grid_ref.rotor_grid(fr, np.asarray(v, dtype=float), n_theta, n_phi)
    for j in range(n_theta):
        for k in range(n_phi):
            E = quat.qmul(fr, quat.from_spherical_coords(th[j], 2 * np.pi * k / n_phi))
            G = np.asarray(R[j, k], dtype=float)
            assert min(np.abs(G - E).max(), np.abs(G + E).max()) < 1e-14, (j, k)
    # aberration formula: tan(Theta / 2) = exp(-+rapidity) tan(theta' / 2) along +-z'
    beta = float(np.linalg.norm(v))
    if beta:
        sign = np.sign(np.dot(quat.rotate_z(fr), v))
        thp = np.linspace(0.0, np.pi, n_theta)
        expect = 2 * np.arctan(np.exp(-sign * np.arctanh(beta)) * np.tan(thp[1:-1] / 2))
        assert np.abs(th[1:-1] - expect).max() < 1e-14
    with pytest.raises(ValueError):
        engine.ring_colatitudes(fr, v, 1, n_phi)


def test_time_intersection_matches_the_oracle():
    """scri_amd.mode_operators.time_intersection (what inner_product(..., allow_times_differ=True) interpolates to) against the
    oracle's restatement of scri/extrapolation.py:47-122, on uniform, jittered and offset axes and with the optional bounds."""
    from oracle import waveform_modes_ref as ref
    from scri_amd.mode_operators import time_intersection

    rng = np.random.default_rng(11)
    for case in range(12):
        n1, n2 = int(rng.integers(5, 80)), int(rng.integers(5, 80))
        t1 = np.sort(rng.uniform(0.0, 10.0, n1)) if case % 2 else np.linspace(0.0, 10.0, n1)
        t2 = np.sort(rng.uniform(-1.0, 12.0, n2)) if case % 3 else np.linspace(0.5, 9.0, n2)
        t1[1:] = np.maximum(t1[1:], t1[:-1] + 1e-3)
        t2[1:] = np.maximum(t2[1:], t2[:-1] + 1e-3)
        kw = {}
        if case % 4 == 1:
            kw = dict(min_step=0.05)
        if case % 4 == 2:
            kw = dict(min_time=1.0, max_time=8.0)
        got, expect = time_intersection(t1, t2, **kw), ref.intersection(t1, t2, **kw)
        assert np.array_equal(got, expect), case
        assert got[0] >= max(t1[0], t2[0]) and got[-1] <= min(t1[-1], t2[-1]) and (np.diff(got) > 0).all()
    with pytest.raises(ValueError):
        time_intersection(np.array([]), np.array([0.0, 1.0]))
    with pytest.raises(ValueError, match="Empty intersection"):
        time_intersection(np.array([0.0, 1.0]), np.array([2.0, 3.0]))
