"""Boosts along the polar axis of the rotated grid keep the separable synthesis: the aberration of
scri/waveform_grid.py:141-161 (== boosted_grid, scri/asymptotic_bondi_data/transformations.py:100-148) rotates every pixel about
r' x v, which for v parallel to the axis lies along the ring's tangent -- the rings move to new colatitudes, nothing else
changes.  The engine checks that form on the rotors themselves and then synthesises ring by ring at
the aberrated colatitudes, with the conformal factor's power applied on the way out of the phi stage.  Checked against the dense
sYlm product of the same library (to rounding) and against the oracle.

The route is taken where it pays (n_modes x n_pix >= 160 000: about l_max >= 13 on the default grids); small shapes are pushed
onto it with SCRI_AMD_AXIS_BOOST_MIN_WORK=0, and the natural-size cases run without the switch."""
import numpy as np
import pytest

from oracle import abd_ref, waveform_grid_ref as grid_ref
from oracle.containers import WM, h as o_h, psi3 as o_psi3, psi4 as o_psi4, sigma as o_sigma
from tests.test_gpu_transform_abd import real_st, smooth_abd

pytestmark = pytest.mark.gpu


def _rotations(ctx):
    """launches of the Wigner rotation since the last call: the separable route rotates the MODES by the frame rotor, the dense
    one never does"""
    return ctx.get_timing(reset=True)["rotate"][1]


def _route_evidence(certain, rotor, n_rot_sep, n_rot_dense, got, ref):
    assert n_rot_dense == 0
    if rotor is None:
        assert n_rot_sep == 0
        if certain:  # no rotation to go by: the two routes round differently, one route twice would be bit-identical
            assert not np.array_equal(got, ref)
    elif certain:
        assert n_rot_sep > 0


def _zrot(a):
    return np.array([np.cos(a / 2), 0.0, 0.0, np.sin(a / 2)])


def _rotate_z(q):
    w, x, y, z = q
    return np.array([2 * (x * z + w * y), 2 * (y * z - w * x), w * w - x * x - y * y + z * z])


GENERAL = np.array([0.4, 1.0, -2.0, 0.3]) / np.linalg.norm([0.4, 1.0, -2.0, 0.3])

# (frame rotation, boost, the separable route is certain).  A general rotor with v = beta F z F^-1 in floating point may or may not
# pass the engine's check of the pole pixels (acos near 1 turns an ulp into 1e-8 rad there, in the reference as well): either
# route has to give the same numbers.
FRAMES = [
    (None, [0.0, 0.0, 0.3], True),
    (None, [0.0, 0.0, -0.45], True),
    (_zrot(0.7), [0.0, 0.0, 0.2], True),
    (np.array([0.0, 1.0, 0.0, 0.0]), [0.0, 0.0, -0.25], True),  # rotation by pi about x: z -> -z exactly
    (GENERAL, 0.3 * _rotate_z(GENERAL), False),
    (GENERAL, -0.15 * _rotate_z(GENERAL), False),
]


@pytest.mark.parametrize("case", range(len(FRAMES)))
@pytest.mark.parametrize("ell_max,n,working,forced", [(4, 120, None, True), (10, 64, None, False), (6, 40, 15, True), (20, 24, None, False)])
def test_abd_axis_boost_separable_equals_dense_and_oracle(ctx, monkeypatch, case, ell_max, n, working, forced, route):
    import scri_amd

    rotor, v, certain = FRAMES[case]
    if ell_max > 10 and case not in (0, 3):
        pytest.skip("large l_max: two frames are enough")
    if forced:
        route("SCRI_AMD_AXIS_BOOST_MIN_WORK", "0")
    else:
        route("SCRI_AMD_AXIS_BOOST_MIN_WORK", None)
    o = smooth_abd(n, ell_max, 300 + ell_max + n)
    kw = dict(supertranslation=real_st(min(ell_max, 3), 7, 0.05), boost_velocity=np.asarray(v, dtype=float))
    if rotor is not None:
        kw["frame_rotation"] = rotor
    if working:
        kw["working_ell_max"] = working

    def run():
        g = scri_amd.AsymptoticBondiData(o.u, o.ell_max, ctx=ctx)
        g._raw_data[:] = o.raw
        return g.transform(**kw)

    route("SCRI_AMD_NO_SEPARABLE_SYNTHESIS", None)
    route("SCRI_AMD_NO_AXIS_BOOST_SEPARABLE", None)
    ctx.enable_timing(True)
    ctx.get_timing(reset=True)
    got = run()
    n_sep = _rotations(ctx)
    route("SCRI_AMD_NO_AXIS_BOOST_SEPARABLE", "1")
    ref = run()
    n_dense = _rotations(ctx)
    ctx.enable_timing(False)
    _route_evidence(certain, rotor, n_sep, n_dense, got._raw_data, ref._raw_data)
    assert got.n_times == ref.n_times > 0 and np.array_equal(got.u, ref.u)
    scale = max(1.0, np.abs(ref._raw_data).max())
    assert np.abs(got._raw_data - ref._raw_data).max() < 1e-13 * scale * max(1.0, ell_max / 8.0)
    if ell_max <= 10:
        expect = abd_ref.transform(o, **kw)
        assert got.n_times == expect.n_times
        assert np.abs(got._raw_data - expect.raw).max() < 1e-12 * scale


@pytest.mark.parametrize("case", range(len(FRAMES)))
@pytest.mark.parametrize("data_type,ell_max,forced", [("h", 8, True), ("sigma", 5, True), ("psi4", 12, True), ("psi3", 6, True), ("h", 16, False),
                                                       ("psi3", 14, False), ("h", 20, False)])
def test_waveform_modes_axis_boost_separable_equals_dense_and_oracle(ctx, monkeypatch, case, data_type, ell_max, forced, route):
    import scri_amd
    from scri_amd import synthetic

    rotor, v, certain = FRAMES[case]
    if ell_max > 12 and case not in (1, 2):
        pytest.skip("large l_max: two frames are enough")
    if forced:
        route("SCRI_AMD_AXIS_BOOST_MIN_WORK", "0")
    else:
        route("SCRI_AMD_AXIS_BOOST_MIN_WORK", None)
    n = 150
    t = np.linspace(-30.0, 40.0, n)
    spins = {"psi3": -1, "psi4": -2, "h": -2, "sigma": 2}
    lmin = abs(spins[data_type])
    rng = np.random.default_rng(ell_max + case)
    data = synthetic.chirp_modes(t, lmin, ell_max, 3 + ell_max)
    st = synthetic.real_supertranslation(0.2 * (rng.normal(size=16) + 1j * rng.normal(size=16)))
    kw = dict(supertranslation=st, boost_velocity=np.asarray(v, dtype=float))
    if rotor is not None:
        kw["frame_rotation"] = rotor
    aux = dict(psi4_modes=synthetic.chirp_modes(t, 2, ell_max, 12)) if data_type == "psi3" else {}

    def wrap(name, d, lo):
        return scri_amd.WaveformModes(t=t, data=d, ell_min=lo, ell_max=ell_max, dataType=getattr(scri_amd, name), frameType=scri_amd.Inertial,
                                      r_is_scaled_out=True, m_is_scaled_out=True, ctx=ctx)

    def run():
        w = wrap(data_type, data, lmin)
        extra = {k: wrap(k[:4], v_, abs(spins[k[:4]])) for k, v_ in aux.items()}
        return w.transform(**kw, **extra)

    route("SCRI_AMD_NO_SEPARABLE_SYNTHESIS", None)
    route("SCRI_AMD_NO_AXIS_BOOST_SEPARABLE", None)
    ctx.enable_timing(True)
    ctx.get_timing(reset=True)
    got = run()
    n_sep = _rotations(ctx)
    route("SCRI_AMD_NO_AXIS_BOOST_SEPARABLE", "1")
    ref = run()
    n_dense = _rotations(ctx)
    ctx.enable_timing(False)
    _route_evidence(certain, rotor, n_sep, n_dense, got.data, ref.data)
    scale = max(1.0, np.abs(ref.data).max())
    assert got.n_times == ref.n_times > 0 and np.array_equal(got.t, ref.t)
    assert np.abs(got.data - ref.data).max() < 2e-13 * scale * max(1.0, ell_max / 8.0)
    if ell_max <= 8:
        otypes = {"psi3": o_psi3, "h": o_h, "psi4": o_psi4, "sigma": o_sigma}
        ow = WM(t=t, data=data, ell_min=lmin, ell_max=ell_max, dataType=otypes[data_type])
        oaux = {k: WM(t=t, data=v_, ell_min=abs(spins[k[:4]]), ell_max=ell_max, dataType=otypes[k[:4]]) for k, v_ in aux.items()}
        expect = grid_ref.transform(ow, **kw, **oaux)
        assert got.n_times == expect.t.size
        assert np.abs(got.data - expect.data).max() < 1e-12 * scale


def test_oblique_boost_keeps_the_dense_route(ctx, monkeypatch, route):
    """A boost a hair off the axis is not separable: the engine's check has to send it to the dense product."""
    import scri_amd

    o = smooth_abd(40, 4, 5)
    route("SCRI_AMD_AXIS_BOOST_MIN_WORK", "0")
    route("SCRI_AMD_NO_SEPARABLE_SYNTHESIS", None)
    route("SCRI_AMD_NO_AXIS_BOOST_SEPARABLE", None)
    g = scri_amd.AsymptoticBondiData(o.u, o.ell_max, ctx=ctx)
    g._raw_data[:] = o.raw
    kw = dict(boost_velocity=[1e-9, 0.0, 0.3], frame_rotation=_zrot(0.4))
    ctx.enable_timing(True)
    ctx.get_timing(reset=True)
    got = g.transform(**kw)
    assert _rotations(ctx) == 0
    ctx.enable_timing(False)
    expect = abd_ref.transform(o, **kw)
    assert np.abs(got._raw_data - expect.raw).max() < 1e-12 * max(1.0, np.abs(expect.raw).max())


def test_small_shapes_keep_the_dense_product(ctx, monkeypatch, route):
    """Below the break-even (l <= 8 on 17 x 17) an axis boost stays on the dense product, which is faster there."""
    import scri_amd
    from scri_amd import synthetic

    route("SCRI_AMD_AXIS_BOOST_MIN_WORK", None)
    route("SCRI_AMD_NO_SEPARABLE_SYNTHESIS", None)
    route("SCRI_AMD_NO_AXIS_BOOST_SEPARABLE", None)
    t = np.linspace(-30.0, 40.0, 100)
    w = scri_amd.WaveformModes(t=t, data=synthetic.chirp_modes(t, 2, 8, 4), ell_min=2, ell_max=8, dataType=scri_amd.h, frameType=scri_amd.Inertial,
                               r_is_scaled_out=True, m_is_scaled_out=True, ctx=ctx)
    ctx.enable_timing(True)
    ctx.get_timing(reset=True)
    w.transform(frame_rotation=_zrot(0.3), boost_velocity=[0.0, 0.0, 0.1])
    assert _rotations(ctx) == 0
    ctx.enable_timing(False)


def test_axis_boost_time_shards_and_pipelined_pieces(ctx, monkeypatch, route):
    """The route under the time shards of the multi-GPU split and under the pipelined pieces of a host caller (every piece builds the
    ring tables of the same transformation): both reassemble to the one-call result."""
    import scri_amd
    from scri_amd import engine, sharding, synthetic

    route("SCRI_AMD_AXIS_BOOST_MIN_WORK", None)
    route("SCRI_AMD_NO_SEPARABLE_SYNTHESIS", None)
    route("SCRI_AMD_NO_AXIS_BOOST_SEPARABLE", None)
    n, ell_max = 5000, 14
    t = np.linspace(0.0, 500.0, n)
    data = synthetic.chirp_modes(t, 2, ell_max, 9)
    st = synthetic.real_supertranslation(0.2 * np.random.default_rng(2).normal(size=9) + 0j)
    n_theta = 2 * ell_max + 1
    tr = engine.make_transformation(st, _zrot(1.1), [0.0, 0.0, -0.02], n_theta, n_theta, ell_max)
    ctx.enable_timing(True)
    ctx.get_timing(reset=True)
    t_ref, d_ref = engine.transform_modes(t, data, 2, ell_max, -2, -1, engine.BMS_TERM_H, tr, ctx=ctx)
    assert _rotations(ctx) > 0
    ctx.enable_timing(False)
    have, need, window = sharding.plan(t, tr, 3)
    ts, ds = [], []
    for r in range(3):
        ext = data[need[r][0] : need[r][1]]
        to, do, first = engine.transform_modes(t, ext, 2, ell_max, -2, -1, engine.BMS_TERM_H, tr, ctx=ctx, shard=(need[r][0], ext.shape[0], have[r][0], have[r][1]))
        ts.append(to), ds.append(do)
    assert np.array_equal(np.concatenate(ts), t_ref)
    assert np.abs(np.concatenate(ds) - d_ref).max() < 1e-13 * max(1.0, np.abs(d_ref).max())

    kw = dict(supertranslation=st, frame_rotation=_zrot(1.1), boost_velocity=[0.0, 0.0, -0.02])

    def run():
        w = scri_amd.WaveformModes(t=t, data=data, ell_min=2, ell_max=ell_max, dataType=scri_amd.h, frameType=scri_amd.Inertial,
                                   r_is_scaled_out=True, m_is_scaled_out=True, ctx=ctx)
        return w.transform(**kw)

    monkeypatch.setenv("SCRI_AMD_NO_PIPELINE", "1")
    plain = run()
    monkeypatch.delenv("SCRI_AMD_NO_PIPELINE")
    calls = []
    real = engine._transform_modes_pipelined
    monkeypatch.setattr(engine, "_transform_modes_pipelined", lambda *a, **k: calls.append(1) or real(*a, **k))
    monkeypatch.setattr(engine, "PIPELINE_MIN_BYTES", 1 << 16)
    piped = run()
    assert calls == [1] and np.array_equal(piped.t, plain.t)
    assert np.abs(piped.data - plain.data).max() < 1e-13 * max(1.0, np.abs(plain.data).max())


def _axis_kwargs(rng, ell_max):
    """tests/test_gpu_fuzz.py's transformation generator with the boost pinned to the axis of the rotated grid"""
    from tests.test_gpu_transform_modes import real_supertranslation

    kw = {}
    if rng.random() < 0.8:
        kw["supertranslation"] = real_supertranslation(int(rng.integers(1, 4)), int(rng.integers(1 << 30)), 10.0 ** rng.uniform(-2.5, -0.7))
    pick = rng.integers(0, 4)
    if pick == 1:
        kw["frame_rotation"] = _zrot(rng.uniform(-3.0, 3.0))
    elif pick == 2:
        kw["frame_rotation"] = np.array([0.0, 1.0, 0.0, 0.0])
    elif pick == 3:
        kw["frame_rotation"] = np.array([0.0, 0.0, 1.0, 0.0])  # pi about y: z -> -z as well
    kw["boost_velocity"] = np.array([0.0, 0.0, 1.0]) * rng.choice([-1.0, 1.0]) * 10.0 ** rng.uniform(-4, np.log10(0.3))
    return kw


@pytest.mark.parametrize("seed", range(24))
def test_random_axis_boost_waveform_transform_against_the_oracle(ctx, monkeypatch, seed, route):
    """The seeded sweep of tests/test_gpu_fuzz.py (data types with their mixing terms, l ranges, grids, jittered time axes) with
    every boost along the grid's axis, all shapes pushed onto the separable route."""
    import tests.test_gpu_fuzz as fuzz

    route("SCRI_AMD_AXIS_BOOST_MIN_WORK", "0")
    route("SCRI_AMD_NO_SEPARABLE_SYNTHESIS", None)
    route("SCRI_AMD_NO_AXIS_BOOST_SEPARABLE", None)
    monkeypatch.setattr(fuzz, "_random_kwargs", _axis_kwargs)
    fuzz.test_random_waveform_transform(ctx, seed)


@pytest.mark.parametrize("seed", range(8))
def test_random_axis_boost_abd_transform_against_the_oracle(ctx, monkeypatch, seed, route):
    import tests.test_gpu_fuzz as fuzz

    route("SCRI_AMD_AXIS_BOOST_MIN_WORK", "0")
    route("SCRI_AMD_NO_SEPARABLE_SYNTHESIS", None)
    route("SCRI_AMD_NO_AXIS_BOOST_SEPARABLE", None)
    monkeypatch.setattr(fuzz, "_random_kwargs", _axis_kwargs)
    fuzz.test_random_abd_transform(ctx, seed)


def test_ring_tables_follow_the_transformation(ctx, monkeypatch, route):
    """The ring tables are kept per shape together with the colatitudes they were built for: a second boost of another size (or
    sign) on the same shape must rebuild them, a repeat of the first must find them again."""
    import scri_amd
    from scri_amd import synthetic

    route("SCRI_AMD_AXIS_BOOST_MIN_WORK", "0")
    route("SCRI_AMD_NO_SEPARABLE_SYNTHESIS", None)
    t = np.linspace(-30.0, 40.0, 120)
    data = synthetic.chirp_modes(t, 2, 10, 21)

    def run(v):
        w = scri_amd.WaveformModes(t=t, data=data, ell_min=2, ell_max=10, dataType=scri_amd.h, frameType=scri_amd.Inertial,
                                   r_is_scaled_out=True, m_is_scaled_out=True, ctx=ctx)
        return w.transform(boost_velocity=[0.0, 0.0, v], frame_rotation=_zrot(0.2))

    ref = {}
    route("SCRI_AMD_NO_AXIS_BOOST_SEPARABLE", "1")
    for v in (0.3, -0.45, 0.05):
        ref[v] = run(v)
    route("SCRI_AMD_NO_AXIS_BOOST_SEPARABLE", None)
    for v in (0.3, -0.45, 0.3, 0.05, 0.05, -0.45):
        got = run(v)
        assert got.n_times == ref[v].n_times > 0
        assert np.abs(got.data - ref[v].data).max() < 2e-13 * max(1.0, np.abs(ref[v].data).max()), v


def test_axis_boost_abd_pipelined_host_path(ctx, monkeypatch, route):
    """AsymptoticBondiData from host memory through the three-stream pipeline (bms_transform_abd_pipelined): every piece takes the
    separable route with the ring tables of the one transformation."""
    from scri_amd import engine

    route("SCRI_AMD_AXIS_BOOST_MIN_WORK", "0")
    route("SCRI_AMD_NO_SEPARABLE_SYNTHESIS", None)
    route("SCRI_AMD_NO_AXIS_BOOST_SEPARABLE", None)
    n, ell_max = 3000, 4
    o = smooth_abd(n, ell_max, 77, t0=0.0, t1=600.0)
    n_theta = 2 * (2 * ell_max) + 1
    tr = engine.make_transformation(np.array([0.3, 0, 0.02, 0], dtype=complex), _zrot(0.6), [0.0, 0.0, 0.01], n_theta, n_theta, ell_max)
    raw = np.ascontiguousarray(o.raw)
    monkeypatch.setenv("SCRI_AMD_NO_PIPELINE", "1")
    ctx.enable_timing(True)
    ctx.get_timing(reset=True)
    u_ref, d_ref = engine.transform_abd(o.u, raw, ell_max, tr, ctx=ctx)
    assert _rotations(ctx) > 0
    ctx.enable_timing(False)
    monkeypatch.delenv("SCRI_AMD_NO_PIPELINE")
    monkeypatch.setattr(engine, "PIPELINE_MIN_BYTES", 1 << 16)
    u_got, d_got = engine.transform_abd(o.u, raw, ell_max, tr, ctx=ctx)
    assert np.array_equal(u_got, u_ref) and d_got.shape == d_ref.shape
    assert np.abs(d_got - d_ref).max() < 1e-13 * max(1.0, np.abs(d_ref).max())
