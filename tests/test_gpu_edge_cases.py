"""Edge cases of the GPU paths: non-uniform time grids, tiny series, user grid sizes, every analysis variant
(fused kernel / phi-DFT GEMM + theta quadrature / dense quadrature GEMM), large boosts (wide time skew), large l in the
rotation (MFMA and VALU kernels), strided inputs."""
import numpy as np
import pytest
from scipy.interpolate import CubicSpline

from oracle import quat, wigner, rotations_ref, spinsfast_ref, abd_ref
from oracle import waveform_grid_ref as grid_ref
from oracle.containers import WM, ABD, h, psi4

pytestmark = pytest.mark.gpu


def _wm(t, ell_max, seed, dataType=h):
    rng = np.random.default_rng(seed)
    LM = wigner.LM_range(2, ell_max)
    a = (rng.normal(size=LM.shape[0]) + 1j * rng.normal(size=LM.shape[0])) * 10.0 ** (-LM[:, 0] / 4.0)
    ph = 0.05 * t + 1e-4 * t**2
    return WM(t=t, data=a[None, :] * np.exp(1j * LM[None, :, 1] * ph[:, None]), ell_min=2, ell_max=ell_max, dataType=dataType)


def _gpu(w, ctx):
    import scri_amd

    return scri_amd.WaveformModes(t=w.t, data=w.data, ell_min=w.ell_min, ell_max=w.ell_max, dataType=w.dataType, frameType=1,
                                  r_is_scaled_out=True, m_is_scaled_out=True, ctx=ctx)


def _check(w, ctx, tol=1e-12, **kw):
    e = grid_ref.transform(w, **kw)
    g = _gpu(w, ctx).transform(**kw)
    assert g.t.shape == e.t.shape and np.abs(g.t - e.t).max() < 1e-12
    assert np.abs(g.data - e.data).max() < tol * max(1.0, np.abs(e.data).max())


def test_nonuniform_time_grid(ctx):
    rng = np.random.default_rng(1)
    t = np.cumsum(rng.uniform(0.02, 0.3, size=700)) - 30.0  # step ratio up to 15
    _check(_wm(t, 5, 2), ctx, space_translation=np.array([0.2, -0.3, 0.1]), boost_velocity=np.array([0.002, 0.001, -0.003]))


@pytest.mark.parametrize("n", [4, 5, 7, 33])
def test_tiny_series(ctx, n):
    t = np.linspace(0.0, 1.0, n)
    _check(_wm(t, 3, 3), ctx, time_translation=0.0)
    _check(_wm(t, 3, 3), ctx, frame_rotation=np.array([1.0, 0.3, -0.2, 0.5]))


def test_series_shorter_than_4_is_rejected(ctx):
    with pytest.raises(ValueError, match="at least 4 time steps"):
        _gpu(_wm(np.linspace(0, 1, 3), 3, 3), ctx).transform(time_translation=0.1)


@pytest.mark.parametrize("n_theta,n_phi", [(15, 15), (21, 17), (15, 33), (45, 47)])
def test_user_grid_sizes_and_analysis_variants(ctx, n_theta, n_phi):
    """(45, 47) exceeds the fused kernel's limits -> phi-DFT GEMM + theta_quadrature_kernel."""
    t = np.linspace(-5, 25, 200)
    _check(_wm(t, 6, 4), ctx, supertranslation=np.array([0.0, 0.01 - 0.02j, 0.03, -0.01 - 0.02j]), n_theta=n_theta, n_phi=n_phi, ell_max=5)


def test_dense_quadrature_path_for_very_fine_grids(ctx):
    """n_theta > 104 uses the dense quadrature GEMM (engine build_analysis)."""
    from scri_amd import engine

    rng = np.random.default_rng(5)
    f = rng.normal(size=(3, 107, 9)) + 1j * rng.normal(size=(3, 107, 9))
    assert np.abs(engine.map2salm(f, -1, 4, ctx=ctx) - spinsfast_ref.map2salm(f, -1, 4)).max() < 1e-12


@pytest.mark.parametrize("n_theta,n_phi,spin,ell_max,ell_min", [(45, 47, -2, 9, 2), (99, 99, 2, 24, 0), (64, 50, 0, 16, 0), (41, 96, -1, 20, 1), (104, 104, 1, 32, 0), (57, 60, -2, 3, 2)])
def test_large_grid_analysis_variants(ctx, monkeypatch, n_theta, n_phi, spin, ell_max, ell_min, route):
    """40 < n_theta <= 104: folded phi-DFT kernel + MFMA theta quadrature (odd and even rings, every template branch);
    the older phi-DFT GEMM + theta_quadrature_kernel pair stays reachable and must agree."""
    from scri_amd import engine

    rng = np.random.default_rng(n_theta + n_phi)
    f = rng.normal(size=(21, n_theta, n_phi)) + 1j * rng.normal(size=(21, n_theta, n_phi))
    ref = spinsfast_ref.map2salm(f, spin, ell_max)[..., ell_min**2 :]
    got = engine.map2salm(f, spin, ell_max, ell_min=ell_min, ctx=ctx)
    assert np.abs(got - ref).max() < 2e-13 * max(1.0, np.abs(ref).max())
    route("SCRI_AMD_NO_LARGE_ANALYSIS", "1")
    old = engine.map2salm(f, spin, ell_max, ell_min=ell_min, ctx=ctx)
    assert np.abs(old - ref).max() < 2e-13 * max(1.0, np.abs(ref).max())


@pytest.mark.parametrize("n_theta,n_phi", [(20, 24), (31, 38), (19, 18), (39, 39), (17, 34)])
def test_separable_synthesis_on_user_grids(ctx, n_theta, n_phi, monkeypatch, route):
    """Boost-free transformations on caller-chosen grids, even n_phi included (a Nyquist column that is its own mirror image,
    side columns k = 17..19 of the 4x4x4 product) against the oracle."""
    route("SCRI_AMD_NO_SMALL_DENSE", "1")  # (shapes this small take the evaluating product by default: the separable kernels are meant)
    t = np.linspace(-5, 25, 180)
    _check(_wm(t, 6, 13), ctx, n_theta=n_theta, n_phi=n_phi, supertranslation=np.array([0.3, 0.1 - 0.2j, 0.15, -0.1 - 0.2j]),
           frame_rotation=np.array([0.8, -0.3, 0.4, 0.2]))
    _check(_wm(t, 8, 14, dataType=psi4), ctx, n_theta=n_theta, n_phi=n_phi, time_translation=0.7)


@pytest.mark.parametrize(
    "n_theta,n_phi,spin,ell_max,ell_min,n_rows",
    [
        (37, 37, -2, 16, 2, 11),  # cfg3's grid: 5 front + 5 back waves
        (37, 37, -2, 4, 2, 6),    # 5 front waves, ONE back wave (fewer than 8 waves in the workgroup)
        (40, 39, 0, 3, 0, 5),     # 5 front waves, one back wave, n_theta at the limit
        (33, 33, 0, 16, 0, 4),    # spin 0: 289 modes
        (21, 21, -2, 8, 2, 9),    # cfg2's grid: 3 + 2 waves, several workgroups per CU
        (17, 20, -1, 8, 1, 3),    # even n_phi (a Nyquist sample that is its own partner), odd number of rows
        (9, 9, 2, 4, 2, 2),       # two rows: one pair
        (25, 27, 1, 12, 1, 1001),
        (3, 5, 0, 1, 0, 7),       # the smallest grid the kernel takes
    ],
)
def test_fused_analysis_two_role_kernel_shapes(ctx, monkeypatch, n_theta, n_phi, spin, ell_max, ell_min, n_rows, route):
    """n_theta <= 40, l_max <= 16: `analysis_split_kernel` (front waves: fold + MFMA, back waves: theta quadrature) for
    every combination of wave counts; the one-role kernel (SCRI_AMD_NO_SPLIT_ANALYSIS) stays reachable and must agree;
    repeated calls are bitwise equal."""
    from scri_amd import engine

    rng = np.random.default_rng(n_theta * n_phi + n_rows)
    f = rng.normal(size=(n_rows, n_theta, n_phi)) + 1j * rng.normal(size=(n_rows, n_theta, n_phi))
    ref = spinsfast_ref.map2salm(f, spin, ell_max)[..., ell_min**2 :]
    got = engine.map2salm(f, spin, ell_max, ell_min=ell_min, ctx=ctx)
    assert np.abs(got - ref).max() < 2e-13 * max(1.0, np.abs(ref).max())
    assert np.array_equal(got, engine.map2salm(f, spin, ell_max, ell_min=ell_min, ctx=ctx))
    route("SCRI_AMD_NO_SPLIT_ANALYSIS", "1")
    old = engine.map2salm(f, spin, ell_max, ell_min=ell_min, ctx=ctx)
    assert np.abs(old - ref).max() < 2e-13 * max(1.0, np.abs(ref).max())


@pytest.mark.parametrize("ell_max,n,data_type", [(8, 1501, "h"), (8, 700, "psi4"), (16, 901, "h"), (12, 333, "sigma"), (5, 64, "psi4")])
@pytest.mark.parametrize("rotated", [False, True])
def test_boost_free_transformations_take_the_separable_synthesis(ctx, monkeypatch, ell_max, n, data_type, rotated, route):
    """Without a boost the modes are rotated by the frame rotor and synthesised ring by ring (`synthesis_split_kernel`:
    theta stage on the VALU, folded phi stage on MFMA) instead of through the dense sYlm matrix; both routes must agree to
    rounding, with and without a frame rotation and with and without the inhomogeneous term of h / sigma."""
    import scri_amd
    from scri_amd import synthetic
    from oracle import containers

    t = np.linspace(-50.0, 60.0, n)
    data = synthetic.chirp_modes(t, 2, ell_max, 17 + ell_max)
    rng = np.random.default_rng(ell_max + n)
    lst = 3
    st = synthetic.real_supertranslation(0.3 * (rng.normal(size=(lst + 1) ** 2) + 1j * rng.normal(size=(lst + 1) ** 2)))
    kw = dict(supertranslation=st)
    if rotated:
        kw["frame_rotation"] = np.array([0.3, -0.5, 0.7, 0.41]) / np.linalg.norm([0.3, -0.5, 0.7, 0.41])
    dt = getattr(containers, data_type)

    def run():
        w = scri_amd.WaveformModes(t=t, data=data, ell_min=2, ell_max=ell_max, dataType=dt, frameType=scri_amd.Inertial,
                                   r_is_scaled_out=True, m_is_scaled_out=True, ctx=ctx)
        return w.transform(**kw)

    route("SCRI_AMD_NO_SEPARABLE_SYNTHESIS", None)  # (the suite may be run with the switch set)
    route("SCRI_AMD_NO_SMALL_DENSE", "1")  # (up to l <= 8 the default is the evaluating product: next test)
    ctx.enable_timing(True)
    ctx.get_timing(reset=True)
    got = run()
    assert rotated == ("rotate" in {k for k, v in ctx.get_timing(reset=True).items() if v[1]})  # the route was the separable one
    route("SCRI_AMD_NO_SEPARABLE_SYNTHESIS", "1")
    ref = run()
    ctx.enable_timing(False)
    assert got.n_times == ref.n_times and np.array_equal(got.t, ref.t)
    assert np.abs(got.data - ref.data).max() < 2e-13 * max(1.0, np.abs(ref.data).max())


def test_large_boost_wide_skew_and_chunks(ctx):
    import scri_amd

    t = np.arange(3000) * 0.1 - 100.0
    w = _wm(t, 4, 6, psi4)
    kw = dict(boost_velocity=np.array([0.05, -0.08, 0.06]), n_theta=15, n_phi=15)  # skew of ~ +-150 samples
    e = grid_ref.transform(w, **kw)
    for limit in (None, 6 << 20):
        c2 = scri_amd.Context(0, workspace_limit=limit) if limit else ctx
        g = _gpu(w, c2).transform(**kw)
        assert g.t.shape == e.t.shape
        assert np.abs(g.data - e.data).max() < 1e-11 * max(1.0, np.abs(e.data).max())


def test_abd_large_working_grid_uses_two_kernel_analysis(ctx):
    import scri_amd

    rng = np.random.default_rng(7)
    L, n = 12, 40
    u = np.linspace(0, 10, n)
    raw = np.zeros((6, n, (L + 1) ** 2), dtype=complex)
    LM = wigner.LM_range(0, L)
    for i, s in enumerate(ABD.spins):
        a = (rng.normal(size=LM.shape[0]) + 1j * rng.normal(size=LM.shape[0])) * 10.0 ** (-LM[:, 0] / 3.0)
        a[: s * s] = 0
        raw[i] = a[None, :] * (1 + 0.02 * u[:, None])
    kw = dict(boost_velocity=np.array([0.01, 0.0, 0.02]), space_translation=np.array([0.1, 0.0, -0.1]))  # grid 51 x 51
    e = abd_ref.transform(ABD(u, raw, L), **kw)
    g = scri_amd.AsymptoticBondiData(u, L, ctx=ctx)
    g._raw_data[:] = raw
    o = g.transform(**kw)
    assert o.n_times == e.n_times
    assert np.abs(o._raw_data - e.raw).max() < 1e-12 * max(1.0, np.abs(e.raw).max())


@pytest.mark.parametrize("ell_max", [24, 33, 40])
def test_rotation_large_ell(ctx, ell_max):
    """l <= 33 runs the MFMA kernel, above that the VALU kernel."""
    from scri_amd import engine

    rng = np.random.default_rng(8)
    n, nm = 70, (ell_max + 1) ** 2
    data = rng.normal(size=(n, nm)) + 1j * rng.normal(size=(n, nm))
    R = rng.normal(size=(n, 4))
    R /= np.linalg.norm(R, axis=1)[:, None]
    sp = quat.as_spinor_array(R)
    got = engine.rotate_series(data.copy(), 0, ell_max, sp, ctx=ctx)
    expect = rotations_ref.rotate_by_series(data, sp, 0, ell_max)
    assert np.abs(got - expect).max() < 1e-13 * ell_max


def test_strided_rows_and_interpolate(ctx):
    from scri_amd import engine

    rng = np.random.default_rng(9)
    big = rng.normal(size=(50, 40)) + 1j * rng.normal(size=(50, 40))
    view = big[:, :21]  # row stride 40 complex, 21 modes of l = 2..4
    q = np.array([0.3, 0.1, -0.7, 0.2])
    q /= np.linalg.norm(q)
    Ra, Rb = quat.as_spinor_array(q)
    expect = rotations_ref.rotate_by_constant(view.copy(), 2, 4, wigner.wigner_D_matrices(Ra, Rb, 2, 4))
    keep = big[:, 21:].copy()
    engine.rotate_const(view, 2, 4, q, ctx=ctx)
    assert np.abs(view - expect).max() < 1e-13 and np.array_equal(big[:, 21:], keep)
    x = np.linspace(0, 1, 50)
    xn = np.array([0.0, 0.013, 0.5, 0.99, 1.0])
    assert np.abs(engine.cubic_spline(x, big, xn, ctx=ctx) - CubicSpline(x, big)(xn)).max() < 1e-12


def test_output_window_matches_the_transforms(ctx):
    """bms_output_window announces exactly the rows the transformations produce (both flavours), so buffers can be sized
    before the data move."""
    from scri_amd import engine

    rng = np.random.default_rng(8)
    for trial in range(6):
        n, ell_max = int(rng.integers(50, 400)), 3
        t = np.cumsum(rng.uniform(0.05, 0.2, size=n)) - 10.0
        st = (rng.normal(size=9) + 1j * rng.normal(size=9)) * 0.3
        st[0] = st[0].real
        st[2], st[6] = st[2].real, st[6].real
        st[1], st[3] = np.conj(st[3]) * -1, st[3]
        st[4], st[8] = np.conj(st[8]), st[8]
        st[5], st[7] = -np.conj(st[7]), st[7]
        v = rng.normal(size=3) * (0.0 if trial == 0 else 0.05)
        n_theta = 2 * (ell_max + 2) + 1
        tr = engine.make_transformation(st, [1, 0, 0, 0], v, n_theta, n_theta, ell_max)
        data = rng.normal(size=(n, (ell_max + 1) ** 2 - 4)) + 0j
        t_new, d_new = engine.transform_modes(t, data, 2, ell_max, -2, -1, engine.BMS_TERM_H, tr, ctx=ctx)
        lo, hi = engine.output_window(t, tr, ctx=ctx)
        assert hi - lo == t_new.size
        raw = rng.normal(size=(6, n, (ell_max + 1) ** 2)) + 0j
        tr2 = engine.make_transformation(st, [1, 0, 0, 0], v, 2 * (2 * ell_max + 1) + 1, 2 * (2 * ell_max + 1) + 1, ell_max)
        u_new, r_new = engine.transform_abd(t, raw, ell_max, tr2, ctx=ctx)
        lo, hi = engine.output_window(t, tr2, abd=True, ctx=ctx)
        assert hi - lo == u_new.size and r_new.shape[1] == u_new.size and r_new.flags.c_contiguous


@pytest.mark.parametrize("n", [8, 9, 10, 13, 40, 320, 321, 353, 700, 1500])
@pytest.mark.parametrize("mesh", ["uniform", "ratio30", "alternating"])
def test_bspline_path_on_short_and_irregular_time_axes(ctx, n, mesh):
    """The B-spline form of the spline (elimination on the modes) against the oracle's scipy splines: series as short as
    the form allows (8 samples: below that the slope form runs), lengths around the 320-knot tile and its 32-knot halo,
    and irregular meshes, where the end rows of the collocation system and the decay of the tiled recurrences matter."""
    import scri_amd
    from oracle import waveform_grid_ref as grid_ref
    from oracle.containers import WM, h

    rng = np.random.default_rng(1000 + n)
    if mesh == "uniform":
        dt = np.full(n, 0.1)
    elif mesh == "ratio30":
        dt = rng.uniform(0.01, 0.3, size=n)
    else:
        dt = np.where(np.arange(n) % 2 == 0, 0.2, 0.01)
    t = np.cumsum(dt)
    ell_max = 3
    nm = (ell_max + 1) ** 2 - 4
    data = (np.sin(np.outer(t, rng.uniform(0.5, 3.0, size=nm))) + 1j * np.cos(np.outer(t, rng.uniform(0.5, 3.0, size=nm)))) * rng.normal(size=nm)
    data += 0.05 * (rng.normal(size=(n, nm)) + 1j * rng.normal(size=(n, nm)))  # rough on purpose: nothing is smoothed away
    st = np.zeros(9, dtype=complex)
    st[0] = 0.37 * dt.mean() * np.sqrt(4 * np.pi)  # a time translation by a fraction of a step ...
    st[2], st[6] = 0.02, -0.01                      # ... plus a direction-dependent part
    kw = dict(supertranslation=st, boost_velocity=np.array([0.01, -0.02, 0.015]))
    expect = grid_ref.transform(WM(t=t, data=data, ell_min=2, ell_max=ell_max, dataType=h), **kw)
    got = scri_amd.WaveformModes(t=t, data=data, ell_min=2, ell_max=ell_max, dataType=scri_amd.h, frameType=scri_amd.Inertial, r_is_scaled_out=True, m_is_scaled_out=True,
                                ctx=ctx).transform(**kw)
    assert got.t.size == expect.t.size and got.t.size > 0
    assert np.abs(got.t - expect.t).max() < 1e-13 * max(1.0, np.abs(t).max())
    assert np.abs(got.data - expect.data).max() < 2e-12 * max(1.0, np.abs(expect.data).max())


def test_pinned_result_arrays(ctx):
    """Results of host-mode transformations sit on page-locked memory owned by the array (freed blocks are reused for the
    next result of the same size); they behave like any numpy array."""
    from scri_amd import _lib

    a = _lib.pinned_empty((1 << 17, 2), np.complex128)  # 4 MiB
    assert a.flags.c_contiguous and a.flags.writeable and a.dtype == np.complex128
    a[:] = 1 + 2j
    assert a.sum() == (1 + 2j) * a.size
    where = a.ctypes.data
    view = a[10:20]  # keeps the block alive through .base
    del a
    assert view[0, 0] == 1 + 2j
    del view
    b = _lib.pinned_empty((1 << 17, 2), np.complex128)
    assert b.ctypes.data == where  # the freed block came back from the pool
    small = _lib.pinned_empty((8,), float)  # small arrays: ordinary memory
    assert small.base is None


@pytest.mark.parametrize("ratio", [1.5, 2.0, 1 / 1.7])
def test_geometrically_graded_time_axis_takes_the_exact_path(ctx, ratio):
    """Steps that grow (or shrink) geometrically over dozens of samples defeat the decay the tiled spline recurrences rely
    on (ratio 2 per step: 1e-5 with a 32-knot halo).  The engine detects such an axis (steps varying more than 1e3-fold
    within 48 samples) and runs the exact single-tile recurrences instead; a time shard of it is refused."""
    import scri_amd
    from oracle import waveform_grid_ref as grid_ref
    from oracle.containers import WM, h
    from scri_amd import engine

    rng = np.random.default_rng(17)
    n, ell_max = 700, 3  # the graded stretch straddles the boundary of the first 320-knot tile
    dt = np.concatenate([np.full(300, 1.0), ratio ** np.arange(1, 41), np.full(360, ratio**40)])
    dt /= dt.mean()
    t = np.cumsum(dt)
    nm = (ell_max + 1) ** 2 - 4
    phase = np.outer(np.log(t + 1.0), rng.uniform(0.5, 3.0, size=nm))
    data = (np.sin(phase) + 1j * np.cos(phase)) * rng.normal(size=nm)
    st = np.zeros(9, dtype=complex)
    st[0], st[2], st[6] = 0.05, 0.02, -0.01
    kw = dict(supertranslation=st, boost_velocity=np.array([0.01, -0.02, 0.015]))
    expect = grid_ref.transform(WM(t=t, data=data, ell_min=2, ell_max=ell_max, dataType=h), **kw)
    got = scri_amd.WaveformModes(t=t, data=data, ell_min=2, ell_max=ell_max, dataType=scri_amd.h, frameType=scri_amd.Inertial,
                                r_is_scaled_out=True, m_is_scaled_out=True, ctx=ctx).transform(**kw)
    assert got.t.size == expect.t.size and got.t.size > 100
    assert np.abs(got.data - expect.data).max() < 2e-12 * max(1.0, np.abs(expect.data).max())
    tr = engine.make_transformation(st, [1, 0, 0, 0], kw["boost_velocity"], 11, 11, ell_max)
    # a "shard" that holds every row is the whole series with a restricted output range: same exact path
    t_w, d_w, first = engine.transform_modes(t, data, 2, ell_max, -2, -1, engine.BMS_TERM_H, tr, ctx=ctx, shard=(0, n, 0, n))
    assert np.array_equal(t_w, got.t) and np.abs(d_w - got.data).max() < 1e-14 * max(1.0, np.abs(got.data).max())
    with pytest.raises(NotImplementedError, match="time steps vary"):
        engine.transform_modes(t, data[50:650], 2, ell_max, -2, -1, engine.BMS_TERM_H, tr, ctx=ctx, shard=(50, 600, 200, 500))


def test_geometrically_graded_time_axis_abd(ctx):
    """The same axis through AsymptoticBondiData.transform (whose host path always names the whole series as a shard to
    size its result exactly): exact single-tile recurrences, against the oracle."""
    import scri_amd
    from oracle import abd_ref
    from oracle.containers import ABD

    rng = np.random.default_rng(18)
    n, ell_max = 500, 3
    dt = np.concatenate([np.full(200, 1.0), 1.5 ** np.arange(1, 41), np.full(260, 1.5**40)])
    dt /= dt.mean()
    u = np.cumsum(dt)
    nm = (ell_max + 1) ** 2
    raw = np.zeros((6, n, nm), dtype=complex)
    for f, s in enumerate(ABD.spins):
        phase = np.outer(np.log(u + 1.0), rng.uniform(0.5, 3.0, size=nm))
        raw[f] = (np.sin(phase) + 1j * np.cos(phase)) * rng.normal(size=nm)
        raw[f, :, : s * s] = 0
    kw = dict(supertranslation=np.array([0.05, 0, 0.02, 0, 0, 0, -0.01, 0, 0], dtype=complex), boost_velocity=np.array([0.01, -0.02, 0.015]))
    expect = abd_ref.transform(ABD(u, raw, ell_max), **kw)
    a = scri_amd.AsymptoticBondiData(u, ell_max, ctx=ctx)
    a._raw_data[:] = raw
    got = a.transform(**kw)
    assert got.n_times == expect.n_times and got.n_times > 100
    assert np.abs(got._raw_data - expect.raw).max() < 2e-12 * max(1.0, np.abs(expect.raw).max())
    # a series long enough for the host pipeline: bms_transform_abd_pipelined refuses the graded axis and the call falls back
    from scri_amd import engine

    old_min, engine.PIPELINE_MIN_BYTES = engine.PIPELINE_MIN_BYTES, 1
    try:
        again = a.transform(**kw)
    finally:
        engine.PIPELINE_MIN_BYTES = old_min
    assert np.array_equal(again._raw_data, got._raw_data)


def test_workspace_limit_too_small_is_reported(ctx):
    """A work space limit that cannot hold a few spline halos of grid rows is an error (MemoryError), not silently exceeded."""
    import scri_amd
    from scri_amd import engine, synthetic

    t, data, spec = synthetic.workload("cfg3", n_times=4000)
    tr = engine.make_transformation(spec["kwargs"]["supertranslation"], spec["kwargs"]["frame_rotation"], spec["kwargs"]["boost_velocity"], 37, 37, 16)
    tiny = scri_amd.Context(0, workspace_limit=4 << 20)  # 4 MiB: ~48 rows of the 1297-column grids
    try:
        with pytest.raises(MemoryError, match="work space limit"):
            engine.transform_modes(t, data, 2, 16, -2, -1, engine.BMS_TERM_H, tr, ctx=tiny)
    finally:
        tiny.close()
    ok = scri_amd.Context(0, workspace_limit=64 << 20)  # several chunks
    try:
        t1, d1 = engine.transform_modes(t, data, 2, 16, -2, -1, engine.BMS_TERM_H, tr, ctx=ok)
    finally:
        ok.close()
    t0, d0 = engine.transform_modes(t, data, 2, 16, -2, -1, engine.BMS_TERM_H, tr, ctx=ctx)
    assert np.array_equal(t0, t1) and np.abs(d0 - d1).max() < 1e-14 * np.abs(d0).max()


def test_reused_host_input_is_page_locked_and_released(ctx):
    """An input array handed in for the second time is page-locked in place (uploads at PCIe rate), the results do not change,
    and the registration goes away with the array."""
    import gc

    from scri_amd import _lib, engine, synthetic

    t, data, spec = synthetic.workload("cfg3", n_times=120_000)  # 547 MB: above the registration threshold
    data = np.ascontiguousarray(data[:, :77])  # l <= 8: 148 MB
    kw = synthetic.CONFIGS["cfg2"]["kwargs"]
    tr = engine.make_transformation(kw["supertranslation"], [1, 0, 0, 0], [0, 0, 0], 21, 21, 8)
    key = id(data)
    outs = []
    for i in range(3):
        outs.append(engine.transform_modes(t, data, 2, 8, -2, -1, engine.BMS_TERM_H, tr, ctx=ctx)[1].copy())
        assert (key in _lib._registered) == (i >= 1)
    assert np.array_equal(outs[0], outs[1]) and np.array_equal(outs[1], outs[2])
    # a one-off array -- a fresh copy per call, as an adapter builds it -- is never page-locked, even when malloc hands out the
    # block of its predecessor again (same address, same size: another object)
    before = set(_lib._registered)
    for i in range(3):
        tmp = data.copy()
        engine.transform_modes(t, tmp, 2, 8, -2, -1, engine.BMS_TERM_H, tr, ctx=ctx)
        assert set(_lib._registered) == before
        del tmp
    del data
    gc.collect()
    assert key not in _lib._registered


@pytest.mark.parametrize("n", [6, 40])
def test_separable_synthesis_reads_nothing_past_the_modes(ctx, monkeypatch, n, route):
    """Boost-free psi-type / slope-form transformations hand the one-kernel synthesis rows WITHOUT the constant column the h / sigma
    route appends: the kernel must not touch the element behind a row's modes (behind the last row: memory past the caller's
    buffer).  A device-resident series whose buffer is followed by NaNs shows it."""
    route("SCRI_AMD_NO_SMALL_DENSE", "1")
    import torch
    from scri_amd import engine, synthetic

    route("SCRI_AMD_NO_SEPARABLE_SYNTHESIS", None)
    route("SCRI_AMD_NO_BSPLINE", "1")  # slope form: the synthesis reads the caller's rows directly
    ell_max = 8
    t = np.linspace(-3.0, 4.0, n)
    data = synthetic.chirp_modes(t, 2, ell_max, 17)
    nm = data.shape[1]
    st = synthetic.real_supertranslation(0.05 * (np.arange(9) + 1j * np.arange(9)[::-1]))
    n_theta = 2 * ell_max + 1
    tr = engine.make_transformation(st, [1.0, 0, 0, 0], [0.0, 0, 0], n_theta, n_theta, ell_max)
    dev = torch.device("cuda", 0)
    buf = torch.full((n * nm + 64,), float("nan"), dtype=torch.complex128, device=dev)
    buf[: n * nm] = torch.from_numpy(data.reshape(-1)).to(dev)
    out = torch.zeros((n, nm), dtype=torch.complex128, device=dev)
    torch.cuda.synchronize()

    def run():
        out.zero_()
        torch.cuda.synchronize()
        n_new = engine.transform_modes(t, buf.data_ptr(), 2, ell_max, -2, -1, engine.BMS_TERM_H, tr, ctx=ctx, device=True, ld=nm,
                                       out_ptr=out.data_ptr(), shard=(0, n, 0, n))[1]
        ctx.synchronize()
        return out[:n_new].cpu().numpy()

    got = run()
    assert got.shape[0] > 0 and np.isfinite(got).all()
    route("SCRI_AMD_NO_SEPARABLE_SYNTHESIS", "1")  # the dense product reads exactly n_modes columns
    ref = run()
    assert got.shape == ref.shape and not np.array_equal(got, ref)
    assert np.abs(got - ref).max() < 1e-13 * max(1.0, np.abs(ref).max())


@pytest.mark.parametrize("defect", ["swap", "repeat", "nan"])
def test_defect_in_the_middle_of_the_time_axis_is_reported_by_the_late_walk(ctx, defect, monkeypatch, route):
    """The host walks the time axis while it waits for the per-direction tables (engine_modes.hip, `walk_later`): by then the time axis is in
    HBM and the spline solve has been queued on it.  A defect found by that walk fails the call exactly as the walk-first order does,
    and the context is usable afterwards."""
    from scri_amd import engine

    rng = np.random.default_rng(5)
    n, ell_max = 4000, 4
    t = np.linspace(-50.0, 150.0, n)
    nm = (ell_max + 1) ** 2 - 4
    data = rng.normal(size=(n, nm)) + 1j * rng.normal(size=(n, nm))
    st = np.zeros(9, dtype=complex)
    st[0] = 0.1
    tr = engine.make_transformation(st, [1, 0, 0, 0], np.array([0.01, 0.02, -0.01]), 13, 13, ell_max)
    good = engine.transform_modes(t, data, 2, ell_max, -2, -1, engine.BMS_TERM_H, tr, ctx=ctx)
    bad = t.copy()
    if defect == "swap":
        bad[2500], bad[2501] = bad[2501], bad[2500]
    elif defect == "repeat":
        bad[2501] = bad[2500]
    else:
        bad[2501] = np.nan
    messages = []
    for first in (False, True):
        if first:
            route("SCRI_AMD_WALK_FIRST", "1")
        with pytest.raises(ValueError, match=r"strictly increasing \(index 250[12]\)") as err:
            engine.transform_modes(bad, data, 2, ell_max, -2, -1, engine.BMS_TERM_H, tr, ctx=ctx)
        messages.append(str(err.value))
    assert messages[0] == messages[1]
    route("SCRI_AMD_WALK_FIRST", None)
    again = engine.transform_modes(t, data, 2, ell_max, -2, -1, engine.BMS_TERM_H, tr, ctx=ctx)
    assert np.array_equal(again[0], good[0]) and np.array_equal(again[1], good[1])



@pytest.mark.parametrize("ell_max,n,expect_dense", [(4, 300, True), (8, 900, True), (8, 9, True), (10, 400, False)])
def test_small_boost_free_shapes_take_the_evaluating_product(ctx, monkeypatch, ell_max, n, expect_dense, route):
    """Since the dense product evaluates the spline itself it beats separable synthesis + back substitution on the grid for small shapes
    (engine_modes.hip, `small_dense`: n_modes x grid <= 40 000, i.e. up to l <= 8 on the default grid, and at least 8 rows): the route is chosen by
    that rule, and both routes agree to rounding."""
    import scri_amd
    from scri_amd import synthetic
    from oracle import containers

    t = np.linspace(-20.0, 30.0, n)
    data = synthetic.chirp_modes(t, 2, ell_max, 3 + ell_max)
    st = synthetic.real_supertranslation(0.2 * (np.arange(9) - 4.0 + 1j * np.arange(9)[::-1]))
    kw = dict(supertranslation=st, frame_rotation=np.array([0.3, -0.5, 0.7, 0.41]) / np.linalg.norm([0.3, -0.5, 0.7, 0.41]))

    def run():
        w = scri_amd.WaveformModes(t=t, data=data, ell_min=2, ell_max=ell_max, dataType=containers.h, frameType=scri_amd.Inertial,
                                   r_is_scaled_out=True, m_is_scaled_out=True, ctx=ctx)
        ctx.enable_timing(True)
        ctx.get_timing(reset=True)
        out = w.transform(**kw)
        tags = {k for k, v in ctx.get_timing(reset=True).items() if v[1]}
        ctx.enable_timing(False)
        return out, tags

    for k in ("SCRI_AMD_NO_SEPARABLE_SYNTHESIS", "SCRI_AMD_NO_SMALL_DENSE", "SCRI_AMD_NO_GEMM_EVAL"):
        route(k, None)
    got, tags = run()
    assert ("rotate" not in tags) == expect_dense  # (the separable route turns the modes into the rotated frame first)
    route("SCRI_AMD_NO_SMALL_DENSE", "1")
    ref, tags_ref = run()
    assert "rotate" in tags_ref
    assert np.array_equal(got.t, ref.t)
    assert np.abs(got.data - ref.data).max() < 1e-13 * max(1.0, np.abs(ref.data).max())


@pytest.mark.parametrize("ell_max,expect", [(8, "dense"), (12, "two-pass"), (14, "two-pass"), (15, "fused"), (16, "fused")])
def test_boost_free_route_follows_the_shape_rules(ctx, route, ell_max, expect):
    """The three boost-free routes of WaveformModes by shape (engine_modes.hip): the evaluating product up to l <= 8
    (`small_dense`), separable synthesis + back substitution on the grid up to l_max = 14, and from l_max = 15 on the separable synthesis
    that evaluates the spline itself (`SYN_EVAL_MIN_ELL`: measured crossover, profiles/r06_b_boost_free_routes_by_ell.txt) -- read off
    the kernels' timing tags; the options force either side and all routes agree to rounding."""
    import scri_amd
    from scri_amd import synthetic
    from oracle import containers

    n = 900
    t = np.linspace(-20.0, 30.0, n)
    data = synthetic.chirp_modes(t, 2, ell_max, 3 + ell_max)
    kw = dict(supertranslation=np.asarray(synthetic.S9) * 30, frame_rotation=np.array([0.3, -0.5, 0.7, 0.41]) / np.linalg.norm([0.3, -0.5, 0.7, 0.41]))

    def run():
        w = scri_amd.WaveformModes(t=t, data=data, ell_min=2, ell_max=ell_max, dataType=containers.h, frameType=scri_amd.Inertial,
                                   r_is_scaled_out=True, m_is_scaled_out=True, ctx=ctx)
        ctx.enable_timing(True)
        ctx.get_timing(reset=True)
        out = w.transform(**kw)
        tags = {k for k, v in ctx.get_timing(reset=True).items() if v[1]}
        ctx.enable_timing(False)
        return out, tags

    def which(tags):
        if "rotate" not in tags:
            return "dense"
        return "two-pass" if "spline_backward" in tags else "fused"

    for k in ("SCRI_AMD_NO_SEPARABLE_SYNTHESIS", "SCRI_AMD_NO_SMALL_DENSE", "SCRI_AMD_NO_GEMM_EVAL", "SCRI_AMD_SYNTHESIS_EVAL", "SCRI_AMD_NO_SYNTHESIS_EVAL"):
        route(k, None)
    got, tags = run()
    assert which(tags) == expect, tags
    # the other side of the rule, forced through the context's options
    route("SCRI_AMD_NO_SMALL_DENSE", "1")
    route("SCRI_AMD_NO_SYNTHESIS_EVAL" if expect == "fused" else "SCRI_AMD_SYNTHESIS_EVAL", "1")
    other, tags_other = run()
    assert which(tags_other) == ("two-pass" if expect == "fused" else "fused"), tags_other
    assert np.array_equal(got.t, other.t)
    assert np.abs(got.data - other.data).max() < 3e-13 * max(1.0, np.abs(other.data).max())


def _fused_case(ell_max, n, mesh, st_scale, seed):
    from scri_amd import synthetic

    rng = np.random.default_rng(seed)
    if mesh == "uniform":
        t = np.linspace(-40.0, 55.0, n)
    elif mesh == "jitter":
        t = (np.arange(n) + 0.3 * rng.uniform(-1, 1, size=n)) * (95.0 / n) - 40.0
    elif mesh == "graded":  # steps shrinking 20x over the series
        steps = 20.0 ** (-np.arange(n) / (n - 1.0))
        t = np.cumsum(steps)
        t = t * (95.0 / t[-1]) - 40.0
    else:  # "rough": steps drawn between 0.2 and 1.8 of the mean
        t = np.cumsum(rng.uniform(0.2, 1.8, size=n)) * (95.0 / n) - 40.0
    data = synthetic.chirp_modes(t, 2, ell_max, 40 + ell_max)
    lst = 3
    st = synthetic.real_supertranslation(st_scale * (rng.normal(size=(lst + 1) ** 2) + 1j * rng.normal(size=(lst + 1) ** 2)))
    rot = np.array([0.3, -0.5, 0.7, 0.41])
    return t, data, dict(supertranslation=st, frame_rotation=rot / np.linalg.norm(rot))


@pytest.mark.parametrize("mesh", ["uniform", "jitter", "graded", "rough"])
@pytest.mark.parametrize("ell_max,n,st_scale", [(16, 700, 0.02), (16, 1900, 0.3), (12, 333, 0.3), (10, 5000, 0.2), (9, 64, 0.05), (16, 9, 0.01),
                                                 (14, 1100, 6.0)])
def test_synthesis_with_the_evaluation_in_it_equals_the_two_pass_route_and_the_oracle(ctx, monkeypatch, mesh, ell_max, n, st_scale, route):
    """`synthesis_eval_kernel` (boost-free WaveformModes: the spline solved on the modes, evaluated by the synthesis kernel from the
    last four coefficient rows of each pixel) against the route it replaces -- elimination on the modes, `synthesis_split_kernel`,
    `bspline_backward_eval_kernel` -- and against the oracle: uniform, jittered, graded and rough time axes; supertranslations from
    a fraction of a step to tens of steps (the pixels' samples then trail their knots by very different numbers of rows; the last
    case exceeds the bound of the route and must fall back by itself); series from 9 samples up."""
    import scri_amd

    t, data, kw = _fused_case(ell_max, n, mesh, st_scale, 7 * ell_max + n)
    route("SCRI_AMD_NO_SMALL_DENSE", "1")
    route("SCRI_AMD_NO_SEPARABLE_SYNTHESIS", None)

    def run():
        w = scri_amd.WaveformModes(t=t, data=data, ell_min=2, ell_max=ell_max, dataType=scri_amd.h, frameType=scri_amd.Inertial,
                                   r_is_scaled_out=True, m_is_scaled_out=True, ctx=ctx)
        ctx.enable_timing(True)
        ctx.get_timing(reset=True)
        out = w.transform(**kw)
        tags = {k for k, v in ctx.get_timing(reset=True).items() if v[1]}
        ctx.enable_timing(False)
        return out, tags

    route("SCRI_AMD_SYNTHESIS_EVAL", "1")
    got, tags = run()
    dt_min = np.diff(t).min()
    if st_scale <= 0.3 and mesh in ("uniform", "jitter"):  # (elsewhere the bound on the spread of the skews may send the call to the old route)
        assert "spline_backward" not in tags, tags  # no pass over a grid of coefficients on this route
    route("SCRI_AMD_SYNTHESIS_EVAL", None)
    route("SCRI_AMD_NO_SYNTHESIS_EVAL", "1")  # (the fused route is the default from l_max = 15 on: the two-pass route is named)
    ref, tags_ref = run()
    assert "spline_backward" in tags_ref
    assert got.n_times == ref.n_times and np.array_equal(got.t, ref.t)
    scale = max(1.0, np.abs(ref.data).max())
    assert np.abs(got.data - ref.data).max() < 3e-13 * scale, (np.abs(got.data - ref.data).max(), dt_min)
    if n <= 2000:
        e = grid_ref.transform(WM(t=t, data=data, ell_min=2, ell_max=ell_max, dataType=h), **kw)
        assert e.t.shape == got.t.shape and np.abs(got.t - e.t).max() < 1e-12
        assert np.abs(got.data - e.data).max() < 1e-12 * max(1.0, np.abs(e.data).max())


def test_synthesis_with_the_evaluation_in_it_chunks_shards_and_grid_output(ctx, monkeypatch, route):
    """The same kernel under what the engine does around it: the time axis walked in chunks of a small work space (each chunk a
    launch with its own run-in rows), time shards with halos (what the ranks of a sharded run compute), a series long enough for
    several segments per workgroup, and the grid output of WaveformGrid.from_modes."""
    import scri_amd
    from scri_amd import engine, sharding

    ell_max, n = 16, 9000
    t, data, kw = _fused_case(ell_max, n, "jitter", 0.8, 99)
    route("SCRI_AMD_NO_SMALL_DENSE", "1")
    route("SCRI_AMD_SYNTHESIS_EVAL", "1")
    n_theta = 2 * (ell_max + 3) + 1
    tr = engine.make_transformation(kw["supertranslation"], kw["frame_rotation"], [0.0, 0.0, 0.0], n_theta, n_theta, ell_max)
    t1, d1 = engine.transform_modes(t, data, 2, ell_max, -2, -1, engine.BMS_TERM_H, tr, ctx=ctx)
    d1 = np.array(d1)
    route("SCRI_AMD_SYNTHESIS_EVAL", None)
    route("SCRI_AMD_NO_SYNTHESIS_EVAL", "1")
    t0, d0 = engine.transform_modes(t, data, 2, ell_max, -2, -1, engine.BMS_TERM_H, tr, ctx=ctx)
    route("SCRI_AMD_NO_SYNTHESIS_EVAL", None)
    route("SCRI_AMD_SYNTHESIS_EVAL", "1")
    scale = np.abs(d0).max()
    assert np.array_equal(t0, t1) and np.abs(d1 - d0).max() < 3e-13 * scale
    small = scri_amd.Context(0, workspace_limit=48 << 20)  # ~ 8 chunks
    t2, d2 = engine.transform_modes(t, data, 2, ell_max, -2, -1, engine.BMS_TERM_H, tr, ctx=small)
    assert np.array_equal(t2, t1) and np.abs(d2 - d1).max() < 1e-14 * scale
    have, need, window = sharding.plan(t, tr, 4)
    parts = []
    for r in range(4):
        ext = data[need[r][0] : need[r][1]]
        parts.append(np.array(engine.transform_modes(t, ext, 2, ell_max, -2, -1, engine.BMS_TERM_H, tr, ctx=small,
                                                     shard=(need[r][0], ext.shape[0], have[r][0], have[r][1]))[1]))
    small.close()
    assert np.abs(np.concatenate(parts) - d1).max() < 1e-14 * scale
    # from_modes on its own: the grid of samples in grid order, new route against old
    tg, g1 = engine.transform_modes(t[:1200], data[:1200], 2, ell_max, -2, -1, engine.BMS_TERM_H, tr, ctx=ctx, grid=True)
    g1 = np.array(g1)
    route("SCRI_AMD_SYNTHESIS_EVAL", None)
    route("SCRI_AMD_NO_SYNTHESIS_EVAL", "1")
    tg0, g0 = engine.transform_modes(t[:1200], data[:1200], 2, ell_max, -2, -1, engine.BMS_TERM_H, tr, ctx=ctx, grid=True)
    assert np.array_equal(tg, tg0) and g1.shape == g0.shape and np.abs(g1 - g0).max() < 3e-13 * max(1.0, np.abs(g0).max())


def test_reserved_slab_serves_the_work_space(ctx):
    """Context.reserve (bms_ctx_reserve): one allocation up front from which the named work-space buffers are carved -- same results
    as a context that allocates on demand; a slab that is too small leaves the larger buffers to allocations of their own; a second
    reserve adds room."""
    import scri_amd

    w = _wm(np.linspace(-30.0, 45.0, 2500), 10, 5)
    kw = dict(supertranslation=np.array([0.0, 0.02, 0.01 + 0.01j, -0.02, 0.005, 0, 0.01, 0, 0.005]),
              frame_rotation=np.array([0.9, 0.1, -0.3, 0.2]) / np.linalg.norm([0.9, 0.1, -0.3, 0.2]), boost_velocity=np.array([1e-3, -2e-3, 1.5e-3]))
    kw["supertranslation"] = kw["supertranslation"].astype(complex)
    from scri_amd import synthetic

    kw["supertranslation"] = synthetic.real_supertranslation(kw["supertranslation"])
    ref = _gpu(w, ctx).transform(**kw)
    for nbytes in (1 << 30, 4 << 20):  # roomy; too small for the grids (they fall back to their own allocations)
        c2 = scri_amd.Context(0)
        c2.reserve(nbytes)
        got = _gpu(w, c2).transform(**kw)
        assert np.array_equal(got.t, ref.t) and np.array_equal(got.data, ref.data)
        c2.reserve(256 << 20)  # a second slab: later growth is served from it
        w2 = _wm(np.linspace(-30.0, 45.0, 5200), 10, 5)
        got2 = _gpu(w2, c2).transform(**kw)
        ref2 = _gpu(w2, ctx).transform(**kw)
        assert np.array_equal(got2.data, ref2.data)
        c2.close()


def test_eval_window_statistics(ctx):
    """bms_ctx_get_eval_stats: tiles of the evaluating product launched since the last reset, how many of them had samples outside
    the window of output times they stage in LDS, how many per-column marches went on from global memory.  A mild boost keeps every
    tile on its window; a boost of 0.3 c (a direction's samples trail its knots by hundreds of rows within one tile's columns)
    does not -- and both give the oracle's numbers."""
    t = np.linspace(-30.0, 40.0, 1500)
    w = _wm(t, 8, 3)
    st = np.zeros(9, dtype=complex)
    st[0], st[2] = 0.3, 0.05
    ctx.eval_stats(reset=True)
    assert ctx.eval_stats(reset=False) == (0, 0, 0)
    _check(w, ctx, supertranslation=st, boost_velocity=np.array([1e-3, -2e-3, 1.5e-3]))
    tiles, off, cont = ctx.eval_stats(reset=True)
    assert tiles > 0 and off == 0 and cont == 0
    _check(w, ctx, tol=4e-12, supertranslation=st, boost_velocity=np.array([0.2, -0.15, 0.18]))
    tiles2, off2, cont2 = ctx.eval_stats(reset=True)
    assert tiles2 > 0 and off2 + cont2 > 0
    assert ctx.eval_stats(reset=False) == (0, 0, 0)
