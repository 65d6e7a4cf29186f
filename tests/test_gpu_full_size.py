"""The BASELINE.json configurations at FULL size on the GPU (1e5 time steps, l_max = 16 / 8; the ABD working grid of
cfg5), checked through properties that do not need an oracle pass over the whole series:

  * locality: the transform of a window of the series depends only on that window + a halo, so windows of the
    full-size output are compared with the oracle run on slices of the input (start, middle, end of the series);
  * linearity in the data (exact for Psi4; affine for h, whose inhomogeneous term cancels in a difference);
  * chunked work space == one chunk, shards == whole series;
  * rotation: R then R^-1 is the identity, and every l-block keeps its norm (unitarity of the Wigner matrices);
  * analytic answer at every time step: boosted Schwarzschild four-momentum m gamma (1, -v) on the cfg5 grid.
"""
import numpy as np
import pytest

from oracle import quat
from oracle import waveform_grid_ref as grid_ref
from oracle.containers import WM, h, psi4

pytestmark = pytest.mark.gpu

N = 100_000


def _gpu_wm(t, data, ell_max, dataType, ctx):
    import scri_amd

    return scri_amd.WaveformModes(t=t, data=data, ell_min=2, ell_max=ell_max, dataType=dataType, frameType=scri_amd.Inertial,
                                  r_is_scaled_out=True, m_is_scaled_out=True, ctx=ctx)


@pytest.fixture(scope="module")
def cfg3(ctx):
    from scri_amd import synthetic

    t, data, spec = synthetic.workload("cfg3")
    assert data.shape == (N, 285)
    out = _gpu_wm(t, data, 16, h, ctx).transform(**spec["kwargs"])
    return t, data, spec["kwargs"], out


def test_cfg3_windows_match_oracle(cfg3):
    t, data, kw, out = cfg3
    assert 99_900 < out.n_times <= N and np.all(np.diff(out.t) > 0)
    st = np.asarray(kw["supertranslation"])
    v = np.asarray(kw["boost_velocity"])
    tt = st[0].real / np.sqrt(4 * np.pi)
    gamma = 1 / np.sqrt(1 - v @ v)
    uprm = (t - tt) / gamma  # output time of input index i
    for i0 in (0, 49_800, N - 400):
        sl = slice(i0, i0 + 400)
        e = grid_ref.transform(WM(t=t[sl], data=data[sl], ell_min=2, ell_max=16, dataType=h), **kw)
        # interior of the window (the oracle's own ends see a truncated spline): 60 samples in from both sides
        keep = e.t[60:-60]
        gi = np.searchsorted(out.t, keep - 1e-9)
        assert np.abs(out.t[gi] - keep).max() < 1e-10
        assert np.abs(uprm[np.searchsorted(uprm, keep - 1e-9)] - keep).max() < 1e-10
        err = np.abs(out.data[gi] - e.data[60:-60]).max()
        assert err < 1e-12 * max(1.0, np.abs(e.data).max()), (i0, err)


def test_cfg3_size_without_boost_separable_equals_dense(cfg3, ctx, monkeypatch, route):
    """The cfg3 series (1e5 steps, l <= 16) through a boost-free transformation -- rotated modes + `synthesis_split_kernel` --
    against the dense sYlm product on the same grid, and a window of it against the oracle."""
    t, data, kw, _ = cfg3
    kw0 = {k: v for k, v in kw.items() if k != "boost_velocity"}
    got = _gpu_wm(t, data, 16, h, ctx).transform(**kw0)
    route("SCRI_AMD_NO_SEPARABLE_SYNTHESIS", "1")
    ref = _gpu_wm(t, data, 16, h, ctx).transform(**kw0)
    route("SCRI_AMD_NO_SEPARABLE_SYNTHESIS", None)
    assert got.n_times == ref.n_times and np.array_equal(got.t, ref.t)
    assert np.abs(got.data - ref.data).max() < 1e-13 * max(1.0, np.abs(ref.data).max())
    i0 = 61_000
    sl = slice(i0, i0 + 400)
    e = grid_ref.transform(WM(t=t[sl], data=data[sl], ell_min=2, ell_max=16, dataType=h), **kw0)
    keep = e.t[60:-60]
    gi = np.searchsorted(got.t, keep - 1e-9)
    assert np.abs(got.t[gi] - keep).max() < 1e-10
    assert np.abs(got.data[gi] - e.data[60:-60]).max() < 1e-12 * max(1.0, np.abs(e.data).max())


def test_cfg3_linearity_and_affinity(cfg3, ctx):
    from scri_amd import synthetic

    t, x, kw, Tx = cfg3
    y = synthetic.chirp_modes(t, 2, 16, 99)
    # Psi4 has no inhomogeneous term: T(2x - 3y) = 2 T(x) - 3 T(y)
    Tpx = _gpu_wm(t, x, 16, psi4, ctx).transform(**kw).data
    Tpy = _gpu_wm(t, y, 16, psi4, ctx).transform(**kw).data
    Tpz = _gpu_wm(t, 2 * x - 3 * y, 16, psi4, ctx).transform(**kw).data
    scale = np.abs(Tpz).max()
    assert np.abs(Tpz - (2 * Tpx - 3 * Tpy)).max() < 1e-13 * scale
    # h is affine: T(x + y) - T(x) - T(y) + T(0) = 0
    T0 = _gpu_wm(t, np.zeros_like(x), 16, h, ctx).transform(**kw).data
    Ty = _gpu_wm(t, y, 16, h, ctx).transform(**kw).data
    Txy = _gpu_wm(t, x + y, 16, h, ctx).transform(**kw).data
    assert np.abs(T0).max() > 1e-8  # the offset is really there
    assert np.abs(Txy - Tx.data - Ty + T0).max() < 1e-13 * np.abs(Txy).max()


def test_cfg3_chunks_and_shards_equal_whole(cfg3):
    import scri_amd
    from scri_amd import engine, sharding

    t, data, kw, out = cfg3
    tr = engine.make_transformation(kw["supertranslation"], kw["frame_rotation"], kw["boost_velocity"], 37, 37, 16)
    small = scri_amd.Context(0, workspace_limit=1 << 30)  # ~ 15 chunks
    t2, d2 = engine.transform_modes(t, data, 2, 16, -2, -1, engine.BMS_TERM_H, tr, ctx=small)
    assert np.array_equal(t2, out.t) and np.abs(d2 - out.data).max() < 1e-14 * np.abs(out.data).max()
    have, need, window = sharding.plan(t, tr, 8)
    parts = []
    for r in range(8):
        ext = data[need[r][0] : need[r][1]]
        parts.append(engine.transform_modes(t, ext, 2, 16, -2, -1, engine.BMS_TERM_H, tr, ctx=small,
                                            shard=(need[r][0], ext.shape[0], have[r][0], have[r][1]))[1])
    small.close()
    assert np.abs(np.concatenate(parts) - out.data).max() < 1e-14 * np.abs(out.data).max()


def test_cfg3_stress_variant_beta_1e_2_windows_shards_and_partition(ctx):
    """SURVEY 8(d)'s stress variant of cfg3 at full size: the same series under a boost of |v| = 1e-2 (26.7 x cfg3's).  A direction's
    time axis is skewed by up to beta |u| / dt = 1000 rows at the end of the series (scri/waveform_grid.py:564-568, 578), which is
    what the chunk plan, the shard halos and the search windows of the evaluating product have to cover: three windows of the
    one-GPU result against the oracle on input slices (the oracle trims beta (t_a + t_b) / dt rows from a slice, so the late
    slices are longer), the 8 time shards of sharding.plan against the whole, and the partition verdict (halo ~ 1000 rows of a
    12 500-row shard: still "rows")."""
    from scri_amd import engine, sharding, synthetic

    t, data, spec = synthetic.workload("cfg3")
    kw = dict(spec["kwargs"])
    kw["boost_velocity"] = list(26.7 * np.asarray(kw["boost_velocity"]))
    beta = float(np.linalg.norm(kw["boost_velocity"]))
    assert 0.0099 < beta < 0.0101
    out = _gpu_wm(t, data, 16, h, ctx).transform(**kw)
    assert N - 2300 < out.n_times < N - 900 and np.all(np.diff(out.t) > 0)  # the window loses ~ beta t_end / dt rows at either end
    worst = _window_check(out.t, out.data, t, data, kw, 16, [(0, 400), (50_000, 1800), (N - 3200, 3200)])
    assert worst < 1e-12
    tr = engine.make_transformation(kw["supertranslation"], kw["frame_rotation"], kw["boost_velocity"], 37, 37, 16)
    have, need, window = sharding.plan(t, tr, 8)
    assert window[1] - window[0] == out.n_times
    halos = [max(h0 - n0, n1 - h1) for (h0, h1), (n0, n1) in zip(have, need)]
    assert 900 < max(halos) < 1200, halos  # skew of the last shard + the spline margin
    assert sharding.choose_partition(have, need) == "rows"
    scale = np.abs(out.data).max()
    row = 0
    for r in range(8):
        ext = data[need[r][0] : need[r][1]]
        tp, dp, first = engine.transform_modes(t, ext, 2, 16, -2, -1, engine.BMS_TERM_H, tr, ctx=ctx,
                                               shard=(need[r][0], ext.shape[0], have[r][0], have[r][1]))
        assert first == window[0] + row
        assert np.array_equal(tp, out.t[row : row + tp.shape[0]])
        assert np.abs(dp - out.data[row : row + dp.shape[0]]).max() < 1e-14 * scale
        row += dp.shape[0]
    assert row == out.n_times


def _window_check(out_t, out_data, t_in, data_in, kw, ell_max, windows, margin=60, tol=1e-12):
    """Windows of a full-size output against the oracle run on slices of the input: `windows` = [(i0, rows)]; the oracle's
    own first / last `margin` outputs see a truncated spline and are left out."""
    worst = 0.0
    for i0, rows in windows:
        sl = slice(i0, i0 + rows)
        e = grid_ref.transform(WM(t=t_in[sl], data=data_in[sl], ell_min=2, ell_max=ell_max, dataType=h), **kw)
        assert e.n_times > 2 * margin + 50, (i0, e.n_times)
        keep = e.t[margin:-margin]
        gi = np.searchsorted(out_t, keep - 1e-9)
        assert np.abs(out_t[gi] - keep).max() < 1e-13 * max(1.0, abs(keep[-1])), i0
        err = np.abs(out_data[gi] - e.data[margin:-margin]).max()
        assert err < tol * max(1.0, np.abs(e.data).max()), (i0, err)
        worst = max(worst, err)
    return worst


@pytest.mark.parametrize("axis", ["jitter", "sxs"])
def test_cfg3_on_non_uniform_time_axes(ctx, axis):
    """cfg3 at full size on the two non-uniform axes of synthetic.time_axis (every BASELINE workload has dt = 0.1; SXS / CCE output is
    adaptively stepped): `jitter` (+-30 % of dt per sample) and `sxs` (steps shrinking 20x over the series).  The spline is the
    not-a-knot interpolant on the REAL knots (scri/waveform_grid.py:574-588), so the oracle on input slices applies unchanged: three
    windows each (the late `sxs` window is long: a skew of 3.7 time units is 240 rows of 0.016 there), the 8 time shards of
    sharding.plan against the whole series, and the share of the evaluating product's tiles that stayed on their LDS window."""
    from scri_amd import engine, sharding, synthetic

    t, data, spec = synthetic.workload("cfg3", axis=axis)
    kw = spec["kwargs"]
    d = np.diff(t)
    assert d.min() > 0 and (d.max() / d.min() > 3.5 if axis == "jitter" else d[0] / d[-1] > 19)
    ctx.eval_stats(reset=True)
    out = _gpu_wm(t, data, 16, h, ctx).transform(**kw)
    tiles, off_lds, continued = ctx.eval_stats(reset=True)
    assert N - 700 < out.n_times <= N and np.all(np.diff(out.t) > 0)
    windows = [(0, 400), (50_000, 600), (N - 400, 400)] if axis == "jitter" else [(0, 400), (50_000, 700), (N - 2000, 2000)]
    worst = _window_check(out.t, out.data, t, data, kw, 16, windows)
    assert worst < 1e-12
    # the product's tiles place their window of output rows from their OWN knots' step: (nearly) all of them stay in LDS
    assert tiles > 30_000 and off_lds <= 0.02 * tiles, (tiles, off_lds, continued)
    # eight time shards (each from its own rows + halo, what eight ranks compute) == the whole series
    tr = engine.make_transformation(kw["supertranslation"], kw["frame_rotation"], kw["boost_velocity"], 37, 37, 16)
    have, need, window = sharding.plan(t, tr, 8)
    parts = []
    for r in range(8):
        ext = data[need[r][0] : need[r][1]]
        parts.append(engine.transform_modes(t, ext, 2, 16, -2, -1, engine.BMS_TERM_H, tr, ctx=ctx,
                                            shard=(need[r][0], ext.shape[0], have[r][0], have[r][1]))[1])
    whole = np.concatenate(parts)
    assert whole.shape == out.data.shape and np.abs(whole - out.data).max() < 1e-14 * np.abs(out.data).max()
    # and without the boost (separable synthesis): one late window against the oracle
    kw0 = {k: v for k, v in kw.items() if k != "boost_velocity"}
    out0 = _gpu_wm(t, data, 16, h, ctx).transform(**kw0)
    assert _window_check(out0.t, out0.data, t, data, kw0, 16, [(N - 700, 700)]) < 1e-12


def test_cfg2_full_size_rotation_then_supertranslation(ctx):
    """BASELINE.json configs[1] at full size (l <= 8, 1e5 steps): the series rotation (seed 4), then the supertranslation.
    Size-independent properties of the rotation (unitarity per l block, R then R^-1), and WINDOWS of both full-size
    results against the oracle chain (rotations_ref.rotate_by_series, then grid_ref.transform) run on input slices."""
    from oracle import rotations_ref
    from scri_amd import synthetic

    t, data, spec = synthetic.workload("cfg2")
    assert data.shape == (N, 77)
    R = synthetic.rotor_series(t, spec["rotation_seed"], omega=8 * np.pi / (t[-1] - t[0]), q0=synthetic.Q1234)
    w = _gpu_wm(t, data.copy(), 8, h, ctx)
    w.rotate_decomposition_basis(R)
    for ell in range(2, 9):  # |f_l|^2 is invariant under rotations
        blk = slice(ell * ell - 4, (ell + 1) ** 2 - 4)
        n0 = (np.abs(data[:, blk]) ** 2).sum(axis=1)
        n1 = (np.abs(w.data[:, blk]) ** 2).sum(axis=1)
        assert np.abs(n1 - n0).max() < 1e-13 * n0.max()
    assert np.abs(w.data - data).max() > 1e-3
    rotated = w.data.copy()
    out = w.transform(**spec["kwargs"])
    assert N - 10 < out.n_times <= N and np.all(np.diff(out.t) > 0)
    for i0 in (0, 31_000, N - 400):
        sl = slice(i0, i0 + 400)
        rot_o = rotations_ref.rotate_by_series(data[sl], quat.as_spinor_array(R[sl]), 2, 8)
        assert np.abs(rotated[sl] - rot_o).max() < 8e-13 * np.abs(data).max()  # the suite's 1e-13 l_max bar
        _window_check(out.t, out.data, t[sl], rot_o, spec["kwargs"], 8, [(0, 400)])
    w.rotate_decomposition_basis(quat.qconj(R))
    assert np.abs(w.data - data).max() < 2e-13 * np.abs(data).max()


def test_cfg1_rotations_match_oracle(ctx):
    """BASELINE.json configs[0]: l = 2..4, 2000 steps, constant rotor and rotor series, through
    bms_rotate_const / bms_rotate_series against the restated numba kernels (scri/rotations.py:346-392)."""
    from oracle import rotations_ref, wigner
    from scri_amd import engine, synthetic

    t, data, rot = synthetic.cfg1()
    assert data.shape == (2000, 21)
    q = rot["constant"]
    Ra, Rb = quat.as_spinor_array(q)
    expect = rotations_ref.rotate_by_constant(data, 2, 4, wigner.wigner_D_matrices(Ra, Rb, 2, 4))
    got = engine.rotate_const(data.copy(), 2, 4, q, ctx=ctx)
    assert np.abs(got - expect).max() < 4e-13 * np.abs(data).max()
    sp = quat.as_spinor_array(rot["series"])
    expect = rotations_ref.rotate_by_series(data, sp, 2, 4)
    got = engine.rotate_series(data.copy(), 2, 4, sp, ctx=ctx)
    assert np.abs(got - expect).max() < 4e-13 * np.abs(data).max()
    # the same through the scri-compatible class (frame bookkeeping of rotations.py:313-335)
    w = _gpu_wm(t, data.copy(), 4, h, ctx)
    w.rotate_decomposition_basis(rot["series"])
    assert np.array_equal(w.data, got) and np.abs(w.frame - rot["series"]).max() == 0.0


@pytest.fixture(scope="module")
def cfg4(ctx):
    """BASELINE.json configs[3] on ONE GPU: l <= 16, 1e6 steps (4.56 GB of modes), walked in chunks of the work space."""
    from scri_amd import synthetic

    t, data, spec = synthetic.workload("cfg4")
    assert data.shape == (1_000_000, 285)
    out = _gpu_wm(t, data, 16, h, ctx).transform(**spec["kwargs"])
    return t, data, spec["kwargs"], out


def test_cfg4_windows_match_oracle(cfg4):
    """Start / middle / end of the 1e6-step output against the oracle on input slices.  The boost skews the time axis of a
    direction by up to beta |u| / dt = 374 rows at the end of the series, which the oracle's window trims from each side of
    its slice: the late slices are longer."""
    t, data, kw, out = cfg4
    assert 999_500 < out.n_times <= 1_000_000 and np.all(np.diff(out.t) > 0)
    _window_check(out.t, out.data, t, data, kw, 16, [(0, 400), (500_000, 900), (1_000_000 - 1300, 1300)])


def test_cfg4_eight_shards_equal_whole(cfg4, ctx):
    """The 8 time shards of sharding.plan (each with its own halo rows only), run one after the other on this GPU,
    reassemble to the one-GPU result."""
    from scri_amd import engine, sharding

    t, data, kw, out = cfg4
    tr = engine.make_transformation(kw["supertranslation"], kw["frame_rotation"], kw["boost_velocity"], 37, 37, 16)
    have, need, window = sharding.plan(t, tr, 8)
    assert window[1] - window[0] == out.n_times
    scale = np.abs(out.data).max()
    row = 0
    for r in range(8):
        ext = data[need[r][0] : need[r][1]]
        tp, dp, first = engine.transform_modes(t, ext, 2, 16, -2, -1, engine.BMS_TERM_H, tr, ctx=ctx,
                                               shard=(need[r][0], ext.shape[0], have[r][0], have[r][1]))
        assert first == window[0] + row
        assert np.array_equal(tp, out.t[row : row + tp.shape[0]])
        assert np.abs(dp - out.data[row : row + dp.shape[0]]).max() < 1e-14 * scale
        row += dp.shape[0]
    assert row == out.n_times


def _abd_window_check(got_u, got_raw, kw, ell_max, windows, margin=60):
    """As _window_check for the six ABD fields of cfg5.  The oracle's input rows are generated on their own from the
    definition of the global series, so a window may straddle the shard's edge: the comparison then covers the rows the
    shard produced (those are the ones that needed its halo)."""
    from oracle import abd_ref
    from oracle.containers import ABD
    from scri_amd import synthetic

    for i0, rows in windows:
        u_in, raw_in, _ = synthetic.abd_workload("cfg5", rows=(i0, i0 + rows))
        e = abd_ref.transform(ABD(u_in[i0 : i0 + rows], raw_in, ell_max), **kw)
        sel = np.zeros(e.n_times, dtype=bool)
        sel[margin:-margin] = True
        sel &= (e.u >= got_u[0] - 1e-9) & (e.u <= got_u[-1] + 1e-9)
        assert sel.sum() > 100, (i0, e.n_times, int(sel.sum()))
        keep = e.u[sel]
        gi = np.searchsorted(got_u, keep - 1e-9)
        assert np.abs(got_u[gi] - keep).max() < 1e-13 * max(1.0, abs(keep[-1])), i0
        scale = max(1.0, np.abs(e.raw).max())
        for f, name in enumerate(("psi0", "psi1", "psi2", "psi3", "psi4", "sigma")):
            err = np.abs(got_raw[f][gi] - e.raw[f][sel]).max()
            # The reference forms each direction's spline abscissae as k (u - alpha) (transformations.py:408): at |u| ~ 2e4
            # their rounding, eps |u|, is a time shift that moves the result by eps |u| |df'/du'| -- 7e-11 for psi0' at
            # l <= 8, ~1e-9 at l <= 24, where the Horner mixing has multiplied psi4 by X^4 ~ (beta u)^4 ~ 2e3 (measured with
            # the oracle alone: one ulp on its abscissae changes its own output by that much).  The reference's result is
            # not defined more sharply than that, so the bar is the parity 1e-12 plus a few of those roundings.
            noise = 8 * np.finfo(float).eps * np.abs(e.u).max() * np.abs(np.gradient(e.raw[f], e.u, axis=0)).max()
            assert err < 1e-12 * scale + noise, (i0, name, err, noise)


@pytest.mark.parametrize("rank", [0, 7])
def test_cfg5_shard_six_fields_match_oracle(ctx, rank):
    """BASELINE.json configs[4]: AsymptoticBondiData, all six fields non-zero, l <= 24, 2e5 steps, working_ell_max = 49 ->
    99 x 99 grid; the 25 000-step time shard of rank 0 and of rank 7 (largest time skew) of the 8-GPU plan, each from its
    own rows + halo, against the oracle (scri/asymptotic_bondi_data/transformations.py:199-431 restated) on input slices."""
    from scri_amd import engine, sharding, synthetic

    spec = synthetic.CONFIGS["cfg5"]
    # working_ell_max = 49 -> the 99 x 99 grid (the default would be 2 l_max + l_max of the
    # supertranslation = 50); passed explicitly to the oracle, built into `tr` for the engine
    kw = dict(spec["kwargs"], working_ell_max=49)
    n, L = spec["n_times"], spec["ell_max"]
    u = np.arange(n) * spec["dt"]
    n_theta = 2 * kw["working_ell_max"] + 1
    tr = engine.make_transformation(kw["supertranslation"], kw["frame_rotation"], kw["boost_velocity"], n_theta, n_theta, L)
    have, need, window = sharding.plan(u, tr, 8)
    assert have[rank][1] - have[rank][0] == 25_000
    _, raw, _ = synthetic.abd_workload("cfg5", rows=need[rank])
    assert all(np.abs(raw[f]).max() > 0.01 for f in range(6))
    got_u, got_raw, first = engine.transform_abd(u, raw, L, tr, ctx=ctx,
                                                 shard=(need[rank][0], raw.shape[1], have[rank][0], have[rank][1]))
    lo, hi = max(have[rank][0], window[0]), min(have[rank][1], window[1])
    assert abs(first - lo) <= 1 and abs(got_u.shape[0] - (hi - lo)) <= 1 and got_raw.shape == (6, got_u.shape[0], 625)
    if rank == 0:
        windows = [(0, 300), (have[0][1] - 380, 480)]  # start of the series; across the shard's last row
    else:
        windows = [(have[7][0] - 100, 480), (n - 480, 480)]  # across the shard's first row; end of the series
    _abd_window_check(got_u, got_raw, kw, L, windows)


def test_cfg5_grid_boosted_schwarzschild_every_time_step(ctx):
    """AsymptoticBondiData at the cfg5 grid (l_max = 24, working_ell_max = 49 -> 99 x 99, 25 000 steps = one GPU's shard):
    the Bondi four-momentum of boosted Schwarzschild data is m gamma (1, -v) at every output time."""
    import scri_amd

    mass, L, n = 0.789, 24, 25_000
    u = np.arange(n) * 0.1
    a = scri_amd.AsymptoticBondiData(u, L, ctx=ctx)
    a._raw_data[2, :, 0] = -mass * np.sqrt(4 * np.pi)
    v = np.array([0.03, -0.02, 0.05])
    ap = a.transform(boost_velocity=v)
    assert ap.n_times > n - 2000 and ap.ell_max == L  # the boost trims |v| u_max / dt ~ 1500 samples
    gamma = 1 / np.sqrt(1 - v @ v)
    P = ap.bondi_four_momentum()
    assert np.abs(P - mass * gamma * np.array([1, *-v])[None, :]).max() < 1e-13
    assert np.abs(ap._raw_data[[3, 4, 5]]).max() < 1e-13  # no shear, no radiation (psi1', psi0' pick up u eth psi2 terms)


def test_cfg5_whole_series_equals_its_eight_shards_and_oracle_at_interior_seams(ctx):
    """BASELINE.json configs[4] as a whole: the full 2e5-step series of the six fields (12 GB of modes, l <= 24, 99 x 99 grid) on ONE
    GPU, resident in HBM, against (i) the 8 time shards of `sharding.plan`, each transformed from its own rows + halo only -- what the
    8 ranks of the sharded run compute -- to 1e-14 of the data's scale, every seam included, and (ii) the oracle
    (scri/asymptotic_bondi_data/transformations.py:199-431 restated) on windows across two interior seams, ranks 3|4 and 5|6."""
    import torch

    from scri_amd import engine, sharding, synthetic

    spec = synthetic.CONFIGS["cfg5"]
    kw = dict(spec["kwargs"], working_ell_max=49)
    n, L = spec["n_times"], spec["ell_max"]
    nm = (L + 1) ** 2
    u = np.arange(n) * spec["dt"]
    n_theta = 2 * kw["working_ell_max"] + 1
    tr = engine.make_transformation(kw["supertranslation"], kw["frame_rotation"], kw["boost_velocity"], n_theta, n_theta, L)
    have, need, window = sharding.plan(u, tr, 8)
    dev = torch.device("cuda", ctx.device)
    d_in = torch.empty((6, n, nm), dtype=torch.complex128, device=dev)
    for r in range(8):  # the series is generated and uploaded shard by shard: 1.5 GB of host memory at a time
        _, rows, _ = synthetic.abd_workload("cfg5", rows=have[r])
        d_in[:, have[r][0] : have[r][1]] = torch.from_numpy(rows).to(dev)
        del rows
    d_out = torch.empty((6, n, nm), dtype=torch.complex128, device=dev)
    torch.cuda.synchronize()
    u_out, n_new = engine.transform_abd(u, d_in.data_ptr(), L, tr, ctx=ctx, device=True, out_ptr=d_out.data_ptr())
    assert n_new == window[1] - window[0] and n_new > n - 1000
    scale = float(d_out[:, :n_new].abs().max())
    assert scale > 0.1
    # (i) the eight shards
    row = 0
    d_part = torch.empty((6, have[0][1] - have[0][0], nm), dtype=torch.complex128, device=dev)
    for r in range(8):
        ext = d_in[:, need[r][0] : need[r][1]].contiguous()
        torch.cuda.synchronize()
        up, nr, first = engine.transform_abd(u, ext.data_ptr(), L, tr, ctx=ctx, device=True, out_ptr=d_part.data_ptr(),
                                             shard=(need[r][0], ext.shape[1], have[r][0], have[r][1]))
        assert first == window[0] + row and np.array_equal(up, u_out[row : row + nr])
        worst = float((d_part[:, :nr] - d_out[:, row : row + nr]).abs().max())
        assert worst <= 1e-14 * scale, (r, worst)
        row += nr
        del ext
    assert row == n_new
    # (ii) the oracle across the seams 3|4 and 5|6 (a slice of the whole result around each seam)
    for seam in (have[4][0], have[6][0]):
        lo, hi = seam - 700 - window[0], seam + 700 - window[0]
        got_raw = d_out[:, lo:hi].cpu().numpy()
        _abd_window_check(u_out[lo:hi], got_raw, kw, L, [(seam - 240, 480)])


def test_cfg5_shard_size_without_boost_fused_route_equals_dense(ctx, monkeypatch, route):
    """One GPU's share of cfg5 (25 000 steps, six fields, l <= 24, 99 x 99 grid) under the workload's supertranslation and frame rotation
    WITHOUT its boost: elimination on the modes + two-kernel separable synthesis + phi stage fused with the mixing
    (kernels_synthesis_large.hip) against the six dense sYlm products + mixing and elimination on the grid (the route the oracle windows
    of the boosted tests above pin at this size), and against the unfused separable route."""
    import torch

    from scri_amd import engine, synthetic

    spec = synthetic.CONFIGS["cfg5"]
    kw = spec["kwargs"]
    n, L = 25_000, spec["ell_max"]
    u, raw, _ = synthetic.abd_workload("cfg5", n_times=n)
    tr = engine.make_transformation(kw["supertranslation"], kw["frame_rotation"], [0, 0, 0], 99, 99, L)
    dev = torch.device("cuda", ctx.device)
    d_in = torch.from_numpy(raw).to(dev)
    outs = {}
    for which, env in (("fused", {}), ("separable", {"SCRI_AMD_NO_FUSED_ABD_MIX": "1"}), ("dense", {"SCRI_AMD_NO_SEPARABLE_SYNTHESIS": "1"})):
        for k in ("SCRI_AMD_NO_FUSED_ABD_MIX", "SCRI_AMD_NO_SEPARABLE_SYNTHESIS"):
            route(k, None)
        for k, v in env.items():
            route(k, v)
        d_out = torch.empty_like(d_in)
        torch.cuda.synchronize()
        u_out, n_new = engine.transform_abd(u, d_in.data_ptr(), L, tr, ctx=ctx, device=True, out_ptr=d_out.data_ptr())
        outs[which] = (u_out, d_out[:, :n_new])
    assert outs["dense"][1].shape[1] > n - 10
    scale = float(outs["dense"][1].abs().max())
    for route in ("fused", "separable"):
        assert np.array_equal(outs[route][0], outs["dense"][0])
        worst = float((outs[route][1] - outs["dense"][1]).abs().max())
        assert worst < 2e-13 * scale, (route, worst / scale)
