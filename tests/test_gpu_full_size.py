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


def test_cfg2_rotation_round_trip_and_unitarity(ctx):
    from scri_amd import synthetic

    t, data, spec = synthetic.workload("cfg2")
    assert data.shape == (N, 77)
    rng = np.random.default_rng(4)
    axis = rng.normal(size=3)
    axis /= np.linalg.norm(axis)
    half = np.zeros((N, 4))
    half[:, 1:] = axis[None, :] * (8 * np.pi / (t[-1] - t[0]) * t / 2)[:, None]
    R = quat.qmul(quat.qexp(half), (np.array([1.0, 2, 3, 4]) / np.sqrt(30))[None, :])
    w = _gpu_wm(t, data.copy(), 8, h, ctx)
    w.rotate_decomposition_basis(R)
    for ell in range(2, 9):  # |f_l|^2 is invariant under rotations
        blk = slice(ell * ell - 4, (ell + 1) ** 2 - 4)
        n0 = (np.abs(data[:, blk]) ** 2).sum(axis=1)
        n1 = (np.abs(w.data[:, blk]) ** 2).sum(axis=1)
        assert np.abs(n1 - n0).max() < 1e-13 * n0.max()
    assert np.abs(w.data - data).max() > 1e-3
    # the second half of cfg2: supertranslate the rotated series; then undo the rotation of the untransformed copy
    out = w.transform(**spec["kwargs"])
    assert out.n_times > N - 10
    w.rotate_decomposition_basis(quat.qconj(R))
    assert np.abs(w.data - data).max() < 2e-13 * np.abs(data).max()


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
