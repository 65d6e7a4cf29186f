"""GPU parity tests of the individual kernels, through the C ABI, against the CPU oracle."""
import numpy as np
import pytest
from scipy.interpolate import CubicSpline

from oracle import quat, wigner, spinsfast_ref, rotations_ref, sample_waveforms_ref as samples
from oracle import waveform_grid_ref as grid_ref

pytestmark = pytest.mark.gpu


def _rotors(seed, n):
    rng = np.random.default_rng(seed)
    q = rng.normal(size=(n, 4))
    return q / np.linalg.norm(q, axis=1)[:, None]


@pytest.mark.parametrize("s", [-2, -1, 0, 1, 2])
def test_swsh_grid_matches_oracle(ctx, s):
    from scri_amd import engine

    R = np.concatenate([_rotors(1, 50), samples.Rs()])
    Y = engine.swsh_grid(R, s, 0, 17, ctx=ctx)
    Yo = wigner.swsh_grid(R, s, 17)
    assert Y.shape == Yo.shape
    assert np.abs(Y - Yo).max() < 2e-14


def test_rotor_grid_matches_oracle(ctx):
    from scri_amd import engine

    for fr, v in [([1, 0, 0, 0], [0, 0, 0]), ([1, 2, 3, 4], [0.01, -0.02, 0.03]), ([0.3, -1, 0.2, 0.5], [0, 0, 0.1])]:
        fr = np.array(fr, dtype=float) / np.linalg.norm(fr)
        R = engine.rotor_grid(fr, v, 9, 11, ctx=ctx)
        Ro = grid_ref.rotor_grid(fr, np.array(v, dtype=float), 9, 11)
        assert np.abs(R - Ro).max() < 1e-15


def test_wigner_D_matches_oracle(ctx):
    from scri_amd import engine

    for q in samples.Rs()[::9]:
        D = engine.wigner_D(q, 0, 8, ctx=ctx)
        Ra, Rb = quat.as_spinor_array(q)
        Do = wigner.wigner_D_matrices(Ra, Rb, 0, 8)
        assert np.abs(D - Do).max() < 1e-14


@pytest.mark.parametrize("ell_min,ell_max", [(2, 4), (0, 8), (2, 16), (0, 24)])
def test_rotate_series_matches_oracle(ctx, ell_min, ell_max):
    from scri_amd import engine

    rng = np.random.default_rng(5)
    n = 333
    nm = wigner.LM_total_size(ell_min, ell_max)
    data = rng.normal(size=(n, nm)) + 1j * rng.normal(size=(n, nm))
    R = _rotors(6, n)
    R[:100] = samples.Rs()  # special rotors: z-rotations, pi flips, identity-like
    sp = quat.as_spinor_array(R)
    expect = rotations_ref.rotate_by_series(data, sp, ell_min, ell_max)
    got = engine.rotate_series(data.copy(), ell_min, ell_max, sp, ctx=ctx)
    assert np.abs(got - expect).max() < 1e-13 * ell_max


@pytest.mark.parametrize("ell_min,ell_max,n", [(0, 0, 50), (1, 1, 17), (0, 3, 37), (3, 7, 16), (5, 11, 1), (7, 12, 100), (12, 15, 33),
                                               (14, 19, 41), (16, 19, 64), (15, 16, 23), (2, 16, 1600), (20, 24, 40), (22, 27, 19), (2, 24, 70),
                                               (26, 30, 21)])
def test_rotate_series_resident_kernel_shapes(ctx, ell_min, ell_max, n):
    """Every shape the LDS-resident rotation kernel distinguishes (kernels_rotate_resident.hip): row slots 4 / 6 / 8 / 10 per lane,
    12- and 8-wave builds, the side columns 16..27 of l >= 16 (one to three 4-column side products), ranges walked in several
    segments (2..19 | 20..24; 26..27 resident | 28..30 staged), single-l ranges (every step a new work unit), l from 0 (class B of
    padding rows only), series shorter than / not a multiple of the 16-step tile; special rotors mixed in."""
    from scri_amd import engine

    rng = np.random.default_rng(100 * ell_min + ell_max)
    nm = wigner.LM_total_size(ell_min, ell_max)
    data = rng.normal(size=(n, nm)) + 1j * rng.normal(size=(n, nm))
    R = _rotors(11, n)
    k = min(n, 100)
    R[:k:3] = samples.Rs()[:k:3][: len(R[:k:3])]
    sp = quat.as_spinor_array(R)
    expect = rotations_ref.rotate_by_series(data, sp, ell_min, ell_max)
    got = engine.rotate_series(data.copy(), ell_min, ell_max, sp, ctx=ctx)
    assert np.abs(got - expect).max() < 1e-13 * max(ell_max, 1)


def test_rotate_const_and_identity_bit_exact(ctx):
    from scri_amd import engine

    rng = np.random.default_rng(7)
    data = rng.normal(size=(1000, 77)) + 1j * rng.normal(size=(1000, 77))
    out = engine.rotate_const(data.copy(), 2, 8, [1.0, 0, 0, 0], ctx=ctx)
    assert np.array_equal(out, data)  # tests/test_rotations.py:14-38 of the reference
    q = np.array([1.0, 2, 3, 4]) / np.sqrt(30)
    Ra, Rb = quat.as_spinor_array(q)
    expect = rotations_ref.rotate_by_constant(data, 2, 8, wigner.wigner_D_matrices(Ra, Rb, 2, 8))
    got = engine.rotate_const(data.copy(), 2, 8, q, ctx=ctx)
    assert np.abs(got - expect).max() < 1e-13


@pytest.mark.parametrize("s", [-2, 0, 1])
def test_map2salm_matches_oracle(ctx, s):
    from scri_amd import engine

    rng = np.random.default_rng(8)
    n_theta, n_phi, L = 13, 15, 5
    f = rng.normal(size=(7, n_theta, n_phi)) + 1j * rng.normal(size=(7, n_theta, n_phi))  # not band limited
    a = engine.map2salm(f, s, L, ctx=ctx)
    ao = spinsfast_ref.map2salm(f, s, L)
    assert np.abs(a - ao).max() < 1e-13


def test_cubic_spline_matches_scipy(ctx):
    from scri_amd import engine

    rng = np.random.default_rng(9)
    for n in (4, 5, 37, 700, 5000):
        x = np.cumsum(rng.uniform(0.05, 0.2, size=n))
        y = np.sin(0.7 * x)[:, None] * (1 + 0.1 * np.arange(6)) + 1j * np.cos(0.3 * x)[:, None] * np.arange(6)
        xn = np.sort(rng.uniform(x[0], x[-1], size=2 * n + 3))
        xn[0], xn[-1] = x[0], x[-1]
        got = engine.cubic_spline(x, y, xn, ctx=ctx)
        expect = CubicSpline(x, y)(xn)
        assert np.abs(got - expect).max() < 2e-13, n


@pytest.mark.parametrize("ell_min,ell_max,n", [(2, 8, 333), (0, 16, 70), (3, 40, 19)])
def test_rotate_const_D_matches_oracle(ctx, ell_min, ell_max, n):
    """bms_rotate_const_D: the reference's numba kernel at its own signature (scri/rotations.py:346-367), the packed D
    matrices handed over by the caller; l = 40 spans two 64-column panels of the GEMM."""
    from oracle import rotations_ref
    from scri_amd import engine

    rng = np.random.default_rng(ell_max)
    data = rng.normal(size=(n, wigner.LM_total_size(ell_min, ell_max))) + 1j * rng.normal(size=(n, wigner.LM_total_size(ell_min, ell_max)))
    q = rng.normal(size=4)
    q /= np.linalg.norm(q)
    Ra, Rb = quat.as_spinor_array(q)
    D = wigner.wigner_D_matrices(Ra, Rb, ell_min, ell_max)
    expect = rotations_ref.rotate_by_constant(data, ell_min, ell_max, D)
    got = engine.rotate_const_D(data.copy(), ell_min, ell_max, D, ctx=ctx)
    assert np.abs(got - expect).max() < 1e-13 * ell_max
    # a strided view: only the mode columns change
    wide = np.zeros((n, data.shape[1] + 7), dtype=complex)
    wide[:, 2 : 2 + data.shape[1]] = data
    engine.rotate_const_D(wide[:, 2 : 2 + data.shape[1]], ell_min, ell_max, D, ctx=ctx)
    assert np.abs(wide[:, 2 : 2 + data.shape[1]] - expect).max() < 1e-13 * ell_max
    assert np.all(wide[:, :2] == 0) and np.all(wide[:, 2 + data.shape[1] :] == 0)
    with pytest.raises(ValueError):
        engine.rotate_const_D(data.copy(), ell_min, ell_max, D[:-1], ctx=ctx)


@pytest.mark.parametrize("ell_max,n", [(16, 100_000), (16, 40_007), (8, 100_000), (8, 70_001), (12, 33_333)])
def test_rotate_series_long_runs_and_the_split_last_round(ctx, ell_max, n):
    """Long series: the (tile, l group) units of the resident-table rotation kernel fill several rounds of the launch's waves and the
    partial last round is cut into per-l pieces (kernels_rotate_resident.hip).  Every row is rotated independently, so the oracle
    (scri/rotations.py:370-392 restated) runs on a sample of rows: the first and last tiles, every row of the tiles dealt in the last
    round, and 400 rows at random; all other rows are checked through the norm of each l block, which a rotation preserves."""
    import torch

    from oracle import quat, rotations_ref, wigner
    from scri_amd import engine

    rng = np.random.default_rng(ell_max + n)
    nm = wigner.LM_total_size(2, ell_max)
    data = rng.normal(size=(n, nm)) + 1j * rng.normal(size=(n, nm))
    q = rng.normal(size=(n, 4))
    q /= np.linalg.norm(q, axis=1)[:, None]
    sp = np.ascontiguousarray(quat.as_spinor_array(q))
    dev = torch.from_numpy(data).cuda()
    sp_dev = torch.from_numpy(sp).cuda()
    engine.rotate_device(dev.data_ptr(), n, nm, 2, ell_max, spinors_ptr=sp_dev.data_ptr(), ctx=ctx)
    ctx.synchronize()
    got = dev.cpu().numpy()
    # a rotation preserves the norm of every l block of every row
    for ell in range(2, ell_max + 1):
        a = ell * ell - 4
        before = np.linalg.norm(data[:, a : a + 2 * ell + 1], axis=1)
        after = np.linalg.norm(got[:, a : a + 2 * ell + 1], axis=1)
        assert np.abs(after - before).max() < 1e-12 * before.max(), ell
    n_tiles = (n + 15) // 16
    tail_tiles = min(n_tiles, 256 * 8)  # (tiles whose units can lie in the partial last round of 2 048 waves)
    rows = np.unique(np.concatenate([np.arange(0, 48), np.arange(max(0, n - 16 * tail_tiles), n)[:: max(1, tail_tiles // 64)],
                                     np.arange(n - 48, n), rng.integers(0, n, 400)]))
    expect = rotations_ref.rotate_by_series(data[rows].copy(), sp[rows], 2, ell_max)
    assert np.abs(got[rows] - expect).max() < 1e-13 * ell_max * np.abs(expect).max()
