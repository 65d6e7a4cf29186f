"""GPU parity of AsymptoticBondiData.transform (bms_transform_abd) against the CPU oracle."""
import numpy as np
import pytest

from oracle import abd_ref, wigner
from oracle.containers import ABD

pytestmark = pytest.mark.gpu


def smooth_abd(n, ell_max, seed, t0=-15.0, t1=25.0):
    rng = np.random.default_rng(seed)
    u = np.linspace(t0, t1, n)
    nm = (ell_max + 1) ** 2
    LM = wigner.LM_range(0, ell_max)
    raw = np.zeros((6, n, nm), dtype=complex)
    phase = 0.07 * u + 3e-4 * u**2
    for i, s in enumerate(ABD.spins):
        a = (rng.normal(size=nm) + 1j * rng.normal(size=nm)) * 10.0 ** (-LM[:, 0] / 4.0)
        a[: s * s] = 0
        raw[i] = a[None, :] * np.exp(1j * LM[None, :, 1] * phase[:, None]) * (1 + 0.01 * u[:, None])
    return ABD(u, raw, ell_max)


def real_st(ell_max, seed, scale):
    rng = np.random.default_rng(seed)
    a = scale * (rng.normal(size=(ell_max + 1) ** 2) + 1j * rng.normal(size=(ell_max + 1) ** 2))
    return a  # the ABD flavour imposes reality itself


CASES = [
    dict(),
    dict(time_translation=0.6),
    dict(space_translation=[0.2, 0.1, -0.3]),
    dict(boost_velocity=[0.02, -0.01, 0.03]),
    dict(supertranslation="st", frame_rotation=[0.4, 1, -2, 0.3], boost_velocity=[3e-3, 1e-3, -2e-3]),
    dict(supertranslation="st", working_ell_max=7, output_ell_max=3),
]


@pytest.mark.parametrize("n", [2, 3])
@pytest.mark.parametrize("case", [1, 3, 4])
def test_abd_transform_of_two_and_three_samples(ctx, n, case):
    """scipy's CubicSpline -- the interpolant of AsymptoticBondiData.transform (transformations.py:398-411) -- takes series of 2
    and 3 samples (the line / parabola through them); so does the engine (`short_series_eval_kernel`).  The window of such a
    series is non-empty only while the time skew stays within its few samples: small transformations."""
    import scri_amd

    kw = dict(CASES[case])
    if kw.get("supertranslation") == "st":
        kw["supertranslation"] = real_st(2, 33, 1e-3)
    if "time_translation" in kw:
        kw["time_translation"] = 1e-3
    o = smooth_abd(n, 4, 70 + case, t0=-1.0, t1=1.0)
    try:
        expect = abd_ref.transform(o, **{k: (np.array(v) if isinstance(v, list) else v) for k, v in kw.items()})
    except Exception as e:  # the oracle's own window came out empty: nothing to compare
        pytest.skip(f"oracle: {e}")
    g = scri_amd.AsymptoticBondiData(o.u, o.ell_max, ctx=ctx)
    g._raw_data[:] = o.raw
    got = g.transform(**kw)
    assert got.n_times == expect.n_times
    if expect.n_times:
        assert np.abs(got.u - expect.u).max() < 1e-13
        scale = max(1.0, np.abs(expect.raw).max())
        assert np.abs(got._raw_data - expect.raw).max() < 1e-12 * scale


@pytest.mark.parametrize("case", range(len(CASES)))
def test_abd_transform_matches_oracle(ctx, case):
    import scri_amd

    kw = dict(CASES[case])
    if kw.get("supertranslation") == "st":
        kw["supertranslation"] = real_st(2, 33, 0.05)
    o = smooth_abd(300, 4, 50 + case)
    expect = abd_ref.transform(o, **{k: (np.array(v) if isinstance(v, list) else v) for k, v in kw.items()})
    g = scri_amd.AsymptoticBondiData(o.u, o.ell_max, ctx=ctx)
    g._raw_data[:] = o.raw
    got = g.transform(**kw)
    assert got.n_times == expect.n_times and got.ell_max == expect.ell_max
    assert np.abs(got.u - expect.u).max() < 1e-13
    scale = max(1.0, np.abs(expect.raw).max())
    for i, name in enumerate(("psi0", "psi1", "psi2", "psi3", "psi4", "sigma")):
        err = np.abs(getattr(got, name) - expect.raw[i]).max()
        assert err < 1e-12 * scale, (name, err)


def test_abd_interpolate_matches_oracle(ctx):
    import scri_amd

    o = smooth_abd(200, 3, 7)
    g = scri_amd.AsymptoticBondiData(o.u, o.ell_max, ctx=ctx)
    g._raw_data[:] = o.raw
    tn = np.linspace(o.u[3], o.u[-5], 333)
    assert np.abs(g.interpolate(tn)._raw_data - o.interpolate(tn).raw).max() < 1e-12


def test_abd_WaveformModes(ctx):
    """The reference's tests/test_asymptoticbondidata.py:165-212 on the GPU: a general BMS transformation of a random
    AsymptoticBondiData object (Kerr-Schild + quadratic-in-time shear, from initial values) agrees with the same
    transformation of its strain h = 2 conj(sigma) as a WaveformModes object -- two different pipelines (six fields with
    Horner mixing vs. one field with the inhomogeneous h term), at the reference's tolerance."""
    import scri_amd
    from tests.test_gpu_transform_modes import real_supertranslation

    tolerance = 4e-12
    rng = np.random.default_rng(123)
    mass, spin, ell_max = 1.23, 0.35, 6
    nm = (ell_max + 1) ** 2
    u = np.linspace(-10, 10, num=1_000)
    psi2, psi1, psi0 = (np.zeros(nm, dtype=complex) for _ in range(3))
    psi2[0] = -mass * np.sqrt(4 * np.pi)
    psi1[2] = -np.sqrt(2) * (3j * spin / 2) * (np.sqrt((8 / 3) * np.pi))
    psi0[6] = 2 * (3 * spin**2 / mass / 2) * (np.sqrt((32 / 15) * np.pi))

    def shear(scale):
        a = scale * ((rng.random(nm) - 0.5) + 1j * (rng.random(nm) - 0.5))
        a[:4] = 0
        return a

    abd = scri_amd.AsymptoticBondiData.from_initial_values(
        u, ell_max=ell_max, sigma0=shear(0.01), sigmadot0=shear(0.0002), sigmaddot0=shear(0.00003), psi2=psi2, psi1=psi1, psi0=psi0, ctx=ctx
    )
    h = scri_amd.WaveformModes(
        t=abd.t, data=2 * abd.sigma.bar.ndarray.copy(), ell_min=0, ell_max=ell_max, frameType=scri_amd.Inertial, dataType=scri_amd.h,
        r_is_scaled_out=True, m_is_scaled_out=True, ctx=ctx,
    )
    alpha = real_supertranslation(ell_max, 7, 0.01)
    R = rng.normal(size=4)
    R /= np.linalg.norm(R)
    v = 0.01 * (rng.random(3) - 0.5)
    abdprime = abd.transform(supertranslation=alpha, frame_rotation=R, boost_velocity=v)
    hprime = h.transform(supertranslation=alpha, frame_rotation=R, boost_velocity=v)
    lo, hi = max(hprime.t[0], abdprime.t[0]), min(hprime.t[-1], abdprime.t[-1])
    t = hprime.t[(hprime.t >= lo) & (hprime.t <= hi)]
    assert t.size > 900
    hprime = hprime.interpolate(t)
    abdprime = abdprime.interpolate(t)
    assert np.allclose(hprime.data, 2 * abdprime.sigma.bar.ndarray[:, 4:], atol=tolerance, rtol=tolerance)


def test_abd_conformal_factors(ctx):
    """The reference's tests/test_asymptoticbondidata.py:33-93: conformal_factors(boost, boosted_grid) against the
    band-limited route -- k and 1/k analysed on the undistorted grid (map2salm on the GPU), eth applied in mode space,
    and the results evaluated on the distorted rotors (SWSH_grid on the GPU) -- at the reference's tolerance."""
    from scri_amd import asymptotic_bondi_data as abd_mod
    from scri_amd import engine

    tolerance = 4e-14
    ell_max = 32
    n_theta = n_phi = 2 * ell_max + 1
    v = np.array([0.01, 0.02, 0.03])
    gamma = 1 / np.sqrt(1 - v @ v)
    rotors = abd_mod.boosted_grid([1.0, 0, 0, 0], v, n_theta, n_phi)
    k, ethk_over_k, one_over_k, one_over_k_cubed = abd_mod.conformal_factors(v, rotors)
    assert k.shape == ethk_over_k.shape == one_over_k.shape == one_over_k_cubed.shape == (1, n_theta, n_phi)
    theta = np.pi * np.arange(n_theta) / (n_theta - 1)
    phi = 2 * np.pi * np.arange(n_phi) / n_phi
    th, ph = np.meshgrid(theta, phi, indexing="ij")
    kinv_grid = gamma * (1 - v[0] * np.sin(th) * np.cos(ph) - v[1] * np.sin(th) * np.sin(ph) - v[2] * np.cos(th))
    kinv_modes = engine.map2salm(kinv_grid + 0j, 0, ell_max, ctx=ctx)
    k_modes = engine.map2salm(1 / kinv_grid + 0j, 0, ell_max, ctx=ctx)
    Y0 = engine.swsh_grid(rotors.reshape(-1, 4), 0, 0, ell_max, ctx=ctx)
    Y1 = engine.swsh_grid(rotors.reshape(-1, 4), 1, 0, ell_max, ctx=ctx)
    one_over_k2 = (Y0 @ kinv_modes).reshape(n_theta, n_phi)
    ell = np.concatenate([np.full(2 * l + 1, l) for l in range(ell_max + 1)])
    eth_k_modes = k_modes * np.sqrt(ell * (ell + 1.0)) / np.sqrt(2.0)  # eth_GHP on spin 0
    ethk2 = (Y1 @ eth_k_modes).reshape(n_theta, n_phi)
    k2 = 1 / one_over_k2
    assert np.allclose(one_over_k[0], one_over_k2, atol=tolerance, rtol=tolerance)
    assert np.allclose(k[0], k2, atol=tolerance, rtol=tolerance)
    assert np.allclose(one_over_k_cubed[0], one_over_k2**3, atol=tolerance, rtol=tolerance)
    # (the band-limited route itself carries ~1e-13 of rounding here: 1089-term sums of the GPU SWSH values)
    assert np.allclose(ethk_over_k[0], ethk2 / k2, atol=2e-13, rtol=tolerance)


@pytest.mark.parametrize("boosted", [True, False])
def test_abd_host_pipeline_equals_one_call(ctx, monkeypatch, boosted):
    """A long AsymptoticBondiData series in host memory goes through bms_transform_abd_pipelined (time shards: upload, kernels,
    download side by side); the result must equal the one-call path (SCRI_AMD_NO_PIPELINE) to rounding, whole window included --
    with a boost (dense products) and without one (elimination on the modes of every piece from the knot tables of the whole series,
    separable synthesis, fused mixing)."""
    import scri_amd
    from scri_amd import engine

    o = smooth_abd(30000, 6, 91, t0=-400.0, t1=500.0)  # 6 x 30000 x 49 x 16 B = 141 MB
    assert o.raw.nbytes >= engine.PIPELINE_MIN_BYTES
    kw = dict(supertranslation=real_st(2, 33, 0.05), frame_rotation=[0.4, 1, -2, 0.3])
    if boosted:
        kw["boost_velocity"] = [3e-3, 1e-3, -2e-3]

    def run():
        g = scri_amd.AsymptoticBondiData(o.u, o.ell_max, ctx=ctx)
        g._raw_data[:] = o.raw
        return g.transform(**kw)

    piped = run()
    monkeypatch.setenv("SCRI_AMD_NO_PIPELINE", "1")
    whole = run()
    assert piped.n_times == whole.n_times and np.array_equal(piped.u, whole.u)
    scale = max(1.0, np.abs(whole._raw_data).max())
    assert np.abs(piped._raw_data - whole._raw_data).max() < 1e-13 * scale


@pytest.mark.parametrize("n,ell_max,beta", [(300, 4, 3e-3), (2500, 8, 2e-2), (9, 3, 1e-3), (700, 12, 0.2)])
def test_sigma_through_the_evaluating_product_equals_the_grid_route(ctx, monkeypatch, n, ell_max, beta, route):
    """With a boost the six fields are synthesised by dense products; sigma' = (sigma - eth eth alpha) / k mixes with nothing, so its
    spline is solved on the modes and evaluated in the epilogue of its product (zgemm3m_eval_kernel), while psi0 .. psi4 go through the
    mixing + elimination pass and the back substitution as before.  Same results as with all six fields on the grid route
    (SCRI_AMD_NO_ABD_SIGMA_EVAL), and the oracle for the small cases."""
    import scri_amd

    o = smooth_abd(n, ell_max, 90 + n)
    kw = dict(supertranslation=real_st(2, 35, 0.05), frame_rotation=np.array([0.4, 1, -2, 0.3]), boost_velocity=np.array([0.3, 0.1, -0.2]) * beta / 0.374)
    route("SCRI_AMD_NO_SEPARABLE_SYNTHESIS", None)

    def run():
        g = scri_amd.AsymptoticBondiData(o.u, o.ell_max, ctx=ctx)
        g._raw_data[:] = o.raw
        ctx.enable_timing(True)
        ctx.get_timing(reset=True)
        out = g.transform(**kw)
        tm = ctx.get_timing(reset=True)
        ctx.enable_timing(False)
        return out, tm

    route("SCRI_AMD_NO_ABD_SIGMA_EVAL", None)
    got, tm = run()
    route("SCRI_AMD_NO_ABD_SIGMA_EVAL", "1")
    ref, tm_ref = run()
    assert got.n_times == ref.n_times and np.array_equal(got.u, ref.u)
    if n >= 8 and got.n_times:
        assert tm["spline_backward"][1] < tm_ref["spline_backward"][1]  # five back substitutions instead of six
    scale = max(1.0, np.abs(ref._raw_data).max())
    assert np.abs(got._raw_data - ref._raw_data).max() < 1e-13 * scale
    if n <= 700:
        e = abd_ref.transform(o, **kw)
        assert got.n_times == e.n_times
        if e.n_times:
            assert np.abs(got._raw_data - e.raw).max() < 1e-12 * max(1.0, np.abs(e.raw).max())
