"""Two-kernel separable synthesis (kernels_synthesis_large.hip) -- the route of every boost-free transformation the one-kernel
form does not take: AsymptoticBondiData.transform on its working grids (scri/asymptotic_bondi_data/transformations.py:312-334
with beta = 0, e.g. the supertranslation and rotation steps of map_to_superrest_frame.py:443,610,641), the psi-mixing
WaveformModes types (scri/waveform_grid.py:504-550) and l_max > 16.  Checked against the dense sYlm product of the same
library (to rounding) and against the oracle."""
import numpy as np
import pytest

from oracle import abd_ref, waveform_grid_ref as grid_ref
from oracle.containers import WM, psi2 as o_psi2, psi3 as o_psi3, h as o_h
from tests.test_gpu_transform_abd import real_st, smooth_abd

pytestmark = pytest.mark.gpu

ROTOR = np.array([0.4, 1.0, -2.0, 0.3]) / np.linalg.norm([0.4, 1.0, -2.0, 0.3])


def _timing_tags(ctx):
    return {k for k, v in ctx.get_timing(reset=True).items() if v[1]}


@pytest.mark.parametrize("ell_max,n,rotated,working", [(4, 300, False, None), (4, 301, True, None), (12, 96, True, None), (24, 40, True, None),
                                                      (24, 9, False, None), (6, 64, True, 9), (3, 3, True, None), (16, 50, True, 51)])
def test_abd_boost_free_separable_equals_dense_and_oracle(ctx, monkeypatch, ell_max, n, rotated, working, route):
    import scri_amd

    o = smooth_abd(n, ell_max, 100 + ell_max + n, t0=-1.0 if n < 4 else -15.0, t1=1.0 if n < 4 else 25.0)
    kw = dict(supertranslation=real_st(min(ell_max, 3), 5, 1e-3 if n < 4 else 0.05))
    if rotated:
        kw["frame_rotation"] = ROTOR
    if working:
        kw["working_ell_max"] = working

    def run():
        g = scri_amd.AsymptoticBondiData(o.u, o.ell_max, ctx=ctx)
        g._raw_data[:] = o.raw
        return g.transform(**kw)

    route("SCRI_AMD_NO_SEPARABLE_SYNTHESIS", None)
    ctx.enable_timing(True)
    ctx.get_timing(reset=True)
    got = run()
    tags = _timing_tags(ctx)
    if got.n_times:  # (a window can come out empty for the 2-sample series: nothing is synthesised then)
        assert ("rotate" in tags) == rotated  # the separable route: the modes were rotated instead of the grid
    route("SCRI_AMD_NO_SEPARABLE_SYNTHESIS", "1")
    ref = run()
    assert "rotate" not in _timing_tags(ctx)
    ctx.enable_timing(False)
    assert got.n_times == ref.n_times > 0 and np.array_equal(got.u, ref.u)
    scale = max(1.0, np.abs(ref._raw_data).max())
    assert np.abs(got._raw_data - ref._raw_data).max() < 1e-13 * scale * max(1.0, ell_max / 8.0)
    if ell_max <= 12:  # (the oracle's per-pixel loop: seconds up to here)
        expect = abd_ref.transform(o, **kw)
        assert got.n_times == expect.n_times
        assert np.abs(got._raw_data - expect.raw).max() < 1e-12 * scale


@pytest.mark.parametrize("data_type,ell_max,rotated", [("psi2", 6, True), ("psi3", 8, False), ("psi2", 18, True), ("h", 20, True), ("h", 24, False)])
def test_waveform_modes_boost_free_psi_types_and_large_ell(ctx, monkeypatch, data_type, ell_max, rotated, route):
    """psi2 needs psi3 and psi4, psi3 needs psi4 (waveform_grid.py:417-426): each companion is synthesised with its own spin; h with
    l_max > 16 takes the same kernels with the elimination on the modes and the offset column."""
    import scri_amd
    from scri_amd import synthetic

    n = 120
    t = np.linspace(-30.0, 40.0, n)
    spins = {"psi2": 0, "psi3": -1, "psi4": -2, "h": -2}
    lmin = abs(spins[data_type])
    rng = np.random.default_rng(ell_max)
    data = synthetic.chirp_modes(t, lmin, ell_max, 3 + ell_max)
    st = synthetic.real_supertranslation(0.2 * (rng.normal(size=16) + 1j * rng.normal(size=16)))
    kw = dict(supertranslation=st)
    if rotated:
        kw["frame_rotation"] = ROTOR
    aux = {}
    if data_type == "psi2":
        aux = dict(psi3_modes=synthetic.chirp_modes(t, 1, ell_max, 11), psi4_modes=synthetic.chirp_modes(t, 2, ell_max, 12))
    elif data_type == "psi3":
        aux = dict(psi4_modes=synthetic.chirp_modes(t, 2, ell_max, 12))

    def wrap(name, d, lo):
        return scri_amd.WaveformModes(t=t, data=d, ell_min=lo, ell_max=ell_max, dataType=getattr(scri_amd, name), frameType=scri_amd.Inertial,
                                      r_is_scaled_out=True, m_is_scaled_out=True, ctx=ctx)

    def run():
        w = wrap(data_type, data, lmin)
        extra = {k: wrap(k[:4], v, abs(spins[k[:4]])) for k, v in aux.items()}
        return w.transform(**kw, **extra)

    route("SCRI_AMD_NO_SEPARABLE_SYNTHESIS", None)
    ctx.enable_timing(True)
    ctx.get_timing(reset=True)
    got = run()
    assert ("rotate" in _timing_tags(ctx)) == rotated
    route("SCRI_AMD_NO_SEPARABLE_SYNTHESIS", "1")
    ref = run()
    ctx.enable_timing(False)
    scale = max(1.0, np.abs(ref.data).max())
    assert got.n_times == ref.n_times and np.array_equal(got.t, ref.t)
    assert np.abs(got.data - ref.data).max() < 2e-13 * scale * max(1.0, ell_max / 8.0)
    if ell_max <= 8:
        otypes = {"psi2": o_psi2, "psi3": o_psi3, "h": o_h}
        from oracle.containers import psi4 as o_psi4

        otypes["psi4"] = o_psi4
        ow = WM(t=t, data=data, ell_min=lmin, ell_max=ell_max, dataType=otypes[data_type])
        oaux = {k: WM(t=t, data=v, ell_min=abs(spins[k[:4]]), ell_max=ell_max, dataType=otypes[k[:4]]) for k, v in aux.items()}
        expect = grid_ref.transform(ow, **kw, **oaux)
        assert got.n_times == expect.t.size
        assert np.abs(got.data - expect.data).max() < 1e-12 * scale


@pytest.mark.parametrize("seed", range(12))
def test_random_boost_free_shapes_equal_the_dense_route(ctx, monkeypatch, seed, route):
    """Random shapes for the two-kernel separable synthesis: WaveformModes (h and psi3 with its psi4 companion) on user grids with odd and
    even n_theta / n_phi up to the kernels' limits, AsymptoticBondiData with random working grids; with and without a frame rotation;
    every case against the dense products of the same library."""
    import scri_amd
    from scri_amd import synthetic

    rng = np.random.default_rng(1000 + seed)
    rotated = bool(rng.integers(0, 2))
    rot = rng.normal(size=4)
    rot /= np.linalg.norm(rot)
    st = synthetic.real_supertranslation(0.1 * (rng.normal(size=9) + 1j * rng.normal(size=9)))
    flavour = ("wm_h", "wm_psi3", "abd")[seed % 3]
    n = int(rng.integers(9, 60))
    t = np.sort(rng.uniform(-20.0, 30.0, n))
    t[1:] = np.maximum(t[1:], t[:-1] + 0.05)
    kw = dict(supertranslation=st)
    if rotated:
        kw["frame_rotation"] = rot

    if flavour == "abd":
        ell_max = int(rng.integers(2, 14))
        o = smooth_abd(n, ell_max, seed)
        o = type(o)(t, o.raw, ell_max)
        kw["working_ell_max"] = int(rng.integers(ell_max + 1, min(51, 3 * ell_max + 4) + 1))

        def run():
            g = scri_amd.AsymptoticBondiData(t, ell_max, ctx=ctx)
            g._raw_data[:] = o.raw
            r = g.transform(**kw)
            return r.u, r._raw_data
    else:
        ell_max = int(rng.integers(3, 30))
        n_theta = int(rng.integers(2 * ell_max + 1, 105))
        n_phi = int(rng.integers(2 * ell_max + 1, 128))
        kw.update(n_theta=n_theta, n_phi=n_phi)
        name = "h" if flavour == "wm_h" else "psi3"
        lmin = 2 if name == "h" else 1
        data = synthetic.chirp_modes(t, lmin, ell_max, seed)
        extra = {}

        def wrap(nm, d, lo):
            return scri_amd.WaveformModes(t=t, data=d, ell_min=lo, ell_max=ell_max, dataType=getattr(scri_amd, nm), frameType=scri_amd.Inertial,
                                          r_is_scaled_out=True, m_is_scaled_out=True, ctx=ctx)

        aux = synthetic.chirp_modes(t, 2, ell_max, seed + 50) if name == "psi3" else None

        def run():
            if aux is not None:
                extra["psi4_modes"] = wrap("psi4", aux, 2)
            r = wrap(name, data, lmin).transform(**kw, **extra)
            return r.t, r.data

    route("SCRI_AMD_NO_SEPARABLE_SYNTHESIS", None)
    t_sep, d_sep = run()
    route("SCRI_AMD_NO_SEPARABLE_SYNTHESIS", "1")
    t_den, d_den = run()
    assert np.array_equal(t_sep, t_den) and t_den.size > 0
    assert np.abs(d_sep - d_den).max() < 3e-13 * max(1.0, np.abs(d_den).max()) * max(1.0, ell_max / 8.0), (flavour, ell_max, kw.get("n_theta"), kw.get("working_ell_max"))
