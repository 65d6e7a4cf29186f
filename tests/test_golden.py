"""Golden vectors (tests/golden/): the oracle reproduces them on the CPU; the GPU engine reproduces them through the C ABI
(-m gpu).  g1-g7 come from tests/golden/make_golden.py (mpmath / sympy / the oracle); g8-g10 were computed by the
reference's OWN source files (/root/reference/scri/*.py imported unmodified on stand-ins for the third-party packages the
image lacks: tests/golden/make_golden_from_reference.py) -- they pin the scri layer of the oracle and of the HIP path:
kwarg handling, mixing signs and term order, trimming, frame bookkeeping.  Nothing here reads the reference checkout."""
import os

import numpy as np
import pytest

from oracle import quat, wigner, spinsfast_ref, abd_ref
from oracle import waveform_grid_ref as grid_ref
from oracle import sample_waveforms_ref as samples
from oracle.containers import ABD, WM, h, psi2, psi3, psi4, sigma

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load(name):
    return np.load(os.path.join(G, name), allow_pickle=False)


# ------------------------------------------------------------------------------------------------ CPU: oracle


def test_g1_oracle_D():
    g = load("g1_wigner_D.npz")
    for q, D in zip(g["rotors"], g["D"]):
        Ra, Rb = quat.as_spinor_array(q)
        assert np.abs(wigner.wigner_D_matrices(Ra, Rb, 0, int(g["ell_max"])) - D).max() < 3e-15


def test_g2_3j_table_matches_oracle_generator():
    g = load("g2_wigner_3j.npz")
    for j1, j2, j3, m1, m2, m3, val in g["table"][::37]:
        assert abs(samples._w3j(int(j1), int(j2), int(j3), int(m1), int(m2), int(m3)) - val) < 1e-15


def test_g3_oracle_swsh_on_boosted_grid():
    g = load("g3_swsh_boosted_grid.npz")
    R = grid_ref.rotor_grid(g["frame_rotation"], g["boost_velocity"], 7, 9)
    assert np.abs(R - g["rotors"]).max() < 1e-15
    for s in range(-2, 3):
        assert np.abs(wigner.swsh_grid(R, s, 6) - g[f"s{s}"]).max() < 3e-15


def test_g4_oracle_map2salm():
    g = load("g4_map2salm.npz")
    for s in range(-2, 3):
        assert np.abs(spinsfast_ref.map2salm(g["maps"], s, int(g["ell_max"])) - g[f"s{s}"]).max() < 1e-14


def test_g5_oracle_translation_vs_analytic():
    g = load("g5_translated_single_mode.npz")
    for i, ((s, ell, m), st) in enumerate(zip(g["meta"], g["translations"])):
        aux = {}
        for k in range(int(s) + 2):
            a = samples.single_mode_proportional_to_time(s=k - 2, ell_max=6)
            a.data *= 0
            aux[f"psi{4-k}_modes"] = a
        w = grid_ref.transform(samples.single_mode_proportional_to_time(s=int(s), ell=int(ell), m=int(m), ell_max=6), space_translation=st, **aux)
        i0 = np.argmin(abs(g["t"] - w.t[0]))
        sel = slice(20, -20)
        assert np.abs(w.data[sel] - g[f"data{i}"][i0 : i0 + w.t.size][sel]).max() < 7.5e-14


def test_g6_oracle_schwarzschild():
    g = load("g6_schwarzschild_boost.npz")
    out = abd_ref.transform(ABD(g["u"], g["raw"], int(g["ell_max"])), boost_velocity=g["boost_velocity"])
    assert np.array_equal(out.u, g["u_out"]) and np.abs(out.raw - g["raw_out"]).max() < 1e-14 * np.abs(g["raw_out"]).max()
    # analytic: P = m gamma (1, -v) from the l <= 1 modes of -Re psi2
    c = -g["raw_out"][2][:, :4]
    P0 = c[:, 0].real / np.sqrt(4 * np.pi)
    Pz = c[:, 2].real / np.sqrt(3) / np.sqrt(4 * np.pi)
    assert np.abs(P0 - g["four_momentum"][0]).max() < 1e-14 and np.abs(Pz - g["four_momentum"][3]).max() < 1e-14


def test_g7_oracle_wm_transform():
    g = load("g7_wm_transform.npz")
    out = grid_ref.transform(WM(t=g["t"], data=g["data"], ell_min=2, ell_max=int(g["ell_max"]), dataType=h),
                             supertranslation=g["supertranslation"], frame_rotation=g["frame_rotation"], boost_velocity=g["boost_velocity"])
    assert np.array_equal(out.t, g["t_out"]) and np.abs(out.data - g["data_out"]).max() < 1e-14


# ------------------------------------------------------------------------------------- CPU: oracle vs the reference's code


def _g8_kw(g):
    return dict(supertranslation=g["supertranslation"], frame_rotation=g["frame_rotation"], boost_velocity=g["boost_velocity"])


def test_g8_oracle_vs_reference_wm_transform():
    """oracle/waveform_grid_ref.py against scri/waveform_grid.py:331-630 run from the reference checkout (h and sigma
    inhomogeneous terms, psi2 mixing with psi3 / psi4 companions, explicit n_theta / n_phi / ell_max)."""
    g = load("g8_ref_wm_transform.npz")
    t, kw = g["t"], _g8_kw(g)
    for name, dt in (("h", h), ("sigma", sigma)):
        o = grid_ref.transform(WM(t=t, data=g[f"{name}_in"], ell_min=2, ell_max=6, dataType=dt), **kw)
        assert np.array_equal(o.t, g[f"{name}_t_out"])
        assert np.abs(o.data - g[f"{name}_out"]).max() < 2e-14 * max(1.0, np.abs(g[f"{name}_out"]).max())
    o = grid_ref.transform(WM(t=t, data=g["psi2_in"], ell_min=0, ell_max=6, dataType=psi2),
                           psi3_modes=WM(t=t, data=g["psi3_in"], ell_min=1, ell_max=5, dataType=psi3),
                           psi4_modes=WM(t=t, data=g["psi4_in"], ell_min=2, ell_max=4, dataType=psi4), **kw)
    assert np.array_equal(o.t, g["psi2_t_out"]) and np.abs(o.data - g["psi2_out"]).max() < 2e-14 * max(1.0, np.abs(g["psi2_out"]).max())
    o = grid_ref.transform(WM(t=t, data=g["h_in"], ell_min=2, ell_max=6, dataType=h), supertranslation=g["supertranslation"],
                           n_theta=23, n_phi=25, ell_max=5)
    assert o.ell_max == 5 and np.array_equal(o.t, g["h_st_t_out"]) and np.abs(o.data - g["h_st_out"]).max() < 2e-14


def _g15_input(g):
    from scri_amd import synthetic

    t = g["t"]
    a = synthetic.chirp_modes(t, 2, 6, int(g["seeds"][0])) * (1 + 0.01 * t[:, None])
    b = synthetic.chirp_modes(t, 2, 6, int(g["seeds"][1]))
    return t, np.stack([a, b], axis=2)


def test_g15_oracle_trailing_dimensions_vs_reference():
    """Extra trailing data dimensions (scri/waveform_grid.py:299-308, 574-594).  The reference's own transform raises on such data
    (recorded in the fixture: its tensordot leaves the extra axis before the grid axes); what its `final_dim` loops spell out --
    every trailing index a series of its own -- is what the oracle restates: data[N, n_modes, 2] transforms to the stack of the two
    separate transforms, each equal to what the reference computes for the series alone, and to_modes of two grids side by side
    equals the reference's."""
    from oracle import spinsfast_ref

    g = load("g15_ref_trailing_dims.npz")
    assert str(g["transform_exception"]).startswith("IndexError")
    t, data = _g15_input(g)
    o = grid_ref.transform(WM(t=t, data=data, ell_min=2, ell_max=6, dataType=h), **_g8_kw(g))
    assert o.data.shape[2:] == (2,) and np.array_equal(o.t, g["t_out"])
    for f, name in enumerate(("modes_a", "modes_b")):
        single = grid_ref.transform(WM(t=t, data=np.ascontiguousarray(data[:, :, f]), ell_min=2, ell_max=6, dataType=h), **_g8_kw(g))
        assert np.array_equal(o.data[:, :, f], single.data)
        assert np.abs(o.data[g["rows_kept_single"], :, f] - g[name]).max() < 2e-14 * max(1.0, np.abs(g[name]).max())
    # the reference's to_modes on [N', n_pix, 2]
    n_theta, n_phi = int(g["n_theta"]), int(g["n_phi"])
    grid = g["grid_two"]
    for f in range(2):
        m = grid_ref.to_modes(g["t_out"][g["rows_kept"]], grid[:, :, f].reshape(-1, n_theta, n_phi), -2, 6)
        assert np.abs(m - g["modes_two"][:, :, f]).max() < 2e-14 * max(1.0, np.abs(g["modes_two"]).max())


def test_g9_oracle_vs_reference_abd_transform():
    """oracle/abd_ref.py against scri/asymptotic_bondi_data/transformations.py:199-431 run from the reference checkout."""
    g = load("g9_ref_abd_transform.npz")
    a = ABD(g["u"], g["raw"], int(g["ell_max"]))
    o = abd_ref.transform(a, **_g8_kw(g))
    assert np.array_equal(o.u, g["u_out"]) and np.abs(o.raw - g["raw_out"]).max() < 2e-14 * np.abs(g["raw_out"]).max()
    o = abd_ref.transform(a, space_translation=g["space_translation_b"], working_ell_max=7, output_ell_max=3)
    assert o.ell_max == 3 and np.array_equal(o.u, g["u_out_b"])
    assert np.abs(o.raw - g["raw_out_b"]).max() < 2e-14 * np.abs(g["raw_out_b"]).max()


def test_g10_oracle_vs_reference_rotations():
    """oracle/rotations_ref.py against scri/rotations.py:268-392 run from the reference checkout (its numba kernels as
    plain Python loops).  The reference leaves the frame of "series after a constant rotor" as an [N, 1] quaternion array
    (rotations.py:316-318: `np.array([W.frame * R for R in R_basis])` with a length-1 frame); the values are compared."""
    from oracle import rotations_ref

    g = load("g10_ref_rotations.npz")
    w = WM(t=g["t"], data=g["data"], ell_min=2, ell_max=4, dataType=h)
    a = rotations_ref.rotate_decomposition_basis(w, g["constant"])
    assert np.abs(a.data - g["const_out"]).max() < 1e-14 and np.abs(a.frame - g["const_frame"]).max() < 1e-15
    b = rotations_ref.rotate_decomposition_basis(a, g["series"])
    assert np.abs(b.data - g["series_after_const_out"]).max() < 1e-14
    assert np.abs(b.frame - g["series_after_const_frame"].reshape(-1, 4)).max() < 1e-15
    c = rotations_ref.rotate_decomposition_basis(w, g["series"])
    assert np.abs(c.data - g["series_out"]).max() < 1e-14 and np.abs(c.frame - g["series_frame"]).max() == 0.0
    d = rotations_ref.rotate_decomposition_basis(w, quat.qconj(g["constant"]))  # rotate_physical_system(R) = basis by ~R
    assert np.abs(d.data - g["physical_out"]).max() < 1e-14 and np.abs(d.frame - g["physical_frame"]).max() < 1e-15


# ------------------------------------------------------------------------------------------------ GPU: engine


@pytest.mark.gpu
def test_g1_gpu_D(ctx):
    from scri_amd import engine

    g = load("g1_wigner_D.npz")
    for q, D in zip(g["rotors"], g["D"]):
        assert np.abs(engine.wigner_D(q, 0, int(g["ell_max"]), ctx=ctx) - D).max() < 5e-15


@pytest.mark.gpu
def test_g3_gpu_swsh(ctx):
    from scri_amd import engine

    g = load("g3_swsh_boosted_grid.npz")
    R = engine.rotor_grid(g["frame_rotation"], g["boost_velocity"], 7, 9, ctx=ctx)
    assert np.abs(R - g["rotors"]).max() < 2e-15
    for s in range(-2, 3):
        assert np.abs(engine.swsh_grid(g["rotors"], s, 0, 6, ctx=ctx) - g[f"s{s}"]).max() < 3e-15


@pytest.mark.gpu
def test_g4_gpu_map2salm(ctx):
    from scri_amd import engine

    g = load("g4_map2salm.npz")
    for s in range(-2, 3):
        assert np.abs(engine.map2salm(g["maps"], s, int(g["ell_max"]), ctx=ctx) - g[f"s{s}"]).max() < 2e-14


@pytest.mark.gpu
def test_g6_g7_gpu_transforms(ctx):
    import scri_amd

    g = load("g6_schwarzschild_boost.npz")
    a = scri_amd.AsymptoticBondiData(g["u"], int(g["ell_max"]), ctx=ctx)
    a._raw_data[:] = g["raw"]
    out = a.transform(boost_velocity=g["boost_velocity"])
    assert np.abs(out.u - g["u_out"]).max() < 1e-14
    assert np.abs(out._raw_data - g["raw_out"]).max() < 1e-14 * np.abs(g["raw_out"]).max()  # |psi0'| reaches 3e2
    g = load("g7_wm_transform.npz")
    w = scri_amd.WaveformModes(t=g["t"], data=g["data"], ell_min=2, ell_max=int(g["ell_max"]), dataType=scri_amd.h,
                               frameType=scri_amd.Inertial, r_is_scaled_out=True, m_is_scaled_out=True, ctx=ctx)
    o = w.transform(supertranslation=g["supertranslation"], frame_rotation=g["frame_rotation"], boost_velocity=g["boost_velocity"])
    assert np.abs(o.t - g["t_out"]).max() < 1e-14 and np.abs(o.data - g["data_out"]).max() < 1e-13


def _gpu_wm(t, data, ell_min, ell_max, dataType, ctx):
    import scri_amd

    return scri_amd.WaveformModes(t=t, data=data, ell_min=ell_min, ell_max=ell_max, dataType=dataType, frameType=scri_amd.Inertial,
                                  r_is_scaled_out=True, m_is_scaled_out=True, ctx=ctx)


@pytest.mark.gpu
def test_g8_gpu_vs_reference_wm_transform(ctx):
    import scri_amd

    g = load("g8_ref_wm_transform.npz")
    t, kw = g["t"], _g8_kw(g)
    for name, dt in (("h", scri_amd.h), ("sigma", scri_amd.sigma)):
        o = _gpu_wm(t, g[f"{name}_in"], 2, 6, dt, ctx).transform(**kw)
        assert o.t.shape == g[f"{name}_t_out"].shape and np.abs(o.t - g[f"{name}_t_out"]).max() < 1e-13
        assert np.abs(o.data - g[f"{name}_out"]).max() < 1e-13 * max(1.0, np.abs(g[f"{name}_out"]).max())
    o = _gpu_wm(t, g["psi2_in"], 0, 6, scri_amd.psi2, ctx).transform(
        psi3_modes=_gpu_wm(t, g["psi3_in"], 1, 5, scri_amd.psi3, ctx), psi4_modes=_gpu_wm(t, g["psi4_in"], 2, 4, scri_amd.psi4, ctx), **kw)
    assert o.t.shape == g["psi2_t_out"].shape and np.abs(o.data - g["psi2_out"]).max() < 1e-13 * max(1.0, np.abs(g["psi2_out"]).max())
    o = _gpu_wm(t, g["h_in"], 2, 6, scri_amd.h, ctx).transform(supertranslation=g["supertranslation"], n_theta=23, n_phi=25, ell_max=5)
    assert o.ell_max == 5 and o.t.shape == g["h_st_t_out"].shape and np.abs(o.data - g["h_st_out"]).max() < 1e-13


@pytest.mark.gpu
def test_g15_gpu_trailing_dimensions(ctx):
    """data[N, n_modes, F] through WaveformModes.transform / WaveformGrid.from_modes / to_modes: the stack of the F separate
    transforms BIT FOR BIT, each equal to what the reference computes for that series alone; to_modes of two grids side by side
    against the reference's own to_modes."""
    import scri_amd

    g = load("g15_ref_trailing_dims.npz")
    t, data = _g15_input(g)
    kw = _g8_kw(g)
    o = _gpu_wm(t, data, 2, 6, scri_amd.h, ctx).transform(**kw)
    assert o.data.shape == (g["t_out"].shape[0], 45, 2) and np.abs(o.t - g["t_out"]).max() < 1e-13
    gr = scri_amd.WaveformGrid.from_modes(_gpu_wm(t, data, 2, 6, scri_amd.h, ctx), **kw)
    assert gr.data.shape == (o.n_times, int(g["n_theta"]) * int(g["n_phi"]), 2)
    back = gr.to_modes(6)
    for f, name in enumerate(("modes_a", "modes_b")):
        w1 = _gpu_wm(t, np.ascontiguousarray(data[:, :, f]), 2, 6, scri_amd.h, ctx)
        single = w1.transform(**kw)
        assert np.array_equal(o.data[:, :, f], single.data)
        assert np.array_equal(gr.data[:, :, f], scri_amd.WaveformGrid.from_modes(w1, **kw).data)
        assert np.abs(o.data[g["rows_kept_single"], :, f] - g[name]).max() < 1e-13 * max(1.0, np.abs(g[name]).max())
        assert np.abs(back.data[:, :, f] - single.data).max() < 1e-13 * max(1.0, np.abs(single.data).max())
    two = scri_amd.WaveformGrid(g["t_out"][g["rows_kept"]], g["grid_two"], int(g["n_theta"]), int(g["n_phi"]), scri_amd.Inertial, scri_amd.h, True, True,
                                ctx=ctx).to_modes(6)
    assert two.data.shape == g["modes_two"].shape and np.abs(two.data - g["modes_two"]).max() < 1e-13 * max(1.0, np.abs(g["modes_two"]).max())
    # a (2, 3) block of trailing dimensions keeps its shape
    d6 = np.stack([data[:, :, 0] * (1 + 0.1 * k) for k in range(6)], axis=2).reshape(data.shape[:2] + (2, 3))
    o6 = _gpu_wm(t, d6, 2, 6, scri_amd.psi4, ctx).transform(**kw)
    assert o6.data.shape == (o.n_times, 45, 2, 3)
    ref = _gpu_wm(t, np.ascontiguousarray(d6[:, :, 1, 2]), 2, 6, scri_amd.psi4, ctx).transform(**kw)
    assert np.array_equal(o6.data[:, :, 1, 2], ref.data)


@pytest.mark.gpu
def test_g9_gpu_vs_reference_abd_transform(ctx):
    import scri_amd

    g = load("g9_ref_abd_transform.npz")
    a = scri_amd.AsymptoticBondiData(g["u"], int(g["ell_max"]), ctx=ctx)
    a._raw_data[:] = g["raw"]
    o = a.transform(**_g8_kw(g))
    assert o.n_times == g["u_out"].shape[0] and np.abs(o.u - g["u_out"]).max() < 1e-13
    assert np.abs(o._raw_data - g["raw_out"]).max() < 1e-13 * np.abs(g["raw_out"]).max()
    o = a.transform(space_translation=g["space_translation_b"], working_ell_max=7, output_ell_max=3)
    assert o.ell_max == 3 and o.n_times == g["u_out_b"].shape[0]
    assert np.abs(o._raw_data - g["raw_out_b"]).max() < 1e-13 * np.abs(g["raw_out_b"]).max()


@pytest.mark.gpu
def test_g10_gpu_vs_reference_rotations(ctx):
    import scri_amd

    g = load("g10_ref_rotations.npz")
    w = _gpu_wm(g["t"], g["data"].copy(), 2, 4, scri_amd.h, ctx)
    w.rotate_decomposition_basis(g["constant"])
    assert np.abs(w.data - g["const_out"]).max() < 4e-13 and np.abs(w.frame - g["const_frame"]).max() < 1e-15
    w.rotate_decomposition_basis(g["series"])
    assert np.abs(w.data - g["series_after_const_out"]).max() < 4e-13
    assert np.abs(w.frame - g["series_after_const_frame"].reshape(-1, 4)).max() < 1e-15
    w = _gpu_wm(g["t"], g["data"].copy(), 2, 4, scri_amd.h, ctx)
    w.rotate_decomposition_basis(g["series"])
    assert np.abs(w.data - g["series_out"]).max() < 4e-13 and np.abs(w.frame - g["series_frame"]).max() == 0.0
    w = _gpu_wm(g["t"], g["data"].copy(), 2, 4, scri_amd.h, ctx)
    w.rotate_physical_system(g["constant"])
    assert np.abs(w.data - g["physical_out"]).max() < 4e-13 and np.abs(w.frame - g["physical_frame"]).max() < 1e-15


# ---- g11: the mode-space operators of scri.WaveformModes, computed by the reference's own scri/waveform_modes.py:458-943 and
# scri/extrapolation.py:47-122 (tests/golden/make_golden_from_reference.py::g11)
G11_TYPES = {"psi1": 2, "psi4": 5, "h": 7, "psi2": 3}  # oracle.containers data type numbers
G11_ETH = (("+", "NP"), ("-", "NP"), ("-+", "NP"), ("+-", "GHP"), ([+1, -1, -1], "NP"))


def _g11_cases(g):
    for name, dt in G11_TYPES.items():
        lmin, lmax = (int(x) for x in g[f"{name}_ells"])
        yield name, dt, lmin, lmax


def test_g11_oracle_vs_reference_mode_operators():
    from oracle import waveform_modes_ref as mref

    g = load("g11_ref_mode_operators.npz")
    t = g["t"]
    for name, dt, lmin, lmax in _g11_cases(g):
        w = WM(t=t, data=g[f"{name}_in"], ell_min=lmin, ell_max=lmax, dataType=dt)
        for d in ("x_", "y_", "z_", ""):
            for part, fn in (("conjugate", mref.parity_conjugate), ("symmetric_part", mref.parity_symmetric_part),
                             ("antisymmetric_part", mref.parity_antisymmetric_part)):
                assert np.array_equal(fn(w, d.rstrip("_")).data, g[f"{name}_{d}parity_{part}"]), (name, d, part)
            # (norms: the vectors were made without numba, whose x ** 2 is x * x; CPython's pow differs by an ulp now and then)
            assert np.allclose(mref.parity_violation_squared(w, d.rstrip("_")), g[f"{name}_{d}parity_violation_squared"], rtol=1e-15, atol=0)
        for k, (ops, conv) in enumerate(G11_ETH):
            assert np.array_equal(mref.apply_eth(w, ops, eth_convention=conv), g[f"{name}_eth{k}"]), (name, ops)
        p = mref.convert_to_conjugate_pairs(w)
        assert np.array_equal(p.data, g[f"{name}_pairs"])
        assert np.array_equal(mref.convert_from_conjugate_pairs(p).data, g[f"{name}_pairs_back"])
        for tol in (1e-10, 1e-3):
            assert np.array_equal(mref.truncate(w, tol).data, g[f"{name}_truncate_{tol:g}"])
        other = WM(t=t, data=g[f"{name}_other"], ell_min=lmin, ell_max=lmax, dataType=dt)
        got = np.array([mref.inner_product(w, other), mref.inner_product(w, other, t1=2.0, t2=7.0)])
        assert np.abs(got - g[f"{name}_inner"]).max() < 1e-13 * np.abs(g[f"{name}_inner"]).max()
    assert np.array_equal(mref.intersection(t, g["t2"]), g["intersection"])
    assert np.array_equal(mref.intersection(t, g["t2"], min_step=0.25), g["intersection_min_step"])
    assert np.array_equal(mref.intersection(t, g["t2"], min_time=1.5, max_time=8.0), g["intersection_bounds"])


def test_g11_time_intersection_vs_reference():
    from scri_amd.mode_operators import time_intersection

    g = load("g11_ref_mode_operators.npz")
    assert np.array_equal(time_intersection(g["t"], g["t2"]), g["intersection"])
    assert np.array_equal(time_intersection(g["t"], g["t2"], min_step=0.25), g["intersection_min_step"])
    assert np.array_equal(time_intersection(g["t"], g["t2"], min_time=1.5, max_time=8.0), g["intersection_bounds"])


@pytest.mark.gpu
@pytest.mark.parametrize("device", [False, True])
def test_g11_gpu_vs_reference_mode_operators(ctx, device):
    import scri_amd

    g = load("g11_ref_mode_operators.npz")
    t = g["t"]

    def make(name, dt, lmin, lmax, key="in"):
        w = scri_amd.WaveformModes(t=t, data=g[f"{name}_{key}"].copy(), ell_min=lmin, ell_max=lmax, dataType=dt, frameType=scri_amd.Inertial,
                                   r_is_scaled_out=True, m_is_scaled_out=True, ctx=ctx)
        return w.to_device() if device else w

    for name, dt, lmin, lmax in _g11_cases(g):
        w = make(name, dt, lmin, lmax)
        for d in ("x_", "y_", "z_", ""):
            for part in ("conjugate", "symmetric_part", "antisymmetric_part"):
                assert np.array_equal(getattr(w, f"{d}parity_{part}").data, g[f"{name}_{d}parity_{part}"]), (name, d, part)
            assert np.allclose(getattr(w, f"{d}parity_violation_squared"), g[f"{name}_{d}parity_violation_squared"], rtol=1e-15, atol=0)
        for k, (ops, conv) in enumerate(G11_ETH):
            assert np.array_equal(w.apply_eth(ops, eth_convention=conv), g[f"{name}_eth{k}"]), (name, ops)
        p = make(name, dt, lmin, lmax)
        p.convert_to_conjugate_pairs()
        scale = np.abs(g[f"{name}_in"]).max()
        assert np.abs(p.data - g[f"{name}_pairs"]).max() < 4e-16 * scale  # (x + y) / sqrt2 here is x / sqrt2 + y / sqrt2
        p.convert_from_conjugate_pairs()
        assert np.abs(p.data - g[f"{name}_pairs_back"]).max() < 8e-16 * scale
        for tol in (1e-10, 1e-3):
            q = make(name, dt, lmin, lmax)
            q.truncate(tol)
            assert np.array_equal(q.data, g[f"{name}_truncate_{tol:g}"])
        other = make(name, dt, lmin, lmax, "other")
        got = np.array([w.inner_product(other), w.inner_product(other, t1=2.0, t2=7.0)])
        assert np.abs(got - g[f"{name}_inner"]).max() < 1e-12 * np.abs(g[f"{name}_inner"]).max()


# ---- g12: the Bondi charges computed by the reference's own scri/asymptotic_bondi_data/bms_charges.py:14-192
def test_g12_oracle_vs_reference_charges():
    from oracle import bms_charges_ref as cref

    g = load("g12_ref_bms_charges.npz")
    u, raw, L = g["u"], g["raw"], int(g["ell_max"])
    psi1, psi2, sigma = raw[1], raw[2], raw[5]
    tol = 2e-13
    assert np.abs(cref.mass_aspect(u, psi2, sigma, L) - g["mass_aspect"]).max() < tol * np.abs(g["mass_aspect"]).max()
    assert np.abs(cref.mass_aspect(u, psi2, sigma, 2) - g["mass_aspect_ell2"]).max() < tol * np.abs(g["mass_aspect"]).max()
    P = cref.four_momentum(u, psi2, sigma)
    assert np.abs(P - g["bondi_four_momentum"]).max() < tol * np.abs(P).max()
    assert np.abs(np.sqrt(P[:, 0] ** 2 - (P[:, 1:] ** 2).sum(axis=1)) - g["bondi_rest_mass"]).max() < tol * np.abs(P).max()
    assert np.abs(cref.angular_momentum(psi1, sigma) - g["bondi_angular_momentum"]).max() < tol
    assert np.abs(cref.com_charge(psi1, sigma) - g["bondi_CoM_charge"]).max() < tol
    assert np.abs(cref.boost_charge(u, psi1, psi2, sigma) - g["bondi_boost_charge"]).max() < 20 * tol * np.abs(g["bondi_boost_charge"]).max()
    assert np.abs(cref.dimensionless_spin(u, psi1, psi2, sigma) - g["bondi_dimensionless_spin"]).max() < 1e-12


@pytest.mark.gpu
@pytest.mark.parametrize("device", [False, True])
def test_g12_gpu_vs_reference_charges(ctx, device):
    import scri_amd

    g = load("g12_ref_bms_charges.npz")
    u, raw, L = g["u"], g["raw"], int(g["ell_max"])
    a = scri_amd.AsymptoticBondiData(u, L, ctx=ctx)
    a._raw_data[:] = raw
    if device:
        a = a.to_device()
    tol = 5e-12
    assert np.abs(np.asarray(a.mass_aspect().ndarray) - g["mass_aspect"]).max() < tol
    assert np.abs(np.asarray(a.mass_aspect(2).ndarray) - g["mass_aspect_ell2"]).max() < tol
    assert np.abs(a.bondi_four_momentum() - g["bondi_four_momentum"]).max() < tol
    assert np.abs(a.bondi_rest_mass() - g["bondi_rest_mass"]).max() < tol
    assert np.abs(a.bondi_angular_momentum() - g["bondi_angular_momentum"]).max() < tol
    assert np.abs(a.bondi_CoM_charge() - g["bondi_CoM_charge"]).max() < tol
    assert np.abs(a.bondi_boost_charge() - g["bondi_boost_charge"]).max() < 20 * tol
    assert np.abs(a.bondi_dimensionless_spin() - g["bondi_dimensionless_spin"]).max() < 1e-9


# ---- g13: LdtVector / LLMatrix / angular_velocity / LLDominantEigenvector by the reference's own scri/mode_calculations.py
def test_g13_oracle_vs_reference_mode_calculations():
    from oracle import mode_calculations_ref as mc

    g = load("g13_ref_mode_calculations.npz")
    for tag in ("a", "b"):
        t, data = g[f"{tag}_t"], g[f"{tag}_data"]
        lmin, lmax = (int(x) for x in g[f"{tag}_ells"])
        dd = mc.data_dot(t, data)
        ldt, ll = mc.LdtVector(data, dd, lmin, lmax), mc.LLMatrix(data, lmin, lmax)
        assert np.abs(ldt - g[f"{tag}_LdtVector"]).max() < 1e-13 * np.abs(ldt).max()
        assert np.abs(ll - g[f"{tag}_LLMatrix"]).max() < 1e-14 * np.abs(ll).max()
        om = mc.angular_velocity(t, data, lmin, lmax)
        assert np.abs(om - g[f"{tag}_angular_velocity"]).max() < 1e-11 * max(1.0, np.abs(om).max())


@pytest.mark.gpu
def test_g13_gpu_vs_reference_mode_calculations(ctx):
    import scri_amd

    g = load("g13_ref_mode_calculations.npz")
    for tag, dt in (("a", scri_amd.h), ("b", scri_amd.psi2)):
        t, data = g[f"{tag}_t"], g[f"{tag}_data"]
        lmin, lmax = (int(x) for x in g[f"{tag}_ells"])
        w = scri_amd.WaveformModes(t=t, data=data.copy(), ell_min=lmin, ell_max=lmax, dataType=dt, frameType=scri_amd.Inertial,
                                   r_is_scaled_out=True, m_is_scaled_out=True, ctx=ctx)
        assert np.abs(w.LdtVector() - g[f"{tag}_LdtVector"]).max() < 1e-11 * np.abs(g[f"{tag}_LdtVector"]).max()
        assert np.abs(w.LLMatrix() - g[f"{tag}_LLMatrix"]).max() < 1e-13 * np.abs(g[f"{tag}_LLMatrix"]).max()
        assert np.abs(w.angular_velocity() - g[f"{tag}_angular_velocity"]).max() < 1e-9 * max(1.0, np.abs(g[f"{tag}_angular_velocity"]).max())
        got, expect = w.LLDominantEigenvector(), g[f"{tag}_LLDominantEigenvector"]
        assert np.abs(got - expect).max() < 1e-9
        # the two-waveform forms (scri/mode_calculations.py:55-260), index slip of the reference's (y, y) / (y, z) elements included
        from scri_amd import mode_calculations as mcalc

        other = scri_amd.WaveformModes(t=t, data=g[f"{tag}_other"].copy(), ell_min=lmin, ell_max=lmax, dataType=dt, frameType=scri_amd.Inertial, ctx=ctx)
        lv, llc = g[f"{tag}_LVector"], g[f"{tag}_LLComparisonMatrix"]
        assert np.abs(mcalc.LVector(w, other) - lv).max() < 1e-13 * np.abs(lv).max()
        assert np.abs(mcalc.LLComparisonMatrix(w, other) - llc).max() < 1e-13 * np.abs(llc).max()
        assert np.all(llc[:, 1, 2] == 0)


# ---- g14: the fluxes of scri/flux.py:182-798 computed by the reference's own matrix elements and loops
def test_g14_oracle_vs_reference_fluxes():
    from oracle import flux_ref

    g = load("g14_ref_fluxes.npz")
    t, data = g["t"], g["data"]
    lmin, lmax = (int(x) for x in g["ells"])
    hdot = flux_ref.data_dot(t, data)
    assert np.abs(hdot - g["hdot"]).max() < 1e-13 * np.abs(g["hdot"]).max()
    rel = lambda a, b: np.abs(a - b).max() / np.abs(b).max()  # noqa: E731
    assert rel(flux_ref.energy_flux(hdot), g["energy_flux"]) < 1e-13
    assert rel(flux_ref.silly_momentum_flux(hdot, lmin, lmax), g["momentum_flux"]) < 1e-12
    assert rel(flux_ref.silly_angular_momentum_flux(data, hdot, lmin, lmax), g["angular_momentum_flux"]) < 1e-12
    assert rel(flux_ref.boost_flux(t, data, hdot, lmin, lmax), g["boost_flux"]) < 1e-12


@pytest.mark.gpu
def test_g14_gpu_vs_reference_fluxes(ctx):
    import scri_amd
    from scri_amd import flux

    g = load("g14_ref_fluxes.npz")
    lmin, lmax = (int(x) for x in g["ells"])
    h = scri_amd.WaveformModes(t=g["t"], data=g["data"].copy(), ell_min=lmin, ell_max=lmax, dataType=scri_amd.h, frameType=scri_amd.Inertial,
                               r_is_scaled_out=True, m_is_scaled_out=True, ctx=ctx)
    rel = lambda a, b: np.abs(a - b).max() / np.abs(b).max()  # noqa: E731
    for s in (1, 2, 3):  # single expectation values, spin -1, -2, -3
        a, b = flux._Field(g["hdot"], -s, lmin, lmax, ctx), flux._Field(g["data"], -s, lmin, lmax, ctx)
        for name, got in zip(("p_plus", "p_minus", "p_z"), flux._chi(a, b)):
            assert rel(got, g[f"{name}_s{s}"]) < 1e-13, (s, name)
    assert rel(h.energy_flux(), g["energy_flux"]) < 1e-13
    assert rel(h.momentum_flux(), g["momentum_flux"]) < 1e-13
    assert rel(h.angular_momentum_flux(), g["angular_momentum_flux"]) < 1e-13
    assert rel(h.boost_flux(), g["boost_flux"]) < 1e-12
    e, p, j, b = scri_amd.poincare_fluxes(h)
    assert rel(e, g["poincare_e"]) < 1e-13 and rel(p, g["poincare_p"]) < 1e-13 and rel(j, g["poincare_j"]) < 1e-13 and rel(b, g["poincare_b"]) < 1e-12
    hdot = h.copy()
    hdot.dataType, hdot.data = scri_amd.hdot, g["hdot"].copy()
    assert rel(scri_amd.momentum_flux(hdot), g["momentum_flux"]) < 1e-13 and rel(scri_amd.energy_flux(hdot), g["energy_flux"]) < 1e-13
    assert rel(scri_amd.boost_flux(h, hdot), g["boost_flux"]) < 1e-12
    with pytest.raises(ValueError, match="expected to have data of type `h`"):
        scri_amd.angular_momentum_flux(hdot)
    with pytest.raises(ValueError, match="can only be calculated from a `WaveformModes` object"):
        scri_amd.energy_flux(g["data"])


# ------------------------------------------------------------------------------------------------ g28: the relativistic regime


def _g28_inputs(g):
    """the fixture holds outputs (every 4th row) and parameters; the inputs are regenerated from the generator's seeds"""
    from scri_amd import synthetic

    t, L = g["wm_t"], int(g["wm_ell_max"])
    kw = dict(supertranslation=g["wm_supertranslation"], frame_rotation=g["wm_frame_rotation"], boost_velocity=g["wm_boost_velocity"])
    wm = {name: synthetic.chirp_modes(t, 2, L, seed) * (1 + 0.004 * t[:, None]) for name, seed in (("h", 282), ("news", 283), ("psi4", 284))}
    d1 = synthetic.chirp_modes(t, 1, 8, 285)
    comp = {k: (s_, 8 - k + 2, synthetic.chirp_modes(t, s_, 8 - k + 2, 285 + k)) for k, s_ in ((2, 0), (3, 1), (4, 2))}
    assert [[v[0], v[1]] for v in comp.values()] == g["psi1_companion_ell"].tolist()
    u, L2 = g["abd_u"], int(g["abd_ell_max"])
    raw = np.zeros((6, u.size, (L2 + 1) ** 2), dtype=complex)
    for f, s_ in enumerate(synthetic.ABD_SPINS):
        raw[f] = synthetic.chirp_modes(u, 0, L2, 290 + f) * (1 + 0.01 * u[:, None])
        raw[f, :, : s_ * s_] = 0
    kw2 = dict(supertranslation=g["abd_supertranslation"], frame_rotation=g["abd_frame_rotation"], boost_velocity=g["abd_boost_velocity"])
    return t, L, kw, wm, d1, comp, u, L2, raw, kw2


def test_g28_oracle_vs_reference_relativistic_transforms():
    """g8 / g9 where nothing is small (|v| = 0.35 / 0.30, supertranslations of order one, l up to 10, non-uniform time axes): the
    oracle against scri/waveform_grid.py:331-630 and scri/asymptotic_bondi_data/transformations.py:199-431 run by the reference's files"""
    from oracle import containers as oc

    g = load("g28_ref_relativistic_transforms.npz")
    t, L, kw, wm, d1, comp, u, L2, raw, kw2 = _g28_inputs(g)
    for name, data in wm.items():
        o = grid_ref.transform(WM(t=t, data=data, ell_min=2, ell_max=L, dataType=getattr(oc, name)), **kw)
        ref = g[f"{name}_out"]
        assert np.array_equal(o.t, g[f"{name}_t_out"]) and np.abs(o.data[::4] - ref).max() < 1e-12 * np.abs(ref).max(), name
    o = grid_ref.transform(WM(t=t, data=d1, ell_min=1, ell_max=8, dataType=oc.psi1),
                           **{f"psi{k}_modes": WM(t=t, data=v[2], ell_min=v[0], ell_max=v[1], dataType=getattr(oc, f"psi{k}")) for k, v in comp.items()}, **kw)
    assert np.array_equal(o.t, g["psi1_t_out"]) and np.abs(o.data[::4] - g["psi1_out"]).max() < 1e-12 * np.abs(g["psi1_out"]).max()
    o = abd_ref.transform(ABD(u, raw, L2), **kw2)
    assert np.array_equal(o.u, g["abd_u_out"])
    for f in range(6):
        assert np.abs(o.raw[f] - g["abd_raw_out"][f]).max() < 1e-12 * np.abs(g["abd_raw_out"][f]).max(), f


@pytest.mark.gpu
def test_g28_gpu_vs_reference_relativistic_transforms(ctx):
    import scri_amd

    g = load("g28_ref_relativistic_transforms.npz")
    t, L, kw, wm, d1, comp, u, L2, raw, kw2 = _g28_inputs(g)
    for name, data in wm.items():
        o = _gpu_wm(t, data, 2, L, getattr(scri_amd, name), ctx).transform(**kw)
        ref = g[f"{name}_out"]
        assert o.t.shape == g[f"{name}_t_out"].shape and np.abs(o.t - g[f"{name}_t_out"]).max() < 1e-12, name
        assert np.abs(o.data[::4] - ref).max() < 1e-12 * np.abs(ref).max(), name
    o = _gpu_wm(t, d1, 1, 8, scri_amd.psi1, ctx).transform(
        **{f"psi{k}_modes": _gpu_wm(t, v[2], v[0], v[1], getattr(scri_amd, f"psi{k}"), ctx) for k, v in comp.items()}, **kw)
    assert o.t.shape == g["psi1_t_out"].shape and np.abs(o.data[::4] - g["psi1_out"]).max() < 1e-12 * np.abs(g["psi1_out"]).max()
    a = scri_amd.AsymptoticBondiData(u, L2, ctx=ctx)
    a._raw_data[:] = raw
    o = a.transform(**kw2)
    assert o.n_times == g["abd_u_out"].shape[0] and np.abs(o.u - g["abd_u_out"]).max() < 1e-12
    for f in range(6):
        assert np.abs(o._raw_data[f] - g["abd_raw_out"][f]).max() < 1e-12 * np.abs(g["abd_raw_out"][f]).max(), f
