"""Golden vectors computed by the reference's OWN source files (run in the build container only: needs /root/reference).

    python tests/golden/make_golden_from_reference.py

imports /root/reference/scri/*.py unmodified on top of the stand-ins of reference_standins.py (the image has no numba,
quaternion, spherical_functions, spinsfast) and writes

  g8_ref_wm_transform.npz   scri.WaveformModes.transform (scri/waveform_grid.py:331-630): h, sigma, psi2 (with psi3/psi4
                            companions), l = 2..6 / 0..6, N = 240, supertranslation + frame rotation + boost
  g9_ref_abd_transform.npz  scri.AsymptoticBondiData.transform (scri/asymptotic_bondi_data/transformations.py:199-431):
                            six fields, l <= 4, N = 120
  g10_ref_rotations.npz     scri.WaveformModes.rotate_decomposition_basis (scri/rotations.py:284-392): constant rotor and
                            rotor series, data and frame

  g11_ref_mode_operators.npz  the mode-space operators of scri.WaveformModes (scri/waveform_modes.py:458-943): apply_eth, the
                            four parity conjugates with their symmetric / antisymmetric parts, the conjugate-pair form, truncate,
                            inner_product; scri.extrapolation.intersection (scri/extrapolation.py:47-122).  The waveforms carry an
                            EMPTY frame, so the numpy-quaternion parity ufuncs the reference applies to it (np.x_parity_conjugate
                            ...: absent here) are the identity on it; inner_product's spline_definite_integral is scipy's
                            CubicSpline.integrate -- both third-party, everything else is the reference's code.

  g12_ref_bms_charges.npz   the Bondi charges of scri/asymptotic_bondi_data/bms_charges.py:14-192 on generic smooth data (l <= 4, N = 90):
                            mass aspect, four-momentum, rest mass, angular momentum, boost and centre-of-mass charge,
                            dimensionless spin.  The formulas are the reference's; the sf.Modes algebra underneath (multiply,
                            eth, bar, real, truncate_ell) and scipy's spline are the stand-ins' / third party.

  g13_ref_mode_calculations.npz  scri/mode_calculations.py:14-141,320-372 on a chirp (l = 2..6 and 0..4, N = 120): LdtVector,
                            LLMatrix, angular_velocity, LLDominantEigenvector -- the reference's loops; the ladder coefficient
                            sqrt(l(l+1) - m(m+1)) and scipy's spline (data_dot) are third party.

  g14_ref_fluxes.npz        scri/flux.py:182-798 on a chirp with all modes l = 2..5 (N = 64): energy, momentum, angular-momentum and boost
                            flux, poincare_fluxes, and single expectation values <a|p_z|b>, <a|p_+|b>, <a|p_-|b> (s = -1, -2, -3) --
                            the reference's matrix elements and loops; Clebsch-Gordan coefficients from sympy (third party).

  g15_ref_trailing_dims.npz  extra trailing data dimensions (scri/waveform_grid.py:299-308, 574-594): WaveformGrid.to_modes of two series
                            side by side, and the exception the reference's own transform raises on such data.

  g16_ref_bms_algebra.npz   scri/bms_transformations.py:183-592 at its default ell_max = 12, |v| = 0.17, a generic real supertranslation
                            (l <= 4): BMSTransformation.reorder for all 36 (input order, output order) pairs, .inverse for the six
                            orders, two compositions; LorentzTransformation reorder (both ways), inverse, product.  The case table,
                            the SL(2,C) bookkeeping and transform_supertranslation's statements are the reference's; the grid /
                            quadrature arithmetic underneath it is the stand-ins'.

  g17_ref_bit_transforms.npz  scri/utilities.py:194-406 and scri/SpEC/file_io/__init__.py:50-70 under the identity `njit`: xor_timeseries(_reverse),
                            multishuffle forward and back (8/16/32/64 bit, six width tuples each), fletcher32 (odd and even
                            lengths beyond one 360-word block), index_is_monotonic.  Integer / bit work: every byte is the
                            reference's own (no arithmetic stand-in is involved).

  g18_ref_modes_time_series.npz  scri/modes_time_series.py:72-202 on a NON-uniform time axis (N = 70; a spin -1, l <= 5 and a spin 2, l <= 4 series):
                            interpolate at every derivative order -2 .. 3, dot / ddot / int / iint, eth_GHP / ethbar_GHP, grid_multiply with
                            its defaults and with explicit working / output l_max -- the reference's bookkeeping (which l_max the grid and
                            the result get, the product's spin weight, where the antiderivative starts); scipy's CubicSpline underneath is
                            live third party, salm2map / map2salm and the ladder factors are the stand-ins'.

  g19_ref_superrest_helpers.npz  the building blocks of scri/asymptotic_bondi_data/map_to_superrest_frame.py:76-507,666-684 on generic smooth data
                            (l <= 4, N = 60): the operator pair, rest mass and conformal factor on the grid, the Moreschi supermomentum in a
                            supertranslated frame, the first-order supertranslation, three iterations of
                            supertranslation_to_map_to_superrest_frame (which transforms with the reference's own abd.transform), the
                            centre-of-mass fit, rotation_from_spin_charge with and without a fixed plane, time_translation.  The
                            statements are the reference's; grids, spline and quaternion arithmetic underneath are stand-ins / scipy.

  g20_ref_map_to_superrest_frame.npz  the frame-fixing loop itself, scri/asymptotic_bondi_data/map_to_superrest_frame.py:369-465, 529-663, 719-1035 on
                            the g19 data: com_transformation_to_map_to_superrest_frame and rotation_to_map_to_superrest_frame (two
                            iterations each), rel_err_for_abd_in_superrest, and map_to_superrest_frame at t_0 = 2 with a padding of 25 and
                            two passes of (supertranslation, rotation, CoM) -- every transformation inside by the reference's own
                            abd.transform, every composition / reordering by its own BMSTransformation; quaternion.calculus'
                            indefinite_integral is restated (scipy's degree-3 InterpolatedUnivariateSpline antiderivative).

  g21_ref_map_to_abd_frame.npz  scri/asymptotic_bondi_data/map_to_abd_frame.py:21-290 on the g19 data against a BMS-transformed copy of itself:
                            rel_err_between_abds, and map_to_abd_frame without the time / phase alignment (that step is sxs' align2d,
                            absent here): the target's and the object's own superrest loops, the composition
                            (transformation2^-1 * transformation1 * BMS) and the final transform, two passes.

  g22_ref_error_behaviour.json  what the reference raises or warns, word for word: scri/waveform_grid.py:20-190 and
                            scri/asymptotic_bondi_data/transformations.py:8-97 on 27 keyword sets (wrong shapes and types, precedence of the
                            translations, grids too small or small, rotors and velocities out of range, unknown keywords), and the
                            transform's own checks (argument type, missing / mismatched psi companions).

  g23_ref_sample_waveforms.npz  the deterministic generators of scri/sample_waveforms.py:195-381, among them the analytic answer of the reference's
                            hyper-translation test (a single mode proportional to time under a supertranslation, :312-381; its 3-j symbols
                            from sympy): three (s, l, m) each, a generic supertranslation and a plain space translation.

  g24_ref_containers.npz    the container operations around the path: WaveformModes.interpolate, data_dot / _ddot /
                            _int / _iint, norm (scri/waveform_base.py:535-551, 685-705, 950-967), slicing along time and along l
                            (scri/waveform_modes.py:976-1004), AsymptoticBondiData.interpolate, its time slice and its strain h
                            (scri/asymptotic_bondi_data/__init__.py:119-131, 218-233).

  g25_ref_supermomenta.npz  AsymptoticBondiData.supermomentum (scri/asymptotic_bondi_data/bms_charges.py:192-286) in its four definitions (Bondi-Sachs,
                            Moreschi, Geroch, Geroch-Winicour), plain and integrated, with the default working l_max and a larger one; the
                            remaining charges of the g19 data (CWWY angular momentum among them) and the text of the error for an unknown name.

  g26_ref_initial_values.npz  AsymptoticBondiData.from_initial_values (scri/asymptotic_bondi_data/from_initial_values.py:1-220) in both of its branches
                            -- sigma as a quadratic in u integrated exactly, and the same sigma given on the time axis and integrated through
                            splines -- with generic initial psi2, psi1, psi0 (l <= 4, N = 41), and the two sides of the six Bondi-gauge
                            relations and their violation norms on the results (constraints.py:9-110).  The products underneath are the
                            stand-ins' Wigner-3j sums (sympy), the splines scipy's.

  g27_ref_grids_and_containers.npz  boosted_grid and conformal_factors (scri/asymptotic_bondi_data/transformations.py:100-196) for a generic frame + boost, a
                            boost alone and a rotation alone on a 9 x 11 grid; WaveformModes.SI_units for three data types and
                            WaveformModes.compare on a resampled copy, with and without its two thresholds (scri/waveform_base.py:553-687).

  g28_ref_relativistic_transforms.npz  g8 and g9 again where nothing is small: |v| = 0.35 / 0.30, a supertranslation of order one (l <= 4 / 3), l = 2..10 (h, news,
                            psi4; psi1 with its three companions) and six fields at l <= 6, on NON-uniform time axes (N = 400 / 150).

Only the .npz and .json files travel; tests/test_golden.py checks the oracle (CPU) and the HIP path (GPU) against them.
"""
import os
import sys
import warnings

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import reference_standins as standins  # noqa: E402

scri = standins.install()

from scri_amd import synthetic  # noqa: E402  (seeded smooth inputs; plain numpy)


def _wm(t, data, ell_min, ell_max, dataType):
    return scri.WaveformModes(t=t, data=data, ell_min=ell_min, ell_max=ell_max, frameType=scri.Inertial, dataType=dataType,
                              r_is_scaled_out=True, m_is_scaled_out=True)


def _real_supertranslation(ell_max, seed, scale):
    rng = np.random.default_rng(seed)
    a = scale * (rng.normal(size=(ell_max + 1) ** 2) + 1j * rng.normal(size=(ell_max + 1) ** 2))
    return synthetic.real_supertranslation(a)


def g8():
    n, L = 240, 6
    t = np.linspace(-12.0, 36.0, n)
    kw = dict(supertranslation=_real_supertranslation(3, 81, 0.05), frame_rotation=np.array([0.4, 1.0, -2.0, 0.3]) / np.linalg.norm([0.4, 1.0, -2.0, 0.3]),
              boost_velocity=np.array([0.012, -0.02, 0.015]))
    out = dict(t=t, ell_max=L, **kw)
    # h (spin -2, l = 2..6) and sigma (spin 2): the inhomogeneous-term branches (waveform_grid.py:485-503)
    for name, dt, seed in (("h", scri.h, 82), ("sigma", scri.sigma, 83)):
        data = synthetic.chirp_modes(t, 2, L, seed) * (1 + 0.01 * t[:, None])
        w = _wm(t, data, 2, L, dt).transform(**kw)
        out[f"{name}_in"], out[f"{name}_t_out"], out[f"{name}_out"] = data, w.t, w.data
        assert w.ell_min == 2 and w.ell_max == L
    # psi2 (spin 0, l = 0..6) with psi3 / psi4 companions: the mixing branch (waveform_grid.py:504-550)
    d2 = synthetic.chirp_modes(t, 0, L, 84)
    d3 = synthetic.chirp_modes(t, 1, 5, 85)
    d4 = synthetic.chirp_modes(t, 2, 4, 86)
    w = _wm(t, d2, 0, L, scri.psi2).transform(psi3_modes=_wm(t, d3, 1, 5, scri.psi3), psi4_modes=_wm(t, d4, 2, 4, scri.psi4), **kw)
    out.update(psi2_in=d2, psi3_in=d3, psi4_in=d4, psi2_t_out=w.t, psi2_out=w.data)
    # a pure supertranslation through explicit n_theta / n_phi / ell_max kwargs (waveform_grid.py:89-110, 615-630)
    w = _wm(t, out["h_in"], 2, L, scri.h).transform(supertranslation=kw["supertranslation"], n_theta=23, n_phi=25, ell_max=5)
    out.update(h_st_t_out=w.t, h_st_out=w.data)
    np.savez_compressed(os.path.join(HERE, "g8_ref_wm_transform.npz"), source="/root/reference/scri (unmodified) on stand-ins", **out)


def g9():
    n, L = 120, 4
    u = np.linspace(-8.0, 16.0, n)
    abd = scri.AsymptoticBondiData(u, L)
    raw = np.zeros((6, n, (L + 1) ** 2), dtype=complex)
    for f, s in enumerate(synthetic.ABD_SPINS):
        raw[f] = synthetic.chirp_modes(u, 0, L, 90 + f) * (1 + 0.02 * u[:, None])
        raw[f, :, : s * s] = 0
    abd.psi0, abd.psi1, abd.psi2, abd.psi3, abd.psi4, abd.sigma = raw
    kw = dict(supertranslation=_real_supertranslation(2, 97, 0.04), frame_rotation=np.array([1.0, 0.5, -0.2, 0.1]) / np.linalg.norm([1.0, 0.5, -0.2, 0.1]),
              boost_velocity=np.array([0.01, 0.02, -0.015]))
    new = abd.transform(**kw)
    out = dict(u=u, raw=raw, ell_max=L, u_out=np.array(new.t), raw_out=np.array(new._raw_data), **kw)
    new = abd.transform(space_translation=np.array([0.2, -0.1, 0.3]), working_ell_max=7, output_ell_max=3)
    out.update(u_out_b=np.array(new.t), raw_out_b=np.array(new._raw_data), space_translation_b=np.array([0.2, -0.1, 0.3]))
    np.savez_compressed(os.path.join(HERE, "g9_ref_abd_transform.npz"), source="/root/reference/scri (unmodified) on stand-ins", **out)


def g10():
    t, data, rot = synthetic.cfg1()
    t, data = t[::8], data[::8]  # 250 steps of cfg1
    R = rot["series"][::8]
    q = rot["constant"]
    w = _wm(t, data.copy(), 2, 4, scri.h)
    w.rotate_decomposition_basis(np.quaternion(*q))
    out = dict(t=t, data=data, constant=q, series=R, const_out=w.data.copy(), const_frame=standins.as_float_array(w.frame))
    w.rotate_decomposition_basis(standins.as_quat_array(R))  # a series on top: frame right-multiplied (rotations.py:313-323)
    out.update(series_after_const_out=w.data.copy(), series_after_const_frame=standins.as_float_array(w.frame))
    w = _wm(t, data.copy(), 2, 4, scri.h)
    w.rotate_decomposition_basis(standins.as_quat_array(R))
    out.update(series_out=w.data.copy(), series_frame=standins.as_float_array(w.frame))
    w = _wm(t, data.copy(), 2, 4, scri.h)
    w.rotate_physical_system(np.quaternion(*q))
    out.update(physical_out=w.data.copy(), physical_frame=standins.as_float_array(w.frame))
    np.savez_compressed(os.path.join(HERE, "g10_ref_rotations.npz"), source="/root/reference/scri (unmodified) on stand-ins", **out)


def g11():
    import types

    from scipy.interpolate import CubicSpline

    # numpy-quaternion's parity ufuncs on an empty frame array: nothing to reflect
    for d in ("x_", "y_", "z_", ""):
        for part in ("conjugate", "symmetric_part", "antisymmetric_part"):
            setattr(np, f"{d}parity_{part}", lambda f: f)
    calculus = types.ModuleType("quaternion.calculus")
    calculus.spline_definite_integral = lambda f, t, t1=None, t2=None, axis=-1: CubicSpline(t, f, axis=axis).integrate(
        t[0] if t1 is None else t1, t[-1] if t2 is None else t2)
    sys.modules["quaternion.calculus"] = calculus
    sys.modules["quaternion"].calculus = calculus
    from scri.extrapolation import intersection

    out = {}
    n = 24
    t = np.linspace(0.0, 9.0, n) + 0.05 * np.sin(np.arange(n))
    out["t"] = t
    for name, dt, lmin, lmax, seed in (("psi1", scri.psi1, 1, 4, 5), ("psi4", scri.psi4, 2, 5, 6), ("h", scri.h, 2, 4, 7), ("psi2", scri.psi2, 0, 3, 8)):
        rng = np.random.default_rng(seed)
        nm = (lmax + 1) ** 2 - lmin**2
        data = rng.normal(size=(n, nm)) + 1j * rng.normal(size=(n, nm))
        out[f"{name}_in"], out[f"{name}_ells"] = data, np.array([lmin, lmax])
        w = _wm(t, data.copy(), lmin, lmax, dt)
        for d in ("x_", "y_", "z_", ""):
            for part in ("conjugate", "symmetric_part", "antisymmetric_part"):
                out[f"{name}_{d}parity_{part}"] = getattr(w, f"{d}parity_{part}").data
            out[f"{name}_{d}parity_violation_squared"] = getattr(w, f"{d}parity_violation_squared")
        for k, (ops, conv) in enumerate((("+", "NP"), ("-", "NP"), ("-+", "NP"), ("+-", "GHP"), ([+1, -1, -1], "NP"))):
            out[f"{name}_eth{k}"] = w.apply_eth(ops, eth_convention=conv)
        p = _wm(t, data.copy(), lmin, lmax, dt)
        p.convert_to_conjugate_pairs()
        out[f"{name}_pairs"] = p.data.copy()
        p.convert_from_conjugate_pairs()
        out[f"{name}_pairs_back"] = p.data.copy()
        for tol in (1e-10, 1e-3):
            q = _wm(t, data.copy(), lmin, lmax, dt)
            q.truncate(tol)
            out[f"{name}_truncate_{tol:g}"] = q.data.copy()
        other = _wm(t, rng.normal(size=(n, nm)) + 1j * rng.normal(size=(n, nm)), lmin, lmax, dt)
        out[f"{name}_other"] = other.data
        out[f"{name}_inner"] = np.array([w.inner_product(other), w.inner_product(other, t1=2.0, t2=7.0)])
    t2 = np.linspace(0.7, 11.0, 31) + 0.08 * np.cos(np.arange(31))
    out["t2"] = t2
    out["intersection"] = intersection(t, t2)
    out["intersection_min_step"] = intersection(t, t2, min_step=0.25)
    out["intersection_bounds"] = intersection(t, t2, min_time=1.5, max_time=8.0)
    np.savez_compressed(os.path.join(HERE, "g11_ref_mode_operators.npz"), source="/root/reference/scri (unmodified) on stand-ins", **out)


def g12():
    n, L = 90, 4
    u = np.linspace(-5.0, 20.0, n)
    raw = np.zeros((6, n, (L + 1) ** 2), dtype=complex)
    for f, s in enumerate(synthetic.ABD_SPINS):
        raw[f] = 0.3 * synthetic.chirp_modes(u, 0, L, 120 + f) * (1 + 0.02 * u[:, None])
        raw[f, :, : s * s] = 0
    raw[2, :, 0] -= 5.0 * np.sqrt(4 * np.pi)  # a dominant mass monopole keeps the four-momentum timelike
    abd = scri.AsymptoticBondiData(u, L)
    abd.psi0, abd.psi1, abd.psi2, abd.psi3, abd.psi4, abd.sigma = raw
    out = dict(u=u, raw=raw, ell_max=L)
    out["mass_aspect"] = np.asarray(abd.mass_aspect()).view(np.ndarray)
    out["mass_aspect_ell2"] = np.asarray(abd.mass_aspect(truncate_ell=2)).view(np.ndarray)
    for name in ("bondi_rest_mass", "bondi_four_momentum", "bondi_angular_momentum", "bondi_boost_charge", "bondi_CoM_charge",
                 "bondi_dimensionless_spin"):
        out[name] = np.asarray(getattr(abd, name)())
    np.savez_compressed(os.path.join(HERE, "g12_ref_bms_charges.npz"), source="/root/reference/scri (unmodified) on stand-ins", **out)


def g13():
    out = {}
    for tag, lmin, lmax, dt, seed in (("a", 2, 6, scri.h, 131), ("b", 0, 4, scri.psi2, 132)):
        n = 120
        t = np.linspace(0.0, 30.0, n) + 0.04 * np.sin(np.arange(n))
        data = synthetic.chirp_modes(t, lmin, lmax, seed) * (1 + 0.01 * t[:, None])
        w = _wm(t, data, lmin, lmax, dt)
        out[f"{tag}_t"], out[f"{tag}_data"], out[f"{tag}_ells"] = t, data, np.array([lmin, lmax])
        out[f"{tag}_LdtVector"] = np.asarray(w.LdtVector())
        out[f"{tag}_LLMatrix"] = np.asarray(w.LLMatrix())
        out[f"{tag}_angular_velocity"] = np.asarray(w.angular_velocity())
        out[f"{tag}_LLDominantEigenvector"] = np.asarray(w.LLDominantEigenvector())
        import scri.mode_calculations as mc

        other = _wm(t, synthetic.chirp_modes(t, lmin, lmax, seed + 50) * (1 - 0.02 * t[:, None]), lmin, lmax, dt)
        out[f"{tag}_other"] = other.data
        out[f"{tag}_LVector"] = np.asarray(mc.LVector(w, other))
        out[f"{tag}_LLComparisonMatrix"] = np.asarray(mc.LLComparisonMatrix(w, other))
    np.savez_compressed(os.path.join(HERE, "g13_ref_mode_calculations.npz"), source="/root/reference/scri (unmodified) on stand-ins", **out)


def g14():
    import functools

    import scri.flux as flux

    n, lmin, lmax = 64, 2, 5
    t = np.linspace(3.0, 35.0, n) + 0.05 * np.sin(np.arange(n))
    data = synthetic.chirp_modes(t, lmin, lmax, 141) * (1 + 0.01 * t[:, None])
    h = _wm(t, data, lmin, lmax, scri.h)
    out = dict(t=t, data=data, ells=np.array([lmin, lmax]))
    out["energy_flux"] = flux.energy_flux(h)
    out["momentum_flux"] = flux.momentum_flux(h)
    out["angular_momentum_flux"] = flux.angular_momentum_flux(h)
    out["boost_flux"] = flux.boost_flux(h)
    e, p, j, b = flux.poincare_fluxes(h)
    out.update(poincare_e=e, poincare_p=p, poincare_j=j, poincare_b=b)
    hdot = h.copy()
    hdot.dataType = scri.hdot
    hdot.data = h.data_dot
    out["hdot"] = hdot.data
    # single expectation values: a = hdot-like, b = h-like objects of spin s (the data are just numbers here)
    for s in (-1, -2, -3):
        for name, gen in (("p_z", flux.p_z), ("p_plus", flux.p_plus), ("p_minus", flux.p_minus)):
            out[f"{name}_s{-s}"] = flux.matrix_expectation_value(hdot, functools.partial(gen, s=s), h)[1]
    for name, gen in (("j_z", flux.j_z), ("j_plus", flux.j_plus), ("j_minus", flux.j_minus)):
        out[name] = flux.matrix_expectation_value(hdot, gen, h)[1]
    np.savez_compressed(os.path.join(HERE, "g14_ref_fluxes.npz"), source="/root/reference/scri (unmodified) on stand-ins", **out)


def g15():
    """Extra trailing data dimensions (scri/waveform_grid.py:299-308, 574-594).  What the reference's own source
    does with data[N, n_modes, 2]: to_modes walks `final_dim` (recorded: the grid of the g8 `h` transformation, two series side by
    side, analysed back); from_modes / transform cannot run -- np.tensordot at :475-484 leaves the extra axis BEFORE the grid axes,
    the loop at :581 indexes axis 1 with the ring number and raises IndexError (recorded as the exception's type and text)."""
    n, L = 240, 6
    t = np.linspace(-12.0, 36.0, n)
    kw = dict(supertranslation=_real_supertranslation(3, 81, 0.05), frame_rotation=np.array([0.4, 1.0, -2.0, 0.3]) / np.linalg.norm([0.4, 1.0, -2.0, 0.3]),
              boost_velocity=np.array([0.012, -0.02, 0.015]))
    a = synthetic.chirp_modes(t, 2, L, 82) * (1 + 0.01 * t[:, None])
    b = synthetic.chirp_modes(t, 2, L, 87)
    data = np.stack([a, b], axis=2)  # [N, n_modes, 2]
    out = dict(t=t, ell_max=L, seeds=np.array([82, 87]), **kw)  # (the input is regenerated from the seeds by the tests)
    try:
        _wm(t, data, 2, L, scri.h).transform(**kw)
        out["transform_exception"] = "none"
    except Exception as e:  # noqa: BLE001
        out["transform_exception"] = f"{type(e).__name__}: {e}"
    # the two series one by one (what the loops say), then both grids side by side through the reference's to_modes
    grids, outs = [], []
    for d in (a, b):
        g = scri.WaveformGrid.from_modes(_wm(t, d, 2, L, scri.h), **kw)
        grids.append(g.data)
        outs.append(g.to_modes(L).data)
    g2 = scri.WaveformGrid(t=g.t, data=np.stack(grids, axis=2), n_theta=g.n_theta, n_phi=g.n_phi, frameType=scri.Inertial, dataType=scri.h,
                           r_is_scaled_out=True, m_is_scaled_out=True)
    m2 = g2.to_modes(L)
    assert m2.data.shape == outs[0].shape + (2,)
    keep = slice(None, None, 10)  # (to_modes works time by time: every tenth row keeps the fixture small)
    out.update(t_out=g.t, rows_kept=np.arange(g.t.size)[keep], n_theta=g.n_theta, n_phi=g.n_phi, grid_two=g2.data[keep], modes_two=m2.data[keep],
               rows_kept_single=np.arange(g.t.size)[::4], modes_a=outs[0][::4], modes_b=outs[1][::4])
    np.savez_compressed(os.path.join(HERE, "g15_ref_trailing_dims.npz"), source="/root/reference/scri (unmodified) on stand-ins", **out)


ORDERS = [
    ["supertranslation", "frame_rotation", "boost_velocity"],
    ["supertranslation", "boost_velocity", "frame_rotation"],
    ["frame_rotation", "supertranslation", "boost_velocity"],
    ["frame_rotation", "boost_velocity", "supertranslation"],
    ["boost_velocity", "supertranslation", "frame_rotation"],
    ["boost_velocity", "frame_rotation", "supertranslation"],
]


def g16():
    import scri.bms_transformations as bt

    L = 12  # the reference's default; its reorder builds intermediate objects at the default and raises on any other (:419 -> :281)
    S = np.zeros((L + 1) ** 2, dtype=complex)
    S[:25] = _real_supertranslation(4, 161, 0.1)
    q = np.array([0.7, -0.3, 0.5, 0.4])
    q /= np.linalg.norm(q)
    v = np.array([0.09, -0.11, 0.095])  # |v| = 0.171: second-order boost terms (3e-2) are far above the comparison bar
    S2 = np.zeros((L + 1) ** 2, dtype=complex)
    S2[:16] = _real_supertranslation(3, 162, 0.07)
    q2 = np.array([-0.2, 0.6, 0.1, 0.75])
    q2 /= np.linalg.norm(q2)
    v2 = np.array([-0.05, 0.12, 0.03])

    def parts(b):
        return np.array(b.supertranslation), np.array(b.frame_rotation.components, dtype=float), np.array(b.boost_velocity, dtype=float)

    out = dict(ell_max=L, S=S, q=q, v=v, S2=S2, q2=q2, v2=v2, orders=np.array(["|".join(o) for o in ORDERS]))
    rS, rq, rv = np.zeros((6, 6, S.size), dtype=complex), np.zeros((6, 6, 4)), np.zeros((6, 6, 3))
    for i, o_in in enumerate(ORDERS):
        B = bt.BMSTransformation(supertranslation=S, frame_rotation=q, boost_velocity=v, order=list(o_in))
        for j, o_out in enumerate(ORDERS):
            R = B.reorder(list(o_out))
            assert R.order == o_out
            rS[i, j], rq[i, j], rv[i, j] = parts(R)
    out.update(reorder_S=rS, reorder_q=rq, reorder_v=rv)
    iS, iq, iv, io = np.zeros((6, S.size), dtype=complex), np.zeros((6, 4)), np.zeros((6, 3)), []
    for i, o_in in enumerate(ORDERS):
        Bi = bt.BMSTransformation(supertranslation=S, frame_rotation=q, boost_velocity=v, order=list(o_in)).inverse()
        iS[i], iq[i], iv[i] = parts(Bi)
        io.append("|".join(Bi.order))
    out.update(inverse_S=iS, inverse_q=iq, inverse_v=iv, inverse_orders=np.array(io))
    # an explicit output order of the inverse
    Bx = bt.BMSTransformation(supertranslation=S, frame_rotation=q, boost_velocity=v, order=list(ORDERS[2])).inverse(output_order=list(ORDERS[4]))
    out["inverse_explicit_S"], out["inverse_explicit_q"], out["inverse_explicit_v"] = parts(Bx)
    # two compositions (other * self): normal order x normal order, and two mixed orders
    for tag, (o1, o2) in (("a", (ORDERS[0], ORDERS[0])), ("b", (ORDERS[3], ORDERS[4]))):
        B1 = bt.BMSTransformation(supertranslation=S, frame_rotation=q, boost_velocity=v, order=list(o1))
        B2 = bt.BMSTransformation(supertranslation=S2, frame_rotation=q2, boost_velocity=v2, order=list(o2))
        C = B1 * B2
        out[f"compose_{tag}_S"], out[f"compose_{tag}_q"], out[f"compose_{tag}_v"] = parts(C)
        out[f"compose_{tag}_orders"] = np.array(["|".join(o1), "|".join(o2), "|".join(C.order)])
    # LorentzTransformation: reorder both ways, inverse (default and explicit order), product
    fb, bf = ["frame_rotation", "boost_velocity"], ["boost_velocity", "frame_rotation"]
    for tag, o_in, o_out in (("fb_bf", fb, bf), ("bf_fb", bf, fb)):
        Lr = bt.LorentzTransformation(frame_rotation=q, boost_velocity=v, order=list(o_in)).reorder(list(o_out))
        out[f"lorentz_reorder_{tag}_q"], out[f"lorentz_reorder_{tag}_v"] = np.array(Lr.frame_rotation.components), np.array(Lr.boost_velocity)
        Li = bt.LorentzTransformation(frame_rotation=q, boost_velocity=v, order=list(o_in)).inverse()
        assert Li.order == o_in[::-1]
        out[f"lorentz_inverse_{tag[:2]}_q"], out[f"lorentz_inverse_{tag[:2]}_v"] = np.array(Li.frame_rotation.components), np.array(Li.boost_velocity)
    Li = bt.LorentzTransformation(frame_rotation=q, boost_velocity=v, order=list(fb)).inverse(output_order=list(fb))
    out["lorentz_inverse_fb_to_fb_q"], out["lorentz_inverse_fb_to_fb_v"] = np.array(Li.frame_rotation.components), np.array(Li.boost_velocity)
    Lp = bt.LorentzTransformation(frame_rotation=q, boost_velocity=v, order=list(bf)) * bt.LorentzTransformation(frame_rotation=q2, boost_velocity=v2)
    out["lorentz_product_q"], out["lorentz_product_v"] = np.array(Lp.frame_rotation.components), np.array(Lp.boost_velocity)
    # transform_supertranslation on its own (bms_transformations.py:151-180)
    out["transformed_S"] = bt.transform_supertranslation(S, bt.LorentzTransformation(frame_rotation=q, boost_velocity=v))
    # the quirk: away from the default ell_max most reorders raise (the intermediate objects are built with ell_max = 12 and the
    # padding of the shorter supertranslation goes negative, :419 -> :281): recorded per (input order, output order) pair at ell_max = 8
    raises = np.zeros((6, 6), dtype=bool)
    for i, o_in in enumerate(ORDERS):
        for j, o_out in enumerate(ORDERS):
            try:
                bt.BMSTransformation(supertranslation=S[:81], frame_rotation=q, boost_velocity=v, ell_max=8, order=list(o_in)).reorder(list(o_out))
            except ValueError:
                raises[i, j] = True
    out["reorder_at_ell_max_8_raises"] = raises
    np.savez_compressed(os.path.join(HERE, "g16_ref_bms_algebra.npz"), source="scri/bms_transformations.py:183-592 (the reference's file, stand-ins underneath)", **out)


def g17():
    import scri.utilities as ut
    from scri.SpEC.file_io import index_is_monotonic

    rng = np.random.default_rng(171)
    out = {}
    # xor_timeseries / _reverse: complex [N, k] (the storage format's use), real [N, k], one row, one column
    for tag, arr in (("c", rng.normal(size=(40, 5)) + 1j * rng.normal(size=(40, 5))), ("f", rng.normal(size=(33, 3))),
                     ("row", rng.normal(size=(1, 6)) + 1j * rng.normal(size=(1, 6))), ("col", rng.normal(size=(17, 1)))):
        x = ut.xor_timeseries(arr.copy())
        back = ut.xor_timeseries_reverse(x.copy())
        assert np.array_equal(back.view(np.uint64), arr.view(np.uint64))
        out[f"xor_{tag}_in"], out[f"xor_{tag}_out"] = arr.view(np.uint64).copy(), x.view(np.uint64).copy()
    # multishuffle: four word sizes, six width tuples each (uniform bits / bytes / whole word, straddling, decreasing, increasing)
    tuples = {
        8: [(1,) * 8, (8,), (4, 4), (3, 5), (2, 2, 4), (1, 7)],
        16: [(1,) * 16, (8, 8), (16,), (5, 3, 7, 1), (8, 4, 2, 2), (1, 1, 2, 4, 8)],
        32: [(1,) * 32, (8,) * 4, (32,), (11, 9, 12), (8, 8, 4, 4, 2, 2, 1, 1, 1, 1), (1, 3, 4, 8, 16)],
        64: [(1,) * 64, (8,) * 8, (64,), (13, 17, 3, 31), (8, 8, 4, 4, 4, 4, 2, 2, 2, 2, 1, 1, 1, 1, 4, 16), (16, 16, 32)],
    }
    for bits, ws in tuples.items():
        dt = np.dtype(f"u{bits // 8}")
        # an odd element count: pieces straddle words everywhere.  (8-bit words: 31 elements -- the reference's unshuffle keeps its bit
        # cursor in the word's own dtype, which numba widens but plain numpy scalars do not: past 255 bits it would wrap HERE only)
        data = rng.integers(0, 2**bits, size=31 if bits == 8 else 61, dtype=dt)
        out[f"shuffle_{bits}_in"] = data
        for k, w in enumerate(ws):
            sh = ut.multishuffle(tuple(w))(data.copy())
            un = ut.multishuffle(tuple(w), forward=False)(sh.copy())
            assert sh.dtype == dt and np.array_equal(un, data), (bits, w)
            out[f"shuffle_{bits}_{k}_widths"], out[f"shuffle_{bits}_{k}_out"], out[f"shuffle_{bits}_{k}_back"] = np.array(w), sh, un
    # fletcher32: lengths below, at and beyond the 360-word block, odd and even; one float array
    for n in (1, 2, 359, 360, 361, 721, 1001, 1440):
        d = rng.integers(0, 2**16, size=n, dtype=np.uint16)
        out[f"fletcher_{n}_in"], out[f"fletcher_{n}_out"] = d, np.uint32(ut.fletcher32(d))
    f = rng.normal(size=(37, 5))
    out["fletcher_f8_in"], out["fletcher_f8_out"] = f, np.uint32(ut.fletcher32(f))
    # index_is_monotonic: increasing with repeats and dips, decreasing, constant ends
    for tag, y in (("up", np.array([0.0, 1, 2, 2, 3, 2.5, 2.9, 3.0, 3.1, 10, 9, 11])), ("down", np.array([5.0, 4, 4.5, 3, 3, 2, 2.5, 1, 0])),
                   ("flat", np.array([1.0, 2, 0.5, 1.0])), ("noise", np.cumsum(rng.normal(0.3, 1.0, size=200)))):
        out[f"mono_{tag}_in"], out[f"mono_{tag}_out"] = y, np.array(index_is_monotonic(y))
    # transition_function (scri/utilities.py:12-58; used by rotations.py when a frame is frozen)
    x = np.linspace(-0.3, 1.4, 173)
    for tag, args in (("a", (0.2, 0.8)), ("b", (0.0, 1.0, 3.0, -1.0)), ("c", (0.35, 0.351))):
        f, i0, i1 = ut.transition_function(x, *args, return_indices=True)
        out[f"transition_{tag}_args"], out[f"transition_{tag}_out"], out[f"transition_{tag}_idx"] = np.array(args), np.array(f), np.array([i0, i1])
    out["transition_x"] = x
    # ... its derivative, the bump built from two of them, and transition_to_constant (utilities.py:60-190: sample_waveforms' helpers)
    xs = np.linspace(-0.3, 1.4, 173)[:-3]  # (the reference's loops index past the end when x never reaches x1 / x3)
    out["transition_deriv_x"] = xs
    out["transition_deriv_out"] = ut.transition_function_derivative(xs, 0.2, 0.8, 1.0, -2.0)
    out["bump_args"] = np.array([0.1, 0.4, 0.5, 1.2, 0.5, 2.0, -1.0])
    out["bump_out"] = ut.bump_function(xs, *out["bump_args"])
    tt = np.linspace(0.0, 10.0, 401)
    ff = np.sin(1.3 * tt) + 0.2 * tt
    out["to_constant_t"], out["to_constant_f"], out["to_constant_out"] = tt, ff, ut.transition_to_constant(ff.copy(), tt, 3.0, 7.5)
    np.savez_compressed(os.path.join(HERE, "g17_ref_bit_transforms.npz"), source="scri/utilities.py:194-406, scri/SpEC/file_io/__init__.py:50-70 (the reference's files, identity njit)", **out)


def g18():
    from scri.modes_time_series import ModesTimeSeries
    import spherical_functions as sf

    rng = np.random.default_rng(181)
    n = 70
    u = np.sort(rng.uniform(-4.0, 11.0, n)) + np.arange(n) * 1e-3
    a = synthetic.chirp_modes(u, 0, 5, 182) * (1 + 0.03 * u[:, None])
    a[:, :1] = 0  # spin -1: nothing below l = 1
    b = synthetic.chirp_modes(u, 0, 4, 183)
    b[:, :4] = 0  # spin 2
    A = ModesTimeSeries(a, time=u, spin_weight=-1, ell_min=0, ell_max=5, multiplication_truncator=max)
    B = ModesTimeSeries(b, time=u, spin_weight=2, ell_min=0, ell_max=4, multiplication_truncator=max)
    new_time = np.concatenate([np.linspace(u[0], u[-1], 41), u[::9]])
    out = dict(u=u, a=a, b=b, new_time=new_time)
    arr = lambda m: np.asarray(m).view(np.ndarray)
    for order in (-2, -1, 0, 1, 2, 3):
        out[f"a_interp_{order}"] = arr(A.interpolate(new_time, derivative_order=order))
    for name in ("dot", "ddot", "int", "iint", "eth_GHP", "ethbar_GHP"):
        r = getattr(A, name)
        out[f"a_{name}"] = arr(r)
        out[f"a_{name}_meta"] = np.array([r.spin_weight, r.ell_min, r.ell_max])
    for tag, kw in (("default", {}), ("wide", dict(working_ell_max=12, output_ell_max=7)), ("narrow", dict(working_ell_max=9, output_ell_max=2))):
        P = A.grid_multiply(B, **kw)
        out[f"ab_{tag}"] = arr(P)
        out[f"ab_{tag}_meta"] = np.array([P.spin_weight, P.ell_min, P.ell_max])
    P = B.grid_multiply(A)  # output l_max follows the FIRST factor
    out["ba_default"], out["ba_default_meta"] = arr(P), np.array([P.spin_weight, P.ell_min, P.ell_max])
    np.savez_compressed(os.path.join(HERE, "g18_ref_modes_time_series.npz"), source="scri/modes_time_series.py:72-202 (the reference's file, stand-ins underneath)", **out)


def g19():
    import scri.asymptotic_bondi_data.map_to_superrest_frame as ms

    L, n = 4, 60
    u = np.linspace(-30.0, 40.0, n)
    raw = np.zeros((6, n, (L + 1) ** 2), dtype=complex)
    for f, s_ in enumerate(synthetic.ABD_SPINS):
        raw[f] = 0.05 * synthetic.chirp_modes(u, 0, L, 190 + f) * (1 + 0.01 * u[:, None])
        raw[f, :, : s_ * s_] = 0
    raw[2, :, 0] -= 1.0 * np.sqrt(4 * np.pi)  # a unit mass monopole: timelike four-momentum
    abd = scri.AsymptoticBondiData(u, L)
    abd.psi0, abd.psi1, abd.psi2, abd.psi3, abd.psi4, abd.sigma = raw
    arr = lambda m: np.asarray(m).view(np.ndarray)
    out = dict(u=u, raw=raw, ell_max=L)
    out["D"], out["Dinv"] = ms.𝔇(np.array(raw[2]), L), ms.𝔇inverse(np.array(raw[2]), L)
    PsiM = abd.supermomentum("Moreschi")
    out["PsiM"] = arr(PsiM)
    M_Grid, K_Grid = ms.compute_bondi_rest_mass_and_conformal_factor(np.array(PsiM), L)
    out["M_Grid"], out["K_Grid"] = np.asarray(M_Grid), np.asarray(K_Grid)
    # a smooth real supertranslation on the (2 L + 1)^2 grid, a few time units in size (the supermomentum is evaluated at u = alpha)
    import spinsfast

    alpha = spinsfast.salm2map(_real_supertranslation(3, 191, 1.5), 0, 3, 2 * L + 1, 2 * L + 1).real
    out["alpha"] = alpha
    moreschi = ms.compute_Moreschi_supermomentum(PsiM, alpha, L)
    out["PsiM_at_alpha"] = arr(moreschi)
    M1, K1 = ms.compute_bondi_rest_mass_and_conformal_factor(np.array(moreschi), L)
    out["M_at_alpha"], out["K_at_alpha"] = np.asarray(M1), np.asarray(K1)
    out["alpha_perturbation"] = np.asarray(ms.compute_alpha_perturbation(moreschi, M1, K1, L))
    B, rel_errs = ms.supertranslation_to_map_to_superrest_frame(abd, N_itr_max=3, ell_max=L)
    out["superrest_supertranslation"], out["superrest_rel_errs"] = np.array(B.supertranslation), np.array(rel_errs[1:])
    G = abd.bondi_CoM_charge() / abd.bondi_four_momentum()[:, 0, None]
    Bc = ms.transformation_from_CoM_charge(G, u)
    out["com_G"], out["com_supertranslation"], out["com_boost"], out["com_order"] = G, np.array(Bc.supertranslation), np.array(Bc.boost_velocity), np.array("|".join(Bc.order))
    chi = np.array([[0.1 + 0.001 * k, -0.2, 0.6] for k in range(n)])
    for tag, kw in (("free", {}), ("xz", dict(fix_xz_plane=True)), ("yz", dict(fix_yz_plane=True))):
        out[f"spin_rotation_{tag}"] = np.array(ms.rotation_from_spin_charge(chi, u, **kw).frame_rotation.components)
    out["chi"] = chi
    tt = ms.time_translation(abd, 3.0)
    out["time_translation_u"], out["time_translation_raw"] = np.array(tt.t), np.array([arr(getattr(tt, f)) for f in ("psi0", "psi1", "psi2", "psi3", "psi4", "sigma")])
    np.savez_compressed(os.path.join(HERE, "g19_ref_superrest_helpers.npz"),
                        source="scri/asymptotic_bondi_data/map_to_superrest_frame.py:76-507,666-684 (the reference's file, stand-ins underneath)", **out)


def _g19_abd():
    L, n = 4, 60
    u = np.linspace(-30.0, 40.0, n)
    raw = np.zeros((6, n, (L + 1) ** 2), dtype=complex)
    for f, s_ in enumerate(synthetic.ABD_SPINS):
        raw[f] = 0.05 * synthetic.chirp_modes(u, 0, L, 190 + f) * (1 + 0.01 * u[:, None])
        raw[f, :, : s_ * s_] = 0
    raw[2, :, 0] -= 1.0 * np.sqrt(4 * np.pi)
    abd = scri.AsymptoticBondiData(u, L)
    abd.psi0, abd.psi1, abd.psi2, abd.psi3, abd.psi4, abd.sigma = raw
    return u, raw, L, abd


def g20():
    import scri.asymptotic_bondi_data.map_to_superrest_frame as ms

    u, raw, L, abd = _g19_abd()
    arr = lambda m: np.asarray(m).view(np.ndarray)
    fields = lambda a: np.array([arr(getattr(a, f)) for f in ("psi0", "psi1", "psi2", "psi3", "psi4", "sigma")])

    def parts(B):
        return np.array(B.supertranslation), np.array(B.frame_rotation.components, dtype=float), np.array(B.boost_velocity, dtype=float), np.array("|".join(B.order))

    out = dict(u=u, raw=raw, ell_max=L)
    B, errs = ms.com_transformation_to_map_to_superrest_frame(abd, N_itr_max=2)
    out["com_S"], out["com_q"], out["com_v"], out["com_order"] = parts(B)
    out["com_rel_errs"] = np.array(errs[1:], dtype=float)
    B, errs = ms.rotation_to_map_to_superrest_frame(abd, N_itr_max=2)
    out["rot_S"], out["rot_q"], out["rot_v"], out["rot_order"] = parts(B)
    out["rot_rel_errs"] = np.array(errs[1:], dtype=float)
    out["rel_err_in_superrest"] = np.array(ms.rel_err_for_abd_in_superrest(abd, None, None), dtype=float)
    iters = {"superrest": 2, "CoM_transformation": 2, "rotation": 2, "supertranslation": 2}
    abd_prime, B, best = ms.map_to_superrest_frame(abd, t_0=2.0, padding_time=25, N_itr_maxes=iters, ell_max=L)
    out["whole_S"], out["whole_q"], out["whole_v"], out["whole_order"] = parts(B)
    out["whole_best_rel_err"] = np.array(best, dtype=float)
    out["whole_u"], out["whole_raw"] = np.array(abd_prime.t), fields(abd_prime)
    # the same loop towards a TARGET Moreschi supermomentum (map_to_superrest_frame.py:268-276, 293-305, 848-859): that of the same data
    # supertranslated a little, handed over as the reference's callers do (a WaveformModes of the supermomentum)
    target_abd = abd.transform(supertranslation=_real_supertranslation(2, 201, 0.4))
    target = ms.MT_to_WM(target_abd.supermomentum("Moreschi"), dataType=scri.psi2)
    out["target_t"], out["target_modes"] = np.array(target.t), np.array(target.data)
    abd_prime, B, best = ms.map_to_superrest_frame(abd, t_0=2.0, target_PsiM_input=target, padding_time=20, N_itr_maxes=iters, ell_max=L)
    out["target_S"], out["target_q"], out["target_v"], out["target_order"] = parts(B)
    out["target_best_rel_err"] = np.array(best, dtype=float)
    out["target_u"], out["target_raw"] = np.array(abd_prime.t), fields(abd_prime)
    np.savez_compressed(os.path.join(HERE, "g20_ref_map_to_superrest_frame.npz"),
                        source="scri/asymptotic_bondi_data/map_to_superrest_frame.py:369-1035 (the reference's file, stand-ins underneath)", **out)


def g21():
    import scri.asymptotic_bondi_data.map_to_abd_frame as ma

    u, raw, L, abd = _g19_abd()
    arr = lambda m: np.asarray(m).view(np.ndarray)
    fields = lambda a: np.array([arr(getattr(a, f)) for f in ("psi0", "psi1", "psi2", "psi3", "psi4", "sigma")])
    kw = dict(supertranslation=_real_supertranslation(2, 211, 0.3), frame_rotation=np.array([0.99, 0.05, -0.08, 0.1]) / np.linalg.norm([0.99, 0.05, -0.08, 0.1]),
              boost_velocity=np.array([1e-4, -2e-4, 1.5e-4]))
    target = abd.transform(**kw)
    out = dict(u=u, raw=raw, ell_max=L, target_u=np.array(target.t), target_raw=fields(target), **{"kw_" + k: v for k, v in kw.items()})
    out["rel_err_between"] = np.array(ma.rel_err_between_abds(abd, target, -10.0, 10.0))
    iters = {"abd": 2, "superrest": 1, "CoM_transformation": 2, "rotation": 2, "supertranslation": 2}
    abd_prime, B, rel_err = abd.map_to_abd_frame(target, t_0=2.0, padding_time=18, N_itr_maxes=iters, ell_max=L, fix_time_phase_freedom=False)
    out["S"], out["q"], out["v"] = np.array(B.supertranslation), np.array(B.frame_rotation.components, dtype=float), np.array(B.boost_velocity, dtype=float)
    out["order"], out["rel_err"] = np.array("|".join(B.order)), np.array(rel_err, dtype=float)
    out["prime_u"], out["prime_raw"] = np.array(abd_prime.t), fields(abd_prime)
    np.savez_compressed(os.path.join(HERE, "g21_ref_map_to_abd_frame.npz"),
                        source="scri/asymptotic_bondi_data/map_to_abd_frame.py:21-290 (the reference's file, stand-ins underneath)", **out)


def g22():
    """error and warning behaviour: (callable, kwargs) -> exception type + text, or the warnings' texts, as the reference produces them"""
    import json
    import scri.waveform_grid as wg
    import scri.asymptotic_bondi_data.transformations as tr_mod

    st9 = (np.array([1, 2 + 4j, 3, -2 + 4j, 7 - 5j, -3 - 2j, 4, 3 - 2j, 7 + 5j]) * 1e-3)
    real9 = synthetic.real_supertranslation(st9)

    def enc(v):
        if isinstance(v, np.ndarray):
            if np.iscomplexobj(v):
                return {"__complex__": [[float(z.real), float(z.imag)] for z in v.ravel()], "shape": list(v.shape)}
            return {"__array__": v.tolist()}
        if isinstance(v, (list, tuple)):
            return [enc(x) for x in v]
        if isinstance(v, (np.floating, np.integer)):
            return v.item()
        return v

    kw_cases = [
        dict(supertranslation=np.zeros(7, dtype=complex)),
        dict(supertranslation=np.array([0, 1.0, 0, 0.5], dtype=complex)),
        dict(supertranslation=st9),
        dict(supertranslation=real9),
        dict(time_translation=1),
        dict(time_translation=1.5),
        dict(time_translation=[1.0, 2.0]),
        dict(space_translation=[1.0, 2.0]),
        dict(space_translation=np.array([0.1, 0.2, -0.3])),
        dict(spacetime_translation=[1.0, 2.0, 3.0]),
        dict(spacetime_translation=np.array([0.5, 0.1, 0.2, 0.3]), time_translation=-1.0),
        dict(supertranslation=real9, space_translation=np.array([0.1, 0.2, -0.3]), time_translation=2.0),
        dict(n_theta=5),
        dict(n_phi=5),
        dict(space_translation=np.array([1.0, 0, 0]), n_theta=17),
        dict(space_translation=np.array([1.0, 0, 0]), n_phi=17),
        dict(supertranslation=real9, n_theta=18, n_phi=19),
        dict(frame_rotation=[0, 0, 0, 0]),
        dict(frame_rotation=[1.0, 2.0, 3.0]),
        dict(frame_rotation=[1, 2, 3, 4]),
        dict(boost_velocity=[1.0, 0.0, 0.0]),
        dict(boost_velocity=[0.1, 0.2]),
        dict(boost_velocity=[0.6, 0.6, 0.6]),
        dict(boost_velocity=np.array([0.1, 0.0, -0.2]), frame_rotation=np.array([1.0, 2, 3, 4]), space_translation=np.array([1.0, 0.0, 0.0])),
        dict(working_ell_max=3),
        dict(working_ell_max=20, output_ell_max=5),
        dict(unknown_keyword=3),
    ]

    def run(fn, *a, **kw):
        with warnings.catch_warnings(record=True) as ws:
            warnings.simplefilter("always")
            try:
                res = fn(*a, **kw)
            except Exception as e:  # noqa: BLE001
                return {"raises": type(e).__name__, "text": str(e), "warnings": [str(w.message) for w in ws]}
        return {"raises": None, "warnings": [str(w.message) for w in ws], "n_returned": len(res) if isinstance(res, tuple) else None}

    cases = []
    for kw in kw_cases:
        if not any(k in kw for k in ("working_ell_max", "output_ell_max")):
            cases.append({"fn": "wm_kwargs", "ell_max": 8, "kwargs": {k: enc(v) for k, v in kw.items()},
                          "outcome": run(wg.process_transformation_kwargs, 8, **{k: (v.copy() if isinstance(v, np.ndarray) else v) for k, v in kw.items()})})
        if not any(k in kw for k in ("n_theta", "n_phi")):
            cases.append({"fn": "abd_kwargs", "ell_max": 8, "kwargs": {k: enc(v) for k, v in kw.items()},
                          "outcome": run(tr_mod._process_transformation_kwargs, 8, **{k: (v.copy() if isinstance(v, np.ndarray) else v) for k, v in kw.items()})})
    # the transform itself: wrong argument type, missing psi companions, unused kwargs
    t = np.linspace(0.0, 10.0, 12)
    w_psi2 = _wm(t, synthetic.chirp_modes(t, 0, 3, 1), 0, 3, scri.psi2)
    w_psi3 = _wm(t, synthetic.chirp_modes(t, 1, 3, 2), 1, 3, scri.psi3)
    w_psi4_short = _wm(t[:10], synthetic.chirp_modes(t[:10], 2, 3, 3), 2, 3, scri.psi4)
    cases.append({"fn": "from_modes_type", "outcome": run(scri.WaveformGrid.from_modes, 3)})
    cases.append({"fn": "transform_type", "outcome": run(scri.WaveformGrid.transform, "not a waveform")})
    cases.append({"fn": "psi2_without_companions", "outcome": run(w_psi2.transform, space_translation=np.array([0.1, 0, 0]))})
    cases.append({"fn": "psi2_with_one_companion", "outcome": run(w_psi2.transform, space_translation=np.array([0.1, 0, 0]), psi3_modes=w_psi3)})
    cases.append({"fn": "psi3_with_short_companion", "outcome": run(w_psi3.transform, space_translation=np.array([0.1, 0, 0]), psi4_modes=w_psi4_short)})
    w_corot = scri.WaveformModes(t=t, data=synthetic.chirp_modes(t, 2, 3, 4), ell_min=2, ell_max=3, frameType=scri.Corotating, dataType=scri.h,
                                 r_is_scaled_out=True, m_is_scaled_out=True)
    cases.append({"fn": "non_inertial_frame", "outcome": run(w_corot.transform, space_translation=np.array([0.1, 0, 0]))})
    # rotate_decomposition_basis' own checks (scri/rotations.py:301-310)
    import quaternion as qmod

    w_h = _wm(t, synthetic.chirp_modes(t, 2, 3, 5), 2, 3, scri.h)
    five = qmod.as_quat_array(synthetic.rotor_series(t[:5], 3))
    cases.append({"fn": "rotate_wrong_length", "outcome": run(w_h.rotate_decomposition_basis, five)})
    cases.append({"fn": "rotate_two_dimensional", "outcome": run(w_h.rotate_decomposition_basis, np.array([list(five), list(five)]))})
    cases.append({"fn": "psi3_with_wrong_type_companion", "outcome": run(w_psi3.transform, space_translation=np.array([0.1, 0, 0]), psi4_modes=w_psi3)})
    with open(os.path.join(HERE, "g22_ref_error_behaviour.json"), "w") as f:
        json.dump({"source": "scri/waveform_grid.py:20-190,417-426,529-532,610-630; scri/asymptotic_bondi_data/transformations.py:8-97 (the reference's files)",
                   "cases": cases}, f, indent=1)


def g23():
    import scri.sample_waveforms as sw

    st = np.zeros(16, dtype=complex)
    st[[0, 2, 5, 7, 12]] = [0.3, 0.1, 0.02 - 0.01j, 0.02 + 0.01j, 0.005]
    out = dict(supertranslation=st, space_translation=np.array([0.2, -0.1, 0.4]))
    kws = (dict(), dict(s=-1, ell=3, m=2), dict(s=0, ell=2, m=-1, ell_max=5, t_0=-3.0, t_1=4.0, dt=0.25))
    for i, kw in enumerate(kws):
        for tag, w in (("rot", sw.single_mode_constant_rotation(omega=0.3 + 0.02j, **kw)), ("prop", sw.single_mode_proportional_to_time(beta=2.0 - 1j, **kw)),
                       ("super", sw.single_mode_proportional_to_time_supertranslated(supertranslation=st, **kw)),
                       ("space", sw.single_mode_proportional_to_time_supertranslated(space_translation=[0.2, -0.1, 0.4], **kw))):
            out[f"{tag}_{i}_t"], out[f"{tag}_{i}_data"] = np.array(w.t), np.array(w.data)
            out[f"{tag}_{i}_meta"] = np.array([w.ell_min, w.ell_max, int(w.dataType), int(w.frameType)])
    c = sw.constant_waveform()
    out["constant_row"], out["constant_shape"] = np.array(c.data[0]), np.array(c.data.shape)
    np.savez_compressed(os.path.join(HERE, "g23_ref_sample_waveforms.npz"), source="scri/sample_waveforms.py:60-381 (the reference's file, stand-ins underneath)", **out)


def g24():
    u, raw, L, abd = _g19_abd()
    tn = np.concatenate([np.linspace(-20.0, 30.0, 33), u[7:40:11]])
    tn.sort()
    arr = lambda m: np.asarray(m).view(np.ndarray)
    fields = lambda a: np.array([arr(getattr(a, f)) for f in ("psi0", "psi1", "psi2", "psi3", "psi4", "sigma")])
    rng = np.random.default_rng(241)
    un = np.sort(rng.uniform(-30.0, 40.0, 60)) + np.arange(60) * 1e-3  # a non-uniform axis for the waveform
    wdata = synthetic.chirp_modes(un, 2, 4, 7) * (1 + 0.01 * un[:, None])
    w = _wm(un, wdata, 2, 4, scri.h)
    out = dict(u=u, raw=raw, ell_max=L, new_times=tn, w_t=un, w_data=wdata)
    wi = w.interpolate(tn)
    out["w_interp_t"], out["w_interp_data"] = np.array(wi.t), np.array(wi.data)
    for name in ("data_dot", "data_ddot", "data_int", "data_iint"):
        out["w_" + name] = np.array(getattr(w, name))
    out["w_norm"], out["w_norm_sqrt"] = np.array(w.norm()), np.array(w.norm(take_sqrt=True))
    sl = w[:, 3:5]
    out["w_ell_slice_data"], out["w_ell_slice_meta"] = np.array(sl.data), np.array([sl.ell_min, sl.ell_max, sl.n_times])
    st = w[5:20]
    out["w_t_slice_t"], out["w_t_slice_data"] = np.array(st.t), np.array(st.data)
    ai = abd.interpolate(tn)
    out["abd_interp_u"], out["abd_interp_raw"] = np.array(ai.t), fields(ai)
    ak = abd[5:20]
    out["abd_slice_u"], out["abd_slice_raw"] = np.array(ak.t), fields(ak)
    h = abd.h
    out["abd_h_t"], out["abd_h_data"], out["abd_h_meta"] = np.array(h.t), np.array(h.data), np.array([h.ell_min, h.ell_max, int(h.dataType), int(h.frameType)])
    np.savez_compressed(os.path.join(HERE, "g24_ref_containers.npz"), source="scri/waveform_base.py, scri/waveform_modes.py, scri/asymptotic_bondi_data/__init__.py (the reference's files)", **out)


def g25():
    u, raw, L, abd = _g19_abd()
    arr = lambda m: np.asarray(m).view(np.ndarray)
    out = dict(u=u, raw=raw, ell_max=L)
    for name in ("Bondi-Sachs", "Moreschi", "Geroch", "GW"):
        for tag, kw in (("plain", {}), ("integrated", dict(integrated=True)), ("wide", dict(working_ell_max=6)), ("integrated_wide", dict(integrated=True, working_ell_max=6))):
            r = abd.supermomentum(name, **kw)
            out[f"{name}_{tag}"] = arr(r)
            out[f"{name}_{tag}_meta"] = np.array([r.spin_weight, r.ell_min, r.ell_max])
    for name in ("bondi_rest_mass", "bondi_four_momentum", "bondi_angular_momentum", "bondi_boost_charge", "bondi_CoM_charge", "bondi_dimensionless_spin",
                 "CWWY_angular_momentum"):
        if hasattr(abd, name):
            out[name] = np.asarray(getattr(abd, name)())
    try:
        abd.supermomentum("Bondi")
        out["unknown_name_error"] = np.array("")
    except ValueError as e:
        out["unknown_name_error"] = np.array(str(e))
    np.savez_compressed(os.path.join(HERE, "g25_ref_supermomenta.npz"), source="scri/asymptotic_bondi_data/bms_charges.py:14-286 (the reference's file, stand-ins underneath)", **out)


def g26():
    L, n = 4, 41
    u = np.linspace(-4.0, 6.0, n)
    rng = np.random.default_rng(260)

    def modes(s_, scale=0.05):
        a = scale * (rng.normal(size=(L + 1) ** 2) + 1j * rng.normal(size=(L + 1) ** 2))
        a[: s_ * s_] = 0
        return a

    sigma0, sigmadot0, sigmaddot0 = modes(2), modes(2, 0.01), modes(2, 0.002)
    psi2, psi1, psi0 = modes(0), modes(1), modes(2)
    psi2[0] -= np.sqrt(4 * np.pi)
    arr = lambda m: np.asarray(m).view(np.ndarray)
    fields = lambda a: np.array([arr(getattr(a, f)) for f in ("psi0", "psi1", "psi2", "psi3", "psi4", "sigma")])
    out = dict(u=u, ell_max=L, sigma0=sigma0, sigmadot0=sigmadot0, sigmaddot0=sigmaddot0, psi2=psi2, psi1=psi1, psi0=psi0)
    sigma_of_u = sigma0[None, :] + u[:, None] * sigmadot0[None, :] + 0.5 * u[:, None] ** 2 * sigmaddot0[None, :]
    out["sigma_of_u"] = sigma_of_u
    for tag, abd in (("exact", scri.AsymptoticBondiData.from_initial_values(u, L, sigma0, sigmadot0, sigmaddot0, psi2, psi1, psi0)),
                     ("numeric", scri.AsymptoticBondiData.from_initial_values(u, L, sigma_of_u, 0.0, 0.0, psi2, psi1, psi0))):
        out[f"{tag}_raw"] = fields(abd)
        out[f"{tag}_violation_norms"] = np.array(abd.bondi_violation_norms)
        cons = abd.bondi_constraints()
        out[f"{tag}_lhs"], out[f"{tag}_rhs"] = np.array([arr(c[0]) for c in cons[:5]]), np.array([arr(c[1]) for c in cons[:5]])
        out[f"{tag}_mass_aspect_lhs"], out[f"{tag}_mass_aspect_rhs"] = arr(cons[5][0]), arr(cons[5][1])
    np.savez_compressed(os.path.join(HERE, "g26_ref_initial_values.npz"),
                        source="scri/asymptotic_bondi_data/from_initial_values.py, constraints.py (the reference's files, stand-ins underneath)", **out)


def g27():
    import quaternion
    import scri.asymptotic_bondi_data.transformations as tr

    out = {}
    # the rotor grid and the conformal factors on it (transformations.py:100-196), a generic frame and boost on a non-square grid
    fr = np.array([0.9, 0.1, -0.3, 0.2])
    fr /= np.linalg.norm(fr)
    v = np.array([0.11, -0.07, 0.2])
    for tag, (q, vel) in (("generic", (fr, v)), ("boost_only", (np.array([1.0, 0, 0, 0]), np.array([0.0, 0.0, 0.3]))), ("rotation_only", (fr, np.zeros(3)))):
        R = tr.boosted_grid(quaternion.quaternion(*q), vel, 9, 11)
        k, ethk_over_k, one_over_k, one_over_k_cubed = tr.conformal_factors(vel, R)
        out[f"{tag}_q"], out[f"{tag}_v"] = q, vel
        out[f"{tag}_rotors"] = quaternion.as_float_array(R)
        out[f"{tag}_k"], out[f"{tag}_ethk_over_k"] = np.asarray(k).view(np.ndarray), np.asarray(ethk_over_k).view(np.ndarray)
        out[f"{tag}_one_over_k"], out[f"{tag}_one_over_k_cubed"] = np.asarray(one_over_k).view(np.ndarray), np.asarray(one_over_k_cubed).view(np.ndarray)
    # SI_units and compare (scri/waveform_base.py:553-687) on a chirp and a resampled, slightly different copy of it
    t = np.linspace(0, 50, 200)
    data = synthetic.chirp_modes(t, 2, 5, 7)
    for dt_name in ("h", "psi4", "news"):
        w = _wm(t, data, 2, 5, getattr(scri, dt_name))
        w = w.SI_units(60.0, 200.0)
        out[f"SI_{dt_name}_t"], out[f"SI_{dt_name}_data"] = np.array(w.t), np.array(w.data)
        out[f"SI_{dt_name}_flags"] = np.array([w.r_is_scaled_out, w.m_is_scaled_out])
    a = _wm(t, data, 2, 5, scri.h)
    t_b = np.sort(np.concatenate([t[::2] + 0.01, [13.3331, 27.7]]))
    b = _wm(t_b, 1.01 * synthetic.chirp_modes(t_b, 2, 5, 7), 2, 5, scri.h)
    c = b.compare(a)
    out["compare_t_a"], out["compare_a"], out["compare_t_b"], out["compare_b"] = t, data, t_b, np.array(b.data)
    out["compare_t"], out["compare_data"], out["compare_frame_size"] = np.array(c.t), np.array(c.data), np.array(np.size(c.frame))
    c2 = b.compare(a, min_time_step=0.3, min_time=5.0)
    out["compare2_t"], out["compare2_data"] = np.array(c2.t), np.array(c2.data)
    np.savez_compressed(os.path.join(HERE, "g27_ref_grids_and_containers.npz"),
                        source="scri/asymptotic_bondi_data/transformations.py:100-196, scri/waveform_base.py:553-687 (the reference's files, stand-ins underneath)", **out)


def g28():
    # the same two transformations as g8 / g9 in the relativistic regime and on a non-uniform time axis: |v| = 0.35 and 0.3, a
    # supertranslation of order one (l <= 4), higher l -- second-order boost terms, k^w weights and the psi mixing are of order one here
    rng = np.random.default_rng(280)
    n, L = 400, 10
    t = np.sort(rng.uniform(-30.0, 90.0, n)) + np.arange(n) * 1e-3
    kw = dict(supertranslation=_real_supertranslation(4, 281, 0.3), frame_rotation=np.array([-0.3, 0.8, 0.4, -0.2]) / np.linalg.norm([-0.3, 0.8, 0.4, -0.2]),
              boost_velocity=np.array([0.21, -0.2, 0.19]))
    out = dict(wm_t=t, wm_ell_max=L, **{"wm_" + k: v for k, v in kw.items()})
    for name, dt, seed in (("h", scri.h, 282), ("news", scri.news, 283), ("psi4", scri.psi4, 284)):
        data = synthetic.chirp_modes(t, 2, L, seed) * (1 + 0.004 * t[:, None])
        w = _wm(t, data, 2, L, dt).transform(**kw)
        out[f"{name}_t_out"], out[f"{name}_out"] = w.t, w.data[::4]  # (inputs: regenerated by the test from the same seeds; every 4th output row)
    d1 = synthetic.chirp_modes(t, 1, 8, 285)
    comp = {f"psi{k}_modes": _wm(t, synthetic.chirp_modes(t, s_, 8 - k + 2, 285 + k), s_, 8 - k + 2, getattr(scri, f"psi{k}")) for k, s_ in ((2, 0), (3, 1), (4, 2))}
    w = _wm(t, d1, 1, 8, scri.psi1).transform(**comp, **kw)
    out.update(psi1_t_out=w.t, psi1_out=w.data[::4], psi1_companion_ell=np.array([[v.ell_min, v.ell_max] for v in comp.values()]))
    # AsymptoticBondiData, six fields, l <= 6
    n2, L2 = 150, 6
    u = np.sort(rng.uniform(-20.0, 60.0, n2)) + np.arange(n2) * 1e-3
    abd = scri.AsymptoticBondiData(u, L2)
    raw = np.zeros((6, n2, (L2 + 1) ** 2), dtype=complex)
    for f, s_ in enumerate(synthetic.ABD_SPINS):
        raw[f] = synthetic.chirp_modes(u, 0, L2, 290 + f) * (1 + 0.01 * u[:, None])
        raw[f, :, : s_ * s_] = 0
    abd.psi0, abd.psi1, abd.psi2, abd.psi3, abd.psi4, abd.sigma = raw
    kw2 = dict(supertranslation=_real_supertranslation(3, 297, 0.25), frame_rotation=np.array([0.2, -0.5, 0.1, 0.9]) / np.linalg.norm([0.2, -0.5, 0.1, 0.9]),
               boost_velocity=np.array([-0.17, 0.16, 0.19]))
    new = abd.transform(**kw2)
    arr = lambda m: np.asarray(m).view(np.ndarray)
    out.update(abd_u=u, abd_ell_max=L2, **{"abd_" + k: v for k, v in kw2.items()}, abd_u_out=np.array(new.t),
               abd_raw_out=np.array([arr(getattr(new, f)) for f in ("psi0", "psi1", "psi2", "psi3", "psi4", "sigma")]))
    np.savez_compressed(os.path.join(HERE, "g28_ref_relativistic_transforms.npz"),
                        source="scri/waveform_grid.py:331-630, scri/asymptotic_bondi_data/transformations.py:199-431 (the reference's files, stand-ins underneath)", **out)


if __name__ == "__main__":
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        every = (g10, g8, g9, g11, g12, g13, g14, g15, g16, g17, g18, g19, g20, g21, g22, g23, g24, g25, g26, g27, g28)
        only = [f for f in every if "--" + f.__name__ in sys.argv]
        for f in only or every:
            f()
            print("wrote", f.__name__)
