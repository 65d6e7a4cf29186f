"""Mapping AsymptoticBondiData to the super rest frame (scri/asymptotic_bondi_data/map_to_superrest_frame.py:76-1035),
the step after the transformation path.  The iteration fixes, at retarded time t_0, the
supertranslation (Moreschi supermomentum -> its rest value), the rotation (spin charge -> z) and the translation + boost
(centre-of-mass charge -> 0), re-transforming the data after every step.

Every pass over the data runs on the GPU through the C ABI: `abd.transform` (the hot path itself), the charges
(`bms_grid_multiply`, `bms_spline_derivative`), grid <-> mode maps (`bms_salm2map`, `bms_map2salm`) and the per-pixel time
interpolation (`bms_spline_derivative`).  What is left here is the control loop and the algebra of the BMS group
(`scri_amd.bms_transformations`).

Built: the target-free iteration, a target Moreschi supermomentum, a target strain (rotation onto its angular velocity)
and the "time_phase" step (scri_amd.alignment.align2d, this package's restatement of sxs.waveforms.alignment.align2d).
"""
import math

import numpy as np

from . import engine, quaternions
from .bms_charges import charge_vector_from_aspect
from .bms_transformations import BMSTransformation
from .mode_algebra import LM_index, vector_as_ell_1_modes, constant_as_ell_0_mode

NORMAL = ["supertranslation", "frame_rotation", "boost_velocity"]


def _ell_factors(ell_max, f):
    return np.concatenate([np.full(2 * ell + 1, f(ell)) for ell in range(ell_max + 1)])


def D_operator(h, ell_max):
    """eth^2 ethbar^2 on spin-0 modes: x (l+2)(l+1)l(l-1)/4 (l >= 2), 0 below (map_to_superrest_frame.py:76-92)"""
    return np.asarray(h) * _ell_factors(ell_max, lambda l: 0.0 if l < 2 else (l + 2) * (l + 1) * l * (l - 1) / 4.0)


def D_inverse(h, ell_max):
    """inverse of D_operator on l >= 2, 0 below (map_to_superrest_frame.py:95-105)"""
    return np.asarray(h) * _ell_factors(ell_max, lambda l: 0.0 if l < 2 else 4.0 / ((l + 2) * (l + 1) * l * (l - 1)))


# the reference's names for the two operators (map_to_superrest_frame.py:76,95; used by bms_charges.py:111)
𝔇 = D_operator
𝔇inverse = D_inverse


def MT_to_WM(h_mts, sxs_version=False, dataType=None):
    """A ModesTimeSeries as a WaveformModes (map_to_superrest_frame.py:18-52): the modes from l = |s| on, Inertial frame, r and M
    scaled out, data type `dataType` (default h).  The sxs flavour needs the `sxs` package, which this image lacks."""
    from . import WaveformModes, Inertial, h

    if sxs_version:
        raise NotImplementedError("sxs.WaveformModes objects need the sxs package")
    s = abs(h_mts.s)
    return WaveformModes(t=np.asarray(h_mts.t), data=np.array(h_mts)[:, LM_index(s, -s, 0):], ell_min=s, ell_max=h_mts.ell_max, frameType=Inertial,
                         dataType=h if dataType is None else dataType, r_is_scaled_out=True, m_is_scaled_out=True, ctx=getattr(h_mts, "_ctx", None))


def WM_to_MT(h_wm):
    """A WaveformModes as a ModesTimeSeries with the same l range (map_to_superrest_frame.py:55-73), products truncated at max"""
    from .modes_time_series import ModesTimeSeries

    return ModesTimeSeries(np.array(h_wm.data), h_wm.t, spin_weight=h_wm.spin_weight, ell_min=h_wm.ell_min, ell_max=h_wm.ell_max,
                           multiplication_truncator=max)


def time_translation(abd, t_0=0):
    """A copy of `abd` on the time axis t - t_0 (map_to_superrest_frame.py:666-684: the fields carry their own time axis, so
    shifting `abd.t` of a plain copy would leave theirs behind)"""
    out = abd.copy()
    out.t = out.t - t_0
    return out


def rotation(abd, phi=0):
    """`abd` with the physical system turned by -phi about z (map_to_superrest_frame.py:687-717): every field goes through
    WaveformModes.rotate_physical_system -- for a rotation the detour of the reference through h = 2 sigma-bar and the
    Newman-Penrose factors drops out (conjugation of the field commutes with turning it), so the six fields are rotated as they
    are, by one constant-rotor kernel launch each; much cheaper than abd.transform()."""
    from . import WaveformModes, Inertial, psi2
    from .rotations import rotate_physical_system

    q = np.array([math.cos(-phi / 2.0), 0.0, 0.0, math.sin(-phi / 2.0)])  # quaternion.from_rotation_vector(-phi z)
    out = abd.copy()
    for name in ("psi0", "psi1", "psi2", "psi3", "psi4", "sigma"):
        field = getattr(abd, name)
        w = WaveformModes(t=abd.t, data=np.array(field.ndarray), ell_min=0, ell_max=abd.ell_max, frameType=Inertial, dataType=psi2,
                          r_is_scaled_out=True, m_is_scaled_out=True, ctx=getattr(abd, "_ctx", None))
        rotate_physical_system(w, q)
        setattr(out, name, w.data)
    return out


def _to_grid(modes, ell_max, ctx=None):
    n = 2 * ell_max + 1
    return engine.salm2map(modes, 0, ell_max, n, n, ctx=ctx)


def _to_modes(grid, ell_max, ctx=None):
    return engine.map2salm(np.asarray(grid, dtype=complex), 0, ell_max, ctx=ctx)


def compute_bondi_rest_mass_and_conformal_factor(PsiM, ell_max, ctx=None):
    """Bondi rest mass M and conformal factor K = M / (P^0 - P.r) on the grid, from the l <= 1 part of the Moreschi
    supermomentum (Eqs. (14), (15) of doi:10.1063/1.532646; map_to_superrest_frame.py:108-152)."""
    PsiM = np.asarray(PsiM)
    P = -charge_vector_from_aspect(PsiM)
    unit_grids = []
    for L, M in [(0, 0), (1, -1), (1, 0), (1, +1)]:
        m = np.zeros((ell_max + 1) ** 2)
        m[LM_index(L, M, 0)] = 1
        unit_grids.append(_to_grid(m, ell_max, ctx))
    r_vector = 4 * np.pi * charge_vector_from_aspect(np.array(unit_grids).transpose()).transpose()
    if PsiM.ndim > 1:
        M_Grid = np.sqrt(P[:, 0] ** 2 - (P[:, 1] ** 2 + P[:, 2] ** 2 + P[:, 3] ** 2))
        K_Grid = M_Grid[:, None, None] / (
            np.tensordot(P[:, 0], r_vector[0], axes=0)
            - (np.tensordot(P[:, 1], r_vector[1], axes=0) + np.tensordot(P[:, 2], r_vector[2], axes=0) + np.tensordot(P[:, 3], r_vector[3], axes=0))
        )
    else:
        M_Grid = np.sqrt(P[0] ** 2 - (P[1] ** 2 + P[2] ** 2 + P[3] ** 2))
        K_Grid = M_Grid / (P[0] * r_vector[0] - (P[1] * r_vector[1] + P[2] * r_vector[2] + P[3] * r_vector[3]))
    return M_Grid, K_Grid


def _interpolate_each_pixel(t, series_grid, times_grid, ctx=None):
    """value at pixel (i, j) of the cubic spline in time through series_grid[:, i, j], evaluated at times_grid[i, j]
    (the per-pixel CubicSpline loops of map_to_superrest_frame.py:176-181): all columns at all requested times on the GPU,
    then the diagonal."""
    n, nt, nphi = series_grid.shape
    cols = np.ascontiguousarray(series_grid.reshape(n, nt * nphi), dtype=complex)
    vals = engine.spline_derivative(t, cols, np.asarray(times_grid, dtype=float).reshape(-1), 0, ctx=ctx)
    return np.real(np.diagonal(vals)).reshape(nt, nphi)


SPLINE_REACH = 64  # knots: a cubic spline forgets its far data like 0.268^n (3e-37 at 64), far below one rounding


def _rows_near(t, times, reach=SPLINE_REACH):
    """Row range of `t` that determines, to rounding, a cubic spline through it at the requested `times`.  The reference
    fits every pixel's spline through the whole series to evaluate it at one time near u = 0
    (map_to_superrest_frame.py:176-181); the knots further than `reach` from the evaluation points contribute < 1e-36."""
    times = np.asarray(times, dtype=float)
    lo = int(np.searchsorted(t, times.min(), side="right")) - 1 - reach
    hi = int(np.searchsorted(t, times.max(), side="left")) + 1 + reach
    return max(lo, 0), min(hi, len(t))


def compute_Moreschi_supermomentum(PsiM, alpha, ell_max, ctx=None):
    """Moreschi supermomentum in the frame supertranslated by alpha (a real grid function): Eq. (9) of
    doi:10.1063/1.532646 (map_to_superrest_frame.py:155-197)."""
    t = PsiM.t
    lo, hi = _rows_near(t, alpha)
    # (a device-resident series hands over the rows around u = 0 only)
    data = PsiM.rows(lo, hi) if hasattr(PsiM, "rows") else np.asarray(PsiM.ndarray if hasattr(PsiM, "ndarray") else PsiM)[lo:hi]
    t = t[lo:hi]
    M_Grid, K_Grid = compute_bondi_rest_mass_and_conformal_factor(data, ell_max, ctx)
    PsiM_Grid = _to_grid(data, ell_max, ctx).real
    PsiM_at_alpha = _interpolate_each_pixel(t, PsiM_Grid, alpha, ctx)
    K_at_alpha = _interpolate_each_pixel(t, K_Grid, alpha, ctx)
    D_alpha_Grid = _to_grid(D_operator(_to_modes(alpha, ell_max, ctx), ell_max), ell_max, ctx)
    return _to_modes((PsiM_at_alpha - D_alpha_Grid) / K_at_alpha**3, ell_max, ctx)


def compute_alpha_perturbation(PsiM, M_Grid, K_Grid, ell_max, ctx=None):
    """Supertranslation that removes the l >= 2 supermomentum to first order: Eq. (10) of doi:10.1063/1.532646
    (map_to_superrest_frame.py:200-224)."""
    PsiM_Grid = _to_grid(np.asarray(PsiM), ell_max, ctx)
    alpha = D_inverse(_to_modes(PsiM_Grid + M_Grid * K_Grid**3, ell_max, ctx), ell_max)
    return _to_grid(alpha, ell_max, ctx).real


def supertranslation_to_map_to_superrest_frame(abd, target_PsiM=None, N_itr_max=10, rel_err_tol=1e-12, ell_max=12, print_conv=False):
    """Iterative solve for the supertranslation that maps the Moreschi supermomentum at u = 0 to zero (or to the target's)
    (map_to_superrest_frame.py:227-319).  target_PsiM: an object with .t, .data [n, modes from l = 0], .ell_max."""
    ctx = getattr(abd, "_ctx", None)
    n = 2 * ell_max + 1
    alpha_Grid = np.zeros((n, n))
    best_alpha_Grid = np.zeros((n, n))
    PsiM = abd.supermomentum("Moreschi")

    def target_at(times_grid):
        lo, hi = _rows_near(target_PsiM.t, times_grid)
        tg = _to_grid(np.asarray(target_PsiM.data)[lo:hi], target_PsiM.ell_max, ctx).real
        return _interpolate_each_pixel(target_PsiM.t[lo:hi], tg, times_grid, ctx)

    itr, rel_err, rel_errs = 0, np.inf, [np.inf]
    PsiM_interp = M_Grid = K_Grid = None
    while itr < N_itr_max and not rel_err < rel_err_tol:
        prev_alpha_Grid = alpha_Grid.copy()
        if itr == 0:
            PsiM_interp = compute_Moreschi_supermomentum(PsiM, alpha_Grid, ell_max, ctx)
            M_Grid, K_Grid = compute_bondi_rest_mass_and_conformal_factor(PsiM_interp, ell_max, ctx)
            if target_PsiM is not None:
                M_Grid = -target_at(prev_alpha_Grid)
        alpha_Grid = alpha_Grid + compute_alpha_perturbation(PsiM_interp, M_Grid, K_Grid, ell_max, ctx)
        PsiM_interp = compute_Moreschi_supermomentum(PsiM, alpha_Grid, ell_max, ctx)
        M_Grid, K_Grid = compute_bondi_rest_mass_and_conformal_factor(PsiM_interp, ell_max, ctx)
        if target_PsiM is not None:
            target_grid = target_at(prev_alpha_Grid)
            M_Grid = -target_grid
            target_interp = _to_modes(target_grid, ell_max, ctx)
            rel_err = np.linalg.norm(PsiM_interp[4:] - target_interp[4:]) / np.linalg.norm(target_interp[4:])
        else:
            rel_err = np.linalg.norm(PsiM_interp[4:])
        if rel_err < min(rel_errs):
            best_alpha_Grid = alpha_Grid.copy()
        rel_errs.append(rel_err)
        itr += 1
    if print_conv:
        if not itr < N_itr_max:
            print(f"supertranslation: maximum number of iterations reached; the min error was {min(rel_errs)}.")
        else:
            print(f"supertranslation: tolerance achieved in {itr} iterations!")
    supertranslation = _to_modes(best_alpha_Grid, ell_max, ctx)
    supertranslation[0:4] = 0
    return BMSTransformation(supertranslation=supertranslation, ell_max=ell_max, ctx=ctx), rel_errs


def transformation_from_CoM_charge(G, t, Gfun=None, Gparams0=None, Gargs=None, ctx=None):
    """Space translation and boost velocity from a fit of the centre-of-mass charge (Eq. (18) of PhysRevD.104.024051;
    map_to_superrest_frame.py:322-366): by default the linear model G(t) = -v t + x0, solved directly; with `Gfun(Gparams,
    time, *Gargs) -> [n, 3]` a nonlinear least-squares fit whose first six parameters are (v, x0)."""
    if Gfun is None and Gargs is None:
        A = np.stack([-np.asarray(t, dtype=float), np.ones(len(t))], axis=1)
        (v, x0), *_ = np.linalg.lstsq(A, np.asarray(G, dtype=float), rcond=None)
    else:
        from scipy.optimize import least_squares

        if Gfun is None:
            Gfun = lambda Gparams, time, *args: -time[:, None] @ Gparams[:3][None, :] + Gparams[3:6][None, :]  # noqa: E731
        Gparams0 = np.zeros(6) if Gparams0 is None else np.asarray(Gparams0)
        if not (Gparams0.ndim == 1 and Gparams0.shape[0] >= 6):
            raise ValueError(
                "The shape of Gparams0 doesn't match with the expected input for Gfun. Refer to the documentation of "
                "com_transformation_to_map_to_superrest_frame for the signature of Gfun."
            )
        G = np.asarray(G, dtype=float)
        fit = least_squares(lambda Gparams, time, *args: (G - Gfun(Gparams, time, *args)).ravel(), Gparams0,
                            args=(np.asarray(t, dtype=float), *(Gargs or ())), method="trf")
        v, x0 = fit.x[0:3], fit.x[3:6]
    return BMSTransformation(
        supertranslation=-np.insert(vector_as_ell_1_modes(x0), 0, 0),
        boost_velocity=-v,
        order=["supertranslation", "boost_velocity", "frame_rotation"],
        ctx=ctx,
    )


def _time_average(values, t, ctx=None):
    """integral of a scalar series over its time range (cubic-spline antiderivative) / the length of the range"""
    f = np.asarray(values, dtype=float)[:, None] + 0j
    total = engine.spline_derivative(t, f, np.array([t[-1]]), -1, ctx=ctx)[0, 0].real
    return total / (t[-1] - t[0])


def _transform(abd, B):
    return abd.transform(
        supertranslation=B.supertranslation, frame_rotation=B.frame_rotation.components, boost_velocity=B.boost_velocity
    )


def com_transformation_to_map_to_superrest_frame(abd, N_itr_max=10, rel_err_tol=1e-12, print_conv=False, Gfun=None, Gparams0=None,
                                                 Gargsfun=None):
    """Iterative solve for the translation and boost that remove the centre-of-mass charge
    (map_to_superrest_frame.py:369-465).  Gfun / Gparams0: model and first guess of the fit (see
    transformation_from_CoM_charge); Gargsfun: callables of the abd object whose values are passed on to Gfun."""
    ctx = getattr(abd, "_ctx", None)
    CoM = BMSTransformation(ctx=ctx)
    best = BMSTransformation(ctx=ctx)
    itr, rel_err, rel_errs = 0, np.inf, [np.inf]
    abd_prime = G_prime = Gargs = None
    while itr < N_itr_max and not rel_err < rel_err_tol:
        if itr == 0:
            abd_prime = abd.copy()
            G_prime = abd_prime.bondi_CoM_charge() / abd_prime.bondi_four_momentum()[:, 0, None]
            Gargs = [func(abd_prime) for func in Gargsfun] if Gargsfun else None
        new = transformation_from_CoM_charge(G_prime, abd_prime.t, Gfun=Gfun, Gparams0=Gparams0, Gargs=Gargs, ctx=ctx)
        CoM = (new * CoM).reorder(NORMAL)
        CoM.supertranslation[4:] *= 0  # keep only the translation ...
        CoM.frame_rotation = type(CoM.frame_rotation)(np.array([1.0, 0, 0, 0]))  # ... and the boost
        abd_prime = _transform(abd, CoM)
        G_prime = abd_prime.bondi_CoM_charge() / abd_prime.bondi_four_momentum()[:, 0, None]
        Gargs = [func(abd_prime) for func in Gargsfun] if Gargsfun else None
        rel_err = _time_average(np.linalg.norm(G_prime, axis=-1), abd_prime.t, ctx)
        if rel_err < min(rel_errs):
            best = CoM.copy()
        rel_errs.append(rel_err)
        itr += 1
    if print_conv:
        if not itr < N_itr_max:
            print(f"CoM: maximum number of iterations reached; the min error was {min(rel_errs)}.")
        else:
            print(f"CoM: tolerance achieved in {itr} iterations!")
    return best, rel_errs


def _qnormalized(q):
    q = np.asarray(q, dtype=float)
    return q / np.linalg.norm(q)


def _rotate_vector(q, v):
    """vector part of q v q^-1"""
    return quaternions.multiply(quaternions.multiply(q, np.concatenate([[0.0], v])), quaternions.conjugate(q))[1:]


def _about_z(angle):
    return np.array([math.cos(angle / 2), 0.0, 0.0, math.sin(angle / 2)])


def rotation_from_spin_charge(chi, t, fix_xz_plane=False, fix_yz_plane=False, ctx=None):
    """Rotor q with q z q^-1 = chi(t ~ 0) / |chi| (map_to_superrest_frame.py:468-507)."""
    chi_f = _qnormalized(np.concatenate([[0.0], chi[np.argmin(abs(np.asarray(t)))]]))
    q = _qnormalized(np.array([1.0, 0, 0, 0]) - quaternions.multiply(chi_f, np.array([0.0, 0, 0, 1])))
    if fix_xz_plane:
        y_rot = _rotate_vector(q, np.array([0.0, 1, 0]))
        q = quaternions.multiply(q, _about_z(np.angle(y_rot[0] + 1j * y_rot[1]) - np.pi / 2))
        if _rotate_vector(q, np.array([1.0, 0, 0]))[0] < 0:
            q = quaternions.multiply(q, _about_z(np.pi))
    elif fix_yz_plane:
        x_rot = _rotate_vector(q, np.array([1.0, 0, 0]))
        q = quaternions.multiply(q, _about_z(np.angle(x_rot[0] + 1j * x_rot[1])))
        if _rotate_vector(q, np.array([0.0, 1, 0]))[1] < 0:
            q = quaternions.multiply(q, _about_z(np.pi))
    return BMSTransformation(frame_rotation=q, ctx=ctx)


def _unit_spin(abd):
    chi = abd.bondi_dimensionless_spin()
    return chi / np.linalg.norm(chi, axis=-1)[:, None]


def rotation_from_vectors(vector, target_vector, t=None, ctx=None):
    """Frame rotation that best aligns `vector` with `target_vector` over the time interval (map_to_superrest_frame.py:510-526)"""
    q = quaternions.optimal_alignment_in_Euclidean_metric(vector, target_vector, t=t)
    return BMSTransformation(frame_rotation=quaternions.conjugate(q), ctx=ctx)


def _news_angular_velocity_direction(abd):
    """unit angular velocity of the news 2 d(sigma-bar)/du of an abd object, as a WaveformModes quantity on the GPU"""
    from . import Inertial, WaveformModes, hdot

    news = (2.0 * abd.sigma.bar).dot
    w = WaveformModes(t=abd.t.copy(), data=np.asarray(news)[:, 4:], ell_min=2, ell_max=abd.ell_max, frameType=Inertial, dataType=hdot,
                      r_is_scaled_out=True, m_is_scaled_out=True, ctx=getattr(abd, "_ctx", None))
    omega = w.angular_velocity()
    return omega / np.linalg.norm(omega, axis=-1)[:, None]


def _target_omega_spline(target_strain):
    """Cubic spline through the unit angular velocity of the target.  As in the reference (:575-590, :724-741) the target
    strain is differentiated TWICE here (`data_dot`, then `.dot` of the series built from it), the abd's shear once."""
    from scipy.interpolate import CubicSpline

    from . import Inertial, WaveformModes, hdot

    ctx = getattr(target_strain, "_ctx", None)
    first = engine.spline_derivative(target_strain.t, target_strain.data, target_strain.t, 1, ctx=ctx)
    second = engine.spline_derivative(target_strain.t, first, target_strain.t, 1, ctx=ctx)
    w = WaveformModes(t=target_strain.t.copy(), data=second, ell_min=target_strain.ell_min, ell_max=target_strain.ell_max,
                      frameType=Inertial, dataType=hdot, r_is_scaled_out=True, m_is_scaled_out=True, ctx=ctx)
    omega = w.angular_velocity()
    return CubicSpline(target_strain.t, omega / np.linalg.norm(omega, axis=-1)[:, None])


def rotation_to_map_to_superrest_frame(abd, target_strain=None, N_itr_max=10, rel_err_tol=1e-12, fix_xz_plane=False,
                                       fix_yz_plane=False, print_conv=False):
    """Iterative solve for the rotation that aligns the spin charge with the z axis -- or, with a target strain, the
    angular velocity of the news with the target's over the whole window (map_to_superrest_frame.py:529-663)."""
    ctx = getattr(abd, "_ctx", None)
    rot = BMSTransformation(ctx=ctx)
    best = BMSTransformation(ctx=ctx)
    itr, rel_err, rel_errs = 0, np.inf, [np.inf]
    if target_strain is not None:
        target = _target_omega_spline(target_strain)
        omega = None
        while itr < N_itr_max and not rel_err < rel_err_tol:
            if itr == 0:
                omega = _news_angular_velocity_direction(abd)
            rot = (rotation_from_vectors(omega, target(abd.t), abd.t, ctx) * rot).reorder(NORMAL)
            rot.supertranslation *= 0
            rot.boost_velocity *= 0
            abd_prime = abd.transform(frame_rotation=rot.frame_rotation.components)
            omega = _news_angular_velocity_direction(abd_prime)
            rel_err = _time_average(np.linalg.norm(omega - target(abd_prime.t), axis=-1), abd_prime.t, ctx)
            if rel_err < min(rel_errs):
                best = rot.copy()
            rel_errs.append(rel_err)
            itr += 1
        if print_conv:
            print(f"rotation: {'maximum number of iterations reached; the min error was ' + str(min(rel_errs)) if not itr < N_itr_max else 'tolerance achieved in ' + str(itr) + ' iterations!'}")
        return best, rel_errs
    abd_prime = chi_prime = None
    while itr < N_itr_max and not rel_err < rel_err_tol:
        if itr == 0:
            abd_prime = abd.copy()
            chi_prime = _unit_spin(abd_prime)
        rot = (rotation_from_spin_charge(chi_prime, abd_prime.t, fix_xz_plane, fix_yz_plane, ctx) * rot).reorder(NORMAL)
        rot.supertranslation *= 0
        rot.boost_velocity *= 0
        abd_prime = abd.transform(frame_rotation=rot.frame_rotation.components)
        chi_prime = _unit_spin(abd_prime)
        rel_err = _time_average(np.linalg.norm(chi_prime - np.array([0.0, 0, 1])[None, :], axis=-1), abd_prime.t, ctx)
        if rel_err < min(rel_errs):
            best = rot.copy()
        rel_errs.append(rel_err)
        itr += 1
    if print_conv:
        if not itr < N_itr_max:
            print(f"rotation: maximum number of iterations reached; the min error was {min(rel_errs)}.")
        else:
            print(f"rotation: tolerance achieved in {itr} iterations!")
    return best, rel_errs


def rel_err_for_abd_in_superrest(abd, target_PsiM, target_strain):
    """(CoM, rotation, supermomentum) residuals of an abd object (map_to_superrest_frame.py:719-765)"""
    ctx = getattr(abd, "_ctx", None)
    G = abd.bondi_CoM_charge() / abd.bondi_four_momentum()[:, 0, None]
    rel_err_CoM = _time_average(np.linalg.norm(G, axis=-1), abd.t, ctx)
    if target_strain is not None:
        omega = _news_angular_velocity_direction(abd)
        rel_err_rot = _time_average(np.linalg.norm(omega - _target_omega_spline(target_strain)(abd.t), axis=-1), abd.t, ctx)
    else:
        rel_err_rot = _time_average(np.linalg.norm(_unit_spin(abd) - np.array([0.0, 0, 1])[None, :], axis=-1), abd.t, ctx)
    PsiM = abd.supermomentum("Moreschi")
    i0 = int(np.argmin(abs(abd.t - 0)))
    PsiM0 = (PsiM.rows(i0, i0 + 1)[0] if hasattr(PsiM, "rows") else PsiM.ndarray[i0])[4:]
    if target_PsiM is not None:
        PsiM0 = PsiM0 - np.asarray(target_PsiM.data)[np.argmin(abs(target_PsiM.t - 0)), 4:]
    return rel_err_CoM, rel_err_rot, np.linalg.norm(PsiM0)


class _Series:
    """times + [n, modes] data + ell_max: what the iteration needs of a target supermomentum"""

    def __init__(self, t, data, ell_max):
        self.t, self.data, self.ell_max = np.array(t, dtype=float), np.array(data, dtype=complex), int(ell_max)


def map_to_superrest_frame(
    self,
    t_0=0,
    target_PsiM_input=None,
    target_strain_input=None,
    padding_time=250,
    N_itr_maxes={"superrest": 2, "CoM_transformation": 10, "rotation": 10, "supertranslation": 10},
    rel_err_tols={"CoM_transformation": 1e-12, "rotation": 1e-12, "supertranslation": 1e-12},
    order=["supertranslation", "rotation", "CoM_transformation"],
    ell_max=None,
    alpha_ell_max=None,
    modes=None,
    fix_xz_plane=False,
    fix_yz_plane=False,
    print_conv=False,
    Gfun=None,
    Gparams0=None,
    Gargsfun=None,
):
    """Transform an abd object to the super rest frame at time t_0 (or to the frame of a target Moreschi
    supermomentum): map_to_superrest_frame.py:768-1035.  Returns (abd_prime, BMSTransformation, best_rel_err)."""
    ctx = getattr(self, "_ctx", None)
    abd = self.copy()
    if order == []:
        return abd, BMSTransformation(ctx=ctx), None
    target_strain = None
    if target_strain_input is not None:
        target_strain = target_strain_input.copy()
        target_strain.t = target_strain.t - t_0
    target_PsiM = None
    if target_PsiM_input is not None:
        tp = target_PsiM_input
        target_PsiM = _Series(np.asarray(tp.t) - t_0, getattr(tp, "ndarray", getattr(tp, "data", tp)), tp.ell_max)
    if ell_max is None:
        ell_max = abd.ell_max
    i1 = np.abs(abd.t - (t_0 - (padding_time + 200))).argmin()
    i2 = np.abs(abd.t - (t_0 + (padding_time + 200))).argmin() + 1
    # the window the iterations work on stays in HBM across their transformations and charge evaluations (a dozen of each):
    # nothing but l <= 1 charge vectors and a few rows around u = 0 comes back to the host
    abd_sliced = abd[i1:i2].to_device(ctx)

    # a time translation first, so that the frame is fixed at u = 0
    time_translation = BMSTransformation(supertranslation=[constant_as_ell_0_mode(t_0)], ell_max=ell_max, ctx=ctx)
    BMS = (time_translation * BMSTransformation(ell_max=ell_max, ctx=ctx)).reorder(NORMAL)

    itr = 0
    rel_err = [np.inf, np.inf, np.inf]
    rel_errs = [[np.inf, np.inf, np.inf]]
    best_rel_err = [np.inf, np.inf, np.inf]
    best_BMS = BMS.copy()
    abd_sliced_prime = None
    while itr < N_itr_maxes["superrest"]:
        if isinstance(rel_err, tuple) and (
            rel_err[0] < rel_err_tols["CoM_transformation"] and rel_err[1] < rel_err_tols["rotation"] and rel_err[2] < rel_err_tols["supertranslation"]
        ):
            break
        if itr == 0:
            abd_sliced_prime = _transform(abd_sliced, BMS)
        for step in order:
            if step == "supertranslation":
                new, _ = supertranslation_to_map_to_superrest_frame(
                    abd_sliced_prime, target_PsiM, N_itr_max=N_itr_maxes["supertranslation"], rel_err_tol=rel_err_tols["supertranslation"],
                    ell_max=ell_max, print_conv=print_conv,
                )
            elif step == "rotation":
                new, _ = rotation_to_map_to_superrest_frame(
                    abd_sliced_prime, target_strain=target_strain, N_itr_max=N_itr_maxes["rotation"], rel_err_tol=rel_err_tols["rotation"], fix_xz_plane=fix_xz_plane,
                    fix_yz_plane=fix_yz_plane, print_conv=print_conv,
                )
            elif step == "CoM_transformation":
                new, _ = com_transformation_to_map_to_superrest_frame(
                    abd_sliced_prime, N_itr_max=N_itr_maxes["CoM_transformation"], rel_err_tol=rel_err_tols["CoM_transformation"],
                    print_conv=print_conv, Gfun=Gfun, Gparams0=Gparams0, Gargsfun=Gargsfun,
                )
            elif step == "time_phase":
                # time translation + turn about z that best match the strain to the target's on [-padding, +padding]
                # (map_to_superrest_frame.py:973-995).  Without a target there is nothing to align to and the step does
                # nothing (the reference re-applies whatever the previous step found, or fails when it is the first step).
                new = BMSTransformation(ell_max=ell_max, ctx=ctx)
                if target_strain is not None:
                    from .alignment import align2d

                    rel_err, _, res = align2d(
                        abd_sliced_prime.h, target_strain, 0 - padding_time, 0 + padding_time, n_brute_force_δt=None,
                        n_brute_force_δϕ=None, include_modes=modes, nprocs=4,
                    )
                    new = BMSTransformation(
                        supertranslation=[constant_as_ell_0_mode(res.x[0])], frame_rotation=_about_z(res.x[1]), ell_max=ell_max, ctx=ctx,
                    )
            else:
                raise ValueError(f"unknown step {step!r}")
            BMS = (new * BMS).reorder(NORMAL)
            abd_sliced_prime = _transform(abd_sliced, BMS)
        if not (target_strain is not None and order[-1] == "time_phase"):  # else: the alignment's own error
            rel_err = rel_err_for_abd_in_superrest(abd_sliced_prime, target_PsiM, target_strain)
        if np.mean(rel_err) < min(np.mean(r) for r in rel_errs):
            best_BMS = BMS.copy()
            best_rel_err = rel_err
        rel_errs.append(rel_err)
        itr += 1
    if print_conv:
        if not itr < N_itr_maxes["superrest"]:
            print(f"superrest: maximum number of iterations reached; the min error was {best_rel_err}.")
        else:
            print(f"superrest: tolerance achieved in {itr} iterations!")
    best_BMS = (time_translation.inverse() * best_BMS).reorder(NORMAL)
    return _transform(abd, best_BMS), best_BMS, best_rel_err
