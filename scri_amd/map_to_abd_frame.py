"""Map an AsymptoticBondiData object to the BMS frame of another one (scri/asymptotic_bondi_data/map_to_abd_frame.py:20-301):
both objects are taken to their super rest frames and the two transformations composed, iterating on the result.  The data
passes (transform, interpolate, charges, norms, time integrals) are the GPU building blocks of this package; the
time/phase fixing step (`fix_time_phase_freedom=True`) uses scri_amd.alignment.align2d, this package's restatement of the
third-party `sxs.waveforms.alignment.align2d`."""
import numpy as np

from . import engine
from .bms_transformations import BMSTransformation
from .mode_algebra import constant_as_ell_0_mode

NORMAL = ["supertranslation", "frame_rotation", "boost_velocity"]


def _integral(values, t, ctx=None):
    """definite integral of a real series over its time range (cubic-spline antiderivative, as quaternion.calculus)"""
    f = np.asarray(values, dtype=float)[:, None] + 0j
    return engine.spline_derivative(t, f, np.array([t[-1]]), -1, ctx=ctx)[0, 0].real


def rel_err_between_abds(abd1, abd2, t1, t2):
    """mean over the six fields of  int |f1 - f2| dt / int |f1| dt  on [t1, t2] (map_to_abd_frame.py:20-56)"""
    ctx = getattr(abd1, "_ctx", None)
    t_array = abd1.t[np.argmin(abs(abd1.t - t1)) : np.argmin(abs(abd1.t - t2)) + 1]
    a = abd1.interpolate(t_array)
    b = abd2.interpolate(t_array)
    rel_err = 0.0
    for name in ("sigma", "psi0", "psi1", "psi2", "psi3", "psi4"):
        fa, fb = getattr(a, name), getattr(b, name)
        diff = fa - fb
        rel_err += _integral(diff.norm(), t_array, ctx) / (_integral(fa.norm(), t_array, ctx) or (t_array[-1] - t_array[0]))
    return rel_err / 6


def map_to_abd_frame(
    self,
    target_abd,
    t_0=0,
    padding_time=250,
    N_itr_maxes={"abd": 2, "superrest": 2, "CoM_transformation": 10, "rotation": 10, "supertranslation": 10},
    rel_err_tols={"CoM_transformation": 1e-12, "rotation": 1e-12, "supertranslation": 1e-12},
    order=["supertranslation", "rotation", "CoM_transformation"],
    ell_max=None,
    alpha_ell_max=None,
    fix_time_phase_freedom=True,
    nprocs=4,
    print_conv=False,
):
    """Transform an abd object to the frame of a target abd object using data around t_0
    (map_to_abd_frame.py:59-301).  Returns (abd_prime, BMSTransformation, rel_err)."""
    ctx = getattr(self, "_ctx", None)
    abd = self.copy()
    target_strain = target_abd.h

    def time_phase(strain, target, t_a, t_b):
        """the time translation + turn about z that carries `strain` onto `target` on [t_a, t_b], and the alignment's error"""
        from .alignment import align2d

        err, _, res = align2d(strain, target, t_a, t_b, n_brute_force_δt=None, n_brute_force_δϕ=None, include_modes=None, nprocs=nprocs)
        half = 0.5 * res.x[1]
        return err, BMSTransformation(
            supertranslation=[constant_as_ell_0_mode(res.x[0])], frame_rotation=np.array([np.cos(half), 0.0, 0.0, np.sin(half)]), ctx=ctx
        )

    time_translation = BMSTransformation(ctx=ctx)
    if fix_time_phase_freedom:
        # start from the time at which the Bondi energy is the target's at t_0, so that the two are reasonably close
        energy = abd.bondi_four_momentum()[:, 0]
        target_energy = target_abd.bondi_four_momentum()[:, 0]
        time_translation = BMSTransformation(
            supertranslation=[constant_as_ell_0_mode(abd.t[np.argmin(abs(energy - target_energy[np.argmin(abs(target_abd.t - t_0))]))] - t_0)],
            ctx=ctx,
        )
    BMS = (time_translation * BMSTransformation(ctx=ctx)).reorder(NORMAL)
    abd_interp = abd.interpolate(
        abd.t[np.argmin(abs(abd.t - (t_0 - 1.5 * padding_time))) : np.argmin(abs(abd.t - (t_0 + 1.5 * padding_time))) + 1]
    )
    superrest_kw = dict(
        t_0=t_0, padding_time=padding_time, N_itr_maxes=N_itr_maxes, rel_err_tols=rel_err_tols, ell_max=ell_max,
        alpha_ell_max=alpha_ell_max, print_conv=print_conv, order=order,
    )
    target_abd_superrest, transformation2, _ = target_abd.map_to_superrest_frame(**superrest_kw)
    target_strain_superrest = target_abd_superrest.h

    def apply(a, B):
        return a.transform(supertranslation=B.supertranslation, frame_rotation=B.frame_rotation.components, boost_velocity=B.boost_velocity)

    itr, rel_errs = 0, [np.inf]
    best, best_rel_err = BMS.copy(), np.inf
    abd_interp_prime = None
    while itr < N_itr_maxes["abd"]:
        if itr == 0:
            abd_interp_prime = apply(abd_interp, BMS)
        # to the super rest frame, and from there to the target's frame
        abd_interp_superrest, transformation1, _ = abd_interp_prime.map_to_superrest_frame(**superrest_kw)
        between = BMSTransformation(ctx=ctx)
        if fix_time_phase_freedom:  # align in the super rest frame, where only time and phase are left free
            _, between = time_phase(abd_interp_superrest.h, target_strain_superrest, t_0 - padding_time, t_0 + padding_time)
        BMS = (transformation2.inverse() * (between * (transformation1 * BMS))).reorder(NORMAL)
        abd_interp_prime = apply(abd_interp, BMS)
        if fix_time_phase_freedom:  # and once more in the target's frame
            rel_err, again = time_phase(abd_interp_prime.h, target_strain, t_0 - padding_time, t_0 + padding_time)
            BMS = (again * BMS).reorder(NORMAL)
            abd_interp_prime = apply(abd_interp, BMS)
        else:
            rel_err = rel_err_between_abds(target_abd, abd_interp_prime, t_0 - padding_time, t_0 + padding_time)
        if rel_err < min(rel_errs):
            best, best_rel_err = BMS.copy(), rel_err
        rel_errs.append(rel_err)
        itr += 1
    if print_conv:
        if not itr < N_itr_maxes["abd"]:
            print(f"BMS: maximum number of iterations reached; the min error was {best_rel_err}.")
        else:
            print(f"BMS: tolerance achieved in {itr} iterations!")
    return apply(abd, best), best, best_rel_err
