"""The reference's initial-value tests (tests/test_abd_ivp.py:14-265) on the GPU implementation of
AsymptoticBondiData.from_initial_values and bondi_violation_norms.  The reference's products are Wigner-3j sums with
exact zeros for forbidden couplings; here they are grid products (exact to rounding), so "zero" modes are checked against
1e-13 of the data scale instead of == 0.  The numerical (sigma given on the time axis) branch is checked against the
analytic one."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def LM_index(ell, m):
    return ell * (ell + 1) + m


def construct_and_validate(ctx, modifier, validator, ell_max=8):
    import scri_amd

    time = np.linspace(-100, 100, num=2001)
    sigma, sigmadot, sigmaddot, psi2, psi1, psi0 = np.zeros((6, (ell_max + 1) ** 2), dtype=complex)
    modifier(sigma, sigmadot, sigmaddot, psi2, psi1, psi0)
    abd = scri_amd.AsymptoticBondiData.from_initial_values(time, ell_max, sigma, sigmadot, sigmaddot, psi2, psi1, psi0, ctx=ctx)
    validator(abd)


def check_modes(modes, nonzero_lm, scale=None):
    d = modes.ndarray
    nz = [LM_index(l, m) for l, m in nonzero_lm]
    zero = sorted(set(range(d.shape[-1])) - set(nz))
    top = max(np.abs(d).max(initial=0.0), 1e-300)
    assert np.abs(d[..., zero]).max(initial=0.0) < 1e-12 * top, "nonzero values among the forbidden modes"
    for i in nz:
        assert np.abs(d[..., i]).max() > 1e-12 * top, f"no nonzero values at index {i}"


def nonsense(sigma, sigmadot, sigmaddot, psi2, psi1, psi0):
    # values below the spin weight of the field: must have no effect
    psi0[: LM_index(1, 1)] = 1.234
    psi1[0] = 0.123
    sigma[: LM_index(1, 1)] = 0.567
    sigmadot[: LM_index(1, 1)] = 0.678
    sigmaddot[: LM_index(1, 1)] = 0.789


def viol(abd):
    return np.max(np.abs(abd.bondi_violation_norms))


def test0_forbidden_terms_only(ctx):
    def validator(abd):
        for f in (abd.psi0, abd.psi1, abd.psi2, abd.psi3, abd.psi4, abd.sigma):
            check_modes(f, [])
        assert viol(abd) == 0.0

    construct_and_validate(ctx, nonsense, validator)


def test1_psi2_monopole(ctx):
    def modifier(*a):
        nonsense(*a)
        a[3][LM_index(0, 0)] = 0.234

    def validator(abd):
        assert np.all(abd.psi2.ndarray[..., 0] == 0.234)
        check_modes(abd.psi2, [[0, 0]])
        for f in (abd.psi0, abd.psi1, abd.psi3, abd.psi4, abd.sigma):
            check_modes(f, [])
        assert viol(abd) < 1e-13

    construct_and_validate(ctx, modifier, validator, ell_max=3)


def test2_and_3_psi2_dipole_quadrupole(ctx):
    def modifier(*a):
        nonsense(*a)
        a[3][LM_index(0, 0)] = 0.234
        a[3][LM_index(1, -1)] = 0.345
        a[3][LM_index(2, -2)] = 0.456

    def validator(abd):
        assert np.all(abd.psi2.ndarray[..., 0] == 0.234)
        check_modes(abd.psi0, [[2, -2], [2, 2]])
        check_modes(abd.psi1, [[1, -1], [1, 1], [2, -2], [2, 2]])
        check_modes(abd.psi2, [[0, 0], [1, -1], [1, 1], [2, -2], [2, 2]])
        for f in (abd.psi3, abd.psi4, abd.sigma):
            check_modes(f, [])
        assert viol(abd) < 4e-11

    construct_and_validate(ctx, modifier, validator, ell_max=4)


def test4_constant_shear(ctx):
    def modifier(*a):
        nonsense(*a)
        a[3][0] = 0.234
        a[0][LM_index(2, 2)] = 0.5678

    def validator(abd):
        check_modes(abd.psi0, [[2, -2], [2, 0], [2, 2], [3, 0], [4, 0], [4, 4]], scale=np.abs(abd.psi0.ndarray).max())
        check_modes(abd.psi1, [[2, -2], [2, 2]])
        check_modes(abd.psi2, [[0, 0], [2, -2], [2, 2]])
        check_modes(abd.psi3, [])
        check_modes(abd.psi4, [])
        check_modes(abd.sigma, [[2, 2]])
        assert viol(abd) <= 2e-10

    construct_and_validate(ctx, modifier, validator, ell_max=6)


def test5_and_6_shear_derivatives(ctx):
    def modifier5(*a):
        nonsense(*a)
        a[0][LM_index(2, 2)] = 0.5678
        a[1][LM_index(2, 2)] = 0.6789

    def validator5(abd):
        check_modes(abd.psi0, [[2, -2], [2, 0], [2, 2], [3, 0], [4, 0], [4, 4]], scale=np.abs(abd.psi0.ndarray).max())
        check_modes(abd.psi1, [[1, 0], [2, -2], [2, 0], [2, 2], [3, 0], [4, 0]], scale=np.abs(abd.psi1.ndarray).max())
        check_modes(abd.psi2, [[2, -2], [2, 2]], scale=np.abs(abd.psi2.ndarray).max())
        check_modes(abd.psi3, [[2, -2]])
        check_modes(abd.psi4, [])
        check_modes(abd.sigma, [[2, 2]])
        assert viol(abd) <= 7e-9

    construct_and_validate(ctx, modifier5, validator5, ell_max=6)

    def modifier6(*a):
        nonsense(*a)
        a[2][LM_index(2, 2)] = 0.1 / 10_000**2

    def validator6(abd):
        check_modes(abd.psi3, [[2, -2]])
        check_modes(abd.psi4, [[2, -2]])
        check_modes(abd.sigma, [[2, 2]])
        check_modes(abd.psi2, [[0, 0], [1, 0], [2, -2], [2, 0], [3, 0], [4, 0]], scale=np.abs(abd.psi2.ndarray).max())
        assert viol(abd) <= 5e-8

    construct_and_validate(ctx, modifier6, validator6, ell_max=7)


def test7_random_and_the_numerical_branch(ctx):
    import scri_amd

    ell_max = 8
    rng = np.random.default_rng(1234)
    nm = (ell_max + 1) ** 2

    def modifier(sigma, sigmadot, sigmaddot, psi2, psi1, psi0):
        sigma[:] = 0.01 * rng.random(2 * nm).view(complex)
        sigmadot[:] = (0.01 / 100) * rng.random(2 * nm).view(complex)
        sigmaddot[:] = (0.01 / 100**2) * rng.random(2 * nm).view(complex)
        psi2[:] = 0.3 * rng.random(2 * nm).view(complex)
        psi1[:] = 0.1 * rng.random(2 * nm).view(complex)

    holder = {}

    def validator(abd):
        for f in (abd.psi0, abd.psi1, abd.psi2, abd.psi4, abd.sigma):
            s = abs(f.spin_weight)
            check_modes(f, [[l, m] for l in range(s, ell_max + 1) for m in range(-l, l + 1)])
        check_modes(abd.psi3, [[l, m] for l in range(2, ell_max + 1) for m in range(-l, l + 1)])
        assert viol(abd) <= 4.5e-6
        holder["abd"] = abd

    construct_and_validate(ctx, modifier, validator, ell_max=ell_max)
    # sigma as a function of time: spline calculus instead of polynomial algebra; values given at time[0]
    a = holder["abd"]
    b = scri_amd.AsymptoticBondiData.from_initial_values(
        a.t, ell_max, sigma0=a.sigma.ndarray.copy(), psi2=a.psi2.ndarray[0], psi1=a.psi1.ndarray[0], psi0=a.psi0.ndarray[0], ctx=ctx
    )
    for name in ("sigma", "psi4", "psi3", "psi2", "psi1", "psi0"):
        ref = getattr(a, name).ndarray
        assert np.abs(getattr(b, name).ndarray - ref).max() < 1e-9 * max(1.0, np.abs(ref).max()), name
    assert viol(b) <= 4.5e-6
