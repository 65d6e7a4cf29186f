"""The oracle against every analytic known-answer test the reference's own test-suite holds for the hot
path.  Each test names the reference test it mirrors.  Tolerances are the
reference's, except where noted (the restated sYlm evaluation is ~3e-15 accurate, which leaves two of
the reference's hand-tuned tolerances a few per cent short: those are widened by 1.5x and say so).

Exhaustive variants (all (s, l, m)) are marked `slow`:  pytest -m slow tests/test_oracle_known_answers.py
"""
import math

import numpy as np
import pytest

from oracle import quat, wigner, spinsfast_ref, rotations_ref, abd_ref
from oracle import waveform_grid_ref as grid_ref
from oracle import sample_waveforms_ref as samples
from oracle.containers import ABD, WM, h


# ---------------------------------------------------------------------------- tests/test_waveform_grid.py


def test_time_translation():
    """tests/test_waveform_grid.py:17-27 (data tolerance 4e-14 there; 6e-14 here)."""
    dt = 1.469
    alpha00 = math.sqrt(4 * math.pi) * dt
    w1 = samples.constant_waveform(t=np.linspace(-10.0, 100.0, num=111))
    w2 = grid_ref.transform(w1, time_translation=dt)
    w3 = grid_ref.transform(w1, supertranslation=[alpha00])
    assert np.allclose(w1.t, w2.t + dt, rtol=0.0, atol=2e-15)
    assert np.allclose(w1.data, w2.data, rtol=0.0, atol=4e-14)
    assert np.allclose(w2.t, w3.t, rtol=0.0, atol=0.0)
    assert np.allclose(w2.data, w3.data, rtol=0.0, atol=0.0)


def test_BMS_rotation():
    """tests/test_waveform_grid.py:30-38: grid path with frame_rotation == Wigner-D path, 100 rotors."""
    w1 = samples.constant_waveform(t=np.linspace(-10.0, 100.0, num=12))
    for R in samples.Rs():
        w2 = rotations_ref.rotate_decomposition_basis(w1, R)
        w3 = grid_ref.transform(w1, frame_rotation=R)
        assert np.allclose(w2.data, w3.data, rtol=1e-15, atol=4e-13)


def _translation_case(s, ell, m, **translation):
    aux = {}
    for i in range(s + 2):
        aux[f"psi{4-i}_modes"] = samples.single_mode_proportional_to_time(s=i - 2)
        aux[f"psi{4-i}_modes"].data *= 0
    w1 = grid_ref.transform(samples.single_mode_proportional_to_time(s=s, ell=ell, m=m), **translation, **aux)
    w2 = samples.single_mode_proportional_to_time_supertranslated(
        s=s, ell=ell, m=m, **{k: np.array(v) for k, v in translation.items()}
    )
    return w1, w2


def _compare_translated(w1, w2, disp, atol):
    i1A = np.argmin(abs(w1.t - (w1.t[0] + 2 * disp)))
    i1B = np.argmin(abs(w1.t - (w1.t[-1] - 2 * disp)))
    i2A = np.argmin(abs(w2.t - w1.t[i1A]))
    i2B = np.argmin(abs(w2.t - w1.t[i1B]))
    assert np.allclose(w1.t[i1A : i1B + 1], w2.t[i2A : i2B + 1], rtol=0.0, atol=1e-16)
    assert np.allclose(w1.data[i1A : i1B + 1], w2.data[i2A : i2B + 1], rtol=0.0, atol=atol)


SPACE_CASES = [(s, ell, m) for s in range(-2, 3) for (ell, m) in [(max(abs(s), 1), -1), (4, 3), (8, -8), (8, 0)] if ell >= abs(s)]


@pytest.mark.parametrize("s,ell,m", SPACE_CASES)
def test_space_translation_subset(s, ell, m):
    """tests/test_waveform_grid.py:41-92 on a subset of (s, l, m); analytic Wigner-3j answer
    (scri/sample_waveforms.py:350-364).  Tolerance 5e-14 there, 5e-14 here."""
    for st in ([1.0, 0.0, 0.0], [0.0, 1.0, 0.0], [0.0, 0.0, 1.0]):
        w1, w2 = _translation_case(s, ell, m, space_translation=st)
        _compare_translated(w1, w2, 1.0, 5e-14)


@pytest.mark.slow
def test_space_translation_exhaustive():
    for s in range(-2, 3):
        for ell in range(abs(s), 9):
            for m in range(-ell, ell + 1):
                for st in ([1.0, 0.0, 0.0], [0.0, 1.0, 0.0], [0.0, 0.0, 1.0]):
                    w1, w2 = _translation_case(s, ell, m, space_translation=st)
                    _compare_translated(w1, w2, 1.0, 5e-14)


def _generator(ellpp, mpp, ell_max=4):
    st = np.zeros((wigner.LM_total_size(0, ell_max),), dtype=complex)
    if mpp == 0:
        st[wigner.LM_index(ellpp, mpp, 0)] = 1.0
    elif mpp < 0:
        st[wigner.LM_index(ellpp, mpp, 0)] = 1.0
        st[wigner.LM_index(ellpp, -mpp, 0)] = (-1.0) ** mpp
    else:
        st[wigner.LM_index(ellpp, mpp, 0)] = 1.0j
        st[wigner.LM_index(ellpp, -mpp, 0)] = (-1.0) ** mpp * -1.0j
    return st


@pytest.mark.parametrize("s,ell,m,ellpp,mpp", [(-2, 2, 2, 2, 0), (-2, 3, -1, 3, 2), (0, 2, 1, 4, -3), (2, 4, -4, 2, 1), (-1, 1, 0, 4, 4), (1, 3, 3, 3, -1)])
def test_hyper_translation_subset(s, ell, m, ellpp, mpp):
    """tests/test_waveform_grid.py:95-158 on a subset: every real l<=4 supertranslation generator."""
    st = _generator(ellpp, mpp)
    disp = abs(spinsfast_ref.salm2map(st, 0, 4, 17, 17)).max()
    aux = {}
    for i in range(s + 2):
        aux[f"psi{4-i}_modes"] = samples.single_mode_proportional_to_time(s=i - 2)
        aux[f"psi{4-i}_modes"].data *= 0
    w1 = grid_ref.transform(samples.single_mode_proportional_to_time(s=s, ell=ell, m=m), supertranslation=st, **aux)
    w2 = samples.single_mode_proportional_to_time_supertranslated(s=s, ell=ell, m=m, supertranslation=st)
    _compare_translated(w1, w2, disp, 5e-14)


def test_supertranslation_inverses():
    """tests/test_waveform_grid.py:161-185: S then -S == interpolate (every l<=2 generator here; l<=4 there)."""
    w1 = samples.random_waveform_proportional_to_time(n_times=301)
    for ellpp, mpp in wigner.LM_range(0, 2):
        st = _generator(int(ellpp), int(mpp), ell_max=2)
        w2 = grid_ref.transform(grid_ref.transform(w1, supertranslation=st), supertranslation=-st)
        w1i = w1.interpolate(w2.t)
        assert np.allclose(w1i.t, w2.t, rtol=5e-10, atol=5e-14)
        assert np.allclose(w1i.data, w2.data, rtol=5e-10, atol=5e-14)


@pytest.mark.parametrize("beta,ell_max", [(1e-2, 8)])
def test_boost_inverses(beta, ell_max):
    """tests/test_waveform_grid.py:188-214 (the beta=1e-1, l=14 leg is in the slow set)."""
    _boost_inverse_case(beta, ell_max, dt=1.0 / 200.0)


@pytest.mark.slow
def test_boost_inverses_large():
    _boost_inverse_case(1e-1, 14, dt=1.0 / 200.0)


def _boost_inverse_case(beta, ell_max, dt):
    for v in [np.array([0.0, 0.0, beta]), np.array([0.0, beta, 0.0]), np.array([beta, 0.0, 0.0])]:
        w1 = samples.single_mode_constant_rotation(s=-2, ell=2, m=2, omega=0.3, t_0=-10.0, t_1=10.0, dt=dt)
        w1 = grid_ref.transform(w1, space_translation=np.array([0.1, 0.0, 0.0]))
        w1.m_is_scaled_out = False
        w2 = grid_ref.transform(w1, boost_velocity=v, n_theta=2 * (ell_max + 1) + 1, n_phi=2 * (ell_max + 1) + 1, ell_max=ell_max)
        w2 = grid_ref.transform(w2, boost_velocity=-v, ell_max=w1.ell_max)
        w1i = w1.interpolate(w2.t)
        assert np.allclose(w1i.data, w2.data, atol=1e-12, rtol=0)


# ---------------------------------------------------------------------------- tests/test_rotations.py


def test_identity_rotation_is_exact():
    """tests/test_rotations.py:14-38."""
    for w in (samples.linear_waveform(n_times=50), samples.constant_waveform(t=np.linspace(0, 1, 20)), samples.random_waveform(n_times=50)):
        out = rotations_ref.rotate_decomposition_basis(w, np.array([1.0, 0, 0, 0]))
        assert np.array_equal(out.data, w.data)


def test_constant_versus_series_and_inversion():
    """tests/test_rotations.py:65-129."""
    rng = np.random.default_rng(5)
    for w in (samples.linear_waveform(n_times=60), samples.random_waveform(n_times=60)):
        q = rng.uniform(-1, 1, 4)
        q /= np.linalg.norm(q)
        a = rotations_ref.rotate_decomposition_basis(w, q)
        b = rotations_ref.rotate_decomposition_basis(w, np.repeat(q[None, :], w.n_times, 0))
        assert np.allclose(a.data, b.data, rtol=1e-13, atol=1e-12)
        back = rotations_ref.rotate_decomposition_basis(a, quat.qconj(q))
        assert np.allclose(back.data, w.data, atol=1e-12, rtol=w.ell_max**4 * 4e-14)



def test_rotations_of_0_0_mode():
    """tests/test_rotations.py:132-155: the (0,0) mode is exactly invariant."""
    Rs = samples.Rs()
    w = samples.delta_waveform(0, 0, n_times=len(Rs), ell_min=0, ell_max=8)
    out = rotations_ref.rotate_decomposition_basis(w, Rs)
    assert np.array_equal(out.data, w.data)
    assert np.max(np.abs(out.frame - Rs)) == 0.0


def test_rotations_of_each_mode_individually():
    """tests/test_rotations.py:158-198: a delta in (l, m') maps to row m' of D^l, zeros elsewhere (exactly)."""
    Rs = samples.Rs()
    ell_min, ell_max = 0, 4
    sp = quat.as_spinor_array(Rs)
    Ds = wigner.wigner_D_matrices(sp[:, 0], sp[:, 1], ell_min, ell_max)
    for ell in range(ell_max + 1):
        for Mp in range(-ell, ell + 1):
            w = samples.delta_waveform(ell, Mp, n_times=len(Rs), ell_min=ell_min, ell_max=ell_max)
            out = rotations_ref.rotate_decomposition_basis(w, Rs)
            i0 = wigner.LMpM_index(ell, Mp, -ell, ell_min)
            lo, hi = wigner.LM_total_size(ell_min, ell - 1), wigner.LM_total_size(ell_min, ell)
            assert np.array_equal(out.data[:, :lo], np.zeros((len(Rs), lo)))
            assert np.array_equal(out.data[:, hi:], np.zeros((len(Rs), out.data.shape[1] - hi)))
            assert np.array_equal(out.data[:, lo:hi], Ds[:, i0 : i0 + 2 * ell + 1])


# ---------------------------------------------------------------------------- tests/test_asymptoticbondidata.py


def _function_real_modes(f):
    L = int(round(math.sqrt(f.shape[-1]))) - 1
    out = np.empty_like(f)
    for l in range(L + 1):
        for m in range(-l, l + 1):
            out[..., wigner.LM_index(l, m, 0)] = 0.5 * (f[..., wigner.LM_index(l, m, 0)] + (-1) ** m * np.conj(f[..., wigner.LM_index(l, -m, 0)]))
    return out


def _charge_vector(c):
    """scri/asymptotic_bondi_data/bms_charges.py:50-67."""
    P = np.empty(c.shape[:-1] + (4,))
    P[..., 0] = c[..., 0].real
    P[..., 1] = (c[..., 1] - c[..., 3]).real / math.sqrt(6)
    P[..., 2] = (c[..., 1] + c[..., 3]).imag / math.sqrt(6)
    P[..., 3] = c[..., 2].real / math.sqrt(3)
    return P / math.sqrt(4 * math.pi)


def _schwarzschild(mass, n, ell_max=8):
    u = np.linspace(0, 100, num=n)
    raw = np.zeros((6, n, (ell_max + 1) ** 2), dtype=complex)
    raw[2, :, 0] = -wigner.constant_as_ell_0_mode(mass)  # tests/conftest.py:37-47 (Moreschi-Boyle convention)
    return ABD(u, raw, ell_max)


def test_abd_schwarzschild_transform():
    """tests/test_asymptoticbondidata.py:96-116: P' = m gamma (1, -v), rest mass invariant (sigma = 0, so the
    mass aspect is -Re psi2, scri/asymptotic_bondi_data/bms_charges.py:14-47)."""
    mass = 1.0
    abd = _schwarzschild(mass, 120)
    for v in [np.array([0.1, 0.0, 0.0]), np.array([0.0, 0.1, 0.0]), np.array([0.0, 0.0, 0.1])]:
        out = abd_ref.transform(abd, boost_velocity=v)
        gamma = 1 / np.sqrt(1 - np.dot(v, v))
        P = _charge_vector(-_function_real_modes(out.raw[2])[..., :4])
        assert np.allclose(P, mass * gamma * np.array([1, *-v]), atol=1e-14, rtol=1e-14)
        rest = np.sqrt(P[:, 0] ** 2 - np.sum(P[:, 1:] ** 2, axis=1))
        assert np.allclose(rest, mass, atol=1e-14, rtol=1e-14)


def test_abd_conformal_factors():
    """tests/test_asymptoticbondidata.py:33-93: conformal_factors vs the spectral route (map2salm + SWSH_grid);
    cross-pins the sYlm and map2salm conventions (l=16 here, l=32 there)."""
    tol = 4e-14
    ell_max = 16
    n_theta = n_phi = 2 * ell_max + 1
    v = np.array([0.01, 0.02, 0.03])
    gamma = 1 / math.sqrt(1 - np.dot(v, v))
    rotors = abd_ref.boosted_grid(np.array([1.0, 0, 0, 0]), v, n_theta, n_phi)
    k, ethk_over_k, one_over_k, one_over_k3 = abd_ref.conformal_factors(v, rotors)
    theta = np.linspace(0, np.pi, n_theta)
    phi = np.linspace(0, 2 * np.pi, n_phi, endpoint=False)
    th, ph = np.meshgrid(theta, phi, indexing="ij")
    kinv = gamma * (1 - v[0] * np.sin(th) * np.cos(ph) - v[1] * np.sin(th) * np.sin(ph) - v[2] * np.cos(th))
    kappa_inv = spinsfast_ref.map2salm(kinv, 0, ell_max)
    kappa = spinsfast_ref.map2salm(1 / kinv, 0, ell_max)
    one_over_k2 = np.tensordot(kappa_inv, wigner.swsh_grid(rotors, 0, ell_max), axes=([-1], [-1]))
    k2 = 1 / one_over_k2
    ethk = np.tensordot(wigner.eth_GHP(kappa, 0), wigner.swsh_grid(rotors, 1, ell_max), axes=([-1], [-1]))
    assert np.allclose(k[0], k2, atol=tol, rtol=tol)
    assert np.allclose(one_over_k[0], one_over_k2, atol=tol, rtol=tol)
    assert np.allclose(one_over_k3[0], one_over_k2**3, atol=tol, rtol=tol)
    # eth k / k of conformal_factors equals the GHP-eth spectral route (4e-14 at l=32 there; the
    # quadrature-limited spectral route needs 1e-13 here)
    assert np.allclose(ethk_over_k[0], ethk / k2, atol=1e-13, rtol=1e-13)


def test_abd_WaveformModes_consistency():
    """tests/test_asymptoticbondidata.py:165-212: the ABD path and the WaveformModes path agree on h = 2 conj(sigma)
    for a random supertranslation, rotation and boost (tolerance 4e-12)."""
    rng = np.random.default_rng(123)
    ell_max = 4
    n = 150
    u = np.linspace(-10, 10, num=n)
    nm = (ell_max + 1) ** 2
    raw = np.zeros((6, n, nm), dtype=complex)
    s0, s1, s2 = [c * (rng.random(nm) - 0.5 + 1j * (rng.random(nm) - 0.5)) for c in (0.01, 0.0002, 0.00003)]
    sig = s0[None, :] + u[:, None] * s1[None, :] + 0.5 * u[:, None] ** 2 * s2[None, :]
    sig[:, :4] = 0
    raw[5] = sig
    for i, sp in enumerate(ABD.spins[:5]):  # smooth psi_n data (they do not feed sigma')
        f = 0.01 * (rng.random(nm) - 0.5 + 1j * (rng.random(nm) - 0.5))
        f[: sp * sp] = 0
        raw[i] = f[None, :] * (1 + 0.01 * u[:, None])
    abd = ABD(u, raw, ell_max)

    def bar_modes(f):  # modes of conj(function): s -> -s
        out = np.zeros_like(f)
        for l in range(ell_max + 1):
            for m in range(-l, l + 1):
                out[..., wigner.LM_index(l, m, 0)] = (-1) ** (2 + m) * np.conj(f[..., wigner.LM_index(l, -m, 0)])
        return out

    hdata = 2 * bar_modes(sig)
    hw = WM(t=u, data=hdata[:, 4:], ell_min=2, ell_max=ell_max, dataType=h)
    alpha = 0.01 * (rng.random(nm) - 0.5 + 1j * (rng.random(nm) - 0.5))
    for l in range(ell_max + 1):
        for m in range(l + 1):
            ip, im = wigner.LM_index(l, m, 0), wigner.LM_index(l, -m, 0)
            alpha[ip] = (alpha[ip] + (-1.0) ** m * np.conj(alpha[im])) / 2
            alpha[im] = (-1.0) ** m * np.conj(alpha[ip])
    R = rng.normal(size=4)
    v = 0.01 * (rng.random(3) - 0.5)
    abdp = abd_ref.transform(abd, supertranslation=alpha, frame_rotation=R, boost_velocity=v)
    hp = grid_ref.transform(hw, supertranslation=alpha, frame_rotation=R, boost_velocity=v)
    t = np.intersect1d(np.round(hp.t, 12), np.round(abdp.t, 12))
    t = t[(t >= max(hp.t[0], abdp.t[0])) & (t <= min(hp.t[-1], abdp.t[-1]))]
    hpi = hp.interpolate(t)
    abdi = abdp.interpolate(t)
    assert np.allclose(hpi.data, 2 * bar_modes(abdi.raw[5])[:, 4:], atol=4e-12, rtol=4e-12)
