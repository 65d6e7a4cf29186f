"""The oracle's restatement of the WaveformModes mode-space operators (oracle/waveform_modes_ref.py), pinned by the reference's own
property tests: tests/test_waveform.py:273-342 (involutions, idempotents, null compositions, parity-violation measures, all with
zero tolerance) and tests/test_parity.py:13-61 (projections, np.array_equal), plus known answers for the ladder factors."""
import numpy as np
import pytest

from oracle import waveform_modes_ref as ref
from oracle.containers import WM, SpinWeights, h, psi0, psi1, psi2, psi3, psi4, sigma, news

DIRECTIONS = ["x", "y", "z", ""]


def random_waveform(dataType=h, ell_max=8, n=50, seed=0):
    """tests/conftest.py random_waveform of the reference: random modes, random unit frame"""
    rng = np.random.default_rng(seed + dataType)
    s = SpinWeights[dataType]
    ell_min = abs(s)
    nm = (ell_max + 1) ** 2 - ell_min**2
    t = np.linspace(0.0, 10.0, n)
    data = rng.normal(size=(n, nm)) + 1j * rng.normal(size=(n, nm))
    frame = rng.normal(size=(n, 4))
    frame /= np.linalg.norm(frame, axis=1)[:, None]
    return WM(t=t, data=data, ell_min=ell_min, ell_max=ell_max, dataType=dataType, frame=frame)


def same(a, b):
    return np.array_equal(a.data, b.data) and np.array_equal(a.frame, b.frame) and np.array_equal(a.t, b.t)


@pytest.mark.parametrize("d", DIRECTIONS)
def test_involutions_are_involute(d):  # tests/test_waveform.py:273-282
    w = random_waveform()
    assert same(ref.parity_conjugate(ref.parity_conjugate(w, d), d), w)


@pytest.mark.parametrize("d", DIRECTIONS)
def test_idempotents(d):  # :285-295
    w = ref.parity_symmetric_part(random_waveform(), d)
    assert same(ref.parity_symmetric_part(w, d), w)
    w = ref.parity_antisymmetric_part(random_waveform(), d)
    assert same(ref.parity_antisymmetric_part(w, d), w)


@pytest.mark.parametrize("d", DIRECTIONS)
def test_null_compositions(d):  # :298-316: data AND frame are zeroed
    w = random_waveform()
    for first, second in ((ref.parity_symmetric_part, ref.parity_antisymmetric_part), (ref.parity_antisymmetric_part, ref.parity_symmetric_part)):
        out = second(first(w, d), d)
        assert np.array_equal(out.data, np.zeros_like(w.data)) and np.array_equal(out.frame, np.zeros_like(w.frame))


@pytest.mark.parametrize("d", DIRECTIONS)
def test_parity_violation_measures(d):  # :319-342
    w = random_waveform()
    zeros, ones = np.zeros(w.n_times), np.ones(w.n_times)
    sym = ref.parity_symmetric_part(w, d)
    assert np.allclose(ref.parity_violation_squared(sym, d), zeros, atol=1e-15)
    assert np.allclose(ref.parity_violation_normalized(sym, d), zeros, atol=1e-15)
    anti = ref.parity_antisymmetric_part(w, d)
    assert np.allclose(ref.parity_violation_squared(w, d), ref.norm(anti), atol=0.0, rtol=1e-15)
    assert np.allclose(ref.parity_violation_normalized(w, d), np.sqrt(ref.norm(anti) / ref.norm(w)), atol=0.0, rtol=1e-15)
    assert np.allclose(ref.parity_violation_normalized(anti, d), ones, atol=0.0, rtol=1e-15)


@pytest.mark.parametrize("dataType", [psi0, psi1, psi2, psi3, psi4, h])
def test_parity_projections(dataType):  # tests/test_parity.py:13-61
    w = random_waveform(dataType=dataType)
    for d in DIRECTIONS:
        x = ref.parity_symmetric_part(w, d)
        assert np.array_equal(x.data, ref.parity_conjugate(x, d).data)
        assert np.array_equal(x.data, ref.parity_symmetric_part(x, d).data)
        assert np.array_equal(np.zeros_like(x.data), ref.parity_antisymmetric_part(x, d).data)
        assert np.array_equal(np.zeros_like(x.t), ref.parity_violation_squared(x, d))
        x = ref.parity_antisymmetric_part(w, d)
        assert np.array_equal(x.data, -ref.parity_conjugate(x, d).data)
        assert np.array_equal(x.data, ref.parity_antisymmetric_part(x, d).data)
        assert np.array_equal(np.zeros_like(x.data), ref.parity_symmetric_part(x, d).data)
        assert np.array_equal(ref.norm(x), ref.parity_violation_squared(x, d))


def test_parity_conjugates_are_the_reflections_of_the_field():
    """What the index games mean (Boyle et al. 2014, the paper the reference cites), checked pointwise with the oracle's own
    spin-weighted harmonics (gamma = 0 rotors): with Y(theta, -phi) = conj Y(theta, phi) the y conjugate is
    conj f(theta, -phi); with sY_lm(pi - theta, phi + pi) = (-1)^l -sY_lm(theta, phi) and conj sY_lm = (-1)^(s+m) -sY_l,-m the z
    conjugate is conj f(pi - theta, phi); the x conjugate is the y conjugate turned by pi about z, conj f(theta, pi - phi); all
    three together conj f(pi - theta, phi + pi), the antipodal point."""
    from oracle import wigner, quat

    for dataType in (psi4, psi1, psi2):
        w = random_waveform(dataType=dataType, ell_max=5, n=2)
        s = w.spin_weight

        def Y(theta, phi):
            return wigner.swsh_grid(quat.from_spherical_coords(theta, phi), s, w.ell_max)[w.ell_min**2 :]

        for theta, phi in ((0.7, 1.9), (2.1, -0.4), (1.3, 3.0)):
            cases = {"y": np.conj(w.data @ Y(theta, -phi)), "x": np.conj(w.data @ Y(theta, np.pi - phi)),
                     "z": np.conj(w.data @ Y(np.pi - theta, phi))}
            for d, rhs in cases.items():
                lhs = ref.parity_conjugate(w, d).data @ Y(theta, phi)
                assert np.abs(lhs - rhs).max() < 1e-13 * np.abs(rhs).max(), (dataType, d)
            lhs = ref.parity_conjugate(w, "").data @ Y(theta, phi)
            rhs = np.conj(w.data @ Y(np.pi - theta, phi + np.pi))
            assert np.abs(lhs - rhs).max() < 1e-13 * np.abs(rhs).max(), (dataType, "all")


def test_ladder_factors_known_answers():
    """eth sYlm = +sqrt((l - s)(l + s + 1)) s+1Ylm, ethbar sYlm = -sqrt((l + s)(l - s + 1)) s-1Ylm (NP); GHP carries 1/sqrt2 per
    operator; operators are applied right to left; below |s| the factor vanishes"""
    for ell in range(0, 7):
        for s in range(-3, 4):
            up = np.sqrt((ell - s) * (ell + s + 1.0)) if ell >= abs(s) else 0.0
            dn = -np.sqrt((ell + s) * (ell - s + 1.0)) if ell >= abs(s) else 0.0
            assert ref.ladder_factor("+", s, ell) == pytest.approx(up, abs=0, rel=1e-15)
            assert ref.ladder_factor("-", s, ell) == pytest.approx(dn, abs=0, rel=1e-15)
            assert ref.ladder_factor("ð", s, ell, "GHP") == pytest.approx(up / np.sqrt(2), abs=0, rel=1e-15)
            assert ref.ladder_factor([-1], s, ell) == pytest.approx(dn, abs=0, rel=1e-15)
    # ethbar eth on spin 0 = -l(l+1) (the Laplacian), eth ethbar on spin -2
    for ell in range(2, 8):
        assert ref.ladder_factor("-+", 0, ell) == pytest.approx(-ell * (ell + 1.0), rel=1e-14)
        assert ref.ladder_factor("+-", -2, ell) == pytest.approx(-(ell - 2.0) * (ell + 3.0), rel=1e-14)
    with pytest.raises(ValueError):
        ref.ladder_factor("+x", 0, 2)
    with pytest.raises(ValueError):
        ref.ladder_factor("+", 0, 2, eth_convention="XY")


def test_conjugate_pairs_round_trip_and_norm():
    w = random_waveform(dataType=h, ell_max=6)
    p = ref.convert_to_conjugate_pairs(w)
    assert np.allclose(ref.norm(p), ref.norm(w), rtol=1e-14, atol=0)  # the sqrt2 keeps the norm (docstring, :659-676)
    back = ref.convert_from_conjugate_pairs(p)
    assert np.abs(back.data - w.data).max() < 4e-16 * np.abs(w.data).max()
    # m = 0 untouched
    for ell in range(w.ell_min, w.ell_max + 1):
        i = ref.LM_index(ell, 0, w.ell_min)
        assert np.array_equal(p.data[:, i], w.data[:, i])


def test_truncate_bounds_the_error_and_is_idempotent():
    w = random_waveform(dataType=h, ell_max=6)
    for tol in (1e-10, 1e-4):
        tr = ref.truncate(w, tol)
        err = np.linalg.norm(tr.data - w.data, axis=1)
        assert (err <= tol * np.linalg.norm(w.data, axis=1)).all()
        assert np.array_equal(ref.truncate(tr, tol).data, tr.data) or np.abs(ref.truncate(tr, tol).data - tr.data).max() < tol
    assert np.array_equal(ref.truncate(w, 0.0).data, w.data)


def test_inner_product_of_polynomial_data_is_exact():
    t = np.linspace(0.0, 2.0, 41)
    a = WM(t=t, data=(t[:, None] * np.array([[1 + 1j]])), ell_min=0, ell_max=0, dataType=psi2)
    b = WM(t=t, data=(t[:, None] ** 2 * np.array([[2 - 1j]])), ell_min=0, ell_max=0, dataType=psi2)
    # integral of conj((1 + i) t) (2 - i) t^2 = (1 - i)(2 - i) t^3 -> (1 - 3i) t^4 / 4
    assert ref.inner_product(a, b) == pytest.approx((1 - 3j) * 2.0**4 / 4, rel=1e-14)
    assert ref.inner_product(a, b, t1=0.5, t2=1.5) == pytest.approx((1 - 3j) * (1.5**4 - 0.5**4) / 4, rel=1e-14)
    with pytest.raises(ValueError):
        ref.inner_product(a, WM(t=t, data=b.data, ell_min=0, ell_max=0, dataType=psi1))
    with pytest.raises(ValueError):
        ref.inner_product(a, WM(t=t + 1, data=b.data, ell_min=0, ell_max=0, dataType=psi2))
