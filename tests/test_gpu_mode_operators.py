"""Mode-space operators of WaveformModes (scri_amd/mode_operators.py, all on `bms_mode_map`) against the oracle's restatement of
scri/waveform_modes.py:458-943, and the reference's own property tests on the GPU class: tests/test_parity.py:13-61 (np.array_equal
throughout) and tests/test_waveform.py:273-342 (zero tolerances)."""
import numpy as np
import pytest

from oracle import waveform_modes_ref as ref
from oracle.containers import WM, SpinWeights, h, psi0, psi1, psi2, psi3, psi4, sigma, news
from tests.test_oracle_mode_operators import random_waveform

pytestmark = pytest.mark.gpu

DIRECTIONS = ["x_", "y_", "z_", ""]


def gpu(o, ctx, device=False):
    import scri_amd

    w = scri_amd.WaveformModes(t=o.t, data=o.data.copy(), ell_min=o.ell_min, ell_max=o.ell_max, dataType=o.dataType, frameType=scri_amd.Inertial,
                               frame=o.frame, r_is_scaled_out=True, m_is_scaled_out=True, ctx=ctx)
    return w.to_device() if device else w


@pytest.mark.parametrize("device", [False, True])
@pytest.mark.parametrize("dataType", [psi0, psi1, psi2, psi3, psi4, h, sigma, news])
def test_parity_operators_equal_the_oracle_bit_for_bit(ctx, dataType, device):
    o = random_waveform(dataType=dataType, ell_max=7, n=40)
    for d in DIRECTIONS:
        w = gpu(o, ctx, device)
        for part, fn in (("conjugate", ref.parity_conjugate), ("symmetric_part", ref.parity_symmetric_part),
                         ("antisymmetric_part", ref.parity_antisymmetric_part)):
            got = getattr(w, f"{d}parity_{part}")
            assert got.is_device_resident == device  # a device-resident object's results stay in HBM
            expect = fn(o, d.rstrip("_"))
            assert np.array_equal(got.data, expect.data), (d, part)
            assert np.array_equal(got.frame, expect.frame) and np.array_equal(got.t, o.t)
            assert (got.ell_min, got.ell_max, got.dataType) == (o.ell_min, o.ell_max, o.dataType)
        assert np.array_equal(getattr(w, f"{d}parity_violation_squared"), ref.parity_violation_squared(o, d.rstrip("_")))
        assert np.array_equal(getattr(w, f"{d}parity_violation_normalized"), ref.parity_violation_normalized(o, d.rstrip("_")))


@pytest.mark.parametrize("dataType", [psi0, psi1, psi2, psi3, psi4, h])
def test_parity_projections(ctx, dataType):  # the reference's tests/test_parity.py:13-61 on the GPU class
    w = gpu(random_waveform(dataType=dataType), ctx)
    for d in DIRECTIONS:
        P = lambda x, part: getattr(x, f"{d}parity_{part}")  # noqa: E731
        x = P(w, "symmetric_part")
        assert np.array_equal(x.data, P(x, "conjugate").data)
        assert np.array_equal(x.data, P(x, "symmetric_part").data)
        assert np.array_equal(np.zeros_like(x.data), P(x, "antisymmetric_part").data)
        assert np.array_equal(np.zeros_like(x.t), getattr(x, f"{d}parity_violation_squared"))
        x = P(w, "antisymmetric_part")
        assert np.array_equal(x.data, -P(x, "conjugate").data)
        assert np.array_equal(x.data, P(x, "antisymmetric_part").data)
        assert np.array_equal(np.zeros_like(x.data), P(x, "symmetric_part").data)
        assert np.array_equal(x.norm(), getattr(x, f"{d}parity_violation_squared"))


def test_involutions_idempotents_null_compositions_and_measures(ctx):  # tests/test_waveform.py:273-342
    o = random_waveform()
    w = gpu(o, ctx)
    zeros, ones = np.zeros(w.n_times), np.ones(w.n_times)
    for d in DIRECTIONS:
        twice = getattr(getattr(w, f"{d}parity_conjugate"), f"{d}parity_conjugate")
        assert np.array_equal(twice.data, w.data) and np.array_equal(twice.frame, w.frame) and np.array_equal(twice.t, w.t)
        for part in ("symmetric_part", "antisymmetric_part"):
            once = getattr(w, f"{d}parity_{part}")
            again = getattr(once, f"{d}parity_{part}")
            assert np.array_equal(again.data, once.data) and np.array_equal(again.frame, once.frame)
        for first, second in (("symmetric_part", "antisymmetric_part"), ("antisymmetric_part", "symmetric_part")):
            out = getattr(getattr(w, f"{d}parity_{first}"), f"{d}parity_{second}")
            assert np.array_equal(out.data, np.zeros_like(w.data)) and np.array_equal(out.frame, np.zeros_like(w.frame))
        sym, anti = getattr(w, f"{d}parity_symmetric_part"), getattr(w, f"{d}parity_antisymmetric_part")
        assert np.allclose(getattr(sym, f"{d}parity_violation_squared"), zeros, atol=1e-15)
        assert np.allclose(getattr(sym, f"{d}parity_violation_normalized"), zeros, atol=1e-15)
        assert np.allclose(getattr(w, f"{d}parity_violation_squared"), anti.norm(), atol=0.0, rtol=1e-15)
        assert np.allclose(getattr(w, f"{d}parity_violation_normalized"), np.sqrt(anti.norm() / w.norm()), atol=0.0, rtol=1e-15)
        assert np.allclose(getattr(anti, f"{d}parity_violation_normalized"), ones, atol=0.0, rtol=1e-15)
    import scri_amd

    u = scri_amd.WaveformModes(t=o.t, data=o.data, ell_min=o.ell_min, ell_max=o.ell_max, ctx=ctx)  # UnknownDataType
    with pytest.raises(ValueError, match="Cannot compute parity type"):
        u.x_parity_conjugate


@pytest.mark.parametrize("device", [False, True])
@pytest.mark.parametrize("dataType", [psi1, psi4, psi2, sigma])
def test_eth_and_ladder_factors(ctx, dataType, device):
    o = random_waveform(dataType=dataType, ell_max=6, n=30)
    w = gpu(o, ctx, device)
    for ops, conv in (("+", "NP"), ("-", "NP"), ("-+", "NP"), ("ð̅ð", "GHP"), ([+1, -1, -1], "NP"), ("++", "GHP")):
        assert np.array_equal(w.apply_eth(ops, eth_convention=conv), ref.apply_eth(o, ops, eth_convention=conv)), (ops, conv)
    assert np.array_equal(w.eth, ref.apply_eth(o, "+")) and np.array_equal(w.ethbar, ref.apply_eth(o, "-"))
    for ell in range(0, 6):
        for s in (-2, 0, 1):
            assert w.ladder_factor("+-", s, ell) == ref.ladder_factor("+-", s, ell)
    with pytest.raises(ValueError, match="operations must be"):
        w.apply_eth("+q")
    with pytest.raises(ValueError, match="eth_convention must be one of"):
        w.apply_eth("+", eth_convention="BS")


def test_conjugate_pairs_truncate_and_inner_product(ctx):
    o = random_waveform(dataType=h, ell_max=6, n=60)
    expect = ref.convert_to_conjugate_pairs(o)
    wd = gpu(o, ctx, device=True)
    wd.convert_to_conjugate_pairs()
    assert wd.is_device_resident and np.abs(wd.data - expect.data).max() < 4e-16 * np.abs(o.data).max()
    w = gpu(o, ctx)
    w.convert_to_conjugate_pairs()
    assert np.abs(w.data - expect.data).max() < 4e-16 * np.abs(o.data).max()
    assert np.allclose(w.norm(), ref.norm(o), rtol=1e-14, atol=0)
    w.convert_from_conjugate_pairs()
    assert np.abs(w.data - o.data).max() < 8e-16 * np.abs(o.data).max()
    # truncation: the same bits as the reference's formula
    for tol in (1e-10, 1e-3):
        w = gpu(o, ctx)
        w.truncate(tol)
        assert np.array_equal(w.data, ref.truncate(o, tol).data)
    w = gpu(o, ctx)
    w.truncate(0.0)
    assert np.array_equal(w.data, o.data)
    # inner product
    o2 = random_waveform(dataType=h, ell_max=6, n=60, seed=5)
    a, b = gpu(o, ctx), gpu(o2, ctx)
    got, expect = a.inner_product(b), ref.inner_product(o, o2)
    assert abs(got - expect) < 1e-12 * abs(expect)
    got, expect = a.inner_product(b, t1=2.0, t2=7.5), ref.inner_product(o, o2, t1=2.0, t2=7.5)
    assert abs(got - expect) < 1e-12 * max(1.0, abs(expect))
    assert abs(a.inner_product(a).imag) < 1e-12 * abs(a.inner_product(a).real) and a.inner_product(a).real > 0
    with pytest.raises(ValueError, match="Spin weights must match"):
        a.inner_product(gpu(random_waveform(dataType=psi2, ell_max=6, n=60), ctx))
    short = gpu(random_waveform(dataType=h, ell_max=4, n=60), ctx)
    with pytest.raises(ValueError, match="ell_min and ell_max must match"):
        a.inner_product(short)
    clipped = a.inner_product(short, allow_LM_differ=True)
    o_clip = WM(t=o.t, data=o.data[:, : short.n_modes], ell_min=2, ell_max=4, dataType=h)
    assert abs(clipped - ref.inner_product(o_clip, random_waveform(dataType=h, ell_max=4, n=60))) < 1e-12 * abs(clipped)
    shifted = gpu(WM(t=o2.t * 0.9 + 0.3, data=o2.data, ell_min=2, ell_max=6, dataType=h), ctx)
    with pytest.raises(ValueError, match="Time samples must match"):
        a.inner_product(shifted)
    both = a.inner_product(shifted, allow_times_differ=True)
    from scri_amd.mode_operators import time_intersection
    from scipy.interpolate import CubicSpline

    tc = time_intersection(o.t, shifted.t)
    assert tc[0] == max(o.t[0], shifted.t[0]) and tc[-1] <= min(o.t[-1], shifted.t[-1]) and (np.diff(tc) > 0).all()
    integrand = np.sum(np.conj(CubicSpline(o.t, o.data)(tc)) * CubicSpline(shifted.t, o2.data)(tc), axis=1)
    assert abs(both - CubicSpline(tc, integrand).integrate(tc[0], tc[-1])) < 1e-11 * abs(both)


@pytest.mark.parametrize("n,ell_min,ell_max", [(1, 2, 2), (63, 0, 3), (64, 2, 8), (65, 1, 4), (1000, 2, 16), (130, 0, 0)])
def test_norm_on_the_gpu_adds_in_the_reference_order(ctx, n, ell_min, ell_max):
    """bms_row_norm (weights resident in HBM) against the host path and the oracle's literal loop (complex_array_norm,
    scri/waveform_base.py:19-35): equal to the bit, with and without the square root; a strided device view as well."""
    import torch
    import scri_amd
    from scri_amd import engine

    rng = np.random.default_rng(n + ell_max)
    nm = (ell_max + 1) ** 2 - ell_min**2
    data = (rng.normal(size=(n, nm)) + 1j * rng.normal(size=(n, nm))) * 10.0 ** rng.uniform(-3, 3, size=(n, 1))
    o = WM(t=np.arange(n, dtype=float), data=data, ell_min=ell_min, ell_max=ell_max, dataType=h)
    expect = ref.norm(o)
    w = scri_amd.WaveformModes(t=o.t, data=data.copy(), ell_min=ell_min, ell_max=ell_max, dataType=scri_amd.h, ctx=ctx)
    assert np.array_equal(w.norm(), expect) and np.array_equal(w.norm(take_sqrt=True), np.sqrt(expect))
    assert np.array_equal(engine.row_norm(data, ctx=ctx), expect)  # host array through the kernel
    w.to_device()
    assert w.is_device_resident
    assert np.array_equal(w.norm(), expect) and np.array_equal(w.norm(take_sqrt=True), np.sqrt(expect))
    assert w.is_device_resident  # the norm did not pull the weights off the GPU
    wide = torch.full((n, nm + 5), float("nan"), dtype=torch.complex128, device="cuda")
    wide[:, :nm] = torch.from_numpy(data).cuda()
    assert np.array_equal(engine.row_norm(None, ctx=ctx, device_tensor=wide[:, :nm]), expect)


def test_compare_two_waveforms(ctx):
    """WaveformBase.compare (scri/waveform_base.py:577-687): A - B on the common time axis; the frame takes B's into A's"""
    import scri_amd
    from oracle import quat
    from scri_amd import quaternions

    oa = random_waveform(dataType=h, ell_max=5, n=70, seed=1)
    ob = random_waveform(dataType=h, ell_max=5, n=55, seed=2)
    ob = WM(t=np.linspace(0.5, 9.0, 55), data=ob.data, ell_min=2, ell_max=5, dataType=h, frame=ob.frame)
    # smooth data (a random series is no test of an interpolant): low-order polynomials in t with random coefficients
    ca, cb = oa.data[:4], ob.data[:4]
    oa = WM(t=oa.t, data=sum(ca[k][None, :] * (oa.t[:, None] / 10.0) ** k for k in range(4)), ell_min=2, ell_max=5, dataType=h, frame=oa.frame)
    ob = WM(t=ob.t, data=sum(cb[k][None, :] * (ob.t[:, None] / 10.0) ** k for k in range(4)), ell_min=2, ell_max=5, dataType=h, frame=ob.frame)
    times, expect = ref.compare_data(ob, oa)
    A, B = gpu(oa, ctx), gpu(ob, ctx)
    C = B.compare(A)
    assert np.array_equal(C.t, times) and (C.ell_min, C.ell_max, C.dataType) == (2, 5, B.dataType)
    assert np.abs(C.data - expect).max() < 1e-12 * np.abs(expect).max()
    assert "B.compare(A)\n" in C.history and C.history[-1] == "### End of old histories from `compare`"
    # frames: a smooth rotor series on either side; C.frame * B(t) = A(t) (up to the overall sign the method fixes)
    ta, tb = oa.t, ob.t
    fa = np.stack([np.cos(0.1 * ta), np.sin(0.1 * ta) * 0.6, np.sin(0.1 * ta) * 0.8, 0 * ta], axis=1)
    fb = np.stack([np.cos(0.07 * tb + 0.2), 0 * tb, np.sin(0.07 * tb + 0.2), 0 * tb], axis=1)
    A.frame, B.frame = fa, fb
    C = B.compare(A)
    Ai, Bi = quaternions.squad(fa, ta, times), quaternions.squad(fb, tb, times)
    assert C.frame.shape == (times.size, 4)
    back = quaternions.multiply(C.frame, Bi)
    assert min(np.abs(back - Ai).max(), np.abs(back + Ai).max()) < 1e-12
    assert np.mean(C.frame[:, 0]) > 0
    A.frame, B.frame = fa[:1], fb[:1]
    C = B.compare(A)
    q = quat.qmul(fa[0], quat.qinverse(fb[0]))
    assert C.frame.shape == (1, 4) and np.abs(C.frame[0] - (q if q[0] >= 0 else -q)).max() < 1e-14
    with pytest.raises(Exception, match="mismatched LM data"):
        B.compare(gpu(random_waveform(dataType=h, ell_max=4, n=70), ctx))
    with pytest.warns(UserWarning, match="Comparing them probably does not make sense"):
        other = gpu(oa, ctx)
        other.frameType = scri_amd.Corotating
        B.compare(other)
