"""GPU parity of LdtVector / LLMatrix / angular_velocity (bms_angular_velocity) against the oracle, and the
reference's analytic tests (tests/test_mode_calculations.py:75-110) with the rotation on the GPU as well."""
import math

import numpy as np
import pytest

from oracle import mode_calculations_ref as mc
from oracle import quat

pytestmark = pytest.mark.gpu


def test_parts_match_oracle_on_generic_data(ctx):
    from scri_amd import engine, synthetic

    for ell_min, ell_max, n in ((2, 8, 700), (0, 5, 300), (2, 16, 400)):
        t = np.sort(np.random.default_rng(ell_max).uniform(0, 50, n)) + np.arange(n) * 1e-3
        data = synthetic.chirp_modes(t, ell_min, ell_max, 40 + ell_max) * 10.0 ** (np.arange(1)[:, None])
        ldt, ll, om = engine.angular_velocity(t, data, ell_min, ell_max, ctx=ctx, parts=True)
        dd = mc.data_dot(t, data)
        ldt_ref = mc.LdtVector(data, dd, ell_min, ell_max)
        ll_ref = mc.LLMatrix(data, ell_min, ell_max)
        assert np.abs(ldt - ldt_ref).max() < 1e-11 * max(1.0, np.abs(ldt_ref).max())
        assert np.abs(ll - ll_ref).max() < 1e-13 * max(1.0, np.abs(ll_ref).max())
        assert np.array_equal(ll, np.swapaxes(ll, 1, 2))
        om_ref = mc.angular_velocity(t, data, ell_min, ell_max)
        assert np.abs(om - om_ref).max() < 1e-9 * max(1.0, np.abs(om_ref).max())


def _constant_waveform(n_times, ctx):
    import scri_amd

    t = np.linspace(-10.0, 10.0, n_times)
    LM = np.array([[l, m] for l in range(2, 9) for m in range(-l, l + 1)])
    data = np.repeat((LM[:, 1] - 1j * LM[:, 1])[None, :], n_times, axis=0).astype(complex)
    return scri_amd.WaveformModes(
        t=t, data=data, ell_min=2, ell_max=8, dataType=scri_amd.h, frameType=scri_amd.Inertial, r_is_scaled_out=True,
        m_is_scaled_out=True, ctx=ctx,
    )


def test_reference_angular_velocity_cases(ctx):
    w = _constant_waveform(10000, ctx)
    assert np.allclose(w.angular_velocity(), 0, atol=1e-15, rtol=0)
    omega = 2 * math.pi / 5.0
    half = np.zeros((w.n_times, 4))
    half[:, 3] = omega / 2 * w.t
    for R0 in (np.array([1.0, 0, 0, 0]), np.array([1.0, 2, 3, 4]) / math.sqrt(30)):
        wr = _constant_waveform(10000, ctx)
        wr.rotate_physical_system(quat.qmul(R0[None, :], quat.qexp(half)))
        Om = quat.qmul(quat.qmul(R0, np.array([0, 0, 0, omega])), quat.qinverse(R0))[1:]
        assert np.allclose(wr.angular_velocity(), Om[None, :], atol=1e-12, rtol=2e-8)
    # the derived data of waveform_base.py:689-703
    from oracle import modes_time_series_ref as mref

    assert np.abs(wr.data_dot - mref.interpolate(wr.t, wr.data, wr.t, 1)).max() < 1e-9
    assert np.abs(wr.data_int - mref.interpolate(wr.t, wr.data, wr.t, -1)).max() < 1e-11


def test_reference_corotating_frame_case(ctx):
    """tests/test_mode_calculations.py:113-128: a constant waveform rotated by R_in(t) has corotating frame R_in, and
    to_corotating_frame brings back the constant waveform."""
    import scri_amd

    w = _constant_waveform(100000, ctx)
    omega = 2 * math.pi / 5.0
    R0 = np.array([1.0, 2, 3, 4]) / math.sqrt(30)
    half = np.zeros((w.n_times, 4))
    half[:, 3] = omega / 2 * w.t
    R_in = quat.qmul(R0[None, :], quat.qexp(half))
    w_rot = _constant_waveform(100000, ctx)
    w_rot.rotate_physical_system(R_in)
    R_out = scri_amd.mode_calculations.corotating_frame(w_rot, R0=R_in[0], tolerance=1e-12)
    assert np.allclose(R_in, R_out, atol=1e-10, rtol=0.0)
    w_rot.to_corotating_frame(R0=R_in[0], tolerance=1e-12)
    assert np.allclose(w_rot.data, w.data, atol=1e-8, rtol=1e-5)
    assert w_rot.frameType == scri_amd.Corotating


def _sign_free_distance(a, b):
    return max(np.amin(np.vstack((np.linalg.norm(a - b, axis=1), np.linalg.norm(a + b, axis=1))), axis=0))


def test_reference_dominant_eigenvector_cases(ctx):
    """tests/test_mode_calculations.py:14-71: the principal axis of <LL> is z for the simple waveforms, and rotates with
    the waveform."""
    import scri_amd
    from oracle import sample_waveforms_ref as sw

    Rs = sw.Rs()
    n = len(Rs)
    t = np.linspace(1.0, 100.0, n)
    LM = np.array([[l, m] for l in range(0, 9) for m in range(-l, l + 1)])
    base = (LM[:, 1] - 1j * LM[:, 1]).astype(complex)
    for data in (np.repeat(base[None, :], n, axis=0), base[None, :] * t[:, None]):  # constant_waveform, linear_waveform
        w = scri_amd.WaveformModes(t=t, data=data.copy(), ell_min=0, ell_max=8, dataType=scri_amd.h, frameType=scri_amd.Inertial,
                                   r_is_scaled_out=True, m_is_scaled_out=True, ctx=ctx)
        dpa = w.LLDominantEigenvector()
        expect = np.zeros_like(dpa)
        expect[:, 2] = 1.0
        assert np.allclose(dpa, expect)
        w.rotate_physical_system(Rs)
        z = np.zeros((n, 4))
        z[:, 3] = 1.0
        expected = quat.qmul(quat.qmul(Rs, z), quat.qconj(Rs))[:, 1:]
        assert _sign_free_distance(w.LLDominantEigenvector(), expected) < 1.0e-13
    # z alignment of the corotating frame: a waveform precessing about a tilted axis ends up with that axis along z
    w = _constant_waveform(20000, ctx)
    omega = 2 * math.pi / 5.0
    R0 = np.array([1.0, 2, 3, 4]) / math.sqrt(30)
    half = np.zeros((w.n_times, 4))
    half[:, 3] = omega / 2 * w.t
    w.rotate_physical_system(quat.qmul(R0[None, :], quat.qexp(half)))
    frame = scri_amd.mode_calculations.corotating_frame(w, z_alignment_region=(0.1, 0.9))
    w.rotate_decomposition_basis(frame)
    dpa = w.LLDominantEigenvector()
    assert np.abs(np.abs(dpa[:, 2]) - 1).max() < 1e-9


def test_to_coprecessing_frame(ctx):
    """scri/rotations.py:14-49: a waveform that is simple in a precessing frame (dominated by (2, +-2), constant) is
    rotated into the inertial frame with a precessing, spinning rotor series; to_coprecessing_frame finds a frame in which
    the dominant eigenvector of <LL> is the z axis again, with no angular velocity about that axis (minimal rotation),
    and in which the modes are the original ones up to the phase e^{i m gamma(t)} of a rotation about z."""
    import scri_amd
    from scri_amd import quaternions as Q

    n = 3000
    t = np.linspace(0.0, 200.0, n)
    LM = np.array([[l, m] for l in range(2, 5) for m in range(-l, l + 1)])
    amp = np.zeros(LM.shape[0], dtype=complex)
    for i, (l, m) in enumerate(LM):
        amp[i] = {(2, 2): 1.0, (2, -2): 1.0, (3, 3): 0.1j, (3, -3): 0.1j, (4, 4): 0.03, (4, -4): 0.03}.get((l, m), 0.0)
    data = np.repeat(amp[None, :], n, axis=0)
    w = scri_amd.WaveformModes(t=t, data=data.copy(), ell_min=2, ell_max=4, dataType=scri_amd.h, frameType=scri_amd.Inertial,
                               r_is_scaled_out=True, m_is_scaled_out=True, ctx=ctx)

    def about(axis, angle):
        axis = np.asarray(axis, dtype=float) / np.linalg.norm(axis)
        return np.concatenate([np.cos(angle / 2)[:, None], np.sin(angle / 2)[:, None] * axis[None, :]], axis=-1)

    # cone precession of the axis (tilt 0.3 about y, carried around z) plus a spin about the body's own axis
    R = Q.multiply(Q.multiply(about([0, 0, 1], 0.02 * t), about([0, 1, 0], 0.3 + 0.0 * t)), about([0, 0, 1], 0.15 * t))
    w.rotate_physical_system(R)
    axis_inertial = Q.multiply(Q.multiply(R, np.array([0.0, 0, 0, 1])), Q.conjugate(R))[:, 1:]
    w.to_coprecessing_frame()
    assert w.frameType == scri_amd.Coprecessing and w.frame.shape == (n, 4)
    # w.frame = R^-1 R_c (rotate_physical_system recorded R^-1): R_c is the coprecessing frame relative to the inertial one
    R_c = Q.multiply(R, w.frame)
    # its z axis is the dominant axis ...
    z_frame = Q.multiply(Q.multiply(R_c, np.array([0.0, 0, 0, 1])), Q.conjugate(R_c))[:, 1:]
    assert np.abs(z_frame - axis_inertial).max() < 1e-9
    # ... and it does not rotate about it (the body's own spin of 0.15 rad per unit time is gone)
    omega = Q.angular_velocity(R_c, t)
    assert np.abs(np.sum(omega * z_frame, axis=-1))[100:-100].max() < 1e-8
    assert np.abs(np.sum(Q.angular_velocity(R, t) * axis_inertial, axis=-1)).min() > 0.1
    # ... and in it the waveform is the simple one again, up to a rotation about z
    dpa = w.LLDominantEigenvector()
    assert np.abs(np.abs(dpa[:, 2]) - 1).max() < 1e-9
    assert np.abs(np.abs(w.data) - np.abs(data)).max() < 1e-9
    w.to_inertial_frame()
    assert w.frameType == scri_amd.Inertial


def test_frame_velocity_and_truncated_log_frame(ctx):
    """angular_velocity(include_frame_velocity=True) adds the angular velocity of the recorded frame
    (scri/mode_calculations.py:426-430): in its corotating frame a uniformly rotating waveform has zero mode velocity and
    the frame carries all of it.  to_corotating_frame(truncate_log_frame=True) rotates with exp of the rounded log-frame
    (scri/rotations.py:86-90) and returns it."""
    import scri_amd

    omega = 2 * math.pi / 5.0
    w = _constant_waveform(6000, ctx)
    half = np.zeros((w.n_times, 4))
    half[:, 3] = omega / 2 * w.t
    w.rotate_physical_system(quat.qexp(half))
    # forget the bookkeeping: the same data as an inertial-frame waveform of a rotating system
    w = scri_amd.WaveformModes(t=w.t, data=w.data, ell_min=2, ell_max=8, dataType=scri_amd.h, frameType=scri_amd.Inertial,
                               r_is_scaled_out=True, m_is_scaled_out=True, ctx=ctx)
    total_before = w.angular_velocity()
    w2, om, log_frame = w.to_corotating_frame(return_omega=True, truncate_log_frame=True)
    assert w2 is w and w.frameType == scri_amd.Corotating and log_frame.shape == (w.n_times, 4)
    power_of_2 = 2 ** int(-np.floor(np.log2(2e-12)))
    assert np.array_equal(log_frame * power_of_2, np.round(log_frame * power_of_2))
    assert np.allclose(w.angular_velocity(), 0.0, atol=1e-6) and np.allclose(w.angular_velocity()[50:-50], 0.0, atol=1e-8)
    total = w.angular_velocity(include_frame_velocity=True)
    assert np.allclose(total[50:-50], [0.0, 0.0, omega], atol=1e-7)
    assert np.allclose(total_before, [0.0, 0.0, omega], atol=1e-9, rtol=3e-7)


def test_align_decomposition_frame_to_modes(ctx):
    """scri/rotations.py:114-265 (no test in the reference): after the alignment the decomposition frame at t_fid has its z axis
    along the dominant eigenvector of <LL> on the side of the angular velocity, the (2, 2) and (2, -2) phases cancel, and x is
    nearer to the given direction than to its opposite; the rotor is a constant right factor of the frame."""
    import scri_amd
    from scri_amd import quaternions

    n = 400
    t = np.linspace(0.0, 40.0, n)
    LM = np.array([[l, m] for l in range(2, 5) for m in range(-l, l + 1)])
    rng = np.random.default_rng(8)
    amp = (rng.normal(size=LM.shape[0]) + 1j * rng.normal(size=LM.shape[0])) * 0.02
    amp[(LM[:, 0] == 2) & (LM[:, 1] == 2)] = 1.0 + 0.3j
    amp[(LM[:, 0] == 2) & (LM[:, 1] == -2)] = 1.0 - 0.3j
    data = amp[None, :] * np.exp(-1j * LM[None, :, 1] * (0.4 * t + 0.002 * t**2)[:, None])
    w = scri_amd.WaveformModes(t=t, data=data, ell_min=2, ell_max=4, dataType=scri_amd.h, frameType=scri_amd.Inertial, r_is_scaled_out=True,
                               m_is_scaled_out=True, ctx=ctx)
    tilt = np.array([np.cos(0.35), np.sin(0.35) * 0.6, np.sin(0.35) * 0.8, 0.0])
    w.rotate_decomposition_basis(tilt)  # the orbital axis is no longer z
    w.frame = np.zeros((0, 4))  # ... and that is the inertial frame we start from
    w.to_corotating_frame()
    before_frame = w.frame.copy()
    t_fid = 17.3
    with pytest.raises(ValueError, match="outside the range"):
        w.get_alignment_of_decomposition_frame_to_modes(100.0)
    R_eps = w.get_alignment_of_decomposition_frame_to_modes(t_fid)
    assert abs(np.linalg.norm(R_eps) - 1.0) < 1e-13
    w.align_decomposition_frame_to_modes(t_fid)
    assert np.abs(w.frame - quaternions.multiply(before_frame, R_eps)).max() < 1e-13
    inst = w.copy().interpolate(np.array([t_fid]))
    V = inst.LLDominantEigenvector()[0]
    assert abs(abs(V[2]) - 1.0) < 1e-9 and np.hypot(V[0], V[1]) < 1e-4  # z of the aligned frame
    d22, d2m2 = inst.data[0, inst.index(2, 2)], inst.data[0, inst.index(2, -2)]
    dphi = np.angle(d22) - np.angle(d2m2)
    assert abs((dphi + np.pi) % (2 * np.pi) - np.pi) < 1e-7  # the two phases cancel (a quarter turn about z moves both by pi)
    Rf = inst.frame[0]
    x_axis = quaternions.multiply(quaternions.multiply(Rf, np.array([0.0, 1.0, 0.0, 0.0])), quaternions.conjugate(Rf))
    assert x_axis[1] > 0  # nearer to +x than to -x
    inertial = scri_amd.WaveformModes(t=t, data=data, ell_min=2, ell_max=4, dataType=scri_amd.h, frameType=scri_amd.Inertial, ctx=ctx)
    with pytest.raises(ValueError, match="only takes Waveforms in the"):
        inertial.get_alignment_of_decomposition_frame_to_modes(t_fid)


def test_time_domain_inner_product(ctx):
    """scri/mode_calculations.py:493-533 against scipy's spline integral, along either axis, with and without the conjugation"""
    from scipy.interpolate import CubicSpline
    import scri_amd

    rng = np.random.default_rng(3)
    t = np.sort(rng.uniform(0.0, 20.0, 80))
    t[1:] = np.maximum(t[1:], t[:-1] + 0.01)
    a = np.sin(0.3 * t)[:, None] * (rng.normal(size=5) + 1j * rng.normal(size=5))[None, :]
    b = np.cos(0.2 * t + 0.1)[:, None] * (rng.normal(size=5) + 1j * rng.normal(size=5))[None, :]
    expect = CubicSpline(t, np.conj(a) * b).integrate(t[0], t[-1])
    assert np.abs(scri_amd.inner_product(t, a, b, apply_conjugate=True, ctx=ctx) - expect).max() < 1e-12 * np.abs(expect).max()
    assert np.abs(scri_amd.inner_product(t, np.conj(a), b, ctx=ctx) - expect).max() < 1e-12 * np.abs(expect).max()
    assert np.abs(scri_amd.inner_product(t, np.conj(a).T.copy(), b.T.copy(), axis=1, ctx=ctx) - expect).max() < 1e-12 * np.abs(expect).max()
    real = scri_amd.inner_product(t, a.real[:, 0], b.real[:, 0], ctx=ctx)
    assert np.isrealobj(real) and abs(real - CubicSpline(t, a.real[:, 0] * b.real[:, 0]).integrate(t[0], t[-1])) < 1e-12 * max(1.0, abs(real))
