"""Pins of oracle/mode_calculations_ref.py: the reference's analytic angular-velocity tests
(tests/test_mode_calculations.py:75-110) -- a constant waveform has omega = 0; rigidly rotated about z, or about a
tilted axis, its angular velocity is the rotation's."""
import math

import numpy as np
import pytest

from oracle import mode_calculations_ref as mc
from oracle import quat, rotations_ref
from oracle import sample_waveforms_ref as sw


def _rotated_constant(R0, omega, n_times=10000):  # the sampling of the reference tests
    t = np.linspace(-10.0, 10.0, n_times)
    w = sw.constant_waveform(t=t)
    half = np.zeros((n_times, 4))
    half[:, 3] = omega / 2 * t
    R = quat.qmul(np.asarray(R0, dtype=float)[None, :], quat.qexp(half))
    # rotate_physical_system(R) == rotate_decomposition_basis(~R)   (scri/rotations.py:268-281)
    return t, rotations_ref.rotate_decomposition_basis(w, quat.qconj(R))


def test_zero_angular_velocity():
    t = np.linspace(-10.0, 10.0, 2000)
    w = sw.constant_waveform(t=t)
    om = mc.angular_velocity(t, w.data, w.ell_min, w.ell_max)
    assert np.allclose(om, 0, atol=1e-15, rtol=0)


def test_z_angular_velocity():
    omega = 2 * math.pi / 5.0
    t, w = _rotated_constant([1.0, 0, 0, 0], omega)
    om = mc.angular_velocity(t, w.data, w.ell_min, w.ell_max)
    expect = np.zeros_like(om)
    expect[:, 2] = omega
    assert np.allclose(expect, om, atol=1e-12, rtol=2e-8)


def test_rotated_angular_velocity():
    omega = 2 * math.pi / 5.0
    R0 = np.array([1.0, 2, 3, 4]) / math.sqrt(30)
    t, w = _rotated_constant(R0, omega)
    Om = quat.qmul(quat.qmul(R0, np.array([0, 0, 0, omega])), quat.qinverse(R0))
    om = mc.angular_velocity(t, w.data, w.ell_min, w.ell_max)
    assert np.allclose(om, Om[1:][None, :], atol=1e-12, rtol=2e-8)


def test_host_integrator_matches_dop853_and_the_analytic_frame():
    """bms_integrate_angular_velocity is host code behind the C ABI (no GPU): against the oracle's scipy DOP853
    integration of a precessing angular velocity, and against exp(axis omega t / 2) for a constant one."""
    from scri_amd import engine

    t = np.linspace(-10.0, 10.0, 2001)
    R0 = np.array([1.0, 2, 3, 4]) / math.sqrt(30)
    om = np.stack([0.3 * np.sin(0.4 * t), 0.2 * np.cos(0.3 * t), 1.0 + 0.01 * t], axis=1)
    R = engine.integrate_angular_velocity(t, om, R0)
    assert np.abs(R - mc.integrate_angular_velocity(t, om, R0)).max() < 5e-12
    assert np.abs(np.linalg.norm(R, axis=1) - 1).max() < 1e-15
    omega = 2 * math.pi / 5.0
    Om = quat.qmul(quat.qmul(R0, np.array([0, 0, 0, omega])), quat.qinverse(R0))[1:]
    half = np.zeros((t.size, 4))
    half[:, 3] = omega / 2 * (t - t[0])
    R = engine.integrate_angular_velocity(t, np.repeat(Om[None, :], t.size, axis=0), R0)
    assert np.abs(R - quat.qmul(R0[None, :], quat.qexp(half))).max() < 1e-13
    # coarse sampling of a fast rotation: sub-stepping keeps the tolerance
    tc = np.linspace(0.0, 40.0, 81)
    omc = np.stack([0.5 * np.sin(0.2 * tc), 0.5 * np.cos(0.2 * tc), 3.0 + 0 * tc], axis=1)
    Rc = engine.integrate_angular_velocity(tc, omc, [1.0, 0, 0, 0])
    assert np.abs(Rc - mc.integrate_angular_velocity(tc, omc, [1.0, 0, 0, 0])).max() < 1e-10
    with pytest.raises(ValueError, match="strictly increasing"):
        engine.integrate_angular_velocity(tc[::-1], omc, [1.0, 0, 0, 0])
