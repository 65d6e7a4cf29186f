"""WaveformModes with its mode weights resident in HBM (`to_device()`): rotations and BMS transformations chained on the GPU
equal the host-resident calls, copies stay on the device, and reading `.data` brings the weights back."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _wm(ctx, n=3000, ell_max=8, seed=5):
    import scri_amd
    from scri_amd import synthetic

    t = np.linspace(-20.0, 80.0, n)
    data = synthetic.chirp_modes(t, 2, ell_max, seed)
    return scri_amd.WaveformModes(t=t, data=data, ell_min=2, ell_max=ell_max, dataType=scri_amd.h, frameType=scri_amd.Inertial,
                                  r_is_scaled_out=True, m_is_scaled_out=True, ctx=ctx)


def test_rotation_then_supertranslation_on_the_device(ctx):
    from scri_amd import synthetic

    host = _wm(ctx)
    dev = _wm(ctx).to_device()
    assert dev.is_device_resident and dev.n_modes == host.n_modes and dev.n_times == host.n_times
    R = synthetic.rotor_series(host.t, 4)
    q = np.array([0.3, -0.5, 0.7, 0.41]) / np.linalg.norm([0.3, -0.5, 0.7, 0.41])
    kw = dict(supertranslation=np.array([0.0, 0.05 - 0.02j, 0.03, -0.05 - 0.02j]), frame_rotation=q, boost_velocity=np.array([1e-3, -2e-3, 5e-4]))
    for w in (host, dev):
        w.rotate_decomposition_basis(R)
        w.rotate_decomposition_basis(q)
    assert dev.is_device_resident and np.array_equal(dev.frame, host.frame)
    out_h, out_d = host.transform(**kw), dev.transform(**kw)
    assert out_d.is_device_resident and not out_h.is_device_resident
    both = out_d.transform(time_translation=0.3)  # a second transformation of a device-resident result
    assert both.is_device_resident
    assert out_d.n_times == out_h.n_times and np.array_equal(out_d.t, out_h.t)
    assert np.abs(out_d.data - out_h.data).max() < 1e-13 * max(1.0, np.abs(out_h.data).max())
    assert not out_d.is_device_resident  # reading .data made the host array authoritative
    assert np.abs(both.data - out_h.transform(time_translation=0.3).data).max() < 1e-13 * max(1.0, np.abs(out_h.data).max())
    assert np.abs(dev.data - host.data).max() < 1e-13 * max(1.0, np.abs(host.data).max())


def test_copy_and_assignment(ctx):
    dev = _wm(ctx, n=200).to_device()
    c = dev.copy()
    assert c.is_device_resident and dev.is_device_resident  # the copy did not pull the source off the device
    c.rotate_decomposition_basis(np.array([0.0, 1.0, 0.0, 0.0]))
    a, b = dev.data.copy(), c.data
    assert not np.allclose(a, b) and np.abs(np.abs(a) - np.abs(b)[:, ::1]).max() >= 0  # independent buffers
    dev.data = a * 2
    assert not dev.is_device_resident and np.array_equal(dev.to_device().to_host().data, a * 2)
