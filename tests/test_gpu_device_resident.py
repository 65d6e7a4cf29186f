"""SURVEY 8(f) rank 2, "kept device-resident": AsymptoticBondiData / ModesTimeSeries with their fields in HBM
(scri_amd/device_series.py, bms_mode_map) give what the host-resident objects give -- operator by operator against the
host class (whose own parity with the oracle is tests/test_gpu_series.py / test_gpu_charges.py), and for the transformation
and the charges against the oracle directly."""
import numpy as np
import pytest

from oracle import abd_ref, bms_charges_ref
from oracle.containers import ABD

pytestmark = pytest.mark.gpu


def _abd_pair(ctx, n=400, ell_max=5, seed=3):
    import scri_amd
    from tests.test_gpu_transform_abd import smooth_abd

    o = smooth_abd(n, ell_max, seed)
    o.raw[2, :, 0] -= 40.0 * np.sqrt(4 * np.pi)  # a dominant mass monopole keeps the four-momentum timelike
    host = scri_amd.AsymptoticBondiData(o.u, ell_max, ctx=ctx)
    host._raw_data[:] = o.raw
    dev = host.to_device()
    assert dev.is_device_resident and not host.is_device_resident
    return o, host, dev


def test_mode_space_operators_match_the_host_class(ctx):
    o, host, dev = _abd_pair(ctx)
    for name in ("psi0", "psi1", "psi2", "psi3", "psi4", "sigma"):
        h, d = getattr(host, name), getattr(dev, name)
        assert np.array_equal(d.ndarray, h.ndarray) and d.spin_weight == h.spin_weight
        for op in ("eth", "ethbar", "eth_GHP", "ethbar_GHP", "bar"):
            a, b = getattr(h, op), getattr(d, op)
            assert b.spin_weight == a.spin_weight and b.ell_max == a.ell_max
            assert np.abs(b.ndarray - a.ndarray).max() <= 4e-16 * np.abs(a.ndarray).max(), (name, op)
        assert np.abs((-d).ndarray + h.ndarray).max() == 0.0
        assert np.abs((d * (0.3 - 2j)).ndarray - (h * (0.3 - 2j)).ndarray).max() < 1e-15 * np.abs(h.ndarray).max()
        assert np.abs((1j * d / 3.0).ndarray - (1j * h / 3.0).ndarray).max() < 1e-15 * np.abs(h.ndarray).max()
        assert np.array_equal(d.truncate_ell(2).ndarray, h.truncate_ell(2).ndarray)
        rows = (host.t[:, np.newaxis] * h).ndarray
        assert np.abs((dev.t[:, np.newaxis] * d).ndarray - rows).max() < 1e-15 * np.abs(rows).max()
        for order in (1, 2, -1):
            a, b = h.derivative(order) if order > 0 else h.antiderivative(-order), d.derivative(order) if order > 0 else d.antiderivative(-order)
            assert np.abs(b.ndarray - a.ndarray).max() < 1e-13 * max(1.0, np.abs(a.ndarray).max())
    # spin-0: real / imag; sums over different l ranges; products
    assert np.abs(dev.psi2.real.ndarray - host.psi2.real.ndarray).max() < 1e-15
    assert np.abs(dev.psi2.imag.ndarray - host.psi2.imag.ndarray).max() < 1e-15
    with pytest.raises(ValueError):
        dev.psi1.real
    s_h, s_d = host.psi2.truncate_ell(3) + host.psi2, dev.psi2.truncate_ell(3) + dev.psi2
    assert s_d.ell_max == s_h.ell_max and np.abs(s_d.ndarray - s_h.ndarray).max() < 1e-15
    d_h, d_d = host.sigma - host.psi0.truncate_ell(2), dev.sigma - dev.psi0.truncate_ell(2)
    assert np.abs(d_d.ndarray - d_h.ndarray).max() < 1e-15
    with pytest.raises(ValueError, match="different spin weights"):
        dev.psi2 + dev.psi1
    p_h = host.sigma.multiply(host.sigma.bar.dot, truncator=max)
    p_d = dev.sigma.multiply(dev.sigma.bar.dot, truncator=max)
    assert p_d.spin_weight == 0 and p_d.ell_max == p_h.ell_max
    assert np.abs(p_d.ndarray - p_h.ndarray).max() < 1e-13 * max(1.0, np.abs(p_h.ndarray).max())
    tn = np.linspace(host.t[5], host.t[-7], 123)
    assert np.abs(dev.psi3.interpolate(tn).ndarray - host.psi3.interpolate(tn).ndarray).max() < 1e-13


def test_device_resident_transform_and_charges_match_oracle(ctx):
    o, host, dev = _abd_pair(ctx, n=300, ell_max=4, seed=9)
    kw = dict(supertranslation=np.array([0.0, 0.02 - 0.01j, 0.03, -0.02 - 0.01j]), frame_rotation=np.array([0.4, 1, -2, 0.3]) / np.linalg.norm([0.4, 1, -2, 0.3]),
              boost_velocity=np.array([3e-3, 1e-3, -2e-3]))
    e = abd_ref.transform(o, **kw)
    got = dev.transform(**kw)
    assert got.is_device_resident and got.n_times == e.n_times and np.abs(got.u - e.u).max() < 1e-13
    assert np.abs(got._raw_data - e.raw).max() < 1e-12 * max(1.0, np.abs(e.raw).max())
    ref = host.transform(**kw)
    assert np.abs(got._raw_data - ref._raw_data).max() < 1e-14 * max(1.0, np.abs(ref._raw_data).max())
    # slices, copies, interpolation stay on the device
    part = got[20:200]
    assert part.is_device_resident and np.array_equal(part._raw_data, got._raw_data[:, 20:200])
    assert np.array_equal(got.copy()._raw_data, got._raw_data)
    tn = np.linspace(got.t[3], got.t[-3], 77)
    assert np.abs(got.interpolate(tn)._raw_data - ref.interpolate(tn)._raw_data).max() < 1e-12
    # the charges: device == host, and == oracle (on the oracle's transformed fields)
    psi1, psi2, sigma = e.raw[1], e.raw[2], e.raw[5]
    oracle_values = dict(
        bondi_four_momentum=bms_charges_ref.four_momentum(e.u, psi2, sigma),
        bondi_angular_momentum=bms_charges_ref.angular_momentum(psi1, sigma),
        bondi_CoM_charge=bms_charges_ref.com_charge(psi1, sigma),
        bondi_boost_charge=bms_charges_ref.boost_charge(e.u, psi1, psi2, sigma),
        CWWY_angular_momentum=bms_charges_ref.cwwy_angular_momentum(e.u, psi1, psi2, sigma),
    )
    for name in ("bondi_four_momentum", "bondi_angular_momentum", "bondi_boost_charge", "bondi_CoM_charge", "bondi_dimensionless_spin",
                 "CWWY_angular_momentum"):
        a, b = getattr(ref, name)(), getattr(got, name)()
        assert isinstance(b, np.ndarray) and np.abs(a - b).max() < 5e-13 * max(1.0, np.abs(a).max()), name
        if name in oracle_values:
            c = oracle_values[name]
            assert np.abs(c - b).max() < 1e-10 * max(1.0, np.abs(c).max()), name
    for definition in ("BS", "Moreschi", "G", "GW"):
        a, b = ref.supermomentum(definition), got.supermomentum(definition)
        assert np.abs(a.ndarray - b.ndarray).max() < 1e-12 * max(1.0, np.abs(a.ndarray).max())
    assert np.abs(got.h.data - ref.h.data).max() < 1e-14 * max(1.0, np.abs(ref.h.data).max())
    back = got.to_host()
    assert not back.is_device_resident and np.array_equal(back._raw_data, got._raw_data)


def test_map_to_superrest_frame_from_a_device_resident_object(ctx):
    """The whole frame-fixing workflow with the input object itself in HBM equals the host-resident call (which moves its
    window to the device internally): same transformation, same residuals."""
    import scri_amd
    from tests.test_oracle_charges import kerr_schild_abd

    u = np.linspace(-400, 400, num=1601)
    a = scri_amd.AsymptoticBondiData(u, 6, ctx=ctx)
    a._raw_data[:] = kerr_schild_abd(2.0, 0.456, 6, u)
    st = np.array([0.0, 3e-2 - 1j * 5e-3, 1e-3, -3e-2 - 1j * 5e-3, 2e-4 + 1j * 1e-4, 1j * 3e-3, 1e-2, 1j * 3e-3, 2e-4 - 1j * 1e-4])
    moved = a.transform(supertranslation=st, frame_rotation=np.array([1.0, 2, 3, 4]) / np.sqrt(30), boost_velocity=np.array([2e-4, -3e-5, 2e-4]))
    rec_h, tr_h, err_h = moved.map_to_superrest_frame(t_0=0, padding_time=100)
    rec_d, tr_d, err_d = moved.to_device().map_to_superrest_frame(t_0=0, padding_time=100)
    assert rec_d.is_device_resident and not rec_h.is_device_resident
    assert np.allclose(tr_d.supertranslation, tr_h.supertranslation, atol=1e-13)
    assert np.allclose(tr_d.boost_velocity, tr_h.boost_velocity, atol=1e-13)
    assert np.allclose(np.asarray(err_d), np.asarray(err_h), atol=1e-13)
    assert np.abs(rec_d._raw_data - rec_h._raw_data).max() < 1e-12 * max(1.0, np.abs(rec_h._raw_data).max())


def test_engine_reads_are_ordered_behind_torch_copies(ctx):
    """torch's default stream has the handle 0, which `bms_ctx_set_stream` used to read as "the context's own (non-blocking) stream": the
    kernels of a device-resident series then raced the asynchronous clone / zero_ / copy_ that produced their input.  `attach` now names the
    null stream (`bms_ctx_use_default_stream`); the stress case: multi-GB copies queued on torch's stream with an engine read right behind
    each of them, and the reverse order (an engine write read back by torch at once)."""
    import torch

    from scri_amd import device_series

    dev = device_series.attach(ctx)
    n, ell_max = 1_500_000, 9  # 100 modes x 1.5e6 rows x 16 B = 2.4 GB per buffer
    nm = (ell_max + 1) ** 2
    src = torch.view_as_complex(torch.randn((n, nm, 2), dtype=torch.float64, device=dev))
    t = np.arange(n, dtype=float)
    for rep in range(4):
        dst = torch.empty_like(src)
        dst.zero_()
        dst.copy_(src if rep % 2 == 0 else src.clone())  # asynchronous on torch's stream
        series = device_series.DeviceModesTimeSeries(dst, t, 0, 0, ell_max, ctx=ctx)
        doubled = series * 2.0  # bms_mode_map reads dst at once
        back = (doubled.buf - 2.0 * src).abs().max()  # torch reads the engine's output at once
        assert float(back) == 0.0, rep
        del dst, series, doubled
