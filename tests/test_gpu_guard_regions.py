"""Guard regions around device-resident buffers: no AddressSanitizer exists for the GPU on this pool, so the entry points that take
device addresses are run on inputs embedded between NaNs (a read outside the rows handed over poisons the result: every spline
solve spreads a NaN over the whole series) and on outputs embedded between sentinels (a write outside the rows the call owns
changes one).  Each case is compared with the same call on host arrays."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

GUARD = 4096  # complex numbers on either side
SENTINEL = 12345.0


def _guarded_input(array):
    import torch

    dev = torch.device("cuda", 0)
    flat = np.ascontiguousarray(array).reshape(-1)
    buf = torch.full((flat.size + 2 * GUARD,), float("nan"), dtype=torch.complex128, device=dev)
    buf[GUARD : GUARD + flat.size] = torch.from_numpy(flat).to(dev)
    return buf, buf.data_ptr() + 16 * GUARD


def _guarded_output(n_complex):
    import torch

    dev = torch.device("cuda", 0)
    buf = torch.full((n_complex + 2 * GUARD,), SENTINEL, dtype=torch.complex128, device=dev)
    return buf, buf.data_ptr() + 16 * GUARD


def _check_guards(buf, n_complex):
    lo = buf[:GUARD].cpu().numpy()
    hi = buf[GUARD + n_complex :].cpu().numpy()
    assert np.all(lo == SENTINEL) and np.all(hi == SENTINEL), "a write outside the output rows"


def _zrot(a):
    return np.array([np.cos(a / 2), 0.0, 0.0, np.sin(a / 2)])


WM_CASES = {
    "boosted (dense product, B-spline on the modes)": dict(boost=[1e-3, 2e-3, -3e-3], rot=[0.9, 0.1, -0.3, 0.2], env={}),
    "boosted, overlapping row tiles of the evaluating product": dict(boost=[1e-3, 2e-3, -3e-3], rot=[0.9, 0.1, -0.3, 0.2], env={"SCRI_AMD_GEMM_EVAL_STEP": "61"}),
    "boosted, two sweeps on the modes": dict(boost=[1e-3, 2e-3, -3e-3], rot=[0.9, 0.1, -0.3, 0.2], env={"SCRI_AMD_TWO_SWEEPS": "1"}),
    "boosted, back substitution on the grid": dict(boost=[1e-3, 2e-3, -3e-3], rot=[0.9, 0.1, -0.3, 0.2], env={"SCRI_AMD_NO_GEMM_EVAL": "1"}),
    "strongly boosted (samples far from their knots: global-memory march)": dict(boost=[0.1, -0.2, 0.15], rot=[0.9, 0.1, -0.3, 0.2], env={}),
    "boost-free (one-kernel separable synthesis with the spline evaluation in it)": dict(boost=[0, 0, 0], rot=[0.9, 0.1, -0.3, 0.2], env={"SCRI_AMD_NO_SMALL_DENSE": "1", "SCRI_AMD_SYNTHESIS_EVAL": "1"}),
    "boost-free (one-kernel separable synthesis, back substitution on the grid)": dict(boost=[0, 0, 0], rot=[0.9, 0.1, -0.3, 0.2], env={"SCRI_AMD_NO_SMALL_DENSE": "1"}),
    "boost-free, two-kernel form": dict(boost=[0, 0, 0], rot=[1.0, 0, 0, 0], env={"SCRI_AMD_NO_SPLIT_SYNTHESIS": "1", "SCRI_AMD_NO_SMALL_DENSE": "1"}),
    "boost-free, small shapes on the evaluating product": dict(boost=[0, 0, 0], rot=[0.9, 0.1, -0.3, 0.2], env={}),
    "axis boost (one-kernel form with the scale)": dict(boost=[0, 0, 0.2], rot=_zrot(0.4), env={"SCRI_AMD_AXIS_BOOST_MIN_WORK": "0"}),
    "slope form": dict(boost=[1e-3, 2e-3, -3e-3], rot=[1.0, 0, 0, 0], env={"SCRI_AMD_NO_BSPLINE": "1"}),
    "slope form, boost-free": dict(boost=[0, 0, 0], rot=[0.9, 0.1, -0.3, 0.2], env={"SCRI_AMD_NO_BSPLINE": "1"}),
    "one-role analysis": dict(boost=[1e-3, 2e-3, -3e-3], rot=[1.0, 0, 0, 0], env={"SCRI_AMD_NO_SPLIT_ANALYSIS": "1"}),
}


@pytest.mark.parametrize("case", list(WM_CASES))
@pytest.mark.parametrize("ell_max,n", [(8, 700), (16, 330), (5, 9)])
def test_transform_modes_between_guards(ctx, monkeypatch, case, ell_max, n, route):
    from scri_amd import engine, synthetic

    spec = WM_CASES[case]
    for k in ("SCRI_AMD_NO_SEPARABLE_SYNTHESIS", "SCRI_AMD_NO_SPLIT_SYNTHESIS", "SCRI_AMD_AXIS_BOOST_MIN_WORK", "SCRI_AMD_NO_BSPLINE",
              "SCRI_AMD_NO_SPLIT_ANALYSIS", "SCRI_AMD_NO_AXIS_BOOST_SEPARABLE", "SCRI_AMD_GEMM_EVAL_STEP", "SCRI_AMD_TWO_SWEEPS", "SCRI_AMD_NO_GEMM_EVAL",
              "SCRI_AMD_NO_SMALL_DENSE", "SCRI_AMD_SYNTHESIS_EVAL"):
        route(k, None)
    for k, v in spec["env"].items():
        route(k, v)
    t = np.linspace(-30.0, 40.0, n)
    data = synthetic.chirp_modes(t, 2, ell_max, 5 + ell_max)
    nm = data.shape[1]
    st = synthetic.real_supertranslation(0.1 * (np.arange(9) - 3.0 + 1j * np.arange(9)[::-1]))
    n_theta = 2 * (ell_max + 2) + 1
    tr = engine.make_transformation(st, spec["rot"], spec["boost"], n_theta, n_theta, ell_max)
    t_ref, d_ref = engine.transform_modes(t, data, 2, ell_max, -2, -1, engine.BMS_TERM_H, tr, ctx=ctx)
    assert d_ref.shape[0] > 0
    src, src_ptr = _guarded_input(data)
    dst, dst_ptr = _guarded_output(n * nm)
    t_out, n_new = engine.transform_modes(t, src_ptr, 2, ell_max, -2, -1, engine.BMS_TERM_H, tr, ctx=ctx, device=True, ld=nm, out_ptr=dst_ptr)
    ctx.synchronize()
    assert n_new == d_ref.shape[0] and np.array_equal(t_out, t_ref)
    got = dst[GUARD : GUARD + n_new * nm].cpu().numpy().reshape(n_new, nm)
    assert np.isfinite(got).all(), "a read outside the input rows"
    assert np.abs(got - d_ref).max() < 1e-13 * max(1.0, np.abs(d_ref).max())
    _check_guards(dst, n * nm)
    assert np.array_equal(src[GUARD : GUARD + n * nm].cpu().numpy().reshape(n, nm), data), "the input rows were modified"


@pytest.mark.parametrize("boost,rot,env", [
    ([1e-3, 2e-3, -3e-3], [0.9, 0.1, -0.3, 0.2], {}),
    ([0, 0, 0], [0.9, 0.1, -0.3, 0.2], {}),
    ([0, 0, 0], [1.0, 0, 0, 0], {"SCRI_AMD_NO_FUSED_ABD_MIX": "1"}),
    ([0, 0, 0.05], _zrot(1.0), {"SCRI_AMD_AXIS_BOOST_MIN_WORK": "0"}),
])
@pytest.mark.parametrize("ell_max,n", [(4, 300), (9, 90), (3, 3)])
def test_transform_abd_between_guards(ctx, monkeypatch, boost, rot, env, ell_max, n, route):
    from scri_amd import engine
    from tests.test_gpu_transform_abd import real_st, smooth_abd

    for k in ("SCRI_AMD_NO_SEPARABLE_SYNTHESIS", "SCRI_AMD_NO_FUSED_ABD_MIX", "SCRI_AMD_AXIS_BOOST_MIN_WORK", "SCRI_AMD_NO_AXIS_BOOST_SEPARABLE"):
        route(k, None)
    for k, v in env.items():
        route(k, v)
    o = smooth_abd(n, ell_max, 40 + n, t0=-1.0 if n < 4 else -15.0, t1=1.0 if n < 4 else 25.0)
    nm = (ell_max + 1) ** 2
    st = np.asarray(real_st(min(ell_max, 2), 3, 1e-3 if n < 4 else 0.05))
    lst = int(round(np.sqrt(st.size))) - 1
    a = st.reshape(-1).copy()  # the ABD flavour wants a real function: impose it as the class does
    for l in range(lst + 1):
        for m in range(0, l + 1):
            ip, im = l * l + l + m, l * l + l - m
            if m == 0:
                a[ip] = a[ip].real
            else:
                a[im] = (-1) ** m * np.conj(a[ip])
    n_theta = 2 * (2 * ell_max) + 1
    tr = engine.make_transformation(a, rot, boost, n_theta, n_theta, ell_max)
    raw = np.ascontiguousarray(o.raw)
    u_ref, d_ref = engine.transform_abd(o.u, raw, ell_max, tr, ctx=ctx)
    if d_ref.shape[1] == 0:
        pytest.skip("empty window")
    src, src_ptr = _guarded_input(raw)
    dst, dst_ptr = _guarded_output(6 * n * nm)
    u_out, n_new = engine.transform_abd(o.u, src_ptr, ell_max, tr, ctx=ctx, device=True, out_ptr=dst_ptr)
    ctx.synchronize()
    assert n_new == d_ref.shape[1] and np.array_equal(u_out, u_ref)
    got = dst[GUARD : GUARD + 6 * n * nm].cpu().numpy().reshape(6, n, nm)
    assert np.isfinite(got[:, :n_new]).all(), "a read outside the input rows"
    assert np.abs(got[:, :n_new] - d_ref).max() < 1e-13 * max(1.0, np.abs(d_ref).max())
    assert np.all(got[:, n_new:] == SENTINEL), "rows beyond the window were written"
    _check_guards(dst, 6 * n * nm)


@pytest.mark.parametrize("ell_min,ell_max,n", [(2, 8, 1000), (0, 16, 333), (2, 24, 47), (0, 30, 17), (3, 5, 1)])
def test_rotations_between_guards(ctx, ell_min, ell_max, n):
    import torch
    from scri_amd import engine

    rng = np.random.default_rng(n + ell_max)
    nm = (ell_max + 1) ** 2 - ell_min**2
    data = rng.normal(size=(n, nm)) + 1j * rng.normal(size=(n, nm))
    q = rng.normal(size=(n, 4))
    q /= np.linalg.norm(q, axis=1)[:, None]
    spinors = np.stack([q[:, 0] + 1j * q[:, 3], q[:, 2] + 1j * q[:, 1]], axis=1)
    expect_series = engine.rotate_series(data.copy(), ell_min, ell_max, spinors, ctx=ctx)
    expect_const = engine.rotate_const(data.copy(), ell_min, ell_max, q[0], ctx=ctx)
    for which, expect in (("series", expect_series), ("const", expect_const)):
        # in place: the buffer is input and output; NaN guards catch reads, and must still be NaN (writes)
        buf, ptr = _guarded_input(data)
        if which == "series":
            sp, sp_ptr = _guarded_input(spinors)
            engine.rotate_device(ptr, n, nm, ell_min, ell_max, spinors_ptr=sp_ptr, ctx=ctx)
        else:
            engine.rotate_device(ptr, n, nm, ell_min, ell_max, quaternion=q[0], ctx=ctx)
        ctx.synchronize()
        got = buf[GUARD : GUARD + n * nm].cpu().numpy().reshape(n, nm)
        assert np.isfinite(got).all(), which
        assert np.abs(got - expect).max() < 1e-13 * ell_max, which
        assert bool(torch.isnan(buf[:GUARD].real).all()) and bool(torch.isnan(buf[GUARD + n * nm :].real).all()), which


@pytest.mark.parametrize("ell_min,ell_max,spin,n", [(0, 5, 0, 200), (2, 8, -2, 77), (1, 3, 1, 9), (0, 12, 2, 33)])
def test_device_series_operators_between_guards(ctx, ell_min, ell_max, spin, n):
    """DeviceModesTimeSeries on a view between NaNs: calculus (spline derivatives of order -2..2, interpolation), the mode-space
    operators (bms_mode_map) and products (bms_grid_multiply) against the host class."""
    from scri_amd import device_series
    from scri_amd.modes_time_series import ModesTimeSeries

    rng = np.random.default_rng(n + ell_max)
    nm = (ell_max + 1) ** 2 - ell_min**2
    t = np.sort(rng.uniform(0.0, 30.0, n))
    t[1:] = np.maximum(t[1:], t[:-1] + 0.02)
    data = rng.normal(size=(n, nm)) + 1j * rng.normal(size=(n, nm))
    for e in range(ell_min, max(abs(spin), ell_min)):  # no modes below |s|
        data[:, e * e - ell_min**2 : (e + 1) ** 2 - ell_min**2] = 0
    host = ModesTimeSeries(data.copy(), t, spin_weight=spin, ell_min=ell_min, ell_max=ell_max)
    buf, _ = _guarded_input(data)
    view = buf[GUARD : GUARD + n * nm].view(n, nm)
    dev = device_series.DeviceModesTimeSeries(view, t, spin, ell_min, ell_max, ctx=ctx)

    def same(a, b, tol=1e-13):
        a, b = np.asarray(a.ndarray), np.asarray(b.ndarray)
        assert a.shape == b.shape and np.isfinite(a).all()
        assert np.abs(a - b).max() <= tol * max(1.0, np.abs(b).max())

    for order in (-2, -1, 0, 1, 2):
        tn = np.linspace(t[1], t[-2], 41)
        same(dev.interpolate(tn, derivative_order=order), host.interpolate(tn, derivative_order=order), 1e-12)
    same(dev.derivative(), host.derivative(), 1e-12)
    for op in ("eth", "ethbar", "bar", "eth_GHP", "ethbar_GHP"):
        same(getattr(dev, op), getattr(host, op))
    same(dev + dev.derivative(), host + host.derivative(), 1e-12)
    same(dev * (0.5 - 1j), host * (0.5 - 1j))
    same(dev.truncate_ell(max(ell_min, abs(spin), ell_max - 1)), host.truncate_ell(max(ell_min, abs(spin), ell_max - 1)))
    same(dev.multiply(dev.bar, truncator=max), host.multiply(host.bar, truncator=max), 1e-12)
    import torch

    assert bool(torch.isnan(buf[:GUARD].real).all()) and bool(torch.isnan(buf[GUARD + n * nm :].real).all())
    assert np.array_equal(view.cpu().numpy(), data)


def test_psi_type_with_device_companions_and_a_shard_between_guards(ctx, monkeypatch, route):
    """psi3 with its psi4 companion from guarded device buffers, whole and as a time shard whose rows start mid-buffer: the
    shard's buffer holds ONLY the rows the plan names, so a read of any other row of the global series lands in the guards."""
    from scri_amd import engine, synthetic

    for k in ("SCRI_AMD_NO_SEPARABLE_SYNTHESIS", "SCRI_AMD_NO_BSPLINE"):
        route(k, None)
    n, ell_max = 1500, 7
    t = np.linspace(0.0, 150.0, n)
    psi3 = synthetic.chirp_modes(t, 1, ell_max, 3)
    psi4 = synthetic.chirp_modes(t, 2, ell_max, 4)
    st = synthetic.real_supertranslation(0.1 * (np.arange(9) - 2.0 + 0.5j * np.arange(9)))
    n_theta = 2 * (ell_max + 2) + 1
    for boost in ([2e-3, -1e-3, 3e-3], [0.0, 0.0, 0.0]):
        tr = engine.make_transformation(st, [0.8, 0.2, -0.5, 0.1], boost, n_theta, n_theta, ell_max)
        aux_h = [(psi4, 2, ell_max, -2, 1.0, 1)]
        t_ref, d_ref = engine.transform_modes(t, psi3, 1, ell_max, -1, -4, engine.BMS_TERM_PSI, tr, aux=aux_h, ctx=ctx)
        n3, n4 = psi3.shape[1], psi4.shape[1]
        s3, p3 = _guarded_input(psi3)
        s4, p4 = _guarded_input(psi4)
        dst, dp = _guarded_output(n * n3)
        t_out, n_new = engine.transform_modes(t, p3, 1, ell_max, -1, -4, engine.BMS_TERM_PSI, tr, aux=[(p4, 2, ell_max, -2, 1.0, 1, n4)], ctx=ctx,
                                              device=True, ld=n3, out_ptr=dp)
        ctx.synchronize()
        got = dst[GUARD : GUARD + n_new * n3].cpu().numpy().reshape(n_new, n3)
        assert n_new == d_ref.shape[0] and np.isfinite(got).all()
        assert np.abs(got - d_ref).max() < 1e-13 * max(1.0, np.abs(d_ref).max())
        _check_guards(dst, n * n3)
        # a shard in the middle of the series
        o0, o1 = 600, 900
        (r0, r1), _ = engine.shard_plan(t, tr, o0, o1)
        s3, p3 = _guarded_input(psi3[r0:r1])
        s4, p4 = _guarded_input(psi4[r0:r1])
        dst, dp = _guarded_output((o1 - o0) * n3)
        t_out, n_new, first = engine.transform_modes(t, p3, 1, ell_max, -1, -4, engine.BMS_TERM_PSI, tr, aux=[(p4, 2, ell_max, -2, 1.0, 1, n4)],
                                                     ctx=ctx, device=True, ld=n3, out_ptr=dp, shard=(r0, r1 - r0, o0, o1))
        ctx.synchronize()
        got = dst[GUARD : GUARD + n_new * n3].cpu().numpy().reshape(n_new, n3)
        i_first = int(np.searchsorted(t_ref, t_out[0] - 1e-9))
        assert n_new > 0 and np.isfinite(got).all()
        assert np.abs(t_ref[i_first : i_first + n_new] - t_out).max() < 1e-12
        assert np.abs(got - d_ref[i_first : i_first + n_new]).max() < 1e-13 * max(1.0, np.abs(d_ref).max())
        _check_guards(dst, (o1 - o0) * n3)


@pytest.mark.parametrize("env", [{}, {"SCRI_AMD_GEMM_EVAL_STEP": "61"}, {"SCRI_AMD_TWO_SWEEPS": "1"}])
@pytest.mark.parametrize("boost_scale", [1.0, 40.0])
def test_h_type_time_shard_through_the_evaluating_product_between_guards(ctx, monkeypatch, env, boost_scale, route):
    """The dense route of the h type (spline solved on the modes in one pass, evaluated in the product's epilogue, straddle kernel)
    on a time shard from the middle of a series whose buffer holds ONLY the planned rows -- the solve kernel's run-in rows, the
    product's row tiles, the staged knot / abscissa windows and the side rows all end at the shard's edges -- and on the whole
    series; a strong boost (samples tens of rows from their knots) as well."""
    from scri_amd import engine, synthetic

    for k in ("SCRI_AMD_GEMM_EVAL_STEP", "SCRI_AMD_TWO_SWEEPS", "SCRI_AMD_NO_GEMM_EVAL", "SCRI_AMD_NO_BSPLINE"):
        route(k, None)
    n, ell_max = 2100, 9
    t = np.linspace(0.0, 210.0, n) + 0.01 * np.sin(np.linspace(0.0, 60.0, n))  # (not quite uniform: the row guess has to be checked)
    data = synthetic.chirp_modes(t, 2, ell_max, 21)
    nm = data.shape[1]
    st = synthetic.real_supertranslation(0.1 * (np.arange(9) - 2.0 + 0.5j * np.arange(9)))
    n_theta = 2 * (ell_max + 2) + 1
    tr = engine.make_transformation(st, [0.8, 0.2, -0.5, 0.1], list(boost_scale * np.array([2e-3, -1e-3, 3e-3])), n_theta, n_theta, ell_max)
    route("SCRI_AMD_NO_GEMM_EVAL", "1")
    t_ref, d_ref = engine.transform_modes(t, data, 2, ell_max, -2, -1, engine.BMS_TERM_H, tr, ctx=ctx)  # back substitution on the grid
    route("SCRI_AMD_NO_GEMM_EVAL", None)
    for k, v in env.items():
        route(k, v)
    scale = max(1.0, np.abs(d_ref).max())
    src, sp = _guarded_input(data)
    dst, dp = _guarded_output(n * nm)
    t_out, n_new = engine.transform_modes(t, sp, 2, ell_max, -2, -1, engine.BMS_TERM_H, tr, ctx=ctx, device=True, ld=nm, out_ptr=dp)
    ctx.synchronize()
    got = dst[GUARD : GUARD + n_new * nm].cpu().numpy().reshape(n_new, nm)
    assert n_new == d_ref.shape[0] and np.array_equal(t_out, t_ref) and np.isfinite(got).all()
    assert np.abs(got - d_ref).max() < 1e-13 * scale
    _check_guards(dst, n * nm)
    for o0, o1 in ((700, 1300), (0, 150), (n - 400, n)):  # (the strong boost trims ~ 270 rows from the end of the output)
        (r0, r1), (w0, w1) = engine.shard_plan(t, tr, o0, o1)
        src, sp = _guarded_input(data[r0:r1])
        dst, dp = _guarded_output((o1 - o0) * nm)
        t_out, n_new, first = engine.transform_modes(t, sp, 2, ell_max, -2, -1, engine.BMS_TERM_H, tr, ctx=ctx, device=True, ld=nm, out_ptr=dp,
                                                     shard=(r0, r1 - r0, o0, o1))
        ctx.synchronize()
        got = dst[GUARD : GUARD + n_new * nm].cpu().numpy().reshape(n_new, nm)
        assert n_new == min(o1, w1) - max(o0, w0) > 50 and first == max(o0, w0) and np.isfinite(got).all()
        assert np.abs(got - d_ref[first - w0 : first - w0 + n_new]).max() < 1e-13 * scale
        _check_guards(dst, (o1 - o0) * nm)


def test_abd_shard_and_column_parts_between_guards(ctx, monkeypatch, route):
    """An AsymptoticBondiData time shard from the middle of the series (its buffer holds only the planned rows of every field) and
    the WaveformModes grid-column parts (every part reads the whole series, writes its contribution to all output rows)."""
    from scri_amd import engine, synthetic
    from tests.test_gpu_transform_abd import smooth_abd

    for k in ("SCRI_AMD_NO_SEPARABLE_SYNTHESIS", "SCRI_AMD_NO_BSPLINE"):
        route(k, None)
    n, ell_max = 1200, 4
    o = smooth_abd(n, ell_max, 9, t0=0.0, t1=240.0)
    nm = (ell_max + 1) ** 2
    n_theta = 2 * (2 * ell_max) + 1
    for boost in ([1e-3, -2e-3, 2e-3], [0.0, 0.0, 0.0]):
        tr = engine.make_transformation(np.array([0.05, 0, 0.01, 0], dtype=complex), [0.8, 0.2, -0.5, 0.1], boost, n_theta, n_theta, ell_max)
        raw = np.ascontiguousarray(o.raw)
        u_ref, d_ref = engine.transform_abd(o.u, raw, ell_max, tr, ctx=ctx)
        o0, o1 = 500, 760
        (r0, r1), _ = engine.shard_plan(o.u, tr, o0, o1)
        src, sp = _guarded_input(raw[:, r0:r1])
        dst, dp = _guarded_output(6 * (o1 - o0) * nm)
        u_out, n_new, first = engine.transform_abd(o.u, sp, ell_max, tr, ctx=ctx, shard=(r0, r1 - r0, o0, o1), device=True, out_ptr=dp)
        ctx.synchronize()
        got = dst[GUARD : GUARD + 6 * (o1 - o0) * nm].cpu().numpy().reshape(6, o1 - o0, nm)
        i_first = int(np.searchsorted(u_ref, u_out[0] - 1e-9))
        assert n_new > 0 and np.isfinite(got[:, :n_new]).all()
        assert np.abs(got[:, :n_new] - d_ref[:, i_first : i_first + n_new]).max() < 1e-13 * max(1.0, np.abs(d_ref).max())
        assert np.all(got[:, n_new:] == SENTINEL)
        _check_guards(dst, 6 * (o1 - o0) * nm)

    n, ell_max = 900, 8
    t = np.linspace(0.0, 90.0, n)
    data = synthetic.chirp_modes(t, 2, ell_max, 6)
    nmw = data.shape[1]
    st = synthetic.real_supertranslation(0.1 * (np.arange(9) + 1.0 + 0j))
    n_theta = 2 * (ell_max + 2) + 1
    tr = engine.make_transformation(st, [0.9, 0.1, -0.3, 0.2], [0.02, -0.01, 0.03], n_theta, n_theta, ell_max)
    t_ref, d_ref = engine.transform_modes(t, data, 2, ell_max, -2, -1, engine.BMS_TERM_H, tr, ctx=ctx)
    total = 0
    for part in range(3):
        src, sp = _guarded_input(data)
        dst, dp = _guarded_output(n * nmw)
        t_out, n_new, first = engine.transform_modes(t, sp, 2, ell_max, -2, -1, engine.BMS_TERM_H, tr, ctx=ctx, device=True, ld=nmw, out_ptr=dp,
                                                     shard=(0, n, 0, n, part, 3))
        ctx.synchronize()
        got = dst[GUARD : GUARD + n_new * nmw].cpu().numpy().reshape(n_new, nmw)
        assert n_new == d_ref.shape[0] and np.isfinite(got).all()
        _check_guards(dst, n * nmw)
        total = total + got
    assert np.abs(total - d_ref).max() < 1e-13 * max(1.0, np.abs(d_ref).max())
