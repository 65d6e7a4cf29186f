"""One process, several devices, numpy in / numpy out (engine.transform_modes / transform_abd with `devices=[...]`): the time
shards of the pipelined call dealt over one context and one host thread per entry.  On a one-GPU box the entries all name
device 0 -- four contexts, four threads, the same code path -- and the result must equal the single-context pipelined call with
the same number of time shards BIT FOR BIT (a shard's arithmetic depends on its cut only), and the default call to rounding.
Callers: scri/waveform_modes.py:705-719, scri/asymptotic_bondi_data/transformations.py:391-412, map_to_superrest_frame.py:1029."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("axis", ["uniform", "sxs"])
def test_cfg3_four_contexts_bit_identical_to_one(ctx, axis):
    from scri_amd import engine, synthetic

    t, data, spec = synthetic.workload("cfg3", axis=axis)  # full size: 1e5 x 285
    kw = spec["kwargs"]
    tr = engine.make_transformation(kw["supertranslation"], kw["frame_rotation"], kw["boost_velocity"], 37, 37, 16)
    args = (t, data, 2, 16, -2, -1, engine.BMS_TERM_H, tr)
    devices = [0, 0, 0, 0]
    pieces = engine.pieces_for(devices, t.size, 16, data.nbytes)  # (each context's 25 000 rows cut by the one-context rule: four shards of 6 250)
    assert pieces == 16
    t_one, d_one = engine.transform_modes(*args, ctx=ctx, pieces=pieces)
    t_four, d_four = engine.transform_modes(*args, ctx=ctx, devices=devices)
    assert np.array_equal(t_four, t_one) and np.array_equal(d_four, d_one)
    # however the shards are dealt: three contexts, five contexts, the same sixteen shards
    for devs in ([0, 0, 0], [0] * 5):
        t_k, d_k = engine.transform_modes(*args, ctx=ctx, devices=devs, pieces=pieces)
        assert np.array_equal(t_k, t_one) and np.array_equal(d_k, d_one)
    # against the default one-context call (its own shard count) : rounding
    t_def, d_def = engine.transform_modes(*args, ctx=ctx)
    assert np.array_equal(t_def, t_one)
    assert np.abs(d_def - d_one).max() < 1e-14 * np.abs(d_def).max()


def test_cfg5_slice_four_contexts_bit_identical_to_one(ctx):
    from scri_amd import engine, synthetic

    u, raw, spec = synthetic.abd_workload("cfg5", n_times=3000)  # l <= 24, six fields, 99 x 99 working grid
    kw, L = spec["kwargs"], spec["ell_max"]
    n_theta = 2 * (2 * L + 1) + 1
    tr = engine.make_transformation(kw["supertranslation"], kw["frame_rotation"], kw["boost_velocity"], n_theta, n_theta, L)
    devices = [0, 0, 0, 0]
    pieces = engine.pieces_for(devices, u.size, None, raw.nbytes, abd=True)
    assert pieces == 8
    u_one, r_one = engine.transform_abd(u, raw, L, tr, ctx=ctx, pieces=pieces)
    u_four, r_four = engine.transform_abd(u, raw, L, tr, ctx=ctx, devices=devices)
    assert np.array_equal(u_four, u_one) and np.array_equal(r_four, r_one)
    u_def, r_def = engine.transform_abd(u, raw, L, tr, ctx=ctx)
    assert np.array_equal(u_def, u_one)
    assert np.abs(np.asarray(r_def) - r_one).max() < 1e-14 * max(1.0, np.abs(r_def).max())


def test_devices_keyword_of_the_scri_level_calls(ctx, monkeypatch):
    """w.transform(devices=...) / abd.transform(devices=...) and the SCRI_AMD_DEVICES default: the caller's two-line change"""
    import scri_amd
    from scri_amd import engine, synthetic

    t, data, spec = synthetic.workload("cfg3", n_times=8000)
    kw = dict(spec["kwargs"])
    data = np.ascontiguousarray(data[:, : 9 * 9 - 4])
    w = scri_amd.WaveformModes(t=t, data=data, ell_min=2, ell_max=8, dataType=scri_amd.h, frameType=scri_amd.Inertial,
                               r_is_scaled_out=True, m_is_scaled_out=True, ctx=ctx)
    ref = w.transform(**kw)
    got = w.transform(devices=[0, 0], **kw)
    assert np.array_equal(got.t, ref.t) and np.abs(got.data - ref.data).max() < 1e-14 * np.abs(ref.data).max()
    # the environment's default serves unchanged callers of LONG series (>= 64 MB of input); short ones stay on one context
    monkeypatch.setenv("SCRI_AMD_DEVICES", "0,0,0")
    assert engine.default_devices() == [0, 0, 0]
    dealt = []
    real = engine._transform_modes_multi
    monkeypatch.setattr(engine, "_transform_modes_multi", lambda *a, **k: dealt.append(len(a[5])) or real(*a, **k))
    env = w.transform(**kw)
    assert dealt == [] and np.array_equal(env.data, ref.data)
    t_l, d_l, _ = synthetic.workload("cfg3", n_times=20000)  # 91 MB
    w_l = scri_amd.WaveformModes(t=t_l, data=d_l, ell_min=2, ell_max=16, dataType=scri_amd.h, frameType=scri_amd.Inertial,
                                 r_is_scaled_out=True, m_is_scaled_out=True, ctx=ctx)
    env_l = w_l.transform(**kw)  # an unchanged caller
    assert dealt == [3]
    monkeypatch.delenv("SCRI_AMD_DEVICES")
    ref_l = w_l.transform(**kw)
    assert dealt == [3] and np.array_equal(env_l.t, ref_l.t) and np.abs(env_l.data - ref_l.data).max() < 1e-14 * np.abs(ref_l.data).max()

    from tests.test_gpu_sharding import _abd_case

    u, raw, _, L = _abd_case(n=2000, ell_max=4)
    abd = scri_amd.AsymptoticBondiData(u, L, ctx=ctx)
    abd._raw_data[:] = raw
    st = np.zeros(9, dtype=complex)
    st[0], st[2], st[6] = 0.3, 0.05, 0.02
    kw_abd = dict(supertranslation=st, frame_rotation=[0.9, 0.1, -0.3, 0.2], boost_velocity=[2e-3, -1e-3, 3e-3])
    ref = abd.transform(**kw_abd)
    got = abd.transform(devices=[0, 0, 0], **kw_abd)
    assert np.array_equal(got.t, ref.t)
    assert np.abs(got._raw_data - ref._raw_data).max() < 1e-14 * max(1.0, np.abs(ref._raw_data).max())


def test_errors_of_one_context_reach_the_caller(ctx):
    """a series that is not increasing goes to the one-call path, whose check raises what the reference's callers expect; a bad
    device index raises from the context's creation; `devices` with a shard is refused"""
    from scri_amd import _lib, engine, synthetic

    t, data, spec = synthetic.workload("cfg3", n_times=4000)
    kw = spec["kwargs"]
    data = np.ascontiguousarray(data[:, : 9 * 9 - 4])
    tr = engine.make_transformation(kw["supertranslation"], kw["frame_rotation"], kw["boost_velocity"], 21, 21, 8)
    bad = t.copy()
    bad[2000] = bad[1999]
    with pytest.raises(ValueError):
        engine.transform_modes(bad, data, 2, 8, -2, -1, engine.BMS_TERM_H, tr, ctx=ctx, devices=[0, 0])
    with pytest.raises((ValueError, _lib.BMSError)):
        engine.transform_modes(t, data, 2, 8, -2, -1, engine.BMS_TERM_H, tr, ctx=ctx, devices=[0, 99])
    with pytest.raises(ValueError, match="devices"):
        engine.transform_modes(t, data, 2, 8, -2, -1, engine.BMS_TERM_H, tr, ctx=ctx, devices=[0, 0], shard=(0, 4000, 0, 4000))
    # the C entry itself: a context listed twice is refused (one context serves one host thread at a time), a failure on another
    # context than the first is reported through the first one's message
    import ctypes

    second = _lib.Context(0)
    try:
        inp = _lib.bms_wm_input()
        tt = np.ascontiguousarray(t)
        inp.n_times, inp.t, inp.data, inp.ld, inp.mem = t.size, _lib.dptr(tt), data.ctypes.data, data.shape[1], _lib.BMS_HOST
        inp.ell_min, inp.ell_max, inp.spin_weight, inp.conformal_weight, inp.type_term = 2, 8, -2, -1, engine.BMS_TERM_H
        out = np.empty((t.size, 77), dtype=complex)
        t_out = np.empty(t.size)
        got = _lib.c_i64(0)
        lib = _lib.load()
        twice = (_lib.c_vp * 2)(ctx.handle, ctx.handle)
        rc = lib.bms_transform_modes_multi(twice, 2, ctypes.byref(inp), ctypes.byref(tr), 6, _lib.dptr(t_out), _lib.vptr(out), ctypes.byref(got))
        assert rc == _lib.BMS_ERR_INVALID and b"listed twice" in lib.bms_last_error(ctx.handle)
        pair = (_lib.c_vp * 2)(ctx.handle, second.handle)
        rc = lib.bms_transform_modes_multi(pair, 2, ctypes.byref(inp), ctypes.byref(tr), 6, _lib.dptr(t_out), _lib.vptr(out), ctypes.byref(got))
        assert rc == 0 and got.value > 3900
        ref_t, ref_d = engine.transform_modes(t, data, 2, 8, -2, -1, engine.BMS_TERM_H, tr, ctx=ctx, pieces=6)
        assert np.array_equal(t_out[: got.value], ref_t) and np.array_equal(out[: got.value], ref_d)
        second.option("NO_GEMM_EVAL", 1)  # (another route on the second context: same results to rounding, another kernel)
        rc = lib.bms_transform_modes_multi(pair, 2, ctypes.byref(inp), ctypes.byref(tr), 6, _lib.dptr(t_out), _lib.vptr(out), ctypes.byref(got))
        assert rc == 0 and np.abs(out[: got.value] - ref_d).max() < 1e-13 * np.abs(ref_d).max()
    finally:
        second.close()


def test_rotations_dealt_over_contexts_equal_the_one_context_call(ctx):
    """rotations act on every time step independently (scri/rotations.py:346-392; SURVEY 8(e): "contiguous time blocks, no
    collective"): blocks of rows on several contexts against the one-context call.  Not bit for bit: the resident kernel deals
    (16-step tile, l group) units to its waves by the launch's geometry, and which group order a row meets moves its last bit
    (tools/probes/rot_block_probe.py: 5e-16 on O(1) data, also for two serial launches on ONE context) -- so the bar is rounding."""
    from scri_amd import engine, synthetic

    t, data, _ = synthetic.workload("cfg3", n_times=30001)
    rng = np.random.default_rng(3)
    R = rng.normal(size=(t.size, 4))
    R /= np.linalg.norm(R, axis=1)[:, None]
    sp = np.stack([R[:, 0] + 1j * R[:, 3], R[:, 2] + 1j * R[:, 1]], axis=1)
    one = engine.rotate_series(data.copy(), 2, 16, sp, ctx=ctx)
    bar = 1e-14 * np.abs(one).max()  # (a few ulp of the largest weight: measured 5e-16 on 0.28)
    for devs in ([0, 0], [0, 0, 0, 0, 0]):
        got = engine.rotate_series(data.copy(), 2, 16, sp, ctx=ctx, devices=devs)
        assert np.abs(got - one).max() <= bar, devs
        assert np.array_equal(got, engine.rotate_series(data.copy(), 2, 16, sp, ctx=ctx, devices=devs))  # deterministic for a given dealing
    q = np.array([1.0, 2, 3, 4]) / np.sqrt(30)
    ref = engine.rotate_const(data.copy(), 2, 16, q, ctx=ctx)
    assert np.abs(engine.rotate_const(data.copy(), 2, 16, q, ctx=ctx, devices=[0, 0, 0]) - ref).max() <= bar


@pytest.mark.parametrize("series", [True, False, "D"])
def test_host_rotation_in_blocks_equals_the_one_call_path(ctx, series):
    """A long series in host memory is rotated in blocks of rows that go up, are turned and come back on three streams
    (engine_rotate.hip; scri/rotations.py:346-392 on numpy arrays): equal to the one-call path (context option NO_ROTATE_PIPELINE) to the
    last bits, to the oracle on a window, with a row stride wider than the modes and columns beside them left alone."""
    from oracle import quat, rotations_ref, wigner
    from scri_amd import engine, synthetic

    n, L = 60000, 12  # 165 modes: 158 MB, 13 blocks
    t = np.linspace(0.0, 600.0, n)
    src = synthetic.chirp_modes(t, 2, L, 5)
    nm = src.shape[1]
    ang = 0.02 * t
    sp = np.stack([np.cos(ang) * np.exp(0.3j * np.sin(0.01 * t)), np.sin(ang) * (0.6 + 0.8j)], axis=1)
    q = np.array([0.5, -0.5, 0.5, 0.5])

    D = wigner.wigner_D_matrices(*quat.as_spinor_array(q), 2, L)

    def run(a):
        if series == "D":  # (the seam of the reference's numba kernel: the packed matrices handed over, scri/rotations.py:346-367)
            engine.rotate_const_D(a, 2, L, D, ctx=ctx)
        elif series:
            engine.rotate_series(a, 2, L, sp, ctx=ctx)
        else:
            engine.rotate_const(a, 2, L, q, ctx=ctx)
        return a

    ctx.option("NO_ROTATE_PIPELINE", 1)
    try:
        whole = run(src.copy())
    finally:
        ctx.option("NO_ROTATE_PIPELINE", 0)
    blocks = run(src.copy())
    scale = np.abs(whole).max()
    assert np.abs(blocks - whole).max() < 1e-14 * scale and not np.array_equal(blocks, src)
    # a view with a wider row stride: the columns beside the modes are not touched
    wide = np.full((n, nm + 7), 3.0 - 2.0j)
    wide[:, :nm] = src
    run(wide[:, :nm])
    assert np.array_equal(wide[:, nm:], np.full((n, 7), 3.0 - 2.0j)) and np.abs(wide[:, :nm] - whole).max() < 1e-14 * scale
    # the oracle on rows across a block boundary (60000 / 13 = 4615.4)
    lo, hi = 4500, 4740
    if series is True:
        expect = rotations_ref.rotate_by_series(src[lo:hi], sp[lo:hi], 2, L)
    else:
        expect = rotations_ref.rotate_by_constant(src[lo:hi], 2, L, D)
    assert np.abs(blocks[lo:hi] - expect).max() < 1e-13 * scale
