"""GPU parity of the WaveformModes BMS transform (bms_transform_modes) against the CPU oracle."""
import math
import numpy as np
import pytest

from oracle import sample_waveforms_ref as samples, waveform_grid_ref as grid_ref, wigner
from oracle.containers import WM, h, sigma, psi4, psi2, psi3, news, SpinWeights

pytestmark = pytest.mark.gpu


def smooth_waveform(n, ell_max, seed, dataType=h, t0=-20.0, t1=60.0):
    rng = np.random.default_rng(seed)
    s = SpinWeights[dataType]
    ell_min = abs(s)
    t = np.linspace(t0, t1, n)
    LM = wigner.LM_range(ell_min, ell_max)
    a = rng.normal(size=LM.shape[0]) + 1j * rng.normal(size=LM.shape[0])
    phase = 0.05 * t + 2e-4 * t**2
    data = a[None, :] * 10.0 ** (-LM[None, :, 0] / 4.0) * np.exp(1j * LM[None, :, 1] * phase[:, None])
    return WM(t=t, data=data, ell_min=ell_min, ell_max=ell_max, dataType=dataType)


def real_supertranslation(ell_max, seed, scale):
    rng = np.random.default_rng(seed)
    a = scale * (rng.normal(size=(ell_max + 1) ** 2) + 1j * rng.normal(size=(ell_max + 1) ** 2))
    for ell in range(ell_max + 1):
        for m in range(ell + 1):
            ip, im = wigner.LM_index(ell, m, 0), wigner.LM_index(ell, -m, 0)
            a[ip] = (a[ip] + (-1.0) ** m * np.conj(a[im])) / 2
            a[im] = (-1.0) ** m * np.conj(a[ip])
    return a


CASES = [
    dict(),
    dict(time_translation=1.469),
    dict(space_translation=[0.3, -0.1, 0.2]),
    dict(frame_rotation=[1, 2, 3, 4]),
    dict(boost_velocity=[0.01, -0.02, 0.015]),
    dict(supertranslation="st3", frame_rotation=[0.5, -1, 0.3, 2], boost_velocity=[1e-3, 2e-3, -3e-3]),
]


@pytest.mark.parametrize("case", range(len(CASES)))
@pytest.mark.parametrize("dataType", [h, sigma, psi4, news])
def test_transform_matches_oracle(ctx, case, dataType):
    from scri_amd import WaveformModes

    kw = dict(CASES[case])
    if kw.get("supertranslation") == "st3":
        kw["supertranslation"] = real_supertranslation(3, 21, 0.1)
    w = smooth_waveform(600, 6, 100 + case, dataType)
    expect = grid_ref.transform(w, **{k: (np.array(v) if isinstance(v, list) else v) for k, v in kw.items()})
    wg = WaveformModes(t=w.t, data=w.data, ell_min=w.ell_min, ell_max=w.ell_max, dataType=dataType, frameType=1,
                       r_is_scaled_out=True, m_is_scaled_out=True, ctx=ctx)
    got = wg.transform(**kw)
    assert got.t.shape == expect.t.shape
    assert np.abs(got.t - expect.t).max() <= 1e-13
    scale = np.abs(expect.data).max()
    assert np.abs(got.data - expect.data).max() < 1e-12 * max(1.0, scale)


def test_waveform_grid_from_modes_and_to_modes(ctx):
    """WaveformGrid.from_modes and .to_modes as separate steps (scri/waveform_grid.py:331-613, 274-329): the grid against
    the oracle's from_modes, and from_modes(...).to_modes(ell_max) against the fused transform."""
    import scri_amd
    from oracle import waveform_grid_ref as grid_ref
    from oracle.containers import WM, h, psi3, psi4

    rng = np.random.default_rng(3)
    n, ell_max = 300, 5
    t = np.linspace(-5.0, 20.0, n)
    nm = (ell_max + 1) ** 2 - 4
    data = (rng.normal(size=nm) + 1j * rng.normal(size=nm))[None, :] * np.exp(1j * np.outer(0.1 * t + 1e-3 * t**2, np.arange(nm) % 5 - 2))
    kw = dict(supertranslation=real_supertranslation(2, 11, 0.05), frame_rotation=np.array([0.9, 0.1, -0.3, 0.2]) / np.linalg.norm([0.9, 0.1, -0.3, 0.2]),
              boost_velocity=np.array([0.02, -0.01, 0.03]))
    w_o = WM(t=t, data=data, ell_min=2, ell_max=ell_max, dataType=h)
    w_g = scri_amd.WaveformModes(t=t, data=data, ell_min=2, ell_max=ell_max, dataType=scri_amd.h, frameType=scri_amd.Inertial,
                                 r_is_scaled_out=True, m_is_scaled_out=True, ctx=ctx)
    uprm, grid, n_theta, n_phi = grid_ref.from_modes(w_o, **kw)
    g = scri_amd.WaveformGrid.from_modes(w_g, **kw)
    assert (g.n_theta, g.n_phi) == (n_theta, n_phi) and g.data.shape == (uprm.size, n_theta * n_phi)
    assert np.abs(g.t - uprm).max() < 1e-13
    assert np.abs(g.data - grid.reshape(uprm.size, -1)).max() < 2e-12 * np.abs(grid).max()
    fused = w_g.transform(**kw)
    two_steps = g.to_modes(ell_max)
    assert two_steps.ell_min == 2 and two_steps.ell_max == ell_max and np.array_equal(two_steps.t, fused.t)
    assert np.abs(two_steps.data - fused.data).max() < 2e-13 * np.abs(fused.data).max()
    assert np.abs(scri_amd.WaveformGrid.transform(w_g, **kw).data - fused.data).max() == 0.0
    # default ell_max of to_modes comes from the grid size; a psi3 waveform needs its psi4 companion here too
    assert g.to_modes().ell_max == (n_theta - 1) // 2
    w3 = scri_amd.WaveformModes(t=t, data=data[:, :], ell_min=2, ell_max=ell_max, dataType=scri_amd.psi3, frameType=scri_amd.Inertial,
                                r_is_scaled_out=True, m_is_scaled_out=True, ctx=ctx)
    with pytest.raises(ValueError, match="requires information from"):
        scri_amd.WaveformGrid.from_modes(w3, **kw)
    with pytest.raises(TypeError):
        scri_amd.WaveformGrid.from_modes(data, **kw)
    # the two capabilities the reference hangs on WaveformModes (scri/waveform_grid.py:640-641), and a copy of a grid object
    g2 = w_g.to_grid(**kw)
    assert np.array_equal(g2.data, g.data) and np.array_equal(g2.t, g.t) and (g2.n_theta, g2.n_phi) == (g.n_theta, g.n_phi)
    back = scri_amd.WaveformModes.from_grid(g, ell_max)
    assert isinstance(back, scri_amd.WaveformModes) and np.array_equal(back.data, two_steps.data)
    g3 = scri_amd.WaveformGrid(g)
    assert np.array_equal(g3.data, g.data) and g3.data is not g.data and g3.num != g.num and g3.dataType == g.dataType


def test_pipelined_host_path_matches_the_one_call_path(ctx, monkeypatch):
    """Host arrays above engine.PIPELINE_MIN_BYTES go through time shards on two contexts (upload, kernels and download of
    neighbouring shards overlapping); the result is the one-call result to rounding, times included, and a series the
    engine does not shard (graded time steps) quietly takes the one-call path."""
    import scri_amd
    from scri_amd import engine

    rng = np.random.default_rng(5)
    n, ell_max = 6000, 6
    t = np.sort(rng.uniform(0.0, 600.0, n))
    t[1:] = np.maximum(t[1:], t[:-1] + 1e-3)
    nm = (ell_max + 1) ** 2 - 4
    data = rng.standard_normal((n, nm)) + 1j * rng.standard_normal((n, nm))
    kw = dict(
        supertranslation=np.array([0.3, 0.02 - 0.01j, 0.05, -0.02 - 0.01j]), boost_velocity=[0.01, -0.02, 0.015],
        frame_rotation=[0.9, 0.1, -0.3, 0.2],
    )

    def run(times):
        w = scri_amd.WaveformModes(t=times, data=data, ell_min=2, ell_max=ell_max, dataType=scri_amd.h, frameType=scri_amd.Inertial,
                                   r_is_scaled_out=True, m_is_scaled_out=True, ctx=ctx)
        return w.transform(**kw)

    monkeypatch.setenv("SCRI_AMD_NO_PIPELINE", "1")
    plain = run(t)
    monkeypatch.delenv("SCRI_AMD_NO_PIPELINE")
    calls = []
    real = engine._transform_modes_pipelined
    monkeypatch.setattr(engine, "_transform_modes_pipelined", lambda *a, **k: calls.append(1) or real(*a, **k))
    monkeypatch.setattr(engine, "PIPELINE_MIN_BYTES", 1 << 16)
    piped = run(t)
    assert calls == [1]
    assert piped.t.shape == plain.t.shape and np.array_equal(piped.t, plain.t)
    assert np.abs(piped.data - plain.data).max() < 1e-12 * np.abs(plain.data).max()
    # without the boost: every piece rotates its eliminated modes (constant rotor through the context's page-locked ring, no stream
    # synchronisation per piece) and takes the separable synthesis
    boost = kw.pop("boost_velocity")
    piped_free = run(t)
    monkeypatch.setenv("SCRI_AMD_NO_PIPELINE", "1")
    plain_free = run(t)
    monkeypatch.delenv("SCRI_AMD_NO_PIPELINE")
    assert calls == [1, 1] and np.array_equal(piped_free.t, plain_free.t)
    assert np.abs(piped_free.data - plain_free.data).max() < 1e-12 * np.abs(plain_free.data).max()
    kw["boost_velocity"] = boost
    # geometric grading: not shardable -> falls back, same answer as with the pipeline switched off
    kw = dict(supertranslation=kw["supertranslation"])
    tg = np.cumsum(np.concatenate([np.full(3000, 0.1), 0.1 * 1.2 ** np.arange(1, 51), np.full(n - 3050, 0.1 * 1.2**50)]))
    monkeypatch.setenv("SCRI_AMD_NO_PIPELINE", "1")
    plain_g = run(tg)
    monkeypatch.delenv("SCRI_AMD_NO_PIPELINE")
    piped_g = run(tg)
    assert len(calls) == 3
    assert np.array_equal(piped_g.data, plain_g.data)


def test_trailing_dimensions_are_one_engine_call(ctx, monkeypatch):
    """Extra trailing data dimensions (scri/waveform_grid.py:299-308, 574-594) are the column blocks of ONE bms_transform_modes_series
    call -- transformation, time axis, tables and window set up once -- not one engine call per trailing index: the entry points are
    counted, the result is the stack of the single-series transforms bit for bit (psi companions with trailing dimensions included),
    and four series cost less than 1.3 x four single-series calls."""
    import time

    import scri_amd
    from scri_amd import _lib, synthetic

    n, L = 20000, 12
    t, data, spec = synthetic.workload("cfg3", n_times=n)
    kw = dict(spec["kwargs"])
    nm = (L + 1) ** 2 - 4
    base = np.ascontiguousarray(data[:, :nm])
    four = np.stack([base * (1 + 0.25 * k) * np.exp(0.3j * k) for k in range(4)], axis=2)

    def wm(d, dt=scri_amd.h, ell_min=2):
        return scri_amd.WaveformModes(t=t, data=d, ell_min=ell_min, ell_max=L, dataType=dt, frameType=scri_amd.Inertial, r_is_scaled_out=True,
                                      m_is_scaled_out=True, ctx=ctx)

    lib = _lib.load()
    calls = {"series": 0, "single": 0}
    real_series, real_single = lib.bms_transform_modes_series, lib.bms_transform_modes_shard
    # (the single-series calls as ONE engine call each: from 16 MB on a host-memory series is otherwise cut into time shards that overlap
    # the transfers with the kernels, engine.auto_pieces -- equal to rounding, not bit for bit, and another entry point)
    monkeypatch.setenv("SCRI_AMD_NO_PIPELINE", "1")

    class Counting:
        def __init__(self, fn, key):
            self.fn, self.key = fn, key

        def __call__(self, *a):
            calls[self.key] += 1
            return self.fn(*a)

    monkeypatch.setattr(lib, "bms_transform_modes_series", Counting(real_series, "series"))
    monkeypatch.setattr(lib, "bms_transform_modes_shard", Counting(real_single, "single"))
    got = wm(four).transform(**kw)
    assert calls == {"series": 1, "single": 0}
    singles = [wm(np.ascontiguousarray(four[:, :, k])).transform(**kw) for k in range(4)]
    assert calls == {"series": 1, "single": 4}
    assert got.data.shape == singles[0].data.shape + (4,)
    for k in range(4):
        assert np.array_equal(got.t, singles[k].t) and np.array_equal(got.data[:, :, k], singles[k].data), k
    # psi3 with a psi4 companion, both [N, modes, 2]
    psi4 = np.stack([base, 0.5j * base], axis=2)
    m3 = synthetic.chirp_modes(t, 1, L, 77)
    psi3 = np.stack([m3, -0.7 * m3], axis=2)
    calls.update(series=0, single=0)
    both = wm(psi3, scri_amd.psi3, 1).transform(psi4_modes=wm(psi4, scri_amd.psi4), **kw)
    assert calls == {"series": 1, "single": 0}
    for k in range(2):
        one = wm(np.ascontiguousarray(psi3[:, :, k]), scri_amd.psi3, 1).transform(psi4_modes=wm(np.ascontiguousarray(psi4[:, :, k]), scri_amd.psi4), **kw)
        assert np.array_equal(both.data[:, :, k], one.data), k
    monkeypatch.undo()
    monkeypatch.setenv("SCRI_AMD_NO_PIPELINE", "1")
    # cost: four series in one call against four calls
    w4, w1 = wm(four), wm(np.ascontiguousarray(four[:, :, 0]))
    for _ in range(2):
        w4.transform(**kw), w1.transform(**kw)

    def best_of(w, reps=5):  # (the fastest of a few runs: the comparison is of costs, not of the box's noise)
        times = []
        for _ in range(reps):
            t0 = time.perf_counter()
            w.transform(**kw)
            times.append(time.perf_counter() - t0)
        return min(times)

    t_four, t_one = best_of(w4), best_of(w1)
    assert t_four < 1.3 * 4 * t_one, (t_four, t_one)


def test_series_call_with_device_resident_blocks(ctx):
    """bms_transform_modes_series with mem = BMS_DEVICE: the block of series stays in HBM (the reference's layout, trailing index
    fastest), the result is written into the caller's device buffer in the same layout -- bit-identical to the host-memory call."""
    import torch

    from scri_amd import engine, synthetic

    n, L, F = 5000, 8, 3
    t, data, spec = synthetic.workload("cfg3", n_times=n)
    kw = spec["kwargs"]
    nm = (L + 1) ** 2 - 4
    block = np.ascontiguousarray(np.stack([data[:, :nm] * (1 + 0.5 * k) for k in range(F)], axis=2))
    tr = engine.make_transformation(kw["supertranslation"], kw["frame_rotation"], kw["boost_velocity"], 2 * (L + 2) + 1, 2 * (L + 2) + 1, L)
    t_ref, ref = engine.transform_modes_series(t, block, 2, L, -2, -1, engine.BMS_TERM_H, tr, ctx=ctx)
    d_in = torch.from_numpy(block).cuda()
    d_out = torch.full((n, nm, F), float("nan"), dtype=torch.complex128, device="cuda")
    torch.cuda.synchronize()
    t_out, n_new = engine.transform_modes_series(t, d_in.data_ptr(), 2, L, -2, -1, engine.BMS_TERM_H, tr, ctx=ctx, device=True, n_series=F,
                                                 out_ptr=d_out.data_ptr())
    assert n_new == t_ref.size and np.array_equal(t_out, t_ref)
    assert np.array_equal(d_out[:n_new].cpu().numpy(), ref)
    assert torch.equal(d_in.cpu(), torch.from_numpy(block))  # the input block is read only
