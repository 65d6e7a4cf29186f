"""The N > 1 path on CPU: two processes (gloo), time-axis sharding plan from the library's host-side planner,
point-to-point halo exchange of input modes, per-shard computation, reassembly == single-process result.

The per-shard arithmetic here is the ORACLE (tests may use it as the checker; on the GPU box the same
plan/exchange code feeds bms_transform_modes_shard -- see tests/test_gpu_sharding.py and bench.py)."""
import os
import socket

import numpy as np
import pytest


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _oracle_shard(t_global, ext, row0, out_i0, out_i1, kw, ell_max):
    """Outputs with global input index in [out_i0, out_i1), computed from rows [row0, row0+len(ext)) only."""
    from oracle import waveform_grid_ref as grid_ref
    from oracle.containers import WM, h
    from oracle.wigner import constant_from_ell_0_mode

    sub_t = t_global[row0 : row0 + ext.shape[0]]
    out = grid_ref.transform(WM(t=sub_t, data=ext, ell_min=2, ell_max=ell_max, dataType=h), **kw)
    # map output times back to global input indices: u' = (t_i - tt) / gamma
    st = np.asarray(kw.get("supertranslation", np.zeros(4)), dtype=complex)
    tt = constant_from_ell_0_mode(st[0]).real
    v = np.asarray(kw.get("boost_velocity", np.zeros(3)), dtype=float)
    gamma = 1 / np.sqrt(1 - np.dot(v, v))
    uprm_global = (1 / gamma) * (t_global - tt)
    idx = np.searchsorted(uprm_global, out.t - 1e-9)
    assert np.abs(uprm_global[idx] - out.t).max() < 1e-12
    keep = (idx >= out_i0) & (idx < out_i1)
    return idx[keep], out.t[keep], out.data[keep]


def _worker(rank, world, port, n_times, ell_max, tmpdir):
    import torch
    import torch.distributed as dist

    from scri_amd import engine, synthetic, sharding

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        t, _, spec = synthetic.workload("cfg3", n_times=n_times)
        kw = dict(spec["kwargs"])
        kw["boost_velocity"] = np.array([1.0, 2.0, 3.0]) * 1e-3  # visible time skew across the shard boundary
        lst = 2
        n_theta = 2 * (ell_max + lst) + 1
        tr = engine.make_transformation(kw["supertranslation"], kw["frame_rotation"], kw["boost_velocity"], n_theta, n_theta, ell_max)
        have, need, window = sharding.plan(t, tr, world)
        nm = (ell_max + 1) ** 2 - 4
        _, full, _ = synthetic.workload("cfg3", n_times=n_times, rows=have[rank])
        local = torch.from_numpy(np.ascontiguousarray(full[:, :nm]))
        ext = sharding.exchange_halos(local, have[rank], need[rank], have, need)
        # the exchanged rows are exactly the global rows [need0, need1)
        _, ref_rows, _ = synthetic.workload("cfg3", n_times=n_times, rows=need[rank])
        assert np.array_equal(ext.numpy(), ref_rows[:, :nm])
        idx, t_out, data = _oracle_shard(t, ext.numpy(), need[rank][0], have[rank][0], have[rank][1], kw, ell_max)
        np.savez(os.path.join(tmpdir, f"rank{rank}.npz"), idx=idx, t=t_out, data=data, window=np.array(window))
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(600)
def test_two_rank_sharded_transform_equals_global(tmp_path):
    import torch.multiprocessing as mp

    from oracle import waveform_grid_ref as grid_ref
    from oracle.containers import WM, h
    from scri_amd import synthetic

    n_times, ell_max, world = 600, 4, 2
    port = _free_port()
    mp.spawn(_worker, args=(world, port, n_times, ell_max, str(tmp_path)), nprocs=world, join=True)
    parts = [np.load(tmp_path / f"rank{r}.npz") for r in range(world)]
    idx = np.concatenate([p["idx"] for p in parts])
    t_sh = np.concatenate([p["t"] for p in parts])
    d_sh = np.concatenate([p["data"] for p in parts])
    # global single-process result
    t, data, spec = synthetic.workload("cfg3", n_times=n_times)
    kw = dict(spec["kwargs"])
    kw["boost_velocity"] = np.array([1.0, 2.0, 3.0]) * 1e-3
    nm = (ell_max + 1) ** 2 - 4
    ref = grid_ref.transform(WM(t=t, data=data[:, :nm], ell_min=2, ell_max=ell_max, dataType=h), **kw)
    window = parts[0]["window"]
    assert ref.t.size == window[1] - window[0]  # the library's planner reproduces the reference's trimming
    assert np.array_equal(idx, np.arange(window[0], window[1]))  # every output produced exactly once, in order
    assert np.abs(t_sh - ref.t).max() < 1e-13
    assert np.abs(d_sh - ref.data).max() < 1e-13 * max(1.0, np.abs(ref.data).max())
