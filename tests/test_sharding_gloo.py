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
        # the same exchange into a preallocated buffer that already holds the own rows: only the halos move
        buf = torch.full((need[rank][1] - need[rank][0], nm), np.nan + 0j, dtype=torch.complex128)
        lo = have[rank][0] - need[rank][0]
        buf[lo : lo + local.shape[0]] = local
        ext2 = sharding.exchange_halos(buf[lo : lo + local.shape[0]], have[rank], need[rank], have, need, out=buf)
        assert ext2 is buf and np.array_equal(buf.numpy(), ref_rows[:, :nm])
        # deferred completion (bench.py --overlap-halo): the interior outputs need own rows only and are computed while
        # the halos travel; the edges use the completed buffer; the three pieces are the shard's outputs, in order
        buf[:] = np.nan + 0j
        buf[lo : lo + local.shape[0]] = local
        pending = sharding.exchange_halos(buf[lo : lo + local.shape[0]], have[rank], need[rank], have, need, out=buf, wait=False)
        i0, i1 = have[rank]
        a = i0 if rank == 0 else i0 + 2 * (i0 - need[rank][0]) + 8
        b = i1 if rank == world - 1 else i1 - 2 * (need[rank][1] - i1) - 8
        (n0, n1), _ = engine.shard_plan(t, tr, a, b)
        assert i0 <= n0 and n1 <= i1 and b - a > 50  # the interior really needs no halo row
        mid = _oracle_shard(t, local.numpy(), i0, a, b, kw, ell_max)
        assert pending() is buf and np.array_equal(buf.numpy(), ref_rows[:, :nm])
        left = _oracle_shard(t, buf.numpy(), need[rank][0], i0, a, kw, ell_max)
        right = _oracle_shard(t, buf.numpy(), need[rank][0], b, i1, kw, ell_max)
        idx, t_out, data = _oracle_shard(t, ext.numpy(), need[rank][0], have[rank][0], have[rank][1], kw, ell_max)
        assert np.array_equal(np.concatenate([left[0], mid[0], right[0]]), idx)
        assert np.abs(np.concatenate([left[2], mid[2], right[2]]) - data).max() < 1e-13 * max(1.0, np.abs(data).max())
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


# ------------------------------------------------------------------------------------------------- ABD flavour
def _abd_inputs(n_times, ell_max):
    from oracle.containers import ABD

    rng = np.random.default_rng(21)
    u = np.arange(n_times) * 0.1
    nm = (ell_max + 1) ** 2
    m = np.concatenate([np.arange(-l, l + 1) for l in range(ell_max + 1)])
    raw = np.zeros((6, n_times, nm), dtype=complex)
    ph = 0.05 * u + 2e-4 * u**2
    for f, s in enumerate(ABD.spins):
        a = rng.normal(size=nm) + 1j * rng.normal(size=nm)
        a[: s * s] = 0
        raw[f] = a[None, :] * np.exp(1j * m[None, :] * ph[:, None])
    kw = dict(supertranslation=np.array([0.3, 0, 0.05, 0], dtype=complex), boost_velocity=np.array([2e-3, -1e-3, 3e-3]))
    return u, raw, kw


def _abd_worker(rank, world, port, n_times, ell_max, tmpdir):
    import torch
    import torch.distributed as dist

    from oracle import abd_ref
    from oracle.containers import ABD
    from scri_amd import engine, sharding

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        u, raw, kw = _abd_inputs(n_times, ell_max)
        n_theta = 2 * (2 * ell_max + 1) + 1
        tr = engine.make_transformation(kw["supertranslation"], [1, 0, 0, 0], kw["boost_velocity"], n_theta, n_theta, ell_max)
        have, need, window = sharding.plan(u, tr, world)
        local = torch.from_numpy(np.ascontiguousarray(raw[:, have[rank][0] : have[rank][1]]))
        ext = sharding.exchange_halos(local, have[rank], need[rank], have, need, dim=1)
        assert ext.shape == (6, need[rank][1] - need[rank][0], raw.shape[2])
        assert np.array_equal(ext.numpy(), raw[:, need[rank][0] : need[rank][1]])
        # this shard's outputs from its rows + halo only (the oracle as the per-shard arithmetic)
        sub = abd_ref.transform(ABD(u[need[rank][0] : need[rank][1]], ext.numpy(), ell_max), **kw)
        tt = kw["supertranslation"][0].real / np.sqrt(4 * np.pi)
        gamma = 1 / np.sqrt(1 - np.dot(kw["boost_velocity"], kw["boost_velocity"]))
        uprm = (u - tt) / gamma
        idx = np.searchsorted(uprm, sub.u - 1e-9)
        assert np.abs(uprm[idx] - sub.u).max() < 1e-12
        keep = (idx >= have[rank][0]) & (idx < have[rank][1])
        np.savez(os.path.join(tmpdir, f"abd{rank}.npz"), idx=idx[keep], u=sub.u[keep], raw=sub.raw[:, keep], window=np.array(window))
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(600)
def test_two_rank_sharded_abd_transform_equals_global(tmp_path):
    import torch.multiprocessing as mp

    from oracle import abd_ref
    from oracle.containers import ABD

    n_times, ell_max, world = 400, 2, 2
    port = _free_port()
    mp.spawn(_abd_worker, args=(world, port, n_times, ell_max, str(tmp_path)), nprocs=world, join=True)
    parts = [np.load(tmp_path / f"abd{r}.npz") for r in range(world)]
    u, raw, kw = _abd_inputs(n_times, ell_max)
    ref = abd_ref.transform(ABD(u, raw, ell_max), **kw)
    window = parts[0]["window"]
    idx = np.concatenate([p["idx"] for p in parts])
    assert ref.u.size == window[1] - window[0]
    assert np.array_equal(idx, np.arange(window[0], window[1]))
    assert np.abs(np.concatenate([p["u"] for p in parts]) - ref.u).max() < 1e-13
    got = np.concatenate([p["raw"] for p in parts], axis=1)
    assert np.abs(got - ref.raw).max() < 1e-12 * max(1.0, np.abs(ref.raw).max())


# ------------------------------------------------------------------------------------------------- plan B: grid columns
def _columns_worker(rank, world, port, n_times, ell_max, tmpdir):
    import torch
    import torch.distributed as dist

    from oracle import waveform_grid_ref as grid_ref
    from oracle.containers import WM, h
    from scri_amd import engine, synthetic, sharding

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        t, _, spec = synthetic.workload("cfg3", n_times=n_times)
        kw = dict(spec["kwargs"])
        kw["boost_velocity"] = np.array([0.06, -0.05, 0.06])  # beta = 0.1: time shards would overlap almost entirely
        n_theta = 2 * (ell_max + 2) + 1
        tr = engine.make_transformation(kw["supertranslation"], kw["frame_rotation"], kw["boost_velocity"], n_theta, n_theta, ell_max)
        have, need, window = sharding.plan(t, tr, world)
        assert sharding.choose_partition(have, need) == "columns"
        nm = (ell_max + 1) ** 2 - 4
        _, mine, _ = synthetic.workload("cfg3", n_times=n_times, rows=have[rank])
        full = sharding.replicate_rows(torch.from_numpy(np.ascontiguousarray(mine[:, :nm])), have)
        _, ref_rows, _ = synthetic.workload("cfg3", n_times=n_times)
        assert np.array_equal(full.numpy(), ref_rows[:, :nm])
        # this rank's columns: every world-th pixel of the grid (any partition of the columns sums to the whole)
        w = WM(t=t, data=full.numpy(), ell_min=2, ell_max=ell_max, dataType=h)
        uprm, grid, n_th, n_ph = grid_ref.from_modes(w, **kw)
        mask = (np.arange(n_th * n_ph) % world == rank).reshape(n_th, n_ph)
        contribution = grid_ref.to_modes(uprm, grid * mask[None], -2, ell_max)
        n_new = uprm.size
        assert n_new == window[1] - window[0]
        total, block = sharding.padded_rows(n_new, world)
        buf = torch.zeros((total, contribution.shape[1]), dtype=torch.complex128)
        buf[:n_new] = torch.from_numpy(contribution)
        rows, (r0, r1) = sharding.reduce_scatter_rows(buf, n_new)
        np.savez(os.path.join(tmpdir, f"col{rank}.npz"), r=np.array([r0, r1]), t=uprm[r0:r1], data=rows.numpy())
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(600)
def test_two_rank_column_partition_equals_global(tmp_path):
    import torch.multiprocessing as mp

    from oracle import waveform_grid_ref as grid_ref
    from oracle.containers import WM, h
    from scri_amd import synthetic

    n_times, ell_max, world = 401, 4, 2
    port = _free_port()
    mp.spawn(_columns_worker, args=(world, port, n_times, ell_max, str(tmp_path)), nprocs=world, join=True)
    parts = [np.load(tmp_path / f"col{r}.npz") for r in range(world)]
    t, data, spec = synthetic.workload("cfg3", n_times=n_times)
    kw = dict(spec["kwargs"])
    kw["boost_velocity"] = np.array([0.06, -0.05, 0.06])
    nm = (ell_max + 1) ** 2 - 4
    ref = grid_ref.transform(WM(t=t, data=data[:, :nm], ell_min=2, ell_max=ell_max, dataType=h), **kw)
    assert parts[0]["r"][0] == 0 and parts[0]["r"][1] == parts[1]["r"][0] and parts[1]["r"][1] == ref.t.size
    assert np.array_equal(np.concatenate([p["t"] for p in parts]), ref.t)
    got = np.concatenate([p["data"] for p in parts])
    assert np.abs(got - ref.data).max() < 1e-13 * max(1.0, np.abs(ref.data).max())


def test_choose_partition():
    from scri_amd import sharding

    have = [(0, 100), (100, 200)]
    assert sharding.choose_partition(have, [(0, 110), (90, 200)]) == "rows"
    assert sharding.choose_partition(have, [(0, 140), (90, 200)]) == "columns"
    assert sharding.choose_partition(have, [(0, 0), (100, 100)]) == "rows"  # nothing to produce
    assert sharding.padded_rows(10, 4) == (12, 3)


# ------------------------------------------------------------------------------------------------- the library functions
# scri_amd.sharding.ShardedTransform / transform_modes_sharded / transform_abd_sharded and the `group=` keyword of
# WaveformModes.transform / AsymptoticBondiData.transform: the plan -> halo -> shard call -> row placement that bench.py and a
# multi-GPU caller run.  No GPU here, so the per-shard arithmetic is the oracle (`compute=`; for the class methods the engine's
# two entry points are replaced by oracle-backed stand-ins inside the worker processes): what is tested is everything AROUND
# the shard call.
def _oracle_compute_modes(kw, ell_max):
    def compute(t_global, ext, shard):
        from oracle import waveform_grid_ref as grid_ref
        from oracle.containers import WM, h

        if len(shard) == 6 and shard[5] > 1:  # a part of the grid columns over all times
            w = WM(t=t_global, data=ext, ell_min=2, ell_max=ell_max, dataType=h)
            uprm, grid, n_th, n_ph = grid_ref.from_modes(w, **kw)
            mask = (np.arange(n_th * n_ph) % shard[5] == shard[4]).reshape(n_th, n_ph)
            return uprm, grid_ref.to_modes(uprm, grid * mask[None], -2, ell_max), None
        idx, t_out, data = _oracle_shard(t_global, ext, shard[0], shard[2], shard[3], kw, ell_max)
        return t_out, data, (idx[0] if idx.size else shard[2])

    return compute


def _oracle_compute_abd(kw, ell_max):
    def compute(u_global, ext, shard):
        from oracle import abd_ref
        from oracle.containers import ABD

        row0, n_rows, o0, o1 = shard[:4]
        sub = abd_ref.transform(ABD(u_global[row0 : row0 + n_rows], ext, ell_max), **kw)
        tt = kw["supertranslation"][0].real / np.sqrt(4 * np.pi)
        gamma = 1 / np.sqrt(1 - np.dot(kw["boost_velocity"], kw["boost_velocity"]))
        uprm = (u_global - tt) / gamma
        idx = np.searchsorted(uprm, sub.u - 1e-9)
        keep = (idx >= o0) & (idx < o1)
        return sub.u[keep], sub.raw[:, keep], (idx[keep][0] if keep.any() else o0)

    return compute


def _library_worker(rank, world, port, n_times, ell_max, tmpdir):
    import torch
    import torch.distributed as dist

    import scri_amd
    from scri_amd import engine, synthetic, sharding

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        t, _, spec = synthetic.workload("cfg3", n_times=n_times)
        kw = dict(spec["kwargs"])
        kw["boost_velocity"] = np.array([1.0, 2.0, 3.0]) * 1e-3
        n_theta = 2 * (ell_max + 2) + 1
        nm = (ell_max + 1) ** 2 - 4
        tr = engine.make_transformation(kw["supertranslation"], kw["frame_rotation"], kw["boost_velocity"], n_theta, n_theta, ell_max)
        compute = _oracle_compute_modes(kw, ell_max)
        results = {}
        # (i) near-equal blocks, one call; (ii) the same with the interior transformed under the exchange; (iii) UNEVEN blocks
        uneven = [(0, 250), (250, n_times)]
        for tag, have, overlap in (("even", None, False), ("overlap", None, True), ("uneven", uneven, False)):
            st = sharding.ShardedTransform("modes", t, tr, 2, ell_max, -2, -1, engine.BMS_TERM_H, have=have, overlap=overlap, compute=compute)
            assert st.partition == "rows" and (st.interior is not None) == overlap
            _, mine, _ = synthetic.workload("cfg3", n_times=n_times, rows=st.have[rank])
            mine = np.ascontiguousarray(mine[:, :nm])
            t_out, rows, first = st(mine)
            assert isinstance(rows, np.ndarray) and rows.shape == (t_out.size, nm) and t_out.size == st.n_out_rows
            # a caller that keeps its rows in the exchange buffer: the same result, and a second call reuses the buffer
            view = st.own_rows_view(like=torch.from_numpy(mine))
            view.copy_(torch.from_numpy(mine))
            t2, rows2, first2 = st(view)
            assert first2 == first and np.array_equal(t2, t_out) and torch.equal(rows2, torch.from_numpy(rows))
            results[tag] = (t_out, rows, first)
        # (iv) the one-shot function; (v) the column partition through the same call
        _, mine, _ = synthetic.workload("cfg3", n_times=n_times, rows=sharding.shard_bounds(n_times, world, rank))
        mine = np.ascontiguousarray(mine[:, :nm])
        t_f, rows_f, first_f = sharding.transform_modes_sharded(mine, t, 2, ell_max, -2, -1, engine.BMS_TERM_H, tr, compute=compute)
        assert first_f == results["even"][2] and np.array_equal(rows_f, results["even"][1])
        t_c, rows_c, first_c = sharding.transform_modes_sharded(mine, t, 2, ell_max, -2, -1, engine.BMS_TERM_H, tr, partition="columns", compute=compute)
        results["columns"] = (t_c, rows_c, first_c)

        # (vi) WaveformModes.transform(group=...) on a rank-local series: the engine's entry point replaced by the oracle
        def fake_transform_modes(t_, data, ell_min, ell_max_, s, cw, term, tr_, aux=(), ctx=None, device=False, ld=None, out_ptr=None, shard=None, grid=False):
            assert not device and shard is not None and (ell_min, ell_max_, s, cw, term) == (2, ell_max, -2, -1, engine.BMS_TERM_H)
            return compute(np.asarray(t_), np.asarray(data), tuple(shard))

        engine.transform_modes = fake_transform_modes
        i0, i1 = sharding.shard_bounds(n_times, world, rank)
        w = scri_amd.WaveformModes(t=t[i0:i1], data=mine, ell_min=2, ell_max=ell_max, dataType=scri_amd.h, frameType=scri_amd.Inertial,
                                   r_is_scaled_out=True, m_is_scaled_out=True)
        got = w.transform(group=dist.group.WORLD, **kw)
        assert got.ell_min == 2 and got.ell_max == ell_max and np.array_equal(got.t, results["even"][0]) and np.array_equal(got.data, results["even"][1])
        np.savez(os.path.join(tmpdir, f"lib{rank}.npz"), **{f"{k}_{i}": v for k, r in results.items() for i, v in enumerate(r)})
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(900)
def test_two_rank_library_functions_equal_global(tmp_path):
    import torch.multiprocessing as mp

    from oracle import waveform_grid_ref as grid_ref
    from oracle.containers import WM, h
    from scri_amd import synthetic

    n_times, ell_max, world = 600, 4, 2
    mp.spawn(_library_worker, args=(world, _free_port(), n_times, ell_max, str(tmp_path)), nprocs=world, join=True)
    parts = [np.load(tmp_path / f"lib{r}.npz") for r in range(world)]
    t, data, spec = synthetic.workload("cfg3", n_times=n_times)
    kw = dict(spec["kwargs"])
    kw["boost_velocity"] = np.array([1.0, 2.0, 3.0]) * 1e-3
    nm = (ell_max + 1) ** 2 - 4
    ref = grid_ref.transform(WM(t=t, data=data[:, :nm], ell_min=2, ell_max=ell_max, dataType=h), **kw)
    scale = max(1.0, np.abs(ref.data).max())
    for tag in ("even", "overlap", "uneven", "columns"):
        t_sh = np.concatenate([p[f"{tag}_0"] for p in parts])
        d_sh = np.concatenate([p[f"{tag}_1"] for p in parts])
        firsts = [int(p[f"{tag}_2"]) for p in parts]
        assert t_sh.shape == ref.t.shape and np.abs(t_sh - ref.t).max() < 1e-13, tag
        assert firsts[1] == firsts[0] + parts[0][f"{tag}_0"].size, tag  # consecutive blocks of output rows, in rank order
        assert np.abs(d_sh - ref.data).max() < 1e-13 * scale, tag


def _library_abd_worker(rank, world, port, n_times, ell_max, tmpdir):
    import torch.distributed as dist

    import scri_amd
    from scri_amd import engine, sharding

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        u, raw, kw = _abd_inputs(n_times, ell_max)
        n_theta = 2 * (2 * ell_max + 1) + 1
        tr = engine.make_transformation(kw["supertranslation"], [1, 0, 0, 0], kw["boost_velocity"], n_theta, n_theta, ell_max)
        compute = _oracle_compute_abd(kw, ell_max)
        i0, i1 = sharding.shard_bounds(n_times, world, rank)
        mine = np.ascontiguousarray(raw[:, i0:i1])
        u_out, raw_out, first = sharding.transform_abd_sharded(mine, u, ell_max, tr, compute=compute)
        assert raw_out.shape == (6, u_out.size, (ell_max + 1) ** 2)

        # AsymptoticBondiData.transform(group=...) on the rank-local object
        def fake_transform_abd(u_, raw_, ell_max_, tr_, ctx=None, shard=None, device=False, out_ptr=None):
            assert not device and shard is not None and ell_max_ == ell_max
            return compute(np.asarray(u_), np.asarray(raw_), tuple(shard))

        engine.transform_abd = fake_transform_abd
        abd = scri_amd.AsymptoticBondiData(u[i0:i1], ell_max)
        abd._raw_data[:] = mine
        got = abd.transform(group=dist.group.WORLD, **kw)
        assert got.n_times == u_out.size and np.array_equal(got.t, u_out) and np.array_equal(got._raw_data, raw_out)
        np.savez(os.path.join(tmpdir, f"libabd{rank}.npz"), u=u_out, raw=raw_out, first=first)
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(900)
def test_two_rank_library_abd_equals_global(tmp_path):
    import torch.multiprocessing as mp

    from oracle import abd_ref
    from oracle.containers import ABD

    n_times, ell_max, world = 400, 2, 2
    mp.spawn(_library_abd_worker, args=(world, _free_port(), n_times, ell_max, str(tmp_path)), nprocs=world, join=True)
    parts = [np.load(tmp_path / f"libabd{r}.npz") for r in range(world)]
    u, raw, kw = _abd_inputs(n_times, ell_max)
    ref = abd_ref.transform(ABD(u, raw, ell_max), **kw)
    assert int(parts[1]["first"]) == int(parts[0]["first"]) + parts[0]["u"].size
    assert np.abs(np.concatenate([p["u"] for p in parts]) - ref.u).max() < 1e-13
    got = np.concatenate([p["raw"] for p in parts], axis=1)
    assert got.shape == ref.raw.shape and np.abs(got - ref.raw).max() < 1e-12 * max(1.0, np.abs(ref.raw).max())


def _trimmed_kw():
    from scri_amd import synthetic

    return dict(supertranslation=np.array(synthetic.S9, dtype=complex), frame_rotation=np.array([1.0, 2, 3, 4]) / np.sqrt(30),
                boost_velocity=np.array([1.0, 2.0, 3.0]) * 1e-3)


def _no_output_worker(rank, world, port, n_times, ell_max, tmpdir):
    import torch.distributed as dist

    from scri_amd import engine, synthetic, sharding

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        t, _, _ = synthetic.workload("cfg3", n_times=n_times)
        kw = _trimmed_kw()
        n_theta = 2 * (ell_max + 2) + 1
        nm = (ell_max + 1) ** 2 - 4
        tr = engine.make_transformation(kw["supertranslation"], kw["frame_rotation"], kw["boost_velocity"], n_theta, n_theta, ell_max)
        # four ranks: rank 0's one row lies outside the window of valid outputs (it produces nothing, but rank 1 needs it), rank 2 holds NO rows
        have = [(0, 1), (1, 420), (420, 420), (420, n_times)]
        st = sharding.ShardedTransform("modes", t, tr, 2, ell_max, -2, -1, engine.BMS_TERM_H, have=have, compute=_oracle_compute_modes(kw, ell_max))
        assert st.window[0] >= 1 and st.need[0][0] == st.need[0][1] and st.need[2][0] == st.need[2][1] and st.need[1][0] == 0
        _, mine, _ = synthetic.workload("cfg3", n_times=n_times, rows=have[rank])
        t_out, rows, first = st(np.ascontiguousarray(mine[:, :nm]))
        assert rows.shape == (t_out.size, nm) and t_out.size == st.n_out_rows
        if rank in (0, 2):
            assert t_out.size == 0
        np.savez(os.path.join(tmpdir, f"trim{rank}.npz"), t=t_out, data=rows, first=first, window=np.array(st.window))
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(900)
def test_ranks_without_output_and_without_rows(tmp_path):
    """the window of valid outputs leaves the first rank (one row) without output; another rank owns no rows at all: both
    still take part in the exchange (their neighbours may need their rows) and return empty blocks"""
    import torch.multiprocessing as mp

    from oracle import waveform_grid_ref as grid_ref
    from oracle.containers import WM, h
    from scri_amd import synthetic

    n_times, ell_max, world = 600, 4, 4
    mp.spawn(_no_output_worker, args=(world, _free_port(), n_times, ell_max, str(tmp_path)), nprocs=world, join=True)
    parts = [np.load(tmp_path / f"trim{r}.npz") for r in range(world)]
    t, data, _ = synthetic.workload("cfg3", n_times=n_times)
    nm = (ell_max + 1) ** 2 - 4
    ref = grid_ref.transform(WM(t=t, data=data[:, :nm], ell_min=2, ell_max=ell_max, dataType=h), **_trimmed_kw())
    window = parts[0]["window"]
    assert ref.t.size == window[1] - window[0] and window[0] >= 1
    t_sh = np.concatenate([p["t"] for p in parts])
    d_sh = np.concatenate([p["data"] for p in parts])
    assert np.abs(t_sh - ref.t).max() < 1e-13
    assert np.abs(d_sh - ref.data).max() < 1e-13 * max(1.0, np.abs(ref.data).max())
    assert int(parts[1]["first"]) == window[0]
