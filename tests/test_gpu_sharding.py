"""bms_transform_modes_shard on the GPU: shards computed one after the other on one device (rows + halo as
the planner prescribes) reassemble to the unsharded result; chunked work space gives the same result too."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _case(n_times=6000, ell_max=8):
    from scri_amd import engine, synthetic

    t, data, spec = synthetic.workload("cfg3", n_times=n_times)
    kw = dict(spec["kwargs"])
    kw["boost_velocity"] = np.array([1.0, 2.0, 3.0]) * 1e-3
    nm = (ell_max + 1) ** 2 - 4
    n_theta = 2 * (ell_max + 2) + 1
    tr = engine.make_transformation(kw["supertranslation"], kw["frame_rotation"], kw["boost_velocity"], n_theta, n_theta, ell_max)
    return t, np.ascontiguousarray(data[:, :nm]), tr, ell_max


def test_shards_reassemble_to_unsharded(ctx):
    from scri_amd import engine, sharding

    t, data, tr, ell_max = _case()
    t_ref, d_ref = engine.transform_modes(t, data, 2, ell_max, -2, -1, engine.BMS_TERM_H, tr, ctx=ctx)
    for world in (2, 5):
        have, need, window = sharding.plan(t, tr, world)
        ts, ds, firsts = [], [], []
        for r in range(world):
            ext = data[need[r][0] : need[r][1]]
            to, do, first = engine.transform_modes(
                t, ext, 2, ell_max, -2, -1, engine.BMS_TERM_H, tr, ctx=ctx, shard=(need[r][0], ext.shape[0], have[r][0], have[r][1])
            )
            ts.append(to), ds.append(do), firsts.append(first)
        assert firsts[0] == window[0]
        assert np.array_equal(np.concatenate(ts), t_ref)
        assert np.abs(np.concatenate(ds) - d_ref).max() < 1e-14 * max(1.0, np.abs(d_ref).max())


def test_column_parts_sum_to_unsharded(ctx):
    """Plan B of SURVEY 8(e): each part synthesises/splines its own grid columns over all times and returns its
    contribution to the modes; the contributions add up to the unsharded result."""
    from scri_amd import engine

    t, data, tr, ell_max = _case()
    n = t.size
    t_ref, d_ref = engine.transform_modes(t, data, 2, ell_max, -2, -1, engine.BMS_TERM_H, tr, ctx=ctx)
    scale = max(1.0, np.abs(d_ref).max())
    for parts in (2, 3, 40):  # 40 > number of 64-column tiles: some parts are empty
        total = 0
        for p in range(parts):
            to, do, first = engine.transform_modes(t, data, 2, ell_max, -2, -1, engine.BMS_TERM_H, tr, ctx=ctx, shard=(0, n, 0, n, p, parts))
            assert np.array_equal(to, t_ref)
            total = total + do
        assert np.abs(total - d_ref).max() < 2e-14 * scale
    # nothing special about a single part of one
    to, do, _ = engine.transform_modes(t, data, 2, ell_max, -2, -1, engine.BMS_TERM_H, tr, ctx=ctx, shard=(0, n, 0, n, 0, 1))
    assert np.array_equal(do, d_ref)
    with pytest.raises(ValueError, match="column part"):
        engine.transform_modes(t, data, 2, ell_max, -2, -1, engine.BMS_TERM_H, tr, ctx=ctx, shard=(0, n, 0, n, 3, 3))


@pytest.mark.parametrize("n_theta_extra", [0, 12])
def test_column_parts_strong_boost_and_psi_terms(ctx, n_theta_extra):
    """beta = 0.1: the case plan B exists for (the row halo of a time shard would span most of the series); a psi3
    waveform with its psi4 companion exercises the per-column tables of the mixing stage.  The larger grid takes the
    unsorted column order (no pole-ring merging) through the same code."""
    from scri_amd import engine, sharding

    rng = np.random.default_rng(5)
    n, ell_max = 3000, 6
    t = np.arange(n) * 0.1
    m3 = np.concatenate([np.arange(-l, l + 1) for l in range(1, ell_max + 1)])
    m4 = np.concatenate([np.arange(-l, l + 1) for l in range(2, ell_max + 1)])
    ph = 0.05 * t + 2e-5 * t**2
    psi3 = (rng.normal(size=m3.size) + 1j * rng.normal(size=m3.size))[None, :] * np.exp(1j * m3[None, :] * ph[:, None])
    psi4 = (rng.normal(size=m4.size) + 1j * rng.normal(size=m4.size))[None, :] * np.exp(1j * m4[None, :] * ph[:, None])
    st = np.zeros(9, dtype=complex)
    st[0], st[2], st[6] = 0.4, 0.1, 0.05
    n_theta = 2 * (ell_max + 2) + 1 + n_theta_extra
    tr = engine.make_transformation(st, [0.9, 0.1, -0.3, 0.2], [0.06, -0.05, 0.06], n_theta, n_theta, ell_max)
    aux = [(psi4, 2, ell_max, -2, 1.0, 1)]
    args = (t, psi3, 1, ell_max, -1, -4, engine.BMS_TERM_PSI, tr)
    t_ref, d_ref = engine.transform_modes(*args, ctx=ctx, aux=aux)
    have, need, _ = sharding.plan(t, tr, 4)
    assert sharding.choose_partition(have, need) == "columns"
    total = 0
    for p in range(4):
        to, do, _ = engine.transform_modes(*args, ctx=ctx, aux=aux, shard=(0, n, 0, n, p, 4))
        total = total + do
    assert np.array_equal(to, t_ref)
    assert np.abs(total - d_ref).max() < 2e-14 * max(1.0, np.abs(d_ref).max())
    # a column part of a time block: rows [a, b) of the contributions
    a, b = 1000, 1700
    part = 0
    for p in range(2):
        to, do, first = engine.transform_modes(*args, ctx=ctx, aux=aux, shard=(0, n, a, b, p, 2))
        part = part + do
    i0 = np.searchsorted(t_ref, to[0] - 1e-9)
    assert np.abs(part - d_ref[i0 : i0 + part.shape[0]]).max() < 2e-14 * max(1.0, np.abs(d_ref).max())


def test_insufficient_halo_is_rejected(ctx):
    from scri_amd import engine, sharding

    t, data, tr, ell_max = _case()
    have, need, _ = sharding.plan(t, tr, 2)
    with pytest.raises(ValueError, match="halo too small"):
        engine.transform_modes(
            t, data[have[1][0] : have[1][1]], 2, ell_max, -2, -1, engine.BMS_TERM_H, tr, ctx=ctx,
            shard=(have[1][0], have[1][1] - have[1][0], have[1][0], have[1][1]),
        )


def test_chunked_workspace_equals_single_chunk():
    from scri_amd import _lib, engine

    t, data, tr, ell_max = _case()
    big = _lib.Context(0)
    small = _lib.Context(0, workspace_limit=40 << 20)  # forces several chunks of the time axis
    t1, d1 = engine.transform_modes(t, data, 2, ell_max, -2, -1, engine.BMS_TERM_H, tr, ctx=big)
    t2, d2 = engine.transform_modes(t, data, 2, ell_max, -2, -1, engine.BMS_TERM_H, tr, ctx=small)
    assert np.array_equal(t1, t2)
    assert np.abs(d1 - d2).max() < 1e-14 * max(1.0, np.abs(d1).max())
    big.close(), small.close()


# ------------------------------------------------------------------------------------------------- ABD flavour (cfg5)
def _abd_case(n=3000, ell_max=4):
    from scri_amd import engine

    rng = np.random.default_rng(77)
    u = np.arange(n) * 0.1
    nm = (ell_max + 1) ** 2
    spins = (2, 1, 0, -1, -2, 2)
    raw = np.zeros((6, n, nm), dtype=complex)
    ph = 0.05 * u + 2e-5 * u**2
    for f, s in enumerate(spins):
        a = rng.normal(size=nm) + 1j * rng.normal(size=nm)
        a[: s * s] = 0
        m = np.concatenate([np.arange(-l, l + 1) for l in range(ell_max + 1)])
        raw[f] = a[None, :] * np.exp(1j * m[None, :] * ph[:, None])
    st = np.zeros(9, dtype=complex)
    st[0], st[2], st[6] = 0.3, 0.05, 0.02  # real supertranslation (m = 0 components)
    n_theta = 2 * (2 * ell_max + 2) + 1
    tr = engine.make_transformation(st, [0.9, 0.1, -0.3, 0.2], [2e-3, -1e-3, 3e-3], n_theta, n_theta, ell_max)
    return u, raw, tr, ell_max


def test_abd_shards_reassemble_to_unsharded(ctx):
    from scri_amd import engine, sharding

    u, raw, tr, ell_max = _abd_case()
    u_ref, r_ref = engine.transform_abd(u, raw, ell_max, tr, ctx=ctx)
    for world in (2, 3):
        have, need, window = sharding.plan(u, tr, world)
        us, rs, firsts = [], [], []
        for r in range(world):
            ext = np.ascontiguousarray(raw[:, need[r][0] : need[r][1]])
            uo, ro, first = engine.transform_abd(u, ext, ell_max, tr, ctx=ctx, shard=(need[r][0], ext.shape[1], have[r][0], have[r][1]))
            us.append(uo), rs.append(ro), firsts.append(first)
        assert firsts[0] == window[0]
        assert np.array_equal(np.concatenate(us), u_ref)
        assert np.abs(np.concatenate(rs, axis=1) - r_ref).max() < 1e-14 * max(1.0, np.abs(r_ref).max())


@pytest.mark.parametrize("ell_max", [4, 9])  # 9: a 39 x 39 grid, past the fused analysis (columns in grid order)
def test_abd_column_parts_sum_to_unsharded(ctx, ell_max):
    """Plan B for the six AsymptoticBondiData fields: per-part contributions add up to the unsharded transformation."""
    from scri_amd import engine

    u, raw, tr, ell_max = _abd_case(n=1500, ell_max=ell_max)
    n = u.size
    u_ref, r_ref = engine.transform_abd(u, raw, ell_max, tr, ctx=ctx)
    scale = max(1.0, np.abs(r_ref).max())
    for parts in (2, 5):
        total = 0
        for p in range(parts):
            uo, ro, first = engine.transform_abd(u, raw, ell_max, tr, ctx=ctx, shard=(0, n, 0, n, p, parts))
            assert np.array_equal(uo, u_ref)
            total = total + ro[:, : u_ref.size]
        assert np.abs(total - r_ref).max() < 5e-14 * scale
    # a column part of a block of output rows
    a, b = 400, 900
    part = 0
    for p in range(3):
        uo, ro, first = engine.transform_abd(u, raw, ell_max, tr, ctx=ctx, shard=(0, n, a, b, p, 3))
        part = part + ro[:, : uo.size]
    i0 = np.searchsorted(u_ref, uo[0] - 1e-9)
    assert np.abs(part - r_ref[:, i0 : i0 + uo.size]).max() < 5e-14 * scale


def test_abd_shard_with_insufficient_halo_is_rejected(ctx):
    from scri_amd import engine, sharding

    u, raw, tr, ell_max = _abd_case()
    have, need, _ = sharding.plan(u, tr, 2)
    own = np.ascontiguousarray(raw[:, have[1][0] : have[1][1]])
    with pytest.raises(ValueError, match="halo too small"):
        engine.transform_abd(u, own, ell_max, tr, ctx=ctx, shard=(have[1][0], own.shape[1], have[1][0], have[1][1]))


def test_abd_device_resident_equals_host(ctx):
    import torch

    from scri_amd import engine

    u, raw, tr, ell_max = _abd_case(n=1200)
    u_ref, r_ref = engine.transform_abd(u, raw, ell_max, tr, ctx=ctx)
    d_in = torch.from_numpy(raw).cuda()
    n_out = (ell_max + 1) ** 2
    d_out = torch.zeros((6, u.size, n_out), dtype=torch.complex128, device="cuda")
    torch.cuda.synchronize()
    u_out, n_new = engine.transform_abd(u, d_in.data_ptr(), ell_max, tr, ctx=ctx, device=True, out_ptr=d_out.data_ptr())
    assert n_new == u_ref.size and np.array_equal(u_out, u_ref)
    assert np.array_equal(d_out[:, :n_new].cpu().numpy(), r_ref)


def test_sharding_helpers_on_rccl_single_rank(tmp_path):
    """The collectives of scri_amd.sharding on the real backend (nccl = RCCL) with device tensors: a one-rank group is all
    a single-GPU box allows, but it exercises the complex-as-real views, the padded reduce-scatter and the preallocated
    halo buffer on the RCCL code path (the multi-rank logic is covered by the gloo tests)."""
    import subprocess
    import sys

    code = r'''
import os, sys
sys.path.insert(0, %r)
import numpy as np, torch, torch.distributed as dist
from scri_amd import sharding
os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = "29533"
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
x = (torch.arange(70, dtype=torch.float64).reshape(10, 7) * (1 + 2j)).cuda()
have = [(0, 10)]
full = sharding.replicate_rows(x, have)
assert torch.equal(full, x)
total, block = sharding.padded_rows(10, 1)
rows, (r0, r1) = sharding.reduce_scatter_rows(x.clone(), 10)
assert (r0, r1) == (0, 10) and torch.equal(rows, x)
buf = torch.empty_like(x); buf[:] = x
ext = sharding.exchange_halos(buf, have[0], have[0], have, have, out=buf)
assert ext is buf and torch.equal(buf, x)
y = torch.stack([x, 2 * x])
assert torch.equal(sharding.replicate_rows(y, have, dim=1), y)
dist.destroy_process_group()
print("ok")
''' % __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__)))
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and "ok" in out.stdout.split(), out.stderr[-2000:]  # (RCCL prints its banner at exit)


def test_interior_and_edges_equal_the_one_call_shard(ctx):
    """bench.py --overlap-halo: a rank's outputs as three engine calls -- the interior from its own rows only (what runs
    while the halos travel), then the two edges from the completed rows -- equal the one-call shard bit for bit in the
    times and to rounding in the data."""
    from scri_amd import engine, sharding, synthetic

    t, data, spec = synthetic.workload("cfg3", n_times=6000)
    kw = dict(spec["kwargs"])
    kw["boost_velocity"] = np.array([1.0, 2.0, 3.0]) * 1e-3
    data = data[:, : 9 * 9 - 4]
    tr = engine.make_transformation(kw["supertranslation"], kw["frame_rotation"], kw["boost_velocity"], 21, 21, 8)
    have, need, window = sharding.plan(t, tr, 3)
    r = 1
    i0, i1 = have[r]
    ext = data[need[r][0] : need[r][1]]
    t_ref, d_ref, first = engine.transform_modes(t, ext, 2, 8, -2, -1, engine.BMS_TERM_H, tr, ctx=ctx, shard=(need[r][0], ext.shape[0], i0, i1))
    a, b = i0 + 2 * (i0 - need[r][0]) + 8, i1 - 2 * (need[r][1] - i1) - 8
    (n0, n1), _ = engine.shard_plan(t, tr, a, b)
    assert i0 <= n0 and n1 <= i1 and b - a > 1000
    own = data[i0:i1]
    pieces = [
        engine.transform_modes(t, ext, 2, 8, -2, -1, engine.BMS_TERM_H, tr, ctx=ctx, shard=(need[r][0], ext.shape[0], i0, a)),
        engine.transform_modes(t, own, 2, 8, -2, -1, engine.BMS_TERM_H, tr, ctx=ctx, shard=(i0, own.shape[0], a, b)),
        engine.transform_modes(t, ext, 2, 8, -2, -1, engine.BMS_TERM_H, tr, ctx=ctx, shard=(need[r][0], ext.shape[0], b, i1)),
    ]
    assert [p[2] for p in pieces] == [first, a, b]
    assert np.array_equal(np.concatenate([p[0] for p in pieces]), t_ref)
    assert np.abs(np.concatenate([p[1] for p in pieces]) - d_ref).max() < 1e-14 * np.abs(d_ref).max()
