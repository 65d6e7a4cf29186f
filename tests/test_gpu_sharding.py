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
