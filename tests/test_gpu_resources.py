"""Resources over many calls: device memory comes back when contexts are closed, repeated transformations of the same and of
changing shapes do not grow the work space without bound, and the host process does not accumulate page-locked or plain memory."""
import gc
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _free_device_bytes():
    import torch

    torch.cuda.synchronize()
    return torch.cuda.mem_get_info(0)[0]


def _rss_bytes():
    with open("/proc/self/status") as f:
        for line in f:
            if line.startswith("VmRSS"):
                return int(line.split()[1]) * 1024
    return 0


def _locked_bytes():
    with open("/proc/self/status") as f:
        for line in f:
            if line.startswith("VmLck") or line.startswith("VmPin"):
                yield int(line.split()[1]) * 1024


def _case(n, ell_max, seed):
    from scri_amd import engine, synthetic

    t = np.linspace(0.0, 0.1 * n, n)
    data = synthetic.chirp_modes(t, 2, ell_max, seed)
    st = synthetic.real_supertranslation(0.05 * (np.arange(9) + 1j))
    tr = engine.make_transformation(st, (0.9, 0.1, -0.3, 0.3), (0.01, -0.02, 0.015), 2 * (ell_max + 2) + 1, 2 * (ell_max + 2) + 1, ell_max)
    return t, data, tr


def test_contexts_give_their_device_memory_back():
    import scri_amd
    from scri_amd import engine

    t, data, tr = _case(20000, 8, 1)
    first = None
    for k in range(12):
        ctx = scri_amd.Context(0)
        out = engine.transform_modes(t, data, 2, 8, -2, -1, engine.BMS_TERM_H, tr, ctx=ctx)
        ctx.close()
        del ctx, out
        gc.collect()
        free = _free_device_bytes()
        if k == 1:
            first = free  # (after the first round trip: the runtime's own pools exist)
        elif k > 1:
            assert free > first - (64 << 20), f"device memory did not come back: round {k}, {first - free} bytes fewer than after round 1"


def test_repeated_and_changing_shapes_do_not_grow_without_bound(ctx):
    from scri_amd import engine

    shapes = [(30000, 8), (12000, 12), (50000, 6), (8000, 16), (30000, 8)]
    cases = [_case(n, L, 10 + i) for i, (n, L) in enumerate(shapes)]
    marks = []
    for lap in range(6):
        for (n, L), (t, data, tr) in zip(shapes, cases):
            engine.transform_modes(t, data, 2, L, -2, -1, engine.BMS_TERM_H, tr, ctx=ctx)
        gc.collect()
        marks.append((_free_device_bytes(), _rss_bytes(), sum(_locked_bytes())))
    # after the second lap every buffer has reached the size of the largest shape it serves: nothing moves any more
    for lap in range(3, 6):
        assert marks[lap][0] > marks[2][0] - (32 << 20), f"device memory keeps shrinking: {[m[0] for m in marks]}"
        assert marks[lap][1] < marks[2][1] + (256 << 20), f"host memory keeps growing: {[m[1] for m in marks]}"
        assert marks[lap][2] < marks[2][2] + (256 << 20), f"page-locked memory keeps growing: {[m[2] for m in marks]}"


def test_device_resident_objects_release_their_memory(ctx):
    import scri_amd
    from scri_amd import synthetic

    n, L = 40000, 8
    t = np.linspace(0.0, 400.0, n)
    before = None
    for k in range(8):
        w = scri_amd.WaveformModes(t=t, data=synthetic.chirp_modes(t, 2, L, 30 + k), ell_min=2, ell_max=L, dataType=scri_amd.h,
                                   frameType=scri_amd.Inertial, r_is_scaled_out=True, m_is_scaled_out=True, ctx=ctx).to_device()
        w2 = w.transform(space_translation=[0.1, -0.2, 0.3], boost_velocity=[0.0, 0.0, 1e-2])
        assert w2.n_times > 0
        del w, w2
        gc.collect()
        import torch

        torch.cuda.empty_cache()
        free = _free_device_bytes()
        if k == 1:
            before = free
        elif k > 1:
            assert free > before - (32 << 20), f"round {k}: {before - free} bytes of HBM fewer than after round 1"
