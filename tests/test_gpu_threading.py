"""The threading contract of the boundary (SURVEY 8(b) "Threading"; include/scri_amd.h: "re-entrant per context ... ctypes releases
the GIL"): two contexts on one GPU driven from two host threads at the same time, and one context shared by two threads under the
caller's lock.  Every result must be bit-identical to the serial run -- the kernels are deterministic (no atomics on the data
path), so any difference is state leaking between contexts: a plan cache, a table buffer, a page-locked pool, a per-kernel
attribute set by the other thread."""
import threading

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ITER = 50


def _wm_case(n=3000):
    """cfg3's shape (ell 2..16, 285 modes, supertranslation + frame rotation + boost, 37 x 37 grid) on a shorter series"""
    from scri_amd import engine, synthetic

    t, data, spec = synthetic.workload("cfg3", n_times=n)
    kw = spec["kwargs"]
    tr = engine.make_transformation(kw["supertranslation"], kw["frame_rotation"], kw["boost_velocity"], 37, 37, 16)
    return t, data, tr


def _wm_boost_free_case(n=1500):
    from scri_amd import engine, synthetic

    t, data, spec = synthetic.workload("cfg3", n_times=n)
    kw = spec["kwargs"]
    tr = engine.make_transformation(kw["supertranslation"], kw["frame_rotation"], [0.0, 0.0, 0.0], 37, 37, 16)
    return t, data, tr


def _abd_case(n=300, ell_max=6):
    from scri_amd import engine, synthetic

    u, raw, spec = synthetic.abd_workload("cfg5", n_times=n, ell_max=ell_max)
    kw = spec["kwargs"]
    w = 2 * ell_max + 1
    tr = engine.make_transformation(kw["supertranslation"], kw["frame_rotation"], kw["boost_velocity"], 2 * w + 1, 2 * w + 1, ell_max)
    return u, raw, tr


def _rotors(n, seed):
    rng = np.random.default_rng(seed)
    R = rng.normal(size=(n, 4))
    R /= np.linalg.norm(R, axis=1)[:, None]
    return np.ascontiguousarray((R[:, 0] + 1j * R[:, 3])[:, None] * [1, 0] + (R[:, 2] + 1j * R[:, 1])[:, None] * [0, 1])


def _job_a(ctx, cases):
    """one turn of thread A: the cfg3-shaped transformation and its boost-free sibling (dense and separable routes)"""
    from scri_amd import engine

    out = []
    for t, data, tr in cases:
        t_new, d = engine.transform_modes(t, data, 2, 16, -2, -1, engine.BMS_TERM_H, tr, ctx=ctx)
        out.append((np.array(t_new), np.array(d)))
    return out


def _job_b(ctx, rot_case, abd_case):
    """one turn of thread B: a rotor series applied to ell 2..8 modes, then a small AsymptoticBondiData transformation"""
    from scri_amd import engine

    data, spinors = rot_case
    rotated = data.copy()
    engine.rotate_series(rotated, 2, 8, spinors, ctx=ctx)
    u, raw, tr = abd_case
    u_new, r = engine.transform_abd(u, raw, 6, tr, ctx=ctx)
    return [(np.zeros(0), rotated), (np.array(u_new), np.array(r))]


def _same(a, b):
    return all(x[0].shape == y[0].shape and np.array_equal(x[0], y[0]) and x[1].shape == y[1].shape and np.array_equal(x[1], y[1])
               for x, y in zip(a, b)) and len(a) == len(b)


def _run_threads(targets):
    errors = []

    def wrap(fn):
        def run():
            try:
                fn()
            except BaseException as e:  # noqa: BLE001 -- reported by the test below
                errors.append(e)
        return run

    threads = [threading.Thread(target=wrap(fn)) for fn in targets]
    for th in threads:
        th.start()
    for th in threads:
        th.join()
    if errors:
        raise errors[0]


@pytest.fixture(scope="module")
def cases():
    from scri_amd import synthetic

    wm = [_wm_case(), _wm_boost_free_case()]
    t8, d8, _ = synthetic.workload("cfg2", n_times=4000)
    rot = (d8, _rotors(4000, 5))
    return wm, rot, _abd_case()


def test_two_contexts_two_threads_bit_identical_to_serial(cases):
    import scri_amd

    wm, rot, abd = cases
    ca, cb = scri_amd.Context(0), scri_amd.Context(0)
    try:
        ref_a, ref_b = _job_a(ca, wm), _job_b(cb, rot, abd)
        # the serial run is itself reproducible (what "bit-identical" below is measured against)
        assert _same(ref_a, _job_a(ca, wm)) and _same(ref_b, _job_b(cb, rot, abd))
        bad = []

        def loop_a():
            for i in range(ITER):
                if not _same(ref_a, _job_a(ca, wm)):
                    bad.append(("A", i))

        def loop_b():
            for i in range(ITER):
                if not _same(ref_b, _job_b(cb, rot, abd)):
                    bad.append(("B", i))

        _run_threads([loop_a, loop_b])
        assert not bad, bad[:5]
        # and with the roles swapped between the contexts (each context now meets the other job's shapes: its plan caches turn over)
        bad.clear()

        def loop_a2():
            for i in range(ITER // 5):
                if not _same(ref_a, _job_a(cb, wm)):
                    bad.append(("A on b", i))

        def loop_b2():
            for i in range(ITER // 5):
                if not _same(ref_b, _job_b(ca, rot, abd)):
                    bad.append(("B on a", i))

        _run_threads([loop_a2, loop_b2])
        assert not bad, bad[:5]
    finally:
        ca.close()
        cb.close()


def test_same_kernels_different_shapes_in_two_threads(cases):
    """Both threads launch the SAME kernels with different dynamic-LDS sizes (separable synthesis and fused analysis of an
    ell <= 16 / 37 x 37 and an ell <= 10 / 25 x 25 transformation): a per-launch function attribute would be a race here."""
    import scri_amd
    from scri_amd import engine, synthetic

    t, data, spec = synthetic.workload("cfg3", n_times=1500)
    kw = spec["kwargs"]
    big = (t, data, 16, engine.make_transformation(kw["supertranslation"], kw["frame_rotation"], [0.0, 0.0, 0.0], 37, 37, 16))
    n10 = (10 + 1) ** 2 - 4
    small = (t, np.ascontiguousarray(data[:, :n10]), 10,
             engine.make_transformation(kw["supertranslation"], kw["frame_rotation"], [0.0, 0.0, 0.0], 25, 25, 10))
    ca, cb = scri_amd.Context(0), scri_amd.Context(0)

    def once(ctx, case):
        tt, dd, lmax, tr = case
        t_new, d = engine.transform_modes(tt, dd, 2, lmax, -2, -1, engine.BMS_TERM_H, tr, ctx=ctx)
        return [(np.array(t_new), np.array(d))]

    try:
        ref_big, ref_small = once(ca, big), once(cb, small)
        bad = []

        def loop(ctx, case, ref, tag):
            def run():
                for i in range(ITER):
                    if not _same(ref, once(ctx, case)):
                        bad.append((tag, i))
            return run

        _run_threads([loop(ca, big, ref_big, "big"), loop(cb, small, ref_small, "small")])
        assert not bad, bad[:5]
    finally:
        ca.close()
        cb.close()


def test_one_context_shared_by_two_threads_under_the_callers_lock(cases):
    """A context is not re-entrant; callers that share one serialise their calls.  Under that lock the interleaving of two
    threads' calls (different shapes: tables and plans are rebuilt or found again) must not change any result."""
    import scri_amd

    wm, rot, abd = cases
    c = scri_amd.Context(0)
    lock = threading.Lock()
    try:
        ref_a, ref_b = _job_a(c, wm), _job_b(c, rot, abd)
        bad = []

        def loop_a():
            for i in range(ITER // 2):
                with lock:
                    got = _job_a(c, wm)
                if not _same(ref_a, got):
                    bad.append(("A", i))

        def loop_b():
            for i in range(ITER // 2):
                with lock:
                    got = _job_b(c, rot, abd)
                if not _same(ref_b, got):
                    bad.append(("B", i))

        _run_threads([loop_a, loop_b])
        assert not bad, bad[:5]
    finally:
        c.close()


def test_two_contexts_on_different_routes_concurrently(monkeypatch):
    """Routes are options of a CONTEXT (bms_ctx_set_option; round 6), not of the process: context A on the dense product and the
    two-pass boost-free route, context B on the separable synthesis and the fused route, driven from two threads at once -- every
    result bit-identical to the same context's serial run, and the two contexts really took different kernels (their timing tags).
    A variable set in the environment AFTER the contexts exist changes nothing (it is read in bms_ctx_create only)."""
    from scri_amd import _lib, engine, synthetic

    a, b = _lib.Context(0), _lib.Context(0)
    try:
        a.option("NO_SEPARABLE_SYNTHESIS", 1)  # dense sYlm product even without a boost
        a.option("NO_SYNTHESIS_EVAL", 1)
        b.option("SYNTHESIS_EVAL", 1)  # fused boost-free route at every l
        b.option("NO_SMALL_DENSE", 1)
        assert (a.option("NO_SEPARABLE_SYNTHESIS"), b.option("NO_SEPARABLE_SYNTHESIS")) == (1, 0)
        with pytest.raises(ValueError, match="no route option"):
            a.option("NO_SUCH_ROUTE", 1)
        monkeypatch.setenv("SCRI_AMD_NO_GEMM_EVAL", "1")  # too late for a and b
        assert a.option("NO_GEMM_EVAL") == 0 and b.option("NO_GEMM_EVAL") == 0

        n, L = 20000, 16
        t, data, spec = synthetic.workload("cfg3", n_times=n)
        kw = spec["kwargs"]
        n_theta = 2 * (L + 2) + 1
        tr_free = engine.make_transformation(kw["supertranslation"], kw["frame_rotation"], [0, 0, 0], n_theta, n_theta, L)
        tr_boost = engine.make_transformation(kw["supertranslation"], kw["frame_rotation"], kw["boost_velocity"], n_theta, n_theta, L)

        def work(ctx, turns):
            outs = []
            for _ in range(turns):
                for tr in (tr_free, tr_boost):
                    _, d = engine.transform_modes(t, data, 2, L, -2, -1, engine.BMS_TERM_H, tr, ctx=ctx)
                    outs.append(np.array(d))
            return outs

        serial = {}
        tags = {}
        for name, ctx in (("a", a), ("b", b)):
            ctx.enable_timing(True)
            ctx.get_timing(reset=True)
            engine.transform_modes(t, data, 2, L, -2, -1, engine.BMS_TERM_H, tr_free, ctx=ctx)
            tags[name] = {k for k, v in ctx.get_timing(reset=True).items() if v[1]}
            ctx.enable_timing(False)
            serial[name] = work(ctx, 1)
        # the dense route has no rotation of the modes and no separable kernels; the fused route no back substitution on a grid
        assert "rotate" not in tags["a"] and "rotate" in tags["b"] and "spline_backward" not in tags["b"]
        assert np.abs(serial["a"][0] - serial["b"][0]).max() < 1e-13 * np.abs(serial["a"][0]).max()  # same results to rounding
        got, errors = {}, []

        def run(name, ctx):
            try:
                got[name] = work(ctx, 6)
            except BaseException as e:  # noqa: BLE001
                errors.append(e)

        threads = [threading.Thread(target=run, args=(name, ctx)) for name, ctx in (("a", a), ("b", b))]
        for th in threads:
            th.start()
        for th in threads:
            th.join()
        assert not errors, errors
        for name in ("a", "b"):
            for k, d in enumerate(got[name]):
                assert np.array_equal(d, serial[name][k % 2]), (name, k)
    finally:
        a.close(), b.close()
