"""A/B of the dense route: spline evaluation in the product's epilogue (default) against elimination on the modes + back
substitution on the grid (SCRI_AMD_NO_GEMM_EVAL=1).  Same process, device-resident input, cfg3 shape at several lengths,
boost scales and shards; prints the largest difference and the time per transform of either route.
usage: python tools/probes/gemm_eval_ab.py [n_times ...]      (SCRI_AMD_GEMM_EVAL_STEP=61|64 selects the tiling)"""
import os, sys, time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch

from scri_amd import _lib, engine, synthetic


def run(n, scale=1.0, shard=None, reps=10, lmax=None):
    t, data, spec = synthetic.workload("cfg3", n_times=n)
    kw, L = spec["kwargs"], spec["ell_max"]
    if lmax is not None:
        L = lmax
        data = np.ascontiguousarray(data[:, : (L + 1) ** 2 - 4])
    nth = 2 * (L + 2) + 1
    v = np.asarray(kw["boost_velocity"]) * scale
    tr = engine.make_transformation(kw["supertranslation"], kw["frame_rotation"], v, nth, nth, L)
    ctx = _lib.Context(0)
    n_modes = data.shape[1]
    n_out = n_modes
    if shard is None:
        rows0, rows = 0, n
        sh = None
    else:
        o0, o1 = shard
        (r0, r1), _ = engine.shard_plan(t, tr, o0, o1, ctx=ctx)
        rows0, rows = r0, r1 - r0
        sh = (r0, r1 - r0, o0, o1, 0, 0)
    d_in = torch.from_numpy(data[rows0 : rows0 + rows].copy()).cuda()
    d_out = torch.empty((n, n_out), dtype=torch.complex128, device="cuda")
    res = {}
    for mode in ("old", "new"):
        if mode == "old":
            os.environ["SCRI_AMD_NO_GEMM_EVAL"] = "1"
        else:
            os.environ.pop("SCRI_AMD_NO_GEMM_EVAL", None)
        d_out.fill_(float("nan"))
        torch.cuda.synchronize()
        out = engine.transform_modes(t, d_in.data_ptr(), 2, L, -2, -1, engine.BMS_TERM_H, tr, ctx=ctx, device=True, ld=n_modes,
                                     out_ptr=d_out.data_ptr(), shard=sh)
        n_new = out[1]
        torch.cuda.synchronize()
        got = d_out[:n_new].cpu().numpy().copy()
        for _ in range(3):
            engine.transform_modes(t, d_in.data_ptr(), 2, L, -2, -1, engine.BMS_TERM_H, tr, ctx=ctx, device=True, ld=n_modes, out_ptr=d_out.data_ptr(), shard=sh)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            engine.transform_modes(t, d_in.data_ptr(), 2, L, -2, -1, engine.BMS_TERM_H, tr, ctx=ctx, device=True, ld=n_modes, out_ptr=d_out.data_ptr(), shard=sh)
        torch.cuda.synchronize()
        res[mode] = (got, (time.perf_counter() - t0) / reps * 1e3, out[0])
        if n >= 50000:  # per-kernel times (HIP events around each launch: they serialise the host a little)
            ctx.enable_timing(True)
            ctx.get_timing(reset=True)
            for _ in range(reps):
                engine.transform_modes(t, d_in.data_ptr(), 2, L, -2, -1, engine.BMS_TERM_H, tr, ctx=ctx, device=True, ld=n_modes, out_ptr=d_out.data_ptr(), shard=sh)
            tm = ctx.get_timing(reset=True)
            ctx.enable_timing(False)
            print("   ", mode, {k: round(v[0] / reps, 3) for k, v in tm.items() if v[1]}, flush=True)
    a, b = res["old"][0], res["new"][0]
    scale_d = np.abs(a).max()
    diff = np.abs(a - b).max() if a.shape == b.shape else float("inf")
    nan = int(np.isnan(b).sum())
    print(f"n={n:8d} lmax={L:2d} boost x{scale:<6g} shard={shard}: rows out {a.shape[0]} / {b.shape[0]}, max|old-new| = {diff:.3e} (scale {scale_d:.3g}), NaNs {nan}; "
          f"old {res['old'][1]:.3f} ms, new {res['new'][1]:.3f} ms", flush=True)
    return diff <= 1e-12 * max(scale_d, 1.0) and nan == 0 and np.array_equal(res["old"][2], res["new"][2])


if __name__ == "__main__":
    ns = [int(a) for a in sys.argv[1:]] or [100, 257, 1000, 5000, 100000]
    ok = True
    for n in ns:
        ok &= run(n)
    ok &= run(3000, lmax=4)
    ok &= run(20000, scale=26.7)
    ok &= run(20000, scale=270.0)
    ok &= run(40000, shard=(15000, 25000))
    ok &= run(40000, shard=(30000, 40000), scale=26.7)
    print("ALL OK" if ok else "MISMATCH")
    sys.exit(0 if ok else 1)
