"""Boost-free WaveformModes transformation through the separable synthesis vs the evaluating dense product (SCRI_AMD_NO_SEPARABLE_SYNTHESIS),
device-resident, by ell_max: where is the crossover now that the dense route no longer pays for a back substitution on the grid?"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from scri_amd import _lib, engine, synthetic

ctx = _lib.Context(0)
ctx.enable_timing(True)
q = np.array([1.0, 2, 3, 4]) / np.sqrt(30)
for L in (4, 6, 8, 10, 12, 16):
    t, data, spec = synthetic.workload("cfg3", n_times=100000)
    nm = (L + 1) ** 2 - 4
    data = np.ascontiguousarray(data[:, :nm])
    st = np.asarray(spec["kwargs"]["supertranslation"])
    nth = 2 * (L + 2) + 1
    for label, rot in (("st only", [1, 0, 0, 0]), ("st + rotation", q)):
        tr = engine.make_transformation(st, rot, np.zeros(3), nth, nth, L)
        src = torch.from_numpy(data).cuda()
        dst = torch.empty_like(src)
        res = {}
        for route in ("separable", "dense"):
            if route == "dense":
                os.environ["SCRI_AMD_NO_SEPARABLE_SYNTHESIS"] = "1"
            else:
                os.environ.pop("SCRI_AMD_NO_SEPARABLE_SYNTHESIS", None)
            for _ in range(3):
                engine.transform_modes(t, src.data_ptr(), 2, L, -2, -1, engine.BMS_TERM_H, tr, ctx=ctx, device=True, ld=nm, out_ptr=dst.data_ptr())
            ctx.synchronize()
            ctx.get_timing(reset=True)
            t0 = time.perf_counter()
            reps = 20
            for _ in range(reps):
                engine.transform_modes(t, src.data_ptr(), 2, L, -2, -1, engine.BMS_TERM_H, tr, ctx=ctx, device=True, ld=nm, out_ptr=dst.data_ptr())
            ctx.synchronize()
            wall = (time.perf_counter() - t0) / reps * 1e3
            tm = {k: round(v[0] / reps, 3) for k, v in ctx.get_timing(reset=True).items() if v[1]}
            res[route] = (wall, tm, dst.cpu().numpy().copy())
        os.environ.pop("SCRI_AMD_NO_SEPARABLE_SYNTHESIS", None)
        err = np.abs(res["dense"][2] - res["separable"][2]).max() / np.abs(res["separable"][2]).max()
        print(f"l<={L:2d} {nth}x{nth} {label:14s}: separable {res['separable'][0]:.3f} ms  dense {res['dense'][0]:.3f} ms   rel diff {err:.1e}")
        print("      separable", res["separable"][1])
        print("      dense    ", res["dense"][1], flush=True)
