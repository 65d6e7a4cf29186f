"""Kernel times of the boost-free WaveformModes transformation (supertranslation + frame rotation of the cfg3 series, resident in HBM).
Usage: python tools/boost_free_rate.py [n_times] [ell_max] [axis]      (env: the route options of scri_amd/csrc/env.h, read by the context at creation; probe builds: SCRI_AMD_SE_KNOCK)"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import scri_amd
from scri_amd import engine, synthetic

n = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
ell_max = int(sys.argv[2]) if len(sys.argv) > 2 else 16
axis = sys.argv[3] if len(sys.argv) > 3 else "uniform"
ctx = scri_amd.Context(0)
ctx.enable_timing(True)
t, data, spec = synthetic.workload("cfg3", n_times=n, axis=axis)
nm = (ell_max + 1) ** 2 - 4
data = np.ascontiguousarray(data[:, :nm])
kw = spec["kwargs"]
n_theta = 2 * (ell_max + 2) + 1
tr = engine.make_transformation(kw["supertranslation"], kw["frame_rotation"], [0, 0, 0], n_theta, n_theta, ell_max)
local = torch.from_numpy(data).cuda()
out = torch.empty_like(local)
go = lambda: engine.transform_modes(t, local.data_ptr(), 2, ell_max, -2, -1, engine.BMS_TERM_H, tr, ctx=ctx, device=True, ld=nm, out_ptr=out.data_ptr())[1]
for _ in range(3):
    go()
ctx.synchronize(); ctx.get_timing(reset=True)
reps = 10
t0 = time.perf_counter()
for _ in range(reps):
    n_new = go()
ctx.synchronize()
wall = (time.perf_counter() - t0) / reps
tm = ctx.get_timing(reset=True)
print(json.dumps({"n": n, "ell_max": ell_max, "axis": axis, "ms_per_transform": round(wall * 1e3, 4), "n_out": int(n_new),
                  "kernels_ms": {k: round(v[0] / reps, 4) for k, v in tm.items() if v[1]},
                  "env": {k: v for k, v in os.environ.items() if k.startswith("SCRI_AMD_")}}))
