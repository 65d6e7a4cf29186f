"""MEASUREMENT: boost-free AsymptoticBondiData.transform (supertranslation + frame rotation) through the two-kernel separable
synthesis vs the six dense sYlm products (SCRI_AMD_NO_SEPARABLE_SYNTHESIS), device-resident fields, HIP-event kernel times.
usage: python tools/separable_probe_abd.py [n_times=25000] [ell_max=24] [reps=3]"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import scri_amd  # noqa: E402
from scri_amd import engine, synthetic  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 25000
ell_max = int(sys.argv[2]) if len(sys.argv) > 2 else 24
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 3
ctx = scri_amd.Context(0)
ctx.enable_timing(True)
u, raw, spec = synthetic.abd_workload("cfg5", n_times=n, ell_max=ell_max)
kw = spec["kwargs"]
n_theta = 2 * (2 * ell_max + 1) + 1
nm = (ell_max + 1) ** 2
dev = torch.device("cuda", 0)
d_in = torch.from_numpy(raw).to(dev)
d_out = torch.empty_like(d_in)
torch.cuda.synchronize()
for label, boost in (("boost-free", [0, 0, 0]), ("with the workload's boost", kw["boost_velocity"])):
    tr = engine.make_transformation(kw["supertranslation"], kw["frame_rotation"], boost, n_theta, n_theta, ell_max)
    res = {}
    for route in ("dense", "separable"):
        if route == "dense":
            ctx.option("NO_SEPARABLE_SYNTHESIS", 1)
        else:
            ctx.option("NO_SEPARABLE_SYNTHESIS", 0)
        engine.transform_abd(u, d_in.data_ptr(), ell_max, tr, ctx=ctx, device=True, out_ptr=d_out.data_ptr())
        ctx.synchronize()
        ctx.get_timing(reset=True)
        t0 = time.perf_counter()
        for _ in range(reps):
            n_new = engine.transform_abd(u, d_in.data_ptr(), ell_max, tr, ctx=ctx, device=True, out_ptr=d_out.data_ptr())[1]
        ctx.synchronize()
        wall = (time.perf_counter() - t0) / reps
        tm = {k: round(v[0] / reps, 3) for k, v in ctx.get_timing(reset=True).items() if v[1]}
        res[route] = d_out[:, :n_new].clone()
        print(f"{label:28s} {route:10s} l<={ell_max} {n_theta}x{n_theta} n={n}: {wall * 1e3:8.2f} ms per transform  {tm}", flush=True)
        if label != "boost-free":
            break
    if len(res) == 2:
        err = float((res["dense"] - res["separable"]).abs().max()) / float(res["dense"].abs().max())
        syn_bytes = 6 * n * 16 * (nm + 2 * (2 * ell_max + 1) * ((n_theta + 7) // 8 * 8) + n_theta * n_theta)
        print(f"   separable vs dense: max rel diff {err:.2e}; synthesis traffic (modes + F twice + grid) {syn_bytes / 1e9:.2f} GB per transform", flush=True)
