"""MEASUREMENT: transformations whose boost points along the polar axis of the rotated grid -- separable synthesis at the aberrated
ring colatitudes vs the dense sYlm products (SCRI_AMD_NO_AXIS_BOOST_SEPARABLE), device-resident, HIP-event kernel times.
usage: python tools/axis_boost_probe.py [reps=3]"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import scri_amd  # noqa: E402
from scri_amd import engine, synthetic  # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 3
ctx = scri_amd.Context(0)
ctx.enable_timing(True)
dev = torch.device("cuda", 0)
zrot = np.array([np.cos(0.35), 0.0, 0.0, np.sin(0.35)])


def routes(call):
    res = {}
    for route in ("dense", "separable"):
        if route == "dense":
            ctx.option("NO_AXIS_BOOST_SEPARABLE", 1)
        else:
            ctx.option("NO_AXIS_BOOST_SEPARABLE", 0)
        call()
        ctx.synchronize()
        ctx.get_timing(reset=True)
        t0 = time.perf_counter()
        for _ in range(reps):
            out = call()
        ctx.synchronize()
        wall = (time.perf_counter() - t0) / reps
        tm = {k: round(v[0] / reps, 3) for k, v in ctx.get_timing(reset=True).items() if v[1]}
        res[route] = (out, wall, tm)
    return res


# ---- AsymptoticBondiData, l <= 24 on 99 x 99, 25 000 steps
n, ell_max = 25000, 24
u, raw, spec = synthetic.abd_workload("cfg5", n_times=n, ell_max=ell_max)
kw = spec["kwargs"]
n_theta = 2 * (2 * ell_max + 1) + 1
d_in = torch.from_numpy(raw).to(dev)
d_out = torch.empty_like(d_in)
speed = float(np.linalg.norm(kw["boost_velocity"]))
tr = engine.make_transformation(kw["supertranslation"], zrot, [0.0, 0.0, speed], n_theta, n_theta, ell_max)


def abd_call():
    n_new = engine.transform_abd(u, d_in.data_ptr(), ell_max, tr, ctx=ctx, device=True, out_ptr=d_out.data_ptr())[1]
    return d_out[:, :n_new].clone()


r = routes(abd_call)
for route, (out, wall, tm) in r.items():
    print(f"ABD l<={ell_max} {n_theta}x{n_theta} n={n} |v|={speed:.3g} along z' {route:10s}: {wall * 1e3:8.2f} ms  {tm}", flush=True)
err = float((r["dense"][0] - r["separable"][0]).abs().max()) / float(r["dense"][0].abs().max())
print(f"   separable vs dense: max rel diff {err:.2e}", flush=True)

# ---- WaveformModes, the headline shape (l <= 16, 10^5 steps) and l <= 8
for name, lmax, nt in (("cfg3", 16, 100000), ("cfg2", 8, 100000)):
    t, data, spec = synthetic.workload(name, n_times=nt)
    kw = spec["kwargs"]
    speed = float(np.linalg.norm(kw.get("boost_velocity", [0.0, 0.0, 1e-3])))
    w = scri_amd.WaveformModes(t=t, data=data, ell_min=2, ell_max=lmax, dataType=scri_amd.h, frameType=scri_amd.Inertial,
                               r_is_scaled_out=True, m_is_scaled_out=True, ctx=ctx).to_device()
    kw2 = dict(supertranslation=kw["supertranslation"], frame_rotation=zrot, boost_velocity=[0.0, 0.0, speed])
    r = routes(lambda: w.transform(**kw2))
    for route, (out, wall, tm) in r.items():
        print(f"WM {name} l<={lmax} n={nt} |v|={speed:.3g} along z' {route:10s}: {wall * 1e3:8.2f} ms  {tm}", flush=True)
    a, b = r["dense"][0].data, r["separable"][0].data
    a = a if isinstance(a, torch.Tensor) else torch.as_tensor(np.asarray(a))
    b = b if isinstance(b, torch.Tensor) else torch.as_tensor(np.asarray(b))
    print(f"   separable vs dense: max rel diff {float((a - b).abs().max()) / float(a.abs().max()):.2e}", flush=True)
