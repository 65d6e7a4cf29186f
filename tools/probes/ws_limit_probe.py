"""cfg5 shard (25 000 steps, six fields, 99 x 99) with the default 32 GB work space (3 chunks) and with 100 GB (1 chunk)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from scri_amd import _lib, engine, synthetic
u, raw, spec = synthetic.abd_workload("cfg5", n_times=25000); kw = spec["kwargs"]
tr = engine.make_transformation(kw["supertranslation"], kw["frame_rotation"], kw["boost_velocity"], 99, 99, 24)
d_in = torch.from_numpy(raw).cuda(); d_out = torch.empty_like(d_in); torch.cuda.synchronize()
for limit in (None, 100 << 30, None, 100 << 30):
    ctx = _lib.Context(0, workspace_limit=limit); ctx.enable_timing(True)
    engine.transform_abd(u, d_in.data_ptr(), 24, tr, ctx=ctx, device=True, out_ptr=d_out.data_ptr()); ctx.synchronize(); ctx.get_timing(reset=True)
    t0 = time.perf_counter()
    for _ in range(3): engine.transform_abd(u, d_in.data_ptr(), 24, tr, ctx=ctx, device=True, out_ptr=d_out.data_ptr())
    ctx.synchronize(); dt = (time.perf_counter() - t0) / 3
    print("workspace limit", limit, f"{dt*1e3:.2f} ms", {k: (round(v[0]/3, 2), v[1] // 3) for k, v in ctx.get_timing(reset=True).items() if v[1]}, flush=True)
    ctx.close()
