import os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np, torch
import scri_amd
from scri_amd import engine, synthetic, _lib
t, data, spec = synthetic.workload("cfg3", n_times=100000)
kw = spec["kwargs"]; ell_max = 16
lst = int(round(np.sqrt(len(kw["supertranslation"])))) - 1
n_theta = 2 * (ell_max + lst) + 1
tr = engine.make_transformation(kw["supertranslation"], kw["frame_rotation"], kw["boost_velocity"], n_theta, n_theta, ell_max)
dev = torch.device("cuda", 0)
d_in = torch.from_numpy(data).to(dev); d_out = torch.empty_like(d_in)
ctx = _lib.Context(0)
nm = data.shape[1]
for i in range(6):
    ctx.synchronize(); t0 = time.perf_counter()
    r = engine.transform_modes(t, d_in.data_ptr(), 2, ell_max, -2, -1, engine.BMS_TERM_H, tr, ctx=ctx, device=True, ld=nm, out_ptr=d_out.data_ptr())
    t1 = time.perf_counter()
    ctx.synchronize(); t2 = time.perf_counter()
    print(f"call {i}: returned after {1e3*(t1-t0):.2f} ms, synchronized after {1e3*(t2-t0):.2f} ms", file=sys.stderr, flush=True)
