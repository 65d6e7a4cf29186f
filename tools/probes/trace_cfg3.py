import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from scri_amd import _lib, engine, synthetic
t, data, spec = synthetic.workload("cfg3"); kw = spec["kwargs"]; L = 16; nth = 37
tr = engine.make_transformation(kw["supertranslation"], kw["frame_rotation"], kw["boost_velocity"], nth, nth, L)
ctx = _lib.Context(0)
d = torch.from_numpy(data).cuda(); o = torch.empty_like(d); torch.cuda.synchronize()
for _ in range(5): engine.transform_modes(t, d.data_ptr(), 2, L, -2, -1, engine.BMS_TERM_H, tr, ctx=ctx, device=True, ld=285, out_ptr=o.data_ptr())
os.environ["SCRI_AMD_TRACE"] = "1"
import time
t0 = time.perf_counter()
engine.transform_modes(t, d.data_ptr(), 2, L, -2, -1, engine.BMS_TERM_H, tr, ctx=ctx, device=True, ld=285, out_ptr=o.data_ptr())
print("whole call", (time.perf_counter() - t0) * 1e3, "ms")
