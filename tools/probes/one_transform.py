"""A few device-resident cfg3 transforms (profiling target): python3 tools/probes/one_transform.py [reps] [n_times] [boost scale]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from scri_amd import _lib, engine, synthetic

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 3
n = int(sys.argv[2]) if len(sys.argv) > 2 else 100000
scale = float(sys.argv[3]) if len(sys.argv) > 3 else 1.0
t, data, spec = synthetic.workload("cfg3", n_times=n)
kw, L = spec["kwargs"], spec["ell_max"]
nth = 2 * (L + 2) + 1
tr = engine.make_transformation(kw["supertranslation"], kw["frame_rotation"], np.asarray(kw["boost_velocity"]) * scale, nth, nth, L)
ctx = _lib.Context(0)
d_in = torch.from_numpy(data).cuda()
d_out = torch.empty((n, data.shape[1]), dtype=torch.complex128, device="cuda")
for _ in range(reps):
    engine.transform_modes(t, d_in.data_ptr(), 2, L, -2, -1, engine.BMS_TERM_H, tr, ctx=ctx, device=True, ld=data.shape[1], out_ptr=d_out.data_ptr())
torch.cuda.synchronize()
print("done", float(torch.view_as_real(d_out[:10]).abs().sum()))
