"""One warm-up + N timed transforms of a workload, for rocprofv3 (kernel trace / PMC passes)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from scri_amd import _lib, engine, synthetic

name = sys.argv[1] if len(sys.argv) > 1 else "cfg3"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
t, data, spec = synthetic.workload(name)
kw = spec["kwargs"]
L = spec["ell_max"]
nth = 2 * (L + 2) + 1
tr = engine.make_transformation(kw["supertranslation"], kw.get("frame_rotation", [1, 0, 0, 0]), kw.get("boost_velocity", [0, 0, 0]), nth, nth, L)
ctx = _lib.Context(0)
d = torch.from_numpy(data).cuda()
nm = data.shape[1]
out = torch.empty((len(t), nm), dtype=torch.complex128, device="cuda")
for _ in range(1 + reps):
    engine.transform_modes(t, d.data_ptr(), 2, L, -2, -1, engine.BMS_TERM_H, tr, ctx=ctx, device=True, ld=nm, out_ptr=out.data_ptr())
ctx.synchronize()
print("done")
