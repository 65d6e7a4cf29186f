for k in 0 1 2 3; do echo "KNOCK=$k (1: no stores, 2: no loads)"; SCRI_AMD_TS_KNOCK=$k SCRI_AMD_NO_FUSED_ABD_MIX=1 python - <<PY 2>&1 | tail -1
import os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np, torch, scri_amd
from scri_amd import engine, synthetic
ctx = scri_amd.Context(0); ctx.enable_timing(True)
u, raw, spec = synthetic.abd_workload("cfg5", n_times=25000, ell_max=24); kw = spec["kwargs"]
tr = engine.make_transformation(kw["supertranslation"], kw["frame_rotation"], [0,0,0], 99, 99, 24)
d_in = torch.from_numpy(raw).cuda(); d_out = torch.empty_like(d_in); torch.cuda.synchronize()
for _ in range(3): engine.transform_abd(u, d_in.data_ptr(), 24, tr, ctx=ctx, device=True, out_ptr=d_out.data_ptr())
print({k: round(v[0]/3, 3) for k, v in ctx.get_timing(reset=True).items() if v[1]})
PY
done
