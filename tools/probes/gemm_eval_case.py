"""One case of gemm_eval_ab.py with the location of the largest difference (debugging aid)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from scri_amd import _lib, engine, synthetic

n, o0, o1, scale = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), float(sys.argv[4])
t, data, spec = synthetic.workload("cfg3", n_times=n)
kw, L = spec["kwargs"], spec["ell_max"]
nth = 2 * (L + 2) + 1
tr = engine.make_transformation(kw["supertranslation"], kw["frame_rotation"], np.asarray(kw["boost_velocity"]) * scale, nth, nth, L)
ctx = _lib.Context(0)
(r0, r1), _ = engine.shard_plan(t, tr, o0, o1, ctx=ctx)
sh = (r0, r1 - r0, o0, o1, 0, 0)
print("rows", r0, r1, "M", r1 - r0)
d_in = torch.from_numpy(data[r0:r1].copy()).cuda()
d_out = torch.empty((n, data.shape[1]), dtype=torch.complex128, device="cuda")
res = {}
for mode in ("old", "new"):
    if mode == "old":
        os.environ["SCRI_AMD_NO_GEMM_EVAL"] = "1"
    else:
        os.environ.pop("SCRI_AMD_NO_GEMM_EVAL", None)
    out = engine.transform_modes(t, d_in.data_ptr(), 2, L, -2, -1, engine.BMS_TERM_H, tr, ctx=ctx, device=True, ld=data.shape[1], out_ptr=d_out.data_ptr(), shard=sh)
    torch.cuda.synchronize()
    res[mode] = d_out[: out[1]].cpu().numpy().copy()
    print(mode, "first index", out[2], "rows", out[1])
d = np.abs(res["old"] - res["new"])
rows = d.max(axis=1)
bad = np.nonzero(rows > 1e-14)[0]
print("max", d.max(), "bad rows:", len(bad), bad[:20], bad[-5:] if len(bad) else "")
print("knot of the first bad row relative to r0:", (o0 + bad[0] - r0) if len(bad) else None, "mod 64:", ((o0 + bad[0] - r0) % 64) if len(bad) else None)
