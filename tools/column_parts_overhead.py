"""Plan B (grid-column partition): time of one part's contribution to a 1e5-step cfg3 transform against the
unsplit transform, on one GPU (no communication): what the dense part-analysis and the narrower kernels cost."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from scri_amd import _lib, engine, synthetic

parts = int(sys.argv[1]) if len(sys.argv) > 1 else 8
beta_scale = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
t, data, spec = synthetic.workload("cfg3")
kw, L = spec["kwargs"], spec["ell_max"]
nth = 2 * (L + 2) + 1
tr = engine.make_transformation(kw["supertranslation"], kw["frame_rotation"], np.asarray(kw["boost_velocity"]) * beta_scale, nth, nth, L)
n, nm = data.shape
d = torch.from_numpy(data).cuda()
out = torch.empty((n, nm), dtype=torch.complex128, device="cuda")
ctx = _lib.Context(0)
ctx.enable_timing(True)

def run(shard):
    def step():
        return engine.transform_modes(t, d.data_ptr(), 2, L, -2, -1, engine.BMS_TERM_H, tr, ctx=ctx, device=True, ld=nm,
                                      out_ptr=out.data_ptr(), shard=shard)
    for _ in range(3): step()
    ctx.get_timing(reset=True)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10): step()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 10
    tm = ctx.get_timing(reset=True)
    return dt, {k: round(v[0] / 10, 3) for k, v in tm.items() if v[1]}

dt0, k0 = run(None)
print(f"unsplit: {dt0 * 1e3:.3f} ms", k0)
tot = 0
for p in sorted({0, parts // 2, parts - 1}):
    dt, k = run((0, n, 0, n, p, parts))
    print(f"part {p} of {parts}: {dt * 1e3:.3f} ms", k)
