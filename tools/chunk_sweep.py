"""cfg3 step time against the work-space limit (rows per chunk): does keeping a chunk's grids inside the 256 MB Infinity
Cache between the kernels pay for the extra launches and the GEMM tail of small chunks?"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from scri_amd import _lib, engine, synthetic

t, data, spec = synthetic.workload("cfg3")
kw, L = spec["kwargs"], spec["ell_max"]
nth = 2 * (L + 2) + 1
tr = engine.make_transformation(kw["supertranslation"], kw["frame_rotation"], kw["boost_velocity"], nth, nth, L)
n, nm = data.shape
d = torch.from_numpy(data).cuda()
out = torch.empty((n, nm), dtype=torch.complex128, device="cuda")
for mb in [int(a) for a in sys.argv[1:]] or [0, 2048, 1024, 512, 256, 128]:
    ctx = _lib.Context(0, workspace_limit=(mb << 20) if mb else None)
    ctx.enable_timing(True)
    def step():
        return engine.transform_modes(t, d.data_ptr(), 2, L, -2, -1, engine.BMS_TERM_H, tr, ctx=ctx, device=True, ld=nm, out_ptr=out.data_ptr())
    for _ in range(3): step()
    ctx.get_timing(reset=True)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10): step()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 10
    tm = ctx.get_timing(reset=True)
    print(f"limit {mb or 'default'} MB: {dt * 1e3:.3f} ms", {k: (round(v[0] / 10, 3), v[1] // 10) for k, v in tm.items() if v[1]})
    ctx.close()
