"""As shard_overhead.py for the AsymptoticBondiData workload (cfg5): one rank's shard of a world x 25 000-step series."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from scri_amd import _lib, engine, synthetic, sharding

world = int(sys.argv[1]) if len(sys.argv) > 1 else 8
rank = int(sys.argv[2]) if len(sys.argv) > 2 else world - 1
per = int(sys.argv[3]) if len(sys.argv) > 3 else 25000
spec = synthetic.CONFIGS["cfg5"]
kw, L = spec["kwargs"], spec["ell_max"]
nth = 2 * (2 * L + 1) + 1
tr = engine.make_transformation(kw["supertranslation"], kw["frame_rotation"], kw["boost_velocity"], nth, nth, L)
n_global = per * world
u = np.arange(n_global) * spec["dt"]
have, need, window = sharding.plan(u, tr, world)
_, raw, _ = synthetic.abd_workload("cfg5", n_times=n_global, rows=need[rank])
d = torch.from_numpy(raw).cuda()
out = torch.empty((6, have[rank][1] - have[rank][0], (L + 1) ** 2), dtype=torch.complex128, device="cuda")
ctx = _lib.Context(0)
ctx.enable_timing(True)
def step():
    return engine.transform_abd(u, d.data_ptr(), L, tr, ctx=ctx, device=True, out_ptr=out.data_ptr(),
                                shard=(need[rank][0], raw.shape[1], have[rank][0], have[rank][1]))
step()
ctx.get_timing(reset=True)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(2): r = step()
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 2
print({k: round(v[0] / 2, 2) for k, v in ctx.get_timing(reset=True).items() if v[1]})
print(f"world {world} rank {rank}: rows held {raw.shape[1]} (halo {raw.shape[1] - per}), outputs {r[1]}, {dt * 1e3:.1f} ms per step")
