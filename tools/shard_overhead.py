"""Per-step time of one rank's shard of an N x 1e5-step series (computed on a single GPU, no communication) against the
unsharded 1e5-step transform: what the planner, the time-window upload and the halo rows cost under weak scaling."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from scri_amd import _lib, engine, synthetic, sharding

world = int(sys.argv[1]) if len(sys.argv) > 1 else 8
rank = int(sys.argv[2]) if len(sys.argv) > 2 else world // 2
spec = synthetic.CONFIGS["cfg3"]
kw, L = spec["kwargs"], spec["ell_max"]
nth = 2 * (L + 2) + 1
tr = engine.make_transformation(kw["supertranslation"], kw["frame_rotation"], kw["boost_velocity"], nth, nth, L)
n_global = int(sys.argv[3]) if len(sys.argv) > 3 else 100_000 * world  # (1000000: a rank's shard of cfg4)
t = np.arange(n_global) * spec["dt"]
have, need, window = sharding.plan(t, tr, world)
_, rows, _ = synthetic.workload("cfg3", n_times=n_global, rows=need[rank])
d = torch.from_numpy(rows).cuda()
nm = rows.shape[1]
out = torch.empty((have[rank][1] - have[rank][0], nm), dtype=torch.complex128, device="cuda")
per_rank = n_global // world
ctx = _lib.Context(0)
ctx.enable_timing(True)
def step():
    return engine.transform_modes(t, d.data_ptr(), 2, L, -2, -1, engine.BMS_TERM_H, tr, ctx=ctx, device=True, ld=nm,
                                  out_ptr=out.data_ptr(), shard=(need[rank][0], rows.shape[0], have[rank][0], have[rank][1]))
for _ in range(3): step()
ctx.get_timing(reset=True)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(10): r = step()
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 10
tm = ctx.get_timing(reset=True)
print({k: round(v[0] / 10, 3) for k, v in tm.items() if v[1]})
print(f"world {world} rank {rank}: rows held {rows.shape[0]} (halo {rows.shape[0] - per_rank}), outputs {r[1]}, {dt * 1e3:.3f} ms per step")
