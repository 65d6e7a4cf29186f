"""As host_path_by_size.py for the six-field call (AsymptoticBondiData, host arrays in and out): one call against the pipelined one at
several shard counts (best of 5, ms).  Usage: python tools/probes/host_path_by_size_abd.py [l_max]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import scri_amd
from scri_amd import engine, synthetic

ctx = scri_amd.Context(0)
kw = synthetic.CONFIGS["cfg5"]["kwargs"]
L = int(sys.argv[1]) if len(sys.argv) > 1 else 12
counts = (0, 2, 3, 4, 6, 8, 10, 14, 20)
print("shards (0 = one call): " + "  ".join(f"{c:7d}" for c in counts))
for n in (1000, 2500, 5000, 10000, 20000, 40000, 100000):
    u, raw, _ = synthetic.abd_workload("cfg5", n_times=n, ell_max=L)
    tr = engine.make_transformation(kw["supertranslation"], kw["frame_rotation"], kw["boost_velocity"], 4 * L + 1, 4 * L + 1, L)
    row = []
    for pieces in counts:
        def run():
            if pieces == 0:
                os.environ["SCRI_AMD_NO_PIPELINE"] = "1"
                try:
                    return engine.transform_abd(u, raw, L, tr, ctx=ctx)
                finally:
                    del os.environ["SCRI_AMD_NO_PIPELINE"]
            return engine.transform_abd(u, raw, L, tr, ctx=ctx, pieces=pieces)
        for _ in range(2):
            out = run()
        del out
        best = 1e9
        for _ in range(5):
            t0 = time.perf_counter()
            out = run()
            best = min(best, time.perf_counter() - t0)
            del out
        row.append(best * 1e3)
    k = int(np.argmin(row))
    print(f"n = {n:6d} ({raw.nbytes / 2**20:7.1f} MB): " + "  ".join(f"{v:7.2f}" for v in row) + f"   best: {counts[k]}", flush=True)
