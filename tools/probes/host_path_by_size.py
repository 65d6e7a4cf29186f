"""Host arrays in and out, l <= 16, by series length: the one-call path against the pipelined one at several piece counts (best of 7
calls each, ms).  Usage: python tools/probes/host_path_by_size.py [l_max]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import scri_amd
from scri_amd import engine, synthetic

ctx = scri_amd.Context(0)
kw = synthetic.CONFIGS["cfg3"]["kwargs"]
L = int(sys.argv[1]) if len(sys.argv) > 1 else 16
counts = (0, 2, 3, 4, 5, 6, 8, 10, 14, 20)
print("pieces (0 = one call): " + "  ".join(f"{c:6d}" for c in counts))
for n in (2500, 5000, 10000, 20000, 30000, 40000, 60000, 80000, 100000, 200000, 400000):
    t = np.linspace(-10.0, 0.1 * n, n)
    data = synthetic.chirp_modes(t, 2, L, 3)
    st = np.asarray(kw["supertranslation"])
    tr = engine.make_transformation(st, kw["frame_rotation"], kw["boost_velocity"], 2 * (L + 2) + 1, 2 * (L + 2) + 1, L)
    row = []
    for pieces in counts:
        def run():
            if pieces == 0:
                os.environ["SCRI_AMD_NO_PIPELINE"] = "1"
                try:
                    return engine.transform_modes(t, data, 2, L, -2, -1, engine.BMS_TERM_H, tr, ctx=ctx)
                finally:
                    del os.environ["SCRI_AMD_NO_PIPELINE"]
            return engine.transform_modes(t, data, 2, L, -2, -1, engine.BMS_TERM_H, tr, ctx=ctx, pieces=pieces)
        for _ in range(3):
            out = run()
        del out
        best = 1e9
        for _ in range(7):
            t0 = time.perf_counter()
            out = run()
            best = min(best, time.perf_counter() - t0)
            del out
        row.append(best * 1e3)
    k = int(np.argmin(row))
    print(f"n = {n:6d} ({data.nbytes / 2**20:7.1f} MB): " + "  ".join(f"{v:6.2f}" for v in row) + f"   best: {counts[k]}", flush=True)
