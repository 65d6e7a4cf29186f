"""The host path on either side of the edge (profiles/r06_r_host_path_by_size.txt): n = 40000, l <= 16, 10 and 14 time shards, a few calls each --
to be run under `rocprofv3 --kernel-trace --memory-copy-trace` for a timeline of copies and kernels.
Usage: python tools/probes/host_path_edge_trace.py [pieces ...]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import scri_amd
from scri_amd import engine, synthetic

ctx = scri_amd.Context(0)
kw = synthetic.CONFIGS["cfg3"]["kwargs"]
L, n = 16, 40000
t = np.linspace(-10.0, 0.1 * n, n)
data = synthetic.chirp_modes(t, 2, L, 3)
tr = engine.make_transformation(np.asarray(kw["supertranslation"]), kw["frame_rotation"], kw["boost_velocity"], 2 * (L + 2) + 1, 2 * (L + 2) + 1, L)
for pieces in [int(a) for a in sys.argv[1:]] or [10, 14]:
    for _ in range(3):
        engine.transform_modes(t, data, 2, L, -2, -1, engine.BMS_TERM_H, tr, ctx=ctx, pieces=pieces)
    ts = []
    for _ in range(5):
        t0 = time.perf_counter()
        engine.transform_modes(t, data, 2, L, -2, -1, engine.BMS_TERM_H, tr, ctx=ctx, pieces=pieces)
        ts.append((time.perf_counter() - t0) * 1e3)
    print(f"pieces {pieces}: " + " ".join(f"{x:.2f}" for x in ts), flush=True)
    time.sleep(0.05)
