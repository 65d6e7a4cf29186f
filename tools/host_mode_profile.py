"""cProfile of the host-memory call path of tools/host_mode_rate.py (where do the milliseconds outside the GPU timeline go?)"""
import cProfile, os, pstats, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from scri_amd import _lib, engine, synthetic

t, data, spec = synthetic.workload("cfg3")
kw, L = spec["kwargs"], spec["ell_max"]
nth = 2 * (L + 2) + 1
tr = engine.make_transformation(kw["supertranslation"], kw["frame_rotation"], kw["boost_velocity"], nth, nth, L)
ctx = _lib.Context(0)
for _ in range(3):
    engine.transform_modes(t, data, 2, L, -2, -1, engine.BMS_TERM_H, tr, ctx=ctx)
pr = cProfile.Profile()
pr.enable()
for _ in range(8):
    out = engine.transform_modes(t, data, 2, L, -2, -1, engine.BMS_TERM_H, tr, ctx=ctx)
pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(14)
