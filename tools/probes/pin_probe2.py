import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from scri_amd import _lib, engine, synthetic

t, data, spec = synthetic.workload("cfg3")
kw, L = spec["kwargs"], spec["ell_max"]
nth = 2 * (L + 2) + 1
tr = engine.make_transformation(kw["supertranslation"], kw["frame_rotation"], kw["boost_velocity"], nth, nth, L)
ctx = _lib.Context(0)
orig = _lib._PinnedBlock.__init__
def timed(self, nbytes):
    t0 = time.perf_counter(); had = {k: len(v) for k, v in _lib._PinnedBlock._pool.items()}
    orig(self, nbytes)
    print(f"   pinned block {nbytes} B: {(time.perf_counter()-t0)*1e3:.2f} ms, pool before {had}")
_lib._PinnedBlock.__init__ = timed
for i in range(5):
    t0 = time.perf_counter()
    t_new, d_new = engine.transform_modes(t, data, 2, L, -2, -1, engine.BMS_TERM_H, tr, ctx=ctx)
    print(i, f"{(time.perf_counter()-t0)*1e3:.2f} ms")
