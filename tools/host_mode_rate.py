"""PCIe-inclusive rate of the headline transform: numpy arrays in host memory in and out through mem = BMS_HOST."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from scri_amd import _lib, engine, synthetic

t, data, spec = synthetic.workload("cfg3")
kw, L = spec["kwargs"], spec["ell_max"]
nth = 2 * (L + 2) + 1
tr = engine.make_transformation(kw["supertranslation"], kw["frame_rotation"], kw["boost_velocity"], nth, nth, L)
ctx = _lib.Context(0)
for _ in range(4):  # (results bound as in the timed loop: the second page-locked result block is allocated here, once)
    t_new, d_new = engine.transform_modes(t, data, 2, L, -2, -1, engine.BMS_TERM_H, tr, ctx=ctx)
t0 = time.perf_counter()
k = 16
for _ in range(k):
    t_new, d_new = engine.transform_modes(t, data, 2, L, -2, -1, engine.BMS_TERM_H, tr, ctx=ctx)
dt = (time.perf_counter() - t0) / k
print(f"host in / host out: {dt * 1e3:.1f} ms per transform = {t.size / dt:.3g} timesteps/s ({data.nbytes / 1e6:.0f} MB in, {d_new.nbytes / 1e6:.0f} MB out, numpy memory, page-locked in place from its second use; SCRI_AMD_NO_REGISTER=1 for pageable)")
