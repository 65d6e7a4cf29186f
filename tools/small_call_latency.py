"""MEASUREMENT: latency of small calls (the fixed cost of a transformation): WaveformModes h, host arrays in and out."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import scri_amd
from scri_amd import synthetic

ctx = scri_amd.Context(0)
ctx.enable_timing(True)
for name, n, lmax in (("cfg2-like", 2000, 8), ("cfg3-like", 2000, 16), ("cfg3-like", 20000, 16)):
    t = np.linspace(-10.0, 100.0, n)
    data = synthetic.chirp_modes(t, 2, lmax, 3)
    kw = synthetic.CONFIGS["cfg3"]["kwargs"]
    for label, kk in (("boost + supertranslation + rotation", kw), ("no boost", {k: v for k, v in kw.items() if k != "boost_velocity"})):
        w = scri_amd.WaveformModes(t=t, data=data, ell_min=2, ell_max=lmax, dataType=scri_amd.h, frameType=scri_amd.Inertial, r_is_scaled_out=True,
                                   m_is_scaled_out=True, ctx=ctx)
        for _ in range(3):
            w.transform(**kk)
        ctx.get_timing(reset=True)
        reps = 20
        t0 = time.perf_counter()
        for _ in range(reps):
            w.transform(**kk)
        dt = (time.perf_counter() - t0) / reps
        tm = ctx.get_timing(reset=True)
        gpu = sum(v[0] for v in tm.values()) / reps
        print(f"{name} n={n} l<={lmax} {label}: {dt * 1e3:.3f} ms per call, kernels {gpu:.3f} ms  {({k: round(v[0] / reps, 3) for k, v in tm.items() if v[1]})}", flush=True)
