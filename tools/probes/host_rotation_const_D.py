"""bms_rotate_const_D (the seam of the reference's numba kernel, scri/rotations.py:346-367) on a host series: one call against blocks on
three streams.  Usage: python tools/probes/host_rotation_const_D.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import scri_amd
from oracle import quat, wigner
from scri_amd import engine, synthetic

ctx = scri_amd.Context(0)
n, L = 100000, 16
t = np.linspace(0, 1e4, n)
src = synthetic.chirp_modes(t, 2, L, 3)
q = np.array([0.5, -0.5, 0.5, 0.5])
D = wigner.wigner_D_matrices(*quat.as_spinor_array(q), 2, L)
a = src.copy()
out = {}
for off in (1, 0):
    ctx.option("NO_ROTATE_PIPELINE", off)
    best = 1e9
    for rep in range(8):
        a[:] = src
        t0 = time.perf_counter()
        engine.rotate_const_D(a, 2, L, D, ctx=ctx)
        dt = time.perf_counter() - t0
        if rep >= 2:
            best = min(best, dt)
    out[off] = a.copy()
    print("rotate_const_D, host series 1e5 x l <= 16:", "one call" if off else "blocks  ", round(best * 1e3, 2), "ms", flush=True)
print("max relative difference:", np.abs(out[0] - out[1]).max() / np.abs(out[1]).max())
