"""How much of the fused analysis at cfg3 is the column gather?  The transform hands the analysis a grid whose columns are
sorted by time-skew rate (good for the spline march), read through col_of_pixel; bms_map2salm runs the same kernel on a
grid in natural order.  Prints the kernel time of each (HIP events)."""
import numpy as np

from scri_amd import _lib, engine

ctx = _lib.Context(0)
ctx.enable_timing(True)
n, nth, nph, L = 100000, 37, 37, 16
rng = np.random.default_rng(0)
g = rng.standard_normal((n, nth, nph)) + 1j * rng.standard_normal((n, nth, nph))
for _ in range(3):
    engine.map2salm(g, -2, L, ell_min=2, ctx=ctx)
    t = ctx.get_timing(reset=True)
    print({k: (round(v[0], 3), v[1]) for k, v in t.items() if v[1]})
