"""DIAGNOSIS: time of the fused analysis at cfg3's shape with phases of the kernel switched off (results are wrong then;
only the time matters).  (the switches this probe drove were removed from the kernel after the diagnosis; it now times the kernel as it is)"""
import os, sys
import numpy as np
from scri_amd import _lib, engine

ctx = _lib.Context(0)
ctx.enable_timing(True)
n, nth, nph, L = 100000, 37, 37, 16
rng = np.random.default_rng(0)
g = rng.standard_normal((n, nth, nph)) + 1j * rng.standard_normal((n, nth, nph))
for knock in [int(a) for a in sys.argv[1:]] or [0, 1, 2, 4, 8, 3, 7, 15, 0]:
    os.environ["SCRI_AMD_FUSED_KNOCK"] = str(knock)
    ts = []
    for _ in range(3):
        engine.map2salm(g, -2, L, ell_min=2, ctx=ctx)
        t = ctx.get_timing(reset=True)
        ts += [round(v[0], 3) for k, v in t.items() if v[1] and "analysis" in k]
    print("knock", knock, ts, flush=True)
