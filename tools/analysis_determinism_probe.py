"""map2salm twice on the same input: bitwise equal?  and against the one-role kernel (SCRI_AMD_NO_SPLIT_ANALYSIS)."""
import os, sys
import numpy as np
from scri_amd import _lib, engine

ctx = _lib.Context(0)
rng = np.random.default_rng(1)
bad = 0
for (nth, nph, L, s) in [(37, 37, 16, -2), (13, 13, 6, -2), (15, 15, 6, 0), (27, 27, 12, 1), (21, 21, 8, -2), (9, 9, 4, 2), (17, 19, 8, -1), (33, 33, 16, 0)]:
    for n in (2, 3, 5, 64, 257, 1001, 4097):
        g = rng.standard_normal((n, nth, nph)) + 1j * rng.standard_normal((n, nth, nph))
        ctx.option("NO_SPLIT_ANALYSIS", 0)
        a = [engine.map2salm(g, s, L, ell_min=abs(s), ctx=ctx) for _ in range(4)]
        ctx.option("NO_SPLIT_ANALYSIS", 1)
        b = engine.map2salm(g, s, L, ell_min=abs(s), ctx=ctx)
        same = all(np.array_equal(a[0], x) for x in a[1:])
        err = np.abs(a[0] - b).max() / np.abs(b).max()
        flag = "" if (same and err < 1e-13) else "   <<<<<<"
        bad += bool(flag)
        print(nth, nph, L, s, n, "deterministic" if same else "DIFFERS", f"{err:.2e}", flag, flush=True)
print("bad:", bad)
