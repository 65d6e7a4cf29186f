"""Time-series rotation of a HOST array in place: the one-call path (context option NO_ROTATE_PIPELINE) against the default, which sends
blocks of rows up, rotates them and brings them back on three streams (engine_rotate.hip); page-locked and ordinary numpy memory.
With the probe build, SCRI_AMD_ROTATE_BLOCKS=k forces the block count.  Best of 7, ms.
Usage: python tools/probes/host_rotation_by_blocks.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import scri_amd
from scri_amd import _lib, engine, synthetic

ctx = scri_amd.Context(0)
print("lib:", os.path.basename(_lib.LIB_PATH), "SCRI_AMD_ROTATE_BLOCKS =", os.environ.get("SCRI_AMD_ROTATE_BLOCKS"))
for L, n in ((8, 100000), (16, 100000), (16, 400000), (4, 400000), (16, 10000)):
    t = np.linspace(0.0, 0.1 * n, n)
    nm = (L + 1) ** 2 - 4
    src = synthetic.chirp_modes(t, 2, L, 3)
    ang = 0.01 * t
    sp = np.stack([np.cos(ang) + 0j, np.sin(ang) * (0.6 + 0.8j)], axis=1)
    row = []
    results = {}
    for memory in ("page-locked", "numpy"):
        block = _lib.pinned_empty((n, nm), np.complex128) if memory == "page-locked" else np.empty((n, nm), dtype=complex)
        for label, off in (("one call", 1), ("blocks", 0)):
            ctx.option("NO_ROTATE_PIPELINE", off)
            best = 1e9
            for rep in range(9):
                block[:] = src
                t0 = time.perf_counter()
                engine.rotate_series(block, 2, L, sp, ctx=ctx)
                dt = time.perf_counter() - t0
                if rep >= 2:
                    best = min(best, dt)
            results[(memory, label)] = block.copy()
            row.append(f"{memory} {label}: {best * 1e3:7.2f}")
        ctx.option("NO_ROTATE_PIPELINE", 0)
    ref = results[("page-locked", "one call")]
    diff = max(np.abs(v - ref).max() for v in results.values()) / np.abs(ref).max()
    print(f"l <= {L:2d}, n = {n} ({src.nbytes / 2**20:7.1f} MB):  " + "   ".join(row) + f"   max relative difference: {diff:.1e}", flush=True)
