"""Duration of every engine.transform_abd call of the first device-resident map_to_superrest_frame of a process."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
import scri_amd
from scri_amd import engine
from tests.test_oracle_charges import kerr_schild_abd

n, ell_max, pad = 100000, 12, 250.0
ctx = scri_amd.Context(0)
u = np.linspace(-1000, 1000, num=n)
a = scri_amd.AsymptoticBondiData(u, ell_max, ctx=ctx)
a._raw_data[:] = kerr_schild_abd(2.0, 0.456, ell_max, u)
st = np.array([0.0, 3e-2 - 1j * 5e-3, 1e-3, -3e-2 - 1j * 5e-3, 2e-4 + 1j * 1e-4, 1j * 3e-3, 1e-2, 1j * 3e-3, 2e-4 - 1j * 1e-4])
moved = a.transform(supertranslation=st, frame_rotation=np.array([1.0, 2, 3, 4]) / np.sqrt(30), boost_velocity=np.array([2e-4, -3e-5, 2e-4]))
if "--host-first" in sys.argv:
    moved.map_to_superrest_frame(t_0=0, padding_time=pad)
moved_d = moved.to_device()
orig = engine.transform_abd
def timed(*args, **kw):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    out = orig(*args, **kw)
    torch.cuda.synchronize()
    shape = getattr(args[1], "shape", None)
    print(f"  transform_abd rows={args[0].size if hasattr(args[0], 'size') else '?'} raw={shape}: {1e3 * (time.perf_counter() - t0):.1f} ms", flush=True)
    return out
engine.transform_abd = timed
import scri_amd.asymptotic_bondi_data as abd_mod
if hasattr(abd_mod, "engine"):
    abd_mod.engine.transform_abd = timed
for rep in range(2):
    t0 = time.perf_counter()
    moved_d.map_to_superrest_frame(t_0=0, padding_time=pad)
    torch.cuda.synchronize()
    print(f"run {rep}: {time.perf_counter() - t0:.2f} s")
