"""Where the host time of the device-resident map_to_superrest_frame goes (cProfile, second run)."""
import os, sys, time, cProfile, pstats
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
import scri_amd
from tests.test_oracle_charges import kerr_schild_abd

n, ell_max, pad = 100000, 12, 250.0
ctx = scri_amd.Context(0)
u = np.linspace(-1000, 1000, num=n)
a = scri_amd.AsymptoticBondiData(u, ell_max, ctx=ctx)
a._raw_data[:] = kerr_schild_abd(2.0, 0.456, ell_max, u)
st = np.array([0.0, 3e-2 - 1j * 5e-3, 1e-3, -3e-2 - 1j * 5e-3, 2e-4 + 1j * 1e-4, 1j * 3e-3, 1e-2, 1j * 3e-3, 2e-4 - 1j * 1e-4])
moved = a.transform(supertranslation=st, frame_rotation=np.array([1.0, 2, 3, 4]) / np.sqrt(30), boost_velocity=np.array([2e-4, -3e-5, 2e-4]))
if "--host-first" in sys.argv:
    t0 = time.perf_counter()
    moved.map_to_superrest_frame(t_0=0, padding_time=pad)
    print(f"host-resident run first: {time.perf_counter() - t0:.2f} s")
moved_d = moved.to_device()
for rep in range(2):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    pr = cProfile.Profile()
    pr.enable()
    rec_d, tr_d, errs_d = moved_d.map_to_superrest_frame(t_0=0, padding_time=pad)
    torch.cuda.synchronize()
    pr.disable()
    print(f"run {rep}: {time.perf_counter() - t0:.2f} s")
    if rep == 0:
        pstats.Stats(pr).sort_stats("tottime").print_stats(8)
pstats.Stats(pr).sort_stats("tottime").print_stats(18)
