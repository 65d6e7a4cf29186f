"""Where the device-resident map_to_superrest_frame spends its wall time: cProfile of the control loop (host side) -- run under
rocprofv3 --kernel-trace --stats for the GPU side.  Usage: python tools/superrest_profile.py [n_times] [ell_max] [padding_time]"""
import os, sys, time, cProfile, pstats, io
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import scri_amd
from tests.test_oracle_charges import kerr_schild_abd

n = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
ell_max = int(sys.argv[2]) if len(sys.argv) > 2 else 12
pad = float(sys.argv[3]) if len(sys.argv) > 3 else 250.0
ctx = scri_amd.Context(0)
u = np.linspace(-1000, 1000, num=n)
a = scri_amd.AsymptoticBondiData(u, ell_max, ctx=ctx)
a._raw_data[:] = kerr_schild_abd(2.0, 0.456, ell_max, u)
st = np.array([0.0, 3e-2 - 1j * 5e-3, 1e-3, -3e-2 - 1j * 5e-3, 2e-4 + 1j * 1e-4, 1j * 3e-3, 1e-2, 1j * 3e-3, 2e-4 - 1j * 1e-4])
moved = a.transform(supertranslation=st, frame_rotation=np.array([1.0, 2, 3, 4]) / np.sqrt(30), boost_velocity=np.array([2e-4, -3e-5, 2e-4]))
moved_d = moved.to_device()
moved_d.map_to_superrest_frame(t_0=0, padding_time=pad)  # warm-up: plans, tables, pools
torch.cuda.synchronize()
ctx.enable_timing(True)
ctx.get_timing(reset=True)
pr = cProfile.Profile()
t0 = time.perf_counter()
pr.enable()
rec_d, tr_d, errs_d = moved_d.map_to_superrest_frame(t_0=0, padding_time=pad)
torch.cuda.synchronize()
pr.disable()
t1 = time.perf_counter()
print(f"map_to_superrest_frame, device-resident, second call: {t1 - t0:.3f} s")
tm = ctx.get_timing()
print("GPU time by tag (ms):", {k: round(v[0], 2) for k, v in tm.items() if v[0] > 0.5}, "sum", round(sum(v[0] for v in tm.values()), 1))
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(45)
print(s.getvalue()[:9000])
