"""Where a HOST-resident map_to_superrest_frame spends its time beyond the device-resident one (tools/superrest_timing.py: 2.0-2.5 s
against 0.9 s): the same call three times on the same host object, with the pieces timed on their own.
Usage: python tools/probes/superrest_host_breakdown.py [n_times] [ell_max] [padding_time]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch  # noqa: F401
import scri_amd
from tests.test_oracle_charges import kerr_schild_abd

n = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
ell_max = int(sys.argv[2]) if len(sys.argv) > 2 else 12
pad = float(sys.argv[3]) if len(sys.argv) > 3 else 250.0
ctx = scri_amd.Context(0)
ctx.reserve()
u = np.linspace(-1000, 1000, num=n)
a = scri_amd.AsymptoticBondiData(u, ell_max, ctx=ctx)
a._raw_data[:] = kerr_schild_abd(2.0, 0.456, ell_max, u)
st = np.array([0.0, 3e-2 - 1j * 5e-3, 1e-3, -3e-2 - 1j * 5e-3, 2e-4 + 1j * 1e-4, 1j * 3e-3, 1e-2, 1j * 3e-3, 2e-4 - 1j * 1e-4])
kw = dict(supertranslation=st, frame_rotation=np.array([1.0, 2, 3, 4]) / np.sqrt(30), boost_velocity=np.array([2e-4, -3e-5, 2e-4]))
for k in range(3):
    t0 = time.perf_counter()
    moved = a.transform(**kw)
    print(f"host transform #{k}: {time.perf_counter() - t0:.3f} s")
for k in range(3):
    t0 = time.perf_counter()
    rec, tr, errs = moved.map_to_superrest_frame(t_0=0, padding_time=pad)
    print(f"host-resident map_to_superrest_frame #{k}: {time.perf_counter() - t0:.3f} s")
# the pieces
i1 = np.abs(moved.t - (0 - (pad + 200))).argmin()
i2 = np.abs(moved.t - (0 + (pad + 200))).argmin() + 1
for k in range(2):
    t0 = time.perf_counter()
    sl = moved[i1:i2]
    t1 = time.perf_counter()
    sd = sl.to_device(ctx)
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    out = moved.transform(**tr.as_kwargs()) if hasattr(tr, "as_kwargs") else moved.transform(
        supertranslation=tr.supertranslation, frame_rotation=tr.frame_rotation, boost_velocity=tr.boost_velocity)
    t3 = time.perf_counter()
    print(f"pieces #{k}: slice {t1 - t0:.3f} s, window to device {t2 - t1:.3f} s, final host transform {t3 - t2:.3f} s  (window rows {i2 - i1})")
md = moved.to_device()
for k in range(2):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    md.map_to_superrest_frame(t_0=0, padding_time=pad)
    torch.cuda.synchronize()
    print(f"device-resident #{k}: {time.perf_counter() - t0:.3f} s")
