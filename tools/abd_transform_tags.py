"""GPU time by stage of one AsymptoticBondiData.transform on resident fields (python tools/abd_transform_tags.py [n_times] [ell_max])."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, scri_amd
from tests.test_oracle_charges import kerr_schild_abd

n = int(sys.argv[1]) if len(sys.argv) > 1 else 45000
ell_max = int(sys.argv[2]) if len(sys.argv) > 2 else 12
ctx = scri_amd.Context(0)
u = np.linspace(-450, 450, num=n)
a = scri_amd.AsymptoticBondiData(u, ell_max, ctx=ctx)
a._raw_data[:] = kerr_schild_abd(2.0, 0.456, ell_max, u)
ad = a.to_device()
st = np.array([0.0, 3e-2 - 1j * 5e-3, 1e-3, -3e-2 - 1j * 5e-3, 2e-4 + 1j * 1e-4, 1j * 3e-3, 1e-2, 1j * 3e-3, 2e-4 - 1j * 1e-4])
cases = {"supertranslation + rotation + boost": dict(supertranslation=st, frame_rotation=np.array([1.0, 2, 3, 4]) / np.sqrt(30), boost_velocity=np.array([2e-4, -3e-5, 2e-4])),
         "supertranslation": dict(supertranslation=st), "rotation": dict(frame_rotation=np.array([1.0, 2, 3, 4]) / np.sqrt(30)),
         "boost": dict(boost_velocity=np.array([2e-4, -3e-5, 2e-4]))}
for name, kw in cases.items():
    ad.transform(**kw)
    torch.cuda.synchronize()
    ctx.enable_timing(True); ctx.get_timing(reset=True)
    t0 = time.perf_counter()
    for _ in range(3):
        r = ad.transform(**kw)
    torch.cuda.synchronize()
    wall = (time.perf_counter() - t0) / 3
    tm = ctx.get_timing()
    print(f"{name}: wall {wall*1e3:.1f} ms; GPU ms by stage:", {k: round(v[0] / 3, 2) for k, v in tm.items() if v[0] > 0.05}, "sum", round(sum(v[0] for v in tm.values()) / 3, 1))
    ctx.enable_timing(False)
