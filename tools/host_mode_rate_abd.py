"""PCIe-inclusive rate of AsymptoticBondiData.transform: numpy arrays in host memory in and out (1e5 steps, l <= 12: 1.6 GB each way)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import scri_amd
from tests.test_oracle_charges import kerr_schild_abd

n, ell_max = 100000, 12
ctx = scri_amd.Context(0)
u = np.linspace(-1000, 1000, num=n)
a = scri_amd.AsymptoticBondiData(u, ell_max, ctx=ctx)
a._raw_data[:] = kerr_schild_abd(2.0, 0.456, ell_max, u)
st = np.array([0.0, 3e-2 - 1j * 5e-3, 1e-3, -3e-2 - 1j * 5e-3, 2e-4 + 1j * 1e-4, 1j * 3e-3, 1e-2, 1j * 3e-3, 2e-4 - 1j * 1e-4])
kw = dict(supertranslation=st, frame_rotation=np.array([1.0, 2, 3, 4]) / np.sqrt(30), boost_velocity=np.array([2e-4, -3e-5, 2e-4]))
for _ in range(3):
    b = a.transform(**kw)
t0 = time.perf_counter()
k = 5
for _ in range(k):
    b = a.transform(**kw)
dt = (time.perf_counter() - t0) / k
print(f"ABD host in / host out: {dt * 1e3:.0f} ms per transform = {n / dt:.3g} timesteps/s ({a._raw_data.nbytes / 1e9:.2f} GB in, {b._raw_data.nbytes / 1e9:.2f} GB out)")
