import cProfile, os, pstats, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, scri_amd
from scri_amd import synthetic
ctx = scri_amd.Context(0)
t = np.linspace(-10.0, 100.0, 2000); data = synthetic.chirp_modes(t, 2, 16, 3); kw = synthetic.CONFIGS["cfg3"]["kwargs"]
w = scri_amd.WaveformModes(t=t, data=data, ell_min=2, ell_max=16, dataType=scri_amd.h, frameType=scri_amd.Inertial, r_is_scaled_out=True, m_is_scaled_out=True, ctx=ctx)
for _ in range(5): w.transform(**kw)
pr = cProfile.Profile(); pr.enable()
for _ in range(50): w.transform(**kw)
pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(18)
