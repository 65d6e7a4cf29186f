"""R then ~R on the reference's linear test waveform (tests/test_rotations.py:14-129: the m = 0 modes are exactly zero and must
come back below 1e-12 next to |data| ~ 1e3): max error at the zero modes over seeds, per kernel variant (env)."""
import json, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import scri_amd
from oracle import quat, sample_waveforms_ref as samples
from scri_amd import engine

ctx = scri_amd.Context(0)
out = {"variant": os.environ.get("SCRI_AMD_ROTATE_STAGED", "resident")}
for ell_max in (8, 16):
    w0 = samples.linear_waveform(n_times=200, ell_max=ell_max)
    worst_c, worst_s = [], []
    for seed in range(20):
        rng = np.random.default_rng(seed)
        for R, sink in ((rng.uniform(-1, 1, 4), worst_c), (rng.uniform(-1, 1, (200, 4)), worst_s)):
            R = R / np.linalg.norm(R, axis=-1, keepdims=True)
            d = w0.data.copy()
            if R.ndim == 1:
                engine.rotate_const(d, 2, ell_max, R, ctx=ctx); engine.rotate_const(d, 2, ell_max, quat.qconj(R), ctx=ctx)
            else:
                engine.rotate_series(d, 2, ell_max, quat.as_spinor_array(R), ctx=ctx); engine.rotate_series(d, 2, ell_max, quat.as_spinor_array(quat.qconj(R)), ctx=ctx)
            sink.append(float(np.abs(d - w0.data)[w0.data == 0].max()))
    out[f"l{ell_max}"] = {"const_max": max(worst_c), "const_median": float(np.median(worst_c)), "series_max": max(worst_s), "series_median": float(np.median(worst_s)), "data_max": float(np.abs(w0.data).max())}
print(json.dumps(out))
