"""Randomised check of the overlapping-shards host path of engine.transform_modes against the one-call path (python tools/pipeline_check.py on a GPU box)."""
import os, sys
sys.path.insert(0, os.getcwd())
import numpy as np
import scri_amd
from scri_amd import engine
engine.PIPELINE_MIN_BYTES = 1 << 12
ctx = scri_amd.Context(0)
rng = np.random.default_rng(11)
bad = 0
for case in range(40):
    n = int(rng.integers(60, 9000))
    ell_max = int(rng.integers(2, 9))
    nm = (ell_max + 1) ** 2 - 4
    t = np.cumsum(rng.uniform(0.05, 0.2, n)) - rng.uniform(0, 400)
    data = rng.standard_normal((n, nm)) + 1j * rng.standard_normal((n, nm))
    v = rng.uniform(-1, 1, 3) * rng.choice([0, 1e-3, 0.05])
    q = rng.standard_normal(4); q /= np.linalg.norm(q)
    st = np.zeros(9, dtype=complex); st[0] = rng.uniform(-3, 3); st[2] = rng.uniform(-0.2, 0.2); st[6] = rng.uniform(-0.1, 0.1)
    st[1] = 0.05 * rng.uniform(-1, 1) + 0.03j; st[3] = -np.conj(st[1])
    kw = dict(supertranslation=st, frame_rotation=q, boost_velocity=v)
    dt = [scri_amd.h, scri_amd.psi4, scri_amd.sigma][case % 3]
    def run():
        w = scri_amd.WaveformModes(t=t, data=data, ell_min=2, ell_max=ell_max, dataType=dt, frameType=scri_amd.Inertial,
                                   r_is_scaled_out=True, m_is_scaled_out=True, ctx=ctx)
        return w.transform(**kw)
    os.environ["SCRI_AMD_NO_PIPELINE"] = "1"
    a = run()
    del os.environ["SCRI_AMD_NO_PIPELINE"]
    b = run()
    ok = a.t.shape == b.t.shape and np.array_equal(a.t, b.t) and (a.data.size == 0 or np.abs(a.data - b.data).max() <= 1e-12 * max(np.abs(a.data).max(), 1e-300))
    if not ok:
        bad += 1
        print("case", case, n, ell_max, "MISMATCH", a.t.shape, b.t.shape)
print("pipelined vs one call: failures", bad, "of 40")
