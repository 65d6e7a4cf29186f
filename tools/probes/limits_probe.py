import os, sys
sys.path.insert(0, os.getcwd())
import numpy as np, scri_amd
from scri_amd import synthetic
ctx = scri_amd.Context(0)
ctx.enable_timing(True)
rot = np.array([0.4, 1.0, -2.0, 0.3]); rot /= np.linalg.norm(rot)
for ell_max, ntheta in ((33, None), (30, 104), (20, 45), (12, 104)):
    n = 60
    t = np.linspace(-30.0, 40.0, n)
    data = synthetic.chirp_modes(t, 2, ell_max, 3)
    st = synthetic.real_supertranslation(0.2 * (np.arange(16) + 1j * np.arange(16)[::-1]))
    kw = dict(supertranslation=st, frame_rotation=rot)
    if ntheta: kw.update(n_theta=ntheta, n_phi=ntheta)
    def run():
        w = scri_amd.WaveformModes(t=t, data=data, ell_min=2, ell_max=ell_max, dataType=scri_amd.h, frameType=scri_amd.Inertial, r_is_scaled_out=True, m_is_scaled_out=True, ctx=ctx)
        return w.transform(**kw)
    os.environ.pop("SCRI_AMD_NO_SEPARABLE_SYNTHESIS", None)
    ctx.get_timing(reset=True); a = run(); tags = {k for k, v in ctx.get_timing(reset=True).items() if v[1]}
    os.environ["SCRI_AMD_NO_SEPARABLE_SYNTHESIS"] = "1"; b = run()
    print(ell_max, ntheta, "separable" if "rotate" in tags else "dense", a.n_times, np.abs(a.data - b.data).max() / np.abs(b.data).max())
