"""GPU against the oracle at boosts far beyond anything a waveform frame has (|v| = 0.6, 0.9, 0.99): h (l = 2..6) and the six
AsymptoticBondiData fields (l <= 4).  Prints the relative difference per case.  Usage: python tools/probes/extreme_boost_probe.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import scri_amd
from oracle import abd_ref, waveform_grid_ref as grid_ref
from oracle.containers import ABD, WM, h
from scri_amd import synthetic

ctx = scri_amd.Context(0)
n, L = 600, 6
t = np.linspace(-40.0, 80.0, n)
data = synthetic.chirp_modes(t, 2, L, 5) * (1 + 0.003 * t[:, None])
st = synthetic.real_supertranslation(0.2 * (np.arange(16) * 0.1 + 1j * np.arange(16)[::-1] * 0.05))
fr = np.array([0.7, -0.2, 0.5, 0.1]); fr /= np.linalg.norm(fr)
u = np.linspace(-30.0, 60.0, 300)
raw = np.zeros((6, 300, 25), dtype=complex)
for f, s_ in enumerate(synthetic.ABD_SPINS):
    raw[f] = synthetic.chirp_modes(u, 0, 4, 40 + f)
    raw[f, :, : s_ * s_] = 0
for speed in (0.6, 0.9, 0.99):
    v = speed * np.array([0.48, -0.6, 0.64])
    kw = dict(supertranslation=st, frame_rotation=fr, boost_velocity=v)
    try:
        o = grid_ref.transform(WM(t=t, data=data, ell_min=2, ell_max=L, dataType=h), **kw)
        w = scri_amd.WaveformModes(t=t, data=data, ell_min=2, ell_max=L, dataType=scri_amd.h, frameType=scri_amd.Inertial, r_is_scaled_out=True,
                                   m_is_scaled_out=True, ctx=ctx).transform(**kw)
        same = w.t.shape == o.t.shape
        err = np.abs(w.data - o.data).max() / np.abs(o.data).max() if same and o.t.size else float("nan")
        print(f"|v| = {speed}: h        n_out oracle {o.t.size} gpu {w.t.size}  max|out| {np.abs(o.data).max() if o.t.size else 0:.3e}  rel diff {err:.2e}")
    except Exception as e:  # noqa: BLE001
        print(f"|v| = {speed}: h        {type(e).__name__}: {str(e)[:200]}")
    try:
        o = abd_ref.transform(ABD(u, raw, 4), **kw)
        a = scri_amd.AsymptoticBondiData(u, 4, ctx=ctx)
        a._raw_data[:] = raw
        g = a.transform(**kw)
        same = g.n_times == o.n_times
        errs = [np.abs(g._raw_data[f] - o.raw[f]).max() / max(np.abs(o.raw[f]).max(), 1e-300) for f in range(6)] if same and o.n_times else []
        print(f"|v| = {speed}: six fields n_out oracle {o.n_times} gpu {g.n_times}  rel diff per field {[float(f'{e:.1e}') for e in errs]}")
    except Exception as e:  # noqa: BLE001
        print(f"|v| = {speed}: six fields {type(e).__name__}: {str(e)[:200]}")
