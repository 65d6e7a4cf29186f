"""GPU against the oracle at l_max = 24, 32, 48, 64 (h, l from 2; supertranslation l <= 2; rotation + boost |v| = 0.05, and the same
without the boost): the relative difference and the route's wall time.  Usage: python tools/probes/high_ell_probe.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import scri_amd
from oracle import waveform_grid_ref as grid_ref
from oracle.containers import WM, h
from scri_amd import synthetic

ctx = scri_amd.Context(0)
n = 160
t = np.linspace(-10.0, 30.0, n)
st = synthetic.real_supertranslation(0.1 * (np.arange(9) * 0.2 + 1j * np.arange(9)[::-1] * 0.1))
fr = np.array([0.7, -0.2, 0.5, 0.1]); fr /= np.linalg.norm(fr)
for L in (24, 32, 48, 64):
    data = synthetic.chirp_modes(t, 2, L, 5)
    for label, v in (("boost", np.array([0.03, -0.02, 0.035])), ("no boost", np.zeros(3))):
        kw = dict(supertranslation=st, frame_rotation=fr, boost_velocity=v)
        try:
            t0 = time.perf_counter()
            o = grid_ref.transform(WM(t=t, data=data, ell_min=2, ell_max=L, dataType=h), **kw)
            t1 = time.perf_counter()
            w = scri_amd.WaveformModes(t=t, data=data, ell_min=2, ell_max=L, dataType=scri_amd.h, frameType=scri_amd.Inertial, r_is_scaled_out=True,
                                       m_is_scaled_out=True, ctx=ctx).transform(**kw)
            t2 = time.perf_counter()
            err = np.abs(w.data - o.data).max() / np.abs(o.data).max() if w.t.shape == o.t.shape else float("nan")
            print(f"l_max = {L:2d} ({label:8s}): n_out {o.t.size} / {w.t.size}  rel diff {err:.2e}   oracle {t1 - t0:6.1f} s, gpu {t2 - t1:6.3f} s", flush=True)
        except Exception as e:  # noqa: BLE001
            print(f"l_max = {L} ({label}): {type(e).__name__}: {str(e)[:300]}", flush=True)
