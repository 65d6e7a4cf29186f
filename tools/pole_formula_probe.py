"""MEASUREMENT: how far the GPU result is from the oracle run with numpy-quaternion's LITERAL pole-angle formula
(2 acos(sqrt((w^2+z^2)/n)), oracle.quat.ROBUST_POLES = False) on boosted transformations -- the GPU parity tests run the oracle
with the well-conditioned atan2 form of the same angle.  Prints max |difference| / scale for both forms of the oracle."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import scri_amd  # noqa: E402
from oracle import abd_ref, quat, waveform_grid_ref as grid_ref  # noqa: E402
from oracle.containers import WM, h  # noqa: E402
from scri_amd import synthetic  # noqa: E402
from tests.test_gpu_transform_abd import smooth_abd  # noqa: E402

ctx = scri_amd.Context(0)
direction = np.array([1.0, 2.0, 3.0]) / np.sqrt(14.0)
ROT = {"frame rotation": np.array([1.0, 2, 3, 4]) / np.sqrt(30), "no frame rotation": np.array([1.0, 0, 0, 0])}
for beta, rot in [(b, r) for b in (3.7417e-4, 1e-2, 0.1) for r in ROT]:
    v = beta * direction
    print(rot)
    for ell_max, n in ((8, 240), (12, 120)):
        t = np.linspace(-40.0, 60.0, n)
        data = synthetic.chirp_modes(t, 2, ell_max, 5)
        kw = dict(boost_velocity=v, supertranslation=synthetic.real_supertranslation(0.1 * np.arange(1, 10) * (1 + 0.5j)),
                  frame_rotation=ROT[rot])
        got = scri_amd.WaveformModes(t=t, data=data, ell_min=2, ell_max=ell_max, dataType=scri_amd.h, frameType=scri_amd.Inertial,
                                     r_is_scaled_out=True, m_is_scaled_out=True, ctx=ctx).transform(**kw)
        res = {}
        for robust in (False, True):
            quat.ROBUST_POLES = robust
            e = grid_ref.transform(WM(t=t, data=data, ell_min=2, ell_max=ell_max, dataType=h), **kw)
            assert e.t.size == got.n_times
            res[robust] = np.abs(got.data - e.data).max() / max(1.0, np.abs(e.data).max())
        print(f"WM  beta={beta:.3g} l<={ell_max:2d}: |gpu - oracle| / scale   literal acos {res[False]:.2e}   atan2 {res[True]:.2e}", flush=True)
    o = smooth_abd(160, 4, 9)
    kw = dict(boost_velocity=v, supertranslation=0.05 * np.arange(1, 10) * (1 + 0.5j), frame_rotation=ROT[rot])
    g = scri_amd.AsymptoticBondiData(o.u, o.ell_max, ctx=ctx)
    g._raw_data[:] = o.raw
    got = g.transform(**kw)
    res = {}
    for robust in (False, True):
        quat.ROBUST_POLES = robust
        e = abd_ref.transform(o, **kw)
        assert e.n_times == got.n_times
        res[robust] = np.abs(got._raw_data - e.raw).max() / max(1.0, np.abs(e.raw).max())
    print(f"ABD beta={beta:.3g} l<= 4: |gpu - oracle| / scale   literal acos {res[False]:.2e}   atan2 {res[True]:.2e}", flush=True)
