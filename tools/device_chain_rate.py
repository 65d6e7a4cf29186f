"""cfg2 as a chain on one object (rotation by a rotor series, then supertranslation + frame rotation): WaveformModes with its
weights in host memory vs resident in HBM (`to_device()`)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import scri_amd
from scri_amd import synthetic

ctx = scri_amd.Context(0)
t, data, spec = synthetic.workload("cfg2")
kw = {k: v for k, v in spec["kwargs"].items() if k in ("supertranslation", "frame_rotation")}
R = synthetic.rotor_series(t, spec.get("rotation_seed", 4))


def wm():
    return scri_amd.WaveformModes(t=t, data=data.copy(), ell_min=2, ell_max=8, dataType=scri_amd.h, frameType=scri_amd.Inertial,
                                  r_is_scaled_out=True, m_is_scaled_out=True, ctx=ctx)


def chain(w):
    w.rotate_decomposition_basis(R)
    w.frameType = scri_amd.Inertial
    return w.transform(**kw)


for label, prep in (("host-resident", lambda w: w), ("device-resident", lambda w: w.to_device())):
    objs = [prep(wm()) for _ in range(8)]
    chain(objs[0]); chain(objs[1])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    outs = [chain(w) for w in objs[2:]]
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / len(outs)
    print(f"{label}: {dt * 1e3:.2f} ms per (rotation series + transformation) of {t.size} steps, l <= 8")
