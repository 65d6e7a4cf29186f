"""DIAGNOSIS: boost-free transformations through the separable synthesis vs the dense product (SCRI_AMD_NO_SEPARABLE_SYNTHESIS)."""
import os, sys
import numpy as np
import scri_amd
from scri_amd import synthetic
from oracle.containers import h, psi4

ctx = scri_amd.Context(0)
ctx.enable_timing(True)


def wm(t, data, lmax, dt=h):
    return scri_amd.WaveformModes(t=t, data=data, ell_min=2, ell_max=lmax, dataType=dt, frameType=scri_amd.Inertial,
                                  r_is_scaled_out=True, m_is_scaled_out=True, ctx=ctx)


def run(t, data, lmax, kw, dt=h):
    out = wm(t, data, lmax, dt).transform(**kw)
    tm = ctx.get_timing(reset=True)
    return out, {k: round(v[0], 3) for k, v in tm.items() if v[1]}


rng = np.random.default_rng(3)
q = np.array([1.0, 2, 3, 4]) / np.sqrt(30)
for name, lmax, n in (("cfg2", 8, 3001), ("cfg2", 8, 100000), ("cfg3", 16, 100000)):
    t, data, spec = synthetic.workload(name, n_times=n)
    st = np.asarray(spec["kwargs"]["supertranslation"])
    for label, kw in (("supertranslation only", dict(supertranslation=st)),
                      ("supertranslation + frame rotation", dict(supertranslation=st, frame_rotation=q))):
        for dt in (h, psi4):
            ctx.option("NO_SEPARABLE_SYNTHESIS", 1)
            ref, tr = run(t, data, lmax, kw, dt)
            ctx.option("NO_SEPARABLE_SYNTHESIS", 0)
            got, tg = run(t, data, lmax, kw, dt)
            got, tg = run(t, data, lmax, kw, dt)
            err = np.abs(got.data - ref.data).max() / np.abs(ref.data).max()
            print(name, n, label, "h" if dt == h else "psi4", f"rel err {err:.1e}", flush=True)
            print("   dense", tr, "\n   separable", tg, flush=True)
