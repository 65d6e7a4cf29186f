"""DIAGNOSIS: the same transformation several times, bitwise equal?  WM flavour (cfg2 / cfg3 shapes at reduced length),
ABD flavour (l_max = 6), each also against the one-role analysis kernel."""
import os, sys
import numpy as np
import scri_amd
from scri_amd import synthetic
from oracle.containers import h
from tests.test_oracle_charges import kerr_schild_abd

ctx = scri_amd.Context(0)
bad = 0


def wm(t, data, lmax):
    return scri_amd.WaveformModes(t=t, data=data, ell_min=2, ell_max=lmax, dataType=h, frameType=scri_amd.Inertial,
                                  r_is_scaled_out=True, m_is_scaled_out=True, ctx=ctx)


for name, lmax, n in (("cfg2", 8, 2001), ("cfg2", 8, 30000), ("cfg3", 16, 1501), ("cfg3", 16, 4000), ("cfg3", 16, 20001)):
    t, data, spec = synthetic.workload(name, n_times=n)
    ctx.option("NO_SPLIT_ANALYSIS", 0)
    outs = [wm(t, data, lmax).transform(**spec["kwargs"]).data.copy() for _ in range(5)]
    ctx.option("NO_SPLIT_ANALYSIS", 1)
    ref = wm(t, data, lmax).transform(**spec["kwargs"]).data
    same = all(np.array_equal(outs[0], x) for x in outs[1:])
    err = max(np.abs(x - ref).max() for x in outs) / np.abs(ref).max()
    flag = "" if same and err < 1e-13 else "  <<<<<<"
    bad += bool(flag)
    print(name, n, "deterministic" if same else "DIFFERS", f"{err:.2e}", flag, flush=True)

u = np.linspace(-400, 400, num=1601)
a = scri_amd.AsymptoticBondiData(u, 6, ctx=ctx)
a._raw_data[:] = kerr_schild_abd(2.0, 0.456, 6, u)
st = np.array([0.0, 3e-2 - 1j * 5e-3, 1e-3, -3e-2 - 1j * 5e-3, 2e-4 + 1j * 1e-4, 1j * 3e-3, 1e-2, 1j * 3e-3, 2e-4 - 1j * 1e-4])
kw = dict(supertranslation=st, frame_rotation=np.array([1.0, 2, 3, 4]) / np.sqrt(30), boost_velocity=np.array([2e-4, -3e-5, 2e-4]))
ctx.option("NO_SPLIT_ANALYSIS", 0)
outs = [a.transform(**kw)._raw_data.copy() for _ in range(8)]
ctx.option("NO_SPLIT_ANALYSIS", 1)
ref = a.transform(**kw)._raw_data
same = all(np.array_equal(outs[0], x) for x in outs[1:])
err = max(np.abs(x - ref).max() for x in outs) / np.abs(ref).max()
print("abd", "deterministic" if same else "DIFFERS", f"{err:.2e}", flush=True)
bad += (not same) or err > 1e-13
print("bad:", bad)
