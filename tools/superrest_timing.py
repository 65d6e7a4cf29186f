"""Wall time of the workflow the reference documents as taking "an hour or two" (docs/tutorial_abd.rst:387-388): an
AsymptoticBondiData object (Kerr, moved by a supertranslation + rotation + boost) mapped to its super rest frame.
Usage: python tools/superrest_timing.py [n_times] [ell_max] [padding_time] [--reserve]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch  # (the device-resident window allocates through torch: its import, ~3 s, is not part of the workflow's time)
import scri_amd
from tests.test_oracle_charges import kerr_schild_abd

reserve = "--reserve" in sys.argv  # warm-up: Context.reserve() (bms_ctx_reserve) before anything else
sys.argv = [a for a in sys.argv if a != "--reserve"]
n = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
ell_max = int(sys.argv[2]) if len(sys.argv) > 2 else 8
pad = float(sys.argv[3]) if len(sys.argv) > 3 else 250.0
ctx = scri_amd.Context(0)
if reserve:
    t0 = time.perf_counter()
    ctx.reserve()
    print(f"Context.reserve(): {time.perf_counter() - t0:.2f} s (once per process: one allocation that the work-space buffers are carved from)")
u = np.linspace(-1000, 1000, num=n)
a = scri_amd.AsymptoticBondiData(u, ell_max, ctx=ctx)
a._raw_data[:] = kerr_schild_abd(2.0, 0.456, ell_max, u)
st = np.array([0.0, 3e-2 - 1j * 5e-3, 1e-3, -3e-2 - 1j * 5e-3, 2e-4 + 1j * 1e-4, 1j * 3e-3, 1e-2, 1j * 3e-3, 2e-4 - 1j * 1e-4])
t0 = time.perf_counter()
moved = a.transform(supertranslation=st, frame_rotation=np.array([1.0, 2, 3, 4]) / np.sqrt(30), boost_velocity=np.array([2e-4, -3e-5, 2e-4]))
t1 = time.perf_counter()
rec, tr, errs = moved.map_to_superrest_frame(t_0=0, padding_time=pad)
t2 = time.perf_counter()
print(f"N = {n}, ell_max = {ell_max}, window = +-{pad}: transform {t1 - t0:.2f} s, map_to_superrest_frame {t2 - t1:.2f} s "
      f"({rec.t.size} output steps), rel_errs {[float(f'{e:.2e}') for e in errs]}")
# the same workflow on an object whose fields already live in HBM: nothing but the control loop's small reads crosses PCIe
moved_d = moved.to_device()
# (twice: the first full-series transformation of a process on device-resident fields grows the engine's work space to ~70 GB; on some
# boxes the driver hands out fresh memory at 40-80 ms per GB -- tools/probes/alloc_rate_probe.py: 0.94 s on one box, 4.9 s on another
# for the same first run; the second run is the workflow itself)
for label in ("first run of the process (the engine's work space grows here)", "again"):
    torch.cuda.synchronize()
    t3 = time.perf_counter()
    rec_d, tr_d, errs_d = moved_d.map_to_superrest_frame(t_0=0, padding_time=pad)
    torch.cuda.synchronize()
    t4 = time.perf_counter()
    print(f"device-resident object, {label}: map_to_superrest_frame {t4 - t3:.2f} s, rel_errs {[float(f'{e:.2e}') for e in errs_d]}, "
          f"max |difference to the host-resident result| = {np.abs(rec_d._raw_data - rec._raw_data).max():.2e}")
