"""Rotation path (scri/rotations.py:346-392) on MI355X: time-series Wigner-D rotation of modes resident in HBM.
Prints one JSON line with the HBM roofline of rotate_modes_kernel and the C port of the numba kernel beside it."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from scri_amd import _lib, engine

ell_min, ell_max, n = 2, int(sys.argv[1]) if len(sys.argv) > 1 else 16, int(sys.argv[2]) if len(sys.argv) > 2 else 1_000_000
nm = (ell_max + 1) ** 2 - ell_min**2
rng = np.random.default_rng(0)
data = torch.from_numpy(rng.normal(size=(n, nm)) + 1j * rng.normal(size=(n, nm))).cuda()
R = rng.normal(size=(n, 4)); R /= np.linalg.norm(R, axis=1)[:, None]
sp = torch.from_numpy(np.stack([R[:, 0] + 1j * R[:, 3], R[:, 2] + 1j * R[:, 1]], axis=1)).cuda()
ctx = _lib.Context(0); ctx.enable_timing(True)
for _ in range(3):
    engine.rotate_device(data.data_ptr(), n, nm, ell_min, ell_max, spinors_ptr=sp.data_ptr(), ctx=ctx)
ctx.get_timing(reset=True)
K = 10
t0 = time.perf_counter()
for _ in range(K):
    engine.rotate_device(data.data_ptr(), n, nm, ell_min, ell_max, spinors_ptr=sp.data_ptr(), ctx=ctx)
ctx.synchronize()
wall = (time.perf_counter() - t0) / K
ms = ctx.get_timing()["rotate"][0] / K
bytes_per_step = 2 * 16 * nm + 32
line = {"metric": "timesteps/s, time-series rotation of modes", "ell": [ell_min, ell_max], "n_times": n, "value": n / wall,
        "kernel_ms": ms, "roofline": {"bound": "hbm", "achieved": n * bytes_per_step / (ms * 1e-3) / 1e9, "peak": 8000.0,
                                      "unit": "GB/s", "frac": n * bytes_per_step / (ms * 1e-3) / 8e12, "bytes_per_step": bytes_per_step}}
if os.environ.get("CPU_BASELINE", "1") == "1":
    from oracle import rotate_port
    ns = 4000
    d = np.ascontiguousarray(data[:ns].cpu().numpy()); s = np.ascontiguousarray(sp[:ns].cpu().numpy())
    t0 = time.perf_counter(); rotate_port.rotate_by_series(d, s, ell_min, ell_max); dt = time.perf_counter() - t0
    line["cpu_baseline"] = {"value": ns / dt, "unit": "timesteps/s", "cores": 1, "kind": "port",
                            "sample": f"{ns} steps, oracle/rotate_port.c (scalar C port of the numba kernel incl. per-step Wigner-D), {dt:.1f} s"}
print(json.dumps(line))
