"""The sharded engine behind the scri API (north star: "shard the time axis ... expose it behind scri.WaveformModes.transform"):
`w.transform(group=g, **kw)` / `abd.transform(group=g, **kw)` on rank-local series, two and three ranks of a gloo group sharing
the one GPU of the box, against the single-context transform of the whole series (scri/waveform_modes.py:705-719,
scri/asymptotic_bondi_data/transformations.py:199-431)."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.parametrize("world", [2, 3])
@pytest.mark.timeout(900)
def test_group_keyword_equals_single_gpu(ctx, tmp_path, world):
    import scri_amd
    from scri_amd import synthetic
    from tests.test_gpu_sharding import _abd_case

    port = _free_port()
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, os.path.join(HERE, "helpers", "sharded_api_worker.py"), str(tmp_path)], env=env,
                                      stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    outs = [p.communicate(timeout=800)[0] for p in procs]
    for p, o in zip(procs, outs):
        assert p.returncode == 0, o[-3000:]
    parts = [np.load(tmp_path / f"rank{r}.npz") for r in range(world)]

    n_times, ell_max = 6000, 8
    t, data, spec = synthetic.workload("cfg3", n_times=n_times)
    kw = dict(spec["kwargs"])
    kw["boost_velocity"] = np.array([1.0, 2.0, 3.0]) * 1e-3
    nm = (ell_max + 1) ** 2 - 4
    w = scri_amd.WaveformModes(t=t, data=np.ascontiguousarray(data[:, :nm]), ell_min=2, ell_max=ell_max, dataType=scri_amd.h,
                               frameType=scri_amd.Inertial, r_is_scaled_out=True, m_is_scaled_out=True, ctx=ctx)
    ref = w.transform(**kw)
    scale = np.abs(ref.data).max()
    for tag in ("host", "device", "overlap", "columns"):
        t_sh = np.concatenate([p[f"wm_{tag}_t"] for p in parts])
        d_sh = np.concatenate([p[f"wm_{tag}_d"] for p in parts])
        assert np.array_equal(t_sh, ref.t), tag
        assert np.abs(d_sh - ref.data).max() < (2e-14 if tag == "columns" else 1e-14) * scale, tag
    # host and device-resident series take the same shard call: identical bits
    for p in parts:
        assert np.array_equal(p["wm_host_d"], p["wm_device_d"])

    u, raw, tr, L = _abd_case(n=3000, ell_max=4)
    abd = scri_amd.AsymptoticBondiData(u, L, ctx=ctx)
    abd._raw_data[:] = raw
    st = np.zeros(9, dtype=complex)
    st[0], st[2], st[6] = 0.3, 0.05, 0.02
    ref = abd.transform(supertranslation=st, frame_rotation=[0.9, 0.1, -0.3, 0.2], boost_velocity=[2e-3, -1e-3, 3e-3], working_ell_max=2 * L + 2)
    for tag in ("host", "device"):
        u_sh = np.concatenate([p[f"abd_{tag}_u"] for p in parts])
        r_sh = np.concatenate([p[f"abd_{tag}_raw"] for p in parts], axis=1)
        assert np.array_equal(u_sh, ref.t), tag
        assert np.abs(r_sh - ref._raw_data).max() < 1e-14 * max(1.0, np.abs(ref._raw_data).max()), tag
