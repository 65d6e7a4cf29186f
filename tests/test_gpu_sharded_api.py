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


def test_group_keyword_on_the_rccl_backend_single_rank(tmp_path):
    """The same calls on the REAL backend (nccl = RCCL): a one-rank group is all a one-GPU box allows, but it runs what the gloo
    groups above cannot -- the gather of the time axis through RCCL's object collectives, a host-resident series staged on the
    rank's GPU, the stream ordering between torch and the engine without a device synchronisation."""
    code = r'''
import os, sys
sys.path.insert(0, %r)
import numpy as np, torch, torch.distributed as dist
import scri_amd
from scri_amd import synthetic, sharding, engine
os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = "29541"
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
t, data, spec = synthetic.workload("cfg3", n_times=5000)
kw = dict(spec["kwargs"]); nm = 9 * 9 - 4
def series():
    return scri_amd.WaveformModes(t=t, data=np.ascontiguousarray(data[:, :nm]), ell_min=2, ell_max=8, dataType=scri_amd.h,
                                  frameType=scri_amd.Inertial, r_is_scaled_out=True, m_is_scaled_out=True)
ref = series().transform(**kw)
g = dist.group.WORLD
host = series().transform(group=g, **kw)
assert not host.is_device_resident and np.array_equal(host.t, ref.t) and np.array_equal(host.data, ref.data)
dev = series().to_device().transform(group=g, **kw)
assert dev.is_device_resident and np.array_equal(dev.data, ref.data)
# the object a repeated caller keeps, on a side stream shared by torch and the engine (bench.py's arrangement)
s = torch.cuda.Stream(); torch.cuda.set_stream(s)
ctx = scri_amd.Context(0, stream=s.cuda_stream)
tr = engine.make_transformation(kw["supertranslation"], kw["frame_rotation"], kw["boost_velocity"], 21, 21, 8)
st = sharding.ShardedTransform("modes", t, tr, 2, 8, -2, -1, engine.BMS_TERM_H, ctx=ctx)
rows = torch.from_numpy(np.ascontiguousarray(data[:, :nm])).cuda()
for _ in range(3):
    t_out, out, first = st(rows)
torch.cuda.synchronize()
assert first == st.window[0] and np.array_equal(t_out, ref.t) and np.abs(out.cpu().numpy() - ref.data).max() < 1e-14 * np.abs(ref.data).max()
u = np.arange(1500) * 0.1
abd = scri_amd.AsymptoticBondiData(u, 3)
rng = np.random.default_rng(2)
for i, sp in enumerate((2, 1, 0, -1, -2, 2)):
    a = (rng.normal(size=16) + 1j * rng.normal(size=16)) * np.exp(0.05j * u[:, None]); a[:, : sp * sp] = 0
    abd._raw_data[i] = a
kw_abd = dict(supertranslation=np.array([0.3, 0, 0.05, 0], dtype=complex), boost_velocity=[2e-3, -1e-3, 3e-3])
r0 = abd.transform(**kw_abd); r1 = abd.transform(group=g, **kw_abd)
assert np.array_equal(r1.t, r0.t) and np.abs(r1._raw_data - r0._raw_data).max() < 1e-14 * max(1.0, np.abs(r0._raw_data).max())
dist.destroy_process_group()
print("ok")
''' % os.path.dirname(HERE)
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600,
                         env=dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0"))
    assert out.returncode == 0 and "ok" in out.stdout.split(), (out.stdout[-1500:], out.stderr[-3000:])
