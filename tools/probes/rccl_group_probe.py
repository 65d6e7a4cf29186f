"""The body of tests/test_gpu_sharded_api.py::test_group_keyword_on_the_rccl_backend_single_rank as a script (prints instead of asserting on the
shard-call comparison): w.transform(group=...) / abd.transform(group=...) and a ShardedTransform on a one-rank nccl (= RCCL) group."""
import os, sys
sys.path.insert(0, "/root/repo")
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
print("first", first, st.window, t_out.shape, ref.t.shape, np.array_equal(t_out, ref.t), out.shape, ref.data.shape, float(np.abs(out.cpu().numpy() - ref.data).max()) if out.shape == ref.data.shape else None)
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
