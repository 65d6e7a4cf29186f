"""Does a row's rotated value depend on where the row sits in the launch?  One series rotated whole, then as two serial launches cut at several rows,
then dealt over two contexts: the last bit moves with the launch geometry (5e-16 on O(1) data; rows beyond 16384 here), never more.
Behind the rounding bar of tests/test_gpu_multi_device.py::test_rotations_dealt_over_contexts_equal_the_one_context_call."""
import sys, os
sys.path.insert(0, os.getcwd())
import numpy as np
from scri_amd import _lib, engine, synthetic
ctx = _lib.Context(0)
t, data, _ = synthetic.workload("cfg3", n_times=30001)
rng = np.random.default_rng(3)
R = rng.normal(size=(t.size, 4)); R /= np.linalg.norm(R, axis=1)[:, None]
sp = np.stack([R[:, 0] + 1j * R[:, 3], R[:, 2] + 1j * R[:, 1]], axis=1)
one = engine.rotate_series(data.copy(), 2, 16, sp, ctx=ctx)
two = engine.rotate_series(data.copy(), 2, 16, sp, ctx=ctx)
print("repeat identical:", np.array_equal(one, two))
for cut in (15000, 15008, 16000, 1, 17):
    d = data.copy()
    engine.rotate_series(d[:cut], 2, 16, sp[:cut], ctx=ctx)
    engine.rotate_series(d[cut:], 2, 16, sp[cut:], ctx=ctx)
    diff = np.abs(d - one)
    bad = np.nonzero(diff.max(axis=1))[0]
    print("serial cut", cut, "identical:", np.array_equal(d, one), "max diff", diff.max(), "bad rows", bad[:5], bad.size)
d = engine.rotate_series(data.copy(), 2, 16, sp, ctx=ctx, devices=[0, 0])
diff = np.abs(d - one); bad = np.nonzero(diff.max(axis=1))[0]
print("threads [0,0]:", np.array_equal(d, one), diff.max(), bad[:5], bad.size)
