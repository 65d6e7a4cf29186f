# Needs the probe build of the library (its switch does not exist in the default one; scri_amd/csrc/env.h):
#   make -C scri_amd/csrc PROBES=1 && export SCRI_AMD_LIB_PATH=$PWD/scri_amd/libscri_amd_probes.so
"""MEASUREMENT: throughput of the cfg3 transform with 1, 2 and 3 transforms in flight (one context + one host thread each, every
context on its own stream; ctypes releases the GIL): does the HBM-bound tail of transform k (back substitution, analysis) hide
under the MFMA-bound synthesis of transform k + 1?  SCRI_AMD_ZGEMM_ST_ROWS_LOG2 selects the GEMM's super-tile shape (6: 64 x 1,
least time alone, most L2 misses; 5: 32 x 2; 3: 8 x 8, fewest misses)."""
import os, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from scri_amd import _lib, engine, synthetic

t, data, spec = synthetic.workload("cfg3")
kw, L = spec["kwargs"], spec["ell_max"]
nth = 2 * (L + 2) + 1
tr = engine.make_transformation(kw["supertranslation"], kw["frame_rotation"], kw["boost_velocity"], nth, nth, L)
nm = data.shape[1]
reps = 40


def worker(ctx, src, dst, n, barrier, out, i):
    for _ in range(3):
        engine.transform_modes(t, src.data_ptr(), 2, L, -2, -1, engine.BMS_TERM_H, tr, ctx=ctx, device=True, ld=nm, out_ptr=dst.data_ptr())
    barrier.wait()
    t0 = time.perf_counter()
    for _ in range(n):
        engine.transform_modes(t, src.data_ptr(), 2, L, -2, -1, engine.BMS_TERM_H, tr, ctx=ctx, device=True, ld=nm, out_ptr=dst.data_ptr())
    out[i] = time.perf_counter() - t0


for k in (1, 2, 3):
    ctxs = [_lib.Context(0) for _ in range(k)]
    srcs = [torch.from_numpy(data).cuda() for _ in range(k)]
    dsts = [torch.empty_like(s) for s in srcs]
    torch.cuda.synchronize()
    barrier = threading.Barrier(k)
    out = [0.0] * k
    th = [threading.Thread(target=worker, args=(ctxs[i], srcs[i], dsts[i], reps, barrier, out, i)) for i in range(k)]
    t0 = time.perf_counter()
    [x.start() for x in th]
    [x.join() for x in th]
    wall = max(out)
    print(f"{k} in flight: {wall / reps * 1e3 / k:.3f} ms per transform (aggregate), {k * reps * t.size / wall:.4g} timesteps/s", flush=True)
    del ctxs, srcs, dsts
