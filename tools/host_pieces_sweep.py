"""Randomized check of the host path's shape rule (engine.auto_pieces): series of 16 .. 300 MB, random l_max, transformation kinds and time
axes; the default call (2 .. 20 time shards) against the one-call path on the same input.  Usage: python tools/host_pieces_sweep.py [last] [first]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import scri_amd
from scri_amd import engine, synthetic

last = int(sys.argv[1]) if len(sys.argv) > 1 else 30
first = int(sys.argv[2]) if len(sys.argv) > 2 else 0
ctx = scri_amd.Context(0)
failures, pieces_seen = 0, set()
for seed in range(first, last):
    rng = np.random.default_rng(9000 + seed)
    L = int(rng.integers(4, 17))
    nm = (L + 1) ** 2 - 4
    mb = float(rng.choice([18, 30, 50, 90, 150, 300]))
    n = int(mb * 2**20 / (nm * 16) * rng.uniform(0.8, 1.2))
    axis = str(rng.choice(["uniform", "jitter", "sxs"]))
    t = synthetic.time_axis(n, 0.1, axis)
    data = synthetic.chirp_modes(t, 2, L, 100 + seed)
    lst = int(rng.integers(1, 4))
    st = synthetic.real_supertranslation(0.05 * (rng.normal(size=(lst + 1) ** 2) + 1j * rng.normal(size=(lst + 1) ** 2)))
    fr = rng.normal(size=4)
    fr /= np.linalg.norm(fr)
    kind = str(rng.choice(["boost", "no boost", "axis boost"]))
    v = {"boost": 0.02 * rng.normal(size=3), "no boost": np.zeros(3), "axis boost": np.zeros(3)}[kind]
    if kind == "axis boost":
        fr = np.array([1.0, 0.0, 0.0, 0.0])
        v = np.array([0.0, 0.0, 0.03])
    n_theta = 2 * (L + lst) + 1
    tr = engine.make_transformation(st, fr, v, n_theta, n_theta, L)
    args = (t, data, 2, L, -2, -1, engine.BMS_TERM_H, tr)
    os.environ["SCRI_AMD_NO_PIPELINE"] = "1"
    try:
        t_one, d_one = engine.transform_modes(*args, ctx=ctx)
    finally:
        del os.environ["SCRI_AMD_NO_PIPELINE"]
    t_def, d_def = engine.transform_modes(*args, ctx=ctx)
    p = engine.auto_pieces(n, L, data.nbytes)
    pieces_seen.add(p)
    ok = t_def.shape == t_one.shape and np.array_equal(t_def, t_one)
    err = np.abs(d_def - d_one).max() / np.abs(d_one).max() if ok else float("nan")
    if not (ok and err < 1e-13):
        failures += 1
        print(f"FAILED seed {seed}: l <= {L} n {n} ({mb} MB) {axis} {kind} pieces {p}: shapes {t_def.shape} {t_one.shape} err {err:.2e}", flush=True)
print(f"done, {last - first} cases, shard counts seen {sorted(pieces_seen)}, failures: {failures}")

# the six-field call (engine.auto_pieces_abd): a third as many cases, 70 .. 400 MB
pieces_seen = set()
abd_failures = 0
for seed in range(first, first + max(1, (last - first) // 3)):
    rng = np.random.default_rng(9500 + seed)
    L = int(rng.integers(4, 13))
    nm = (L + 1) ** 2
    mb = float(rng.choice([70, 100, 160, 250, 400]))
    n = int(mb * 2**20 / (6 * nm * 16) * rng.uniform(0.9, 1.1))
    u, raw, _ = synthetic.abd_workload("cfg5", n_times=n, ell_max=L, axis=str(rng.choice(["uniform", "sxs"])))
    st = synthetic.real_supertranslation(0.05 * (rng.normal(size=9) + 1j * rng.normal(size=9)))
    fr = rng.normal(size=4)
    fr /= np.linalg.norm(fr)
    v = 0.01 * rng.normal(size=3) if rng.uniform() < 0.6 else np.zeros(3)
    tr = engine.make_transformation(st, fr, v, 4 * L + 1, 4 * L + 1, L)
    os.environ["SCRI_AMD_NO_PIPELINE"] = "1"
    try:
        u_one, r_one = engine.transform_abd(u, raw, L, tr, ctx=ctx)
    finally:
        del os.environ["SCRI_AMD_NO_PIPELINE"]
    u_def, r_def = engine.transform_abd(u, raw, L, tr, ctx=ctx)
    pieces_seen.add(engine.auto_pieces_abd(raw.nbytes))
    ok = u_def.shape == u_one.shape and np.array_equal(u_def, u_one)
    err = np.abs(np.asarray(r_def) - np.asarray(r_one)).max() / np.abs(np.asarray(r_one)).max() if ok else float("nan")
    if not (ok and err < 1e-13):
        abd_failures += 1
        print(f"FAILED six fields, seed {seed}: l <= {L} n {n} ({mb} MB) boost {bool(np.any(v))}: err {err:.2e}", flush=True)
print(f"six fields: done, shard counts seen {sorted(pieces_seen)}, failures: {abd_failures}")
sys.exit(1 if failures or abd_failures else 0)
