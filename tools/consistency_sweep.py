"""Seeded sweep at MEDIUM size (4e3 .. 6e4 time steps, l_max 2 .. 16, three kinds of time axis, boosts up to 0.3 c) of the
properties the full-size tests check on the BASELINE.json shapes only (tests/test_gpu_full_size.py):

  * a small work space (3 .. 12 chunks of the time axis) gives the result of one chunk,
  * weights resident in HBM (to_device) give the host path's result (bit for bit below the size at which host arrays travel in pieces),
  * the shards of sharding.plan (2 .. 8 ranks), each run from its own rows + halo, reassemble to the whole,
  * random windows of the output match the oracle run on slices of the input (the slice is widened by the boost's time skew).

tests/test_gpu_fuzz.py covers the parameter space against the oracle on SMALL series (n <= 260, l <= 7); this sweep is where
chunk seams, tile windows of the evaluating product and halos meet random skews.  Usage:
    python tools/consistency_sweep.py [last_seed [first_seed]]        (prints failures; exit code = their number)
SWEEP_ABD=1: the same for AsymptoticBondiData (six fields with their mixing; 2e3 .. 1.2e4 steps, l <= 8)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

import scri_amd
from oracle import quat
from oracle import waveform_grid_ref as grid_ref
from oracle.containers import WM, SpinWeights, h, news, psi4, sigma
from scri_amd import engine, sharding, synthetic
from scri_amd.waveform_grid import process_transformation_kwargs
from tests.test_gpu_transform_modes import real_supertranslation

quat.ROBUST_POLES = True
DONE = {"chunked runs": 0, "device-resident runs": 0, "... of them bit for bit the host run": 0, "work-space limit reported too small": 0, "sharded runs": 0, "shards": 0, "column partitions": 0, "oracle windows": 0}


def _gpu(t, data, ell_max, dataType, ctx):
    return scri_amd.WaveformModes(t=t, data=data, ell_min=2, ell_max=ell_max, dataType=dataType, frameType=scri_amd.Inertial,
                                  r_is_scaled_out=True, m_is_scaled_out=True, ctx=ctx)


def one(seed, ctx):
    rng = np.random.default_rng(77_000 + seed)
    ell_max = int(rng.integers(2, 17))
    n = int(rng.integers(4_000, 60_000 if ell_max <= 10 else 30_000))
    dt = 0.1
    axis = ["uniform", "jitter", "sxs"][int(rng.integers(3))]
    t = synthetic.time_axis(n, dt, axis)
    data = synthetic.chirp_modes(t, 2, ell_max, 10 + seed)
    dataType = [h, sigma, news, psi4][int(rng.integers(4))]
    kw = {}
    if rng.random() < 0.85:
        kw["supertranslation"] = real_supertranslation(int(rng.integers(1, 4)), int(rng.integers(1 << 30)), 10.0 ** rng.uniform(-2.0, 0.3))
    if rng.random() < 0.7:
        q = rng.normal(size=4)
        kw["frame_rotation"] = q / np.linalg.norm(q)
    beta = 0.0
    if rng.random() < 0.8:
        v = rng.normal(size=3)
        # (time skew of a direction: up to beta |u| / dt rows; keep it below ~2 500 rows so that the oracle's slices stay cheap)
        beta = min(10.0 ** rng.uniform(-4, np.log10(0.3)), 2_500 * dt / max(abs(t[0]), abs(t[-1])))
        kw["boost_velocity"] = v / np.linalg.norm(v) * beta
    what = f"seed {seed}: n={n} l<={ell_max} {axis} type={dataType} beta={beta:.2e} keys={sorted(kw)}"
    whole = _gpu(t, data, ell_max, dataType, ctx).transform(**kw)
    scale = max(1.0, np.abs(whole.data).max())
    bad = []
    if whole.n_times < 50:
        return bad, what + " (window too small, skipped)"

    # --- chunks of the time axis
    n_pix = (2 * (ell_max + 3) + 1) ** 2
    per_row = 5.0 * 2 * 64 * ((n_pix + 63) // 64) * 8.0
    chunks = int(rng.integers(3, 13))
    limit = int(max(per_row * (n / chunks + 300), 24 << 20))
    small = scri_amd.Context(0, workspace_limit=limit)
    try:
        part = _gpu(t, data, ell_max, dataType, small).transform(**kw)
        if part.n_times != whole.n_times or not np.array_equal(part.t, whole.t):
            bad.append("chunks: time axis differs")
        else:
            DONE["chunked runs"] += 1
            err = np.abs(part.data - whole.data).max()
            if not err < 1e-13 * scale:
                bad.append(f"chunks ({chunks} wanted): {err / scale:.2e}")
    except scri_amd.BMSError as e:  # (a limit below what the halos need is reported, not a failure)
        DONE["work-space limit reported too small"] += 1
        if "work space limit" not in str(e):
            bad.append(f"chunks: {e}")
    finally:
        del small

    # --- weights resident in HBM (to_device): the same kernels on the caller's device memory
    dev = _gpu(t, data.copy(), ell_max, dataType, ctx).to_device().transform(**kw)
    DONE["device-resident runs"] += 1
    if not dev.is_device_resident:
        bad.append("device-resident input gave a host result")
    elif dev.n_times != whole.n_times or not np.array_equal(dev.t, whole.t):
        bad.append("device-resident run: time axis differs from the host run's")
    else:
        # (host arrays of 64 MB and more travel in pieces, each with its own tiles of the spline solve: equal to rounding, not bit for bit)
        err = np.abs(dev.data - whole.data).max()
        DONE["... of them bit for bit the host run"] += int(err == 0.0)
        if not err < 1e-13 * scale:
            bad.append(f"device-resident run: {err / scale:.2e} from the host run")

    # --- shards (single-field type h only: the engine call below is the one sharding.py makes)
    if dataType == h:
        # (the defaults of the class: grid of 2 (l_max + l_max of the supertranslation) + 1 points each way)
        st, _, _, n_theta, n_phi, rot, boost, _ = process_transformation_kwargs(ell_max, **kw)
        tr = engine.make_transformation(st, rot, boost, n_theta, n_phi, ell_max)
        ranks = int(rng.integers(2, 9))
        have, need, window = sharding.plan(t, tr, ranks)
        if window[1] - window[0] != whole.n_times:
            bad.append(f"shards: window {window} against {whole.n_times} outputs")
        else:
            row = 0
            DONE["sharded runs"] += 1
            for r in range(ranks):
                DONE["shards"] += 1
                ext = data[need[r][0]:need[r][1]]
                tp, dp, first = engine.transform_modes(t, ext, 2, ell_max, -2, -1, engine.BMS_TERM_H, tr, ctx=ctx,
                                                       shard=(need[r][0], ext.shape[0], have[r][0], have[r][1]))
                # (a rank whose outputs all fall outside the output window returns no rows)
                if (tp.shape[0] and first != window[0] + row) or not np.array_equal(tp, whole.t[row:row + tp.shape[0]]):
                    bad.append(f"shard {r} of {ranks}: rows misplaced")
                    break
                err = np.abs(dp - whole.data[row:row + dp.shape[0]]).max() if dp.shape[0] else 0.0
                if not err < 1e-13 * scale:
                    bad.append(f"shard {r} of {ranks}: {err / scale:.2e}")
                row += dp.shape[0]
            if not bad and row != whole.n_times:
                bad.append(f"shards: {row} rows against {whole.n_times}")
        # halos beyond a quarter of a shard: sharding.choose_partition turns to grid-column parts, whose contributions add up
        if sharding.choose_partition(have, need) == "columns":
            DONE["column partitions"] += 1
            total = None
            for p in range(ranks):
                tp, dp, first = engine.transform_modes(t, data, 2, ell_max, -2, -1, engine.BMS_TERM_H, tr, ctx=ctx, shard=(0, n, 0, n, p, ranks))
                if not np.array_equal(tp, whole.t):
                    bad.append(f"column part {p} of {ranks}: time axis differs")
                    break
                total = dp.copy() if total is None else total + dp
            else:
                err = np.abs(total - whole.data).max()
                if not err < 1e-13 * scale:
                    bad.append(f"column parts ({ranks}): {err / scale:.2e}")

    # --- oracle windows
    skew_rows = int(np.ceil(beta * max(abs(t[0]), abs(t[-1])) / np.diff(t).min())) if beta else 0
    half = 200 + int(1.3 * skew_rows) + 70
    if 2 * half + 50 < n and (2 * half) * n_pix < 6e6:
        for _ in range(2):
            i0 = int(rng.integers(0, n - 2 * half))
            sl = slice(i0, i0 + 2 * half)
            e = grid_ref.transform(WM(t=t[sl], data=data[sl], ell_min=2, ell_max=ell_max, dataType=dataType), **kw)
            if e.t.shape[0] < 140:
                continue
            keep = e.t[60:-60]
            keep = keep[(keep >= whole.t[0]) & (keep <= whole.t[-1])]
            if keep.size == 0:
                continue
            gi = np.searchsorted(whole.t, keep - 1e-9)
            gi = np.minimum(gi, whole.t.shape[0] - 1)
            if not np.abs(whole.t[gi] - keep).max() < 1e-9 * max(1.0, abs(keep).max()):
                bad.append(f"oracle window at {i0}: time samples differ by {np.abs(whole.t[gi] - keep).max():.2e}")
                continue
            DONE["oracle windows"] += 1
            sel = np.isin(e.t, keep)
            err = np.abs(whole.data[gi] - e.data[sel]).max()
            if not err < 1e-12 * max(1.0, np.abs(e.data).max()):
                bad.append(f"oracle window at {i0}: {err:.2e}")
    return bad, what


def one_abd(seed, ctx):
    """AsymptoticBondiData (all six fields, Horner mixing): whole = chunks = shards, and windows against the oracle"""
    from oracle import abd_ref
    from oracle.containers import ABD
    from scri_amd.asymptotic_bondi_data import _process_transformation_kwargs as abd_kwargs

    rng = np.random.default_rng(99_000 + seed)
    L = int(rng.integers(2, 9))
    n = int(rng.integers(2_000, 12_000))
    axis = ["uniform", "jitter", "sxs"][int(rng.integers(3))]
    u, raw, _ = synthetic.abd_workload("cfg5", n_times=n, ell_max=L, axis=axis)
    kw = {}
    if rng.random() < 0.85:
        kw["supertranslation"] = real_supertranslation(int(rng.integers(1, 4)), int(rng.integers(1 << 30)), 10.0 ** rng.uniform(-2.0, 0.0))
    if rng.random() < 0.7:
        q = rng.normal(size=4)
        kw["frame_rotation"] = q / np.linalg.norm(q)
    beta = 0.0
    if rng.random() < 0.8:
        v = rng.normal(size=3)
        beta = min(10.0 ** rng.uniform(-4, np.log10(0.3)), 1_200 * 0.1 / max(abs(u[0]), abs(u[-1])))
        kw["boost_velocity"] = v / np.linalg.norm(v) * beta
    what = f"ABD seed {seed}: n={n} l<={L} {axis} beta={beta:.2e} keys={sorted(kw)}"
    rot, boost, st, wl, out_l = abd_kwargs(L, **dict(kw))
    tr = engine.make_transformation(st, rot, boost, 2 * wl + 1, 2 * wl + 1, out_l)
    u_w, raw_w = engine.transform_abd(u, raw, L, tr, ctx=ctx)
    bad = []
    if u_w.shape[0] < 50:
        return bad, what + " (window too small, skipped)"
    scale = max(1.0, np.abs(raw_w).max())

    # --- chunks
    n_pix = (2 * wl + 1) ** 2
    per_row = 16.0 * 2 * 64 * ((n_pix + 63) // 64) * 8.0
    chunks = int(rng.integers(3, 9))
    small = scri_amd.Context(0, workspace_limit=int(max(per_row * (n / chunks + 300), 32 << 20)))
    try:
        u_c, raw_c = engine.transform_abd(u, raw, L, tr, ctx=small)
        DONE["chunked runs"] += 1
        if not np.array_equal(u_c, u_w):
            bad.append("chunks: time axis differs")
        elif not np.abs(raw_c - raw_w).max() < 1e-13 * scale:
            bad.append(f"chunks ({chunks} wanted): {np.abs(raw_c - raw_w).max() / scale:.2e}")
    except scri_amd.BMSError as e:
        DONE["work-space limit reported too small"] += 1
        if "work space limit" not in str(e):
            bad.append(f"chunks: {e}")
    finally:
        del small

    # --- shards
    ranks = int(rng.integers(2, 7))
    have, need, window = sharding.plan(u, tr, ranks)
    if window[1] - window[0] != u_w.shape[0]:
        bad.append(f"shards: window {window} against {u_w.shape[0]} outputs")
    else:
        row = 0
        DONE["sharded runs"] += 1
        for r in range(ranks):
            DONE["shards"] += 1
            ext = np.ascontiguousarray(raw[:, need[r][0]:need[r][1]])
            up, rp, first = engine.transform_abd(u, ext, L, tr, ctx=ctx, shard=(need[r][0], ext.shape[1], have[r][0], have[r][1]))
            if (up.shape[0] and first != window[0] + row) or not np.array_equal(up, u_w[row:row + up.shape[0]]):
                bad.append(f"shard {r} of {ranks}: rows misplaced")
                break
            err = np.abs(rp - raw_w[:, row:row + up.shape[0]]).max() if up.shape[0] else 0.0
            if not err < 1e-13 * scale:
                bad.append(f"shard {r} of {ranks}: {err / scale:.2e}")
            row += up.shape[0]
        if not bad and row != u_w.shape[0]:
            bad.append(f"shards: {row} rows against {u_w.shape[0]}")

    # --- one oracle window (the bar of tests/test_gpu_full_size.py::_abd_window_check: 1e-12 of the scale + the rounding of the
    # reference's own abscissae k (u - alpha), which a time derivative carries into the result)
    skew_rows = int(np.ceil(beta * max(abs(u[0]), abs(u[-1])) / np.diff(u).min())) if beta else 0
    half = 160 + int(1.3 * skew_rows) + 70
    if 2 * half + 50 < n and (2 * half) * n_pix < 2.5e6:
        i0 = int(rng.integers(0, n - 2 * half))
        e = abd_ref.transform(ABD(u[i0:i0 + 2 * half], raw[:, i0:i0 + 2 * half], L), **kw)
        if e.n_times >= 140:
            keep = e.u[60:-60]
            keep = keep[(keep >= u_w[0]) & (keep <= u_w[-1])]
            if keep.size:
                gi = np.minimum(np.searchsorted(u_w, keep - 1e-9), u_w.shape[0] - 1)
                if not np.abs(u_w[gi] - keep).max() < 1e-9 * max(1.0, abs(keep).max()):
                    bad.append(f"oracle window at {i0}: time samples differ")
                else:
                    DONE["oracle windows"] += 1
                    sel = np.isin(e.u, keep)
                    osc = max(1.0, np.abs(e.raw).max())
                    for f in range(6):
                        noise = 8 * np.finfo(float).eps * np.abs(e.u).max() * np.abs(np.gradient(e.raw[f], e.u, axis=0)).max()
                        err = np.abs(raw_w[f][gi] - e.raw[f][sel]).max()
                        if not err < 1e-12 * osc + noise:
                            bad.append(f"oracle window at {i0}, field {f}: {err:.2e} (bar {1e-12 * osc + noise:.2e})")
    return bad, what


def main():
    last = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    first = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    ctx = scri_amd.Context(0)
    failures = 0
    fn = one_abd if os.environ.get("SWEEP_ABD") else one
    for seed in range(first, last):
        try:
            bad, what = fn(seed, ctx)
        except Exception as e:  # noqa: BLE001  (a sweep: report and go on)
            bad, what = [f"{type(e).__name__}: {str(e)[:300]}"], f"seed {seed}"
        if bad:
            failures += 1
            print("FAILED", what, "|", "; ".join(bad), flush=True)
        elif os.environ.get("SWEEP_VERBOSE"):
            print("ok", what, flush=True)
    print("checked:", DONE)
    print("done, failures:", failures)
    return failures


if __name__ == "__main__":
    sys.exit(min(main(), 100))
