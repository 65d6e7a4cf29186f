#!/usr/bin/env python
"""Benchmark of the BMS-transformation hot path on MI355X (contract: see the task description).

  python bench.py --gpus N --steps K --warmup W

N > 1 without WORLD_SIZE in the environment: this process starts its own N ranks (a child `python -m torch.distributed.run`,
spawned before anything here touches the GPU) and returns their exit code; under a launcher (WORLD_SIZE set) it is one rank.

A "step" is one full BMS transformation (supertranslation + frame rotation + boost) of one synthetic
WaveformModes series that is already resident in HBM.  N = 1: workload cfg3 of BASELINE.json / SURVEY section 8(d)
(h, ell = 2..16, 285 modes, 1e5 time steps, 37 x 37 grid).  N > 1: workload cfg4, the SAME transformation of a
1e6-step series whose time axis is sharded over the N ranks (STRONG scaling: total work fixed; BASELINE.json
configs[3] / north star): the input-mode halos are exchanged over RCCL point-to-point inside the timed region, `value`
is the whole-job rate, and before the sharded loop rank 0 transforms the whole 1e6-step series on its own GPU so that
the line carries its own single-GPU reference (`strong_scaling`).  `--workload` overrides (cfg3 with N > 1 = the weak
scaling of round 1: 1e5 steps per rank).  Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FP64_MATRIX_PEAK_TFLOPS = 78.6  # MI355X fp64 matrix (= vector) peak, AMD public spec
# the synthesis product: with the spline evaluation in its epilogue on the WaveformModes route (kernels_gemm_eval.hip; round 4), the
# plain product for AsymptoticBondiData (time-dependent mixing between synthesis and spline) and with SCRI_AMD_NO_GEMM_EVAL
DOMINANT_KERNEL = "bms::zgemm3m_mfma_kernel"


def dominant_kernel(abd):
    # (before a context exists: its default for the option is what the environment says at bms_ctx_create; the bench never changes it)
    return "bms::zgemm3m_mfma_kernel" if abd or os.environ.get("SCRI_AMD_NO_GEMM_EVAL", "0") not in ("0", "") else "bms::zgemm3m_eval_kernel"
# HBM traffic of the dominant kernel: measured BY THIS RUN where rocprofv3 is on the PATH -- before the parent process touches
# the GPU it runs two short child passes of this same script under `rocprofv3 --pmc FETCH_SIZE` / `--pmc WRITE_SIZE` (separate
# passes, as MI355X_MICROARCH.md prescribes; traffic = 2 x FETCH_SIZE + WRITE_SIZE KiB on gfx950) and reads their counter CSVs.
# Fallback: the committed summary of tools/run_profiles.sh, used only if it was taken on the same kernel sources.
PMC_SUMMARY = os.path.join(ROOT, "profiles", "r06_v_pmc_cfg3.json")


def csrc_hash():
    """sha256 over the kernel sources and the ABI header: identifies the build a profile was taken on (the GPU box has no .git)"""
    import glob
    import hashlib

    h = hashlib.sha256()
    files = sorted(glob.glob(os.path.join(ROOT, "scri_amd", "csrc", "*.hip")) + glob.glob(os.path.join(ROOT, "scri_amd", "csrc", "*.h"))
                   + [os.path.join(ROOT, "scri_amd", "csrc", "Makefile"), os.path.join(ROOT, "include", "scri_amd.h")])
    for f in files:
        h.update(os.path.basename(f).encode())
        with open(f, "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


def _pmc_counter_mean(directory, counter):
    """mean of `counter` over the launches of the dominant kernel in a rocprofv3 counter_collection.csv (first launch dropped)"""
    import csv
    import glob

    vals = []
    for f in glob.glob(os.path.join(directory, "**", "*counter_collection.csv"), recursive=True):
        with open(f) as fh:
            for row in csv.DictReader(fh):
                if DOMINANT_KERNEL.split("::")[1] in row["Kernel_Name"] and row["Counter_Name"] == counter:
                    vals.append(float(row["Counter_Value"]))
    vals = vals[1:] if len(vals) > 1 else vals
    return sum(vals) / len(vals) if vals else None


def live_pmc_traffic(argv):
    """Two child passes of this script under rocprofv3 --pmc (never combined with a trace option); called before this process
    initialises the GPU.  Returns (bytes per launch or None, note)."""
    import shutil
    import subprocess
    import tempfile

    prof = shutil.which("rocprofv3") or ("/opt/rocm/bin/rocprofv3" if os.path.exists("/opt/rocm/bin/rocprofv3") else None)
    if prof is None:
        return None, "rocprofv3 not found"
    keep = [a for a in argv if a not in ("--live-pmc",)]
    got = {}
    with tempfile.TemporaryDirectory(dir="/tmp") as tmp:
        env = dict(os.environ, TMPDIR="/tmp", SCRI_AMD_BENCH_PMC_CHILD="1")
        for counter in ("FETCH_SIZE", "WRITE_SIZE"):
            out_dir = os.path.join(tmp, counter)
            cmd = [prof, "--pmc", counter, "--output-format", "csv", "-d", out_dir, "--", sys.executable, os.path.abspath(__file__)] + keep
            try:
                r = subprocess.run(cmd, cwd="/tmp", env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=420)
            except (subprocess.TimeoutExpired, OSError) as e:
                return None, f"rocprofv3 --pmc {counter} pass failed: {type(e).__name__}"
            if r.returncode != 0:
                return None, f"rocprofv3 --pmc {counter} pass exited with {r.returncode}: {r.stdout.decode(errors='replace')[-200:]}"
            got[counter] = _pmc_counter_mean(out_dir, counter)
            if got[counter] is None:
                return None, f"no {counter} rows for {DOMINANT_KERNEL} in the pass's counter_collection.csv"
    traffic = (2.0 * got["FETCH_SIZE"] + got["WRITE_SIZE"]) * 1024.0
    return traffic, ("measured by this run: two child passes of the same command (3 timed steps) under rocprofv3 --pmc FETCH_SIZE / --pmc "
                     "WRITE_SIZE, mean over the kernel's launches; bytes = (2 x FETCH_SIZE + WRITE_SIZE) KiB (gfx950 correction of "
                     "MI355X_MICROARCH.md); reads are L2 misses, Infinity-Cache hits included")


def committed_pmc_traffic(workload, world, per_gpu):
    """HBM bytes per launch of the dominant kernel from the committed PMC summary -- only if that summary was taken on the kernel
    sources of the running build (its `csrc_hash` stamp) and for this very workload.  Returns (bytes or None, note)."""
    if workload != "cfg3" or world != 1 or per_gpu != 100_000:
        return None, "no PMC pass for this workload"
    if not os.path.exists(PMC_SUMMARY):
        return None, "no committed PMC summary"
    with open(PMC_SUMMARY) as f:
        summary = json.load(f)
    stamp = summary.get("_meta", {}).get("csrc_hash")
    name = os.path.relpath(PMC_SUMMARY, ROOT)
    if stamp != csrc_hash():
        return None, f"{name} was taken on kernel sources {stamp}, this build is {csrc_hash()}: not reported"
    for kname, counters in summary.items():  # (the kernel is a template: its name carries "<false>")
        if kname.startswith(DOMINANT_KERNEL) and "traffic_bytes" in counters:
            return counters["traffic_bytes"], f"{name} (tools/run_profiles.sh on kernel sources {stamp} = this build), 2 x FETCH_SIZE + WRITE_SIZE"
    return None, f"{name} has no row for {DOMINANT_KERNEL}"


def launch_ranks(n, argv):
    """`python bench.py --gpus N` outside a launcher: start N ranks as a CHILD process group (torch.distributed.run) before this
    process has touched the GPU, wait, hand back the exit code.  With fewer GPUs than ranks the run is a gloo dry run (ranks share
    devices, halos through the host: plumbing only, the line says so)."""
    import subprocess

    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if "SCRI_AMD_BENCH_BACKEND" not in env:
        try:
            import torch  # (device_count() does not initialise the GPU)

            if torch.cuda.device_count() < n:
                env["SCRI_AMD_BENCH_BACKEND"] = "gloo"
        except Exception:  # noqa: BLE001
            pass
    # (--standalone: the launcher binds a free port itself and hands it to the ranks as MASTER_PORT -- a port found here by
    # bind(0) + close could be taken by another process before the launcher listens on it)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--standalone", "--local-addr", "127.0.0.1", "--nnodes=1", f"--nproc-per-node={n}",
           os.path.abspath(__file__)] + list(argv)
    return subprocess.call(cmd, env=env)


def cpu_baseline(spec, n_sample):
    """Reference-structured CPU path (numpy zgemm + per-pixel scipy spline loop + dense map2salm): the oracle, timed on a bounded
    sample (first n_sample time steps of the same workload) on ONE core -- the same worker, pinned to one core with one BLAS thread,
    that the all-cores leg runs P of (one definition for both legs)."""
    import multiprocessing as mp

    cpus = _one_socket_cpus()
    saved = {k: os.environ.get(k) for k in ("OMP_NUM_THREADS", "OPENBLAS_NUM_THREADS", "MKL_NUM_THREADS")}
    for k in saved:
        os.environ[k] = "1"
    try:
        with mp.get_context("spawn").Pool(1) as pool:
            pool.map(_cpu_worker, [(spec["name"], 200, spec.get("axis", "uniform"), None)])  # imports paid before the clock starts
            dt, n_out = pool.map(_cpu_worker, [(spec["name"], n_sample, spec.get("axis", "uniform"), cpus[0])])[0]
    finally:
        for k, v in saved.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
    return {
        "value": n_sample / dt,
        "unit": "timesteps/s",
        "cores": 1,
        "kind": "port",
        "sample": f"first {n_sample} of the 1e5 time steps of the same workload, oracle/waveform_grid_ref.transform "
        f"(numpy tensordot + per-pixel scipy InterpolatedUnivariateSpline loop + dense map2salm) in one process pinned to one core, {dt:.1f} s, "
        f"{n_out} output steps",
    }


def _host_cores():
    """cores this process can actually use: affinity mask, cut by the cgroup CPU quota when there is one, capped at 64"""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    for path, parse in (
        ("/sys/fs/cgroup/cpu.max", lambda s: (s.split()[0], s.split()[1])),
        ("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", lambda s: (s.strip(), open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read().strip())),
    ):
        try:
            quota, period = parse(open(path).read())
            if quota not in ("max", "-1"):
                n = min(n, max(1, int(int(quota) / int(period))))
            break
        except (OSError, ValueError, IndexError):
            continue
    return max(1, min(n, 64))


def _one_socket_cpus():
    """Logical CPUs of this process's affinity mask that are the FIRST hardware thread of a physical core of ONE socket (the socket
    of the first allowed CPU): the "single-socket" denominator of the north star.  Falls back to the usable-core count."""
    try:
        allowed = sorted(os.sched_getaffinity(0))
    except AttributeError:
        allowed = list(range(os.cpu_count() or 1))
    topo = {}
    try:
        cpu = None
        for line in open("/proc/cpuinfo"):
            k, _, v = line.partition(":")
            k, v = k.strip(), v.strip()
            if k == "processor":
                cpu = int(v)
                topo[cpu] = [0, cpu]
            elif k == "physical id" and cpu is not None:
                topo[cpu][0] = int(v)
            elif k == "core id" and cpu is not None:
                topo[cpu][1] = int(v)
    except (OSError, ValueError):
        topo = {}
    usable = [c for c in allowed if c in topo] or allowed
    socket = topo.get(usable[0], [0, 0])[0]
    seen, cpus = set(), []
    for c in usable:
        s_, core = topo.get(c, [socket, c])
        if s_ == socket and core not in seen:
            seen.add(core)
            cpus.append(c)
    return cpus[: _host_cores()]


def _cpu_worker(job):
    """One process of the all-cores CPU baseline: the oracle on the SAME sample the one-core leg transforms (spawned, pinned to one
    core of the socket, never touches the GPU)."""
    name, n_sample, axis, cpu = job
    if cpu is not None:
        try:
            os.sched_setaffinity(0, {cpu})
        except (AttributeError, OSError):
            pass
    from oracle import waveform_grid_ref as grid_ref
    from oracle.containers import WM, h
    from scri_amd import synthetic

    spec = dict(synthetic.CONFIGS[name])
    t, data, _ = synthetic.workload(name, n_times=n_sample, axis=axis)
    w = WM(t=t, data=data, ell_min=2, ell_max=spec["ell_max"], dataType=h)
    t0 = time.perf_counter()
    out = grid_ref.transform(w, **spec["kwargs"])
    return time.perf_counter() - t0, int(out.t.size)


def cpu_baseline_all_cores(spec, n_sample):
    """The same oracle on every physical core of one socket at once, ONE definition with the one-core leg: each of the P processes
    transforms the same `n_sample`-step sample (first n_sample steps of the workload, same time axis), BLAS threads pinned to 1 per
    process; throughput = P x n_sample / wall.  P is also bounded by host memory (the reference-shaped transform of 30 000 steps of
    cfg3 holds about 3 GB of grids and spline temporaries per process)."""
    import multiprocessing as mp

    cpus = _one_socket_cpus()
    per_proc = 3.5e9 * n_sample / 30000.0
    try:
        import psutil

        fit = int(psutil.virtual_memory().available * 0.7 // max(per_proc, 1.0))
    except Exception:  # noqa: BLE001
        fit = len(cpus)
    procs = max(1, min(len(cpus), fit))
    cpus = cpus[:procs]
    axis = spec.get("axis", "uniform")
    saved = {k: os.environ.get(k) for k in ("OMP_NUM_THREADS", "OPENBLAS_NUM_THREADS", "MKL_NUM_THREADS")}
    for k in saved:
        os.environ[k] = "1"
    try:
        with mp.get_context("spawn").Pool(procs) as pool:
            pool.map(_cpu_worker, [(spec["name"], 200, axis, None)] * procs)  # imports paid before the clock starts
            t0 = time.perf_counter()
            each = [e[0] for e in pool.map(_cpu_worker, [(spec["name"], n_sample, axis, c) for c in cpus], chunksize=1)]
            dt = time.perf_counter() - t0
    finally:
        for k, v in saved.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
    return {
        "value": procs * n_sample / dt,
        "unit": "timesteps/s",
        "cores": procs,
        "kind": "port",
        "sample": f"{procs} processes (one per physical core of one socket, pinned), EACH the first {n_sample} of the 1e5 time steps of the same "
        f"workload -- the one-core leg's sample -- oracle/waveform_grid_ref.transform, {dt:.1f} s wall (slowest process {max(each):.1f} s, "
        f"fastest {min(each):.1f} s)",
    }


def rotation_line(local, ell_max, ctx, cpu_steps):
    """Secondary measurement (configs[0] of BASELINE.json at the cfg3 size): in-place time-series rotation of the resident
    modes by one rotor per time step (scri/rotations.py:370-392), HIP-event kernel time against the HBM roofline
    (read + write the modes once: 2 x 16 n_modes + 32 B per step), with the scalar C port of the numba kernel beside it."""
    import torch

    from scri_amd import engine

    n, nm = local.shape
    rng = np.random.default_rng(0)
    R = rng.normal(size=(n, 4))
    R /= np.linalg.norm(R, axis=1)[:, None]
    sp_host = np.stack([R[:, 0] + 1j * R[:, 3], R[:, 2] + 1j * R[:, 1]], axis=1)
    sp = torch.from_numpy(sp_host).to(local.device)
    data = local.clone()
    for _ in range(3):
        engine.rotate_device(data.data_ptr(), n, nm, 2, ell_max, spinors_ptr=sp.data_ptr(), ctx=ctx)
    ctx.synchronize()
    ctx.get_timing(reset=True)
    reps = 10
    t0 = time.perf_counter()
    for _ in range(reps):
        engine.rotate_device(data.data_ptr(), n, nm, 2, ell_max, spinors_ptr=sp.data_ptr(), ctx=ctx)
    ctx.synchronize()
    wall = (time.perf_counter() - t0) / reps
    ms = ctx.get_timing(reset=True)["rotate"][0] / reps
    bytes_per_step = 2 * 16 * nm + 32
    out = {
        "metric": "timesteps/s, time-series rotation of modes in HBM",
        "value": n / wall,
        "kernel_ms": ms,
        "roofline": {"bound": "hbm", "achieved": n * bytes_per_step / (ms * 1e-3) / 1e9, "peak": 8000.0, "unit": "GB/s",
                     "frac": n * bytes_per_step / (ms * 1e-3) / 8e12, "bytes_per_step": bytes_per_step},
    }
    if cpu_steps > 0:
        from oracle import rotate_port

        ns = min(cpu_steps, n)
        d = np.ascontiguousarray(local[:ns].cpu().numpy())
        t0 = time.perf_counter()
        rotate_port.rotate_by_series(d, np.ascontiguousarray(sp_host[:ns]), 2, ell_max)
        dt = time.perf_counter() - t0
        out["cpu_baseline"] = {"value": ns / dt, "unit": "timesteps/s", "cores": 1, "kind": "port",
                               "sample": f"{ns} steps, oracle/rotate_port.c (scalar C port of the numba kernel incl. per-step Wigner-D), {dt:.1f} s"}
        nsa = min(8 * ns, n)
        da = np.ascontiguousarray(local[:nsa].cpu().numpy())
        t0 = time.perf_counter()
        _, used = rotate_port.rotate_by_series_omp(da, np.ascontiguousarray(sp_host[:nsa]), 2, ell_max, n_threads=_host_cores())
        dt = time.perf_counter() - t0
        out["cpu_baseline_all_cores"] = {"value": nsa / dt, "unit": "timesteps/s", "cores": used, "kind": "port",
                                         "sample": f"{nsa} steps, the same C port with the time loop under OpenMP, {dt:.1f} s"}
    return out


def boost_free_line(local, t_global, kw, n_theta, ell_max, ctx):
    """Secondary measurement: the same resident series through a transformation WITHOUT a boost (the workload's
    supertranslation and frame rotation): the modes are rotated once and synthesised ring by ring (synthesis_split_kernel)
    instead of through the dense sYlm product.  HIP-event time of the synthesis kernel against the HBM roofline
    (read the modes once, write the grid once: 16 (n_modes + 1 + n_pix) B per step)."""
    import torch

    from scri_amd import engine

    n, nm = local.shape
    tr = engine.make_transformation(kw["supertranslation"], kw.get("frame_rotation", [1, 0, 0, 0]), [0, 0, 0], n_theta, n_theta, ell_max)
    out = torch.empty_like(local)

    def go():
        return engine.transform_modes(t_global, local.data_ptr(), 2, ell_max, -2, -1, engine.BMS_TERM_H, tr, ctx=ctx, device=True,
                                      ld=nm, out_ptr=out.data_ptr())[1]

    reps = 10

    def measure():
        for _ in range(3):
            go()
        ctx.synchronize()
        ctx.get_timing(reset=True)
        t0 = time.perf_counter()
        for _ in range(reps):
            n = go()
        ctx.synchronize()
        return (time.perf_counter() - t0) / reps, ctx.get_timing(reset=True), n

    wall, tm, n_new = measure()
    # Default route since round 6 (shape rule of the engine: l_max >= 15): bspline_solve_modes_kernel + synthesis_eval_kernel -- the spline
    # evaluated inside the separable synthesis, no pass over a grid of coefficients, no `spline_backward` kernel.  The route it replaced
    # (elimination + synthesis_split_kernel + bspline_backward_eval_kernel) is measured beside it through the context's option, so that the
    # choice can be checked on every run.
    with ctx.options(NO_SYNTHESIS_EVAL=1):
        wall_2, tm_2, _ = measure()
    two_pass = {"ms_per_step": wall_2 * 1e3, "kernels": {k: v[0] / reps for k, v in tm_2.items() if v[1]},
                "route": "context option NO_SYNTHESIS_EVAL=1: elimination + synthesis_split_kernel + bspline_backward_eval_kernel"}
    fused_default = tm.get("spline_backward", (0.0, 0))[1] == 0
    ms = tm["gemm_synthesis"][0] / max(tm["gemm_synthesis"][1], 1)
    bytes_per_step = 16 * (nm + 1 + n_theta * n_theta)
    return {
        "two_pass_route": two_pass,
        "metric": "timesteps/s, transformation without a boost (supertranslation + frame rotation), separable synthesis",
        "value": n / wall,
        "ms_per_step": wall * 1e3,
        "n_out": int(n_new),
        "kernels": {k: v[0] / reps for k, v in tm.items() if v[1]},
        "synthesis_roofline": {"bound": "hbm", "kernel": "synthesis_eval_kernel (modes in, SAMPLES out)" if fused_default else "synthesis_split_kernel", "achieved": n * bytes_per_step / (ms * 1e-3) / 1e9,
                               "peak": 8000.0, "unit": "GB/s", "frac": n * bytes_per_step / (ms * 1e-3) / 8e12,
                               "bytes_per_step": bytes_per_step, "ms_per_launch": ms},
    }


def plumbing_only(args, rank, world, backend_note=None):
    """The N > 1 path minus the kernels, for a box without GPUs (CPU test of the launcher): shard plan from the library's host
    planner, one halo exchange of the synthetic input rows over the process group, every rank's rows checked."""
    import torch
    import torch.distributed as dist

    from scri_amd import engine, sharding, synthetic

    spec = dict(synthetic.CONFIGS[args.workload])
    kw = spec["kwargs"]
    ell_max = spec["ell_max"]
    lst = int(round(np.sqrt(len(kw["supertranslation"])))) - 1
    abd = args.workload == "cfg5"
    n_theta = 2 * (2 * ell_max + 1) + 1 if abd else 2 * (ell_max + lst) + 1
    n_global = int(args.n_times or 8000)
    tr = engine.make_transformation(kw["supertranslation"], kw.get("frame_rotation", [1, 0, 0, 0]), kw.get("boost_velocity", [0, 0, 0]),
                                    n_theta, n_theta, ell_max)
    have, need, window = sharding.plan(synthetic.time_axis(n_global, spec["dt"], args.time_axis), tr, world)
    gen = synthetic.abd_workload if abd else synthetic.workload
    _, mine, _ = gen(args.workload, n_times=n_global, rows=have[rank])
    ext = sharding.exchange_halos(torch.from_numpy(mine), have[rank], need[rank], have, need, dim=1 if abd else 0)  # (six fields: rows = axis 1)
    _, expect, _ = gen(args.workload, n_times=n_global, rows=need[rank])
    ok = torch.tensor([1 if np.array_equal(ext.numpy(), expect) else 0])
    dist.all_reduce(ok, op=dist.ReduceOp.MIN)
    if rank == 0:
        print(json.dumps({"plumbing_only": True, "n_gpus": world, "backend": dist.get_backend(), "backend_note": backend_note,
                          "halo_rows_exact": bool(int(ok)),
                          "have": have, "need": need, "window": list(window), "workload": args.workload, "n_times_total": n_global}))
    dist.barrier()
    dist.destroy_process_group()
    return 0 if int(ok) else 1


def abd_boost_free_line(ctx, n_rows=25_000):
    """Secondary measurement: AsymptoticBondiData (psi0..psi4 + sigma, l <= 24, 99 x 99 working grid: `n_rows` = 25 000 is one GPU's
    share of cfg5, the shape of profiles/r03_c_separable_probe_abd.txt) under cfg5's supertranslation + frame rotation WITHOUT its boost -- elimination on the modes, two-kernel
    separable synthesis (theta stage per field, phi stage of the six fields fused with their mixing) -- and the same through the six
    dense products.  HIP-event time of the synthesis kernels against the HBM roofline with their algorithmic bytes: modes read,
    F written and read, grids written."""
    import torch

    from scri_amd import engine, synthetic

    spec = synthetic.CONFIGS["cfg5"]
    kw, L = spec["kwargs"], spec["ell_max"]
    u, raw, _ = synthetic.abd_workload("cfg5", n_times=n_rows)
    n_theta = 2 * (2 * L + 1) + 1
    tr = engine.make_transformation(kw["supertranslation"], kw["frame_rotation"], [0, 0, 0], n_theta, n_theta, L)
    d_in = torch.from_numpy(raw).to(torch.device("cuda", ctx.device))
    d_out = torch.empty_like(d_in)
    out = {"metric": "timesteps/s, AsymptoticBondiData transformation without a boost (cfg5 fields and grid)", "n_times": n_rows}
    saved = ctx.option("NO_SEPARABLE_SYNTHESIS")
    try:
        for route in ("separable", "dense"):
            ctx.option("NO_SEPARABLE_SYNTHESIS", 1 if route == "dense" else 0)
            reps = 3
            engine.transform_abd(u, d_in.data_ptr(), L, tr, ctx=ctx, device=True, out_ptr=d_out.data_ptr())
            ctx.synchronize()
            ctx.get_timing(reset=True)
            t0 = time.perf_counter()
            for _ in range(reps):
                engine.transform_abd(u, d_in.data_ptr(), L, tr, ctx=ctx, device=True, out_ptr=d_out.data_ptr())
            ctx.synchronize()
            wall = (time.perf_counter() - t0) / reps
            tm = {k: v[0] / reps for k, v in ctx.get_timing(reset=True).items() if v[1]}
            out[route] = {"ms_per_step": wall * 1e3, "value": n_rows / wall, "kernels": tm}
    finally:
        ctx.option("NO_SEPARABLE_SYNTHESIS", saved)
    nm, jp = (L + 1) ** 2, (n_theta + 7) // 8 * 8
    bytes_per_step = 6 * 16 * (nm + 1 + 2 * (2 * L + 1) * jp + n_theta * n_theta)
    ms = out["separable"]["kernels"].get("gemm_synthesis")
    if ms:
        out["synthesis_roofline"] = {"bound": "hbm", "kernel": "theta_synthesis_mfma_kernel x 6 + phi_synthesis_mix6_kernel", "bytes_per_step": bytes_per_step,
                                     "achieved": n_rows * bytes_per_step / (ms * 1e-3) / 1e9, "peak": 8000.0, "unit": "GB/s",
                                     "frac": n_rows * bytes_per_step / (ms * 1e-3) / 8e12, "ms": ms}
    out["speedup_vs_dense_products"] = out["dense"]["ms_per_step"] / out["separable"]["ms_per_step"]
    return out


def inprocess_line(args):
    """`--inprocess N`: ONE process, N devices, numpy in / numpy out -- the callers the reference has (scri/waveform_modes.py:705-719,
    scri/asymptotic_bondi_data/transformations.py:391-412 take host arrays in one process).  engine.transform_modes / transform_abd
    with `devices=[...]` deal the time shards of the pipelined call over one context and one host thread per device; every device
    gets its rows + halo at upload time, so there is no process group and no GPU-to-GPU traffic.  The rate is PCIe-INCLUSIVE (host
    arrays in, host arrays out: not comparable with the device-resident `value` of the default line).  With fewer devices than N
    the contexts share devices (plumbing + parity only; the line says so)."""
    import torch

    from scri_amd import engine, synthetic

    n = args.inprocess
    n_dev = max(torch.cuda.device_count(), 1)
    devices = [i % n_dev for i in range(n)]
    workload = args.workload or ("cfg3" if n == 1 else "cfg4")
    abd = workload == "cfg5"
    spec = dict(synthetic.CONFIGS[workload])
    kw, ell_max = spec["kwargs"], spec["ell_max"]
    lst = int(round(np.sqrt(len(kw["supertranslation"])))) - 1
    n_global = int(args.n_times or (100_000 if workload in ("cfg2", "cfg3") else spec["n_times"]))
    n_theta = 2 * int(args.working_ell_max or 2 * ell_max + 1) + 1 if abd else 2 * (ell_max + lst) + 1
    tr = engine.make_transformation(kw["supertranslation"], kw.get("frame_rotation", [1, 0, 0, 0]), kw.get("boost_velocity", [0, 0, 0]),
                                    n_theta, n_theta, ell_max)
    gen = synthetic.abd_workload if abd else synthetic.workload
    t, data, _ = gen(workload, n_times=n_global, axis=args.time_axis)
    pieces = engine.pieces_for(devices, n_global, ell_max, data.nbytes, abd=abd)

    def go(devs, pcs):
        if abd:
            return engine.transform_abd(t, data, ell_max, tr, devices=devs, pieces=pcs)
        return engine.transform_modes(t, data, 2, ell_max, -2, -1, engine.BMS_TERM_H, tr, devices=devs, pieces=pcs)

    for _ in range(max(args.warmup, 2)):  # (the second sighting page-locks the input in place: _lib.register_if_reused)
        t_out, out = go(devices, pieces)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        t_out, out = go(devices, pieces)
    elapsed = time.perf_counter() - t0
    out = np.array(out)
    # parity inside the run: the same number of time shards on ONE context (bit for bit), and the default one-context call (rounding)
    t_one, out_one = go(None, pieces)
    same_bits = bool(np.array_equal(t_one, t_out) and np.array_equal(out_one, out))
    for _ in range(2):  # (warm: the default call's own page-locked result block and work space)
        t_def, out_def = go(None, None)
    t1 = time.perf_counter()
    reps1 = max(1, min(3, args.steps))
    for _ in range(reps1):
        t_def, out_def = go(None, None)
    one_ctx_ms = 1e3 * (time.perf_counter() - t1) / reps1
    scale = float(np.abs(out_def).max())
    diff = float(np.abs(np.asarray(out_def) - out).max())
    shared = n_dev < n
    line = {
        "metric": "timesteps/sec for full BMS transform, host arrays in and out (PCIe-inclusive), one process over several devices; fp64",
        "value": n_global * args.steps / elapsed, "unit": "timesteps/s", "n_gpus": n, "steps": args.steps, "warmup": max(args.warmup, 2),
        "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True, "scaling": "strong" if n > 1 else None, "vs_baseline": None,
        "dtype": "f64", "data": "synthetic",
        "config": {
            "workload": f"{workload}: {'AsymptoticBondiData psi0..psi4 + sigma' if abd else 'WaveformModes h'}, ell_max {ell_max}, {n_global} time steps in "
                        f"host memory, {n_theta}x{n_theta} grid, {t_out.size} output steps",
            "sharding": f"in-process: {pieces} time shards of the pipelined call dealt over {n} contexts on devices {devices}, one host thread each, rows + "
                        "halo shipped at upload time (no process group, no GPU-to-GPU traffic); engine.transform_" + ("abd" if abd else "modes") + "(devices=...)",
            "devices_shared": shared,
            "note": ("the contexts share devices: plumbing and parity only, NO timing claim is possible on this box" if shared else
                     "one context per device"),
        },
        "parity": {"bit_identical_to_one_context_with_the_same_shards": same_bits, "max_abs_diff_vs_default_one_context_call": diff,
                   "scale_max_abs": scale, "bar": 1e-14 * scale, "within_bar": bool(same_bits and diff <= 1e-14 * scale)},
        "one_context_same_run": {"ms_per_step": one_ctx_ms, "pieces": engine.PIPELINE_PIECES if abd else engine.auto_pieces(n_global, ell_max, data.nbytes)},
    }
    print("\n" + json.dumps(line), flush=True)
    return 0 if line["parity"]["within_bar"] else 1


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--workload", default=None, choices=["cfg2", "cfg3", "cfg4", "cfg5"],
                    help="default: cfg3 on one GPU, cfg4 (1e6 steps in all, strong scaling) on several")
    ap.add_argument("--n-times", type=int, default=None,
                    help="time steps PER GPU (default: 1e5; cfg5 on one GPU: 2e5 / 8); cfg4, and cfg5 on several GPUs: time steps IN ALL "
                    "(defaults 1e6 / 2e5)")
    ap.add_argument("--overlap-halo", action="store_true",
                    help="N > 1, time shards: transform the outputs that need own rows only while the halos travel, then the "
                    "two edges (three engine calls per step instead of one)")
    ap.add_argument("--no-n1-reference", action="store_true", help="cfg4: skip the single-GPU pass of the whole series on rank 0")
    ap.add_argument("--working-ell-max", type=int, default=None, help="cfg5 only (default 2 ell_max + 1 = 49 -> 99 x 99 grid)")
    ap.add_argument("--cpu-sample", type=int, default=30000, help="time steps of the CPU-baseline sample (0: skip)")
    ap.add_argument("--partition", default="auto", choices=["auto", "rows", "columns"],
                    help="N > 1: time shards + halo exchange (rows) or grid-column parts + reduce-scatter (columns, for strong "
                    "boosts); auto = sharding.choose_partition (rows for the BASELINE.json workloads)")
    ap.add_argument("--boost-scale", type=float, default=1.0, help="multiplies the workload's boost velocity (stress variants)")
    ap.add_argument("--time-axis", default="uniform", choices=["uniform", "jitter", "sxs"],
                    help="time samples of the synthetic series (scri_amd/synthetic.py::time_axis): BASELINE.json's uniform dt, every sample "
                    "jittered by +-30 %% of dt, or steps shrinking 20x over the series as an inspiral -> merger run's do (same span, same signals)")
    ap.add_argument("--no-live-pmc", action="store_true",
                    help="N = 1: do not run the two rocprofv3 --pmc child passes that measure roofline.traffic (the committed summary "
                    "is reported instead if it was taken on this build's kernel sources, otherwise null)")
    ap.add_argument("--plumbing-only", action="store_true",
                    help="N > 1 without any GPU: launcher, rendezvous, backend agreement, shard plan and one halo exchange on host "
                    "tensors, checked against the synthetic series; prints a line with \"plumbing_only\": true and no measurement")
    ap.add_argument("--no-parity", action="store_true", help="cfg4, N > 1: skip the comparison of the reassembled shard outputs with "
                    "rank 0's single-GPU result of the same run")
    ap.add_argument("--inprocess", type=int, default=0, metavar="N",
                    help="ONE process, N devices, host arrays in and out: engine.transform_modes(devices=[0..N-1]) (no process group; with fewer "
                    "devices than N the contexts share them: parity only).  Prints its own line; PCIe-inclusive, not the headline `value`")
    args = ap.parse_args()
    if args.inprocess:
        return inprocess_line(args)
    if args.workload is None:
        args.workload = "cfg3" if args.gpus == 1 else "cfg4"
    pmc_child = os.environ.get("SCRI_AMD_BENCH_PMC_CHILD") == "1"
    global DOMINANT_KERNEL
    DOMINANT_KERNEL = dominant_kernel(args.workload == "cfg5")

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # no launcher around us: be the launcher (child processes; nothing in this process has touched the GPU)
        raise SystemExit(launch_ranks(args.gpus, sys.argv[1:]))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} does not match WORLD_SIZE {world}")

    # roofline.traffic of the dominant kernel, measured by this run: PMC child passes BEFORE this process initialises the GPU
    traffic, traffic_note = None, "not measured"
    under_profiler = any("rocprof" in os.environ.get(k, "").lower() for k in ("LD_PRELOAD", "ROCP_TOOL_LIBRARIES", "ROCPROFILER_LIBRARY_CTOR"))
    if under_profiler:  # (a profiler's preloaded library has initialised the GPU already: no child programs from here)
        traffic_note = "not measured: this run is itself under a profiler"
    if world == 1 and not pmc_child and not args.no_live_pmc and args.workload == "cfg3" and not under_profiler:
        child = ["--gpus", "1", "--steps", "3", "--warmup", "1", "--cpu-sample", "0", "--no-live-pmc", "--workload", args.workload,
                 "--boost-scale", str(args.boost_scale), "--time-axis", args.time_axis] + (["--n-times", str(args.n_times)] if args.n_times else [])
        traffic, traffic_note = live_pmc_traffic(child)

    import datetime

    import torch
    import torch.distributed as dist

    from scri_amd import _lib, engine, synthetic, sharding

    # SCRI_AMD_BENCH_BACKEND=gloo: dry run of the multi-rank path on a box with fewer GPUs than ranks (ranks share
    # devices, halos travel through the host); the measured configuration is nccl = RCCL, one rank per GPU
    backend = os.environ.get("SCRI_AMD_BENCH_BACKEND", "nccl")
    if args.plumbing_only and os.environ.get("SCRI_AMD_BENCH_BACKEND") != "nccl":
        backend = "gloo"  # (with nccl asked for explicitly the bring-up below is attempted, fails without GPUs and falls back: its test)
    n_dev = max(torch.cuda.device_count(), 1)
    dev_index = local_rank % n_dev
    if not args.plumbing_only:
        torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    backend_note = None
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            # RCCL, one rank per GPU.  The ranks AGREE on the outcome of the bring-up through a key-value store of their own
            # (a failure seen by a subset of the ranks must not leave the others waiting in a collective): every rank
            # publishes ok / failed after its init + first all_reduce, reads everybody's flag, and only a unanimous ok keeps
            # RCCL; otherwise all of them fall back to gloo with the halos staged through the host rather than produce no
            # line at all, and the line says so (`config.ranks`).
            port = int(os.environ.get("MASTER_PORT", "29500"))
            if os.environ.get("TORCHELASTIC_USE_AGENT_STORE") == "True":  # under torchrun: the agent's store at MASTER_PORT, as a client
                store = dist.PrefixStore("bench_agree", dist.TCPStore(os.environ["MASTER_ADDR"], port, world, is_master=False,
                                                                      timeout=datetime.timedelta(seconds=300)))
            else:
                store = dist.TCPStore(os.environ["MASTER_ADDR"], port + 23, world, is_master=(rank == 0), timeout=datetime.timedelta(seconds=300),
                                      wait_for_workers=False)
            err = None
            try:
                if n_dev < world:
                    raise RuntimeError(f"{world} ranks on {n_dev} visible GPU(s)")
                dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev, timeout=datetime.timedelta(seconds=240))
                probe = torch.ones(1, device=dev)
                dist.all_reduce(probe)
                torch.cuda.synchronize()
                if int(probe.item()) != world:
                    raise RuntimeError(f"all_reduce probe returned {probe.item()} on {world} ranks")
            except Exception as e:  # noqa: BLE001 -- any failure of the RCCL bring-up
                err = f"{type(e).__name__}: {str(e)[:200]}"
            store.set(f"nccl_{rank}", "ok" if err is None else err)
            flags = [store.get(f"nccl_{r}").decode(errors="replace") for r in range(world)]
            if any(f != "ok" for f in flags):
                bad = next(r for r, f in enumerate(flags) if f != "ok")
                backend_note = f"nccl bring-up failed on rank {bad} ({flags[bad]}); all ranks on gloo, halos through the host"
                if dist.is_initialized():
                    try:
                        dist.destroy_process_group()
                    except Exception:  # noqa: BLE001
                        pass
                backend = "gloo"
                # (rendezvous through the ranks' own store: under torchrun the env:// store lives in the launcher's agent at
                # MASTER_PORT and must stay where it is)
                dist.init_process_group("gloo", store=dist.PrefixStore("gloo_fallback", store), rank=rank, world_size=world)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    # One stream for torch's work (allocations, copies, RCCL's completion waits) AND the engine's kernels: a side stream
    # of torch's made current for the whole run, handed to the context.  (torch's default stream has the handle 0; engine
    # kernels on the context's own non-blocking stream would not be ordered behind copies queued there.)
    if args.plumbing_only:
        return plumbing_only(args, rank, world, backend_note)
    run_stream = torch.cuda.Stream(device=dev)
    torch.cuda.set_stream(run_stream)

    spec = dict(synthetic.CONFIGS[args.workload])
    spec["name"] = args.workload
    spec["axis"] = args.time_axis  # (both CPU baselines run on the same axis)
    abd = args.workload == "cfg5"
    # total work fixed, sharded `world` ways: cfg4 (1e6 steps), and cfg5 on several GPUs (BASELINE.json configs[4]: 2e5 steps, 8 GPUs;
    # scri/asymptotic_bondi_data/transformations.py:391-412 is what the shards reproduce) -- on one GPU cfg5 stays one rank's share
    strong = args.workload == "cfg4" or (abd and world > 1)
    if strong:
        n_global = int(args.n_times or spec["n_times"])
        per_gpu = -(-n_global // world)
    else:
        per_gpu = int(args.n_times or (spec["n_times"] // 8 if abd else 100_000))
        n_global = per_gpu * world
    kw = dict(spec["kwargs"])
    if args.boost_scale != 1.0:
        kw["boost_velocity"] = np.asarray(kw["boost_velocity"], dtype=float) * args.boost_scale
        spec["kwargs"] = kw
    ell_max = spec["ell_max"]
    lst = int(round(np.sqrt(len(kw["supertranslation"])))) - 1
    if abd:
        n_theta = 2 * int(args.working_ell_max or 2 * ell_max + 1) + 1
        n_modes = (ell_max + 1) ** 2
    else:
        n_theta = 2 * (ell_max + lst) + 1
        n_modes = (ell_max + 1) ** 2 - 4
    tr = engine.make_transformation(
        kw["supertranslation"], kw.get("frame_rotation", [1, 0, 0, 0]), kw.get("boost_velocity", [0, 0, 0]), n_theta, n_theta, ell_max
    )
    n_pix = n_theta * n_theta
    n_fields = 6 if abd else 1

    # this rank's rows of the global series, resident in HBM before the timed region.  Plan, halo exchange, shard call and row
    # placement are the library's (scri_amd/sharding.py::ShardedTransform -- what WaveformModes.transform(group=...) runs too)
    ctx = _lib.Context(dev_index, stream=run_stream.cuda_stream)
    ctx.enable_timing(True)
    st = sharding.ShardedTransform("abd" if abd else "modes", synthetic.time_axis(n_global, spec["dt"], args.time_axis), tr, 2, ell_max, -2, -1,
                                   engine.BMS_TERM_H, partition=args.partition, overlap=args.overlap_halo, ctx=ctx)
    have, need, window = st.have, st.need, st.window
    own = st.own
    columns = st.partition == "columns" and world > 1
    if abd:
        t_global, local_host, _ = synthetic.abd_workload(args.workload, n_times=n_global, rows=have[rank], axis=args.time_axis)
        out = torch.empty((6, own, n_modes), dtype=torch.complex128, device=dev)
    else:
        t_global, local_host, _ = synthetic.workload(args.workload, n_times=n_global, rows=have[rank], axis=args.time_axis)
        out = torch.empty((own, n_modes), dtype=torch.complex128, device=dev)
    local = torch.from_numpy(local_host).to(dev)
    del local_host
    # N > 1, WaveformModes, time shards: the series lives inside the exchange buffer (own rows placed once, every step only moves the halos)
    view = st.own_rows_view(like=local)
    if view is not None:
        view.copy_(local)
        local = view

    # ---- cfg4 / cfg5 on several GPUs: the whole series on ONE GPU (rank 0's), in the same run: the reference of the strong-scaling line
    n1 = None
    whole_out = None
    if strong and world > 1 and not args.no_n1_reference:
        if rank == 0:
            if abd:
                _, whole_host, _ = synthetic.abd_workload(args.workload, n_times=n_global, axis=args.time_axis)
            else:
                _, whole_host, _ = synthetic.workload(args.workload, n_times=n_global, axis=args.time_axis)
            whole = torch.from_numpy(whole_host).to(dev)
            del whole_host
            whole_out = torch.empty(((6, n_global, n_modes) if abd else (n_global, n_modes)), dtype=torch.complex128, device=dev)
            reps = 3
            for i in range(reps + 1):
                if i == 1:
                    ctx.synchronize()
                    t1 = time.perf_counter()
                if abd:
                    n1_rows = engine.transform_abd(t_global, whole.data_ptr(), ell_max, tr, ctx=ctx, device=True, out_ptr=whole_out.data_ptr())[1]
                else:
                    n1_rows = engine.transform_modes(t_global, whole.data_ptr(), 2, ell_max, -2, -1, engine.BMS_TERM_H, tr, ctx=ctx, device=True,
                                                     ld=n_modes, out_ptr=whole_out.data_ptr())[1]
            ctx.synchronize()
            n1 = {"ms_per_step": 1e3 * (time.perf_counter() - t1) / reps, "steps": reps, "where": "rank 0's GPU, before the sharded loop"}
            del whole
            if args.no_parity:
                whole_out = None
            torch.cuda.empty_cache()
            ctx.get_timing(reset=True)
        dist.barrier()

    # N > 1 on RCCL: the engine runs on torch's current stream (`run_stream`), so the halo rows (received on RCCL's stream,
    # which the current stream waits for in req.wait(); copied into place on the current stream) are ordered before the
    # kernels that read them without a host-side synchronisation of the device
    stream_ordered = world > 1 and backend == "nccl"

    interior = st.interior  # --overlap-halo: outputs [a, b) of this rank need its own rows only and are transformed while the halos travel

    last = {}

    def step():
        _, rows, _ = st(local, out=out)
        last["rows"] = rows  # (time shards: a view of `out`; grid columns: this rank's block of the reduce-scatter)
        return rows.shape[1 if abd else 0]

    def fence():
        torch.cuda.synchronize()
        ctx.synchronize()
        if world > 1:
            dist.barrier()

    for _ in range(args.warmup):
        step()
    fence()
    ctx.get_timing(reset=True)
    t0 = time.perf_counter()
    n_out = 0
    for _ in range(args.steps):
        n_out = step()
    fence()
    elapsed = time.perf_counter() - t0
    timing = ctx.get_timing(reset=True)
    eval_stats = ctx.eval_stats(reset=True)  # (the warm-up steps count too: the same launches)

    el = torch.tensor([elapsed], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
    if world > 1:
        dist.all_reduce(el, op=dist.ReduceOp.MAX)
    elapsed = float(el.item())

    # ---- cfg4 / cfg5, N > 1: the shards' outputs of the LAST timed step, reassembled on rank 0, against rank 0's own single-GPU
    # transform of the whole series from the same run (the check of the RCCL path that only hardware can give; the same
    # comparison on one device is tests/test_gpu_full_size.py::test_cfg4_eight_shards_equal_whole, bar 1e-14 x scale)
    parity = None
    if strong and world > 1 and not args.no_parity and not args.no_n1_reference:
        comm_dev = dev if backend == "nccl" else torch.device("cpu")
        counts = torch.zeros(world, dtype=torch.int64, device=comm_dev)
        counts[rank] = int(n_out)
        dist.all_reduce(counts)
        counts = [int(c) for c in counts.cpu()]
        worst, scale, offset = 0.0, 0.0, 0
        for r in range(world):
            if counts[r] == 0:
                continue
            if r == rank:
                piece = (last["rows"][:, : counts[r]] if abd else last["rows"][: counts[r]]).contiguous().to(comm_dev)
            else:
                piece = torch.empty(((6, counts[r], n_modes) if abd else (counts[r], n_modes)), dtype=torch.complex128, device=comm_dev)
            piece_real = torch.view_as_real(piece)
            dist.broadcast(piece_real, src=r)
            if rank == 0:
                ref = whole_out[:, offset : offset + counts[r]] if abd else whole_out[offset : offset + counts[r]]
                got = piece.to(dev)
                worst = max(worst, float((got - ref).abs().max()))
                scale = max(scale, float(ref.abs().max()))
                del got
            offset += counts[r]
            del piece, piece_real
        if rank == 0:
            parity = {"sharded_vs_n1_max_abs_diff": worst, "scale_max_abs": scale, "rows_compared": offset,
                      "rows_of_n1_result": int(n1_rows),
                      "bar": (2e-14 if columns else 1e-14) * scale,
                      "within_bar": bool(worst <= (2e-14 if columns else 1e-14) * scale and offset == int(n1_rows))}

    if rank == 0:
        ms_per_step = 1e3 * elapsed / args.steps
        value = n_global * args.steps / elapsed
        # roofline of the dominant kernel (zgemm3m_mfma_kernel, synthesis launches): algorithmic flops
        # 8 * n_modes * n_pix per time row x rows per launch / HIP-event duration per launch
        rows_in = n_global if columns or world == 1 else (need[0][1] - need[0][0])
        g_ms, g_calls = timing["gemm_synthesis"]
        # (cfg5: the work space is walked in chunks, 6 launches each; per-launch figures are averages over them and
        # ignore the few halo rows that neighbouring chunks both synthesise)
        launches_per_step = max(g_calls, 1) / args.steps
        flops_per_launch = 8.0 * n_modes * n_pix * rows_in * n_fields / launches_per_step / (world if columns else 1)
        achieved = flops_per_launch / (g_ms / max(g_calls, 1) * 1e-3) / 1e12 if g_ms > 0 else None
        kernels = {k: {"ms_per_step": v[0] / args.steps, "launches_per_step": v[1] / args.steps} for k, v in timing.items() if v[1]}
        executed = 0.75 * achieved * ((n_pix - 2 * (n_theta - 1)) / n_pix if n_theta <= 40 else 1.0) if achieved else None
        if traffic is None and not pmc_child:
            traffic, fallback_note = committed_pmc_traffic(args.workload, world, per_gpu)
            traffic_note = fallback_note if traffic_note == "not measured" else f"{traffic_note}; {fallback_note}"
        # what the workload's transformation is made of, and with it the route and the kernel that dominates: with a boost the dense
        # product (MFMA-bound); without one (cfg2: a rotor series first, then supertranslation only) the separable synthesis, HBM-bound
        v_kw = np.asarray(kw.get("boost_velocity", np.zeros(3)), dtype=float)
        has_boost = bool(np.any(v_kw != 0.0))
        workload_terms = " + ".join(
            [f"supertranslation(l<={int(round(np.sqrt(len(kw['supertranslation'])))) - 1})"] * ("supertranslation" in kw)
            + ["frame_rotation"] * ("frame_rotation" in kw)
            + [f"boost |v|={float(np.linalg.norm(v_kw)):.3g}"] * has_boost) or "identity"
        line = {
            "metric": {
                "cfg3": "timesteps/sec for full BMS transform, l_max=16, 1e5 steps; fp64",
                "cfg4": "timesteps/sec for full BMS transform, l_max=16, 1e6 steps sharded over the GPUs; fp64",
                "cfg2": "timesteps/sec for BMS transform (cfg2)",
                "cfg5": "timesteps/sec for AsymptoticBondiData BMS transform (cfg5: psi0..psi4 + sigma, l_max=24)",
            }[args.workload],
            "value": value,
            "unit": "timesteps/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": ms_per_step,
            "higher_is_better": True,
            "scaling": None if world == 1 else ("strong" if strong else "weak"),
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {
                "workload": (
                    f"{args.workload}: AsymptoticBondiData psi0..psi4 + sigma, ell 0..{ell_max} (6 x {n_modes} modes), " if abd
                    else f"{args.workload}: WaveformModes h, ell 2..{ell_max} ({n_modes} modes), "
                )
                + f"{per_gpu} time steps per GPU ({n_global} total"
                + ("" if args.time_axis == "uniform" else f"; NON-UNIFORM time axis '{args.time_axis}': synthetic.time_axis") + "), " + workload_terms
                + (" (cfg2's second step; its first, the rotor series, is the `rotation` line of the default run at this l range: tools/bench_rotation.py 8)"
                   if args.workload == "cfg2" else "") + ", "
                f"{n_theta}x{n_theta} grid, {n_out} output steps on rank 0",
                "ranks": dict({"world_size": dist.get_world_size() if world > 1 else 1, "backend": dist.get_backend() if world > 1 else None},
                              **({"note": backend_note} if backend_note else {})),
                "sharding": st.describe(),
            },
            "roofline": {
                "bound": "mfma",
                "kernel": (f"{DOMINANT_KERNEL.split('::')[1]} (synthesis: modes -> grid" +
                           ("; the spline is solved on the modes and evaluated in this kernel's epilogue, spline_straddle_eval_kernel for the "
                            "3 of 64 windows that straddle two row tiles is inside the timed launch)" if "eval" in DOMINANT_KERNEL else ")")),
                # `achieved` / `frac`: flops the matrix pipe EXECUTES (3 real products per complex one, one column per pole ring) over
                # the kernel's time; `achieved_nominal` / `frac_nominal`: SURVEY 8(d)'s algorithmic 8 n_modes n_pix per row, which
                # counts work the kernel does not do and can exceed what the chip sustains
                "achieved": executed,
                "peak": FP64_MATRIX_PEAK_TFLOPS,
                "unit": "TFLOP/s",
                "frac": (executed / FP64_MATRIX_PEAK_TFLOPS) if executed else None,
                "achieved_nominal": achieved,
                "frac_nominal": (achieved / FP64_MATRIX_PEAK_TFLOPS) if achieved else None,
                "traffic": traffic,
                "traffic_source": traffic_note,
                "traffic_unit": "HBM-side bytes per launch of the kernel (L2 misses: Infinity-Cache hits included); the A operand passes "
                "each of the column panels (DESIGN.md section 4, 'Dense route')" + ("; `ms_per_launch` covers zgemm3m_eval_kernel AND spline_straddle_eval_kernel "
                "(one timing tag), `traffic` is zgemm3m_eval_kernel's alone (the straddle kernel moves 0.31 GB more)" if "eval" in DOMINANT_KERNEL else ""),
                "csrc_hash": csrc_hash(),
                "flops_per_launch": flops_per_launch,
                "ms_per_launch": g_ms / max(g_calls, 1),
                # what the MFMA pipe actually executes: 3 real products per complex one (not 4), and, when the fused analysis
                # is in use (grids up to 40 x 40), one column per pole ring instead of n_phi
                "executed_tflops": executed,
                "frac_executed": (executed / FP64_MATRIX_PEAK_TFLOPS) if executed else None,
            },
            "kernels": kernels,
        }
        if eval_stats[0] > 0:
            # the evaluating product stages a window of output times per tile in LDS; tiles whose samples leave it search global memory
            line["eval_window"] = {"time_axis": args.time_axis, "tiles": eval_stats[0], "tiles_off_the_lds_path": eval_stats[1],
                                   "marches_continued_from_global_memory": eval_stats[2],
                                   "share_off_the_lds_path": eval_stats[1] / eval_stats[0]}
        if not has_boost and not abd and g_ms > 0 and timing.get("spline_backward", (0.0, 0))[1] > 0:
            # no dense product on this route (a boost-free shape too large for the engine's small-shape rule, which keeps l <= 8 on the evaluating
            # product: then there is no back substitution on the grid and the MFMA roofline above is the right one): the dominant kernel is the
            # separable synthesis (reads the eliminated modes, writes the grid)
            bytes_per_row = 16 * (n_modes + 1 + n_pix)
            ms_launch = g_ms / max(g_calls, 1)
            gbs = bytes_per_row * rows_in / (ms_launch * 1e-3) / 1e9
            line["roofline"] = {"bound": "hbm", "kernel": "synthesis_split_kernel (separable synthesis: no boost in this workload)", "achieved": gbs, "peak": 8000.0,
                                "unit": "GB/s", "frac": gbs / 8000.0, "traffic": None, "traffic_source": "no PMC pass for this workload",
                                "bytes_per_step": bytes_per_row, "ms_per_launch": ms_launch, "csrc_hash": csrc_hash()}
        if world > 1 and not columns:
            all_halo = [(have[r][0] - need[r][0], need[r][1] - have[r][1]) for r in range(world)]
            row_bytes = 16 * n_modes * n_fields
            line["halo"] = {
                "rows_per_rank_before_after": all_halo,
                "bytes_received_per_rank": [row_bytes * (a + b) for a, b in all_halo],
                "exchange": "RCCL point-to-point (batch_isend_irecv) of input-mode rows; torch and the engine share one side stream, "
                "so the rows are ordered before the kernels without a device synchronisation" if stream_ordered
                else "gloo dry run through host memory",
                "overlap": (f"interior outputs [{interior[0]}, {interior[1]}) of rank 0 transformed under the exchange, edges after it"
                            if interior is not None else "none (one engine call per step after the exchange)"),
            }
        if strong:
            line["strong_scaling"] = {
                "n_times_total": n_global,
                "n1_same_run": n1,
                "speedup_vs_n1_same_run": (n1["ms_per_step"] / ms_per_step) if n1 else None,
                "parity": parity,
                "sharded_vs_n1_max_abs_diff": parity["sharded_vs_n1_max_abs_diff"] if parity else None,
            }
        if not abd:
            # the HBM-bound stages against the 8 TB/s peak, with the algorithmic bytes of SURVEY 8(d) (grid = the columns
            # actually stored, n_cols) over the HIP-event kernel times; and the whole transform against the same table
            n_cols_b = n_pix - 2 * (n_theta - 1) if n_theta <= 40 else n_pix
            rows_b = rows_in / (world if columns else 1)
            n_modes_out = n_modes

            def stage(tag, bytes_per_row):
                ms, calls = timing.get(tag, (0.0, 0))
                if not calls or ms <= 0:
                    return None
                gbs = bytes_per_row * rows_b / (ms / calls * 1e-3) / 1e9  # per launch
                return {"algorithmic_bytes_per_step": bytes_per_row, "achieved_GBps": gbs, "frac_of_8TBps": gbs / 8000.0}

            eval_route = "eval" in DOMINANT_KERNEL
            two_sweeps = bool(ctx.option("TWO_SWEEPS"))
            if eval_route:
                # both sweeps of the spline solve on the n_modes + 1 mode columns in ONE pass over memory (read once, written once:
                # bspline_solve_modes_kernel; two passes with SCRI_AMD_TWO_SWEEPS), the product reads the solved modes and writes the
                # SAMPLES (its epilogue evaluates the spline), the analysis reads them: the grid crosses HBM once each way
                total_bytes = (2 if two_sweeps else 1) * 32 * (n_modes + 1) + 16 * (n_modes + 1 + n_cols_b) + 16 * (n_cols_b + n_modes_out)
            else:
                total_bytes = 16 * (n_modes + n_cols_b) + 32 * n_cols_b + 16 * (n_cols_b + n_modes_out)
            line["hbm_stages"] = {
                "analysis": stage("analysis_fused", 16 * (n_cols_b + n_modes_out)),
                ("spline_solve_on_modes" if eval_route and not two_sweeps else "spline_elimination_on_modes"): stage("spline_forward", 32 * (n_modes + 1)),
                **({"spline_back_substitution": stage("spline_backward", 32 * n_cols_b)} if not eval_route else
                   ({"spline_back_substitution_on_modes": stage("spline_backward", 32 * (n_modes + 1))} if two_sweeps else {})),
                "whole_transform_materialised_grid": {
                    "algorithmic_bytes_per_step": total_bytes,
                    "achieved_GBps": total_bytes * (n_global / world) / (ms_per_step * 1e-3) / 1e9,
                    "frac_of_8TBps": total_bytes * (n_global / world) / (ms_per_step * 1e-3) / 8e12,
                },
                "ideal_fusion_bytes_per_step": 2 * (16 * n_modes + 8),
            }
        if world == 1 and args.cpu_sample > 0 and not abd:
            line["cpu_baseline"] = cpu_baseline(spec, args.cpu_sample)
            line["cpu_baseline_all_cores"] = cpu_baseline_all_cores(spec, args.cpu_sample)
        if world == 1 and not abd and not pmc_child:
            line["rotation"] = rotation_line(local, ell_max, ctx, 3000 if args.cpu_sample > 0 else 0)
            line["boost_free"] = boost_free_line(local, t_global, kw, n_theta, ell_max, ctx)
            if args.workload == "cfg3" and args.cpu_sample > 0:  # (the default run; the profiling commands pass --cpu-sample 0)
                del local
                torch.cuda.empty_cache()
                line["abd_boost_free"] = abd_boost_free_line(ctx)
        print("\n" + json.dumps(line), flush=True)  # (on a line of its own whatever a backend wrote to stdout before)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    sys.exit(main())
