#!/usr/bin/env python
"""Benchmark of the BMS-transformation hot path on MI355X (contract: see the task description).

  python bench.py --gpus N --steps K --warmup W        (N > 1: launched under torch.distributed.run)

A "step" is one full BMS transformation (supertranslation + frame rotation + boost) of one synthetic
WaveformModes series that is already resident in HBM: workload cfg3 of BASELINE.json / SURVEY section 8(d)
(h, ell = 2..16, 285 modes, 1e5 time steps, 37 x 37 grid).  With N > 1 the time axis is sharded (weak scaling:
every rank owns 1e5 steps of an N x 1e5 series), the input-mode halos are exchanged over RCCL inside the timed
region, and `value` is the whole-job rate.  Rank 0 prints ONE JSON line.
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


def cpu_baseline(spec, n_sample):
    """Reference-structured CPU path (numpy zgemm + per-pixel scipy spline loop + dense map2salm): the oracle,
    timed on a bounded sample (first n_sample time steps of the same workload), 1 thread of Python/FITPACK."""
    from oracle import waveform_grid_ref as grid_ref
    from oracle.containers import WM, h
    from scri_amd import synthetic

    t, data, _ = synthetic.workload(spec["name"], n_times=n_sample)
    w = WM(t=t, data=data, ell_min=2, ell_max=spec["ell_max"], dataType=h)
    t0 = time.perf_counter()
    out = grid_ref.transform(w, **spec["kwargs"])
    dt = time.perf_counter() - t0
    return {
        "value": n_sample / dt,
        "unit": "timesteps/s",
        "cores": 1,
        "kind": "port",
        "sample": f"first {n_sample} of the 1e5 time steps of the same workload, oracle/waveform_grid_ref.transform "
        f"(numpy tensordot + per-pixel scipy InterpolatedUnivariateSpline loop + dense map2salm), {dt:.1f} s, "
        f"{out.t.size} output steps",
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="cfg3", choices=["cfg2", "cfg3"])
    ap.add_argument("--n-times", type=int, default=None, help="time steps PER GPU (default: the workload's 1e5)")
    ap.add_argument("--cpu-sample", type=int, default=4000, help="time steps of the CPU-baseline sample (0: skip)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit(
                "launch multi-GPU runs with: python -m torch.distributed.run --nnodes=1 --nproc-per-node N "
                "--master-addr 127.0.0.1 --master-port P bench.py --gpus N ..."
            )
        raise SystemExit(f"--gpus {args.gpus} does not match WORLD_SIZE {world}")

    import torch
    import torch.distributed as dist

    from scri_amd import _lib, engine, synthetic, sharding

    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    spec = dict(synthetic.CONFIGS[args.workload])
    spec["name"] = args.workload
    per_gpu = int(args.n_times or spec["n_times"])
    n_global = per_gpu * world
    kw = spec["kwargs"]
    ell_max = spec["ell_max"]
    lst = int(round(np.sqrt(len(kw["supertranslation"])))) - 1
    n_theta = 2 * (ell_max + lst) + 1
    tr = engine.make_transformation(
        kw["supertranslation"], kw.get("frame_rotation", [1, 0, 0, 0]), kw.get("boost_velocity", [0, 0, 0]), n_theta, n_theta, ell_max
    )
    n_modes = (ell_max + 1) ** 2 - 4
    n_pix = n_theta * n_theta

    # this rank's rows of the global series, resident in HBM before the timed region
    have, need, window = sharding.plan(np.arange(n_global) * spec["dt"], tr, world)
    t_global, local_host, _ = synthetic.workload(args.workload, n_times=n_global, rows=have[rank])
    local = torch.from_numpy(local_host).to(dev)
    out = torch.empty((have[rank][1] - have[rank][0], n_modes), dtype=torch.complex128, device=dev)
    ctx = _lib.Context(local_rank)
    ctx.enable_timing(True)

    def step():
        if world > 1:
            ext = sharding.exchange_halos(local, have[rank], need[rank], have, need)
            torch.cuda.synchronize()
        else:
            ext = local
        row0 = need[rank][0] if world > 1 else 0
        t_out, n_new, first = engine.transform_modes(
            t_global, ext.data_ptr(), 2, ell_max, -2, -1, engine.BMS_TERM_H, tr, ctx=ctx, device=True, ld=n_modes,
            out_ptr=out.data_ptr(), shard=(row0, ext.shape[0], have[rank][0], have[rank][1]),
        )
        return n_new

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

    el = torch.tensor([elapsed], dtype=torch.float64, device=dev)
    if world > 1:
        dist.all_reduce(el, op=dist.ReduceOp.MAX)
    elapsed = float(el.item())

    if rank == 0:
        ms_per_step = 1e3 * elapsed / args.steps
        value = n_global * args.steps / elapsed
        # roofline of the dominant kernel (zgemm3m_mfma_kernel, synthesis launches): algorithmic flops
        # 8 * n_modes * n_pix per time row x rows per launch / HIP-event duration per launch
        rows_in = (need[0][1] - need[0][0]) if world > 1 else n_global
        g_ms, g_calls = timing["gemm_synthesis"]
        flops_per_launch = 8.0 * n_modes * n_pix * rows_in
        achieved = flops_per_launch / (g_ms / max(g_calls, 1) * 1e-3) / 1e12 if g_ms > 0 else None
        kernels = {k: {"ms_per_step": v[0] / args.steps, "launches_per_step": v[1] / args.steps} for k, v in timing.items() if v[1]}
        line = {
            "metric": "timesteps/sec for full BMS transform, l_max=16, 1e5 steps; fp64" if args.workload == "cfg3" else "timesteps/sec for BMS transform (cfg2)",
            "value": value,
            "unit": "timesteps/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": ms_per_step,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {
                "workload": f"{args.workload}: WaveformModes h, ell 2..{ell_max} ({n_modes} modes), {per_gpu} time steps per GPU "
                f"({n_global} total), supertranslation(l<=2) + frame_rotation + boost |v|=3.7e-4, {n_theta}x{n_theta} grid, "
                f"{n_out} output steps on rank 0",
                "sharding": f"time axis x{world}, RCCL point-to-point halo exchange of input modes" if world > 1 else "none",
            },
            "roofline": {
                "bound": "mfma",
                "kernel": "zgemm3m_mfma_kernel (synthesis: modes -> grid)",
                "achieved": achieved,
                "peak": FP64_MATRIX_PEAK_TFLOPS,
                "unit": "TFLOP/s",
                "frac": (achieved / FP64_MATRIX_PEAK_TFLOPS) if achieved else None,
                "traffic": None,
                "flops_per_launch": flops_per_launch,
                "ms_per_launch": g_ms / max(g_calls, 1),
                # the kernel forms each complex product from 3 real MFMA products (not 4): flops it actually executes
                "executed_tflops": 0.75 * achieved if achieved else None,
            },
            "kernels": kernels,
        }
        if world == 1 and args.cpu_sample > 0:
            line["cpu_baseline"] = cpu_baseline(spec, args.cpu_sample)
        print(json.dumps(line))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
