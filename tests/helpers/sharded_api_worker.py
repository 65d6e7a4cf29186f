"""One rank of tests/test_gpu_sharded_api.py: a rank-local series through WaveformModes.transform(group=...) /
AsymptoticBondiData.transform(group=...) on the GPU (gloo group, the ranks share device 0: the halos take the round trip
through the host that a one-GPU box forces; with one rank per GPU and backend nccl the same calls move them over xGMI).
Usage: RANK=r WORLD_SIZE=w MASTER_PORT=p python sharded_api_worker.py <out_dir>"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main(out_dir):
    import torch
    import torch.distributed as dist

    import scri_amd
    from scri_amd import sharding, synthetic
    from test_gpu_sharding import _abd_case

    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    res = {}
    try:
        n_times, ell_max = 6000, 8
        t, data, spec = synthetic.workload("cfg3", n_times=n_times)
        kw = dict(spec["kwargs"])
        kw["boost_velocity"] = np.array([1.0, 2.0, 3.0]) * 1e-3
        nm = (ell_max + 1) ** 2 - 4
        # deliberately uneven blocks: the group= path takes whatever rows the ranks hold
        cuts = [0] + [int(n_times * (0.2 + 0.6 * (r + 1) / world)) if r + 1 < world else n_times for r in range(world)]
        i0, i1 = cuts[rank], cuts[rank + 1]

        def series():
            return scri_amd.WaveformModes(t=t[i0:i1], data=np.ascontiguousarray(data[i0:i1, :nm]), ell_min=2, ell_max=ell_max, dataType=scri_amd.h,
                                          frameType=scri_amd.Inertial, r_is_scaled_out=True, m_is_scaled_out=True)

        g = dist.group.WORLD
        for tag, extra, resident in (("host", {}, False), ("device", {}, True), ("overlap", dict(overlap_halo=True), True),
                                     ("columns", dict(partition="columns"), True)):
            w = series()
            if resident:
                w.to_device()
            got = w.transform(group=g, **kw, **extra)
            assert got.is_device_resident == resident
            res[f"wm_{tag}_t"], res[f"wm_{tag}_d"] = got.t, np.array(got.data)
        # AsymptoticBondiData
        u, raw, tr, L = _abd_case(n=3000, ell_max=4)
        j0, j1 = sharding.shard_bounds(u.size, world, rank)
        st = np.zeros(9, dtype=complex)
        st[0], st[2], st[6] = 0.3, 0.05, 0.02
        kw_abd = dict(supertranslation=st, frame_rotation=[0.9, 0.1, -0.3, 0.2], boost_velocity=[2e-3, -1e-3, 3e-3], working_ell_max=2 * L + 2)
        for tag, resident in (("host", False), ("device", True)):
            abd = scri_amd.AsymptoticBondiData(u[j0:j1], L)
            abd._raw_data[:] = raw[:, j0:j1]
            if resident:
                abd = abd.to_device()
            got = abd.transform(group=g, **kw_abd)
            assert got.is_device_resident == resident
            res[f"abd_{tag}_u"], res[f"abd_{tag}_raw"] = got.t, np.array(got._raw_data)
        np.savez(os.path.join(out_dir, f"rank{rank}.npz"), **res)
    finally:
        dist.destroy_process_group()


if __name__ == "__main__":
    main(sys.argv[1])
