"""Time-axis sharding of a BMS transformation over several GPUs (one process per GPU, torch.distributed;
backend "nccl" = RCCL over xGMI on the GPU box, "gloo" in the CPU tests).

Each rank owns a contiguous block of input rows [i0, i1) and produces the output samples with the same global
indices.  The per-pixel cubic spline and the boost/supertranslation time skew make an output sample depend on
input rows up to `H = H_spline + H_skew` away; `bms_shard_plan` returns the exact row
range a rank needs, and `exchange_halos` fetches the missing rows from the neighbouring ranks with
point-to-point sends -- the only communication of the path (no collective on the data).
"""


def shard_bounds(n_times, world_size, rank):
    """Contiguous, near-equal split of [0, n_times)."""
    base, rem = divmod(int(n_times), int(world_size))
    i0 = rank * base + min(rank, rem)
    return i0, i0 + base + (1 if rank < rem else 0)


def exchange_halos(local, have, need, all_have, all_need, group=None, dim=0, out=None):
    """Return rows [need[0], need[1]) of the global array, given this rank's rows `local` = [have[0], have[1]).

    local: torch tensor [rows, cols] float64/complex128 (CPU for gloo, GPU for nccl); `dim` names the time axis
    when it is not the first one (AsymptoticBondiData storage is [6, rows, modes]: dim=1).
    all_have / all_need: per-rank (start, stop) lists known to every rank (static plan, no communication).
    Rows outside every rank's range are never requested.
    out: optional preallocated result (time axis first, shape of the return value) whose own-row block already holds
    `local` (e.g. `local` is a view of it): then only the halo rows move."""
    import torch
    import torch.distributed as dist

    rank = dist.get_rank(group)
    world = dist.get_world_size(group)
    is_complex = local.is_complex()
    if dim != 0:
        local = local.movedim(dim, 0)
    # gloo moves host memory only: device tensors take a round trip through the host (dry runs of the multi-rank
    # path on a box without RCCL peers; the production backend is nccl = RCCL, device to device)
    device = local.device
    if device.type != "cpu" and dist.get_backend(group) == "gloo":
        local = local.cpu()
    loc = torch.view_as_real(local) if is_complex else local
    out_shape = (need[1] - need[0],) + tuple(loc.shape[1:])
    prefilled = out is not None
    if prefilled:
        if dim != 0 or out.device != loc.device:
            raise ValueError("a preallocated result needs the time axis first and the device of `local`")
        result = out
        out = torch.view_as_real(out) if is_complex else out
        if tuple(out.shape) != out_shape:
            raise ValueError(f"preallocated result has shape {tuple(out.shape)}, expected {out_shape}")
    else:
        out = torch.empty(out_shape, dtype=loc.dtype, device=loc.device)
        # own rows
        a, b = max(need[0], have[0]), min(need[1], have[1])
        if b > a:
            out[a - need[0] : b - need[0]] = loc[a - have[0] : b - have[0]]
    ops, recv_bufs = [], []
    for peer in range(world):
        if peer == rank:
            continue
        # rows I need that the peer owns
        a, b = max(need[0], all_have[peer][0]), min(need[1], all_have[peer][1])
        if b > a:
            buf = torch.empty((b - a,) + tuple(loc.shape[1:]), dtype=loc.dtype, device=loc.device)
            recv_bufs.append((a, b, buf))
            ops.append(dist.P2POp(dist.irecv, buf, peer, group))
        # rows the peer needs that I own
        a, b = max(all_need[peer][0], have[0]), min(all_need[peer][1], have[1])
        if b > a:
            ops.append(dist.P2POp(dist.isend, loc[a - have[0] : b - have[0]].contiguous(), peer, group))
    if ops:
        for req in dist.batch_isend_irecv(ops):
            req.wait()
    for a, b, buf in recv_bufs:
        out[a - need[0] : b - need[0]] = buf
    if prefilled:
        return result
    out = torch.view_as_complex(out) if is_complex else out
    if out.device != device:
        out = out.to(device)
    return out.movedim(0, dim).contiguous() if dim != 0 else out


def plan(t_global, transformation, world_size):
    """Static plan shared by all ranks: per-rank owned rows, needed rows and the global output window."""
    from . import engine

    n = len(t_global)
    have = [shard_bounds(n, world_size, r) for r in range(world_size)]
    need, window = [], None
    for r in range(world_size):
        nr, window = engine.shard_plan(t_global, transformation, have[r][0], have[r][1])
        # a rank with no output still "needs" nothing
        need.append(tuple(nr) if nr[1] > nr[0] else (have[r][0], have[r][0]))
    return have, need, tuple(window)
