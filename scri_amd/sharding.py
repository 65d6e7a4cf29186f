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


def exchange_halos(local, have, need, all_have, all_need, group=None, dim=0, out=None, wait=True):
    """Return rows [need[0], need[1]) of the global array, given this rank's rows `local` = [have[0], have[1]).

    local: torch tensor [rows, cols] float64/complex128 (CPU for gloo, GPU for nccl); `dim` names the time axis
    when it is not the first one (AsymptoticBondiData storage is [6, rows, modes]: dim=1).
    all_have / all_need: per-rank (start, stop) lists known to every rank (static plan, no communication).
    Rows outside every rank's range are never requested.
    out: optional preallocated result (time axis first, shape of the return value) whose own-row block already holds
    `local` (e.g. `local` is a view of it): then only the halo rows move.
    wait=False: start the transfers and return a function that completes them and returns the rows -- the caller works on
    its own rows in between (bench.py --overlap-halo)."""
    import torch
    import torch.distributed as dist

    rank = dist.get_rank(group)
    world = dist.get_world_size(group)
    is_complex = local.is_complex()
    if dim != 0:
        local = local.movedim(dim, 0)
    # gloo moves host memory only: there the halo rows (not the whole shard) take a round trip through the host (dry runs of
    # the multi-rank path on a box without RCCL peers; the production backend is nccl = RCCL, device to device)
    device = local.device
    via_host = device.type != "cpu" and dist.get_backend(group) == "gloo"
    wire = torch.device("cpu") if via_host else device
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
            buf = torch.empty((b - a,) + tuple(loc.shape[1:]), dtype=loc.dtype, device=wire)
            recv_bufs.append((a, b, buf))
            ops.append(dist.P2POp(dist.irecv, buf, peer, group))
        # rows the peer needs that I own
        a, b = max(all_need[peer][0], have[0]), min(all_need[peer][1], have[1])
        if b > a:
            ops.append(dist.P2POp(dist.isend, loc[a - have[0] : b - have[0]].contiguous().to(wire), peer, group))
    reqs = dist.batch_isend_irecv(ops) if ops else []

    def finish():
        for req in reqs:
            req.wait()
        o = out
        for a, b, buf in recv_bufs:
            o[a - need[0] : b - need[0]] = buf.to(o.device)
        if prefilled:
            return result
        o = torch.view_as_complex(o) if is_complex else o
        return o.movedim(0, dim).contiguous() if dim != 0 else o

    return finish() if wait else finish


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


# ------------------------------------------------------------------------------------------------------------------
# Plan B of SURVEY section 8(e): partition the grid COLUMNS instead of the time axis.  A strong boost skews the time
# axis of each direction by up to beta |u|, so the row halo of a time shard grows until the shards overlap almost
# entirely (beta = 0.1 at |u| = 1e4 dt: 1e3 rows each side for every 1e4 rows of shard).  Splitting the columns needs no
# halo at all: every rank holds the whole input series, synthesises and splines its own columns over all times, and
# contributes its part of the (linear) analysis; one reduce-scatter over the ranks sums the parts and leaves each rank
# with a block of output rows.


def choose_partition(have, need, max_halo_fraction=0.25):
    """"rows" (time shards + halo exchange) unless some rank's halo exceeds `max_halo_fraction` of its shard: "columns"."""
    for (h0, h1), (n0, n1) in zip(have, need):
        own = h1 - h0
        if n1 > n0 and own > 0 and max(h0 - n0, n1 - h1) > max_halo_fraction * own:
            return "columns"
    return "rows"


def replicate_rows(local, have, group=None, dim=0):
    """Every rank's rows [have[r][0], have[r][1]) -> the whole series on every rank (one all_gather; the input modes are
    a few GB at most against 288 GB of HBM)."""
    import torch
    import torch.distributed as dist

    world = dist.get_world_size(group)
    is_complex = local.is_complex()
    if dim != 0:
        local = local.movedim(dim, 0)
    loc = (torch.view_as_real(local) if is_complex else local).contiguous()
    widest = max(h1 - h0 for h0, h1 in have)
    pad = torch.zeros((widest,) + tuple(loc.shape[1:]), dtype=loc.dtype, device=loc.device)
    pad[: loc.shape[0]] = loc
    gathered = torch.empty((world * widest,) + tuple(loc.shape[1:]), dtype=loc.dtype, device=loc.device)
    dist.all_gather_into_tensor(gathered, pad, group=group)
    out = torch.cat([gathered[r * widest : r * widest + (have[r][1] - have[r][0])] for r in range(world)])
    out = torch.view_as_complex(out) if is_complex else out
    return out.movedim(0, dim).contiguous() if dim != 0 else out


def padded_rows(n_rows, world_size):
    """Row count of the buffer a rank's contribution is written into: equal blocks for the reduce-scatter."""
    block = -(-int(n_rows) // int(world_size))
    return block * world_size, block


def reduce_scatter_rows(partial, n_rows, group=None):
    """Sum the ranks' contributions and leave rows [r0, r1) of the sum on this rank.

    partial: torch complex128/float64 [padded_rows(n_rows, world)[0], ...] with this rank's contribution in the first
    n_rows rows (the padding is ignored).  Returns (block [r1 - r0, ...], (r0, r1))."""
    import torch
    import torch.distributed as dist

    rank = dist.get_rank(group)
    world = dist.get_world_size(group)
    total, block = padded_rows(n_rows, world)
    if partial.shape[0] != total:
        raise ValueError(f"contribution has {partial.shape[0]} rows, expected the padded count {total}")
    is_complex = partial.is_complex()
    src = torch.view_as_real(partial) if is_complex else partial
    r0, r1 = min(n_rows, rank * block), min(n_rows, (rank + 1) * block)
    if dist.get_backend(group) == "gloo":  # gloo has no reduce-scatter: all-reduce and keep the own block (CPU tests)
        host = src.cpu() if src.device.type != "cpu" else src.clone()
        dist.all_reduce(host, group=group)
        mine = host[rank * block : (rank + 1) * block].to(src.device)
    else:
        mine = torch.empty((block,) + tuple(src.shape[1:]), dtype=src.dtype, device=src.device)
        dist.reduce_scatter_tensor(mine, src, group=group)
    mine = torch.view_as_complex(mine) if is_complex else mine
    return mine[: r1 - r0], (r0, r1)
