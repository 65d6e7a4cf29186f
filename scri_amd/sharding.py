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

    def finish(_in_flight=ops):  # (the send buffers stay referenced until the exchange has been waited for)
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


def plan(t_global, transformation, world_size, have=None):
    """Static plan shared by all ranks: per-rank owned rows, needed rows and the global output window.
    have: the rows each rank owns, [(i0, i1)] contiguous in rank order (default: near-equal blocks, shard_bounds)."""
    from . import engine

    n = len(t_global)
    if have is None:
        have = [shard_bounds(n, world_size, r) for r in range(world_size)]
    else:
        have = [(int(a), int(b)) for a, b in have]
        if len(have) != world_size or have[0][0] != 0 or have[-1][1] != n or any(have[r][1] != have[r + 1][0] for r in range(world_size - 1)) \
                or any(b < a for a, b in have):
            raise ValueError(f"`have` must cut [0, {n}) into {world_size} contiguous blocks in rank order; got {have}")
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
    device = loc.device
    if device.type != "cpu" and dist.get_backend(group) == "gloo":  # gloo moves host memory (dry runs; the production backend is RCCL)
        loc = loc.cpu()
    widest = max(h1 - h0 for h0, h1 in have)
    pad = torch.zeros((widest,) + tuple(loc.shape[1:]), dtype=loc.dtype, device=loc.device)
    pad[: loc.shape[0]] = loc
    gathered = torch.empty((world * widest,) + tuple(loc.shape[1:]), dtype=loc.dtype, device=loc.device)
    dist.all_gather_into_tensor(gathered, pad, group=group)
    out = torch.cat([gathered[r * widest : r * widest + (have[r][1] - have[r][0])] for r in range(world)]).to(device)
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


# ------------------------------------------------------------------------------------------------------------------
# The sharded engine behind one call (north star: "shard the time axis across 8 GPUs ... expose it behind
# scri.WaveformModes.transform so existing callers are unchanged").  One process per GPU holds a contiguous block of rows
# of the series; `ShardedTransform` plans once (host only), and every call exchanges the halos of input modes, runs this
# rank's shard through the C ABI (bms_transform_modes_shard / bms_transform_abd_shard) and returns this rank's output rows
# with their global offset.  `transform_modes_sharded` / `transform_abd_sharded` are the one-shot forms;
# WaveformModes.transform(group=...) and AsymptoticBondiData.transform(group=...) forward to them.


def gather_time_axis(t_local, group=None, ctx=None):
    """Every rank's block of the time axis -> (t_global, have): the global axis and the rows [i0, i1) each rank owns, in rank
    order (8 bytes per sample: the one piece of metadata a rank-local series lacks).  ctx: this rank's engine context -- RCCL
    moves objects through the CURRENT device, which a caller who never called torch.cuda.set_device leaves at 0 on every rank."""
    import contextlib

    import numpy as np
    import torch.distributed as dist

    world = dist.get_world_size(group)
    blocks = [None] * world
    on_device = contextlib.nullcontext()
    if dist.get_backend(group) == "nccl":
        import torch

        from ._lib import default_context

        on_device = torch.cuda.device(int((ctx or default_context()).device))
    with on_device:
        dist.all_gather_object(blocks, np.ascontiguousarray(t_local, dtype=float), group=group)
    have, i0 = [], 0
    for b in blocks:
        have.append((i0, i0 + b.shape[0]))
        i0 += b.shape[0]
    return np.concatenate(blocks), have


class ShardedTransform:
    """This rank's part of ONE BMS transformation of a series whose time axis is split over the ranks of `group`.

    kind "modes": WaveformModes data, rows [own, n_modes]; kind "abd": AsymptoticBondiData storage [6, own, (ell_max+1)^2].
    t_global: the whole time axis (every rank holds it: 8 B per sample); have: per-rank owned rows [(i0, i1)] in rank order
    (default: near-equal contiguous blocks, shard_bounds).  partition: "rows" (time shards + point-to-point halo exchange of
    input modes, plan A of SURVEY 8(e)), "columns" (every rank gathers the whole input, transforms its part of the grid
    columns over all times, one reduce-scatter of the output rows: plan B, for boosts whose time skew makes the row halos
    overlap; WaveformModes only) or "auto" (choose_partition).  overlap=True ("rows", WaveformModes): the outputs that need
    own rows only are transformed while the halos travel, the two edges afterwards (three engine calls instead of one).

    Calling the object with this rank's rows returns (t_out, rows_out, first): rows_out[k] is the output sample whose global
    INPUT index is first + k ("rows": the rank's own outputs, consecutive over the ranks; "columns": block `rank` of the
    reduce-scatter).  Rows may be a torch tensor on the context's device (stays in HBM, RCCL moves the halos device to
    device), or a host array / CPU tensor (the shard call uploads it).  `compute`: the per-shard arithmetic,
    compute(t_global, ext_rows, shard) -> (t_out, data_out, first) with shard = (row0, n_rows, out_i0, out_i1[, part, parts]);
    the default is the engine (C ABI) -- the CPU tests of the multi-rank logic pass the oracle."""

    def __init__(self, kind, t_global, transformation, ell_min=None, ell_max=None, spin_weight=None, conformal_weight=None, type_term=None,
                 group=None, have=None, partition="auto", overlap=False, ctx=None, compute=None):
        import numpy as np
        import torch.distributed as dist

        if kind not in ("modes", "abd"):
            raise ValueError(f"kind {kind!r}: 'modes' or 'abd'")
        self.kind, self.group, self.ctx, self.tr = kind, group, ctx, transformation
        self.t_global = np.ascontiguousarray(t_global, dtype=float)
        self.ell_min, self.ell_max = ell_min, ell_max
        self.spin_weight, self.conformal_weight, self.type_term = spin_weight, conformal_weight, type_term
        if group is None and not dist.is_initialized():  # one process, no group: the degenerate one-rank case (same code path)
            self.rank, self.world, self.backend = 0, 1, None
        else:
            self.rank, self.world = dist.get_rank(group), dist.get_world_size(group)
            self.backend = dist.get_backend(group)
        if have is None:
            self.have, self.need, self.window = plan(self.t_global, transformation, self.world)
        else:
            self.have, self.need, self.window = plan(self.t_global, transformation, self.world, have=have)
        if partition == "auto":
            partition = choose_partition(self.have, self.need) if kind == "modes" and self.world > 1 else "rows"
        if partition not in ("rows", "columns"):
            raise ValueError(f"partition {partition!r}: 'auto', 'rows' or 'columns'")
        if partition == "columns" and kind != "modes":
            raise NotImplementedError("the column partition is wired in for WaveformModes series only")
        self.partition = partition
        self._compute = compute
        i0, i1 = self.have[self.rank]
        self.own = i1 - i0
        self.halo_rows = (i0 - self.need[self.rank][0], self.need[self.rank][1] - i1) if self.need[self.rank][1] > self.need[self.rank][0] else (0, 0)
        self.interior = None
        if overlap and partition == "rows" and kind == "modes" and self.world > 1:
            self.interior = self._plan_interior()
        self._ext = None  # rows [need) of the series, own rows placed once: only the halos move per call
        self._part = None  # "columns": this rank's contribution to all output rows, padded to equal blocks
        self.n_out_rows = max(0, min(i1, self.window[1]) - max(i0, self.window[0]))

    # -- planning helpers
    def _plan_interior(self):
        from . import engine

        i0, i1 = self.have[self.rank]
        a, b = i0 + 2 * self.halo_rows[0] + 8, i1 - 2 * self.halo_rows[1] - 8
        if self.rank == 0:
            a = i0
        if self.rank == self.world - 1:
            b = i1
        while b - a > 64:
            (n0, n1), _ = engine.shard_plan(self.t_global, self.tr, a, b)
            if n0 >= i0 and n1 <= i1:
                break
            a, b = (a + 16 if n0 < i0 else a), (b - 16 if n1 > i1 else b)
        return (a, b) if b - a > 64 else None

    @property
    def n_modes_in(self):
        return (self.ell_max + 1) ** 2 - (0 if self.kind == "abd" else self.ell_min**2)

    @property
    def n_modes_out(self):
        L = self.tr.ell_max_out
        return (L + 1) ** 2 - (0 if self.kind == "abd" else abs(self.spin_weight) ** 2)

    def describe(self):
        """what bench.py prints as `config.sharding`"""
        if self.world == 1:
            return "none"
        if self.partition == "columns":
            return f"grid columns x{self.world}: all-gather of input modes, reduce-scatter of output modes ({self.backend}); scri_amd.sharding.ShardedTransform"
        return f"time axis x{self.world}, point-to-point halo exchange of input modes ({self.backend}); scri_amd.sharding.ShardedTransform"

    # -- the per-shard arithmetic: the engine through the C ABI
    def _engine_call(self, ext, shard, out=None):
        """ext: torch tensor (device of the context or CPU) or numpy rows -> (t_out, rows_out, first)"""
        import numpy as np
        import torch

        from . import engine

        if self._compute is not None:
            host = ext.cpu().numpy() if isinstance(ext, torch.Tensor) else np.asarray(ext)
            return self._compute(self.t_global, host, tuple(int(x) for x in shard))
        on_device = isinstance(ext, torch.Tensor) and ext.device.type == "cuda"
        if on_device:
            self._sync_before_engine(ext)  # (whatever torch queued on its stream -- copies, halo rows -- is in place before the kernels read it)
        if not on_device:
            host = np.ascontiguousarray(ext.numpy() if isinstance(ext, torch.Tensor) else ext)
            if self.kind == "abd":
                return engine.transform_abd(self.t_global, host, self.ell_max, self.tr, ctx=self.ctx, shard=shard)
            return engine.transform_modes(self.t_global, host, self.ell_min, self.ell_max, self.spin_weight, self.conformal_weight,
                                          self.type_term, self.tr, ctx=self.ctx, shard=shard)
        n_alloc = max(0, min(len(self.t_global), int(shard[3])) - max(0, int(shard[2])))
        if self.kind == "abd":
            # (the engine writes field f at out_ptr + f * (out_i1 - out_i0) * n_out and fills its first n_new rows)
            shape = (6, max(int(shard[3]) - int(shard[2]), 1), self.n_modes_out)
            if out is None:
                out = torch.empty(shape, dtype=torch.complex128, device=ext.device)
            elif tuple(out.shape) != shape or not out.is_contiguous():
                raise ValueError(f"`out` must be a contiguous complex128 tensor of shape {shape}")
            t_out, n_new, first = engine.transform_abd(self.t_global, ext.data_ptr(), self.ell_max, self.tr, ctx=self.ctx, shard=shard[:4],
                                                       device=True, out_ptr=out.data_ptr())
            return t_out, out[:, :n_new], first
        if out is None:
            out = torch.empty((max(n_alloc, 1), self.n_modes_out), dtype=torch.complex128, device=ext.device)
        t_out, n_new, first = engine.transform_modes(
            self.t_global, ext.data_ptr(), self.ell_min, self.ell_max, self.spin_weight, self.conformal_weight, self.type_term, self.tr,
            ctx=self.ctx, device=True, ld=ext.stride(0), out_ptr=out.data_ptr(), shard=shard)
        return t_out, out[:n_new], first

    def _sync_before_engine(self, tensor):
        """The halo rows were put in place by torch (RCCL completion waits and copies on torch's current stream).  With the
        context on that same stream the engine's kernels are ordered behind them; otherwise the device is drained first."""
        import torch

        if not isinstance(tensor, torch.Tensor) or tensor.device.type != "cuda" or self._compute is not None:
            return
        same = getattr(self.ctx, "stream_handle", None) == torch.cuda.current_stream(tensor.device).cuda_stream
        if not same:
            torch.cuda.synchronize(tensor.device)

    def own_rows_view(self, like=None):
        """"rows", WaveformModes: the place of this rank's own rows inside the exchange buffer.  A caller that keeps its
        series THERE (fills this view once, passes it to every call) saves the copy of its rows per call."""
        import torch

        if self.kind != "modes" or self.partition != "rows" or self.world == 1:
            return None
        n0, n1 = self.need[self.rank]
        if self._ext is None or (like is not None and (self._ext.device != like.device)):
            device = like.device if like is not None else "cpu"
            self._ext = torch.empty((max(n1 - n0, self.own), self.n_modes_in), dtype=torch.complex128, device=device)
        lo = self.have[self.rank][0] - n0 if n1 > n0 else 0
        return self._ext[lo : lo + self.own]

    def __call__(self, local, out=None):
        import numpy as np
        import torch

        as_numpy = not isinstance(local, torch.Tensor)
        loc = torch.from_numpy(np.ascontiguousarray(local, dtype=np.complex128)) if as_numpy else local
        if self.backend == "nccl" and loc.device.type == "cpu" and self._compute is None:
            # RCCL moves device memory: a host-resident series goes to this rank's GPU once and its result comes back to the host
            from . import _lib

            ctx = self.ctx if self.ctx is not None else _lib.default_context()
            self.ctx = ctx
            loc = loc.to(torch.device("cuda", ctx.device))
        rows_axis = 1 if self.kind == "abd" else 0
        if loc.shape[rows_axis] != self.own:
            raise ValueError(f"rank {self.rank} owns rows [{self.have[self.rank][0]}, {self.have[self.rank][1]}): got {loc.shape[rows_axis]} rows")
        rank, world = self.rank, self.world
        i0, i1 = self.have[rank]

        def back(res):
            t_out, rows, first = res
            if as_numpy and isinstance(rows, torch.Tensor):
                rows = rows.cpu().numpy()
            elif not as_numpy and not isinstance(rows, torch.Tensor):
                rows = torch.from_numpy(np.ascontiguousarray(rows)).to(loc.device)
            return np.asarray(t_out), rows, int(first)

        if world == 1:
            return back(self._engine_call(loc, (0, self.own, 0, self.own), out=out))
        if self.partition == "columns":
            full = replicate_rows(loc, self.have, group=self.group)
            n = len(self.t_global)
            n_new_all = self.window[1] - self.window[0]
            total, _ = padded_rows(n_new_all, world)
            t_all, part, first = self._engine_call(full, (0, n, 0, n, rank, world))
            if not isinstance(part, torch.Tensor):
                part = torch.from_numpy(np.ascontiguousarray(part))
            if self._part is None or self._part.device != part.device or self._part.shape[1] != part.shape[1]:
                self._part = torch.zeros((total, part.shape[1]), dtype=torch.complex128, device=part.device)
            self._part[: part.shape[0]] = part
            rows, (r0, r1) = reduce_scatter_rows(self._part, n_new_all, group=self.group)
            return back((np.asarray(t_all)[r0:r1], rows, self.window[0] + r0))
        n0, n1 = self.need[rank]
        if n1 <= n0:  # no output falls into this rank's rows: it only serves its neighbours' halos
            exchange_halos(loc, self.have[rank], (i0, i0), self.have, self.need, group=self.group, dim=rows_axis)
            shape = (6, 0, self.n_modes_out) if self.kind == "abd" else (0, self.n_modes_out)
            return back((np.empty(0), torch.empty(shape, dtype=torch.complex128, device=loc.device), max(i0, self.window[0])))
        if self.kind == "abd":
            ext = exchange_halos(loc, self.have[rank], self.need[rank], self.have, self.need, group=self.group, dim=1)
            return back(self._engine_call(ext, (n0, ext.shape[1], i0, i1), out=out))
        view = self.own_rows_view(like=loc)
        if view.data_ptr() != loc.data_ptr():
            view.copy_(loc)
        ext_buf = self._ext[: n1 - n0]
        if self.interior is None:
            ext = exchange_halos(view, self.have[rank], self.need[rank], self.have, self.need, group=self.group, out=ext_buf)
            return back(self._engine_call(ext, (n0, ext.shape[0], i0, i1), out=out))
        # interior outputs from own rows while the halos travel; then the two edges from the completed rows
        a, b = self.interior
        pending = exchange_halos(view, self.have[rank], self.need[rank], self.have, self.need, group=self.group, out=ext_buf, wait=False)
        if out is None and self._compute is None and view.device.type == "cuda":
            out = torch.empty((max(self.n_out_rows, 1), self.n_modes_out), dtype=torch.complex128, device=view.device)
        first_all = max(i0, self.window[0])

        def piece(src, row0, o0, o1):
            o0c, o1c = max(o0, self.window[0]), min(o1, self.window[1])
            if o1c <= o0c:
                return None
            dst = out[o0c - first_all : o1c - first_all] if out is not None else None
            return self._engine_call(src, (row0, src.shape[0], o0, o1), out=dst)

        mid = piece(view, i0, a, b)  # own rows only: runs under the exchange
        ext = pending()
        parts = [p for p in (piece(ext, n0, i0, a), mid, piece(ext, n0, b, i1)) if p is not None]
        t_out = np.concatenate([np.asarray(p[0]) for p in parts]) if parts else np.empty(0)
        if out is not None:
            rows = out[: t_out.shape[0]]
        elif parts:
            rows = torch.cat([p[1] if isinstance(p[1], torch.Tensor) else torch.from_numpy(np.ascontiguousarray(p[1])) for p in parts])
        else:
            rows = torch.empty((0, self.n_modes_out), dtype=torch.complex128, device=loc.device)
        return back((t_out, rows, parts[0][2] if parts else first_all))


def transform_modes_sharded(local_rows, t_global, ell_min, ell_max, spin_weight, conformal_weight, type_term, transformation, group=None,
                            have=None, partition="auto", overlap=False, ctx=None, compute=None):
    """This rank's output rows of the BMS transformation of a WaveformModes series sharded over `group` (one call; a caller
    that repeats the transformation keeps a ShardedTransform).  Returns (t_out, rows_out, first_global_index)."""
    st = ShardedTransform("modes", t_global, transformation, ell_min, ell_max, spin_weight, conformal_weight, type_term, group=group, have=have,
                          partition=partition, overlap=overlap, ctx=ctx, compute=compute)
    return st(local_rows)


def transform_abd_sharded(local_raw, u_global, ell_max, transformation, group=None, have=None, ctx=None, compute=None):
    """The same for AsymptoticBondiData storage: local_raw [6, own rows, (ell_max+1)^2] -> (u_out, raw_out [6, n', n_out], first)."""
    st = ShardedTransform("abd", u_global, transformation, ell_max=ell_max, group=group, have=have, partition="rows", ctx=ctx, compute=compute)
    return st(local_raw)
