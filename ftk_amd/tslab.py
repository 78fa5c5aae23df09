"""Multi-GPU host logic: the space-time lattice is cut into contiguous timestep slabs, one process per GPU.

The reference only decomposes SPACE over MPI ranks (include/ftk/filters/regular_tracker.hh:127-149,
include/ftk/mesh/lattice_partitioner.hh:97-196) and leaves time "uncuttable" (include/ftk/mesh/lattice.hh:32,59).  Every
(timestep, scope) sweep is independent given slices t, t+1 and the scalar factor(t), so here rank r owns timesteps
[r*nt/G, (r+1)*nt/G), sweeps ordinal(t) and interval[t, t+1] for each of them, and needs two things from its neighbours:
  * the FIRST slice of rank r+1 (one point-to-point transfer over xGMI; no ring, no all-reduce);
  * every slice's min|V != 0| -- an all_gather of nt doubles -- because the reference's quantisation factor is a sticky running
    minimum over all slices pushed so far (include/ftk/filters/critical_point_tracker.hh:850-864): at the sweep of
    current = c it is min over slices 0 .. min(c+1, nt-1).
torch.distributed is the plumbing (backend "nccl" = RCCL on ROCm; "gloo" in the CPU tests)."""
import math

import numpy as np

DBL_MAX = float(np.finfo(np.float64).max)
E_NOSLICE, E_UNSUPPORTED = -4, -5      # include/ftkx.h


def slab_range(nt, world, rank):
    """timesteps [t0, t1) owned by `rank`: contiguous, sizes differ by at most one, empty slabs allowed when world > nt"""
    return (rank * nt) // world, ((rank + 1) * nt) // world


def owner_of(t, nt, world):
    for r in range(world):
        t0, t1 = slab_range(nt, world, r)
        if t0 <= t < t1:
            return r
    raise ValueError(t)


def scaling_factor(resolution, minbits=8, maxbits=21):
    """critical_point_tracker.hh:850-864"""
    nbits = int(math.ceil(math.log2(1.0 / resolution)))
    return 1 << max(minbits, min(nbits, maxbits))


def factors_from_resolutions(res):
    """res[t] = ndarray::resolution() of slice t for ALL nt slices -> factor used by the sweep of current_timestep == t"""
    nt = len(res)
    run = np.minimum.accumulate(np.minimum(np.asarray(res, dtype=np.float64), DBL_MAX))
    vals = run[np.minimum(np.arange(1, nt + 1), nt - 1)].tolist()
    out, last_v, last_f = [], None, None
    for v in vals:                  # (the running minimum changes a handful of times: one log2 per change, not per timestep)
        if v != last_v:
            last_v, last_f = v, scaling_factor(v)
        out.append(last_f)
    return out


def global_factors(local_res, nt, group=None, local_max=None):
    """local_res: {t: resolution} of the slices this rank owns (local_max: {t: max finite |V|}, optional).  One all_gather of
    2*nt doubles; unowned entries are DBL_MAX / 0 and the ranks' vectors are combined with an elementwise min / max.
    Returns (factors[nt], res[nt]) or, with local_max, (factors, res, maxabs[nt])."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group)
    dev = "cuda" if dist.get_backend(group) == "nccl" else "cpu"
    # built on the host and moved in ONE transfer each way (element-wise writes to a device tensor are a launch apiece)
    host = np.zeros((2 * nt,), dtype=np.float64)
    host[:nt] = DBL_MAX
    for t, r in local_res.items():
        host[t] = r
    for t, v in (local_max or {}).items():
        host[nt + t] = v
    mine = torch.from_numpy(host).to(dev)
    gathered = torch.empty((world * 2 * nt,), dtype=torch.float64, device=dev)
    dist.all_gather_into_tensor(gathered, mine, group=group)
    allv = gathered.cpu().numpy().reshape(world, 2 * nt)
    res = allv[:, :nt].min(axis=0)
    mx = allv[:, nt:].max(axis=0)
    if local_max is None:
        return factors_from_resolutions(res), res
    return factors_from_resolutions(res), res, mx


def exchange_halo(first_slice, recv_buffer, nt, group=None):
    """Each rank that owns timesteps sends its FIRST slice to the owner of the preceding timestep and receives the first
    slice of the following slab into recv_buffer.  Returns True if recv_buffer was filled (i.e. this rank's slab is not
    the last one).  Point-to-point only: xGMI is a mesh of links, a neighbour transfer uses one of them at full rate."""
    import torch
    import torch.distributed as dist
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    t0, t1 = slab_range(nt, world, rank)
    # gloo (CPU tests, or the single-GPU dry run of bench.py) cannot move device tensors: stage through host memory
    staged = dist.get_backend(group) != "nccl" and (first_slice.is_cuda or recv_buffer.is_cuda)
    send_t = first_slice.cpu() if staged else first_slice
    recv_t = torch.empty(recv_buffer.shape, dtype=recv_buffer.dtype) if staged else recv_buffer
    ops = []
    got = False
    if t1 > t0:
        if t0 > 0:
            ops.append(dist.P2POp(dist.isend, send_t, owner_of(t0 - 1, nt, world), group))
        if t1 < nt:
            ops.append(dist.P2POp(dist.irecv, recv_t, owner_of(t1, nt, world), group))
            got = True
    if ops:
        for req in dist.batch_isend_irecv(ops):
            req.wait()
    if staged and got:
        recv_buffer.copy_(recv_t)
    return got


def _p2p(ops):
    import torch.distributed as dist
    if ops:
        for req in dist.batch_isend_irecv(ops):
            req.wait()


CELL_CAP = 4096      # cells a patch request carries; more survivors than that and the whole slice is cheaper anyway


def compact_halo_masks(ctx, own, nt, scalar_input, group=None, mask_factor=256, max_abs=None):
    """Compact halo, step 1 (instead of exchange_halo's whole slice): every rank sends the sign masks of its FIRST slice -- the
    summary array and the mask words the summaries do not describe (the slice must have been prepared) -- to the owner of the
    preceding timestep, which pushes them as a masks-only slice.  ONE message each way, of a size both sides know from the mesh
    (ftkx_packed_masks_bytes); its header is written and read on the device, so neither side waits for the other's numbers.
    mask_factor: the hint the slices were prepared under; max_abs: {t: max |v|} of every slice (the all_gather of the reductions).
    Returns (t_masked or None, bytes sent, bytes received)."""
    import torch
    import torch.distributed as dist
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    dev = torch.device("cuda", torch.cuda.current_device()) if dist.get_backend(group) == "nccl" else torch.device("cpu")
    t0, t1 = (own[0], own[-1] + 1) if own else (0, 0)
    send_to = owner_of(t0 - 1, nt, world) if own and t0 > 0 else None
    recv_from = owner_of(t1, nt, world) if own and t1 < nt else None
    nbytes, _cap = ctx.packed_masks_bytes()
    if nbytes == 0:
        raise RuntimeError("compact halo: this mesh has no summarised masks")
    sent = received = 0
    ops = []
    if send_to is not None:
        out = torch.empty((nbytes,), dtype=torch.uint8, device=dev)
        ctx.export_masks_packed(t0, out)
        ops.append(dist.P2POp(dist.isend, out, send_to, group)); sent += nbytes
    if recv_from is not None:
        inn = torch.empty((nbytes,), dtype=torch.uint8, device=dev)
        ops.append(dist.P2POp(dist.irecv, inn, recv_from, group)); received += nbytes
    _p2p(ops)
    if recv_from is not None:
        ctx.push_masked_slice_packed(t1, scalar_input, inn, mask_factor, 0.0 if max_abs is None else float(max_abs[t1]))
        return t1, sent, received
    return None, sent, received


def compact_halo_patches(ctx, own, nt, t_masked, first_slice=None, halo_buffer=None, push_full=None, group=None):
    """Compact halo, step 2 (after sweep_enqueue, before sweep_collect): cull; the cells whose exact test reads the masked slice go to
    its owner, which answers with the input values around them (6^nd vertices per cell); they are scattered into the masked slice.
    ONE round trip: a fixed-size request (count + up to CELL_CAP cells), a reply whose size both sides derive from the count.  Where
    patches would move MORE than the slice itself (hit-dense data on small slices: thousands of surviving cells), or the packed masks
    did not fit their message, the receiver asks for the whole slice instead (count -1): the owner sends `first_slice`, the receiver
    takes it into `halo_buffer` and hands it to `push_full(t, tensor)`, which must cancel the pending sweeps (ftkx_sweep_cancel), push
    it like any other slice and leave it to the caller to enqueue the sweeps again.
    Returns (cells requested or -1, bytes sent, bytes received)."""
    import torch
    import torch.distributed as dist
    from . import FtkxError
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    dev = torch.device("cuda", torch.cuda.current_device()) if dist.get_backend(group) == "nccl" else torch.device("cpu")
    t0, t1 = (own[0], own[-1] + 1) if own else (0, 0)
    serve = owner_of(t0 - 1, nt, world) if own and t0 > 0 else None          # the rank that holds OUR first slice as masks only
    ask = owner_of(t1, nt, world) if t_masked is not None else None
    sent = received = 0
    cells, want_full, fatal = None, False, None
    if ask is not None:
        try:
            cells = ctx.sweep_cull(t_masked, torch, dev)
        except FtkxError as e:         # the masks did not fit their message, or do not serve this factor: the slice itself
            if e.code not in (E_NOSLICE, E_UNSUPPORTED):
                raise                  # (a device error is not a reason to ask for the slice)
            cells, want_full = torch.zeros((0,), dtype=torch.int64, device=dev), True
        if halo_buffer is not None and push_full is not None:
            want_full = want_full or len(cells) > CELL_CAP or len(cells) * ctx.patch_doubles() * 8 > halo_buffer.numel() * halo_buffer.element_size() // 2
        elif want_full or len(cells) > CELL_CAP:
            # nobody can take the slice here: say so to the owner too (a request of -2 = "giving up"), so that it does not block in its receive
            fatal = RuntimeError("compact halo: the whole slice is needed but no halo_buffer / push_full was given")
    ops = []
    if ask is not None:
        req = torch.zeros((1 + CELL_CAP,), dtype=torch.int64, device=dev)
        req[0] = -2 if fatal is not None else (-1 if want_full else len(cells))
        if not want_full and len(cells):
            req[1:1 + len(cells)] = cells
        ops.append(dist.P2POp(dist.isend, req, ask, group)); sent += req.numel() * 8
    if serve is not None:
        theirs = torch.zeros((1 + CELL_CAP,), dtype=torch.int64, device=dev)
        ops.append(dist.P2POp(dist.irecv, theirs, serve, group)); received += theirs.numel() * 8
    _p2p(ops)
    n_theirs = int(theirs[0].item()) if serve is not None else 0              # (the one number the owner has to see on the host: it sizes the reply)
    if fatal is not None:
        raise fatal
    if n_theirs == -2:
        raise RuntimeError("compact halo: the neighbour below gave up (it needs the whole slice and cannot take it)")
    staged = dist.get_backend(group) != "nccl"
    ops = []
    if serve is not None and n_theirs < 0:
        src = first_slice.cpu() if (staged and first_slice.is_cuda) else first_slice
        ops.append(dist.P2POp(dist.isend, src, serve, group)); sent += src.numel() * src.element_size()
    elif serve is not None and n_theirs > 0:
        out = ctx.gather_patches(t0, theirs[1:1 + n_theirs].contiguous(), torch)
        ops.append(dist.P2POp(dist.isend, out, serve, group)); sent += out.numel() * 8
    mine = dst = None
    if ask is not None and want_full:
        dst = torch.empty(halo_buffer.shape, dtype=halo_buffer.dtype) if (staged and halo_buffer.is_cuda) else halo_buffer
        ops.append(dist.P2POp(dist.irecv, dst, ask, group)); received += dst.numel() * dst.element_size()
    elif ask is not None and len(cells):
        mine = torch.empty((len(cells) * ctx.patch_doubles(),), dtype=torch.float64, device=dev)
        ops.append(dist.P2POp(dist.irecv, mine, ask, group)); received += mine.numel() * 8
    _p2p(ops)
    if dst is not None:
        if dst is not halo_buffer:
            halo_buffer.copy_(dst)
        push_full(t_masked, halo_buffer)
    if mine is not None:
        ctx.scatter_patches(t_masked, cells, mine)
    return (-1 if want_full else (len(cells) if cells is not None else 0)), sent, received


class SlabSeries:
    """One rank's DEVICE-DRIVEN pass over its timestep slab (include/ftkx.h: ftkx_series_dist_*), with the ranks' links -- the sticky
    running minimum across slabs (critical_point_tracker.hh:850-864) and the compact halo -- as collectives queued between the stages:

        begin (masks, reduction, contribution, outgoing masks)  | all_gather of 4 doubles per rank; masks -> lower neighbour
        cull  (masks imported, factors, cull, request)          | request -> upper neighbour
        serve (patches around the neighbour's cells)            | reply -> lower neighbour
        finish (patches scattered, exact test, records)         | complete(): the ONE host wait of the pass

    Backend nccl (= RCCL): every collective is queued on torch's current stream, which must be the context's stream
    (`torch.cuda.set_stream(s); ctx.set_stream(s.cuda_stream)`): nothing waits on the host.  Backend gloo (dry runs, tests): the same
    stages with the messages staged through host memory.  Two passes may be in flight (submit, submit, complete, ...).
    Where the halo is needed as a whole slice (request -1) both sides learn it from the same number, exchange `first_slice`, and the
    asker sweeps again with the full slice (`push` = ctx.push_scalar_slice / push_slice); the next pass starts compact again."""

    def __init__(self, ctx, nt, own, scalar_input, torch, device, first_slice=None, group=None):
        import torch.distributed as dist
        self.ctx, self.nt, self.own, self.scalar, self.torch, self.dist, self.group = ctx, nt, list(own), scalar_input, torch, dist, group
        self.world, self.rank = dist.get_world_size(group), dist.get_rank(group)
        self.staged = dist.get_backend(group) != "nccl"
        self.dev = device
        t0, t1 = (self.own[0], self.own[-1] + 1) if self.own else (0, 0)
        self.t_halo = t1 if (self.own and t1 < nt) else None
        self.lower = owner_of(t0 - 1, nt, self.world) if self.own and t0 > 0 else None      # the rank whose last interval sweep reads OUR first slice
        self.upper = owner_of(t1, nt, self.world) if self.t_halo is not None else None
        self.first_slice = first_slice
        self.full_halo = None
        self.ts = np.array(self.own, dtype=np.int32)
        self.scopes = np.array([3 if t + 1 < nt else 1 for t in self.own], dtype=np.int32)      # FTKX_SCOPE_BOTH / _ORDINAL
        nbytes, _ = ctx.packed_masks_bytes() if self.own else (0, 0)
        cells = ctx.series_dist_cells() if self.own else 0
        pd = ctx.patch_doubles() if self.own else 0
        if self.own and (nbytes == 0 or cells == 0) and (self.lower is not None or self.upper is not None):
            raise RuntimeError("slab pass: this mesh has no summarised masks (use the host-driven protocol)")
        f64, i64, u8 = torch.float64, torch.int64, torch.uint8
        z = lambda n, dt: torch.zeros((max(int(n), 1),), dtype=dt, device=device)
        self.sets = []
        for _ in range(2):
            self.sets.append(dict(contrib=torch.tensor([DBL_MAX, 0.0, DBL_MAX, 0.0], dtype=f64, device=device), gathered=z(4 * self.world, f64),
                                  masks_out=z(nbytes, u8) if self.lower is not None else None, masks_in=z(nbytes, u8) if self.upper is not None else None,
                                  req_out=z(1 + cells, i64) if self.upper is not None else None, req_in=z(1 + cells, i64) if self.lower is not None else None,
                                  reply_out=z(cells * pd, f64) if self.lower is not None else None, reply_in=z(cells * pd, f64) if self.upper is not None else None))
        # (nccl) the masks' way to the lower neighbour runs on a stream of its own: it starts as soon as the first slice's masks are packed,
        # next to the mask kernel of the slab's other slices; the context's stream waits for it in front of the cull
        self.side = torch.cuda.Stream(device=device) if (not self.staged and self.own and (self.lower is not None or self.upper is not None)) else None
        self.k = 0
        self.open = []              # buffer sets of the passes in flight, oldest first
        self.stash = []             # outcomes of passes completed early (a recovery needed the context free)
        self.bytes_sent = self.bytes_received = 0
        self.fallbacks = 0

    # ---- the collectives: queued on the stream (nccl) or staged through the host (gloo) ----
    def _all_gather(self, out, mine):
        if self.staged:
            self.torch.cuda.current_stream().synchronize() if mine.is_cuda else None
            h = self.torch.empty((out.numel(),), dtype=out.dtype)
            self.dist.all_gather_into_tensor(h, mine.cpu(), group=self.group)
            out.copy_(h)
        else:
            self.dist.all_gather_into_tensor(out, mine, group=self.group)

    def _exchange(self, send, to, recv, frm):
        ops, back = [], None
        if to is not None:
            if self.staged and send.is_cuda:
                self.torch.cuda.current_stream().synchronize()
            ops.append(self.dist.P2POp(self.dist.isend, send.cpu() if self.staged else send, to, self.group))
            self.bytes_sent += send.numel() * send.element_size()
        if frm is not None:
            back = self.torch.empty(recv.shape, dtype=recv.dtype) if self.staged else recv
            ops.append(self.dist.P2POp(self.dist.irecv, back, frm, self.group))
            self.bytes_received += recv.numel() * recv.element_size()
        if ops:
            for r in self.dist.batch_isend_irecv(ops):
                r.wait()               # (nccl: the CURRENT STREAM waits, not the host)
        if frm is not None and self.staged:
            recv.copy_(back)

    def submit(self, running_resolution=None):
        b = self.sets[self.k]
        self.k ^= 1
        if not self.own:               # (more ranks than timesteps: this rank only takes part in the all_gather)
            self._all_gather(b["gathered"], b["contrib"])
            self.open.append(b)
            return
        ctx = self.ctx
        ctx.series_dist_begin(self.ts, self.scopes, running_resolution, self.rank, self.world, self.upper, b["contrib"], b["gathered"], b["masks_out"],
                              side_stream=self.side.cuda_stream if self.side is not None else None)
        if self.side is not None:
            main = self.torch.cuda.current_stream()
            with self.torch.cuda.stream(self.side):
                self._exchange(b["masks_out"], self.lower, b["masks_in"], self.upper)
            self._all_gather(b["gathered"], b["contrib"])
            main.wait_stream(self.side)
        else:
            self._all_gather(b["gathered"], b["contrib"])
            self._exchange(b["masks_out"], self.lower, b["masks_in"], self.upper)
        ctx.series_dist_cull(b["masks_in"], b["req_out"])
        self._exchange(b["req_out"], self.upper, b["req_in"], self.lower)
        ctx.series_dist_serve(b["req_in"], b["reply_out"])
        self._exchange(b["reply_out"], self.lower, b["reply_in"], self.upper)
        ctx.series_dist_finish(b["reply_in"])
        b["running_in"] = DBL_MAX if running_resolution is None else float(running_resolution)
        self.open.append(b)

    def _complete_local(self, copy):
        from . import FtkxError
        try:
            recs, f, run = self.ctx.sweep_series_complete(copy=copy)
            failed = False
        except FtkxError as e:
            if e.code != E_NOSLICE:
                raise
            recs, f, run, failed = None, None, None, True
        asked, served, g = self.ctx.series_dist_status(self.world)
        return dict(recs=recs, f=f, run=run, failed=failed, asked=asked, served=served, gathered=g, path=self.ctx.series_last_path())

    def complete(self, copy=True, push=None):
        """the oldest pass in flight -> (records, factors, running resolution)"""
        b = self.open.pop(0)
        if not self.own:
            return np.zeros((0,), dtype=_empty_records_dtype()), np.zeros((0,), dtype=np.uint64), DBL_MAX
        o = self.stash.pop(0) if self.stash else self._complete_local(copy)
        need, give = o["asked"] < 0 and self.upper is not None, o["served"] < 0 and self.lower is not None
        if need or give:
            # the whole slice after all: both sides know from the same number.  The context must be free for the second sweep: the pass
            # queued behind this one (if any) is collected first, its outcome kept for the next call
            if need and self.open and not self.stash:
                self.stash.append(self._complete_local(True))
            self.fallbacks += 1 if need else 0
            buf = None
            if need:
                if self.full_halo is None:
                    self.full_halo = self.torch.empty_like(self.first_slice)
                buf = self.full_halo
            self._exchange(self.first_slice if give else None, self.lower if give else None, buf, self.upper if need else None)
            if need:
                self.torch.cuda.current_stream().synchronize() if buf.is_cuda else None
                ctx = self.ctx
                try:
                    ctx.drop_slice(self.t_halo)      # the masks-only slice (gone already if the pass before this one needed the slice too)
                except Exception as e:               # noqa: BLE001
                    if getattr(e, "code", None) != E_NOSLICE:
                        raise
                (ctx.push_scalar_slice if self.scalar else ctx.push_slice)(self.t_halo, buf)
                run_in = min([b.get("running_in", DBL_MAX)] + [float(v) for v in o["gathered"][:self.rank, 0]])
                recs, f, run = ctx.sweep_series(self.ts, self.scopes, run_in, copy=True)
                o.update(recs=recs, f=f, run=run, path=ctx.series_last_path())
                ctx.drop_slice(self.t_halo)          # (the next pass starts compact again)
        self.last_path = o["path"]
        self.last_asked = o["asked"]
        return o["recs"], o["f"], o["run"]


def _empty_records_dtype():
    from . import CP_DTYPE
    return CP_DTYPE


def gather_records(recs, dst=0, group=None):
    """The merge step of the t-slab partition (what critical_point_tracker.hh:689 does with `diy::mpi::gather` of
    the discrete points): every rank's hit records go to rank `dst`, which gets them as ONE array sorted by tag -- ready for
    ftkx_trace_curves, whose neighbourhoods cross slab boundaries like any other cell boundary.  Other ranks get None.
    Variable-length point-to-point transfers (a rank's hit count is data dependent); with 64-bit element tags the slabs are
    already in tag order (time is the slowest axis of the tag), the final sort only makes that independent of the tag mode."""
    import torch
    import torch.distributed as dist
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    recs = np.ascontiguousarray(recs)
    dev = "cuda" if dist.get_backend(group) == "nccl" else "cpu"
    mine = torch.tensor([len(recs)], dtype=torch.int64, device=dev)
    counts = [torch.zeros_like(mine) for _ in range(world)]
    dist.all_gather(counts, mine, group=group)
    counts = [int(c.item()) for c in counts]
    item = recs.dtype.itemsize
    if rank != dst:
        if len(recs):
            dist.send(torch.from_numpy(recs.view(np.uint8).reshape(-1).copy()).to(dev), dst, group=group)
        return None
    parts = []
    for r in range(world):
        if r == dst:
            parts.append(recs)
        elif counts[r]:
            buf = torch.empty((counts[r] * item,), dtype=torch.uint8, device=dev)
            dist.recv(buf, src=r, group=group)
            parts.append(buf.cpu().numpy().view(recs.dtype))
    merged = np.concatenate(parts) if parts else recs
    return merged[np.argsort(merged["tag"], kind="stable")]


def count_simplices(nd, dims, nt, scalar_input=True):
    """work items of the whole job: corners x (n_ord*nt + n_int*(nt-1))   (simplicial_regular_mesh.hh:1042)"""
    n_ord, n_int = (2, 10) if nd == 2 else (6, 54)
    corners = 1
    for d in dims:
        corners *= (d - 3) if scalar_input else (d - 2)
    return corners * (n_ord * nt + n_int * (nt - 1))
