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
    """One rank's DEVICE-DRIVEN pass over its timestep slab -- a thin caller of the C++ host, include/ftkx_slab.h (ftk_amd/csrc/slab.cpp):

        begin (masks, reduction, contribution, outgoing masks)  | all_gather of 4 doubles per rank; masks -> lower neighbour
        cull  (masks imported, factors, cull, request)          | request -> upper neighbour
        serve (patches around the neighbour's cells)            | reply -> lower neighbour
        finish (patches scattered, exact test, records)         | complete(): the ONE host wait of the pass

    The stages, the order of the messages, two passes in flight and the whole-slice recovery live THERE, once.  What this class adds is
    the transport.  Backend nccl: the library's own RCCL transport (ncclAllGather, grouped ncclSend / ncclRecv on the context's stream and a
    side stream; the communicator is made here from an id broadcast over torch.distributed).  Backend gloo (dry runs, tests): a table of
    two Python callbacks that stage the messages through host memory and torch.distributed.  A context stand-in with a `slab_backend()`
    method (tests/test_tslab.py: the oracle on the CPU) is driven through ftkx_slab_create_custom."""

    def __init__(self, ctx, nt, own, scalar_input, torch, device, first_slice=None, group=None):
        import ctypes as C
        import torch.distributed as dist
        from . import _lib
        self.ctx, self.nt, self.own, self.scalar, self.torch, self.dist, self.group = ctx, nt, list(own), scalar_input, torch, dist, group
        self.world, self.rank = dist.get_world_size(group), dist.get_rank(group)
        self.staged = dist.get_backend(group) != "nccl"
        self._L = L = _lib.load()
        self._C = C
        t0, t1 = slab_range(nt, self.world, self.rank)
        assert list(range(t0, t1)) == self.own, "the slab of this rank: slab_range(nt, world, rank)"
        self._err = None
        self._comm = None
        self._h = C.c_void_p()
        custom = hasattr(ctx, "slab_backend")
        if custom:
            self._backend = ctx.slab_backend()
            self._copy_in, self._copy_out = ctx.slab_upload, ctx.slab_download
        else:
            self._copy_in = lambda dst, src_np: _lib.check(L.ftkx_upload(ctx._h, dst, src_np.ctypes.data, src_np.nbytes), ctx._h)
            self._copy_out = lambda dst_np, src: _lib.check(L.ftkx_download(ctx._h, dst_np.ctypes.data, src, dst_np.nbytes), ctx._h)
        self.transport = "callbacks (torch.distributed, staged through host memory)"
        self._coll_dev = "cpu" if self.staged else device        # where torch.distributed wants its tensors (nccl: on the device)
        rc = None
        if not self.staged and not custom:
            # RCCL inside the library: a communicator of our own over the same ranks (torch does not hand its ncclComm_t out).  Every rank
            # learns whether EVERY rank got one (an all_reduce): either all of them take the library's transport, or all fall back
            idbuf = torch.zeros((128,), dtype=torch.uint8, device=device)
            if self.rank == 0:
                raw = (C.c_ubyte * 128)()
                _lib.check(L.ftkx_rccl_unique_id(raw))
                idbuf.copy_(torch.frombuffer(bytearray(raw), dtype=torch.uint8))
            dist.broadcast(idbuf, 0, group=group)
            raw = (C.c_ubyte * 128).from_buffer_copy(bytes(idbuf.cpu().numpy().tobytes()))
            comm = C.c_void_p()
            ok = L.ftkx_rccl_comm_create(raw, self.rank, self.world, int(device.index or 0), C.byref(comm)) == 0
            why = "" if ok else _lib.last_error(None)
            agreed = torch.tensor([1 if ok else 0], dtype=torch.int32, device=device)
            dist.all_reduce(agreed, op=dist.ReduceOp.MIN, group=group)
            if int(agreed.item()) == 1:
                self._comm = comm
                self.transport = "rccl (ftk_amd/csrc/slab_rccl.cpp: ncclAllGather, grouped ncclSend / ncclRecv), RCCL %d" % L.ftkx_rccl_version()
                rc = L.ftkx_slab_create_rccl(ctx._h, nt, self.rank, self.world, comm, None, C.byref(self._h))
            else:
                if ok:
                    L.ftkx_rccl_comm_destroy(comm)
                self.transport += "; the library's RCCL communicator could not be made on every rank (%s)" % why
        if rc is None:
            self._tr = _lib.SlabTransport(None, _lib.AG_FN(self._cb_all_gather), _lib.XCHG_FN(self._cb_exchange), 0, _lib.DESTROY_FN())
            rc = (L.ftkx_slab_create_custom(C.byref(self._backend), nt, self.rank, self.world, C.byref(self._tr), C.byref(self._h)) if custom
                  else L.ftkx_slab_create(ctx._h, nt, self.rank, self.world, C.byref(self._tr), C.byref(self._h)))
        if rc != 0:
            msg = _lib.last_error(None)
            if rc == E_UNSUPPORTED:
                raise RuntimeError("slab pass: " + msg)
            raise _lib.FtkxError(rc, msg)
        self._base = (0, 0, 0)
        self.last_path, self.last_asked = (0, 0), 0

    @classmethod
    def local(cls, ctx, nt, rank, world, hub):
        """a rank of ONE process (include/ftkx_slab.h: ftkx_slab_hub_*): peer copies between the ranks' devices, every rank driven by a thread
        of its own.  hub: the handle ftkx_slab_hub_create(world) gave."""
        import ctypes as C
        from . import _lib
        self = object.__new__(cls)
        self.ctx, self.nt, self.world, self.rank, self.group = ctx, nt, world, rank, None
        self._L, self._C, self._err, self._comm = _lib.load(), C, None, None
        self.transport = "hub (ranks of one process: peer copies ordered by events)"
        self.own = list(range(*slab_range(nt, world, rank)))
        self._h = C.c_void_p()
        _lib.check(self._L.ftkx_slab_create_local(ctx._h, nt, rank, hub, C.byref(self._h)))
        self._base = (0, 0, 0)
        self.last_path, self.last_asked = (0, 0), 0
        return self

    @classmethod
    def rccl(cls, ctx, nt, rank, world, comm, side_comm=None, periodic=False):
        """over a caller's ncclComm_t (ftkx_slab_create_rccl).  periodic: slice nt is slice 0 again (ftkx_slab_set_periodic)"""
        import ctypes as C
        from . import _lib
        self = object.__new__(cls)
        self.ctx, self.nt, self.world, self.rank, self.group = ctx, nt, world, rank, None
        self._L, self._C, self._err, self._comm = _lib.load(), C, None, None
        self.transport = "rccl (the caller's communicator)"
        self.own = list(range(*slab_range(nt, world, rank)))
        self._h = C.c_void_p()
        _lib.check(self._L.ftkx_slab_create_rccl(ctx._h, nt, rank, world, comm, side_comm, C.byref(self._h)))
        if periodic:
            _lib.check(self._L.ftkx_slab_set_periodic(self._h, 1))
        self._base = (0, 0, 0)
        self.last_path, self.last_asked = (0, 0), 0
        return self

    # ---- the host-staged transport (gloo): called back from ftkx_slab_submit / _complete ----
    def _guard(self, fn):
        try:
            fn()
            return 0
        except BaseException as e:      # noqa: BLE001  (an exception cannot cross the C frames: kept, re-raised by the caller of the C function)
            self._err = e
            return -2

    def _cb_all_gather(self, user, send, recv, nbytes, stream):
        def run():
            torch = self.torch
            mine = np.empty((nbytes,), dtype=np.uint8)
            self._copy_out(mine, send)
            out = torch.empty((nbytes * self.world,), dtype=torch.uint8, device=self._coll_dev)
            self.dist.all_gather_into_tensor(out, torch.from_numpy(mine).to(self._coll_dev), group=self.group)
            self._copy_in(recv, out.cpu().numpy())
        return self._guard(run)

    def _cb_exchange(self, user, send, sb, to, recv, rb, frm, stream):
        def run():
            torch, dist = self.torch, self.dist
            ops, back = [], None
            if to >= 0:
                out = np.empty((sb,), dtype=np.uint8)
                self._copy_out(out, send)
                ops.append(dist.P2POp(dist.isend, torch.from_numpy(out).to(self._coll_dev), to, self.group))
            if frm >= 0:
                back = torch.empty((rb,), dtype=torch.uint8, device=self._coll_dev)
                ops.append(dist.P2POp(dist.irecv, back, frm, self.group))
            for r in dist.batch_isend_irecv(ops):
                r.wait()
            if back is not None:
                if back.is_cuda:
                    self.torch.cuda.current_stream().synchronize()
                self._copy_in(recv, back.cpu().numpy())
        return self._guard(run)

    def _ck(self, rc):
        err, self._err = self._err, None
        if err is not None:
            raise err
        if rc != 0:
            from . import _lib
            raise _lib.FtkxError(rc, (self._L.ftkx_slab_last_error(self._h) or b"").decode(errors="replace"))

    def _info(self):
        from . import _lib
        i = _lib.SlabInfo()
        self._L.ftkx_slab_get_info(self._h, self._C.byref(i))
        return i

    # (bench.py sets these to 0 in front of its timed region)
    bytes_sent = property(lambda self: self._info().bytes_sent - self._base[0], lambda self, v: self._rebase(0, v))
    bytes_received = property(lambda self: self._info().bytes_received - self._base[1], lambda self, v: self._rebase(1, v))
    fallbacks = property(lambda self: self._info().fallbacks - self._base[2], lambda self, v: self._rebase(2, v))

    def _rebase(self, k, v):
        i = self._info()
        b = list(self._base)
        b[k] = (i.bytes_sent, i.bytes_received, i.fallbacks)[k] - int(v)
        self._base = tuple(b)

    def submit(self, running_resolution=None):
        C = self._C
        run = C.c_double(DBL_MAX if running_resolution is None else float(running_resolution))
        self._ck(self._L.ftkx_slab_submit(self._h, C.byref(run)))

    def complete(self, copy=True, push=None):
        """the oldest pass in flight -> (records, factors, running resolution)"""
        from . import _lib
        C = self._C
        run = C.c_double(0.0)
        n = len(self.own)
        f = np.zeros((max(1, n),), dtype=np.uint64)
        out, cnt = C.c_void_p(), C.c_size_t()
        self._ck(self._L.ftkx_slab_complete(self._h, C.byref(run), f.ctypes.data, C.byref(out), C.byref(cnt)))
        i = self._info()
        self.last_path, self.last_asked = (int(i.last_path), int(i.last_status)), int(i.last_asked)
        return _lib.records_from(out.value, cnt.value, copy), f[:n], run.value

    def gather_records(self, recs, root=0):
        """ftkx_slab_gather_records: every rank's records on `root`, sorted by tag (None elsewhere); over the slab's own transport"""
        from . import _lib
        C = self._C
        recs = np.ascontiguousarray(recs)
        out, cnt = C.c_void_p(), C.c_size_t()
        self._ck(self._L.ftkx_slab_gather_records(self._h, recs.ctypes.data if len(recs) else None, len(recs), int(root), C.byref(out), C.byref(cnt)))
        if self.rank != root:
            return None
        merged = _lib.records_from(out.value, cnt.value, True)
        self._L.ftkx_free(out)
        return merged

    def close(self):
        if self._h:
            self._L.ftkx_slab_destroy(self._h)
            self._h = self._C.c_void_p()
        if self._comm is not None:
            self._L.ftkx_rccl_comm_destroy(self._comm)
            self._comm = None

    def __del__(self):
        try:
            self.close()
        except Exception:   # noqa: BLE001
            pass


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
