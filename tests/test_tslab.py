"""Host logic of the t-slab partition; the world_size-2 tests run on CPU over gloo."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from ftk_amd import tslab


def test_slab_ranges_cover_time_exactly_once():
    for nt in (1, 2, 5, 16, 32, 33):
        for world in (1, 2, 3, 4, 8):
            seen = []
            for r in range(world):
                t0, t1 = tslab.slab_range(nt, world, r)
                assert 0 <= t0 <= t1 <= nt
                seen += list(range(t0, t1))
            assert seen == list(range(nt))
            for t in range(nt):
                t0, t1 = tslab.slab_range(nt, world, tslab.owner_of(t, nt, world))
                assert t0 <= t < t1


def test_factor_sequence_reproduces_reference(oracle):
    """the sticky running minimum must give the reference's per-step factors (fixtures come from the real reference)"""
    from common import load_golden
    for name in ("woven_31x37x32", "merger_2d_32x32x100", "woven_128x128x10", "random_2d_scalar_29x24x6"):
        g = load_golden(name)
        res = [oracle.resolution(oracle.gradient2D(s)) for s in g["steps"]]
        assert tslab.factors_from_resolutions(res) == [int(f) for f in g["factors"]], name
    assert tslab.scaling_factor(0.25) == oracle.scaling_factor(0.25)[0] == 256


def test_simplex_count_matches_baseline_table():
    assert tslab.count_simplices(2, (128, 128), 10) == 1718750
    assert tslab.count_simplices(2, (1024, 1024), 64) == 790170278
    assert tslab.count_simplices(3, (256, 256, 256), 16) == 16194277 * (6 * 16 + 54 * 15)
    assert tslab.count_simplices(2, (2048, 1024), 128, scalar_input=False) == 2091012 * (2 * 128 + 10 * 127)


def test_neighbours_of_a_series_that_is_periodic_in_time():
    """ftkx_slab_set_periodic (include/ftkx_slab.h): slice nt is slice 0 again -- the rank that owns the last timestep has the owner of
    timestep 0 for its upper neighbour, which has it for its lower one; with one rank both are the rank itself; ranks in between keep
    their neighbours; empty slabs have none.  Checked on the C++ host's own bookkeeping (no pass is queued: no GPU)."""
    import ctypes as C
    from ftk_amd import _lib
    L = _lib.load()
    keep = [_lib.BEGIN_FN(lambda *a: -1), _lib.CULL_FN(lambda *a: -1), _lib.CULL_FN(lambda *a: -1), _lib.FINISH_FN(lambda *a: -1), _lib.COMPLETE_FN(lambda *a: -1),
            _lib.STATUS_FN(lambda *a: -1), _lib.RECOVER_FN(lambda *a: -1), _lib.FIRST_FN(lambda *a: None), _lib.ALLOC_FN(lambda *a: None), _lib.RELEASE_FN(lambda *a: None),
            _lib.COPY_FN(lambda *a: -1), _lib.COPY_FN(lambda *a: -1), _lib.ABORT_FN(lambda u: None)]
    be = _lib.SlabBackend(None, *keep, 64, 4, 4, 64, None, 0)
    tr = _lib.SlabTransport(None, _lib.AG_FN(lambda *a: -1), _lib.XCHG_FN(lambda *a: -1), 0, _lib.DESTROY_FN(0))
    for nt, world in ((5, 1), (6, 2), (7, 3), (3, 5), (32, 8)):
        for periodic in (False, True):
            info = {}
            for r in range(world):
                h = C.c_void_p()
                _lib.check(L.ftkx_slab_create_custom(C.byref(be), nt, r, world, C.byref(tr), C.byref(h)))
                if periodic:
                    _lib.check(L.ftkx_slab_set_periodic(h, 1))
                i = _lib.SlabInfo()
                L.ftkx_slab_get_info(h, C.byref(i))
                info[r] = (i.t0, i.t1, i.lower, i.upper)
                L.ftkx_slab_destroy(h)
            owners = [r for r in range(world) if info[r][1] > info[r][0]]
            for k, r in enumerate(owners):
                t0, t1, lower, upper = info[r]
                want_lower = owners[k - 1] if k > 0 else (owners[-1] if periodic else -1)
                want_upper = owners[k + 1] if k + 1 < len(owners) else (owners[0] if periodic else -1)
                assert (lower, upper) == (want_lower, want_upper), (nt, world, periodic, r, info[r])
            for r in range(world):
                if r not in owners:
                    assert info[r][2:] == (-1, -1), (nt, world, periodic, r, info[r])


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _worker(rank, world, port, nt, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        t0, t1 = tslab.slab_range(nt, world, rank)
        # slice t is filled with the value t; resolution of slice t is 1/(t+2)
        slices = {t: torch.full((4, 5), float(t), dtype=torch.float64) for t in range(t0, t1)}
        buf = torch.full((4, 5), -1.0, dtype=torch.float64)
        got = tslab.exchange_halo(slices[t0] if t1 > t0 else buf, buf, nt)
        halo = float(buf[0, 0]) if got else None
        factors, res, mx = tslab.global_factors({t: 1.0 / (t + 2) for t in range(t0, t1)}, nt, local_max={t: 10.0 + t for t in range(t0, t1)})
        assert mx.tolist() == [10.0 + t for t in range(nt)]
        q.put((rank, t0, t1, halo, factors, res.tolist()))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("nt", [5, 8])
def test_halo_and_factor_exchange_world2_gloo(nt):
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, nt, q)) for r in range(world)]
    for p in procs:
        p.start()
    out = sorted(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    expect_res = [1.0 / (t + 2) for t in range(nt)]
    expect_f = tslab.factors_from_resolutions(expect_res)
    for rank, t0, t1, halo, factors, res in out:
        assert res == expect_res and factors == expect_f
        if t1 < nt:
            assert halo == float(t1)        # first slice of the next slab
        else:
            assert halo is None


# ---------------------------------------------------------------------------------------------------------------
# merge of the slabs' records -> pass 2 (critical_point_tracker.hh:689-717): every rank sweeps ITS timesteps (here with the
# oracle standing in for the GPU sweep: this is a CPU test of the host logic), receives its halo slice and the global factors
# exactly as bench.py does, the records are gathered on rank 0 and traced there.  The merged set must be the single-rank record
# set and the curves the reference's own traced curves (fixtures from the real reference), bit for bit.
# ---------------------------------------------------------------------------------------------------------------
def _merge_worker(rank, world, port, name, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import pyoracle as oracle
        import ftk_amd
        from common import load_golden
        g = load_golden(name)
        nd, nv, D, nt = g["nd"], g["nv"], g["dims"], g["DT"]
        scalar = nv == 1
        t0, t1 = tslab.slab_range(nt, world, rank)
        own = list(range(t0, t1))

        def derive(a):
            if scalar:
                V = oracle.gradient2D(a) if nd == 2 else oracle.gradient3D(a)
                return V, (oracle.jacobian2D(V, True) if nd == 2 else oracle.jacobian3D(V)), a
            return a, (oracle.jacobian2D(a, False) if nd == 2 else oracle.jacobian3D(a)), None

        slices = {t: torch.from_numpy(np.ascontiguousarray(g["steps"][t])) for t in own}       # this rank sees ONLY its slab ...
        buf = torch.empty_like(torch.from_numpy(np.ascontiguousarray(g["steps"][0])))
        got = tslab.exchange_halo(slices[t0] if own else buf, buf, nt)                          # ... plus the first slice of the next one
        if got:
            slices[t1] = buf
        fields = {t: derive(s.numpy()) for t, s in slices.items()}
        factors, _, _ = tslab.global_factors({t: oracle.resolution(fields[t][0]) for t in own}, nt, local_max={t: 1.0 for t in own})
        assert factors == [int(f) for f in g["factors"]]
        lo = 2 if scalar else 1
        dom = ([lo] * nd, [d - (3 if scalar else 2) for d in D])
        parts = []
        for t in own:
            for scope in (1, 2):
                if scope == 2 and t + 1 >= nt:
                    continue
                f0, f1 = fields[t], fields.get(t + 1)
                r = oracle.sweep(nd, scope, t, dom, dom, ([0] * nd, D), (f0[0], f1[0] if f1 else None), (f0[1], f1[1] if f1 else None),
                                 (f0[2], f1[2] if f1 else None) if scalar else None, factors[t], jacobian_symmetric=scalar, tag_mode=oracle.TAG_EXACT64)
                c = np.zeros(len(r), dtype=ftk_amd.CP_DTYPE)
                for fld in ("x", "t", "type", "tag"):
                    c[fld] = r[fld]
                c["scalar"] = r["scalar"]
                c["aux"] = (r["ordinal"].astype(np.uint32) & 1) | (np.uint32(t) << 1)
                parts.append(c)
        mine = np.concatenate(parts) if parts else np.zeros(0, dtype=ftk_amd.CP_DTYPE)
        merged = tslab.gather_records(mine, 0)
        if rank != 0:
            assert merged is None
            q.put((rank, len(mine), None))
            return
        ref = g["records"]
        order = np.argsort(ref["tag"], kind="stable")
        assert len(merged) == len(ref) and np.array_equal(merged["tag"], ref["tag"][order]) and np.array_equal(merged["type"], ref["type"][order])
        assert np.array_equal(merged["x"], ref["x"][order]) and np.array_equal(merged["t"], ref["t"][order])
        assert np.array_equal(merged["aux"] >> 1, ref["timestep"][order].astype(np.uint32))
        curves, loop, _ = ftk_amd.trace_curves(nd, dom, merged)
        gotc = sorted((tuple(merged["tag"][c].tolist()), int(l)) for c, l in zip(curves, loop))
        expc = sorted((tuple(tg.tolist()), int(l)) for l, tg in g["curves"])
        assert gotc == expc
        # curves that no single slab could have formed: points from more than one rank's timesteps
        crossing = sum(1 for c in curves if len(set(tslab.owner_of(int(ts), nt, world) for ts in (merged["aux"][c] >> 1))) > 1)
        trajs = ftk_amd.trace_and_post_process(nd, dom, merged)
        gotp = sorted((tuple(merged["tag"][i].tolist()), tuple(ty.tolist()), tuple(tt.tolist()), lp) for i, ty, tt, lp in trajs)
        expp = sorted((tuple(tg.tolist()), tuple(ty.tolist()), tuple(tt.tolist()), lp) for lp, tg, ty, tt in g["pp"])
        assert gotp == expp
        q.put((rank, len(mine), (len(merged), len(curves), crossing, len(trajs))))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("name,world,expect", [("woven_31x37x32", 2, (4491, 56, 56)), ("woven_31x37x32", 3, (4491, 56, 56)),
                                               ("moving_extremum_3d_21x21x21x32", 3, (126, 1, 1)), ("double_gyre_64x32x50", 2, (879, 2, 2)),
                                               ("random_3d_scalar_13x12x11x4", 5, None)])     # more ranks than timesteps: an empty slab
def test_merged_slabs_trace_to_the_reference_curves_gloo(oracle, name, world, expect):
    from ftk_amd import build
    build.build()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_merge_worker, args=(r, world, port, name, q)) for r in range(world)]
    for p in procs:
        p.start()
    out = sorted(q.get(timeout=300) for _ in range(world))
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    n_records, n_curves, crossing, n_trajs = out[0][2]
    if expect is None:      # (the worker has compared records, curves and trajectories with the fixture; here: the counts add up)
        from common import load_golden
        g = load_golden(name)
        assert world > g["DT"] and (n_records, n_curves, n_trajs) == (len(g["records"]), len(g["curves"]), len(g["pp"]))
        assert sum(o[1] for o in out) == n_records and sum(1 for o in out if o[1] == 0) >= world - g["DT"]
        return
    assert (n_records, n_curves, n_trajs) == expect
    assert sum(o[1] for o in out) == n_records and all(o[1] > 0 for o in out)     # every rank contributed
    assert crossing > 0                                                           # and curves do cross the slab boundaries


# ---------------------------------------------------------------------------------------------------------------
# tslab.SlabSeries -- the driver of the device-driven slab pass (ftkx_series_dist_*) -- over gloo on the CPU, with the ORACLE standing in
# for the context: the stages' contract (include/ftkx.h) restated on host tensors.  What is under test is the host logic: which
# collective is issued between which stages, by whom and to whom (ranks without a lower / upper neighbour, empty slabs), two passes
# in flight, and the whole-slice recovery both sides derive from the same number -- also when two passes in flight both need it.
# ---------------------------------------------------------------------------------------------------------------
class OracleSlabCtx:
    """include/ftkx.h: ftkx_series_dist_begin / _cull / _serve / _finish, ftkx_sweep_series_complete, ftkx_series_dist_status -- on the CPU.
    The "mask message" of this stand-in carries the slice itself; a request is 0 cells (everything needed came with the masks) or -1."""

    def __init__(self, oracle, ftk_amd, nd, nv, dims, nt, ask_full=()):
        self.o, self.f, self.nd, self.nv, self.D, self.nt = oracle, ftk_amd, nd, nv, dims, nt
        self.scalar = nv == 1
        self.slices, self.passes, self.done = {}, [], None
        self.ask_full = set(ask_full)           # pass numbers (0, 1, ...) in which this rank asks for the whole slice
        self.npass = 0
        self.nvals = int(np.prod(dims)) * (1 if self.scalar else nd)

    def _derive(self, a):
        o, nd = self.o, self.nd
        if self.scalar:
            V = o.gradient2D(a) if nd == 2 else o.gradient3D(a)
            return V, (o.jacobian2D(V, True) if nd == 2 else o.jacobian3D(V)), a
        return a, (o.jacobian2D(a, False) if nd == 2 else o.jacobian3D(a)), None

    def push_scalar_slice(self, t, a): self.slices[t] = np.array(a.numpy() if hasattr(a, "numpy") else a, dtype=np.float64).reshape(self._shape())
    push_slice = push_scalar_slice

    def _shape(self):
        return tuple(reversed(self.D)) + (() if self.scalar else (self.nd,))

    def drop_slice(self, t):
        if t not in self.slices:
            raise self.f.FtkxError(tslab.E_NOSLICE, "not resident")
        del self.slices[t]

    def packed_masks_bytes(self): return self.nvals * 8, 0
    def series_dist_cells(self): return 4
    def patch_doubles(self): return 1
    def series_last_path(self): return (1, 0)

    def series_dist_begin(self, ts, scopes, running, rank, world, upper, contrib, gathered, masks_out, side_stream=None):
        halo = upper is not None
        assert upper is None or upper == tslab.owner_of(int(ts[-1]) + 1, self.nt, world)      # (the rank that owns the next timestep: not rank + 1 where slabs are empty)
        assert side_stream is None
        self.cur = dict(ts=[int(t) for t in ts], scopes=[int(s) for s in scopes], run=tslab.DBL_MAX if running is None else running, rank=rank, world=world, halo=halo, gathered=gathered,
                        n=self.npass)
        self.npass += 1
        res = [self.o.resolution(self._derive(self.slices[t])[0]) for t in self.cur["ts"]]
        contrib[0], contrib[1], contrib[2], contrib[3] = min(res), 1.0, res[0], 1.0
        if masks_out is not None:
            masks_out.copy_(torch.from_numpy(self.slices[self.cur["ts"][0]].reshape(-1).view(np.uint8)))

    def series_dist_cull(self, masks_in, req_out):
        c = self.cur
        assert (masks_in is not None) == bool(c["halo"]) == (req_out is not None)
        c["asked"] = 0
        if masks_in is not None:
            c["halo_data"] = masks_in.numpy().view(np.float64).reshape(self._shape()).copy()
            c["asked"] = -1 if c["n"] in self.ask_full else 0
            req_out[0] = c["asked"]

    def series_dist_serve(self, req_in, reply_out):
        self.cur["served"] = int(req_in[0]) if req_in is not None else 0

    def series_dist_finish(self, reply_in):
        assert (reply_in is not None) == bool(self.cur["halo"])
        self.cur["G"] = self.cur["gathered"].numpy().reshape(-1, 4).copy()      # (what the all_gather put there: read behind it, as the cull stage does on the device)
        self.passes.append(self.cur)

    def _sweep(self, ts, scopes, run_in, halo_data):
        # (every pass of a test sweeps the same data: once is enough)
        key = (tuple(ts), tuple(scopes), run_in, None if halo_data is None else halo_data.tobytes(), self.slices[ts[-1] + 1].tobytes() if ts[-1] + 1 in self.slices else None)
        if not hasattr(self, "_memo"):
            self._memo = {}
        if key not in self._memo:
            self._memo[key] = self._sweep_now(ts, scopes, run_in, halo_data)
        return self._memo[key]

    def _sweep_now(self, ts, scopes, run_in, halo_data):
        o, nd = self.o, self.nd
        lo = 2 if self.scalar else 1
        dom = ([lo] * nd, [d - (3 if self.scalar else 2) for d in self.D])
        fields = {t: self._derive(self.slices[t]) for t in ts}
        if halo_data is not None:
            fields[ts[-1] + 1] = self._derive(halo_data)
        elif ts[-1] + 1 in self.slices:
            fields[ts[-1] + 1] = self._derive(self.slices[ts[-1] + 1])      # (the whole slice, pushed after a request of -1)
        run, parts, factors = run_in, [], []
        for t, sc in zip(ts, scopes):
            run = min([run] + [o.resolution(fields[u][0]) for u in (t, t + 1) if u in fields])
            factor = tslab.scaling_factor(run)
            factors.append(factor)
            for scope in (1, 2):
                if not (sc & scope):
                    continue
                f0, f1 = fields[t], fields.get(t + 1) if scope == 2 else None
                r = o.sweep(nd, scope, t, dom, dom, ([0] * nd, self.D), (f0[0], f1[0] if f1 else None), (f0[1], f1[1] if f1 else None),
                            (f0[2], f1[2] if f1 else None) if self.scalar else None, factor, jacobian_symmetric=self.scalar, tag_mode=o.TAG_EXACT64)
                c = np.zeros(len(r), dtype=self.f.CP_DTYPE)
                for fld in ("x", "t", "type", "tag", "scalar"):
                    c[fld] = r[fld]
                c["aux"] = (r["ordinal"].astype(np.uint32) & 1) | (np.uint32(t) << 1)
                parts.append(c)
        recs = np.concatenate(parts) if parts else np.zeros(0, dtype=self.f.CP_DTYPE)
        return recs[np.argsort(recs["tag"], kind="stable")], np.array(factors, dtype=np.uint64), run

    def sweep_series_complete(self, copy=True):
        c = self.passes.pop(0)
        self.done = c
        if c["asked"] < 0:
            raise self.f.FtkxError(tslab.E_NOSLICE, "the halo slice is needed as a whole")
        run_in = min([c["run"]] + [float(v) for v in c["G"][:c["rank"], 0]])
        return self._sweep(c["ts"], c["scopes"], run_in, c.get("halo_data"))

    def series_dist_status(self, world):
        return self.done["asked"], self.done["served"], self.done["G"]

    def sweep_series(self, ts, scopes, running, copy=True):
        assert not self.passes, "the context must be free for the second sweep"
        ts = [int(t) for t in ts]
        assert ts[-1] + 1 in self.slices, "the whole halo slice was pushed"
        recs, f, run = self._sweep(ts, [int(s) for s in scopes], running, None)
        return recs, f, run

    # ---- include/ftkx_slab.h: this stand-in as a ftkx_slab_backend -- the table of calls the C++ host (ftk_amd/csrc/slab.cpp) drives.  Buffers
    # are host memory handed out by `alloc`; a stage callback sees them as torch tensors over that memory. ----
    def slab_upload(self, dst, src_np):
        import ctypes as C
        C.memmove(dst, src_np.ctypes.data, src_np.nbytes)

    def slab_download(self, dst_np, src):
        import ctypes as C
        C.memmove(dst_np.ctypes.data, src, dst_np.nbytes)

    def slab_backend(self):
        import ctypes as C
        from ftk_amd import _lib
        self._bufs, self._cb_err = {}, None
        nb, cells, pd = self.nvals * 8, self.series_dist_cells(), self.patch_doubles()

        def view(ptr, n, dt):
            if not ptr:
                return None
            raw = (C.c_char * (n * np.dtype(dt).itemsize)).from_address(ptr)
            return torch.from_numpy(np.frombuffer(raw, dtype=dt))

        def guard(fn):
            try:
                r = fn()
                return 0 if r is None else r
            except self.f.FtkxError as e:
                return e.code
            except BaseException as e:      # noqa: BLE001
                import traceback
                traceback.print_exc()
                self._cb_err = e
                return -1

        def begin(u, ts, scopes, n, running, rank, nranks, upper, contrib, gathered, masks_out, side):
            return guard(lambda: self.series_dist_begin([ts[i] for i in range(n)], [scopes[i] for i in range(n)], running[0], rank, nranks, None if upper < 0 else upper,
                                                        view(contrib, 4, np.float64), view(gathered, 4 * nranks, np.float64), view(masks_out, nb, np.uint8), side))

        def cull(u, masks_in, req_out):
            return guard(lambda: self.series_dist_cull(view(masks_in, nb, np.uint8), view(req_out, 1 + cells, np.int64)))

        def serve(u, req_in, reply_out):
            return guard(lambda: self.series_dist_serve(view(req_in, 1 + cells, np.int64), view(reply_out, cells * pd, np.float64)))

        def finish(u, reply_in):
            return guard(lambda: self.series_dist_finish(view(reply_in, cells * pd, np.float64)))

        def hand_out(recs, f, run, running, factors, out, n_out):
            self._last = np.ascontiguousarray(recs)
            for i, v in enumerate(f):
                factors[i] = int(v)
            running[0] = float(run)
            out[0] = self._last.ctypes.data if len(self._last) else None
            n_out[0] = len(self._last)

        def complete(u, running, factors, out, n_out):
            return guard(lambda: hand_out(*self.sweep_series_complete(), running, factors, out, n_out))

        def status(u, asked, served, gathered, nranks, path, path_status):
            def run():
                a, s_, G = self.series_dist_status(nranks)
                asked[0], served[0] = a, s_
                for i, v in enumerate(np.asarray(G, dtype=np.float64).reshape(-1)):
                    gathered[i] = float(v)
                path[0], path_status[0] = 1, 0
            return guard(run)

        def recover(u, t_halo, full, ts, scopes, n, running, factors, out, n_out):
            def run():
                try:
                    self.drop_slice(t_halo)
                except self.f.FtkxError as e:
                    assert e.code == tslab.E_NOSLICE
                self.push_scalar_slice(t_halo, view(full, self.nvals, np.float64).numpy().copy())
                hand_out(*self.sweep_series([ts[i] for i in range(n)], [scopes[i] for i in range(n)], running[0]), running, factors, out, n_out)
                self.drop_slice(t_halo)
            return guard(run)

        def first_slice(u, t):
            self.slices[t] = np.ascontiguousarray(self.slices[t], dtype=np.float64)
            return self.slices[t].ctypes.data

        def alloc(u, n):
            buf = (C.c_char * max(int(n), 8))()
            self._bufs[C.addressof(buf)] = buf
            return C.addressof(buf)

        def release(u, p):
            self._bufs.pop(p, None)

        def copy(u, dst, src, n):
            C.memmove(dst, src, n)
            return 0

        self._fns = [_lib.BEGIN_FN(begin), _lib.CULL_FN(cull), _lib.CULL_FN(serve), _lib.FINISH_FN(finish), _lib.COMPLETE_FN(complete), _lib.STATUS_FN(status), _lib.RECOVER_FN(recover),
                     _lib.FIRST_FN(first_slice), _lib.ALLOC_FN(alloc), _lib.RELEASE_FN(release), _lib.COPY_FN(copy), _lib.COPY_FN(copy), _lib.ABORT_FN(lambda u: None)]
        return _lib.SlabBackend(None, *self._fns, nb, cells, pd, nb, None, 0)


def _slab_worker(rank, world, port, name, ask_full, pipelined, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import pyoracle as oracle
        import ftk_amd
        from common import load_golden
        g = load_golden(name)
        nd, nv, D, nt = g["nd"], g["nv"], g["dims"], g["DT"]
        t0, t1 = tslab.slab_range(nt, world, rank)
        own = list(range(t0, t1))
        ctx = OracleSlabCtx(oracle, ftk_amd, nd, nv, D, nt, ask_full=ask_full.get(rank, ()))
        for t in own:
            ctx.push_scalar_slice(t, np.ascontiguousarray(g["steps"][t]))
        first = torch.from_numpy(np.ascontiguousarray(g["steps"][t0], dtype=np.float64)) if own else None
        slab = tslab.SlabSeries(ctx, nt, own, nv == 1, torch, torch.device("cpu"), first_slice=first)
        outs = []
        npass = 3
        if pipelined:
            slab.submit()
            for i in range(1, npass + 1):
                if i < npass:
                    slab.submit()
                outs.append(slab.complete())
        else:
            for _ in range(npass):
                slab.submit()
                outs.append(slab.complete())
        for recs, f, run in outs[1:]:
            assert recs.tobytes() == outs[0][0].tobytes() and list(f) == list(outs[0][1])
        merged = slab.gather_records(np.array(outs[-1][0]), 0)          # (ftkx_slab_gather_records, over the same transport)
        also = tslab.gather_records(np.array(outs[-1][0]), 0)
        assert (merged is None) == (also is None) and (merged is None or merged.tobytes() == also.tobytes())
        assert ctx._cb_err is None
        if own:
            assert [int(v) for v in outs[-1][1]] == [int(g["factors"][t]) for t in own], (rank, outs[-1][1])
        if rank == 0:
            ref = g["records"]
            order = np.argsort(ref["tag"], kind="stable")
            assert len(merged) == len(ref) and np.array_equal(merged["tag"], ref["tag"][order]) and np.array_equal(merged["type"], ref["type"][order])
            assert np.array_equal(merged["x"], ref["x"][order]) and np.array_equal(merged["t"], ref["t"][order])
        q.put((rank, len(outs[-1][0]), slab.fallbacks, slab.bytes_sent, slab.bytes_received))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("name,world,ask_full,pipelined", [
    ("woven_31x37x32", 2, {}, True), ("woven_31x37x32", 3, {}, False),
    ("woven_31x37x32", 3, {0: (1,), 1: (0, 1, 2)}, True),          # whole-slice recovery: rank 0 in its second pass, rank 1 in every pass -- with two passes in flight
    ("moving_extremum_3d_21x21x21x32", 4, {2: (0,)}, False),
    ("random_3d_scalar_13x12x11x4", 5, {}, True),                  # more ranks than timesteps: an empty slab takes part in the all_gather only
])
def test_slab_series_protocol_gloo(oracle, name, world, ask_full, pipelined):
    from ftk_amd import build
    build.build()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_slab_worker, args=(r, world, port, name, ask_full, pipelined, q)) for r in range(world)]
    for p in procs:
        p.start()
    out = sorted(q.get(timeout=300) for _ in range(world))
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    for rank, nrec, fallbacks, sent, received in out:
        assert fallbacks == len(ask_full.get(rank, ())), (rank, fallbacks)
    assert sum(o[1] for o in out) > 0
