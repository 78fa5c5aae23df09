"""The slab pass (include/ftkx.h: ftkx_series_dist_begin / _cull / _serve / _finish) with ALL ranks in ONE process: R contexts on one GPU,
one per rank of a series cut into timestep slabs, and the messages that would cross xGMI -- the gathered contributions, the halo's sign
masks, the request, the reply -- handed from context to context as they are.  No process group, no process guard: up to eight ranks, on
random fields (smooth, dyadic, rough, plateaus, tiny and huge values, NaN / Inf), 2D and 3D, scalar and vector input, sizes that are and are
not multiples of the tile sizes, slabs of one timestep, requests too small for the survivors (the whole-slice recovery), two passes in
flight.  The merged records and the per-step factors must be those of ONE context sweeping the whole series (ftkx_sweep_series), byte for
byte -- the reference's update_vector_field_scaling_factor (critical_point_tracker.hh:850-864) across slabs, and its interval sweep
(critical_point_tracker_3d_regular.hh:150-308) across a slab boundary."""
import os

import numpy as np
import pytest

from test_gpu_fuzz import _field, _vector_series

pytestmark = pytest.mark.gpu

DBL_MAX = float(np.finfo(np.float64).max)
E_NOSLICE, E_UNSUPPORTED = -4, -5


@pytest.fixture(scope="module")
def gpu():
    import torch
    assert torch.cuda.is_available()
    import ftk_amd
    from ftk_amd import build
    build.build()
    return ftk_amd


def _make_ctx(gpu, nd, nv, dims, stream):
    scalar = nv == 1
    lo = 2 if scalar else 1
    dom = ([lo] * nd, [d - (3 if scalar else 2) for d in dims])
    ctx = gpu.Context(nd)
    ctx.set_stream(stream.cuda_stream)
    ctx.set_mesh(dom, dom, ([0] * nd, list(dims)))
    ctx.set_options(jacobian_symmetric=int(scalar), derive_jacobian=1, tag_mode=gpu.TAG_EXACT64)
    return ctx


class Rank:
    def __init__(self, gpu, torch, dev, stream, nd, nv, dims, steps, nt, world, rank):
        from ftk_amd import tslab
        self.gpu, self.torch, self.rank, self.world, self.nt = gpu, torch, rank, world, nt
        t0, t1 = tslab.slab_range(nt, world, rank)
        self.own = list(range(t0, t1))
        self.scalar = nv == 1
        self.ctx = _make_ctx(gpu, nd, nv, dims, stream) if self.own else None
        self.dev_slices = {}
        for t in self.own:
            a = torch.from_numpy(np.ascontiguousarray(steps[t])).to(dev)
            self.dev_slices[t] = a
            (self.ctx.push_scalar_slice if self.scalar else self.ctx.push_slice)(t, a)
        self.open, self.stash = 0, []            # passes in flight; outcomes of passes collected early (a recovery needs the context free)
        self.t_halo = t1 if (self.own and t1 < nt) else None
        self.lower = tslab.owner_of(t0 - 1, nt, world) if self.own and t0 > 0 else None
        self.upper = tslab.owner_of(t1, nt, world) if self.t_halo is not None else None
        self.ts = np.array(self.own, dtype=np.int32)
        self.scopes = np.array([gpu.SCOPE_BOTH if t + 1 < nt else gpu.SCOPE_ORDINAL for t in self.own], dtype=np.int32)
        f64, i64, u8 = torch.float64, torch.int64, torch.uint8
        z = lambda n, dt: torch.zeros((max(int(n), 1),), dtype=dt, device=dev)
        self.sets = []
        if self.own:
            nbytes, _ = self.ctx.packed_masks_bytes()
            cells, pd = self.ctx.series_dist_cells(), self.ctx.patch_doubles()
            self.nbytes = nbytes
            for _ in range(2):
                self.sets.append(dict(contrib=z(4, f64), gathered=z(4 * world, f64), masks_out=z(nbytes, u8), masks_in=z(nbytes, u8), req_out=z(1 + cells, i64),
                                      req_in=z(1 + cells, i64), reply_out=z(cells * pd, f64), reply_in=z(cells * pd, f64)))
        else:
            for _ in range(2):
                self.sets.append(dict(contrib=torch.tensor([DBL_MAX, 0.0, DBL_MAX, 0.0], dtype=f64, device=dev), gathered=z(4 * world, f64)))


def _slab_pass(ranks, k, running=None):
    """one pass of every rank, stage by stage, the messages handed over in between (what tslab.SlabSeries.submit does over torch.distributed)"""
    torch = ranks[0].torch
    live = [r for r in ranks if r.own]
    for r in live:
        b = r.sets[k]
        r.ctx.series_dist_begin(r.ts, r.scopes, running, r.rank, r.world, r.upper, b["contrib"], b["gathered"], b["masks_out"] if r.lower is not None else None)
    allc = torch.cat([r.sets[k]["contrib"] for r in ranks])                # the all_gather
    for r in ranks:
        r.sets[k]["gathered"].copy_(allc)
    for r in live:
        if r.upper is not None:
            r.sets[k]["masks_in"].copy_(ranks[r.upper].sets[k]["masks_out"])
    for r in live:
        b = r.sets[k]
        r.ctx.series_dist_cull(b["masks_in"] if r.upper is not None else None, b["req_out"] if r.upper is not None else None)
    for r in live:
        if r.lower is not None:
            r.sets[k]["req_in"].copy_(ranks[r.lower].sets[k]["req_out"])
    for r in live:
        b = r.sets[k]
        r.ctx.series_dist_serve(b["req_in"] if r.lower is not None else None, b["reply_out"] if r.lower is not None else None)
    for r in live:
        if r.upper is not None:
            r.sets[k]["reply_in"].copy_(ranks[r.upper].sets[k]["reply_out"])
    for r in live:
        r.ctx.series_dist_finish(r.sets[k]["reply_in"] if r.upper is not None else None)
        r.open += 1


def _collect(r):
    """the oldest open pass of rank r -> (asked, served, gathered, (records, factors) or None)"""
    gpu = r.gpu
    if r.stash:
        return r.stash.pop(0)
    try:
        recs, f, run = r.ctx.sweep_series_complete(copy=True)
        res = (recs, [int(v) for v in f])
    except gpu.FtkxError as e:
        assert e.code == E_NOSLICE, e
        res = None
    r.open -= 1
    asked, served, g = r.ctx.series_dist_status(r.world)
    return asked, served, g, res


def _complete(ranks, running=None):
    """every rank collects its oldest pass; where a request said -1 the owner's first slice goes over as a whole and the asker sweeps again
    (with another pass still open behind it, that one is collected first and kept: the second sweep needs the context free)"""
    gpu = ranks[0].gpu
    out, recovered = [], 0
    status = {r.rank: _collect(r) for r in ranks if r.own}
    for r in ranks:
        if not r.own:
            continue
        asked, served, g, res = status[r.rank]
        if r.upper is not None:
            assert status[r.upper][1] == asked, "the owner was asked what the asker asked"
        if asked < 0:
            assert res is None
            recovered += 1
            while r.open > 0:
                r.stash.append(_collect(r))
            owner = ranks[r.upper]
            full = owner.dev_slices[owner.own[0]]
            try:
                r.ctx.drop_slice(r.t_halo)
            except gpu.FtkxError as e:
                assert e.code == E_NOSLICE
            (r.ctx.push_scalar_slice if r.scalar else r.ctx.push_slice)(r.t_halo, full)
            run_in = min([DBL_MAX if running is None else running] + [float(v) for v in g[:r.rank, 0]])
            recs, f, _ = r.ctx.sweep_series(r.ts, r.scopes, run_in, copy=True)
            res = (recs, [int(v) for v in f])
            r.ctx.drop_slice(r.t_halo)
        else:
            assert res is not None
        out.append((r, res))
    return out, recovered


def _one_case(gpu, rng, what, world, nd, nv, dims, nt, kind, pipelined, env_cells):
    import torch
    dev = torch.device("cuda", 0)
    sp = tuple(reversed(dims))
    steps = _field(rng, (nt,) + sp, kind) if nv == 1 else _vector_series(rng, nt, sp, kind)
    stream = torch.cuda.Stream(device=dev)
    torch.cuda.set_stream(stream)
    try:
        # the whole series on one context: what the slabs must add up to
        one = _make_ctx(gpu, nd, nv, dims, stream)
        for t in range(nt):
            (one.push_scalar_slice if nv == 1 else one.push_slice)(t, steps[t])
        scopes = [gpu.SCOPE_BOTH if t + 1 < nt else gpu.SCOPE_ORDINAL for t in range(nt)]
        want, wf, _ = one.sweep_series(range(nt), scopes, copy=True)
        wf = [int(v) for v in wf]
        path_one = one.series_last_path()
        one.close()
        ranks = [Rank(gpu, torch, dev, stream, nd, nv, dims, steps, nt, world, r) for r in range(world)]
        try:
            try:
                _slab_pass(ranks, 0)
            except gpu.FtkxError as e:
                assert e.code == E_UNSUPPORTED, (what, e)          # a mesh without summarised masks: every rank says so alike, at _begin
                for r in ranks:
                    if r.ctx is not None:
                        r.ctx.sweep_series_abort()
                return "unsupported", 0
            if pipelined:
                _slab_pass(ranks, 1)
            results, recovered = _complete(ranks)
            if pipelined:
                results2, rec2 = _complete(ranks)
                recovered += rec2
                for (r, a), (r2, b) in zip(results, results2):
                    assert a[1] == b[1] and a[0].tobytes() == b[0].tobytes(), (what, "the second pass in flight differs from the first")
            merged = np.concatenate([res[0] for _, res in results]) if results else np.zeros(0, dtype=want.dtype)
            merged = merged[np.argsort(merged["tag"], kind="stable")]
            got_f = {}
            for r, res in results:
                for t, f in zip(r.own, res[1]):
                    got_f[t] = f
            assert [got_f[t] for t in range(nt)] == wf, (what, [got_f[t] for t in range(nt)], wf)
            assert len(merged) == len(want), (what, len(merged), len(want), path_one)
            assert merged.tobytes() == np.ascontiguousarray(want).tobytes(), (what, "merged records differ")
            return "ok", recovered
        finally:
            for r in ranks:
                if r.ctx is not None:
                    r.ctx.close()
    finally:
        torch.cuda.set_stream(torch.cuda.default_stream())


KINDS = ["smooth", "dyadic", "rough", "plateau", "tiny", "huge", "spikes"]
TALLY = {"ok": 0, "unsupported": 0, "recovered": 0, "records": 0}


@pytest.mark.parametrize("seed", range(int(os.environ.get("FTKX_SLAB_FUZZ_SEEDS", "60"))))
def test_slabs_in_one_process_add_up_to_the_whole_series(gpu, seed, monkeypatch):
    rng = np.random.default_rng(4200 + seed)
    # (the request's capacity is read once per process: all seeds of one run share it -- small, so that recoveries happen)
    for case in range(4):
        nd = int(rng.choice([2, 3]))
        nv = int(rng.choice([1, 1, nd]))
        nt = int(rng.integers(2, 10))
        world = int(rng.choice([2, 3, 4, 8]))
        if nd == 2:
            dims = (int(rng.choice([16, 24, 40, 64, 136, 256])) + (int(rng.integers(0, 2)) if rng.random() < 0.2 else 0), int(rng.integers(9, 70)))
        else:
            dims = (int(rng.choice([8, 16, 24, 40, 128])) + (int(rng.integers(0, 2)) if rng.random() < 0.2 else 0), int(rng.integers(7, 36)), int(rng.integers(7, 20)))
        kind = str(rng.choice(KINDS))
        pipelined = bool(rng.random() < 0.5)
        what = f"seed {seed} case {case}: world {world} nd {nd} nv {nv} dims {dims} nt {nt} {kind} pipelined {pipelined}"
        verdict, recovered = _one_case(gpu, rng, what, world, nd, nv, dims, nt, kind, pipelined, None)
        TALLY[verdict] += 1
        TALLY["recovered"] += recovered


def test_the_in_process_slabs_took_every_way(gpu):
    """(runs behind the seeds) the device-driven form was what most cases ran, some meshes have no summarised masks (odd row lengths:
    FTKX_E_UNSUPPORTED at _begin, on every rank alike), and the whole-slice recovery happened (NaN / Inf and huge fields: masks the host rebuilds)"""
    assert TALLY["ok"] >= 150, TALLY
    assert TALLY["unsupported"] >= 1, TALLY
    assert TALLY["recovered"] >= 1, TALLY


def test_small_requests_force_the_whole_slice(gpu):
    """requests of at most two cells (FTKX_DIST_CELLS is read once per process: a child process): hit-dense data asks for the whole slice in
    every pass -- the same records"""
    import subprocess
    import sys
    code = (
        "import os, sys, numpy as np\n"
        "sys.path.insert(0, %r); sys.path.insert(0, %r); sys.path.insert(0, %r)\n"
        "import ftk_amd, test_gpu_slab_inprocess as T\n"
        "rng = np.random.default_rng(7)\n"
        "tot = 0\n"
        "for world, nd, nv, dims, nt, kind, pip in ((3, 2, 1, (64, 40), 6, 'rough', True), (2, 3, 1, (24, 20, 12), 5, 'smooth', False), (4, 2, 2, (40, 33), 7, 'smooth', True)):\n"
        "    v, rec = T._one_case(ftk_amd, rng, 'forced', world, nd, nv, dims, nt, kind, pip, None)\n"
        "    assert v == 'ok', v\n"
        "    tot += rec\n"
        "assert tot >= 3, tot\n"
        "print('recovered', tot)\n"
    ) % (os.path.dirname(os.path.dirname(os.path.abspath(__file__))), os.path.dirname(os.path.abspath(__file__)),
         os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600, env=dict(os.environ, FTKX_DIST_CELLS="2"))
    assert r.returncode == 0, r.stderr[-3000:]
    assert "recovered" in r.stdout


@pytest.mark.parametrize("dense_boundary", [False, True])
def test_sparse_slabs_turn_dense_with_two_passes_in_flight(gpu, dense_boundary):
    """ADVICE r04 (series.hip, series_complete): a slab pass queued as a SHORT chain -- the fused tail finished the pass before it, so nothing
    is queued behind the fused tail -- on data that has turned dense meanwhile, with another slab pass in flight behind it.  The fused tail
    declines, the rest of the chain cannot be queued (the counters are the next pass's), the host-driven batch sweeps the steps: it must
    start from the running minimum the LOWER ranks' contributions give, the halo's reduction must be recorded, ftkx_series_dist_status must
    answer for THIS pass, and a halo that is needed as a whole (dense_boundary: request -1) must be reported as such, not swept empty."""
    import torch
    dev = torch.device("cuda", 0)
    rng = np.random.default_rng(99 + int(dense_boundary))
    nd, nv, dims, nt, world = 2, 1, (256, 64), 6, 2
    sp = tuple(reversed(dims))
    # sparse: one moving bowl -- a gradient that is large everywhere but around its one zero (a handful of cells survive the cull)
    gy, gx = np.meshgrid(np.linspace(-1.0, 1.0, sp[0]), np.linspace(-1.0, 1.0, sp[1]), indexing="ij")
    # (its zero stays where it is: a zero that crosses a grid row between two slices flips the one sign bit whole 8 x 4 summary blocks have
    # in common, and the cells of those blocks -- more than a request of this size holds -- would ask for the slice as a whole)
    sparse = [np.ascontiguousarray((1.0 + 0.125 * k) * ((gx - 0.13) ** 2 + (gy + 0.07) ** 2)) for k in range(nt)]
    rough = _field(rng, (nt,) + sp, "rough")
    # tiny values in the LATER slab: its resolution must reach the factors of ... nobody before it, but the first slab's must reach the second's
    rough[0] = rough[0] * 2.0 ** -7
    dense = list(rough) if dense_boundary else [rough[0], rough[1], sparse[2], sparse[3], rough[4], rough[5]]
    stream = torch.cuda.Stream(device=dev)
    torch.cuda.set_stream(stream)
    try:
        scopes = [gpu.SCOPE_BOTH if t + 1 < nt else gpu.SCOPE_ORDINAL for t in range(nt)]
        one = _make_ctx(gpu, nd, nv, dims, stream)
        for t in range(nt):
            one.push_scalar_slice(t, dense[t])
        want, wf, _ = one.sweep_series(range(nt), scopes, copy=True)
        wf = [int(v) for v in wf]
        one.close()
        assert len(want) > 4096
        ranks = [Rank(gpu, torch, dev, stream, nd, nv, dims, sparse, nt, world, r) for r in range(world)]
        try:
            _slab_pass(ranks, 0)
            _, recovered = _complete(ranks)
            assert recovered == 0
            for r in ranks:
                assert r.ctx.series_last_path()[0] == 2, "the sparse pass was finished by the fused tail: the next one is queued as a short chain"
            for r in ranks:                                  # the data turns dense
                for t in r.own:
                    a = torch.from_numpy(np.ascontiguousarray(dense[t])).to(dev)
                    r.dev_slices[t] = a
                    r.ctx.push_scalar_slice(t, a)
            _slab_pass(ranks, 0)
            _slab_pass(ranks, 1)
            first, rec1 = _complete(ranks)
            paths = [r.ctx.series_last_path()[0] for r in ranks]
            second, rec2 = _complete(ranks)
            if dense_boundary:
                assert rec1 >= 1, "the halo was needed as a whole: reported by the declining fused tail, recovered by the caller"
            else:
                assert rec1 == 0 and 0 in paths, (rec1, paths, "a rank's short-chain pass fell back to the host-driven batch with its halo patched")
            for results in (first, second):
                merged = np.concatenate([res[0] for _, res in results])
                merged = merged[np.argsort(merged["tag"], kind="stable")]
                got_f = {}
                for r, res in results:
                    for t, f in zip(r.own, res[1]):
                        got_f[t] = f
                assert [got_f[t] for t in range(nt)] == wf, ([got_f[t] for t in range(nt)], wf)
                assert merged.tobytes() == np.ascontiguousarray(want).tobytes()
        finally:
            for r in ranks:
                if r.ctx is not None:
                    r.ctx.close()
    finally:
        torch.cuda.set_stream(torch.cuda.default_stream())
