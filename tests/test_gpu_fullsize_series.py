"""The TIMED path at the TIMED sizes.  bench.py's `passes()` (bench.py: job) is `invalidate_masks` + `ftkx_sweep_series_submit`, two passes in
flight, `ftkx_sweep_series_complete`; its latency figure is `ftkx_sweep_series` on its own.  These tests run exactly that on BASELINE.json's
configurations C2 .. C5 at their full sizes and hold EVERY pass -- the first pipelined one, the last, the one on its own -- to

  * the oracle on the very arrays the GPU swept (C2 woven 1024^2 x 64, C5 double_gyre 2048 x 1024 x 128),
  * the analytic trajectory and an `exact_only` sweep of a 64^3 core around it (C3 256^3 x 16, C4 512^3 x 32),
  * record-for-record equality with the host-driven batch (ftkx_slices_prepare / _enqueue / _collect) on the same context,

and assert `ftkx_series_last_path`: a pass that silently fell back to the host-driven batch does not pass as the device-driven one.
Reference behaviour: critical_point_tracker_2d_regular.hh:263-433, critical_point_tracker_3d_regular.hh:150-308 (update_timestep)."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

SERIES_EARLY = 32          # csrc/sweep_params.hpp: the fused tail kernel finished the pass


@pytest.fixture(scope="module")
def gpu():
    import torch
    assert torch.cuda.is_available()
    import ftk_amd
    from ftk_amd import build
    build.build()
    return ftk_amd


def _bytes_equal(a, b):
    return len(a) == len(b) and np.ascontiguousarray(a).tobytes() == np.ascontiguousarray(b).tobytes()


class Resident:
    """a BASELINE configuration resident on the GPU, driven the way bench.py drives it"""

    def __init__(self, gpu, case, dims, nt, nv=1, keep_host=False, core=None, exact_only=False):
        import torch
        from ftk_amd import synthetic
        self.gpu, self.torch, self.nt, self.nd, self.nv = gpu, torch, nt, len(dims), nv
        scalar = nv == 1
        dev = torch.device("cuda", 0)
        lo = 2 if scalar else 1
        dom = ([lo] * self.nd, [d - (3 if scalar else 2) for d in dims])
        stream = torch.cuda.Stream(device=dev)          # a real stream shared with the library, as in bench.py
        torch.cuda.set_stream(stream)
        self.ctx = gpu.Context(self.nd)
        self.ctx.set_stream(stream.cuda_stream)
        self.ctx.set_mesh(dom, core or dom, ([0] * self.nd, list(dims)))
        self.ctx.set_options(jacobian_symmetric=scalar, derive_jacobian=1, tag_mode=gpu.TAG_EXACT64, exact_only=exact_only)
        self.keep, self.host = [], []
        for t in range(nt):
            a = synthetic.generate(case, dims, t, nt, torch, dev)
            torch.cuda.synchronize()
            self.keep.append(a)
            if keep_host:
                self.host.append(a.cpu().numpy())
            (self.ctx.push_scalar_slice if scalar else self.ctx.push_slice)(t, a)
        self.ts = np.arange(nt, dtype=np.int32)
        self.scopes = np.array([gpu.SCOPE_BOTH if t + 1 < nt else gpu.SCOPE_ORDINAL for t in range(nt)], dtype=np.int32)

    def pipelined(self, k):
        """bench.py's passes(k): -> [(records, factors, path)] of every pass"""
        ctx, out = self.ctx, []
        ctx.invalidate_masks()
        ctx.sweep_series_submit(self.ts, self.scopes)
        for i in range(1, k + 1):
            if i < k:
                ctx.invalidate_masks()
                ctx.sweep_series_submit(self.ts, self.scopes)
            recs, f, _ = ctx.sweep_series_complete(copy=True)
            out.append((recs, [int(v) for v in f], ctx.series_last_path()))
        return out

    def alone(self):
        """bench.py's one_pass(): ftkx_sweep_series with nothing else in flight"""
        self.ctx.invalidate_masks()
        recs, f, _ = self.ctx.sweep_series(self.ts, self.scopes, copy=True)
        return recs, [int(v) for v in f], self.ctx.series_last_path(), self.ctx.stats()

    def batch(self):
        """the host-driven batch on the same context: slices_prepare, factors on the host, enqueue, collect"""
        from ftk_amd import tslab
        self.ctx.invalidate_masks()
        rm = self.ctx.slices_prepare(range(self.nt), 0)
        factors = tslab.factors_from_resolutions([rm[t][0] for t in range(self.nt)])
        self.ctx.sweep_enqueue_many(self.ts, self.scopes, factors)
        return self.ctx.sweep_collect(copy=True), [int(f) for f in factors], self.ctx.stats()

    def close(self):
        self.ctx.close()
        self.keep = None
        self.torch.cuda.set_stream(self.torch.cuda.default_stream())
        self.torch.cuda.empty_cache()


def _timed_path(gpu, case, dims, nt, nv, want_path, keep_host=False, k=5):
    """-> (records, factors, host arrays, stats of the pass on its own) after every form of the pass has been held to every other"""
    R = Resident(gpu, case, dims, nt, nv, keep_host=keep_host)
    try:
        R.pipelined(3)                                   # bench.py's warm-up: both buffer sets, the copy stream, the record count known
        runs = R.pipelined(k)
        for i, (recs, f, path) in enumerate(runs):
            # (pipelined passes over sparse data whose mask kernel is long enough: the split pass, path 5 -- its tail next to the next mask kernel)
            assert path == want_path or path == (5, 0), (i, path, "a pass of the timed loop left the device-driven form")
            assert f == runs[0][1], (i, "factors differ between passes")
            assert _bytes_equal(recs, runs[0][0]), (i, len(recs), len(runs[0][0]), "pass %d differs from the first pipelined pass" % i)
        recs, f, path, st = R.alone()
        assert path == want_path, path
        assert f == runs[0][1] and _bytes_equal(recs, runs[0][0]), "ftkx_sweep_series on its own differs from the pipelined passes"
        b_recs, b_f, b_st = R.batch()
        assert b_f == f, (b_f[:4], f[:4])
        assert _bytes_equal(b_recs, recs), (len(b_recs), len(recs), "the host-driven batch differs")
        assert b_st["work_items"] == st["work_items"]
        # and back: the context that has just run the batch takes the device-driven form again
        again = R.pipelined(2)
        assert all((p == want_path or p == (5, 0)) and _bytes_equal(r, recs) for r, _, p in again), [p for _, _, p in again]
        return recs, f, R.host, st
    finally:
        R.close()


def _assert_equals_oracle(oracle, recs, host_steps, nd, nv, factors, what):
    from gpu_common import oracle_track_cached
    ref, rf, secs = oracle_track_cached(oracle, host_steps, nd, nv)      # (sorted by tag; shared with tests/test_gpu_fullsize.py)
    assert [int(f) for f in rf] == [int(f) for f in factors], what
    assert len(ref) == len(recs), (what, len(ref), len(recs))
    assert np.array_equal(ref["tag"], recs["tag"]) and np.array_equal(ref["type"], recs["type"]), what
    assert np.array_equal(ref["ordinal"].astype(np.uint32), recs["aux"] & 1) and np.array_equal(ref["timestep"].astype(np.uint32), recs["aux"] >> 1), what
    for f in ("x", "t"):
        assert np.array_equal(ref[f], recs[f]), (what, f)           # bit-identical (north_star asks for 1e-6)
    assert np.array_equal(ref["scalar"][:, 0], recs["scalar"][:, 0]), what


def _analytic_3d(gpu, recs, st, f, dims, nt):
    from ftk_amd import synthetic, tslab
    assert st["work_items"] == tslab.count_simplices(3, dims, nt) and st["cull_enabled"] == 1
    assert set(f) == {256}                                                             # dyadic parameters: nbits 8
    x0, dv = synthetic.moving_extremum_params(dims)
    assert len(recs) >= 2 * nt - 1 and set(recs["type"].tolist()) == {2}               # one minimum, every record a MIN
    for a in range(3):
        assert np.abs(recs["x"][:, a] - (x0[a] + dv[a] * recs["t"])).max() < 1e-6      # north_star tolerance (observed ~1e-13)
    assert np.all(np.diff(recs["tag"].astype(np.uint64)) > 0)                         # in tag order, unique
    ordinal = recs[(recs["aux"] & 1) == 1]
    assert np.array_equal(ordinal["t"], (ordinal["aux"] >> 1).astype(float))
    assert np.array_equal(np.unique(recs["aux"] >> 1), np.arange(nt))                  # the trajectory crosses every slice
    # every simplex of a 64^3 core around the path through the integer test: exactly the records of the culled, device-driven pass
    lo = [int(x0[a]) - 24 for a in range(3)]
    E = Resident(gpu, "moving_extremum_3d", dims, nt, core=(lo, [64, 64, 64]), exact_only=True)
    try:
        sub, sub_f, path, st_e = E.alone()
    finally:
        E.close()
    assert path[0] == 0                                 # (exact_only is the host-driven batch's: the tile kernel)
    assert st_e["cull_enabled"] == 0 and st_e["simplices_tested"] > 1000 * st["simplices_tested"]
    c = recs["x"]
    assert all(lo[a] <= c[:, a].min() and c[:, a].max() < lo[a] + 64 for a in range(3))
    assert sub_f == f and _bytes_equal(sub, recs)


def test_c2_series_woven_1024x1024x64(gpu, oracle):
    """BASELINE configs[1] through the timed path: hit-dense 2D, the bucket-ordering chain (path 1), records over the copy kernel"""
    from ftk_amd import tslab
    dims, nt = (1024, 1024), 64
    big_host = (os.cpu_count() or 1) >= 32
    recs, f, host, st = _timed_path(gpu, "woven", dims, nt, 1, (1, 0), keep_host=big_host)
    assert st["work_items"] == tslab.count_simplices(2, dims, nt) == 790170278 and st["cull_enabled"] == 1
    assert len(recs) > 50000 and set(recs["type"].tolist()) <= {1, 2, 4, 8}
    assert np.all(np.diff(recs["tag"].astype(np.uint64)) > 0)
    if big_host:
        _assert_equals_oracle(oracle, recs, host, 2, 1, f, "c2 series vs oracle")


def test_c5_series_double_gyre_2048x1024x128(gpu, oracle):
    """BASELINE configs[4] through the timed path: vector input, nbits 21"""
    from ftk_amd import tslab
    dims, nt = (2048, 1024), 128
    big_host = (os.cpu_count() or 1) >= 32
    recs, f, host, st = _timed_path(gpu, "double_gyre", dims, nt, 2, (1, 0), keep_host=big_host)
    assert st["work_items"] == tslab.count_simplices(2, dims, nt, scalar_input=False) == 3190884312 and st["cull_enabled"] == 1
    assert set(f) == {1 << 21}
    assert len(recs) > nt and set(recs["type"].tolist()) == {4}
    assert np.all(np.diff(recs["tag"].astype(np.uint64)) > 0)
    if big_host:
        _assert_equals_oracle(oracle, recs, host, 2, 2, f, "c5 series vs oracle")


def test_c3_series_moving_extremum_256cubed_x16(gpu):
    """BASELINE configs[2] through the timed path: sparse 3D, the fused tail (path 2) on its own, the split pass (path 5) with two in flight"""
    dims, nt = (256, 256, 256), 16
    recs, f, _, st = _timed_path(gpu, "moving_extremum_3d", dims, nt, 1, (2, SERIES_EARLY))
    _analytic_3d(gpu, recs, st, f, dims, nt)


def test_c4_series_moving_extremum_512cubed_x32(gpu):
    """BASELINE configs[3], the headline: 512^3 x 32 through the timed path (tail-chunked mask launch, fused tail)"""
    dims, nt = (512, 512, 512), 32
    recs, f, _, st = _timed_path(gpu, "moving_extremum_3d", dims, nt, 1, (2, SERIES_EARLY), k=4)
    assert st["work_items"] == 246073579314
    _analytic_3d(gpu, recs, st, f, dims, nt)


BIG_HOST = (os.cpu_count() or 1) >= 128        # the oracle at these sizes: minutes on a laptop, about a minute each on the GPU box's 256 host threads


def _bumpy_3d(torch, dev, dims, nt, seed):
    """a 3D series that is neither smooth nor rough everywhere: a tilted background (a strict sign almost everywhere), a few hundred moving
    bumps (isolated critical points, some of them merging), and a box of noise (hit-dense: every kind of mask word, many records per cell)"""
    g = torch.Generator(device="cpu").manual_seed(seed)
    DW, DH, DD = dims
    z, y, x = torch.meshgrid(torch.arange(DD, dtype=torch.float64, device=dev), torch.arange(DH, dtype=torch.float64, device=dev),
                             torch.arange(DW, dtype=torch.float64, device=dev), indexing="ij")
    nb = 240
    c0 = torch.rand((nb, 3), generator=g, dtype=torch.float64) * torch.tensor([DW, DH, DD], dtype=torch.float64)
    vel = (torch.rand((nb, 3), generator=g, dtype=torch.float64) - 0.5) * 1.5
    sig = 5.0 + 7.0 * torch.rand((nb,), generator=g, dtype=torch.float64)
    # (nbits is 21 on any smooth field with critical points: max |gradient| stays below 727041 / 2^21 = 0.347, or every vertex could overflow a
    # determinant and the masks would need the per-vertex rule -- the host-driven batch's)
    amp = (0.6 * torch.rand((nb,), generator=g, dtype=torch.float64) + 0.4) * torch.where(torch.rand((nb,), generator=g) < 0.5, -1.0, 1.0)
    noise = torch.randn((nt, 24, 70, 66), generator=g, dtype=torch.float64)
    out = []
    for k in range(nt):
        a = 0.021 * x + 0.017 * y + 0.013 * z
        for b in range(nb):
            cx, cy, cz = (c0[b] + vel[b] * k).tolist()
            s = float(sig[b])
            r = int(4 * s) + 1
            x0, x1 = max(0, int(cx) - r), min(DW, int(cx) + r + 1)
            y0, y1 = max(0, int(cy) - r), min(DH, int(cy) + r + 1)
            z0, z1 = max(0, int(cz) - r), min(DD, int(cz) + r + 1)
            if x0 >= x1 or y0 >= y1 or z0 >= z1:
                continue
            sub = (slice(z0, z1), slice(y0, y1), slice(x0, x1))
            a[sub] += float(amp[b]) * torch.exp(-((x[sub] - cx) ** 2 + (y[sub] - cy) ** 2 + (z[sub] - cz) ** 2) / (2 * s * s))
        a[13:37, 101:171, 97:163] += 0.04 * noise[k].to(dev)
        out.append(a.contiguous())
    return out


@pytest.mark.skipif(not BIG_HOST, reason="the oracle over 1.6e9 simplices of a hit-dense series needs the GPU box's host threads")
def test_bumpy_3d_series_vs_oracle(gpu, oracle):
    """A 3D series with partial tiles in x and y (517 x 515), several ZPlan pieces per tile column (50 planes) and every kind of mask block --
    uniform, stand-in, written -- through ftkx_sweep_series_submit / _complete and on its own, against the ORACLE: records bit-identical,
    factors equal."""
    import torch
    dims, nt = (517, 515, 50), 3
    dev = torch.device("cuda", 0)
    steps = _bumpy_3d(torch, dev, dims, nt, 77)
    torch.cuda.synchronize()
    stream = torch.cuda.Stream(device=dev)
    torch.cuda.set_stream(stream)
    try:
        ctx = gpu.Context(3)
        ctx.set_stream(stream.cuda_stream)
        dom = ([2] * 3, [d - 3 for d in dims])
        ctx.set_mesh(dom, dom, ([0] * 3, list(dims)))
        ctx.set_options(jacobian_symmetric=1, derive_jacobian=1, tag_mode=gpu.TAG_EXACT64)
        for t in range(nt):
            ctx.push_scalar_slice(t, steps[t])
        ts = np.arange(nt, dtype=np.int32)
        scopes = np.array([gpu.SCOPE_BOTH if t + 1 < nt else gpu.SCOPE_ORDINAL for t in range(nt)], dtype=np.int32)
        outs = []
        ctx.sweep_series(ts, scopes)                     # (a first pass sizes the survivor lists and the record buffers: bench.py's warm-up)
        ctx.invalidate_masks()
        ctx.sweep_series_submit(ts, scopes)
        for i in range(1, 4):
            if i < 3:
                ctx.invalidate_masks()
                ctx.sweep_series_submit(ts, scopes)
            recs, f, _ = ctx.sweep_series_complete(copy=True)
            outs.append((recs, [int(v) for v in f], ctx.series_last_path()))
        ctx.invalidate_masks()
        recs, f, _ = ctx.sweep_series(ts, scopes, copy=True)
        outs.append((recs, [int(v) for v in f], ctx.series_last_path()))
        ctx.close()
    finally:
        torch.cuda.set_stream(torch.cuda.default_stream())
    recs, f, path = outs[0]
    assert path[0] in (1, 2), path                       # device-driven
    assert len(recs) > 20000, len(recs)
    for r2, f2, p2 in outs[1:]:
        assert f2 == f and _bytes_equal(r2, recs), (p2, len(r2), len(recs))
    host = [a.cpu().numpy() for a in steps]
    _assert_equals_oracle(oracle, recs, host, 3, 1, f, "bumpy 3D 517 x 515 x 50 x 3 vs oracle")


def test_c4_series_eight_slabs_in_one_process(gpu):
    """BASELINE configs[3] in its literal form -- moving_extremum_3d 512^3 x 32 cut into EIGHT timestep slabs -- at full size: eight contexts
    on this one GPU, one per rank, each with its four slices, running the device-driven slab pass (ftkx_series_dist_*) with the messages
    handed from context to context as they would cross xGMI (tests/test_gpu_slab_inprocess.py).  The merged records are the one-context
    pass's, byte for byte; every rank took the fused tail; what crossed the slab boundaries was a few dozen cells' patches."""
    import torch
    from ftk_amd import synthetic
    import test_gpu_slab_inprocess as S
    dims, nt, world = (512, 512, 512), 32, 8
    dev = torch.device("cuda", 0)
    stream = torch.cuda.Stream(device=dev)
    torch.cuda.set_stream(stream)
    try:
        slices = []
        for t in range(nt):
            slices.append(synthetic.generate("moving_extremum_3d", dims, t, nt, torch, dev))
            torch.cuda.synchronize()
        one = S._make_ctx(gpu, 3, 1, dims, stream)
        for t in range(nt):
            one.push_scalar_slice(t, slices[t])
        scopes = [gpu.SCOPE_BOTH if t + 1 < nt else gpu.SCOPE_ORDINAL for t in range(nt)]
        want, wf, _ = one.sweep_series(range(nt), scopes, copy=True)
        assert one.series_last_path() == (2, SERIES_EARLY) and len(want) >= 2 * nt - 1
        one.close()

        class DeviceRank(S.Rank):            # (the slices are on the device already: adopt them instead of uploading numpy arrays)
            def __init__(self, rank):
                self._steps = None
                import numpy as np_
                from ftk_amd import tslab
                self.gpu, self.torch, self.rank, self.world, self.nt = gpu, torch, rank, world, nt
                t0, t1 = tslab.slab_range(nt, world, rank)
                self.own = list(range(t0, t1))
                self.scalar = True
                self.ctx = S._make_ctx(gpu, 3, 1, dims, stream)
                self.dev_slices = {t: slices[t] for t in self.own}
                for t in self.own:
                    self.ctx.push_scalar_slice(t, slices[t])
                self.open, self.stash = 0, []
                self.t_halo = t1 if t1 < nt else None
                self.lower = tslab.owner_of(t0 - 1, nt, world) if t0 > 0 else None
                self.upper = tslab.owner_of(t1, nt, world) if self.t_halo is not None else None
                self.ts = np_.array(self.own, dtype=np_.int32)
                self.scopes = np_.array([gpu.SCOPE_BOTH if t + 1 < nt else gpu.SCOPE_ORDINAL for t in self.own], dtype=np_.int32)
                nbytes, _ = self.ctx.packed_masks_bytes()
                cells, pd = self.ctx.series_dist_cells(), self.ctx.patch_doubles()
                z = lambda n, dt: torch.zeros((max(int(n), 1),), dtype=dt, device=dev)
                self.sets = [dict(contrib=z(4, torch.float64), gathered=z(4 * world, torch.float64), masks_out=z(nbytes, torch.uint8), masks_in=z(nbytes, torch.uint8),
                                  req_out=z(1 + cells, torch.int64), req_in=z(1 + cells, torch.int64), reply_out=z(cells * pd, torch.float64), reply_in=z(cells * pd, torch.float64))
                             for _ in range(2)]

        ranks = [DeviceRank(r) for r in range(world)]
        try:
            S._slab_pass(ranks, 0)
            S._slab_pass(ranks, 1)                   # two passes in flight
            first, rec1 = S._complete(ranks)
            second, rec2 = S._complete(ranks)
            assert rec1 == 0 and rec2 == 0           # nobody needed a whole slice
            for (r, a), (_, b) in zip(first, second):
                assert a[1] == b[1] and a[0].tobytes() == b[0].tobytes()
                assert r.ctx.series_last_path() == (2, SERIES_EARLY), (r.rank, r.ctx.series_last_path())
                asked, served, g = r.ctx.series_dist_status(world)
                assert 0 <= asked <= 512 and (asked > 0) == (r.upper is not None), (r.rank, asked)      # a few dozen cells around the extremum cross each boundary
            merged = np.concatenate([res[0] for _, res in first])
            merged = merged[np.argsort(merged["tag"], kind="stable")]
            got_f = [f for r, res in first for f in res[1]]
            assert got_f == [int(v) for v in wf]
            assert merged.tobytes() == np.ascontiguousarray(want).tobytes()
        finally:
            for r in ranks:
                r.ctx.close()
    finally:
        torch.cuda.set_stream(torch.cuda.default_stream())
        torch.cuda.empty_cache()
