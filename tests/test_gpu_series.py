"""ftkx_sweep_series -- the device-driven pass over a resident series (factors formed on the device, records ordered without a sort and
written into the pinned host buffer by the record kernel, one host wait) -- against the reference's fixtures, against the host-driven
batch, and through its own fallbacks."""
import os

import numpy as np
import pytest

from common import assert_records_equal, golden_names, load_golden

pytestmark = pytest.mark.gpu

DBL_MAX = float(np.finfo(np.float64).max)


@pytest.fixture(scope="module")
def gpu():
    import torch
    assert torch.cuda.is_available(), "these tests need the MI355X"
    import ftk_amd
    from ftk_amd import build
    build.build()
    return ftk_amd


def _ctx(gpu, g_or_dims, nd, nv, **opts):
    dims = g_or_dims
    scalar = nv == 1
    lo = 2 if scalar else 1
    dom = ([lo] * nd, [d - (3 if scalar else 2) for d in dims])
    ctx = gpu.Context(nd)
    ctx.set_mesh(dom, dom, ([0] * nd, list(dims)))
    o = dict(jacobian_symmetric=int(scalar), derive_jacobian=1, tag_mode=gpu.TAG_REFERENCE)
    o.update(opts)
    ctx.set_options(**o)
    return ctx


def _as_fixture(recs):
    out = np.zeros(len(recs), dtype=[("tag", "<u8"), ("type", "<u4"), ("ordinal", "<i4"), ("timestep", "<i4"), ("x", "<f8", (3,)), ("t", "<f8"), ("scalar", "<f8", (3,))])
    for f in ("tag", "type", "x", "t", "scalar"):
        out[f] = recs[f]
    out["ordinal"] = recs["aux"] & 1
    out["timestep"] = recs["aux"] >> 1
    return out


def _push_all(ctx, steps, nv, only=None):
    for t, a in enumerate(steps):
        if only is None or t in only:
            (ctx.push_scalar_slice if nv == 1 else ctx.push_slice)(t, a)


def _same(a, b):
    """record arrays identical byte for byte (NaN coordinates included)"""
    return len(a) == len(b) and np.ascontiguousarray(a).tobytes() == np.ascontiguousarray(b).tobytes()


PATHS = {}      # (path, status) -> fixtures that went that way


def _plain(g):
    return g["bounds"] is None and g["rectilinear"] is None and g["explicit"] is None


@pytest.fixture(autouse=True)
def _the_chain_not_the_one_launch_pass(monkeypatch):
    """The fixtures are small series: by default the ONE-LAUNCH pass (csrc/one_kernel.hip) would sweep them.  This module is about the kernel
    chain, the fused tail and their fallbacks -- FTKX_SERIES_HOOKS one=0 for its tests; the one-launch pass has its own tests at the end."""
    monkeypatch.setenv("FTKX_SERIES_HOOKS", "one=0")


@pytest.mark.parametrize("name", [n for n in golden_names()])
def test_series_pass_matches_reference_fixture(gpu, name):
    """the whole series in ONE call: records, their order, the per-step factors and the running resolution"""
    g = load_golden(name)
    if not _plain(g):
        pytest.skip("physical coordinates are set on the tracker")
    nd, nv, nt = g["nd"], g["nv"], g["DT"]
    opts = dict(robust=int(g["robust"]), compute_degrees=int(g["degrees"]))
    if g["type_filter"] is not None:
        opts.update(use_type_filter=1, type_filter=g["type_filter"])
    ctx = _ctx(gpu, g["dims"], nd, nv, **opts)
    _push_all(ctx, g["steps"], nv)
    scopes = [gpu.SCOPE_BOTH if t + 1 < nt else gpu.SCOPE_ORDINAL for t in range(nt)]
    recs, factors, run = ctx.sweep_series(range(nt), scopes)
    path, status = ctx.series_last_path()
    PATHS.setdefault((path, status), []).append(name)
    assert np.array_equal(factors, g["factors"]), f"{name}: factors {factors} vs {g['factors']} (path {path}, status {status})"
    assert np.all(recs["tag"][1:] >= recs["tag"][:-1]), f"{name}: records not in tag order (path {path}, status {status})"
    assert_records_equal(_as_fixture(recs), g["records"], coord_tol=0.0, what=f"{name} (path {path}, status {status})")
    # a second pass over the same (now masked and reduced) slices: nothing left to mask, same result
    recs2, factors2, run2 = ctx.sweep_series(range(nt), scopes)
    assert np.array_equal(factors2, factors) and run2 == run
    assert _same(recs2, recs)
    ctx.close()


def test_every_way_through_the_series_pass_was_taken(gpu):
    """(runs after the fixtures above) the fixtures exercised the kernel chain (path 1) -- taken up front and, where the fused tail found
    more than the 1024 records it orders, after that kernel had declined late (status bit 64) --, the fused tail kernel (path 2) and the
    hand-over to the host-driven batch, both decided up front (status 0) and raised by the kernels (masks that need the per-vertex
    overflow rule: bit 2)"""
    if sum(len(v) for v in PATHS.values()) < 20:
        pytest.skip("the fixture tests above did not run in this process")
    print({k: len(v) for k, v in PATHS.items()})
    assert any(p == 1 for p, _ in PATHS), PATHS
    assert any(p == 1 and not (st & 64) for p, st in PATHS), PATHS
    assert any(p == 1 and (st & 64) for p, st in PATHS), PATHS
    assert any(p == 2 for p, st in PATHS) and not any(p == 2 and (st & 64) for p, st in PATHS), PATHS
    assert any(p == 0 and st == 0 for p, st in PATHS) or any(p == 0 for p, _ in PATHS), PATHS
    assert any(p == 0 and (st & 2) for p, st in PATHS), PATHS


def test_the_device_driven_form_is_what_runs(gpu):
    """on the data the path was built for the call must not quietly take the host-driven batch"""
    taken = {}
    for name in ("woven_128x128x10", "woven_31x37x32", "double_gyre_64x32x50", "moving_extremum_3d_32x32x32x8_dyadic", "merger_2d_32x32x100"):
        g = load_golden(name)
        nt = g["DT"]
        ctx = _ctx(gpu, g["dims"], g["nd"], g["nv"], tag_mode=gpu.TAG_EXACT64)
        _push_all(ctx, g["steps"], g["nv"])
        scopes = [gpu.SCOPE_BOTH if t + 1 < nt else gpu.SCOPE_ORDINAL for t in range(nt)]
        recs, factors, _ = ctx.sweep_series(range(nt), scopes)
        taken[name] = ctx.series_last_path()
        st = ctx.stats()
        assert st["hits"] == len(recs) == len(g["records"]) and st["cull_enabled"] == 1
        assert np.array_equal(factors, g["factors"])
        ctx.close()
    assert all(p[0] in (1, 2) for p in taken.values()), taken


@pytest.mark.parametrize("name", ["woven_31x37x32", "merger_2d_32x32x100", "moving_extremum_3d_21x21x21x32", "double_gyre_64x32x50"])
def test_series_step_by_step_is_the_streaming_tracker(gpu, name):
    """one call per timestep with the running resolution handed on -- what update_timestep does -- gives the factor sequence and the
    records of the reference's streaming run; slice t of a step keeps the masks it got as slice t + 1 of the step before"""
    g = load_golden(name)
    nd, nv, nt = g["nd"], g["nv"], g["DT"]
    ctx = _ctx(gpu, g["dims"], nd, nv)
    run = DBL_MAX
    got, factors = [], []
    for k in range(nt):
        (ctx.push_scalar_slice if nv == 1 else ctx.push_slice)(k, g["steps"][k])
        if k == 0:
            continue
        r, f, run = ctx.sweep_series([k - 1], [gpu.SCOPE_BOTH], run)
        got.append(r); factors.append(int(f[0]))
        ctx.drop_slice(k - 1)
    r, f, run = ctx.sweep_series([nt - 1], [gpu.SCOPE_ORDINAL], run)
    got.append(r); factors.append(int(f[0]))
    assert factors == [int(v) for v in g["factors"]]
    assert_records_equal(_as_fixture(np.concatenate(got)), g["records"], coord_tol=0.0, what=name)
    ctx.close()


@pytest.mark.parametrize("seed", range(12))
def test_series_equals_the_host_driven_batch_on_random_fields(gpu, seed):
    """seeded random series (smooth / rough / plateaus / tiny / huge / NaN), 2D and 3D, scalar and vector input, every tag mode:
    ftkx_sweep_series == slices_prepare + host factors + enqueue + collect, record for record and in the same order"""
    from ftk_amd import tslab
    from test_gpu_fuzz import _field, _vector_series, KINDS
    rng = np.random.default_rng(7000 + seed)
    for case in range(5):
        nd = int(rng.choice([2, 3]))
        nv = int(rng.choice([1, nd]))
        nt = int(rng.integers(2, 7))
        dims = ((int(rng.choice([24, 40, 64, 136, 257])) + int(rng.integers(0, 2)), int(rng.integers(9, 90))) if nd == 2 else
                (int(rng.choice([16, 24, 40, 130])) + int(rng.integers(0, 2)), int(rng.integers(7, 36)), int(rng.integers(7, 24))))
        sp = tuple(reversed(dims))
        kind = str(rng.choice(KINDS))
        steps = _field(rng, (nt,) + sp, kind) if nv == 1 else _vector_series(rng, nt, sp, kind)
        tag_mode = int(rng.choice([gpu.TAG_REFERENCE, gpu.TAG_EXACT64]))
        what = f"seed {seed} case {case}: nd {nd} nv {nv} dims {dims} nt {nt} {kind} tag {tag_mode}"
        scopes = [gpu.SCOPE_BOTH if t + 1 < nt else gpu.SCOPE_ORDINAL for t in range(nt)]
        ctx = _ctx(gpu, dims, nd, nv, tag_mode=tag_mode)
        _push_all(ctx, steps, nv)
        rm = ctx.slices_prepare(range(nt), 0)
        ref_f = tslab.factors_from_resolutions([rm[t][0] for t in range(nt)])
        ctx.sweep_enqueue_many(range(nt), scopes, ref_f)
        ref = ctx.sweep_collect()
        ctx.invalidate_masks()
        got, f, _ = ctx.sweep_series(range(nt), scopes)
        path, status = ctx.series_last_path()
        assert [int(v) for v in f] == [int(v) for v in ref_f], what + f" factors (path {path} status {status})"
        assert _same(got, ref), what + f" (path {path} status {status}): {len(got)} vs {len(ref)} records"
        ctx.close()


def test_series_fallbacks_give_the_same_records(gpu, monkeypatch):
    """the ways out of the device-driven form -- buffers that are too small (tiny initial capacities cannot be forced from outside, so: a
    hit-dense series larger than the default buffers), buckets too full to rank on the device (FTKX_SERIES_HOOKS rank_max=0: the host orders
    every bucket), the form switched off (FTKX_SERIES=0) -- all return what the device-driven form returns"""
    g = load_golden("woven_128x128x10")
    nt = g["DT"]
    scopes = [gpu.SCOPE_BOTH if t + 1 < nt else gpu.SCOPE_ORDINAL for t in range(nt)]

    def run():
        ctx = _ctx(gpu, g["dims"], 2, 1, tag_mode=gpu.TAG_EXACT64)
        _push_all(ctx, g["steps"], 1)
        recs, f, _ = ctx.sweep_series(range(nt), scopes)
        p = ctx.series_last_path()
        ctx.close()
        return recs, f, p
    base, bf, bp = run()
    assert bp[0] in (1, 2)
    assert_records_equal(_as_fixture(base), g["records"], coord_tol=0.0, what="device-driven")
    monkeypatch.setenv("FTKX_SERIES_HOOKS", "one=0,rank_max=0")
    r, f, p = run()
    assert p[0] == 1 and (p[1] & 16), p                   # SERIES_FIX_ORDER raised, ordered on the host
    assert _same(r, base) and np.array_equal(f, bf)
    monkeypatch.setenv("FTKX_SERIES_HOOKS", "one=0")
    monkeypatch.setenv("FTKX_SERIES", "0")
    r, f, p = run()
    assert p[0] == 0
    assert _same(r, base) and np.array_equal(f, bf)


def test_short_chain_and_its_way_back(gpu, monkeypatch):
    """Sparse data: the fused tail kernel finishes a pass, and the NEXT pass is queued without the kernels behind it (refine, exact, scan,
    scatter, rank, records, finish).  When that pass then has more survivors than the fused tail takes, the kernel says so itself and the
    host queues the rest: sparse, dense, sparse, dense, sparse on one context -- every pass equals the same sweep on a fresh context with
    the short chain (and the factor job inside the cull kernel) switched off."""
    import torch
    from ftk_amd import synthetic
    dev = torch.device("cuda", 0)
    dims, nt = (96, 80), 6
    scopes = [gpu.SCOPE_BOTH if t + 1 < nt else gpu.SCOPE_ORDINAL for t in range(nt)]
    sparse = [synthetic.generate("moving_extremum_2d", dims, t, nt, torch, dev).cpu().numpy() for t in range(nt)]
    dense = [synthetic.generate("woven", dims, t, nt, torch, dev).cpu().numpy() * 1.0 for t in range(nt)]
    rng = np.random.default_rng(5)
    dense = [d + 0.3 * rng.standard_normal(d.shape) for d in dense]          # thousands of critical points
    series = [sparse, dense, sparse, dense, sparse]

    def sweep(ctx, steps):
        for t in list(range(nt)):
            ctx.push_scalar_slice(t, steps[t])
        recs, f, _ = ctx.sweep_series(range(nt), scopes)
        return recs.copy(), [int(v) for v in f], ctx.series_last_path()

    monkeypatch.setenv("FTKX_SERIES_HOOKS", "one=0,short=0,fold=0")
    want = []
    for steps in series:
        ctx = _ctx(gpu, dims, 2, 1, tag_mode=gpu.TAG_EXACT64)
        want.append(sweep(ctx, steps))
        ctx.close()
    monkeypatch.setenv("FTKX_SERIES_HOOKS", "one=0")
    ctx = _ctx(gpu, dims, 2, 1, tag_mode=gpu.TAG_EXACT64)
    paths = []
    for i, steps in enumerate(series):
        got = sweep(ctx, steps)
        paths.append(got[2][0])
        assert got[1] == want[i][1], (i, got[1], want[i][1])
        assert _same(got[0], want[i][0]), (i, len(got[0]), len(want[i][0]), got[2], want[i][2])
    ctx.close()
    assert paths[0] == 2 and paths[2] == 2 and paths[4] == 2, (paths, [w[2] for w in want])      # the fused tail did the sparse passes
    # ... and declined the dense ones, both times out of the short chain: the first outgrew the buffers on top (replayed by the host-driven
    # batch, which grows them), the second had its seven kernels queued behind the report
    assert paths[1] in (0, 1) and paths[3] == 1, (paths, [w[2] for w in want])
    assert len(want[1][0]) > 2000

def test_two_declined_short_chain_passes_completed_back_to_back(gpu, monkeypatch):
    """ADVICE r03 (series.hip): sparse data puts the context into the short chain; then TWO dense passes are submitted, both fused tails
    decline, and both are completed with no submit in between.  The first falls to the host-driven batch (a pass is open behind it), which
    takes the counters and survivor lists the second pass's cull left -- the second must not queue the rest of its chain on them."""
    import torch
    from ftk_amd import synthetic
    dev = torch.device("cuda", 0)
    dims, nt = (96, 80), 6
    scopes = [gpu.SCOPE_BOTH if t + 1 < nt else gpu.SCOPE_ORDINAL for t in range(nt)]
    sparse = [synthetic.generate("moving_extremum_2d", dims, t, nt, torch, dev).cpu().numpy() for t in range(nt)]
    rng = np.random.default_rng(5)
    dense_a = [synthetic.generate("woven", dims, t, nt, torch, dev).cpu().numpy() + 0.3 * rng.standard_normal(dims[::-1]) for t in range(nt)]
    dense_b = [d + 0.2 * rng.standard_normal(d.shape) for d in dense_a]

    def push(ctx, steps, t0=0):
        for t in range(nt):
            ctx.push_scalar_slice(t0 + t, steps[t])

    monkeypatch.setenv("FTKX_SERIES_HOOKS", "one=0,short=0")
    want = []
    for steps in (dense_a, dense_b):
        ctx = _ctx(gpu, dims, 2, 1, tag_mode=gpu.TAG_EXACT64)
        push(ctx, steps)
        r, f, _ = ctx.sweep_series(range(nt), scopes)
        want.append((r.copy(), [int(v) for v in f]))
        ctx.close()
    monkeypatch.setenv("FTKX_SERIES_HOOKS", "one=0")
    ctx = _ctx(gpu, dims, 2, 1, tag_mode=gpu.TAG_EXACT64)
    push(ctx, sparse)
    ctx.sweep_series(range(nt), scopes)
    assert ctx.series_last_path()[0] == 2                       # the fused tail finished it: the next passes go out as short chains
    # the two dense series live side by side at timesteps 100.. and 200.. (tags carry the timestep: compare everything else)
    push(ctx, dense_a, 100); push(ctx, dense_b, 200)
    ctx.sweep_series_submit(np.arange(100, 100 + nt), scopes)
    ctx.sweep_series_submit(np.arange(200, 200 + nt), scopes)
    got = []
    for _ in range(2):
        r, f, _ = ctx.sweep_series_complete()
        got.append((r.copy(), [int(v) for v in f], ctx.series_last_path()))
    ctx.close()
    for i, t0 in enumerate((100, 200)):
        g, w = got[i][0], want[i][0]
        assert got[i][1] == want[i][1], (i, got[i][1], want[i][1])
        assert len(g) == len(w) and len(w) > 2000, (i, len(g), len(w), got[i][2])
        assert np.array_equal(g["type"], w["type"]) and np.array_equal(g["x"][:, :2], w["x"][:, :2]) and np.allclose(g["t"], w["t"] + t0, rtol=0, atol=1e-9), (i, got[i][2])
        assert np.array_equal(g["scalar"], w["scalar"]) and np.array_equal(g["aux"] & 1, w["aux"] & 1) and np.array_equal(g["aux"] >> 1, (w["aux"] >> 1) + t0)


def test_abort_discards_the_open_passes_and_leaves_the_context_usable(gpu):
    """ftkx_sweep_series_abort: two passes submitted, none completed, aborted -- the context then sweeps as if nothing had
    been queued (the masks the discarded passes were building are rebuilt), other sweep calls are let through again, and aborting with
    nothing open is a no-op.  The streaming tracker uses it after a failed completion."""
    import torch
    from ftk_amd import synthetic
    dev = torch.device("cuda", 0)
    dims, nt = (96, 80), 6
    scopes = [gpu.SCOPE_BOTH if t + 1 < nt else gpu.SCOPE_ORDINAL for t in range(nt)]
    steps = [synthetic.generate("woven", dims, t, nt, torch, dev).cpu().numpy() for t in range(nt)]
    ref = _ctx(gpu, dims, 2, 1, tag_mode=gpu.TAG_EXACT64)
    _push_all(ref, steps, 1)
    want, wf, _ = ref.sweep_series(range(nt), scopes)
    want = want.copy()
    ref.close()
    ctx = _ctx(gpu, dims, 2, 1, tag_mode=gpu.TAG_EXACT64)
    _push_all(ctx, steps, 1)
    ctx.sweep_series_abort()                                    # nothing open: fine
    ctx.sweep_series_submit(range(nt), scopes)
    ctx.sweep_series_submit(range(nt), scopes)
    with pytest.raises(gpu.FtkxError):
        ctx.slices_prepare(range(nt), 0)                        # passes open: refused
    ctx.sweep_series_abort()
    rm = ctx.slices_prepare(range(nt), 0)                       # let through again
    assert len(rm) == nt
    got, f, _ = ctx.sweep_series(range(nt), scopes)
    assert _same(got, want) and [int(v) for v in f] == [int(v) for v in wf]
    ctx.sweep_series_submit(range(nt), scopes)                  # and the two-in-flight form works as before
    ctx.sweep_series_submit(range(nt), scopes)
    for _ in range(2):
        got, f, _ = ctx.sweep_series_complete()
        assert _same(got, want)
    with pytest.raises((gpu.FtkxError, IndexError)):
        ctx.sweep_series_complete()                             # nothing left
    ctx.close()



@pytest.mark.parametrize("name", ["woven_128x128x10", "woven_31x37x32", "double_gyre_64x32x50", "merger_2d_32x32x100", "moving_extremum_3d_21x21x21x32",
                                  "adversarial_3d_scalar_9x9x9x4", "random_2d_scalar_29x24x6_saddles", "adversarial_3d_scalar_9x9x9x3_norobust"])
@pytest.mark.parametrize("split,coarse_out", [(False, False), (True, False), (True, True)], ids=["in_order", "split", "split_coarse_grained_records"])
def test_pipelined_passes_equal_the_plain_ones(gpu, name, split, coarse_out, monkeypatch):
    """ftkx_sweep_series_submit / _complete, two passes open at a time: (a) the whole series swept again and again, masks dropped in
    between (what bench.py times), (b) the series in consecutive pieces, each continuing on the device from the running minimum of the one
    before it, while the host has not even seen that one yet (a streaming caller).  Records, factors and running minima are those of
    ftkx_sweep_series on the same steps; the copy engine carries the records of the later passes (more than 4096 of them, where the
    fixture has that many).
    split: FTKX_SERIES_HOOKS split=2 -- the SPLIT pass whatever the size of the mask launch (by default: 2 GB and more): where the pass before
    was sparse, the tail of a pass (the kernel chain, its record kernel held to a third of a SIMD's registers) runs on a stream of its own
    next to the begin and mask kernels of the pass queued behind; counters zeroed there, mask arrays that the next pass rebuilds replaced."""
    if split:
        monkeypatch.setenv("FTKX_SERIES_HOOKS", "one=0,split=2")
    # the record buffer: coherent host memory by default (the HIP memory model's guarantee for records read behind a flag); the switch back
    # to coarse-grained memory (rounds 3-5, gfx950 behaviour) stays covered
    if coarse_out:
        monkeypatch.setenv("FTKX_SERIES_OUT_COHERENT", "0")
    g = load_golden(name)
    nd, nv, nt = g["nd"], g["nv"], g["DT"]
    scopes = [gpu.SCOPE_BOTH if t + 1 < nt else gpu.SCOPE_ORDINAL for t in range(nt)]
    opts = dict(tag_mode=gpu.TAG_EXACT64, robust=int(g["robust"]), compute_degrees=int(g["degrees"]))
    if g["type_filter"] is not None:      # (a type filter, non-robust 3D: not the device-driven form -- the passes are swept by the host-driven batch when collected)
        opts.update(use_type_filter=1, type_filter=g["type_filter"])
    ctx = _ctx(gpu, g["dims"], nd, nv, **opts)
    _push_all(ctx, g["steps"], nv)
    want, wf, wrun = ctx.sweep_series(range(nt), scopes)
    # (a)
    ctx.invalidate_masks()
    ctx.sweep_series_submit(range(nt), scopes)
    for it in range(4):
        ctx.invalidate_masks()
        ctx.sweep_series_submit(range(nt), scopes)
        got, f, run = ctx.sweep_series_complete()
        assert _same(got, want) and np.array_equal(f, wf) and run == wrun, (name, it, len(got), len(want))
    got, f, run = ctx.sweep_series_complete()
    assert _same(got, want) and np.array_equal(f, wf) and run == wrun
    if split and name == "moving_extremum_3d_21x21x21x32":
        assert ctx.series_last_path()[0] == 5, (name, ctx.series_last_path())      # (sparse fixtures: the passes after the first were split)
    with pytest.raises(Exception):
        ctx.sweep_series_complete()
    # (b) pieces of the series; the reference: the same pieces through ftkx_sweep_series, the running minimum carried by the caller
    pieces = [list(range(a, min(a + 3, nt))) for a in range(0, nt, 3)]
    ctx.invalidate_masks()
    ref, run = [], None
    for p in pieces:
        sc = [gpu.SCOPE_BOTH if t + 1 < nt else gpu.SCOPE_ORDINAL for t in p]
        r, f, run = ctx.sweep_series(p, sc, running_resolution=run)
        ref.append((r, [int(v) for v in f], run))
    assert sum(len(r[0]) for r in ref) == len(want)
    ctx.invalidate_masks()
    done = []
    for i, p in enumerate(pieces):
        sc = [gpu.SCOPE_BOTH if t + 1 < nt else gpu.SCOPE_ORDINAL for t in p]
        ctx.sweep_series_submit(p, sc, chain=i > 0)
        if i > 0:
            done.append(ctx.sweep_series_complete())
    done.append(ctx.sweep_series_complete())
    for i, (r, f, run) in enumerate(done):
        # (the running minimum is as exact as the pass's hint makes it -- components at or above 1 / hint cannot change nbits and are not looked at --
        # and a chained pass takes its hint from what the HOST knew when the chain began: the same factor, a value that may be more exact)
        from ftk_amd import tslab
        assert _same(r, ref[i][0]) and [int(v) for v in f] == ref[i][1], (name, i, len(r), len(ref[i][0]))
        assert run <= ref[i][2] and tslab.scaling_factor(run) == tslab.scaling_factor(ref[i][2]), (name, i, run, ref[i][2])
    # the plain call still works, other sweeps are refused while a pass is open
    ctx.sweep_series_submit(range(nt), scopes)
    with pytest.raises(Exception):
        ctx.slices_prepare(range(nt), 0)
    got, f, run = ctx.sweep_series_complete()
    assert _same(got, want)
    ctx.close()


@pytest.mark.parametrize("each_step", [True, False], ids=["factor_read_each_step", "never_looked_at"])
@pytest.mark.parametrize("name", ["woven_128x128x10", "woven_31x37x32", "double_gyre_64x32x50", "merger_2d_32x32x100", "moving_extremum_3d_21x21x21x32",
                                  "adversarial_3d_scalar_9x9x9x4", "adversarial_3d_vector_8x8x8x3"])
def test_tracker_with_deferred_collection(gpu, name, each_step):
    """critical_point_tracker_regular with set_deferred_collection(True): every step's sweep is queued -- continuing on the device from the
    running minimum of the step before it, whose records the host has not seen yet, and masking each snapshot once -- before the step before
    it is collected.  Same records and factors as the fixture, whether the caller looks at the tracker after every step (which collects
    what is out) or only at the end; device-resident snapshots."""
    from gpu_common import run_tracker
    g = load_golden(name)
    if not _plain(g):
        pytest.skip("physical coordinates are set on the tracker")
    out = run_tracker(g["steps"], g["nd"], g["nv"], robust=g["robust"], type_filter=g["type_filter"], compute_degrees=g["degrees"],
                      device=True, deferred=True, factor_each_step=each_step)
    recs, factors = out[0], out[1]
    if each_step:
        assert np.array_equal(np.asarray(factors, dtype=np.uint64), g["factors"]), (factors, g["factors"])
    else:
        assert int(factors[-1]) == int(g["factors"][-1])
    assert_records_equal(recs, g["records"], coord_tol=0.0, what=f"{name} deferred")


@pytest.mark.parametrize("depth", [2, 3, 5, -3])
@pytest.mark.parametrize("name", ["woven_128x128x10", "woven_31x37x32", "double_gyre_64x32x50", "merger_2d_32x32x100", "moving_extremum_3d_21x21x21x32",
                                  "adversarial_3d_scalar_9x9x9x4", "adversarial_3d_vector_8x8x8x3"])
def test_tracker_with_deferred_batches(gpu, name, depth, monkeypatch):
    """set_deferred_collection(True, depth): the sweeps of `depth` consecutive update_timestep() calls go out as ONE device-driven pass (the
    snapshots popped meanwhile stay resident until it has been queued), two such passes in flight; a series whose length is not a multiple
    of the depth ends in a partial batch.  Same records (each with the timestep of ITS step) and the same final factor as the fixture."""
    from gpu_common import run_tracker
    if depth < 0:      # batches of 3 as SPLIT passes (FTKX_SERIES_HOOKS split=2: whatever their size): the snapshots the tracker pops while a batch's tail
        depth = -depth                       # still runs on its own stream are parked with that pass, not recycled under it (free_slice)
        monkeypatch.setenv("FTKX_SERIES_HOOKS", "one=0,split=2")
    g = load_golden(name)
    if not _plain(g):
        pytest.skip("physical coordinates are set on the tracker")
    out = run_tracker(g["steps"], g["nd"], g["nv"], robust=g["robust"], type_filter=g["type_filter"], compute_degrees=g["degrees"],
                      device=True, deferred=True, depth=depth, factor_each_step=False)
    recs, factors = out[0], out[1]
    assert int(factors[-1]) == int(g["factors"][-1])
    assert_records_equal(recs, g["records"], coord_tol=0.0, what=f"{name} deferred in batches of {depth}")


def test_tracker_with_deferred_batches_looked_at_midway(gpu):
    """a getter between the steps of a batch submits the partial batch and collects what is out: the factor sequence read every OTHER step is the fixture's"""
    from gpu_common import run_tracker
    import ftk_amd, torch
    g = load_golden("double_gyre_64x32x50")
    tr = ftk_amd.CriticalPointTracker2DRegular()
    D = [g["steps"][0].shape[1], g["steps"][0].shape[0]]
    tr.set_scalar_field_source(ftk_amd.SOURCE_NONE); tr.set_vector_field_source(ftk_amd.SOURCE_GIVEN)
    tr.set_jacobian_field_source(ftk_amd.SOURCE_DERIVED); tr.set_jacobian_symmetric(False)
    tr.set_domain([1, 1], [d - 2 for d in D]); tr.set_array_domain([0, 0], D)
    tr.initialize()
    tr.set_deferred_collection(True, 4)
    seen = {}
    for k, a in enumerate(g["steps"]):
        tr.push_vector_field_snapshot(torch.from_numpy(np.ascontiguousarray(a)).cuda())
        if k:
            tr.advance_timestep()
            if k % 2 == 0:
                seen[k - 1] = tr.get_vector_field_scaling_factor()
    tr.update_timestep()
    seen[len(g["steps"]) - 1] = tr.get_vector_field_scaling_factor()
    recs, o, ts = tr.get_critical_points()
    tr.close()
    for k, f in seen.items():
        assert int(f) == int(g["factors"][k]), (k, f, g["factors"][k])
    assert len(recs) == len(g["records"]) and np.array_equal(np.sort(recs["tag"]), np.sort(g["records"]["tag"]))
    assert np.array_equal(ts[np.argsort(recs["tag"], kind="stable")], g["records"]["timestep"][np.argsort(g["records"]["tag"], kind="stable")])


@pytest.mark.parametrize("hooks,coarse_out", [("one=0", False), ("one=0,split=2", False), ("one=0,split=2", True)], ids=["in_order", "split", "split_coarse_grained_records"])
@pytest.mark.parametrize("name", ["moving_extremum_3d_21x21x21x32", "woven_128x128x10", "double_gyre_64x32x50"])
def test_three_passes_in_flight(gpu, name, hooks, coarse_out, monkeypatch):
    """three passes open at a time (the third lets the host run one pass ahead of a split pass's tail; split passes alternate between two sets
    of counters and lists, on two tail streams): the whole series again and again, masks dropped in between, and a chain of pieces that
    each continue on the device from the one before -- records, factors and running minima of ftkx_sweep_series; a fourth submit is refused"""
    monkeypatch.setenv("FTKX_SERIES_HOOKS", hooks)
    if coarse_out:
        monkeypatch.setenv("FTKX_SERIES_OUT_COHERENT", "0")      # (the record buffer in coarse-grained host memory: the switch of rounds 3-5's default)
    g = load_golden(name)
    nd, nv, nt = g["nd"], g["nv"], g["DT"]
    scopes = [gpu.SCOPE_BOTH if t + 1 < nt else gpu.SCOPE_ORDINAL for t in range(nt)]
    ctx = _ctx(gpu, g["dims"], nd, nv, tag_mode=gpu.TAG_EXACT64, robust=int(g["robust"]), compute_degrees=int(g["degrees"]))
    _push_all(ctx, g["steps"], nv)
    want, wf, wrun = ctx.sweep_series(range(nt), scopes)
    for _ in range(2):
        ctx.invalidate_masks(); ctx.sweep_series_submit(range(nt), scopes)
    for it in range(6):
        ctx.invalidate_masks(); ctx.sweep_series_submit(range(nt), scopes)
        if it == 0:
            with pytest.raises(Exception):
                ctx.sweep_series_submit(range(nt), scopes)
        got, f, run = ctx.sweep_series_complete()
        assert _same(got, want) and np.array_equal(f, wf) and run == wrun, (name, it, len(got), len(want))
    for _ in range(2):
        got, f, run = ctx.sweep_series_complete()
        assert _same(got, want) and np.array_equal(f, wf) and run == wrun
    # pieces of two steps, chained on the device, three open
    pieces = [list(range(a, min(a + 2, nt))) for a in range(0, nt, 2)]
    ctx.invalidate_masks()
    ref, run = [], None
    for p in pieces:
        sc = [gpu.SCOPE_BOTH if t + 1 < nt else gpu.SCOPE_ORDINAL for t in p]
        r, f, run = ctx.sweep_series(p, sc, running_resolution=run)
        ref.append((r, [int(v) for v in f]))
    ctx.invalidate_masks()
    done = []
    for i, p in enumerate(pieces):
        sc = [gpu.SCOPE_BOTH if t + 1 < nt else gpu.SCOPE_ORDINAL for t in p]
        ctx.sweep_series_submit(p, sc, chain=i > 0)
        if i >= 2:
            done.append(ctx.sweep_series_complete())
    while len(done) < len(pieces):
        done.append(ctx.sweep_series_complete())
    for i, (r, f, run) in enumerate(done):
        assert _same(r, ref[i][0]) and [int(v) for v in f] == ref[i][1], (name, i, len(r), len(ref[i][0]))
    ctx.close()


@pytest.mark.parametrize("name", ["moving_extremum_3d_21x21x21x32", "woven_128x128x10"])
def test_unsplit_pass_behind_two_open_tails(gpu, name, monkeypatch):
    """Two split passes open, their tails on the two tail streams (tail sets 0 and 1), then a pass that is NOT split -- it zeroes and reuses
    the context's own counters, lists and ordering arrays, and what is dropped while it is the newest pass goes straight back to the pools:
    the context's stream has to wait for BOTH tails, not only for the pass queued last (round-5 advisor finding, series.hip
    wait_for_open_tails).  Cycles of split, split, unsplit (chain, one-launch and host-driven batch in turn), a slice dropped and pushed
    again behind the unsplit pass; every pass returns the records of ftkx_sweep_series."""
    g = load_golden(name)
    nd, nv, nt = g["nd"], g["nv"], g["DT"]
    scopes = [gpu.SCOPE_BOTH if t + 1 < nt else gpu.SCOPE_ORDINAL for t in range(nt)]
    ctx = _ctx(gpu, g["dims"], nd, nv, tag_mode=gpu.TAG_EXACT64, robust=int(g["robust"]), compute_degrees=int(g["degrees"]))
    _push_all(ctx, g["steps"], nv)
    monkeypatch.setenv("FTKX_SERIES_HOOKS", "one=0")
    want, wf, wrun = ctx.sweep_series(range(nt), scopes)
    third = ["one=0,split=0", "one=1,split=0", "one=0,split=0"]
    for it in range(6):
        for k in range(3):
            monkeypatch.setenv("FTKX_SERIES_HOOKS", "one=0,split=2" if k < 2 else third[it % 3])
            if k == 2 and it % 3 == 2:
                monkeypatch.setenv("FTKX_SERIES", "0")                 # the host-driven batch as the unsplit pass
            ctx.invalidate_masks(); ctx.sweep_series_submit(range(nt), scopes)
            monkeypatch.delenv("FTKX_SERIES", raising=False)
        # the newest pass is not split: a dropped slice is not parked with it -- its arrays go to the pools and come straight back, here
        # under another timestep and full of other values, while the three passes that read the slice are still out
        # (not behind the host-driven batch: that one reads its slices when it is completed)
        last = nt - 1
        swap = it % 3 != 2
        if swap:
            ctx.drop_slice(last)
            junk = np.full_like(g["steps"][last], 1234.5)
            (ctx.push_scalar_slice if nv == 1 else ctx.push_slice)(nt + 5, junk)
        for k in range(3):
            got, f, run = ctx.sweep_series_complete()
            assert _same(got, want) and np.array_equal(f, wf) and run == wrun, (name, it, k, len(got), len(want))
        if swap:
            ctx.drop_slice(nt + 5)
            _push_all(ctx, g["steps"], nv, only=[last])
    ctx.close()


def test_split_forced_on_and_off_give_identical_bytes_at_256_cubed(gpu, monkeypatch):
    """The split pass pinned both ways (FTKX_SERIES_HOOKS split=0 / split=2) on a C3-sized volume, 256^3 x 4, pipelined: the same bytes, the
    same factors, path 5 only where it was asked for -- and ftkx_series_split_decision says which setting a context ran under (auto: what the
    self-check measured; forced: that it was forced)."""
    import torch
    from ftk_amd import synthetic
    dev = torch.device("cuda", 0)
    dims, nt = (256, 256, 256), 4
    scopes = [gpu.SCOPE_BOTH if t + 1 < nt else gpu.SCOPE_ORDINAL for t in range(nt)]
    slices = [synthetic.generate("moving_extremum_3d", dims, t, 16, torch, dev) for t in range(nt)]
    torch.cuda.synchronize()
    got = {}
    for mode, hooks in (("off", "one=0,split=0"), ("on", "one=0,split=2"), ("auto", "one=0")):
        monkeypatch.setenv("FTKX_SERIES_HOOKS", hooks)
        ctx = _ctx(gpu, dims, 3, 1, tag_mode=gpu.TAG_EXACT64)
        for t in range(nt):
            ctx.push_scalar_slice(t, slices[t])
        outs, paths = [], set()
        for _ in range(2):
            ctx.invalidate_masks(); ctx.sweep_series_submit(range(nt), scopes)
        for it in range(6):
            ctx.invalidate_masks(); ctx.sweep_series_submit(range(nt), scopes)
            r, f, run = ctx.sweep_series_complete()
            outs.append((np.ascontiguousarray(r).tobytes(), [int(v) for v in f], run)); paths.add(ctx.series_last_path()[0])
        for _ in range(2):
            r, f, run = ctx.sweep_series_complete()
            outs.append((np.ascontiguousarray(r).tobytes(), [int(v) for v in f], run)); paths.add(ctx.series_last_path()[0])
        d = ctx.series_split_decision()
        ctx.close()
        assert all(o == outs[0] for o in outs), mode
        got[mode] = (outs[0], paths, d)
    assert got["off"][0] == got["on"][0] == got["auto"][0] and len(got["off"][0][0]) > 0
    assert 5 not in got["off"][1] and 5 in got["on"][1]
    assert got["off"][2]["state"] == "forced off" and got["on"][2]["state"] == "forced on" and got["auto"][2]["state"].startswith("auto")


def test_pipelined_records_through_the_copy_engine(gpu):
    """more than 4096 records per pass: from the second pipelined pass on the record kernel leaves them in device memory and the copy
    engine brings them over while the next pass runs; records and factors as ftkx_sweep_series returns them"""
    import torch
    from ftk_amd import synthetic
    dev = torch.device("cuda", 0)
    dims, nt = (1024, 512), 16
    scopes = [gpu.SCOPE_BOTH if t + 1 < nt else gpu.SCOPE_ORDINAL for t in range(nt)]
    ctx = _ctx(gpu, dims, 2, 1, tag_mode=gpu.TAG_EXACT64)
    for t in range(nt):
        ctx.push_scalar_slice(t, synthetic.generate("woven", dims, t, nt, torch, dev))
    want, wf, wrun = ctx.sweep_series(range(nt), scopes)
    assert len(want) > 4096
    ctx.invalidate_masks()
    ctx.sweep_series_submit(range(nt), scopes)
    for it in range(3):
        ctx.invalidate_masks()
        ctx.sweep_series_submit(range(nt), scopes)
        got, f, run = ctx.sweep_series_complete()
        assert _same(got, want) and np.array_equal(f, wf) and run == wrun, (it, len(got), len(want))
    got, f, run = ctx.sweep_series_complete()
    assert _same(got, want) and np.array_equal(f, wf) and run == wrun
    ctx.close()


def test_pipelined_passes_through_buffers_that_are_too_small(gpu):
    """two passes in flight on a series with more records than the initial hit buffer holds: the first is flagged, swept again by the
    host-driven batch (which grows the buffers) while the second is still out; the second is flagged as well; from the third on the passes
    fit -- every pass returns the records of the plain call"""
    import torch
    from ftk_amd import synthetic
    dev = torch.device("cuda", 0)
    dims, nt = (2048, 2048), 12
    scopes = [gpu.SCOPE_BOTH if t + 1 < nt else gpu.SCOPE_ORDINAL for t in range(nt)]
    ref = _ctx(gpu, dims, 2, 1, tag_mode=gpu.TAG_EXACT64)
    snaps = [synthetic.generate("woven", dims, t, nt, torch, dev) for t in range(nt)]
    for t in range(nt):
        ref.push_scalar_slice(t, snaps[t])
    want, wf, wrun = ref.sweep_series(range(nt), scopes)
    ref.close()
    if len(want) <= 65536:
        pytest.skip(f"only {len(want)} records: the buffers would fit")
    ctx = _ctx(gpu, dims, 2, 1, tag_mode=gpu.TAG_EXACT64)
    for t in range(nt):
        ctx.push_scalar_slice(t, snaps[t])
    paths = []
    ctx.sweep_series_submit(range(nt), scopes)
    for it in range(4):
        ctx.invalidate_masks()
        ctx.sweep_series_submit(range(nt), scopes)
        got, f, run = ctx.sweep_series_complete()
        paths.append(ctx.series_last_path())
        assert _same(got, want) and np.array_equal(f, wf) and run == wrun, (it, len(got), len(want), paths)
    got, f, run = ctx.sweep_series_complete()
    paths.append(ctx.series_last_path())
    assert _same(got, want) and np.array_equal(f, wf) and run == wrun, paths
    assert paths[0][0] == 0 and (paths[0][1] & 8), paths            # flagged SERIES_OVERFLOW, replayed by the batch
    assert paths[-1][0] == 1, paths                                  # ... and the later passes fit
    ctx.close()


def test_series_overflowing_buffers_replays_through_the_batch(gpu):
    """more records than the initial hit buffer holds (65 536): flagged by the finish kernel, swept again by the host-driven batch, and
    the NEXT call fits"""
    from ftk_amd import synthetic
    import torch
    dims, nt = (2048, 2048), 12
    ctx = _ctx(gpu, dims, 2, 1, tag_mode=gpu.TAG_EXACT64)
    dev = torch.device("cuda", 0)
    keep = []
    for t in range(nt):
        s = synthetic.generate("woven", dims, t, nt, torch, dev); keep.append(s)
        torch.cuda.synchronize()
        ctx.push_scalar_slice(t, s)
    scopes = [gpu.SCOPE_BOTH if t + 1 < nt else gpu.SCOPE_ORDINAL for t in range(nt)]
    r1, f1, _ = ctx.sweep_series(range(nt), scopes)
    p1 = ctx.series_last_path()
    if len(r1) <= 65536:
        pytest.skip(f"only {len(r1)} records: the initial buffer holds them")
    assert p1[0] == 0 and (p1[1] & 8), p1
    ctx.invalidate_masks()
    r2, f2, _ = ctx.sweep_series(range(nt), scopes)
    p2 = ctx.series_last_path()
    assert p2[0] == 1, p2
    assert _same(r1, r2) and np.array_equal(f1, f2)
    assert np.all(r2["tag"][1:] > r2["tag"][:-1])
    ctx.close()


def test_a_short_chain_pass_that_declines_late(gpu):
    """The fused tail with nothing queued behind it (the pass before was finished by it: short chain) meets few coarse cells with more
    records in them than it orders: it takes back what it counted, says so (SERIES_LATE_DECLINE) and hands the pass to the host, which
    queues the chain -- the records are those of the host-driven batch, and the passes after it go out as whole chains again."""
    from ftk_amd import tslab
    g = load_golden("woven_128x128x10")
    nt = g["DT"]
    late = []
    for nsub in (2, 3, 4, 5):
        ctx = _ctx(gpu, g["dims"], g["nd"], g["nv"], tag_mode=gpu.TAG_EXACT64)
        _push_all(ctx, g["steps"], g["nv"])
        ts, scopes = list(range(nsub)), [gpu.SCOPE_BOTH] * nsub
        # the records of these steps through the host-driven batch
        rm = ctx.slices_prepare(range(nsub + 1), 0)
        factors = tslab.factors_from_resolutions([rm[t][0] for t in range(nsub + 1)])[:nsub]
        ctx.sweep_enqueue_many(ts, scopes, factors)
        want = np.array(ctx.sweep_collect())
        ctx.invalidate_masks()
        # a sparse pass first: one ordinal sweep (a few hundred records), which the fused tail finishes
        recs0, _, _ = ctx.sweep_series([0], [gpu.SCOPE_ORDINAL])
        seen = [ctx.series_last_path()]
        assert len(recs0) > 0 and seen[0][0] == 2, seen
        for _ in range(3):
            got, f, _ = ctx.sweep_series(ts, scopes)
            seen.append(ctx.series_last_path())
            assert _same(np.array(got), want), (nsub, seen)
            assert [int(v) for v in f] == [int(v) for v in factors]
        print(nsub, len(want), seen)
        if any(st & 64 for _, st in seen[1:]):
            late.append(nsub)
            assert seen[-1][0] == 1 and not (seen[-1][1] & 64), seen      # after a late decline the fused tail is not tried again for a while
        ctx.close()
    assert late, "none of the sub-series made the fused tail decline late"


# ---- the one-launch pass for small series (csrc/one_kernel.hip) ------------------------------------------------------------------------------
ONE_TAKEN = {}


@pytest.mark.parametrize("name", [n for n in golden_names()])
def test_one_launch_pass_matches_reference_fixture(gpu, monkeypatch, name):
    """every reference fixture through ftkx_sweep_series with the one-launch pass on (the default): records, their order, factors and the
    running resolution are what the kernel chain gives -- path 4 wherever the series is small and the options are the device-driven pass's,
    on its own, two in flight, and chained on the device (the second pass continues from the first's running minimum)"""
    g = load_golden(name)
    if not _plain(g):
        pytest.skip("physical coordinates are set on the tracker")
    nd, nv, nt = g["nd"], g["nv"], g["DT"]
    opts = dict(robust=int(g["robust"]), compute_degrees=int(g["degrees"]))
    if g["type_filter"] is not None:
        opts.update(use_type_filter=1, type_filter=g["type_filter"])
    scopes = [gpu.SCOPE_BOTH if t + 1 < nt else gpu.SCOPE_ORDINAL for t in range(nt)]
    ctx = _ctx(gpu, g["dims"], nd, nv, **opts)
    _push_all(ctx, g["steps"], nv)
    want, wf, wrun = ctx.sweep_series(range(nt), scopes)          # (FTKX_SERIES_HOOKS one=0: this module's default)
    want, wpath = want.copy(), ctx.series_last_path()
    ctx.close()
    monkeypatch.setenv("FTKX_SERIES_HOOKS", "one=1")
    ctx = _ctx(gpu, g["dims"], nd, nv, **opts)
    _push_all(ctx, g["steps"], nv)
    recs, factors, run = ctx.sweep_series(range(nt), scopes)
    path, status = ctx.series_last_path()
    ONE_TAKEN[name] = (path, status, wpath)
    assert np.array_equal(factors, g["factors"]) and np.array_equal(factors, wf), f"{name}: factors {factors} (path {path}, status {status})"
    assert run == wrun, (name, run, wrun)
    assert _same(recs, want), f"{name}: {len(recs)} vs {len(want)} records (path {path}, status {status}; the chain took {wpath})"
    assert_records_equal(_as_fixture(recs), g["records"], coord_tol=0.0, what=f"{name} (path {path}, status {status})")
    # two in flight; then a pass in two halves, the second continuing ON THE DEVICE from the first's running minimum
    ctx.sweep_series_submit(range(nt), scopes)
    ctx.sweep_series_submit(range(nt), scopes)
    for _ in range(2):
        r2, f2, run2 = ctx.sweep_series_complete()
        assert _same(r2, want) and np.array_equal(f2, wf) and run2 == wrun, name
    if nt >= 4:
        h = nt // 2
        ctx.sweep_series_submit(range(h), [gpu.SCOPE_BOTH] * h)
        ctx.sweep_series_submit(range(h, nt), scopes[h:], chain=True)
        a, fa, _ = ctx.sweep_series_complete()
        a = a.copy()
        b, fb, runb = ctx.sweep_series_complete()
        both = np.concatenate([a, b])
        assert np.array_equal(np.concatenate([fa, fb]), wf) and runb == wrun and _same(both, want), name
    ctx.close()


def test_the_one_launch_pass_took_the_small_series(gpu):
    if len(ONE_TAKEN) < 20:
        pytest.skip("the fixture tests above did not run in this process")
    took = [n for n, (p, st, wp) in ONE_TAKEN.items() if p == 4]
    print({n: v for n, v in ONE_TAKEN.items() if v[0] != 4})
    assert len(took) >= 20, ONE_TAKEN
    # what the device-driven pass does not cover goes to the host-driven batch with or without it; what the chain handed to the batch for a
    # kernel-raised flag (an ambiguous factor, ...) the one-launch pass hands over as well
    assert all(p == 4 or wp[0] == 0 or p in (0, 1, 2) for n, (p, st, wp) in ONE_TAKEN.items())
    for n in ("woven_128x128x10", "woven_31x37x32", "moving_extremum_3d_21x21x21x32", "moving_extremum_3d_32x32x32x8_dyadic"):      # (double_gyre_64x32x50: 50 steps, more than the kernel's argument block holds)
        if n in ONE_TAKEN:
            assert ONE_TAKEN[n][0] == 4, (n, ONE_TAKEN[n])
