"""Size-independent properties at (or towards) BASELINE.json's full sizes, where the CPU oracle is too slow to be the checker:
analytic positions, cull == exact_only, partition of `core` == whole, idempotence, sortedness/uniqueness of tags."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def gpu():
    import torch
    assert torch.cuda.is_available()
    import ftk_amd
    from ftk_amd import build
    build.build()
    return ftk_amd


def _run(gpu, case, dims, nt, *, exact_only=False, core=None, steps=None, nv=1, tag_mode=None, prepass="fused", hint=0, announce=None):
    import torch
    from ftk_amd import synthetic, tslab
    nd = len(dims)
    scalar = nv == 1
    dev = torch.device("cuda", 0)
    lo = 2 if scalar else 1
    dom = ([lo] * nd, [d - (3 if scalar else 2) for d in dims])
    ctx = gpu.Context(nd)
    ctx.set_mesh(dom, core or dom, ([0] * nd, list(dims)))
    ctx.set_options(jacobian_symmetric=scalar, derive_jacobian=1, tag_mode=tag_mode or gpu.TAG_EXACT64, exact_only=exact_only)
    keep, res = [], []
    for t in range(nt):
        a = synthetic.generate(case, dims, t, nt, torch, dev) if steps is None else torch.from_numpy(np.ascontiguousarray(steps[t])).to(dev)
        torch.cuda.synchronize()
        keep.append(a)
        (ctx.push_scalar_slice if scalar else ctx.push_slice)(t, a)
        if prepass == "exact":      # the separate pre-pass: ndarray::resolution() of every slice, masks built later under the true factor
            res.append(ctx.slice_resolution(t)[0])
    if prepass == "fused":          # the product's one-pass form: masks (under `hint`) and the capped reduction from one kernel
        if announce is not None:    # cull-ahead: [(t, scope)] the caller says it will sweep ("all" = what is enqueued below)
            ann = [(t, gpu.SCOPE_BOTH if t + 1 < nt else gpu.SCOPE_ORDINAL) for t in range(nt)] if announce == "all" else announce
            ctx.sweep_announce([a[0] for a in ann], [a[1] for a in ann])
        rm = ctx.slices_prepare(range(nt), hint)
        res = [rm[t][0] for t in range(nt)]
    factors = tslab.factors_from_resolutions(res)
    for t in range(nt):
        ctx.sweep_enqueue(t, gpu.SCOPE_BOTH if t + 1 < nt else gpu.SCOPE_ORDINAL, factors[t])
    recs = ctx.sweep_collect()
    st = ctx.stats()
    ctx.close()
    return recs, st, factors


def _same(a, b):
    assert len(a) == len(b)
    for f in ("tag", "type", "aux"):
        assert np.array_equal(a[f], b[f]), f
    for f in ("x", "t", "scalar"):
        assert np.array_equal(a[f], b[f], equal_nan=True), f


def test_c3_moving_extremum_256cubed_analytic(gpu):
    """BASELINE configs[2]: 256^3 x 16, 1.47e10 simplices.  One extremum on x0 + dir*t, every record a minimum."""
    from ftk_amd import synthetic, tslab
    dims, nt = (256, 256, 256), 16
    recs, st, factors = _run(gpu, "moving_extremum_3d", dims, nt)
    assert st["work_items"] == tslab.count_simplices(3, dims, nt) and st["cull_enabled"] == 1
    assert set(factors) == {256}
    x0, dv = synthetic.moving_extremum_params(dims)
    assert len(recs) > 0 and set(recs["type"].tolist()) == {2}
    for a in range(3):
        assert np.abs(recs["x"][:, a] - (x0[a] + dv[a] * recs["t"])).max() < 1e-6      # north_star tolerance (observed ~1e-14)
    assert np.all(np.diff(recs["tag"].astype(np.uint64)) > 0)                         # sorted, unique
    ordinal = recs[(recs["aux"] & 1) == 1]
    assert np.array_equal(ordinal["t"], (ordinal["aux"] >> 1).astype(float))           # ordinal points sit exactly on their timestep
    assert len(np.unique(ordinal["aux"] >> 1)) == nt                                   # the trajectory crosses every slice
    assert recs["scalar"][:, 0].min() >= 0 and recs["scalar"][:, 0].max() < 1.0        # S = |x - xc|^2 near the extremum


def test_cull_equals_exact_only_128cubed(gpu):
    recs_c, st_c, _ = _run(gpu, "moving_extremum_3d", (128, 128, 128), 4)
    recs_e, st_e, _ = _run(gpu, "moving_extremum_3d", (128, 128, 128), 4, exact_only=True)
    assert st_c["cull_enabled"] == 1 and st_e["cull_enabled"] == 0
    assert st_e["simplices_tested"] > 100 * max(1, st_c["simplices_tested"])
    _same(recs_c, recs_e)


@pytest.mark.parametrize("dims", [(200, 150, 37), (136, 49, 18)])
def test_rough_3d_cull_equals_exact_only(gpu, dims):
    """Plateaus, ties and noise in 3D -- most 8 x 16 blocks have no common sign bit, many of a wavefront's 8 x 4 sub-blocks do (their
    summaries go out as stand-in mask bytes one step late), tiles that end inside the array in x and y: the culled sweep against the
    integer fallback that tests every simplex, and the series pass against both."""
    nt = 3
    rng = np.random.default_rng(23)
    shape = tuple(reversed(dims))
    steps = [np.round(rng.standard_normal(shape) * 2) * 0.25 + rng.integers(-2, 3, size=shape) / 64.0 for _ in range(nt)]
    for z in range(0, shape[0], 5):       # smooth slabs in between: uniform blocks next to rough ones
        steps[1][z] = np.linspace(1.0, 2.0, shape[2])[None, :] + np.linspace(0.0, 0.5, shape[1])[:, None]
    recs_c, st_c, f_c = _run(gpu, "moving_extremum_3d", dims, nt, steps=steps)
    recs_e, st_e, f_e = _run(gpu, "moving_extremum_3d", dims, nt, steps=steps, exact_only=True)
    assert st_c["cull_enabled"] == 1 and st_e["cull_enabled"] == 0 and list(f_c) == list(f_e)
    assert len(recs_c) > 1000
    _same(recs_c, recs_e)
    import torch
    ctx = gpu.Context(3)
    dom = ([2] * 3, [d - 3 for d in dims])
    ctx.set_mesh(dom, dom, ([0] * 3, list(dims)))
    ctx.set_options(jacobian_symmetric=1, derive_jacobian=1, tag_mode=gpu.TAG_EXACT64)
    keep = [torch.from_numpy(np.ascontiguousarray(a)).to("cuda") for a in steps]
    for t in range(nt):
        ctx.push_scalar_slice(t, keep[t])
    got, f, _ = ctx.sweep_series(range(nt), [gpu.SCOPE_BOTH if t + 1 < nt else gpu.SCOPE_ORDINAL for t in range(nt)])
    assert [int(v) for v in f] == [int(v) for v in f_c]
    _same(np.array(got), recs_c)
    ctx.close()


def test_cull_equals_exact_only_woven_1024(gpu):
    """BASELINE configs[1] geometry (1024 x 1024), 6 timesteps: ~5e4 records"""
    recs_c, st_c, f = _run(gpu, "woven", (1024, 1024), 6)
    recs_e, st_e, _ = _run(gpu, "woven", (1024, 1024), 6, exact_only=True)
    assert len(recs_c) > 3000 and st_c["cull_enabled"] == 1
    _same(recs_c, recs_e)
    assert set(recs_c["type"].tolist()) <= {1, 2, 4, 8}


def test_double_gyre_vector_path_cull_equals_exact(gpu):
    """BASELINE configs[4] kind (vector input, non-symmetric Jacobian, nbits 21) at 512 x 256 x 6"""
    recs_c, st_c, f = _run(gpu, "double_gyre", (512, 256), 6, nv=2)
    recs_e, st_e, _ = _run(gpu, "double_gyre", (512, 256), 6, nv=2, exact_only=True)
    assert set(f) == {1 << 21}
    _same(recs_c, recs_e)
    assert len(recs_c) > 0 and set(recs_c["type"].tolist()) == {4}      # the derived off-diagonals are 0 -> saddles


def test_partition_of_core_equals_whole(gpu):
    """sweeping two halves of `core` (what a spatially partitioned caller would do) and merging == sweeping the whole domain"""
    dims, nt = (96, 80, 64), 4
    whole, _, _ = _run(gpu, "moving_extremum_3d", dims, nt)
    lo, sz = [2, 2, 2], [d - 3 for d in dims]
    h = sz[0] // 2
    left, _, _ = _run(gpu, "moving_extremum_3d", dims, nt, core=(lo, [h, sz[1], sz[2]]))
    right, _, _ = _run(gpu, "moving_extremum_3d", dims, nt, core=([lo[0] + h, 2, 2], [sz[0] - h, sz[1], sz[2]]))
    merged = np.concatenate([left, right])
    merged = merged[np.argsort(merged["tag"], kind="stable")]
    _same(whole, merged)
    rng = np.random.default_rng(1)
    steps = [np.cumsum(rng.standard_normal((40, 56)), axis=1) * 0.25 for _ in range(3)]
    whole, _, _ = _run(gpu, None, (56, 40), 3, steps=steps)
    a, _, _ = _run(gpu, None, (56, 40), 3, steps=steps, core=([2, 2], [53, 17]))
    b, _, _ = _run(gpu, None, (56, 40), 3, steps=steps, core=([2, 19], [53, 20]))
    merged = np.concatenate([a, b]); merged = merged[np.argsort(merged["tag"], kind="stable")]
    assert len(whole) > 50
    _same(whole, merged)


def test_idempotent_and_deterministic(gpu):
    a, _, _ = _run(gpu, "woven", (300, 260), 5)
    b, _, _ = _run(gpu, "woven", (300, 260), 5)
    _same(a, b)


def test_reference_and_exact_tags_agree_without_overflow(gpu):
    a, _, _ = _run(gpu, "moving_extremum_3d", (64, 64, 64), 4, tag_mode=gpu.TAG_REFERENCE)
    b, _, _ = _run(gpu, "moving_extremum_3d", (64, 64, 64), 4, tag_mode=gpu.TAG_EXACT64)
    assert np.array_equal(a["tag"], b["tag"])


@pytest.mark.parametrize("case,dims,nt,rough", [("moving_extremum_3d", (256, 128, 72), 4, False), ("moving_extremum_3d", (130, 70, 40), 3, True),
                                                ("moving_extremum_3d", (126, 33, 6), 3, True), ("moving_extremum_3d", (386, 50, 35), 2, True),
                                                ("woven", (1024, 512), 5, False), ("woven", (258, 100), 4, True)])
def test_mask_kernel_generations_agree(gpu, case, dims, nt, rough):
    """The marching mask kernel (3D: mask_march6_kernel, 2D: mask_march4_kernel) under every workgroup placement, every way of cutting a
    tile column into pieces of planes, with and without summaries, with summaries per word, per 8 x 4 block and (3D) per 8 x 16 block:
    same results, the same fused reduction AND, within a geometry, the same cull statistics = the same mask / summary bytes where it matters.
    (Rounds 1-3 carried five generations of this kernel side by side; round 4 keeps one per dimension and one generic form.)"""
    import os
    steps = None
    if rough:   # a field with plateaus, ties and noise: many non-uniform words, masks actually written and refined
        rng = np.random.default_rng(7)
        shape = tuple(reversed(dims))
        # (values on a 1/64 grid: the resolution stays at 2^-7, so the determinants cannot overflow and the cull stays legal)
        steps = [np.round(rng.standard_normal(shape) * 2) * 0.25 + rng.integers(-2, 3, size=shape) / 64.0 for _ in range(nt)]
    variants = [{},                                                                              # mask_march6_kernel<2, 4, 4, false>, its default pieces
                {"FTKX_MASK_SWIZZLE": "0"}, {"FTKX_MASK_SWIZZLE": "1"}, {"FTKX_MASK_SWIZZLE": "24", "FTKX_MASK_YG": "3"}, {"FTKX_MASK_YG": "4"},
                {"FTKX_MASK_ZCHUNK": "16"}, {"FTKX_MASK_ZCHUNK": "7"}, {"FTKX_TWO_LEVEL": "0"}, {"FTKX_U_ROWS": "1"}, {"FTKX_U_ROWS": "1", "FTKX_MASK_ZCHUNK": "6"},
                # the pieces a column is marched in: short ones only, one piece per column, many tiny ones (more than the table holds: merged)
                {"FTKX_MASK_LMIN": "1", "FTKX_MASK_LCAP": "3"}, {"FTKX_MASK_LCAP": "100000"}, {"FTKX_MASK_LMIN": "1", "FTKX_MASK_LCAP": "1"}, {"FTKX_MASK_ZCHUNK": "1"},
                {"FTKX_MASK_ZCHUNK": "32"}, {"FTKX_MASK_LMIN": "6", "FTKX_MASK_LCAP": "24"}, {"FTKX_MASK_ORDER": "0"}, {"FTKX_MASK_ORDER": "1"}, {"FTKX_MASK_ORDER": "0", "FTKX_MASK_ZCHUNK": "5"}]
    variants += [{"FTKX_U_ROWS": "4"}, {"FTKX_U_ROWS": "4", "FTKX_MASK_ZCHUNK": "7"}, {"FTKX_U_ROWS": "4", "FTKX_MASK_LCAP": "3", "FTKX_MASK_LMIN": "1"}]
    # 2D: mask_rows2_kernel (a wavefront marches down `rows` groups of 8 rows; 0: mask_march4_kernel, one group per wavefront)
    variants += [{"FTKX_MASK_ROWS": "0"}, {"FTKX_MASK_ROWS": "1"}, {"FTKX_MASK_ROWS": "3"}, {"FTKX_MASK_ROWS": "5", "FTKX_U_ROWS": "1"},
                 {"FTKX_MASK_ROWS": "2", "FTKX_TWO_LEVEL": "0"}, {"FTKX_MASK_ROWS": "7", "FTKX_MASK_SWIZZLE": "0"}, {"FTKX_MASK_ROWS": "0", "FTKX_U_ROWS": "1"}]
    base = None
    by_geometry = {}
    def hooks(env):
        """the variants above name one knob each; the library reads the mask kernels' launch geometry from ONE variable, FTKX_MASK_PLAN
        ("name=value,..."), and the summary geometry from FTKX_U_ROWS (-1: no summaries, the one-level cull)"""
        plan = ",".join("%s=%s" % (k[len("FTKX_MASK_"):].lower(), v) for k, v in env.items() if k.startswith("FTKX_MASK_"))
        out = {"FTKX_MASK_PLAN": plan} if plan else {}
        if env.get("FTKX_TWO_LEVEL") == "0":
            out["FTKX_U_ROWS"] = "-1"
        elif "FTKX_U_ROWS" in env:
            out["FTKX_U_ROWS"] = env["FTKX_U_ROWS"]
        return out
    for env in variants:
        real = hooks(env)
        old = {k: os.environ.get(k) for k in real}
        os.environ.update(real)
        try:
            recs, st, factors = _run(gpu, case, dims, nt, steps=steps)
        finally:
            for k, v in old.items():
                if v is None:
                    os.environ.pop(k, None)
                else:
                    os.environ[k] = v
        key = (st["cells_survived"], st["simplices_tested"], tuple(factors))
        if base is None:
            base = recs
            assert st["cull_enabled"] == 1
        _same(recs, base)                                      # the records never differ
        # The cull statistics are those of the summary geometry -- none (one-level cull), one byte per word of 8, per 8 x 4 block, per
        # 8 x 16 block (3D, the default: the workgroup's sixteen rows) --: every way of placing workgroups and cutting columns gives the
        # same numbers within a geometry, and a coarser stand-in for unwritten mask words can only let MORE simplices through to the test
        default_rows = 16 if len(dims) == 3 else 4
        rows = 0 if env.get("FTKX_TWO_LEVEL") == "0" else int(env.get("FTKX_U_ROWS", default_rows))
        rows = min(rows, default_rows)
        by_geometry.setdefault(rows, (key, env))
        assert key == by_geometry[rows][0], (env, key, by_geometry[rows])
    assert set(by_geometry) >= ({0, 1, 4, 16} if len(dims) == 3 else {0, 1, 4}), sorted(by_geometry)
    order = sorted(by_geometry)
    for fine, coarse in zip(order, order[1:]):
        assert by_geometry[fine][0][2] == by_geometry[coarse][0][2]
        assert by_geometry[fine][0][1] <= by_geometry[coarse][0][1], (fine, coarse, by_geometry[fine], by_geometry[coarse])


def test_slices_beyond_4GiB_take_the_64bit_kernels(gpu):
    """Maximum sizes: a 1024 x 1024 x 520 slice is 4.06 GiB, past the 32-bit byte offsets of the marching mask kernels
    (march2_supported is false) -- the sweep then runs the size_t-indexed kernels.  Same analytic trajectory, same records with
    and without the cull, and the same records as the 32-bit path gives on the sub-volume that fits it."""
    from ftk_amd import synthetic
    dims, nt = (1024, 1024, 520), 3
    assert dims[0] * dims[1] * dims[2] * 8 >= 1 << 32
    recs, st, factors = _run(gpu, "moving_extremum_3d", dims, nt)
    assert st["cull_enabled"] == 1 and set(factors) == {256}
    x0, dv = synthetic.moving_extremum_params(dims)
    assert len(recs) >= nt and set(recs["type"].tolist()) == {2}
    for a in range(3):
        assert np.abs(recs["x"][:, a] - (x0[a] + dv[a] * recs["t"])).max() < 1e-6
    assert np.all(np.diff(recs["tag"].astype(np.uint64)) > 0)
    # the extremum sits near the centre: a core of 64^3 cells around it, swept without the cull, must give the same records
    lo = [int(x0[a]) - 32 for a in range(3)]
    core = (lo, [64, 64, 64])
    sub, st_e, _ = _run(gpu, "moving_extremum_3d", dims, nt, exact_only=True, core=core)
    assert st_e["cull_enabled"] == 0
    inside = np.ones(len(recs), dtype=bool)
    _same(recs[inside], sub)


def test_batched_resolution_equals_per_slice(gpu, oracle):
    """ftkx_slices_resolution (one round trip for the whole series) == ftkx_slice_resolution slice by slice, bit for bit:
    marching reduction (3D, 2D), the generic scalar reduction (odd row length) and vector input"""
    import torch
    from ftk_amd import synthetic
    for case, dims, nt, nv in (("moving_extremum_3d", (128, 96, 40), 5, 1), ("woven", (256, 128), 6, 1), ("moving_extremum_3d", (31, 17, 9), 3, 1),
                               ("double_gyre", (128, 64), 5, 2)):
        nd = len(dims)
        lo = 2 if nv == 1 else 1
        dom = ([lo] * nd, [d - (3 if nv == 1 else 2) for d in dims])
        got = []
        for batched in (False, True):
            ctx = gpu.Context(nd)
            ctx.set_mesh(dom, dom, ([0] * nd, list(dims)))
            ctx.set_options(jacobian_symmetric=int(nv == 1), derive_jacobian=1)
            keep = []
            for t in range(nt):
                a = synthetic.generate(case, dims, t, nt, torch, torch.device("cuda", 0)); torch.cuda.synchronize(); keep.append(a)
                (ctx.push_scalar_slice if nv == 1 else ctx.push_slice)(t, a)
            got.append(ctx.slices_resolution(range(nt)) if batched else {t: ctx.slice_resolution(t) for t in range(nt)})
            ctx.close()
        assert got[0] == got[1], (case, dims)
        assert all(r > 0 and m > 0 for r, m in got[0].values())
        # and both equal ndarray::resolution() of the derived (or given) vector field as the oracle computes it
        for t in range(nt):
            a = keep[t].cpu().numpy()
            V = a if nv == 2 else (oracle.gradient2D(a) if nd == 2 else oracle.gradient3D(a))
            assert got[1][t][0] == oracle.resolution(V), (case, dims, t)
            fin = np.abs(V[np.isfinite(V)])
            assert got[1][t][1] == fin.max()


def test_one_pass_prepare_equals_separate_prepass(gpu, oracle):
    """ftkx_slices_prepare (masks + reduction from ONE kernel, masks built under a factor that may be smaller than the final one)
    against the two-pass form (exact ndarray::resolution() first, masks under the true factor): same factors, same records;
    res_below / max_abs against the oracle's ndarray::resolution() of the same field."""
    import torch
    from ftk_amd import synthetic
    DBL_MAX = float(np.finfo(np.float64).max)
    cases = [("moving_extremum_3d", (128, 96, 40), 5, 1, None), ("woven", (256, 128), 6, 1, None), ("moving_extremum_3d", (31, 17, 9), 3, 1, None),
             ("double_gyre", (128, 64), 5, 2, None), ("double_gyre", (101, 64), 4, 2, None), ("woven", (1024, 512), 5, 1, None)]
    rng = np.random.default_rng(11)
    rough3 = [np.round(rng.standard_normal((20, 36, 130)) * 2) * 0.25 + rng.integers(-2, 3, size=(20, 36, 130)) / 64.0 for _ in range(3)]
    tiny2 = [rng.standard_normal((48, 64)) * 1e-5 for _ in range(3)]                  # every gradient below 2^-8: nothing strictly signed under the hint
    cases += [(None, (130, 36, 20), 3, 1, rough3), (None, (64, 48), 3, 1, tiny2)]
    for case, dims, nt, nv, steps in cases:
        a, sa, fa = _run(gpu, case, dims, nt, nv=nv, steps=steps, prepass="fused")
        b, sb, fb = _run(gpu, case, dims, nt, nv=nv, steps=steps, prepass="exact")
        assert fa == fb, (case, dims, fa, fb)
        _same(a, b)
        for hint in (1 << 12, 1 << 21):
            if hint <= min(fa):      # any hint up to the true factor is legal
                c, sc, fc = _run(gpu, case, dims, nt, nv=nv, steps=steps, prepass="fused", hint=hint)
                assert fc == fa
                _same(a, c)
        # the reduction itself
        nd = len(dims)
        lo = 2 if nv == 1 else 1
        dom = ([lo] * nd, [d - (3 if nv == 1 else 2) for d in dims])
        ctx = gpu.Context(nd)
        ctx.set_mesh(dom, dom, ([0] * nd, list(dims)))
        ctx.set_options(jacobian_symmetric=int(nv == 1), derive_jacobian=1)
        keep = []
        for t in range(nt):
            x = synthetic.generate(case, dims, t, nt, torch, torch.device("cuda", 0)) if steps is None else torch.from_numpy(np.ascontiguousarray(steps[t])).cuda()
            torch.cuda.synchronize(); keep.append(x)
            (ctx.push_scalar_slice if nv == 1 else ctx.push_slice)(t, x)
        for hint in (0, 1 << 14):
            ctx.invalidate_masks()
            got = ctx.slices_prepare(range(nt), hint)
            cap = 1.0 / (hint or 256)
            for t in range(nt):
                h = keep[t].cpu().numpy()
                V = h if nv == 2 else (oracle.gradient2D(h) if nd == 2 else oracle.gradient3D(h))
                r = oracle.resolution(V)
                assert got[t][0] == (r if r < cap else DBL_MAX), (case, dims, t, hint, got[t][0], r)
                assert got[t][1] == np.abs(V[np.isfinite(V)]).max(), (case, dims, t)
        ctx.close()


def test_big_vertices_are_handled_per_cell(gpu):
    """A few outliers large enough to overflow the determinants (|q| + 1 > safe_m): only the cells that touch them lose the cull,
    the records equal the exact_only sweep (the wrapped results included), and the rest of the field is still culled."""
    from ftk_amd import synthetic
    import torch
    dims, nt = (96, 80, 64), 4
    steps = [synthetic.generate("moving_extremum_3d", dims, t, nt, torch, torch.device("cuda", 0)).cpu().numpy() for t in range(nt)]
    rng = np.random.default_rng(3)
    for t in range(nt):
        for _ in range(6):
            k, j, i = (int(rng.integers(4, d - 4)) for d in reversed(dims))
            steps[t][k, j, i] += 3e6 * (1 + rng.random())      # gradient ~1.5e6 next to it: * 2^8 > 727041
    a, sa, fa = _run(gpu, None, dims, nt, steps=steps)
    b, sb, fb = _run(gpu, None, dims, nt, steps=steps, exact_only=True)
    assert fa == fb and sa["cull_enabled"] == 1 and sb["cull_enabled"] == 0
    _same(a, b)
    assert 0 < sa["cells_survived"] < 0.01 * sa["cells"]
    # 2D: same with a vector field and nbits 21
    dims2 = (256, 128)
    v = [synthetic.generate("double_gyre", dims2, t, 3, torch, torch.device("cuda", 0)).cpu().numpy() for t in range(3)]
    for t in range(3):
        v[t][40 + t, 100, 0] = 700.0; v[t][41 + t, 101, 1] = -650.0      # 700 * 2^21 > 1 239 850 262
    a, sa, fa = _run(gpu, None, dims2, 3, steps=v, nv=2)
    b, sb, fb = _run(gpu, None, dims2, 3, steps=v, nv=2, exact_only=True)
    assert fa == fb == [1 << 21] * 3
    _same(a, b)
    assert sa["cells_survived"] < 0.05 * sa["cells"]


def test_dense_survivors_fall_back_to_the_tile_kernel(gpu):
    """The int64-overflow regime at a size where the survivor list would not fit (all 3.9e6 cells per step survive): the batch is
    replayed through the tile kernel with its in-tile cull; records equal exact_only (wrapped determinants reproduced)."""
    import torch
    from ftk_amd import synthetic
    dims, nt = (160, 160, 160), 2
    x0 = [80 + 1e-7, 80 + 2e-7, 80 + 3e-7]
    ax = [torch.arange(d, dtype=torch.float64) for d in dims]
    steps = []
    for k in range(nt):
        c = [x0[a] + 0.1 * k for a in range(3)]
        steps.append((((ax[0] - c[0]) ** 2)[None, None, :] + ((ax[1] - c[1]) ** 2)[None, :, None] + ((ax[2] - c[2]) ** 2)[:, None, None]).numpy())
    a, sa, fa = _run(gpu, None, dims, nt, steps=steps)
    b, sb, fb = _run(gpu, None, dims, nt, steps=steps, exact_only=True)
    assert fa == fb == [1 << 21] * nt
    assert sa["cull_enabled"] == 1 and sa["simplices_tested"] > 0.5 * sa["work_items"]
    _same(a, b)
    assert len(a) > 10000       # mostly bogus records of wrapped determinants, like the reference's


@pytest.mark.parametrize("scale", [1.0, 3e4, 3e6, 3e9], ids=["25bit", "32bit", "mixed", "64bit"])
def test_tile_kernel_forms_agree_2d(gpu, monkeypatch, scale):
    """the same for the 2D tile kernel (12 triangles per corner; fp64 exact below 2^25, integer below 2^31, pairs above)"""
    rng = np.random.default_rng(23)
    dims, nt = (70, 41), 4
    steps = []
    for t in range(nt):
        f = rng.standard_normal(tuple(reversed(dims))) * 0.05 * scale
        f[5:12, 6:30] = np.round(f[5:12, 6:30] / (0.05 * scale) * 4) / 4 * scale           # many equal values: zero determinants
        f[20:36, 2:66] = np.round(f[20:36, 2:66] / (0.05 * scale)) * scale                  # (more per tile than its list holds)
        if scale == 3e6:
            f[:, :35] *= 1e-4                                                             # half the tiles small, half large
        f[15, 17] = np.nan
        steps.append(f)
    got = {}
    for fan in ("0", "1", "2"):
        monkeypatch.setenv("FTKX_TILE_FAN", fan)
        got[fan] = _run(gpu, None, dims, nt, steps=steps, exact_only=True)
    monkeypatch.delenv("FTKX_TILE_FAN")
    for fan in ("1", "2"):
        _same(got["0"][0], got[fan][0])
        assert got["0"][2] == got[fan][2]
        assert got["0"][1]["simplices_tested"] == got[fan][1]["simplices_tested"], fan
    assert len(got["0"][0]) > 50
    culled = _run(gpu, None, dims, nt, steps=steps)
    _same(culled[0], got["0"][0])


@pytest.mark.parametrize("scale", [1.0, 300.0, 3e5, 3e9], ids=["16bit", "32bit", "mixed", "64bit"])
def test_tile_kernel_forms_agree(gpu, monkeypatch, scale):
    """The 3D tile kernel has three forms of the same integer test: (corner, type) pairs over the lanes with 64-bit multiplies; one corner
    per lane with the fan's shared minors in registers, 32-bit operands; the same with 16-bit operands.  Which one a tile takes depends
    on the magnitudes in it: the same series at four scales (all three forms, tiles of different forms side by side, degenerate
    values from planted zeros, NaN vertices), swept under FTKX_TILE_FAN = 0 / 1 / 2, exact_only and with the in-tile cull: equal
    records, equal counts of simplices tested."""
    rng = np.random.default_rng(17)
    dims, nt = (40, 33, 21), 3
    steps = []
    for t in range(nt):
        f = rng.standard_normal(tuple(reversed(dims))) * 0.05 * scale
        f[4:9, 5:12, 6:14] = np.round(f[4:9, 5:12, 6:14] / (0.05 * scale) * 4) / 4 * scale      # many equal values: zero determinants
        f[12:20, 14:30, 2:38] = np.round(f[12:20, 14:30, 2:38] / (0.05 * scale)) * scale         # (thousands per tile: more than its list holds)
        if scale == 3e5:
            f[:, :, :20] *= 1e-4                                                     # half the tiles small, half large
        f[10, 11, 12] = np.nan
        steps.append(f)
    got = {}
    for fan in ("0", "1", "2"):
        monkeypatch.setenv("FTKX_TILE_FAN", fan)
        got[fan] = _run(gpu, None, dims, nt, steps=steps, exact_only=True)
    monkeypatch.delenv("FTKX_TILE_FAN")
    for fan in ("1", "2"):
        _same(got["0"][0], got[fan][0])
        assert got["0"][2] == got[fan][2]
        assert got["0"][1]["simplices_tested"] == got[fan][1]["simplices_tested"], fan
    assert len(got["0"][0]) > 50
    culled = _run(gpu, None, dims, nt, steps=steps)
    _same(culled[0], got["0"][0])


@pytest.mark.parametrize("nd", [2, 3])
def test_tile_kernel_walking_the_steps_equals_a_launch_per_step(gpu, monkeypatch, nd):
    """Round 6: one tile-kernel launch covers the consecutive tile requests of a batch; a workgroup keeps its tile for all of them and the
    slice two consecutive steps share is staged once (moved from LDS slot 1 to slot 0).  FTKX_TILE_WALK=0 is the old way, a launch per
    step with both slices staged.  Series whose factor changes half way (the shared slice is then staged again under the new factor), with
    planted zeros and a NaN, all three forms: equal records, factors and counts -- and equal to the culled pass."""
    rng = np.random.default_rng(31 + nd)
    dims, nt = ((40, 33, 21), 6) if nd == 3 else ((70, 41), 7)
    steps = []
    for t in range(nt):
        K = 16.0 if t < 3 else 4096.0        # values on a grid: the gradient's smallest non-zero magnitude is 1 / 2K -- nbits 8, then 13
        f = np.round(rng.standard_normal(tuple(reversed(dims))) * K) / K
        if nd == 3:
            f[10, 11, 12] = np.nan
        else:
            f[15, 17] = np.nan
        steps.append(f)
    got = {}
    for walk in ("1", "0"):
        for fan in ("0", "1", "2"):
            monkeypatch.setenv("FTKX_TILE_WALK", walk); monkeypatch.setenv("FTKX_TILE_FAN", fan)
            got[walk, fan] = _run(gpu, None, dims, nt, steps=steps, exact_only=True)
    monkeypatch.delenv("FTKX_TILE_WALK"); monkeypatch.delenv("FTKX_TILE_FAN")
    ref = got["0", "0"]
    assert len(ref[0]) > 50
    if nd == 3:
        assert len(set(int(v) for v in ref[2])) > 1, "the factor is meant to change inside the series (the shared slice is staged again)"
    for key, g in got.items():
        _same(ref[0], g[0])
        assert ref[2] == g[2] and ref[1]["simplices_tested"] == g[1]["simplices_tested"], key
    culled = _run(gpu, None, dims, nt, steps=steps)
    _same(culled[0], ref[0])


def test_cull_ahead_changes_nothing_but_the_schedule(gpu):
    """ftkx_sweep_announce: the cull queued behind the mask kernel, before the factors exist.  Same records and statistics as the
    plain order -- when the announcement is what gets swept, when it is not (other scopes, other steps, a subset: the list is
    dropped), when the masks cannot serve the factor after all (vertices that can overflow a determinant), on 2D, 3D and vector
    input, and over repeated passes on one context."""
    import torch
    from ftk_amd import synthetic, tslab
    cases = [("moving_extremum_3d", (96, 80, 64), 5, 1), ("woven", (512, 256), 6, 1), ("double_gyre", (256, 128), 5, 2), ("moving_extremum_3d", (31, 17, 9), 3, 1)]
    for case, dims, nt, nv in cases:
        a, sa, fa = _run(gpu, case, dims, nt, nv=nv)
        b, sb, fb = _run(gpu, case, dims, nt, nv=nv, announce="all")
        assert fa == fb and sa == sb, (case, sa, sb)
        _same(a, b)
        wrong = [[(t, gpu.SCOPE_ORDINAL) for t in range(nt)],                       # other scopes
                 [(t, gpu.SCOPE_BOTH) for t in range(nt - 1)],                      # one step short
                 [(t, gpu.SCOPE_BOTH if t + 1 < nt else gpu.SCOPE_ORDINAL) for t in reversed(range(nt))],   # another order
                 [(nt + 5, gpu.SCOPE_ORDINAL)]]                                    # a slice that is not resident
        for ann in wrong:
            c, sc, fc = _run(gpu, case, dims, nt, nv=nv, announce=ann)
            assert fc == fa and sc == sa
            _same(a, c)
    # masks built under the hint turn out unusable (big vertices): the cull-ahead is dropped, the sweep rebuilds them with the rule on
    dims, nt = (96, 80, 64), 4
    steps = [synthetic.generate("moving_extremum_3d", dims, t, nt, torch, torch.device("cuda", 0)).cpu().numpy() for t in range(nt)]
    rng = np.random.default_rng(5)
    for t in range(nt):
        for _ in range(4):
            k, j, i = (int(rng.integers(4, d - 4)) for d in reversed(dims))
            steps[t][k, j, i] += 3e6 * (1 + rng.random())
    a, sa, fa = _run(gpu, None, dims, nt, steps=steps)
    b, sb, fb = _run(gpu, None, dims, nt, steps=steps, announce="all")
    assert fa == fb and sa == sb
    _same(a, b)
    # repeated passes on one context, announced and not, with a slice replaced in between
    dims, nt = (128, 64, 48), 4
    dev = torch.device("cuda", 0)
    dom = ([2] * 3, [d - 3 for d in dims])
    ctx = gpu.Context(3)
    ctx.set_mesh(dom, dom, ([0] * 3, list(dims)))
    ctx.set_options(jacobian_symmetric=1, derive_jacobian=1, tag_mode=gpu.TAG_EXACT64)
    keep = [synthetic.generate("moving_extremum_3d", dims, t, nt, torch, dev) for t in range(nt)]
    torch.cuda.synchronize()
    for t in range(nt):
        ctx.push_scalar_slice(t, keep[t])
    scopes = [gpu.SCOPE_BOTH if t + 1 < nt else gpu.SCOPE_ORDINAL for t in range(nt)]

    def one(announce, touch=False):
        ctx.invalidate_masks()
        if announce:
            ctx.sweep_announce(range(nt), scopes)
        rm = ctx.slices_prepare(range(nt), 0)
        if touch:                   # a call that touches a slice between prepare and collect: the list must not be used
            ctx.push_scalar_slice(1, keep[1])
            rm.update(ctx.slices_prepare([1], 0))
        f = tslab.factors_from_resolutions([rm[t][0] for t in range(nt)])
        ctx.sweep_enqueue_many(range(nt), scopes, f)
        return ctx.sweep_collect(), ctx.stats()
    ref, sref = one(False)
    assert len(ref) > 0
    for announce, touch in ((True, False), (True, False), (False, False), (True, True), (True, False)):
        got, sg = one(announce, touch)
        assert sg == sref
        _same(ref, got)
    ctx.close()


@pytest.mark.parametrize("kind", ["smooth", "tiny"])
@pytest.mark.parametrize("halo", ["full", "masked"])
@pytest.mark.parametrize("announce", [False, True])
def test_announced_x_masked_halo_x_factor_change(gpu, kind, halo, announce):
    """The product the batched state machine has to get right: sweeps {announced before slices_prepare, not announced} x the slab's
    boundary slice {resident in full, resident as sign masks + patches only} x the factor {the hint the masks were built under (256),
    a larger one that only the other slices' reduction reveals (nbits 21)}.  Two contexts on one GPU play two ranks; the records of
    the first rank's steps must be those of one context that holds the whole series."""
    import torch
    from ftk_amd import tslab
    dims, nt, split = (72, 40, 24), 5, 3                      # rank A: steps 0 .. 2 (its last interval reads slice 3), rank B: slices 3, 4
    rng = np.random.default_rng(11)
    grids = np.meshgrid(*[np.linspace(-1.0, 1.0, n) for n in reversed(dims)], indexing="ij")
    steps = []
    for k in range(nt):
        a = np.exp(-3.0 * sum((g - 0.07 * k) ** 2 for g in grids)) + 0.2 * np.sin(3.0 * grids[-1] + 0.3 * k)
        if kind == "tiny" and k >= split:                     # tiny gradients arrive with rank B's slices: the factor of A's last step is not the hint's
            a = a + 1e-6 * rng.standard_normal(a.shape)
        steps.append(np.ascontiguousarray(a))
    dom = ([2] * 3, [d - 3 for d in dims])
    dev = torch.device("cuda", 0)

    def make():
        c = gpu.Context(3)
        c.set_mesh(dom, dom, ([0] * 3, list(dims)))
        c.set_options(jacobian_symmetric=1, derive_jacobian=1, tag_mode=gpu.TAG_EXACT64)
        return c
    # the whole series on one context
    whole = make()
    for t in range(nt):
        whole.push_scalar_slice(t, steps[t])
    rm = whole.slices_prepare(range(nt), 0)
    factors = tslab.factors_from_resolutions([rm[t][0] for t in range(nt)])
    scopes = [gpu.SCOPE_BOTH if t + 1 < nt else gpu.SCOPE_ORDINAL for t in range(nt)]
    whole.sweep_enqueue_many(range(nt), scopes, factors)
    ref = np.array(whole.sweep_collect())
    ref = ref[(ref["aux"] >> 1) < split]
    if kind == "tiny":
        assert factors[split - 1] > 256, factors               # the factor of A's boundary step really differs from the hint
    # two ranks
    A, B = make(), make()
    for t in range(split):
        A.push_scalar_slice(t, steps[t])
    for t in range(split, nt):
        B.push_scalar_slice(t, steps[t])
    rmB = B.slices_prepare(range(split, nt), 0)
    own = list(range(split))
    if announce:
        A.sweep_announce(own, [gpu.SCOPE_BOTH] * split)
    if halo == "full":
        A.push_scalar_slice(split, steps[split])
        A.set_slice_resolution(split, rm[split][0] if rm[split][0] < 1.0 / 256 else 1e300, rm[split][1])
        A.slices_prepare(own + [split], 0)
    else:
        A.slices_prepare(own, 0)
        nbytes, _cap = B.packed_masks_bytes()
        buf = torch.empty((nbytes,), dtype=torch.uint8, device=dev)
        B.export_masks_packed(split, buf)
        A.push_masked_slice_packed(split, True, buf, 256, rmB[split][1])
    A.sweep_enqueue_many(own, [gpu.SCOPE_BOTH] * split, [factors[t] for t in own])
    if halo == "masked":
        try:
            cells = A.sweep_cull(split, torch, dev)
            if len(cells):
                A.scatter_patches(split, cells, B.gather_patches(split, cells, torch))
        except gpu.FtkxError:                                  # the masks do not serve this factor: the slice itself, as the protocol does
            A.sweep_cancel()
            A.push_scalar_slice(split, steps[split])
            A.sweep_enqueue_many(own, [gpu.SCOPE_BOTH] * split, [factors[t] for t in own])
    got = np.array(A.sweep_collect())
    assert len(got) == len(ref) > 0, (kind, halo, announce, len(got), len(ref))
    assert got.tobytes() == ref.tobytes(), (kind, halo, announce)
    for c in (whole, A, B):
        c.close()


def _vector_field(rng, dims, kind):
    """a vector field (components last) with what the vector mask kernels treat specially"""
    nd = len(dims)
    shape = tuple(reversed(dims)) + (nd,)
    grids = np.meshgrid(*[np.linspace(-1.0, 1.0, n) for n in reversed(dims)], indexing="ij")
    v = np.stack([np.sin(2.1 * grids[(c + 1) % nd] + 0.3 * c) + 0.5 * np.cos(1.7 * grids[c]) for c in range(nd)], axis=-1)
    if kind == "rough":          # plateaus, zeros, values right at and just below the threshold 1 / 256, negative zero
        v = np.round(v * 4.0) * 0.25
        pick = rng.random(shape) < 0.08
        v[pick] = rng.choice([0.0, -0.0, 1.0 / 256, -1.0 / 256, 1.0 / 512, -1.0 / 1024, 3e-7, np.nextafter(1.0 / 256, 0.0), -np.nextafter(1.0 / 256, 0.0)], size=int(pick.sum()))
    elif kind == "odd":          # NaNs (quiet and signalling), infinities and values past `big`, alone and in clusters
        flat = v.reshape(-1)
        n = flat.size
        for val in (np.nan, np.inf, -np.inf, 1e300, -3e299):
            flat[rng.integers(0, n, size=5)] = val
        snan = np.frombuffer(np.array([0x7ff0000000000001], dtype=np.uint64).tobytes(), dtype=np.float64)[0]
        u = flat.view(np.uint64)
        u[rng.integers(0, n, size=4)] = np.array([snan]).view(np.uint64)[0]
        v[tuple(0 for _ in range(nd))] = np.nan                  # the slice's very first and very last vertex
        v[tuple(s - 1 for s in shape[:-1])] = np.inf
    return np.ascontiguousarray(v)


@pytest.mark.parametrize("dims,core", [((2048, 14), None), ((264, 42), None), ((512, 33), ((5, 3), (500, 29))), ((256, 17, 7), None), ((328, 9, 6), ((2, 1, 1), (320, 7, 4)))])
@pytest.mark.parametrize("kind", ["smooth", "rough", "odd"])
def test_vector_mask_kernels_agree(gpu, dims, core, kind):
    """mask_vec2_kernel (units of 4 rows x 64 groups, one summary byte per 8 x 4 block, wave-uniform addressing, one compare per component,
    the vertices one by one only where a wavefront meets the domain's border / a non-finite or big value) against mask_vec_kernel
    (FTKX_MASK_PLAN lean=0: one summary byte per word): the same mask words wherever both write them, block summaries = the AND of the word
    summaries, the same fused reduction, the same records and number of simplices tested -- rows of 64 groups exactly, rows whose last
    chunk is ragged, row counts that are no multiple of 4, a domain inside the array, fields with plateaus / zeros / values at the threshold
    and fields with NaNs, infinities and values past `big`."""
    import os
    import torch
    from ftk_amd import tslab
    nd, nt = len(dims), 3
    dev = torch.device("cuda", 0)
    rng = np.random.default_rng(5)
    steps = [_vector_field(rng, dims, kind) for _ in range(nt)]
    dom = core or ([1] * nd, [d - 2 for d in dims])
    out = {}
    for lean in ("0", "1"):
        old = os.environ.get("FTKX_MASK_PLAN")
        os.environ["FTKX_MASK_PLAN"] = "lean=" + lean
        try:
            ctx = gpu.Context(nd)
            ctx.set_mesh(dom, dom, ([0] * nd, list(dims)))
            ctx.set_options(jacobian_symmetric=0, derive_jacobian=1, tag_mode=gpu.TAG_EXACT64)
            keep = [torch.from_numpy(s).to(dev) for s in steps]
            for t in range(nt):
                ctx.push_slice(t, keep[t])
            rm = ctx.slices_prepare(range(nt), 0)
            name = (ctx._L.ftkx_last_mask_kernel() or b"").decode()
            # the sweep first: where a slice holds big values the masks of slices_prepare do not stand and the sweep builds them again, under
            # its factor and with the per-vertex rule (job.big finite) -- those are then the masks exported below
            res = [rm[t][0] for t in range(nt)]
            factors = tslab.factors_from_resolutions(res)
            for t in range(nt):
                ctx.sweep_enqueue(t, gpu.SCOPE_BOTH if t + 1 < nt else gpu.SCOPE_ORDINAL, factors[t])
            recs = np.array(ctx.sweep_collect())
            st = ctx.stats()
            name2 = (ctx._L.ftkx_last_mask_kernel() or b"").decode()
            nbytes, cap = ctx.packed_masks_bytes()
            packed = []
            for t in range(nt):
                buf = torch.zeros((nbytes,), dtype=torch.uint8, device=dev)
                try:
                    ctx.export_masks_packed(t, buf)
                except gpu.FtkxError as e:      # (no summarised masks for this slice: then for neither kernel)
                    packed.append(e.code)
                    continue
                torch.cuda.synchronize()
                h = buf.cpu().numpy()
                head = h[:32].view(np.uint64)
                count, ub = int(head[0]), int(head[1])
                assert count <= cap, (count, cap)
                off_idx = 32 + (ub + 7) // 8 * 8
                off_words = off_idx + (cap * 4 + 7) // 8 * 8
                idx = h[off_idx:off_idx + 4 * count].view(np.uint32)
                words = h[off_words:off_words + 8 * count].view(np.uint64)
                order = np.argsort(idx, kind="stable")
                packed.append((int(head[2] >> 48) & 0xff, h[32:32 + ub].copy(), idx[order].copy(), words[order].copy()))
            assert name2 == name, (name, name2)
            ctx.close()
        finally:
            if old is None:
                os.environ.pop("FTKX_MASK_PLAN", None)
            else:
                os.environ["FTKX_MASK_PLAN"] = old
        out[lean] = (name, {t: (np.float64(a).tobytes(), np.float64(b).tobytes()) for t, (a, b) in rm.items()}, packed, recs, st)
    assert "mask_vec_kernel" in out["0"][0] and "mask_vec2_kernel" in out["1"][0], (out["0"][0], out["1"][0])
    assert out["0"][1] == out["1"][1], "fused reduction"
    # the masks: mask_vec_kernel writes one summary byte per word of 8 vertices (the AND of its mask bytes) and the words whose summary is 0;
    # mask_vec2_kernel one byte per 8 x 4 block and ALL words of the blocks whose summary is 0
    DW, DH, DD = dims[0], dims[1], (dims[2] if nd == 3 else 1)
    PW, UP, ncols, nby = ((DW + 7) // 8 * 8 + 8) // 8, ((DW + 7) // 8 + 7) // 8 * 8 + 8, DW // 8, (DH + 3) // 4
    for t in range(nt):
        a, b = out["0"][2][t], out["1"][2][t]
        if isinstance(a, int) or isinstance(b, int):
            assert a == b, (t, a, b)
            continue
        assert (a[0], b[0]) == (1, 4), (a[0], b[0])
        Ua, Ub = a[1].reshape(DD, DH, UP)[:, :, :ncols], b[1].reshape(DD, nby, UP)[:, :, :ncols]
        pad = np.full((DD, nby * 4, ncols), 0x3f, dtype=np.uint8)
        pad[:, :DH] = Ua
        assert np.array_equal(np.bitwise_and.reduce(pad.reshape(DD, nby, 4, ncols), axis=2), Ub), (t, "block summaries")
        wa, wb = dict(zip(a[2].tolist(), a[3].tolist())), dict(zip(b[2].tolist(), b[3].tolist()))
        assert len(wa) == len(a[2]) and len(wb) == len(b[2])
        for w, val in wa.items():
            assert wb.get(w) == val, (t, "a word both kernels write", w, val, wb.get(w))
        flatUa = Ua.reshape(DD * DH, ncols)
        for w, val in wb.items():
            row, g = divmod(w, PW)
            bytes8 = np.frombuffer(np.uint64(val).tobytes(), dtype=np.uint8)
            assert g < ncols and int(np.bitwise_and.reduce(bytes8)) == int(flatUa[row, g]), (t, "a word only the block kernel writes", w)
            assert Ub.reshape(DD * nby, ncols)[(row // DH) * nby + (row % DH) // 4, g] == 0
    assert out["0"][3].tobytes() == out["1"][3].tobytes()
    assert out["0"][4]["simplices_tested"] == out["1"][4]["simplices_tested"]
    # (cells_survived counts coarse cells of the cull -- words of 8 vertices there, blocks of 8 x 4 here: not comparable)
    if kind == "smooth":
        assert len(out["1"][3]) > 0 and not any(isinstance(x, int) for x in out["1"][2])
