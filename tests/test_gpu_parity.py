"""GPU parity tests proper: the HIP path, called through the C ABI, against
  (a) the golden fixtures captured from the real reference CPU path (tests/golden), and
  (b) the oracle on the same inputs.
Bar: record sets identical; tag / type / ordinal / timestep exact; x, t, scalar within 1e-6 (north_star) -- and we additionally
assert they are BIT-identical, which holds because the FP64 hit path uses the same IEEE operations in the same order."""
import numpy as np
import pytest

from common import assert_records_equal, golden_names, load_golden, wrap_golden_names

pytestmark = pytest.mark.gpu

COORD_TOL = 1e-6   # BASELINE.json north_star: "within 1e-6 on interpolated (x,y,z,t)"


@pytest.fixture(scope="module")
def gpu():
    import torch
    assert torch.cuda.is_available(), "these tests need the MI355X"
    import ftk_amd
    from ftk_amd import build
    build.build()
    return ftk_amd


@pytest.mark.parametrize("exact_only", [False, True], ids=["cull", "exact_only"])
@pytest.mark.parametrize("name", golden_names())
def test_tracker_matches_reference_fixture(gpu, name, exact_only):
    from gpu_common import run_tracker
    g = load_golden(name)
    recs, factors, stats = run_tracker(g["steps"], g["nd"], g["nv"], robust=g["robust"], type_filter=g["type_filter"], exact_only=exact_only,
                                       compute_degrees=g["degrees"], bounds=g["bounds"], rectilinear=g["rectilinear"], explicit=g["explicit"])
    assert np.array_equal(factors, g["factors"]), "per-step quantisation factor differs from the reference"
    assert_records_equal(recs, g["records"], coord_tol=COORD_TOL, what=name)
    assert_records_equal(recs, g["records"], coord_tol=0.0, what=name + " (bit-exact)")
    if exact_only:
        assert all(s["cull_enabled"] == 0 for s in stats)


@pytest.mark.parametrize("exact_only", [False, True], ids=["cull", "exact_only"])
@pytest.mark.parametrize("name", wrap_golden_names())
def test_reference_tags_and_vertex_ids_where_int32_wraps(gpu, name, exact_only):
    """FTKX_TAG_REFERENCE on a mesh whose tag products exceed 2^31 (simplicial_regular_mesh.hh:496-502) and whose SoS vertex ids
    wrap when truncated to int (regular_tracker.hh:188-194): fixtures from the real reference started at a large timestep."""
    from gpu_common import run_tracker
    g = load_golden(name)
    recs, factors, stats = run_tracker(g["steps"], g["nd"], g["nv"], exact_only=exact_only, tag_mode=gpu.TAG_REFERENCE, t0=g["t0"])
    assert np.array_equal(factors, g["factors"])
    assert_records_equal(recs, g["records"], coord_tol=0.0, what=name)
    # and the 64-bit tags of the same run differ: the comparison above really exercised the wrap
    exact, _, _ = run_tracker(g["steps"], g["nd"], g["nv"], tag_mode=gpu.TAG_EXACT64, t0=g["t0"])
    assert len(exact) == len(recs) and not np.array_equal(np.sort(exact["tag"]), np.sort(recs["tag"]))


@pytest.mark.parametrize("name", ["woven_31x37x32", "moving_extremum_3d_21x21x21x32", "random_3d_scalar_13x12x11x4"])
def test_tracker_with_device_resident_input(gpu, name):
    from gpu_common import run_tracker
    g = load_golden(name)
    recs, factors, _ = run_tracker(g["steps"], g["nd"], g["nv"], device=True)
    assert_records_equal(recs, g["records"], coord_tol=0.0, what=name)


def test_cull_is_actually_on_where_it_is_legal(gpu):
    from gpu_common import run_tracker
    g = load_golden("moving_extremum_3d_32x32x32x8_dyadic")
    _, _, stats = run_tracker(g["steps"], 3, 1)
    assert all(s["cull_enabled"] == 1 for s in stats)
    assert sum(s["simplices_tested"] for s in stats) < 0.05 * sum(s["work_items"] for s in stats)
    # nbits 21, M = 2^25: determinants wrap and the wrapped sign is the reference's answer (33 662 mostly bogus records).  The cull
    # stays formally on, but every vertex is "big" (|q| + 1 > safe_m, sweep_params.hpp), carries no sign bits and supports no
    # cull: all but the cells with a vertex outside ... no cell at all may be dropped
    g = load_golden("moving_extremum_3d_21x21x21x4_overflow")
    recs, _, stats = run_tracker(g["steps"], 3, 1)
    assert all(s["cull_enabled"] == 1 for s in stats)
    assert all(s["cells_survived"] == s["cells"] for s in stats)
    assert len(recs) == 33662


@pytest.mark.parametrize("name", ["woven_31x37x32", "double_gyre_64x32x50", "adversarial_2d_vector_15x12x5",
                                  "moving_extremum_3d_12x10x9x5_aligned", "adversarial_3d_scalar_9x9x9x4", "adversarial_3d_vector_8x8x8x3"])
def test_boundary_call_with_given_fields(gpu, oracle, name):
    """extract_cp{2,3}dt with the reference boundary's argument list: host V/J/S given, tag = work index inside core."""
    g = load_golden(name)
    nd, nv, D = g["nd"], g["nv"], g["dims"]
    lo = 2 if nv == 1 else 1
    dom = ([lo] * nd, [d - (3 if nv == 1 else 2) for d in D])
    fields = []
    for k in (0, 1):
        a = g["steps"][k]
        if nv == 1:
            V = oracle.gradient2D(a) if nd == 2 else oracle.gradient3D(a)
            J = oracle.jacobian2D(V, True) if nd == 2 else oracle.jacobian3D(V)
            fields.append((V, J, a))
        else:
            J = oracle.jacobian2D(a, False) if nd == 2 else oracle.jacobian3D(a)
            fields.append((a, J, None))
    res = min(oracle.resolution(fields[0][0]), oracle.resolution(fields[1][0]))
    factor, _ = oracle.scaling_factor(res)
    opt = gpu.default_options(jacobian_symmetric=(nv == 1), tag_mode=gpu.TAG_WORK_INDEX)
    for scope in (gpu.SCOPE_ORDINAL, gpu.SCOPE_INTERVAL):
        ref = oracle.sweep(nd, scope, 0, dom, dom, ([0] * nd, D), (fields[0][0], fields[1][0]), (fields[0][1], fields[1][1]),
                           (fields[0][2], fields[1][2]) if nv == 1 else None, factor, jacobian_symmetric=(nv == 1),
                           tag_mode=oracle.TAG_WORK_INDEX)
        f = gpu.extract_cp2dt if nd == 2 else gpu.extract_cp3dt
        nxt = scope == gpu.SCOPE_INTERVAL
        got = f(scope, 0, (dom[0] + [0], dom[1] + [2 ** 31 - 1]), (dom[0] + [0], dom[1] + [1]), ([0] * nd, D),
                fields[0][0], fields[1][0] if nxt else None, fields[0][1], fields[1][1] if nxt else None,
                fields[0][2], fields[1][2] if nxt else None, factor, opt)
        refr = np.zeros(len(ref), dtype=got.dtype)
        for fld in ("x", "t", "scalar", "type", "tag"):
            refr[fld] = ref[fld]
        assert_records_equal(got, refr, coord_tol=0.0, what=f"{name} scope {scope}")
        assert np.array_equal(got["aux"] & 1, np.full(len(got), 1 if scope == gpu.SCOPE_ORDINAL else 0))


def test_boundary_call_with_explicit_coords(gpu, oracle):
    """extract_cp2dt(..., use_explicit_coords = true, coords): the boundary's (2, DW, DH) vertex coordinates"""
    g = load_golden("random_2d_scalar_29x24x6_explicit2")
    D = g["dims"]
    dom = ([2, 2], [d - 3 for d in D])
    S0, S1 = g["steps"][0], g["steps"][1]
    V0, V1 = oracle.gradient2D(S0), oracle.gradient2D(S1)
    J0, J1 = oracle.jacobian2D(V0, True), oracle.jacobian2D(V1, True)
    factor, _ = oracle.scaling_factor(min(oracle.resolution(V0), oracle.resolution(V1)))
    opt = gpu.default_options(jacobian_symmetric=1, tag_mode=gpu.TAG_WORK_INDEX)
    for scope in (gpu.SCOPE_ORDINAL, gpu.SCOPE_INTERVAL):
        ref = oracle.sweep(2, scope, 0, dom, dom, ([0, 0], D), (V0, V1), (J0, J1), (S0, S1), factor, jacobian_symmetric=True,
                           tag_mode=oracle.TAG_WORK_INDEX, explicit=g["explicit"])
        nxt = scope == gpu.SCOPE_INTERVAL
        got = gpu.extract_cp2dt(scope, 0, (dom[0] + [0], dom[1] + [2 ** 31 - 1]), (dom[0] + [0], dom[1] + [1]), ([0, 0], D),
                                V0, V1 if nxt else None, J0, J1 if nxt else None, S0, S1 if nxt else None, factor, opt, coords=g["explicit"])
        refr = np.zeros(len(ref), dtype=got.dtype)
        for fld in ("x", "t", "scalar", "type", "tag"):
            refr[fld] = ref[fld]
        assert len(got) > 20
        assert_records_equal(got, refr, coord_tol=0.0, what=f"explicit coords, scope {scope}")
        assert not np.array_equal(got["x"][:, 0], np.round(got["x"][:, 0]))      # really not lattice coordinates


def test_derived_fields_bit_identical(gpu, oracle):
    import torch
    rng = np.random.default_rng(5)
    ctx = gpu.Context(2)
    S = rng.standard_normal((37, 53)); V = rng.standard_normal((37, 53, 2))
    dS = torch.from_numpy(S).cuda(); dV = torch.empty((37, 53, 2), dtype=torch.float64, device="cuda")
    ctx.gradient2D(dS.data_ptr(), 53, 37, dV.data_ptr())
    assert np.array_equal(dV.cpu().numpy(), oracle.gradient2D(S))
    for sym in (True, False):
        dVin = torch.from_numpy(V).cuda(); dJ = torch.zeros((37, 53, 2, 2), dtype=torch.float64, device="cuda")
        torch.cuda.synchronize()
        ctx.jacobian2D(dVin.data_ptr(), 53, 37, sym, dJ.data_ptr())
        assert np.array_equal(dJ.cpu().numpy(), oracle.jacobian2D(V, sym)), f"jacobian2D symmetric={sym}"
    ctx.close()
    ctx = gpu.Context(3)
    S = rng.standard_normal((11, 13, 17)); V = rng.standard_normal((11, 13, 17, 3))
    dS = torch.from_numpy(S).cuda(); dV = torch.empty((11, 13, 17, 3), dtype=torch.float64, device="cuda")
    ctx.gradient3D(dS.data_ptr(), 17, 13, 11, dV.data_ptr())
    assert np.array_equal(dV.cpu().numpy(), oracle.gradient3D(S))
    dVin = torch.from_numpy(V).cuda(); dJ = torch.empty((11, 13, 17, 3, 3), dtype=torch.float64, device="cuda")
    ctx.jacobian3D(dVin.data_ptr(), 17, 13, 11, dJ.data_ptr())
    assert np.array_equal(dJ.cpu().numpy(), oracle.jacobian3D(V))
    ctx.close()


def test_slice_resolution_matches_ndarray_resolution(gpu, oracle):
    rng = np.random.default_rng(9)
    V = rng.standard_normal((19, 23, 2)) * 10
    V[3, 4, 0] = 0.0; V[5, 6, 1] = np.nan; V[7, 8, 0] = np.inf; V[9, 9, 1] = 1e-11; V[1, 1, 0] = -4e3
    ctx = gpu.Context(2)
    ctx.set_mesh(([1, 1], [21, 17]), ([1, 1], [21, 17]), ([0, 0], [23, 19]))
    ctx.push_slice(0, V)
    res, mx = ctx.slice_resolution(0)
    assert res == oracle.resolution(V) == 1e-11
    assert mx == 4e3
    ctx.close()


def test_error_behaviour(gpu):
    ctx = gpu.Context(2)
    with pytest.raises(gpu.FtkxError) as e:
        ctx.push_slice(0, np.zeros((4, 4, 2)))
    assert e.value.code == -1                 # set_mesh first
    ctx.set_mesh(([1, 1], [2, 2]), ([1, 1], [2, 2]), ([0, 0], [4, 4]))
    with pytest.raises(gpu.FtkxError) as e:
        ctx.sweep(0, gpu.SCOPE_ORDINAL, 256)
    assert e.value.code == -4                 # slice not resident
    ctx.push_slice(0, np.ones((4, 4, 2)))
    with pytest.raises(gpu.FtkxError):
        ctx.sweep(0, gpu.SCOPE_INTERVAL, 256)  # needs slice 1
    assert len(ctx.sweep(0, gpu.SCOPE_ORDINAL, 256)) == 0
    ctx.close()


def test_device_generators_match_reference_generators(gpu, oracle):
    """bench.py generates its inputs in HBM (ftk_amd/synthetic.py); moving_extremum with the dyadic bench parameters must be
    bit-identical to the reference generator, the transcendental cases equal to rounding."""
    import torch
    from ftk_amd import synthetic
    dev = torch.device("cuda", 0)
    dims = (40, 36, 32)
    x0, dv = synthetic.moving_extremum_params(dims)
    for k in (0, 3, 7):
        a = synthetic.generate("moving_extremum_3d", dims, k, 8, torch, dev).cpu().numpy()
        assert np.array_equal(a, oracle.synthetic("moving_extremum_3d", list(dims), k, 8, x0, dv))
    a = synthetic.generate("woven", (50, 41), 3, 9, torch, dev).cpu().numpy()
    assert np.allclose(a, oracle.synthetic("woven", [50, 41], 3, 9), rtol=0, atol=1e-13)
    a = synthetic.generate("double_gyre", (48, 24), 5, 9, torch, dev).cpu().numpy()
    assert np.allclose(a, oracle.synthetic("double_gyre", [48, 24], 5, 9), rtol=0, atol=1e-13)


def _batched_context_run(gpu, steps, nd, nv, dims):
    """all timesteps resident, every sweep enqueued, ONE collect (what bench.py does): exercises the batched launches, the
    sub-batching on factor changes and the device-side sort of the hit buffer"""
    from ftk_amd import tslab
    scalar = nv == 1
    lo = 2 if scalar else 1
    dom = ([lo] * nd, [d - (3 if scalar else 2) for d in dims])
    ctx = gpu.Context(nd)
    ctx.set_mesh(dom, dom, ([0] * nd, list(dims)))
    ctx.set_options(jacobian_symmetric=scalar, derive_jacobian=1, tag_mode=gpu.TAG_REFERENCE)
    nt = len(steps)
    res = []
    for t in range(nt):
        (ctx.push_scalar_slice if scalar else ctx.push_slice)(t, steps[t])
        res.append(ctx.slice_resolution(t)[0])
    factors = tslab.factors_from_resolutions(res)
    for t in range(nt):
        ctx.sweep_enqueue(t, gpu.SCOPE_BOTH if t + 1 < nt else gpu.SCOPE_ORDINAL, factors[t])
    recs = ctx.sweep_collect()
    st = ctx.stats()
    ctx.close()
    out = np.zeros(len(recs), dtype=[("tag", "<u8"), ("type", "<u4"), ("ordinal", "<i4"), ("timestep", "<i4"),
                                     ("x", "<f8", (3,)), ("t", "<f8"), ("scalar", "<f8", (3,))])
    for f in ("tag", "type", "x", "t", "scalar"):
        out[f] = recs[f]
    out["ordinal"] = recs["aux"] & 1
    out["timestep"] = recs["aux"] >> 1
    return out, factors, st


@pytest.mark.parametrize("name", ["woven_31x37x32", "merger_2d_32x32x100", "double_gyre_64x32x50", "random_2d_scalar_29x24x6",
                                  "moving_extremum_3d_21x21x21x32", "random_3d_scalar_13x12x11x4", "moving_extremum_3d_21x21x21x4_overflow"])
def test_batched_context_matches_reference_fixture(gpu, name):
    g = load_golden(name)
    recs, factors, st = _batched_context_run(gpu, g["steps"], g["nd"], g["nv"], g["dims"])
    assert [int(f) for f in g["factors"]] == factors
    assert_records_equal(recs, g["records"], coord_tol=0.0, what=name)


def test_larger_2d_batch_vs_oracle(gpu, oracle):
    """woven 256x256x8: ~15k hits in one collect (device radix sort path), factor changes inside the batch"""
    dims, nt = (256, 256), 8
    steps = [oracle.synthetic("woven", list(dims), k, nt) for k in range(nt)]
    ref, rf, _ = oracle.track(steps, 2, 1, tag_mode=oracle.TAG_REFERENCE, nthreads=8)
    recs, factors, st = _batched_context_run(gpu, steps, 2, 1, dims)
    assert factors == [int(f) for f in rf]
    assert len(ref) > 4096
    assert_records_equal(recs, ref, coord_tol=0.0, what="woven 256x256x8")
    assert st["cull_enabled"] == 1 and st["simplices_tested"] < 0.1 * st["work_items"]


def test_larger_3d_vs_oracle_with_odd_sizes(gpu, oracle):
    """sizes that are not multiples of the tile / word sizes, extremum near a corner of the domain"""
    dims, nt = (45, 38, 27), 5
    steps = [oracle.synthetic("moving_extremum_3d", list(dims), k, nt, [40.25, 3.375, 22.125], [0.5, 0.25, 0.125]) for k in range(nt)]
    ref, rf, _ = oracle.track(steps, 3, 1, tag_mode=oracle.TAG_REFERENCE, nthreads=8)
    recs, factors, st = _batched_context_run(gpu, steps, 3, 1, dims)
    assert factors == [int(f) for f in rf]
    assert_records_equal(recs, ref, coord_tol=0.0, what="moving_extremum 45x38x27x5")
    rng = np.random.default_rng(3)
    steps = [np.cumsum(rng.standard_normal((19, 22, 35)), axis=2) * 0.125 for _ in range(3)]
    ref, rf, _ = oracle.track(steps, 3, 1, tag_mode=oracle.TAG_REFERENCE, nthreads=8)
    recs, factors, st = _batched_context_run(gpu, steps, 3, 1, (35, 22, 19))
    assert factors == [int(f) for f in rf]
    assert_records_equal(recs, ref, coord_tol=0.0, what="random walk 35x22x19x3")


@pytest.mark.parametrize("name", ["woven_31x37x32", "merger_2d_32x32x100", "double_gyre_64x32x50", "moving_extremum_3d_21x21x21x32",
                                  "random_3d_scalar_13x12x11x4"])
def test_tracker_end_to_end_curves(gpu, name):
    """sweep on the GPU + finalize(): the traced curves are the reference's (56 for the woven test of the reference itself)"""
    from gpu_common import run_tracker
    g = load_golden(name)
    recs, factors, stats, (curves, loop) = run_tracker(g["steps"], g["nd"], g["nv"], want_curves=True)
    got = sorted((tuple(c.tolist()), int(l)) for c, l in zip(curves, loop))
    exp = sorted((tuple(t.tolist()), int(l)) for l, t in g["curves"])
    assert got == exp
    if name == "woven_31x37x32":
        assert len(curves) == 56      # tests/test_critical_point_tracking_woven.cpp:32-37


@pytest.mark.parametrize("nd", [2, 3])
def test_empty_and_minimal_inputs(gpu, oracle, nd):
    """edge cases around nothing: an empty core, arrays so small that the domain [2, D-2] is empty or a single cell, a single
    timestep (ordinal sweep only) -- the tracker must agree with the oracle (usually: no records) and never fail"""
    from gpu_common import run_tracker
    rng = np.random.default_rng(11)
    ctx = gpu.Context(nd)
    ctx.set_mesh(([2] * nd, [4] * nd), ([2] * nd, [0] + [4] * (nd - 1)), ([0] * nd, [8] * nd))      # core of size 0 along x
    ctx.push_scalar_slice(0, rng.standard_normal((8,) * nd))
    assert len(ctx.sweep(0, gpu.SCOPE_ORDINAL, 256)) == 0
    ctx.close()
    for D in (3, 4, 5, 6):
        for nt in (1, 2, 3):
            steps = [rng.standard_normal((D,) * nd) for _ in range(nt)]
            ref, rfac, _ = oracle.track(steps, nd, 1, tag_mode=oracle.TAG_REFERENCE)
            got, gfac, _ = run_tracker(steps, nd, 1)
            assert len(got) == len(ref), (D, nt)
            if D == 3:
                assert len(got) == 0                            # the domain has no vertices
            if len(ref):
                assert np.array_equal(np.sort(got["tag"]), np.sort(ref["tag"]))
                assert np.array_equal(gfac, rfac)


@pytest.mark.parametrize("name", ["woven_31x37x32", "moving_extremum_3d_21x21x21x32", "merger_2d_32x32x100", "double_gyre_64x32x50",
                                  "random_3d_scalar_13x12x11x4", "moving_extremum_3d_21x21x21x4_overflow"])
@pytest.mark.parametrize("ndev,block", [(2, 1), (2, 3), (3, 2), (4, 1)])
def test_multi_device_tracker_matches_reference_fixture(gpu, name, ndev, block):
    """ONE tracker over several devices (here the same GPU listed ndev times: own context, stream and host thread each): timesteps
    dealt in blocks, steps queued and swept concurrently, the sticky factor formed from reductions published across devices.
    Records and final factor identical to the reference's sequential run; with a sync after every step also the factor sequence."""
    from gpu_common import run_tracker
    g = load_golden(name)
    recs, factors, _ = run_tracker(g["steps"], g["nd"], g["nv"], device_ids=[0] * ndev, block=block, factor_each_step=False)
    assert factors[g["DT"] - 1] == g["factors"][g["DT"] - 1]
    assert_records_equal(recs, g["records"], coord_tol=0.0, what=f"{name} on {ndev} devices, block {block}")
    if ndev == 2:
        recs, factors, _ = run_tracker(g["steps"], g["nd"], g["nv"], device_ids=[0] * ndev, block=block, device=True, want_curves=False)
        assert np.array_equal(factors, g["factors"])
        assert_records_equal(recs, g["records"], coord_tol=0.0, what=name)


def test_streaming_trajectories_equal_the_reference(gpu):
    """enable_streaming_trajectories end to end on the GPU tracker: trajectories grown after every interval sweep from the HIP
    sweep's records == the trajectories the real reference grew in the same mode (tests/golden/streaming_*.npz): same order of birth,
    same point sequences, same loop flags; the last step's ordinal points stay behind as discrete points; post_process then gives the
    reference's trajectory count."""
    from gpu_common import run_tracker
    from common import streaming_golden_names, load_streaming_golden
    for name in streaming_golden_names():
        g, sg = load_golden(name), load_streaming_golden(name)
        state = {}

        def after(tr):
            tr.finalize()
            state["curves"] = tr.get_traced_critical_points()
            tr.post_process()
            state["pp"] = len(tr.get_traced_critical_points()[0])
        left, factors, _ = run_tracker(g["steps"], g["nd"], g["nv"], streaming=True, after=after)
        assert np.array_equal(factors, g["factors"])
        curves, loop = state["curves"]
        got = [(tuple(c.tolist()), int(l)) for c, l in zip(curves, loop)]
        exp = [(tuple(t.tolist()), int(l)) for l, t in sg["curves"]]
        assert got == exp, name
        assert np.array_equal(np.sort(left["tag"]), sg["leftover_tags"]), name
        assert state["pp"] == sg["pp_count"], name
