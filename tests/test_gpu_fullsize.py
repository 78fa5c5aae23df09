"""BASELINE.json's configurations at their FULL sizes (C1 is the golden fixture woven_128x128x10 and
C3 is tests/test_gpu_properties.py::test_c3_moving_extremum_256cubed_analytic).  Where the host has the cores for it the oracle
itself is the checker (C2: 7.9e8 simplices, C5: 3.2e9 -- seconds on the GPU box's 256 hardware threads); everywhere the
size-independent properties are asserted: cull == exact_only, analytic trajectory, sorted unique tags, types, factor sequence."""
import os

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


def _sweep_series(gpu, case, dims, nt, *, nv=1, exact_only=False, core=None, keep_host=False, tag_mode=None, prepass="fused"):
    """whole series resident, one pass for masks + reduction (the product's path; prepass="exact": the separate exact pre-pass), one
    batched sweep -> (records, stats, factors, host copies or None)"""
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
    keep, host = [], []
    for t in range(nt):
        a = synthetic.generate(case, dims, t, nt, torch, dev)
        torch.cuda.synchronize()
        keep.append(a)
        if keep_host:
            host.append(a.cpu().numpy())
        (ctx.push_scalar_slice if scalar else ctx.push_slice)(t, a)
    rm = ctx.slices_prepare(range(nt), 0) if prepass == "fused" else ctx.slices_resolution(range(nt))
    factors = tslab.factors_from_resolutions([rm[t][0] for t in range(nt)])
    for t in range(nt):
        ctx.sweep_enqueue(t, gpu.SCOPE_BOTH if t + 1 < nt else gpu.SCOPE_ORDINAL, factors[t])
    recs = ctx.sweep_collect()
    st = ctx.stats()
    ctx.close()
    del keep
    torch.cuda.empty_cache()
    return recs, st, factors, (host if keep_host else None)


def _same(a, b):
    assert len(a) == len(b)
    for f in ("tag", "type", "aux"):
        assert np.array_equal(a[f], b[f]), f
    for f in ("x", "t", "scalar"):
        assert np.array_equal(a[f], b[f], equal_nan=True), f


def _assert_equals_oracle(oracle, recs, host_steps, nd, nv, factors, what):
    """the oracle on the very arrays the GPU swept (generated on the device: torch's sin / cos are not libm's)"""
    from gpu_common import oracle_track_cached
    ref, rf, secs = oracle_track_cached(oracle, host_steps, nd, nv)      # (sorted by tag; shared with tests/test_gpu_fullsize_series.py)
    assert [int(f) for f in rf] == [int(f) for f in factors], what
    assert len(ref) == len(recs), (what, len(ref), len(recs))
    assert np.array_equal(ref["tag"], recs["tag"]) and np.array_equal(ref["type"], recs["type"]), what
    assert np.array_equal(ref["ordinal"].astype(np.uint32), recs["aux"] & 1) and np.array_equal(ref["timestep"].astype(np.uint32), recs["aux"] >> 1), what
    for f in ("x", "t"):
        assert np.array_equal(ref[f], recs[f]), (what, f)       # bit-identical (north_star asks for 1e-6)
    assert np.array_equal(ref["scalar"][:, 0], recs["scalar"][:, 0]), what
    return secs


def test_c4_moving_extremum_512cubed_x32(gpu):
    """BASELINE configs[3] = the configuration the metric is quoted on: 512^3 x 32, 2.46e11 simplices, 34 GB of S resident."""
    from ftk_amd import synthetic, tslab
    dims, nt = (512, 512, 512), 32
    recs, st, factors, _ = _sweep_series(gpu, "moving_extremum_3d", dims, nt)
    assert st["work_items"] == tslab.count_simplices(3, dims, nt) == 246073579314 and st["cull_enabled"] == 1
    assert set(factors) == {256}                                                       # dyadic parameters: nbits 8
    x0, dv = synthetic.moving_extremum_params(dims)
    assert len(recs) >= 2 * nt - 1 and set(recs["type"].tolist()) == {2}               # one minimum, every record a MIN
    for a in range(3):
        assert np.abs(recs["x"][:, a] - (x0[a] + dv[a] * recs["t"])).max() < 1e-6      # north_star tolerance (observed ~1e-13)
    assert np.all(np.diff(recs["tag"].astype(np.uint64)) > 0)                         # sorted, unique
    ordinal = recs[(recs["aux"] & 1) == 1]
    assert np.array_equal(ordinal["t"], (ordinal["aux"] >> 1).astype(float))
    assert len(np.unique(ordinal["aux"] >> 1)) == nt                                   # the trajectory crosses every slice
    assert np.array_equal(np.unique(recs["aux"] >> 1), np.arange(nt))
    # the extremum moves from x0 to x0 + 31 dir: a core of 64 x 64 x 64 cells around the path, swept WITHOUT the cull (every
    # simplex through the integer test), must give exactly the records of the culled sweep of the whole domain
    lo = [int(x0[a]) - 24 for a in range(3)]
    sub, st_e, _, _ = _sweep_series(gpu, "moving_extremum_3d", dims, nt, exact_only=True, core=(lo, [64, 64, 64]))
    assert st_e["cull_enabled"] == 0 and st_e["simplices_tested"] > 1000 * st["simplices_tested"]
    c = recs["x"]
    assert all(lo[a] <= c[:, a].min() and c[:, a].max() < lo[a] + 64 for a in range(3))
    _same(recs, sub)


def test_c2_woven_1024x1024x64(gpu, oracle):
    """BASELINE configs[1]: woven 1024 x 1024 x 64, 7.9e8 simplices, ~62 000 records"""
    from ftk_amd import tslab
    dims, nt = (1024, 1024), 64
    recs, st, factors, host = _sweep_series(gpu, "woven", dims, nt, keep_host=True)
    assert st["work_items"] == tslab.count_simplices(2, dims, nt) == 790170278 and st["cull_enabled"] == 1
    assert len(recs) > 50000 and set(recs["type"].tolist()) <= {1, 2, 4, 8}
    assert np.all(np.diff(recs["tag"].astype(np.uint64)) > 0)
    assert np.array_equal(np.unique(recs["aux"] >> 1), np.arange(nt))
    recs_e, st_e, _, _ = _sweep_series(gpu, "woven", dims, nt, exact_only=True)
    assert st_e["cull_enabled"] == 0 and st_e["simplices_tested"] > 100 * st["simplices_tested"]
    _same(recs, recs_e)
    recs_p, st_p, f_p, _ = _sweep_series(gpu, "woven", dims, nt, prepass="exact")      # masks under the true factors: fewer survivors, same records
    assert f_p == factors and st_p["cells_survived"] <= st["cells_survived"]
    _same(recs, recs_p)
    if (os.cpu_count() or 1) >= 32:     # ~2 s of the oracle on the GPU box's host; hours-long nowhere, but skip on small hosts
        _assert_equals_oracle(oracle, recs, host, 2, 1, factors, "c2 vs oracle")


def test_c5_double_gyre_2048x1024x128(gpu, oracle):
    """BASELINE configs[4]: vector input, Jacobian derived at hits, non-symmetric classification, nbits 21"""
    from ftk_amd import tslab
    dims, nt = (2048, 1024), 128
    big_host = (os.cpu_count() or 1) >= 32
    recs, st, factors, host = _sweep_series(gpu, "double_gyre", dims, nt, nv=2, keep_host=big_host)
    assert st["work_items"] == tslab.count_simplices(2, dims, nt, scalar_input=False) == 3190884312 and st["cull_enabled"] == 1
    assert set(factors) == {1 << 21}
    assert len(recs) > nt and set(recs["type"].tolist()) == {4}        # derived off-diagonals are 0 -> saddles
    assert np.all(np.diff(recs["tag"].astype(np.uint64)) > 0)
    recs_e, st_e, _, _ = _sweep_series(gpu, "double_gyre", dims, nt, nv=2, exact_only=True)
    assert st_e["cull_enabled"] == 0
    _same(recs, recs_e)
    if big_host:
        _assert_equals_oracle(oracle, recs, host, 2, 2, factors, "c5 vs oracle")
