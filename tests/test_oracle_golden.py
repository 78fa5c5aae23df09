"""Pins the CPU oracle (oracle/ftk_oracle.c) against the REAL reference CPU path.

The fixtures in tests/golden/ were produced by oracle/_ref/ftk_ref_driver (the reference's own
critical_point_tracker_{2d,3d}_regular compiled from /root/reference) -- see tests/golden/make_golden.py.
Bar: record sets identical, tag/type/ordinal/timestep exact, x/t/scalar BIT-identical, and the per-step
quantisation factor identical."""
import numpy as np
import pytest

from common import assert_records_equal, golden_names, load_golden, wrap_golden_names

# known-answer record counts from BASELINE.md section 3 (reference CLI, --output-type discrete)
KNOWN_COUNTS = {
    "woven_31x37x32": (4491, 1205, {2: 1138, 4: 2248, 8: 1105}),
    "woven_128x128x10": (7357, 422, {2: 1861, 4: 3657, 8: 1839}),
    "moving_extremum_3d_32x32x32x8_dyadic": (34, 8, {2: 34}),
    "double_gyre_64x32x50": (879, 100, {4: 879}),
    "moving_extremum_2d_21x21x32": (94, 32, {2: 94}),
    "moving_extremum_3d_21x21x21x32": (126, 32, {2: 126}),
    "merger_2d_32x32x100": (1263, 286, {8: 915, 4: 348}),
    "moving_extremum_3d_21x21x21x4_overflow": (33662, None, {2: 32950, 1: 712}),
    "moving_extremum_2d_21x21x9_aligned": (27, None, {2: 27}),
}


@pytest.mark.parametrize("name", golden_names())
def test_oracle_matches_reference_records(oracle, name):
    g = load_golden(name)
    recs, factors, _ = oracle.track(g["steps"], g["nd"], g["nv"], robust=g["robust"], type_filter=g["type_filter"],
                                    compute_degrees=g["degrees"], bounds=g["bounds"], tag_mode=oracle.TAG_REFERENCE, nthreads=4,
                                   rectilinear=g["rectilinear"], explicit=g["explicit"])
    assert np.array_equal(factors, g["factors"]), "per-step quantisation factor differs"
    assert_records_equal(recs, g["records"], coord_tol=0.0, what=name)


@pytest.mark.parametrize("name", wrap_golden_names())
def test_oracle_matches_reference_where_int32_wraps(oracle, name):
    """series starting at a timestep so large that element::to_integer's int products and the int
    truncation of the SoS vertex ids wrap in the reference (tests/golden/make_golden_wrap.py).  Tags are then NOT the 64-bit
    element tags, and the wrapped (partly negative) ids change the order the SoS cascade sees -- reproduced bit for bit."""
    g = load_golden(name)
    assert g["t0"] > 0
    recs, factors, _ = oracle.track(g["steps"], g["nd"], g["nv"], tag_mode=oracle.TAG_REFERENCE, nthreads=4, t0=g["t0"])
    assert np.array_equal(factors, g["factors"])
    assert_records_equal(recs, g["records"], coord_tol=0.0, what=name)
    assert recs["timestep"].min() == g["t0"]
    exact, _, _ = oracle.track(g["steps"], g["nd"], g["nv"], tag_mode=oracle.TAG_EXACT64, nthreads=4, t0=g["t0"])
    assert len(exact) == len(recs) and not np.array_equal(np.sort(exact["tag"]), np.sort(recs["tag"]))   # the tags really wrapped


@pytest.mark.parametrize("name", sorted(KNOWN_COUNTS))
def test_known_answer_counts(oracle, name):
    n, n_ord, types = KNOWN_COUNTS[name]
    g = load_golden(name)
    recs, _, _ = oracle.track(g["steps"], g["nd"], g["nv"], nthreads=4)
    assert len(recs) == n
    if n_ord is not None:
        assert int(recs["ordinal"].sum()) == n_ord
    t, c = np.unique(recs["type"], return_counts=True)
    assert dict(zip(t.tolist(), c.tolist())) == types


def test_scaling_factor_sequence_is_sticky(oracle):
    # woven 31x37 -> 256 at step 0 then 4096; merger -> 256, 2048 (step 1), 8192 (step 60)
    f = load_golden("woven_31x37x32")["factors"]
    assert f[0] == 256 and set(f[1:].tolist()) == {4096}
    f = load_golden("merger_2d_32x32x100")["factors"]
    assert f[0] == 256 and set(f[1:60].tolist()) == {2048} and set(f[60:].tolist()) == {8192}


@pytest.mark.parametrize("name", ["woven_31x37x32", "merger_2d_32x32x100", "double_gyre_64x32x50",
                                  "moving_extremum_2d_21x21x9_aligned", "moving_extremum_3d_21x21x21x4_overflow",
                                  "moving_extremum_3d_32x32x32x8_dyadic"])
def test_synthetic_generators_match_reference_inputs(oracle, name):
    """oracle's restatement of ndarray/synthetic.hh vs the arrays the reference's own generators produced."""
    g = load_golden(name)
    x0dir = g["x0dir"]
    x0 = list(x0dir[:g["nd"]]) if x0dir is not None and len(x0dir) else None
    dv = list(x0dir[3:3 + g["nd"]]) if x0dir is not None and len(x0dir) else None
    for k in range(g["DT"]):
        a = oracle.synthetic(g["case"], g["dims"], k, g["DT"], x0, dv)
        assert np.array_equal(a, g["steps"][k]), f"{name} step {k}: max diff {np.abs(a - g['steps'][k]).max()}"


def test_moving_extremum_positions_are_analytic(oracle):
    """The reference's own assertion (tests/test_critical_point_tracking_moving_extremum_{2d,3d}.cpp): every point of the
    single trajectory lies on x0 + dir * t."""
    for name, tol in (("moving_extremum_3d_32x32x32x8_dyadic", 1e-12), ("moving_extremum_3d_21x21x21x32", 0.01),
                      ("moving_extremum_2d_21x21x32", 0.01)):
        g = load_golden(name)
        nd = g["nd"]
        x0dir = g["x0dir"] if len(g["x0dir"]) else np.array([10, 10, 10, 0.1, 0.11, 0.1] if nd == 3 else [10, 10, 0, 0.1, 0.1, 0])
        recs, _, _ = oracle.track(g["steps"], nd, 1)
        for d in range(nd):
            assert np.allclose(recs["x"][:, d], x0dir[d] + x0dir[3 + d] * recs["t"], atol=tol, rtol=0)


def test_work_index_and_exact_tags(oracle):
    g = load_golden("moving_extremum_3d_12x10x9x5_aligned")
    r_ref, _, _ = oracle.track(g["steps"], 3, 1, tag_mode=oracle.TAG_REFERENCE)
    r_64, _, _ = oracle.track(g["steps"], 3, 1, tag_mode=oracle.TAG_EXACT64)
    assert np.array_equal(r_ref["tag"], r_64["tag"])          # no int32 overflow on small meshes
    nx, ny, nz = [d - 3 for d in g["dims"]]
    c = r_64["corner"]
    idx = (c[:, 0] - 2) + nx * ((c[:, 1] - 2) + ny * ((c[:, 2] - 2) + nz * c[:, 3].astype(np.int64)))
    assert np.array_equal(r_64["tag"], idx.astype(np.uint64) * 60 + r_64["etype"].astype(np.uint64))
