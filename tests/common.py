"""Helpers shared by the parity tests."""
import glob
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def golden_names():
    return sorted(n for n in (os.path.basename(p)[:-4] for p in glob.glob(os.path.join(GOLDEN, "*.npz"))) if not n.startswith(("io_", "wrap_", "streaming_")))


def wrap_golden_names():
    """fixtures whose series starts at a large timestep so that the reference's int32 tag / vertex-id arithmetic wraps
    (tests/golden/make_golden_wrap.py); their tags are not element tags any more, so the pass-2 tests do not take them"""
    return sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(GOLDEN, "wrap_*.npz")))


def io_golden_names():
    return sorted(os.path.basename(p)[3:-4] for p in glob.glob(os.path.join(GOLDEN, "io_*.npz")))


def streaming_golden_names():
    """the reference's enable_streaming_trajectories runs (tests/golden/make_golden_streaming.py): trajectories as tag sequences"""
    return sorted(os.path.basename(p)[10:-4] for p in glob.glob(os.path.join(GOLDEN, "streaming_*.npz")))


def load_streaming_golden(name):
    z = np.load(os.path.join(GOLDEN, "streaming_" + name + ".npz"))
    offs = z["curve_offsets"]
    return dict(curves=[(int(z["curve_loop"][i]), z["curve_tags"][offs[i]:offs[i + 1]]) for i in range(len(z["curve_loop"]))],
                leftover_tags=z["leftover_tags"], pp_count=int(z["pp_count"]))


def load_io_golden(name):
    """the six files the reference wrote for one run (tests/golden/make_golden_io.py), as bytes"""
    z = np.load(os.path.join(GOLDEN, "io_" + name + ".npz"))
    return {k: z[k].tobytes() for k in z.files if k != "records_fixture"}, str(z["records_fixture"])


def load_golden(name):
    z = np.load(os.path.join(GOLDEN, name + ".npz"))
    d = {k: z[k] for k in z.files}      # NpzFile re-reads (and inflates) an array on every access
    return dict(
        nd=int(d["nd"]), nv=int(d["nv"]), dims=[int(x) for x in d["dims"]], DT=int(d["DT"]),
        steps=list(d["steps"]), factors=d["factors"], records=d["records"],
        robust=bool(int(d["robust"])) if "robust" in d else True,
        type_filter=int(d["type_filter"]) if "type_filter" in d else None,
        case=str(d["case"]) if "case" in d else None,
        degrees=bool(int(d["degrees"])) if "degrees" in d else False,
        bounds=[float(v) for v in d["bounds"]] if "bounds" in d else None,
        curves=([(int(d["curve_loop"][i]), d["curve_tags"][d["curve_offsets"][i]:d["curve_offsets"][i + 1]]) for i in range(len(d["curve_loop"]))]
                if "curve_loop" in d else None),
        pp=([(int(d["pp_loop"][i]), d["pp_tags"][d["pp_offsets"][i]:d["pp_offsets"][i + 1]], d["pp_types"][d["pp_offsets"][i]:d["pp_offsets"][i + 1]],
              d["pp_t"][d["pp_offsets"][i]:d["pp_offsets"][i + 1]]) for i in range(len(d["pp_loop"]))] if "pp_loop" in d else None),
        x0dir=d["x0dir"] if "x0dir" in d else None,
        rectilinear=[d[f"rect{i}"] for i in range(int(d["nd"]))] if "rect0" in d else None,
        explicit=d["explicit"] if "explicit" in d else None,
        t0=int(d["t0"]) if "t0" in d else 0,
    )


def by_tag(recs):
    return recs[np.argsort(recs["tag"], kind="stable")]


def assert_records_equal(got, ref, *, coord_tol=0.0, what=""):
    """tag / type / (ordinal, timestep when present) exact; coordinates and scalar within coord_tol
    (0.0 = bit-identical, NaN == NaN).  `got` and `ref` are structured arrays with fields
    tag, type, x[3], t and scalar (either [3] or scalar0)."""
    got, ref = by_tag(got), by_tag(ref)
    assert len(got) == len(ref), f"{what}: {len(got)} records, expected {len(ref)}"
    assert np.array_equal(got["tag"], ref["tag"]), f"{what}: tag sets differ"
    assert np.array_equal(got["type"], ref["type"]), \
        f"{what}: {int((got['type'] != ref['type']).sum())} type mismatches"
    for f in ("ordinal", "timestep"):
        if f in got.dtype.names and f in ref.dtype.names:
            assert np.array_equal(got[f], ref[f]), f"{what}: {f} differs"
    gs = got["scalar"][:, 0] if got["scalar"].ndim == 2 else got["scalar"]
    rs = ref["scalar"][:, 0] if ref["scalar"].ndim == 2 else ref["scalar"]
    for a, b, nm in ((got["x"], ref["x"], "x"), (got["t"], ref["t"], "t"), (gs, rs, "scalar")):
        if coord_tol == 0.0:
            assert np.array_equal(a, b, equal_nan=True), f"{what}: {nm} not bit-identical (max diff {np.nanmax(np.abs(a - b))})"
        else:
            assert np.allclose(a, b, rtol=0, atol=coord_tol, equal_nan=True), f"{what}: {nm} differs by {np.nanmax(np.abs(a - b))}"
