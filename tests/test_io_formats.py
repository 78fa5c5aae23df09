"""Record-stream formats: ftkx_{write,read}_[traced_]critical_points against the files the reference itself wrote
for the same run (tests/golden/io_*.npz, made by tests/golden/make_golden_io.py through the reference's own writers).
  binary  byte-identical                      json  same structure, every number parses to the identical double
  text    byte-identical                      readers: records equal to the record fixture of the same run
(JSON digits: this library prints the shortest round-trip spelling; the reference's JSON library uses Grisu2, which for about
1 double in 10^4 emits one digit more -- 20.231935889834048 for 20.23193588983405.  Same double either way.)
and the whole chain discrete records -> trace -> post-process -> traced file reproduces the reference's traced files.
Host-side code: runs without a GPU."""
import json
import re

import numpy as np
import pytest

from common import io_golden_names, load_golden, load_io_golden

FORMATS = [("bin", "binary"), ("json", "json"), ("txt", "text")]


def _lib():
    import ftk_amd
    from ftk_amd import build
    build.build()
    return ftk_amd


def _materialise(tmp_path, files):
    paths = {}
    for k, b in files.items():
        kind, ext = k.split("_")
        p = tmp_path / f"ref.{kind}.{ext}"
        p.write_bytes(b)
        paths[(kind, ext)] = str(p)
    return paths


_NUM = re.compile(rb"-?\d+(?:\.\d+)?(?:e[+-]\d+)?")


def _assert_json_equivalent(got, ref):
    assert json.loads(got) == json.loads(ref)                            # same members, same doubles, same 64-bit tags
    assert _NUM.sub(b"#", got) == _NUM.sub(b"#", ref)                    # same bytes apart from number spellings
    a, b = _NUM.findall(got), _NUM.findall(ref)
    differ = sum(x != y for x, y in zip(a, b))
    assert differ <= max(1, len(a) // 1000), f"{differ} of {len(a)} numbers spelled differently"


def _records_of_fixture(ftk_amd, g):
    ref = g["records"]
    recs = np.zeros(len(ref), dtype=ftk_amd.CP_DTYPE)
    recs["tag"] = ref["tag"]; recs["type"] = ref["type"]; recs["x"] = ref["x"]; recs["t"] = ref["t"]; recs["scalar"][:, 0] = ref["scalar"]
    recs["aux"] = (ref["timestep"].astype(np.uint32) << 1) | ref["ordinal"].astype(np.uint32)
    return recs


def _domain(g):
    scalar = g["nv"] == 1
    return ([2 if scalar else 1] * g["nd"], [d - (3 if scalar else 2) for d in g["dims"]])


@pytest.mark.parametrize("name", io_golden_names())
def test_read_discrete_equals_record_fixture(name, tmp_path):
    ftk_amd = _lib()
    files, of = load_io_golden(name)
    paths = _materialise(tmp_path, files)
    want = _records_of_fixture(ftk_amd, load_golden(of))
    for ext in ("bin", "json"):
        recs, v, ids = ftk_amd.read_critical_points(paths[("discrete", ext)])
        assert recs.tobytes() == want.tobytes(), ext           # every field, bit for bit, in the reference's order
        assert not v.any() and not ids.any()                    # discrete points carry no velocity / id


@pytest.mark.parametrize("name", io_golden_names())
@pytest.mark.parametrize("ext,fmt", FORMATS)
def test_write_discrete_equals_reference_file(name, ext, fmt, tmp_path):
    ftk_amd = _lib()
    files, of = load_io_golden(name)
    recs = _records_of_fixture(ftk_amd, load_golden(of))
    out = tmp_path / ("out." + ext)                             # format chosen from the file name, like json_interface.hh:225-231
    ftk_amd.write_critical_points(str(out), recs)
    got, ref = out.read_bytes(), files["discrete_" + ext]
    if fmt == "json":
        _assert_json_equivalent(got, ref)
    else:
        assert got == ref


def _canon_traced(data, ext):
    """Curves of a traced file as a sorted list, labels removed.  The reference appends curves from a thread pool
    (critical_point_tracker.hh:709-737: parallel_for_container + mutex), so their order and hence their labels and the per-point
    `id` vary from run to run; everything else is determined."""
    if ext == "json":
        curves = json.loads(data)["trajs"]
        for c in curves:
            c.pop("id")
            for q in c["traj"]:
                q.pop("id")
        return sorted(json.dumps(c, sort_keys=True) for c in curves)
    if ext == "txt":
        lines = data.decode().split("\n")
        assert lines[0].startswith("#trajectories=")
        blocks = []
        for ln in lines[1:]:
            if ln.startswith("--trajectory "):
                blocks.append([ln.split(", ", 1)[1]])
            elif ln:
                assert ln.startswith("---")
                blocks[-1].append(ln.rsplit(", id=", 1)[0])
        assert len(blocks) == int(lines[0].split("=")[1])
        return sorted("\n".join(b) for b in blocks)
    n = int(np.frombuffer(data, dtype="<u8", count=1)[0])
    off, out = 8, []
    for _ in range(n):
        head = data[off + 4:off + 4 + 1 + 17 * 8 + 4]                     # complete, 5 x double[3], tmin, tmax, consistent_type
        off += 4 + 1 + 17 * 8 + 4
        npts = int(np.frombuffer(data, dtype="<u8", count=1, offset=off)[0]); off += 8
        pts = np.frombuffer(data, dtype=np.uint8, count=npts * 105, offset=off).reshape(npts, 105)[:, :97]   # drop the trailing id
        off += npts * 105
        out.append((head, pts.tobytes()))
    assert off == len(data)
    return sorted(out)


@pytest.mark.parametrize("name", io_golden_names())
@pytest.mark.parametrize("ext,fmt", FORMATS)
def test_chain_records_to_traced_file_equals_reference(name, ext, fmt, tmp_path):
    """sweep records -> ftkx_trace_curves -> ftkx_post_process_curves -> ftkx_write_traced_critical_points == the file the
    reference writes after finalize() + post_process(): the same trajectories (points, smoothed types, adjusted times, statistics)
    in every format; byte-identical files where the reference's curve order is determined (a single curve)."""
    ftk_amd = _lib()
    files, of = load_io_golden(name)
    g = load_golden(of)
    recs = _records_of_fixture(ftk_amd, g)
    ts = ftk_amd.post_process(g["nd"], _domain(g), recs)
    out = tmp_path / ("traced." + ext)
    ftk_amd.write_traced_critical_points(str(out), recs, ts)
    got, ref = out.read_bytes(), files["traced_" + ext]
    assert len(got) == len(ref) or fmt != "binary"                     # labels are decimal in json / text
    assert _canon_traced(got, ext) == _canon_traced(ref, ext)
    if len(ts) == 1:
        if fmt == "json":
            _assert_json_equivalent(got, ref)
        else:
            assert got == ref


@pytest.mark.parametrize("name", io_golden_names())
def test_read_traced_round_trip(name, tmp_path):
    ftk_amd = _lib()
    files, of = load_io_golden(name)
    paths = _materialise(tmp_path, files)
    rb, tb = ftk_amd.read_traced_critical_points(paths[("traced", "bin")])
    rj, tj = ftk_amd.read_traced_critical_points(paths[("traced", "json")])
    assert rb.tobytes() == rj.tobytes() and np.array_equal(tb.offsets, tj.offsets)
    assert np.array_equal(tb.type, rb["type"]) and np.array_equal(tb.t, rb["t"])
    # binary keeps the multimap labels (split pieces share one); json renumbers on load (feature_curve_set.hh:130-134)
    assert np.array_equal(tj.id, np.arange(len(tj)))
    assert np.all(np.diff(tb.id) >= 0)
    out = tmp_path / "again.bin"
    ftk_amd.write_traced_critical_points(str(out), rb, tb)
    assert out.read_bytes() == files["traced_bin"]


def test_json_numbers_follow_nlohmann(tmp_path):
    """spelling rules of the reference's JSON library for the doubles a record can hold"""
    ftk_amd = _lib()
    vals = [0.0, -0.0, 1.0, -2.5, 10.1, 1e14, 1e15, 123456789012345.0, 1234567890123456.0, 0.001, 0.0001, 0.00001, 1e-7, 5e-324, 1.7976931348623157e308,
            0.1 + 0.2, 1 / 3, 2.0 ** 53, 1234.5e10, float("nan"), float("inf"), -float("inf")]
    want = ["0.0", "-0.0", "1.0", "-2.5", "10.1", "100000000000000.0", "1e+15", "123456789012345.0", "1.234567890123456e+15", "0.001", "0.0001", "1e-05", "1e-07",
            "5e-324", "1.7976931348623157e+308", "0.30000000000000004", "0.3333333333333333", "9.007199254740992e+15", "12345000000000.0", "null", "null", "null"]
    recs = np.zeros(len(vals), dtype=ftk_amd.CP_DTYPE)
    recs["t"] = vals
    recs["tag"] = [2 ** 63 + 5] + [0] * (len(vals) - 1)                # tags are 64-bit: not representable as doubles
    p = tmp_path / "n.json"
    ftk_amd.write_critical_points(str(p), recs)
    text = p.read_text()
    got = [o.split('"t":')[1].split(',"tag"')[0] for o in text.split("},{")]
    assert got == want
    assert '"tag":9223372036854775813' in text
    back, _, _ = ftk_amd.read_critical_points(str(p))
    assert back["tag"][0] == 2 ** 63 + 5
    finite = np.isfinite(vals)
    assert np.array_equal(back["t"][finite], np.array(vals)[finite]) and np.all(np.isnan(back["t"][~finite]))
    assert np.signbit(back["t"][1])


def test_io_errors(tmp_path):
    ftk_amd = _lib()
    with pytest.raises(ftk_amd.FtkxError):
        ftk_amd.read_critical_points(str(tmp_path / "missing.bin"))
    (tmp_path / "bad.json").write_text('[{"x": [0, 0, 0]}]')
    with pytest.raises(ftk_amd.FtkxError):
        ftk_amd.read_critical_points(str(tmp_path / "bad.json"))
    (tmp_path / "trunc.bin").write_bytes(np.uint64(5).tobytes() + b"\0" * 100)
    with pytest.raises(ftk_amd.FtkxError):
        ftk_amd.read_critical_points(str(tmp_path / "trunc.bin"))
    with pytest.raises(ftk_amd.FtkxError):                            # the reference has no text reader
        ftk_amd.read_critical_points(str(tmp_path / "x.txt"))
    recs = np.zeros(0, dtype=ftk_amd.CP_DTYPE)                        # empty sets are valid files
    for ext in ("bin", "json", "txt"):
        ftk_amd.write_critical_points(str(tmp_path / ("empty." + ext)), recs)
    assert (tmp_path / "empty.json").read_text() == "[]" and (tmp_path / "empty.bin").read_bytes() == bytes(8)
    assert len(ftk_amd.read_critical_points(str(tmp_path / "empty.bin"))[0]) == 0
    assert len(ftk_amd.read_critical_points(str(tmp_path / "empty.json"))[0]) == 0


@pytest.mark.gpu
@pytest.mark.parametrize("name", io_golden_names())
def test_gpu_tracker_writes_reference_files(name, tmp_path):
    """the whole product chain on the GPU: sweep (HIP) -> finalize -> post_process -> the tracker's write_* members, against the
    files the reference wrote for the same input.  Integer content exact; coordinates within 1e-6 (they are in fact equal)."""
    from gpu_common import run_tracker
    ftk_amd = _lib()
    files, of = load_io_golden(name)
    g = load_golden(of)
    paths = _materialise(tmp_path, files)
    out = {k: str(tmp_path / ("gpu." + k)) for k in ("discrete.bin", "discrete.json", "discrete.txt", "traced.bin", "traced.json", "traced.txt")}

    def after(tr):
        tr.write_critical_points_binary(out["discrete.bin"]); tr.write_critical_points_json(out["discrete.json"]); tr.write_critical_points_text(out["discrete.txt"])
        tr.finalize(); tr.post_process()
        tr.write_traced_critical_points_binary(out["traced.bin"]); tr.write_traced_critical_points_json(out["traced.json"]); tr.write_traced_critical_points_text(out["traced.txt"])

    run_tracker(g["steps"], g["nd"], g["nv"], after=after)
    got, _, _ = ftk_amd.read_critical_points(out["discrete.bin"])
    ref, _, _ = ftk_amd.read_critical_points(paths[("discrete", "bin")])
    assert np.array_equal(got["tag"], ref["tag"]) and np.array_equal(got["type"], ref["type"]) and np.array_equal(got["aux"], ref["aux"])
    for f in ("x", "t", "scalar"):
        assert np.allclose(got[f], ref[f], rtol=0, atol=1e-6, equal_nan=True)
    gj, _, _ = ftk_amd.read_critical_points(out["discrete.json"])
    assert gj.tobytes() == got.tobytes()                                 # json and binary carry the same records
    bit_exact = got.tobytes() == ref.tobytes()
    if bit_exact:                                                        # then the files themselves must match
        assert open(out["discrete.bin"], "rb").read() == files["discrete_bin"]
        assert open(out["discrete.txt"], "rb").read() == files["discrete_txt"]
        _assert_json_equivalent(open(out["discrete.json"], "rb").read(), files["discrete_json"])
        for ext in ("bin", "json", "txt"):
            assert _canon_traced(open(out["traced." + ext], "rb").read(), ext) == _canon_traced(files["traced_" + ext], ext)
    # traced: same trajectories as tag / type sequences, times within tolerance
    gr, gt = ftk_amd.read_traced_critical_points(out["traced.bin"])
    rr, rt = ftk_amd.read_traced_critical_points(paths[("traced", "bin")])
    key = lambda r, t: sorted((tuple(r["tag"][a:b].tolist()), tuple(t.type[a:b].tolist()), tuple(np.round(t.t[a:b], 6).tolist()))  # noqa: E731
                              for a, b in zip(t.offsets[:-1], t.offsets[1:]))
    assert key(gr, gt) == key(rr, rt)
    print(f"{name}: {len(got)} records, {len(gt)} trajectories, bit-exact={bit_exact}")
    assert bit_exact, "coordinates differ in the last bits from the reference's (allowed by the 1e-6 bar, but unexpected)"
