"""include/ftkx_shim.hh -- the C++ shim with the reference's accelerator signatures.
CPU: it compiles and links against the REAL ftk::lattice / ftk::feature_point_lite_t (build container only; skipped where the
reference tree is absent).  GPU: instantiated with this repo's own types and run from a C++ program, it returns the records the
oracle computes for the same call (tag = work index inside core)."""
import os
import subprocess

import numpy as np
import pytest

from common import golden_names, load_golden, wrap_golden_names

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference/include"
REFCFG = os.path.join(ROOT, "oracle", "_ref", "include")
SHIM_DRIVER = os.path.join(ROOT, "oracle", "_ref", "ftk_shim_driver")


def _lib_dir():
    from ftk_amd import build
    build.build()
    return os.path.join(ROOT, "ftk_amd")


@pytest.mark.skipif(not (os.path.exists(os.path.join(REF, "ftk", "mesh", "lattice.hh")) and os.path.exists(os.path.join(REFCFG, "ftk", "config.hh"))),
                    reason="needs the reference headers (build container) and oracle/_ref (make -C oracle ref)")
def test_shim_compiles_and_links_against_reference_headers(tmp_path):
    exe = tmp_path / "shim_ref"
    cmd = ["g++", "-std=c++17", "-O1", "-w", "-I" + REF, "-I" + REFCFG, "-I" + os.path.join(ROOT, "include"),
           os.path.join(ROOT, "tests", "shim", "shim_against_reference_headers.cpp"), "-o", str(exe),
           "-L" + _lib_dir(), "-lftkx", "-Wl,-rpath," + _lib_dir(), "-Wl,-rpath-link,/opt/rocm/lib", "-lpthread"]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]
    syms = subprocess.run(["nm", "-C", "--undefined-only", str(exe)], capture_output=True, text=True).stdout
    assert "ftkx_extract_cp2dt" in syms and "ftkx_extract_cp3dt" in syms      # the shim really binds the C ABI
    assert subprocess.run([str(exe)]).returncode == 0                           # main() only checks layouts; no GPU call


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["woven_31x37x32", "moving_extremum_3d_12x10x9x5_aligned", "adversarial_3d_scalar_9x9x9x4"])
def test_shim_returns_oracle_records(name, tmp_path, oracle):
    g = load_golden(name)
    nd, D = g["nd"], g["dims"]
    exe = tmp_path / "shim_run"
    r = subprocess.run(["g++", "-std=c++17", "-O1", "-I" + os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "shim", "shim_run.cpp"), "-o", str(exe),
                        "-L" + _lib_dir(), "-lftkx", "-Wl,-rpath," + _lib_dir(), "-Wl,-rpath-link,/opt/rocm/lib"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]
    fields = []
    for k in (0, 1):
        a = g["steps"][k]
        V = oracle.gradient2D(a) if nd == 2 else oracle.gradient3D(a)
        J = oracle.jacobian2D(V, True) if nd == 2 else oracle.jacobian3D(V)
        fields.append((V, J, a))
    factor, _ = oracle.scaling_factor(min(oracle.resolution(fields[0][0]), oracle.resolution(fields[1][0])))
    dom = ([2] * nd, [d - 3 for d in D])
    for scope in (1, 2):
        inp, out = tmp_path / f"in{scope}.bin", tmp_path / f"out{scope}.bin"
        with open(inp, "wb") as f:
            f.write(np.array([nd, 1, D[0], D[1], D[2] if nd == 3 else 1, 0, scope], dtype=np.int32).tobytes())
            f.write(np.uint64(factor).tobytes())
            for arr in (fields[0][0], fields[1][0], fields[0][1], fields[1][1], fields[0][2], fields[1][2]):
                f.write(np.ascontiguousarray(arr, dtype=np.float64).tobytes())
        r = subprocess.run([str(exe), str(inp), str(out)], capture_output=True, text=True)
        assert r.returncode == 0, r.stderr
        raw = open(out, "rb").read()
        n = int(np.frombuffer(raw, dtype=np.uint64, count=1)[0])
        import ftk_amd
        got = np.frombuffer(raw, dtype=ftk_amd.CP_DTYPE, count=n, offset=8)
        ref = oracle.sweep(nd, scope, 0, dom, dom, ([0] * nd, D), (fields[0][0], fields[1][0]), (fields[0][1], fields[1][1]),
                           (fields[0][2], fields[1][2]), factor, jacobian_symmetric=True, tag_mode=oracle.TAG_WORK_INDEX)
        assert len(got) == len(ref) and len(ref) > 0
        o = np.argsort(got["tag"], kind="stable"); p = np.argsort(ref["tag"], kind="stable")
        for fld in ("tag", "type"):
            assert np.array_equal(got[fld][o], ref[fld][p]), fld
        for fld in ("x", "t"):
            assert np.array_equal(got[fld][o], ref[fld][p], equal_nan=True), fld


def _driver_env(name, g, mode):
    """the options a fixture was made with (tests/golden/make_golden*.py), as oracle/ref_driver.cpp reads them"""
    env = dict(os.environ)
    for k in list(env):
        if k.startswith(("FTK_REF_", "FTK_SHIM_")):
            del env[k]
    if not g["robust"]:
        env["FTK_REF_NO_ROBUST"] = "1"
    if g["type_filter"]:
        env["FTK_REF_TYPE_FILTER"] = str(g["type_filter"])
    if g["degrees"]:
        env["FTK_REF_DEGREES"] = "1"
    if g["bounds"]:
        env["FTK_REF_BOUNDS"] = ",".join(repr(float(v)) for v in g["bounds"])
    for m in ("rect", "explicit2", "explicit3"):
        if name.endswith("_" + m):
            env["FTK_REF_COORDS"] = m
    if g["t0"]:
        env["FTK_REF_T0"] = str(g["t0"])
    if mode == "oneshot":
        env["FTK_SHIM_ONESHOT"] = "1"
    if mode == "resident_all_given":
        env["FTK_REF_PUSH_ALL"] = "1"
    if mode == "resident_streaming":
        env["FTK_REF_STREAMING"] = "1"
    return env


def _run_shim_driver(name, mode, tmp_path):
    import json
    from refdump import read_dump, write_input
    g = load_golden(name)
    inp, out = tmp_path / "in.bin", tmp_path / "o.bin"
    write_input(str(inp), g["steps"], g["nd"], g["nv"])
    r = subprocess.run([SHIM_DRIVER, "file", str(inp), str(out)], capture_output=True, text=True, timeout=600, env=_driver_env(name, g, mode))
    assert r.returncode == 0, r.stderr[-2000:]
    info = json.loads(r.stdout.strip().splitlines()[-1])
    assert info["hip_resident"] == (mode != "oneshot")
    return g, read_dump(str(out)), info


def _assert_dump_equals_fixture(d, g):
    assert np.array_equal(d["factors"], g["factors"])
    got, ref = d["records"], g["records"]
    assert len(got) == len(ref)
    o, p = np.argsort(got["tag"], kind="stable"), np.argsort(ref["tag"], kind="stable")
    for f in ("tag", "type", "ordinal", "timestep"):
        assert np.array_equal(got[f][o], ref[f][p]), f
    for f in ("x", "t", "scalar"):
        assert np.array_equal(got[f][o], ref[f][p], equal_nan=True), f            # bit-identical (north_star asks for 1e-6)


@pytest.mark.gpu
@pytest.mark.parametrize("name", golden_names() + wrap_golden_names())
def test_patched_reference_tracker_is_resident_and_matches_every_fixture(name, tmp_path):
    """patches/ftk-xl-hip.patch, executed on every record fixture: oracle/_ref/ftk_shim_driver is the REAL
    ftk::critical_point_tracker_{2d,3d}_regular built from the patched headers, told use_accelerator("hip") and nothing else.  Its
    push_*_field_snapshot hand each ndarray to libftkx.so once (no host gradient / jacobian), update_timestep() sweeps the resident
    slices with the factor formed on the device, pop_field_data_snapshot drops the slice; the reference's own from_work_index /
    to_integer loops, std::map and finalize() consume the records.  Result: the CPU fixture of the same reference -- per-step factors,
    records (bit for bit), traced curves -- over every option the fixtures cover (robust off, type filter, degrees, the four
    coordinate modes, int32-wrapped tags, near-singular Hessians incl. 72x64x56x8)."""
    if not os.path.exists(SHIM_DRIVER):
        pytest.skip("oracle/_ref/ftk_shim_driver not built (needs the build container: make -C oracle ref)")
    g, d, _ = _run_shim_driver(name, "resident", tmp_path)
    _assert_dump_equals_fixture(d, g)
    if g["curves"] is not None:
        assert sorted((tuple(t.tolist()), l) for l, t in d["curves"]) == sorted((tuple(t.tolist()), l) for l, t in g["curves"])


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["woven_31x37x32", "random_3d_scalar_13x12x11x4", "moving_extremum_3d_12x10x9x5_aligned"])
def test_patched_reference_tracker_with_scalar_vector_and_jacobian_given(name, tmp_path):
    """critical_point_tracker::push_field_data_snapshot(scalar, vector, jacobian) (critical_point_tracker.hh:202-213), the third way a snapshot
    reaches the tracker: all three arrays GIVEN -- here derived by the driver with the reference's own gradient / jacobian functions -- go to
    HBM once through the patch's override (ftkx_push_slice with V, J and S); the records are those of the fixture."""
    if not os.path.exists(SHIM_DRIVER):
        pytest.skip("oracle/_ref/ftk_shim_driver not built (needs the build container: make -C oracle ref)")
    g, d, _ = _run_shim_driver(name, "resident_all_given", tmp_path)
    _assert_dump_equals_fixture(d, g)


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["woven_31x37x32", "double_gyre_64x32x50", "moving_extremum_3d_21x21x21x32"])
def test_patched_reference_tracker_with_streaming_trajectories(name, tmp_path):
    """enable_streaming_trajectories (critical_point_tracker.hh:38; update_timestep 2d:326-330, 3d:197-201): the reference's own grow() --
    trace_critical_points_online, critical_point_tracker.hh:523-639 -- consumes the resident sweep's records after every interval sweep
    inside the patched update_timestep().  The trajectories and the discrete points left over are those the reference grew on the CPU
    (tests/golden/streaming_*.npz)."""
    if not os.path.exists(SHIM_DRIVER):
        pytest.skip("oracle/_ref/ftk_shim_driver not built (needs the build container: make -C oracle ref)")
    from common import load_streaming_golden
    sg = load_streaming_golden(name)
    g, d, _ = _run_shim_driver(name, "resident_streaming", tmp_path)
    assert np.array_equal(d["factors"], g["factors"])
    assert np.array_equal(np.sort(d["records"]["tag"]), sg["leftover_tags"])
    assert len(d["curves"]) == len(sg["curves"])
    assert sorted((tuple(t.tolist()), l) for l, t in d["curves"]) == sorted((tuple(t.tolist()), l) for l, t in sg["curves"])
    assert len(d["pp"]) == sg["pp_count"]


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["woven_31x37x32", "random_3d_scalar_13x12x11x4", "random_2d_vector_23x20x5", "random_2d_scalar_29x24x6_saddles"])
def test_patched_reference_tracker_one_shot_mode(name, tmp_path):
    """set_hip_resident(false): the same patched tracker through the literal reference boundary (extract_cp{2,3}dt_hip with host V, J, S)"""
    if not os.path.exists(SHIM_DRIVER):
        pytest.skip("oracle/_ref/ftk_shim_driver not built (needs the build container: make -C oracle ref)")
    g, d, _ = _run_shim_driver(name, "oneshot", tmp_path)
    _assert_dump_equals_fixture(d, g)


@pytest.mark.gpu
@pytest.mark.parametrize("name,case,args,curves,trajs", [
    ("woven_31x37x32", "woven", (31, 37, 1, 32), 56, 56),                                  # tests/test_critical_point_tracking_woven.cpp:32-37
    ("moving_extremum_3d_21x21x21x32", "moving_extremum_3d", (21, 21, 21, 32), 1, 1),
    ("double_gyre_64x32x50", "double_gyre", (64, 32, 1, 50), 2, 2),                        # vector input, reference-derived Jacobian
    ("merger_2d_32x32x100", "merger_2d", (32, 32, 1, 100), 5, 9),
    ("moving_extremum_3d_21x21x21x4_overflow", "moving_extremum_3d", (21, 21, 21, 4), 23208, None),   # wrapped determinants included
])
def test_real_reference_tracker_runs_through_the_shim(name, case, args, curves, trajs, tmp_path):
    """The drop-in, executed: oracle/_ref/ftk_shim_driver is the REAL ftk::critical_point_tracker_{2d,3d}_regular (compiled from
    /root/reference in the build container) whose update_timestep() takes the reference's accelerator branch with the call sites
    bound to libftkx.so through include/ftkx_shim.hh; the reference's own from_work_index / to_integer, its own std::map and its own
    finalize() consume what the HIP kernels return.  Result: the CPU fixture of the same reference, record for record, the
    per-step factors, the traced curves and the post-processed trajectories (56 on the woven test, as the reference's test asserts)."""
    if not os.path.exists(SHIM_DRIVER):
        pytest.skip("oracle/_ref/ftk_shim_driver not built (needs the build container: make -C oracle ref)")
    from refdump import read_dump
    g = load_golden(name)
    out = tmp_path / "o.bin"
    cmd = [SHIM_DRIVER, "synthetic", case] + [str(a) for a in args] + [str(out)]
    if len(g["x0dir"]):
        cmd += [repr(float(v)) for v in g["x0dir"]]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    d = read_dump(str(out))
    assert np.array_equal(d["factors"], g["factors"])
    got, ref = d["records"], g["records"]
    assert len(got) == len(ref)
    o, p = np.argsort(got["tag"], kind="stable"), np.argsort(ref["tag"], kind="stable")
    for f in ("tag", "type", "ordinal", "timestep"):
        assert np.array_equal(got[f][o], ref[f][p]), f
    for f in ("x", "t", "scalar"):
        assert np.array_equal(got[f][o], ref[f][p], equal_nan=True), f            # bit-identical (north_star asks for 1e-6)
    assert len(d["curves"]) == curves == len(g["curves"])
    assert sorted((tuple(t.tolist()), l) for l, t in d["curves"]) == sorted((tuple(t.tolist()), l) for l, t in g["curves"])
    if trajs is not None:
        assert len(d["pp"]) == trajs
