"""patches/ftk-xl-hip.patch -- the reference-side change a maintainer of hguo/ftk applies to get `use_accelerator("hip")` (INTEGRATION.md
sections 2-4) -- as an artefact that is checked: it applies cleanly to the reference tree (`git apply --check` on a scratch copy outside
this repository), the line ranges INTEGRATION.md cites still hold what it says they hold, and oracle/_ref/ftk_shim_driver -- which
tests/test_shim.py runs on the GPU against the CPU fixtures of the same reference -- was built from the PATCHED headers plus the patch's
new source file, with no update_timestep() override of its own."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"
PATCH = os.path.join(ROOT, "patches", "ftk-xl-hip.patch")
PATCHED = ["include/ftk/object.hh", "include/ftk/config.hh.in", "include/ftk/filters/filter.hh",
           "include/ftk/filters/critical_point_tracker_2d_regular.hh", "include/ftk/filters/critical_point_tracker_3d_regular.hh",
           "include/ftk/filters/critical_point_tracker_regular.hh"]

needs_reference = pytest.mark.skipif(not os.path.exists(os.path.join(REF, "include", "ftk", "object.hh")), reason="needs the reference tree (build container)")


def test_patch_touches_only_the_boundary():
    files = [l.split()[1][2:] for l in open(PATCH) if l.startswith("+++ b/")]
    assert sorted(files) == sorted(PATCHED + ["src/filters/critical_point_tracer_regular_hip.cpp"])


@needs_reference
def test_patch_applies_to_the_reference(tmp_path):
    for f in PATCHED:
        os.makedirs(tmp_path / os.path.dirname(f), exist_ok=True)
        shutil.copy(os.path.join(REF, f), tmp_path / f)
    r = subprocess.run(["git", "apply", "--check", "-p1", PATCH], cwd=tmp_path, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    r = subprocess.run(["git", "apply", "-p1", PATCH], cwd=tmp_path, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    new = (tmp_path / "src/filters/critical_point_tracer_regular_hip.cpp").read_text()
    assert "ftkx_shim.hh" in new and "extract_cp2dt_hip" in new and "extract_cp3dt_hip" in new
    assert "FTK_XL_HIP = 6" in (tmp_path / "include/ftk/object.hh").read_text()
    assert 'acc == "hip"' in (tmp_path / "include/ftk/filters/filter.hh").read_text()
    t2 = (tmp_path / "include/ftk/filters/critical_point_tracker_2d_regular.hh").read_text()
    t3 = (tmp_path / "include/ftk/filters/critical_point_tracker_3d_regular.hh").read_text()
    # the two caller defects the HIP branch must not inherit: domain size - 1, snapshot 0 twice as Sl
    assert "domain.size(0) - (xl == FTK_XL_HIP ? 0 : 1)" in t2 and "vector_field_scaling_factor, is_jacobian_field_symmetric, use_type_filter, type_filter" in t2
    assert "field_data_snapshots[xl == FTK_XL_HIP ? 1 : 0].scalar" in t3 and "enable_robust_detection" in t3
    # the resident form (round 6): the tracker owns the device context; push / update_timestep / pop reach the resident C ABI through it
    tr = (tmp_path / "include/ftk/filters/critical_point_tracker_regular.hh").read_text()
    assert "ftkx::resident_sweep hip;" in tr and "bool pop_field_data_snapshot();" in tr and "hip.pop_front()" in tr
    assert "if (hip_push_snapshot(&scalar, &vector, &jacobian))" in tr          # (all three given: critical_point_tracker.hh:202-213)
    assert "hip.push_scalar(t, scalar->data())" in tr and "hip.sweep(current_timestep, field_data_snapshots.size() >= 2, vector_field_resolution" in tr
    for t in (t2, t3):
        assert t.count("if (hip_push_snapshot(&s, NULL, NULL))") == 1 and t.count("if (hip_push_snapshot(NULL, &v, NULL))") == 1
        assert "if (resident) hip_sweep_current_timestep();" in t
        assert "resident ? hip_take_results(ELEMENT_SCOPE_ORDINAL) :" in t and "resident ? hip_take_results(ELEMENT_SCOPE_INTERVAL) :" in t
        assert "if (!hip_is_resident())" in t
        # the host derivation stays where it was, behind the early return: gradient / jacobian are NOT evaluated for a resident snapshot
        push = t[t.index("::push_scalar_field_snapshot(const ndarray<double>& s)"):]
        assert push.index("hip_push_snapshot(&s") < push.index("gradient")
    shim = open(os.path.join(ROOT, "include", "ftkx_shim.hh")).read()
    for sym in ("ftkx_push_scalar_slice", "ftkx_push_slice", "ftkx_sweep_series", "ftkx_drop_slice", "ftkx_set_mesh"):
        assert sym in shim, sym
    # reversible: the patch and nothing else
    r = subprocess.run(["git", "apply", "-R", "-p1", PATCH], cwd=tmp_path, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    for f in PATCHED:
        assert open(os.path.join(REF, f), "rb").read() == (tmp_path / f).read_bytes(), f


@needs_reference
@pytest.mark.parametrize("path,first,last,anchors", [
    # (file, first line, last line, what INTEGRATION.md / DESIGN.md say is there)
    ("include/ftk/filters/filter.hh", 26, 34, ["use_accelerator(int i)"]),
    ("include/ftk/filters/filter.hh", 74, 79, ['acc == "cuda"', "FTK_XL_SYCL"]),
    ("include/ftk/filters/filter.hh", 47, 61, ["set_device_ids", "device_ids"]),
    ("include/ftk/filters/critical_point_tracker_2d_regular.hh", 33, 63, ["extract_cp2dt_cuda(", "extract_cp2dt_sycl("]),
    ("include/ftk/filters/critical_point_tracker_2d_regular.hh", 65, 101, ["extract_cp2dt_xl_wrapper(", "FTK_ERR_ACCELERATOR_UNSUPPORTED"]),
    ("include/ftk/filters/critical_point_tracker_2d_regular.hh", 333, 347, ["ftk::lattice domain3", "domain.size(0)-1", "ordinal_core"]),
    ("include/ftk/filters/critical_point_tracker_2d_regular.hh", 369, 384, ["extract_cp2dt_xl_wrapper(", "ELEMENT_SCOPE_ORDINAL"]),
    ("include/ftk/filters/critical_point_tracker_2d_regular.hh", 387, 395, ["from_work_index", "to_integer"]),
    ("include/ftk/filters/critical_point_tracker_2d_regular.hh", 399, 414, ["ELEMENT_SCOPE_INTERVAL", "field_data_snapshots[1].scalar"]),
    ("include/ftk/filters/critical_point_tracker_3d_regular.hh", 42, 56, ["extract_cp3dt_cuda("]),
    ("include/ftk/filters/critical_point_tracker_3d_regular.hh", 203, 304, ["FTK_XL_CUDA", "ftk::lattice domain4", "extract_cp3dt_cuda(", "from_work_index"]),
    ("include/ftk/filters/critical_point_tracker_3d_regular.hh", 284, 285, ["field_data_snapshots[0].scalar.data()"]),
    ("include/ftk/object.hh", 33, 38, ["FTK_XL_NONE", "FTK_XL_CUDA"]),
    # what the resident form replaces (patch header)
    ("include/ftk/filters/critical_point_tracker_2d_regular.hh", 238, 261, ["push_scalar_field_snapshot", "gradient2D(s)", "jacobian2D<double, true>", "push_vector_field_snapshot"]),
    ("include/ftk/filters/critical_point_tracker_3d_regular.hh", 125, 148, ["push_scalar_field_snapshot", "gradient3D(s)", "jacobian3D(snapshot.vector)"]),
    ("include/ftk/filters/critical_point_tracker.hh", 231, 237, ["pop_field_data_snapshot", "pop_front"]),
    ("include/ftk/filters/critical_point_tracker.hh", 841, 848, ["advance_timestep", "update_timestep();", "pop_field_data_snapshot();"]),
    ("include/ftk/filters/critical_point_tracker.hh", 850, 864, ["update_vector_field_scaling_factor", "s.vector.resolution()", "1 << nbits"]),
    ("include/ftk/filters/critical_point_tracker.hh", 155, 159, ["field_data_snapshot_t", "std::deque<field_data_snapshot_t> field_data_snapshots"]),
    ("include/ftk/filters/critical_point_tracker.hh", 202, 213, ["push_field_data_snapshot", "snapshot.jacobian = jacobian"]),
    ("include/ftk/filters/regular_tracker.hh", 38, 40, ["set_coords_bounds", "set_coords_rectilinear", "set_coords_explicit"]),
])
def test_cited_line_ranges_hold_what_the_docs_say(path, first, last, anchors):
    lines = open(os.path.join(REF, path)).read().split("\n")
    text = "\n".join(lines[first - 1:last])
    for a in anchors:
        assert a in text, (path, first, last, a)


@needs_reference
def test_shim_driver_is_the_patched_reference():
    drv = os.path.join(ROOT, "oracle", "_ref", "ftk_shim_driver")
    if not os.path.exists(drv):
        pytest.skip("oracle/_ref/ftk_shim_driver not built (make -C oracle shim, after ftk_amd/libftkx.so)")
    assert os.path.getmtime(drv) >= os.path.getmtime(PATCH), "rebuild: make -C oracle shim"
    defined = subprocess.run(["nm", "-C", "--defined-only", drv], capture_output=True, text=True).stdout
    undefined = subprocess.run(["nm", "-C", "--undefined-only", drv], capture_output=True, text=True).stdout
    assert "extract_cp2dt_hip(" in defined and "extract_cp3dt_hip(" in defined            # the patch's new source file is linked in
    assert "ftkx_extract_cp2dt" in undefined and "ftkx_extract_cp3dt" in undefined        # ... and binds the C ABI of libftkx.so
    for sym in ("ftkx_push_scalar_slice", "ftkx_push_slice", "ftkx_sweep_series", "ftkx_drop_slice"):       # the resident calls too
        assert sym in undefined, sym
    assert "hip_tracker" not in defined                                                   # no subclass, no update_timestep() override
    src = open(os.path.join(ROOT, "oracle", "ref_driver.cpp")).read()
    assert 'use_accelerator("hip")' in src and "hip_tracker_2d" not in src
