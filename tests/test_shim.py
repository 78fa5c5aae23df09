"""include/ftkx_shim.hh -- the C++ shim with the reference's accelerator signatures.
CPU: it compiles and links against the REAL ftk::lattice / ftk::feature_point_lite_t (build container only; skipped where the
reference tree is absent).  GPU: instantiated with this repo's own types and run from a C++ program, it returns the records the
oracle computes for the same call (tag = work index inside core)."""
import os
import subprocess

import numpy as np
import pytest

from common import load_golden

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference/include"
REFCFG = os.path.join(ROOT, "oracle", "_ref", "include")


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
