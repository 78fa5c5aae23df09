"""Sanitizers on the CPU build (GPU AddressSanitizer is not available on the pool): the host-side C++ of the library -- trace.cpp and
io.cpp -- compiled with g++ -fsanitize=address,undefined and driven over every fixture plus malformed inputs."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _runtime(name):
    p = subprocess.run(["g++", "-print-file-name=" + name], capture_output=True, text=True).stdout.strip()
    return p if os.path.isabs(p) and os.path.exists(p) else None


@pytest.mark.skipif(_runtime("libasan.so") is None or _runtime("libubsan.so") is None, reason="sanitizer runtimes not installed")
def test_host_code_is_clean_under_asan_ubsan(tmp_path):
    lib = tmp_path / "libftkx_host_san.so"
    csrc = os.path.join(ROOT, "ftk_amd", "csrc")
    r = subprocess.run(["g++", "-std=c++17", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined", "-fno-omit-frame-pointer",
                        "-fPIC", "-shared", "-I" + os.path.join(ROOT, "include"), "-I" + csrc, "-o", str(lib),
                        os.path.join(csrc, "io.cpp"), os.path.join(csrc, "trace.cpp"), os.path.join(ROOT, "tests", "sanitize", "host_stub.cpp")],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]
    env = dict(os.environ, FTKX_HOST_SAN_LIB=str(lib), LD_PRELOAD=_runtime("libasan.so") + ":" + _runtime("libubsan.so"),
               ASAN_OPTIONS="detect_leaks=0:abort_on_error=1", UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "sanitize", "run_host_code.py")], capture_output=True, text=True, env=env, timeout=900)
    assert r.returncode == 0 and "sanitizer run complete" in r.stdout, (r.stdout[-2000:], r.stderr[-4000:])


@pytest.mark.skipif(_runtime("libubsan.so") is None, reason="UBSan runtime not installed")
def test_device_arithmetic_headers_are_clean_under_ubsan(tmp_path):
    """cp_device.hpp / fan_tables.hpp -- the code the HIP kernels execute per simplex -- compiled for the host with UBSan"""
    so = tmp_path / "libhostcheck_ubsan.so"
    r = subprocess.run(["g++", "-O1", "-g", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off", "-fsanitize=undefined", "-fno-sanitize-recover=undefined",
                        "-o", str(so), os.path.join(ROOT, "tests", "hostcheck", "hostcheck.cpp")], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]
    env = dict(os.environ, UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "sanitize", "run_hostcheck_ubsan.py"), str(so)], capture_output=True, text=True, env=env, timeout=900)
    assert r.returncode == 0 and "ubsan run complete" in r.stdout, (r.stdout[-2000:], r.stderr[-4000:])


@pytest.mark.skipif(_runtime("libasan.so") is None or _runtime("libubsan.so") is None, reason="sanitizer runtimes not installed")
def test_oracle_is_clean_under_asan_ubsan(tmp_path):
    lib = tmp_path / "libftk_oracle_san.so"
    r = subprocess.run(["gcc", "-O1", "-g", "-std=c11", "-fPIC", "-shared", "-ffp-contract=off", "-fno-fast-math", "-fwrapv", "-pthread",
                        "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined", "-fno-omit-frame-pointer", "-o", str(lib),
                        os.path.join(ROOT, "oracle", "ftk_oracle.c"), "-lm"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]
    env = dict(os.environ, FTKO_LIB=str(lib), LD_PRELOAD=_runtime("libasan.so") + ":" + _runtime("libubsan.so"),
               ASAN_OPTIONS="detect_leaks=0:abort_on_error=1", UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "sanitize", "run_oracle.py")], capture_output=True, text=True, env=env, timeout=1200)
    assert r.returncode == 0 and "oracle sanitizer run complete" in r.stdout, (r.stdout[-2000:], r.stderr[-4000:])


@pytest.mark.skipif(_runtime("libtsan.so") is None, reason="TSan runtime not installed")
def test_pass2_threads_are_clean_under_tsan(tmp_path):
    """trace.cpp's worker pool, the parallel (compare-and-swap) hash build and the threaded neighbour phase under ThreadSanitizer, on a
    hit set large enough to be split over threads (woven 128 x 128 x 10: 7 357 records), also with two callers at once"""
    import numpy as np
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from common import load_golden
    g = load_golden("woven_128x128x10")
    ref = g["records"]
    rec = np.zeros(len(ref), dtype=[("x", "<f8", (3,)), ("t", "<f8"), ("scalar", "<f8", (3,)), ("type", "<u4"), ("aux", "<u4"), ("tag", "<u8")])
    for f in ("tag", "type", "x", "t"):
        rec[f] = ref[f]
    rec["aux"] = (ref["ordinal"].astype(np.uint32) & 1) | (ref["timestep"].astype(np.uint32) << 1)
    rec = rec[np.argsort(rec["tag"], kind="stable")]
    assert rec.dtype.itemsize == 72 and len(rec) > 4096
    raw = tmp_path / "records.raw"
    rec.tofile(raw)
    exe = tmp_path / "tsan_trace"
    csrc = os.path.join(ROOT, "ftk_amd", "csrc")
    r = subprocess.run(["g++", "-std=c++17", "-O1", "-g", "-fsanitize=thread", "-pthread", "-I" + os.path.join(ROOT, "include"), "-I" + csrc, "-o", str(exe),
                        os.path.join(ROOT, "tests", "sanitize", "tsan_trace_main.cpp"), os.path.join(csrc, "trace.cpp"),
                        os.path.join(ROOT, "tests", "sanitize", "host_stub.cpp")], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]
    D = g["dims"]
    env = dict(os.environ, FTKX_TRACE_THREADS="8", TSAN_OPTIONS="halt_on_error=1")
    r = subprocess.run([str(exe), str(raw), "2", "2", str(D[0] - 3), str(D[1] - 3)], capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0 and "tsan run complete" in r.stdout and "WARNING: ThreadSanitizer" not in r.stderr, (r.stdout[-1000:], r.stderr[-4000:])
    assert "rc 0 curves" in r.stdout
