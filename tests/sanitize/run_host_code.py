"""Drives the host-only parts of the library (curve tracing, trajectory post-processing, record-file writers and readers) built with
AddressSanitizer + UBSan over every fixture and over malformed files.  Run by tests/test_sanitizers.py in a subprocess with the
sanitizer runtimes preloaded."""
import ctypes as C, sys, os, json, numpy as np, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import ftk_amd._lib as L
# point the loader at the sanitized host-only library: only the host entry points are used below
L.LIB_PATH = os.environ["FTKX_HOST_SAN_LIB"]
lib = C.CDLL(L.LIB_PATH)
vp = C.c_void_p
lib.ftkx_trace_curves.argtypes = [C.c_int, C.POINTER(C.c_longlong), C.POINTER(C.c_longlong), vp, C.c_size_t, C.POINTER(L.Curves)]
lib.ftkx_post_process_curves.argtypes = [vp, C.c_size_t, C.POINTER(L.Curves), C.POINTER(L.Trajectories)]
lib.ftkx_free_curves.argtypes = [C.POINTER(L.Curves)]; lib.ftkx_free_trajectories.argtypes = [C.POINTER(L.Trajectories)]
lib.ftkx_write_critical_points.argtypes = [C.c_char_p, C.c_int, vp, C.c_size_t, vp, vp, C.POINTER(C.c_char_p), C.c_int]
lib.ftkx_read_critical_points.argtypes = [C.c_char_p, C.c_int, C.POINTER(vp), C.POINTER(C.c_size_t), C.POINTER(vp), C.POINTER(vp)]
lib.ftkx_write_traced_critical_points.argtypes = [C.c_char_p, C.c_int, vp, C.c_size_t, C.POINTER(L.Trajectories), C.POINTER(C.c_char_p), C.c_int]
lib.ftkx_read_traced_critical_points.argtypes = [C.c_char_p, C.c_int, C.POINTER(vp), C.POINTER(C.c_size_t), C.POINTER(L.Trajectories)]
lib.ftkx_free.argtypes = [vp]
from common import golden_names, load_golden, io_golden_names, load_io_golden
tmp = tempfile.mkdtemp()
for name in golden_names():
    g = load_golden(name); ref = g["records"]
    recs = np.zeros(len(ref), dtype=L.CP_DTYPE)
    for f in ("tag", "type", "x", "t"): recs[f] = ref[f]
    recs["aux"] = (ref["timestep"].astype(np.uint32) << 1) | ref["ordinal"].astype(np.uint32)
    scalar = g["nv"] == 1; lo = 2 if scalar else 1
    st = (C.c_longlong * 3)(*([lo] * g["nd"] + [0] * (3 - g["nd"]))); sz = (C.c_longlong * 3)(*([d - (3 if scalar else 2) for d in g["dims"]] + [1] * (3 - g["nd"])))
    cur = L.Curves(); assert lib.ftkx_trace_curves(g["nd"], st, sz, recs.ctypes.data, len(recs), C.byref(cur)) == 0
    tr = L.Trajectories(); assert lib.ftkx_post_process_curves(recs.ctypes.data, len(recs), C.byref(cur), C.byref(tr)) == 0
    for fmt, ext in ((0, "bin"), (1, "json"), (2, "txt")):
        p = os.path.join(tmp, "t." + ext).encode()
        assert lib.ftkx_write_critical_points(p, fmt, recs.ctypes.data, len(recs), None, None, None, -1) == 0
        assert lib.ftkx_write_traced_critical_points(p + b".tr", fmt, recs.ctypes.data, len(recs), C.byref(tr), None, -1) == 0
        if fmt != 2:
            r, n, v, i = vp(), C.c_size_t(), vp(), vp()
            assert lib.ftkx_read_critical_points(p, fmt, C.byref(r), C.byref(n), C.byref(v), C.byref(i)) == 0 and n.value == len(recs)
            for q in (r, v, i): lib.ftkx_free(q)
            r2, n2, t2 = vp(), C.c_size_t(), L.Trajectories()
            assert lib.ftkx_read_traced_critical_points(p + b".tr", fmt, C.byref(r2), C.byref(n2), C.byref(t2)) == 0
            lib.ftkx_free(r2); lib.ftkx_free_trajectories(C.byref(t2))
    lib.ftkx_free_curves(C.byref(cur)); lib.ftkx_free_trajectories(C.byref(tr))
# malformed inputs
for bad in (b"", b"[", b"[{", b'[{"x":[1,2,3]}]', b'{"trajs": 5}', b"[1,2", b'[{"x":[1,2],"t":1}]', b"\x05\0\0\0\0\0\0\0abc"):
    p = os.path.join(tmp, "bad"); open(p, "wb").write(bad)
    for fmt in (0, 1):
        r, n, v, i = vp(), C.c_size_t(), vp(), vp()
        lib.ftkx_read_critical_points(p.encode(), fmt, C.byref(r), C.byref(n), C.byref(v), C.byref(i))
        t2 = L.Trajectories(); lib.ftkx_read_traced_critical_points(p.encode(), fmt, C.byref(r), C.byref(n), C.byref(t2))
print("sanitizer run complete")
