"""TEST INFRASTRUCTURE -- ctypes binding of oracle/libftk_oracle.so (the CPU restatement of the
reference sweep).  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this module; the product package ftk_amd never does."""
import ctypes as C
import os
import subprocess
import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

REC_DTYPE = np.dtype([
    ("x", "<f8", (3,)), ("t", "<f8"), ("scalar", "<f8", (3,)), ("type", "<u4"), ("_pad0", "<u4"),
    ("tag", "<u8"), ("ordinal", "<i4"), ("timestep", "<i4"), ("etype", "<i4"), ("_pad1", "<i4"),
    ("corner", "<i4", (4,)),
])
assert REC_DTYPE.itemsize == 104


class SweepArgs(C.Structure):
    _fields_ = [
        ("nd", C.c_int), ("scope", C.c_int), ("current_timestep", C.c_int),
        ("domain_st", C.c_longlong * 3), ("domain_sz", C.c_longlong * 3),
        ("core_st", C.c_longlong * 3), ("core_sz", C.c_longlong * 3),
        ("ext_st", C.c_longlong * 3), ("ext_sz", C.c_longlong * 3),
        ("V", C.c_void_p * 2), ("J", C.c_void_p * 2), ("S", C.c_void_p * 2),
        ("factor", C.c_ulonglong),
        ("jacobian_symmetric", C.c_int), ("robust", C.c_int), ("use_type_filter", C.c_int),
        ("type_filter", C.c_uint), ("compute_degrees", C.c_int), ("tag_mode", C.c_int), ("nthreads", C.c_int),
        ("coords_mode", C.c_int), ("bounds", C.c_double * 6),
        ("rect", C.c_void_p * 3), ("expl", C.c_void_p), ("expl_ncomp", C.c_int), ("expl_n0", C.c_int),
    ]


class TrackArgs(C.Structure):
    _fields_ = [
        ("nd", C.c_int), ("nv", C.c_int), ("D", C.c_int * 3), ("DT", C.c_int),
        ("steps", C.POINTER(C.c_void_p)),
        ("robust", C.c_int), ("use_type_filter", C.c_int), ("type_filter", C.c_uint),
        ("compute_degrees", C.c_int), ("tag_mode", C.c_int), ("nthreads", C.c_int),
        ("coords_mode", C.c_int), ("bounds", C.c_double * 6),
        ("rect", C.c_void_p * 3), ("expl", C.c_void_p), ("expl_ncomp", C.c_int), ("expl_n0", C.c_int), ("t0", C.c_int),
    ]


TAG_WORK_INDEX, TAG_REFERENCE, TAG_EXACT64 = 0, 1, 2
SCOPE_ORDINAL, SCOPE_INTERVAL = 1, 2


def build():
    subprocess.check_call(["make", "-s", "-C", _HERE, "libftk_oracle.so"])


def lib():
    global _LIB
    if _LIB is None:
        path = os.environ.get("FTKO_LIB") or os.path.join(_HERE, "libftk_oracle.so")   # FTKO_LIB: e.g. a sanitizer build (tests/test_sanitizers.py)
        if not os.path.exists(path):
            build()
        L = C.CDLL(path)
        L.ftko_sweep.restype = C.c_size_t
        L.ftko_sweep.argtypes = [C.POINTER(SweepArgs), C.POINTER(C.c_void_p)]
        L.ftko_track.restype = C.c_size_t
        L.ftko_track.argtypes = [C.POINTER(TrackArgs), C.POINTER(C.c_void_p), C.c_void_p, C.POINTER(C.c_double)]
        L.ftko_free.argtypes = [C.c_void_p]
        L.ftko_num_work_items.restype = C.c_ulonglong
        L.ftko_num_work_items.argtypes = [C.POINTER(SweepArgs)]
        L.ftko_resolution.restype = C.c_double
        L.ftko_resolution.argtypes = [C.c_void_p, C.c_size_t]
        L.ftko_scaling_factor.restype = C.c_ulonglong
        L.ftko_scaling_factor.argtypes = [C.c_double, C.POINTER(C.c_int)]
        _LIB = L
    return _LIB


def _take(ptr, n):
    if n == 0:
        out = np.zeros(0, dtype=REC_DTYPE)
    else:
        buf = (C.c_char * (n * REC_DTYPE.itemsize)).from_address(ptr.value)
        out = np.frombuffer(buf, dtype=REC_DTYPE).copy()
    lib().ftko_free(ptr)
    return out


def _f64(a):
    a = np.ascontiguousarray(a, dtype=np.float64)
    return a


def unit_simplices(n):
    verts = (C.c_int * (60 * 4 * 4))()
    is_ord = (C.c_int * 60)()
    nt = lib().ftko_unit_simplices(n, verts, is_ord)
    v = np.frombuffer(verts, dtype=np.int32).reshape(60, 4, 4)[:nt, :n, :n].copy()
    return v, np.frombuffer(is_ord, dtype=np.int32)[:nt].astype(bool)


def side_of(n, t):
    ct = (C.c_int * 2)()
    co = (C.c_int * 8)()
    k = lib().ftko_side_of(n, t, ct, co)
    return [(ct[i], tuple(co[i * 4 + a] for a in range(n))) for i in range(k)]


def sides(n, t):
    ft = (C.c_int * 5)()
    fo = (C.c_int * 20)()
    k = lib().ftko_sides(n, t, ft, fo)
    return [(ft[i], tuple(fo[i * 4 + a] for a in range(n))) for i in range(k)]


def gradient2D(S):
    DH, DW = S.shape
    S = _f64(S); V = np.empty((DH, DW, 2))
    lib().ftko_gradient2D(C.c_void_p(S.ctypes.data), DW, DH, C.c_void_p(V.ctypes.data))
    return V


def jacobian2D(V, symmetric):
    DH, DW, _ = V.shape
    V = _f64(V); J = np.empty((DH, DW, 2, 2))
    lib().ftko_jacobian2D(C.c_void_p(V.ctypes.data), DW, DH, int(symmetric), C.c_void_p(J.ctypes.data))
    return J


def gradient3D(S):
    DD, DH, DW = S.shape
    S = _f64(S); V = np.empty((DD, DH, DW, 3))
    lib().ftko_gradient3D(C.c_void_p(S.ctypes.data), DW, DH, DD, C.c_void_p(V.ctypes.data))
    return V


def jacobian3D(V):
    DD, DH, DW, _ = V.shape
    V = _f64(V); J = np.empty((DD, DH, DW, 3, 3))
    lib().ftko_jacobian3D(C.c_void_p(V.ctypes.data), DW, DH, DD, C.c_void_p(J.ctypes.data))
    return J


def resolution(a):
    a = _f64(a)
    return lib().ftko_resolution(C.c_void_p(a.ctypes.data), a.size)


def scaling_factor(res):
    nb = C.c_int()
    f = lib().ftko_scaling_factor(res, C.byref(nb))
    return f, nb.value


def synthetic(name, dims, k, DT, x0=None, dirv=None):
    """One timestep of a reference synthetic case with the parameterisation of
    ndarray/stream.hh:1444-1567.  Arrays are numpy C-order with the reference's first index LAST
    (shape (DH, DW) / (DD, DH, DW) / (..., ncomp))."""
    L = lib()
    if name == "woven":
        DW, DH = dims
        t = 0.0 if DT == 1 else float(k) / (DT - 1)
        S = np.empty((DH, DW)); L.ftko_synthetic_woven_2D(DW, DH, C.c_double(t), C.c_void_p(S.ctypes.data)); return S
    if name == "merger_2d":
        DW, DH = dims
        S = np.empty((DH, DW)); L.ftko_synthetic_merger_2D(DW, DH, C.c_double(float(k) * 0.1), C.c_void_p(S.ctypes.data)); return S
    if name in ("moving_extremum_2d", "moving_extremum_3d"):
        nd = len(dims)
        D = (C.c_int * 3)(*list(dims) + [1] * (3 - nd))
        x0 = list(x0 if x0 is not None else ([10.0, 10.0] if nd == 2 else [10.0, 10.0, 10.0]))
        dirv = list(dirv if dirv is not None else ([0.1, 0.1] if nd == 2 else [0.1, 0.11, 0.1]))
        S = np.empty(tuple(reversed(dims)))
        L.ftko_synthetic_moving_extremum(nd, D, (C.c_double * 3)(*(x0 + [0.0] * (3 - nd))), (C.c_double * 3)(*(dirv + [0.0] * (3 - nd))),
                                         C.c_double(float(k)), C.c_void_p(S.ctypes.data))
        return S
    if name == "double_gyre":
        DW, DH = dims
        V = np.empty((DH, DW, 2))
        L.ftko_synthetic_double_gyre(DW, DH, C.c_double(k * 0.1), C.c_double(0.1), C.c_double(np.pi * 2), C.c_double(0.25), C.c_void_p(V.ctypes.data))
        return V
    raise ValueError(name)


def sweep(nd, scope, t, domain, core, ext, V, J, S, factor, jacobian_symmetric=True, robust=True,
          type_filter=None, compute_degrees=False, tag_mode=TAG_EXACT64, nthreads=1, explicit=None):
    """domain/core/ext = (starts, sizes) spatial.  V/J/S = (cur, next) numpy arrays or None.
    explicit: REGULAR_COORDS_EXPLICIT array of shape (n1, n0, ncomp)."""
    a = SweepArgs()
    a.nd, a.scope, a.current_timestep = nd, scope, t
    for name, (st, sz) in (("domain", domain), ("core", core), ("ext", ext)):
        for d in range(nd):
            getattr(a, name + "_st")[d] = int(st[d]); getattr(a, name + "_sz")[d] = int(sz[d])
    keep = []
    for name, pair in (("V", V), ("J", J), ("S", S)):
        for i in range(2):
            arr = pair[i] if pair is not None else None
            if arr is not None:
                arr = _f64(arr); keep.append(arr)
                getattr(a, name)[i] = arr.ctypes.data
            else:
                getattr(a, name)[i] = None
    a.factor = int(factor)
    a.jacobian_symmetric, a.robust = int(jacobian_symmetric), int(robust)
    a.use_type_filter, a.type_filter = int(type_filter is not None), int(type_filter or 0)
    a.compute_degrees, a.tag_mode, a.nthreads = int(compute_degrees), tag_mode, nthreads
    if explicit is not None:
        e = _f64(explicit); keep.append(e)
        a.coords_mode, a.expl, a.expl_ncomp, a.expl_n0 = 3, e.ctypes.data, e.shape[-1], e.shape[-2]
    out = C.c_void_p()
    n = lib().ftko_sweep(C.byref(a), C.byref(out))
    return _take(out, n)


def track(steps, nd, nv, robust=True, type_filter=None, compute_degrees=False, tag_mode=TAG_REFERENCE, nthreads=1, bounds=None,
          rectilinear=None, explicit=None, t0=0):
    """steps: list of DT numpy arrays (scalar: shape reversed dims; vector: (..., nd)).
    Returns (records, factors[DT], sweep_seconds)."""
    steps = [_f64(s) for s in steps]
    DT = len(steps)
    shp = steps[0].shape[:nd]
    a = TrackArgs()
    a.nd, a.nv, a.DT, a.t0 = nd, nv, DT, int(t0)
    for d in range(nd):
        a.D[d] = shp[nd - 1 - d]
    a.D[2] = a.D[2] if nd == 3 else 1
    ptrs = (C.c_void_p * DT)(*[s.ctypes.data for s in steps])
    a.steps = C.cast(ptrs, C.POINTER(C.c_void_p))
    a.robust, a.use_type_filter, a.type_filter = int(robust), int(type_filter is not None), int(type_filter or 0)
    a.compute_degrees, a.tag_mode, a.nthreads = int(compute_degrees), tag_mode, nthreads
    if bounds is not None:
        a.coords_mode = 1
        for i, b in enumerate(bounds):
            a.bounds[i] = float(b)
    keep = []
    if rectilinear is not None:       # REGULAR_COORDS_RECTILINEAR: one 1-D array per axis
        a.coords_mode = 2
        for d, r in enumerate(rectilinear):
            r = _f64(r); keep.append(r); a.rect[d] = r.ctypes.data
    if explicit is not None:          # REGULAR_COORDS_EXPLICIT: numpy array of shape (n1, n0, ncomp) = ndarray (ncomp, n0, n1)
        e = _f64(explicit); keep.append(e)
        a.coords_mode = 3
        a.expl, a.expl_ncomp, a.expl_n0 = e.ctypes.data, e.shape[-1], e.shape[-2]
    factors = np.zeros(DT, dtype=np.uint64)
    secs = C.c_double()
    out = C.c_void_p()
    n = lib().ftko_track(C.byref(a), C.byref(out), C.c_void_p(factors.ctypes.data), C.byref(secs))
    return _take(out, n), factors, secs.value
