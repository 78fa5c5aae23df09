"""ftk_amd -- MI355X-native critical-point space-time simplex sweep behind FTK's tracker API.

The product is the shared library ftk_amd/libftkx.so (HIP kernels for gfx950 + C ABI `ftkx_*`, see include/ftkx.h, and the C++
tracker of include/ftkx_tracker.hh).  This package is the Python plumbing over that C ABI, mirroring the reference's tracker
interface (include/ftk/filters/critical_point_tracker_{2d,3d}_regular.hh; python/pyftk.cpp:93-142)."""
import ctypes as C

import numpy as np

from . import _lib
from ._lib import (CP_DTYPE, FORMAT_BINARY, FORMAT_JSON, FORMAT_TEXT, SCOPE_BOTH, SCOPE_INTERVAL, SCOPE_ORDINAL, SOURCE_DERIVED, SOURCE_GIVEN, SOURCE_NONE,  # noqa: F401
                   TAG_EXACT64, TAG_REFERENCE, TAG_WORK_INDEX, FtkxError, Options, Stats)

__all__ = ["trace_curves", "pass2", "trace_and_post_process", "post_process", "TrajectorySet", "write_critical_points", "read_critical_points",
           "write_traced_critical_points", "read_traced_critical_points", "Context", "CriticalPointTracker2DRegular", "CriticalPointTracker3DRegular", "extract_cp2dt", "extract_cp3dt",
           "scaling_factor", "CP_DTYPE", "FtkxError"]


def _ptr(a):
    """host numpy array / torch tensor / raw int -> (address, keepalive, on_device)"""
    if a is None:
        return None, None, 0
    if isinstance(a, int):
        return a, None, 1
    if hasattr(a, "data_ptr"):          # torch tensor (device memory is torch's job: plumbing, not the product)
        if a.dtype.is_floating_point and a.element_size() != 8:
            raise TypeError("fields must be float64")
        a = a.contiguous()
        return a.data_ptr(), a, 1 if a.is_cuda else 0
    a = np.ascontiguousarray(a, dtype=np.float64)
    return a.ctypes.data, a, 0


def default_options(**kw):
    o = Options()
    _lib.load().ftkx_default_options(C.byref(o))
    for k, v in kw.items():
        if k == "coords_bounds":
            o.coords_mode = 1
            for i, b in enumerate(v):
                o.coords_bounds[i] = float(b)
        else:
            setattr(o, k, int(v))
    return o


def scaling_factor(resolution):
    nb = C.c_int()
    f = _lib.load().ftkx_scaling_factor(float(resolution), C.byref(nb))
    return int(f), nb.value


class Context:
    """ftkx_ctx: slices resident in HBM + sweeps (include/ftkx.h)."""

    def __init__(self, nd, device_id=0):
        self._L = _lib.load()
        self.nd = nd
        self._h = C.c_void_p()
        _lib.check(self._L.ftkx_create(C.byref(self._h), nd, device_id))
        self._keep = {}

    def close(self):
        if self._h:
            self._L.ftkx_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _ck(self, rc):
        _lib.check(rc, self._h)

    def set_stream(self, stream_ptr):
        self._ck(self._L.ftkx_set_stream(self._h, C.c_void_p(stream_ptr)))

    def set_options(self, **kw):
        self._opt = default_options(**kw)
        self._ck(self._L.ftkx_set_options(self._h, C.byref(self._opt)))

    def set_coords_rectilinear(self, arrays):
        arrs = [np.ascontiguousarray(a, dtype=np.float64) for a in arrays] + [None] * (3 - len(arrays))
        args = []
        for a in arrs:
            args += [a.ctypes.data if a is not None else None, len(a) if a is not None else 0]
        self._ck(self._L.ftkx_set_coords_rectilinear(self._h, *args))

    def set_coords_explicit(self, coords):
        e = np.ascontiguousarray(coords, dtype=np.float64)
        self._ck(self._L.ftkx_set_coords_explicit(self._h, e.ctypes.data, e.shape[-1], e.shape[-2], e.shape[-3]))

    def set_mesh(self, domain, core, ext):
        """each = (starts, sizes), spatial axes only (x, y[, z])"""
        args = []
        for st, sz in (domain, core, ext):
            args += [_lib.ll(st), _lib.ll(sz, fill=1)]
        self._ck(self._L.ftkx_set_mesh(self._h, *args))

    def push_slice(self, t, V, J=None, S=None):
        pv, kv, dv = _ptr(V); pj, kj, dj = _ptr(J); ps, ks, ds = _ptr(S)
        devs = {d for p, d in ((pv, dv), (pj, dj), (ps, ds)) if p is not None}
        if len(devs) != 1:
            raise ValueError("V, J, S must all be host arrays or all device tensors")
        on_dev = devs.pop()
        self._ck(self._L.ftkx_push_slice(self._h, t, pv, pj, ps, on_dev))
        self._keep[t] = (kv, kj, ks) if on_dev else None

    def push_scalar_slice(self, t, S):
        ps, ks, ds = _ptr(S)
        self._ck(self._L.ftkx_push_scalar_slice(self._h, t, ps, ds))
        self._keep[t] = ks if ds else None

    def drop_slice(self, t):
        self._ck(self._L.ftkx_drop_slice(self._h, t))
        self._keep.pop(t, None)

    def slice_resolution(self, t):
        r, m = C.c_double(), C.c_double()
        self._ck(self._L.ftkx_slice_resolution(self._h, t, C.byref(r), C.byref(m)))
        return r.value, m.value

    def slices_resolution(self, ts):
        """slice_resolution for several resident slices with one launch and one synchronise -> {t: (resolution, max_abs)}"""
        ts = [int(t) for t in ts]
        n = len(ts)
        tt = (C.c_int * max(1, n))(*ts); r = (C.c_double * max(1, n))(); m = (C.c_double * max(1, n))()
        self._ck(self._L.ftkx_slices_resolution(self._h, tt, n, r, m))
        return {ts[i]: (r[i], m[i]) for i in range(n)}

    def slices_prepare(self, ts, factor_hint=0):
        """ftkx_slices_prepare: ONE pass over the slices builds the sweep's sign masks (under factor_hint, 0 = 256) and returns
        {t: (res_below, max_abs)} -- res_below = smallest non-zero |v| below 1 / factor_hint (DBL_MAX if none)"""
        ts = [int(t) for t in ts]
        n = len(ts)
        tt = (C.c_int * max(1, n))(*ts); r = (C.c_double * max(1, n))(); m = (C.c_double * max(1, n))()
        self._ck(self._L.ftkx_slices_prepare(self._h, tt, n, int(factor_hint), r, m))
        return {ts[i]: (r[i], m[i]) for i in range(n)}

    def sweep_announce(self, ts, scopes):
        """ftkx_sweep_announce: the sweeps that will be enqueued after the next slices_prepare, whose cull that call then queues
        right behind the mask kernel (a hint: collect uses the survivor list only if the pending sweeps are exactly these)"""
        n = len(ts)
        # (contiguous int32 numpy arrays are handed over as they are: a caller that announces the same sweeps every pass builds them once)
        if isinstance(ts, np.ndarray) and isinstance(scopes, np.ndarray) and ts.dtype == np.int32 and scopes.dtype == np.int32 and ts.flags.c_contiguous and scopes.flags.c_contiguous:
            self._ck(self._L.ftkx_sweep_announce(self._h, ts.ctypes.data, scopes.ctypes.data, n))
            return
        self._ck(self._L.ftkx_sweep_announce(self._h, (C.c_int * max(1, n))(*[int(t) for t in ts]), (C.c_int * max(1, n))(*[int(v) for v in scopes]), n))

    # ---- compact t-slab halo (torch tensors on this context's device, or on the host) ----
    def export_masks(self, t, torch, device):
        """owner side: (U uint8 tensor, word_index int32 tensor, words int64 tensor, mask_factor, max_abs) of a prepared slice"""
        ub, nw, mf, mx = C.c_size_t(), C.c_size_t(), C.c_ulonglong(), C.c_double()
        self._ck(self._L.ftkx_export_masks_size(self._h, int(t), C.byref(ub), C.byref(nw), C.byref(mf), C.byref(mx)))
        U = torch.empty((ub.value,), dtype=torch.uint8, device=device)
        idx = torch.empty((max(1, nw.value),), dtype=torch.int32, device=device)
        words = torch.empty((max(1, nw.value),), dtype=torch.int64, device=device)
        self._ck(self._L.ftkx_export_masks(self._h, int(t), U.data_ptr(), idx.data_ptr(), words.data_ptr(), 1 if U.is_cuda else 0))
        return U, idx[:nw.value], words[:nw.value], mf.value, mx.value

    def push_masked_slice(self, t, scalar_input, U, idx, words, mask_factor, max_abs):
        self._keep[("masked", t)] = (U, idx, words)
        self._ck(self._L.ftkx_push_masked_slice(self._h, int(t), int(bool(scalar_input)), U.data_ptr(), U.numel() * U.element_size(), idx.data_ptr() if len(idx) else None,
                                                words.data_ptr() if len(words) else None, len(idx), int(mask_factor), float(max_abs), 1 if U.is_cuda else 0))

    def packed_masks_bytes(self):
        """size of a packed mask message for this context's mesh (0: the mesh has no summarised masks), word capacity"""
        cap = C.c_size_t()
        n = self._L.ftkx_packed_masks_bytes(self._h, C.byref(cap))
        return int(n), int(cap.value)

    def export_masks_packed(self, t, out):
        """owner side: a prepared slice's masks as ONE message in `out` (uint8 tensor of packed_masks_bytes() bytes); a device tensor is
        filled by work queued on the context's stream -- nothing is waited for"""
        self._ck(self._L.ftkx_export_masks_packed(self._h, int(t), out.data_ptr(), 1 if out.is_cuda else 0))

    def push_masked_slice_packed(self, t, scalar_input, buf, mask_factor, max_abs):
        self._keep[("masked", t)] = buf
        self._ck(self._L.ftkx_push_masked_slice_packed(self._h, int(t), int(bool(scalar_input)), buf.data_ptr(), 1 if buf.is_cuda else 0, int(mask_factor), float(max_abs)))

    def sweep_cull(self, t_masked, torch, device):
        """receiver side, after sweep_enqueue: the cells (int64 tensor of core-linear indices) whose exact test reads the masked slice"""
        n = C.c_size_t()
        self._ck(self._L.ftkx_sweep_cull(self._h, int(t_masked), C.byref(n)))
        cells = torch.empty((max(1, n.value),), dtype=torch.int64, device=device)
        self._ck(self._L.ftkx_get_sparse_cells(self._h, cells.data_ptr(), 1 if cells.is_cuda else 0))
        return cells[:n.value]

    def sweep_enqueue_many(self, ts, scopes, factors):
        n = len(ts)
        if isinstance(ts, np.ndarray) and isinstance(scopes, np.ndarray) and ts.dtype == np.int32 and scopes.dtype == np.int32 and ts.flags.c_contiguous and scopes.flags.c_contiguous:
            f = np.ascontiguousarray(factors, dtype=np.uint64)     # (timesteps and scopes handed over as they are: built once by the caller)
            self._ck(self._L.ftkx_sweep_enqueue_many(self._h, ts.ctypes.data, scopes.ctypes.data, f.ctypes.data, n))
            return
        self._ck(self._L.ftkx_sweep_enqueue_many(self._h, (C.c_int * n)(*[int(t) for t in ts]), (C.c_int * n)(*[int(v) for v in scopes]),
                                                 (C.c_ulonglong * n)(*[int(f) for f in factors]), n))

    def sweep_series(self, ts, scopes, running_resolution=None, copy=True):
        """ftkx_sweep_series: the steps (ts[i], scopes[i]) over resident slices under the reference's sticky factor, formed on the
        device; one host wait.  -> (records, factors (uint64 array), running resolution after these slices).
        ts / scopes: int32 arrays (handed over as they are) or sequences."""
        n = len(ts)
        if not (isinstance(ts, np.ndarray) and ts.dtype == np.int32 and ts.flags.c_contiguous):
            ts = np.ascontiguousarray(ts, dtype=np.int32)
        if not (isinstance(scopes, np.ndarray) and scopes.dtype == np.int32 and scopes.flags.c_contiguous):
            scopes = np.ascontiguousarray(scopes, dtype=np.int32)
        run = C.c_double(np.finfo(np.float64).max if running_resolution is None else float(running_resolution))
        f = np.empty((max(1, n),), dtype=np.uint64)
        out, cnt = C.c_void_p(), C.c_size_t()
        self._ck(self._L.ftkx_sweep_series(self._h, ts.ctypes.data, scopes.ctypes.data, n, C.byref(run), f.ctypes.data, C.byref(out), C.byref(cnt)))
        return _lib.records_from(out.value, cnt.value, copy), f[:n], run.value

    def sweep_series_submit(self, ts, scopes, running_resolution=None, chain=False):
        """ftkx_sweep_series_submit: queue the pass and return.  chain=True: continue from the pass queued before it (still open)."""
        n = len(ts)
        ts = np.ascontiguousarray(ts, dtype=np.int32)
        scopes = np.ascontiguousarray(scopes, dtype=np.int32)
        if chain:
            run_p = None
        else:
            run = C.c_double(np.finfo(np.float64).max if running_resolution is None else float(running_resolution))
            run_p = C.addressof(run)
        self._ck(self._L.ftkx_sweep_series_submit(self._h, ts.ctypes.data, scopes.ctypes.data, n, run_p))
        self._open_series = getattr(self, "_open_series", [])
        self._open_series.append(n)

    def sweep_series_complete(self, copy=True):
        """ftkx_sweep_series_complete: the oldest open pass -> (records, factors, running resolution), as sweep_series returns them"""
        n = self._open_series.pop(0)
        run = C.c_double(0.0)
        f = np.empty((max(1, n),), dtype=np.uint64)
        out, cnt = C.c_void_p(), C.c_size_t()
        self._ck(self._L.ftkx_sweep_series_complete(self._h, C.byref(run), f.ctypes.data, C.byref(out), C.byref(cnt)))
        return _lib.records_from(out.value, cnt.value, copy), f[:n], run.value

    # ---- the slab pass (ftkx_series_dist_*): one rank's device-driven pass, queued in stages; ftk_amd/tslab.py drives it ----
    def series_dist_cells(self):
        return int(self._L.ftkx_series_dist_cells(self._h))

    def series_dist_begin(self, ts, scopes, running_resolution, rank, nranks, upper, contrib, gathered, masks_out, side_stream=None):
        """upper: the rank that owns the slice behind this slab (None: no halo).  Tensors: device memory of this context's device
        (masks_out: None without a lower neighbour); side_stream: a hipStream_t (int) that is made to wait for masks_out only"""
        n = len(ts)
        ts = np.ascontiguousarray(ts, dtype=np.int32)
        scopes = np.ascontiguousarray(scopes, dtype=np.int32)
        run = C.c_double(np.finfo(np.float64).max if running_resolution is None else float(running_resolution))
        self._ck(self._L.ftkx_series_dist_begin(self._h, ts.ctypes.data, scopes.ctypes.data, n, C.byref(run), int(rank), int(nranks), -1 if upper is None else int(upper),
                                                contrib.data_ptr(), gathered.data_ptr(), masks_out.data_ptr() if masks_out is not None else None,
                                                C.c_void_p(side_stream) if side_stream else None))
        self._dist_n = n

    def series_dist_cull(self, masks_in, request_out):
        self._ck(self._L.ftkx_series_dist_cull(self._h, masks_in.data_ptr() if masks_in is not None else None, request_out.data_ptr() if request_out is not None else None))

    def series_dist_serve(self, request_in, reply_out):
        self._ck(self._L.ftkx_series_dist_serve(self._h, request_in.data_ptr() if request_in is not None else None, reply_out.data_ptr() if reply_out is not None else None))

    def series_dist_finish(self, reply_in):
        self._ck(self._L.ftkx_series_dist_finish(self._h, reply_in.data_ptr() if reply_in is not None else None))
        self._open_series = getattr(self, "_open_series", [])
        self._open_series.append(self._dist_n)

    def series_dist_status(self, nranks):
        """of the slab pass completed last -> (asked, served, gathered[nranks, 4])"""
        a, s_ = C.c_longlong(), C.c_longlong()
        g = np.zeros((nranks, 4), dtype=np.float64)
        self._ck(self._L.ftkx_series_dist_status(self._h, C.byref(a), C.byref(s_), g.ctypes.data, int(nranks)))
        return int(a.value), int(s_.value), g

    def sweep_series_abort(self):
        """ftkx_sweep_series_abort: discard the open passes (after a failed submit / complete, or to give up)"""
        self._open_series = []
        self._ck(self._L.ftkx_sweep_series_abort(self._h))

    def series_last_path(self):
        """(path, status bits) of the last sweep_series: 1 device-driven (kernel chain), 2 finished by the fused tail, 4 one launch, 5 split (the
        chain next to the next pass's mask kernel), 0 host-driven batch"""
        st = C.c_ulonglong()
        return int(self._L.ftkx_series_last_path(self._h, C.byref(st))), int(st.value)

    def series_split_decision(self):
        """how this context decides on the split pass: {"state": "auto: measuring" | "auto: split" | "auto: in order" | "forced on" | "forced off",
        "median_in_order_ms", "median_split_ms"} (ftkx_series_split_decision)"""
        st, a, b = C.c_int(), C.c_double(), C.c_double()
        self._ck(self._L.ftkx_series_split_decision(self._h, C.byref(st), C.byref(a), C.byref(b)))
        names = ["auto: measuring", "auto: split", "auto: in order", "forced on", "forced off"]
        return {"state": names[st.value], "median_in_order_ms": a.value, "median_split_ms": b.value}

    def sweep_cancel(self):
        self._ck(self._L.ftkx_sweep_cancel(self._h))

    def patch_doubles(self):
        return int(self._L.ftkx_patch_doubles(self._h))

    def gather_patches(self, t, cells, torch):
        out = torch.empty((max(1, len(cells)) * self.patch_doubles(),), dtype=torch.float64, device=cells.device)
        if len(cells):
            self._ck(self._L.ftkx_gather_patches(self._h, int(t), cells.data_ptr(), len(cells), out.data_ptr(), 1 if cells.is_cuda else 0))
        return out[:len(cells) * self.patch_doubles()]

    def scatter_patches(self, t, cells, patches):
        if len(cells):
            self._ck(self._L.ftkx_scatter_patches(self._h, int(t), cells.data_ptr(), len(cells), patches.data_ptr(), 1 if cells.is_cuda else 0))

    def set_slice_resolution(self, t, resolution, max_abs):
        self._ck(self._L.ftkx_set_slice_resolution(self._h, t, float(resolution), float(max_abs)))

    def sweep(self, t, scope, factor):
        out, n = C.c_void_p(), C.c_size_t()
        self._ck(self._L.ftkx_sweep(self._h, t, scope, int(factor), C.byref(out), C.byref(n)))
        return _lib.records_from(out.value, n.value)

    def sweep_enqueue(self, t, scope, factor):
        self._ck(self._L.ftkx_sweep_enqueue(self._h, t, scope, int(factor)))

    def sweep_collect(self, copy=True):
        """copy=False: a view of the library's pinned host buffer, valid until the next sweep / collect on this context"""
        out, n = C.c_void_p(), C.c_size_t()
        self._ck(self._L.ftkx_sweep_collect(self._h, C.byref(out), C.byref(n)))
        return _lib.records_from(out.value, n.value, copy)

    def stats(self):
        s = Stats()
        self._ck(self._L.ftkx_get_stats(self._h, C.byref(s)))
        return {k: getattr(s, k) for k, _ in Stats._fields_}

    KERNELS = ("mask_kernel", "cull_kernel", "exact_kernel", "tile_kernel")

    def invalidate_masks(self):
        self._ck(self._L.ftkx_invalidate_masks(self._h))

    def set_profiling(self, on=True):
        """0 / False off, 1 / True every kernel family, 2 the mask kernel only"""
        self._ck(self._L.ftkx_set_profiling(self._h, int(on)))

    def debug_tile_repeat(self, repeat):
        """profiling aid: the tile kernel's fan phase `repeat` times per tile and step from the next sweep on (1 = off)"""
        self._ck(self._L.ftkx_debug_tile_repeat(self._h, int(repeat)))

    def upload_counts(self):
        """(staged, direct): host arrays of this context that went up through the library's pinned staging / the runtime's own copy"""
        a = C.c_ulonglong(0); b = C.c_ulonglong(0)
        self._ck(self._L.ftkx_debug_upload_counts(self._h, C.byref(a), C.byref(b)))
        return int(a.value), int(b.value)

    def kernel_times(self):
        """{kernel: (summed device ms, launches)} measured with HIP events on the context's stream"""
        ms = (C.c_double * 4)(); n = (C.c_ulonglong * 4)()
        self._ck(self._L.ftkx_get_kernel_times(self._h, ms, n))
        return {k: (ms[i], n[i]) for i, k in enumerate(self.KERNELS)}

    # derived fields on device tensors (ndarray/grad.hh)
    def gradient2D(self, S_ptr, DW, DH, V_ptr): self._ck(self._L.ftkx_gradient2D(self._h, S_ptr, DW, DH, V_ptr))
    def jacobian2D(self, V_ptr, DW, DH, symmetric, J_ptr): self._ck(self._L.ftkx_jacobian2D(self._h, V_ptr, DW, DH, int(symmetric), J_ptr))
    def gradient3D(self, S_ptr, DW, DH, DD, V_ptr): self._ck(self._L.ftkx_gradient3D(self._h, S_ptr, DW, DH, DD, V_ptr))
    def jacobian3D(self, V_ptr, DW, DH, DD, J_ptr): self._ck(self._L.ftkx_jacobian3D(self._h, V_ptr, DW, DH, DD, J_ptr))


def _extract(nd, scope, current_timestep, domain, core, ext, Vc, Vn, Jc, Jn, Sc, Sn, factor, options, device_id, coords=None):
    L = _lib.load()
    keep = []
    ptrs = []
    for a in (Vc, Vn, Jc, Jn, Sc, Sn):
        if a is None:
            ptrs.append(None)
        else:
            a = np.ascontiguousarray(a, dtype=np.float64); keep.append(a); ptrs.append(a.ctypes.data)
    n4 = nd + 1
    lat = [_lib.ll(domain[0], n4), _lib.ll(domain[1], n4, 1), _lib.ll(core[0], n4), _lib.ll(core[1], n4, 1), _lib.ll(ext[0], 3), _lib.ll(ext[1], 3, 1)]
    out, n = C.c_void_p(), C.c_size_t()
    opt = C.byref(options) if options is not None else None
    if nd == 2:
        cc = None if coords is None else np.ascontiguousarray(coords, dtype=np.float64)
        rc = L.ftkx_extract_cp2dt(scope, current_timestep, *lat, *ptrs, 0 if cc is None else 1, None if cc is None else cc.ctypes.data, int(factor), opt, device_id,
                                  C.byref(out), C.byref(n))
    else:
        rc = L.ftkx_extract_cp3dt(scope, current_timestep, *lat, *ptrs, int(factor), opt, device_id, C.byref(out), C.byref(n))
    _lib.check(rc)
    recs = _lib.records_from(out.value, n.value)
    L.ftkx_free(out)
    return recs


def extract_cp2dt(scope, current_timestep, domain, core, ext, Vc, Vn, Jc, Jn, Sc, Sn, factor, options=None, device_id=0, coords=None):
    """extract_cp2dt_cuda's argument list (critical_point_tracker_2d_regular.hh:33-63); lattices as (starts, sizes) incl. time.
    coords: the boundary's explicit vertex coordinates, shape (DH, DW, 2) (use_explicit_coords = true), or None."""
    return _extract(2, scope, current_timestep, domain, core, ext, Vc, Vn, Jc, Jn, Sc, Sn, factor, options, device_id, coords)


def extract_cp3dt(scope, current_timestep, domain, core, ext, Vc, Vn, Jc, Jn, Sc, Sn, factor, options=None, device_id=0):
    """extract_cp3dt_cuda's argument list (critical_point_tracker_3d_regular.hh:42-56)."""
    return _extract(3, scope, current_timestep, domain, core, ext, Vc, Vn, Jc, Jn, Sc, Sn, factor, options, device_id)


def trace_curves(nd, domain, records, ctx=None):
    """Pass 2 (ftkx_trace_curves): records (CP_DTYPE, element tags) -> (list of index arrays into `records`, loop flags, n_special).
    ctx: a Context whose GPU does the neighbour search and the component labelling (ftkx_trace_curves_ctx); same curves."""
    L = _lib.load()
    recs = np.ascontiguousarray(records, dtype=CP_DTYPE)
    out = _lib.Curves()
    if ctx is not None:
        _lib.check(L.ftkx_trace_curves_ctx(ctx._h, nd, _lib.ll(domain[0]), _lib.ll(domain[1], fill=1), recs.ctypes.data, len(recs), C.byref(out)), ctx._h)
    else:
        _lib.check(L.ftkx_trace_curves(nd, _lib.ll(domain[0]), _lib.ll(domain[1], fill=1), recs.ctypes.data, len(recs), C.byref(out)))
    offs = np.ctypeslib.as_array(out.offsets, shape=(out.n_curves + 1,)).copy()
    idx = np.ctypeslib.as_array(out.indices, shape=(max(1, out.n_points),))[:out.n_points].copy()
    loop = np.ctypeslib.as_array(out.loop, shape=(max(1, out.n_curves),))[:out.n_curves].copy()
    nspecial = out.n_special
    L.ftkx_free_curves(C.byref(out))
    return [idx[offs[i]:offs[i + 1]] for i in range(len(offs) - 1)], loop, nspecial


class OnlineTracer:
    """ftkx_online_tracer: trace_critical_points_online (enable_streaming_trajectories) on record arrays"""

    def __init__(self, nd, domain):
        self._L = _lib.load()
        self._h = C.c_void_p()
        _lib.check(self._L.ftkx_online_tracer_create(C.byref(self._h), nd, _lib.ll(domain[0]), _lib.ll(domain[1], fill=1)))

    def grow(self, records):
        recs = np.ascontiguousarray(records, dtype=CP_DTYPE)
        _lib.check(self._L.ftkx_online_tracer_grow(self._h, recs.ctypes.data, len(recs)))

    def curves(self):
        """-> (list of record arrays, one per trajectory in order of birth; loop flags)"""
        pts, out = C.c_void_p(), _lib.Curves()
        _lib.check(self._L.ftkx_online_tracer_curves(self._h, C.byref(pts), C.byref(out)))
        n = out.n_points
        recs = np.frombuffer(C.string_at(pts.value, max(1, n) * CP_DTYPE.itemsize), dtype=CP_DTYPE)[:n].copy()
        offs = np.ctypeslib.as_array(out.offsets, shape=(out.n_curves + 1,)).copy()
        loop = np.ctypeslib.as_array(out.loop, shape=(max(1, out.n_curves),))[:out.n_curves].copy()
        self._L.ftkx_free(pts); self._L.ftkx_free_curves(C.byref(out))
        return [recs[offs[i]:offs[i + 1]] for i in range(len(offs) - 1)], loop

    def close(self):
        if self._h:
            self._L.ftkx_online_tracer_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class TrajectorySet:
    """ftkx_trajectories as numpy arrays: curve c owns points offsets[c]:offsets[c+1]; per point the index into the record
    array, the (smoothed) type and the (adjusted) time; per curve the loop flag and its label in the reference's multimap."""

    def __init__(self, offsets, indices, type, t, loop, id):
        self.offsets = np.ascontiguousarray(offsets, dtype=np.int64)
        self.indices = np.ascontiguousarray(indices, dtype=np.int64)
        self.type = np.ascontiguousarray(type, dtype=np.uint32)
        self.t = np.ascontiguousarray(t, dtype=np.float64)
        self.loop = np.ascontiguousarray(loop, dtype=np.int32)
        self.id = np.ascontiguousarray(id, dtype=np.int32)

    def __len__(self):
        return len(self.offsets) - 1

    def curve(self, c):
        a, b = self.offsets[c], self.offsets[c + 1]
        return self.indices[a:b], self.type[a:b], self.t[a:b], int(self.loop[c])

    @classmethod
    def _from_c(cls, out):
        n, npts = out.n_curves, out.n_points
        grab = lambda p, k: np.ctypeslib.as_array(p, shape=(max(1, k),))[:k].copy()  # noqa: E731
        return cls(np.ctypeslib.as_array(out.offsets, shape=(n + 1,)).copy(), grab(out.indices, npts), grab(out.type, npts), grab(out.t, npts),
                   grab(out.loop, n), grab(out.id, n))

    def _to_c(self):
        out = _lib.Trajectories()
        out.n_curves, out.n_points = len(self), len(self.indices)
        out.offsets = self.offsets.ctypes.data_as(C.POINTER(C.c_longlong)); out.indices = self.indices.ctypes.data_as(C.POINTER(C.c_longlong))
        out.loop = self.loop.ctypes.data_as(C.POINTER(C.c_int)); out.type = self.type.ctypes.data_as(C.POINTER(C.c_uint))
        out.t = self.t.ctypes.data_as(C.POINTER(C.c_double)); out.id = self.id.ctypes.data_as(C.POINTER(C.c_int))
        return out


def post_process(nd, domain, records):
    """ftkx_trace_curves followed by ftkx_post_process_curves (json_interface::post_process defaults) -> TrajectorySet."""
    L = _lib.load()
    recs = np.ascontiguousarray(records, dtype=CP_DTYPE)
    cur = _lib.Curves()
    _lib.check(L.ftkx_trace_curves(nd, _lib.ll(domain[0]), _lib.ll(domain[1], fill=1), recs.ctypes.data, len(recs), C.byref(cur)))
    out = _lib.Trajectories()
    rc = L.ftkx_post_process_curves(recs.ctypes.data, len(recs), C.byref(cur), C.byref(out))
    L.ftkx_free_curves(C.byref(cur))
    _lib.check(rc)
    ts = TrajectorySet._from_c(out)
    L.ftkx_free_trajectories(C.byref(out))
    return ts


def pass2(nd, domain, records, ctx=None):
    """ftkx_trace_curves (ctx given: ftkx_trace_curves_ctx, its data-parallel half on that context's GPU), then ftkx_post_process_curves
    on its result, each timed by itself (the C calls only)
    -> (curves as index arrays, loop flags, n_special, TrajectorySet, ms_trace, ms_post_process)"""
    import time
    L = _lib.load()
    recs = np.ascontiguousarray(records, dtype=CP_DTYPE)
    cur = _lib.Curves()
    t0 = time.perf_counter()
    if ctx is not None:
        _lib.check(L.ftkx_trace_curves_ctx(ctx._h, nd, _lib.ll(domain[0]), _lib.ll(domain[1], fill=1), recs.ctypes.data, len(recs), C.byref(cur)), ctx._h)
    else:
        _lib.check(L.ftkx_trace_curves(nd, _lib.ll(domain[0]), _lib.ll(domain[1], fill=1), recs.ctypes.data, len(recs), C.byref(cur)))
    t1 = time.perf_counter()
    out = _lib.Trajectories()
    rc = L.ftkx_post_process_curves(recs.ctypes.data, len(recs), C.byref(cur), C.byref(out))
    t2 = time.perf_counter()
    offs = np.ctypeslib.as_array(cur.offsets, shape=(cur.n_curves + 1,)).copy()
    idx = np.ctypeslib.as_array(cur.indices, shape=(max(1, cur.n_points),))[:cur.n_points].copy()
    loop = np.ctypeslib.as_array(cur.loop, shape=(max(1, cur.n_curves),))[:cur.n_curves].copy()
    nspecial = cur.n_special
    L.ftkx_free_curves(C.byref(cur))
    _lib.check(rc)
    ts = TrajectorySet._from_c(out)
    L.ftkx_free_trajectories(C.byref(out))
    return [idx[offs[i]:offs[i + 1]] for i in range(len(offs) - 1)], loop, nspecial, ts, (t1 - t0) * 1e3, (t2 - t1) * 1e3


def trace_and_post_process(nd, domain, records):
    """post_process() as a list of (indices, types, t, loop) per trajectory."""
    ts = post_process(nd, domain, records)
    return [ts.curve(c) for c in range(len(ts))]


def _format(path, format):
    if format is None:
        return _lib.load().ftkx_format_from_path(str(path).encode())
    return {"binary": FORMAT_BINARY, "json": FORMAT_JSON, "text": FORMAT_TEXT}.get(format, format)


def _names(scalar_names):
    if scalar_names is None:
        return None, -1
    arr = (C.c_char_p * max(1, len(scalar_names)))(*[s.encode() for s in scalar_names])
    return arr, len(scalar_names)


def write_critical_points(path, records, format=None, v=None, id=None, scalar_names=None):
    """critical_point_tracker::write_critical_points_{json,binary,text}; format None = by file name (txt / json / else binary)."""
    recs = np.ascontiguousarray(records, dtype=CP_DTYPE)
    vv = None if v is None else np.ascontiguousarray(v, dtype=np.float64).reshape(len(recs), 3)
    ii = None if id is None else np.ascontiguousarray(id, dtype=np.uint64)
    names, nn = _names(scalar_names)
    _lib.check(_lib.load().ftkx_write_critical_points(str(path).encode(), _format(path, format), recs.ctypes.data, len(recs),
                                                      None if vv is None else vv.ctypes.data, None if ii is None else ii.ctypes.data, names, nn))


def read_critical_points(path, format=None):
    """read_critical_points_{json,binary} -> (records[CP_DTYPE] with the aux word, v[n,3], id[n])"""
    L = _lib.load()
    r, v, i, n = C.c_void_p(), C.c_void_p(), C.c_void_p(), C.c_size_t()
    _lib.check(L.ftkx_read_critical_points(str(path).encode(), _format(path, format), C.byref(r), C.byref(n), C.byref(v), C.byref(i)))
    k = n.value
    recs = np.frombuffer(C.string_at(r.value, k * CP_DTYPE.itemsize), dtype=CP_DTYPE).copy()
    vv = np.frombuffer(C.string_at(v.value, k * 24), dtype=np.float64).reshape(k, 3).copy()
    ii = np.frombuffer(C.string_at(i.value, k * 8), dtype=np.uint64).copy()
    for p in (r, v, i):
        L.ftkx_free(p)
    return recs, vv, ii


def write_traced_critical_points(path, records, trajectories, format=None, scalar_names=None):
    """write_traced_critical_points_{json,binary,text} of a TrajectorySet over `records`."""
    recs = np.ascontiguousarray(records, dtype=CP_DTYPE)
    tr = trajectories._to_c()
    names, nn = _names(scalar_names)
    _lib.check(_lib.load().ftkx_write_traced_critical_points(str(path).encode(), _format(path, format), recs.ctypes.data, len(recs), C.byref(tr), names, nn))


def read_traced_critical_points(path, format=None):
    """read_traced_critical_points_{json,binary} -> (records in file order, TrajectorySet indexing them)"""
    L = _lib.load()
    r, n, out = C.c_void_p(), C.c_size_t(), _lib.Trajectories()
    _lib.check(L.ftkx_read_traced_critical_points(str(path).encode(), _format(path, format), C.byref(r), C.byref(n), C.byref(out)))
    recs = np.frombuffer(C.string_at(r.value, n.value * CP_DTYPE.itemsize), dtype=CP_DTYPE).copy()
    ts = TrajectorySet._from_c(out)
    L.ftkx_free(r); L.ftkx_free_trajectories(C.byref(out))
    return recs, ts


class _TrackerRegular:
    """ftkx::critical_point_tracker_regular (include/ftkx_tracker.hh) through its C handle; method names are the reference's."""
    ND = 0

    def __init__(self, device_id=0, device_ids=None, block=2):
        """device_ids: several GPUs behind one tracker (timesteps dealt to them in blocks of `block` steps, one host thread per
        device; a device may be listed more than once)"""
        self._L = _lib.load()
        self._h = C.c_void_p()
        if device_ids is not None:
            ids = (C.c_int * len(device_ids))(*[int(d) for d in device_ids])
            _lib.check(self._L.ftkx_tracker_create_multi(C.byref(self._h), self.ND, ids, len(device_ids), int(block)), None, True)
        else:
            _lib.check(self._L.ftkx_tracker_create(C.byref(self._h), self.ND, device_id), None, True)
        self._src = [SOURCE_NONE, SOURCE_NONE, SOURCE_NONE, 0]
        self._flags = dict(robust=1, use_type_filter=0, type_filter=0, compute_degrees=0, exact_only=0, tag_mode=TAG_REFERENCE)
        self._keep = []

    def close(self):
        if self._h:
            self._L.ftkx_tracker_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _ck(self, rc):
        _lib.check(rc, self._h, True)

    def set_domain(self, starts, sizes): self._ck(self._L.ftkx_tracker_set_domain(self._h, _lib.ll(starts), _lib.ll(sizes, fill=1)))
    def set_array_domain(self, starts, sizes): self._ck(self._L.ftkx_tracker_set_array_domain(self._h, _lib.ll(starts), _lib.ll(sizes, fill=1)))
    def set_scalar_field_source(self, s): self._src[0] = s
    def set_vector_field_source(self, s): self._src[1] = s
    def set_jacobian_field_source(self, s): self._src[2] = s
    def set_jacobian_symmetric(self, b): self._src[3] = int(b)
    def set_enable_robust_detection(self, b): self._flags["robust"] = int(b)
    def set_enable_computing_degrees(self, b): self._flags["compute_degrees"] = int(b)
    def set_type_filter(self, f): self._flags["use_type_filter"] = 1; self._flags["type_filter"] = int(f)
    def set_exact_only(self, b): self._flags["exact_only"] = int(b)
    def set_tag_mode(self, m): self._flags["tag_mode"] = int(m)
    def set_stream(self, ptr): self._ck(self._L.ftkx_tracker_set_stream(self._h, C.c_void_p(ptr)))
    def set_current_timestep(self, t): self._ck(self._L.ftkx_tracker_set_current_timestep(self._h, int(t)))
    def set_deferred_collection(self, b, depth=1):
        """the sweep of step t + 1 is queued before the records of step t are collected (ftkx_tracker.hh); depth > 1: the sweeps of `depth`
        consecutive steps are queued as one pass.  Pushed device tensors are kept alive until their batch has been collected"""
        self._deferred = bool(b)
        self._deferred_depth = max(1, int(depth)) if b else 1
        self._ck(self._L.ftkx_tracker_set_deferred_collection(self._h, self._deferred_depth if b else 0))

    # several ranks behind the tracker (include/ftkx_tracker.hh: slab mode): this rank's tracker takes the snapshots of its timestep slab
    # (tslab.slab_range), sweeps it as one device-driven pass, finalize() gathers the points on rank 0.  Pushed device tensors are kept alive.
    def set_communicator(self, nccl_comm, rank, nranks, nt):
        self._slab_keep = []
        self._ck(self._L.ftkx_tracker_set_communicator(self._h, nccl_comm, int(rank), int(nranks), int(nt)))

    def set_slab_hub(self, hub, rank, nt):
        self._slab_keep = []
        self._ck(self._L.ftkx_tracker_set_slab_hub(self._h, hub, int(rank), int(nt)))

    def set_enable_streaming_trajectories(self, b): self._ck(self._L.ftkx_tracker_set_enable_streaming_trajectories(self._h, int(bool(b))))
    def set_coords_bounds(self, b): self._ck(self._L.ftkx_tracker_set_coords_bounds(self._h, (C.c_double * len(b))(*[float(x) for x in b])))

    def set_coords_rectilinear(self, arrays):
        """REGULAR_COORDS_RECTILINEAR: one 1-D float64 array per axis, indexed by the vertex coordinate"""
        arrs = [np.ascontiguousarray(a, dtype=np.float64) for a in arrays] + [None] * (3 - len(arrays))
        args = []
        for a in arrs:
            args += [a.ctypes.data if a is not None else None, len(a) if a is not None else 0]
        self._ck(self._L.ftkx_tracker_set_coords_rectilinear(self._h, *args))

    def set_coords_explicit(self, coords):
        """REGULAR_COORDS_EXPLICIT: numpy array of shape (n1, n0, ncomp) = the reference's ndarray (ncomp, n0, n1)"""
        e = np.ascontiguousarray(coords, dtype=np.float64)
        self._ck(self._L.ftkx_tracker_set_coords_explicit(self._h, e.ctypes.data, e.shape[-1], e.shape[-2], e.shape[-3]))

    def initialize(self):
        self._ck(self._L.ftkx_tracker_set_sources(self._h, *self._src))
        f = self._flags
        self._ck(self._L.ftkx_tracker_set_flags(self._h, f["robust"], f["use_type_filter"], f["type_filter"], f["compute_degrees"], f["exact_only"], f["tag_mode"]))
        self._ck(self._L.ftkx_tracker_initialize(self._h))

    def push_scalar_field_snapshot(self, s):
        p, k, d = _ptr(s); self._keep.append(k)
        self._ck(self._L.ftkx_tracker_push_scalar_field_snapshot(self._h, p, d))

    def push_vector_field_snapshot(self, v):
        p, k, d = _ptr(v); self._keep.append(k)
        self._ck(self._L.ftkx_tracker_push_vector_field_snapshot(self._h, p, d))

    def push_field_data_snapshot(self, s, v, j):
        ps, ks, ds = _ptr(s); pv, kv, dv = _ptr(v); pj, kj, dj = _ptr(j); self._keep.append((ks, kv, kj))
        self._ck(self._L.ftkx_tracker_push_field_data_snapshot(self._h, ps, pv, pj, dv))

    def advance_timestep(self):
        self._ck(self._L.ftkx_tracker_advance_timestep(self._h))
        if hasattr(self, "_slab_keep"):             # slab mode: the snapshots stay resident until the slab's pass has run
            self._slab_keep += self._keep
        self._keep = self._keep[-((3 * getattr(self, "_deferred_depth", 1) + 2) if getattr(self, "_deferred", False) else 2):]

    def update_timestep(self): self._ck(self._L.ftkx_tracker_update_timestep(self._h))
    def sync(self): self._ck(self._L.ftkx_tracker_sync(self._h))

    def get_critical_points(self):
        """(records[CP_DTYPE], ordinal[int32], timestep[int32]) in the reference's std::map order (by element tag)"""
        n = C.c_size_t()
        self._ck(self._L.ftkx_tracker_num_critical_points(self._h, C.byref(n)))
        recs = np.zeros(n.value, dtype=CP_DTYPE); o = np.zeros(n.value, dtype=np.int32); ts = np.zeros(n.value, dtype=np.int32)
        if n.value:
            self._ck(self._L.ftkx_tracker_get_critical_points(self._h, recs.ctypes.data, o.ctypes.data, ts.ctypes.data, n.value))
        return recs, o, ts

    def finalize(self): self._ck(self._L.ftkx_tracker_finalize(self._h))

    def get_traced_critical_points(self):
        """after finalize(): (list of tag arrays, one per curve, in curve order; loop flags)"""
        nc, npts = C.c_size_t(), C.c_size_t()
        self._ck(self._L.ftkx_tracker_num_curves(self._h, C.byref(nc), C.byref(npts)))
        offs = np.zeros(nc.value + 1, dtype=np.int64); tags = np.zeros(max(1, npts.value), dtype=np.uint64); loop = np.zeros(max(1, nc.value), dtype=np.int32)
        self._ck(self._L.ftkx_tracker_get_curves(self._h, offs.ctypes.data, tags.ctypes.data, loop.ctypes.data))
        return [tags[offs[i]:offs[i + 1]] for i in range(nc.value)], loop[:nc.value]

    def post_process(self):
        """json_interface::post_process with its default options, on the traced curves"""
        self._ck(self._L.ftkx_tracker_post_process(self._h))

    def get_traced_trajectories(self):
        """after finalize() [+ post_process()]: list of (tags, types, t, loop, label) per curve"""
        curves, loop = self.get_traced_critical_points()
        npts = sum(len(c) for c in curves)
        ty = np.zeros(max(1, npts), dtype=np.uint32); tt = np.zeros(max(1, npts), dtype=np.float64); ids = np.zeros(max(1, len(curves)), dtype=np.int32)
        self._ck(self._L.ftkx_tracker_get_curve_points(self._h, ty.ctypes.data, tt.ctypes.data, ids.ctypes.data))
        out, k = [], 0
        for i, c in enumerate(curves):
            out.append((c, ty[k:k + len(c)], tt[k:k + len(c)], int(loop[i]), int(ids[i]))); k += len(c)
        return out

    def _write(self, path, fmt, traced): self._ck(self._L.ftkx_tracker_write(self._h, str(path).encode(), fmt, traced))
    def write_critical_points_json(self, path): self._write(path, FORMAT_JSON, 0)
    def write_critical_points_binary(self, path): self._write(path, FORMAT_BINARY, 0)
    def write_critical_points_text(self, path): self._write(path, FORMAT_TEXT, 0)
    def write_traced_critical_points_json(self, path): self._write(path, FORMAT_JSON, 1)
    def write_traced_critical_points_binary(self, path): self._write(path, FORMAT_BINARY, 1)
    def write_traced_critical_points_text(self, path): self._write(path, FORMAT_TEXT, 1)
    def read_critical_points_json(self, path): self._ck(self._L.ftkx_tracker_read_critical_points(self._h, str(path).encode(), FORMAT_JSON))
    def read_critical_points_binary(self, path): self._ck(self._L.ftkx_tracker_read_critical_points(self._h, str(path).encode(), FORMAT_BINARY))

    def get_vector_field_scaling_factor(self):
        f, r = C.c_ulonglong(), C.c_double()
        self._ck(self._L.ftkx_tracker_get_scaling(self._h, C.byref(f), C.byref(r)))
        return f.value

    def get_last_stats(self):
        s = Stats()
        self._ck(self._L.ftkx_tracker_get_stats(self._h, C.byref(s)))
        return {k: getattr(s, k) for k, _ in Stats._fields_}


class CriticalPointTracker2DRegular(_TrackerRegular):
    ND = 2


class CriticalPointTracker3DRegular(_TrackerRegular):
    ND = 3
