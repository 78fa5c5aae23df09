"""ctypes binding of ftk_amd/libftkx.so -- plumbing only.  There is no Python or CPU implementation of the sweep behind
these calls: if the shared library is missing the import fails, and if no GPU is visible ftkx_create() fails."""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libftkx.so")

# == ftkx_cp_t / ftk::feature_point_lite_t (72 bytes); `aux` sits in the C struct's padding (include/ftkx.h)
CP_DTYPE = np.dtype([("x", "<f8", (3,)), ("t", "<f8"), ("scalar", "<f8", (3,)), ("type", "<u4"), ("aux", "<u4"), ("tag", "<u8")])
assert CP_DTYPE.itemsize == 72

OK, E_INVALID, E_DEVICE, E_NOMEM, E_NOSLICE, E_UNSUPPORTED = 0, -1, -2, -3, -4, -5
SCOPE_ORDINAL, SCOPE_INTERVAL, SCOPE_BOTH = 1, 2, 3
TAG_WORK_INDEX, TAG_REFERENCE, TAG_EXACT64 = 0, 1, 2
FORMAT_BINARY, FORMAT_JSON, FORMAT_TEXT = 0, 1, 2
SOURCE_NONE, SOURCE_GIVEN, SOURCE_DERIVED = 0, 1, 2


class Options(C.Structure):
    _fields_ = [("jacobian_symmetric", C.c_int), ("robust", C.c_int), ("use_type_filter", C.c_int), ("type_filter", C.c_uint),
                ("compute_degrees", C.c_int), ("tag_mode", C.c_int), ("exact_only", C.c_int), ("derive_jacobian", C.c_int),
                ("coords_mode", C.c_int), ("coords_bounds", C.c_double * 6)]


class Stats(C.Structure):
    _fields_ = [("work_items", C.c_ulonglong), ("cells", C.c_ulonglong), ("cells_survived", C.c_ulonglong),
                ("simplices_tested", C.c_ulonglong), ("hits", C.c_ulonglong), ("cull_enabled", C.c_int), ("reclassified", C.c_ulonglong)]


class Curves(C.Structure):
    _fields_ = [("n_curves", C.c_size_t), ("n_points", C.c_size_t), ("n_special", C.c_size_t),
                ("offsets", C.POINTER(C.c_longlong)), ("indices", C.POINTER(C.c_longlong)), ("loop", C.POINTER(C.c_int))]


class Trajectories(C.Structure):
    _fields_ = [("n_curves", C.c_size_t), ("n_points", C.c_size_t), ("offsets", C.POINTER(C.c_longlong)), ("indices", C.POINTER(C.c_longlong)),
                ("loop", C.POINTER(C.c_int)), ("type", C.POINTER(C.c_uint)), ("t", C.POINTER(C.c_double)), ("id", C.POINTER(C.c_int))]


# include/ftkx_slab.h: the tables a caller hands to ftkx_slab_create / ftkx_slab_create_custom
AG_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p)
XCHG_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p)
DESTROY_FN = C.CFUNCTYPE(None, C.c_void_p)


class SlabTransport(C.Structure):
    _fields_ = [("user", C.c_void_p), ("all_gather", AG_FN), ("exchange", XCHG_FN), ("queued", C.c_int), ("destroy", DESTROY_FN)]


BEGIN_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_int), C.c_int, C.POINTER(C.c_double), C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p)
CULL_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_void_p)
FINISH_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p)
COMPLETE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_ulonglong), C.POINTER(C.c_void_p), C.POINTER(C.c_size_t))
STATUS_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(C.c_longlong), C.POINTER(C.c_longlong), C.POINTER(C.c_double), C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_ulonglong))
RECOVER_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_int), C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_ulonglong), C.POINTER(C.c_void_p),
                         C.POINTER(C.c_size_t))
FIRST_FN = C.CFUNCTYPE(C.c_void_p, C.c_void_p, C.c_int)
ALLOC_FN = C.CFUNCTYPE(C.c_void_p, C.c_void_p, C.c_size_t)
RELEASE_FN = C.CFUNCTYPE(None, C.c_void_p, C.c_void_p)
COPY_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t)
ABORT_FN = C.CFUNCTYPE(None, C.c_void_p)


class SlabBackend(C.Structure):
    _fields_ = [("user", C.c_void_p), ("begin", BEGIN_FN), ("cull", CULL_FN), ("serve", CULL_FN), ("finish", FINISH_FN), ("complete", COMPLETE_FN), ("status", STATUS_FN),
                ("recover", RECOVER_FN), ("first_slice", FIRST_FN), ("alloc", ALLOC_FN), ("release", RELEASE_FN), ("upload", COPY_FN), ("download", COPY_FN), ("abort", ABORT_FN),
                ("masks_bytes", C.c_size_t), ("cells", C.c_size_t), ("patch_doubles", C.c_size_t), ("slice_bytes", C.c_size_t), ("stream", C.c_void_p), ("device", C.c_int)]


class SlabInfo(C.Structure):
    _fields_ = [("rank", C.c_int), ("nranks", C.c_int), ("t0", C.c_int), ("t1", C.c_int), ("lower", C.c_int), ("upper", C.c_int), ("bytes_sent", C.c_ulonglong),
                ("bytes_received", C.c_ulonglong), ("fallbacks", C.c_int), ("last_asked", C.c_longlong), ("last_served", C.c_longlong), ("last_path", C.c_int),
                ("last_status", C.c_ulonglong), ("open", C.c_int)]


class FtkxError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"ftkx error {code}: {msg}")
        self.code = code


EXPORTS = [
    "ftkx_create", "ftkx_destroy", "ftkx_last_error", "ftkx_set_stream", "ftkx_set_options", "ftkx_default_options", "ftkx_set_mesh",
    "ftkx_push_slice", "ftkx_push_scalar_slice", "ftkx_drop_slice", "ftkx_slice_resolution", "ftkx_slices_resolution", "ftkx_slices_prepare", "ftkx_sweep_announce", "ftkx_set_slice_resolution", "ftkx_export_masks_size", "ftkx_export_masks", "ftkx_push_masked_slice", "ftkx_packed_masks_bytes", "ftkx_export_masks_packed", "ftkx_push_masked_slice_packed", "ftkx_sweep_cull", "ftkx_get_sparse_cells", "ftkx_patch_doubles", "ftkx_gather_patches", "ftkx_scatter_patches", "ftkx_scaling_factor",
    "ftkx_sweep", "ftkx_sweep_series", "ftkx_sweep_series_submit", "ftkx_sweep_series_complete", "ftkx_sweep_series_abort", "ftkx_series_dist_cells", "ftkx_series_dist_begin", "ftkx_series_dist_cull", "ftkx_series_dist_serve", "ftkx_series_dist_finish", "ftkx_series_dist_status", "ftkx_series_last_path", "ftkx_series_split_decision", "ftkx_sweep_enqueue", "ftkx_sweep_enqueue_many", "ftkx_sweep_collect", "ftkx_sweep_cancel", "ftkx_get_stats", "ftkx_invalidate_masks", "ftkx_debug_stream_read", "ftkx_debug_tile_repeat", "ftkx_set_profiling", "ftkx_get_kernel_times", "ftkx_extract_cp2dt", "ftkx_extract_cp3dt", "ftkx_free",
    "ftkx_trace_curves", "ftkx_trace_curves_ctx", "ftkx_trace_curves_tags_ctx", "ftkx_free_curves", "ftkx_post_process_curves", "ftkx_free_trajectories", "ftkx_format_from_path", "ftkx_write_critical_points", "ftkx_read_critical_points", "ftkx_write_traced_critical_points", "ftkx_read_traced_critical_points", "ftkx_gradient2D", "ftkx_jacobian2D", "ftkx_gradient3D", "ftkx_jacobian3D", "ftkx_version", "ftkx_device_count", "ftkx_pointer_device", "ftkx_context_device", "ftkx_last_mask_kernel", "ftkx_debug_mask_kernel_launches", "ftkx_debug_upload_counts", "ftkx_debug_mask_relaunch",
    "ftkx_tracker_post_process", "ftkx_tracker_get_curve_points", "ftkx_tracker_write", "ftkx_tracker_read_critical_points",
    "ftkx_tracker_create", "ftkx_tracker_create_multi", "ftkx_tracker_sync", "ftkx_tracker_destroy", "ftkx_tracker_last_error", "ftkx_tracker_set_domain", "ftkx_tracker_set_array_domain",
    "ftkx_tracker_set_sources", "ftkx_tracker_set_flags", "ftkx_tracker_set_stream", "ftkx_tracker_set_current_timestep", "ftkx_tracker_set_enable_streaming_trajectories", "ftkx_tracker_set_deferred_collection", "ftkx_tracker_set_communicator", "ftkx_tracker_set_slab_transport", "ftkx_tracker_set_slab_hub", "ftkx_online_tracer_create", "ftkx_online_tracer_destroy", "ftkx_online_tracer_grow", "ftkx_online_tracer_curves", "ftkx_tracker_set_coords_bounds", "ftkx_tracker_set_coords_rectilinear", "ftkx_tracker_set_coords_explicit", "ftkx_set_coords_rectilinear", "ftkx_set_coords_explicit", "ftkx_tracker_initialize",
    "ftkx_tracker_push_scalar_field_snapshot", "ftkx_tracker_push_vector_field_snapshot", "ftkx_tracker_push_field_data_snapshot",
    "ftkx_tracker_advance_timestep", "ftkx_tracker_update_timestep", "ftkx_tracker_num_critical_points",
    "ftkx_tracker_get_critical_points", "ftkx_tracker_get_scaling", "ftkx_tracker_get_stats",
    "ftkx_tracker_finalize", "ftkx_tracker_num_curves", "ftkx_tracker_get_curves",
    # include/ftkx_slab.h
    "ftkx_slab_range", "ftkx_slab_owner", "ftkx_slab_create", "ftkx_slab_create_custom", "ftkx_slab_create_rccl", "ftkx_slab_destroy", "ftkx_slab_set_periodic", "ftkx_slab_submit", "ftkx_slab_complete",
    "ftkx_slab_gather_records", "ftkx_slab_get_info", "ftkx_slab_last_error", "ftkx_rccl_unique_id", "ftkx_rccl_comm_create", "ftkx_rccl_comm_destroy", "ftkx_rccl_version",
    "ftkx_slab_transport_rccl", "ftkx_upload", "ftkx_download", "ftkx_slab_hub_create", "ftkx_slab_hub_destroy", "ftkx_slab_create_local", "ftkx_slab_hub_abort",
]

_L = None


def load():
    """Loads the shared library (no GPU needed for that) and declares the prototypes."""
    global _L
    if _L is not None:
        return _L
    if not os.path.exists(LIB_PATH):
        raise ImportError(f"{LIB_PATH} is missing: build it with `python -m ftk_amd.build` (hipcc --offload-arch=gfx950)")
    # PyTorch (this package's plumbing: device memory, streams, torch.distributed) ships a HIP runtime of its own.  Whichever runtime is
    # loaded first serves the whole process, and with the system's loaded first torch finds no device afterwards ("no ROCm-capable device",
    # seen when __graft_entry__.build() loaded this library before smoke() imported torch): torch goes first.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    L = C.CDLL(LIB_PATH)
    vp, ll3, dbl = C.c_void_p, C.POINTER(C.c_longlong), C.c_void_p
    L.ftkx_create.argtypes = [C.POINTER(vp), C.c_int, C.c_int]
    L.ftkx_destroy.argtypes = [vp]; L.ftkx_destroy.restype = None
    L.ftkx_last_error.argtypes = [vp, C.c_char_p, C.c_size_t]
    L.ftkx_set_stream.argtypes = [vp, vp]
    L.ftkx_set_options.argtypes = [vp, C.POINTER(Options)]
    L.ftkx_default_options.argtypes = [C.POINTER(Options)]; L.ftkx_default_options.restype = None
    L.ftkx_set_mesh.argtypes = [vp] + [ll3] * 6
    L.ftkx_push_slice.argtypes = [vp, C.c_int, dbl, dbl, dbl, C.c_int]
    L.ftkx_push_scalar_slice.argtypes = [vp, C.c_int, dbl, C.c_int]
    L.ftkx_drop_slice.argtypes = [vp, C.c_int]
    L.ftkx_slice_resolution.argtypes = [vp, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_double)]
    L.ftkx_set_slice_resolution.argtypes = [vp, C.c_int, C.c_double, C.c_double]
    L.ftkx_scaling_factor.argtypes = [C.c_double, C.POINTER(C.c_int)]; L.ftkx_scaling_factor.restype = C.c_ulonglong
    L.ftkx_sweep.argtypes = [vp, C.c_int, C.c_int, C.c_ulonglong, C.POINTER(vp), C.POINTER(C.c_size_t)]
    L.ftkx_sweep_enqueue.argtypes = [vp, C.c_int, C.c_int, C.c_ulonglong]
    L.ftkx_sweep_collect.argtypes = [vp, C.POINTER(vp), C.POINTER(C.c_size_t)]
    L.ftkx_get_stats.argtypes = [vp, C.POINTER(Stats)]
    L.ftkx_set_profiling.argtypes = [vp, C.c_int]
    L.ftkx_invalidate_masks.argtypes = [vp]
    L.ftkx_debug_stream_read.argtypes = [vp, vp, C.c_size_t]
    L.ftkx_debug_tile_repeat.argtypes = [vp, C.c_int]
    L.ftkx_debug_upload_counts.argtypes = [vp, C.POINTER(C.c_ulonglong), C.POINTER(C.c_ulonglong)]
    L.ftkx_debug_mask_relaunch.argtypes = [vp, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_double)]
    L.ftkx_series_split_decision.argtypes = [vp, C.POINTER(C.c_int), C.POINTER(C.c_double), C.POINTER(C.c_double)]
    L.ftkx_get_kernel_times.argtypes = [vp, C.POINTER(C.c_double), C.POINTER(C.c_ulonglong)]
    L.ftkx_extract_cp2dt.argtypes = [C.c_int, C.c_int] + [ll3] * 6 + [dbl] * 6 + [C.c_int, dbl, C.c_ulonglong, C.POINTER(Options), C.c_int, C.POINTER(vp), C.POINTER(C.c_size_t)]
    L.ftkx_extract_cp3dt.argtypes = [C.c_int, C.c_int] + [ll3] * 6 + [dbl] * 6 + [C.c_ulonglong, C.POINTER(Options), C.c_int, C.POINTER(vp), C.POINTER(C.c_size_t)]
    L.ftkx_free.argtypes = [vp]; L.ftkx_free.restype = None
    L.ftkx_trace_curves.argtypes = [C.c_int, ll3, ll3, vp, C.c_size_t, C.POINTER(Curves)]
    L.ftkx_trace_curves_ctx.argtypes = [vp, C.c_int, ll3, ll3, vp, C.c_size_t, C.POINTER(Curves)]
    L.ftkx_free_curves.argtypes = [C.POINTER(Curves)]; L.ftkx_free_curves.restype = None
    L.ftkx_post_process_curves.argtypes = [vp, C.c_size_t, C.POINTER(Curves), C.POINTER(Trajectories)]
    L.ftkx_free_trajectories.argtypes = [C.POINTER(Trajectories)]; L.ftkx_free_trajectories.restype = None
    L.ftkx_format_from_path.argtypes = [C.c_char_p]
    L.ftkx_write_critical_points.argtypes = [C.c_char_p, C.c_int, vp, C.c_size_t, vp, vp, C.POINTER(C.c_char_p), C.c_int]
    L.ftkx_read_critical_points.argtypes = [C.c_char_p, C.c_int, C.POINTER(vp), C.POINTER(C.c_size_t), C.POINTER(vp), C.POINTER(vp)]
    L.ftkx_write_traced_critical_points.argtypes = [C.c_char_p, C.c_int, vp, C.c_size_t, C.POINTER(Trajectories), C.POINTER(C.c_char_p), C.c_int]
    L.ftkx_read_traced_critical_points.argtypes = [C.c_char_p, C.c_int, C.POINTER(vp), C.POINTER(C.c_size_t), C.POINTER(Trajectories)]
    L.ftkx_gradient2D.argtypes = [vp, dbl, C.c_int, C.c_int, dbl]
    L.ftkx_jacobian2D.argtypes = [vp, dbl, C.c_int, C.c_int, C.c_int, dbl]
    L.ftkx_gradient3D.argtypes = [vp, dbl, C.c_int, C.c_int, C.c_int, dbl]
    L.ftkx_jacobian3D.argtypes = [vp, dbl, C.c_int, C.c_int, C.c_int, dbl]
    L.ftkx_version.restype = C.c_char_p
    L.ftkx_tracker_create.argtypes = [C.POINTER(vp), C.c_int, C.c_int]
    L.ftkx_tracker_destroy.argtypes = [vp]; L.ftkx_tracker_destroy.restype = None
    L.ftkx_tracker_last_error.argtypes = [vp, C.c_char_p, C.c_size_t]
    L.ftkx_tracker_set_domain.argtypes = [vp, ll3, ll3]
    L.ftkx_tracker_set_array_domain.argtypes = [vp, ll3, ll3]
    L.ftkx_tracker_set_sources.argtypes = [vp, C.c_int, C.c_int, C.c_int, C.c_int]
    L.ftkx_tracker_set_flags.argtypes = [vp, C.c_int, C.c_int, C.c_uint, C.c_int, C.c_int, C.c_int]
    L.ftkx_tracker_set_stream.argtypes = [vp, vp]
    L.ftkx_tracker_create_multi.argtypes = [C.POINTER(vp), C.c_int, C.POINTER(C.c_int), C.c_int, C.c_int]
    L.ftkx_tracker_sync.argtypes = [vp]
    L.ftkx_tracker_set_current_timestep.argtypes = [vp, C.c_int]
    L.ftkx_tracker_set_enable_streaming_trajectories.argtypes = [vp, C.c_int]
    L.ftkx_tracker_set_deferred_collection.argtypes = [vp, C.c_int]
    L.ftkx_tracker_set_communicator.argtypes = [vp, vp, C.c_int, C.c_int, C.c_int]
    L.ftkx_tracker_set_slab_transport.argtypes = [vp, C.POINTER(SlabTransport), C.c_int, C.c_int, C.c_int]
    L.ftkx_tracker_set_slab_hub.argtypes = [vp, vp, C.c_int, C.c_int]
    L.ftkx_online_tracer_create.argtypes = [C.POINTER(vp), C.c_int, ll3, ll3]
    L.ftkx_online_tracer_destroy.argtypes = [vp]
    L.ftkx_online_tracer_destroy.restype = None
    L.ftkx_online_tracer_grow.argtypes = [vp, vp, C.c_size_t]
    L.ftkx_online_tracer_curves.argtypes = [vp, C.POINTER(vp), C.POINTER(Curves)]
    L.ftkx_tracker_set_coords_bounds.argtypes = [vp, C.POINTER(C.c_double)]
    L.ftkx_tracker_initialize.argtypes = [vp]
    L.ftkx_tracker_push_scalar_field_snapshot.argtypes = [vp, dbl, C.c_int]
    L.ftkx_tracker_push_vector_field_snapshot.argtypes = [vp, dbl, C.c_int]
    L.ftkx_tracker_push_field_data_snapshot.argtypes = [vp, dbl, dbl, dbl, C.c_int]
    L.ftkx_tracker_advance_timestep.argtypes = [vp]
    L.ftkx_tracker_update_timestep.argtypes = [vp]
    L.ftkx_tracker_num_critical_points.argtypes = [vp, C.POINTER(C.c_size_t)]
    L.ftkx_tracker_get_critical_points.argtypes = [vp, vp, vp, vp, C.c_size_t]
    L.ftkx_tracker_get_scaling.argtypes = [vp, C.POINTER(C.c_ulonglong), C.POINTER(C.c_double)]
    L.ftkx_tracker_get_stats.argtypes = [vp, C.POINTER(Stats)]
    L.ftkx_tracker_finalize.argtypes = [vp]
    L.ftkx_tracker_num_curves.argtypes = [vp, C.POINTER(C.c_size_t), C.POINTER(C.c_size_t)]
    L.ftkx_tracker_get_curves.argtypes = [vp, vp, vp, vp]
    L.ftkx_tracker_post_process.argtypes = [vp]
    L.ftkx_last_mask_kernel.restype = C.c_char_p; L.ftkx_last_mask_kernel.argtypes = []
    L.ftkx_slices_resolution.argtypes = [vp, vp, C.c_int, vp, vp]
    L.ftkx_slices_prepare.argtypes = [vp, vp, C.c_int, C.c_ulonglong, vp, vp]
    L.ftkx_sweep_announce.argtypes = [vp, vp, vp, C.c_int]
    L.ftkx_export_masks_size.argtypes = [vp, C.c_int, C.POINTER(C.c_size_t), C.POINTER(C.c_size_t), C.POINTER(C.c_ulonglong), C.POINTER(C.c_double)]
    L.ftkx_export_masks.argtypes = [vp, C.c_int, vp, vp, vp, C.c_int]
    L.ftkx_push_masked_slice.argtypes = [vp, C.c_int, C.c_int, vp, C.c_size_t, vp, vp, C.c_size_t, C.c_ulonglong, C.c_double, C.c_int]
    L.ftkx_packed_masks_bytes.argtypes = [vp, C.POINTER(C.c_size_t)]
    L.ftkx_packed_masks_bytes.restype = C.c_size_t
    L.ftkx_export_masks_packed.argtypes = [vp, C.c_int, vp, C.c_int]
    L.ftkx_push_masked_slice_packed.argtypes = [vp, C.c_int, C.c_int, vp, C.c_int, C.c_ulonglong, C.c_double]
    L.ftkx_sweep_cull.argtypes = [vp, C.c_int, C.POINTER(C.c_size_t)]
    L.ftkx_sweep_cancel.argtypes = [vp]
    L.ftkx_sweep_enqueue_many.argtypes = [vp, vp, vp, vp, C.c_int]
    L.ftkx_sweep_series.argtypes = [vp, vp, vp, C.c_int, C.POINTER(C.c_double), vp, C.POINTER(vp), C.POINTER(C.c_size_t)]
    L.ftkx_sweep_series_submit.argtypes = [vp, vp, vp, C.c_int, vp]
    L.ftkx_sweep_series_complete.argtypes = [vp, C.POINTER(C.c_double), vp, C.POINTER(vp), C.POINTER(C.c_size_t)]
    L.ftkx_sweep_series_abort.argtypes = [vp]
    L.ftkx_series_dist_cells.argtypes = [vp]
    L.ftkx_series_dist_cells.restype = C.c_size_t
    L.ftkx_series_dist_begin.argtypes = [vp, vp, vp, C.c_int, C.POINTER(C.c_double), C.c_int, C.c_int, C.c_int, vp, vp, vp, vp]
    L.ftkx_series_dist_cull.argtypes = [vp, vp, vp]
    L.ftkx_series_dist_serve.argtypes = [vp, vp, vp]
    L.ftkx_series_dist_finish.argtypes = [vp, vp]
    L.ftkx_series_dist_status.argtypes = [vp, C.POINTER(C.c_longlong), C.POINTER(C.c_longlong), vp, C.c_int]
    L.ftkx_series_last_path.argtypes = [vp, C.POINTER(C.c_ulonglong)]
    L.ftkx_get_sparse_cells.argtypes = [vp, vp, C.c_int]
    L.ftkx_patch_doubles.argtypes = [vp]
    L.ftkx_patch_doubles.restype = C.c_size_t
    L.ftkx_gather_patches.argtypes = [vp, C.c_int, vp, C.c_size_t, vp, C.c_int]
    L.ftkx_scatter_patches.argtypes = [vp, C.c_int, vp, C.c_size_t, vp, C.c_int]
    L.ftkx_set_coords_rectilinear.argtypes = [vp, vp, C.c_size_t, vp, C.c_size_t, vp, C.c_size_t]
    L.ftkx_set_coords_explicit.argtypes = [vp, vp, C.c_int, C.c_size_t, C.c_size_t]
    L.ftkx_tracker_set_coords_rectilinear.argtypes = [vp, vp, C.c_size_t, vp, C.c_size_t, vp, C.c_size_t]
    L.ftkx_tracker_set_coords_explicit.argtypes = [vp, vp, C.c_int, C.c_size_t, C.c_size_t]
    L.ftkx_tracker_get_curve_points.argtypes = [vp, vp, vp, vp]
    L.ftkx_tracker_write.argtypes = [vp, C.c_char_p, C.c_int, C.c_int]
    L.ftkx_tracker_read_critical_points.argtypes = [vp, C.c_char_p, C.c_int]
    L.ftkx_slab_range.argtypes = [C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int)]; L.ftkx_slab_range.restype = None
    L.ftkx_slab_owner.argtypes = [C.c_int, C.c_int, C.c_int]
    L.ftkx_slab_create.argtypes = [vp, C.c_int, C.c_int, C.c_int, C.POINTER(SlabTransport), C.POINTER(vp)]
    L.ftkx_slab_create_custom.argtypes = [C.POINTER(SlabBackend), C.c_int, C.c_int, C.c_int, C.POINTER(SlabTransport), C.POINTER(vp)]
    L.ftkx_slab_create_rccl.argtypes = [vp, C.c_int, C.c_int, C.c_int, vp, vp, C.POINTER(vp)]
    L.ftkx_slab_destroy.argtypes = [vp]; L.ftkx_slab_destroy.restype = None
    L.ftkx_slab_set_periodic.argtypes = [vp, C.c_int]
    L.ftkx_slab_submit.argtypes = [vp, C.POINTER(C.c_double)]
    L.ftkx_slab_complete.argtypes = [vp, C.POINTER(C.c_double), vp, C.POINTER(vp), C.POINTER(C.c_size_t)]
    L.ftkx_slab_gather_records.argtypes = [vp, vp, C.c_size_t, C.c_int, C.POINTER(vp), C.POINTER(C.c_size_t)]
    L.ftkx_slab_get_info.argtypes = [vp, C.POINTER(SlabInfo)]
    L.ftkx_slab_last_error.argtypes = [vp]; L.ftkx_slab_last_error.restype = C.c_char_p
    L.ftkx_rccl_unique_id.argtypes = [vp]
    L.ftkx_rccl_comm_create.argtypes = [vp, C.c_int, C.c_int, C.c_int, C.POINTER(vp)]
    L.ftkx_upload.argtypes = [vp, vp, vp, C.c_size_t]
    L.ftkx_download.argtypes = [vp, vp, vp, C.c_size_t]
    L.ftkx_rccl_comm_destroy.argtypes = [vp]; L.ftkx_rccl_comm_destroy.restype = None
    L.ftkx_rccl_version.argtypes = []
    L.ftkx_slab_transport_rccl.argtypes = [vp, vp, C.POINTER(SlabTransport)]
    L.ftkx_slab_hub_create.argtypes = [C.c_int]; L.ftkx_slab_hub_create.restype = vp
    L.ftkx_slab_hub_destroy.argtypes = [vp]; L.ftkx_slab_hub_destroy.restype = None
    L.ftkx_slab_hub_abort.argtypes = [vp]; L.ftkx_slab_hub_abort.restype = None
    L.ftkx_slab_create_local.argtypes = [vp, C.c_int, C.c_int, vp, C.POINTER(vp)]
    _L = L
    return L


def ll(values, n=3, fill=0):
    v = [int(x) for x in values] + [fill] * (n - len(values))
    return (C.c_longlong * n)(*v)


def last_error(handle=None, tracker=False):
    L = load()
    buf = C.create_string_buffer(512)
    (L.ftkx_tracker_last_error if tracker else L.ftkx_last_error)(handle, buf, 512)
    return buf.value.decode(errors="replace")


def check(rc, handle=None, tracker=False):
    if rc != OK:
        raise FtkxError(rc, last_error(handle, tracker))


def records_from(ptr, n, copy=True):
    if n == 0 or not ptr:
        return np.zeros(0, dtype=CP_DTYPE)
    buf = (C.c_char * (n * CP_DTYPE.itemsize)).from_address(ptr)
    a = np.frombuffer(buf, dtype=CP_DTYPE)
    return a.copy() if copy else a
