/* ftkx -- C ABI of the MI355X-native critical-point space-time simplex sweep.
 *
 * This is the drop-in boundary for ONE hot path of hguo/ftk: the per-timestep sweep of
 *   ftk::critical_point_tracker_2d_regular::update_timestep()   include/ftk/filters/critical_point_tracker_2d_regular.hh:263-433
 *   ftk::critical_point_tracker_3d_regular::update_timestep()   include/ftk/filters/critical_point_tracker_3d_regular.hh:150-308
 * i.e. what the reference hands to its accelerator back-ends through
 *   extract_cp2dt_cuda / extract_cp2dt_sycl   critical_point_tracker_2d_regular.hh:33-63  (call sites 369-384, 399-414)
 *   extract_cp3dt_cuda                        critical_point_tracker_3d_regular.hh:42-56  (call sites 248-260, 274-286)
 * Results follow the reference's CPU path (check_simplex, 2d:584-685, 3d:425-514), not its CUDA/SYCL kernels
 * (those use a different domain and quantisation).
 *
 * Plain C: opaque context, raw pointers and sizes, integer status codes.  No STL, no torch types.  All entry points
 * return FTKX_OK (0) or a negative FTKX_E_* code and never call exit(); ftkx_last_error() gives the message.
 * (The reference's own boundary returns nothing and prints CUDA errors: src/filters/utils.cuh:13-22.)
 *
 * Array layouts are the reference's ndarray layouts (first index fastest, include/ftk/ndarray.hh:103-117):
 *   V[c + nd*(x + DW*(y + DH*z))]            vector field, nd components
 *   J[k + nd*j + nd*nd*(x + DW*(y + DH*z))]  jacobian; the tracker reads Js[j][k] = J(k, j, ...)  (2d:566-582, 3d:405-422)
 *   S[x + DW*(y + DH*z)]                     scalar
 */
#ifndef FTKX_H
#define FTKX_H

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

/* == ftk::feature_point_lite_t (include/ftk/features/feature_point_lite.hh:8-15), 72 bytes */
typedef struct ftkx_cp_t {
  double x[3];              /* lattice coordinates (REGULAR_COORDS_SIMPLE); 2D: x[2] = 0 */
  double t;
  double scalar[3];         /* scalar[0] filled iff a scalar field was given (FTK_CP_MAX_NUM_VARS = 3) */
  unsigned int type;        /* include/ftk/numeric/critical_point_type.hh:10-36 */
  unsigned long long tag;   /* see ftkx_options.tag_mode */
} ftkx_cp_t;

/* The 4 padding bytes between `type` and `tag` (offset 60; feature_point_lite_t has the same hole) carry what the caller
 * of the reference boundary otherwise has to remember per call (2d:387-395): bit 0 = the simplex type is ordinal,
 * bits 1..31 = current_timestep of the sweep that emitted the record.  Consumers that ignore padding are unaffected. */
static inline unsigned int ftkx_cp_aux(const ftkx_cp_t *cp) { return ((const unsigned int *)cp)[15]; }
static inline int ftkx_cp_ordinal(const ftkx_cp_t *cp) { return (int)(ftkx_cp_aux(cp) & 1u); }
static inline int ftkx_cp_timestep(const ftkx_cp_t *cp) { return (int)(ftkx_cp_aux(cp) >> 1); }

typedef struct ftkx_ctx ftkx_ctx;

enum {
  FTKX_OK = 0,
  FTKX_E_INVALID = -1,      /* bad argument / call order */
  FTKX_E_DEVICE = -2,       /* HIP runtime error */
  FTKX_E_NOMEM = -3,
  FTKX_E_NOSLICE = -4,      /* sweep of a timestep whose slice is not resident */
  FTKX_E_UNSUPPORTED = -5
};

/* ELEMENT_SCOPE_* of include/ftk/mesh/simplicial_regular_mesh.hh:39-43; BOTH = one launch doing both sweeps of a step */
enum { FTKX_SCOPE_ORDINAL = 1, FTKX_SCOPE_INTERVAL = 2, FTKX_SCOPE_BOTH = 3 };

enum {
  FTKX_TAG_WORK_INDEX = 0,  /* index inside `core` for the given scope: what extract_cp*dt_cuda returns (src/filters/critical_point_tracer_2d_regular.cu:162-164).
                               FTKX_SCOPE_BOTH: ftkx_sweep_series only; every record counts inside its OWN scope (ftkx_cp_ordinal says which),
                               so tags of the two scopes may coincide and the records come in element order, not in tag order */
  FTKX_TAG_REFERENCE  = 1,  /* e.to_integer(m) bit-for-bit, including its int32 products (simplicial_regular_mesh.hh:496-502) */
  FTKX_TAG_EXACT64    = 2   /* same formula in 64-bit arithmetic: equal to REFERENCE whenever that does not overflow */
};

typedef struct ftkx_options {
  int jacobian_symmetric;   /* is_jacobian_field_symmetric (critical_point_tracker.hh:176) */
  int robust;               /* enable_robust_detection, 3D only (3d:439-467); 2D always runs the robust test (2d:618-622) */
  int use_type_filter;      /* 2D only (2d:280) */
  unsigned int type_filter;
  int compute_degrees;      /* enable_computing_degrees, 2D only (2d:653-662) */
  int tag_mode;             /* FTKX_TAG_* */
  int exact_only;           /* 1: never apply the sign cull (every simplex gets the full integer test) */
  int derive_jacobian;      /* 1: when a slice has no J, derive it at hit vertices from V exactly like ndarray/grad.hh
                               jacobian2D/3D would (jacobian_field_source == SOURCE_DERIVED); 0: treat J as absent (zeros) */
  int coords_mode;          /* REGULAR_COORDS_* (regular_tracker.hh:12-17; simplex_coordinates 2d:494-527, 3d:342-378):
                               0 SIMPLE (lattice integers), 1 BOUNDS (coords_bounds below), 2 RECTILINEAR and 3 EXPLICIT --
                               the last two are selected by ftkx_set_coords_rectilinear / ftkx_set_coords_explicit, which
                               also carry the arrays */
  double coords_bounds[6];  /* x0,x1,y0,y1[,z0,z1] */
} ftkx_options;

/* ---- context ------------------------------------------------------------------------------------------------- */
int  ftkx_create(ftkx_ctx **ctx, int nd /*2|3*/, int device_id);
void ftkx_destroy(ftkx_ctx *ctx);
int  ftkx_last_error(const ftkx_ctx *ctx, char *buf, size_t n);   /* ctx may be NULL: last error of the calling thread */
int  ftkx_set_stream(ftkx_ctx *ctx, void *hip_stream);            /* NULL = the context's own stream */
int  ftkx_set_options(ftkx_ctx *ctx, const ftkx_options *opt);
void ftkx_default_options(ftkx_options *opt);

/* mesh: `domain` = vertex validity box == tracker `domain` (regular_tracker.hh:116-124; inclusive upper bound st+sz-1,
 * time is [0, INT_MAX]); `core` = corners to enumerate == local_domain; `ext` = array lattice == local_array_domain. */
int ftkx_set_mesh(ftkx_ctx *ctx, const long long domain_st[3], const long long domain_sz[3],
                  const long long core_st[3], const long long core_sz[3],
                  const long long ext_st[3], const long long ext_sz[3]);
/* regular_tracker::set_coords_rectilinear / set_coords_explicit (regular_tracker.hh:39-40).  Host arrays, copied to the device;
 * they replace opt.coords_mode.  Rectilinear: one array per spatial axis, indexed by the vertex coordinate (so n_d must cover the
 * array lattice).  Explicit: ndarray (ncomp, n0, n1) with the first index fastest, read as coords[c + ncomp * (x + n0 * y)];
 * ncomp >= 2 (2D; a third component becomes x[2]) or 3 (3D -- where the reference reads only this z = 0 plane and reports the
 * vertex's z index as its time; reproduced).  ftkx_set_options with coords_mode 0 or 1 switches back. */
int ftkx_set_coords_rectilinear(ftkx_ctx *ctx, const double *x, size_t nx, const double *y, size_t ny, const double *z, size_t nz);
int ftkx_set_coords_explicit(ftkx_ctx *ctx, const double *coords, int ncomp, size_t n0, size_t n1);

/* ---- slices resident in HBM (== field_data_snapshots, critical_point_tracker.hh:155-159) --------------------- */
/* V required; J, S nullable.  on_device = 0: host pointers, copied to the device; 1: device pointers (of this context's device),
 * adopted without a copy and owned by the caller until ftkx_drop_slice(); 2: device pointers of any device, copied (peer copy).
 * Host arrays are in HBM when the call returns (the caller may reuse or free them); pageable ones of 32 MiB and more are staged by a few
 * threads the call starts and joins (FTKX_UPLOAD_THREADS, default 4; 0: the runtime's own copy), pinned ones go by DMA as they are. */
int ftkx_push_slice(ftkx_ctx *ctx, int t, const double *V, const double *J, const double *S, int on_device);
/* scalar input (vector_field_source == SOURCE_DERIVED, 2d:238-250, 3d:125-137): uploads S and derives V = gradient2D/3D(S)
 * on the device bit-for-bit like ndarray/grad.hh.  With options.derive_jacobian the Jacobian at hit vertices is the one
 * jacobian2D<T, true> / jacobian3D would give for that V. */
int ftkx_push_scalar_slice(ftkx_ctx *ctx, int t, const double *S, int on_device);
int ftkx_drop_slice(ftkx_ctx *ctx, int t);
/* ndarray::resolution() of the slice's V (ndarray.hh:770-778): min |v| over non-zero entries (DBL_MAX if none);
 * max_abs (nullable) = max finite |v|, used for the no-overflow guard of the cull. */
int ftkx_slice_resolution(ftkx_ctx *ctx, int t, double *resolution, double *max_abs);
/* A slice received from another GPU (t-slab halo) comes with the reduction its owner already did: hand it over instead of
 * reducing the slice again. */
/* ftkx_slice_resolution for n resident slices with one launch and one synchronise; res / max_abs: n doubles each (nullable) */
int ftkx_slices_resolution(ftkx_ctx *ctx, const int *timesteps, int n, double *res, double *max_abs);
int ftkx_set_slice_resolution(ftkx_ctx *ctx, int t, double resolution, double max_abs);
/* The ONE-PASS form of the pre-pass (what the tracker and bench.py use): a single kernel reads each slice once and produces
 * both the vertex sign masks of the sweep and the reduction update_vector_field_scaling_factor needs, so a slice is read from
 * HBM once per sweep instead of twice.  The masks are built under `factor_hint`, a power of two that must not exceed the factor
 * the sweeps will be given (0 = 256, the smallest factor there is; a streaming caller passes the factor in force before these
 * slices arrived: the reference's factor is a sticky running minimum and only grows).  Masks built under a smaller factor only
 * ever cull less, never wrongly; ftkx_sweep rebuilds them by itself in the one case that is not covered (a larger factor
 * under which the slice has vertices that could overflow a determinant).
 *   res_below[i] = smallest non-zero |v| of slice i that is < 1 / factor_hint, DBL_MAX if there is none.  That is all the scaling
 *                  factor needs: running = min(running, res_below[i]) gives the same nbits = clamp(ceil(log2(1 / running)), 8, 21)
 *                  as the running minimum of ndarray::resolution() (a value >= 1 / factor_hint cannot raise nbits past log2 hint).
 *   max_abs[i]   = max finite |v| (both nullable). */
int ftkx_slices_prepare(ftkx_ctx *ctx, const int *timesteps, int n, unsigned long long factor_hint, double *res_below, double *max_abs);
/* Cull-ahead (optional): the sweeps that will be enqueued after the NEXT ftkx_slices_prepare, in that order.  That call then queues
 * their cull right behind the mask kernel -- the cull needs the masks and the list of steps, not the factor -- so it runs while the
 * host still waits for the reduction and forms the factors (update_vector_field_scaling_factor, critical_point_tracker.hh:850-864).
 * A hint, never an obligation: ftkx_sweep_collect takes the survivor list over only if the pending sweeps are exactly the announced
 * ones and every mask serves its sweep's factor; otherwise, and after any call that touches slices or masks, it culls as usual. */
int ftkx_sweep_announce(ftkx_ctx *ctx, const int *timesteps, const int *scopes, int n);
/* update_vector_field_scaling_factor (critical_point_tracker.hh:850-864): nbits = clamp(ceil(log2(1/res)), 8, 21) */
unsigned long long ftkx_scaling_factor(double resolution, int *nbits);

/* ---- compact t-slab halo (multi-GPU, one process per GPU: DESIGN.md 6) ------------------------------------------------------
 * A rank's last interval sweep [t-1, t] reads the first slice of the NEXT rank's slab.  Instead of that slice (1 GiB for 512^3) its
 * owner can hand over what the sweep really uses of it: the sign masks -- factor-free since round 2: the summary array as it is
 * plus the compacted mask words the summaries do not describe -- and, after the receiver's cull, the input values around the few
 * cells that survived there (6^nd vertices per cell: corner - 2 .. corner + 3 on every axis).
 *   owner:     ftkx_slices_prepare(t) ... ftkx_export_masks_size(t) -> sizes; ftkx_export_masks(t) -> buffers to send
 *   receiver:  ftkx_push_masked_slice(t, ...); ftkx_sweep_enqueue(...) as usual; ftkx_sweep_cull(t) -> n cells;
 *              ftkx_get_sparse_cells -> send to the owner
 *   owner:     ftkx_gather_patches(t, cells) -> send back          receiver: ftkx_scatter_patches(t, cells, patches); ftkx_sweep_collect
 * `*_on_device` = 1: the caller's buffers are device memory of this context's device; 0: host memory.
 * Limits: meshes whose masks carry summaries (even row length multiple of 8, slices < 4 GiB) and sweeps that use the cull; a masked
 * slice whose masks cannot serve the factor of the sweep (FTKX_E_NOSLICE) or any other limit (FTKX_E_UNSUPPORTED) means: send the
 * slice itself (ftkx_push_scalar_slice). */
int ftkx_export_masks_size(ftkx_ctx *ctx, int t, size_t *u_bytes, size_t *n_words, unsigned long long *mask_factor, double *max_abs);
int ftkx_export_masks(ftkx_ctx *ctx, int t, void *U_dst, unsigned *word_index_dst, unsigned long long *words_dst, int dst_on_device);
/* (u_bytes: the size of the summary array as the SENDER exported it; it must equal this context's -- same extents, same mask settings) */
int ftkx_push_masked_slice(ftkx_ctx *ctx, int t, int scalar_input, const void *U, size_t u_bytes, const unsigned *word_index, const unsigned long long *words, size_t n_words,
                           unsigned long long mask_factor, double max_abs, int on_device);
/* The same hand-over as ONE message of a size both sides know from the mesh alone (ftkx_packed_masks_bytes: 32-byte header, the summary
 * array, a capacity-bounded list of mask words).  The owner's export is queued on its stream and the receiver's import reads the header
 * -- word count, geometry -- ON THE DEVICE: neither side waits on the host for the other's numbers.  A message that did not fit (more
 * words than its capacity, another geometry, an index outside the mask array) is found by ftkx_sweep_cull: FTKX_E_NOSLICE, send the
 * slice itself.  mask_factor / max_abs: what the owner prepared the slice under (the factor hint) and its max |v| (it is part of the
 * all_gather of the reductions every rank takes part in anyway). */
size_t ftkx_packed_masks_bytes(const ftkx_ctx *ctx, size_t *word_capacity);
int ftkx_export_masks_packed(ftkx_ctx *ctx, int t, void *dst, int dst_on_device);
int ftkx_push_masked_slice_packed(ftkx_ctx *ctx, int t, int scalar_input, const void *src, int src_on_device, unsigned long long mask_factor, double max_abs);
int ftkx_sweep_cull(ftkx_ctx *ctx, int t_masked, size_t *n_cells);
int ftkx_get_sparse_cells(ftkx_ctx *ctx, unsigned long long *dst, int dst_on_device);
size_t ftkx_patch_doubles(const ftkx_ctx *ctx);      /* doubles per cell in a patch buffer */
int ftkx_gather_patches(ftkx_ctx *ctx, int t, const unsigned long long *cells, size_t n, double *patches, int on_device);
int ftkx_scatter_patches(ftkx_ctx *ctx, int t, const unsigned long long *cells, size_t n, const double *patches, int on_device);

/* ---- the sweep ------------------------------------------------------------------------------------------------- */
/* Sweeps the simplices of `scope` whose corner lies in `core` at time t (interval: [t, t+1], needs slice t+1).
 * `factor` = vector_field_scaling_factor: any non-zero value is honoured (quantisation is trunc(v * factor) like 2d:605-616), but the
 * sign cull that makes the sweep memory-bound needs a power of two <= 2^53 (what update_vector_field_scaling_factor produces);
 * other factors run the full integer test on every simplex.  Records are returned in a context-owned pinned host buffer, sorted by tag,
 * valid until the next call on this context.  Synchronous on the context's stream.
 * Everything in a record is computed on the device; the one exception is the TYPE of a 3D record whose Hessian has an eigenvalue
 * that is zero up to rounding (plateaus, lattice-aligned data): the kernels flag it and hand its Jacobian over, and the class is
 * computed here, with the host's pow / acos / cos -- the libm the reference's eigen_solver3.hh:20-47 runs on, whose last-place
 * rounding is what such a class hangs on (ftkx_stats.reclassified counts them). */
int ftkx_sweep(ftkx_ctx *ctx, int t, int scope, unsigned long long factor, const ftkx_cp_t **out, size_t *n_out);

/* batched form: enqueue only records the request; ftkx_sweep_collect() launches the whole batch (one mask / cull / exact
 * launch covers every enqueued timestep), synchronises and returns the records of ALL of them (sorted by tag). */
int ftkx_sweep_enqueue(ftkx_ctx *ctx, int t, int scope, unsigned long long factor);
int ftkx_sweep_enqueue_many(ftkx_ctx *ctx, const int *timesteps, const int *scopes, const unsigned long long *factors, int n);   /* all or nothing */
int ftkx_sweep_collect(ftkx_ctx *ctx, const ftkx_cp_t **out, size_t *n_out);
int ftkx_sweep_cancel(ftkx_ctx *ctx);    /* forgets the enqueued, not yet collected sweeps */

/* The device-driven pass over a series of resident slices -- what the tracker and bench.py use; the calls above remain for callers
 * that form the factors themselves (several ranks: ftk_amd/tslab.py).  Sweeps the steps (timesteps[i], scopes[i]), timesteps strictly
 * ascending, each under the reference's sticky factor (update_vector_field_scaling_factor, critical_point_tracker.hh:850-864):
 *   factor(i) = scaling_factor(min(*running_resolution, resolution of every slice the steps read with timestep <= timesteps[i] + 1))
 * with the running minimum and nbits formed ON THE DEVICE between the mask kernel and the exact test: the whole pass -- masks and
 * reduction, cull, factors, exact test, records, their ordering by tag and their transfer into the pinned host buffer -- is queued at
 * once and the host waits once.  *running_resolution: in = the minimum before these slices (DBL_MAX: none), out = the minimum after
 * them (as exact as ftkx_slices_prepare's res_below makes it: exact once it is below 2^-8).  factors (nullable): n values out.
 * Whatever the device-driven form does not cover (options that need the tile path, a type filter, wrapped reference tags, a factor that
 * hangs on the last bit of the host's log2, slices with vertices that can overflow a determinant, ...) is swept by the calls above
 * inside this one, with the same result.  Records as for ftkx_sweep_collect: sorted by tag, valid until the next call. */
int ftkx_sweep_series(ftkx_ctx *ctx, const int *timesteps, const int *scopes, int n, double *running_resolution,
                      unsigned long long *factors, const ftkx_cp_t **out, size_t *n_out);
/* The same pass in two halves, for callers that keep the device busy across passes (a streaming tracker, bench.py): _submit queues a
 * pass and returns; _complete waits for the OLDEST open pass and returns what ftkx_sweep_series returns.  At most three passes are open at
 * a time; two keep the device busy -- submit(N), submit(N + 1), complete(N), submit(N + 2), complete(N + 1), ... --, the third lets the
 * host run one pass ahead of a split pass's tail (ftkx_series_last_path = 5), which ends behind the mask kernel of the pass after it
 * (slab passes, ftkx_series_dist_*: two).  While pass N + 1's mask kernel runs, the host
 * collects pass N, and the records of a pass with many of them cross PCIe on a copy engine instead of holding the stream.
 * running_resolution of _submit: the running minimum before this pass, or NULL = continue from the pass queued before it, still open
 * (the minimum is handed on ON THE DEVICE; _complete then reports the chained value).  Between _submit and _complete only slices may be
 * pushed or dropped and further passes submitted; the sweeps above and ftkx_slices_prepare fail until every open pass is complete.
 * Records of a pass: valid until the next _submit OR the next _complete (or any other sweep call) after its own _complete -- a pass that
 * the host-driven batch had to sweep returns the context's shared host buffer, which the next such completion overwrites; a device-driven
 * pass's buffers are taken over by the second _submit after it.  Copy what must live longer.  Results are those of ftkx_sweep_series on
 * the same steps: tests/test_gpu_series.py::test_pipelined_passes_equal_the_plain_ones.
 * _abort: after a failed _submit / _complete (or to give up): waits for whatever the open passes queued, discards them and leaves the
 * context as if no pass had been submitted (masks they were building are rebuilt by the next sweep); FTKX_OK with nothing open. */
int ftkx_sweep_series_submit(ftkx_ctx *ctx, const int *timesteps, const int *scopes, int n, const double *running_resolution);
int ftkx_sweep_series_complete(ftkx_ctx *ctx, double *running_resolution, unsigned long long *factors, const ftkx_cp_t **out, size_t *n_out);
int ftkx_sweep_series_abort(ftkx_ctx *ctx);
/* ---- the slab pass: the device-driven pass of ONE RANK of a series cut into timestep slabs (one process per GPU) ----------
 * The reference's sticky factor (critical_point_tracker.hh:850-864) runs over ALL slices in time order and rank r's last interval sweep
 * reads the first slice of rank r + 1.  Both links are small and both are closed ON THE DEVICE: the pass is queued in four stages, and
 * between them the caller queues its collectives on the context's stream (RCCL: ncclAllGather / ncclSend / ncclRecv, or
 * torch.distributed with that stream current) -- nothing is waited for on the host until ftkx_sweep_series_complete:
 *   _begin   masks + reduction of this rank's slices; `contrib` (4 doubles: min resolution and max |v| of the slab, and of its FIRST slice)
 *            and, with masks_out, the first slice's sign masks as one message of ftkx_packed_masks_bytes() bytes -- built and packed FIRST:
 *            with a side_stream, that stream is made to wait for the message only, so a send queued on it crosses xGMI while the masks
 *            of the slab's other slices are still being built (the caller makes the context's stream wait for the side stream's receive
 *            before _cull)
 *      caller: all_gather(contrib -> gathered, 4 doubles per rank); masks_out -> lower neighbour, upper neighbour's -> masks_in
 *   _cull    the halo's masks imported, the running minimum before this slab from `gathered`, cull + factors, and `request_out`:
 *            1 + ftkx_series_dist_cells() words -- the count of surviving cells whose exact test reads the halo slice and their indices
 *            (-1: send the slice itself)
 *      caller: request_out -> upper neighbour, lower neighbour's -> request_in
 *   _serve   the patches around the lower neighbour's cells, gathered from this rank's first slice into `reply_out`:
 *            ftkx_series_dist_cells() * ftkx_patch_doubles() doubles, a FIXED size (the count is read on the device)
 *      caller: reply_out -> lower neighbour, upper neighbour's -> reply_in
 *   _finish  patches scattered into the halo slice; exact test, records, finish.  The pass is open: ftkx_sweep_series_complete as usual
 * All buffers are device memory of this context's device and must stay untouched until the pass is complete; ranks without a lower /
 * upper neighbour pass NULL for the corresponding buffers (upper = -1: the last step's slices are all this rank's own; with more ranks than
 * timesteps the neighbours are the nearest ranks that own timesteps, not rank - 1 / rank + 1).  Two slab passes
 * may be in flight like any two passes (separate buffers each).  ftkx_sweep_series_complete returns FTKX_E_NOSLICE when the request said
 * -1 (nothing was swept: fetch the slice itself, push it, and sweep with ftkx_sweep_series from the running minimum
 * ftkx_series_dist_status reports); the owner learns the same from `served` = -1.  Options the device-driven pass does not cover:
 * FTKX_E_UNSUPPORTED from _begin (same on every rank: they share options and mesh) -- use ftkx_slices_prepare / ftkx_sweep_enqueue /
 * ftkx_sweep_cull / ftkx_sweep_collect. */
size_t ftkx_series_dist_cells(const ftkx_ctx *ctx);
int ftkx_series_dist_begin(ftkx_ctx *ctx, const int *timesteps, const int *scopes, int n, const double *running_resolution, int rank, int nranks,
                           int upper /* the rank that owns the slice behind this slab, -1: none (every slice the steps read is this rank's own) */,
                           void *contrib, const void *gathered, void *masks_out, void *side_stream /* hipStream_t, nullable */);
int ftkx_series_dist_cull(ftkx_ctx *ctx, const void *masks_in, void *request_out);
int ftkx_series_dist_serve(ftkx_ctx *ctx, const void *request_in, void *reply_out);
int ftkx_series_dist_finish(ftkx_ctx *ctx, const void *reply_in);
/* of the slab pass completed last: what it asked its upper neighbour for and what its lower neighbour asked of it (cells, -1: the whole
 * slice, 0: nothing), and the gathered contributions (4 * nranks doubles, nullable) */
int ftkx_series_dist_status(const ftkx_ctx *ctx, long long *asked, long long *served, double *gathered, int nranks);

/* which way the last ftkx_sweep_series went: 1 = device-driven (the kernel chain), 2 = device-driven and finished by the fused tail
 * kernel (sparse data), 4 = the whole pass in one launch (small series), 5 = split: the kernel chain on a stream of its own next to the
 * mask kernel of the pass queued behind (pipelined passes over sparse data whose mask kernel is long enough to hide it), 0 = host-driven
 * batch; *status (nullable) = the SERIES_* bits the kernels raised (csrc/sweep_params.hpp) */
int ftkx_series_last_path(const ftkx_ctx *ctx, unsigned long long *status);

/* The split pass (path 5) is taken where a pass's mask launch is long enough to hide its tail (sparse data: 1 GB and more; hit-dense data:
 * 4 GB and more) AND -- in the default setting, "auto" -- the context's own measurement found it no slower: the first qualifying passes of
 * a shape run five in order, five split, the host's time between completions is sampled, and the split pass is kept unless it was more
 * than 2 % slower (how two queues share the device is the hardware's business; GPU_MAX_HW_QUEUES alone can turn it).  Deterministic
 * settings: FTKX_SERIES_HOOKS="split=4" (the size rule alone) / "split=0" (never).  This call says which way the context went:
 * *state 0 auto, still measuring; 1 auto, split; 2 auto, in order; 3 forced on; 4 forced off; medians in ms per pass (0: none taken). */
int ftkx_series_split_decision(const ftkx_ctx *ctx, int *state, double *median_in_order_ms, double *median_split_ms);

/* counters of the last collect: simplices visited (work items), cells/simplices surviving the cull, device-side hits */
typedef struct ftkx_stats {
  unsigned long long work_items, cells, cells_survived, simplices_tested, hits;
  int cull_enabled;
  unsigned long long reclassified;   /* 3D records whose Hessian has an eigenvalue that is zero up to rounding: their class was computed
                                        on the host, with the host's libm (the device library's pow / acos / cos differ from it in the
                                        last place, which is all such a class hangs on) */
} ftkx_stats;
int ftkx_get_stats(const ftkx_ctx *ctx, ftkx_stats *st);

/* The vertex sign masks a sweep builds for a slice are kept with the slice (slice t+1 of step t is slice t of step t+1).
 * A caller that rewrites a slice's device memory in place, or a benchmark that must redo all work, drops them with this. */
int ftkx_invalidate_masks(ftkx_ctx *ctx);

/* optional kernel timing with HIP events on the context's stream (for bench.py's roofline figure).  Index: 0 mask_kernel,
 * 1 cull_kernel, 2 exact_kernel (+ everything behind it: the FP64 half of the records, their ordering), 3 tile_kernel (the integer test of
 * every simplex: exact_only, the overflow regime); ms[] = summed device time, launches[] = launches timed since set_profiling(1). */
int ftkx_set_profiling(ftkx_ctx *ctx, int on);      /* 0 off, 1 every kernel family, 2 the mask kernel only (an event pair costs the stream a few us) */
int ftkx_get_kernel_times(const ftkx_ctx *ctx, double ms[4], unsigned long long launches[4]);

/* ---- stateless one-shot calls with the reference boundary's argument list ------------------------------------ */
/* extract_cp2dt_{cuda,sycl}(scope, current_timestep, domain, core, ext, Vc, Vn, Jc, Jn, Sc, Sn, use_explicit_coords, coords)
 * critical_point_tracker_2d_regular.hh:33-63.  Lattices are passed as (start, size) triples; host pointers; the records
 * are malloc'd into *out (release with ftkx_free).  tag = work index inside `core` unless opt->tag_mode says otherwise. */
int ftkx_extract_cp2dt(int scope, int current_timestep,
                       const long long domain_st[3], const long long domain_sz[3],
                       const long long core_st[3], const long long core_sz[3],
                       const long long ext_st[2], const long long ext_sz[2],
                       const double *Vc, const double *Vn, const double *Jc, const double *Jn,
                       const double *Sc, const double *Sn, int use_explicit_coords, const double *coords,
                       unsigned long long factor, const ftkx_options *opt, int device_id,
                       ftkx_cp_t **out, size_t *n_out);
/* extract_cp3dt_cuda(scope, current_timestep, domain4, core4, ext3, Vc, Vl, Jc, Jl, Sc, Sl)  critical_point_tracker_3d_regular.hh:42-56 */
int ftkx_extract_cp3dt(int scope, int current_timestep,
                       const long long domain_st[4], const long long domain_sz[4],
                       const long long core_st[4], const long long core_sz[4],
                       const long long ext_st[3], const long long ext_sz[3],
                       const double *Vc, const double *Vn, const double *Jc, const double *Jn,
                       const double *Sc, const double *Sn,
                       unsigned long long factor, const ftkx_options *opt, int device_id,
                       ftkx_cp_t **out, size_t *n_out);
void ftkx_free(void *p);

/* ---- pass 2 on the hit set: records -> traced curves (host side, like the reference's) --------------------------------
 * Replaces critical_point_tracker::trace_critical_points_offline (include/ftk/filters/critical_point_tracker.hh:668-817) with
 * the neighbourhood of critical_point_tracker_2d_regular.hh:189-197 and geometry/cc2curves.hh:10-122: two hits are neighbours
 * iff their simplices share a (d+1)-cell; hits with more than two neighbours are dropped; every connected component of the rest
 * is one curve whose points come in the reference's order.  `recs[i].tag` must be element tags (FTKX_TAG_EXACT64, or
 * FTKX_TAG_REFERENCE where that did not overflow) of the mesh whose spatial box is (domain_st, domain_sz). */
typedef struct ftkx_curves {
  size_t n_curves, n_points, n_special;   /* n_special: hits dropped for having more than two neighbours */
  long long *offsets;                     /* n_curves + 1 */
  long long *indices;                     /* n_points indices into recs[], curve after curve */
  int *loop;                              /* n_curves: first and last point are neighbours */
} ftkx_curves;
int  ftkx_trace_curves(int nd, const long long domain_st[3], const long long domain_sz[3], const ftkx_cp_t *recs, size_t n, ftkx_curves *out);
void ftkx_free_curves(ftkx_curves *c);
/* The same, with the two phases that are independent per record -- the neighbour search and the component labelling -- on the context's
 * GPU (tags up, neighbours and component roots down; what is serial per curve stays on host threads): identical curves.  Record sets
 * below 4 096 records, or whose tags do not come strictly ascending, take ftkx_trace_curves as it is; ctx == NULL likewise. */
int  ftkx_trace_curves_ctx(ftkx_ctx *ctx, int nd, const long long domain_st[3], const long long domain_sz[3], const ftkx_cp_t *recs, size_t n, ftkx_curves *out);
/* the same on the tags alone (8 bytes per record): a caller that holds its points in another form need not build records around them */
int  ftkx_trace_curves_tags_ctx(ftkx_ctx *ctx, int nd, const long long domain_st[3], const long long domain_sz[3], const unsigned long long *tags, size_t n, ftkx_curves *out);

/* enable_streaming_trajectories (critical_point_tracker.hh:38; update_timestep 2d:326-330, 3d:197-201): trajectories that grow while
 * the sweep streams -- trace_critical_points_online (critical_point_tracker.hh:523-639).  After every interval sweep the caller hands
 * over the discrete points found since the last call (they are consumed); open trajectories are continued greedily through the
 * new points, the rest starts new trajectories.  ftkx_online_tracer_curves copies the trajectories out: *points (malloc'ed, release
 * with ftkx_free) holds them one after the other, `out` the offsets / loop flags (indices = 0 .. n_points-1). */
typedef struct ftkx_online_tracer ftkx_online_tracer;
int  ftkx_online_tracer_create(ftkx_online_tracer **out, int nd, const long long domain_st[3], const long long domain_sz[3]);
void ftkx_online_tracer_destroy(ftkx_online_tracer *);
int  ftkx_online_tracer_grow(ftkx_online_tracer *, const ftkx_cp_t *recs, size_t n);
int  ftkx_online_tracer_curves(const ftkx_online_tracer *, ftkx_cp_t **points, ftkx_curves *out);

/* Trajectory post-processing with the defaults of json_interface::post_process (include/ftk/filters/json_interface.hh:758-800):
 * smooth_ordinal_types / smooth_interval_types / rotate, split_all, reorder / adjust_time (features/feature_curve.hh:220-348,
 * features/feature_curve_set.hh:514-532).  recs[] must carry the aux word (ordinal, timestep) the sweep writes.  Per point the
 * smoothed type and adjusted time come back next to the index into recs[]. */
typedef struct ftkx_trajectories {
  size_t n_curves, n_points;
  long long *offsets;
  long long *indices;
  int *loop;
  unsigned int *type;
  double *t;
  int *id;                                /* n_curves: label of the curve in the reference's multimap = index of the traced curve
                                             it came from (split pieces share their parent's label, feature_curve_set.hh:530-531) */
} ftkx_trajectories;
int  ftkx_post_process_curves(const ftkx_cp_t *recs, size_t n, const ftkx_curves *in, ftkx_trajectories *out);
void ftkx_free_trajectories(ftkx_trajectories *c);

/* ---- record-stream formats: what `ftk --output-type discrete|traced` writes and reads -------- */
/* binary = diy::serializeToFile, byte-identical to the reference's files; json = nlohmann's compact dump (parses to the
 * same values); text = the ostream print-outs.  Host only.  Errors: ftkx_last_error(NULL, ...). */
enum { FTKX_FORMAT_BINARY = 0, FTKX_FORMAT_JSON = 1, FTKX_FORMAT_TEXT = 2 };
int ftkx_format_from_path(const char *path);     /* "...txt" text, "...json" json, else binary (filters/json_interface.hh:225-231) */
/* critical_point_tracker::write_critical_points_{json,binary,text} (filters/critical_point_tracker.hh:106-108, 339-343,
 * 392-396; point members features/feature_point.hh:143-205).  recs[] carry ordinal/timestep in their aux word.  v (3 per
 * point) and id are nullable: the sweep does not produce them and the reference writes zeros for discrete points.
 * scalar_names: labels of the text format; n_scalar_names < 0 = the tracker's default {"scalar"}. */
int ftkx_write_critical_points(const char *path, int format, const ftkx_cp_t *recs, size_t n, const double *v, const unsigned long long *id,
                               const char *const *scalar_names, int n_scalar_names);
/* read_critical_points_{json,binary} (critical_point_tracker.hh:345-352, 498-508).  *recs (and *v, *id when asked for) are
 * malloc'ed: release with ftkx_free.  There is no text reader (the reference has none): FTKX_E_UNSUPPORTED. */
int ftkx_read_critical_points(const char *path, int format, ftkx_cp_t **recs, size_t *n, double **v, unsigned long long **id);
/* write_traced_critical_points_{json,binary,text} (critical_point_tracker.hh:354-373, 475-484; features/feature_curve.hh:436-510,
 * features/feature_curve_set.hh:75-118, 180-228) of trajectories as ftkx_post_process_curves returns them (or of raw traced
 * curves: fill type/t with NULL to take them from recs).  Per-curve statistics = feature_curve_t::update_statistics. */
int ftkx_write_traced_critical_points(const char *path, int format, const ftkx_cp_t *recs, size_t n, const ftkx_trajectories *trajs,
                                      const char *const *scalar_names, int n_scalar_names);
/* read_traced_critical_points_{json,binary}: points come back as records in file order (indices = 0..n-1), curve labels as
 * the reference assigns them on load.  Release with ftkx_free(*recs) and ftkx_free_trajectories. */
int ftkx_read_traced_critical_points(const char *path, int format, ftkx_cp_t **recs, size_t *n, ftkx_trajectories *trajs);

/* ---- derived fields on the device (ndarray/grad.hh), exposed for callers that keep V/J themselves ------------- */
/* all pointers are DEVICE pointers; results are bit-identical to the reference's host loops */
int ftkx_gradient2D(ftkx_ctx *ctx, const double *S, int DW, int DH, double *V);                    /* grad.hh:10-31   */
int ftkx_jacobian2D(ftkx_ctx *ctx, const double *V, int DW, int DH, int symmetric, double *J);     /* grad.hh:54-86   */
int ftkx_gradient3D(ftkx_ctx *ctx, const double *S, int DW, int DH, int DD, double *V);            /* grad.hh:130-149 */
int ftkx_jacobian3D(ftkx_ctx *ctx, const double *V, int DW, int DH, int DD, double *J);            /* grad.hh:175-212 */

/* profiling aid: streams `bytes` of device memory with the mask kernel's load shape (16 B per lane) and nothing else, so that
 * rocprofv3's FETCH_SIZE can be calibrated on a known byte count (tools/calibrate_fetch.py) */
int ftkx_debug_stream_read(ftkx_ctx *ctx, const void *device_ptr, size_t bytes);

/* profiling aid (bench.py's int-VALU yardstick): from the next sweep on the tile kernel -- exact_only, the overflow regime -- repeats its fan
 * phase, the predicate arithmetic on a tile staged in LDS, `repeat` times per tile and step; records are those of one.  The time the extra
 * repetitions add is the time of the arithmetic alone.  1 = off. */
int ftkx_debug_tile_repeat(ftkx_ctx *ctx, int repeat);

/* profiling aid: the mask-kernel instantiation of the most recent sweep in this process, spelled as rocprofv3 lists it */
const char *ftkx_last_mask_kernel(void);

/* profiling aid: launches per mask-kernel instantiation family in this process (seven of them; returns that number): which shapes the
 * suite / a workload reached -- every kernel behind launch_masks is reachable by some mesh, none is there for history's sake */
int ftkx_debug_mask_kernel_launches(unsigned long long *launches, const char **names, int n);

/* profiling aid: how the host arrays this context was handed (on_device = 0) reached HBM -- staged by the library's copy threads through
 * pinned pieces (pageable sources of 32 MiB and more; FTKX_UPLOAD_THREADS, default 4, 0 = never) or by the runtime's own copy */
int ftkx_debug_upload_counts(const ftkx_ctx *ctx, unsigned long long *staged, unsigned long long *direct);

/* profiling aid (tools/mask_overlap.py): slice t's mask job -- its masks must exist, ftkx_slices_prepare -- launched `reps` times back to
 * back on the context's stream (nstreams 1) or alternately on it and a second stream (2), with a small dependent kernel in front of each
 * launch when with_begin; device time per launch.  What a streaming tracker's one-slice launch pays for its ramp. */
int ftkx_debug_mask_relaunch(ftkx_ctx *ctx, int t, int reps, int nstreams, int with_begin, double *ms_per_launch);

/* library / device identification */
const char *ftkx_version(void);
int ftkx_device_count(void);
int ftkx_pointer_device(const void *p);          /* ordinal of the device the pointer lives on, -1: host memory or unknown */
int ftkx_context_device(const ftkx_ctx *ctx);

#ifdef __cplusplus
}
#endif
#endif
