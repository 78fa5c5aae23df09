/* ftkx_slab -- the several-rank host of the sweep: one rank's part of a series cut into timestep slabs, driven from C/C++.
 *
 * The reference distributes a tracker over MPI ranks inside the tracker itself (include/ftk/filters/regular_tracker.hh:127-149: the
 * partitioner; include/ftk/filters/critical_point_tracker.hh:689: the gather of the discrete points on the root) and keeps its device list
 * on the filter (include/ftk/filters/filter.hh:47-61).  Here the series is cut in TIME (north_star): rank r owns the
 * timesteps ftkx_slab_range() gives it, sweeps ordinal(t) and interval[t, t + 1] for them, and is linked to its neighbours by two small
 * things only -- the sticky running minimum of update_vector_field_scaling_factor (critical_point_tracker.hh:850-864) across slabs, and the
 * first slice of the next slab, which its last interval sweep reads.  include/ftkx.h closes both links on the device in four stages
 * (ftkx_series_dist_begin / _cull / _serve / _finish); THIS header is the host that drives the stages and queues the messages between them:
 *
 *     begin   (masks, reduction, contribution, outgoing masks) | all_gather of 4 doubles per rank; masks -> lower neighbour (side stream)
 *     cull    (masks imported, factors, cull, request)         | request -> upper neighbour
 *     serve   (patches around the neighbour's cells)           | reply -> lower neighbour
 *     finish  (patches scattered, exact test, records)         | ftkx_slab_complete: the ONE host wait of the pass
 *
 * Where the halo slice is needed as a whole (request -1: too many surviving cells, masks that did not fit) both sides learn it from the
 * same number when they complete, the owner sends its first slice, the asker sweeps again with it (ftkx_sweep_series) and the next pass
 * starts compact again.  Two passes may be in flight: submit, submit, complete, submit, complete, ...
 *
 * The messages travel over a TRANSPORT, a table of two calls.  Built in: RCCL (ftkx_slab_create_rccl: ncclAllGather and grouped
 * ncclSend / ncclRecv on the context's stream and a side stream -- nothing waits on the host), and a hub for ranks that live in ONE process
 * (ftkx_slab_hub_*: peer copies between the ranks' devices, each rank driven by its own thread -- several GPUs behind one tracker, and the
 * tests on one GPU).  A caller's own transport (MPI, torch.distributed from Python, ...) is a ftkx_slab_transport of its own.
 * Plain C like include/ftkx.h: opaque handles, raw pointers, status codes. */
#ifndef FTKX_SLAB_H
#define FTKX_SLAB_H

#include "ftkx.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct ftkx_slab ftkx_slab;

/* how the ranks' messages travel.  Buffers are memory of the slab's backend (device memory of the context's device for a real context);
 * `stream` is the hipStream_t the call's work must be ordered on (the context's stream, or the slab's side stream).  Return FTKX_OK or a
 * negative code.  Every rank issues the same sequence of calls per pass: [exchange masks] all_gather, exchange request, exchange reply
 * -- and, where a slice goes as a whole, one more exchange when the pass is completed. */
typedef struct ftkx_slab_transport {
  void *user;
  /* `bytes` from `send` of every rank into recv + r * bytes of every rank */
  int (*all_gather)(void *user, const void *send, void *recv, size_t bytes, void *stream);
  /* at most one message each way: send_bytes from `send` to rank `to` (to < 0: none), recv_bytes into `recv` from rank `from` (from < 0: none) */
  int (*exchange)(void *user, const void *send, size_t send_bytes, int to, void *recv, size_t recv_bytes, int from, void *stream);
  /* 1: the calls only QUEUE work on `stream` (RCCL, peer copies): the pass never waits on the host, and the first slice's masks travel on a
   * side stream next to the mask kernel of the slab's other slices; 0: they return when the data has arrived (host-staged transports) */
  int queued;
  void (*destroy)(void *user);     /* nullable: called by ftkx_slab_destroy */
} ftkx_slab_transport;

/* what the host drives: ftkx_slab_create fills this table with the calls of include/ftkx.h on a context.  A table of the caller's own
 * (ftkx_slab_create_custom) lets the same host logic run without a device -- tests/test_tslab.py runs it over gloo on the CPU with the
 * oracle standing in for the stages.  Stage calls: the arguments of ftkx_series_dist_* / ftkx_sweep_series_complete / ftkx_series_dist_status. */
typedef struct ftkx_slab_backend {
  void *user;
  int (*begin)(void *user, const int *ts, const int *scopes, int n, const double *running, int rank, int nranks, int upper, void *contrib, const void *gathered,
               void *masks_out, void *side_stream);
  int (*cull)(void *user, const void *masks_in, void *request_out);
  int (*serve)(void *user, const void *request_in, void *reply_out);
  int (*finish)(void *user, const void *reply_in);
  int (*complete)(void *user, double *running, unsigned long long *factors, const ftkx_cp_t **out, size_t *n_out);
  int (*status)(void *user, long long *asked, long long *served, double *gathered, int nranks, int *path, unsigned long long *path_status);
  /* the whole-slice recovery: the halo slice t_halo replaced by `full_slice` (backend memory), the steps swept again from *running, the slice dropped */
  int (*recover)(void *user, int t_halo, const void *full_slice, const int *ts, const int *scopes, int n, double *running, unsigned long long *factors,
                 const ftkx_cp_t **out, size_t *n_out);
  const void *(*first_slice)(void *user, int t);       /* this rank's slice t as the neighbour needs it (S, or V for vector input) */
  void *(*alloc)(void *user, size_t bytes);            /* zero-filled */
  void (*release)(void *user, void *p);
  int (*upload)(void *user, void *dst, const void *host_src, size_t bytes);      /* synchronous */
  int (*download)(void *user, void *host_dst, const void *src, size_t bytes);    /* synchronous, behind everything queued on the stream */
  void (*abort)(void *user);                           /* after a failed stage: ftkx_sweep_series_abort */
  size_t masks_bytes, cells, patch_doubles, slice_bytes;   /* ftkx_packed_masks_bytes, ftkx_series_dist_cells, ftkx_patch_doubles, bytes of one input slice */
  void *stream;                                        /* hipStream_t of the stage calls (NULL: none) */
  int device;                                          /* 1: the buffers are device memory and a side stream may be created */
} ftkx_slab_backend;

/* the t-slab partition: rank r of `nranks` owns [*t0, *t1) = [r nt / nranks, (r + 1) nt / nranks) of nt timesteps (contiguous, sizes
 * differing by at most one); the owner of timestep t (-1: none).  With more ranks than timesteps some ranks own nothing: they take part
 * in the all_gather only, and a rank's neighbours are the nearest ranks that DO own timesteps. */
void ftkx_slab_range(int nt, int nranks, int rank, int *t0, int *t1);
int ftkx_slab_owner(int t, int nt, int nranks);

/* One rank's slab host over a context whose own slices [t0, t1) are (or will be, before the first submit) resident.  The transport table
 * is copied; its `user` must outlive the slab (destroy is called by ftkx_slab_destroy). */
int ftkx_slab_create(ftkx_ctx *ctx, int nt, int rank, int nranks, const ftkx_slab_transport *tr, ftkx_slab **out);
int ftkx_slab_create_custom(const ftkx_slab_backend *backend, int nt, int rank, int nranks, const ftkx_slab_transport *tr, ftkx_slab **out);
/* RCCL: `comm` is the caller's ncclComm_t over the ranks (rank / nranks must be its); messages are ncclAllGather and grouped ncclSend /
 * ncclRecv on the context's stream (the masks: on the slab's side stream).  side_comm (nullable): a second communicator for the side
 * stream's traffic -- with one communicator RCCL runs the masks' message and the all_gather one after the other, in the order issued.
 * Two communicators on one device run CONCURRENTLY: that needs an RCCL that makes progress on both (its kernels co-resident; every rank
 * issues on both in the same order, which the protocol does) -- a rank without neighbours takes no part in the side exchange. */
int ftkx_slab_create_rccl(ftkx_ctx *ctx, int nt, int rank, int nranks, void *comm, void *side_comm, ftkx_slab **out);
void ftkx_slab_destroy(ftkx_slab *s);
/* A series that is PERIODIC in time: slice nt is slice 0 again.  The last timestep's sweep is then an interval sweep as well, [nt - 1, nt], and
 * the rank that owns it has the owner of timestep 0 for its upper neighbour -- with one rank, itself: it sends the masks of its first slice,
 * its request and its patches to itself, over the same transport calls (RCCL: grouped ncclSend / ncclRecv with its own rank as the peer).
 * The records of that last sweep carry t in [nt - 1, nt].  Call before the first pass.  (Also how a ONE-GPU box executes every message of
 * the slab protocol over RCCL, the whole-slice recovery included: tests/test_gpu_slab_host.py.) */
int ftkx_slab_set_periodic(ftkx_slab *s, int on);

/* queues one pass over this rank's slab.  running_resolution: the running minimum before the SERIES (NULL: none yet = DBL_MAX); the
 * minimum before this slab comes from the lower ranks' contributions, on the device. */
int ftkx_slab_submit(ftkx_slab *s, const double *running_resolution);
/* waits for the oldest pass in flight.  Records (sorted by tag), factors (one per own timestep; nullable) and *running_resolution (the
 * minimum after this slab) as ftkx_sweep_series returns them; the records stay valid until the next call on this slab. */
int ftkx_slab_complete(ftkx_slab *s, double *running_resolution, unsigned long long *factors, const ftkx_cp_t **out, size_t *n_out);
/* every rank's records on rank `root`, sorted by tag (critical_point_tracker.hh:689: the gather in front of pass 2); *merged (root only,
 * else NULL) is released with ftkx_free.  Collective: every rank calls it. */
int ftkx_slab_gather_records(ftkx_slab *s, const ftkx_cp_t *mine, size_t n, int root, ftkx_cp_t **merged, size_t *n_merged);

typedef struct ftkx_slab_info {
  int rank, nranks, t0, t1, lower, upper;               /* neighbours: the nearest ranks that own timesteps; -1: none */
  unsigned long long bytes_sent, bytes_received;        /* through the transport, since creation */
  int fallbacks;                                        /* passes in which this rank needed its halo slice as a whole */
  long long last_asked, last_served;                    /* of the pass completed last */
  int last_path; unsigned long long last_status;        /* ftkx_series_last_path of the pass completed last */
  int open;                                             /* passes in flight */
} ftkx_slab_info;
int ftkx_slab_get_info(const ftkx_slab *s, ftkx_slab_info *info);
const char *ftkx_slab_last_error(const ftkx_slab *s);

/* ---- RCCL helpers for callers without a communicator of their own (Python: the 128-byte id travels over any channel) ---- */
int ftkx_rccl_unique_id(void *id128);
int ftkx_rccl_comm_create(const void *id128, int rank, int nranks, int device, void **comm);      /* ncclCommInitRank on `device`: collective over the ranks */
void ftkx_rccl_comm_destroy(void *comm);
int ftkx_rccl_version(void);                                                          /* ncclGetVersion */
int ftkx_slab_transport_rccl(void *comm, void *side_comm, ftkx_slab_transport *out);  /* the table ftkx_slab_create_rccl uses */

/* synchronous copies between host memory and memory of the context's device, behind everything queued on the context's stream (what a
 * host-staged transport needs: ftk_amd/tslab.py over gloo) */
int ftkx_upload(ftkx_ctx *ctx, void *dst, const void *host_src, size_t bytes);
int ftkx_download(ftkx_ctx *ctx, void *host_dst, const void *src, size_t bytes);

/* ---- ranks of ONE process (several GPUs behind one tracker; tests): a hub the ranks' transports rendezvous on.  Every rank must be
 * driven by a thread of its own -- a call blocks until the peer has posted its side.  Messages are hipMemcpyPeerAsync between the ranks'
 * devices, ordered by events: queued, like RCCL. ---- */
typedef struct ftkx_slab_hub ftkx_slab_hub;
ftkx_slab_hub *ftkx_slab_hub_create(int nranks);
void ftkx_slab_hub_destroy(ftkx_slab_hub *hub);
int ftkx_slab_create_local(ftkx_ctx *ctx, int nt, int rank, ftkx_slab_hub *hub, ftkx_slab **out);
/* a rank that gives up tells the hub, so that peers blocked in a rendezvous return FTKX_E_DEVICE instead of waiting for ever */
void ftkx_slab_hub_abort(ftkx_slab_hub *hub);

#ifdef __cplusplus
}
#endif
#endif
