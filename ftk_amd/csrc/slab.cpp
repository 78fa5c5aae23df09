// ftkx_slab: the several-rank host of the sweep (include/ftkx_slab.h).  One rank's device-driven pass over its timestep slab, the ranks'
// messages queued between its stages over a transport -- RCCL (slab_rccl.cpp), the in-process hub below, or the caller's own table.
//
// Reference counterparts: the reference distributes its tracker over MPI ranks inside the tracker (include/ftk/filters/regular_tracker.hh:127-149)
// and gathers the discrete points on the root in front of pass 2 (include/ftk/filters/critical_point_tracker.hh:689); its scaling factor is a
// sticky running minimum in time (critical_point_tracker.hh:850-864).  Here the cut is in TIME, and both links between neighbouring slabs --
// the running minimum and the first slice of the next slab -- are closed on the device by the stage calls of include/ftkx.h; this file only
// decides which message goes where between them, and recovers where a halo slice is needed as a whole.
//
// This is the ONE implementation of the protocol: ftk_amd/tslab.py (SlabSeries), the C++ tracker's slab mode (tracker.cpp) and bench.py call it.
#include "ctx.hpp"
#include "../../include/ftkx_slab.h"

#include <condition_variable>
#include <deque>
#include <mutex>

using namespace ftkxh;

namespace {

constexpr double kDblMax = DBL_MAX;

struct Outcome {
  std::vector<ftkx_cp_t> copy;            // the records of a pass collected early (the context had to be free for the second sweep of the pass before it)
  const ftkx_cp_t *recs = nullptr; size_t n = 0;
  std::vector<unsigned long long> f;
  double run = kDblMax;
  bool failed = false;
  long long asked = 0, served = 0;
  std::vector<double> gathered;
  int path = 0; unsigned long long path_status = 0;
};

struct Set {
  double *contrib = nullptr, *gathered = nullptr;
  void *masks_out = nullptr, *masks_in = nullptr;
  unsigned long long *req_out = nullptr, *req_in = nullptr;
  double *reply_out = nullptr, *reply_in = nullptr;
  double running_in = kDblMax;
};

}  // namespace

struct ftkx_slab {
  ftkx_slab_backend be;
  ftkx_slab_transport tr;
  int nt = 0, rank = 0, nranks = 1, t0 = 0, t1 = 0, lower = -1, upper = -1, t_halo = -1;
  std::vector<int> ts, scopes;
  bool sized = false;
  size_t masks_bytes = 0, cells = 0, pd = 0, slice_bytes = 0;
  hipStream_t side = nullptr;
  hipEvent_t ev_side = nullptr;
  Set sets[2];
  bool sets_ready = false;
  int k = 0;
  std::deque<int> open;                   // buffer sets of the passes in flight, oldest first
  std::deque<Outcome> stash;              // outcomes of passes completed early
  void *full_halo = nullptr;
  unsigned long long bytes_sent = 0, bytes_received = 0;
  int fallbacks = 0;
  long long last_asked = 0, last_served = 0;
  int last_path = 0; unsigned long long last_status = 0;
  std::vector<ftkx_cp_t> held;            // records handed out by the last complete, where they had to be copied
  std::vector<unsigned long long> held_f;
  bool own_ctx_backend = false;
  // a mesh without summarised masks (rows that are not a multiple of 8 vertices: the generic mask kernel) has no compact halo: such a slab
  // is swept the plain way -- exact reductions all_gathered, the neighbour's first slice as a whole, ftkx_sweep_series -- inside submit
  bool plain = false;
  bool periodic = false;                  // the series is periodic in time: slice nt is slice 0 again (ftkx_slab_set_periodic)
  std::deque<Outcome> plain_out;
  std::string err;
};

namespace {

int sfail(ftkx_slab *s, int code, const char *fmt, ...)
{
  char buf[640];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof(buf), fmt, ap);
  va_end(ap);
  if (s) s->err = buf;
  return fail(nullptr, code, "%s", buf);
}

// a stage or transport call failed: its own message (the context's, where there is one) becomes the slab's
int carry(ftkx_slab *s, int rc, const char *where)
{
  char buf[512] = "";
  ftkx_last_error(nullptr, buf, sizeof(buf));              // (the calling thread's last error: every fail() of the library leaves it there)
  s->err = std::string(where) + ": " + buf;
  return rc;
}

// ---- the real backend: the calls of include/ftkx.h on a context -----------------------------------------------------------------------
int rb_begin(void *u, const int *ts, const int *scopes, int n, const double *running, int rank, int nranks, int upper, void *contrib, const void *gathered, void *masks_out, void *side)
{ return ftkx_series_dist_begin((ftkx_ctx *)u, ts, scopes, n, running, rank, nranks, upper, contrib, gathered, masks_out, side); }
int rb_cull(void *u, const void *masks_in, void *request_out) { return ftkx_series_dist_cull((ftkx_ctx *)u, masks_in, request_out); }
int rb_serve(void *u, const void *request_in, void *reply_out) { return ftkx_series_dist_serve((ftkx_ctx *)u, request_in, reply_out); }
int rb_finish(void *u, const void *reply_in) { return ftkx_series_dist_finish((ftkx_ctx *)u, reply_in); }
int rb_complete(void *u, double *running, unsigned long long *factors, const ftkx_cp_t **out, size_t *n_out) { return ftkx_sweep_series_complete((ftkx_ctx *)u, running, factors, out, n_out); }
int rb_status(void *u, long long *asked, long long *served, double *gathered, int nranks, int *path, unsigned long long *path_status)
{
  ftkx_ctx *c = (ftkx_ctx *)u;
  if (path) *path = ftkx_series_last_path(c, path_status);
  return ftkx_series_dist_status(c, asked, served, gathered, nranks);
}
int rb_recover(void *u, int t_halo, const void *full, const int *ts, const int *scopes, int n, double *running, unsigned long long *factors, const ftkx_cp_t **out, size_t *n_out)
{
  ftkx_ctx *c = (ftkx_ctx *)u;
  int rc = ftkx_drop_slice(c, t_halo);                      // the masks-only slice (gone already if the pass before this one needed the slice too)
  if (rc != FTKX_OK && rc != FTKX_E_NOSLICE) return rc;
  rc = c->scalar_mode == 1 ? ftkx_push_scalar_slice(c, t_halo, (const double *)full, 1) : ftkx_push_slice(c, t_halo, (const double *)full, nullptr, nullptr, 1);
  if (rc) return rc;
  rc = ftkx_sweep_series(c, ts, scopes, n, running, factors, out, n_out);
  const int rc2 = ftkx_drop_slice(c, t_halo);                // (the next pass starts compact again)
  return rc ? rc : rc2;
}
const void *rb_first_slice(void *u, int t)
{
  ftkx_ctx *c = (ftkx_ctx *)u;
  auto it = c->slices.find(t);
  if (it == c->slices.end() || it->second.sparse) return nullptr;
  return c->scalar_mode == 1 ? (const void *)it->second.S : (const void *)it->second.V;
}
void *rb_alloc(void *u, size_t bytes)
{
  ftkx_ctx *c = (ftkx_ctx *)u;
  void *p = nullptr;
  if (hipSetDevice(c->device) != hipSuccess || hipMalloc(&p, std::max<size_t>(bytes, 8)) != hipSuccess) return nullptr;
  // (on the context's stream and waited for: a memset on the null stream is not ordered with a non-blocking stream's kernels, and could
  // land in a buffer a stage has written already)
  if (hipMemsetAsync(p, 0, std::max<size_t>(bytes, 8), c->stream) != hipSuccess || hipStreamSynchronize(c->stream) != hipSuccess) { (void)hipFree(p); return nullptr; }
  return p;
}
void rb_release(void *u, void *p) { (void)u; if (p) (void)hipFree(p); }
int rb_upload(void *u, void *dst, const void *src, size_t bytes)
{
  ftkx_ctx *c = (ftkx_ctx *)u;
  HIP_TRY(c, hipSetDevice(c->device));
  HIP_TRY(c, hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, c->stream));
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  return FTKX_OK;
}
int rb_download(void *u, void *dst, const void *src, size_t bytes)
{
  ftkx_ctx *c = (ftkx_ctx *)u;
  HIP_TRY(c, hipSetDevice(c->device));
  HIP_TRY(c, hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  return FTKX_OK;
}
void rb_abort(void *u) { (void)ftkx_sweep_series_abort((ftkx_ctx *)u); }

void real_backend(ftkx_ctx *c, ftkx_slab_backend *b)
{
  memset(b, 0, sizeof(*b));
  b->user = c;
  b->begin = rb_begin; b->cull = rb_cull; b->serve = rb_serve; b->finish = rb_finish; b->complete = rb_complete; b->status = rb_status; b->recover = rb_recover;
  b->first_slice = rb_first_slice; b->alloc = rb_alloc; b->release = rb_release; b->upload = rb_upload; b->download = rb_download; b->abort = rb_abort;
  b->stream = c->stream; b->device = 1;
}

// sizes of the messages: known once the mesh is set and a slice has been pushed (the real backend: asked of the context at the first submit)
int size_up(ftkx_slab *s)
{
  if (s->sized) return FTKX_OK;
  if (s->ts.empty()) { s->sized = true; return FTKX_OK; }    // (a rank without timesteps takes part in the all_gather only: no slices, no messages of its own)
  if (s->own_ctx_backend) {
    ftkx_ctx *c = (ftkx_ctx *)s->be.user;
    if (!c->mesh_set || c->scalar_mode < 0) return sfail(s, FTKX_E_INVALID, "ftkx_slab: set the mesh and push this rank's slices before the first pass");
    s->be.masks_bytes = ftkx_packed_masks_bytes(c, nullptr);
    s->be.cells = ftkx_series_dist_cells(c);
    s->be.patch_doubles = ftkx_patch_doubles(c);
    s->be.slice_bytes = n_vertices(c) * (size_t)(c->scalar_mode == 1 ? 1 : c->nd) * sizeof(double);
    s->be.stream = c->stream;
  }
  s->masks_bytes = s->be.masks_bytes; s->cells = s->be.cells; s->pd = s->be.patch_doubles; s->slice_bytes = s->be.slice_bytes;
  if (s->own_ctx_backend && (s->masks_bytes == 0 || s->cells == 0)) { s->plain = true; s->sized = true; return FTKX_OK; }
  if ((s->lower >= 0 || s->upper >= 0) && (s->masks_bytes == 0 || s->cells == 0))
    return sfail(s, FTKX_E_UNSUPPORTED, "ftkx_slab: this mesh has no summarised masks (use the host-driven calls: ftkx_slices_prepare / ftkx_sweep_cull / ftkx_sweep_collect)");
  s->sized = true;
  return FTKX_OK;
}

int make_sets(ftkx_slab *s)
{
  if (s->sets_ready) return FTKX_OK;
  const double neutral[4] = {kDblMax, 0.0, kDblMax, 0.0};
  auto take = [&](size_t bytes) { return s->be.alloc(s->be.user, bytes); };
  for (Set &b : s->sets) {
    b.contrib = (double *)take(4 * sizeof(double));
    b.gathered = (double *)take(4 * sizeof(double) * (size_t)s->nranks);
    if (!b.contrib || !b.gathered) return sfail(s, FTKX_E_NOMEM, "ftkx_slab: out of memory");
    int rc = s->be.upload(s->be.user, b.contrib, neutral, sizeof(neutral));
    if (rc) return carry(s, rc, "ftkx_slab");
    if (s->plain) continue;
    if (s->lower >= 0) {
      b.masks_out = take(s->masks_bytes); b.req_in = (unsigned long long *)take((1 + s->cells) * 8); b.reply_out = (double *)take(s->cells * s->pd * 8);
      if (!b.masks_out || !b.req_in || !b.reply_out) return sfail(s, FTKX_E_NOMEM, "ftkx_slab: out of memory");
    }
    if (s->upper >= 0) {
      b.masks_in = take(s->masks_bytes); b.req_out = (unsigned long long *)take((1 + s->cells) * 8); b.reply_in = (double *)take(s->cells * s->pd * 8);
      if (!b.masks_in || !b.req_out || !b.reply_in) return sfail(s, FTKX_E_NOMEM, "ftkx_slab: out of memory");
    }
  }
  // queued transports: the masks' way to the lower neighbour runs on a stream of its own -- it starts as soon as the first slice's masks are
  // packed, next to the mask kernel of the slab's other slices; the context's stream waits for it in front of the cull
  if (s->tr.queued && s->be.device && !s->plain && !s->ts.empty() && (s->lower >= 0 || s->upper >= 0)) {
    if (hipStreamCreateWithFlags(&s->side, hipStreamNonBlocking) != hipSuccess || hipEventCreateWithFlags(&s->ev_side, hipEventDisableTiming) != hipSuccess)
      return sfail(s, FTKX_E_DEVICE, "ftkx_slab: no side stream");
  }
  s->sets_ready = true;
  return FTKX_OK;
}

int xchg(ftkx_slab *s, const void *send, size_t sb, int to, void *recv, size_t rb, int from, void *stream)
{
  if (to < 0 && from < 0) return FTKX_OK;
  if (to >= 0) s->bytes_sent += sb;
  if (from >= 0) s->bytes_received += rb;
  const int rc = s->tr.exchange(s->tr.user, to >= 0 ? send : nullptr, to >= 0 ? sb : 0, to, from >= 0 ? recv : nullptr, from >= 0 ? rb : 0, from, stream);
  return rc ? carry(s, rc, "ftkx_slab: exchange") : FTKX_OK;
}

int complete_local(ftkx_slab *s, Outcome &o, bool copy)
{
  const int n = (int)s->ts.size();
  o.f.assign((size_t)n, 0ull);
  o.run = kDblMax;
  const int rc = s->be.complete(s->be.user, &o.run, o.f.data(), &o.recs, &o.n);
  o.failed = false;
  if (rc == FTKX_E_NOSLICE) { o.failed = true; o.recs = nullptr; o.n = 0; }
  else if (rc) return carry(s, rc, "ftkx_slab_complete");
  o.gathered.assign(4 * (size_t)s->nranks, 0.0);
  const int rs = s->be.status(s->be.user, &o.asked, &o.served, o.gathered.data(), s->nranks, &o.path, &o.path_status);
  if (rs) return carry(s, rs, "ftkx_slab_complete (status)");
  if (copy && o.recs) { o.copy.assign(o.recs, o.recs + o.n); o.recs = o.copy.data(); }
  return FTKX_OK;
}

// the plain way (meshes without summarised masks): everything of the pass inside submit, its outcome kept for complete
int submit_plain(ftkx_slab *s, Set &b, double run)
{
  ftkx_ctx *c = (ftkx_ctx *)s->be.user;
  const int n = (int)s->ts.size();
  int rc;
  std::vector<double> res((size_t)n), mx((size_t)n);
  if ((rc = ftkx_slices_resolution(c, s->ts.data(), n, res.data(), mx.data()))) return carry(s, rc, "ftkx_slab_submit: resolution");
  double contrib[4] = {kDblMax, 0.0, res[0], mx[0]};
  for (int i = 0; i < n; i ++) { contrib[0] = std::min(contrib[0], res[(size_t)i]); contrib[1] = std::max(contrib[1], mx[(size_t)i]); }
  if ((rc = s->be.upload(s->be.user, b.contrib, contrib, sizeof(contrib)))) return carry(s, rc, "ftkx_slab_submit");
  if ((rc = s->tr.all_gather(s->tr.user, b.contrib, b.gathered, 4 * sizeof(double), s->be.stream))) return carry(s, rc, "ftkx_slab_submit: all_gather");
  std::vector<double> G(4 * (size_t)s->nranks);
  if ((rc = s->be.download(s->be.user, G.data(), b.gathered, G.size() * sizeof(double)))) return carry(s, rc, "ftkx_slab_submit");
  if (s->upper >= 0 && !s->full_halo && !(s->full_halo = s->be.alloc(s->be.user, s->slice_bytes))) return sfail(s, FTKX_E_NOMEM, "ftkx_slab_submit: out of memory for the halo slice");
  const void *first = s->lower >= 0 ? s->be.first_slice(s->be.user, s->t0) : nullptr;
  if (s->lower >= 0 && !first) return sfail(s, FTKX_E_NOSLICE, "ftkx_slab_submit: this rank's first slice %d is not resident", s->t0);
  if ((rc = xchg(s, first, s->slice_bytes, s->lower, s->full_halo, s->slice_bytes, s->upper, s->be.stream))) return rc;
  Outcome o;
  o.f.assign((size_t)n, 0ull);
  o.run = run;
  for (int r = 0; r < s->rank; r ++) o.run = std::min(o.run, G[4 * (size_t)r]);
  o.gathered = G;
  if (s->upper >= 0) rc = s->be.recover(s->be.user, s->t_halo, s->full_halo, s->ts.data(), s->scopes.data(), n, &o.run, o.f.data(), &o.recs, &o.n);
  else rc = ftkx_sweep_series(c, s->ts.data(), s->scopes.data(), n, &o.run, o.f.data(), &o.recs, &o.n);
  if (rc) return carry(s, rc, "ftkx_slab_submit: sweep");
  o.path = ftkx_series_last_path(c, &o.path_status);
  if (o.recs) o.copy.assign(o.recs, o.recs + o.n);
  // (the lower neighbour reads this rank's first slice on ITS stream -- the hub -- or on ours -- RCCL: either way it must have happened before
  // the caller may drop the slice; a pass of the plain kind ends with everybody's sweep done, which the next all_gather orders)
  s->plain_out.push_back(std::move(o));
  return FTKX_OK;
}

// this rank's steps and neighbours.  Periodic: slice nt is slice 0 again -- the last timestep's sweep is an interval sweep too, its halo is
// the first slice of the rank that owns timestep 0, which may be this rank itself (one rank: its own lower and upper neighbour)
void layout(ftkx_slab *s)
{
  const int nt = s->nt, nranks = s->nranks;
  s->ts.clear(); s->scopes.clear();
  for (int t = s->t0; t < s->t1; t ++) { s->ts.push_back(t); s->scopes.push_back((t + 1 < nt || s->periodic) ? FTKX_SCOPE_BOTH : FTKX_SCOPE_ORDINAL); }
  const bool own = s->t1 > s->t0;
  s->t_halo = (own && (s->t1 < nt || s->periodic)) ? s->t1 : -1;
  s->lower = (own && s->t0 > 0) ? ftkx_slab_owner(s->t0 - 1, nt, nranks) : (own && s->periodic) ? ftkx_slab_owner(nt - 1, nt, nranks) : -1;   // the rank whose last interval sweep reads OUR first slice
  s->upper = s->t_halo < 0 ? -1 : s->t1 < nt ? ftkx_slab_owner(s->t1, nt, nranks) : ftkx_slab_owner(0, nt, nranks);
}

}  // namespace

extern "C" {

void ftkx_slab_range(int nt, int nranks, int rank, int *t0, int *t1)
{
  if (t0) *t0 = (int)(((long long)rank * nt) / nranks);
  if (t1) *t1 = (int)(((long long)(rank + 1) * nt) / nranks);
}

int ftkx_slab_owner(int t, int nt, int nranks)
{
  for (int r = 0; r < nranks; r ++) { int a, b; ftkx_slab_range(nt, nranks, r, &a, &b); if (a <= t && t < b) return r; }
  return -1;
}

int ftkx_slab_create_custom(const ftkx_slab_backend *backend, int nt, int rank, int nranks, const ftkx_slab_transport *tr, ftkx_slab **out)
{
  if (!backend || !tr || !out || nt <= 0 || nranks <= 0 || rank < 0 || rank >= nranks) return fail(nullptr, FTKX_E_INVALID, "ftkx_slab_create: bad argument");
  if (!tr->all_gather || !tr->exchange) return fail(nullptr, FTKX_E_INVALID, "ftkx_slab_create: the transport needs all_gather and exchange");
  if (!backend->begin || !backend->cull || !backend->serve || !backend->finish || !backend->complete || !backend->status || !backend->recover || !backend->first_slice ||
      !backend->alloc || !backend->release || !backend->upload || !backend->download) return fail(nullptr, FTKX_E_INVALID, "ftkx_slab_create: incomplete backend table");
  ftkx_slab *s = new ftkx_slab();
  s->be = *backend; s->tr = *tr;
  s->nt = nt; s->rank = rank; s->nranks = nranks;
  ftkx_slab_range(nt, nranks, rank, &s->t0, &s->t1);
  layout(s);
  *out = s;
  return FTKX_OK;
}

int ftkx_slab_set_periodic(ftkx_slab *s, int on)
{
  if (!s) return FTKX_E_INVALID;
  if (s->sized || !s->open.empty()) return sfail(s, FTKX_E_INVALID, "ftkx_slab_set_periodic: before the first pass");
  s->periodic = on != 0;
  layout(s);
  return FTKX_OK;
}

int ftkx_slab_create(ftkx_ctx *ctx, int nt, int rank, int nranks, const ftkx_slab_transport *tr, ftkx_slab **out)
{
  if (!ctx) return fail(nullptr, FTKX_E_INVALID, "ftkx_slab_create: null context");
  ftkx_slab_backend b;
  real_backend(ctx, &b);
  const int rc = ftkx_slab_create_custom(&b, nt, rank, nranks, tr, out);
  if (rc == FTKX_OK) (*out)->own_ctx_backend = true;
  return rc;
}

void ftkx_slab_destroy(ftkx_slab *s)
{
  if (!s) return;
  if (!s->open.empty() && s->be.abort) s->be.abort(s->be.user);
  for (Set &b : s->sets)
    for (void *p : {(void *)b.contrib, (void *)b.gathered, b.masks_out, b.masks_in, (void *)b.req_out, (void *)b.req_in, (void *)b.reply_out, (void *)b.reply_in})
      if (p) s->be.release(s->be.user, p);
  if (s->full_halo) s->be.release(s->be.user, s->full_halo);
  if (s->ev_side) (void)hipEventDestroy(s->ev_side);
  if (s->side) (void)hipStreamDestroy(s->side);
  if (s->tr.destroy) s->tr.destroy(s->tr.user);
  delete s;
}

int ftkx_slab_submit(ftkx_slab *s, const double *running_resolution)
{
  if (!s) return FTKX_E_INVALID;
  if (s->open.size() >= 2) return sfail(s, FTKX_E_INVALID, "ftkx_slab_submit: two passes are open, complete one first");
  int rc;
  if ((rc = size_up(s)) || (rc = make_sets(s))) return rc;
  const int idx = s->k;
  Set &b = s->sets[idx];
  if (s->own_ctx_backend) s->be.stream = ((ftkx_ctx *)s->be.user)->stream;     // (ftkx_set_stream may have been called since)
  void *main = s->be.stream;
  const double run = running_resolution ? *running_resolution : kDblMax;
  if (s->ts.empty()) {             // (more ranks than timesteps: this rank only takes part in the all_gather)
    if ((rc = s->tr.all_gather(s->tr.user, b.contrib, b.gathered, 4 * sizeof(double), main))) return carry(s, rc, "ftkx_slab_submit: all_gather");
    s->k ^= 1; s->open.push_back(idx);
    return FTKX_OK;
  }
  if (s->plain) {
    if ((rc = submit_plain(s, b, run))) return rc;
    s->k ^= 1; s->open.push_back(idx);
    return FTKX_OK;
  }
  if ((rc = s->be.begin(s->be.user, s->ts.data(), s->scopes.data(), (int)s->ts.size(), &run, s->rank, s->nranks, s->upper, b.contrib, b.gathered,
                        s->lower >= 0 ? b.masks_out : nullptr, s->side))) return carry(s, rc, "ftkx_slab_submit: begin");
  auto bail = [&](int code) { if (s->be.abort) s->be.abort(s->be.user); return code; };
  if (s->side) {
    if ((rc = xchg(s, b.masks_out, s->masks_bytes, s->lower, b.masks_in, s->masks_bytes, s->upper, s->side))) return bail(rc);
    if ((rc = s->tr.all_gather(s->tr.user, b.contrib, b.gathered, 4 * sizeof(double), main))) return bail(carry(s, rc, "ftkx_slab_submit: all_gather"));
    if (hipEventRecord(s->ev_side, s->side) != hipSuccess || hipStreamWaitEvent((hipStream_t)main, s->ev_side, 0) != hipSuccess) return bail(sfail(s, FTKX_E_DEVICE, "ftkx_slab_submit: side stream"));
  } else {
    if ((rc = s->tr.all_gather(s->tr.user, b.contrib, b.gathered, 4 * sizeof(double), main))) return bail(carry(s, rc, "ftkx_slab_submit: all_gather"));
    if ((rc = xchg(s, b.masks_out, s->masks_bytes, s->lower, b.masks_in, s->masks_bytes, s->upper, main))) return bail(rc);
  }
  if ((rc = s->be.cull(s->be.user, s->upper >= 0 ? b.masks_in : nullptr, s->upper >= 0 ? b.req_out : nullptr))) return bail(carry(s, rc, "ftkx_slab_submit: cull"));
  if ((rc = xchg(s, b.req_out, (1 + s->cells) * 8, s->upper, b.req_in, (1 + s->cells) * 8, s->lower, main))) return bail(rc);
  if ((rc = s->be.serve(s->be.user, s->lower >= 0 ? b.req_in : nullptr, s->lower >= 0 ? b.reply_out : nullptr))) return bail(carry(s, rc, "ftkx_slab_submit: serve"));
  if ((rc = xchg(s, b.reply_out, s->cells * s->pd * 8, s->lower, b.reply_in, s->cells * s->pd * 8, s->upper, main))) return bail(rc);
  if ((rc = s->be.finish(s->be.user, s->upper >= 0 ? b.reply_in : nullptr))) return bail(carry(s, rc, "ftkx_slab_submit: finish"));
  b.running_in = run;
  s->k ^= 1; s->open.push_back(idx);
  return FTKX_OK;
}

int ftkx_slab_complete(ftkx_slab *s, double *running_resolution, unsigned long long *factors, const ftkx_cp_t **out, size_t *n_out)
{
  if (!s) return FTKX_E_INVALID;
  if (out) *out = nullptr;
  if (n_out) *n_out = 0;
  if (s->open.empty()) return sfail(s, FTKX_E_INVALID, "ftkx_slab_complete: no pass open");
  const int idx = s->open.front();
  s->open.pop_front();
  Set &b = s->sets[idx];
  if (s->ts.empty()) { if (running_resolution) *running_resolution = kDblMax; return FTKX_OK; }
  int rc;
  Outcome o;
  if (s->plain) {
    o = std::move(s->plain_out.front()); s->plain_out.pop_front();
    o.recs = o.copy.empty() ? nullptr : o.copy.data();
    s->fallbacks += s->upper >= 0 ? 1 : 0;
  } else
  if (!s->stash.empty()) { o = std::move(s->stash.front()); s->stash.pop_front(); if (!o.copy.empty()) o.recs = o.copy.data(); }
  else if ((rc = complete_local(s, o, false))) return rc;
  const bool need = o.asked < 0 && s->upper >= 0, give = o.served < 0 && s->lower >= 0;
  if (need || give) {
    // the whole slice after all: both sides know from the same number.  The context must be free for the second sweep: the pass queued
    // behind this one (if any) is collected first, its outcome kept for the next call
    if (need && !s->open.empty() && s->stash.empty()) {
      if (o.recs && o.copy.empty()) { o.copy.assign(o.recs, o.recs + o.n); o.recs = o.copy.data(); }
      Outcome nx;
      if ((rc = complete_local(s, nx, true))) return rc;
      s->stash.push_back(std::move(nx));
      if (!s->stash.back().copy.empty()) s->stash.back().recs = s->stash.back().copy.data();
    }
    if (need) {
      s->fallbacks ++;
      if (!s->full_halo && !(s->full_halo = s->be.alloc(s->be.user, s->slice_bytes))) return sfail(s, FTKX_E_NOMEM, "ftkx_slab_complete: out of memory for the halo slice");
    }
    const void *first = give ? s->be.first_slice(s->be.user, s->t0) : nullptr;
    if (give && !first) return sfail(s, FTKX_E_NOSLICE, "ftkx_slab_complete: this rank's first slice %d is not resident any more (the lower neighbour needs it as a whole)", s->t0);
    if ((rc = xchg(s, first, s->slice_bytes, give ? s->lower : -1, s->full_halo, s->slice_bytes, need ? s->upper : -1, s->be.stream))) return rc;
    if (need) {
      double run_in = b.running_in;
      for (int r = 0; r < s->rank; r ++) run_in = std::min(run_in, o.gathered[4 * (size_t)r]);
      o.f.assign(s->ts.size(), 0ull);
      o.run = run_in;
      // (recover pushes the slice: the stream's order puts that behind the transport's copy)
      if ((rc = s->be.recover(s->be.user, s->t_halo, s->full_halo, s->ts.data(), s->scopes.data(), (int)s->ts.size(), &o.run, o.f.data(), &o.recs, &o.n))) return carry(s, rc, "ftkx_slab_complete: recovery");
      o.copy.clear();
      if (s->own_ctx_backend) o.path = ftkx_series_last_path((ftkx_ctx *)s->be.user, &o.path_status);
    }
  }
  s->last_path = o.path; s->last_status = o.path_status; s->last_asked = o.asked; s->last_served = o.served;
  // the records: the context's own buffer where it is still the last thing the context did, our copy otherwise
  if (!o.copy.empty()) { s->held.swap(o.copy); o.recs = s->held.data(); }
  s->held_f = o.f;
  if (running_resolution) *running_resolution = o.run;
  if (factors) for (size_t i = 0; i < o.f.size(); i ++) factors[i] = o.f[i];
  if (out) *out = o.recs;
  if (n_out) *n_out = o.n;
  return FTKX_OK;
}

int ftkx_slab_gather_records(ftkx_slab *s, const ftkx_cp_t *mine, size_t n, int root, ftkx_cp_t **merged, size_t *n_merged)
{
  if (!s || root < 0 || root >= s->nranks || (n && !mine)) return sfail(s, FTKX_E_INVALID, "ftkx_slab_gather_records: bad argument");
  if (merged) *merged = nullptr;
  if (n_merged) *n_merged = 0;
  int rc;
  const ftkx_slab_backend &be = s->be;
  void *stream = be.stream;
  unsigned long long *d_mine = (unsigned long long *)be.alloc(be.user, 8), *d_all = (unsigned long long *)be.alloc(be.user, 8 * (size_t)s->nranks);
  std::vector<unsigned long long> counts((size_t)s->nranks, 0ull);
  void *d_recs = nullptr;
  std::vector<std::vector<ftkx_cp_t>> parts;
  auto done = [&](int code) { for (void *p : {(void *)d_mine, (void *)d_all, d_recs}) if (p) be.release(be.user, p); return code; };
  if (!d_mine || !d_all) return done(sfail(s, FTKX_E_NOMEM, "ftkx_slab_gather_records: out of memory"));
  const unsigned long long cnt = (unsigned long long)n;
  if ((rc = be.upload(be.user, d_mine, &cnt, 8))) return done(carry(s, rc, "ftkx_slab_gather_records"));
  if ((rc = s->tr.all_gather(s->tr.user, d_mine, d_all, 8, stream))) return done(carry(s, rc, "ftkx_slab_gather_records: all_gather"));
  if ((rc = be.download(be.user, counts.data(), d_all, 8 * (size_t)s->nranks))) return done(carry(s, rc, "ftkx_slab_gather_records"));
  if (s->rank != root) {
    if (n) {
      if (!(d_recs = be.alloc(be.user, n * sizeof(ftkx_cp_t)))) return done(sfail(s, FTKX_E_NOMEM, "ftkx_slab_gather_records: out of memory"));
      if ((rc = be.upload(be.user, d_recs, mine, n * sizeof(ftkx_cp_t)))) return done(carry(s, rc, "ftkx_slab_gather_records"));
      if ((rc = xchg(s, d_recs, n * sizeof(ftkx_cp_t), root, nullptr, 0, -1, stream))) return done(rc);
    }
    // (a sender's buffer may go only when the root has taken the message: with a transport that copies on the RECEIVER's stream -- the hub --
    // the sender's own stream does not say so.  The root joins this all_gather after it has downloaded every message.)
    if ((rc = s->tr.all_gather(s->tr.user, d_mine, d_all, 8, stream))) return done(carry(s, rc, "ftkx_slab_gather_records: all_gather"));
    if ((rc = be.download(be.user, counts.data(), d_all, 8 * (size_t)s->nranks))) return done(carry(s, rc, "ftkx_slab_gather_records"));
    return done(FTKX_OK);
  }
  size_t total = 0, largest = 0;
  for (int r = 0; r < s->nranks; r ++) { total += (size_t)counts[(size_t)r]; if (r != root) largest = std::max(largest, (size_t)counts[(size_t)r]); }
  ftkx_cp_t *all = (ftkx_cp_t *)malloc(std::max<size_t>(total, 1) * sizeof(ftkx_cp_t));
  if (!all) return done(sfail(s, FTKX_E_NOMEM, "ftkx_slab_gather_records: out of memory"));
  if (largest && !(d_recs = be.alloc(be.user, largest * sizeof(ftkx_cp_t)))) { free(all); return done(sfail(s, FTKX_E_NOMEM, "ftkx_slab_gather_records: out of memory")); }
  size_t at = 0;
  for (int r = 0; r < s->nranks; r ++) {            // slabs in rank order: with 64-bit tags already the order of the tags (time is their slowest axis)
    const size_t c = (size_t)counts[(size_t)r];
    if (r == root) { if (n) memcpy(all + at, mine, n * sizeof(ftkx_cp_t)); at += n; continue; }
    if (!c) continue;
    if ((rc = xchg(s, nullptr, 0, -1, d_recs, c * sizeof(ftkx_cp_t), r, stream))) { free(all); return done(rc); }
    if ((rc = be.download(be.user, all + at, d_recs, c * sizeof(ftkx_cp_t)))) { free(all); return done(carry(s, rc, "ftkx_slab_gather_records")); }
    at += c;
  }
  if ((rc = s->tr.all_gather(s->tr.user, d_mine, d_all, 8, stream)) || (rc = be.download(be.user, counts.data(), d_all, 8 * (size_t)s->nranks))) { free(all); return done(carry(s, rc, "ftkx_slab_gather_records")); }
  auto less = [](const ftkx_cp_t &a, const ftkx_cp_t &b) { return a.tag < b.tag; };
  if (!std::is_sorted(all, all + total, less)) std::stable_sort(all, all + total, less);     // (independent of the tag mode)
  if (merged) *merged = all; else free(all);
  if (n_merged) *n_merged = total;
  return done(FTKX_OK);
}

/* synchronous copies between host memory and memory of the context's device, behind everything queued on the context's stream (host-staged
 * transports: ftk_amd/tslab.py over gloo) */
int ftkx_upload(ftkx_ctx *c, void *dst, const void *host_src, size_t bytes) { return (c && dst && host_src) ? rb_upload(c, dst, host_src, bytes) : FTKX_E_INVALID; }
int ftkx_download(ftkx_ctx *c, void *host_dst, const void *src, size_t bytes) { return (c && host_dst && src) ? rb_download(c, host_dst, src, bytes) : FTKX_E_INVALID; }

int ftkx_slab_get_info(const ftkx_slab *s, ftkx_slab_info *info)
{
  if (!s || !info) return FTKX_E_INVALID;
  memset(info, 0, sizeof(*info));
  info->rank = s->rank; info->nranks = s->nranks; info->t0 = s->t0; info->t1 = s->t1; info->lower = s->lower; info->upper = s->upper;
  info->bytes_sent = s->bytes_sent; info->bytes_received = s->bytes_received; info->fallbacks = s->fallbacks;
  info->last_asked = s->last_asked; info->last_served = s->last_served; info->last_path = s->last_path; info->last_status = s->last_status;
  info->open = (int)s->open.size();
  return FTKX_OK;
}

const char *ftkx_slab_last_error(const ftkx_slab *s) { return s ? s->err.c_str() : ""; }

}  // extern "C"

// ---- ranks of one process: the hub ---------------------------------------------------------------------------------------------------------
// A rank posts what it has to say -- pointer, device, an event recorded on its stream behind the kernel that produced the data -- and the
// receiver, once the post is there, makes ITS stream wait for that event and queues a peer copy.  Nothing waits for the device on the host;
// what a host thread waits for is the peer's CALL.  A sender's buffer is written again two passes later, behind kernels of the pass in
// between that wait (through the same chain of events) for messages the receiver produced after its copy: the stream order carries it.
struct ftkx_slab_hub {
  int n = 0;
  std::mutex mu;
  std::condition_variable cv;
  bool aborted = false;
  struct Post { const void *ptr; size_t bytes; int device; hipEvent_t ev; };
  struct Round { std::vector<Post> posts; int posted = 0, taken = 0; };
  std::map<unsigned long long, Round> rounds;             // all_gather number g -> every rank's post
  std::map<std::pair<int, int>, std::deque<Post>> box;    // (from, to) -> messages in order
  struct Side { int rank = 0, device = 0; std::vector<hipEvent_t> ring; size_t next = 0; unsigned long long gathers = 0; ftkx_slab_hub *hub = nullptr; };
  std::vector<Side> sides;
};

namespace {

hipEvent_t hub_event(ftkx_slab_hub::Side &S, hipStream_t st)
{
  if (S.ring.empty()) { S.ring.resize(64, nullptr); for (auto &e : S.ring) (void)hipEventCreateWithFlags(&e, hipEventDisableTiming); }
  hipEvent_t e = S.ring[S.next ++ % S.ring.size()];        // (64 posts back: long consumed -- a pass posts at most four)
  (void)hipEventRecord(e, st);
  return e;
}

int hub_all_gather(void *user, const void *send, void *recv, size_t bytes, void *stream)
{
  ftkx_slab_hub::Side &S = *(ftkx_slab_hub::Side *)user;
  ftkx_slab_hub *H = S.hub;
  if (hipSetDevice(S.device) != hipSuccess) return fail(nullptr, FTKX_E_DEVICE, "hub: hipSetDevice");
  hipStream_t st = (hipStream_t)stream;
  const ftkx_slab_hub::Post mine{send, bytes, S.device, hub_event(S, st)};
  std::vector<ftkx_slab_hub::Post> all;
  {
    std::unique_lock<std::mutex> lk(H->mu);
    const unsigned long long g = S.gathers ++;             // this rank's g-th all_gather meets everybody's g-th
    ftkx_slab_hub::Round &R = H->rounds[g];
    if (R.posts.empty()) R.posts.resize((size_t)H->n);
    R.posts[(size_t)S.rank] = mine; R.posted ++;
    H->cv.notify_all();
    H->cv.wait(lk, [&] { return H->aborted || H->rounds[g].posted == H->n; });
    if (H->aborted) return fail(nullptr, FTKX_E_DEVICE, "hub: a peer gave up");
    ftkx_slab_hub::Round &R2 = H->rounds[g];
    all = R2.posts;
    if (++ R2.taken == H->n) H->rounds.erase(g);
  }
  for (int r = 0; r < H->n; r ++) {
    const ftkx_slab_hub::Post &p = all[(size_t)r];
    if (p.bytes != bytes) return fail(nullptr, FTKX_E_INVALID, "hub: all_gather of %zu bytes meets one of %zu", bytes, p.bytes);
    if (hipStreamWaitEvent(st, p.ev, 0) != hipSuccess) return fail(nullptr, FTKX_E_DEVICE, "hub: hipStreamWaitEvent");
    if (hipMemcpyPeerAsync((char *)recv + (size_t)r * bytes, S.device, p.ptr, p.device, bytes, st) != hipSuccess) return fail(nullptr, FTKX_E_DEVICE, "hub: hipMemcpyPeerAsync");
  }
  return FTKX_OK;
}

int hub_exchange(void *user, const void *send, size_t sb, int to, void *recv, size_t rb, int from, void *stream)
{
  ftkx_slab_hub::Side &S = *(ftkx_slab_hub::Side *)user;
  ftkx_slab_hub *H = S.hub;
  if (hipSetDevice(S.device) != hipSuccess) return fail(nullptr, FTKX_E_DEVICE, "hub: hipSetDevice");
  hipStream_t st = (hipStream_t)stream;
  ftkx_slab_hub::Post got{nullptr, 0, 0, nullptr};
  {
    std::unique_lock<std::mutex> lk(H->mu);
    if (to >= 0) { H->box[{S.rank, to}].push_back(ftkx_slab_hub::Post{send, sb, S.device, hub_event(S, st)}); H->cv.notify_all(); }
    if (from >= 0) {
      auto &q = H->box[{from, S.rank}];
      H->cv.wait(lk, [&] { return H->aborted || !q.empty(); });
      if (H->aborted) return fail(nullptr, FTKX_E_DEVICE, "hub: a peer gave up");
      got = q.front(); q.pop_front();
    }
  }
  if (from >= 0) {
    if (got.bytes != rb) return fail(nullptr, FTKX_E_INVALID, "hub: a message of %zu bytes where %zu were expected (rank %d <- %d)", got.bytes, rb, S.rank, from);
    if (hipStreamWaitEvent(st, got.ev, 0) != hipSuccess) return fail(nullptr, FTKX_E_DEVICE, "hub: hipStreamWaitEvent");
    if (hipMemcpyPeerAsync(recv, S.device, got.ptr, got.device, rb, st) != hipSuccess) return fail(nullptr, FTKX_E_DEVICE, "hub: hipMemcpyPeerAsync");
  }
  return FTKX_OK;
}

}  // namespace

extern "C" {

ftkx_slab_hub *ftkx_slab_hub_create(int nranks)
{
  if (nranks <= 0) return nullptr;
  ftkx_slab_hub *H = new ftkx_slab_hub();
  H->n = nranks;
  H->sides.resize((size_t)nranks);
  for (int r = 0; r < nranks; r ++) { H->sides[(size_t)r].rank = r; H->sides[(size_t)r].hub = H; }
  return H;
}

void ftkx_slab_hub_abort(ftkx_slab_hub *H)
{
  if (!H) return;
  { std::lock_guard<std::mutex> g(H->mu); H->aborted = true; }
  H->cv.notify_all();
}

void ftkx_slab_hub_destroy(ftkx_slab_hub *H)
{
  if (!H) return;
  for (auto &S : H->sides) for (hipEvent_t e : S.ring) if (e) (void)hipEventDestroy(e);
  delete H;
}

int ftkx_slab_create_local(ftkx_ctx *ctx, int nt, int rank, ftkx_slab_hub *H, ftkx_slab **out)
{
  if (!ctx || !H || rank < 0 || rank >= H->n) return fail(nullptr, FTKX_E_INVALID, "ftkx_slab_create_local: bad argument");
  ftkx_slab_hub::Side &S = H->sides[(size_t)rank];
  S.device = ctx->device;
  ftkx_slab_transport tr;
  memset(&tr, 0, sizeof(tr));
  tr.user = &S; tr.all_gather = hub_all_gather; tr.exchange = hub_exchange; tr.queued = 1; tr.destroy = nullptr;      // (the hub outlives its slabs: ftkx_slab_hub_destroy)
  return ftkx_slab_create(ctx, nt, rank, H->n, &tr, out);
}

}  // extern "C"
