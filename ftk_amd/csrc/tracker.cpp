// Host-side tracker with the reference's operator interface (include/ftkx_tracker.hh).  Pure host logic over the C ABI:
// snapshot window, sticky quantisation factor, ordinal + interval sweep per step, record bookkeeping.
// Reference counterparts are cited per method.
#include "../../include/ftkx_tracker.hh"
#include "host_sort.hpp"

#include <algorithm>
#include <chrono>
#include <cmath>
#include <condition_variable>
#include <cstring>
#include <deque>
#include <functional>
#include <mutex>
#include <set>
#include <thread>

namespace ftkx {

critical_point_tracker_regular::critical_point_tracker_regular(int nd_, int device_id) : nd(nd_)
{
  std::memset(&last_stats, 0, sizeof(last_stats));
  int rc = ftkx_create(&ctx, nd, device_id);
  if (rc != FTKX_OK) {
    char buf[512];
    ftkx_last_error(nullptr, buf, sizeof(buf));
    throw ftkx_error(rc, buf);   // no device, no tracker: there is no CPU path behind this class
  }
}

// ---- several devices behind one tracker ---------------------------------------------------------------------------------------
struct critical_point_tracker_regular::multi_engine {
  struct worker {
    ftkx_ctx *ctx = nullptr;
    std::thread th;
    std::mutex mu;
    std::condition_variable cv;
    std::deque<std::function<void()>> q;
    bool stop = false, busy = false;
    unsigned long long posted = 0, done = 0;     // jobs handed in / finished (FIFO: `done >= ticket` says a particular job is through)
  };
  std::vector<std::unique_ptr<worker>> w;
  int block = 2, t_first = -1;
  std::map<int, std::set<int>> resident;       // timestep -> workers that hold the slice
  // the reduction board: what each slice contributes to the sticky running minimum, published by whichever step reduces it first
  std::mutex bmu;
  std::condition_variable bcv;
  std::map<int, double> res_below;
  double base_resolution = std::numeric_limits<double>::max();   // the running minimum before this series (never reset, like the reference's)
  // errors raised inside jobs surface at the next sync()
  std::mutex emu;
  int error_code = 0;
  std::string error;
  std::mutex rmu;                              // guards the tracker's result members while jobs write them

  int dev_of(int t) const { return ((t - t_first) / block) % (int)w.size(); }

  void run(worker *W)
  {
    for (;;) {
      std::function<void()> f;
      {
        std::unique_lock<std::mutex> lk(W->mu);
        W->cv.wait(lk, [&] { return W->stop || !W->q.empty(); });
        if (W->q.empty()) return;
        f = std::move(W->q.front()); W->q.pop_front(); W->busy = true;
      }
      try { f(); }
      catch (const ftkx_error &e) { raise(e.code, e.what()); }
      catch (const std::exception &e) { raise(FTKX_E_INVALID, e.what()); }
      { std::lock_guard<std::mutex> lk(W->mu); W->busy = false; W->done ++; }
      W->cv.notify_all();
    }
  }
  // A failing job wakes every step that waits on the board.  The error is set under emu, the notify goes out under bmu: a waiter that
  // has evaluated its predicate (failed() == false) and not yet blocked holds bmu, so the notify cannot fall into that gap.
  void raise(int code, const char *what)
  {
    { std::lock_guard<std::mutex> g(emu); if (!error_code) { error_code = code; error = what; } }
    { std::lock_guard<std::mutex> g(bmu); }
    bcv.notify_all();
  }
  unsigned long long post(int d, std::function<void()> f)
  {
    unsigned long long ticket;
    { std::lock_guard<std::mutex> lk(w[d]->mu); w[d]->q.push_back(std::move(f)); ticket = ++ w[d]->posted; }
    w[d]->cv.notify_all();
    return ticket;
  }
  // waits for ONE job (its ticket), not for the device's queue to drain: a push only has to know that its own copy is through, while the
  // sweeps queued behind it on that device run on
  void wait_job(int d, unsigned long long ticket)
  {
    std::unique_lock<std::mutex> lk(w[d]->mu);
    w[d]->cv.wait(lk, [&] { return w[d]->done >= ticket; });
  }
  void wait(int d)
  {
    std::unique_lock<std::mutex> lk(w[d]->mu);
    w[d]->cv.wait(lk, [&] { return w[d]->q.empty() && !w[d]->busy; });
  }
  void wait_all() { for (size_t d = 0; d < w.size(); d ++) wait((int)d); }
  bool failed() { std::lock_guard<std::mutex> g(emu); return error_code != 0; }
  void rethrow()
  {
    std::lock_guard<std::mutex> g(emu);
    if (error_code) { const int c = error_code; const std::string m = error; error_code = 0; error.clear(); throw ftkx_error(c, m); }
  }
  // running minimum over what has been published of the slices BEFORE t (any subset of them bounds the factor from below)
  double known_before(int t)
  {
    std::lock_guard<std::mutex> g(bmu);
    double r = base_resolution;
    for (const auto &kv : res_below) if (kv.first < t) r = std::min(r, kv.second);
    return r;
  }
  void publish(const std::vector<int> &ts, const std::vector<double> &below)
  {
    {
      std::lock_guard<std::mutex> g(bmu);
      for (size_t i = 0; i < ts.size(); i ++) {
        auto it = res_below.find(ts[i]);
        if (it == res_below.end()) res_below[ts[i]] = below[i]; else it->second = std::min(it->second, below[i]);
      }
    }
    bcv.notify_all();
  }
  // the reference's vector_field_resolution at the step whose last snapshot is t_last: needs every slice up to it
  double wait_running_min(int t_last)
  {
    std::unique_lock<std::mutex> lk(bmu);
    bcv.wait(lk, [&] {
      if (failed()) return true;
      for (int s = t_first; s <= t_last; s ++) if (!res_below.count(s)) return false;
      return true;
    });
    double r = base_resolution;
    for (int s = t_first; s <= t_last; s ++) { auto it = res_below.find(s); if (it != res_below.end()) r = std::min(r, it->second); }
    return r;
  }
  ~multi_engine()
  {
    for (auto &W : w) { { std::lock_guard<std::mutex> lk(W->mu); W->stop = true; } W->cv.notify_all(); }
    for (auto &W : w) if (W->th.joinable()) W->th.join();
    for (auto &W : w) ftkx_destroy(W->ctx);
  }
};

critical_point_tracker_regular::critical_point_tracker_regular(int nd_, const std::vector<int> &device_ids, int block) : nd(nd_)
{
  std::memset(&last_stats, 0, sizeof(last_stats));
  if (device_ids.empty() || block < 1) throw ftkx_error(FTKX_E_INVALID, "critical_point_tracker_regular: need at least one device and block >= 1");
  if (device_ids.size() == 1) {
    const int rc = ftkx_create(&ctx, nd, device_ids[0]);
    if (rc != FTKX_OK) { char buf[512]; ftkx_last_error(nullptr, buf, sizeof(buf)); throw ftkx_error(rc, buf); }
    return;
  }
  multi.reset(new multi_engine());
  multi->block = block;
  for (int dev : device_ids) {
    std::unique_ptr<multi_engine::worker> W(new multi_engine::worker());
    const int rc = ftkx_create(&W->ctx, nd, dev);
    if (rc != FTKX_OK) { char buf[512]; ftkx_last_error(nullptr, buf, sizeof(buf)); multi.reset(); throw ftkx_error(rc, buf); }
    multi->w.push_back(std::move(W));
  }
  for (auto &W : multi->w) { multi_engine *e = multi.get(); multi_engine::worker *wp = W.get(); W->th = std::thread([e, wp] { e->run(wp); }); }
  ctx = multi->w[0]->ctx;     // (context() of a multi-device tracker: its first device's)
}

critical_point_tracker_regular::~critical_point_tracker_regular()
{
  ftkx_online_tracer_destroy(online);
  if (slab) ftkx_slab_destroy(slab);
  if (multi) multi.reset();   // joins the workers, destroys their contexts (ctx is one of them)
  else ftkx_destroy(ctx);
}

// ---- several ranks behind the tracker: slab mode -----------------------------------------------------------------------------------------
void critical_point_tracker_regular::enter_slab_mode(int rank, int nranks, int nt)
{
  if (multi) throw ftkx_error(FTKX_E_UNSUPPORTED, "slab mode: a tracker with one device (several devices of one process: one tracker per device over a ftkx_slab_hub)");
  if (slab || !field_data_snapshots.empty()) throw ftkx_error(FTKX_E_INVALID, "slab mode: set it once, before the first snapshot is pushed");
  if (enable_streaming_trajectories) throw ftkx_error(FTKX_E_UNSUPPORTED, "slab mode: not with streaming trajectories");
  if (nt <= 0 || nranks <= 0 || rank < 0 || rank >= nranks) throw ftkx_error(FTKX_E_INVALID, "slab mode: bad rank / nranks / nt");
  slab_nt = nt; slab_rank = rank; slab_nranks = nranks;
  ftkx_slab_range(nt, nranks, rank, &slab_t0, &slab_t1);
  current_timestep = next_push_timestep = slab_t0;          // the first snapshot this rank pushes is its slab's first timestep
  slab_steps.clear(); slab_swept = false;
}

void critical_point_tracker_regular::set_communicator(void *comm, int rank, int nranks, int nt)
{
  enter_slab_mode(rank, nranks, nt);
  check(ftkx_slab_create_rccl(ctx, nt, rank, nranks, comm, nullptr, &slab));
}

void critical_point_tracker_regular::set_slab_transport(const ftkx_slab_transport &tr, int rank, int nranks, int nt)
{
  enter_slab_mode(rank, nranks, nt);
  check(ftkx_slab_create(ctx, nt, rank, nranks, &tr, &slab));
}

void critical_point_tracker_regular::set_slab_hub(ftkx_slab_hub *hub, int rank, int nt)
{
  if (!hub) throw ftkx_error(FTKX_E_INVALID, "set_slab_hub: null hub");
  ftkx_slab *probe = nullptr;
  // (the hub knows how many ranks it has: the slab says so)
  check(ftkx_slab_create_local(ctx, nt, rank, hub, &probe));
  ftkx_slab_info info;
  ftkx_slab_get_info(probe, &info);
  try { enter_slab_mode(rank, info.nranks, nt); } catch (...) { ftkx_slab_destroy(probe); throw; }
  slab = probe;
}

// the slab's pass: every step recorded so far must be the slab's -- t0 .. t1 - 1 in order -- and every snapshot still resident
void critical_point_tracker_regular::run_slab() const
{
  critical_point_tracker_regular *self = const_cast<critical_point_tracker_regular *>(this);
  const int nown = slab_t1 - slab_t0;
  if ((int)slab_steps.size() != nown || (int)field_data_snapshots.size() != nown)
    throw ftkx_error(FTKX_E_INVALID, "slab mode: " + std::to_string(slab_steps.size()) + " steps recorded and " + std::to_string(field_data_snapshots.size()) +
                     " snapshots pushed of this rank's " + std::to_string(nown) + " (push the slab's snapshots, advance_timestep() between them, update_timestep() after the last)");
  for (int i = 0; i < nown; i ++) if (slab_steps[(size_t)i] != slab_t0 + i) throw ftkx_error(FTKX_E_INVALID, "slab mode: the steps must be the slab's timesteps in order");
  auto slab_check = [&](int rc) { if (rc != FTKX_OK) throw ftkx_error(rc, ftkx_slab_last_error(slab)); };
  slab_check(ftkx_slab_submit(slab, &vector_field_resolution));
  std::vector<unsigned long long> f((size_t)std::max(nown, 1), 0ull);
  const ftkx_cp_t *recs = nullptr;
  size_t n = 0;
  double run = vector_field_resolution;
  slab_check(ftkx_slab_complete(slab, &run, f.data(), &recs, &n));
  self->slab_swept = true;
  if (nown > 0) {
    self->vector_field_resolution = std::min(vector_field_resolution, run);
    self->vector_field_scaling_factor = f[(size_t)nown - 1];
    for (size_t i = 0; i < n; i ++) self->take_records(recs + i, 1, ftkx_cp_timestep(recs + i));
    check(ftkx_get_stats(ctx, &self->last_stats));
  }
}

void critical_point_tracker_regular::wait_devices() const
{
  if (slab) { if (!slab_swept) run_slab(); return; }
  if (multi) {
    multi->wait_all();
    multi->rethrow();
  }
  if (!batch_ts.empty()) submit_batch();                       // deferred collection in batches: the steps recorded since the last full batch
  while (!open_steps.empty()) collect_open_step();             // deferred collection: what is still out
}

// deferred collection: the oldest queued sweep -> its records, factor, running minimum and statistics, as update_timestep leaves them
void critical_point_tracker_regular::collect_open_step() const
{
  critical_point_tracker_regular *self = const_cast<critical_point_tracker_regular *>(this);
  const int t = open_steps.front();
  open_steps.erase(open_steps.begin());
  const ftkx_cp_t *recs = nullptr;
  size_t n = 0;
  unsigned long long f = 0;
  double res = vector_field_resolution;
  const int rc = ftkx_sweep_series_complete(ctx, &res, &f, &recs, &n);
  if (rc != FTKX_OK) {
    // the other open pass (if any) is discarded with it: the context is left as if nothing had been queued, so that the tracker's later
    // calls do not fail with "passes open" or chain from a stale running minimum
    char why[512] = "";
    ftkx_last_error(ctx, why, sizeof(why));
    open_steps.clear();
    (void)ftkx_sweep_series_abort(ctx);
    batch_ts.clear(); batch_scopes.clear();
    for (int d : batch_drops) (void)ftkx_drop_slice(ctx, d);
    batch_drops.clear();
    throw ftkx_error(rc, why);
  }
  self->vector_field_resolution = res;
  self->vector_field_scaling_factor = f;
  if (t >= 0) self->take_records(recs, n, t);
  else for (size_t i = 0; i < n;) {                             // a batch: runs of one timestep (the records are ordered by tag, not by step)
    size_t j = i + 1;
    while (j < n && ftkx_cp_timestep(&recs[j]) == ftkx_cp_timestep(&recs[i])) j ++;
    self->take_records(recs + i, j - i, ftkx_cp_timestep(&recs[i]));
    i = j;
  }
  check(ftkx_get_stats(ctx, &self->last_stats));
}

// deferred collection in batches: the recorded steps as one pass (continuing, on the device, from the pass queued before it if that one is
// still out); the snapshots popped since the batch began are dropped behind it -- their buffers are reused in stream order
void critical_point_tracker_regular::submit_batch() const
{
  critical_point_tracker_regular *self = const_cast<critical_point_tracker_regular *>(this);
  const bool chained = !open_steps.empty();
  std::vector<int> ts, scopes, drops;
  ts.swap(batch_ts); scopes.swap(batch_scopes); drops.swap(batch_drops);
  int rc = ftkx_sweep_series_submit(ctx, ts.data(), scopes.data(), (int)ts.size(), chained ? nullptr : &self->vector_field_resolution);
  if (rc == FTKX_OK) open_steps.push_back(-1);
  for (int t : drops) { const int rc2 = ftkx_drop_slice(ctx, t); if (rc == FTKX_OK) rc = rc2; }
  check(rc);
  if (open_steps.size() >= 3) collect_open_step();      // (three in flight: the host runs one pass ahead of a split pass's tail)
}

void critical_point_tracker_regular::sync() const
{
  wait_devices();
  flush_points();
}

// the parked records into the flat store: ordered by the element order's integer key, merged with what is there (a later point with
// the same tag replaces the earlier one, like operator[] of the reference's map)
void critical_point_tracker_regular::flush_points() const
{
  if (pending_points.empty()) return;
  typedef std::pair<std::pair<unsigned long long, unsigned long long>, size_t> keyed;
  std::vector<keyed> order(pending_points.size());
  for (size_t i = 0; i < pending_points.size(); i ++) order[i] = {order_.key(pending_points[i].tag), i};
  ftkx::sort_on_threads(order);                                  // (equal keys = equal tags: their insertion order, the index, breaks the tie)
  std::vector<feature_point_t> merged;
  std::vector<std::pair<unsigned long long, unsigned long long>> keys;
  merged.reserve(points.size() + order.size()); keys.reserve(points.size() + order.size());
  size_t a = 0, b = 0;
  while (a < points.size() || b < order.size()) {
    if (b < order.size() && b + 1 < order.size() && order[b + 1].first == order[b].first) { b ++; continue; }   // the later of two pending points with one tag
    if (b == order.size() || (a < points.size() && point_keys[a] < order[b].first)) { merged.push_back(points[a]); keys.push_back(point_keys[a]); a ++; }
    else {
      if (a < points.size() && point_keys[a] == order[b].first) a ++;                                            // replaced
      merged.push_back(pending_points[order[b].second]); keys.push_back(order[b].first); b ++;
    }
  }
  points.swap(merged); point_keys.swap(keys);
  pending_points.clear();
  pending_ascending = true;
  map_valid = false;
}

const discrete_map_t &critical_point_tracker_regular::get_discrete_critical_points() const
{
  sync();
  if (!map_valid) {
    discrete_map_t fresh(order_);
    for (const feature_point_t &cp : points) fresh.emplace_hint(fresh.end(), cp.tag, cp);     // (already in the map's order: linear)
    discrete_critical_points.swap(fresh);
    map_valid = true;
  }
  return discrete_critical_points;
}

int critical_point_tracker_regular::num_devices() const { return multi ? (int)multi->w.size() : 1; }

void critical_point_tracker_regular::check(int rc) const
{
  if (rc == FTKX_OK) return;
  char buf[512];
  ftkx_last_error(ctx, buf, sizeof(buf));
  throw ftkx_error(rc, buf);
}

void critical_point_tracker_regular::set_stream(void *s)
{
  if (multi) throw ftkx_error(FTKX_E_UNSUPPORTED, "set_stream: a multi-device tracker runs each device on its context's own stream");
  check(ftkx_set_stream(ctx, s));
}

// regular_tracker::initialize, regular_tracker.hh:105-149.  Single process per GPU: the partitioner returns the whole domain
// (local_domain == domain) and, with is_input_array_partial == false, local_array_domain == array_domain.
void critical_point_tracker_regular::apply_configuration(ftkx_ctx *c)
{
  auto ck = [&](int rc) { if (rc != FTKX_OK) { char buf[512]; ftkx_last_error(c, buf, sizeof(buf)); throw ftkx_error(rc, buf); } };
  long long dst[3] = {0, 0, 0}, dsz[3] = {1, 1, 1}, est[3] = {0, 0, 0}, esz[3] = {1, 1, 1};
  for (int d = 0; d < nd; d ++) { dst[d] = domain.start(d); dsz[d] = domain.size(d); est[d] = array_domain.start(d); esz[d] = array_domain.size(d); }
  ck(ftkx_set_mesh(c, dst, dsz, dst, dsz, est, esz));
  ftkx_options o;
  ftkx_default_options(&o);
  o.jacobian_symmetric = is_jacobian_field_symmetric;
  o.robust = enable_robust_detection;
  o.use_type_filter = use_type_filter;
  o.type_filter = type_filter;
  o.compute_degrees = enable_computing_degrees;
  o.tag_mode = tag_mode;
  o.exact_only = exact_only;
  o.derive_jacobian = jacobian_field_source == SOURCE_DERIVED;
  if (mode_phys_coords == 2) {
    if ((int)rectilinear_coords.size() < nd) throw ftkx_error(FTKX_E_INVALID, "initialize: set_coords_rectilinear needs one array per axis");
    const std::vector<double> none;
    const std::vector<double> &z = nd == 3 ? rectilinear_coords[2] : none;
    ck(ftkx_set_coords_rectilinear(c, rectilinear_coords[0].data(), rectilinear_coords[0].size(), rectilinear_coords[1].data(), rectilinear_coords[1].size(),
                                   z.empty() ? nullptr : z.data(), z.size()));
  } else if (mode_phys_coords == 3)
    ck(ftkx_set_coords_explicit(c, explicit_coords.data(), explicit_ncomp, explicit_n0, explicit_n1));
  o.coords_mode = mode_phys_coords;
  for (size_t i = 0; i < 6 && i < bounds_coords.size(); i ++) o.coords_bounds[i] = bounds_coords[i];
  ck(ftkx_set_options(c, &o));
}

void critical_point_tracker_regular::initialize()
{
  if ((int)domain.nd() != nd || (int)array_domain.nd() != nd) throw ftkx_error(FTKX_E_INVALID, "initialize: set_domain / set_array_domain first");
  local_domain = domain;
  local_array_domain = array_domain;
  sync();
  if (multi) { for (auto &W : multi->w) apply_configuration(W->ctx); }
  else apply_configuration(ctx);
  // the discrete points are kept in the reference's element order, which needs the mesh sizes: what is there is ordered afresh
  order_.nd = nd;
  for (int d = 0; d < nd; d ++) order_.n[d] = domain.size(d);
  if (!points.empty()) {
    std::vector<feature_point_t> again;
    again.swap(points); point_keys.clear();
    again.insert(again.end(), pending_points.begin(), pending_points.end());
    pending_points.swap(again);
    pending_ascending = false;
  }
  discrete_critical_points = discrete_map_t(order_);
  map_valid = false;
  initialized = true;
}

// critical_point_tracker_2d_regular::reset, 2d:227-236 (+ critical_point_tracker::reset, critical_point_tracker.hh:31-34).
// Like the reference, the running resolution is NOT reset.
void critical_point_tracker_regular::reset()
{
  if (!slab) sync();                            // (slab mode: sync() is the slab's pass, a collective -- a reset does not sweep)
  ftkx_online_tracer_destroy(online); online = nullptr;
  if (multi) { multi->base_resolution = vector_field_resolution; multi->res_below.clear(); multi->t_first = -1; }
  current_timestep = 0;
  while (pop_field_data_snapshot()) {}
  next_push_timestep = 0;
  if (slab) { current_timestep = next_push_timestep = slab_t0; slab_steps.clear(); slab_swept = false; }
  pending_points.clear(); pending_ascending = true;
  points.clear(); point_keys.clear();
  discrete_critical_points.clear(); map_valid = true;
}

// one snapshot -> the context(s) whose steps read it.  kind: 0 scalar (V derived), 1 vector, 2 all three given
static int push_to(ftkx_ctx *c, int kind, int t, const double *s, const double *v, const double *j, int on_device)
{
  if (kind == 0) return ftkx_push_scalar_slice(c, t, s, on_device);
  if (kind == 1) return ftkx_push_slice(c, t, v, nullptr, nullptr, on_device);
  return ftkx_push_slice(c, t, v, j, s, on_device);
}

void critical_point_tracker_regular::push_everywhere(int kind, int t, const double *s, const double *v, const double *j, bool device)
{
  auto fail = [](ftkx_ctx *c, int rc) { char buf[512]; ftkx_last_error(c, buf, sizeof(buf)); throw ftkx_error(rc, buf); };
  if (slab && (t < slab_t0 || t >= slab_t1)) throw ftkx_error(FTKX_E_INVALID, "slab mode: timestep " + std::to_string(t) + " is not in this rank's slab [" + std::to_string(slab_t0) + ", " + std::to_string(slab_t1) + ")");
  if (!multi) { const int rc = push_to(ctx, kind, t, s, v, j, device ? 1 : 0); if (rc) fail(ctx, rc); return; }
  if (multi->t_first < 0) multi->t_first = t;
  std::set<int> targets;
  targets.insert(multi->dev_of(t));
  if (t > multi->t_first) targets.insert(multi->dev_of(t - 1));        // the interval sweep [t-1, t] of the previous block reads it too
  // Device memory is COPIED into each context (a peer copy where the devices differ; the contexts recycle their slice buffers, so
  // no allocation per step): the steps run later than the calls that queue them, and the caller's buffer is free again on return
  // -- adopting the pointer would tie its lifetime to a queue the caller cannot see.
  std::vector<std::pair<int, unsigned long long>> tickets;
  for (int d : targets) {
    ftkx_ctx *c = multi->w[d]->ctx;
    tickets.push_back({d, multi->post(d, [=] { const int rc = push_to(c, kind, t, s, v, j, device ? 2 : 0); if (rc) { char buf[512]; ftkx_last_error(c, buf, sizeof(buf)); throw ftkx_error(rc, buf); } })});
  }
  for (const auto &tk : tickets) multi->wait_job(tk.first, tk.second);     // (the caller's buffer is free again once these copies are through)
  multi->rethrow();
  multi->resident[t] = targets;
}

void critical_point_tracker_regular::push_scalar_field_snapshot(const double *s, bool device)
{
  if (!initialized) throw ftkx_error(FTKX_E_INVALID, "push: initialize() first");
  if (vector_field_source != SOURCE_DERIVED) throw ftkx_error(FTKX_E_INVALID, "push_scalar_field_snapshot: vector_field_source must be SOURCE_DERIVED");
  const int t = next_push_timestep;
  push_everywhere(0, t, s, nullptr, nullptr, device);       // V = gradientND(s) on the device
  field_data_snapshots.push_back(t);
  next_push_timestep ++;
}

void critical_point_tracker_regular::push_vector_field_snapshot(const double *v, bool device)
{
  if (!initialized) throw ftkx_error(FTKX_E_INVALID, "push: initialize() first");
  const int t = next_push_timestep;
  push_everywhere(1, t, nullptr, v, nullptr, device);       // J derived at hits when jacobian_field_source == SOURCE_DERIVED
  field_data_snapshots.push_back(t);
  next_push_timestep ++;
}

void critical_point_tracker_regular::push_field_data_snapshot(const double *s, const double *v, const double *j, bool device)
{
  if (!initialized) throw ftkx_error(FTKX_E_INVALID, "push: initialize() first");
  const int t = next_push_timestep;
  push_everywhere(2, t, s, v, j, device);
  field_data_snapshots.push_back(t);
  next_push_timestep ++;
}

bool critical_point_tracker_regular::pop_field_data_snapshot()
{
  if (field_data_snapshots.empty()) return false;
  const int t = field_data_snapshots.front();
  if (multi) {
    for (int d : multi->resident[t]) {           // queued behind the steps that still read the slice
      ftkx_ctx *c = multi->w[d]->ctx;
      multi->post(d, [=] { const int rc = ftkx_drop_slice(c, t); if (rc) { char buf[512]; ftkx_last_error(c, buf, sizeof(buf)); throw ftkx_error(rc, buf); } });
    }
    multi->resident.erase(t);
  } else if (!batch_ts.empty() && t >= batch_ts.front()) batch_drops.push_back(t);      // (a recorded step still reads it: dropped behind its batch)
  else check(ftkx_drop_slice(ctx, t));
  field_data_snapshots.erase(field_data_snapshots.begin());
  return true;
}

// critical_point_tracker::update_vector_field_scaling_factor, critical_point_tracker.hh:850-864: sticky running minimum of
// ndarray::resolution() over every queued snapshot.  The per-slice reduction is fused into the pass that builds the slice's
// sign masks (ftkx_slices_prepare: each snapshot is read from HBM once, when it first takes part in a step), under the factor
// in force so far -- which the running minimum can only raise.  What comes back per slice is its smallest non-zero |v| BELOW
// 1 / that factor (or DBL_MAX): values at or above it cannot change nbits, so the factor sequence is the reference's, while
// vector_field_resolution itself (never observable in the reference either) is exact only once it is below 2^-8.
void critical_point_tracker_regular::update_vector_field_scaling_factor(int minbits, int maxbits)
{
  const unsigned long long hint = std::max(vector_field_scaling_factor, 1ull << minbits);   // never above the factor this step ends up with
  std::vector<double> below(field_data_snapshots.size());
  if (!field_data_snapshots.empty())
    check(ftkx_slices_prepare(ctx, field_data_snapshots.data(), (int)field_data_snapshots.size(), hint, below.data(), nullptr));
  for (double r : below) vector_field_resolution = std::min(vector_field_resolution, r);
  int nbits = (int)std::ceil(std::log2(1.0 / vector_field_resolution));
  nbits = std::max(minbits, std::min(nbits, maxbits));
  vector_field_scaling_factor = 1ull << nbits;
}

// critical_point_tracker_{2d,3d}_regular::update_timestep (2d:263-433, 3d:150-308): ordinal sweep at current_timestep and,
// when two snapshots are queued, the interval sweep [current, current+1] -- here one launch, one download.
void critical_point_tracker_regular::take_records(const ftkx_cp_t *recs, size_t n, int timestep)
{
  for (size_t i = 0; i < n; i ++) {
    feature_point_t cp;
    for (int k = 0; k < 3; k ++) { cp.x[k] = recs[i].x[k]; cp.scalar[k] = recs[i].scalar[k]; }
    cp.t = recs[i].t;
    cp.type = recs[i].type;
    cp.tag = recs[i].tag;
    cp.ordinal = ftkx_cp_ordinal(&recs[i]) != 0;
    cp.timestep = timestep;
    if (scalar_field_source == SOURCE_NONE) cp.scalar[0] = 0.0;   // 2d:642-646: scalar only when a scalar field exists
    if (!pending_points.empty() && pending_points.back().tag >= cp.tag) pending_ascending = false;
    pending_points.push_back(cp);                                 // -> the flat store at the next sync()
  }
}

void critical_point_tracker_regular::update_timestep()
{
  if (field_data_snapshots.empty()) return;
  if (slab) {                                   // slab mode: the step is recorded; the slab is swept as one pass (run_slab)
    if (slab_swept) throw ftkx_error(FTKX_E_INVALID, "slab mode: the slab has been swept (reset() starts a new series)");
    if (slab_steps.empty() || slab_steps.back() != current_timestep) slab_steps.push_back(current_timestep);
    return;
  }
  const int scope = field_data_snapshots.size() >= 2 ? FTKX_SCOPE_BOTH : FTKX_SCOPE_ORDINAL;
  if (!multi) {
    // (the step that follows is known before its factor is: announced, its cull is queued right behind the mask kernel of the
    // newly arrived snapshot and runs while the host waits for the reduction)
    const ftkx_cp_t *recs = nullptr;
    size_t n = 0;
    if (deferred_collection && !enable_streaming_trajectories && field_data_snapshots.size() <= 2 && field_data_snapshots.front() == current_timestep) {
      // queue this step (continuing, on the device, from the running minimum of the step queued before it if that one is still out),
      // then collect the step before it
      if (deferred_depth > 1) {                 // batches: recorded; queued with the batch's last step (submit_batch)
        batch_ts.push_back(current_timestep); batch_scopes.push_back(scope);
        if ((int)batch_ts.size() >= deferred_depth) submit_batch();
        return;
      }
      const bool chained = !open_steps.empty();
      check(ftkx_sweep_series_submit(ctx, &current_timestep, &scope, 1, chained ? nullptr : &vector_field_resolution));
      open_steps.push_back(current_timestep);
      if (open_steps.size() >= 3) collect_open_step();      // (three in flight: the host runs one pass ahead of a split pass's tail)
      return;
    }
    if (!batch_ts.empty()) submit_batch();
    while (!open_steps.empty()) collect_open_step();
    if (field_data_snapshots.size() <= 2 && field_data_snapshots.front() == current_timestep) {
      // The device-driven pass (ftkx_sweep_series): the newly arrived snapshot's masks and reduction, the sticky factor (formed on the
      // device from the running minimum handed in), cull, exact test and records are queued at once and waited for once.
      unsigned long long f = 0;
      check(ftkx_sweep_series(ctx, &current_timestep, &scope, 1, &vector_field_resolution, &f, &recs, &n));
      vector_field_scaling_factor = f;
    } else {
      // (more than two snapshots queued: the reference's factor takes every queued snapshot into account, critical_point_tracker.hh:853)
      check(ftkx_sweep_announce(ctx, &current_timestep, &scope, 1));
      update_vector_field_scaling_factor();
      check(ftkx_sweep(ctx, current_timestep, scope, vector_field_scaling_factor, &recs, &n));
    }
    take_records(recs, n, current_timestep);
    check(ftkx_get_stats(ctx, &last_stats));
    if (enable_streaming_trajectories && scope == FTKX_SCOPE_BOTH) grow();      // 2d:326-330, 3d:197-201: only after an interval sweep
    return;
  }
  if (enable_streaming_trajectories) throw ftkx_error(FTKX_E_UNSUPPORTED, "enable_streaming_trajectories: single-device trackers only");
  // Several devices: the step is queued on the device that owns its timestep and this call returns.  Inside the job: reduce
  // (and mask) the step's slices under the factor known so far, publish their contribution, wait for the contributions of ALL
  // earlier slices -- other devices publish theirs before they sweep, so this is a wait for reductions only -- and sweep under
  // the factor the reference would have at this step.
  multi->rethrow();
  const int t = current_timestep, d = multi->dev_of(t);
  multi_engine *e = multi.get();
  ftkx_ctx *c = e->w[d]->ctx;
  std::vector<int> ts(field_data_snapshots.begin(), field_data_snapshots.begin() + (scope == FTKX_SCOPE_BOTH ? 2 : 1));
  e->post(d, [this, e, c, t, scope, ts] {
    auto ck = [&](int rc) { if (rc != FTKX_OK) { char buf[512]; ftkx_last_error(c, buf, sizeof(buf)); throw ftkx_error(rc, buf); } };
    const int minbits = 8, maxbits = 21;
    auto factor_of = [&](double res) { int nb = (int)std::ceil(std::log2(1.0 / res)); nb = std::max(minbits, std::min(nb, maxbits)); return 1ull << nb; };
    std::vector<double> below(ts.size());
    ck(ftkx_sweep_announce(c, &t, &scope, 1));
    ck(ftkx_slices_prepare(c, ts.data(), (int)ts.size(), factor_of(e->known_before(t)), below.data(), nullptr));
    e->publish(ts, below);
    const double res = e->wait_running_min(ts.back());
    if (e->failed()) return;                    // another step failed: do not sweep with a partial minimum
    const unsigned long long factor = factor_of(res);
    const ftkx_cp_t *recs = nullptr;
    size_t n = 0;
    ck(ftkx_sweep(c, t, scope, factor, &recs, &n));
    ftkx_stats st;
    ck(ftkx_get_stats(c, &st));
    std::lock_guard<std::mutex> g(e->rmu);
    take_records(recs, n, t);
    // steps finish out of order: the members keep the values of the LATEST timestep, as a sequential run would leave them
    if (t >= result_timestep) { result_timestep = t; last_stats = st; vector_field_scaling_factor = factor; vector_field_resolution = res; }
  });
}

// critical_point_tracker::advance_timestep, critical_point_tracker.hh:841-848
bool critical_point_tracker_regular::advance_timestep()
{
  update_timestep();
  if (slab) { current_timestep ++; return current_timestep < next_push_timestep; }      // (slab mode: the snapshots stay until the slab's pass has run)
  pop_field_data_snapshot();
  current_timestep ++;
  return field_data_snapshots.size() > 0;
}

// critical_point_tracker_{2d,3d}_regular::finalize (2d:143-225, 3d:86-117) without streaming trajectories:
// traced_critical_points = trace_critical_points_offline(discrete_critical_points, neighbours-sharing-a-cell)
// trace_critical_points_online (critical_point_tracker.hh:523-639) on everything in discrete_critical_points, which it consumes
void critical_point_tracker_regular::grow()
{
  flush_points();
  long long dst[3] = {0, 0, 0}, dsz[3] = {1, 1, 1};
  for (int d = 0; d < nd; d ++) { dst[d] = domain.start(d); dsz[d] = domain.size(d); }
  if (!online) { const int rc = ftkx_online_tracer_create(&online, nd, dst, dsz); if (rc != FTKX_OK) throw ftkx_error(rc, "online tracer"); }
  std::vector<ftkx_cp_t> recs;
  recs.reserve(points.size());
  for (const feature_point_t &cp : points) {
    ftkx_cp_t r;
    std::memset(&r, 0, sizeof(r));
    for (int k = 0; k < 3; k ++) { r.x[k] = cp.x[k]; r.scalar[k] = cp.scalar[k]; }
    r.t = cp.t; r.type = cp.type; r.tag = cp.tag;
    reinterpret_cast<unsigned int *>(&r)[15] = ((unsigned)cp.timestep << 1) | (cp.ordinal ? 1u : 0u);
    recs.push_back(r);
  }
  const int rc = ftkx_online_tracer_grow(online, recs.data(), recs.size());
  if (rc != FTKX_OK) throw ftkx_error(rc, "grow: ftkx_online_tracer_grow failed (element tags needed: FTKX_TAG_EXACT64, or REFERENCE where it does not wrap)");
  points.clear(); point_keys.clear();
  discrete_critical_points.clear(); map_valid = true;
}

void critical_point_tracker_regular::finalize()
{
  wait_devices();
  if (slab) {
    // critical_point_tracker.hh:689: the ranks' discrete points gathered on the root, which traces them -- curves cross slab boundaries like
    // any other cell boundary.  The other ranks keep their own points and end without trajectories.
    flush_points();
    std::vector<ftkx_cp_t> mine(points.size());
    for (size_t i = 0; i < points.size(); i ++) {
      const feature_point_t &cp = points[i];
      ftkx_cp_t r;
      std::memset(&r, 0, sizeof(r));
      for (int k = 0; k < 3; k ++) { r.x[k] = cp.x[k]; r.scalar[k] = cp.scalar[k]; }
      r.t = cp.t; r.type = cp.type; r.tag = cp.tag;
      reinterpret_cast<unsigned int *>(&r)[15] = ((unsigned)cp.timestep << 1) | (cp.ordinal ? 1u : 0u);
      mine[i] = r;
    }
    ftkx_cp_t *merged = nullptr;
    size_t nm = 0;
    const int rc = ftkx_slab_gather_records(slab, mine.data(), mine.size(), 0, &merged, &nm);
    if (rc != FTKX_OK) throw ftkx_error(rc, ftkx_slab_last_error(slab));
    if (slab_rank != 0) {
      traced_points.clear(); traced_offsets.assign(1, 0); traced_nested_valid = false; traced_loop.clear(); traced_id.clear();
      return;
    }
    points.clear(); point_keys.clear(); pending_points.clear(); pending_ascending = true;
    discrete_critical_points.clear(); map_valid = false;
    for (size_t i = 0; i < nm; i ++) take_records(merged + i, 1, ftkx_cp_timestep(merged + i));
    ftkx_free(merged);
  }
  if (enable_streaming_trajectories) {
    flush_points();             // 2d:150-151: "done" -- the trajectories are what grow() built
    traced_points.clear(); traced_offsets.assign(1, 0); traced_nested_valid = false; traced_loop.clear(); traced_id.clear();
    if (!online) return;
    ftkx_cp_t *pts = nullptr;
    ftkx_curves c{};
    const int rc = ftkx_online_tracer_curves(online, &pts, &c);
    if (rc != FTKX_OK) { ftkx_free(pts); ftkx_free_curves(&c); throw ftkx_error(rc, "finalize: ftkx_online_tracer_curves failed"); }
    for (size_t i = 0; i < c.n_curves; i ++) {
      for (long long k = c.offsets[i]; k < c.offsets[i + 1]; k ++) {
        const ftkx_cp_t &r = pts[k];
        feature_point_t cp;
        for (int q = 0; q < 3; q ++) { cp.x[q] = r.x[q]; cp.scalar[q] = r.scalar[q]; }
        cp.t = r.t; cp.type = r.type; cp.tag = r.tag;
        cp.ordinal = ftkx_cp_ordinal(&r) != 0; cp.timestep = ftkx_cp_timestep(&r);
        traced_points.push_back(cp);
      }
      traced_offsets.push_back((long long)traced_points.size());
      traced_loop.push_back(c.loop[i]);
      traced_id.push_back((int)i);
    }
    ftkx_free(pts); ftkx_free_curves(&c);
    return;
  }
  // Nothing has to be ordered for this: ftkx_trace_curves takes the points in any order (it indexes them by tag) and returns the curves
  // in the reference's order.  Sweeps in time order with 64-bit tags deliver ascending tags: the pending points are traced as they are.
  constexpr bool timing = false;      // (phase timing to stderr: a debugging aid, compiled out)
  typedef std::chrono::steady_clock clk;
  const clk::time_point tq0 = clk::now();
  if (!(points.empty() && pending_ascending)) flush_points();
  const clk::time_point tq1 = clk::now();
  const std::vector<feature_point_t> &src = points.empty() ? pending_points : points;
  // (the flat store is in the reference's element order; the trace's device phases want ascending tags: an index sorted on a few
  // threads, and the curves' indices mapped back through it)
  std::vector<std::pair<unsigned long long, size_t>> by_tag;
  if (!points.empty()) {
    by_tag.resize(src.size());
    for (size_t i = 0; i < src.size(); i ++) by_tag[i] = {src[i].tag, i};
    ftkx::sort_on_threads(by_tag);
  }
  std::vector<unsigned long long> tags(src.size());           // (the trace reads nothing but the tags)
  for (size_t i = 0; i < src.size(); i ++) tags[i] = by_tag.empty() ? src[i].tag : by_tag[i].first;
  long long dst[3] = {0, 0, 0}, dsz[3] = {1, 1, 1};
  for (int d = 0; d < nd; d ++) { dst[d] = domain.start(d); dsz[d] = domain.size(d); }
  ftkx_curves c{};
  const clk::time_point tq2 = clk::now();
  // (neighbour search and component labelling on the tracker's GPU where the record set is large enough to pay for the round trip)
  const int rc = ftkx_trace_curves_tags_ctx(ctx, nd, dst, dsz, tags.data(), tags.size(), &c);
  const clk::time_point tq3 = clk::now();
  if (rc != FTKX_OK) { ftkx_free_curves(&c); throw ftkx_error(rc, "finalize: ftkx_trace_curves failed (tags must not have overflowed int32: use FTKX_TAG_EXACT64 on very large meshes)"); }
  // the curves stay flat -- the points of all curves one after the other; one vector per curve is built only if somebody asks for it
  traced_points.resize(c.n_points); traced_offsets.assign(c.offsets, c.offsets + c.n_curves + 1);
  traced_loop.assign(c.loop, c.loop + c.n_curves); traced_id.resize(c.n_curves);
  ftkx::for_ranges_on_threads(c.n_points, [&](size_t b, size_t e) {
    for (size_t k = b; k < e; k ++) traced_points[k] = src[by_tag.empty() ? (size_t)c.indices[k] : by_tag[(size_t)c.indices[k]].second];
  });
  for (size_t i = 0; i < c.n_curves; i ++) traced_id[i] = (int)i;
  traced_nested_valid = false;
  ftkx_free_curves(&c);
  if (timing) {
    auto us = [](clk::time_point a, clk::time_point b) { return std::chrono::duration<double, std::micro>(b - a).count(); };
    fprintf(stderr, "tracker finalize: flush %.0f us, tag index + records %.0f us, trace %.0f us, curves -> points %.0f us\n", us(tq0, tq1), us(tq1, tq2), us(tq2, tq3), us(tq3, clk::now()));
  }
}

namespace {
ftkx_cp_t record_of(const feature_point_t &cp)
{
  ftkx_cp_t r;
  std::memset(&r, 0, sizeof(r));
  for (int k = 0; k < 3; k ++) { r.x[k] = cp.x[k]; r.scalar[k] = cp.scalar[k]; }
  r.t = cp.t; r.type = cp.type; r.tag = cp.tag;
  reinterpret_cast<unsigned int *>(&r)[15] = ((unsigned)cp.timestep << 1) | (cp.ordinal ? 1u : 0u);
  return r;
}
feature_point_t point_of(const ftkx_cp_t &r)
{
  feature_point_t cp;
  for (int k = 0; k < 3; k ++) { cp.x[k] = r.x[k]; cp.scalar[k] = r.scalar[k]; }
  cp.t = r.t; cp.type = r.type; cp.tag = r.tag;
  cp.ordinal = ftkx_cp_ordinal(&r) != 0; cp.timestep = ftkx_cp_timestep(&r);
  return cp;
}
struct flat_curves {   // traced curves as one record array (points in curve order) + the identity as index list
  std::vector<ftkx_cp_t> recs;
  std::vector<long long> indices;
  flat_curves(const std::vector<feature_point_t> &pts) : recs(pts.size()), indices(pts.size())
  {
    ftkx::for_ranges_on_threads(pts.size(), [&](size_t b, size_t e) { for (size_t i = b; i < e; i ++) { recs[i] = record_of(pts[i]); indices[i] = (long long)i; } });
  }
};
std::string io_error()
{
  char msg[512] = {0};
  ftkx_last_error(nullptr, msg, sizeof msg);
  return msg;
}
}  // namespace

// json_interface::post_process (filters/json_interface.hh:758-800) with the options it defaults to
void critical_point_tracker_regular::post_process()
{
  flat_curves f(traced_points);
  ftkx_curves in;
  std::memset(&in, 0, sizeof(in));
  in.n_curves = num_traced_curves(); in.n_points = f.recs.size();
  in.offsets = traced_offsets.data(); in.indices = f.indices.data(); in.loop = traced_loop.data();
  ftkx_trajectories out{};
  const int rc = ftkx_post_process_curves(f.recs.data(), f.recs.size(), &in, &out);
  if (rc != FTKX_OK) { ftkx_free_trajectories(&out); throw ftkx_error(rc, "post_process failed"); }
  std::vector<feature_point_t> pts(out.n_points);
  std::vector<int> loop(out.n_curves), ids(out.n_curves);
  ftkx::for_ranges_on_threads(out.n_points, [&](size_t b, size_t e) {
    for (size_t k = b; k < e; k ++) {
      pts[k] = traced_points[(size_t)out.indices[k]];
      pts[k].type = out.type[k]; pts[k].t = out.t[k];
    }
  });
  // a split piece keeps its parent's label; labels of traced curves are their own (possibly already post-processed) ids
  for (size_t c = 0; c < out.n_curves; c ++) { loop[c] = out.loop[c]; ids[c] = traced_id[out.id[c]]; }
  traced_offsets.assign(out.offsets, out.offsets + out.n_curves + 1);
  ftkx_free_trajectories(&out);
  traced_points.swap(pts); traced_loop.swap(loop); traced_id.swap(ids);
  traced_nested_valid = false;
}

const std::vector<std::vector<feature_point_t>> &critical_point_tracker_regular::get_traced_critical_points() const
{
  if (!traced_nested_valid) {
    traced_critical_points.assign(num_traced_curves(), std::vector<feature_point_t>());
    for (size_t c = 0; c < num_traced_curves(); c ++)
      traced_critical_points[c].assign(traced_points.begin() + traced_offsets[c], traced_points.begin() + traced_offsets[c + 1]);
    traced_nested_valid = true;
  }
  return traced_critical_points;
}

void critical_point_tracker_regular::write_discrete(const std::string &filename, int format) const
{
  sync();
  std::vector<ftkx_cp_t> recs;
  recs.reserve(points.size());
  for (const feature_point_t &cp : points) recs.push_back(record_of(cp));
  // text labels: the reference's scalar_components default to {"scalar"} whether or not a scalar field exists
  const int rc = ftkx_write_critical_points(filename.c_str(), format, recs.data(), recs.size(), nullptr, nullptr, nullptr, -1);
  if (rc != FTKX_OK) throw ftkx_error(rc, io_error());
}
void critical_point_tracker_regular::write_critical_points_json(const std::string &f) const { write_discrete(f, FTKX_FORMAT_JSON); }
void critical_point_tracker_regular::write_critical_points_binary(const std::string &f) const { write_discrete(f, FTKX_FORMAT_BINARY); }
void critical_point_tracker_regular::write_critical_points_text(const std::string &f) const { write_discrete(f, FTKX_FORMAT_TEXT); }

// critical_point_tracker_regular::put_critical_points (critical_point_tracker_regular.hh:40-46): tags are trusted as keys
void critical_point_tracker_regular::put_critical_points(const std::vector<feature_point_t> &cps)
{
  for (const auto &cp : cps) {
    if (!pending_points.empty() && pending_points.back().tag >= cp.tag) pending_ascending = false;
    pending_points.push_back(cp);
  }
}

void critical_point_tracker_regular::read_discrete(const std::string &filename, int format)
{
  ftkx_cp_t *recs = nullptr;
  size_t n = 0;
  const int rc = ftkx_read_critical_points(filename.c_str(), format, &recs, &n, nullptr, nullptr);
  if (rc != FTKX_OK) throw ftkx_error(rc, io_error());
  std::vector<feature_point_t> cps;
  cps.reserve(n);
  for (size_t i = 0; i < n; i ++) cps.push_back(point_of(recs[i]));
  ftkx_free(recs);
  put_critical_points(cps);
}
void critical_point_tracker_regular::read_critical_points_json(const std::string &f) { read_discrete(f, FTKX_FORMAT_JSON); }
void critical_point_tracker_regular::read_critical_points_binary(const std::string &f) { read_discrete(f, FTKX_FORMAT_BINARY); }

void critical_point_tracker_regular::write_traced(const std::string &filename, int format) const
{
  flat_curves f(traced_points);
  std::vector<unsigned> type(f.recs.size() ? f.recs.size() : 1);
  std::vector<double> t(f.recs.size() ? f.recs.size() : 1);
  for (size_t i = 0; i < f.recs.size(); i ++) { type[i] = f.recs[i].type; t[i] = f.recs[i].t; }
  std::vector<int> loop(traced_loop), ids(traced_id);
  loop.resize(num_traced_curves() + 1); ids.resize(num_traced_curves() + 1);
  std::vector<long long> offsets(traced_offsets);
  ftkx_trajectories tr;
  std::memset(&tr, 0, sizeof(tr));
  tr.n_curves = num_traced_curves(); tr.n_points = f.recs.size();
  tr.offsets = offsets.data(); tr.indices = f.indices.data(); tr.loop = loop.data(); tr.type = type.data(); tr.t = t.data(); tr.id = ids.data();
  const int rc = ftkx_write_traced_critical_points(filename.c_str(), format, f.recs.data(), f.recs.size(), &tr, nullptr, -1);
  if (rc != FTKX_OK) throw ftkx_error(rc, io_error());
}
void critical_point_tracker_regular::write_traced_critical_points_json(const std::string &f) const { write_traced(f, FTKX_FORMAT_JSON); }
void critical_point_tracker_regular::write_traced_critical_points_binary(const std::string &f) const { write_traced(f, FTKX_FORMAT_BINARY); }
void critical_point_tracker_regular::write_traced_critical_points_text(const std::string &f) const { write_traced(f, FTKX_FORMAT_TEXT); }

std::vector<feature_point_t> critical_point_tracker_regular::get_critical_points() const
{
  sync();
  return points;
}

}  // namespace ftkx

// ---- C handles ---------------------------------------------------------------------------------------------------
struct ftkx_tracker {
  ftkx::critical_point_tracker_regular *t = nullptr;
  int nd = 0;
  std::string err;
};

namespace {
thread_local std::string g_tracker_error;

template <class F>
int guarded(ftkx_tracker *h, F f)
{
  if (!h || !h->t) { g_tracker_error = "null tracker"; return FTKX_E_INVALID; }
  try { f(); return FTKX_OK; }
  catch (const ftkx::ftkx_error &e) { h->err = e.what(); g_tracker_error = e.what(); return e.code; }
  catch (const std::exception &e) { h->err = e.what(); g_tracker_error = e.what(); return FTKX_E_INVALID; }
}
}  // namespace

extern "C" {

int ftkx_tracker_create(ftkx_tracker **out, int nd, int device_id)
{
  if (!out || (nd != 2 && nd != 3)) { g_tracker_error = "ftkx_tracker_create: nd must be 2 or 3"; return FTKX_E_INVALID; }
  ftkx_tracker *h = new ftkx_tracker();
  try {
    h->nd = nd;
    h->t = new ftkx::critical_point_tracker_regular(nd, device_id);
    *out = h;
    return FTKX_OK;
  } catch (const ftkx::ftkx_error &e) { delete h; g_tracker_error = e.what(); return e.code; }
  catch (const std::exception &e) { delete h; g_tracker_error = e.what(); return FTKX_E_INVALID; }
}

int ftkx_tracker_create_multi(ftkx_tracker **out, int nd, const int *device_ids, int ndev, int block)
{
  if (!out || (nd != 2 && nd != 3) || !device_ids || ndev < 1) { g_tracker_error = "ftkx_tracker_create_multi: bad arguments"; return FTKX_E_INVALID; }
  ftkx_tracker *h = new ftkx_tracker();
  try {
    h->nd = nd;
    h->t = new ftkx::critical_point_tracker_regular(nd, std::vector<int>(device_ids, device_ids + ndev), block);
    *out = h;
    return FTKX_OK;
  } catch (const ftkx::ftkx_error &e) { delete h; g_tracker_error = e.what(); return e.code; }
  catch (const std::exception &e) { delete h; g_tracker_error = e.what(); return FTKX_E_INVALID; }
}

int ftkx_tracker_sync(ftkx_tracker *h) { return guarded(h, [&] { h->t->sync(); }); }

void ftkx_tracker_destroy(ftkx_tracker *h) { if (h) { delete h->t; delete h; } }

int ftkx_tracker_last_error(const ftkx_tracker *h, char *buf, size_t n)
{
  const std::string &e = h ? h->err : g_tracker_error;
  if (buf && n) { strncpy(buf, e.c_str(), n - 1); buf[n - 1] = 0; }
  return (int)e.size();
}

int ftkx_tracker_set_domain(ftkx_tracker *h, const long long *st, const long long *sz)
{ return guarded(h, [&] { h->t->set_domain(ftkx::lattice(std::vector<long long>(st, st + h->nd), std::vector<long long>(sz, sz + h->nd))); }); }
int ftkx_tracker_set_array_domain(ftkx_tracker *h, const long long *st, const long long *sz)
{ return guarded(h, [&] { h->t->set_array_domain(ftkx::lattice(std::vector<long long>(st, st + h->nd), std::vector<long long>(sz, sz + h->nd))); }); }
int ftkx_tracker_set_sources(ftkx_tracker *h, int s, int v, int j, int sym)
{ return guarded(h, [&] { h->t->set_scalar_field_source(s); h->t->set_vector_field_source(v); h->t->set_jacobian_field_source(j); h->t->set_jacobian_symmetric(sym != 0); }); }
int ftkx_tracker_set_flags(ftkx_tracker *h, int robust, int use_tf, unsigned tf, int degrees, int exact_only, int tag_mode)
{
  return guarded(h, [&] {
    h->t->set_enable_robust_detection(robust != 0);
    if (use_tf) h->t->set_type_filter(tf);
    h->t->set_enable_computing_degrees(degrees != 0);
    h->t->set_exact_only(exact_only != 0);
    h->t->set_tag_mode(tag_mode);
  });
}
int ftkx_tracker_set_stream(ftkx_tracker *h, void *s) { return guarded(h, [&] { h->t->set_stream(s); }); }
int ftkx_tracker_set_deferred_collection(ftkx_tracker *h, int on) { return guarded(h, [&] { h->t->set_deferred_collection(on != 0, on); }); }
int ftkx_tracker_set_communicator(ftkx_tracker *h, void *comm, int rank, int nranks, int nt) { return guarded(h, [&] { h->t->set_communicator(comm, rank, nranks, nt); }); }
int ftkx_tracker_set_slab_transport(ftkx_tracker *h, const ftkx_slab_transport *tr, int rank, int nranks, int nt)
{ return guarded(h, [&] { if (!tr) throw ftkx::ftkx_error(FTKX_E_INVALID, "null transport"); h->t->set_slab_transport(*tr, rank, nranks, nt); }); }
int ftkx_tracker_set_slab_hub(ftkx_tracker *h, ftkx_slab_hub *hub, int rank, int nt) { return guarded(h, [&] { h->t->set_slab_hub(hub, rank, nt); }); }
int ftkx_tracker_set_enable_streaming_trajectories(ftkx_tracker *h, int on) { return guarded(h, [&] { h->t->set_enable_streaming_trajectories(on != 0); }); }
int ftkx_tracker_set_current_timestep(ftkx_tracker *h, int t)
{ return guarded(h, [&] { if (t < 0) throw ftkx::ftkx_error(FTKX_E_INVALID, "set_current_timestep: negative timestep"); h->t->set_current_timestep(t); }); }
int ftkx_tracker_set_coords_bounds(ftkx_tracker *h, const double *b) { return guarded(h, [&] { h->t->set_coords_bounds(std::vector<double>(b, b + 2 * h->nd)); }); }
int ftkx_tracker_set_coords_rectilinear(ftkx_tracker *h, const double *x, size_t nx, const double *y, size_t ny, const double *z, size_t nz)
{
  return guarded(h, [&] {
    std::vector<std::vector<double>> c;
    c.emplace_back(x, x + nx); c.emplace_back(y, y + ny);
    if (h->nd == 3) c.emplace_back(z, z + nz);
    h->t->set_coords_rectilinear(c);
  });
}
int ftkx_tracker_set_coords_explicit(ftkx_tracker *h, const double *c, int ncomp, size_t n0, size_t n1)
{ return guarded(h, [&] { if (!c || ncomp < 2) throw ftkx::ftkx_error(FTKX_E_INVALID, "set_coords_explicit: bad arguments"); h->t->set_coords_explicit(c, ncomp, n0, n1); }); }
int ftkx_tracker_initialize(ftkx_tracker *h) { return guarded(h, [&] { h->t->initialize(); }); }
int ftkx_tracker_push_scalar_field_snapshot(ftkx_tracker *h, const double *s, int dev) { return guarded(h, [&] { h->t->push_scalar_field_snapshot(s, dev != 0); }); }
int ftkx_tracker_push_vector_field_snapshot(ftkx_tracker *h, const double *v, int dev) { return guarded(h, [&] { h->t->push_vector_field_snapshot(v, dev != 0); }); }
int ftkx_tracker_push_field_data_snapshot(ftkx_tracker *h, const double *s, const double *v, const double *j, int dev)
{ return guarded(h, [&] { h->t->push_field_data_snapshot(s, v, j, dev != 0); }); }
int ftkx_tracker_advance_timestep(ftkx_tracker *h) { return guarded(h, [&] { h->t->advance_timestep(); }); }
int ftkx_tracker_update_timestep(ftkx_tracker *h) { return guarded(h, [&] { h->t->update_timestep(); }); }

int ftkx_tracker_num_critical_points(const ftkx_tracker *h, size_t *n)
{
  if (!n) return FTKX_E_INVALID;
  return guarded(const_cast<ftkx_tracker *>(h), [&] { *n = h->t->num_discrete_critical_points(); });
}

int ftkx_tracker_get_critical_points(const ftkx_tracker *h, ftkx_cp_t *out, int *ordinal, int *timestep, size_t cap)
{
  if (!h || !h->t || !out) return FTKX_E_INVALID;
  size_t i = 0;
  std::vector<ftkx::feature_point_t> pts;
  const int rc = guarded(const_cast<ftkx_tracker *>(h), [&] { pts = h->t->get_critical_points(); });
  if (rc) return rc;
  for (const ftkx::feature_point_t &cp : pts) {
    if (i >= cap) break;
    std::memset(&out[i], 0, sizeof(ftkx_cp_t));
    for (int k = 0; k < 3; k ++) { out[i].x[k] = cp.x[k]; out[i].scalar[k] = cp.scalar[k]; }
    out[i].t = cp.t; out[i].type = cp.type; out[i].tag = cp.tag;
    if (ordinal) ordinal[i] = cp.ordinal;
    if (timestep) timestep[i] = cp.timestep;
    i ++;
  }
  return FTKX_OK;
}

int ftkx_tracker_get_scaling(const ftkx_tracker *h, unsigned long long *factor, double *resolution)
{
  // (a getter waits for what is still out -- a several-device tracker's queues, a slab's pass: whatever that raises is the call's error)
  return guarded(const_cast<ftkx_tracker *>(h), [&] {
    if (factor) *factor = h->t->get_vector_field_scaling_factor();
    if (resolution) *resolution = h->t->get_vector_field_resolution();
  });
}

int ftkx_tracker_finalize(ftkx_tracker *h) { return guarded(h, [&] { h->t->finalize(); }); }

int ftkx_tracker_num_curves(const ftkx_tracker *h, size_t *n_curves, size_t *n_points)
{
  if (!h || !h->t || !n_curves || !n_points) return FTKX_E_INVALID;
  *n_curves = h->t->num_traced_curves();
  *n_points = h->t->get_traced_points().size();
  return FTKX_OK;
}

int ftkx_tracker_get_curves(const ftkx_tracker *h, long long *offsets, unsigned long long *tags, int *loop)
{
  if (!h || !h->t || !offsets || !tags || !loop) return FTKX_E_INVALID;
  const auto &pts = h->t->get_traced_points();
  const auto &off = h->t->get_traced_offsets();
  for (size_t k = 0; k < pts.size(); k ++) tags[k] = pts[k].tag;
  for (size_t i = 0; i < off.size(); i ++) offsets[i] = off[i];
  for (size_t i = 0; i + 1 < off.size(); i ++) loop[i] = h->t->get_traced_loop_flags()[i];
  return FTKX_OK;
}

int ftkx_tracker_post_process(ftkx_tracker *h) { return guarded(h, [&] { h->t->post_process(); }); }

int ftkx_tracker_get_curve_points(const ftkx_tracker *h, unsigned int *type, double *t, int *ids)
{
  if (!h || !h->t) return FTKX_E_INVALID;
  const auto &pts = h->t->get_traced_points();
  for (size_t k = 0; k < pts.size(); k ++) { if (type) type[k] = pts[k].type; if (t) t[k] = pts[k].t; }
  if (ids) for (size_t i = 0; i < h->t->num_traced_curves(); i ++) ids[i] = h->t->get_traced_ids()[i];
  return FTKX_OK;
}

int ftkx_tracker_write(const ftkx_tracker *h, const char *path, int format, int traced)
{
  if (!path) return FTKX_E_INVALID;
  return guarded(const_cast<ftkx_tracker *>(h), [&] {
    const std::string f(path);
    const ftkx::critical_point_tracker_regular &t = *h->t;
    if (format == FTKX_FORMAT_JSON) { if (traced) t.write_traced_critical_points_json(f); else t.write_critical_points_json(f); }
    else if (format == FTKX_FORMAT_TEXT) { if (traced) t.write_traced_critical_points_text(f); else t.write_critical_points_text(f); }
    else if (format == FTKX_FORMAT_BINARY) { if (traced) t.write_traced_critical_points_binary(f); else t.write_critical_points_binary(f); }
    else throw ftkx::ftkx_error(FTKX_E_INVALID, "ftkx_tracker_write: unknown format");
  });
}

int ftkx_tracker_read_critical_points(ftkx_tracker *h, const char *path, int format)
{
  if (!path) return FTKX_E_INVALID;
  return guarded(h, [&] {
    if (format == FTKX_FORMAT_JSON) h->t->read_critical_points_json(path);
    else if (format == FTKX_FORMAT_BINARY) h->t->read_critical_points_binary(path);
    else throw ftkx::ftkx_error(FTKX_E_UNSUPPORTED, "ftkx_tracker_read_critical_points: json or binary");
  });
}

int ftkx_tracker_get_stats(const ftkx_tracker *h, ftkx_stats *st)
{
  if (!st) return FTKX_E_INVALID;
  return guarded(const_cast<ftkx_tracker *>(h), [&] { *st = h->t->get_last_stats(); });
}

}  // extern "C"
