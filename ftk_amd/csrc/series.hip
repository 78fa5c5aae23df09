// ftkx_sweep_series: one pass over a series of resident slices, queued without the host in the loop.
//
// The reference's per-step sequence (critical_point_tracker_{2d,3d}_regular::update_timestep, 2d:263-433, 3d:150-308) starts on the
// host: update_vector_field_scaling_factor (critical_point_tracker.hh:850-864) folds the newest snapshot's resolution into a sticky
// running minimum and derives nbits from it; only then can a simplex be quantised.  The batched form of this library used to mirror
// that: mask kernel -> host waits for the reduction -> host forms the factors -> exact test -> host waits for the counters -> device
// sort -> download -> host waits again.  Here the running minimum and nbits are formed by a kernel (series_kernels.hip), the records are
// ordered without a sort and written by the record kernel straight into the pinned buffer the caller reads, and the host waits ONCE,
// for a sequence number stored behind the results.  Anything the device-driven form does not cover -- found up front on the host, or
// flagged by the kernels (a factor that hangs on the last bit of the host's log2, slices whose masks need the per-vertex overflow rule,
// buffers that were too small) -- is swept by the host-driven batch (ftkx_slices_prepare / ftkx_sweep_enqueue / ftkx_sweep_collect)
// inside the same call, with the same result.
#include "ctx.hpp"
#include "cp_device.hpp"   // classify3 on the HOST (fragile 3D records)

using namespace ftkxh;

namespace {

unsigned long long factor_of(double resolution)
{
  int nbits = (int)std::ceil(std::log2(1.0 / resolution));      // critical_point_tracker.hh:850-864
  nbits = std::max(8, std::min(nbits, 21));
  return 1ull << nbits;
}

// the host-driven batch: what the series pass falls back on, and the definition of what it must return
int series_by_host(ftkx_ctx *c, const int *ts, const int *scopes, int n, const std::vector<int> &slice_ts, double *running, unsigned long long *factors,
                   const ftkx_cp_t **out, size_t *n_out)
{
  c->sr_last_path = 0;
  const unsigned long long hint = std::max<unsigned long long>(factor_of(*running), 256ull);
  std::vector<double> below(slice_ts.size());
  int rc = ftkx_sweep_announce(c, ts, scopes, n);
  if (rc) return rc;
  if ((rc = ftkx_slices_prepare(c, slice_ts.data(), (int)slice_ts.size(), hint, below.data(), nullptr))) return rc;
  std::vector<unsigned long long> f((size_t)n);
  double run = *running;
  size_t j = 0;
  for (int i = 0; i < n; i ++) {
    while (j < slice_ts.size() && slice_ts[j] <= ts[i] + 1) { run = std::min(run, below[j]); j ++; }
    f[(size_t)i] = factor_of(run);
  }
  for (; j < slice_ts.size(); j ++) run = std::min(run, below[j]);
  if ((rc = ftkx_sweep_enqueue_many(c, ts, scopes, f.data(), n))) return rc;
  if ((rc = ftkx_sweep_collect(c, out, n_out))) return rc;
  *running = run;
  if (factors) for (int i = 0; i < n; i ++) factors[i] = f[(size_t)i];
  return FTKX_OK;
}

template <class T> int grow_device(ftkx_ctx *c, T **p, size_t *cap, size_t want)
{
  if (*cap >= want) return FTKX_OK;
  if (*p) { HIP_TRY(c, hipFree(*p)); *p = nullptr; *cap = 0; }
  HIP_TRY(c, hipMalloc((void **)p, want * sizeof(T)));
  *cap = want;
  return FTKX_OK;
}

size_t align256(size_t v) { return (v + 255) / 256 * 256; }

}  // namespace

extern "C" {

int ftkx_sweep_series(ftkx_ctx *c, const int *ts, const int *scopes, int n, double *running_resolution, unsigned long long *factors,
                      const ftkx_cp_t **out, size_t *n_out)
{
  if (!c || (n > 0 && (!ts || !scopes)) || !running_resolution) return fail(c, FTKX_E_INVALID, "null argument");
  if (out) *out = nullptr;
  if (n_out) *n_out = 0;
  if (!c->mesh_set) return fail(c, FTKX_E_INVALID, "sweep: call ftkx_set_mesh first");
  if (!c->pending.empty()) return fail(c, FTKX_E_INVALID, "ftkx_sweep_series: sweeps pending, collect first");
  if (!(*running_resolution > 0)) return fail(c, FTKX_E_INVALID, "ftkx_sweep_series: the running resolution must be positive (DBL_MAX: none yet)");
  if (n == 0) return FTKX_OK;
  c->ahead.clear(); c->announced.clear();
  HIP_TRY(c, hipSetDevice(c->device));
  const int nd = c->nd;

  // ---- the steps, and the slices they read in time order -----------------------------------------------------------------------------
  std::vector<int> slice_ts;
  for (int i = 0; i < n; i ++) {
    if (scopes[i] < FTKX_SCOPE_ORDINAL || scopes[i] > FTKX_SCOPE_BOTH) return fail(c, FTKX_E_INVALID, "ftkx_sweep_series: bad scope %d", scopes[i]);
    if (i > 0 && ts[i] <= ts[i - 1]) return fail(c, FTKX_E_INVALID, "ftkx_sweep_series: timesteps must be strictly ascending");
    if (slice_ts.empty() || slice_ts.back() != ts[i]) slice_ts.push_back(ts[i]);
    if (scopes[i] & FTKX_SCOPE_INTERVAL) slice_ts.push_back(ts[i] + 1);
  }
  std::vector<Slice *> sl(slice_ts.size());
  for (size_t j = 0; j < slice_ts.size(); j ++) {
    auto it = c->slices.find(slice_ts[j]);
    if (it == c->slices.end()) return fail(c, FTKX_E_NOSLICE, "ftkx_sweep_series: slice %d not resident", slice_ts[j]);
    sl[j] = &it->second;
  }
  const size_t k = sl.size();

  // ---- is the device-driven form applicable? ---------------------------------------------------------------------------------------
  Mesh m; fill_mesh(c, m);
  const bool two_level = ftkx::masks_have_summary(m);
  u64 cells = 1;
  for (int d = 0; d < nd; d ++) cells *= (u64)c->core_sz[d];
  bool ok = !c->opt.exact_only && (nd == 2 || c->opt.robust) && c->dense_collects == 0 && !c->opt.use_type_filter && cells > 0 && n <= ftkx::kSeriesMaxSlices && k <= (size_t)ftkx::kSeriesMaxSlices;
  if (const char *e = getenv("FTKX_SERIES")) ok = ok && atoi(e) != 0;
  // the order of (step, corner, type) must be the order of the tags
  if (ok && c->opt.tag_mode == FTKX_TAG_WORK_INDEX) ok = n == 1 && scopes[0] != FTKX_SCOPE_BOTH;
  if (ok && c->opt.tag_mode == FTKX_TAG_REFERENCE) {         // int32 products: equal to the 64-bit formula only while nothing wraps
    long double bound = nd == 2 ? 12.0L : 60.0L;
    for (int d = 0; d < nd; d ++) bound *= (long double)c->dom_sz[d];
    ok = bound * (long double)(ts[n - 1] + 2) < 2147483648.0L;
  }
  if (ok && (long double)n * (long double)cells * 64.0L >= 4611686018427387904.0L) ok = false;   // the order key must fit
  for (size_t j = 0; ok && j < k; j ++) ok = !sl[j]->sparse;
  for (int i = 0; ok && i < n; i ++) {                       // the same consistency rule as ftkx_sweep_enqueue
    if (!(scopes[i] & FTKX_SCOPE_INTERVAL)) continue;
    const Slice &a = c->slices[ts[i]], &b = c->slices[ts[i] + 1];
    if ((a.J == nullptr) != (b.J == nullptr) || (a.S == nullptr) != (b.S == nullptr)) ok = false;
  }
  if (ok && (c->opt.coords_mode == 2 || c->opt.coords_mode == 3)) ok = false;     // (their bounds checks live in ftkx_sweep_enqueue)
  const unsigned long long hint = std::max<unsigned long long>(factor_of(*running_resolution), 256ull);
  // slices whose masks and reduction stand from an earlier call (a streaming tracker: slice t of this step was slice t + 1 of the last)
  std::vector<int> red_index(k, -1);
  size_t ntodo = 0;
  for (size_t j = 0; ok && j < k; j ++) {
    const Slice &s = *sl[j];
    const bool ready = s.M && (!two_level || (s.U && s.u_rows == m.u_rows)) && s.mask_factor != 0 && s.mask_factor <= hint && !s.mask_big && (s.have_fused || s.have_res);
    if (!ready) red_index[j] = (int)ntodo ++;
  }
  if (!ok) return series_by_host(c, ts, scopes, n, slice_ts, running_resolution, factors, out, n_out);

  // ---- buffers (persistent; they only ever grow) -----------------------------------------------------------------------------------
  int rc;
  if ((rc = ensure_hit_buffer(c, std::max<u64>(c->capacity, 1u << 16)))) return rc;
  if ((rc = ensure_fragile(c, std::max<u64>(c->fragile_capacity, 1u << 12)))) return rc;
  if ((rc = ensure_list(c, std::max<u64>(c->list_capacity, 1u << 20))) || (rc = ensure_refine(c, std::max<u64>(c->refine_capacity, 1u << 20)))) return rc;
  if ((rc = ensure_host_buffer(c, (size_t)c->capacity))) return rc;
  for (size_t j = 0; j < k; j ++) if (red_index[j] >= 0 && (rc = ensure_mask_arrays(c, *sl[j], two_level))) return rc;
  if (c->red_cap < std::max<size_t>(ntodo, 1)) {
    if (c->d_red) { (void)hipFree(c->d_red); c->d_red = nullptr; c->red_cap = 0; }
    HIP_TRY(c, hipMalloc((void **)&c->d_red, std::max<size_t>(ntodo, 1) * 128 * sizeof(u64)));
    c->red_cap = std::max<size_t>(ntodo, 1);
  }
  // order key -> bucket: at most 2^16 buckets over the keys this pass can produce
  const u64 max_key = (u64)n * cells * 64ull;
  int key_bits = 1;
  while (key_bits < 63 && (1ull << key_bits) < max_key) key_bits ++;
  const int shift = std::max(0, key_bits - 16);
  const size_t nbins = (size_t)((max_key - 1) >> shift) + 1;
  if (c->sr_bins_cap < nbins + 1) {
    const size_t cap = std::max<size_t>(nbins + 1, (1u << 16) + 1);
    for (void *p : {(void *)c->sr_hist, (void *)c->sr_boff}) if (p) (void)hipFree(p);
    c->sr_hist = nullptr; c->sr_boff = nullptr; c->sr_bins_cap = 0;
    HIP_TRY(c, hipMalloc((void **)&c->sr_hist, cap * sizeof(unsigned)));
    HIP_TRY(c, hipMalloc((void **)&c->sr_boff, cap * sizeof(unsigned)));
    c->sr_bins_cap = cap;
  }
  if ((rc = grow_device(c, &c->sr_bucketed, &c->sr_bucketed_cap, (size_t)c->capacity))) return rc;
  if ((rc = grow_device(c, &c->sr_sorted, &c->sr_sorted_cap, (size_t)c->capacity))) return rc;
  const size_t nwords = (size_t)ftkx::SR_HEAD + (size_t)n + 2 * k;
  if ((rc = grow_device(c, &c->sr_results, &c->sr_results_cap, std::max<size_t>(nwords, 1024)))) return rc;
  const size_t h_words = nwords + (size_t)c->fragile_capacity * 10;
  if (c->sr_h_results_cap < h_words) {
    if (c->sr_h_results) { HIP_TRY(c, hipStreamSynchronize(c->stream)); (void)hipHostFree(c->sr_h_results); c->sr_h_results = nullptr; c->sr_h_results_cap = 0; }
    const size_t cap = h_words + h_words / 4 + 1024;
    HIP_TRY(c, hipHostMalloc((void **)&c->sr_h_results, (cap + 8) * sizeof(u64), hipHostMallocCoherent));
    c->sr_h_results_cap = cap;
    *reinterpret_cast<volatile unsigned *>(c->sr_h_results + cap) = 0u;
    c->sr_seq = 0;
  }
  fill_mesh(c, m);                                           // (the buffers may have moved)
  m.hist = c->sr_hist; m.hist_shift = shift; m.core_cells = cells;

  // ---- descriptors: mask jobs | steps | slice table | step table, one pinned block fetched by a kernel -------------------------------
  const size_t off_jobs = 0, off_steps = align256(ntodo * sizeof(MaskJob)), off_slices = off_steps + align256((size_t)n * sizeof(Fields)),
               off_sinfo = off_slices + align256(k * sizeof(ftkx::SeriesSlice)), total = off_sinfo + align256((size_t)n * sizeof(ftkx::SeriesStep));
  if ((rc = ensure_desc(c, total))) return rc;
  {
    MaskJob *jobs = (MaskJob *)((char *)c->h_desc + off_jobs);
    Fields *steps = (Fields *)((char *)c->h_desc + off_steps);
    ftkx::SeriesSlice *ss = (ftkx::SeriesSlice *)((char *)c->h_desc + off_slices);
    ftkx::SeriesStep *si = (ftkx::SeriesStep *)((char *)c->h_desc + off_sinfo);
    const double cap = 1.0 / (double)hint;
    for (size_t j = 0; j < k; j ++) {
      const Slice &s = *sl[j];
      ss[j].t = slice_ts[j]; ss[j].red_index = red_index[j];
      ss[j].known_res = DBL_MAX; ss[j].known_max = 0.0;
      if (s.have_res) { ss[j].known_res = s.res < cap ? s.res : DBL_MAX; ss[j].known_max = s.maxabs; }
      else if (red_index[j] < 0) { ss[j].known_res = s.res_below; ss[j].known_max = s.maxabs; }
      if (red_index[j] >= 0)
        jobs[red_index[j]] = MaskJob{s.S, s.V, s.M, two_level ? s.U : nullptr, c->d_red + (size_t)red_index[j] * 128, cap, HUGE_VAL};   // rule off: validated by the factor kernel
    }
    size_t last = 0;
    for (int i = 0; i < n; i ++) {
      const size_t j0 = (size_t)(std::lower_bound(slice_ts.begin(), slice_ts.end(), ts[i]) - slice_ts.begin());
      const bool interval = (scopes[i] & FTKX_SCOPE_INTERVAL) != 0;
      const Slice &s0 = *sl[j0];
      Fields f;
      memset(&f, 0, sizeof(f));
      f.S[0] = s0.S; f.V[0] = s0.V; f.J[0] = s0.J; f.M[0] = s0.M; f.U[0] = two_level ? s0.U : nullptr;
      if (interval) { const Slice &s1 = *sl[j0 + 1]; f.S[1] = s1.S; f.V[1] = s1.V; f.J[1] = s1.J; f.M[1] = s1.M; f.U[1] = two_level ? s1.U : nullptr; }
      f.factor = 0.0; f.t = ts[i]; f.scope_mask = scopes[i];
      steps[i] = f;
      while (last + 1 < k && slice_ts[last + 1] <= ts[i] + 1) last ++;
      si[i].slice0 = (int)j0; si[i].slice1 = interval ? (int)j0 + 1 : -1; si[i].last = (int)last; si[i].pad = 0;
    }
  }
  const MaskJob *d_jobs = (const MaskJob *)((char *)c->d_desc + off_jobs);
  Fields *d_steps = (Fields *)((char *)c->d_desc + off_steps);
  const ftkx::SeriesSlice *d_slices = (const ftkx::SeriesSlice *)((char *)c->d_desc + off_slices);
  const ftkx::SeriesStep *d_sinfo = (const ftkx::SeriesStep *)((char *)c->d_desc + off_sinfo);

  // ---- the whole pass, queued ------------------------------------------------------------------------------------------------------
  unsigned *flag = reinterpret_cast<unsigned *>(c->sr_h_results + c->sr_h_results_cap);
  const unsigned seq = ++ c->sr_seq;
  // (marks set before the masks exist are taken back on every error exit, like ftkx_slices_prepare does)
  struct MarkGuard { std::vector<Slice *> *v; std::vector<int> *todo; bool armed; ~MarkGuard() { if (armed) for (size_t j = 0; j < v->size(); j ++) if ((*todo)[j] >= 0) { (*v)[j]->mask_factor = 0; (*v)[j]->have_fused = false; } } } marks{&sl, &red_index, true};
  for (size_t j = 0; j < k; j ++) if (red_index[j] >= 0) { sl[j]->mask_factor = 0; sl[j]->have_fused = false; }
  ftkx::launch_series_begin(c->d_counters, c->d_red, ntodo * 64, c->sr_hist, nbins + 1, c->sr_results, nwords, c->stream);
  launch_fetch_desc(c->h_desc, c->d_desc, total, c->stream);
  if (ntodo) { ev_begin(c, K_MASK); ftkx::launch_masks(m, d_jobs, (int)ntodo, c->stream); ev_end(c); }
  ev_begin(c, K_CULL);
  if (two_level) ftkx::launch_cull_coarse(m, d_steps, n, c->d_refine, c->refine_capacity, c->stream);
  else ftkx::launch_cull(m, d_steps, n, c->d_list, c->list_capacity, c->stream);
  ftkx::launch_series_factors(d_steps, n, d_slices, (int)k, d_sinfo, c->d_red, *running_resolution, (double)(nd == 3 ? ftkx::kSafeM3 : ftkx::kSafeM2),
                              c->sr_results, c->stream);
  ev_end(c);
  ev_begin(c, K_EXACT);
  // sparse data: one workgroup does the rest of the pass (and the kernels below leave at once)
  static const bool small_on = !(getenv("FTKX_SERIES_SMALL") && atoi(getenv("FTKX_SERIES_SMALL")) == 0);
  // (a pass that has just found far more survivors than the fused kernel takes does not launch it for a while: finding nothing to do
  // costs its 256 workgroups of 256-VGPR wavefronts ~15 us)
  const bool small_now = small_on && c->sr_skip_small == 0;
  if (c->sr_skip_small > 0) c->sr_skip_small --;
  if (small_now) ftkx::launch_series_small(m, two_level ? ftkx::coarse_view(m) : m, d_steps, two_level, c->d_refine, c->d_list, c->h_hits, c->sr_results, nwords,
                                          c->sr_h_results, flag, seq, reinterpret_cast<unsigned *>(c->d_counters + ftkx::CNT_SMALL_DONE), c->stream);
  if (two_level) ftkx::launch_refine(m, d_steps, c->d_refine, c->refine_capacity, c->d_list, c->list_capacity, c->stream);
  ftkx::launch_exact(m, d_steps, 0, c->d_list, c->list_capacity, c->stream);
  ftkx::launch_bucket_scan(c->sr_hist, c->sr_boff, (unsigned)nbins, c->d_counters, c->stream);
  ftkx::launch_bucket_scatter(m, c->sr_boff, c->sr_bucketed, c->stream);
  ftkx::launch_bucket_rank(m, c->sr_bucketed, c->sr_boff, c->sr_sorted, c->sr_results, c->stream);
  ftkx::launch_series_records(m, d_steps, c->sr_sorted, c->h_hits, c->stream);
  ev_end(c);
  ftkx::launch_series_finish(m, c->sr_results, nwords, c->list_capacity, c->refine_capacity, c->sr_h_results, flag, seq, c->stream);
  HIP_TRY(c, hipGetLastError());
  if (const char *why = ftkx::wait_flag(flag, seq, c->stream)) return fail(c, FTKX_E_DEVICE, "ftkx_sweep_series: %s", why);
  ev_harvest(c, false);

  // ---- what came back ----------------------------------------------------------------------------------------------------------------
  const u64 *R = c->sr_h_results;
  const unsigned long long status = R[ftkx::SR_STATUS];
  c->sr_last_status = status;
  // the masks and reductions of this pass stand whichever way the records are made: the slices are marked like ftkx_slices_prepare marks them
  marks.armed = false;
  for (size_t j = 0; j < k; j ++) {
    if (red_index[j] < 0) continue;
    Slice &s = *sl[j];
    double r, x;
    memcpy(&r, &R[ftkx::SR_HEAD + n + j], 8); memcpy(&x, &R[ftkx::SR_HEAD + n + k + j], 8);
    if (std::isinf(x)) { s.mask_factor = 0; s.have_fused = false; continue; }     // the fused max is not the max FINITE |v|: the host-driven path reduces it exactly
    s.res_below = r; s.fused_factor = hint; s.have_fused = true;
    if (!s.have_res) s.maxabs = x;
    s.mask_factor = overflow_free(nd, s.maxabs, hint) ? hint : 0;
    s.mask_big = false; s.u_rows = m.u_rows;
  }
  const unsigned long long redo = ftkx::SERIES_AMBIGUOUS | ftkx::SERIES_MASKS_INVALID | ftkx::SERIES_INF | ftkx::SERIES_OVERFLOW;
  if (status & redo) {
    if (status & ftkx::SERIES_OVERFLOW) {                    // grow what was too small (the host-driven batch would find out the same way, one replay later)
      const u64 *cnt = R + ftkx::SR_COUNTERS;
      const u64 hits = cnt[ftkx::CNT_PASS], listed = cnt[ftkx::CNT_LIST_PEAK], refined = cnt[ftkx::CNT_REFINE_PEAK], fragile = cnt[ftkx::CNT_FRAGILE];
      if (hits > c->capacity && (rc = ensure_hit_buffer(c, hits + hits / 8 + 1024))) return rc;
      if (listed > c->list_capacity && (rc = ensure_list(c, listed + listed / 8 + 1024))) return rc;
      if (refined > c->refine_capacity && (rc = ensure_refine(c, refined + refined / 8 + 1024))) return rc;
      if (fragile > c->fragile_capacity && (rc = ensure_fragile(c, fragile + fragile / 8 + 1024))) return rc;
    }
    return series_by_host(c, ts, scopes, n, slice_ts, running_resolution, factors, out, n_out);
  }
  c->sr_last_path = (status & ftkx::SERIES_EARLY) ? 2 : 1;
  const u64 *cnt = R + ftkx::SR_COUNTERS;
  if (!(status & ftkx::SERIES_EARLY) && (two_level ? cnt[ftkx::CNT_REFINE_PEAK] : cnt[ftkx::CNT_LIST_PEAK]) > 4 * 2048ull) c->sr_skip_small = 16;
  const size_t nrec = (size_t)R[ftkx::SR_NHITS];
  memset(&c->stats, 0, sizeof(c->stats));
  {
    const u64 n_ord = nd == 2 ? 2 : 6, n_int = nd == 2 ? 10 : 54;
    for (int i = 0; i < n; i ++) { c->stats.cells += cells; c->stats.work_items += cells * (((scopes[i] & 1) ? n_ord : 0) + ((scopes[i] & 2) ? n_int : 0)); }
  }
  c->stats.cull_enabled = 1;
  c->stats.hits = nrec;
  c->stats.cells_survived = cnt[ftkx::CNT_CELLS_SURVIVED];
  c->stats.simplices_tested = cnt[ftkx::CNT_SIMPLICES_TESTED];
  // 3D records whose class hangs on the last bits of pow / acos / cos: classified with the libm the reference runs on (collect.hip does
  // the same), patched in place -- the records are in host memory already
  const size_t nf = (size_t)R[ftkx::SR_NFRAGILE];
  for (size_t i = 0; i < nf; i ++) {
    const u64 *e = R + nwords + i * 10;
    double A[3][3];
    memcpy(A, e + 1, sizeof(A));
    if (e[0] < nrec) c->h_hits[e[0]].type = (unsigned)ftkx::classify3(A, c->opt.jacobian_symmetric != 0);
  }
  c->stats.reclassified = nf;
  if (status & ftkx::SERIES_UNORDERED) {                     // (the fused tail with more records than its last workgroup ranks: rare) sort an index, move once
    std::vector<std::pair<unsigned long long, size_t>> order(nrec);
    for (size_t i = 0; i < nrec; i ++) order[i] = {c->h_hits[i].tag, i};
    std::sort(order.begin(), order.end());
    std::vector<ftkx_cp_t> tmp(nrec);
    for (size_t i = 0; i < nrec; i ++) tmp[i] = c->h_hits[order[i].second];
    if (nrec) memcpy(c->h_hits, tmp.data(), nrec * sizeof(ftkx_cp_t));
  }
  if (status & ftkx::SERIES_FIX_ORDER) {
    // A bucket too full to rank on the device: its records sit in their own run of the output, unordered among themselves; everything
    // before the run is smaller, everything behind it larger.  Find each such run from an inversion, widen it until both ends are in
    // order with their neighbours, sort it.
    auto less = [](const ftkx_cp_t &p, const ftkx_cp_t &q) { return p.tag < q.tag; };
    ftkx_cp_t *h = c->h_hits;
    for (size_t a = 0; a + 1 < nrec; a ++) {
      if (h[a].tag <= h[a + 1].tag) continue;
      size_t lo = a, hi = a + 2;
      unsigned long long mn = std::min(h[a].tag, h[a + 1].tag), mx = std::max(h[a].tag, h[a + 1].tag);
      for (bool grown = true; grown;) {
        grown = false;
        while (lo > 0 && h[lo - 1].tag > mn) { lo --; mn = std::min(mn, h[lo].tag); mx = std::max(mx, h[lo].tag); grown = true; }
        while (hi < nrec && h[hi].tag < mx) { mn = std::min(mn, h[hi].tag); mx = std::max(mx, h[hi].tag); hi ++; grown = true; }
      }
      std::sort(h + lo, h + hi, less);
      a = hi - 2;                                            // (the loop's increment moves on to the run's last element)
    }
  }
  double run;
  memcpy(&run, &R[ftkx::SR_RUNNING], 8);
  *running_resolution = run;
  if (factors) for (int i = 0; i < n; i ++) factors[i] = R[ftkx::SR_HEAD + i];
  if (out) *out = c->h_hits;
  if (n_out) *n_out = nrec;
  return FTKX_OK;
}

int ftkx_series_last_path(const ftkx_ctx *c, unsigned long long *status)
{
  if (!c) return -1;
  if (status) *status = c->sr_last_status;
  return c->sr_last_path;
}

}  // extern "C"
