// The batched sweep: ftkx_sweep_enqueue / ftkx_sweep_collect (masks where missing -> cull -> exact test -> records -> device sort ->
// download) and the kernel timing that goes with it.  Reference counterpart: the per-call host wrapper extract_cp2dt<scope> /
// extract_cp3dt<scope> (src/filters/critical_point_tracer_2d_regular.cu:168-272, ..._3d_regular.cu:144-250).
#include "ctx.hpp"
#include "cp_device.hpp"   // classify3 on the HOST (fragile 3D records, see there)

using namespace ftkxh;

namespace ftkxh {

// types re-computed on the host written back into the hit buffer: pairs (slot, type)
__global__ void patch_types_kernel(ftkx_cp_t *hits, const u64 *__restrict__ pairs, size_t n)
{
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) hits[pairs[2 * i]].type = (unsigned)pairs[2 * i + 1];
}

__global__ void sort_keys_kernel(const ftkx_cp_t *__restrict__ hits, size_t n, u64 *__restrict__ keys, unsigned *__restrict__ idx)
{
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) { keys[i] = hits[i].tag; idx[i] = (unsigned)i; }
}

__global__ void sort_gather_kernel(const ftkx_cp_t *__restrict__ hits, const unsigned *__restrict__ idx, size_t n, ftkx_cp_t *__restrict__ out)
{
  // 72-byte records moved as nine 8-byte words by nine consecutive lanes: coalesced stores
  const size_t w = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (w < n * 9) {
    const size_t r = w / 9, k = w % 9;
    reinterpret_cast<u64 *>(out)[w] = reinterpret_cast<const u64 *>(hits)[(size_t)idx[r] * 9 + k];
  }
}

// ---- the records of a batch in tag order: a radix sort of (tag, index) pairs, least significant digit first ---------------------------
// 4-bit digits, three launches per digit: per-tile digit counts, ONE exclusive scan over (digit, tile), a stable scatter.  A tile is
// 256 threads x 8 consecutive keys each; inside it a key's place among its digit's keys = the keys of that digit in the threads before
// (a prefix over per-thread counts in LDS) + those before it in its own thread -- stable without a single atomic.  Only the digits a tag
// of this batch can have take part (key_bits).  This is the host-driven batch's ordering step (the device-driven pass orders by buckets,
// series_kernels.hip); rounds 1-3 called hipcub::DeviceRadixSort here.
constexpr int kRsThreads = 256, kRsPer = 8, kRsTile = kRsThreads * kRsPer;

__global__ __launch_bounds__(kRsThreads) void rs_count_kernel(const u64 *__restrict__ keys, size_t n, int shift, unsigned *__restrict__ counts, unsigned ntiles)
{
  __shared__ unsigned s_cnt[16];
  if (threadIdx.x < 16) s_cnt[threadIdx.x] = 0u;
  __syncthreads();
  const size_t base = (size_t)blockIdx.x * kRsTile;
  for (int e = 0; e < kRsPer; e ++) {
    const size_t i = base + (size_t)e * kRsThreads + threadIdx.x;
    if (i < n) atomicAdd(&s_cnt[(unsigned)(keys[i] >> shift) & 15u], 1u);
  }
  __syncthreads();
  if (threadIdx.x < 16) counts[(size_t)threadIdx.x * ntiles + blockIdx.x] = s_cnt[threadIdx.x];
}

// exclusive scan in place, one workgroup: a thread's contiguous share serially, the shares' totals by wavefront and through LDS
__global__ __launch_bounds__(1024) void rs_scan_kernel(unsigned *__restrict__ v, unsigned n)
{
  __shared__ unsigned s_wave[16];
  const unsigned tid = threadIdx.x, lane = tid & 63u, wv = tid >> 6;
  const unsigned per = (n + 1023u) / 1024u, lo = tid * per, hi = lo + per < n ? lo + per : n;
  unsigned sum = 0;
  for (unsigned i = lo; i < hi; i ++) sum += v[i];
  unsigned incl = sum;
  for (int o = 1; o < 64; o <<= 1) { const unsigned up = __shfl_up(incl, o); if ((int)lane >= o) incl += up; }
  if (lane == 63u) s_wave[wv] = incl;
  __syncthreads();
  unsigned run = incl - sum;
  for (unsigned q = 0; q < wv; q ++) run += s_wave[q];
  for (unsigned i = lo; i < hi; i ++) { const unsigned c = v[i]; v[i] = run; run += c; }
}

__global__ __launch_bounds__(kRsThreads) void rs_scatter_kernel(const u64 *__restrict__ keys, const unsigned *__restrict__ idx, size_t n, int shift,
                                                                const unsigned *__restrict__ offs, unsigned ntiles, u64 *__restrict__ keys_out, unsigned *__restrict__ idx_out)
{
  __shared__ unsigned short s_cnt[16][kRsThreads];          // column t: thread t's keys per digit, then its first place per digit inside the tile
  __shared__ unsigned s_base[16];
  const unsigned tid = threadIdx.x;
  const size_t first = (size_t)blockIdx.x * kRsTile + (size_t)tid * kRsPer;      // this thread's eight consecutive keys
  u64 k[kRsPer];
  unsigned v[kRsPer];
  for (int d = 0; d < 16; d ++) s_cnt[d][tid] = 0;
  for (int e = 0; e < kRsPer; e ++) {
    const size_t i = first + (size_t)e;
    k[e] = 0ull; v[e] = 0u;
    if (i < n) { k[e] = keys[i]; v[e] = idx[i]; s_cnt[(unsigned)(k[e] >> shift) & 15u][tid] ++; }
  }
  if (tid < 16) s_base[tid] = offs[(size_t)tid * ntiles + blockIdx.x];
  __syncthreads();
  {
    // exclusive prefix over the 256 threads, per digit: thread (d, seg) = (tid / 16, tid % 16) walks sixteen columns, the sixteen
    // segments of a digit are sixteen consecutive lanes
    const unsigned d = tid >> 4, seg = tid & 15u;
    unsigned sum = 0;
    for (unsigned j = 0; j < 16; j ++) sum += s_cnt[d][seg * 16u + j];
    unsigned incl = sum;
    for (int o = 1; o < 16; o <<= 1) { const unsigned up = __shfl_up(incl, o, 16); if ((int)seg >= o) incl += up; }
    unsigned run = incl - sum;
    for (unsigned j = 0; j < 16; j ++) { const unsigned c = s_cnt[d][seg * 16u + j]; s_cnt[d][seg * 16u + j] = (unsigned short)run; run += c; }
  }
  __syncthreads();
  for (int e = 0; e < kRsPer; e ++) {
    if (first + (size_t)e < n) {
      const unsigned d = (unsigned)(k[e] >> shift) & 15u;
      const size_t at = (size_t)s_base[d] + (size_t)(s_cnt[d][tid] ++);
      keys_out[at] = k[e]; idx_out[at] = v[e];
    }
  }
}

// the reference keeps hits in a std::map ordered by element; device append order is arbitrary
int sort_hits_on_device(ftkx_ctx *c, size_t n, int key_bits)
{
  if (n == 0) return FTKX_OK;                                // (no tiles: nothing to launch)
  if (n >= (size_t)1 << 31) return fail(c, FTKX_E_UNSUPPORTED, "sort_hits_on_device: %zu records (indices are 32-bit)", n);
  if (c->sort_cap < n) {
    for (void *p : {(void *)c->d_sorted, (void *)c->d_keys, (void *)c->d_idx, c->d_sort_tmp}) if (p) (void)hipFree(p);
    c->d_sorted = nullptr; c->d_keys = nullptr; c->d_idx = nullptr; c->d_sort_tmp = nullptr; c->sort_cap = 0;
    const size_t cap = n + n / 4 + 1024;
    HIP_TRY(c, hipMalloc((void **)&c->d_sorted, cap * sizeof(ftkx_cp_t)));
    HIP_TRY(c, hipMalloc((void **)&c->d_keys, 2 * cap * sizeof(u64)));
    HIP_TRY(c, hipMalloc((void **)&c->d_idx, 2 * cap * sizeof(unsigned)));
    c->sort_tmp_bytes = 16 * ((cap + kRsTile - 1) / kRsTile) * sizeof(unsigned);      // digit counts per tile
    HIP_TRY(c, hipMalloc(&c->d_sort_tmp, c->sort_tmp_bytes));
    c->sort_cap = cap;
  }
  const size_t cap = c->sort_cap;
  hipLaunchKernelGGL(sort_keys_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, c->stream, c->d_hits, n, c->d_keys, c->d_idx);
  const unsigned ntiles = (unsigned)((n + kRsTile - 1) / kRsTile);
  unsigned *counts = (unsigned *)c->d_sort_tmp;
  int from = 0;                                              // which half of the ping-pong buffers holds the pairs
  for (int shift = 0; shift < key_bits; shift += 4) {
    const u64 *kin = c->d_keys + (from ? cap : 0);
    const unsigned *iin = c->d_idx + (from ? cap : 0);
    u64 *kout = c->d_keys + (from ? 0 : cap);
    unsigned *iout = c->d_idx + (from ? 0 : cap);
    hipLaunchKernelGGL(rs_count_kernel, dim3(ntiles), dim3(kRsThreads), 0, c->stream, kin, n, shift, counts, ntiles);
    hipLaunchKernelGGL(rs_scan_kernel, dim3(1), dim3(1024), 0, c->stream, counts, 16u * ntiles);
    hipLaunchKernelGGL(rs_scatter_kernel, dim3(ntiles), dim3(kRsThreads), 0, c->stream, kin, iin, n, shift, counts, ntiles, kout, iout);
    from ^= 1;
  }
  hipLaunchKernelGGL(sort_gather_kernel, dim3((unsigned)((n * 9 + 255) / 256)), dim3(256), 0, c->stream, c->d_hits, c->d_idx + (from ? cap : 0), n, c->d_sorted);
  HIP_TRY(c, hipGetLastError());
  return FTKX_OK;
}

// (events are recycled: creating and destroying a pair per kernel cost a hit-dense 2D pass several per cent)
hipEvent_t ev_take(ftkx_ctx *c)
{
  if (!c->event_pool.empty()) { hipEvent_t e = c->event_pool.back(); c->event_pool.pop_back(); return e; }
  hipEvent_t e = nullptr;
  return hipEventCreate(&e) == hipSuccess ? e : nullptr;
}
void ev_give(ftkx_ctx *c, hipEvent_t e) { if (e) c->event_pool.push_back(e); }
void ev_begin(ftkx_ctx *c, int kind)
{
  c->ev_open = false;
  if (!c->profiling || (c->profiling == 2 && kind != K_MASK)) return;     // level 2: the dominant kernel only (an event pair costs the stream ~10 us of idle time)
  hipEvent_t a = ev_take(c), b = ev_take(c);
  if (!a || !b) { ev_give(c, a); ev_give(c, b); return; }
  (void)hipEventRecord(a, c->stream);
  c->events.push_back({kind, {a, b}});
  c->ev_open = true;
}
void ev_end(ftkx_ctx *c)
{
  if (!c->profiling || !c->ev_open || c->events.empty()) return;
  c->ev_open = false;
  (void)hipEventRecord(c->events.back().second.second, c->stream);
}
void ev_harvest(ftkx_ctx *c, bool all)
{
  std::vector<std::pair<int, std::pair<hipEvent_t, hipEvent_t>>> later;
  for (auto &e : c->events) {
    if (!all && hipEventQuery(e.second.second) != hipSuccess) { later.push_back(e); continue; }
    float ms = 0;
    if (hipEventElapsedTime(&ms, e.second.first, e.second.second) == hipSuccess) { c->k_ms[e.first] += ms; c->k_launches[e.first] ++; }
    ev_give(c, e.second.first); ev_give(c, e.second.second);
  }
  c->events.swap(later);
}

// launches everything the pending requests need; counters must have been zeroed.
// Fast-path requests are grouped into sub-batches (one mask / cull / exact launch each); a new sub-batch starts whenever a
// slice's masks would be needed under a second quantisation factor (the factor is a running minimum, so it changes a few
// times at the start of a series and then stays put).
int run_batch(ftkx_ctx *c, const double *sparse_field, bool cull_done)
{
  const bool cull_only = sparse_field != nullptr;   // ftkx_sweep_cull: stop after the cull and list the survivors that read `sparse_field`
  Mesh m;
  fill_mesh(c, m);
  const int nd = c->nd;
  struct Sub { std::vector<MaskJob> jobs; std::vector<Fields> steps; };
  std::vector<Sub> subs(1);
  const bool two_level = ftkx::masks_have_summary(m);
  // the tile requests: one launch per run of consecutive requests with the same cull setting -- a workgroup walks the run's steps in time
  // order with its tile staged in LDS (tile_kernels.hip) -- under the smallest form any of them asks for (every form is correct everywhere)
  std::vector<TileParams> tiles;
  std::vector<Fields> tile_fields;
  const bool tile_walk = !(getenv("FTKX_TILE_WALK") && atoi(getenv("FTKX_TILE_WALK")) == 0);      // (test hook: 0 = a launch per step)
  for (const Request &r : c->pending) {
    Slice &s0 = c->slices[r.t];
    Slice *s1 = (r.scope & FTKX_SCOPE_INTERVAL) ? &c->slices[r.t + 1] : nullptr;
    Fields f;
    memset(&f, 0, sizeof(f));
    f.S[0] = s0.S; f.V[0] = s0.V; f.J[0] = s0.J;
    if (s1) { f.S[1] = s1->S; f.V[1] = s1->V; f.J[1] = s1->J; }
    f.factor = (double)r.factor; f.t = r.t; f.scope_mask = r.scope;
    if (r.mode == MODE_FAST) {
      for (Slice *s : {&s0, s1}) {
        if (!s) continue;
        if (masks_valid(c, *s, r.factor, two_level, m.u_rows)) continue;     // e.g. built by ftkx_slices_prepare, or by an earlier step
        if (s->sparse) return fail(c, FTKX_E_NOSLICE, "sweep: the masks of halo slice (masks only) do not serve factor %llu: send the slice itself", r.factor);
        int rc = ensure_mask_arrays(c, *s, two_level);
        if (rc) return rc;
        // masks of this slice already (re)built or used in the current sub-batch under another factor -> close it
        bool touched = false;
        for (const MaskJob &j : subs.back().jobs) touched = touched || j.M == s->M;
        for (const Fields &g : subs.back().steps) touched = touched || g.M[0] == s->M || g.M[1] == s->M;
        if (touched) subs.emplace_back();
        bool rule_on;
        const double big = job_big(c, *s, r.factor, &rule_on);
        subs.back().jobs.push_back(with_lean_thresholds(MaskJob{s->S, s->V, s->M, two_level ? s->U : nullptr, nullptr, 1.0 / (double)r.factor, big}, m));
        s->mask_factor = r.factor; s->mask_big = rule_on; s->u_rows = m.u_rows;
      }
      f.M[0] = s0.M; f.M[1] = s1 ? s1->M : nullptr;
      f.U[0] = two_level ? s0.U : nullptr; f.U[1] = (two_level && s1) ? s1->U : nullptr;
      subs.back().steps.push_back(f);
    } else {
      TileParams p;
      p.m = m; p.steps = nullptr; p.nsteps = 1; p.repeat = c->tile_repeat; p.cull = r.mode == MODE_TILE_CULL ? 1 : 0; p.step = (int)tile_fields.size();
      tile_fields.push_back(f);
      { const char *e = getenv("FTKX_TILE_FAN"); p.fan = e ? atoi(e) : 2; }
      {
        // largest |quantised component| the request can meet, where the slices' maxima are known (the kernel checks every tile anyway)
        double bound = -1.0;
        if (s0.max_known() && (!s1 || s1->max_known())) bound = std::max(s0.maxabs, s1 ? s1->maxabs : 0.0) * (double)r.factor;
        const double fp_bound = nd == 3 ? 524287.0 : 33554431.0;     // below 2^19 (3D: error-bounded fp64) / 2^25 (2D: exact fp64)
        p.form = (nd == 3 && !m.robust) ? 0 : bound < 0 ? 1 : bound < fp_bound ? 2 : bound < 2147483647.0 ? 1 : 0;
        p.form = std::min(p.form, p.fan == 9 ? 2 : p.fan);
      }
      int tile[3];
      ftkx::tile_dims(nd, tile);
      for (int d = 0; d < 3; d ++) p.ntiles[d] = d < nd ? (int)((c->core_sz[d] + tile[d] - 1) / tile[d]) : 1;
      if (tile_walk && !tiles.empty() && tiles.back().cull == p.cull && tiles.back().fan == p.fan && tiles.back().step + tiles.back().nsteps == p.step) {
        tiles.back().nsteps ++;
        tiles.back().form = std::min(tiles.back().form, p.form);
      } else tiles.push_back(p);
    }
  }
  // one upload for all descriptors: the mask jobs of each sub-batch, then ONE array of Fields for the whole batch -- the steps of
  // sub-batch 0, 1, ... back to back (each cull / exact launch gets its slice of it) and the tile requests behind them; the
  // record kernel looks a simplex's request up in that array by the index its pass descriptor carries
  size_t total = 0;
  std::vector<size_t> job_off, step_base;
  for (const Sub &sb : subs) { job_off.push_back(total); total += (sb.jobs.size() * sizeof(MaskJob) + 255) / 256 * 256; }
  const size_t fields_off = total;
  size_t nfields = 0;
  for (const Sub &sb : subs) { step_base.push_back(nfields); nfields += sb.steps.size(); }
  const size_t tile_base = nfields;
  nfields += tile_fields.size();
  total += (nfields * sizeof(Fields) + 255) / 256 * 256;
  if ((nfields >> (64 - ftkx::kPassStepShift)) != 0) return fail(c, FTKX_E_INVALID, "sweep: too many requests in one batch (%zu)", nfields);
  if (total) {
    int rc = ensure_desc(c, total);
    if (rc) return rc;
    Fields *hf = (Fields *)((char *)c->h_desc + fields_off);
    for (size_t i = 0; i < subs.size(); i ++) {
      if (!subs[i].jobs.empty()) memcpy((char *)c->h_desc + job_off[i], subs[i].jobs.data(), subs[i].jobs.size() * sizeof(MaskJob));
      if (!subs[i].steps.empty()) memcpy(hf + step_base[i], subs[i].steps.data(), subs[i].steps.size() * sizeof(Fields));
    }
    for (size_t i = 0; i < tile_fields.size(); i ++) hf[tile_base + i] = tile_fields[i];
    HIP_TRY(c, hipMemcpyAsync(c->d_desc, c->h_desc, total, hipMemcpyHostToDevice, c->stream));
  }
  const Fields *d_fields = (const Fields *)((char *)c->d_desc + fields_off);
  for (TileParams &p : tiles) { p.step += (int)tile_base; p.steps = d_fields + p.step; }
  if (cull_only && (subs.size() > 1 || !tiles.empty()))
    return fail(c, FTKX_E_UNSUPPORTED, "ftkx_sweep_cull: the batch needs masks under two factors or the tile path (send the slice itself)");
  for (size_t i = 0; i < subs.size(); i ++) {
    const Sub &sb = subs[i];
    if (sb.steps.empty()) continue;
    const MaskJob *d_jobs = (const MaskJob *)((char *)c->d_desc + job_off[i]);
    const Fields *d_steps = d_fields + step_base[i];
    if (!sb.jobs.empty()) { ev_begin(c, K_MASK); ftkx::launch_masks(m, d_jobs, (int)sb.jobs.size(), c->stream); ev_end(c); }
    // the survivor list is shared by the sub-batches of one collect: the exact kernel of sub-batch i must not re-test the
    // survivors of sub-batch i-1, so each sub-batch gets its own list segment by resetting the list counter in between
    if (i > 0) {
      HIP_TRY(c, hipMemsetAsync(c->d_counters + ftkx::CNT_SURVIVOR_LIST, 0, sizeof(u64), c->stream));
      HIP_TRY(c, hipMemsetAsync(c->d_counters + ftkx::CNT_REFINE_LIST, 0, sizeof(u64), c->stream));
    }
    if (cull_done) {    // the survivor list of exactly these steps is on the device already (cull-ahead, see ftkx_ctx::ahead)
      if (subs.size() != 1 || !sb.jobs.empty()) return fail(c, FTKX_E_DEVICE, "internal: cull-ahead taken over by a batch that rebuilds masks");
    } else {
      ev_begin(c, K_CULL);
      if (two_level) ftkx::launch_cull_two_level(m, d_steps, (int)sb.steps.size(), c->d_refine, c->refine_capacity, c->d_list, c->list_capacity, c->stream);
      else ftkx::launch_cull(m, d_steps, (int)sb.steps.size(), c->d_list, c->list_capacity, c->stream);
      ev_end(c);
    }
    if (cull_only) {
      // (the exact kernel is what publishes the list peak; without it the host reads the list counter itself)
      ftkx::launch_sparse_cells(m, d_steps, c->d_list, c->list_capacity, sparse_field, c->d_cells, c->cells_cap, c->stream);
      continue;
    }
    ev_begin(c, K_EXACT); ftkx::launch_exact(m, d_steps, (int)step_base[i], c->d_list, c->list_capacity, c->stream); ev_end(c);
  }
  if (cull_only) { HIP_TRY(c, hipGetLastError()); return FTKX_OK; }
  if (!tiles.empty()) {
    if (!c->d_tile_stats) {
      HIP_TRY(c, hipMalloc((void **)&c->d_tile_stats, 512 * sizeof(u64)));
      HIP_TRY(c, hipMemsetAsync(c->d_tile_stats, 0, 512 * sizeof(u64), c->stream));
    }
    for (TileParams &p : tiles) { p.stats = c->d_tile_stats; ev_begin(c, K_TILE); ftkx::launch_tile(p, c->stream); ev_end(c); }
    ftkx::launch_tile_stats_fold(c->d_tile_stats, m.counters, c->stream);
  }
  // the FP64 half, once for the whole batch: records of every simplex that passed (timed with the kernel family that fed it)
  // (K_EXACT also behind tile requests: K_TILE is the integer test of every simplex and nothing else -- bench.py's int-VALU figure)
  if (nfields) { ev_begin(c, K_EXACT); ftkx::launch_records(m, d_fields, c->stream); ev_end(c); }
  HIP_TRY(c, hipGetLastError());
  return FTKX_OK;
}

// may ftkx_sweep_collect take the cull-ahead's survivor list over?
bool ahead_serves_pending(const ftkx_ctx *c, const Mesh &m, bool two_level)
{
  if (c->ahead.empty() || c->ahead.size() != c->pending.size()) return false;
  for (size_t i = 0; i < c->pending.size(); i ++) {
    const Request &r = c->pending[i];
    const ftkx_ctx::AheadStep &a = c->ahead[i];
    if (r.t != a.t || r.scope != a.scope || r.mode != MODE_FAST) return false;
    auto s0 = c->slices.find(r.t);
    if (s0 == c->slices.end() || s0->second.M != a.M[0] || (two_level ? s0->second.U : nullptr) != a.U[0] || !masks_valid(c, s0->second, r.factor, two_level, m.u_rows)) return false;
    if (r.scope & FTKX_SCOPE_INTERVAL) {
      auto s1 = c->slices.find(r.t + 1);
      if (s1 == c->slices.end() || s1->second.M != a.M[1] || (two_level ? s1->second.U : nullptr) != a.U[1] || !masks_valid(c, s1->second, r.factor, two_level, m.u_rows)) return false;
    }
  }
  return true;
}

}  // namespace ftkxh

extern "C" {

int ftkx_sweep_enqueue(ftkx_ctx *c, int t, int scope, unsigned long long factor)
{
  if (!c) return fail(nullptr, FTKX_E_INVALID, "null context");
  if (!c->mesh_set) return fail(c, FTKX_E_INVALID, "sweep: call ftkx_set_mesh first");
  if (c->sr_open && !c->sr_internal) return fail(c, FTKX_E_INVALID, "ftkx_sweep_enqueue: series passes open (ftkx_sweep_series_submit), complete them first");
  if (scope < FTKX_SCOPE_ORDINAL || scope > FTKX_SCOPE_BOTH) return fail(c, FTKX_E_INVALID, "sweep: bad scope %d", scope);
  if (scope == FTKX_SCOPE_BOTH && c->opt.tag_mode == FTKX_TAG_WORK_INDEX)
    return fail(c, FTKX_E_INVALID, "sweep: FTKX_SCOPE_BOTH needs an element tag (work indices of the two scopes collide)");
  if (factor == 0) return fail(c, FTKX_E_INVALID, "sweep: factor must be non-zero");
  auto it0 = c->slices.find(t);
  if (it0 == c->slices.end()) return fail(c, FTKX_E_NOSLICE, "sweep: slice %d not resident", t);
  Slice *s0 = &it0->second, *s1 = nullptr;
  if (scope & FTKX_SCOPE_INTERVAL) {
    auto it1 = c->slices.find(t + 1);
    if (it1 == c->slices.end()) return fail(c, FTKX_E_NOSLICE, "sweep: interval [%d, %d] needs slice %d", t, t + 1, t + 1);
    s1 = &it1->second;
  }
  if (s1 && ((s0->J == nullptr) != (s1->J == nullptr) || (s0->S == nullptr) != (s1->S == nullptr)))
    return fail(c, FTKX_E_INVALID, "sweep: slices %d and %d disagree on which of J / S are given", t, t + 1);
  // coordinate arrays are indexed by vertex coordinates: they must cover the vertex box
  if (c->opt.coords_mode == 2)
    for (int d = 0; d < c->nd; d ++)
      if (c->dom_st[d] < 0 || (size_t)(c->dom_st[d] + c->dom_sz[d]) > c->rect_n[d])
        return fail(c, FTKX_E_INVALID, "sweep: rectilinear coordinates of axis %d have %zu entries, vertices reach %lld", d, c->rect_n[d], c->dom_st[d] + c->dom_sz[d] - 1);
  if (c->opt.coords_mode == 3 && (c->dom_st[0] < 0 || c->dom_st[1] < 0 || (size_t)(c->dom_st[0] + c->dom_sz[0]) > c->expl_n0 || (size_t)(c->dom_st[1] + c->dom_sz[1]) > c->expl_n1))
    return fail(c, FTKX_E_INVALID, "sweep: explicit coordinates are %zu x %zu, the vertex box needs %lld x %lld", c->expl_n0, c->expl_n1, c->dom_st[0] + c->dom_sz[0], c->dom_st[1] + c->dom_sz[1]);
  for (int d = 0; d < c->nd; d ++)
    if (c->core_sz[d] == 0) return FTKX_OK;    // empty core: nothing to enumerate
  HIP_TRY(c, hipSetDevice(c->device));
  const int nd = c->nd;

  // Is the strict-sign cull usable?  Only with the robust integer test (the FP64 test of the non-robust 3D mode has no such
  // property) and a power-of-two factor: the masks test v >= 1/factor on doubles, which equals trunc(v * factor) >= 1 only then
  // (the tracker always passes 1 << nbits); any other factor a direct caller hands over takes the tile path, which quantises like
  // the reference.  Determinants that could leave int64 are dealt with per vertex (MaskJob::big), not per request.
  const bool fast = !c->opt.exact_only && pow2_factor(factor) && (nd == 2 || c->opt.robust);
  if ((s0->sparse || (s1 && s1->sparse)) && (!fast || c->dense_collects > 0))
    return fail(c, FTKX_E_UNSUPPORTED, "sweep: a masked halo slice only serves sweeps that use the cull (send the slice itself)");
  if (c->pending.empty()) memset(&c->stats, 0, sizeof(c->stats));
  c->pending.push_back(Request{t, scope, factor, fast ? (c->dense_collects > 0 ? MODE_TILE_CULL : MODE_FAST) : MODE_TILE});

  u64 cells = 1;
  for (int d = 0; d < nd; d ++) cells *= (u64)c->core_sz[d];
  const u64 n_ord = nd == 2 ? 2 : 6, n_int = nd == 2 ? 10 : 54;
  c->stats.cells += cells;
  c->stats.work_items += cells * (((scope & 1) ? n_ord : 0) + ((scope & 2) ? n_int : 0));
  c->stats.cull_enabled = fast ? 1 : 0;
  return FTKX_OK;
}

int ftkx_sweep_collect(ftkx_ctx *c, const ftkx_cp_t **out, size_t *n_out)
{
  if (!c) return fail(nullptr, FTKX_E_INVALID, "null context");
  HIP_TRY(c, hipSetDevice(c->device));
  if (out) *out = nullptr;
  if (n_out) *n_out = 0;
  if (c->pending.empty()) return FTKX_OK;
  int rc;
  if ((rc = ensure_hit_buffer(c, std::max<u64>(c->capacity, 1u << 16)))) { c->pending.clear(); return rc; }
  if (c->nd == 3 && (rc = ensure_fragile(c, std::max<u64>(c->fragile_capacity, 1u << 12)))) { c->pending.clear(); return rc; }
  bool any_fast = false;
  u64 fast_cells = 0;
  {
    u64 cells = 1;
    for (int d = 0; d < c->nd; d ++) cells *= (u64)c->core_sz[d];
    for (const Request &r : c->pending) if (r.mode == MODE_FAST) { any_fast = true; fast_cells += cells; }
  }
  if (c->dense_collects > 0) c->dense_collects --;          // the fast path is probed again after a while
  if (any_fast && (rc = ensure_list(c, std::max<u64>(c->list_capacity, 1u << 20)))) { c->pending.clear(); return rc; }
  if (any_fast && (rc = ensure_refine(c, std::max<u64>(c->refine_capacity, 1u << 20)))) { c->pending.clear(); return rc; }
  // upper bound of the tags this batch can emit -> number of key bits for the device sort
  int key_bits = 64;
  if (c->opt.tag_mode != FTKX_TAG_REFERENCE) {            // REFERENCE tags go through int32 products and may wrap to anything
    int t_max = 0;
    for (const Request &r : c->pending) t_max = std::max(t_max, r.t);
    long double bound = c->nd == 2 ? 12.0L : 60.0L;
    const bool work_index = c->opt.tag_mode == FTKX_TAG_WORK_INDEX;
    for (int d = 0; d < c->nd; d ++) bound *= (long double)(work_index ? c->core_sz[d] : c->dom_sz[d]);
    if (!work_index) bound *= (long double)(t_max + 2);
    int b = 1;
    while (b < 64 && ldexpl(1.0L, b) <= bound) b ++;
    key_bits = b;
  }
  bool use_ahead = false;
  if (!c->ahead.empty()) {
    Mesh m; fill_mesh(c, m);
    use_ahead = ahead_serves_pending(c, m, ftkx::masks_have_summary(m));
    c->ahead.clear();                                        // one use; and a replay below culls afresh
  }
  for (int attempt = 0; ; attempt ++) {
    // (cull-ahead: the counters were zeroed before that cull and hold its list counts)
    if (!use_ahead) HIP_TRY(c, hipMemsetAsync(c->d_counters, 0, ftkx::CNT_N * sizeof(u64), c->stream));
    if ((rc = run_batch(c, nullptr, use_ahead))) { c->pending.clear(); return rc; }
    use_ahead = false;
    HIP_TRY(c, hipMemcpyAsync(c->h_counters, c->d_counters, ftkx::CNT_N * sizeof(u64), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    c->ahead_staged = false;
    // (records <= simplices that passed: the 2D type filter may drop some; the pass list shares the hit buffer's capacity)
    const u64 hits = std::max(c->h_counters[ftkx::CNT_HITS], c->h_counters[ftkx::CNT_PASS]);
    const u64 listed = c->h_counters[ftkx::CNT_LIST_PEAK], refined = c->h_counters[ftkx::CNT_REFINE_PEAK];
    const u64 fragile = c->h_counters[ftkx::CNT_FRAGILE];
    if (hits <= c->capacity && listed <= c->list_capacity && refined <= c->refine_capacity && fragile <= c->fragile_capacity) { ev_harvest(c); break; }
    // a buffer was too small (records / survivors beyond capacity were only counted): grow to what this batch needs, replay it
    for (auto &e : c->events) { ev_give(c, e.second.first); ev_give(c, e.second.second); }
    c->events.clear();
    if (attempt == 4) { c->pending.clear(); return fail(c, FTKX_E_DEVICE, "buffer overflow persisted after regrowing four times"); }
    // Most cells survive the cull (data whose quantised magnitudes can overflow the determinants almost everywhere):
    // a survivor list would be as large as the input.  Such a batch goes through the tile kernel instead, which stages each
    // tile's vertices once and applies the same cull rule in LDS.
    if (any_fast && (listed > c->list_capacity || refined > c->refine_capacity) && (listed > fast_cells / 8 || refined * 8 > fast_cells / 8)) {
      for (const Request &r : c->pending) {
        auto a = c->slices.find(r.t), b = c->slices.find(r.t + 1);
        if ((a != c->slices.end() && a->second.sparse) || ((r.scope & FTKX_SCOPE_INTERVAL) && b != c->slices.end() && b->second.sparse)) {
          c->pending.clear();
          return fail(c, FTKX_E_UNSUPPORTED, "sweep: most cells survive the cull and a masked halo slice is involved (send the slice itself)");
        }
      }
      for (Request &r : c->pending) if (r.mode == MODE_FAST) r.mode = MODE_TILE_CULL;
      any_fast = false;
      c->dense_collects = 16;
      if (hits > c->capacity && (rc = ensure_hit_buffer(c, 2 * hits + 1024))) { c->pending.clear(); return rc; }
      continue;
    }
    if (fragile > c->fragile_capacity && (rc = ensure_fragile(c, fragile + fragile / 8 + 1024))) { c->pending.clear(); return rc; }
    if (refined > c->refine_capacity && (rc = ensure_refine(c, refined + refined / 8 + 1024))) { c->pending.clear(); return rc; }
    if (listed > c->list_capacity && (rc = ensure_list(c, listed + listed / 8 + 1024))) { c->pending.clear(); return rc; }
    // with a truncated survivor list the hit count is a lower bound: leave generous room
    const u64 want_hits = std::max<u64>(hits + hits / 8 + 1024, (listed > c->list_capacity || refined > c->refine_capacity) ? 2 * hits + 1024 : 0);
    if (want_hits > c->capacity && (rc = ensure_hit_buffer(c, want_hits))) { c->pending.clear(); return rc; }
  }
  c->pending.clear();
  const size_t n = (size_t)c->h_counters[ftkx::CNT_HITS];
  c->stats.hits = n;
  c->stats.cells_survived = c->h_counters[ftkx::CNT_CELLS_SURVIVED];
  c->stats.simplices_tested = c->h_counters[ftkx::CNT_SIMPLICES_TESTED];
  if ((rc = ensure_host_buffer(c, n))) return rc;
  // 3D records whose class hangs on the last bits of pow / acos / cos (an eigenvalue of the Hessian that is zero up to rounding):
  // classified again here, with the libm the reference itself runs on, and written back before the records are sorted.  Rare -- an
  // exactly singular Hessian takes plateaus or lattice-aligned data -- and then one small round trip.
  if (const u64 nf = c->h_counters[ftkx::CNT_FRAGILE]) {
    std::vector<u64> frag((size_t)nf * 10), pairs((size_t)nf * 2);
    HIP_TRY(c, hipMemcpyAsync(frag.data(), c->d_fragile, frag.size() * sizeof(u64), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    for (size_t i = 0; i < (size_t)nf; i ++) {
      double A[3][3];
      memcpy(A, &frag[i * 10 + 1], sizeof(A));
      pairs[2 * i] = frag[i * 10];
      pairs[2 * i + 1] = (u64)ftkx::classify3(A, c->opt.jacobian_symmetric != 0);
    }
    if ((rc = ensure_desc(c, pairs.size() * sizeof(u64)))) return rc;
    memcpy(c->h_desc, pairs.data(), pairs.size() * sizeof(u64));
    HIP_TRY(c, hipMemcpyAsync(c->d_desc, c->h_desc, pairs.size() * sizeof(u64), hipMemcpyHostToDevice, c->stream));
    hipLaunchKernelGGL(patch_types_kernel, dim3((unsigned)((nf + 255) / 256)), dim3(256), 0, c->stream, c->d_hits, (const u64 *)c->d_desc, (size_t)nf);
    HIP_TRY(c, hipGetLastError());
    c->stats.reclassified = nf;
  }
  if (n >= 4096 && n < (1ull << 31)) {
    if ((rc = sort_hits_on_device(c, n, key_bits))) return rc;
    HIP_TRY(c, hipMemcpyAsync(c->h_hits, c->d_sorted, n * sizeof(ftkx_cp_t), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
  } else if (n) {
    HIP_TRY(c, hipMemcpyAsync(c->h_hits, c->d_hits, n * sizeof(ftkx_cp_t), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    std::sort(c->h_hits, c->h_hits + n, [](const ftkx_cp_t &a, const ftkx_cp_t &b) { return a.tag < b.tag; });
  }
  if (out) *out = c->h_hits;
  if (n_out) *n_out = n;
  return FTKX_OK;
}

int ftkx_sweep_enqueue_many(ftkx_ctx *c, const int *ts, const int *scopes, const unsigned long long *factors, int n)
{
  if (!c || (n > 0 && (!ts || !scopes || !factors))) return fail(c, FTKX_E_INVALID, "null argument");
  for (int i = 0; i < n; i ++) { const int rc = ftkx_sweep_enqueue(c, ts[i], scopes[i], factors[i]); if (rc) { c->pending.clear(); return rc; } }
  return FTKX_OK;
}

int ftkx_sweep_cancel(ftkx_ctx *c)
{
  if (!c) return fail(nullptr, FTKX_E_INVALID, "null context");
  c->pending.clear();
  return FTKX_OK;
}

int ftkx_sweep(ftkx_ctx *c, int t, int scope, unsigned long long factor, const ftkx_cp_t **out, size_t *n_out)
{
  if (!c) return fail(nullptr, FTKX_E_INVALID, "null context");
  if (!c->pending.empty()) return fail(c, FTKX_E_INVALID, "ftkx_sweep: asynchronous sweeps pending, collect first");
  if (c->sr_open && !c->sr_internal) return fail(c, FTKX_E_INVALID, "ftkx_sweep: series passes open (ftkx_sweep_series_submit), complete them first");
  int rc = ftkx_sweep_enqueue(c, t, scope, factor);
  if (rc) return rc;
  return ftkx_sweep_collect(c, out, n_out);
}

}  // extern "C"
