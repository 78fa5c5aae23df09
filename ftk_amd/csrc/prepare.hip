// The one-pass pre-pass of a sweep: ftkx_slices_prepare (sign masks + the reduction update_vector_field_scaling_factor needs, one
// kernel, each slice read once) and the cull-ahead that is queued behind it (ftkx_sweep_announce).  Reference counterpart:
// critical_point_tracker::update_vector_field_scaling_factor, include/ftk/filters/critical_point_tracker.hh:850-864
// (ndarray::resolution, include/ftk/ndarray.hh:770-778).
#include "ctx.hpp"

using namespace ftkxh;

namespace ftkxh {

__global__ void init_red_kernel(u64 *red, size_t nslots, u64 *counters)
{
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < nslots) { red[2 * i] = 0x7fefffffffffffffull; red[2 * i + 1] = 0ull; }   // {min = DBL_MAX, max = 0} as bit patterns
  if (counters && i < (size_t)ftkx::CNT_N) counters[i] = 0ull;                       // (cull-ahead: the sweep's counters, zeroed here)
}

// The reduction, folded per slice (64 {min, max} slots -> one pair; bit patterns of non-negative doubles order like the values) and
// written into coherent pinned host memory by the GPU itself, with a sequence number stored behind it with system scope.  The host
// spins on that word (ftkx_slices_prepare with a cull queued behind it: a stream or event wait would, in practice, also wait for
// work queued AFTER this point).  ONE workgroup, a wavefront per slice and eight slices in flight per wavefront: a system-scope
// release writes the L2 back, so the fewer wavefronts execute one the better (a wavefront per slice in its own workgroup cost
// ~0.4 us per slice).
__global__ __launch_bounds__(256) void readback_kernel(const u64 *__restrict__ red, u64 *dst, unsigned k, unsigned *flag, unsigned seq)
{
  const unsigned wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
  for (unsigned base = wv * 8; base < k; base += 32) {
    u64 mn[8], mx[8];
#pragma unroll
    for (int j = 0; j < 8; j ++) {
      const unsigned i = base + j < k ? base + j : k - 1;
      mn[j] = red[(size_t)i * 128 + 2 * lane]; mx[j] = red[(size_t)i * 128 + 2 * lane + 1];
    }
#pragma unroll
    for (int j = 0; j < 8; j ++) {
      for (int o = 32; o > 0; o >>= 1) {
        const u64 a = __shfl_down(mn[j], o), b = __shfl_down(mx[j], o);
        mn[j] = a < mn[j] ? a : mn[j]; mx[j] = b > mx[j] ? b : mx[j];
      }
      if (lane == 0 && base + j < k) { dst[2 * (base + j)] = mn[j]; dst[2 * (base + j) + 1] = mx[j]; }
    }
  }
  __threadfence_system();
  __syncthreads();
  if (threadIdx.x == 0) __hip_atomic_store(flag, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

// descriptors from pinned host memory into device memory, by a kernel: a launch never holds the host, whereas a copy or fill queued
// behind a running kernel was seen to (cull-ahead: everything queued behind the mask kernel is a kernel)
__global__ __launch_bounds__(256) void fetch_desc_kernel(const u64 *__restrict__ src, u64 *__restrict__ dst, size_t n)
{
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) dst[i] = src[i];
}

void launch_init_red(u64 *red, size_t nslots, u64 *counters, hipStream_t st)
{ hipLaunchKernelGGL(init_red_kernel, dim3((unsigned)((std::max<size_t>(nslots, ftkx::CNT_N) + 255) / 256)), dim3(256), 0, st, red, nslots, counters); }
void launch_fetch_desc(const void *pinned_src, void *device_dst, size_t bytes, hipStream_t st)
{ hipLaunchKernelGGL(fetch_desc_kernel, dim3(4), dim3(256), 0, st, (const u64 *)pinned_src, (u64 *)device_dst, bytes / 8); }

}  // namespace ftkxh

namespace {

// ftkx_slices_prepare with announced sweeps.  Everything the host must hand over goes up BEFORE the mask kernel, with the mask jobs
// (copies and fills queued behind a running kernel were seen to hold the host until it finished); behind the mask kernel only
// kernels are queued.  ahead_steps: the announced sweeps' descriptors, or nothing whenever something is not as the fast path needs
// it -- the sweep then culls at collect time as before.
bool ahead_steps(ftkx_ctx *c, bool two_level, u64 hint, std::vector<Fields> &steps, std::vector<ftkx_ctx::AheadStep> &rec)
{
  steps.clear(); rec.clear();
  if (c->announced.empty() || c->dense_collects > 0) return false;
  for (int d = 0; d < c->nd; d ++) if (c->core_sz[d] == 0) return false;
  for (const auto &ts : c->announced) {
    auto a = c->slices.find(ts.first);
    if (a == c->slices.end() || a->second.sparse || a->second.mask_factor != hint || a->second.mask_big) return false;
    const Slice *s1 = nullptr;
    if (ts.second & FTKX_SCOPE_INTERVAL) {
      auto b = c->slices.find(ts.first + 1);
      if (b == c->slices.end() || b->second.sparse || b->second.mask_factor != hint || b->second.mask_big) return false;
      s1 = &b->second;
    }
    if (ts.second == FTKX_SCOPE_BOTH && c->opt.tag_mode == FTKX_TAG_WORK_INDEX) return false;
    Fields f;
    memset(&f, 0, sizeof(f));
    f.t = ts.first; f.scope_mask = ts.second;
    f.M[0] = a->second.M; f.M[1] = s1 ? s1->M : nullptr;
    f.U[0] = two_level ? a->second.U : nullptr; f.U[1] = (two_level && s1) ? s1->U : nullptr;
    steps.push_back(f);
    rec.push_back({ts.first, ts.second, {f.M[0], f.M[1]}, {f.U[0], f.U[1]}});
  }
  return !steps.empty();
}

void ahead_launch(ftkx_ctx *c, const Mesh &m, bool two_level, const Fields *d_steps, int nsteps)
{
  ev_begin(c, K_CULL);
  if (two_level) ftkx::launch_cull_two_level(m, d_steps, nsteps, c->d_refine, c->refine_capacity, c->d_list, c->list_capacity, c->stream);
  else ftkx::launch_cull(m, d_steps, nsteps, c->d_list, c->list_capacity, c->stream);
  ev_end(c);
}

}  // namespace

extern "C" {

// One pass over the slices for the whole sweep: the sign masks (built under factor_hint, which must not exceed the factor the
// sweeps will use -- the scaling factor only grows, so the factor in force BEFORE these slices arrived qualifies) and, fused into
// the same kernel, what update_vector_field_scaling_factor needs of each slice.  See MaskJob in sweep_params.hpp.
int ftkx_slices_prepare(ftkx_ctx *c, const int *ts, int n, unsigned long long factor_hint, double *res_below, double *max_abs)
{
  if (!c || (n > 0 && !ts)) return fail(c, FTKX_E_INVALID, "null argument");
  if (!c->pending.empty()) return fail(c, FTKX_E_INVALID, "ftkx_slices_prepare: sweeps pending, collect first");
  if (c->sr_open && !c->sr_internal) return fail(c, FTKX_E_INVALID, "ftkx_slices_prepare: series passes open (ftkx_sweep_series_submit), complete them first");
  const u64 hint = factor_hint ? factor_hint : 256;          // the smallest factor there is (minbits = 8)
  if (!pow2_factor(hint)) return fail(c, FTKX_E_INVALID, "ftkx_slices_prepare: factor_hint must be a power of two");
  c->ahead.clear();
  HIP_TRY(c, hipSetDevice(c->device));
  const double cap = 1.0 / (double)hint;
  std::vector<Slice *> all, todo;
  for (int i = 0; i < n; i ++) {
    auto it = c->slices.find(ts[i]);
    if (it == c->slices.end()) return fail(c, FTKX_E_NOSLICE, "ftkx_slices_prepare: timestep %d not resident", ts[i]);
    all.push_back(&it->second);
  }
  Mesh m; fill_mesh(c, m);
  // contexts that never cull (exact_only, non-robust 3D) have no use for masks: the plain pre-pass serves them
  const bool want_masks = !c->opt.exact_only && (c->nd == 2 || c->opt.robust) && c->dense_collects == 0;
  const bool two_level = ftkx::masks_have_summary(m);
  int rc;
  for (Slice *s : all) {
    if (s->sparse && !s->have_res) return fail(c, FTKX_E_NOSLICE, "ftkx_slices_prepare: a masked halo slice has no data to reduce (its owner's reduction: ftkx_set_slice_resolution)");
    if (s->sparse) continue;
    if (!want_masks) { if ((rc = slice_resolution(c, *s))) return rc; continue; }
    // already reduced under this hint: nothing to do -- also when its masks were then found unusable (vertices that can overflow a
    // determinant): the sweep rebuilds those with the per-vertex rule, another pass here would only repeat the finding
    if (s->have_fused && s->fused_factor == hint) continue;
    if (std::find(todo.begin(), todo.end(), s) == todo.end()) todo.push_back(s);
  }
  if (!todo.empty()) {
    const size_t k = todo.size();
    if (c->red_cap < k) {
      if (c->d_red) { (void)hipFree(c->d_red); c->d_red = nullptr; c->red_cap = 0; }
      HIP_TRY(c, hipMalloc((void **)&c->d_red, k * 128 * sizeof(u64)));
      c->red_cap = k;
    }
    for (Slice *s : todo) if ((rc = ensure_mask_arrays(c, *s, two_level))) return rc;
    // cull-ahead: the masks are about to be built under the hint -- mark them so (the validation further down may take that back).
    // Everything that could synchronise the device happens before the mask launch; the announced sweeps' descriptors are put
    // together AFTER it, while the mask kernel runs (per-step host work in front of the launch would delay the kernel by as much)
    bool ahead_ok = !c->announced.empty() && c->dense_collects == 0;
    // (the marks below are set BEFORE the masks exist: every error exit between here and the point where the reduction has arrived
    // takes them back -- a slice that still held masks from an earlier prepare under a larger factor would otherwise pass masks_valid
    // for the smaller hint and cull with thresholds that are too tight)
    struct MarkGuard { std::vector<Slice *> *v; bool armed; ~MarkGuard() { if (armed) for (Slice *s : *v) { s->mask_factor = 0; s->have_fused = false; } } } marks{&todo, false};
    if (ahead_ok) {
      marks.armed = true;
      for (Slice *s : todo) { s->mask_factor = hint; s->mask_big = false; s->u_rows = m.u_rows; }
      const size_t bytes = c->announced.size() * sizeof(Fields);
      rc = FTKX_OK;
      if (c->ahead_staged) { HIP_TRY(c, hipStreamSynchronize(c->stream)); c->ahead_staged = false; }   // (prepare after prepare, no collect in between)
      if (c->ahead_cap < bytes) {
        if (c->h_ahead) { HIP_TRY(c, hipStreamSynchronize(c->stream)); (void)hipHostFree(c->h_ahead); c->h_ahead = nullptr; }
        if (c->d_ahead) { (void)hipFree(c->d_ahead); c->d_ahead = nullptr; }
        c->ahead_cap = 0;
        const size_t capb = (bytes * 2 + 4095) / 4096 * 4096;
        HIP_TRY(c, hipHostMalloc(&c->h_ahead, capb, hipHostMallocCoherent));
        HIP_TRY(c, hipMalloc(&c->d_ahead, capb));
        c->ahead_cap = capb;
      }
      if (c->h_red_cap < k * 2) {
        if (c->h_red) { HIP_TRY(c, hipStreamSynchronize(c->stream)); (void)hipHostFree(c->h_red); c->h_red = nullptr; c->h_red_cap = 0; }
        const size_t slots = (k + k / 4 + 8) * 2;
        HIP_TRY(c, hipHostMalloc((void **)&c->h_red, (slots + 8) * sizeof(u64), hipHostMallocCoherent));
        c->h_red_cap = slots;
        *reinterpret_cast<volatile unsigned *>(c->h_red + slots) = 0u;
        c->red_seq = 0;
      }
      if ((rc = ensure_list(c, std::max<u64>(c->list_capacity, 1u << 20))) || (rc = ensure_refine(c, std::max<u64>(c->refine_capacity, 1u << 20)))) return rc;
    }
    else {                         // (no cull-ahead: the masks are rewritten all the same -- whatever they were valid for is gone)
      marks.armed = true;
      for (Slice *s : todo) s->mask_factor = 0;
    }
    if ((rc = ensure_desc(c, std::max(k * sizeof(MaskJob), k * 128 * sizeof(u64))))) return rc;
    launch_init_red(c->d_red, k * 64, ahead_ok ? c->d_counters : nullptr, c->stream);
    MaskJob *jobs = (MaskJob *)c->h_desc;
    for (size_t i = 0; i < k; i ++)
      jobs[i] = with_lean_thresholds(MaskJob{todo[i]->S, todo[i]->V, todo[i]->M, two_level ? todo[i]->U : nullptr, c->d_red + i * 128, cap, HUGE_VAL}, m);   // rule off: validated below
    HIP_TRY(c, hipMemcpyAsync(c->d_desc, c->h_desc, k * sizeof(MaskJob), hipMemcpyHostToDevice, c->stream));
    ev_begin(c, K_MASK); ftkx::launch_masks(m, (const MaskJob *)c->d_desc, (int)k, c->stream); ev_end(c);
    HIP_TRY(c, hipGetLastError());
    const u64 *host = (const u64 *)c->h_desc;
    std::vector<Fields> a_steps;
    std::vector<ftkx_ctx::AheadStep> a_rec;
    if (ahead_ok) ahead_ok = ahead_steps(c, two_level, hint, a_steps, a_rec);
    if (ahead_ok) {
      // behind the mask kernel, kernels only: the reduction folded and written to pinned memory with a flag behind it, the
      // descriptors fetched from pinned memory, the cull.  The host waits for the flag ONLY; the cull runs while it forms the factors
      unsigned *flag = reinterpret_cast<unsigned *>(c->h_red + c->h_red_cap);
      const unsigned seq = ++ c->red_seq;
      hipLaunchKernelGGL(readback_kernel, dim3(1), dim3(256), 0, c->stream, (const u64 *)c->d_red, c->h_red, (unsigned)k, flag, seq);
      const size_t bytes = a_steps.size() * sizeof(Fields);
      static_assert(sizeof(Fields) % 8 == 0, "descriptors are fetched as 8-byte words");
      memcpy(c->h_ahead, a_steps.data(), bytes);
      c->ahead_staged = true;
      launch_fetch_desc(c->h_ahead, c->d_ahead, bytes, c->stream);
      ahead_launch(c, m, two_level, (const Fields *)c->d_ahead, (int)a_steps.size());
      HIP_TRY(c, hipGetLastError());
      c->ahead = a_rec;
      // spin on the flag; a device error would leave it unset: look at the stream now and then, give up after a generous while
      if (const char *why = ftkx::wait_flag(flag, seq, c->stream)) {      // nothing of this call stands: no masks (MarkGuard), no cull-ahead, no announcement
        c->ahead.clear(); c->announced.clear();
        return fail(c, FTKX_E_DEVICE, "ftkx_slices_prepare: %s", why);
      }
      ev_harvest(c, false);
    } else {
      HIP_TRY(c, hipMemcpyAsync(c->h_desc, c->d_red, k * 128 * sizeof(u64), hipMemcpyDeviceToHost, c->stream));
      HIP_TRY(c, hipStreamSynchronize(c->stream));
      ev_harvest(c);
    }
    std::vector<Slice *> with_inf;
    for (size_t i = 0; i < k; i ++) {
      u64 mn, mx;
      if (ahead_ok) { mn = c->h_red[2 * i]; mx = c->h_red[2 * i + 1]; }     // folded on the device
      else {
        mn = host[i * 128]; mx = host[i * 128 + 1];
        for (int q = 1; q < 64; q ++) { mn = std::min(mn, host[i * 128 + 2 * q]); mx = std::max(mx, host[i * 128 + 2 * q + 1]); }
      }
      Slice &s = *todo[i];
      memcpy(&s.res_below, &mn, 8);
      double mxd; memcpy(&mxd, &mx, 8);
      s.mask_factor = hint; s.mask_big = false; s.fused_factor = hint; s.have_fused = true; s.u_rows = m.u_rows;
      if (std::isinf(mxd)) with_inf.push_back(&s);        // the fused max cannot skip an Inf: the exact pre-pass gives max FINITE |v|
      else if (!s.have_res) s.maxabs = mxd;
    }
    marks.armed = false;                                          // the masks exist and the reduction has arrived
    for (Slice *s : with_inf) if ((rc = slice_resolution(c, *s))) return rc;
    // The masks were built without the per-vertex overflow rule.  They stand only if no vertex of the slice is big under the hint
    // (then under no smaller factor either); otherwise the sweep rebuilds them, rule on, under its factor (masks_valid).
    for (Slice *s : todo) if (!overflow_free(c->nd, s->maxabs, hint)) s->mask_factor = 0;
  }
  c->announced.clear();                                       // (an announcement holds for one prepare)
  for (int i = 0; i < n; i ++) {
    const Slice &s = *all[i];
    if (res_below) res_below[i] = s.have_res ? (s.res < cap ? s.res : DBL_MAX) : s.res_below;
    if (max_abs) max_abs[i] = s.maxabs;
  }
  return FTKX_OK;
}

// PROBE (tools/mask_overlap.py): slice t's mask job launched `reps` times back to back -- on the context's stream alone (nstreams 1) or
// alternately on it and on a second stream of the same priority (2), with a small dependent kernel in front of each launch when
// with_begin (the series pass's begin kernel) -- and the device time per launch.  The slice's masks must exist (ftkx_slices_prepare).
int ftkx_debug_mask_relaunch(ftkx_ctx *c, int t, int reps, int nstreams, int with_begin, double *ms_per_launch)
{
  if (!c || !ms_per_launch || reps < 1 || nstreams < 1 || nstreams > 2) return fail(c, FTKX_E_INVALID, "ftkx_debug_mask_relaunch: bad argument");
  auto it = c->slices.find(t);
  if (it == c->slices.end()) return fail(c, FTKX_E_NOSLICE, "ftkx_debug_mask_relaunch: timestep %d not resident", t);
  Slice &s = it->second;
  Mesh m; fill_mesh(c, m);
  const bool two_level = ftkx::masks_have_summary(m);
  if (!s.M || (two_level && !s.U) || !c->d_red || s.sparse) return fail(c, FTKX_E_INVALID, "ftkx_debug_mask_relaunch: ftkx_slices_prepare first");
  HIP_TRY(c, hipSetDevice(c->device));
  int rc;
  if ((rc = ensure_desc(c, 2 * 4096))) return rc;
  MaskJob *job = (MaskJob *)c->h_desc;
  *job = with_lean_thresholds(MaskJob{s.S, s.V, s.M, two_level ? s.U : nullptr, c->d_red, 1.0 / 256.0, HUGE_VAL}, m);
  HIP_TRY(c, hipMemcpyAsync(c->d_desc, c->h_desc, sizeof(MaskJob), hipMemcpyHostToDevice, c->stream));
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  hipStream_t st[2] = {c->stream, nullptr};
  if (nstreams == 2 && (rc = aux_stream_get(c, false, &st[1]))) return rc;
  hipEvent_t e0, e1, eb;
  HIP_TRY(c, hipEventCreate(&e0)); HIP_TRY(c, hipEventCreate(&e1)); HIP_TRY(c, hipEventCreateWithFlags(&eb, hipEventDisableTiming));
  HIP_TRY(c, hipEventRecord(e0, st[0]));
  if (nstreams == 2) HIP_TRY(c, hipStreamWaitEvent(st[1], e0, 0));
  for (int r = 0; r < reps; r ++) {
    hipStream_t q = st[r % nstreams];
    if (with_begin) launch_fetch_desc(c->h_desc, (char *)c->d_desc + 4096, 4096, q);
    ftkx::launch_masks(m, (const MaskJob *)c->d_desc, 1, q);
  }
  if (nstreams == 2) { HIP_TRY(c, hipEventRecord(eb, st[1])); HIP_TRY(c, hipStreamWaitEvent(st[0], eb, 0)); }
  HIP_TRY(c, hipEventRecord(e1, st[0]));
  HIP_TRY(c, hipEventSynchronize(e1));
  float ms = 0.f;
  HIP_TRY(c, hipEventElapsedTime(&ms, e0, e1));
  (void)hipEventDestroy(e0); (void)hipEventDestroy(e1); (void)hipEventDestroy(eb);
  if (st[1]) aux_stream_put(c, false, st[1]);
  *ms_per_launch = (double)ms / reps;
  return FTKX_OK;
}

// The sweeps that will follow the next ftkx_slices_prepare, in the order they will be enqueued: that call then queues their cull
// right behind the mask kernel (it needs the masks, not the factor), so that it runs while the host still waits for the reduction
// and forms the factors.  A hint, never an obligation: ftkx_sweep_collect uses the list only if the pending sweeps are exactly these.
int ftkx_sweep_announce(ftkx_ctx *c, const int *ts, const int *scopes, int n)
{
  if (!c || (n > 0 && (!ts || !scopes))) return fail(c, FTKX_E_INVALID, "null argument");
  c->announced.clear();
  for (int i = 0; i < n; i ++) {
    if (scopes[i] < FTKX_SCOPE_ORDINAL || scopes[i] > FTKX_SCOPE_BOTH) { c->announced.clear(); return fail(c, FTKX_E_INVALID, "ftkx_sweep_announce: bad scope %d", scopes[i]); }
    c->announced.push_back({ts[i], scopes[i]});
  }
  return FTKX_OK;
}

}  // extern "C"
