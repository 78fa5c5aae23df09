// Host side of the C ABI declared in include/ftkx.h: context, HBM-resident slices, launch bookkeeping, hit download.
// The reference's counterpart is the per-call host wrapper extract_cp2dt<scope> / extract_cp3dt<scope>
// (src/filters/critical_point_tracer_2d_regular.cu:168-272, ..._3d_regular.cu:144-250), which re-allocates, re-uploads and
// frees everything on every call and synchronises the whole device; here slices stay resident, launches go to a stream,
// and the hit buffer is persistent (grown and the batch replayed if a launch overflows it).
#include <map>
#include <mutex>
#include "ctx.hpp"
#include "cp_device.hpp"
#include <sched.h>

using namespace ftkxh;

namespace { thread_local std::string g_last_error; }

namespace ftkx { void set_global_error(const char *msg) { g_last_error = msg ? msg : ""; } }

namespace ftkx {
// Waits for a sequence number a kernel stores, with system scope, into coherent pinned memory.  A short spin (the common case: the
// value is microseconds away), then the core is given up between polls -- a multi-device tracker waits like this on one thread per
// device while the trace's worker pool may want the same cores -- and the stream is looked at once per millisecond, so that a queue
// that faulted or drained without the store is noticed promptly.  Returns nullptr, or what went wrong.
const char *wait_flag(const unsigned *flag, unsigned seq, hipStream_t stream)
{
  const auto t_start = std::chrono::steady_clock::now();
  auto t_poll = t_start;
  for (unsigned long long spins = 0; __atomic_load_n(flag, __ATOMIC_ACQUIRE) != seq; spins ++) {
#if defined(__x86_64__) || defined(__i386__)
    __builtin_ia32_pause();                                  // (a spin-wait hint: the sibling hyperthread keeps its issue slots)
#endif
    if ((spins & 0x3ffull) != 0x3ffull) continue;
    const auto now = std::chrono::steady_clock::now();
    if (now - t_start > std::chrono::microseconds(200)) sched_yield();
    if (now - t_poll < std::chrono::milliseconds(1)) continue;
    t_poll = now;
    const hipError_t q = hipStreamQuery(stream);
    if (q != hipSuccess && q != hipErrorNotReady) return hipGetErrorString(q);
    if (q == hipSuccess && __atomic_load_n(flag, __ATOMIC_ACQUIRE) != seq) return "the stream drained without the result arriving";
    if (now - t_start > std::chrono::seconds(120)) return "timed out waiting for the device";
  }
  return nullptr;
}
}  // namespace ftkx

namespace ftkxh {

int fail(ftkx_ctx *c, int code, const char *fmt, ...)
{
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof(buf), fmt, ap);
  va_end(ap);
  g_last_error = buf;
  if (c) c->err = buf;
  return code;
}

size_t n_vertices(const ftkx_ctx *c)
{
  size_t n = 1;
  for (int d = 0; d < c->nd; d ++) n *= (size_t)c->ext_sz[d];
  return n;
}

int mask_pitch(const ftkx_ctx *c) { return (int)(((c->ext_sz[0] + 7) / 8) * 8 + 8); }
int u_pitch(const ftkx_ctx *c) { return (int)((((c->ext_sz[0] + 7) / 8 + 7) / 8) * 8 + 8); }
size_t u_bytes(const ftkx_ctx *c) { return (size_t)u_pitch(c) * (size_t)c->ext_sz[1] * (size_t)(c->nd == 3 ? c->ext_sz[2] : 1); }   // allocation (u_rows = 1)
// the part of a summary array that carries data: ceil(rows / u_rows) rows per plane, planes back to back
size_t u_bytes_used(const ftkx_ctx *c, const Mesh &m) { return (size_t)u_pitch(c) * (size_t)((c->ext_sz[1] + m.u_rows - 1) / m.u_rows) * (size_t)(c->nd == 3 ? c->ext_sz[2] : 1); }
size_t mask_bytes(const ftkx_ctx *c) { return (size_t)mask_pitch(c) * (size_t)c->ext_sz[1] * (size_t)(c->nd == 3 ? c->ext_sz[2] : 1); }

void free_slice(Slice &s, ftkx_ctx *pool_owner)
{
  // The newest open pass is a SPLIT pass: its tail runs on a stream of its own and may still read this slice's arrays, while whatever takes
  // them out of the pools next -- a push, the mask kernel of the next pass -- runs on the context's stream, in no order with that tail.
  // The slice is parked with that pass and comes back here when it has been completed (series.hip, release_retired).
  if (pool_owner && pool_owner->sr_open > 0) {
    ftkx_series_pending &N = pool_owner->sr_pend[pool_owner->sr_place(pool_owner->sr_open - 1)];
    if (N.open && N.split) { N.parked.push_back(s); s = Slice(); return; }
  }
  // owned copies go back to the context's pool: a streaming caller pushes and pops one slice per step, and hipMalloc + hipFree of a
  // slice-sized array cost more than sweeping a 256^3 slice
  auto give_back = [&](double *p, size_t count) {
    if (pool_owner && pool_owner->pool_F.size() < 12) pool_owner->pool_F.push_back({p, count});      // (12: a batch of steps parked with a split pass comes back at once)
    else (void)hipFree(p);
  };
  size_t nv = 0;
  if (pool_owner) { nv = 1; for (int d = 0; d < pool_owner->nd; d ++) nv *= (size_t)pool_owner->ext_sz[d]; }
  const size_t nd_ = pool_owner ? (size_t)pool_owner->nd : 0;
  if (s.ownV && s.V) give_back(s.V, nv * nd_);
  if (s.ownJ && s.J) give_back(s.J, nv * nd_ * nd_);
  if (s.ownS && s.S) give_back(s.S, nv);
  s.ownV = s.ownJ = s.ownS = false;
  if (s.M) { if (pool_owner && pool_owner->pool_M.size() < 12) pool_owner->pool_M.push_back(s.M); else (void)hipFree(s.M); }
  if (s.U) { if (pool_owner && pool_owner->pool_U.size() < 12) pool_owner->pool_U.push_back(s.U); else (void)hipFree(s.U); }
  s = Slice();
}

void release_pools(ftkx_ctx *c)
{
  for (unsigned char *p : c->pool_M) (void)hipFree(p);
  for (unsigned char *p : c->pool_U) (void)hipFree(p);
  for (auto &p : c->pool_F) (void)hipFree(p.first);
  c->pool_M.clear(); c->pool_U.clear(); c->pool_F.clear();
}

int ensure_hit_buffer(ftkx_ctx *c, u64 want)
{
  if (c->capacity >= want) return FTKX_OK;
  if (c->d_hits) { HIP_TRY(c, hipFree(c->d_hits)); c->d_hits = nullptr; c->capacity = 0; }
  if (c->d_pass) { HIP_TRY(c, hipFree(c->d_pass)); c->d_pass = nullptr; }
  HIP_TRY(c, hipMalloc((void **)&c->d_hits, want * sizeof(ftkx_cp_t)));
  HIP_TRY(c, hipMalloc((void **)&c->d_pass, want * sizeof(u64)));
  c->capacity = want;
  return FTKX_OK;
}

int ensure_fragile(ftkx_ctx *c, u64 want)
{
  if (c->fragile_capacity >= want) return FTKX_OK;
  if (c->d_fragile) { HIP_TRY(c, hipFree(c->d_fragile)); c->d_fragile = nullptr; c->fragile_capacity = 0; }
  HIP_TRY(c, hipMalloc((void **)&c->d_fragile, want * 10 * sizeof(u64)));
  c->fragile_capacity = want;
  return FTKX_OK;
}

int ensure_list(ftkx_ctx *c, u64 want)
{
  if (c->list_capacity >= want) return FTKX_OK;
  if (c->d_list) { HIP_TRY(c, hipFree(c->d_list)); c->d_list = nullptr; c->list_capacity = 0; }
  HIP_TRY(c, hipMalloc((void **)&c->d_list, want * sizeof(u64)));
  c->list_capacity = want;
  return FTKX_OK;
}

int ensure_refine(ftkx_ctx *c, u64 want)
{
  if (c->refine_capacity >= want) return FTKX_OK;
  if (c->d_refine) { HIP_TRY(c, hipFree(c->d_refine)); c->d_refine = nullptr; c->refine_capacity = 0; }
  HIP_TRY(c, hipMalloc((void **)&c->d_refine, want * sizeof(u64)));
  c->refine_capacity = want;
  return FTKX_OK;
}

int ensure_desc(ftkx_ctx *c, size_t bytes)
{
  if (c->desc_cap >= bytes) return FTKX_OK;
  if (c->h_desc) { HIP_TRY(c, hipHostFree(c->h_desc)); c->h_desc = nullptr; }
  if (c->d_desc) { HIP_TRY(c, hipFree(c->d_desc)); c->d_desc = nullptr; }
  const size_t cap = std::max<size_t>(bytes, 1 << 16);
  HIP_TRY(c, hipHostMalloc(&c->h_desc, cap, hipHostMallocDefault));
  HIP_TRY(c, hipMalloc(&c->d_desc, cap));
  c->desc_cap = cap;
  return FTKX_OK;
}

int ensure_host_buffer(ftkx_ctx *c, size_t want)
{
  if (c->h_cap >= want) return FTKX_OK;
  if (c->h_hits) { HIP_TRY(c, hipHostFree(c->h_hits)); c->h_hits = nullptr; c->h_cap = 0; }
  const size_t cap = std::max<size_t>(want, 4096);
  // non-coherent = ordinary cached host memory for the CPU (it only reads the records after a stream synchronise);
  // the default coherent mapping is uncached on this platform and made every consumer crawl (5 GB/s)
  HIP_TRY(c, hipHostMalloc((void **)&c->h_hits, cap * sizeof(ftkx_cp_t), hipHostMallocNonCoherent));
  c->h_cap = cap;
  return FTKX_OK;
}

void fill_mesh(const ftkx_ctx *c, Mesh &m)
{
  memset(&m, 0, sizeof(m));
  const int nd = c->nd;
  m.nd = nd;
  for (int d = 0; d < 3; d ++) {
    m.dom_lb[d] = (int)c->dom_st[d]; m.dom_ub[d] = (int)(c->dom_st[d] + c->dom_sz[d] - 1);
    m.core_st[d] = (int)c->core_st[d]; m.core_sz[d] = (int)c->core_sz[d];
    m.ext_st[d] = (int)c->ext_st[d]; m.ext_sz[d] = (int)c->ext_sz[d];
  }
  // lattice::prod_ of the mesh lattice (lattice.hh:156-167) and simplicial_regular_mesh::dimprod_ (int; simplicial_regular_mesh.hh:930-947)
  m.mesh_prod[0] = 1; m.dimprod[0] = 1; m.exact_prod[0] = 1;
  for (int d = 1; d <= nd; d ++) {
    m.mesh_prod[d] = m.mesh_prod[d - 1] * (u64)c->dom_sz[d - 1];
    m.exact_prod[d] = m.exact_prod[d - 1] * (u64)c->dom_sz[d - 1];
    m.dimprod[d] = (int)((u64)c->dom_sz[d - 1] * (u64)(long long)m.dimprod[d - 1]);
  }
  m.mask_pitch = mask_pitch(c);
  m.u_pitch = u_pitch(c);
  m.u_rows = 1;
  m.jacobian_symmetric = c->opt.jacobian_symmetric; m.robust = c->opt.robust;
  m.use_type_filter = c->opt.use_type_filter; m.type_filter = c->opt.type_filter;
  m.compute_degrees = c->opt.compute_degrees; m.tag_mode = c->opt.tag_mode;
  m.scalar_mode = c->scalar_mode == 1;
  m.derive_jacobian = c->opt.derive_jacobian;
  { const char *e = getenv("FTKX_RECORD_GENERAL"); m.record_general = (e && atoi(e) != 0) ? 1 : 0; }
  m.coords_mode = c->opt.coords_mode;
  for (int i = 0; i < 6; i ++) m.coords_bounds[i] = c->opt.coords_bounds[i];
  for (int d = 0; d < 3; d ++) m.coords_rect[d] = c->d_rect[d];
  m.coords_expl = c->d_expl; m.coords_expl_ncomp = c->expl_ncomp; m.coords_expl_n0 = (int)c->expl_n0;
  m.hits = c->d_hits; m.pass = c->d_pass; m.counters = c->d_counters; m.capacity = c->capacity;
  m.fragile = c->d_fragile; m.fragile_capacity = c->fragile_capacity;
  m.u_rows = ftkx::mask_summary_rows(m);
}

int slice_resolution(ftkx_ctx *c, Slice &s)
{
  if (s.have_res) return FTKX_OK;
  u64 *d = c->d_counters + ftkx::CNT_N;
  u64 init[128];
  for (int i = 0; i < 64; i ++) { init[2 * i] = 0x7fefffffffffffffull; init[2 * i + 1] = 0ull; }
  HIP_TRY(c, hipMemcpyAsync(d, init, sizeof(init), hipMemcpyHostToDevice, c->stream));
  if (c->scalar_mode == 1) {
    Mesh m; fill_mesh(c, m);
    if (ftkx::march2_supported(m)) {
      // the marching stencil kernel in reduce-only mode: same single pass over S as the mask kernel, nothing stored
      int rc = ensure_desc(c, sizeof(MaskJob));
      if (rc) return rc;
      const MaskJob job{s.S, nullptr, nullptr, nullptr, d, 1.0};
      HIP_TRY(c, hipMemcpyAsync(c->d_desc, &job, sizeof(job), hipMemcpyHostToDevice, c->stream));
      ftkx::launch_reduce_march(m, (const MaskJob *)c->d_desc, 1, c->stream);
    } else ftkx::launch_resolution_scalar(m, s.S, d, c->stream);
  }
  else ftkx::launch_resolution(s.V, n_vertices(c) * (size_t)c->nd, d, c->stream);
  HIP_TRY(c, hipGetLastError());
  u64 out[128];
  HIP_TRY(c, hipMemcpyAsync(out, d, sizeof(out), hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  u64 mn = out[0], mx = out[1];
  for (int i = 1; i < 64; i ++) { mn = std::min(mn, out[2 * i]); mx = std::max(mx, out[2 * i + 1]); }
  memcpy(&s.res, &mn, 8);
  memcpy(&s.maxabs, &mx, 8);
  s.have_res = true;
  return FTKX_OK;
}

// the strict-sign cull is exact only while no determinant of the predicate can leave int64:
// |det4| <= 24 M^3 (3D), |det3| <= 6 M^2 (2D) with M = max |quantised component|   
bool overflow_free(int nd, double maxabs, u64 factor)
{
  const long double M = floorl((long double)maxabs * (long double)factor) + 1.0L;
  const long double lim = 9223372036854775807.0L;
  return nd == 3 ? 24.0L * M * M * M < lim : 6.0L * M * M < lim;
}

double big_threshold(int nd, u64 factor) { return (double)(nd == 3 ? ftkx::kSafeM3 : ftkx::kSafeM2) / (double)factor; }
bool pow2_factor(u64 factor) { return factor != 0 && (factor & (factor - 1)) == 0 && factor <= (1ull << 53); }

// Masks built under mask_factor serve a sweep under `factor` when they can only cull less than masks built under `factor`
// itself: the sign thresholds need mask_factor <= factor; the per-vertex overflow rule (MaskJob::big) is factor-specific, so a
// larger factor is accepted only when the slice's max |v| shows that no vertex is big under it either.
bool masks_valid(const ftkx_ctx *c, const Slice &s, u64 factor, bool two_level, int u_rows)
{
  if (!s.M || (two_level && !s.U) || s.mask_factor == 0 || s.mask_factor > factor) return false;
  if (two_level && s.u_rows != u_rows) return false;           // summaries of another geometry (FTKX_MASK_* changed since)
  if (s.mask_big && s.mask_factor == factor) return true;
  return s.max_known() && overflow_free(c->nd, s.maxabs, factor);   // no vertex is big under `factor`: the rule would change nothing
}

// the per-vertex rule costs the marching kernels a few instructions per row: it is switched on only when it can matter
double job_big(const ftkx_ctx *c, const Slice &s, u64 factor, bool *rule_on)
{
  const bool off = s.max_known() && overflow_free(c->nd, s.maxabs, factor);
  *rule_on = !off;
  return off ? HUGE_VAL : big_threshold(c->nd, factor);
}

// (host memory -> HBM: upload.cpp)

// ---- the library's auxiliary streams ----------------------------------------------------------------------------------------------
// A pass's tail (series.hip) and the copy of its records run on streams of the library's own.  They are kept for the PROCESS, per device and
// priority, and handed from a context that is destroyed to the next one that asks: what a stream is mapped to -- the hardware queue, its
// priority -- is decided by the runtime when the stream is made, and a process that makes and destroys contexts (bench.py's configurations,
// a test session) otherwise gets a different mapping for every context (round 5: the split pass of hit-dense data ran at 0.78 ms in a
// fresh process and at 0.93-0.97 behind other contexts).  FTKX_STREAM_POOL=0: streams made and destroyed with the context, as before.
namespace {
std::mutex g_aux_mutex;
std::map<std::pair<int, int>, std::vector<hipStream_t>> g_aux_free;      // (device, high priority?) -> idle streams
bool aux_pool_on() { const char *e = getenv("FTKX_STREAM_POOL"); return !e || atoi(e) != 0; }
}
int aux_stream_get(ftkx_ctx *c, bool high, hipStream_t *out)
{
  if (aux_pool_on()) {
    std::lock_guard<std::mutex> g(g_aux_mutex);
    auto &v = g_aux_free[{c->device, high ? 1 : 0}];
    if (!v.empty()) { *out = v.back(); v.pop_back(); return FTKX_OK; }
  }
  if (high) {
    int lo = 0, hi = 0;
    (void)hipDeviceGetStreamPriorityRange(&lo, &hi);
    HIP_TRY(c, hipStreamCreateWithPriority(out, hipStreamNonBlocking, hi));
  } else HIP_TRY(c, hipStreamCreateWithFlags(out, hipStreamNonBlocking));
  return FTKX_OK;
}
void aux_stream_put(ftkx_ctx *c, bool high, hipStream_t st)
{
  if (!st) return;
  (void)hipStreamSynchronize(st);
  if (!aux_pool_on()) { (void)hipStreamDestroy(st); return; }
  std::lock_guard<std::mutex> g(g_aux_mutex);
  g_aux_free[{c->device, high ? 1 : 0}].push_back(st);
}

int ensure_mask_arrays(ftkx_ctx *c, Slice &s, bool two_level)
{
  s.mask_gen = ++ c->mask_epoch;    // (every builder of masks comes through here: a series pass collected later leaves this slice's marks alone)
  if (!s.M) {
    if (!c->pool_M.empty()) { s.M = c->pool_M.back(); c->pool_M.pop_back(); }   // padding still neutral from its first life
    else {
      HIP_TRY(c, hipMalloc((void **)&s.M, mask_bytes(c)));
      // row padding and anything a kernel does not write is cull-neutral
      HIP_TRY(c, hipMemsetAsync(s.M, 0x3f, mask_bytes(c), c->stream));
    }
    s.mask_factor = 0;
  }
  if (two_level && !s.U) {
    if (!c->pool_U.empty()) { s.U = c->pool_U.back(); c->pool_U.pop_back(); }
    else {
      HIP_TRY(c, hipMalloc((void **)&s.U, u_bytes(c)));
      HIP_TRY(c, hipMemsetAsync(s.U, 0x3f, u_bytes(c), c->stream));
    }
    s.mask_factor = 0;      // summaries must be produced together with the masks
  }
  return FTKX_OK;
}

}  // namespace ftkxh

extern "C" {

const char *ftkx_last_mask_kernel(void) { return ftkx::last_mask_kernel(); }
int ftkx_debug_mask_kernel_launches(unsigned long long *launches, const char **names, int n)
{
  unsigned long long l[ftkx::kMaskKernels]; const char *nm[ftkx::kMaskKernels];
  ftkx::mask_kernel_launches(l, nm);
  for (int i = 0; i < n && i < ftkx::kMaskKernels; i ++) { if (launches) launches[i] = l[i]; if (names) names[i] = nm[i]; }
  return ftkx::kMaskKernels;
}

const char *ftkx_version(void) { return "ftkx 0.1 (gfx950)"; }

// which device a pointer lives on: its ordinal, or -1 for host memory / unknown pointers
int ftkx_pointer_device(const void *p)
{
  hipPointerAttribute_t a;
  if (!p || hipPointerGetAttributes(&a, p) != hipSuccess) { (void)hipGetLastError(); return -1; }
  return a.type == hipMemoryTypeDevice ? a.device : -1;
}

int ftkx_context_device(const ftkx_ctx *c) { return c ? c->device : -1; }

int ftkx_device_count(void)
{
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

void ftkx_default_options(ftkx_options *o)
{
  memset(o, 0, sizeof(*o));
  o->jacobian_symmetric = 1;
  o->robust = 1;
  o->tag_mode = FTKX_TAG_EXACT64;
}

int ftkx_last_error(const ftkx_ctx *ctx, char *buf, size_t n)
{
  const std::string &e = ctx ? ctx->err : g_last_error;
  if (buf && n) { strncpy(buf, e.c_str(), n - 1); buf[n - 1] = 0; }
  return (int)e.size();
}

int ftkx_create(ftkx_ctx **out, int nd, int device_id)
{
  if (!out || (nd != 2 && nd != 3)) return fail(nullptr, FTKX_E_INVALID, "ftkx_create: nd must be 2 or 3");
  int ndev = 0;
  hipError_t e = hipGetDeviceCount(&ndev);
  if (e != hipSuccess || ndev == 0) return fail(nullptr, FTKX_E_DEVICE, "ftkx_create: no HIP device (%s)", hipGetErrorString(e));
  if (device_id < 0 || device_id >= ndev) return fail(nullptr, FTKX_E_INVALID, "ftkx_create: device %d of %d", device_id, ndev);
  ftkx_ctx *c = new ftkx_ctx();
  c->nd = nd;
  c->device = device_id;
  ftkx_default_options(&c->opt);
  memset(&c->stats, 0, sizeof(c->stats));
  // a blocking stream: it orders itself against the legacy default stream, which is where a caller that never heard of
  // streams (and torch's default stream) puts its copies and fills
  auto init = [&]() -> int {
    HIP_TRY(c, hipSetDevice(device_id));
    HIP_TRY(c, hipStreamCreateWithFlags(&c->own_stream, hipStreamDefault));
    c->stream = c->own_stream;
    // CNT_N counters, 128 words of reduction slots, one word that says "a halo message did not fit this mesh" (halo.hip; survives the sweeps' resets)
    HIP_TRY(c, hipMalloc((void **)&c->d_counters, (ftkx::CNT_N + 128 + 8) * sizeof(u64)));
    HIP_TRY(c, hipMemset(c->d_counters, 0, (ftkx::CNT_N + 128 + 8) * sizeof(u64)));
    HIP_TRY(c, hipHostMalloc((void **)&c->h_counters, ftkx::CNT_N * sizeof(u64), hipHostMallocDefault));
    return FTKX_OK;
  };
  const int rc = init();
  if (rc != FTKX_OK) { ftkx_destroy(c); return rc; }     // nothing half-built is left behind
  *out = c;
  return FTKX_OK;
}

void ftkx_destroy(ftkx_ctx *c)
{
  if (!c) return;
  (void)hipSetDevice(c->device);
  if (c->stream) (void)hipStreamSynchronize(c->stream);
  for (hipStream_t st : {c->sr_tail_stream, c->sr_tail_stream2, c->sr_copy_stream}) if (st) (void)hipStreamSynchronize(st);      // (passes left open: their tails and copies read what is freed below)
  for (auto &kv : c->slices) free_slice(kv.second);
  for (ftkx_series_pending &P : c->sr_pend) { for (Slice &sl : P.parked) free_slice(sl); P.parked.clear(); }
  release_pools(c);
  for (auto &e : c->events) { (void)hipEventDestroy(e.second.first); (void)hipEventDestroy(e.second.second); }
  for (hipEvent_t e : c->event_pool) (void)hipEventDestroy(e);
  if (c->h_red) (void)hipHostFree(c->h_red);
  if (c->h_ahead) (void)hipHostFree(c->h_ahead);
  if (c->d_ahead) (void)hipFree(c->d_ahead);
  if (c->d_red) (void)hipFree(c->d_red);
  if (c->d_tile_stats) (void)hipFree(c->d_tile_stats);
  if (c->d_hits) (void)hipFree(c->d_hits);
  if (c->d_pass) (void)hipFree(c->d_pass);
  if (c->d_fragile) (void)hipFree(c->d_fragile);
  for (void *p : {(void *)c->d_word_idx, (void *)c->d_words, (void *)c->d_cells, (void *)c->d_patch_cells, (void *)c->d_patches, c->d_packed}) if (p) (void)hipFree(p);
  for (void *p : {(void *)c->sr_hist, (void *)c->sr_boff, (void *)c->sr_bucketed, (void *)c->sr_sorted}) if (p) (void)hipFree(p);
  for (ftkx_series_buffers &B : c->sr_buf) {
    for (void *p : {(void *)B.results, (void *)B.d_out, B.d_desc, (void *)B.copy_done, (void *)B.dist_block}) if (p) (void)hipFree(p);
    if (B.ev_copied) (void)hipEventDestroy(B.ev_copied);
    if (B.ev_export) (void)hipEventDestroy(B.ev_export);
    for (hipEvent_t e : {B.ev_masks, B.ev_factors, B.ev_tail}) if (e) (void)hipEventDestroy(e);
    if (B.red) (void)hipFree(B.red);
    for (void *p : {(void *)B.h_results, (void *)B.out, B.h_desc}) if (p) (void)hipHostFree(p);
  }
  aux_stream_put(c, false, c->sr_copy_stream);
  aux_stream_put(c, true, c->sr_tail_stream);
  aux_stream_put(c, true, c->sr_tail_stream2);
  for (void *q : {(void *)c->sr_set1.counters, (void *)c->sr_set1.list, (void *)c->sr_set1.refine, (void *)c->sr_set1.pass, (void *)c->sr_set1.fragile,
                  (void *)c->sr_set1.bucketed, (void *)c->sr_set1.sorted, (void *)c->sr_set1.hist, (void *)c->sr_set1.boff}) if (q) (void)hipFree(q);
  if (c->sr_one_scratch) (void)hipFree(c->sr_one_scratch);
  if (c->sr_fetch_flag) (void)hipFree(c->sr_fetch_flag);
  if (c->sr_ev_fetched) (void)hipEventDestroy(c->sr_ev_fetched);
  if (c->sr_fetch_stream) (void)hipStreamDestroy(c->sr_fetch_stream);
  for (void *p : {c->tr_dev, c->tr_parent, c->tr_tables}) if (p) (void)hipFree(p);
  if (c->tr_host) (void)hipHostFree(c->tr_host);
  if (c->d_list) (void)hipFree(c->d_list);
  if (c->d_refine) (void)hipFree(c->d_refine);
  if (c->d_sorted) (void)hipFree(c->d_sorted);
  if (c->d_keys) (void)hipFree(c->d_keys);
  if (c->d_idx) (void)hipFree(c->d_idx);
  if (c->d_sort_tmp) (void)hipFree(c->d_sort_tmp);
  if (c->d_desc) (void)hipFree(c->d_desc);
  for (int d = 0; d < 3; d ++) if (c->d_rect[d]) (void)hipFree(c->d_rect[d]);
  if (c->d_expl) (void)hipFree(c->d_expl);
  if (c->h_desc) (void)hipHostFree(c->h_desc);
  if (c->d_counters) (void)hipFree(c->d_counters);
  if (c->h_counters) (void)hipHostFree(c->h_counters);
  if (c->h_hits) (void)hipHostFree(c->h_hits);
  if (c->own_stream) (void)hipStreamDestroy(c->own_stream);
  delete c;
}

int ftkx_set_stream(ftkx_ctx *c, void *s)
{
  if (c) c->ahead.clear();
  if (!c) return fail(nullptr, FTKX_E_INVALID, "null context");
  if (!c->pending.empty()) return fail(c, FTKX_E_INVALID, "ftkx_set_stream: sweeps pending, collect first");
  if (c->sr_open && !c->sr_internal) return fail(c, FTKX_E_INVALID, "ftkx_set_stream: series passes open (ftkx_sweep_series_submit), complete them first");
  c->stream = s ? (hipStream_t)s : c->own_stream;
  return FTKX_OK;
}

int ftkx_set_options(ftkx_ctx *c, const ftkx_options *o)
{
  if (c) c->ahead.clear();
  if (!c || !o) return fail(c, FTKX_E_INVALID, "null argument");
  if (!c->pending.empty()) return fail(c, FTKX_E_INVALID, "ftkx_set_options: sweeps pending, collect first");
  if (c->sr_open && !c->sr_internal) return fail(c, FTKX_E_INVALID, "ftkx_set_options: series passes open (ftkx_sweep_series_submit), complete them first");
  if (o->tag_mode < FTKX_TAG_WORK_INDEX || o->tag_mode > FTKX_TAG_EXACT64) return fail(c, FTKX_E_INVALID, "bad tag_mode %d", o->tag_mode);
  if (o->coords_mode < 0 || o->coords_mode > 3) return fail(c, FTKX_E_INVALID, "bad coords_mode %d", o->coords_mode);
  if (o->coords_mode == 2 && !c->d_rect[0]) return fail(c, FTKX_E_INVALID, "coords_mode RECTILINEAR: call ftkx_set_coords_rectilinear");
  if (o->coords_mode == 3 && !c->d_expl) return fail(c, FTKX_E_INVALID, "coords_mode EXPLICIT: call ftkx_set_coords_explicit");
  c->opt = *o;
  return FTKX_OK;
}

int ftkx_set_coords_rectilinear(ftkx_ctx *c, const double *x, size_t nx, const double *y, size_t ny, const double *z, size_t nz)
{
  if (!c) return fail(nullptr, FTKX_E_INVALID, "null context");
  if (!c->pending.empty()) return fail(c, FTKX_E_INVALID, "ftkx_set_coords_rectilinear: sweeps pending, collect first");
  const double *src[3] = {x, y, z};
  const size_t n[3] = {nx, ny, nz};
  for (int d = 0; d < c->nd; d ++) if (!src[d] || !n[d]) return fail(c, FTKX_E_INVALID, "ftkx_set_coords_rectilinear: axis %d missing", d);
  HIP_TRY(c, hipSetDevice(c->device));
  for (int d = 0; d < c->nd; d ++) {
    if (c->d_rect[d]) { (void)hipFree(c->d_rect[d]); c->d_rect[d] = nullptr; }
    HIP_TRY(c, hipMalloc((void **)&c->d_rect[d], n[d] * sizeof(double)));
    HIP_TRY(c, hipMemcpy(c->d_rect[d], src[d], n[d] * sizeof(double), hipMemcpyHostToDevice));
    c->rect_n[d] = n[d];
  }
  c->opt.coords_mode = 2;
  return FTKX_OK;
}

int ftkx_set_coords_explicit(ftkx_ctx *c, const double *coords, int ncomp, size_t n0, size_t n1)
{
  if (!c || !coords) return fail(c, FTKX_E_INVALID, "null argument");
  if (!c->pending.empty()) return fail(c, FTKX_E_INVALID, "ftkx_set_coords_explicit: sweeps pending, collect first");
  if (ncomp < 2 || (c->nd == 3 && ncomp < 3) || !n0 || !n1) return fail(c, FTKX_E_INVALID, "ftkx_set_coords_explicit: need %d components and a non-empty array", c->nd == 3 ? 3 : 2);
  HIP_TRY(c, hipSetDevice(c->device));
  if (c->d_expl) { (void)hipFree(c->d_expl); c->d_expl = nullptr; }
  const size_t count = (size_t)ncomp * n0 * n1;
  HIP_TRY(c, hipMalloc((void **)&c->d_expl, count * sizeof(double)));
  HIP_TRY(c, hipMemcpy(c->d_expl, coords, count * sizeof(double), hipMemcpyHostToDevice));
  c->expl_ncomp = ncomp; c->expl_n0 = n0; c->expl_n1 = n1;
  c->opt.coords_mode = 3;
  return FTKX_OK;
}

int ftkx_set_mesh(ftkx_ctx *c, const long long dst[3], const long long dsz[3], const long long cst[3], const long long csz[3],
                  const long long est[3], const long long esz[3])
{
  if (c) c->ahead.clear();
  if (!c) return fail(nullptr, FTKX_E_INVALID, "null context");
  if (!c->slices.empty()) return fail(c, FTKX_E_INVALID, "ftkx_set_mesh: drop all slices first");
  (void)hipSetDevice(c->device);
  release_pools(c);                       // pooled mask arrays have the old lattice's size
  for (int d = 0; d < c->nd; d ++) {
    if (dsz[d] < 0 || csz[d] < 0 || esz[d] <= 0) return fail(c, FTKX_E_INVALID, "ftkx_set_mesh: negative size on axis %d", d);
    if (dst[d] + dsz[d] > 2147483647LL || est[d] + esz[d] > 2147483647LL || cst[d] + csz[d] > 2147483647LL)
      return fail(c, FTKX_E_INVALID, "ftkx_set_mesh: axis %d exceeds int range", d);
    // every corner enumerated must be addressable: the kernel reads a vertex only when it is inside domain AND ext
    c->dom_st[d] = dst[d]; c->dom_sz[d] = dsz[d];
    c->core_st[d] = cst[d]; c->core_sz[d] = csz[d];
    c->ext_st[d] = est[d]; c->ext_sz[d] = esz[d];
  }
  for (int d = c->nd; d < 3; d ++) { c->dom_st[d] = 0; c->dom_sz[d] = 1; c->core_st[d] = 0; c->core_sz[d] = 1; c->ext_st[d] = 0; c->ext_sz[d] = 1; }
  c->mesh_set = true;
  c->scalar_mode = -1;
  c->dense_collects = 0;
  return FTKX_OK;
}

static int push_common(ftkx_ctx *c, int t, const double *V, const double *J, const double *S, int on_device, bool scalar_only)
{
  if (c) c->ahead.clear();
  if (!c) return fail(nullptr, FTKX_E_INVALID, "null context");
  if (!c->mesh_set) return fail(c, FTKX_E_INVALID, "push: call ftkx_set_mesh first");
  if (t < 0) return fail(c, FTKX_E_INVALID, "push: negative timestep");
  if (on_device < 0 || on_device > 2) return fail(c, FTKX_E_INVALID, "push: on_device must be 0, 1 or 2");
  if (scalar_only ? !S : !V) return fail(c, FTKX_E_INVALID, "push: missing field pointer");
  if (!c->pending.empty()) return fail(c, FTKX_E_INVALID, "push: sweeps pending, collect first");
  if (c->slices.empty()) c->scalar_mode = -1;
  if (c->scalar_mode >= 0 && c->scalar_mode != (scalar_only ? 1 : 0))
    return fail(c, FTKX_E_INVALID, "push: scalar and vector slices cannot be mixed in one context");
  HIP_TRY(c, hipSetDevice(c->device));
  auto it = c->slices.find(t);
  if (it != c->slices.end()) { free_slice(it->second, c); c->slices.erase(it); }
  Slice s;
  const size_t n = n_vertices(c);
  const int nd = c->nd;
  auto take = [&](const double *src, size_t count, double **dst, bool *own) -> int {
    if (!src) { *dst = nullptr; *own = false; return FTKX_OK; }
    if (on_device == 1) { *dst = const_cast<double *>(src); *own = false; return FTKX_OK; }
    *dst = nullptr;
    for (size_t i = 0; i < c->pool_F.size(); i ++)
      if (c->pool_F[i].second == count) { *dst = c->pool_F[i].first; c->pool_F.erase(c->pool_F.begin() + (long)i); break; }
    if (!*dst) HIP_TRY(c, hipMalloc((void **)dst, count * sizeof(double)));
    *own = true;
    // 0: host memory; 2: device memory of ANY device (a multi-device tracker hands one snapshot to two contexts), copied
    if (on_device == 0) return upload_from_host(c, *dst, src, count * sizeof(double));
    HIP_TRY(c, hipMemcpyAsync(*dst, src, count * sizeof(double), hipMemcpyDefault, c->stream));
    return FTKX_OK;
  };
  int rc;
  if ((rc = take(S, n, &s.S, &s.ownS))) { free_slice(s); return rc; }
  if (!scalar_only) {
    if ((rc = take(V, n * nd, &s.V, &s.ownV))) { free_slice(s); return rc; }      // what was already allocated goes back
    if ((rc = take(J, n * nd * nd, &s.J, &s.ownJ))) { free_slice(s); return rc; }
  }
  // scalar input: V = gradient2D/3D(S) is never materialised -- every kernel evaluates it where it needs it, with the
  // reference's exact operations (ndarray/grad.hh), so the slice costs 8 bytes per vertex of HBM instead of 8 + 8*nd.
  // the source buffers may be reused by the caller on return: a device source (2) has to be read first; a host source has been staged
  // completely by upload_from_host (nothing to wait for: the DMAs run on while the caller produces its next snapshot)
  if (on_device == 2) HIP_TRY(c, hipStreamSynchronize(c->stream));
  s.mask_gen = ++ c->mask_epoch;
  c->slices[t] = s;
  c->scalar_mode = scalar_only ? 1 : 0;
  return FTKX_OK;
}

int ftkx_push_slice(ftkx_ctx *c, int t, const double *V, const double *J, const double *S, int on_device)
{ return push_common(c, t, V, J, S, on_device, false); }

int ftkx_push_scalar_slice(ftkx_ctx *c, int t, const double *S, int on_device)
{ return push_common(c, t, nullptr, nullptr, S, on_device, true); }

int ftkx_drop_slice(ftkx_ctx *c, int t)
{
  if (c) c->ahead.clear();
  if (!c) return fail(nullptr, FTKX_E_INVALID, "null context");
  auto it = c->slices.find(t);
  if (it == c->slices.end()) return fail(c, FTKX_E_NOSLICE, "ftkx_drop_slice: timestep %d not resident", t);
  if (!c->pending.empty()) return fail(c, FTKX_E_INVALID, "ftkx_drop_slice: sweeps pending, collect first");
  (void)hipSetDevice(c->device);
  free_slice(it->second, c);
  c->slices.erase(it);
  return FTKX_OK;
}

int ftkx_slice_resolution(ftkx_ctx *c, int t, double *res, double *max_abs)
{
  if (!c) return fail(nullptr, FTKX_E_INVALID, "null context");
  auto it = c->slices.find(t);
  if (it == c->slices.end()) return fail(c, FTKX_E_NOSLICE, "ftkx_slice_resolution: timestep %d not resident", t);
  HIP_TRY(c, hipSetDevice(c->device));
  int rc = slice_resolution(c, it->second);
  if (rc) return rc;
  if (res) *res = it->second.res;
  if (max_abs) *max_abs = it->second.maxabs;
  return FTKX_OK;
}

// the same reduction for several slices at once: one launch, one download, one synchronise (a time series that is already
// resident does not need a round trip per slice)
int ftkx_slices_resolution(ftkx_ctx *c, const int *ts, int n, double *res, double *max_abs)
{
  if (!c || (n > 0 && !ts)) return fail(c, FTKX_E_INVALID, "null argument");
  HIP_TRY(c, hipSetDevice(c->device));
  std::vector<Slice *> todo;
  for (int i = 0; i < n; i ++) {
    auto it = c->slices.find(ts[i]);
    if (it == c->slices.end()) return fail(c, FTKX_E_NOSLICE, "ftkx_slices_resolution: timestep %d not resident", ts[i]);
    if (!it->second.have_res) todo.push_back(&it->second);
  }
  Mesh m; fill_mesh(c, m);
  if (todo.size() > 1) {
    const size_t k = todo.size();
    if (c->red_cap < k) {
      if (c->d_red) { (void)hipFree(c->d_red); c->d_red = nullptr; c->red_cap = 0; }
      HIP_TRY(c, hipMalloc((void **)&c->d_red, k * 128 * sizeof(u64)));
      c->red_cap = k;
    }
    // descriptors and results share the pinned staging buffer (stream order: upload, kernel, download)
    int rc = ensure_desc(c, std::max(k * sizeof(MaskJob), k * 128 * sizeof(u64)));
    if (rc) return rc;
    launch_init_red(c->d_red, k * 64, nullptr, c->stream);
    if (c->scalar_mode == 1 && ftkx::march2_supported(m)) {
      // the marching stencil kernel in reduce-only mode over all slices at once
      MaskJob *jobs = (MaskJob *)c->h_desc;
      for (size_t i = 0; i < k; i ++) jobs[i] = MaskJob{todo[i]->S, nullptr, nullptr, nullptr, c->d_red + i * 128, 1.0};
      HIP_TRY(c, hipMemcpyAsync(c->d_desc, c->h_desc, k * sizeof(MaskJob), hipMemcpyHostToDevice, c->stream));
      ftkx::launch_reduce_march(m, (const MaskJob *)c->d_desc, (int)k, c->stream);
    } else {
      // one launch per slice, back to back, each into its own slots
      for (size_t i = 0; i < k; i ++) {
        if (c->scalar_mode == 1) ftkx::launch_resolution_scalar(m, todo[i]->S, c->d_red + i * 128, c->stream);
        else ftkx::launch_resolution(todo[i]->V, n_vertices(c) * (size_t)c->nd, c->d_red + i * 128, c->stream);
      }
    }
    HIP_TRY(c, hipGetLastError());
    HIP_TRY(c, hipMemcpyAsync(c->h_desc, c->d_red, k * 128 * sizeof(u64), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    const u64 *host = (const u64 *)c->h_desc;
    for (size_t i = 0; i < k; i ++) {
      u64 mn = host[i * 128], mx = host[i * 128 + 1];
      for (int q = 1; q < 64; q ++) { mn = std::min(mn, host[i * 128 + 2 * q]); mx = std::max(mx, host[i * 128 + 2 * q + 1]); }
      memcpy(&todo[i]->res, &mn, 8);
      memcpy(&todo[i]->maxabs, &mx, 8);
      todo[i]->have_res = true;
    }
  } else {
    for (Slice *s : todo) { int rc = slice_resolution(c, *s); if (rc) return rc; }
  }
  for (int i = 0; i < n; i ++) {
    const Slice &s = c->slices.find(ts[i])->second;
    if (res) res[i] = s.res;
    if (max_abs) max_abs[i] = s.maxabs;
  }
  return FTKX_OK;
}

int ftkx_set_slice_resolution(ftkx_ctx *c, int t, double resolution, double max_abs)
{
  if (!c) return fail(nullptr, FTKX_E_INVALID, "null context");
  auto it = c->slices.find(t);
  if (it == c->slices.end()) return fail(c, FTKX_E_NOSLICE, "ftkx_set_slice_resolution: timestep %d not resident", t);
  if (!(resolution > 0) || !(max_abs >= 0)) return fail(c, FTKX_E_INVALID, "ftkx_set_slice_resolution: bad values");
  it->second.res = resolution; it->second.maxabs = max_abs; it->second.have_res = true;
  return FTKX_OK;
}

unsigned long long ftkx_scaling_factor(double resolution, int *nbits_out)
{
  // critical_point_tracker.hh:850-864
  int nbits = (int)std::ceil(std::log2(1.0 / resolution));
  nbits = std::max(8, std::min(nbits, 21));
  if (nbits_out) *nbits_out = nbits;
  return 1ull << nbits;
}

int ftkx_get_stats(const ftkx_ctx *c, ftkx_stats *st)
{
  if (!c || !st) return fail(nullptr, FTKX_E_INVALID, "null argument");
  *st = c->stats;
  return FTKX_OK;
}

int ftkx_invalidate_masks(ftkx_ctx *c)
{
  if (!c) return fail(nullptr, FTKX_E_INVALID, "null context");
  if (!c->pending.empty()) return fail(c, FTKX_E_INVALID, "ftkx_invalidate_masks: sweeps pending, collect first");
  for (auto &kv : c->slices) { kv.second.mask_factor = 0; kv.second.have_fused = false; kv.second.mask_gen = ++ c->mask_epoch; }
  c->ahead.clear();
  c->dense_collects = 0;
  return FTKX_OK;
}


// profiling aid (bench.py's int-VALU yardstick): from the next sweep on, the tile kernel runs its fan phase -- the predicate arithmetic on
// the tile staged in LDS -- `repeat` times per tile and step; records and statistics are those of one.  1 = off.
int ftkx_debug_tile_repeat(ftkx_ctx *c, int repeat)
{
  if (!c || repeat < 1) return fail(c, FTKX_E_INVALID, "ftkx_debug_tile_repeat: repeat >= 1");
  c->tile_repeat = repeat;
  return FTKX_OK;
}
int ftkx_debug_stream_read(ftkx_ctx *c, const void *device_ptr, size_t bytes)
{
  if (!c || !device_ptr) return fail(c, FTKX_E_INVALID, "null argument");
  HIP_TRY(c, hipSetDevice(c->device));
  ftkx::launch_calib_read(device_ptr, bytes, (double *)(c->d_counters + ftkx::CNT_N), c->stream);
  HIP_TRY(c, hipGetLastError());
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  return FTKX_OK;
}

int ftkx_set_profiling(ftkx_ctx *c, int on)
{
  if (!c) return fail(nullptr, FTKX_E_INVALID, "null context");
  c->profiling = on < 0 ? 0 : (on > 2 ? 1 : on);
  for (int k = 0; k < K_N; k ++) { c->k_ms[k] = 0; c->k_launches[k] = 0; }
  return FTKX_OK;
}

int ftkx_get_kernel_times(const ftkx_ctx *c, double ms[4], unsigned long long launches[4])
{
  if (!c || !ms || !launches) return fail(nullptr, FTKX_E_INVALID, "null argument");
  ev_harvest(const_cast<ftkx_ctx *>(c), false);          // (pairs that have completed since the last call that waited)
  for (int k = 0; k < K_N; k ++) { ms[k] = c->k_ms[k]; launches[k] = c->k_launches[k]; }
  return FTKX_OK;
}

void ftkx_free(void *p) { free(p); }

static int extract_common(int nd, int scope, int t, const long long *dst, const long long *dsz, const long long *cst, const long long *csz,
                          const long long *est, const long long *esz, const double *Vc, const double *Vn, const double *Jc, const double *Jn,
                          const double *Sc, const double *Sn, unsigned long long factor, const ftkx_options *opt, int device_id,
                          ftkx_cp_t **out, size_t *n_out, const double *explicit_coords = nullptr)
{
  if (!out || !n_out) return fail(nullptr, FTKX_E_INVALID, "extract: null output");
  *out = nullptr; *n_out = 0;
  if (scope != FTKX_SCOPE_ORDINAL && scope != FTKX_SCOPE_INTERVAL) return fail(nullptr, FTKX_E_INVALID, "extract: scope must be 1 or 2");
  if (!Vc || (scope == FTKX_SCOPE_INTERVAL && !Vn)) return fail(nullptr, FTKX_E_INVALID, "extract: missing vector field");
  // the time axis of core must be the single step `t` (element_for builds it that way, regular_tracker.hh:196-211)
  if (cst[nd] != t || csz[nd] != 1) return fail(nullptr, FTKX_E_INVALID, "extract: core must cover exactly timestep %d", t);
  if (dst[nd] != 0) return fail(nullptr, FTKX_E_UNSUPPORTED, "extract: domain must start at time 0");
  ftkx_ctx *c = nullptr;
  int rc = ftkx_create(&c, nd, device_id);
  if (rc) return rc;
  ftkx_options o;
  if (opt) o = *opt; else { ftkx_default_options(&o); o.tag_mode = FTKX_TAG_WORK_INDEX; }
  long long e3[3] = {est[0], est[1], nd == 3 ? est[2] : 0}, s3[3] = {esz[0], esz[1], nd == 3 ? esz[2] : 1};
  if (explicit_coords) {   // the boundary's `coords`: (2, DW, DH) doubles over `ext` (critical_point_tracer_2d_regular.cu:194-198)
    if (est[0] != 0 || est[1] != 0) { ftkx_destroy(c); return fail(nullptr, FTKX_E_UNSUPPORTED, "extract: explicit coordinates need an array lattice starting at 0"); }
    if ((rc = ftkx_set_coords_explicit(c, explicit_coords, 2, (size_t)esz[0], (size_t)esz[1]))) { g_last_error = c->err; ftkx_destroy(c); return rc; }
    o.coords_mode = 3;
  }
  if ((rc = ftkx_set_options(c, &o)) || (rc = ftkx_set_mesh(c, dst, dsz, cst, csz, e3, s3)) ||
      (rc = ftkx_push_slice(c, t, Vc, Jc, Sc, 0)) ||
      (scope == FTKX_SCOPE_INTERVAL && (rc = ftkx_push_slice(c, t + 1, Vn, Jn, Sn, 0)))) {
    g_last_error = c->err; ftkx_destroy(c); return rc;
  }
  const ftkx_cp_t *recs = nullptr; size_t n = 0;
  rc = ftkx_sweep(c, t, scope, factor, &recs, &n);
  if (rc) { g_last_error = c->err; ftkx_destroy(c); return rc; }
  ftkx_cp_t *copy = (ftkx_cp_t *)malloc((n ? n : 1) * sizeof(ftkx_cp_t));
  if (!copy) { ftkx_destroy(c); return fail(nullptr, FTKX_E_NOMEM, "extract: out of host memory"); }
  if (n) memcpy(copy, recs, n * sizeof(ftkx_cp_t));
  ftkx_destroy(c);
  *out = copy; *n_out = n;
  return FTKX_OK;
}

int ftkx_extract_cp2dt(int scope, int current_timestep, const long long domain_st[3], const long long domain_sz[3],
                       const long long core_st[3], const long long core_sz[3], const long long ext_st[2], const long long ext_sz[2],
                       const double *Vc, const double *Vn, const double *Jc, const double *Jn, const double *Sc, const double *Sn,
                       int use_explicit_coords, const double *coords, unsigned long long factor, const ftkx_options *opt, int device_id,
                       ftkx_cp_t **out, size_t *n_out)
{
  if (use_explicit_coords && !coords) return fail(nullptr, FTKX_E_INVALID, "extract: use_explicit_coords without coords");
  return extract_common(2, scope, current_timestep, domain_st, domain_sz, core_st, core_sz, ext_st, ext_sz, Vc, Vn, Jc, Jn, Sc, Sn, factor, opt, device_id, out, n_out,
                        use_explicit_coords ? coords : nullptr);
}

int ftkx_extract_cp3dt(int scope, int current_timestep, const long long domain_st[4], const long long domain_sz[4],
                       const long long core_st[4], const long long core_sz[4], const long long ext_st[3], const long long ext_sz[3],
                       const double *Vc, const double *Vn, const double *Jc, const double *Jn, const double *Sc, const double *Sn,
                       unsigned long long factor, const ftkx_options *opt, int device_id, ftkx_cp_t **out, size_t *n_out)
{
  return extract_common(3, scope, current_timestep, domain_st, domain_sz, core_st, core_sz, ext_st, ext_sz, Vc, Vn, Jc, Jn, Sc, Sn, factor, opt, device_id, out, n_out);
}

#define DERIVE_PROLOGUE(c) do { if (!(c)) return fail(nullptr, FTKX_E_INVALID, "null context"); HIP_TRY((c), hipSetDevice((c)->device)); } while (0)

int ftkx_gradient2D(ftkx_ctx *c, const double *S, int DW, int DH, double *V)
{ DERIVE_PROLOGUE(c); ftkx::launch_gradient2d(S, DW, DH, V, c->stream); HIP_TRY(c, hipGetLastError()); HIP_TRY(c, hipStreamSynchronize(c->stream)); return FTKX_OK; }
int ftkx_jacobian2D(ftkx_ctx *c, const double *V, int DW, int DH, int symmetric, double *J)
{ DERIVE_PROLOGUE(c); ftkx::launch_jacobian2d(V, DW, DH, symmetric, J, c->stream); HIP_TRY(c, hipGetLastError()); HIP_TRY(c, hipStreamSynchronize(c->stream)); return FTKX_OK; }
int ftkx_gradient3D(ftkx_ctx *c, const double *S, int DW, int DH, int DD, double *V)
{ DERIVE_PROLOGUE(c); ftkx::launch_gradient3d(S, DW, DH, DD, V, c->stream); HIP_TRY(c, hipGetLastError()); HIP_TRY(c, hipStreamSynchronize(c->stream)); return FTKX_OK; }
int ftkx_jacobian3D(ftkx_ctx *c, const double *V, int DW, int DH, int DD, double *J)
{ DERIVE_PROLOGUE(c); ftkx::launch_jacobian3d(V, DW, DH, DD, J, c->stream); HIP_TRY(c, hipGetLastError()); HIP_TRY(c, hipStreamSynchronize(c->stream)); return FTKX_OK; }

}  // extern "C"
