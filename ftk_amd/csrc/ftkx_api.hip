// Host side of the C ABI declared in include/ftkx.h: context, HBM-resident slices, launch bookkeeping, hit download.
// The reference's counterpart is the per-call host wrapper extract_cp2dt<scope> / extract_cp3dt<scope>
// (src/filters/critical_point_tracer_2d_regular.cu:168-272, ..._3d_regular.cu:144-250), which re-allocates, re-uploads and
// frees everything on every call and synchronises the whole device; here slices stay resident, launches go to a stream,
// and the hit buffer is persistent (grown and the batch replayed if a launch overflows it).
#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>

#include <algorithm>
#include <chrono>
#include <cfloat>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <string>
#include <vector>

#include "sweep_params.hpp"
#include "cp_device.hpp"   // classify3 on the HOST (fragile 3D records, see there)
#include "internal.hpp"

namespace ftkx {
void launch_tile(const TileParams &p, hipStream_t stream);
void tile_dims(int nd, int tile[3]);
void launch_masks(const Mesh &m, const MaskJob *d_jobs, int njobs, hipStream_t stream);
void launch_cull(const Mesh &m, const Fields *d_steps, int nsteps, u64 *d_list, u64 cap, hipStream_t stream);
void launch_exact(const Mesh &m, const Fields *d_steps, int step_base, const u64 *d_list, u64 cap, hipStream_t stream);
void launch_records(const Mesh &m, const Fields *d_fields, hipStream_t stream);
void launch_compact_words(const Mesh &m, const unsigned char *U, const unsigned char *M, unsigned *idx, u64 *words, u64 capacity, u64 *counter, hipStream_t st);
void launch_scatter_words(const unsigned *idx, const u64 *words, size_t n, unsigned char *M, hipStream_t st);
void launch_sparse_cells(const Mesh &m, const Fields *d_steps, const u64 *d_list, u64 cap, const double *sparse, u64 *cells, u64 cells_cap, hipStream_t st);
void launch_patches(const Mesh &m, bool scatter, const u64 *cells, size_t n, int ncomp, double *field, double *patches, hipStream_t st);
bool masks_have_summary(const Mesh &m);
int mask_summary_rows(const Mesh &m);
bool march2_supported(const Mesh &m);
bool masks_fuse_reduction(const Mesh &m);
void launch_reduce_march(const Mesh &m, const MaskJob *d_jobs, int njobs, hipStream_t stream);
void launch_cull_two_level(const Mesh &m, const Fields *d_steps, int nsteps, u64 *d_refine, u64 refine_cap, u64 *d_list, u64 cap, hipStream_t stream);
void launch_resolution_scalar(const Mesh &m, const double *S, u64 *out2, hipStream_t stream);
void launch_gradient2d(const double *S, int DW, int DH, double *V, hipStream_t st);
void launch_jacobian2d(const double *V, int DW, int DH, int symmetric, double *J, hipStream_t st);
void launch_gradient3d(const double *S, int DW, int DH, int DD, double *V, hipStream_t st);
void launch_jacobian3d(const double *V, int DW, int DH, int DD, double *J, hipStream_t st);
void launch_resolution(const double *p, size_t n, u64 *out2, hipStream_t st);
void launch_calib_read(const void *p, size_t bytes, double *scratch, hipStream_t stream);
const char *last_mask_kernel();
}  // namespace ftkx

using ftkx::Fields;
using ftkx::MaskJob;
using ftkx::Mesh;
using ftkx::TileParams;
using ftkx::u64;

namespace {

thread_local std::string g_last_error;

struct Slice {
  double *V = nullptr, *J = nullptr, *S = nullptr;
  unsigned char *M = nullptr;       // vertex sign masks, built lazily for `mask_factor`
  unsigned char *U = nullptr;       // per-8-vertex summaries of M (two-level cull)
  bool ownV = false, ownJ = false, ownS = false;
  unsigned long long mask_factor = 0;   // the (power-of-two) factor M / U were built under; 0 = not built
  int u_rows = 1;                   // rows a byte of U stands for (Mesh::u_rows at the time the masks were built)
  bool mask_big = false;            // built with the per-vertex overflow rule of that factor (MaskJob::big finite)
  bool have_res = false;            // res = ndarray::resolution() of the slice's vector field (exact pre-pass), maxabs with it
  double res = 0, maxabs = 0;
  bool have_fused = false;          // reduction fused into the mask pass (ftkx_slices_prepare): maxabs, and
  double res_below = 0;             //   the smallest non-zero |v| below 1 / fused_factor (DBL_MAX if none)
  unsigned long long fused_factor = 0;
  bool sparse = false;              // a halo slice that exists as masks only: its field array holds just the patches scattered into it
  bool max_known() const { return have_res || have_fused || sparse; }
};

// how a request is swept: MODE_TILE tests every simplex (exact_only, non-robust 3D, odd factors); MODE_FAST = masks -> cull ->
// survivor list -> exact kernel; MODE_TILE_CULL = the tile kernel with its in-tile cull (same per-vertex legality rule), for
// data on which most cells survive the cull anyway (the int64-overflow regime: a survivor list would be as large as the input)
enum { MODE_TILE = 0, MODE_FAST = 1, MODE_TILE_CULL = 2 };
struct Request { int t, scope; unsigned long long factor; int mode; };

enum { K_MASK = 0, K_CULL = 1, K_EXACT = 2, K_TILE = 3, K_N = 4 };

}  // namespace

struct ftkx_ctx {
  int nd = 0, device = 0;
  hipStream_t own_stream = nullptr, stream = nullptr;
  ftkx_options opt;
  long long dom_st[3] = {0, 0, 0}, dom_sz[3] = {1, 1, 1}, core_st[3] = {0, 0, 0}, core_sz[3] = {1, 1, 1}, ext_st[3] = {0, 0, 0}, ext_sz[3] = {1, 1, 1};
  bool mesh_set = false;
  int scalar_mode = -1;             // -1 undecided, 0 vector slices, 1 scalar slices (V = gradient(S) evaluated in flight)
  std::map<int, Slice> slices;
  ftkx_cp_t *d_hits = nullptr;
  u64 *d_pass = nullptr;            // simplices that passed the integer test, awaiting the record kernel (same capacity)
  u64 *d_fragile = nullptr;         // 3D records to be re-classified on the host (slot, J[9]): cp_device.hpp, classify3
  u64 fragile_capacity = 0;
  u64 capacity = 0;
  u64 *d_list = nullptr;            // surviving corners of the fast path
  u64 list_capacity = 0;
  u64 *d_refine = nullptr;          // words the summary level could not rule out (two-level cull)
  u64 refine_capacity = 0;
  u64 *d_counters = nullptr;        // CNT_N counters + 128 words (64 {min, max} slots) for the resolution reduction
  u64 *h_counters = nullptr;        // pinned
  ftkx_cp_t *h_hits = nullptr;      // pinned
  size_t h_cap = 0;
  // device-side ordering of the hit records by tag (radix sort of (tag, index) pairs + one gather)
  ftkx_cp_t *d_sorted = nullptr;
  u64 *d_keys = nullptr;            // 2 * sort_cap
  unsigned *d_idx = nullptr;        // 2 * sort_cap
  void *d_sort_tmp = nullptr;
  size_t sort_cap = 0, sort_tmp_bytes = 0;
  // per-batch descriptors: pinned staging + device copies
  void *h_desc = nullptr, *d_desc = nullptr;
  size_t desc_cap = 0;
  // mask / summary arrays of dropped slices, kept for the next slice (a streaming tracker pushes and pops one slice per step:
  // hipMalloc + hipFree per step cost more than the sweep itself).  Their padding bytes stay valid: kernels never write them.
  std::vector<unsigned char *> pool_M, pool_U;
  std::vector<std::pair<double *, size_t>> pool_F;   // owned field arrays (S / V / J copies) of dropped slices, by size in doubles
  u64 *d_red = nullptr;             // {min, max} slots of a batched resolution reduction: 128 words per slice
  size_t red_cap = 0;
  // physical coordinates (REGULAR_COORDS_RECTILINEAR / _EXPLICIT): device copies
  double *d_rect[3] = {nullptr, nullptr, nullptr};
  size_t rect_n[3] = {0, 0, 0};
  double *d_expl = nullptr;
  int expl_ncomp = 0;
  size_t expl_n0 = 0, expl_n1 = 0;
  std::vector<Request> pending;
  // Cull-ahead: the sweeps the caller announced (ftkx_sweep_announce) for the slices of the next ftkx_slices_prepare, and -- once that
  // call has queued their cull right behind the mask kernel -- the survivor list it left on the device.  The cull needs the masks
  // and the list of steps, not the factor: it runs while the host still waits for the reduction, forms the factors and queues the
  // sweeps.  ftkx_sweep_collect takes the list over if the pending sweeps are exactly the announced ones and every mask serves its
  // factor; anything else (and any call that touches slices or masks in between) drops it.
  std::vector<std::pair<int, int>> announced;
  struct AheadStep { int t, scope; const unsigned char *M[2], *U[2]; };
  std::vector<AheadStep> ahead;     // non-empty: survivor list + counters on the device belong to these steps
  void *h_ahead = nullptr, *d_ahead = nullptr;   // the cull-ahead's own descriptors: pinned staging (read by fetch_desc_kernel) + device copy
  size_t ahead_cap = 0;
  bool ahead_staged = false;        // a fetch out of h_ahead may still be queued (cleared by every full stream synchronise of collect)
  u64 *h_red = nullptr;             // coherent pinned copy of the reduction slots + one flag word, written by readback_kernel
  size_t h_red_cap = 0;             //   (slots it can hold; the flag lives behind them)
  unsigned red_seq = 0;
  int dense_collects = 0;           // > 0: the last fast collect found most cells surviving; fast requests run MODE_TILE_CULL for a while
  // compact halo: the compacted mask words of the last ftkx_export_masks_size, the surviving cells of the last ftkx_sweep_cull
  unsigned *d_word_idx = nullptr; u64 *d_words = nullptr; size_t words_cap = 0, n_words = 0; int words_t = -1;
  u64 *d_cells = nullptr; size_t cells_cap = 0, n_cells = 0;
  u64 *d_patch_cells = nullptr; double *d_patches = nullptr; size_t patch_cap = 0;   // staging for host-side callers
  ftkx_stats stats;
  // optional kernel timing (hipEvents on the context's stream)
  int profiling = 0;
  std::vector<std::pair<int, std::pair<hipEvent_t, hipEvent_t>>> events;
  std::vector<hipEvent_t> event_pool;
  double k_ms[K_N] = {0, 0, 0, 0};
  unsigned long long k_launches[K_N] = {0, 0, 0, 0};
  std::string err;
};

namespace ftkx { void set_global_error(const char *msg) { g_last_error = msg ? msg : ""; } }

namespace {

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

#define HIP_TRY(c, call)                                                                                   \
  do {                                                                                                     \
    hipError_t e_ = (call);                                                                                \
    if (e_ != hipSuccess) return fail((c), e_ == hipErrorOutOfMemory ? FTKX_E_NOMEM : FTKX_E_DEVICE,       \
                                      "%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), __FILE__, __LINE__); \
  } while (0)

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

void free_slice(Slice &s, ftkx_ctx *pool_owner = nullptr)
{
  // owned copies go back to the context's pool: a streaming caller pushes and pops one slice per step, and hipMalloc + hipFree of a
  // slice-sized array cost more than sweeping a 256^3 slice
  auto give_back = [&](double *p, size_t count) {
    if (pool_owner && pool_owner->pool_F.size() < 6) pool_owner->pool_F.push_back({p, count});
    else (void)hipFree(p);
  };
  size_t nv = 0;
  if (pool_owner) { nv = 1; for (int d = 0; d < pool_owner->nd; d ++) nv *= (size_t)pool_owner->ext_sz[d]; }
  const size_t nd_ = pool_owner ? (size_t)pool_owner->nd : 0;
  if (s.ownV && s.V) give_back(s.V, nv * nd_);
  if (s.ownJ && s.J) give_back(s.J, nv * nd_ * nd_);
  if (s.ownS && s.S) give_back(s.S, nv);
  s.ownV = s.ownJ = s.ownS = false;
  if (s.M) { if (pool_owner && pool_owner->pool_M.size() < 4) pool_owner->pool_M.push_back(s.M); else (void)hipFree(s.M); }
  if (s.U) { if (pool_owner && pool_owner->pool_U.size() < 4) pool_owner->pool_U.push_back(s.U); else (void)hipFree(s.U); }
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

int ensure_mask_arrays(ftkx_ctx *c, Slice &s, bool two_level)
{
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

__global__ void init_red_kernel(u64 *red, size_t nslots, u64 *counters = nullptr)
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

// the reference keeps hits in a std::map ordered by element; device append order is arbitrary
int sort_hits_on_device(ftkx_ctx *c, size_t n, int key_bits)
{
  if (c->sort_cap < n) {
    for (void *p : {(void *)c->d_sorted, (void *)c->d_keys, (void *)c->d_idx, c->d_sort_tmp}) if (p) (void)hipFree(p);
    c->d_sorted = nullptr; c->d_keys = nullptr; c->d_idx = nullptr; c->d_sort_tmp = nullptr; c->sort_cap = 0;
    const size_t cap = n + n / 4 + 1024;
    HIP_TRY(c, hipMalloc((void **)&c->d_sorted, cap * sizeof(ftkx_cp_t)));
    HIP_TRY(c, hipMalloc((void **)&c->d_keys, 2 * cap * sizeof(u64)));
    HIP_TRY(c, hipMalloc((void **)&c->d_idx, 2 * cap * sizeof(unsigned)));
    size_t tmp = 0;
    HIP_TRY(c, hipcub::DeviceRadixSort::SortPairs(nullptr, tmp, c->d_keys, c->d_keys + cap, c->d_idx, c->d_idx + cap, (int)cap, 0, 64, c->stream));
    HIP_TRY(c, hipMalloc(&c->d_sort_tmp, tmp));
    c->sort_tmp_bytes = tmp;
    c->sort_cap = cap;
  }
  const size_t cap = c->sort_cap;
  hipLaunchKernelGGL(sort_keys_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, c->stream, c->d_hits, n, c->d_keys, c->d_idx);
  size_t tmp = c->sort_tmp_bytes;
  // only the bits a tag of this batch can have take part: an 8-bit digit pass less per byte saved
  HIP_TRY(c, hipcub::DeviceRadixSort::SortPairs(c->d_sort_tmp, tmp, c->d_keys, c->d_keys + cap, c->d_idx, c->d_idx + cap, (int)n, 0, key_bits, c->stream));
  hipLaunchKernelGGL(sort_gather_kernel, dim3((unsigned)((n * 9 + 255) / 256)), dim3(256), 0, c->stream, c->d_hits, c->d_idx + cap, n, c->d_sorted);
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
  if (!c->profiling) return;
  hipEvent_t a = ev_take(c), b = ev_take(c);
  if (!a || !b) { ev_give(c, a); ev_give(c, b); return; }
  (void)hipEventRecord(a, c->stream);
  c->events.push_back({kind, {a, b}});
}
void ev_end(ftkx_ctx *c)
{
  if (!c->profiling || c->events.empty()) return;
  (void)hipEventRecord(c->events.back().second.second, c->stream);
}
void ev_harvest(ftkx_ctx *c, bool all = true)   // all: after a stream synchronise; otherwise only the pairs that have completed
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
int run_batch(ftkx_ctx *c, const double *sparse_field = nullptr, bool cull_done = false)
{
  const bool cull_only = sparse_field != nullptr;   // ftkx_sweep_cull: stop after the cull and list the survivors that read `sparse_field`
  Mesh m;
  fill_mesh(c, m);
  const int nd = c->nd;
  struct Sub { std::vector<MaskJob> jobs; std::vector<Fields> steps; };
  std::vector<Sub> subs(1);
  const bool two_level = ftkx::masks_have_summary(m);
  std::vector<TileParams> tiles;
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
        subs.back().jobs.push_back(MaskJob{s->S, s->V, s->M, two_level ? s->U : nullptr, nullptr, 1.0 / (double)r.factor, big});
        s->mask_factor = r.factor; s->mask_big = rule_on; s->u_rows = m.u_rows;
      }
      f.M[0] = s0.M; f.M[1] = s1 ? s1->M : nullptr;
      f.U[0] = two_level ? s0.U : nullptr; f.U[1] = (two_level && s1) ? s1->U : nullptr;
      subs.back().steps.push_back(f);
    } else {
      TileParams p;
      p.m = m; p.f = f; p.cull = r.mode == MODE_TILE_CULL ? 1 : 0; p.step = 0;
      int tile[3];
      ftkx::tile_dims(nd, tile);
      for (int d = 0; d < 3; d ++) p.ntiles[d] = d < nd ? (int)((c->core_sz[d] + tile[d] - 1) / tile[d]) : 1;
      tiles.push_back(p);
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
  nfields += tiles.size();
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
    for (size_t i = 0; i < tiles.size(); i ++) { hf[tile_base + i] = tiles[i].f; tiles[i].step = (int)(tile_base + i); }
    HIP_TRY(c, hipMemcpyAsync(c->d_desc, c->h_desc, total, hipMemcpyHostToDevice, c->stream));
  }
  const Fields *d_fields = (const Fields *)((char *)c->d_desc + fields_off);
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
  for (const TileParams &p : tiles) { ev_begin(c, K_TILE); ftkx::launch_tile(p, c->stream); ev_end(c); }
  // the FP64 half, once for the whole batch: records of every simplex that passed (timed with the kernel family that fed it)
  if (nfields) { ev_begin(c, tiles.empty() ? K_EXACT : K_TILE); ftkx::launch_records(m, d_fields, c->stream); ev_end(c); }
  HIP_TRY(c, hipGetLastError());
  return FTKX_OK;
}

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

}  // namespace

extern "C" {

const char *ftkx_last_mask_kernel(void) { return ftkx::last_mask_kernel(); }

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
    HIP_TRY(c, hipMalloc((void **)&c->d_counters, (ftkx::CNT_N + 128) * sizeof(u64)));
    HIP_TRY(c, hipMemset(c->d_counters, 0, (ftkx::CNT_N + 128) * sizeof(u64)));
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
  for (auto &kv : c->slices) free_slice(kv.second);
  release_pools(c);
  for (auto &e : c->events) { (void)hipEventDestroy(e.second.first); (void)hipEventDestroy(e.second.second); }
  for (hipEvent_t e : c->event_pool) (void)hipEventDestroy(e);
  if (c->h_red) (void)hipHostFree(c->h_red);
  if (c->h_ahead) (void)hipHostFree(c->h_ahead);
  if (c->d_ahead) (void)hipFree(c->d_ahead);
  if (c->d_red) (void)hipFree(c->d_red);
  if (c->d_hits) (void)hipFree(c->d_hits);
  if (c->d_pass) (void)hipFree(c->d_pass);
  if (c->d_fragile) (void)hipFree(c->d_fragile);
  for (void *p : {(void *)c->d_word_idx, (void *)c->d_words, (void *)c->d_cells, (void *)c->d_patch_cells, (void *)c->d_patches}) if (p) (void)hipFree(p);
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
  c->stream = s ? (hipStream_t)s : c->own_stream;
  return FTKX_OK;
}

int ftkx_set_options(ftkx_ctx *c, const ftkx_options *o)
{
  if (c) c->ahead.clear();
  if (!c || !o) return fail(c, FTKX_E_INVALID, "null argument");
  if (!c->pending.empty()) return fail(c, FTKX_E_INVALID, "ftkx_set_options: sweeps pending, collect first");
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
    HIP_TRY(c, hipMemcpyAsync(*dst, src, count * sizeof(double), on_device == 2 ? hipMemcpyDefault : hipMemcpyHostToDevice, c->stream));
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
  if (on_device != 1) HIP_TRY(c, hipStreamSynchronize(c->stream));   // the source buffers may be reused by the caller on return
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
    hipLaunchKernelGGL(init_red_kernel, dim3((unsigned)((k * 64 + 255) / 256)), dim3(256), 0, c->stream, c->d_red, k * 64);
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

// One pass over the slices for the whole sweep: the sign masks (built under factor_hint, which must not exceed the factor the
// sweeps will use -- the scaling factor only grows, so the factor in force BEFORE these slices arrived qualifies) and, fused into
// the same kernel, what update_vector_field_scaling_factor needs of each slice.  See MaskJob in sweep_params.hpp.
int ftkx_slices_prepare(ftkx_ctx *c, const int *ts, int n, unsigned long long factor_hint, double *res_below, double *max_abs)
{
  if (!c || (n > 0 && !ts)) return fail(c, FTKX_E_INVALID, "null argument");
  if (!c->pending.empty()) return fail(c, FTKX_E_INVALID, "ftkx_slices_prepare: sweeps pending, collect first");
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
    if (ahead_ok) {
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
      if ((rc = ensure_list(c, std::max<u64>(c->list_capacity, 1u << 20))) || (rc = ensure_refine(c, std::max<u64>(c->refine_capacity, 1u << 20)))) {
        for (Slice *s : todo) s->mask_factor = 0;
        return rc;
      }
    }
    if ((rc = ensure_desc(c, std::max(k * sizeof(MaskJob), k * 128 * sizeof(u64))))) return rc;
    hipLaunchKernelGGL(init_red_kernel, dim3((unsigned)((k * 64 + 255) / 256)), dim3(256), 0, c->stream, c->d_red, k * 64, ahead_ok ? c->d_counters : nullptr);
    MaskJob *jobs = (MaskJob *)c->h_desc;
    for (size_t i = 0; i < k; i ++)
      jobs[i] = MaskJob{todo[i]->S, todo[i]->V, todo[i]->M, two_level ? todo[i]->U : nullptr, c->d_red + i * 128, cap, HUGE_VAL};   // rule off: validated below
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
      hipLaunchKernelGGL(fetch_desc_kernel, dim3(4), dim3(256), 0, c->stream, (const u64 *)c->h_ahead, (u64 *)c->d_ahead, bytes / 8);
      ahead_launch(c, m, two_level, (const Fields *)c->d_ahead, (int)a_steps.size());
      HIP_TRY(c, hipGetLastError());
      c->ahead = a_rec;
      // spin on the flag; a device error would leave it unset: look at the stream now and then, give up after a generous while
      const auto t_start = std::chrono::steady_clock::now();
      unsigned long long spins = 0;
      while (__atomic_load_n(flag, __ATOMIC_ACQUIRE) != seq) {
#if defined(__x86_64__) || defined(__i386__)
        __builtin_ia32_pause();                                  // (a spin-wait hint: the sibling hyperthread keeps its issue slots)
#endif
        if ((++ spins & 0xfffffull) == 0) {
          const hipError_t q = hipStreamQuery(c->stream);
          const char *why = nullptr;
          if (q != hipSuccess && q != hipErrorNotReady) why = hipGetErrorString(q);
          else if (q == hipSuccess && __atomic_load_n(flag, __ATOMIC_ACQUIRE) != seq) why = "the stream drained without the reduction arriving";
          else if (std::chrono::steady_clock::now() - t_start > std::chrono::seconds(120)) why = "timed out waiting for the reduction";
          if (why) {      // nothing of this call stands: no masks, no cull-ahead, no announcement
            for (Slice *s : todo) { s->mask_factor = 0; s->have_fused = false; }
            c->ahead.clear(); c->announced.clear();
            return fail(c, FTKX_E_DEVICE, "ftkx_slices_prepare: %s", why);
          }
        }
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

// ---- compact t-slab halo ------------------------------------------------------------------------------------------------------
static int copy_out(ftkx_ctx *c, void *dst, const void *src, size_t bytes, int dst_on_device)
{
  if (!bytes) return FTKX_OK;
  HIP_TRY(c, hipMemcpyAsync(dst, src, bytes, dst_on_device ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost, c->stream));
  return FTKX_OK;
}

int ftkx_export_masks_size(ftkx_ctx *c, int t, size_t *u_bytes_out, size_t *n_words, unsigned long long *mask_factor, double *max_abs)
{
  if (c) c->ahead.clear();
  if (!c) return fail(nullptr, FTKX_E_INVALID, "null context");
  auto it = c->slices.find(t);
  if (it == c->slices.end()) return fail(c, FTKX_E_NOSLICE, "ftkx_export_masks_size: timestep %d not resident", t);
  Slice &s = it->second;
  if (!s.M || !s.U || !s.mask_factor || !s.max_known())
    return fail(c, FTKX_E_UNSUPPORTED, "ftkx_export_masks_size: slice %d has no summarised masks (ftkx_slices_prepare first; needs a mesh the two-level cull supports)", t);
  HIP_TRY(c, hipSetDevice(c->device));
  Mesh m; fill_mesh(c, m);
  for (int attempt = 0; attempt < 2; attempt ++) {
    HIP_TRY(c, hipMemsetAsync(c->d_counters + ftkx::CNT_SPARSE, 0, sizeof(u64), c->stream));
    ftkx::launch_compact_words(m, s.U, s.M, c->d_word_idx, c->d_words, c->words_cap, c->d_counters + ftkx::CNT_SPARSE, c->stream);
    HIP_TRY(c, hipGetLastError());
    HIP_TRY(c, hipMemcpyAsync(c->h_counters, c->d_counters + ftkx::CNT_SPARSE, sizeof(u64), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    const size_t n = (size_t)c->h_counters[0];
    if (n <= c->words_cap) { c->n_words = n; c->words_t = t; break; }
    if (c->d_word_idx) (void)hipFree(c->d_word_idx);
    if (c->d_words) (void)hipFree(c->d_words);
    c->d_word_idx = nullptr; c->d_words = nullptr; c->words_cap = 0;
    const size_t cap = n + n / 8 + 1024;
    HIP_TRY(c, hipMalloc((void **)&c->d_word_idx, cap * sizeof(unsigned)));
    HIP_TRY(c, hipMalloc((void **)&c->d_words, cap * sizeof(u64)));
    c->words_cap = cap;
  }
  if (u_bytes_out) *u_bytes_out = u_bytes_used(c, m);
  if (n_words) *n_words = c->n_words;
  if (mask_factor) *mask_factor = s.mask_factor;
  if (max_abs) *max_abs = s.maxabs;
  return FTKX_OK;
}

int ftkx_export_masks(ftkx_ctx *c, int t, void *U_dst, unsigned *word_index_dst, unsigned long long *words_dst, int dst_on_device)
{
  if (!c || !U_dst) return fail(c, FTKX_E_INVALID, "null argument");
  auto it = c->slices.find(t);
  if (it == c->slices.end() || c->words_t != t) return fail(c, FTKX_E_INVALID, "ftkx_export_masks: call ftkx_export_masks_size for timestep %d first", t);
  HIP_TRY(c, hipSetDevice(c->device));
  int rc;
  Mesh m; fill_mesh(c, m);
  if ((rc = copy_out(c, U_dst, it->second.U, u_bytes_used(c, m), dst_on_device))) return rc;
  if (c->n_words && (!word_index_dst || !words_dst)) return fail(c, FTKX_E_INVALID, "ftkx_export_masks: null list buffers");
  if ((rc = copy_out(c, word_index_dst, c->d_word_idx, c->n_words * sizeof(unsigned), dst_on_device))) return rc;
  if ((rc = copy_out(c, words_dst, c->d_words, c->n_words * sizeof(u64), dst_on_device))) return rc;
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  return FTKX_OK;
}

int ftkx_push_masked_slice(ftkx_ctx *c, int t, int scalar_input, const void *U, const unsigned *word_index, const unsigned long long *words, size_t n_words,
                           unsigned long long mask_factor, double max_abs, int on_device)
{
  if (c) c->ahead.clear();
  if (!c || !U) return fail(c, FTKX_E_INVALID, "null argument");
  if (!c->mesh_set) return fail(c, FTKX_E_INVALID, "push: call ftkx_set_mesh first");
  if (!c->pending.empty()) return fail(c, FTKX_E_INVALID, "push: sweeps pending, collect first");
  if (t < 0 || !pow2_factor(mask_factor) || !(max_abs >= 0)) return fail(c, FTKX_E_INVALID, "ftkx_push_masked_slice: bad arguments");
  if (c->slices.empty()) c->scalar_mode = -1;
  if (c->scalar_mode >= 0 && c->scalar_mode != (scalar_input ? 1 : 0)) return fail(c, FTKX_E_INVALID, "push: scalar and vector slices cannot be mixed in one context");
  HIP_TRY(c, hipSetDevice(c->device));
  const int saved_mode = c->scalar_mode;
  c->scalar_mode = scalar_input ? 1 : 0;
  Mesh m; fill_mesh(c, m);
  if (!ftkx::masks_have_summary(m)) { c->scalar_mode = saved_mode; return fail(c, FTKX_E_UNSUPPORTED, "ftkx_push_masked_slice: this mesh has no summarised masks"); }
  auto it = c->slices.find(t);
  Slice s;
  if (it != c->slices.end() && it->second.sparse) { s = it->second; c->slices.erase(it); }          // the same halo slice again: keep its arrays
  else if (it != c->slices.end()) { free_slice(it->second, c); c->slices.erase(it); }
  const size_t n = n_vertices(c), ncomp = scalar_input ? 1 : (size_t)c->nd;
  // everything below that can fail runs inside `fill`: on failure the half-built slice is released, not leaked
  auto fill = [&]() -> int {
  int rc;
  if (!s.sparse) {
    double **field = scalar_input ? &s.S : &s.V;
    HIP_TRY(c, hipMalloc((void **)field, n * ncomp * sizeof(double)));
    (scalar_input ? s.ownS : s.ownV) = true;
    HIP_TRY(c, hipMemsetAsync(*field, 0, n * ncomp * sizeof(double), c->stream));      // only patches are ever read; zeros elsewhere, not garbage
    if ((rc = ensure_mask_arrays(c, s, true))) return rc;
    s.sparse = true;
  }
  const hipMemcpyKind kind = on_device ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice;
  HIP_TRY(c, hipMemcpyAsync(s.U, U, u_bytes_used(c, m), kind, c->stream));
  if (n_words) {
    if (c->words_cap < n_words) {
      if (c->d_word_idx) (void)hipFree(c->d_word_idx);
      if (c->d_words) (void)hipFree(c->d_words);
      c->d_word_idx = nullptr; c->d_words = nullptr; c->words_cap = 0;
      HIP_TRY(c, hipMalloc((void **)&c->d_word_idx, n_words * sizeof(unsigned)));
      HIP_TRY(c, hipMalloc((void **)&c->d_words, n_words * sizeof(u64)));
      c->words_cap = n_words;
    }
    c->words_t = -1;
    HIP_TRY(c, hipMemcpyAsync(c->d_word_idx, word_index, n_words * sizeof(unsigned), kind, c->stream));
    HIP_TRY(c, hipMemcpyAsync(c->d_words, words, n_words * sizeof(u64), kind, c->stream));
    ftkx::launch_scatter_words(c->d_word_idx, c->d_words, n_words, s.M, c->stream);
    HIP_TRY(c, hipGetLastError());
  }
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  return FTKX_OK;
  };
  const int frc = fill();
  if (frc != FTKX_OK) { free_slice(s, c); c->scalar_mode = saved_mode; return frc; }
  s.mask_factor = mask_factor; s.mask_big = false; s.u_rows = m.u_rows;
  s.maxabs = max_abs;                                                                  // (all a masked slice knows of its values)
  c->slices[t] = s;
  return FTKX_OK;
}

int ftkx_sweep_cull(ftkx_ctx *c, int t_sparse, size_t *n_cells)
{
  if (c) c->ahead.clear();
  if (!c || !n_cells) return fail(c, FTKX_E_INVALID, "null argument");
  *n_cells = 0;
  auto it = c->slices.find(t_sparse);
  if (it == c->slices.end() || !it->second.sparse) return fail(c, FTKX_E_INVALID, "ftkx_sweep_cull: timestep %d is not a masked halo slice", t_sparse);
  if (c->pending.empty()) return FTKX_OK;
  for (const Request &r : c->pending) if (r.mode != MODE_FAST) return fail(c, FTKX_E_UNSUPPORTED, "ftkx_sweep_cull: the pending sweeps do not use the cull (send the slice itself)");
  HIP_TRY(c, hipSetDevice(c->device));
  const double *field = it->second.S ? it->second.S : it->second.V;
  int rc;
  if ((rc = ensure_hit_buffer(c, std::max<u64>(c->capacity, 1u << 16)))) return rc;
  if ((rc = ensure_list(c, std::max<u64>(c->list_capacity, 1u << 20))) || (rc = ensure_refine(c, std::max<u64>(c->refine_capacity, 1u << 20)))) return rc;
  for (int attempt = 0; attempt < 4; attempt ++) {
    if (c->cells_cap < c->list_capacity) {
      if (c->d_cells) (void)hipFree(c->d_cells);
      c->d_cells = nullptr; c->cells_cap = 0;
      HIP_TRY(c, hipMalloc((void **)&c->d_cells, c->list_capacity * sizeof(u64)));
      c->cells_cap = c->list_capacity;
    }
    HIP_TRY(c, hipMemsetAsync(c->d_counters, 0, ftkx::CNT_N * sizeof(u64), c->stream));
    if ((rc = run_batch(c, field))) return rc;
    HIP_TRY(c, hipMemcpyAsync(c->h_counters, c->d_counters, ftkx::CNT_N * sizeof(u64), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    for (auto &e : c->events) { ev_give(c, e.second.first); ev_give(c, e.second.second); }
    c->events.clear();
    const u64 listed = c->h_counters[ftkx::CNT_SURVIVOR_LIST], refined = std::max(c->h_counters[ftkx::CNT_REFINE_LIST], c->h_counters[ftkx::CNT_REFINE_PEAK]);
    if (listed <= c->list_capacity && refined <= c->refine_capacity) { c->n_cells = (size_t)c->h_counters[ftkx::CNT_SPARSE]; *n_cells = c->n_cells; return FTKX_OK; }
    if (refined > c->refine_capacity && (rc = ensure_refine(c, refined + refined / 8 + 1024))) return rc;
    if (listed > c->list_capacity && (rc = ensure_list(c, 2 * listed + 1024))) return rc;
  }
  return fail(c, FTKX_E_DEVICE, "ftkx_sweep_cull: survivor lists kept overflowing");
}

int ftkx_get_sparse_cells(ftkx_ctx *c, unsigned long long *dst, int dst_on_device)
{
  if (!c || (c->n_cells && !dst)) return fail(c, FTKX_E_INVALID, "null argument");
  HIP_TRY(c, hipSetDevice(c->device));
  int rc = copy_out(c, dst, c->d_cells, c->n_cells * sizeof(u64), dst_on_device);
  if (rc) return rc;
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  return FTKX_OK;
}

size_t ftkx_patch_doubles(const ftkx_ctx *c) { return c ? (size_t)(c->nd == 3 ? 216 : 36) * (size_t)(c->scalar_mode == 1 ? 1 : c->nd) : 0; }

static int patches_common(ftkx_ctx *c, int t, const unsigned long long *cells, size_t n, double *patches, int on_device, bool scatter)
{
  if (!c || (n && (!cells || !patches))) return fail(c, FTKX_E_INVALID, "null argument");
  auto it = c->slices.find(t);
  if (it == c->slices.end()) return fail(c, FTKX_E_NOSLICE, "patches: timestep %d not resident", t);
  if (scatter && !it->second.sparse) return fail(c, FTKX_E_INVALID, "ftkx_scatter_patches: timestep %d is not a masked halo slice", t);
  if (!n) return FTKX_OK;
  HIP_TRY(c, hipSetDevice(c->device));
  Mesh m; fill_mesh(c, m);
  const int ncomp = c->scalar_mode == 1 ? 1 : c->nd;
  double *field = c->scalar_mode == 1 ? it->second.S : it->second.V;
  const size_t pd = ftkx_patch_doubles(c);
  const u64 *d_cells = cells; double *d_patches = patches;
  if (!on_device) {                              // host-side callers (gloo tests): stage through device buffers
    if (c->patch_cap < n) {
      if (c->d_patch_cells) (void)hipFree(c->d_patch_cells);
      if (c->d_patches) (void)hipFree(c->d_patches);
      c->d_patch_cells = nullptr; c->d_patches = nullptr; c->patch_cap = 0;
      HIP_TRY(c, hipMalloc((void **)&c->d_patch_cells, n * sizeof(u64)));
      HIP_TRY(c, hipMalloc((void **)&c->d_patches, n * pd * sizeof(double)));
      c->patch_cap = n;
    }
    HIP_TRY(c, hipMemcpyAsync(c->d_patch_cells, cells, n * sizeof(u64), hipMemcpyHostToDevice, c->stream));
    if (scatter) HIP_TRY(c, hipMemcpyAsync(c->d_patches, patches, n * pd * sizeof(double), hipMemcpyHostToDevice, c->stream));
    d_cells = c->d_patch_cells; d_patches = c->d_patches;
  }
  ftkx::launch_patches(m, scatter, d_cells, n, ncomp, field, d_patches, c->stream);
  HIP_TRY(c, hipGetLastError());
  if (!on_device && !scatter) HIP_TRY(c, hipMemcpyAsync(patches, c->d_patches, n * pd * sizeof(double), hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  return FTKX_OK;
}

int ftkx_gather_patches(ftkx_ctx *c, int t, const unsigned long long *cells, size_t n, double *patches, int on_device)
{ return patches_common(c, t, cells, n, patches, on_device, false); }
int ftkx_scatter_patches(ftkx_ctx *c, int t, const unsigned long long *cells, size_t n, const double *patches, int on_device)
{ return patches_common(c, t, cells, n, const_cast<double *>(patches), on_device, true); }

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

int ftkx_sweep_enqueue(ftkx_ctx *c, int t, int scope, unsigned long long factor)
{
  if (!c) return fail(nullptr, FTKX_E_INVALID, "null context");
  if (!c->mesh_set) return fail(c, FTKX_E_INVALID, "sweep: call ftkx_set_mesh first");
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
  int rc = ftkx_sweep_enqueue(c, t, scope, factor);
  if (rc) return rc;
  return ftkx_sweep_collect(c, out, n_out);
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
  for (auto &kv : c->slices) { kv.second.mask_factor = 0; kv.second.have_fused = false; }
  c->ahead.clear();
  c->dense_collects = 0;
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
  c->profiling = on != 0;
  for (int k = 0; k < K_N; k ++) { c->k_ms[k] = 0; c->k_launches[k] = 0; }
  return FTKX_OK;
}

int ftkx_get_kernel_times(const ftkx_ctx *c, double ms[4], unsigned long long launches[4])
{
  if (!c || !ms || !launches) return fail(nullptr, FTKX_E_INVALID, "null argument");
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
