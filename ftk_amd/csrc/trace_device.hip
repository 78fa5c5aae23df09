// Pass 2 on the hit set, its data-parallel half on the device: ftkx_trace_curves_ctx.
//
// The reference traces on the host (critical_point_tracker::trace_critical_points_offline, include/ftk/filters/critical_point_tracker.hh:
// 668-817: a std::set of elements, union-find over the neighbours that share a (d+1)-cell, geometry/cc2curves.hh:10-111 for the order
// of the points).  ftkx_trace_curves (trace.cpp) does that on host threads in 1.5 ms for the 62 181 records of woven 1024^2 x 64 -- four
// times the sweep that produced them.  Two of its phases are independent per record: the neighbour search (a handful of tag look-ups per
// record) and the component labelling (a union-find over the neighbour edges).  Those run here, on the GPU the records came from: the
// tags go up (8 bytes per record), a kernel finds every record's neighbours by binary search in the sorted tags, a lock-free union-find
// labels the components, and neighbours, degrees and roots come back (29-37 bytes per record).  What stays on the host is what is
// serial per curve -- seeds in the reference's element order, the walk along each curve -- in trace.cpp, unchanged.
#include <chrono>
#include "ctx.hpp"

using namespace ftkxh;

namespace {

struct TraceGeom { long long lb[3], sz[3]; unsigned long long prod[4]; int nd, ntypes, maxnb; };

// neighbours of record i inside the set, in the order of the candidate table (= the reference's element order)
__global__ __launch_bounds__(256) void trace_neighbours_kernel(const TraceGeom g, const u64 *__restrict__ tags, int n, const int *__restrict__ cand_off, const int *__restrict__ cand,
                                                               int *__restrict__ nbr, unsigned char *__restrict__ deg, int *__restrict__ parent)
{
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const u64 tag = tags[i];
  const int type = (int)(tag % (u64)g.ntypes);
  u64 ci = tag / (u64)g.ntypes;
  long long cc[4] = {0, 0, 0, 0};
  for (int d = 0; d < g.nd; d ++) { cc[d] = g.lb[d] + (long long)(ci % (u64)g.sz[d]); ci /= (u64)g.sz[d]; }
  cc[g.nd] = (long long)ci;
  int cnt = 0;
  for (int q = cand_off[type]; q < cand_off[type + 1]; q ++) {
    const int *c = cand + 5 * q;
    u64 idx = 0;
    bool ok = true;
    for (int d = 0; d < g.nd; d ++) {
      const long long rel = cc[d] + c[1 + d] - g.lb[d];
      ok = ok && rel >= 0 && rel < g.sz[d];
      idx += (u64)rel * g.prod[d];
    }
    const long long tt = cc[g.nd] + c[1 + g.nd];
    if (!ok || tt < 0) continue;
    idx += (u64)tt * g.prod[g.nd];
    const u64 want = idx * (u64)g.ntypes + (u64)c[0];
    int lo = 0, hi = n;                                   // first position with tags[pos] >= want
    while (lo < hi) { const int mid = (lo + hi) >> 1; if (tags[mid] < want) lo = mid + 1; else hi = mid; }
    if (lo < n && tags[lo] == want && cnt < g.maxnb) nbr[(size_t)i * g.maxnb + cnt ++] = lo;
  }
  for (int q = cnt; q < g.maxnb; q ++) nbr[(size_t)i * g.maxnb + q] = -1;
  deg[i] = (unsigned char)cnt;
  parent[i] = i;
}

__device__ inline int uf_find(int *parent, int x)
{
  for (;;) {
    const int p = __atomic_load_n(&parent[x], __ATOMIC_RELAXED);
    if (p == x) return x;
    const int pp = __atomic_load_n(&parent[p], __ATOMIC_RELAXED);
    if (pp != p) __atomic_store_n(&parent[x], pp, __ATOMIC_RELAXED);     // path halving (a benign race: any ancestor will do)
    x = p;
  }
}

// curves = connected components of the ordinary records (at most two neighbours): larger roots are linked under smaller ones only
__global__ __launch_bounds__(256) void trace_unite_kernel(int n, int maxnb, const int *__restrict__ nbr, const unsigned char *__restrict__ deg, int *parent)
{
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n || deg[i] > 2) return;
  for (int q = 0; q < deg[i]; q ++) {
    const int j = nbr[(size_t)i * maxnb + q];
    if (j < 0 || j >= i || deg[j] > 2) continue;           // (every edge is seen from both ends: once is enough)
    int a = i, b = j;
    for (;;) {
      a = uf_find(parent, a); b = uf_find(parent, b);
      if (a == b) break;
      if (a < b) { const int t = a; a = b; b = t; }
      if (atomicCAS(&parent[a], a, b) == a) break;
    }
  }
}

__global__ __launch_bounds__(256) void trace_roots_kernel(int n, int *parent, int *__restrict__ root)
{
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < n) root[i] = uf_find(parent, i);
}

}  // namespace

extern "C" {

static int trace_curves_ctx_impl(ftkx_ctx *c, int nd, const long long domain_st[3], const long long domain_sz[3], const ftkx_cp_t *recs, const unsigned long long *tags, size_t n, ftkx_curves *out)
{
  auto tag_of = [&](size_t i) { return tags ? tags[i] : recs[i].tag; };
  auto on_host = [&]() { return tags ? ftkx::trace_curves_tags(nd, domain_st, domain_sz, tags, n, out) : ftkx_trace_curves(nd, domain_st, domain_sz, recs, n, out); };
  if (!c) return on_host();
  constexpr bool timing = false;      // (phase timing to stderr: a debugging aid, compiled out)
  const auto tp0 = std::chrono::steady_clock::now();
  if ((nd != 2 && nd != 3) || !domain_st || !domain_sz || (!recs && !tags && n) || !out) return fail(c, FTKX_E_INVALID, "ftkx_trace_curves_ctx: bad arguments");
  // few records, or tags that do not come strictly ascending (the sweep delivers them so): the host does it all
  bool ascending = n < (1u << 30);
  for (size_t i = 1; i < n && ascending; i ++) ascending = tag_of(i - 1) < tag_of(i);
  if (n < 4096 || !ascending) return on_host();
  HIP_TRY(c, hipSetDevice(c->device));
  static thread_local std::vector<int> cand_off[2], cand_flat[2];
  static thread_local int maxnb_of[2] = {0, 0};
  const int w = nd - 2;
  if (cand_off[w].empty()) maxnb_of[w] = ftkx::trace_candidates(nd, cand_off[w], cand_flat[w]);
  const int maxnb = maxnb_of[w];
  // device / pinned buffers, kept with the context: tags | nbr | root | deg on both sides, parent and the tables on the device
  const size_t per = 8 + (size_t)maxnb * 4 + 4 + 1;
  const size_t bytes = n * per + 64, tbytes = (cand_off[w].size() + cand_flat[w].size()) * sizeof(int);
  if (c->tr_cap < bytes) {
    if (c->tr_dev) (void)hipFree(c->tr_dev);
    if (c->tr_host) (void)hipHostFree(c->tr_host);
    if (c->tr_parent) (void)hipFree(c->tr_parent);
    c->tr_dev = nullptr; c->tr_host = nullptr; c->tr_parent = nullptr; c->tr_cap = 0;
    const size_t cap = bytes + bytes / 4;
    HIP_TRY(c, hipMalloc(&c->tr_dev, cap));
    HIP_TRY(c, hipHostMalloc(&c->tr_host, cap, hipHostMallocNonCoherent));
    HIP_TRY(c, hipMalloc(&c->tr_parent, (cap / per + 1) * sizeof(int)));
    c->tr_cap = cap;
  }
  if (c->tr_tables_nd != nd) {
    if (c->tr_tables) (void)hipFree(c->tr_tables);
    c->tr_tables = nullptr; c->tr_tables_nd = 0;
    HIP_TRY(c, hipMalloc(&c->tr_tables, tbytes));
    HIP_TRY(c, hipMemcpy(c->tr_tables, cand_off[w].data(), cand_off[w].size() * sizeof(int), hipMemcpyHostToDevice));
    HIP_TRY(c, hipMemcpy((int *)c->tr_tables + cand_off[w].size(), cand_flat[w].data(), cand_flat[w].size() * sizeof(int), hipMemcpyHostToDevice));
    c->tr_tables_nd = nd;
  }
  // layout (8-byte aligned pieces): tags u64[n] | nbr int[n * maxnb] | root int[n] | deg u8[n]
  const size_t off_nbr = n * 8, off_root = off_nbr + n * (size_t)maxnb * 4, off_deg = off_root + n * 4;
  u64 *h_tags = (u64 *)c->tr_host;
  if (tags) memcpy(h_tags, tags, n * sizeof(u64)); else for (size_t i = 0; i < n; i ++) h_tags[i] = recs[i].tag;
  char *d = (char *)c->tr_dev;
  const auto tp1 = std::chrono::steady_clock::now();
  HIP_TRY(c, hipMemcpyAsync(d, c->tr_host, n * 8, hipMemcpyHostToDevice, c->stream));
  TraceGeom g;
  memset(&g, 0, sizeof(g));
  g.nd = nd; g.ntypes = nd == 2 ? 12 : 60; g.maxnb = maxnb;
  g.prod[0] = 1;
  for (int a = 0; a < nd; a ++) { g.lb[a] = domain_st[a]; g.sz[a] = domain_sz[a]; g.prod[a + 1] = g.prod[a] * (unsigned long long)domain_sz[a]; }
  const unsigned grid = (unsigned)((n + 255) / 256);
  const int *d_off = (const int *)c->tr_tables, *d_cand = d_off + cand_off[w].size();
  hipLaunchKernelGGL(trace_neighbours_kernel, dim3(grid), dim3(256), 0, c->stream, g, (const u64 *)d, (int)n, d_off, d_cand, (int *)(d + off_nbr), (unsigned char *)(d + off_deg), (int *)c->tr_parent);
  hipLaunchKernelGGL(trace_unite_kernel, dim3(grid), dim3(256), 0, c->stream, (int)n, maxnb, (const int *)(d + off_nbr), (const unsigned char *)(d + off_deg), (int *)c->tr_parent);
  hipLaunchKernelGGL(trace_roots_kernel, dim3(grid), dim3(256), 0, c->stream, (int)n, (int *)c->tr_parent, (int *)(d + off_root));
  HIP_TRY(c, hipGetLastError());
  HIP_TRY(c, hipMemcpyAsync((char *)c->tr_host + off_nbr, d + off_nbr, off_deg + n - off_nbr, hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  const auto tp2 = std::chrono::steady_clock::now();
  const char *h = (const char *)c->tr_host;
  const int rc = ftkx::trace_curves_with(nd, domain_st, domain_sz, h_tags, n, out, (const int *)(h + off_nbr), (const unsigned char *)(h + off_deg), (const int *)(h + off_root), maxnb);
  if (timing) {
    const auto tp3 = std::chrono::steady_clock::now();
    auto us = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) { return std::chrono::duration<double, std::micro>(b - a).count(); };
    fprintf(stderr, "ftkx_trace_curves_ctx: %zu records, maxnb %d: checks + tags %.0f us, device (up, 3 kernels, down %zu bytes) %.0f us, host (seeds, walks, curves) %.0f us\n",
            n, maxnb, us(tp0, tp1), off_deg + n - off_nbr, us(tp1, tp2), us(tp2, tp3));
  }
  if (rc != FTKX_OK) return fail(c, rc, "ftkx_trace_curves_ctx: tracing failed (%d)", rc);
  return FTKX_OK;
}


int ftkx_trace_curves_ctx(ftkx_ctx *c, int nd, const long long domain_st[3], const long long domain_sz[3], const ftkx_cp_t *recs, size_t n, ftkx_curves *out)
{ return trace_curves_ctx_impl(c, nd, domain_st, domain_sz, recs, nullptr, n, out); }

int ftkx_trace_curves_tags_ctx(ftkx_ctx *c, int nd, const long long domain_st[3], const long long domain_sz[3], const unsigned long long *tags, size_t n, ftkx_curves *out)
{ return trace_curves_ctx_impl(c, nd, domain_st, domain_sz, nullptr, tags, n, out); }

}  // extern "C"
