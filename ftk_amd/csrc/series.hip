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
#include <algorithm>
#include <chrono>
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
  c->sr_lists_owner = 0;                                      // (the batch takes the counters and the survivor lists over)
  // (... from whatever still runs on the tail stream: the tail of a split pass queued behind the one the batch sweeps for shares them in
  // STREAM order only with its own stream -- it must be through before the batch's kernels start on the context's stream)
  if (c->sr_tail_stream) { HIP_TRY(c, hipStreamSynchronize(c->sr_tail_stream)); if (c->sr_tail_stream2) HIP_TRY(c, hipStreamSynchronize(c->sr_tail_stream2)); }
  struct Through { ftkx_ctx *c; bool was; ~Through() { c->sr_internal = was; } } through{c, c->sr_internal};
  c->sr_internal = true;
  const unsigned long long hint = std::max<unsigned long long>(factor_of(*running), 256ull);
  std::vector<double> below(slice_ts.size());
  // work-index tags count inside one scope (simplicial_regular_mesh.hh:480-493): the batch takes a step of both scopes as two requests
  // (their tags may coincide; ftkx_cp_ordinal tells the records apart -- what the patched tracker's resident step asks for)
  std::vector<int> ets(ts, ts + n), esc(scopes, scopes + n), origin((size_t)n);
  for (int i = 0; i < n; i ++) origin[(size_t)i] = i;
  if (c->opt.tag_mode == FTKX_TAG_WORK_INDEX)
    for (size_t i = 0; i < esc.size(); i ++)
      if (esc[i] == FTKX_SCOPE_BOTH) {
        esc[i] = FTKX_SCOPE_ORDINAL;
        ets.insert(ets.begin() + (long)i + 1, ets[i]); esc.insert(esc.begin() + (long)i + 1, FTKX_SCOPE_INTERVAL); origin.insert(origin.begin() + (long)i + 1, origin[i]);
        i ++;
      }
  int rc = ftkx_sweep_announce(c, ets.data(), esc.data(), (int)ets.size());
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
  std::vector<unsigned long long> ef(ets.size());
  for (size_t i = 0; i < ets.size(); i ++) ef[i] = f[(size_t)origin[i]];
  if ((rc = ftkx_sweep_enqueue_many(c, ets.data(), esc.data(), ef.data(), (int)ets.size()))) return rc;
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

// buckets of the ordering step: 2^this over the keys a pass can produce (at most kSeriesMaxBins).  2^14: the scan of the counts is a
// one-workgroup kernel (18 us over 2^16 of them), the ranking inside a bucket costs next to nothing more with four records than with one
int bins_log2()
{
  return 14;
}

// ---- the device-driven pass in two halves -------------------------------------------------------------------------------------------
// what of a pass can be checked without the device: the steps, the slices they read in time order
int series_steps(ftkx_ctx *c, const int *ts, const int *scopes, int n, std::vector<int> &slice_ts)
{
  for (int i = 0; i < n; i ++) {
    if (scopes[i] < FTKX_SCOPE_ORDINAL || scopes[i] > FTKX_SCOPE_BOTH) return fail(c, FTKX_E_INVALID, "ftkx_sweep_series: bad scope %d", scopes[i]);
    if (i > 0 && ts[i] <= ts[i - 1]) return fail(c, FTKX_E_INVALID, "ftkx_sweep_series: timesteps must be strictly ascending");
    if (slice_ts.empty() || slice_ts.back() != ts[i]) slice_ts.push_back(ts[i]);
    if (scopes[i] & FTKX_SCOPE_INTERVAL) slice_ts.push_back(ts[i] + 1);
  }
  for (size_t j = 0; j < slice_ts.size(); j ++)
    if (c->slices.find(slice_ts[j]) == c->slices.end()) return fail(c, FTKX_E_NOSLICE, "ftkx_sweep_series: slice %d not resident", slice_ts[j]);
  return FTKX_OK;
}

// does the device-driven form cover these steps?  (What it does not is swept by the host-driven batch.)
bool series_applicable(ftkx_ctx *c, const int *ts, const int *scopes, int n, const std::vector<int> &slice_ts, const std::vector<Slice *> &sl, u64 cells, int t_halo = -1)
{
  const int nd = c->nd;
  const size_t k = sl.size();
  bool ok = !c->opt.exact_only && (nd == 2 || c->opt.robust) && c->dense_collects == 0 && !c->opt.use_type_filter && cells > 0 && n <= ftkx::kSeriesMaxSlices && k <= (size_t)ftkx::kSeriesMaxSlices;
  if (const char *e = getenv("FTKX_SERIES")) ok = ok && atoi(e) != 0;
  // the order of (step, corner, type) must be the order of the tags
  // (work-index tags: one step.  With both scopes the records come in element order -- step, corner, type -- which is not the order of
  // their tags any more: each counts inside its own scope)
  if (ok && c->opt.tag_mode == FTKX_TAG_WORK_INDEX) ok = n == 1;
  if (ok && c->opt.tag_mode == FTKX_TAG_REFERENCE) {         // int32 products: equal to the 64-bit formula only while nothing wraps
    long double bound = nd == 2 ? 12.0L : 60.0L;
    for (int d = 0; d < nd; d ++) bound *= (long double)c->dom_sz[d];
    ok = bound * (long double)(ts[n - 1] + 2) < 2147483648.0L;
  }
  if (ok && (long double)n * (long double)cells * 64.0L >= 4611686018427387904.0L) ok = false;   // the order key must fit
  for (size_t j = 0; ok && j < k; j ++) ok = !sl[j]->sparse || slice_ts[j] == t_halo;      // (a slab pass's halo slice: masks + patches arrive inside the pass)
  for (int i = 0; ok && i < n; i ++) {                       // the same consistency rule as ftkx_sweep_enqueue
    if (!(scopes[i] & FTKX_SCOPE_INTERVAL)) continue;
    const Slice &a = c->slices[ts[i]], &b = c->slices[ts[i] + 1];
    if (b.sparse) { if (a.J != nullptr) ok = false; continue; }      // (a masks-only slice carries S or V, never J)
    if ((a.J == nullptr) != (b.J == nullptr) || (a.S == nullptr) != (b.S == nullptr)) ok = false;
  }
  if (ok && (c->opt.coords_mode == 2 || c->opt.coords_mode == 3)) ok = false;     // (their bounds checks live in ftkx_sweep_enqueue)
  (void)slice_ts;
  return ok;
}

int ensure_series_buffers(ftkx_ctx *c, ftkx_series_buffers &B, size_t nwords, size_t desc_bytes, bool to_device)
{
  int rc;
  if ((rc = grow_device(c, &B.results, &B.results_cap, std::max<size_t>(nwords, 1024)))) return rc;
  const size_t h_words = nwords + (size_t)c->fragile_capacity * 10;
  if (B.h_results_cap < h_words) {
    if (B.h_results) { HIP_TRY(c, hipStreamSynchronize(c->stream)); (void)hipHostFree(B.h_results); B.h_results = nullptr; B.h_results_cap = 0; }
    const size_t cap = h_words + h_words / 4 + 1024;
    HIP_TRY(c, hipHostMalloc((void **)&B.h_results, (cap + 8) * sizeof(u64), hipHostMallocCoherent));
    B.h_results_cap = cap;
    *reinterpret_cast<volatile unsigned *>(B.h_results + cap) = 0u;
    *(reinterpret_cast<volatile unsigned *>(B.h_results + cap) + 2) = 0u;          // (the copy kernel's flag)
    B.seq = 0;
  }
  if (B.out_cap < (size_t)c->capacity) {
    if (B.out) { HIP_TRY(c, hipStreamSynchronize(c->stream)); HIP_TRY(c, hipHostFree(B.out)); B.out = nullptr; B.out_cap = 0; }
    // The host reads the records after it has seen the flag, which a LATER kernel on the same stream (series_finish / the copy kernel's
    // last workgroup) stores with system scope behind a __threadfence_system() -- there is no synchronisation point of the runtime in
    // between.  The HIP programming model promises visibility of such stores for COHERENT (fine-grained) host memory; that is what the
    // buffer is (round 6: it costs nothing measurable -- woven 1024^2 x 64 0.2033 against 0.2075 ms per pass, double_gyre 0.760 / 0.763,
    // 512^3 x 32 5.479 / 5.486, the host's copy of 62 181 records 0.83 / 0.81 ms either way: profiles/r06_coherent_cost.txt).
    // FTKX_SERIES_OUT_COHERENT=0: coarse-grained as in rounds 3-5, which relies on gfx950 behaviour -- the record kernel's system-scope
    // stores have left the device when the later kernel starts, and PCIe writes snoop the CPU's caches.
    const bool coherent_out = !(getenv("FTKX_SERIES_OUT_COHERENT") && atoi(getenv("FTKX_SERIES_OUT_COHERENT")) == 0);
    HIP_TRY(c, hipHostMalloc((void **)&B.out, (size_t)c->capacity * sizeof(ftkx_cp_t), coherent_out ? hipHostMallocCoherent : hipHostMallocNonCoherent));
    B.out_cap = (size_t)c->capacity;
  }
  if (to_device && B.d_out_cap < (size_t)c->capacity) {
    if (B.d_out) { HIP_TRY(c, hipFree(B.d_out)); B.d_out = nullptr; B.d_out_cap = 0; }
    HIP_TRY(c, hipMalloc((void **)&B.d_out, (size_t)c->capacity * sizeof(ftkx_cp_t)));
    B.d_out_cap = (size_t)c->capacity;
  }
  if (to_device && !B.copy_done) {
    HIP_TRY(c, hipMalloc((void **)&B.copy_done, sizeof(unsigned)));
    HIP_TRY(c, hipMemsetAsync(B.copy_done, 0, sizeof(unsigned), c->stream));
    HIP_TRY(c, hipEventCreateWithFlags(&B.ev_copied, hipEventDisableTiming));
  }
  if (B.desc_cap < desc_bytes) {
    if (B.h_desc) { HIP_TRY(c, hipStreamSynchronize(c->stream)); HIP_TRY(c, hipHostFree(B.h_desc)); B.h_desc = nullptr; }
    if (B.d_desc) { HIP_TRY(c, hipFree(B.d_desc)); B.d_desc = nullptr; }
    const size_t cap = std::max<size_t>(desc_bytes + desc_bytes / 4, 1 << 16);
    HIP_TRY(c, hipHostMalloc(&B.h_desc, cap, hipHostMallocDefault));
    HIP_TRY(c, hipMalloc(&B.d_desc, cap));
    B.desc_cap = cap;
  }
  return FTKX_OK;
}

void series_queue_copy(ftkx_ctx *c, ftkx_series_pending &P, const unsigned *wait_flag, unsigned wait_val);

// ---- the tail next to the next mask kernel (round 5) -------------------------------------------------------------------------------------
// A sparse pass ends in a latency chain -- cull + factors, refine, exact test, ordering, records: ~70 us for a handful of workgroups -- and the
// mask kernel of the pass queued behind it used to wait for all of it.  A SPLIT pass queues only its begin and mask kernels on the context's
// stream; its tail goes to a stream of its own behind an event and runs NEXT TO the mask kernel of the pass queued behind.  What makes that
// work is the shape of the tail: every kernel of it must fit next to two of the mask kernel's three wavefronts per SIMD (168 registers) --
// a kernel that needs more gets no wavefront slot before the mask kernel drains (tools/probe/anyorder.hip; the fused tail, 388 registers in
// 3D, was tried first: it had to be let in FRONT of the mask kernel by a gate, which cost more than the overlap gained -- NOTES.md).  So the
// tail of a split pass is the kernel chain with the record kernel held to 168 registers (series_record_lean_kernel: its few records do not
// care about the scratch) and the bucket scan as one workgroup of four wavefronts.  Under the mask kernel's memory traffic every dependent
// load of the chain takes several times as long (256^3 x 16: the chain 340 us instead of 70) -- hidden as long as the mask kernel is longer:
// taken by pipelined, single-rank passes whose mask kernel reads at least kSplitMinBytes -- after a sparse pass, and in 2D after any.
// What the two sides share is kept apart: the reduction slots are the pass's own (ftkx_series_buffers::red), the counters and the histogram
// are zeroed on the tail stream, and a slice whose masks the next pass rebuilds while this pass's tail still reads them gets fresh arrays
// (`retired`: back to the pool when this pass is completed).
using ftkxh::kSplitMinBytes;
hipStream_t tail_stream(ftkx_ctx *c, const ftkx_series_pending &P) { return P.split ? (P.tail_set ? c->sr_tail_stream2 : c->sr_tail_stream) : c->stream; }

// the counters, lists and ordering arrays a pass's tail works on: the context's own, or the second set
struct TailView { u64 *counters, *list, *refine, *pass, *fragile, *bucketed, *sorted; unsigned *hist, *boff; };
TailView tail_view(ftkx_ctx *c, const ftkx_series_pending &P)
{
  if (P.tail_set == 0) return TailView{c->d_counters, c->d_list, c->d_refine, c->d_pass, c->d_fragile, c->sr_bucketed, c->sr_sorted, c->sr_hist, c->sr_boff};
  const ftkx_ctx::tail_set &S = c->sr_set1;
  return TailView{S.counters, S.list, S.refine, S.pass, S.fragile, S.bucketed, S.sorted, S.hist, S.boff};
}
// the second set at the context's capacities (they change only where everything has been waited for; its stream is drained before anything
// of it is freed)
int ensure_set1(ftkx_ctx *c)
{
  ftkx_ctx::tail_set &S = c->sr_set1;
  if (S.counters && S.capacity == c->capacity && S.list_capacity == c->list_capacity && S.refine_capacity == c->refine_capacity &&
      S.fragile_capacity == c->fragile_capacity && S.bins_cap == c->sr_bins_cap) return FTKX_OK;
  if (c->sr_tail_stream2) HIP_TRY(c, hipStreamSynchronize(c->sr_tail_stream2));
  for (void **q : {(void **)&S.counters, (void **)&S.list, (void **)&S.refine, (void **)&S.pass, (void **)&S.fragile, (void **)&S.bucketed, (void **)&S.sorted, (void **)&S.hist, (void **)&S.boff})
    if (*q) { HIP_TRY(c, hipFree(*q)); *q = nullptr; }
  S = ftkx_ctx::tail_set();
  HIP_TRY(c, hipMalloc((void **)&S.counters, (ftkx::CNT_N + 128 + 8) * sizeof(u64)));
  HIP_TRY(c, hipMemsetAsync(S.counters, 0, (ftkx::CNT_N + 128 + 8) * sizeof(u64), c->stream));
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  HIP_TRY(c, hipMalloc((void **)&S.list, (size_t)c->list_capacity * sizeof(u64)));
  HIP_TRY(c, hipMalloc((void **)&S.refine, (size_t)c->refine_capacity * sizeof(u64)));
  HIP_TRY(c, hipMalloc((void **)&S.pass, (size_t)c->capacity * sizeof(u64)));
  HIP_TRY(c, hipMalloc((void **)&S.fragile, (size_t)c->fragile_capacity * 10 * sizeof(u64)));
  HIP_TRY(c, hipMalloc((void **)&S.bucketed, (size_t)c->capacity * sizeof(u64)));
  HIP_TRY(c, hipMalloc((void **)&S.sorted, (size_t)c->capacity * sizeof(u64)));
  HIP_TRY(c, hipMalloc((void **)&S.hist, c->sr_bins_cap * sizeof(unsigned)));
  HIP_TRY(c, hipMalloc((void **)&S.boff, c->sr_bins_cap * sizeof(unsigned)));
  S.capacity = c->capacity; S.list_capacity = c->list_capacity; S.refine_capacity = c->refine_capacity; S.fragile_capacity = c->fragile_capacity; S.bins_cap = c->sr_bins_cap;
  return FTKX_OK;
}

void release_retired(ftkx_ctx *c, ftkx_series_pending &P)
{
  for (auto &mu : P.retired) { if (mu.first) c->pool_M.push_back(mu.first); if (mu.second) c->pool_U.push_back(mu.second); }
  P.retired.clear();
  // (slices dropped while this pass was the newest one open: called with P.open already false, so that free_slice does not park them again
  // with THIS pass; a pass queued behind it was planned after the drop and does not know them)
  std::vector<Slice> parked;
  parked.swap(P.parked);
  const bool was_open = P.open;
  P.open = false;
  for (Slice &sl : parked) {
    // (free_slice parks with the newest open split pass: not this one, and a newer one never read these -- straight to the pools)
    const int open_was = c->sr_open;
    c->sr_open = 0;
    free_slice(sl, c);
    c->sr_open = open_was;
  }
  P.open = was_open;
}

bool short_chain_now(const ftkx_ctx *c, bool to_device, bool *small_now)
{
  const bool small_on = ftkx::env_hook("FTKX_SERIES_HOOKS", "small", 1) != 0, short_on = ftkx::env_hook("FTKX_SERIES_HOOKS", "short", 1) != 0;
  const bool sn = small_on && c->sr_skip_small == 0 && !to_device;
  if (small_now) *small_now = sn;
  return sn && short_on && c->sr_short_chain;
}

// the kernels behind the fused tail: refine, exact test, ordering, records, finish
void series_queue_rest(ftkx_ctx *c, const ftkx_series_pending &P, const Mesh &m, unsigned seq)
{
  ftkx_series_buffers &B = c->sr_buf[P.buf];
  Fields *d_steps = (Fields *)((char *)B.d_desc + P.off_steps);
  unsigned *flag = reinterpret_cast<unsigned *>(B.h_results + B.h_results_cap);
  hipStream_t st = tail_stream(c, P);
  // (a split pass: sparse data next to a mask kernel -- every workgroup of these kernels waits for a wavefront slot: few of them)
  const bool few = P.split_sparse;
  const TailView T = tail_view(c, P);
  if (P.two_level && !P.refined) ftkx::launch_refine(m, d_steps, T.refine, c->refine_capacity, T.list, c->list_capacity, st, few ? 64 : 0);   // (a slab pass refines before it asks for patches)
  ftkx::launch_exact(m, d_steps, 0, T.list, c->list_capacity, st, few ? 64 : 0);
  ftkx::launch_bucket_scan(T.hist, T.boff, (unsigned)P.nbins, T.counters, st, P.split);
  ftkx::launch_bucket_scatter(m, T.boff, T.bucketed, st, few ? 16 : 0);
  ftkx::launch_bucket_rank(m, T.bucketed, T.boff, T.sorted, B.results, st, few ? 16 : 0);
  if (B.copy_out) { (void)hipStreamWaitEvent(st, B.ev_copied, 0); B.copy_out = false; }   // (the copy of the pass that used these buffers last: long through)
  ftkx::launch_series_records(m, d_steps, T.sorted, P.to_device ? B.d_out : B.out, st, P.split);
  if (!P.split) ev_end(c);
  ftkx::launch_series_finish(m, B.results, P.nwords, c->list_capacity, c->refine_capacity, B.h_results, flag, seq, st);
}

// the records' way over PCIe: a small kernel on its own stream, behind the finish kernel (the count is final) and next to whatever the
// context's stream does then -- the mask kernel of the pass queued behind this one
void series_queue_copy(ftkx_ctx *c, ftkx_series_pending &P, const unsigned *wait_flag, unsigned wait_val)
{
  ftkx_series_buffers &B = c->sr_buf[P.buf];
  unsigned *flag = reinterpret_cast<unsigned *>(B.h_results + B.h_results_cap);
  // (no event between the pass and its copy: with a pass queued behind, the word its begin kernel stores says that THIS pass's finish kernel
  // is through as well -- same stream, in order; without one, ftkx_sweep_series_complete has already waited for the finish kernel's flag.
  // An event recorded behind the finish kernel stood between it and the next pass's begin kernel: ~5 us per pipelined pass)
  ftkx::launch_series_copy_out(B.d_out, B.out, (u64)B.d_out_cap, B.results, B.copy_done, flag + 2, P.seq, c->sr_copy_stream, wait_flag, wait_val);
  (void)hipEventRecord(B.ev_copied, c->sr_copy_stream);
  B.copy_out = true;
  P.copy_pending = false;
}

// A pass that is NOT split works on the context's counters, lists and ordering arrays in the order of the context's stream -- and so does
// whatever recycles a dropped slice's arrays behind it.  Every split pass still open has its tail on a stream of its own (two of them may
// be out at once: tail sets 0 and 1): the context's stream waits for ALL of them, not only for the pass queued last -- with two tails
// open, the newest one's event says nothing about the older one's.
int wait_for_open_tails(ftkx_ctx *c, const ftkx_series_pending *self)
{
  for (ftkx_series_pending &X : c->sr_pend)
    if (&X != self && X.open && X.split && c->sr_buf[X.buf].ev_tail) HIP_TRY(c, hipStreamWaitEvent(c->stream, c->sr_buf[X.buf].ev_tail, 0));
  return FTKX_OK;
}

// ---- the one-launch pass for small series (one_kernel.hip) --------------------------------------------------------------------------------
bool series_one_eligible(ftkx_ctx *c, const ftkx_series_pending &P, int n, size_t k, u64 cells, bool dist)
{
  const bool one_on = ftkx::env_hook("FTKX_SERIES_HOOKS", "one", 1) != 0;
  if (!one_on || dist || c->sr_one_off > 0) { if (c->sr_one_off > 0 && !dist) c->sr_one_off --; return false; }
  if (n > ftkx::kOneMaxSteps || k > (size_t)ftkx::kOneMaxSlices) return false;
  // small: at most kOneMaxBlocks staging batches of (step, corner) -- 128 corners a batch in 2D, 64 in 3D --, and slices whose reduction is a few
  // chunks' worth of reading
  const u64 bs = c->nd == 2 ? 128 : 64;
  if ((cells * (u64)n + bs - 1) / bs > (u64)ftkx::kOneMaxBlocks || (u64)n_vertices(c) * (u64)k > (1ull << 22)) return false;
  // ... and where it wins.  Measured (tools: in-kernel stamps, NOTES.md round 5): a device-wide barrier costs ~10 us on this part, two of them plus
  // launch and hand-over ~35 us before any work; moving_extremum_3d 32^3 x 8 (sparse): 138 against 192 us for the kernel chain; woven 128^2 x 10
  // (7 357 records, BASELINE config 1): 92-105 against 91 -- hit-dense series of that size stay with the chain, whose kernels overlap nothing
  // either but whose ten launches cost no more than this kernel's barriers and its one wavefront per SIMD.  The last pass's record count decides.
  if (cells * (u64)n > (1ull << 17) && c->stats.hits > 2048) return false;
  (void)P;
  return true;
}

int series_plan_one(ftkx_ctx *c, ftkx_series_pending &P, const int *ts, const int *scopes, int n, const std::vector<Slice *> &sl, const ftkx_series_pending *prev,
                    ftkx_series_pending *before)
{
  const size_t k = P.k;
  int rc;
  P.one = true; P.split = false; P.to_device = false; P.short_chain = false; P.small_now = false;
  P.red_index.assign(k, -1); P.gen.assign(k, 0);               // (no masks are built: nothing to mark when the pass is collected)
  P.ntodo = 0;
  if ((rc = ensure_hit_buffer(c, std::max<u64>(c->capacity, 1u << 16)))) return rc;
  if ((rc = ensure_fragile(c, std::max<u64>(c->fragile_capacity, 1u << 12)))) return rc;
  const size_t nwords = (size_t)ftkx::SR_HEAD + (size_t)n + 2 * k;
  P.nwords = nwords;
  P.buf = (int)(&P - c->sr_pend);
  ftkx_series_buffers &B = c->sr_buf[P.buf];
  if ((rc = ensure_series_buffers(c, B, nwords, 256, false))) return rc;
  if (!c->sr_one_scratch) {
    HIP_TRY(c, hipMalloc((void **)&c->sr_one_scratch, (size_t)ftkx::ONE_WORDS * sizeof(u64)));
    HIP_TRY(c, hipMemsetAsync(c->sr_one_scratch, 0, (size_t)ftkx::ONE_WORDS * sizeof(u64), c->stream));
  }
  Mesh m; fill_mesh(c, m);
  m.core_cells = P.cells;
  ftkx::OneArgs a;
  memset(&a, 0, sizeof(a));
  a.nsteps = n; a.nslices = (int)k;
  for (size_t j = 0; j < k; j ++) { const Slice &s = *sl[j]; a.slice[j] = ftkx::OneSlice{s.S, s.V, s.J, P.slice_ts[j], 0}; }
  size_t last = 0;
  for (int i = 0; i < n; i ++) {
    const size_t j0 = (size_t)(std::lower_bound(P.slice_ts.begin(), P.slice_ts.end(), ts[i]) - P.slice_ts.begin());
    const bool interval = (scopes[i] & FTKX_SCOPE_INTERVAL) != 0;
    while (last + 1 < k && P.slice_ts[last + 1] <= ts[i] + 1) last ++;
    a.step[i] = ftkx::OneStep{ts[i], scopes[i], (int)j0, interval ? (int)j0 + 1 : -1, (int)last, 0};
  }
  a.running_in = prev ? DBL_MAX : P.running_in;
  a.cap = 1.0 / (double)P.hint;
  a.running_from = prev ? c->sr_buf[prev->buf].results : nullptr;
  a.scratch = c->sr_one_scratch;
  a.results = B.results; a.h_results = B.h_results; a.nwords = nwords;
  a.flag = reinterpret_cast<unsigned *>(B.h_results + B.h_results_cap);
  a.seq = ++ B.seq;
  P.seq = a.seq;
  a.out = B.out; a.capacity = std::min<u64>(c->capacity, (u64)B.out_cap);
  a.fragile = c->d_fragile; a.fragile_capacity = c->fragile_capacity;
  // (the pass before left its records in device memory: their copy needs this pass's launch position, no more)
  if (before && before->open && before->copy_pending) {
    if (!c->sr_ev_fetched) HIP_TRY(c, hipEventCreateWithFlags(&c->sr_ev_fetched, hipEventDisableTiming));
    HIP_TRY(c, hipEventRecord(c->sr_ev_fetched, c->stream));
    HIP_TRY(c, hipStreamWaitEvent(c->sr_copy_stream, c->sr_ev_fetched, 0));
    series_queue_copy(c, *before, nullptr, 0);
  }
  if ((rc = wait_for_open_tails(c, &P))) return rc;         // (their tails share the fragile list and the counters)
  if (B.copy_out) { HIP_TRY(c, hipStreamWaitEvent(c->stream, B.ev_copied, 0)); B.copy_out = false; }
  const u64 bs = c->nd == 2 ? 128 : 64, nblocks = (P.cells * (u64)n + bs - 1) / bs;
  const int nwg = (int)std::max<u64>(8, std::min<u64>(256, (nblocks + 3) / 4));      // (four or more blocks per workgroup, kOneOwnBlocks at most)
  ev_begin(c, K_EXACT);
  ftkx::launch_series_one(m, a, nwg, c->stream);
  ev_end(c);
  HIP_TRY(c, hipGetLastError());
  P.uid = ++ c->sr_pass_uid;
  P.pipelined = true;
  P.open = true;
  return FTKX_OK;
}

// First half: everything of the pass is queued on the context's stream.  `prev`: the pass queued before this one and not yet collected,
// whose running minimum this one continues from (on the device), or nullptr: *running_in is the value.
// what a slab pass (ftkx_series_dist_*) adds to the plan of a pass: the halo slice and where the gathered contributions will be
struct DistPlan { int t_halo; int rank, nranks, upper; const u64 *gathered; u64 *contrib; void *masks_out; hipStream_t side; };

int series_queue_cull(ftkx_ctx *c, ftkx_series_pending &P);
int series_queue_tail(ftkx_ctx *c, ftkx_series_pending &P);

// First half, stage 1: the pass is planned -- buffers, descriptors -- and its begin and mask kernels are queued.
int series_plan(ftkx_ctx *c, ftkx_series_pending &P, const int *ts, const int *scopes, int n, double running_in, const ftkx_series_pending *prev, bool pipelined,
                ftkx_series_pending *before = nullptr /* the pass queued before this one, if it is still open */, const DistPlan *dist = nullptr)
{
  const int nd = c->nd;
  release_retired(c, P);                                     // (a slot that was abandoned with arrays still parked in it)
  P = ftkx_series_pending();
  if (dist) { P.dist = true; P.t_halo = dist->t_halo; P.dist_rank = dist->rank; P.dist_nranks = dist->nranks; P.dist_upper = dist->upper; P.gathered = dist->gathered; }
  P.ts.assign(ts, ts + n); P.scopes.assign(scopes, scopes + n); P.n = n;
  P.running_in = running_in; P.chained = prev != nullptr;
  int rc;
  if ((rc = series_steps(c, ts, scopes, n, P.slice_ts))) return rc;
  const std::vector<int> &slice_ts = P.slice_ts;
  const size_t k = slice_ts.size();
  P.k = k;
  std::vector<Slice *> sl(k);
  for (size_t j = 0; j < k; j ++) sl[j] = &c->slices.find(slice_ts[j])->second;

  // ---- is the device-driven form applicable? ---------------------------------------------------------------------------------------
  Mesh m; fill_mesh(c, m);
  const bool two_level = ftkx::masks_have_summary(m);
  u64 cells = 1;
  for (int d = 0; d < nd; d ++) cells *= (u64)c->core_sz[d];
  bool ok = series_applicable(c, ts, scopes, n, slice_ts, sl, cells, dist ? dist->t_halo : -1);
  if (prev && prev->by_host) ok = false;                     // (its running minimum will not be on the device)
  if (dist && !two_level) ok = false;                        // (the halo's masks travel as summaries + the words they do not describe)
  const unsigned long long hint = std::max<unsigned long long>(factor_of(running_in), 256ull);
  P.hint = hint; P.two_level = two_level; P.cells = cells; P.u_rows = m.u_rows;
  if (!ok) { P.by_host = true; P.open = true; return FTKX_OK; }
  if (series_one_eligible(c, P, n, k, cells, dist != nullptr)) return series_plan_one(c, P, ts, scopes, n, sl, prev, before);
  // slices whose masks and reduction stand from an earlier call (a streaming tracker: slice t of this step was slice t + 1 of the last)
  P.red_index.assign(k, -1);
  P.gen.assign(k, 0);
  std::vector<int> &red_index = P.red_index;
  // ... or are being built by the pass queued before this one, which is still out: their masks will be there in stream order, and what the
  // factor job needs of their reductions it reads from that pass's results block on the device (a streaming caller queues step t + 1
  // before it has collected step t: slice t + 1 is masked ONCE)
  std::vector<const u64 *> from_res(k, nullptr), from_max(k, nullptr);
  size_t ntodo = 0;
  for (size_t j = 0; j < k; j ++) {
    const Slice &s = *sl[j];
    bool ready = s.M && (!two_level || (s.U && s.u_rows == m.u_rows)) && s.mask_factor != 0 && s.mask_factor <= hint && !s.mask_big && (s.have_fused || s.have_res);
    if (!ready && before && before->open && !before->by_host && before->hint <= hint && before->two_level == two_level && before->u_rows == m.u_rows) {
      const auto it = std::lower_bound(before->slice_ts.begin(), before->slice_ts.end(), slice_ts[j]);
      if (it != before->slice_ts.end() && *it == slice_ts[j]) {
        const size_t jj = (size_t)(it - before->slice_ts.begin());
        if (before->red_index[jj] >= 0 && before->gen[jj] == s.mask_gen && s.M && (!two_level || s.U)) {
          const u64 *R = c->sr_buf[before->buf].results;
          from_res[j] = R + ftkx::SR_HEAD + (size_t)before->n + jj;
          from_max[j] = R + ftkx::SR_HEAD + (size_t)before->n + before->k + jj;
          ready = true;
        }
      }
    }
    if (dist && slice_ts[j] == dist->t_halo && s.sparse) {
      // the halo slice: its masks are imported behind the mask kernel (ftkx_series_dist_cull), its reduction is its owner's -- words 2, 3 of
      // the upper neighbour's contribution to the all_gather (the rank that owns the next timestep: not rank + 1 where slabs are empty)
      ready = true;
      from_res[j] = dist->gathered + (size_t)ftkx::kDistContrib * (size_t)dist->upper + 2;
      from_max[j] = from_res[j] + 1;
    }
    if (!ready) red_index[j] = (int)ntodo ++;
  }
  P.ntodo = ntodo;

  // ---- buffers (persistent; they only ever grow) -----------------------------------------------------------------------------------
  if ((rc = ensure_hit_buffer(c, std::max<u64>(c->capacity, 1u << 16)))) return rc;
  if ((rc = ensure_fragile(c, std::max<u64>(c->fragile_capacity, 1u << 12)))) return rc;
  if ((rc = ensure_list(c, std::max<u64>(c->list_capacity, 1u << 20))) || (rc = ensure_refine(c, std::max<u64>(c->refine_capacity, 1u << 20)))) return rc;
  // split?  (decided here: the begin kernel of a split pass leaves the counters to the tail stream)
  P.to_device = false;
  const unsigned long long mask_bytes_of_pass = (unsigned long long)ntodo * (unsigned long long)n_vertices(c) * 8ull * (c->scalar_mode == 1 ? 1ull : (unsigned long long)nd);
  {
    P.to_device = pipelined && c->stats.hits > 4096;
    // split?  FTKX_SERIES_HOOKS split = 0 never | 1 auto (default): where the size rule below says so AND the self-check found it no slower
    // | 4 on: the size rule alone, no self-check -- the deterministic setting | 2 whatever the size, no self-check (tests) | 3 = 4.
    // ftkx_series_split_decision says which way a context went and on what numbers.  (Profiling level 2 times the mask kernel only, with
    // events on the context's stream: they do not stand between the tail and anything.)
    const long split_mode = ftkx::env_hook("FTKX_SERIES_HOOKS", "split", 1);
    const unsigned long long mask_bytes = mask_bytes_of_pass;
    // hit-dense passes as well, where the mask launch reads 4 GB and more -- their chain at full grids, the records by way of the copy kernel:
    // double_gyre 2048 x 1024 x 128 0.83 -> 0.77 ms (its mask kernel 692 -> 752 us next to the chain's 550).  Round 5 saw 0.93-0.97 behind other
    // hit-dense contexts of the same process and kept them in order; round 6 (every open tail waited for by an unsplit pass, the library's
    // streams kept for the process) measures 0.764-0.774 alone, behind three other configurations and inside the driver's full line
    // (tools/dense_split.py) -- and auto's self-check keeps a context in order where it is not so.  Smaller hit-dense passes lose: woven
    // 1024^2 x 64, tail = mask kernel = 100 us, 0.205 -> 0.279 split.
    const bool sparse_now = c->sr_sparse && !P.to_device;
    // (a hit-dense chain is ~550 us next to a mask kernel -- double_gyre's 56 766 records --: only mask launches of 4 GB and more hide it.)
    // The decision, and in "auto" the self-check behind it: split_policy.hpp
    ftkxh::split_inputs in;
    in.mode = split_mode; in.pipelined = pipelined; in.dist = dist != nullptr; in.profiling_ok = c->profiling == 0 || c->profiling == 2;
    in.sparse_now = sparse_now; in.ntodo = ntodo; in.mask_bytes = mask_bytes;
    in.signature = ((unsigned long long)n << 48) ^ ((unsigned long long)ntodo << 32) ^ (unsigned long long)cells;
    const ftkxh::split_verdict v = ftkxh::split_decide(c->sr_cal, in);
    P.split = v.split; P.cal_kind = v.cal_kind; c->sr_split_forced = v.forced;
    P.split_sparse = P.split && c->sr_sparse && !P.to_device;
  }
  const bool before_split = before && before->open && before->split;
  // a slice whose masks this pass rebuilds while the tail of the pass before it -- on its own stream -- still reads them: fresh arrays here,
  // the old ones parked with that pass until it is completed
  if (before_split)
    for (size_t j = 0; j < k; j ++) {
      if (red_index[j] < 0 || !(sl[j]->M || sl[j]->U)) continue;
      if (!std::binary_search(before->slice_ts.begin(), before->slice_ts.end(), slice_ts[j])) continue;
      before->retired.push_back({sl[j]->M, sl[j]->U});
      sl[j]->M = nullptr; sl[j]->U = nullptr;
    }
  for (size_t j = 0; j < k; j ++) if (red_index[j] >= 0 && (rc = ensure_mask_arrays(c, *sl[j], two_level))) return rc;
  // order key -> bucket: at most 2^16 buckets over the keys this pass can produce
  const u64 max_key = (u64)n * cells * 64ull;
  int key_bits = 1;
  while (key_bits < 63 && (1ull << key_bits) < max_key) key_bits ++;
  // (a split pass: few records, and its scan -- one workgroup of four wavefronts next to a mask kernel -- pays several us per round of loads: 2^10)
  const int shift = std::max(0, key_bits - (P.split_sparse ? 10 : bins_log2()));
  const size_t nbins = (size_t)((max_key - 1) >> shift) + 1;
  P.nbins = nbins;
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
  const size_t nwords = (size_t)ftkx::SR_HEAD + (size_t)n + 2 * k + (dist ? (size_t)ftkx::kDistContrib * (size_t)dist->nranks : 0);
  P.nwords = nwords;
  // ---- descriptors: mask jobs | steps | slice table | step table, one pinned block fetched by a kernel -------------------------------
  const size_t off_jobs = 0, off_steps = align256(ntodo * sizeof(MaskJob)), off_slices = off_steps + align256((size_t)n * sizeof(Fields)),
               off_sinfo = off_slices + align256(k * sizeof(ftkx::SeriesSlice)), total = off_sinfo + align256((size_t)n * sizeof(ftkx::SeriesStep));
  P.off_steps = off_steps; P.off_slices = off_slices; P.off_sinfo = off_sinfo; P.total_desc = total; P.shift = shift;
  // A pass with many records, queued while another is still out: the record kernel leaves them in device memory and a small kernel on a
  // stream of its own takes them over PCIe -- next to the mask kernel of the pass queued behind.  (A record kernel that writes through
  // PCIe itself holds its stream for the transfer: 106 us of woven 1024^2 x 64's 363.)
  P.buf = (int)(&P - c->sr_pend);                             // (a pass's buffers go with its place in sr_pend: nothing to undo when a step below fails)
  ftkx_series_buffers &B = c->sr_buf[P.buf];
  if ((rc = ensure_series_buffers(c, B, nwords, total, P.to_device))) return rc;
  if (B.red_cap < std::max<size_t>(ntodo, 1)) {              // (the pass's own reduction slots: a buffer in use by an open pass is never this one -- two buffers, two passes)
    if (B.red) { HIP_TRY(c, hipFree(B.red)); B.red = nullptr; B.red_cap = 0; }
    HIP_TRY(c, hipMalloc((void **)&B.red, std::max<size_t>(ntodo, 1) * 128 * sizeof(u64)));
    B.red_cap = std::max<size_t>(ntodo, 1);
  }
  if (P.split || before_split) {
    if (!c->sr_tail_stream && (rc = aux_stream_get(c, true, &c->sr_tail_stream))) return rc;    // (high priority: the tail is a latency chain, it goes first wherever a slot frees up)
    for (ftkx_series_buffers &X : c->sr_buf)
      for (hipEvent_t *e : {&X.ev_masks, &X.ev_factors, &X.ev_tail}) if (!*e) HIP_TRY(c, hipEventCreateWithFlags(e, hipEventDisableTiming));
  }
  // (every other split pass of a SHORT mask launch works on the second set of counters and lists, on the second tail stream: two tails at a
  // time, each hidden behind two mask kernels.  Where one mask kernel hides a whole tail -- 2 GB and more -- the tails stay one behind the
  // other: two of them at once slow each other and the mask kernel down until the tails are what a pass takes: 256^3 x 16 0.41 -> 0.49 ms,
  // double_gyre 0.77 -> 0.97, in some runs and not in others)
  const bool two_tails = P.split_sparse && mask_bytes_of_pass < 2 * kSplitMinBytes;
  P.tail_set = two_tails ? (int)(c->sr_split_seq ++ & 1u) : 0;
  // (only where it is used: streams beyond the runtime's hardware queues share them, and a tail that shares its mask kernel's queue runs behind it)
  if (two_tails && !c->sr_tail_stream2 && (rc = aux_stream_get(c, true, &c->sr_tail_stream2))) return rc;
  P.before_buf = before_split ? before->buf : -1;
  if (P.tail_set == 1 && (rc = ensure_set1(c))) return rc;
  if (P.to_device && !c->sr_copy_stream && (rc = aux_stream_get(c, false, &c->sr_copy_stream))) return rc;
  fill_mesh(c, m);                                           // (the buffers may have moved)
  m.hist = tail_view(c, P).hist; m.hist_shift = shift; m.core_cells = cells;
  {
    MaskJob *jobs = (MaskJob *)((char *)B.h_desc + off_jobs);
    Fields *steps = (Fields *)((char *)B.h_desc + off_steps);
    ftkx::SeriesSlice *ss = (ftkx::SeriesSlice *)((char *)B.h_desc + off_slices);
    ftkx::SeriesStep *si = (ftkx::SeriesStep *)((char *)B.h_desc + off_sinfo);
    const double cap = 1.0 / (double)hint;
    for (size_t j = 0; j < k; j ++) {
      const Slice &s = *sl[j];
      ss[j].t = slice_ts[j]; ss[j].red_index = red_index[j];
      ss[j].known_res = DBL_MAX; ss[j].known_max = 0.0;
      ss[j].from_res = from_res[j]; ss[j].from_max = from_max[j];
      if (s.have_res) { ss[j].known_res = s.res < cap ? s.res : DBL_MAX; ss[j].known_max = s.maxabs; }
      else if (red_index[j] < 0 && !from_res[j]) { ss[j].known_res = s.res_below; ss[j].known_max = s.maxabs; }
      if (from_res[j] && s.sparse) { ss[j].known_res = DBL_MAX; ss[j].known_max = 0.0; }      // (the halo slice: nothing of an earlier pass stands)
      if (red_index[j] >= 0)
        jobs[red_index[j]] = with_lean_thresholds(MaskJob{s.S, s.V, s.M, two_level ? s.U : nullptr, B.red + (size_t)red_index[j] * 128, cap, HUGE_VAL}, m);   // rule off: validated by the factor kernel
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
  const MaskJob *d_jobs = (const MaskJob *)((char *)B.d_desc + off_jobs);
  Fields *d_steps = (Fields *)((char *)B.d_desc + off_steps);
  const ftkx::SeriesSlice *d_slices = (const ftkx::SeriesSlice *)((char *)B.d_desc + off_slices);
  const ftkx::SeriesStep *d_sinfo = (const ftkx::SeriesStep *)((char *)B.d_desc + off_sinfo);

  // ---- the whole pass, queued ------------------------------------------------------------------------------------------------------
  unsigned *flag = reinterpret_cast<unsigned *>(B.h_results + B.h_results_cap);
  const unsigned seq = ++ B.seq;
  P.seq = seq;
  // (the masks of the slices this pass rebuilds are nobody's until it has been collected; whatever else touches masks meanwhile bumps the
  // epoch, and the marks of this pass are then not applied)
  for (size_t j = 0; j < k; j ++) {
    if (red_index[j] >= 0) { sl[j]->mask_factor = 0; sl[j]->have_fused = false; sl[j]->mask_gen = ++ c->mask_epoch; }
    if (dist && slice_ts[j] == dist->t_halo && sl[j]->sparse) {
      // (the halo's masks of this pass: built by its owner under the owner's hint, which the import checks against ours on the device)
      sl[j]->mask_factor = hint; sl[j]->mask_big = false; sl[j]->u_rows = m.u_rows; sl[j]->have_res = false; sl[j]->have_fused = false;
      sl[j]->mask_gen = ++ c->mask_epoch;
    }
    P.gen[j] = sl[j]->mask_gen;
  }
  // (the pass queued before this one left its records in device memory: their way over PCIe starts behind this pass's descriptor fetch,
  // which the begin kernel announces in a word of device memory -- series_copy_out_kernel)
  const bool copy_behind = before && before->open && before->copy_pending;
  // (single-rank passes: the copy waits on a word of device memory.  Slab passes: their begin kernel may sit behind the pass before it with
  // its messages from other ranks for as long as a peer lags -- sixteen workgroups would spin for that long, and give up in the end; the
  // copy is ordered behind the begin kernel by an event instead, 5 us on a path that waits for the network anyway)
  const bool copy_by_tail = copy_behind && before->split;      // (its finish kernel is on the tail stream: the copy goes behind that stream's event)
  const bool copy_by_event = copy_behind && !copy_by_tail && (dist != nullptr || before->dist);
  const bool copy_by_flag = copy_behind && !copy_by_event && !copy_by_tail;
  if (copy_by_flag && !c->sr_fetch_flag) { HIP_TRY(c, hipMalloc((void **)&c->sr_fetch_flag, 2 * sizeof(unsigned))); HIP_TRY(c, hipMemsetAsync(c->sr_fetch_flag, 0, 2 * sizeof(unsigned), c->stream)); HIP_TRY(c, hipStreamSynchronize(c->stream)); }   // (once per context; waited for: the copy stream reads it)
  const unsigned fetch_val = copy_by_flag ? ++ c->sr_fetch_seq : 0u;
  // (the pass before this one has its tail on the tail stream: a pass that is not split itself shares the counters with it in STREAM order,
  // so the context's stream waits for that tail; a split pass only needs that pass's cull -- its mask kernel must not start before the fused
  // tail behind that cull can be placed -- and zeroes the counters on the tail stream, behind it)
  if (!P.split && (rc = wait_for_open_tails(c, &P))) return rc;
  // (a split pass: what its TAIL owns -- counters, histogram, the results block, which the tail of the pass before may still be reading as the
  // block it continues from -- is zeroed on the tail stream)
  ftkx::launch_series_begin(P.split ? nullptr : c->d_counters, B.red, ntodo * 64, P.split ? nullptr : c->sr_hist, P.split ? 0 : nbins + 1, P.split ? nullptr : B.results, P.split ? 0 : nwords,
                            c->stream, B.h_desc, B.d_desc, total, copy_by_flag ? c->sr_fetch_flag : nullptr, fetch_val);
  c->sr_lists_owner = 0;                                      // (the begin kernel zeroes the counters and the histogram: they are nobody's until this pass's cull is queued)
  if (copy_by_flag) series_queue_copy(c, *before, c->sr_fetch_flag, fetch_val);
  if (copy_by_tail) {
    HIP_TRY(c, hipStreamWaitEvent(c->sr_copy_stream, c->sr_buf[before->buf].ev_tail, 0));
    series_queue_copy(c, *before, nullptr, 0);
  }
  if (copy_by_event) {
    if (!c->sr_ev_fetched) HIP_TRY(c, hipEventCreateWithFlags(&c->sr_ev_fetched, hipEventDisableTiming));
    HIP_TRY(c, hipEventRecord(c->sr_ev_fetched, c->stream));
    HIP_TRY(c, hipStreamWaitEvent(c->sr_copy_stream, c->sr_ev_fetched, 0));
    series_queue_copy(c, *before, nullptr, 0);
  }
  if (dist && dist->masks_out) {
    // A slab pass with a lower neighbour: the FIRST slice's masks are that neighbour's halo.  They are built first, by a launch of their
    // own, and packed into the message right behind it -- the message can then cross xGMI (on the caller's side stream, which is made to
    // wait for the export here) while the masks of the slab's other slices are still being built.
    if (!B.dist_block) { HIP_TRY(c, hipMalloc((void **)&B.dist_block, (size_t)ftkx::DB_N * sizeof(u64))); HIP_TRY(c, hipMemsetAsync(B.dist_block, 0, (size_t)ftkx::DB_N * sizeof(u64), c->stream)); }
    size_t ub, cap, off_idx, off_words, total_msg;
    if (!packed_layout(c, m, &ub, &cap, &off_idx, &off_words, &total_msg)) return fail(c, FTKX_E_UNSUPPORTED, "slab pass: this mesh has no summarised masks");
    const bool first_now = red_index[0] == 0;              // (its masks are built in this pass: job 0)
    size_t done = 0;
    if (first_now) { ev_begin(c, K_MASK); ftkx::launch_masks(m, d_jobs, 1, c->stream); ev_end(c); done = 1; }
    char *out = (char *)dist->masks_out;
    int flog = 0; while (flog < 63 && (1ull << flog) < hint) flog ++;
    ftkx::launch_dist_export(m, sl[0]->U, sl[0]->M, ub, (u64 *)out, (unsigned *)(out + off_idx), (u64 *)(out + off_words), cap, flog, B.dist_block, c->stream);
    if (dist->side) {
      if (!B.ev_export) HIP_TRY(c, hipEventCreateWithFlags(&B.ev_export, hipEventDisableTiming));
      HIP_TRY(c, hipEventRecord(B.ev_export, c->stream));
      HIP_TRY(c, hipStreamWaitEvent(dist->side, B.ev_export, 0));
    }
    if (ntodo > done) { ev_begin(c, K_MASK); ftkx::launch_masks(m, d_jobs + done, (int)(ntodo - done), c->stream); ev_end(c); }
  } else if (ntodo) { ev_begin(c, K_MASK); ftkx::launch_masks(m, d_jobs, (int)ntodo, c->stream); ev_end(c); }
  if (P.split) {
    // the tail's side: behind the masks (an event), the counters and the histogram zeroed there
    HIP_TRY(c, hipEventRecord(B.ev_masks, c->stream));
    const TailView T = tail_view(c, P);
    HIP_TRY(c, hipStreamWaitEvent(tail_stream(c, P), B.ev_masks, 0));
    ftkx::launch_series_tail_begin(T.counters, T.hist, nbins + 1, B.results, nwords, tail_stream(c, P));
  }
  P.running_from = prev ? c->sr_buf[prev->buf].results : nullptr;
  P.pipelined = pipelined;
  HIP_TRY(c, hipGetLastError());
  (void)d_steps; (void)d_slices; (void)d_sinfo; (void)flag;
  return FTKX_OK;
}

void series_mesh(ftkx_ctx *c, const ftkx_series_pending &P, Mesh &m)
{
  fill_mesh(c, m);
  const TailView T = tail_view(c, P);
  m.counters = T.counters; m.pass = T.pass; m.fragile = T.fragile;
  m.hist = T.hist; m.hist_shift = P.shift; m.core_cells = P.cells;
}

// stage 2: the cull, with the factor job riding in it
int series_queue_cull(ftkx_ctx *c, ftkx_series_pending &P)
{
  const int nd = c->nd, n = P.n;
  const size_t k = P.k;
  ftkx_series_buffers &B = c->sr_buf[P.buf];
  Mesh m; series_mesh(c, P, m);
  Fields *d_steps = (Fields *)((char *)B.d_desc + P.off_steps);
  const ftkx::SeriesSlice *d_slices = (const ftkx::SeriesSlice *)((char *)B.d_desc + P.off_slices);
  const ftkx::SeriesStep *d_sinfo = (const ftkx::SeriesStep *)((char *)B.d_desc + P.off_sinfo);
  hipStream_t st = tail_stream(c, P);
  const TailView T = tail_view(c, P);
  if (!P.split) ev_begin(c, K_CULL);
  // (two tails at a time: this pass's factor job continues from the running minimum the factor job of the pass before it leaves -- on the
  // other tail stream)
  if (P.split && P.before_buf >= 0 && tail_stream(c, c->sr_pend[P.before_buf]) != st) HIP_TRY(c, hipStreamWaitEvent(st, c->sr_buf[P.before_buf].ev_factors, 0));
  P.uid = ++ c->sr_pass_uid;
  c->sr_lists_owner = P.uid;                                  // (from here on the counters and lists hold this pass's cull)
  {
    // the factors need what the mask kernel left, the cull does not need the factors: one extra workgroup of the cull kernel forms them
    // (series_device.hpp) -- a kernel boundary and a one-workgroup launch less than the factor kernel behind the cull
    const double safe_m = (double)(nd == 3 ? ftkx::kSafeM3 : ftkx::kSafeM2);
    const u64 *running_from = P.running_from;
    ftkx::FactorJob fj;
    memset(&fj, 0, sizeof(fj));
    fj.steps = d_steps; fj.slices = d_slices; fj.sinfo = d_sinfo; fj.red = B.red; fj.running_from = running_from; fj.results = B.results; fj.counters = T.counters;
    fj.running_in = running_from ? DBL_MAX : P.running_in; fj.safe_m = safe_m; fj.nsteps = n; fj.nslices = (int)k;
    const bool fold_on = ftkx::env_hook("FTKX_SERIES_HOOKS", "fold", 1) != 0;
    fj.enabled = (fold_on && k <= (size_t)ftkx::kFoldMaxSlices) ? 1 : 0;
    if (P.two_level) ftkx::launch_cull_coarse(m, d_steps, n, T.refine, c->refine_capacity, st, &fj);
    else ftkx::launch_cull(m, d_steps, n, T.list, c->list_capacity, st, &fj);
    if (!fj.enabled) ftkx::launch_series_factors(d_steps, n, d_slices, (int)k, d_sinfo, B.red, fj.running_in, running_from, safe_m, B.results, T.counters, st);
  }
  if (P.split) HIP_TRY(c, hipEventRecord(B.ev_factors, st));
  else ev_end(c);
  HIP_TRY(c, hipGetLastError());
  return FTKX_OK;
}

// stage 3: everything behind the cull -- the fused tail for sparse data, or refine, exact test, ordering, records, finish
int series_queue_tail(ftkx_ctx *c, ftkx_series_pending &P)
{
  ftkx_series_buffers &B = c->sr_buf[P.buf];
  Mesh m; series_mesh(c, P, m);
  Fields *d_steps = (Fields *)((char *)B.d_desc + P.off_steps);
  unsigned *flag = reinterpret_cast<unsigned *>(B.h_results + B.h_results_cap);
  const unsigned seq = P.seq;
  const bool two_level = P.two_level;
  const size_t nwords = P.nwords;
  hipStream_t st = tail_stream(c, P);
  if (!P.split) ev_begin(c, K_EXACT);
  // sparse data: one kernel does the rest of the pass (and the kernels below leave at once).  (A pass that has just found far more
  // survivors than the fused kernel takes does not launch it for a while: finding nothing to do costs its 256 workgroups of 256-VGPR
  // wavefronts ~15 us.)  The fused tail finished the last pass too: this one is queued WITHOUT the seven kernels behind it -- each of them
  // costs a few us just to find out that it has nothing to do.  Should the fused tail decline this time, it says so itself and the rest is
  // queued then (or, with another pass queued behind already, the host-driven batch sweeps the steps).
  bool small_now = false;
  P.short_chain = short_chain_now(c, P.to_device, &small_now);
  if (P.split) { small_now = false; P.short_chain = false; }      // (the fused tail does not fit next to a mask kernel: the chain)
  if (c->sr_skip_small > 0) c->sr_skip_small --;
  P.small_now = small_now;
  if (small_now) ftkx::launch_series_small(m, two_level ? ftkx::coarse_view(m) : m, d_steps, two_level, c->d_refine, c->d_list, B.out, B.results, nwords,
                                          B.h_results, flag, seq, reinterpret_cast<unsigned *>(c->d_counters + ftkx::CNT_SMALL_DONE), P.short_chain, st);
  if (!P.short_chain) series_queue_rest(c, P, m, seq);
  if (P.split) HIP_TRY(c, hipEventRecord(B.ev_tail, st));
  P.copy_pending = P.to_device;
  HIP_TRY(c, hipGetLastError());
  return FTKX_OK;
}

// First half: everything of the pass is queued on the context's stream.  `prev`: the pass queued before this one and not yet collected,
// whose running minimum this one continues from (on the device), or nullptr: *running_in is the value.
int series_submit(ftkx_ctx *c, ftkx_series_pending &P, const int *ts, const int *scopes, int n, double running_in, const ftkx_series_pending *prev, bool pipelined,
                  ftkx_series_pending *before = nullptr /* the pass queued before this one, if it is still open */)
{
  int rc = series_plan(c, P, ts, scopes, n, running_in, prev, pipelined, before, nullptr);
  if (rc || P.by_host || P.one) return rc;
  if ((rc = series_queue_cull(c, P)) || (rc = series_queue_tail(c, P))) return rc;
  P.open = true;
  return FTKX_OK;
}

// Second half: the host waits (once) for the pass, marks the slices, patches what has to be patched on the host, or has the host-driven
// batch sweep the steps again where a kernel raised a flag.
int series_complete(ftkx_ctx *c, ftkx_series_pending &P, double *running_resolution, unsigned long long *factors, const ftkx_cp_t **out, size_t *n_out)
{
  const int nd = c->nd, n = P.n;
  const size_t k = P.k;
  P.open = false;
  double running = *running_resolution;                       // (what the pass started from: the host-driven batch, if it comes to that, starts there too)
  // (the split pass's self-check samples the time between two completions by the plain way out of this function: any other way out breaks the chain)
  const double last_complete_s = c->sr_last_complete_s;
  const int last_complete_kind = c->sr_last_complete_kind;
  c->sr_last_complete_s = 0; c->sr_last_complete_kind = 0;
  if (P.by_host) {
    int rc = series_by_host(c, P.ts.data(), P.scopes.data(), n, P.slice_ts, &running, factors, out, n_out);
    if (rc == FTKX_OK) *running_resolution = running;
    return rc;
  }
  ftkx_series_buffers &B = c->sr_buf[P.buf];
  unsigned *flag = reinterpret_cast<unsigned *>(B.h_results + B.h_results_cap);
  Mesh m; fill_mesh(c, m);
  { const TailView T = tail_view(c, P); m.counters = T.counters; m.pass = T.pass; m.fragile = T.fragile; m.hist = T.hist; }
  m.hist_shift = P.shift; m.core_cells = P.cells;
  if (const char *why = ftkx::wait_flag(flag, P.seq, tail_stream(c, P))) return fail(c, FTKX_E_DEVICE, "ftkx_sweep_series: %s", why);
  release_retired(c, P);                                     // (the tail that read them is through)
  // (with no pass left open the tail stream is at its end: waited for, so that whatever the caller does next on the context's stream --
  // a host-driven batch, a pass that is not split -- finds the counters and lists idle)
  if (P.split && c->sr_open == 0) { HIP_TRY(c, hipStreamSynchronize(c->sr_tail_stream)); if (c->sr_tail_stream2) HIP_TRY(c, hipStreamSynchronize(c->sr_tail_stream2)); }

  // ---- what came back ----------------------------------------------------------------------------------------------------------------
  // (whoever stored the flag -- the fused tail, finishing or declining, or the finish kernel -- copied the whole results block first: the
  // reductions, the status, a slab pass's gathered contributions.  Everything below that does not depend on HOW the records get made comes
  // first, so that every way out of this function has done it.)
  const u64 *R = B.h_results;
  unsigned long long status = R[ftkx::SR_STATUS];
  c->sr_last_status = status;
  // the masks and reductions of this pass stand whichever way the records are made: the slices are marked like ftkx_slices_prepare marks
  // them -- slice by slice, unless something has rebuilt or dropped its masks, or replaced it, since the pass was queued
  {
    for (size_t j = 0; j < k; j ++) {
      if (P.red_index[j] < 0) continue;
      auto it = c->slices.find(P.slice_ts[j]);
      if (it == c->slices.end() || it->second.mask_gen != P.gen[j]) continue;
      Slice &s = it->second;
      double r, x;
      memcpy(&r, &R[ftkx::SR_HEAD + n + j], 8); memcpy(&x, &R[ftkx::SR_HEAD + n + k + j], 8);
      if (std::isinf(x)) { s.mask_factor = 0; s.have_fused = false; continue; }     // the fused max is not the max FINITE |v|: the host-driven path reduces it exactly
      s.res_below = r; s.fused_factor = P.hint; s.have_fused = true;
      if (!s.have_res) s.maxabs = x;
      s.mask_factor = overflow_free(nd, s.maxabs, P.hint) ? P.hint : 0;
      s.mask_big = false; s.u_rows = P.u_rows;
    }
  }
  c->sr_last_buf = P.buf; c->sr_last_nranks = 0;
  if (P.dist) {
    // a slab pass: what the other ranks contributed came back with the results -- the running minimum before this slab and the halo
    // slice's reduction are what the host-driven batch needs, should it have to take the pass over
    c->sr_last_nranks = P.dist_nranks; c->sr_last_gathered_off = (size_t)ftkx::SR_HEAD + (size_t)n + 2 * k;
    const u64 *G = R + c->sr_last_gathered_off;
    for (int r = 0; r < P.dist_rank; r ++) { double v; memcpy(&v, &G[(size_t)ftkx::kDistContrib * r], 8); running = std::min(running, v); }
    auto it = P.t_halo >= 0 ? c->slices.find(P.t_halo) : c->slices.end();
    if (it != c->slices.end() && it->second.sparse && P.dist_upper >= 0 && P.dist_upper < P.dist_nranks) {
      Slice &h = it->second;
      double r, x;
      memcpy(&r, &G[(size_t)ftkx::kDistContrib * P.dist_upper + 2], 8); memcpy(&x, &G[(size_t)ftkx::kDistContrib * P.dist_upper + 3], 8);
      h.have_res = true; h.res = r; h.maxabs = x; h.res_below = r;
    }
    if (status & ftkx::SERIES_HALO_FULL) {
      c->sr_short_chain = false;
      if (P.short_chain && !P.split) ev_end(c);
      ev_harvest(c, false);
      *running_resolution = running;
      return fail(c, FTKX_E_NOSLICE, "slab pass: the halo slice %d is needed as a whole (request -1: too many surviving cells, a mask message that did not fit, or masks the host rebuilds): nothing was swept", P.t_halo);
    }
  }
  if (P.short_chain) {
    if ((status & ftkx::SERIES_TAIL_PENDING) && (c->sr_open > 0 || c->sr_lists_owner != P.uid)) {
      // (the rest of this pass cannot be queued: the counters and lists are not this pass's any more -- the pass queued behind it has them
      // now, or the host-driven batch took them when the pass BEFORE this one fell back (two short-chain passes that both declined,
      // completed back to back).  The host-driven batch sweeps the steps; it happens when sparse data turns dense.  A slab pass: from the
      // running minimum the lower ranks' contributions give, over a halo slice that has its patches -- both settled above)
      c->sr_short_chain = false;
      if (!P.split) ev_end(c);
      int rc = series_by_host(c, P.ts.data(), P.scopes.data(), n, P.slice_ts, &running, factors, out, n_out);
      if (rc == FTKX_OK) *running_resolution = running;
      return rc;
    }
    if (status & ftkx::SERIES_TAIL_PENDING) {
      const unsigned seq2 = ++ B.seq;
      series_queue_rest(c, P, m, seq2);
      if (P.split) HIP_TRY(c, hipEventRecord(B.ev_tail, tail_stream(c, P)));
      HIP_TRY(c, hipGetLastError());
      if (const char *why = ftkx::wait_flag(flag, seq2, tail_stream(c, P))) return fail(c, FTKX_E_DEVICE, "ftkx_sweep_series: %s", why);
      status = R[ftkx::SR_STATUS];                           // (the finish kernel's copy of the block: the same reductions, the final counters)
      c->sr_last_status = status;
    } else if (!P.split) ev_end(c);
  }
  ev_harvest(c, false);
  const unsigned long long redo = ftkx::SERIES_AMBIGUOUS | ftkx::SERIES_MASKS_INVALID | ftkx::SERIES_INF | ftkx::SERIES_OVERFLOW;
  if (status & redo) {
    c->sr_short_chain = false; c->sr_sparse = false;
    if (P.one) c->sr_one_off = 16;                           // (more hits than a workgroup parks, or the device was not this kernel's alone: the usual way for a while)
    int rc;
    if (status & ftkx::SERIES_OVERFLOW) {                    // grow what was too small (the host-driven batch would find out the same way, one replay later)
      const u64 *cnt = R + ftkx::SR_COUNTERS;
      const u64 hits = cnt[ftkx::CNT_PASS], listed = cnt[ftkx::CNT_LIST_PEAK], refined = cnt[ftkx::CNT_REFINE_PEAK], fragile = cnt[ftkx::CNT_FRAGILE];
      if (c->sr_open > 0) {                                  // (a pass queued behind this one still uses the buffers)
        HIP_TRY(c, hipStreamSynchronize(c->stream));
        if (c->sr_tail_stream) { HIP_TRY(c, hipStreamSynchronize(c->sr_tail_stream)); if (c->sr_tail_stream2) HIP_TRY(c, hipStreamSynchronize(c->sr_tail_stream2)); }
      }
      if (hits > c->capacity && (rc = ensure_hit_buffer(c, hits + hits / 8 + 1024))) return rc;
      if (listed > c->list_capacity && (rc = ensure_list(c, listed + listed / 8 + 1024))) return rc;
      if (refined > c->refine_capacity && (rc = ensure_refine(c, refined + refined / 8 + 1024))) return rc;
      if (fragile > c->fragile_capacity && (rc = ensure_fragile(c, fragile + fragile / 8 + 1024))) return rc;
    }
    rc = series_by_host(c, P.ts.data(), P.scopes.data(), n, P.slice_ts, &running, factors, out, n_out);
    if (rc == FTKX_OK) *running_resolution = running;
    return rc;
  }
  c->sr_last_path = (status & ftkx::SERIES_ONE) ? 4 : P.split ? 5 : (status & ftkx::SERIES_EARLY) ? 2 : 1;
  {
    // the split pass's self-check: the time since the last completion is a sample of this pass's form if the pipeline was full all the while
    // (another pass is open now and one was when the last one completed) and the last completion was of the same form
    const double now = std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
    ftkxh::split_sample(c->sr_cal, P.cal_kind, now, last_complete_s, last_complete_kind, c->sr_open > 0);
    c->sr_last_complete_kind = P.cal_kind;
    c->sr_last_complete_s = c->sr_open > 0 ? now : 0.0;
  }
  if (!P.one && !P.split) c->sr_short_chain = (status & ftkx::SERIES_EARLY) != 0;      // (a split pass says nothing about the fused tail)
  if (!P.one) {
    const u64 *cn = R + ftkx::SR_COUNTERS;
    c->sr_sparse = (status & ftkx::SERIES_EARLY) != 0 || (R[ftkx::SR_NHITS] <= 1024 && (P.two_level ? cn[ftkx::CNT_REFINE_PEAK] : cn[ftkx::CNT_LIST_PEAK]) <= 4 * 2048ull);
  }
  const u64 *cnt = R + ftkx::SR_COUNTERS;
  if (!(status & ftkx::SERIES_EARLY) && (P.two_level ? cnt[ftkx::CNT_REFINE_PEAK] : cnt[ftkx::CNT_LIST_PEAK]) > 4 * 2048ull) c->sr_skip_small = 16;
  // (few coarse cells, many records in them: the fused tail declined late, tens of microseconds lost.  Twice in a row: the series is like that)
  c->sr_late_streak = (status & ftkx::SERIES_LATE_DECLINE) ? c->sr_late_streak + 1 : 0;
  if (c->sr_late_streak >= 2) c->sr_skip_small = 16;
  const size_t nrec = (size_t)R[ftkx::SR_NHITS];
  ftkx_cp_t *H = B.out;
  if (P.to_device) {                                         // (the copy kernel; the mask kernel of the pass queued behind this one is running meanwhile)
    if (P.copy_pending) series_queue_copy(c, P, nullptr, 0);    // (no pass was queued behind this one)
    if (const char *why = ftkx::wait_flag(flag + 2, P.seq, c->sr_copy_stream)) {
      // the copy stream drained without the flag: the copy kernel gave up waiting for the begin kernel of the pass queued behind this one (a
      // caller that took seconds to submit it).  This pass itself is through -- its own flag was waited for above --, so the copy needs no
      // wait now: once more, unconditionally
      if (hipStreamQuery(c->sr_copy_stream) != hipSuccess) return fail(c, FTKX_E_DEVICE, "ftkx_sweep_series: %s", why);
      series_queue_copy(c, P, nullptr, 0);
      if (const char *why2 = ftkx::wait_flag(flag + 2, P.seq, c->sr_copy_stream)) return fail(c, FTKX_E_DEVICE, "ftkx_sweep_series: %s", why2);
    }
    B.copy_out = false;                                      // (the copy is through: the pass that takes these buffers next need not wait for its event)
  }
  memset(&c->stats, 0, sizeof(c->stats));
  {
    const u64 n_ord = nd == 2 ? 2 : 6, n_int = nd == 2 ? 10 : 54;
    for (int i = 0; i < n; i ++) { c->stats.cells += P.cells; c->stats.work_items += P.cells * (((P.scopes[i] & 1) ? n_ord : 0) + ((P.scopes[i] & 2) ? n_int : 0)); }
  }
  c->stats.cull_enabled = 1;
  c->stats.hits = nrec;
  c->stats.cells_survived = cnt[ftkx::CNT_CELLS_SURVIVED];
  c->stats.simplices_tested = cnt[ftkx::CNT_SIMPLICES_TESTED];
  // 3D records whose class hangs on the last bits of pow / acos / cos: classified with the libm the reference runs on (collect.hip does
  // the same), patched in place -- the records are in host memory already
  const size_t nf = (size_t)R[ftkx::SR_NFRAGILE];
  for (size_t i = 0; i < nf; i ++) {
    const u64 *e = R + P.nwords + i * 10;
    double A[3][3];
    memcpy(A, e + 1, sizeof(A));
    if (e[0] < nrec) H[e[0]].type = (unsigned)ftkx::classify3(A, c->opt.jacobian_symmetric != 0);
  }
  c->stats.reclassified = nf;
  if (status & ftkx::SERIES_FIX_ORDER) {
    // A bucket too full to rank on the device: its records sit in their own run of the output, unordered among themselves; everything
    // before the run is smaller, everything behind it larger.  Find each such run from an inversion, widen it until both ends are in
    // order with their neighbours, sort it.
    auto less = [](const ftkx_cp_t &p, const ftkx_cp_t &q) { return p.tag < q.tag; };
    ftkx_cp_t *h = H;
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
  if (out) *out = H;
  if (n_out) *n_out = nrec;
  return FTKX_OK;
}

}  // namespace

// a slab pass between ftkx_series_dist_begin and _finish: its begin kernel has wiped the counters, its slot of sr_pend is taken, and it does
// not count in sr_open yet -- no other pass may be queued, swept or completed until it has been finished (or aborted)
static bool slab_half_queued(const ftkx_ctx *c)
{
  const ftkx_series_pending &Q = c->sr_pend[c->sr_place(c->sr_open)];
  return Q.dist && Q.dist_stage > 0 && Q.dist_stage < 4;
}

extern "C" {

int ftkx_sweep_series_submit(ftkx_ctx *c, const int *ts, const int *scopes, int n, const double *running_resolution)
{
  if (!c || n <= 0 || !ts || !scopes) return fail(c, FTKX_E_INVALID, "ftkx_sweep_series_submit: null argument or no steps");
  if (!c->mesh_set) return fail(c, FTKX_E_INVALID, "sweep: call ftkx_set_mesh first");
  if (!c->pending.empty()) return fail(c, FTKX_E_INVALID, "ftkx_sweep_series_submit: sweeps pending, collect first");
  if (c->sr_open >= ftkx_ctx::kPlaces) return fail(c, FTKX_E_INVALID, "ftkx_sweep_series_submit: three passes are open, complete one first");
  if (slab_half_queued(c)) return fail(c, FTKX_E_INVALID, "ftkx_sweep_series_submit: a slab pass is half queued (ftkx_series_dist_finish or ftkx_sweep_series_abort first)");
  if (!running_resolution && c->sr_open == 0) return fail(c, FTKX_E_INVALID, "ftkx_sweep_series_submit: no pass open to continue from, give the running resolution");
  if (running_resolution && !(*running_resolution > 0)) return fail(c, FTKX_E_INVALID, "ftkx_sweep_series_submit: the running resolution must be positive (DBL_MAX: none yet)");
  c->ahead.clear(); c->announced.clear();
  HIP_TRY(c, hipSetDevice(c->device));
  ftkx_series_pending *before = c->sr_open > 0 ? &c->sr_pend[c->sr_place(c->sr_open - 1)] : nullptr;
  const ftkx_series_pending *prev = (!running_resolution) ? before : nullptr;
  ftkx_series_pending &P = c->sr_pend[c->sr_place(c->sr_open)];
  // (a chain starts here: what the host knows of its running minimum is what the caller says -- not what an earlier series left behind, whose
  // smaller value would make the hint of the passes chained behind this one larger than their factor)
  if (running_resolution) c->sr_last_running = *running_resolution;
  const double running_in = running_resolution ? *running_resolution : (c->sr_last_running > 0 ? c->sr_last_running : DBL_MAX);
  int rc = series_submit(c, P, ts, scopes, n, running_in, prev, true, before);
  if (rc) { P.open = false; return rc; }
  c->sr_open ++;
  return FTKX_OK;
}

int ftkx_sweep_series_complete(ftkx_ctx *c, double *running_resolution, unsigned long long *factors, const ftkx_cp_t **out, size_t *n_out)
{
  if (!c || !running_resolution) return fail(c, FTKX_E_INVALID, "null argument");
  if (out) *out = nullptr;
  if (n_out) *n_out = 0;
  if (c->sr_open == 0) return fail(c, FTKX_E_INVALID, "ftkx_sweep_series_complete: no pass open");
  if (slab_half_queued(c)) return fail(c, FTKX_E_INVALID, "ftkx_sweep_series_complete: a slab pass is half queued (ftkx_series_dist_finish or ftkx_sweep_series_abort first)");
  HIP_TRY(c, hipSetDevice(c->device));
  ftkx_series_pending &P = c->sr_pend[c->sr_head];
  c->sr_head = (c->sr_head + 1) % ftkx_ctx::kPlaces; c->sr_open --;
  *running_resolution = P.chained ? c->sr_last_running : P.running_in;   // (a chained pass continues from what the pass before it returned)
  int rc = series_complete(c, P, running_resolution, factors, out, n_out);
  if (rc == FTKX_OK) c->sr_last_running = *running_resolution;
  return rc;
}

// ---- the slab pass: one rank's part of a series cut into timestep slabs, queued in stages -----------------------------------------------
namespace {
int factor_log2_of(unsigned long long f) { int b = 0; while (b < 63 && (1ull << b) < f) b ++; return b; }
size_t dist_cells_cap(const ftkx_ctx *c)
{
  // cells a request carries: the reply has a FIXED size (the owner never sees the count on the host) -- at most 1 MiB of patches
  // (~15 us of one xGMI link) and at most a sixteenth of the slice it stands for
  const size_t pd = ftkx_patch_doubles(c) * sizeof(double);
  const long forced = getenv("FTKX_DIST_CELLS") ? atol(getenv("FTKX_DIST_CELLS")) : 0;      // (tests: a small request forces the whole-slice way, a large one keeps small meshes compact; read by every context afresh)
  if (forced > 0) return (size_t)forced;
  const size_t slice_bytes = n_vertices(c) * (size_t)(c->scalar_mode == 1 ? 1 : c->nd) * sizeof(double);
  return pd ? std::max<size_t>(16, std::min<size_t>(4096, std::min<size_t>((size_t)1 << 20, slice_bytes / 16) / pd)) : 0;
}
ftkx_series_pending *dist_pending(ftkx_ctx *c, int stage, const char *who)
{
  ftkx_series_pending &P = c->sr_pend[c->sr_place(c->sr_open)];
  if (!P.dist || P.dist_stage != stage) { fail(c, FTKX_E_INVALID, "%s: no slab pass at that stage (ftkx_series_dist_begin, _cull, _serve, _finish in this order)", who); return nullptr; }
  return &P;
}
}

size_t ftkx_series_dist_cells(const ftkx_ctx *c) { return c && c->mesh_set && c->scalar_mode >= 0 ? dist_cells_cap(c) : 0; }

int ftkx_series_dist_begin(ftkx_ctx *c, const int *ts, const int *scopes, int n, const double *running_resolution, int rank, int nranks, int upper,
                           void *contrib, const void *gathered, void *masks_out, void *side_stream)
{
  const bool halo = upper >= 0;
  if (!c || n <= 0 || !ts || !scopes || !running_resolution || !contrib || !gathered) return fail(c, FTKX_E_INVALID, "ftkx_series_dist_begin: null argument or no steps");
  if (!c->mesh_set) return fail(c, FTKX_E_INVALID, "sweep: call ftkx_set_mesh first");
  if (!c->pending.empty()) return fail(c, FTKX_E_INVALID, "ftkx_series_dist_begin: sweeps pending, collect first");
  if (c->sr_open >= 2) return fail(c, FTKX_E_INVALID, "ftkx_series_dist_begin: two passes are open, complete one first");      // (slab passes: two, as their callers keep)
  if (rank < 0 || rank >= nranks) return fail(c, FTKX_E_INVALID, "ftkx_series_dist_begin: rank %d of %d", rank, nranks);
  if (!(*running_resolution > 0)) return fail(c, FTKX_E_INVALID, "ftkx_series_dist_begin: the running resolution must be positive (DBL_MAX: none yet)");
  // (upper <= rank: a series that is periodic in time -- the slice behind the last slab is the first slab's first, include/ftkx_slab.h)
  if (halo && (upper >= nranks || !(scopes[n - 1] & FTKX_SCOPE_INTERVAL))) return fail(c, FTKX_E_INVALID, "ftkx_series_dist_begin: the upper neighbour must be a rank of the series, and the last step an interval sweep");
  if (c->scalar_mode < 0) return fail(c, FTKX_E_INVALID, "ftkx_series_dist_begin: push this rank's slices first");
  c->ahead.clear(); c->announced.clear();
  HIP_TRY(c, hipSetDevice(c->device));
  ftkx_series_pending &Q = c->sr_pend[c->sr_place(c->sr_open)];
  if (slab_half_queued(c)) return fail(c, FTKX_E_INVALID, "ftkx_series_dist_begin: a slab pass is half queued (finish or abort it)");
  int rc;
  {
    Mesh m0; fill_mesh(c, m0);                               // (before anything is allocated for a halo slice that will not be used)
    if (!ftkx::masks_have_summary(m0)) return fail(c, FTKX_E_UNSUPPORTED, "ftkx_series_dist_begin: this mesh has no summarised masks (use the host-driven calls)");
  }
  const int t_halo = halo ? ts[n - 1] + 1 : -1;
  if (halo && (rc = ensure_sparse_slice(c, t_halo, c->scalar_mode))) return rc;
  ftkx_series_pending *before = c->sr_open > 0 ? &c->sr_pend[c->sr_place(c->sr_open - 1)] : nullptr;
  DistPlan dp{t_halo, rank, nranks, upper, (const u64 *)gathered, (u64 *)contrib, masks_out, (hipStream_t)side_stream};
  if ((rc = series_plan(c, Q, ts, scopes, n, *running_resolution, nullptr, true, before, &dp))) { Q.open = false; Q.dist = false; return rc; }
  if (Q.by_host) {      // (options the device-driven form does not cover: a slab pass has no host-driven form of its own -- the caller's protocol does)
    Q.open = false; Q.dist = false;
    return fail(c, FTKX_E_UNSUPPORTED, "ftkx_series_dist_begin: these options / this mesh are not covered by the device-driven pass (use the host-driven calls)");
  }
  ftkx_series_buffers &B = c->sr_buf[Q.buf];
  if (!B.dist_block) { HIP_TRY(c, hipMalloc((void **)&B.dist_block, (size_t)ftkx::DB_N * sizeof(u64))); HIP_TRY(c, hipMemsetAsync(B.dist_block, 0, (size_t)ftkx::DB_N * sizeof(u64), c->stream)); }
  Q.running_from = B.dist_block + ftkx::DB_PSEUDO;
  // this rank's contribution to the all_gather: its slab's reductions folded (the first slice's masks went out inside the plan)
  const ftkx::SeriesSlice *d_slices = (const ftkx::SeriesSlice *)((char *)B.d_desc + Q.off_slices);
  const int nown = (int)Q.k - (halo ? 1 : 0);
  ftkx::launch_dist_contrib(d_slices, nown, B.red, (u64 *)contrib, B.dist_block, c->stream);
  HIP_TRY(c, hipGetLastError());
  Q.dist_stage = 1;
  return FTKX_OK;
}

int ftkx_series_dist_cull(ftkx_ctx *c, const void *masks_in, void *request_out)
{
  if (!c) return FTKX_E_INVALID;
  ftkx_series_pending *Pp = dist_pending(c, 1, "ftkx_series_dist_cull");
  if (!Pp) return FTKX_E_INVALID;
  ftkx_series_pending &P = *Pp;
  if ((P.t_halo >= 0) != (masks_in != nullptr) || (P.t_halo >= 0) != (request_out != nullptr)) return fail(c, FTKX_E_INVALID, "ftkx_series_dist_cull: masks_in / request_out go with a halo, and only with one");
  HIP_TRY(c, hipSetDevice(c->device));
  ftkx_series_buffers &B = c->sr_buf[P.buf];
  Mesh m; series_mesh(c, P, m);
  // the running minimum before this slab (and the gathered block into the results, for the host); the halo's masks into its slice
  {
    u64 *tail = B.results + (size_t)ftkx::SR_HEAD + (size_t)P.n + 2 * P.k;
    if (P.t_halo >= 0) {
      size_t ub, cap, off_idx, off_words, total;
      if (!packed_layout(c, m, &ub, &cap, &off_idx, &off_words, &total)) return fail(c, FTKX_E_UNSUPPORTED, "ftkx_series_dist_cull: this mesh has no summarised masks");
      auto hit = c->slices.find(P.t_halo);
      if (hit == c->slices.end() || !hit->second.sparse) return fail(c, FTKX_E_NOSLICE, "ftkx_series_dist_cull: the halo slice %d was dropped or replaced between the stages", P.t_halo);
      Slice &h = hit->second;
      const char *in = (const char *)masks_in;
      ftkx::launch_dist_import(P.gathered, P.dist_rank, P.dist_nranks, P.running_in, B.dist_block, tail, (const u64 *)in, (const unsigned *)(in + off_idx), (const u64 *)(in + off_words), ub, cap,
                               m.u_rows, factor_log2_of(P.hint), h.U, h.M, mask_bytes(c) / 8, c->stream);
    } else
      ftkx::launch_dist_import(P.gathered, P.dist_rank, P.dist_nranks, P.running_in, B.dist_block, tail, nullptr, nullptr, nullptr, 0, 0, m.u_rows, 0, nullptr, nullptr, 0, c->stream);
  }
  int rc;
  if ((rc = series_queue_cull(c, P))) return rc;
  if (P.t_halo >= 0) {
    // the cells whose exact test reads the halo slice: refine now (the rest of the chain will not refine again), list them, write the request
    Fields *d_steps = (Fields *)((char *)B.d_desc + P.off_steps);
    const Slice &h = c->slices.find(P.t_halo)->second;          // (checked above)
    ftkx::launch_refine(m, d_steps, c->d_refine, c->refine_capacity, c->d_list, c->list_capacity, c->stream);
    P.refined = true;
    u64 *req = (u64 *)request_out;
    const size_t cap = dist_cells_cap(c);
    ftkx::launch_dist_cells(m, d_steps, c->d_list, c->list_capacity, c->refine_capacity, h.S ? h.S : h.V, req, cap, B.dist_block, B.results, c->stream);
    P.request_out = req;
  }
  HIP_TRY(c, hipGetLastError());
  P.dist_stage = 2;
  return FTKX_OK;
}

int ftkx_series_dist_serve(ftkx_ctx *c, const void *request_in, void *reply_out)
{
  if (!c) return FTKX_E_INVALID;
  ftkx_series_pending *Pp = dist_pending(c, 2, "ftkx_series_dist_serve");
  if (!Pp) return FTKX_E_INVALID;
  ftkx_series_pending &P = *Pp;
  if ((request_in != nullptr) != (reply_out != nullptr)) return fail(c, FTKX_E_INVALID, "ftkx_series_dist_serve: a request and a reply buffer, or neither");
  HIP_TRY(c, hipSetDevice(c->device));
  if (request_in) {
    // the lower neighbour's cells around THIS rank's first slice: count read on the device, reply of fixed size
    ftkx_series_buffers &B = c->sr_buf[P.buf];
    Mesh m; series_mesh(c, P, m);
    auto it0 = c->slices.find(P.slice_ts[0]);
    if (it0 == c->slices.end() || it0->second.sparse) return fail(c, FTKX_E_NOSLICE, "ftkx_series_dist_serve: this rank's first slice %d was dropped between the stages", P.slice_ts[0]);
    const Slice &s0 = it0->second;
    const int ncomp = c->scalar_mode == 1 ? 1 : c->nd;
    ftkx::launch_dist_patches(m, false, (const u64 *)request_in, dist_cells_cap(c), ncomp, c->scalar_mode == 1 ? s0.S : s0.V, (double *)reply_out, B.results + ftkx::SR_HALO_SERVED, c->stream);
    HIP_TRY(c, hipGetLastError());
  }
  P.dist_stage = 3;
  return FTKX_OK;
}

int ftkx_series_dist_finish(ftkx_ctx *c, const void *reply_in)
{
  if (!c) return FTKX_E_INVALID;
  ftkx_series_pending *Pp = dist_pending(c, 3, "ftkx_series_dist_finish");
  if (!Pp) return FTKX_E_INVALID;
  ftkx_series_pending &P = *Pp;
  if ((P.t_halo >= 0) != (reply_in != nullptr)) return fail(c, FTKX_E_INVALID, "ftkx_series_dist_finish: a reply goes with a halo, and only with one");
  HIP_TRY(c, hipSetDevice(c->device));
  if (P.t_halo >= 0) {
    Mesh m; series_mesh(c, P, m);
    auto hit = c->slices.find(P.t_halo);
    if (hit == c->slices.end() || !hit->second.sparse) return fail(c, FTKX_E_NOSLICE, "ftkx_series_dist_finish: the halo slice %d was dropped or replaced between the stages", P.t_halo);
    Slice &h = hit->second;
    const int ncomp = c->scalar_mode == 1 ? 1 : c->nd;
    ftkx::launch_dist_patches(m, true, P.request_out, dist_cells_cap(c), ncomp, c->scalar_mode == 1 ? h.S : h.V, const_cast<double *>((const double *)reply_in), nullptr, c->stream);
  }
  int rc;
  if ((rc = series_queue_tail(c, P))) return rc;
  P.dist_stage = 4;
  P.open = true;
  c->sr_open ++;
  return FTKX_OK;
}

int ftkx_series_dist_status(const ftkx_ctx *c, long long *asked, long long *served, double *gathered, int nranks)
{
  if (!c) return FTKX_E_INVALID;
  const ftkx_series_buffers &B = c->sr_buf[c->sr_last_buf];
  if (!B.h_results || c->sr_last_nranks == 0) return FTKX_E_INVALID;
  if (asked) *asked = (long long)B.h_results[ftkx::SR_HALO_ASKED];
  if (served) *served = (long long)B.h_results[ftkx::SR_HALO_SERVED];
  if (gathered) {
    if (nranks != c->sr_last_nranks) return FTKX_E_INVALID;
    memcpy(gathered, B.h_results + c->sr_last_gathered_off, (size_t)ftkx::kDistContrib * (size_t)nranks * sizeof(double));
  }
  return FTKX_OK;
}

int ftkx_sweep_series_abort(ftkx_ctx *c)
{
  if (!c) return FTKX_E_INVALID;
  if (c->sr_open == 0 && !slab_half_queued(c)) return FTKX_OK;
  (void)hipSetDevice(c->device);
  // whatever the open passes queued runs to its end (their kernels write buffers that stay allocated); nothing of it is read
  hipError_t e = hipStreamSynchronize(c->stream);
  if (c->sr_copy_stream && e == hipSuccess) e = hipStreamSynchronize(c->sr_copy_stream);
  if (c->sr_tail_stream && e == hipSuccess) e = hipStreamSynchronize(c->sr_tail_stream);
  if (c->sr_tail_stream2 && e == hipSuccess) e = hipStreamSynchronize(c->sr_tail_stream2);
  for (ftkx_series_pending &P : c->sr_pend) {
    if (P.dist && P.dist_stage > 0 && P.dist_stage < 4) { P.open = true; }      // (a slab pass that was never finished: its masks are nobody's either)
    if (!P.open) continue;
    // the masks this pass was building are nobody's: built, but never marked
    for (size_t j = 0; j < P.k && j < P.red_index.size(); j ++) {
      auto it = c->slices.find(P.slice_ts[j]);
      if (P.red_index[j] >= 0 && it != c->slices.end() && it->second.mask_gen == P.gen[j]) { it->second.mask_factor = 0; it->second.have_fused = false; }
    }
    release_retired(c, P);
    P.open = false; P.copy_pending = false; P.dist_stage = 0;
  }
  for (ftkx_series_buffers &B : c->sr_buf) B.copy_out = false;
  c->sr_open = 0; c->sr_head = 0;
  c->sr_short_chain = false; c->sr_sparse = false; c->sr_lists_owner = 0; c->sr_last_running = 0;
  c->ahead.clear(); c->announced.clear();
  if (e != hipSuccess) return fail(c, FTKX_E_DEVICE, "ftkx_sweep_series_abort: %s", hipGetErrorString(e));
  return FTKX_OK;
}

int ftkx_sweep_series(ftkx_ctx *c, const int *ts, const int *scopes, int n, double *running_resolution, unsigned long long *factors,
                      const ftkx_cp_t **out, size_t *n_out)
{
  if (!c || (n > 0 && (!ts || !scopes)) || !running_resolution) return fail(c, FTKX_E_INVALID, "null argument");
  if (out) *out = nullptr;
  if (n_out) *n_out = 0;
  if (!c->mesh_set) return fail(c, FTKX_E_INVALID, "sweep: call ftkx_set_mesh first");
  if (!c->pending.empty()) return fail(c, FTKX_E_INVALID, "ftkx_sweep_series: sweeps pending, collect first");
  if (c->sr_open) return fail(c, FTKX_E_INVALID, "ftkx_sweep_series: passes open (ftkx_sweep_series_submit), complete them first");
  if (slab_half_queued(c)) return fail(c, FTKX_E_INVALID, "ftkx_sweep_series: a slab pass is half queued (ftkx_series_dist_finish or ftkx_sweep_series_abort first)");
  if (!(*running_resolution > 0)) return fail(c, FTKX_E_INVALID, "ftkx_sweep_series: the running resolution must be positive (DBL_MAX: none yet)");
  if (n == 0) return FTKX_OK;
  c->ahead.clear(); c->announced.clear();
  HIP_TRY(c, hipSetDevice(c->device));
  int rc;

  // (the pass in chunks, the tail of one chunk next to the mask kernel of the next -- round 3, FTKX_SERIES_CHUNKS -- was measured slower on every
  // configuration and has been removed: NOTES.md)
  ftkx_series_pending &P = c->sr_pend[0];
  if ((rc = series_submit(c, P, ts, scopes, n, *running_resolution, nullptr, false))) { P.open = false; return rc; }
  return series_complete(c, P, running_resolution, factors, out, n_out);
}

// The split pass (path 5) of this context: how it is decided and on what.  *state: 0 auto, still measuring (or no pass of a qualifying shape
// yet); 1 auto, decided for the split pass; 2 auto, decided against it (in order; measured again after 4 096 passes); 3 forced on; 4 forced off
// (FTKX_SERIES_HOOKS split=...).  The medians (ms per pass, in order and split, the host's time between completions with the pipeline full)
// are those of the last decision, 0 where none was taken.
int ftkx_series_split_decision(const ftkx_ctx *c, int *state, double *median_in_order_ms, double *median_split_ms)
{
  if (!c) return fail(nullptr, FTKX_E_INVALID, "null context");
  const ftkxh::split_cal &K = c->sr_cal;
  if (state) *state = ftkxh::split_state(K, c->sr_split_forced);
  if (median_in_order_ms) *median_in_order_ms = K.median_order * 1e3;
  if (median_split_ms) *median_split_ms = K.median_split * 1e3;
  return FTKX_OK;
}

int ftkx_series_last_path(const ftkx_ctx *c, unsigned long long *status)
{
  if (!c) return -1;
  if (status) *status = c->sr_last_status;
  return c->sr_last_path;
}

}  // extern "C"
