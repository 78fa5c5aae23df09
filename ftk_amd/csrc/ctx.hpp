// Shared between the translation units of libftkx.so that implement the C ABI of include/ftkx.h (ftkx_api.hip: context, slices,
// options, one-shot calls; prepare.hip: the one-pass mask + reduction pre-pass; collect.hip: the batched sweep; halo.hip: the compact
// t-slab halo; series.hip: the device-driven pass over a resident series).  Not part of the ABI.
#pragma once
#include <hip/hip_runtime.h>

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
#include "split_policy.hpp"
#include "internal.hpp"

namespace ftkx {
void mask_kernel_launches(unsigned long long *out, const char **names);
void launch_tile(const TileParams &p, hipStream_t stream);
void launch_tile_stats_fold(u64 *slots, u64 *counters, hipStream_t stream);
void tile_dims(int nd, int tile[3]);
void launch_masks(const Mesh &m, const MaskJob *d_jobs, int njobs, hipStream_t stream);
void launch_cull(const Mesh &m, const Fields *d_steps, int nsteps, u64 *d_list, u64 cap, hipStream_t stream, const FactorJob *job = nullptr);
void launch_exact(const Mesh &m, const Fields *d_steps, int step_base, const u64 *d_list, u64 cap, hipStream_t stream, int few_wgs = 0);
void launch_records(const Mesh &m, const Fields *d_fields, hipStream_t stream);
void launch_compact_words(const Mesh &m, const unsigned char *U, const unsigned char *M, unsigned *idx, u64 *words, u64 capacity, u64 *counter, hipStream_t st);
void launch_scatter_words(const unsigned *idx, const u64 *words, size_t n, unsigned char *M, size_t mask_words, u64 *bad, hipStream_t st);
void launch_pack_masks(u64 *hdr, const u64 *counter, const unsigned char *U, u64 u_bytes, u64 capacity, int u_rows, int factor_log2, hipStream_t st);
void launch_scatter_packed(const u64 *hdr, const unsigned *idx, const u64 *words, u64 u_bytes, u64 capacity, int u_rows, int max_factor_log2, unsigned char *U, unsigned char *M,
                           size_t mask_words, u64 *bad, hipStream_t st);
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
void launch_cull_coarse(const Mesh &m, const Fields *d_steps, int nsteps, u64 *d_refine, u64 refine_cap, hipStream_t stream, const FactorJob *job = nullptr);
void launch_refine(const Mesh &m, const Fields *d_steps, const u64 *d_refine, u64 refine_cap, u64 *d_list, u64 cap, hipStream_t stream, int few_wgs = 0);
void launch_series_begin(u64 *counters, u64 *red, size_t nslots, unsigned *hist, size_t nbins, u64 *results, size_t nresults, hipStream_t st,
                         const void *desc_src = nullptr, void *desc_dst = nullptr, size_t desc_bytes = 0, unsigned *fetched = nullptr, unsigned fetched_val = 0);
void launch_series_one(const Mesh &m, const OneArgs &a, int nwg, hipStream_t st);
void launch_series_tail_begin(u64 *counters, unsigned *hist, size_t nbins, u64 *results, size_t nresults, hipStream_t st);
void launch_series_factors(Fields *steps, int nsteps, const SeriesSlice *slices, int nslices, const SeriesStep *sinfo, const u64 *red, double running_in, const u64 *running_from,
                           double safe_m, u64 *results, u64 *counters, hipStream_t st);
void launch_bucket_rank(const Mesh &m, const u64 *bucketed, const unsigned *boff, u64 *sorted, u64 *results, hipStream_t st, int few_wgs = 0);
void launch_bucket_scan(unsigned *hist, unsigned *boff, unsigned nbins, u64 *counters, hipStream_t st, bool small_wg = false);
void launch_bucket_scatter(const Mesh &m, unsigned *boff, u64 *bucketed, hipStream_t st, int few_wgs = 0);
void launch_series_records(const Mesh &m, const Fields *d_fields, const u64 *sorted, ftkx_cp_t *out, hipStream_t st, bool lean = false);
Mesh coarse_view(const Mesh &m);
void launch_series_small(const Mesh &m, const Mesh &mc, const Fields *d_steps, bool two_level, const u64 *d_refine, const u64 *d_list, ftkx_cp_t *out,
                         u64 *results, size_t nwords, u64 *h_results, unsigned *flag, unsigned seq, unsigned *done, bool report_decline, hipStream_t st);
void launch_series_copy_out(const ftkx_cp_t *src, ftkx_cp_t *dst, u64 capacity, const u64 *results, unsigned *done, unsigned *flag, unsigned seq, hipStream_t st,
                            const unsigned *wait_flag = nullptr, unsigned wait_val = 0);
void launch_series_finish(const Mesh &m, u64 *results, size_t nwords, u64 list_capacity, u64 refine_capacity, u64 *h_results, unsigned *flag, unsigned seq, hipStream_t st);
// dist_kernels.hip: the slab pass
void launch_dist_contrib(const SeriesSlice *slices, int nown, const u64 *red, u64 *contrib, u64 *block, hipStream_t st);
void launch_dist_export(const Mesh &m, const unsigned char *U, const unsigned char *M, u64 u_bytes, u64 *hdr, unsigned *idx, u64 *words, u64 capacity, int factor_log2, u64 *block, hipStream_t st);
void launch_dist_import(const u64 *gathered, int rank, int nranks, double running_in, u64 *block, u64 *results_tail, const u64 *hdr, const unsigned *idx, const u64 *words, u64 u_bytes,
                        u64 capacity, int u_rows, int max_factor_log2, unsigned char *U, unsigned char *M, u64 mask_words, hipStream_t st);
void launch_dist_cells(const Mesh &m, const Fields *d_steps, const u64 *d_list, u64 list_capacity, u64 refine_capacity, const double *halo_field, u64 *request, u64 cap, u64 *block, u64 *results, hipStream_t st);
void launch_dist_patches(const Mesh &m, bool scatter, const u64 *request, u64 cap, int ncomp, double *field, double *patches, u64 *served, hipStream_t st);
}  // namespace ftkx

using ftkx::Fields;
using ftkx::MaskJob;
using ftkx::Mesh;
using ftkx::TileParams;
using ftkx::u64;

namespace ftkxh {

// 2D scalar slices: gradient2D's (D - 1) scaling folded into exact thresholds of the unscaled differences (MaskJob::tx, ty)
inline MaskJob with_lean_thresholds(MaskJob j, const Mesh &m)
{
  if (m.nd == 2 && m.scalar_mode && j.threshold > 0.0) {
    j.tx = ftkx::exact_threshold(j.threshold, (double)(m.ext_sz[0] - 1));
    j.ty = ftkx::exact_threshold(j.threshold, (double)(m.ext_sz[1] - 1));
  }
  return j;
}

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
  unsigned long long mask_gen = 0;  // changes whenever the masks are (re)built, dropped or the slice replaced: a series pass collected later marks only what is still its own
  bool max_known() const { return have_res || have_fused || sparse; }
};

// how a request is swept: MODE_TILE tests every simplex (exact_only, non-robust 3D, odd factors); MODE_FAST = masks -> cull ->
// survivor list -> exact kernel; MODE_TILE_CULL = the tile kernel with its in-tile cull (same per-vertex legality rule), for
// data on which most cells survive the cull anyway (the int64-overflow regime: a survivor list would be as large as the input)
enum { MODE_TILE = 0, MODE_FAST = 1, MODE_TILE_CULL = 2 };
struct Request { int t, scope; unsigned long long factor; int mode; };

enum { K_MASK = 0, K_CULL = 1, K_EXACT = 2, K_TILE = 3, K_N = 4 };


}  // namespace ftkxh

// A series pass that has been queued (ftkx_sweep_series_submit) and not yet collected (ftkx_sweep_series_complete): what the second half
// of the call needs.  Two may be open at a time: the host prepares and queues pass N + 1 while the device still works on pass N, and the
// records of pass N cross PCIe (a copy engine, not a kernel) while the mask kernel of pass N + 1 runs.
struct ftkx_series_pending {
  bool open = false;
  bool by_host = false;             // not queued: the host-driven batch sweeps it when it is collected
  std::vector<int> ts, scopes, slice_ts, red_index;
  std::vector<unsigned long long> gen;   // per slice: Slice::mask_gen as this pass left it
  int n = 0, buf = 0;
  size_t k = 0, nwords = 0, nbins = 0, total_desc = 0;
  unsigned long long hint = 0;
  bool two_level = false, short_chain = false, small_now = false, to_device = false;
  bool copy_pending = false;        // to_device: the copy kernel has not been queued yet (it goes behind the descriptor fetch of the next pass, or is queued when the pass is collected)
  int u_rows = 1;
  u64 cells = 0;
  unsigned seq = 0;
  unsigned long long uid = 0;       // ftkx_ctx::sr_pass_uid when the pass was queued (owner stamp of the counters and lists)
  double running_in = 0;
  bool chained = false;             // the running minimum came from the pass before it on the device (running_in: what the host knew)
  size_t off_steps = 0, off_slices = 0, off_sinfo = 0, ntodo = 0;
  int shift = 0;                    // order key -> bucket
  const u64 *running_from = nullptr;   // a results block on the device whose SR_RUNNING word this pass continues from (the pass before it, or a slab pass's stub)
  bool pipelined = false;
  bool refined = false;             // the refine kernel has been queued already (a slab pass lists the halo's cells from its output)
  // split pass (series.hip, "the tail next to the next mask kernel"): begin + masks on the context's stream, the tail -- counters, cull + factors,
  // the kernel chain -- on the tail stream behind an event, next to the mask kernel of the pass queued behind it
  bool split = false;
  int before_buf = -1;              // the place of the split pass that was open when this one was planned (its factor job is waited for by this pass's tail), or -1
  int cal_kind = 0;                 // a pass of the split calibration: 1 measured in order, 2 measured split
  int tail_set = 0;                 // 0: the context's own counters and lists, 1: sr_set1
  bool split_sparse = false;        // ... of a sparse pass: few workgroups per chain kernel, 2^10 buckets
  bool one = false;                 // the one-launch pass for small series (one_kernel.hip)
  std::vector<std::pair<unsigned char *, unsigned char *>> retired;   // (M, U) arrays this pass still reads, replaced in their slices by the pass queued behind it
  std::vector<ftkxh::Slice> parked;        // slices dropped (or replaced) while this split pass was the newest one open: its tail -- on a stream of its own -- may still read
                                    // their arrays, which go back to the pools when it has been completed (free_slice)
  // slab pass (ftkx_series_dist_*): one rank's part of a series cut into timestep slabs, queued in stages with the caller's collectives between them
  bool dist = false;
  int dist_stage = 0;               // 1 begun (masks, contribution, outgoing masks), 2 culled (request written), 3 served (reply written), 4 finished = open
  int t_halo = -1;                  // the slice this rank's last interval sweep reads and does not own: masks + patches arrive inside the pass; -1: none
  int dist_rank = 0, dist_nranks = 1, dist_upper = -1;     // dist_upper: the rank that owns t_halo (not rank + 1 where slabs are empty)
  const u64 *gathered = nullptr;    // device, kDistContrib words per rank: where the caller's all_gather puts the contributions
  u64 *request_out = nullptr;       // device: this rank's request to its upper neighbour (count, cells)
};

// what one of the (two) passes in flight writes that the host reads, or that a copy engine reads after the pass
struct ftkx_series_buffers {
  u64 *results = nullptr, *h_results = nullptr;       // device block; coherent pinned copy with the flag word behind it
  size_t results_cap = 0, h_results_cap = 0;
  unsigned seq = 0;
  ftkx_cp_t *out = nullptr; size_t out_cap = 0;       // pinned: the records as the caller reads them
  ftkx_cp_t *d_out = nullptr; size_t d_out_cap = 0;   // device: the records of a pass whose way over PCIe is left to the copy kernel on its own stream
  unsigned *copy_done = nullptr;                       // that kernel's workgroup counter
  hipEvent_t ev_copied = nullptr, ev_export = nullptr;
  hipEvent_t ev_masks = nullptr, ev_factors = nullptr, ev_tail = nullptr;   // split pass: masks done (stream), cull + factor job / tail done (tail stream)
  u64 *red = nullptr; size_t red_cap = 0;             // the reduction slots of this pass's mask jobs (128 words per slice): its own, the next pass's begin kernel must not wipe them
  bool copy_out = false;                               // a copy has been queued since the buffers were last used: the next record kernel waits for it
  void *h_desc = nullptr, *d_desc = nullptr; size_t desc_cap = 0;
  u64 *dist_block = nullptr;                           // slab pass: DB_N words (sweep_params.hpp)
};

struct ftkx_ctx {
  int nd = 0, device = 0;
  hipStream_t own_stream = nullptr, stream = nullptr;
  ftkx_options opt;
  long long dom_st[3] = {0, 0, 0}, dom_sz[3] = {1, 1, 1}, core_st[3] = {0, 0, 0}, core_sz[3] = {1, 1, 1}, ext_st[3] = {0, 0, 0}, ext_sz[3] = {1, 1, 1};
  bool mesh_set = false;
  int scalar_mode = -1;             // -1 undecided, 0 vector slices, 1 scalar slices (V = gradient(S) evaluated in flight)
  std::map<int, ftkxh::Slice> slices;
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
  u64 *d_tile_stats = nullptr;      // 512 words: the tile kernels' statistics in 256 slots (TileParams::stats)
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
  std::vector<ftkxh::Request> pending;
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
  unsigned long long uploads_staged = 0, uploads_direct = 0;   // ftkx_debug_upload_counts (upload.cpp)
  int tile_repeat = 1;              // ftkx_debug_tile_repeat: the tile kernel's fan phase that many times per tile (the int-VALU yardstick of bench.py)
  // compact halo: the compacted mask words of the last ftkx_export_masks_size, the surviving cells of the last ftkx_sweep_cull
  unsigned *d_word_idx = nullptr; u64 *d_words = nullptr; size_t words_cap = 0, n_words = 0; int words_t = -1;
  u64 *d_cells = nullptr; size_t cells_cap = 0, n_cells = 0;
  u64 *d_patch_cells = nullptr; double *d_patches = nullptr; size_t patch_cap = 0;   // staging for host-side callers
  void *d_packed = nullptr; size_t packed_cap = 0;                                   // staging of a packed mask message for host-side callers
  // series pass (series.hip): per pass in flight the results block (device + coherent pinned copy: it also holds the fragile list, the flag
  // lives behind it), the record buffers and the descriptors; shared, in stream order: the ordering buffers
  static constexpr int kPlaces = 3;   // passes in flight at most: two keep the stream fed; the third lets the host run one pass ahead of a split pass's
                                      // tail, which ends behind the mask kernel of the pass after it
  ftkx_series_buffers sr_buf[kPlaces];
  ftkx_series_pending sr_pend[kPlaces];
  int sr_place(int k) const { return (sr_head + k) % kPlaces; }      // the k-th oldest open pass's place
  int sr_open = 0, sr_head = 0;       // passes open, and which of sr_pend is the oldest
  bool sr_internal = false;           // the host-driven batch is sweeping for a series pass: its calls are let through while passes are open
  // which pass the context's counters and survivor lists (d_counters, d_list, d_refine, d_pass) belong to right now: stamped when a pass's
  // cull is queued, cleared whenever the host-driven batch takes them (series.hip: a pass whose fused tail declined may queue the rest of
  // its chain only while they are still its own)
  unsigned long long sr_pass_uid = 0, sr_lists_owner = 0;
  hipStream_t sr_copy_stream = nullptr, sr_fetch_stream = nullptr, sr_tail_stream = nullptr, sr_tail_stream2 = nullptr;
  // what the tail of a split pass works on besides the context's own counters, lists, pass descriptors, fragile list and ordering arrays: a second
  // set (same capacities) for every other split pass, on the second tail stream, so that the tails of two passes can run at the same time -- a
  // mask launch shorter than one tail next to it (a single 512^3 slice) then still hides half a tail behind every mask kernel
  struct tail_set {
    u64 *counters = nullptr, *list = nullptr, *refine = nullptr, *pass = nullptr, *fragile = nullptr, *bucketed = nullptr, *sorted = nullptr;
    unsigned *hist = nullptr, *boff = nullptr;
    u64 capacity = 0, list_capacity = 0, refine_capacity = 0, fragile_capacity = 0;
    size_t bins_cap = 0;
  } sr_set1;
  ftkxh::split_cal sr_cal;              // the split pass's self-check (split_policy.hpp)
  int sr_split_forced = 0;              // the last plan's FTKX_SERIES_HOOKS split setting: 0 auto, 1 forced on (2, 3, 4), 2 forced off (0)
  double sr_last_complete_s = 0;        // host clock of the last completion (0: the pipeline ran empty since)
  int sr_last_complete_kind = 0;        // 1 in order / 2 split, of a calibration pass; 0 otherwise
  unsigned sr_split_seq = 0;          // split passes queued so far: their parity picks the set
  int sr_one_off = 0;                  // passes for which the one-launch form is not tried (it declined a moment ago)
  u64 *sr_one_scratch = nullptr;       // the one-launch pass's barrier counters, partial reductions and per-workgroup counts (ONE_WORDS)
  unsigned *sr_fetch_flag = nullptr;   // device: [0] the number of the last pass whose descriptors have been fetched (series_begin_kernel), [1] its arrival counter
  unsigned sr_fetch_seq = 0;
  hipEvent_t sr_ev_fetched = nullptr;  // slab passes: recorded behind the begin kernel, waited for by the copy of the pass before (no spin-wait there)
  unsigned long long mask_epoch = 0;  // source of Slice::mask_gen values
  double sr_last_running = 0;         // the running minimum the host knew when it last collected a pass (hint of a chained pass)
  unsigned *sr_hist = nullptr, *sr_boff = nullptr;
  size_t sr_bins_cap = 0;
  u64 *sr_bucketed = nullptr;
  size_t sr_bucketed_cap = 0;
  u64 *sr_sorted = nullptr;
  size_t sr_sorted_cap = 0;
  int sr_skip_small = 0;             // passes for which the fused tail kernel is not launched (the data was hit-dense a moment ago)
  int sr_late_streak = 0;            // consecutive passes the fused tail declined late
  bool sr_short_chain = false;       // the last pass was finished by the fused tail kernel: the next one is queued without the kernels behind it
  bool sr_sparse = false;            // the last pass had few survivors and records (the fused tail's range): a pass whose mask kernel is long enough is split
  int sr_last_buf = 0;               // the buffers of the pass completed last (ftkx_series_dist_status reads its results block)
  size_t sr_last_gathered_off = 0; int sr_last_nranks = 0;   // where its gathered contributions sit in that block (0 ranks: not a slab pass)
  int sr_last_path = 0;              // which way the last ftkx_sweep_series went: 1 device-driven, 2 early single-workgroup tail, 0 the host-driven batch
  unsigned long long sr_last_status = 0;
  // pass 2 on the device (trace_device.hip): tags up, neighbours / degrees / roots down
  void *tr_dev = nullptr, *tr_host = nullptr, *tr_parent = nullptr, *tr_tables = nullptr;
  size_t tr_cap = 0;
  int tr_tables_nd = 0;
  ftkx_stats stats;
  // optional kernel timing (hipEvents on the context's stream)
  int profiling = 0;               // 0 off, 1 every kernel family, 2 the mask kernel only
  bool ev_open = false;
  std::vector<std::pair<int, std::pair<hipEvent_t, hipEvent_t>>> events;
  std::vector<hipEvent_t> event_pool;
  double k_ms[ftkxh::K_N] = {0, 0, 0, 0};
  unsigned long long k_launches[ftkxh::K_N] = {0, 0, 0, 0};
  std::string err;
};

namespace ftkx {
// waits for a sequence number a kernel stores, with system scope, into coherent pinned memory (ftkx_api.hip); nullptr or what went wrong
const char *wait_flag(const unsigned *flag, unsigned seq, hipStream_t stream);
}

namespace ftkxh {

int fail(ftkx_ctx *c, int code, const char *fmt, ...);

#define HIP_TRY(c, call)                                                                                   \
  do {                                                                                                     \
    hipError_t e_ = (call);                                                                                \
    if (e_ != hipSuccess) return fail((c), e_ == hipErrorOutOfMemory ? FTKX_E_NOMEM : FTKX_E_DEVICE,       \
                                      "%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), __FILE__, __LINE__); \
  } while (0)

// ftkx_api.hip
size_t n_vertices(const ftkx_ctx *c);
int mask_pitch(const ftkx_ctx *c);
int u_pitch(const ftkx_ctx *c);
size_t u_bytes(const ftkx_ctx *c);
size_t u_bytes_used(const ftkx_ctx *c, const Mesh &m);
size_t mask_bytes(const ftkx_ctx *c);
void free_slice(Slice &s, ftkx_ctx *pool_owner = nullptr);
void release_pools(ftkx_ctx *c);
int ensure_hit_buffer(ftkx_ctx *c, u64 want);
int ensure_fragile(ftkx_ctx *c, u64 want);
int ensure_list(ftkx_ctx *c, u64 want);
int ensure_refine(ftkx_ctx *c, u64 want);
int ensure_desc(ftkx_ctx *c, size_t bytes);
int ensure_host_buffer(ftkx_ctx *c, size_t want);
void fill_mesh(const ftkx_ctx *c, Mesh &m);
int slice_resolution(ftkx_ctx *c, Slice &s);
bool overflow_free(int nd, double maxabs, u64 factor);
double big_threshold(int nd, u64 factor);
bool pow2_factor(u64 factor);
bool masks_valid(const ftkx_ctx *c, const Slice &s, u64 factor, bool two_level, int u_rows);
double job_big(const ftkx_ctx *c, const Slice &s, u64 factor, bool *rule_on);
int ensure_mask_arrays(ftkx_ctx *c, Slice &s, bool two_level);
int upload_from_host(ftkx_ctx *c, void *dst, const void *src, size_t bytes);       // upload.cpp
int aux_stream_get(ftkx_ctx *c, bool high_priority, hipStream_t *out);       // the library's own streams, kept for the process (ftkx_api.hip)
void aux_stream_put(ftkx_ctx *c, bool high_priority, hipStream_t st);
// halo.hip
bool packed_layout(const ftkx_ctx *c, const Mesh &m, size_t *ub, size_t *cap, size_t *off_idx, size_t *off_words, size_t *total);
int ensure_sparse_slice(ftkx_ctx *c, int t, int scalar_input);
// prepare.hip
void launch_init_red(u64 *red, size_t nslots, u64 *counters, hipStream_t st);        // {min = DBL_MAX, max = 0} slots; counters (nullable) zeroed
void launch_fetch_desc(const void *pinned_src, void *device_dst, size_t bytes, hipStream_t st);   // pinned -> device by a kernel (bytes % 8 == 0)
// collect.hip
hipEvent_t ev_take(ftkx_ctx *c);
void ev_give(ftkx_ctx *c, hipEvent_t e);
void ev_begin(ftkx_ctx *c, int kind);
void ev_end(ftkx_ctx *c);
void ev_harvest(ftkx_ctx *c, bool all = true);   // all: after a stream synchronise; otherwise only the pairs that have completed
int run_batch(ftkx_ctx *c, const double *sparse_field = nullptr, bool cull_done = false);
bool ahead_serves_pending(const ftkx_ctx *c, const Mesh &m, bool two_level);
int sort_hits_on_device(ftkx_ctx *c, size_t n, int key_bits);
int finish_records(ftkx_ctx *c, size_t n, int key_bits);

}  // namespace ftkxh
