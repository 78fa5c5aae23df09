// Kernels of the device-driven pass over a resident series (host side: series.hip).
//
// What the reference does at the top of every update_timestep on the HOST -- update_vector_field_scaling_factor,
// include/ftk/filters/critical_point_tracker.hh:850-864: the sticky running minimum of ndarray::resolution() and nbits from it -- is
// done here on the device, between the mask kernel (whose fused reduction it folds) and the exact test (which is the first kernel to
// quantise), so that a whole pass is queued without the host in the loop.  The records are put in tag order without a sort: the
// simplices that passed are counted per bucket of their order key by the exact kernel, one scan turns the counts into offsets, one
// scatter puts the descriptors into their buckets, and the record kernel picks, per output position, the descriptor of that rank
// inside its bucket.  It writes the finished records straight into the caller-visible pinned host buffer, a wavefront's 64 records as
// one contiguous run, so that the download runs while records are still being built.
#include "internal.hpp"
#include "sweep_device.hpp"
#include "series_device.hpp"

namespace ftkx {

// ---- begin: every counter, reduction slot and histogram bin of the pass in one launch ------------------------------------------------
__global__ __launch_bounds__(256) void series_begin_kernel(u64 *counters, u64 *red, size_t nslots, unsigned *hist, size_t nbins, u64 *results, size_t nresults,
                                                           const u64 *__restrict__ desc_src /* pinned; nullable */, u64 *__restrict__ desc_dst, size_t desc_words,
                                                           unsigned *fetched /* nullable */, unsigned fetched_val)
{
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  // (the pass's descriptors, pinned host -> device: the first workgroups, so that the read over PCIe is under way at once)
  for (size_t w = i; w < desc_words; w += (size_t)gridDim.x * 256) desc_dst[w] = desc_src[w];
  if (counters && i < (size_t)CNT_N) counters[i] = 0ull;      // (nullptr: a split pass zeroes them on its tail stream -- series.hip)
  if (i < nslots) { red[2 * i] = 0x7fefffffffffffffull; red[2 * i + 1] = 0ull; }   // {min = DBL_MAX, max = 0} as bit patterns
  if (i < nbins) hist[i] = 0u;
  if (i < nresults) results[i] = 0ull;
  // The descriptors have arrived (every lane's load had returned before its store was issued): said to series_copy_out_kernel of the
  // pass before, which starts its writes over PCIe only now -- a read over PCIe queued behind them took 65 us instead of 4.  A word in
  // device memory, not an event: an event recorded here held the mask kernel back by 5 us.
  if (fetched) {
    __syncthreads();
    if (threadIdx.x == 0 && atomicAdd(&fetched[1], 1u) + 1u == gridDim.x) {
      fetched[1] = 0u;
      __hip_atomic_store(&fetched[0], fetched_val, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
}

// a split pass (series.hip): what the pass's TAIL owns -- the counters and the histogram -- zeroed on the tail stream, behind the tail of the pass before it
// -- and the pass's results block: the pass before this one may still read the block's last contents (it is the block of the pass before
// THAT, whose running minimum and reductions it continues from) until its own tail is through, which on the tail stream it is
__global__ __launch_bounds__(256) void series_tail_begin_kernel(u64 *counters, unsigned *hist, size_t nbins, u64 *results, size_t nresults)
{
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i < (size_t)CNT_N) counters[i] = 0ull;
  if (i < nbins) hist[i] = 0u;
  if (i < nresults) results[i] = 0ull;
}

// ---- the sticky factor: series_device.hpp ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void series_factors_kernel(Fields *__restrict__ steps, int nsteps, const SeriesSlice *__restrict__ slices, int nslices,
                                                             const SeriesStep *__restrict__ sinfo, const u64 *__restrict__ red, double running_in,
                                                             const u64 *__restrict__ running_from, double safe_m, u64 *__restrict__ results, u64 *__restrict__ counters)
{
  series_factors_body<1024, kSeriesMaxSlices>(steps, nsteps, slices, nslices, sinfo, red, running_in, running_from, safe_m, results, counters);
}

// inclusive prefix sum over the 64 lanes of a wavefront with DPP row shifts and row broadcasts (no LDS crossbar: __shfl_up is a
// ds_bpermute, ~100 cycles a step, and the scan kernel does 64 of these scans back to back)
__device__ inline unsigned wave_inclusive_sum(unsigned v)
{
  v += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, false);   // row_shr:1
  v += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xf, 0xf, false);   // row_shr:2
  v += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xf, 0xf, false);   // row_shr:4
  v += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xf, 0xf, false);   // row_shr:8: every row of 16 lanes scanned
  v += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xa, 0xf, false);   // row_bcast:15 into rows 1 and 3
  v += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xc, 0xf, false);   // row_bcast:31 into rows 2 and 3
  return v;
}

// ---- ordering without a sort ---------------------------------------------------------------------------------------------------------
// counts per bucket -> offsets (exclusive scan).  One workgroup of 16 wavefronts; a wavefront owns a contiguous sixteenth of the bins
// and walks it in rows of 64 consecutive bins (every load and store is one contiguous 256-byte run): a first pass for the wavefront's
// total, a second one -- the counts come from the L2 now -- for the offsets.  The counts are zeroed for their second life as scatter
// cursors; the fullest bucket is published (CNT_BUCKET_MAX).  nbins <= kSeriesMaxBins.
template <int NW>
__global__ __launch_bounds__(NW * 64) void bucket_scan_kernel(unsigned *__restrict__ hist, unsigned *__restrict__ boff, unsigned nbins, u64 *__restrict__ counters)
{
  __shared__ unsigned s_wave[NW];
  __shared__ unsigned s_max;
  if (counters[CNT_SERIES_DONE]) return;
  const unsigned tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const unsigned seg = (nbins + (unsigned)NW - 1u) / (unsigned)NW, rows = (seg + 63u) / 64u;      // rows of 64 bins per wavefront
  const unsigned base = wv * rows * 64u;
  if (tid == 0) s_max = 0;
  __syncthreads();
  unsigned sum = 0, mxc = 0;
#pragma unroll 32
  for (unsigned r = 0; r < rows; r ++) {                 // (an L2 hit is ~1 us away on this part: as many loads in flight as registers allow)
    const unsigned i = base + 64u * r + lane;
    const unsigned v = i < nbins ? hist[i] : 0u;
    sum += v; mxc = v > mxc ? v : mxc;
  }
  for (int o = 32; o > 0; o >>= 1) { sum += __shfl_down(sum, o); const unsigned w = __shfl_down(mxc, o); mxc = w > mxc ? w : mxc; }
  if (lane == 0) { s_wave[wv] = sum; if (mxc) atomicMax(&s_max, mxc); }
  __syncthreads();
  unsigned carry = 0;
  for (unsigned q = 0; q < wv; q ++) carry += s_wave[q];
  for (unsigned r0 = 0; r0 < rows; r0 += 16) {           // sixteen rows' loads in flight, then their scans
    unsigned v[16];
#pragma unroll
    for (int k = 0; k < 16; k ++) { const unsigned i = base + 64u * (r0 + (unsigned)k) + lane; v[k] = (r0 + (unsigned)k < rows && i < nbins) ? hist[i] : 0u; }
#pragma unroll
    for (int k = 0; k < 16; k ++) {
      const unsigned i = base + 64u * (r0 + (unsigned)k) + lane;
      const unsigned incl = wave_inclusive_sum(v[k]);
      if (r0 + (unsigned)k < rows && i < nbins) { boff[i] = carry + incl - v[k]; hist[i] = 0u; }
      carry += (unsigned)__builtin_amdgcn_readlane((int)incl, 63);
    }
  }
  if (tid == (unsigned)(NW * 64 - 1)) boff[nbins] = carry;   // (the last wavefront's carry is the total)
  if (tid == 0) counters[CNT_BUCKET_MAX] = s_max;
}

// every simplex that passed, into its bucket (unordered inside it): the order key itself is what the record kernel needs
__global__ __launch_bounds__(256) void bucket_scatter_kernel(const u64 *__restrict__ pass, u64 capacity, const unsigned *__restrict__ boff, unsigned *__restrict__ cursor,
                                                             int shift, u64 core_cells, u64 *__restrict__ bucketed, const u64 *__restrict__ counters)
{
  if (counters[CNT_SERIES_DONE]) return;
  u64 count = counters[CNT_PASS];
  if (count > capacity) count = capacity;
  for (u64 i = (u64)blockIdx.x * 256 + threadIdx.x; i < count; i += (u64)gridDim.x * 256) {
    const u64 key = order_key(pass[i], core_cells);
    const unsigned b = (unsigned)(key >> shift);
    bucketed[boff[b] + atomicAdd(&cursor[b], 1u)] = key;
  }
}

// the descriptors of a bucket in order: every descriptor counts the smaller ones of its bucket and takes that place.  A workgroup takes
// 256 consecutive positions; the buckets they lie in form one contiguous span of the array, which is staged in LDS when it fits (hits
// cluster: a bucket of a hit-dense series holds hundreds of descriptors, and every one of them reads all of them).  Buckets fuller
// than rank_max stay as they are (SERIES_FIX_ORDER: the host orders those runs of the output).
__global__ __launch_bounds__(256) void bucket_rank_kernel(const u64 *__restrict__ bucketed, u64 capacity, const unsigned *__restrict__ boff, int shift, unsigned rank_max,
                                                          u64 *__restrict__ sorted, const u64 *__restrict__ counters, u64 *__restrict__ results)
{
  constexpr unsigned SPAN = 4096;
  __shared__ u64 s_keys[SPAN];
  __shared__ unsigned s_lo, s_hi;
  if (counters[CNT_SERIES_DONE]) return;
  u64 count = counters[CNT_PASS];
  if (count > capacity) count = capacity;
  if (blockIdx.x == 0 && threadIdx.x == 0 && counters[CNT_BUCKET_MAX] > rank_max) atomicOr((unsigned long long *)&results[SR_STATUS], (unsigned long long)SERIES_FIX_ORDER);
  for (u64 p0 = (u64)blockIdx.x * 256; p0 < count; p0 += (u64)gridDim.x * 256) {      // (block-uniform trip count)
    const u64 p = p0 + threadIdx.x;
    const bool mine = p < count;
    const u64 key = mine ? bucketed[p] : 0ull;
    unsigned lo = 0, hi = 0;
    if (mine) { const unsigned b = (unsigned)(key >> shift); lo = boff[b]; hi = boff[b + 1]; }
    __syncthreads();                                       // (the previous round's readers of s_keys are done)
    if (threadIdx.x == 0) s_lo = lo;                       // the first position's bucket starts the span ...
    const u64 plast = (p0 + 255 < count ? p0 + 255 : count - 1);
    if (p == plast) s_hi = hi;                             // ... the last position's bucket ends it
    __syncthreads();
    const unsigned span_lo = s_lo, span_hi = s_hi;
    const bool staged = span_hi - span_lo <= SPAN;
    if (staged) for (unsigned q = span_lo + threadIdx.x; q < span_hi; q += 256) s_keys[q - span_lo] = bucketed[q];
    __syncthreads();
    if (!mine) continue;
    u64 at = p;
    if (hi - lo > 1 && hi - lo <= rank_max) {
      unsigned r = 0;
      if (staged) for (unsigned q = lo; q < hi; q ++) r += s_keys[q - span_lo] < key ? 1u : 0u;
      else for (unsigned q = lo; q < hi; q ++) r += bucketed[q] < key ? 1u : 0u;
      at = lo + r;
    }
    sorted[at] = key;
  }
}

// ---- records, in order, streamed to the host -------------------------------------------------------------------------------------------

template <int ND>
__device__ __forceinline__ void series_record_body(const Mesh &m, const Fields *__restrict__ fields, const u64 *__restrict__ sorted, ftkx_cp_t *__restrict__ out /* pinned host memory */)
{
  constexpr int N = ND + 1;
  __shared__ u64 s_rec[kThreads / 64][64 * 9];
  if (m.counters[CNT_SERIES_DONE]) return;
  u64 count = m.counters[CNT_PASS];
  if (count > m.capacity) count = m.capacity;           // overflow: the host sees CNT_PASS and takes the pass over
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const u64 padded = (count + 63) / 64 * 64;
  for (u64 p = (u64)blockIdx.x * kThreads + threadIdx.x; p < padded; p += (u64)gridDim.x * kThreads) {
    bool fragile = false;
    double Jfrag[9];
    ftkx_cp_t rec;
    if (p < count) {
      const u64 key = sorted[p];                            // (bucket_rank_kernel: position p holds the p-th descriptor in tag order)
      const int type = (int)(key & 63u);
      const u64 q = key >> 6, step = q / m.core_cells;
      u64 lin = q - step * m.core_cells;
      const Fields &f = fields[step];
      int corner[N];
      core_corner<ND>(m, lin, corner);
      corner[ND] = f.t;
      u64 X[N][ND];
      int ids[N];
      if (ND == 2 && m.compute_degrees)
      for (int v = 0; v < N; v ++) {
        const unsigned vm = dev_fan<ND>().vert[type][v];
        int vx[3] = {0, 0, 0};
        for (int a = 0; a < ND; a ++) vx[a] = corner[a] + (int)((vm >> a) & 1u);
        const int sl = (int)((vm >> ND) & 1u);
        i64 qq[ND];
        classify_vertex<ND>(m, f.S[sl], f.V[sl], f.factor, vx, qq);
        for (int c = 0; c < ND; c ++) X[v][c] = (u64)qq[c];
        ids[v] = vertex_id<ND>(m, corner, vm);
      }
      // (the series pass is not taken with a type filter: every simplex that passed yields a record)
      if (record_is_fast<ND>(m, f, corner)) make_record_impl<ND, true>(m, f, corner, type, X, ids, false, nullptr, &rec, &fragile, Jfrag);
      else make_record_general<ND>(m, f, corner, type, X, ids, false, nullptr, &rec, &fragile, Jfrag);
      const u64 *w = reinterpret_cast<const u64 *>(&rec);
#pragma unroll
      for (int k = 0; k < 9; k ++) s_rec[wv][lane * 9 + k] = w[k];
      if (ND == 3 && fragile) {                            // (rare) the host classifies it with ITS libm and patches the type in place
        const u64 e = atomicAdd(&m.counters[CNT_FRAGILE], 1ull);
        if (e < m.fragile_capacity) {
          u64 *dst = m.fragile + e * 10;
          dst[0] = p;
          for (int k = 0; k < 9; k ++) dst[1 + k] = (u64)__double_as_longlong(Jfrag[k]);
        }
      }
    }
    // the wavefront's records are positions p0 .. p0 + 63: one contiguous run of the output, written as such (512-byte pieces, system
    // scope so that they leave for the host now and not when the kernel ends)
    __builtin_amdgcn_wave_barrier();
    const u64 p0 = p - (u64)lane;
    const unsigned nvalid = count - p0 >= 64 ? 64u : (unsigned)(count - p0);
    u64 *dst = reinterpret_cast<u64 *>(out) + p0 * 9;
    for (unsigned w8 = lane; w8 < nvalid * 9; w8 += 64)
      __hip_atomic_store(dst + w8, s_rec[wv][w8], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);   // (16-byte and non-temporal stores: the same 45 GB/s)
    __builtin_amdgcn_wave_barrier();
  }
}

template <int ND>
__global__ __launch_bounds__(kThreads) void series_record_kernel(const Mesh m, const Fields *__restrict__ fields, const u64 *__restrict__ sorted, ftkx_cp_t *__restrict__ out)
{ series_record_body<ND>(m, fields, sorted, out); }
// the same held to a third of a SIMD's registers (3D: 168 + 1.1 KB of scratch per lane instead of 369): a wavefront of it fits next to two of
// the mask kernel's -- the tail of a split pass, whose few records do not care
template <int ND>
__global__ __launch_bounds__(kThreads) __attribute__((amdgpu_waves_per_eu(3, 3))) void series_record_lean_kernel(const Mesh m, const Fields *__restrict__ fields, const u64 *__restrict__ sorted, ftkx_cp_t *__restrict__ out)
{ series_record_body<ND>(m, fields, sorted, out); }

// ---- the whole tail in one kernel (sparse data) -----------------------------------------------------------------------------------------
// Where almost everything is culled -- one moving extremum in 512^3 x 32: a few hundred coarse cells survive -- refine, exact test,
// ordering, records and the hand-over to the host are each a kernel that waits for a memory latency or two and does next to nothing; the
// kernel boundaries between them (launch, drain, ~10 us apiece) were most of the tail of a pass.  This kernel runs right behind the
// coarse cull and the factor kernel.  If few cells survived, every workgroup takes a handful of them through ALL of it on its own --
// refine, exact test of the surviving corners, records of the simplices that passed (appended to the device hit buffer, unordered) --
// and the workgroup that finishes last puts the records in tag order while it copies them into the pinned host buffer, publishes the
// results block, stores the flag the host waits for and raises CNT_SERIES_DONE, on which the kernels queued behind this one leave at
// once.  With more survivors than that it changes nothing and leaves the pass to those kernels.
constexpr int kSmallGrid = 256;                          // workgroups: one per CU (256 VGPRs + 100 AGPRs: one wavefront per SIMD -- a second round of
                                                         // workgroups would start when the first has finished)
constexpr unsigned kSmallPer = 8;                        // coarse cells (two-level) per workgroup at most, dealt round-robin; without summaries: 8 x 32 corners
                                                         // (dealt as single ROWS of coarse cells the refine phase took twice as long and the rest no less)
constexpr unsigned kSmallRank = 1024;                    // records the last workgroup ranks in LDS (more: the kernel declines after all and the bucket chain orders them)

#ifdef FTKX_SMALL_STAMPS
__device__ unsigned long long g_small_stamps[16];      // [0] earliest start (min), [k] the latest passage of phase boundary k (max)
#define SMALL_STAMP(k) do { if (threadIdx.x == 0) atomicMax(&g_small_stamps[k], (unsigned long long)wall_clock64()); } while (0)
extern "C" void ftkx_debug_small_stamps(unsigned long long *out, int reset)
{
  (void)hipDeviceSynchronize();
  (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_small_stamps), sizeof(unsigned long long) * 16);
  if (reset) { unsigned long long z[16]; for (auto &x : z) x = 0; z[0] = ~0ull; (void)hipMemcpyToSymbol(HIP_SYMBOL(g_small_stamps), z, sizeof(z)); }
}
#else
#define SMALL_STAMP(k) do { } while (0)
#endif

template <int ND>
__device__ __forceinline__ void series_small_body(const Mesh &m, const Mesh &mc, const Fields *__restrict__ steps, int two_level,
                                                  const u64 *__restrict__ refine, const u64 *__restrict__ list,
                                                  ftkx_cp_t *__restrict__ out /* pinned */, u64 *__restrict__ results, size_t nwords,
                                                  u64 *__restrict__ h_results /* pinned, coherent */, unsigned *flag, unsigned seq, unsigned *__restrict__ done,
                                                  int report_decline /* nothing is queued behind this kernel: if it declines, it says so itself */)
{
  constexpr int N = ND + 1, NVC = 1 << N, G = kThreads / NVC, NTYPES = fan_table<N>::NTYPES;
  constexpr unsigned LIST_CAP = kSmallPer * 128, PASS_CAP = 2048;   // a coarse cell is 8 x u_rows corners, u_rows <= 16 (mask_summary_rows)
  __shared__ unsigned s_rank[kSmallRank];
  static_assert(G * NTYPES <= (int)PASS_CAP / 2, "a batch's worst case must fit twice");
  __shared__ u64 s_list[LIST_CAP];
  __shared__ u64 s_pass[PASS_CAP];                       // order keys of the simplices that passed; the last workgroup: all keys of the pass
  constexpr int SUB = 4;                                 // rounds of G corners staged together
  __shared__ i64 s_vf[SUB * G][NVC][ND];
  __shared__ unsigned char s_flag[SUB * G][NVC];
  __shared__ unsigned s_tab[NTYPES];
  __shared__ unsigned short s_deg[G * NTYPES];           // (corner of the round, type) of the simplices with a degenerate value
  __shared__ unsigned s_nlist, s_npass, s_tested, s_last, s_ndeg;
  __shared__ u64 s_base;
  const int tid = threadIdx.x;
#ifdef FTKX_SMALL_STAMPS
  if (tid == 0) atomicMin(&g_small_stamps[0], (unsigned long long)wall_clock64());
#endif
  const fan_table<N> &fan = dev_fan<ND>();
  const u64 redo = (u64)(SERIES_AMBIGUOUS | SERIES_MASKS_INVALID | SERIES_INF);
  const u64 count = m.counters[two_level ? CNT_REFINE_LIST : CNT_SURVIVOR_LIST];
  // What is dealt to the workgroups (round-robin): a coarse cell of up to four rows whole, a taller one (8 x 16: u_rows == 16) in quarters of
  // four rows -- with whole cells of 128 corners a sparse series would keep a third of the workgroups busy, each with four times the corners
  const u64 SR = two_level ? (u64)(m.u_rows < 4 ? m.u_rows : 4) : 1ull, UQ = two_level ? (u64)m.u_rows / SR : 1ull;
  const u64 per = two_level ? (u64)kSmallPer : (u64)(kSmallPer * 32);
  const u64 units = count * UQ;
  // the host takes this pass over anyway: a flag of the factor job, or a slab pass abandoned by dist_cells_kernel (the halo slice is needed as
  // a whole: it has no patches, or its mask import was rejected -- nothing may be swept, and nothing more than the block is reported)
  const bool is_redo = (results[SR_STATUS] & (redo | (u64)SERIES_HALO_FULL)) != 0 || m.counters[CNT_SERIES_DONE] == 2ull;
  if (is_redo || units > per * (u64)kSmallGrid) {         // (the same for every workgroup) too much for this kernel: nothing has been changed
    if (report_decline && blockIdx.x == 0) {
      for (size_t i = tid; i < nwords; i += kThreads) h_results[i] = (i == (size_t)SR_STATUS && !is_redo) ? (results[i] | (u64)SERIES_TAIL_PENDING) : results[i];
      __threadfence_system();
      __syncthreads();
      if (tid == 0) __hip_atomic_store(flag, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    return;
  }
  const unsigned nwork = units ? (units < (u64)kSmallGrid ? (unsigned)units : (unsigned)kSmallGrid) : 1u;   // workgroups that take part (workgroup 0 always does: somebody must finish)
  if (blockIdx.x >= nwork) return;
  if (tid == 0) { s_nlist = 0; s_npass = 0; s_tested = 0; s_ndeg = 0; }
  if (tid < NTYPES) {
    unsigned w = 0;
    for (int i = 0; i < N; i ++) w |= (unsigned)fan.vert[tid][i] << (8 * i);
    s_tab[tid] = w;
  }
  __syncthreads();
  SMALL_STAMP(1);
  // this workgroup's units: blockIdx.x, + kSmallGrid, + 2 kSmallGrid, ... (at most `per` of them)
  const u64 mine_n = units > (u64)blockIdx.x ? (units - 1 - (u64)blockIdx.x) / (u64)kSmallGrid + 1 : 0ull;

  // ---- refine (the second level of the cull, as refine_kernel does it): this workgroup's coarse cells, one lane per row of a cell ----
  if (two_level) {
    const int DH = m.ext_sz[1], DD = (ND == 3) ? m.ext_sz[2] : 1, P = m.mask_pitch;
    const u64 k7f = 0x7f7f7f7f7f7f7f7full, k80 = 0x8080808080808080ull;
    const u64 urows = (u64)((DH + m.u_rows - 1) / m.u_rows);
    for (u64 idx = tid; idx < mine_n * SR; idx += kThreads) {
      const u64 unit = (u64)blockIdx.x + (idx / SR) * (u64)kSmallGrid;      // = (coarse cell, quarter of it)
      const u64 e = refine[unit / UQ];
      const int step = (int)(e >> 44);
      const unsigned want = (unsigned)((e >> 40) & 3);
      u64 lin = e & 0xffffffffffull;
      const int g = mc.core_st[0] + (int)(lin % (u64)mc.core_sz[0]); lin /= (u64)mc.core_sz[0];
      const int cyc = mc.core_st[1] + (int)(lin % (u64)mc.core_sz[1]); lin /= (u64)mc.core_sz[1];
      const int cz = (ND == 3) ? mc.core_st[2] + (int)lin : 0;
      const int j = cyc * m.u_rows + (int)((unit % UQ) * SR + idx % SR), k = cz - m.ext_st[2];
      const int cy = j + m.ext_st[1];
      if (!(cy >= m.core_st[1] && cy < m.core_st[1] + m.core_sz[1])) continue;
      const Fields f = steps[step];
      const bool need_next = (f.scope_mask & FTKX_SCOPE_INTERVAL) != 0 && (want & 2);
      // (all summary bytes first, then all mask words, unconditionally -- a word whose summary is non-zero was not written and its
      // summary, replicated, stands in: three memory latencies per lane instead of ten dependent ones)
      constexpr int NR = (ND == 3) ? 4 : 2;
      unsigned uu[2][NR][2];
      u64 ww[2][NR][2];
      bool rok[NR];
#pragma unroll
      for (int r = 0; r < NR; r ++) {
        const int dy = r & 1, dz = r >> 1;
        rok[r] = j + dy < DH && k + dz < DD;
        const int jj = rok[r] ? j + dy : j, kk = rok[r] ? k + dz : k;
#pragma unroll
        for (int sl = 0; sl < 2; sl ++) {
          const unsigned char *Up = (sl && need_next) ? f.U[1] : f.U[0];
          const unsigned char *u = Up + (size_t)m.u_pitch * ((size_t)(jj / m.u_rows) + (size_t)urows * (size_t)kk) + g;
          uu[sl][r][0] = u[0]; uu[sl][r][1] = u[1];
        }
      }
#pragma unroll
      for (int r = 0; r < NR; r ++) {
        const int dy = r & 1, dz = r >> 1;
        const int jj = rok[r] ? j + dy : j, kk = rok[r] ? k + dz : k;
        const size_t row = (size_t)jj + (size_t)DH * (size_t)kk;
#pragma unroll
        for (int sl = 0; sl < 2; sl ++) {
          const unsigned char *Mp = (sl && need_next) ? f.M[1] : f.M[0];
          const u64 *w = reinterpret_cast<const u64 *>(Mp + (size_t)P * row) + g;
          ww[sl][r][0] = w[0]; ww[sl][r][1] = w[1];
        }
      }
      u64 a0 = ~0ull, a1 = ~0ull;
#pragma unroll
      for (int r = 0; r < NR; r ++) {
        if (!rok[r]) continue;                                            // row outside the array: invalid vertices, neutral
#pragma unroll
        for (int sl = 0; sl < 2; sl ++) {
          const u64 w0 = uu[sl][r][0] ? (u64)uu[sl][r][0] * 0x0101010101010101ull : ww[sl][r][0];
          const u64 w1 = uu[sl][r][1] ? (u64)uu[sl][r][1] * 0x0101010101010101ull : ww[sl][r][1];
          const u64 v = w0 & ((w0 >> 8) | (w1 << 56));
          if (sl == 0) a0 &= v; else if (need_next) a1 &= v;
        }
      }
      u64 in_core = 0;
      for (int b = 0; b < 8; b ++) {
        const int cx = g * 8 + b + m.ext_st[0];
        if (cx >= m.core_st[0] && cx < m.core_st[0] + m.core_sz[0]) in_core |= 0x80ull << (8 * b);
      }
      u64 surv_o = 0, surv_i = 0;
      if ((f.scope_mask & FTKX_SCOPE_ORDINAL) && (want & 1)) surv_o = ~(a0 + k7f) & k80 & in_core;
      if (need_next) surv_i = ~((a0 & a1) + k7f) & k80 & in_core;
      if (!(surv_o | surv_i)) continue;
      const u64 row_lin = (u64)(cy - m.core_st[1]) * (u64)m.core_sz[0] + (ND == 3 ? (u64)(cz - m.core_st[2]) * (u64)m.core_sz[0] * (u64)m.core_sz[1] : 0ull);
      for (int b = 0; b < 8; b ++) {
        const unsigned fl = (unsigned)((surv_o >> (8 * b + 7)) & 1) | ((unsigned)((surv_i >> (8 * b + 7)) & 1) << 1);
        if (!fl) continue;
        s_list[atomicAdd(&s_nlist, 1u)] = (row_lin + (u64)(g * 8 + b + m.ext_st[0] - m.core_st[0])) | ((u64)fl << 40) | ((u64)step << 44);   // (at most 8 x u_rows <= 128 corners per coarse cell: fits)
      }
    }
  } else {
    for (u64 i = tid; i < mine_n; i += kThreads) s_list[i] = list[(u64)blockIdx.x + i * (u64)kSmallGrid];
    if (tid == 0) s_nlist = (unsigned)mine_n;
  }
  __syncthreads();
  SMALL_STAMP(2);
  const unsigned nlist = s_nlist;

  // records of the simplices parked in s_pass: appended to the device hit buffer (unordered; their order keys next to them in m.pass)
  auto flush_records = [&]() {                           // called by the whole workgroup, after a barrier that made s_npass final
    const unsigned np = s_npass;
    if (tid == 0) s_base = np ? atomicAdd(&m.counters[CNT_HITS], (u64)np) : 0ull;
    __syncthreads();
    const u64 slot0 = s_base;
    for (unsigned p = tid; p < np; p += kThreads) {
      const u64 key = s_pass[p];
      const int type = (int)(key & 63u);
      const u64 q = key >> 6, step = q / m.core_cells;
      u64 lin = q - step * m.core_cells;
      const Fields &f = steps[step];
      int corner[N];
      core_corner<ND>(m, lin, corner);
      corner[ND] = f.t;
      u64 X[N][ND];
      int ids[N];
      if (ND == 2 && m.compute_degrees)
      for (int v = 0; v < N; v ++) {
        const unsigned vm = fan.vert[type][v];
        int vx[3] = {0, 0, 0};
        for (int a = 0; a < ND; a ++) vx[a] = corner[a] + (int)((vm >> a) & 1u);
        const int sl = (int)((vm >> ND) & 1u);
        i64 qq[ND];
        classify_vertex<ND>(m, f.S[sl], f.V[sl], f.factor, vx, qq);
        for (int c = 0; c < ND; c ++) X[v][c] = (u64)qq[c];
        ids[v] = vertex_id<ND>(m, corner, vm);
      }
      bool fragile = false;
      double Jfrag[9];
      ftkx_cp_t rec;
      if (record_is_fast<ND>(m, f, corner)) make_record_impl<ND, true>(m, f, corner, type, X, ids, false, nullptr, &rec, &fragile, Jfrag);
      else make_record_general<ND>(m, f, corner, type, X, ids, false, nullptr, &rec, &fragile, Jfrag);
      const u64 slot = slot0 + p;
      if (slot < m.capacity) { m.hits[slot] = rec; m.pass[slot] = key; }
      if (ND == 3 && fragile) {
        const u64 e = atomicAdd(&m.counters[CNT_FRAGILE], 1ull);
        if (e < m.fragile_capacity) {
          u64 *fd = m.fragile + e * 10;
          fd[0] = slot;
          for (int k = 0; k < 9; k ++) fd[1 + k] = (u64)__double_as_longlong(Jfrag[k]);
        }
      }
    }
    __syncthreads();
    if (tid == 0) s_npass = 0;
    __syncthreads();
  };

  // ---- exact test (as exact_kernel does it: one lane per hypercube vertex while staging, then (corner, type) pairs over all lanes) ----
  // Staging is a chain of two memory round trips (the step's descriptor, then the field values) and a workgroup has the CU to itself: the
  // vertices of SUB x G corners are fetched in one go -- descriptors of all of them, then field values of all of them, then the
  // arithmetic -- instead of G corners per round trip pair (the cell-richest workgroup of 256^3 x 16 walked four of those: 32 of the
  // kernel's 63 us).  The test itself still goes G corners at a time, so that s_pass can be emptied in between.
  unsigned tested = 0;
  bool narrow = false;
  for (unsigned base = 0; base < nlist; base += SUB * G) {
    __syncthreads();                                     // (the staged vertices of the previous round are no longer read)
    {
      const int vtx = tid % NVC, sl = (vtx >> ND) & 1;
      u64 ent[SUB];
      const double *pS[SUB], *pV[SUB];
      double factor[SUB];
      bool live[SUB];
#pragma unroll
      for (int r = 0; r < SUB; r ++) {                   // descriptors
        const unsigned gi = (unsigned)r * G + (unsigned)(tid / NVC);
        ent[r] = base + gi < nlist ? s_list[base + gi] : ~0ull;
        live[r] = false; pS[r] = nullptr; pV[r] = nullptr; factor[r] = 0.0;
        if (ent[r] != ~0ull) {
          const Fields &f = steps[ent[r] >> 44];
          live[r] = sl == 0 || (f.scope_mask & FTKX_SCOPE_INTERVAL);
          pS[r] = f.S[sl]; pV[r] = f.V[sl]; factor[r] = f.factor;
        }
      }
      double raw[SUB][6];
      int vxs[SUB][3];
      bool usable[SUB], inner[SUB];
#pragma unroll
      for (int r = 0; r < SUB; r ++) {                   // field values: every load of the round in flight before the first is used
        for (int k = 0; k < 6; k ++) raw[r][k] = 0.0;
        for (int d = 0; d < 3; d ++) vxs[r][d] = 0;
        core_corner<ND>(m, ent[r] & 0xffffffffffull, vxs[r]);
        for (int d = 0; d < ND; d ++) vxs[r][d] += (vtx >> d) & 1;
        usable[r] = live[r] && vertex_usable<ND>(m, vxs[r]);
        inner[r] = false;
        if (usable[r]) {
          const int i = vxs[r][0] - m.ext_st[0], j = vxs[r][1] - m.ext_st[1], k = ND == 3 ? vxs[r][2] - m.ext_st[2] : 0;
          const int DW = m.ext_sz[0], DH = m.ext_sz[1];
          if (!m.scalar_mode) {
            const size_t at = arr_index<ND>(m, i, j, k) * ND;
            for (int c = 0; c < ND; c ++) raw[r][c] = pV[r][at + c];
          } else if constexpr (ND == 2) {                // gradient2D (grad.hh:17-28): clamped indices
            const int ip = clampi(i + 1, 0, DW - 1), im = clampi(i - 1, 0, DW - 1), jp = clampi(j + 1, 0, DH - 1), jm = clampi(j - 1, 0, DH - 1);
            const int ic = clampi(i, 0, DW - 1), jc = clampi(j, 0, DH - 1);
            raw[r][0] = pS[r][(size_t)ip + (size_t)DW * jc]; raw[r][1] = pS[r][(size_t)im + (size_t)DW * jc];
            raw[r][2] = pS[r][(size_t)ic + (size_t)DW * jp]; raw[r][3] = pS[r][(size_t)ic + (size_t)DW * jm];
          } else {                                       // gradient3D (grad.hh:138-146): interior vertices only
            const int DD = m.ext_sz[2];
            inner[r] = i >= 1 && i < DW - 1 && j >= 1 && j < DH - 1 && k >= 1 && k < DD - 1;
            if (inner[r]) {
              const size_t sy = (size_t)DW, sz = (size_t)DW * DH, c = (size_t)i + sy * j + sz * k;
              raw[r][0] = pS[r][c + 1]; raw[r][1] = pS[r][c - 1]; raw[r][2] = pS[r][c + sy]; raw[r][3] = pS[r][c - sy]; raw[r][4] = pS[r][c + sz]; raw[r][5] = pS[r][c - sz];
            }
          }
        }
      }
      bool mine_narrow = true;
#pragma unroll
      for (int r = 0; r < SUB; r ++) {                   // the same operations as vector_at / gradient_at on the same values, then classify_vertex's
        const unsigned gi = (unsigned)r * G + (unsigned)(tid / NVC);
        i64 q[ND];
        for (int c = 0; c < ND; c ++) q[c] = 0;
        unsigned char fl = kInvalid;
        if (usable[r]) {
          double v[ND];
          if (!m.scalar_mode) { for (int c = 0; c < ND; c ++) v[c] = raw[r][c]; }
          else if constexpr (ND == 2) { v[0] = (raw[r][0] - raw[r][1]) * (double)(m.ext_sz[0] - 1); v[1] = (raw[r][2] - raw[r][3]) * (double)(m.ext_sz[1] - 1); }
          else {
            if (inner[r]) { v[0] = 0.5 * (raw[r][0] - raw[r][1]); v[1] = 0.5 * (raw[r][2] - raw[r][3]); v[2] = 0.5 * (raw[r][4] - raw[r][5]); }
            else { v[0] = 0.0; v[1] = 0.0; v[2] = 0.0; }
          }
          fl = classify_value<ND>(v, factor[r], q);
        }
        s_flag[gi][vtx] = fl;
        for (int c = 0; c < ND; c ++) { s_vf[gi][vtx][c] = q[c]; mine_narrow = mine_narrow && fits_s32(q[c]); }
      }
      narrow = __syncthreads_and(mine_narrow) != 0;
    }
    SMALL_STAMP(10);
    for (unsigned sub = 0; sub < (unsigned)SUB && base + sub * G < nlist; sub ++) {
      if (sub) __syncthreads();
      if (s_npass > PASS_CAP - G * NTYPES) flush_records();   // (workgroup-uniform: s_npass was final at the barrier above)
      const unsigned sbase = base + sub * G;
      for (int wb = 0; wb < G * NTYPES; wb += kThreads) {
        const int w = wb + tid;
        if (w >= G * NTYPES) continue;
        const int gi = w / NTYPES, type = w % NTYPES;
        if (sbase + gi >= nlist) continue;
        const u64 e = s_list[sbase + gi];
        const unsigned scope_flags = (unsigned)((e >> 40) & 3);
        const bool wanted = fan.ordinal[type] ? (scope_flags & 1) : (scope_flags & 2);
        if (!wanted) continue;
        const Fields &f = steps[e >> 44];
        int corner[N];
        core_corner<ND>(m, e & 0xffffffffffull, corner);
        corner[ND] = f.t;
        const unsigned tab = s_tab[type];
        unsigned char flags[N];
        u64 X[N][ND];
        for (int i = 0; i < N; i ++) {
          const unsigned vm = (tab >> (8 * i)) & 0xffu;
          flags[i] = s_flag[sub * G + gi][vm];
          for (int c = 0; c < ND; c ++) X[i][c] = (u64)s_vf[sub * G + gi][vm][c];
        }
        int ids[N]; double mu[N]; bool presolved, degenerate = false;
        if (simplex_inside<ND>(m, f, 1, corner, tab, flags, X, tested, ids, mu, &presolved, narrow, &degenerate))
          s_pass[atomicAdd(&s_npass, 1u)] = order_key((e & kPassLinMask) | ((u64)type << kPassTypeShift) | ((e >> 44) << kPassStepShift), m.core_cells);
        if (degenerate) s_deg[atomicAdd(&s_ndeg, 1u)] = (unsigned short)((gi << 6) | type);   // (G * NTYPES entries at most: fits)
      }
      SMALL_STAMP(11);
      // A degenerate value takes the literal cascade -- thousands of instructions.  In line it held its wavefront in every one of the
      // loop's rounds that had one (the bowl of moving_extremum has them everywhere: 28 of this kernel's 63 us on 256^3 x 16); listed
      // and dealt over all lanes they all run at once.
      __syncthreads();
      const unsigned ndeg = s_ndeg;
      for (unsigned it = tid; it < ndeg; it += kThreads) {
        const unsigned item = s_deg[it];
        const int gi = (int)(item >> 6), type = (int)(item & 63u);
        const u64 e = s_list[sbase + gi];
        const Fields &f = steps[e >> 44];
        int corner[N];
        core_corner<ND>(m, e & 0xffffffffffull, corner);
        corner[ND] = f.t;
        const unsigned tab = s_tab[type];
        u64 X[N][ND];
        int ids[N];
        for (int i = 0; i < N; i ++) {
          const unsigned vm = (tab >> (8 * i)) & 0xffu;
          for (int c = 0; c < ND; c ++) X[i][c] = (u64)s_vf[sub * G + gi][vm][c];
          ids[i] = vertex_id<ND>(m, corner, vm);
        }
        if (sos_origin_in_simplex_resolved<ND>(X, ids))
          s_pass[atomicAdd(&s_npass, 1u)] = order_key((e & kPassLinMask) | ((u64)type << kPassTypeShift) | ((e >> 44) << kPassStepShift), m.core_cells);
      }
      __syncthreads();
      if (tid == 0) s_ndeg = 0;
    }
  }
  __syncthreads();
  SMALL_STAMP(3);
  if (s_npass) flush_records();                          // (workgroup-uniform)
  SMALL_STAMP(4);
  {
    unsigned t_sum = tested;
    for (int o = 32; o > 0; o >>= 1) t_sum += __shfl_down(t_sum, o);
    if ((tid & 63) == 0 && t_sum) atomicAdd(&s_tested, t_sum);
    __syncthreads();
    if (tid == 0) {
      if (s_tested) atomicAdd(&m.counters[CNT_SIMPLICES_TESTED], (u64)s_tested);
      if (nlist) atomicAdd(&m.counters[CNT_CELLS_SURVIVED], (u64)nlist);
    }
  }

  // ---- the workgroup that finishes last hands the pass over to the host ----
  __threadfence();
  __syncthreads();
  if (tid == 0) s_last = atomicAdd(done, 1u) == nwork - 1u ? 1u : 0u;
  __syncthreads();
  SMALL_STAMP(5);
  if (!s_last) return;
  __threadfence();
  SMALL_STAMP(6);
  const u64 nrec_all = m.counters[CNT_HITS], nfrag = m.counters[CNT_FRAGILE];
  const bool over = nrec_all > m.capacity || nfrag > m.fragile_capacity;
  const u64 nrec = over ? 0ull : nrec_all;
  const bool ranked = nrec <= (u64)kSmallRank && nrec <= (u64)PASS_CAP;
  if (!ranked && !over) {
    // More records than this workgroup can put in order (few coarse cells, many hits in them: small hit-dense 2D series).  Declined after
    // all: what this kernel counted is taken back and the pass is left to the kernels behind it, which walk the same lists again and
    // order with buckets -- tens of microseconds lost, against a host that would have to sort (rounds 3-4a: SERIES_UNORDERED; the bit now says that this happened)
    if (tid == 0) {
      m.counters[CNT_HITS] = 0ull; m.counters[CNT_FRAGILE] = 0ull; m.counters[CNT_SIMPLICES_TESTED] = 0ull; m.counters[CNT_CELLS_SURVIVED] = 0ull;
      results[SR_STATUS] |= (u64)SERIES_LATE_DECLINE;       // (for the record: the finish kernel hands it to the host with the rest)
    }
    __syncthreads();
    if (report_decline) {
      for (size_t i = tid; i < nwords; i += kThreads) h_results[i] = (i == (size_t)SR_STATUS) ? (results[i] | (u64)SERIES_TAIL_PENDING) : results[i];
      __threadfence_system();
      __syncthreads();
      if (tid == 0) __hip_atomic_store(flag, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    return;
  }
  u64 status = (u64)SERIES_EARLY | (over ? (u64)SERIES_OVERFLOW : 0ull);
  // rank of every record (by counting the smaller order keys), records copied into the pinned host buffer at their rank
  if (ranked) {
    for (u64 i = tid; i < nrec; i += kThreads) s_pass[i] = m.pass[i];
    __syncthreads();
    for (u64 i = tid; i < nrec; i += kThreads) {
      const u64 key = s_pass[i];
      unsigned r = 0;
      for (u64 q = 0; q < nrec; q ++) r += s_pass[q] < key ? 1u : 0u;
      s_rank[i] = r;
    }
    __syncthreads();
  }
  SMALL_STAMP(7);
  for (u64 w = tid; w < nrec * 9; w += kThreads) {                       // nine consecutive lanes move one record
    const u64 i = w / 9, k = w - i * 9;
    const u64 at = ranked ? (u64)s_rank[i] : i;
    __hip_atomic_store(reinterpret_cast<u64 *>(out + at) + k, reinterpret_cast<const u64 *>(m.hits + i)[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  }
  const u64 nf = over ? 0ull : nfrag;
  for (u64 w = tid; w < nf * 10; w += kThreads) {
    const u64 e = w / 10, k = w - e * 10;
    u64 v = m.fragile[w];
    if (k == 0 && ranked) v = s_rank[v];
    h_results[nwords + w] = v;
  }
  for (size_t i = tid; i < nwords; i += kThreads) {
    u64 v = results[i];
    if (i == (size_t)SR_STATUS) v |= status;
    else if (i == (size_t)SR_NHITS) v = nrec_all;
    else if (i == (size_t)SR_NFRAGILE) v = nfrag;
    else if (i >= (size_t)SR_COUNTERS && i < (size_t)SR_HEAD) {
      const int cidx = (int)i - SR_COUNTERS;
      v = m.counters[cidx];
      if (cidx == CNT_PASS) v = nrec_all;
      if (cidx == CNT_LIST_PEAK) v = m.counters[CNT_CELLS_SURVIVED];
      if (cidx == CNT_REFINE_PEAK) v = two_level ? count : 0ull;
    }
    h_results[i] = v;
  }
  SMALL_STAMP(8);
  __threadfence_system();
  __syncthreads();
  SMALL_STAMP(9);
  if (tid == 0) {
    m.counters[CNT_SERIES_DONE] = 1ull;                  // the kernels queued behind this one leave at once
    __hip_atomic_store(flag, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
  }
}

template <int ND>
__global__ __launch_bounds__(kThreads) void series_small_kernel(const Mesh m, const Mesh mc, const Fields *__restrict__ steps, int two_level, const u64 *__restrict__ refine,
                                                                const u64 *__restrict__ list, ftkx_cp_t *__restrict__ out, u64 *__restrict__ results, size_t nwords,
                                                                u64 *__restrict__ h_results, unsigned *flag, unsigned seq, unsigned *__restrict__ done, int report_decline)
{ series_small_body<ND>(m, mc, steps, two_level, refine, list, out, results, nwords, h_results, flag, seq, done, report_decline); }

// ---- finish: counters, factors and reductions to the host, then the flag --------------------------------------------------------------
// One workgroup.  Runs behind the record kernel (a kernel boundary: its stores have been released); copies the device results block
// into coherent pinned memory and stores the sequence number behind it with system scope -- the ONE thing the host waits for.
__global__ __launch_bounds__(256) void series_finish_kernel(const u64 *__restrict__ counters, u64 *__restrict__ results, size_t nwords, u64 capacity, u64 list_capacity, u64 refine_capacity,
                                                            const u64 *__restrict__ fragile, u64 fragile_capacity, u64 *__restrict__ h_results, unsigned *flag, unsigned seq)
{
  if (counters[CNT_SERIES_DONE] == 1) return;           // (the fused tail has published everything already; 2 = abandoned by the factor kernel: reported here)
  __shared__ unsigned s_over;
  const unsigned tid = threadIdx.x;
  if (tid == 0) {
    unsigned over = 0;
    const u64 hits = counters[CNT_PASS];
    if (hits > capacity || counters[CNT_LIST_PEAK] > list_capacity || counters[CNT_REFINE_PEAK] > refine_capacity || counters[CNT_FRAGILE] > fragile_capacity) over = SERIES_OVERFLOW;
    s_over = over;
    results[SR_NHITS] = hits;
    results[SR_NFRAGILE] = counters[CNT_FRAGILE];
  }
  __syncthreads();
  if (tid < (unsigned)CNT_N) results[SR_COUNTERS + tid] = counters[tid];
  __syncthreads();
  u64 nf = counters[CNT_FRAGILE];
  if (nf > fragile_capacity) nf = fragile_capacity;
  for (size_t i = tid; i < nwords; i += 256) h_results[i] = i == (size_t)SR_STATUS ? (results[i] | (u64)s_over) : results[i];
  for (size_t i = tid; i < (size_t)nf * 10; i += 256) h_results[nwords + i] = fragile[i];
  __threadfence_system();
  __syncthreads();
  if (tid == 0) __hip_atomic_store(flag, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

// The records of a pass that left them in device memory, into the pinned host buffer: a FEW workgroups on a stream of their own, next to
// the mask kernel of the pass queued behind (a copy queued through the runtime turned out to be a blit kernel with a grid that took the
// mask kernel's slots: 109 -> 183 us).  How many there are is on the device (results[SR_NHITS], final: this runs behind the finish
// kernel); the last workgroup stores the sequence number the host waits for.
__global__ __launch_bounds__(256) void series_copy_out_kernel(const ftkx_cp_t *__restrict__ src, ftkx_cp_t *__restrict__ dst /* pinned */, u64 capacity,
                                                              const u64 *__restrict__ results, unsigned *__restrict__ done, unsigned *flag, unsigned seq,
                                                              const unsigned *wait_flag /* nullable */, unsigned wait_val)
{
  // With a pass queued behind the one whose records these are: not before that pass's begin kernel has stored its number (wait_val) --
  // the records' pass is then through (same stream, in order: no event needed, and none is recorded) and the next pass's descriptors have
  // been fetched over PCIe (a read queued behind this kernel's writes took 65 us instead of 4).  This kernel may well start while the
  // records' pass is still running: sixteen workgroups whose first lanes sleep and poll -- for as long as the rest of the records' pass and
  // the next pass's begin kernel take, and for at most 2^16 polls (some 0.1 s: what keeps a begin kernel that never ran from hanging the
  // device; round 5: 2^23, eight seconds).  Slab passes do not take this form: their begin kernel can sit behind a peer's messages, and
  // the copy is ordered behind it by an event instead (series.hip, series_plan).
  // A workgroup that gives up still arrives at the counter below (marked), so that the counter is left at zero and the flag is NOT stored:
  // the host, which has waited for the records' pass itself by then, sees the copy stream drain without the flag and queues the copy again
  // without a wait (series.hip, series_complete) -- a late begin kernel delays the records, it does not lose them.
  __shared__ int s_gave_up;
  if (threadIdx.x == 0) s_gave_up = 0;
  if (wait_flag) {
    if (threadIdx.x == 0) {
      unsigned it = 0;
      while (it < (1u << 16) && (int)(__hip_atomic_load(wait_flag, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) - wait_val) < 0) { __builtin_amdgcn_s_sleep(32); it ++; }
      s_gave_up = it == (1u << 16);
    }
  }
  __syncthreads();
  const bool gave_up = s_gave_up != 0;
  u64 n = gave_up ? 0ull : results[SR_NHITS];
  if (n > capacity) n = capacity;
  const size_t nvec = (size_t)n * sizeof(ftkx_cp_t) / 8;     // (72 bytes a record: nine words)
  const u64 *s = reinterpret_cast<const u64 *>(src);
  u64 *d = reinterpret_cast<u64 *>(dst);
  // a wavefront moves runs of 64 x 8 x 8 bytes: eight loads in flight per lane, 512-byte bursts over PCIe
  const size_t per = (size_t)gridDim.x * 256 * 8;
  for (size_t base = (size_t)blockIdx.x * 256 * 8; base < nvec; base += per) {
    u64 v[8];
#pragma unroll
    for (int k = 0; k < 8; k ++) { const size_t i = base + (size_t)k * 256 + threadIdx.x; v[k] = i < nvec ? s[i] : 0ull; }
#pragma unroll
    for (int k = 0; k < 8; k ++) { const size_t i = base + (size_t)k * 256 + threadIdx.x; if (i < nvec) __builtin_nontemporal_store(v[k], &d[i]); }
  }
  __threadfence_system();
  __syncthreads();
  if (threadIdx.x == 0) {
    const unsigned before = atomicAdd(done, gave_up ? 0x10001u : 1u);      // (low half: arrivals; high half: workgroups that gave up)
    if ((before & 0xffffu) + 1 == gridDim.x) {
      *done = 0;
      __threadfence_system();
      if ((before >> 16) == 0 && !gave_up) __hip_atomic_store(flag, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
  }
}

void launch_series_copy_out(const ftkx_cp_t *src, ftkx_cp_t *dst, u64 capacity, const u64 *results, unsigned *done, unsigned *flag, unsigned seq, hipStream_t st,
                            const unsigned *wait_flag, unsigned wait_val)
{
  constexpr int wgs = 16;      // (256 / 64 / 16 / 4 workgroups cost the mask kernel next to it 94 / 55 / 3 / 0 us; with 4 the copy becomes the long pole: NOTES.md)
  hipLaunchKernelGGL(series_copy_out_kernel, dim3((unsigned)wgs), dim3(256), 0, st, src, dst, capacity, results, done, flag, seq, wait_flag, wait_val);
}

// ---- launchers -----------------------------------------------------------------------------------------------------------------------
void launch_series_begin(u64 *counters, u64 *red, size_t nslots, unsigned *hist, size_t nbins, u64 *results, size_t nresults, hipStream_t st,
                         const void *desc_src, void *desc_dst, size_t desc_bytes, unsigned *fetched, unsigned fetched_val)
{
  size_t n = (size_t)CNT_N;
  n = n > nslots ? n : nslots; n = n > nbins ? n : nbins; n = n > nresults ? n : nresults;
  if (desc_src && n < 1024) n = 1024;
  hipLaunchKernelGGL(series_begin_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, counters, red, nslots, hist, nbins, results, nresults,
                     (const u64 *)desc_src, (u64 *)desc_dst, desc_src ? desc_bytes / 8 : (size_t)0, fetched, fetched_val);
}

void launch_series_tail_begin(u64 *counters, unsigned *hist, size_t nbins, u64 *results, size_t nresults, hipStream_t st)
{
  size_t n = nbins > (size_t)CNT_N ? nbins : (size_t)CNT_N;
  n = n > nresults ? n : nresults;
  hipLaunchKernelGGL(series_tail_begin_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, counters, hist, nbins, results, nresults);
}

void launch_series_factors(Fields *steps, int nsteps, const SeriesSlice *slices, int nslices, const SeriesStep *sinfo, const u64 *red, double running_in, const u64 *running_from,
                           double safe_m, u64 *results, u64 *counters, hipStream_t st)
{ hipLaunchKernelGGL(series_factors_kernel, dim3(1), dim3(1024), 0, st, steps, nsteps, slices, nslices, sinfo, red, running_in, running_from, safe_m, results, counters); }

void launch_bucket_scan(unsigned *hist, unsigned *boff, unsigned nbins, u64 *counters, hipStream_t st, bool small_wg)
{
  if (small_wg) hipLaunchKernelGGL(bucket_scan_kernel<4>, dim3(1), dim3(256), 0, st, hist, boff, nbins, counters);
  else hipLaunchKernelGGL(bucket_scan_kernel<16>, dim3(1), dim3(1024), 0, st, hist, boff, nbins, counters);
}

void launch_bucket_scatter(const Mesh &m, unsigned *boff, u64 *bucketed, hipStream_t st, int few_wgs)
{ hipLaunchKernelGGL(bucket_scatter_kernel, dim3(few_wgs > 0 ? (unsigned)few_wgs : 256u), dim3(256), 0, st, m.pass, m.capacity, boff, m.hist, m.hist_shift, m.core_cells, bucketed, m.counters); }

void launch_bucket_rank(const Mesh &m, const u64 *bucketed, const unsigned *boff, u64 *sorted, u64 *results, hipStream_t st, int few_wgs)
{
  unsigned rank_max = 4096;
  rank_max = (unsigned)env_hook("FTKX_SERIES_HOOKS", "rank_max", 4096);
  hipLaunchKernelGGL(bucket_rank_kernel, dim3(few_wgs > 0 ? (unsigned)few_wgs : 256u), dim3(256), 0, st, bucketed, m.capacity, boff, m.hist_shift, rank_max, sorted, m.counters, results);
}

void launch_series_records(const Mesh &m, const Fields *d_fields, const u64 *sorted, ftkx_cp_t *out, hipStream_t st, bool lean)
{
  const dim3 grid(lean ? 64u : 256u * 2u);
  if (lean) {
    if (m.nd == 2) hipLaunchKernelGGL(series_record_lean_kernel<2>, grid, dim3(kThreads), 0, st, m, d_fields, sorted, out);
    else hipLaunchKernelGGL(series_record_lean_kernel<3>, grid, dim3(kThreads), 0, st, m, d_fields, sorted, out);
    return;
  }
  if (m.nd == 2) hipLaunchKernelGGL(series_record_kernel<2>, grid, dim3(kThreads), 0, st, m, d_fields, sorted, out);
  else hipLaunchKernelGGL(series_record_kernel<3>, grid, dim3(kThreads), 0, st, m, d_fields, sorted, out);
}

void launch_series_small(const Mesh &m, const Mesh &mc, const Fields *d_steps, bool two_level, const u64 *d_refine, const u64 *d_list, ftkx_cp_t *out,
                         u64 *results, size_t nwords, u64 *h_results, unsigned *flag, unsigned seq, unsigned *done, bool report_decline, hipStream_t st)
{
  const int rd = report_decline ? 1 : 0;
  if (m.nd == 2) hipLaunchKernelGGL(series_small_kernel<2>, dim3(kSmallGrid), dim3(kThreads), 0, st, m, mc, d_steps, two_level ? 1 : 0, d_refine, d_list, out, results, nwords, h_results, flag, seq, done, rd);
  else hipLaunchKernelGGL(series_small_kernel<3>, dim3(kSmallGrid), dim3(kThreads), 0, st, m, mc, d_steps, two_level ? 1 : 0, d_refine, d_list, out, results, nwords, h_results, flag, seq, done, rd);
}

void launch_series_finish(const Mesh &m, u64 *results, size_t nwords, u64 list_capacity, u64 refine_capacity, u64 *h_results, unsigned *flag, unsigned seq, hipStream_t st)
{
  hipLaunchKernelGGL(series_finish_kernel, dim3(1), dim3(256), 0, st, m.counters, results, nwords, m.capacity, list_capacity, refine_capacity, m.fragile, m.fragile_capacity,
                     h_results, flag, seq);
}

}  // namespace ftkx
