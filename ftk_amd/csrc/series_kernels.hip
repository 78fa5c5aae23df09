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
#include "sweep_device.hpp"

namespace ftkx {

// ---- begin: every counter, reduction slot and histogram bin of the pass in one launch ------------------------------------------------
__global__ __launch_bounds__(256) void series_begin_kernel(u64 *counters, u64 *red, size_t nslots, unsigned *hist, size_t nbins, u64 *results, size_t nresults)
{
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i < (size_t)CNT_N) counters[i] = 0ull;
  if (i < nslots) { red[2 * i] = 0x7fefffffffffffffull; red[2 * i + 1] = 0ull; }   // {min = DBL_MAX, max = 0} as bit patterns
  if (i < nbins) hist[i] = 0u;
  if (i < nresults) results[i] = 0ull;
}

// ---- the sticky factor ---------------------------------------------------------------------------------------------------------------
// nbits = clamp(ceil(log2(1 / resolution)), 8, 21) exactly as the host computes it with glibc's log2 -- except where the last bit of
// that log2 could decide, which is flagged instead (the host then takes the pass over): 1 / resolution = 2^e (1 + d) with 0 < d < 2^-32
// has log2 = e + 1.44 d, which a double rounds to e when d is small enough, and then ceil gives e where the exact value gives e + 1.
constexpr double DBL_MAX_D = 1.7976931348623157e308;

__device__ inline int nbits_of(double resolution, bool &ambiguous)
{
  const double y = 1.0 / resolution;                 // IEEE division, like the host's
  if (!(y > 256.0)) return 8;                         // log2 y <= 8 (exact at 256)
  if (y > 2097152.0) return 21;                       // log2 y >= 21 whichever way it rounds
  const u64 bits = (u64)__double_as_longlong(y);
  const int e = (int)((bits >> 52) & 0x7ffu) - 1023;
  const u64 frac = bits & ((1ull << 52) - 1ull);
  if (frac == 0) return e;                            // a power of two: log2 is exact
  if (frac < (1ull << 20)) ambiguous = true;
  return e + 1;
}

// One workgroup.  Folds the fused reductions of the slices that were masked in this pass, forms the running minimum over the slices in
// time order, and writes every step's factor into its descriptor.  Masks were built without the per-vertex overflow rule: they stand
// only if no vertex of a step's slices is big under the step's factor (max |v| * factor < safe_m) -- otherwise the pass is flagged.
// nslices <= kSeriesMaxSlices: the per-slice values live in LDS (a chain of dependent global loads per slice cost 0.4 us apiece).
__global__ __launch_bounds__(256) void series_factors_kernel(Fields *__restrict__ steps, int nsteps, const SeriesSlice *__restrict__ slices, int nslices,
                                                             const SeriesStep *__restrict__ sinfo, const u64 *__restrict__ red, double running_in,
                                                             double safe_m, u64 *__restrict__ results)
{
  __shared__ double res[kSeriesMaxSlices], mx[kSeriesMaxSlices];
  __shared__ double s_lane_min[64];
  __shared__ unsigned s_status;
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  if (tid == 0) s_status = 0;
  __syncthreads();
  for (int j = wv; j < nslices; j += 4) {              // a wavefront per slice
    const SeriesSlice sl = slices[j];
    u64 mn = 0x7fefffffffffffffull, mxb = 0ull;
    if (sl.red_index >= 0) {
      mn = red[(size_t)sl.red_index * 128 + 2 * lane]; mxb = red[(size_t)sl.red_index * 128 + 2 * lane + 1];
      for (int o = 32; o > 0; o >>= 1) {               // bit patterns of non-negative doubles order like the values
        const u64 a = __shfl_down(mn, o), b = __shfl_down(mxb, o);
        mn = a < mn ? a : mn; mxb = b > mxb ? b : mxb;
      }
    }
    if (lane == 0) {
      double r = __longlong_as_double((long long)mn), x = __longlong_as_double((long long)mxb);
      if (sl.known_res < r) r = sl.known_res;
      if (sl.known_max > x) x = sl.known_max;
      res[j] = r; mx[j] = x;
      if (isinf(x)) atomicOr(&s_status, (unsigned)SERIES_INF);
    }
  }
  __syncthreads();
  for (int j = tid; j < nslices; j += 256) {
    results[SR_HEAD + nsteps + j] = (u64)__double_as_longlong(res[j]);
    results[SR_HEAD + nsteps + nslices + j] = (u64)__double_as_longlong(mx[j]);
  }
  __syncthreads();
  // running minimum in time order: wavefront 0, a contiguous run of slices per lane, the lanes' minima scanned across the wavefront
  if (wv == 0) {
    const int per = (nslices + 63) / 64, lo = lane * per, hi = lo + per < nslices ? lo + per : nslices;
    double mine = DBL_MAX_D;
    for (int j = lo; j < hi; j ++) mine = res[j] < mine ? res[j] : mine;
    double incl = mine;
    for (int o = 1; o < 64; o <<= 1) { const double up = __shfl_up(incl, o); if (lane >= o && up < incl) incl = up; }
    double run = __shfl_up(incl, 1);                   // the minimum of everything before this lane's run
    if (lane == 0) run = DBL_MAX_D;
    run = running_in < run ? running_in : run;
    for (int j = lo; j < hi; j ++) { run = res[j] < run ? res[j] : run; res[j] = run; }
    if (lane == 63) { const double total = incl < running_in ? incl : running_in; results[SR_RUNNING] = (u64)__double_as_longlong(total); }
  }
  __syncthreads();
  for (int i = tid; i < nsteps; i += 256) {
    const SeriesStep st = sinfo[i];
    bool amb = false;
    const int nbits = nbits_of(res[st.last], amb);
    const double factor = (double)(1ull << nbits);
    bool ok = mx[st.slice0] * factor < safe_m;
    if (st.slice1 >= 0) ok = ok && mx[st.slice1] * factor < safe_m;
    if (amb) atomicOr(&s_status, (unsigned)SERIES_AMBIGUOUS);
    if (!ok) atomicOr(&s_status, (unsigned)SERIES_MASKS_INVALID);
    steps[i].factor = factor;
    results[SR_HEAD + i] = 1ull << nbits;
  }
  __syncthreads();
  if (tid == 0 && s_status) atomicOr((unsigned long long *)&results[SR_STATUS], (unsigned long long)s_status);
}

// ---- ordering without a sort ---------------------------------------------------------------------------------------------------------
// counts per bucket -> offsets (exclusive scan).  One workgroup of 16 wavefronts; a wavefront owns a contiguous sixteenth of the bins
// and holds it in registers, 64 consecutive bins per row (row r, lane l = bin base + 64 r + l: every load and store is one contiguous
// 256-byte run), so that the whole scan costs two passes over memory.  The counts are zeroed for their second life as scatter cursors;
// the fullest bucket is published (CNT_BUCKET_MAX).  nbins <= kSeriesMaxBins.
__global__ __launch_bounds__(1024) void bucket_scan_kernel(unsigned *__restrict__ hist, unsigned *__restrict__ boff, unsigned nbins, u64 *__restrict__ counters)
{
  constexpr int ROWS = kSeriesMaxBins / 1024;          // rows of 64 bins per wavefront
  __shared__ unsigned s_wave[16];
  __shared__ unsigned s_max;
  if (counters[CNT_SERIES_DONE]) return;
  const unsigned tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const unsigned seg = (nbins + 15u) / 16u, seg64 = (seg + 63u) / 64u * 64u;      // bins per wavefront, rounded up to whole rows
  const unsigned base = wv * seg64;
  if (tid == 0) s_max = 0;
  __syncthreads();
  unsigned v[ROWS];
  unsigned sum = 0, mxc = 0;
#pragma unroll
  for (int r = 0; r < ROWS; r ++) {
    const unsigned i = base + 64u * (unsigned)r + lane;
    v[r] = (64u * (unsigned)r < seg64 && i < nbins) ? hist[i] : 0u;
    sum += v[r]; mxc = v[r] > mxc ? v[r] : mxc;
  }
  unsigned tot = sum;
  for (int o = 32; o > 0; o >>= 1) { tot += __shfl_down(tot, o); const unsigned w = __shfl_down(mxc, o); mxc = w > mxc ? w : mxc; }
  if (lane == 0) { s_wave[wv] = tot; if (mxc) atomicMax(&s_max, mxc); }
  __syncthreads();
  unsigned carry = 0;
  for (unsigned q = 0; q < wv; q ++) carry += s_wave[q];
#pragma unroll
  for (int r = 0; r < ROWS; r ++) {
    if (64u * (unsigned)r >= seg64) break;               // (wave-uniform)
    unsigned incl = v[r];
    for (int o = 1; o < 64; o <<= 1) { const unsigned up = __shfl_up(incl, o); if (lane >= (unsigned)o) incl += up; }
    const unsigned i = base + 64u * (unsigned)r + lane;
    if (i < nbins) { boff[i] = carry + incl - v[r]; hist[i] = 0u; }
    carry += __shfl(incl, 63);
  }
  if (tid == 1023) boff[nbins] = carry;                  // (the last wavefront's carry is the total)
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

// the descriptors of a bucket in order: every descriptor counts the smaller ones of its bucket and takes that place.  Buckets fuller
// than rank_max stay as they are (SERIES_FIX_ORDER: the host orders those runs of the output).
__global__ __launch_bounds__(256) void bucket_rank_kernel(const u64 *__restrict__ bucketed, u64 capacity, const unsigned *__restrict__ boff, int shift, unsigned rank_max,
                                                          u64 *__restrict__ sorted, const u64 *__restrict__ counters, u64 *__restrict__ results)
{
  if (counters[CNT_SERIES_DONE]) return;
  u64 count = counters[CNT_PASS];
  if (count > capacity) count = capacity;
  if (blockIdx.x == 0 && threadIdx.x == 0 && counters[CNT_BUCKET_MAX] > rank_max) atomicOr((unsigned long long *)&results[SR_STATUS], (unsigned long long)SERIES_FIX_ORDER);
  for (u64 p = (u64)blockIdx.x * 256 + threadIdx.x; p < count; p += (u64)gridDim.x * 256) {
    const u64 key = bucketed[p];
    const unsigned b = (unsigned)(key >> shift), lo = boff[b], hi = boff[b + 1];
    u64 at = p;
    if (hi - lo > 1 && hi - lo <= rank_max) {
      unsigned r = 0;
      for (unsigned q = lo; q < hi; q ++) r += bucketed[q] < key ? 1u : 0u;
      at = lo + r;
    }
    sorted[at] = key;
  }
}

// ---- records, in order, streamed to the host -------------------------------------------------------------------------------------------

template <int ND>
__global__ __launch_bounds__(kThreads) void series_record_kernel(const Mesh m, const Fields *__restrict__ fields, const u64 *__restrict__ sorted,
                                                                 ftkx_cp_t *__restrict__ out /* pinned host memory */)
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
      for (int a = 0; a < ND; a ++) { corner[a] = m.core_st[a] + (int)(lin % (u64)m.core_sz[a]); lin /= (u64)m.core_sz[a]; }
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
      __hip_atomic_store(dst + w8, s_rec[wv][w8], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __builtin_amdgcn_wave_barrier();
  }
}

// ---- finish: counters, factors and reductions to the host, then the flag --------------------------------------------------------------
// One workgroup.  Runs behind the record kernel (a kernel boundary: its stores have been released); copies the device results block
// into coherent pinned memory and stores the sequence number behind it with system scope -- the ONE thing the host waits for.
__global__ __launch_bounds__(256) void series_finish_kernel(const u64 *__restrict__ counters, u64 *__restrict__ results, size_t nwords, u64 capacity, u64 list_capacity, u64 refine_capacity,
                                                            const u64 *__restrict__ fragile, u64 fragile_capacity, u64 *__restrict__ h_results, unsigned *flag, unsigned seq)
{
  if (counters[CNT_SERIES_DONE]) return;                // (the early tail has published everything already)
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

// ---- launchers -----------------------------------------------------------------------------------------------------------------------
void launch_series_begin(u64 *counters, u64 *red, size_t nslots, unsigned *hist, size_t nbins, u64 *results, size_t nresults, hipStream_t st)
{
  size_t n = (size_t)CNT_N;
  n = n > nslots ? n : nslots; n = n > nbins ? n : nbins; n = n > nresults ? n : nresults;
  hipLaunchKernelGGL(series_begin_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, counters, red, nslots, hist, nbins, results, nresults);
}

void launch_series_factors(Fields *steps, int nsteps, const SeriesSlice *slices, int nslices, const SeriesStep *sinfo, const u64 *red, double running_in, double safe_m,
                           u64 *results, hipStream_t st)
{ hipLaunchKernelGGL(series_factors_kernel, dim3(1), dim3(256), 0, st, steps, nsteps, slices, nslices, sinfo, red, running_in, safe_m, results); }

void launch_bucket_scan(unsigned *hist, unsigned *boff, unsigned nbins, u64 *counters, hipStream_t st)
{ hipLaunchKernelGGL(bucket_scan_kernel, dim3(1), dim3(1024), 0, st, hist, boff, nbins, counters); }

void launch_bucket_scatter(const Mesh &m, unsigned *boff, u64 *bucketed, hipStream_t st)
{ hipLaunchKernelGGL(bucket_scatter_kernel, dim3(256), dim3(256), 0, st, m.pass, m.capacity, boff, m.hist, m.hist_shift, m.core_cells, bucketed, m.counters); }

void launch_bucket_rank(const Mesh &m, const u64 *bucketed, const unsigned *boff, u64 *sorted, u64 *results, hipStream_t st)
{
  unsigned rank_max = 4096;
  if (const char *e = getenv("FTKX_SERIES_RANK_MAX")) rank_max = (unsigned)atoi(e);
  hipLaunchKernelGGL(bucket_rank_kernel, dim3(256), dim3(256), 0, st, bucketed, m.capacity, boff, m.hist_shift, rank_max, sorted, m.counters, results);
}

void launch_series_records(const Mesh &m, const Fields *d_fields, const u64 *sorted, ftkx_cp_t *out, hipStream_t st)
{
  const dim3 grid(256u * 2u);
  if (m.nd == 2) hipLaunchKernelGGL(series_record_kernel<2>, grid, dim3(kThreads), 0, st, m, d_fields, sorted, out);
  else hipLaunchKernelGGL(series_record_kernel<3>, grid, dim3(kThreads), 0, st, m, d_fields, sorted, out);
}

void launch_series_finish(const Mesh &m, u64 *results, size_t nwords, u64 list_capacity, u64 refine_capacity, u64 *h_results, unsigned *flag, unsigned seq, hipStream_t st)
{
  hipLaunchKernelGGL(series_finish_kernel, dim3(1), dim3(256), 0, st, m.counters, results, nwords, m.capacity, list_capacity, refine_capacity, m.fragile, m.fragile_capacity,
                     h_results, flag, seq);
}

}  // namespace ftkx
