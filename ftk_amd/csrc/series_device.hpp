// The sticky quantisation factor on the device: shared by series_factors_kernel (series_kernels.hip) and the workgroup of the cull
// kernel that does the same job while the others cull (cull_exact_kernels.hip, FactorJob).
#pragma once
#include "sweep_device.hpp"

namespace ftkx {

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

// One workgroup (of THREADS lanes, for at most MAXSLICES slices).  Folds the fused reductions of the slices that were masked in this pass, forms the running minimum over the slices in
// time order, and writes every step's factor into its descriptor.  Masks were built without the per-vertex overflow rule: they stand
// only if no vertex of a step's slices is big under the step's factor (max |v| * factor < safe_m) -- otherwise the pass is flagged.
// nslices <= kSeriesMaxSlices: the per-slice values live in LDS (a chain of dependent global loads per slice cost 0.4 us apiece).
template <int THREADS, int MAXSLICES>
__device__ __forceinline__ void series_factors_body(Fields *__restrict__ steps, int nsteps, const SeriesSlice *__restrict__ slices, int nslices,
                                                    const SeriesStep *__restrict__ sinfo, const u64 *__restrict__ red, double running_in,
                                                    const u64 *__restrict__ running_from /* a previous chunk's results block, or nullptr */,
                                                    double safe_m, u64 *__restrict__ results, u64 *__restrict__ counters)
{
  constexpr int LPS = THREADS / 64, SLOTS = 64 / LPS;   // lanes per slice (64 slices per round), {min, max} slots per lane
  static_assert(THREADS % 64 == 0 && (LPS & (LPS - 1)) == 0 && LPS <= 64, "lanes per slice: a power of two");
  if (running_from) { const double r = __longlong_as_double((long long)running_from[SR_RUNNING]); running_in = r < running_in ? r : running_in; }
  __shared__ double res[MAXSLICES], mx[MAXSLICES];
  __shared__ double s_lane_min[64];
  __shared__ unsigned s_status;
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  if (tid == 0) s_status = 0;
  __syncthreads();
  for (int j0 = 0; j0 < nslices; j0 += 64) {           // LPS lanes per slice, SLOTS {min, max} slots per lane: 64 slices per round
    const int j = j0 + tid / LPS, sub = tid % LPS;
    u64 mn = 0x7fefffffffffffffull, mxb = 0ull;
    SeriesSlice sl;
    sl.red_index = -1; sl.known_res = 0; sl.known_max = 0; sl.t = 0; sl.from_res = nullptr; sl.from_max = nullptr;
    if (j < nslices) sl = slices[j];
    if (j < nslices && sl.red_index >= 0) {
      const u64 *r = red + (size_t)sl.red_index * 128 + (size_t)sub * (2 * SLOTS);
#pragma unroll
      for (int q = 0; q < SLOTS; q ++) { const u64 a = r[2 * q], b = r[2 * q + 1]; mn = a < mn ? a : mn; mxb = b > mxb ? b : mxb; }   // (bit patterns of non-negative doubles order like the values)
    }
    for (int o = LPS / 2; o > 0; o >>= 1) {
      const u64 a = __shfl_xor(mn, o), b = __shfl_xor(mxb, o);
      mn = a < mn ? a : mn; mxb = b > mxb ? b : mxb;
    }
    if (sub == 0 && j < nslices) {
      double r = __longlong_as_double((long long)mn), x = __longlong_as_double((long long)mxb);
      if (sl.from_res) { sl.known_res = __longlong_as_double((long long)*sl.from_res); sl.known_max = __longlong_as_double((long long)*sl.from_max); }
      if (sl.known_res < r) r = sl.known_res;
      if (sl.known_max > x) x = sl.known_max;
      res[j] = r; mx[j] = x;
      if (isinf(x)) atomicOr(&s_status, (unsigned)SERIES_INF);
    }
  }
  __syncthreads();
  for (int j = tid; j < nslices; j += THREADS) {
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
  for (int i = tid; i < nsteps; i += THREADS) {
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
  if (tid == 0 && s_status) {
    atomicOr((unsigned long long *)&results[SR_STATUS], (unsigned long long)s_status);
    // the host takes this pass over: the kernels queued behind this one (refine, exact test, ordering, records) leave at once -- on data that
    // needs the per-vertex overflow rule they would chew through a survivor list as long as the input -- and only the finish kernel reports
    if (counters) counters[CNT_SERIES_DONE] = 2ull;
  }
}


}  // namespace ftkx
