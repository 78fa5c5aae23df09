// The whole pass over a SMALL series in ONE launch (round 5).
//
// Every size the reference's own tests use (tests/test_critical_point_tracking_woven.cpp:32-37: woven 31 x 37 x 32; moving_extremum 21^3) and
// BASELINE config 1 (woven 128 x 128 x 10) is a few MB: the ten launches of the device-driven pass -- begin, masks, cull, refine, exact test,
// scan, scatter, rank, records, finish -- are ten kernel boundaries of ~8 us around microseconds of work.  Here ONE kernel of at most one
// workgroup per CU does what the reference's update_timestep does for every step of the series (critical_point_tracker_2d_regular.hh:263-433,
// critical_point_tracker_3d_regular.hh:150-308), with two device-wide barriers between its phases:
//
//   1  the slices' reductions (ndarray::resolution(), ndarray.hh:770-778: the smallest non-zero |v| below 1 / hint, and max |v|), in chunks
//      --------------------------------------------------------------------------------------------------- barrier
//   2  every workgroup for itself: the sticky running minimum and nbits per step (update_vector_field_scaling_factor,
//      critical_point_tracker.hh:850-864; series_device.hpp: nbits_of)
//   3  a contiguous range of (step, corner) per workgroup: vertices quantised once per corner, the strict-sign cull per simplex, the robust
//      integer test (cp_device.hpp), degenerate simplices dealt over all lanes; the order keys of the simplices that passed stay in LDS
//      --------------------------------------------------------------------------------------------------- barrier
//   4  offsets from the workgroups' counts; own keys ranked in LDS (ranges are contiguous in the order key: ranks within a workgroup +
//      the offset = the place in tag order); records built in that order (the FP64 half: sweep_device.hpp, make_record_*) and written
//      straight into the pinned host buffer; the workgroup that finishes last publishes the results block and the flag the host waits for.
//
// No masks are built (nothing is left for a later pass to reuse -- at these sizes there is nothing to save), no survivor lists, no sort.
// Anything this kernel cannot decide -- a factor that hangs on the last bit of log2, more hits than a workgroup parks -- is flagged and the
// host sweeps the steps the usual way (series.hip).  Host side: series.hip, series_one_*.
#include "sweep_device.hpp"
#include "series_device.hpp"

namespace ftkx {

namespace {

// Device-wide barrier over the kernel's workgroups.  They are all resident when the kernel has the device to itself (at most one workgroup
// per CU) -- but nothing guarantees that: next to another context's kernels some workgroups may not get a CU until resident ones leave, and
// those would wait here for ever.  So the wait is bounded (2 ms against microseconds of work): whoever gives up says so in `abort`, everybody
// leaves through the kernel's common exit, and the host sweeps the steps the usual way.  false: give up.
__device__ inline bool grid_barrier(unsigned *ctr, unsigned nwg, unsigned *abort)
{
  __shared__ int s_ok;
  __syncthreads();
  if (threadIdx.x == 0) {
    __threadfence();
    __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    const unsigned long long t0 = wall_clock64();
    int ok = 1;
    while (__hip_atomic_load(ctr, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < nwg) {
      if (__hip_atomic_load(abort, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) || wall_clock64() - t0 > 200000ull) {      // (100 MHz: 2 ms)
        __hip_atomic_store(abort, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        ok = 0;
        break;
      }
      __builtin_amdgcn_s_sleep(2);
    }
    s_ok = ok;
  }
  __syncthreads();
  return s_ok != 0;
}

}  // namespace

template <int ND>
__global__ __launch_bounds__(kThreads) void series_one_kernel(const Mesh m, const OneArgs a)
{
  constexpr int N = ND + 1, NVC = 1 << N, G = kThreads / NVC, NTYPES = fan_table<N>::NTYPES, SUB = 4;
  constexpr unsigned KCAP = kOneKeys;
  __shared__ Fields s_fields[kOneMaxSteps];
  __shared__ double s_res[kOneMaxSlices], s_mx[kOneMaxSlices], s_pm[kOneMaxSlices];
  __shared__ u64 s_keys[KCAP], s_sorted[KCAP];
  __shared__ i64 s_vf[SUB * G][NVC][ND];
  __shared__ unsigned char s_flag[SUB * G][NVC];
  __shared__ unsigned s_tab[NTYPES];
  __shared__ unsigned short s_deg[G * NTYPES];
  __shared__ unsigned s_nkeys, s_ndeg, s_status, s_tested, s_cells, s_last;
  __shared__ double s_rmn[kThreads / 64], s_rmx[kThreads / 64];
  __shared__ u64 s_base, s_total;
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const unsigned nwg = gridDim.x, w = blockIdx.x;
  const fan_table<N> &fan = dev_fan<ND>();
  unsigned *bar = reinterpret_cast<unsigned *>(a.scratch + ONE_BAR);      // [0], [1]: the two barriers, [2]: arrivals at the exit, [3]: somebody gave up
  u64 *parts = a.scratch + ONE_PARTS;                          // [nslices * kOneParts][2]
  u64 *counts = a.scratch + ONE_COUNTS;                        // [nwg]
  if (tid == 0) { s_nkeys = 0; s_ndeg = 0; s_status = 0; s_tested = 0; s_cells = 0; s_base = 0; s_total = 0; }
  if (tid < NTYPES) {
    unsigned t = 0;
    for (int i = 0; i < N; i ++) t |= (unsigned)fan.vert[tid][i] << (8 * i);
    s_tab[tid] = t;
  }

  // ---- 1: reductions, chunk = (slice, part) --------------------------------------------------------------------------------------------
  {
    const int DW = m.ext_sz[0], DH = m.ext_sz[1], DD = (ND == 3) ? m.ext_sz[2] : 1;
    const size_t nv = (size_t)DW * DH * DD, per = (nv + kOneParts - 1) / kOneParts;
    const int nchunks = a.nslices * kOneParts;
    for (int c = (int)w; c < nchunks; c += (int)nwg) {
      const OneSlice sl = a.slice[c / kOneParts];
      const size_t lo = (size_t)(c % kOneParts) * per, hi = lo + per < nv ? lo + per : nv;
      double mn = DBL_MAX_D, mx = 0.0;
      for (size_t idx = lo + tid; idx < hi; idx += kThreads) {
        const int i = (int)(idx % DW), j = (int)((idx / DW) % DH), k = (int)(idx / ((size_t)DW * DH));
        double v[ND];
        vector_at<ND>(m, sl.S, sl.V, i, j, k, v);
        for (int q = 0; q < ND; q ++) {                          // (mask_kernel's fused reduction, sweep_kernels.hip: the WHOLE array, like ndarray::resolution())
          const double x = fabs(v[q]);
          mn = fmin(mn, (x == 0.0 || !(x < a.cap)) ? DBL_MAX_D : x);
          mx = fmax(mx, x);
        }
      }
      for (int o = 32; o > 0; o >>= 1) { mn = fmin(mn, __shfl_down(mn, o)); mx = fmax(mx, __shfl_down(mx, o)); }
      __syncthreads();
      if (lane == 0) { s_rmn[wv] = mn; s_rmx[wv] = mx; }
      __syncthreads();
      if (tid == 0) {
        for (int q = 1; q < kThreads / 64; q ++) { mn = fmin(mn, s_rmn[q]); mx = fmax(mx, s_rmx[q]); }
        parts[2 * c] = (u64)__double_as_longlong(mn); parts[2 * c + 1] = (u64)__double_as_longlong(mx);
      }
    }
  }
  bool alive = grid_barrier(bar + 0, nwg, bar + 3);

  // ---- 2: factors (every workgroup for itself) -----------------------------------------------------------------------------------------
  if (alive && tid < a.nslices) {
    double mn = DBL_MAX_D, mx = 0.0;
    for (int p = 0; p < kOneParts; p ++) {
      const u64 x = __hip_atomic_load(&parts[2 * (tid * kOneParts + p)], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const u64 y = __hip_atomic_load(&parts[2 * (tid * kOneParts + p) + 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      mn = fmin(mn, __longlong_as_double((long long)x)); mx = fmax(mx, __longlong_as_double((long long)y));
    }
    s_res[tid] = mn; s_mx[tid] = mx;
    if (isinf(mx)) atomicOr(&s_status, (unsigned)SERIES_INF);
  }
  __syncthreads();
  if (alive && tid == 0) {
    double run = a.running_in;
    if (a.running_from) { const double r = __longlong_as_double((long long)__hip_atomic_load(&a.running_from[SR_RUNNING], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)); run = r < run ? r : run; }
    for (int j = 0; j < a.nslices; j ++) { run = s_res[j] < run ? s_res[j] : run; s_pm[j] = run; }
  }
  __syncthreads();
  if (alive && tid < a.nsteps) {
    const OneStep st = a.step[tid];
    bool amb = false;
    const int nbits = nbits_of(s_pm[st.last], amb);
    if (amb) atomicOr(&s_status, (unsigned)SERIES_AMBIGUOUS);
    Fields f;
    f.S[0] = a.slice[st.slice0].S; f.V[0] = a.slice[st.slice0].V; f.J[0] = a.slice[st.slice0].J; f.M[0] = nullptr; f.U[0] = nullptr;
    f.S[1] = nullptr; f.V[1] = nullptr; f.J[1] = nullptr; f.M[1] = nullptr; f.U[1] = nullptr;
    if (st.slice1 >= 0) { f.S[1] = a.slice[st.slice1].S; f.V[1] = a.slice[st.slice1].V; f.J[1] = a.slice[st.slice1].J; }
    f.factor = (double)(1ull << nbits); f.t = st.t; f.scope_mask = st.scope;
    s_fields[tid] = f;
  }
  __syncthreads();
  const unsigned flagged = s_status;                             // (the same in every workgroup: computed from the same numbers)

  // ---- 3: the cells of this workgroup: [lo, hi) of (step * cells + corner) -------------------------------------------------------------
  const u64 cells = m.core_cells, total_cs = cells * (u64)a.nsteps;
  const u64 per_wg = (total_cs + nwg - 1) / nwg;
  const u64 c_lo = (u64)w * per_wg < total_cs ? (u64)w * per_wg : total_cs, c_hi = c_lo + per_wg < total_cs ? c_lo + per_wg : total_cs;
  unsigned tested = 0;
  bool overflow = false;
  if (alive && !flagged) {
    for (u64 base = c_lo; base < c_hi; base += SUB * G) {
      __syncthreads();                                           // (the staged vertices of the round before are no longer read)
      bool narrow;
      {
        const int vtx = tid % NVC, sl = (vtx >> ND) & 1;
        bool mine_narrow = true;
#pragma unroll
        for (int r = 0; r < SUB; r ++) {
          const unsigned gi = (unsigned)r * G + (unsigned)(tid / NVC);
          const u64 cs = base + gi;
          i64 q[ND];
          for (int c = 0; c < ND; c ++) q[c] = 0;
          unsigned char fl = kInvalid;
          if (cs < c_hi) {
            const u64 step = cs / cells;
            const Fields &f = s_fields[step];
            if (sl == 0 || (f.scope_mask & FTKX_SCOPE_INTERVAL)) {
              int vx[3] = {0, 0, 0};
              core_corner<ND>(m, cs - step * cells, vx);
              for (int d = 0; d < ND; d ++) vx[d] += (vtx >> d) & 1;
              fl = classify_vertex<ND>(m, f.S[sl], f.V[sl], f.factor, vx, q);
            }
          }
          s_flag[gi][vtx] = fl;
          for (int c = 0; c < ND; c ++) { s_vf[gi][vtx][c] = q[c]; mine_narrow = mine_narrow && fits_s32(q[c]); }
        }
        narrow = __syncthreads_and(mine_narrow) != 0;
      }
      // (statistics: the cells the mask cull of the kernel chain would have let through -- no strict sign bit common to all the vertices a
      // scope reads; vertices outside the domain and non-finite ones are neutral, as their mask bytes are)
      for (unsigned gi = tid; gi < (unsigned)(SUB * G); gi += kThreads) {
        const u64 cs = base + gi;
        if (cs >= c_hi) continue;
        const int scope = s_fields[cs / cells].scope_mask;
        unsigned a0 = 0x3f, a1 = 0x3f;
        for (int v = 0; v < NVC; v ++) {
          const unsigned char fl = s_flag[gi][v];
          const unsigned b = (fl & (kInvalid | kNonFinite)) ? 0x3fu : (unsigned)(fl & 0x3f);
          if ((v >> ND) & 1) a1 &= b; else a0 &= b;
        }
        if (((scope & FTKX_SCOPE_ORDINAL) && a0 == 0) || ((scope & FTKX_SCOPE_INTERVAL) && (a0 & a1) == 0)) atomicAdd(&s_cells, 1u);
      }
      for (unsigned sub = 0; sub < (unsigned)SUB && base + sub * G < c_hi; sub ++) {
        if (sub) __syncthreads();
        const u64 sbase = base + sub * G;
        for (int wb = 0; wb < G * NTYPES; wb += kThreads) {
          const int wi = wb + tid;
          if (wi >= G * NTYPES) continue;
          const int gi = wi / NTYPES, type = wi % NTYPES;
          const u64 cs = sbase + (u64)gi;
          if (cs >= c_hi) continue;
          const u64 step = cs / cells, lin = cs - step * cells;
          const Fields &f = s_fields[step];
          const bool wanted = fan.ordinal[type] ? (f.scope_mask & FTKX_SCOPE_ORDINAL) : (f.scope_mask & FTKX_SCOPE_INTERVAL);
          if (!wanted) continue;
          int corner[N];
          core_corner<ND>(m, lin, corner);
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
          if (simplex_inside<ND>(m, f, 1, corner, tab, flags, X, tested, ids, mu, &presolved, narrow, &degenerate)) {
            const unsigned at = atomicAdd(&s_nkeys, 1u);
            if (at < KCAP) s_keys[at] = ((step * cells + lin) << 6) | (u64)type;
          }
          if (degenerate) s_deg[atomicAdd(&s_ndeg, 1u)] = (unsigned short)((gi << 6) | type);
        }
        __syncthreads();
        const unsigned ndeg = s_ndeg;
        for (unsigned it = tid; it < ndeg; it += kThreads) {       // degenerate values: the literal cascade, dealt over all lanes
          const unsigned item = s_deg[it];
          const int gi = (int)(item >> 6), type = (int)(item & 63u);
          const u64 cs = sbase + (u64)gi;
          const u64 step = cs / cells, lin = cs - step * cells;
          const Fields &f = s_fields[step];
          int corner[N];
          core_corner<ND>(m, lin, corner);
          corner[ND] = f.t;
          const unsigned tab = s_tab[type];
          u64 X[N][ND];
          int ids[N];
          for (int i = 0; i < N; i ++) {
            const unsigned vm = (tab >> (8 * i)) & 0xffu;
            for (int c = 0; c < ND; c ++) X[i][c] = (u64)s_vf[sub * G + gi][vm][c];
            ids[i] = vertex_id<ND>(m, corner, vm);
          }
          if (sos_origin_in_simplex_resolved<ND>(X, ids)) {
            const unsigned at = atomicAdd(&s_nkeys, 1u);
            if (at < KCAP) s_keys[at] = ((step * cells + lin) << 6) | (u64)type;
          }
        }
        __syncthreads();
        if (tid == 0) s_ndeg = 0;
      }
    }
    __syncthreads();
    overflow = s_nkeys > KCAP;
    unsigned t_sum = tested;
    for (int o = 32; o > 0; o >>= 1) t_sum += __shfl_down(t_sum, o);
    if (lane == 0 && t_sum) atomicAdd(&s_tested, t_sum);
    __syncthreads();
  }
  if (tid == 0) {
    __hip_atomic_store(&counts[w], (u64)(overflow ? KCAP : s_nkeys) | (overflow ? (1ull << 63) : 0ull), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (s_tested) atomicAdd((unsigned long long *)(a.scratch + ONE_TESTED), (unsigned long long)s_tested);
    if (s_cells) atomicAdd((unsigned long long *)(a.scratch + ONE_CELLS), (unsigned long long)s_cells);
  }
  alive = alive && grid_barrier(bar + 1, nwg, bar + 3);
  if (!alive && tid == 0) s_status |= (unsigned)SERIES_OVERFLOW;      // (given up: the host sweeps the steps the usual way)
  __syncthreads();

  // ---- 4: offsets, ranks, records ------------------------------------------------------------------------------------------------------
  if (alive && tid == 0) {
    u64 run = 0, mine = 0;
    bool any_over = false;
    for (unsigned q = 0; q < nwg; q ++) {
      const u64 cq = __hip_atomic_load(&counts[q], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (q == w) mine = run;
      run += cq & ~(1ull << 63);
      any_over = any_over || (cq >> 63);
    }
    s_base = mine; s_total = run;
    if (any_over || run > a.capacity) s_status |= (unsigned)SERIES_OVERFLOW;
  }
  __syncthreads();
  const unsigned status = s_status;
  const unsigned nk = (status || overflow) ? 0u : s_nkeys;
  if (!status) {
    for (unsigned i = tid; i < nk; i += kThreads) {               // rank = the number of smaller keys (keys are unique)
      const u64 key = s_keys[i];
      unsigned r = 0;
      for (unsigned q = 0; q < nk; q ++) r += s_keys[q] < key ? 1u : 0u;
      s_sorted[r] = key;
    }
    __syncthreads();
    for (unsigned p = tid; p < nk; p += kThreads) {
      const u64 key = s_sorted[p];
      const int type = (int)(key & 63u);
      const u64 q = key >> 6, step = q / cells, lin = q - step * cells;
      const Fields &f = s_fields[step];
      int corner[N];
      core_corner<ND>(m, lin, corner);
      corner[ND] = f.t;
      u64 X[N][ND];
      int ids[N];
      if (ND == 2 && m.compute_degrees)
      for (int v = 0; v < N; v ++) {
        const unsigned vm = fan.vert[type][v];
        int vx[3] = {0, 0, 0};
        for (int d = 0; d < ND; d ++) vx[d] = corner[d] + (int)((vm >> d) & 1u);
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
      const u64 at = s_base + p;
      const u64 *src = reinterpret_cast<const u64 *>(&rec);
      u64 *dst = reinterpret_cast<u64 *>(a.out + at);
#pragma unroll
      for (int k = 0; k < 9; k ++) __hip_atomic_store(dst + k, src[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      if (ND == 3 && fragile) {
        const u64 e = atomicAdd((unsigned long long *)(a.scratch + ONE_NFRAG), 1ull);
        if (e < a.fragile_capacity) {
          u64 *fd = a.fragile + e * 10;
          fd[0] = at;
          for (int k = 0; k < 9; k ++) fd[1 + k] = (u64)__double_as_longlong(Jfrag[k]);
        }
      }
    }
  }
  // ---- the workgroup that finishes last hands the pass over to the host ----
  __threadfence_system();
  __syncthreads();
  if (tid == 0) s_last = __hip_atomic_fetch_add(bar + 2, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT) == nwg - 1u ? 1u : 0u;
  __syncthreads();
  if (!s_last) return;
  const u64 nfrag_all = __hip_atomic_load(a.scratch + ONE_NFRAG, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  const u64 tested_all = __hip_atomic_load(a.scratch + ONE_TESTED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  const u64 cells_all = __hip_atomic_load(a.scratch + ONE_CELLS, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  u64 st = (u64)status;
  if (nfrag_all > a.fragile_capacity || __hip_atomic_load(bar + 3, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) st |= (u64)SERIES_OVERFLOW;
  const bool good = st == 0;
  const u64 nrec = good ? s_total : 0ull, nf = good ? nfrag_all : 0ull;
  const int n = a.nsteps, k = a.nslices;
  for (size_t i = tid; i < a.nwords; i += kThreads) {
    u64 v = 0ull;
    if (i == (size_t)SR_STATUS) v = st | (good ? (u64)(SERIES_EARLY | SERIES_ONE) : 0ull);
    else if (i == (size_t)SR_RUNNING) v = (u64)__double_as_longlong(k ? s_pm[k - 1] : a.running_in);
    else if (i == (size_t)SR_NHITS) v = nrec;
    else if (i == (size_t)SR_NFRAGILE) v = nf;
    else if (i == (size_t)(SR_COUNTERS + CNT_PASS) || i == (size_t)(SR_COUNTERS + CNT_HITS)) v = s_total;
    else if (i == (size_t)(SR_COUNTERS + CNT_SIMPLICES_TESTED)) v = tested_all;
    else if (i == (size_t)(SR_COUNTERS + CNT_CELLS_SURVIVED)) v = cells_all;
    else if (i == (size_t)(SR_COUNTERS + CNT_FRAGILE)) v = nfrag_all;
    else if (i >= (size_t)SR_HEAD && i < (size_t)(SR_HEAD + n)) v = (u64)s_fields[i - SR_HEAD].factor;
    else if (i >= (size_t)(SR_HEAD + n) && i < (size_t)(SR_HEAD + n + k)) v = (u64)__double_as_longlong(s_res[i - SR_HEAD - n]);
    else if (i >= (size_t)(SR_HEAD + n + k) && i < (size_t)(SR_HEAD + n + 2 * k)) v = (u64)__double_as_longlong(s_mx[i - SR_HEAD - n - k]);
    a.results[i] = v;                                            // (the device copy: a pass chained behind this one reads SR_RUNNING there)
    a.h_results[i] = v;
  }
  for (u64 wd = tid; wd < nf * 10; wd += kThreads) a.h_results[a.nwords + wd] = a.fragile[wd];
  if (tid < 4) bar[tid] = 0u;                                    // as found, for the next launch
  if (tid == 4) { a.scratch[ONE_NFRAG] = 0ull; a.scratch[ONE_TESTED] = 0ull; a.scratch[ONE_CELLS] = 0ull; }
  __threadfence_system();
  __syncthreads();
  if (tid == 0) __hip_atomic_store(a.flag, a.seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

void launch_series_one(const Mesh &m, const OneArgs &a, int nwg, hipStream_t st)
{
  if (m.nd == 2) hipLaunchKernelGGL(series_one_kernel<2>, dim3((unsigned)nwg), dim3(kThreads), 0, st, m, a);
  else hipLaunchKernelGGL(series_one_kernel<3>, dim3((unsigned)nwg), dim3(kThreads), 0, st, m, a);
}

}  // namespace ftkx
