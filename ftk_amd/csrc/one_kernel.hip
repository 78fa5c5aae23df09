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
//   3  the (step, corner) range in BLOCKS of one staging batch, a run of consecutive blocks per workgroup: vertices quantised once per corner, the cell-level and the per-simplex strict-sign cull, the
//      robust integer test (cp_device.hpp), degenerate simplices dealt over all lanes; the order keys of the simplices that passed stay
//      in LDS, a count per block goes to device memory
//      --------------------------------------------------------------------------------------------------- barrier
//   4  every workgroup scans the blocks' counts for the offsets of its own blocks; a block's keys ranked among themselves in LDS (a block is
//      a contiguous range of the order key: rank + offset = the place in tag order); records built in that order (the FP64 half:
//      sweep_device.hpp, make_record_*) and written into the pinned host buffer in contiguous runs; the workgroup that finishes last
//      publishes the results block and the flag the host waits for.
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
      __builtin_amdgcn_s_sleep(1);
    }
    s_ok = ok;
  }
  __syncthreads();
  return s_ok != 0;
}

}  // namespace

#ifdef FTKX_ONE_STAMPS
__device__ unsigned long long g_one_stamps[16];      // [k]: the latest passage of phase boundary k over all workgroups (diagnostic builds: -DFTKX_ONE_STAMPS)
#define ONE_STAMP(k) do { if (threadIdx.x == 0) atomicMax(&g_one_stamps[k], (unsigned long long)wall_clock64()); } while (0)
extern "C" void ftkx_debug_one_stamps(unsigned long long *out, int reset)
{
  (void)hipDeviceSynchronize();
  (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_one_stamps), sizeof(unsigned long long) * 16);
  if (reset) { unsigned long long z[16]; for (auto &x : z) x = 0; (void)hipMemcpyToSymbol(HIP_SYMBOL(g_one_stamps), z, sizeof(z)); }
}
#else
#define ONE_STAMP(k) do { } while (0)
#endif

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
  __shared__ unsigned short s_deg[G * NTYPES];                  // (gi << 6) | type: gi < SUB * G <= 128
  __shared__ unsigned s_nkeys, s_ndeg, s_status, s_tested, s_cells, s_last, s_nsurv;
  __shared__ unsigned short s_surv[SUB * G];
  __shared__ double s_rmn[kThreads / 64], s_rmx[kThreads / 64];
  __shared__ unsigned s_wsum[kThreads / 64], s_lanepre[kThreads], s_bstart[kOneOwnBlocks], s_bcnt[kOneOwnBlocks], s_boff[kOneOwnBlocks], s_pos[KCAP];
  __shared__ unsigned s_bc[kOneMaxBlocks];                        // every block's count (phase 4)
  __shared__ u64 s_rec[kThreads / 64][64 * 9];                   // a wavefront's records, staged for contiguous stores
  __shared__ u64 s_base, s_total;
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const unsigned nwg = gridDim.x, w = blockIdx.x;
  const fan_table<N> &fan = dev_fan<ND>();
  unsigned *bar = reinterpret_cast<unsigned *>(a.scratch + ONE_BAR);      // [0], [1]: the two barriers, [2]: arrivals at the exit, [3]: somebody gave up
  u64 *parts = a.scratch + ONE_PARTS;                          // [nslices * kOneParts][2]
  unsigned *bcount = reinterpret_cast<unsigned *>(a.scratch + ONE_BCOUNT);      // [nblocks]
  if (tid == 0) { s_nkeys = 0; s_ndeg = 0; s_status = 0; s_tested = 0; s_cells = 0; s_base = 0; s_total = 0; }
  if (tid < NTYPES) {
    unsigned t = 0;
    for (int i = 0; i < N; i ++) t |= (unsigned)fan.vert[tid][i] << (8 * i);
    s_tab[tid] = t;
  }

#ifdef FTKX_ONE_STAMPS
  if (threadIdx.x == 0 && blockIdx.x == 0) g_one_stamps[0] = wall_clock64();
#endif
  // ---- 1: reductions, chunk = (slice, part): as many chunks as there are workgroups, at least ---------------------------------------------
  const int parts_per_slice = (int)min((unsigned)kOneParts, max(1u, (nwg + (unsigned)a.nslices - 1u) / (unsigned)a.nslices));
  {
    const unsigned DW = (unsigned)m.ext_sz[0], DH = (unsigned)m.ext_sz[1], DD = (ND == 3) ? (unsigned)m.ext_sz[2] : 1u;
    const unsigned nv = DW * DH * DD, per = (nv + (unsigned)parts_per_slice - 1u) / (unsigned)parts_per_slice;      // (small series: far below 2^32 vertices)
    const int nchunks = a.nslices * parts_per_slice;
    for (int c = (int)w; c < nchunks; c += (int)nwg) {
      const OneSlice sl = a.slice[c / parts_per_slice];
      const unsigned lo = (unsigned)(c % parts_per_slice) * per, hi = lo + per < nv ? lo + per : nv;
      double mn = DBL_MAX_D, mx = 0.0;
      for (unsigned idx = lo + (unsigned)tid; idx < hi; idx += kThreads) {
        const unsigned row = idx / DW, i = idx - row * DW, k = row / DH, j = row - k * DH;
        double v[ND];
        vector_at<ND>(m, sl.S, sl.V, (int)i, (int)j, (int)k, v);
        for (int q = 0; q < ND; q ++) {                          // (mask_kernel's fused reduction, mask_kernels.hip: the WHOLE array, like ndarray::resolution())
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
  ONE_STAMP(1);
  bool alive = grid_barrier(bar + 0, nwg, bar + 3);
  ONE_STAMP(2);

  // ---- 2: factors (every workgroup for itself) -----------------------------------------------------------------------------------------
  if (alive) {
    // every part of every slice by its own lane (the loads of a lane that walked a slice's parts went out one L2 round trip after the other)
    if (tid < a.nslices) { s_res[tid] = DBL_MAX_D; s_mx[tid] = 0.0; }
    __syncthreads();
    const int nparts = a.nslices * parts_per_slice;
    for (int c = tid; c < nparts; c += kThreads) {
      const u64 x = __hip_atomic_load(&parts[2 * c], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const u64 y = __hip_atomic_load(&parts[2 * c + 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      // (bit patterns of non-negative doubles order like the values: LDS atomics on the patterns)
      atomicMin(reinterpret_cast<unsigned long long *>(&s_res[c / parts_per_slice]), (unsigned long long)x);
      atomicMax(reinterpret_cast<unsigned long long *>(&s_mx[c / parts_per_slice]), (unsigned long long)y);
    }
    __syncthreads();
    if (tid < a.nslices && isinf(s_mx[tid])) atomicOr(&s_status, (unsigned)SERIES_INF);
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
  ONE_STAMP(3);

  // ---- 3: the cells: blocks b = w, w + nwg, ... of SUB x G consecutive (step * cells + corner) -----------------------------------------
  const u64 cells = m.core_cells, total_cs = cells * (u64)a.nsteps;
  constexpr unsigned BS = SUB * G;
  const unsigned nblocks = (unsigned)((total_cs + BS - 1) / BS);
  unsigned tested = 0;
  bool overflow = false;
  unsigned nown = 0;
  if (alive && !flagged) {
    // (a workgroup's blocks are CONSECUTIVE: dealt round-robin -- hits cluster, and a contiguous share leaves some workgroups with most of them --
    // the workgroups finished closer together, 14 instead of 20 us of waiting at the barrier on woven 128^2 x 10, but each of them took longer
    // over blocks that share no rows: 38 instead of 25 us)
    const unsigned bpw = (nblocks + nwg - 1) / nwg;
    for (unsigned blk = w * bpw; blk < nblocks && blk < (w + 1) * bpw; blk ++, nown ++) {
      const u64 base = (u64)blk * BS, c_hi = base + BS < total_cs ? base + BS : total_cs;
      const unsigned k_before = s_nkeys;                         // (workgroup-uniform: read behind the barrier that closed the block before)
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
      // the cell-level cull (what the mask cull of the kernel chain does): a corner none of whose scopes lacks a strict sign bit common to all the
      // vertices it reads has no simplex to test -- vertices outside the domain and non-finite ones are neutral, as their mask bytes are.  The
      // survivors of the SUB x G corners staged go on a list; the (corner, type) pairs below walk that list only
      if (tid == 0) s_nsurv = 0;
      __syncthreads();
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
        const unsigned want = (((scope & FTKX_SCOPE_ORDINAL) && a0 == 0) ? 1u : 0u) | (((scope & FTKX_SCOPE_INTERVAL) && (a0 & a1) == 0) ? 2u : 0u);
        if (want) s_surv[atomicAdd(&s_nsurv, 1u)] = (unsigned short)(gi | (want << 8));
      }
      __syncthreads();
      const unsigned nsurv = s_nsurv;
      if (tid == 0) s_cells += nsurv;
      for (unsigned sb = 0; sb < nsurv; sb += G) {               // G surviving corners x NTYPES simplices over the lanes
        if (sb) __syncthreads();
        for (int wb = 0; wb < G * NTYPES; wb += kThreads) {
          const int wi = wb + tid;
          if (wi >= G * NTYPES) continue;
          const unsigned si = sb + (unsigned)(wi / NTYPES);
          const int type = wi % NTYPES;
          if (si >= nsurv) continue;
          const unsigned ent = s_surv[si], gi = ent & 0xffu, want = ent >> 8;
          if (!(fan.ordinal[type] ? (want & 1u) : (want & 2u))) continue;
          const unsigned tab = s_tab[type];
          unsigned char flags[N];
          for (int i = 0; i < N; i ++) flags[i] = s_flag[gi][(tab >> (8 * i)) & 0xffu];
          unsigned m_and = 0x3f, m_or = 0;
          for (int i = 0; i < N; i ++) { m_and &= flags[i]; m_or |= flags[i]; }
          if ((m_or & (kInvalid | kNonFinite)) || (m_and & 0x3f)) continue;      // (simplex_inside's own first test: before anything is fetched for it)
          const u64 cs = base + gi;
          const u64 step = cs / cells, lin = cs - step * cells;
          const Fields &f = s_fields[step];
          int corner[N];
          core_corner<ND>(m, lin, corner);
          corner[ND] = f.t;
          u64 X[N][ND];
          for (int i = 0; i < N; i ++) {
            const unsigned vm = (tab >> (8 * i)) & 0xffu;
            for (int c = 0; c < ND; c ++) X[i][c] = (u64)s_vf[gi][vm][c];
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
          const unsigned gi = item >> 6;
          const int type = (int)(item & 63u);
          const u64 cs = base + gi;
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
            for (int c = 0; c < ND; c ++) X[i][c] = (u64)s_vf[gi][vm][c];
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
      __syncthreads();
      if (tid == 0) {
        const unsigned k_now = s_nkeys < KCAP ? s_nkeys : KCAP, kb = k_before < KCAP ? k_before : KCAP;
        if (nown < kOneOwnBlocks) { s_bstart[nown] = kb; s_bcnt[nown] = k_now - kb; }
        __hip_atomic_store(&bcount[blk], k_now - kb, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
    __syncthreads();
    overflow = s_nkeys > KCAP || nown > kOneOwnBlocks;
    unsigned t_sum = tested;
    for (int o = 32; o > 0; o >>= 1) t_sum += __shfl_down(t_sum, o);
    if (lane == 0 && t_sum) atomicAdd(&s_tested, t_sum);
    __syncthreads();
  }
  if (tid == 0) {
    if (overflow) __hip_atomic_store(a.scratch + ONE_OVER, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (s_tested) atomicAdd((unsigned long long *)(a.scratch + ONE_TESTED), (unsigned long long)s_tested);
    if (s_cells) atomicAdd((unsigned long long *)(a.scratch + ONE_CELLS), (unsigned long long)s_cells);
  }
  ONE_STAMP(4);
  alive = alive && grid_barrier(bar + 1, nwg, bar + 3);
  ONE_STAMP(5);
  if (!alive && tid == 0) s_status |= (unsigned)SERIES_OVERFLOW;      // (given up: the host sweeps the steps the usual way)
  __syncthreads();

  // ---- 4: offsets, ranks, records ------------------------------------------------------------------------------------------------------
  if (alive) {
    // offsets: every workgroup scans ALL the blocks' counts (a contiguous share per lane, the lanes' sums scanned across the workgroup) and keeps
    // the exclusive prefix of its own blocks
    const unsigned per = (nblocks + kThreads - 1) / kThreads, lo = (unsigned)tid * per, hi = lo + per < nblocks ? lo + per : nblocks;
    unsigned sum = 0;
    for (unsigned b = lo; b < hi; b ++) { const unsigned cb = __hip_atomic_load(&bcount[b], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); s_bc[b] = cb; sum += cb; }
    unsigned incl = sum;
    for (int o = 1; o < 64; o <<= 1) { const unsigned up = __shfl_up(incl, o); if (lane >= o) incl += up; }
    if (lane == 63) s_wsum[wv] = incl;
    __syncthreads();
    unsigned before = 0;
    for (int q = 0; q < wv; q ++) before += s_wsum[q];
    s_lanepre[tid] = before + incl - sum;                        // the blocks before this lane's share
    if (tid == kThreads - 1) {
      s_total = (u64)(before + incl);
      if ((u64)(before + incl) > a.capacity || __hip_atomic_load(a.scratch + ONE_OVER, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicOr(&s_status, (unsigned)SERIES_OVERFLOW);
    }
    __syncthreads();
    for (unsigned i = tid; i < nown && i < kOneOwnBlocks; i += kThreads) {
      const unsigned b = w * ((nblocks + nwg - 1) / nwg) + i, ln = b / per;
      unsigned off = s_lanepre[ln];
      for (unsigned q = ln * per; q < b; q ++) off += s_bc[q];
      s_boff[i] = off;
    }
  }
  __syncthreads();
  const unsigned status = s_status;
  const unsigned nk = (status || overflow) ? 0u : s_nkeys;
  if (!status) {
    // a block's keys among themselves: rank = the number of smaller keys of the block (keys are unique), place = the block's offset + rank
    for (unsigned i = tid; i < nk; i += kThreads) {
      unsigned bi = 0;
      while (bi + 1 < nown && s_bstart[bi + 1] <= i) bi ++;
      const unsigned b0 = s_bstart[bi], bn = s_bcnt[bi];
      const u64 key = s_keys[i];
      unsigned r = 0;
      for (unsigned q = b0; q < b0 + bn; q ++) r += s_keys[q] < key ? 1u : 0u;
      s_sorted[b0 + r] = key;
      s_pos[b0 + r] = s_boff[bi] + r;
    }
    __syncthreads();
    const unsigned nk_pad = (nk + 63u) / 64u * 64u;
    for (unsigned p = tid; p < nk_pad; p += kThreads) {           // (wave-uniform trip count: a wavefront's 64 positions are one contiguous run of the output)
      if (p < nk) {
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
        const u64 *src = reinterpret_cast<const u64 *>(&rec);
#pragma unroll
        for (int k = 0; k < 9; k ++) s_rec[wv][lane * 9 + k] = src[k];
        if (ND == 3 && fragile) {
          const u64 e = atomicAdd((unsigned long long *)(a.scratch + ONE_NFRAG), 1ull);
          if (e < a.fragile_capacity) {
            u64 *fd = a.fragile + e * 10;
            fd[0] = (u64)s_pos[p];
            for (int k = 0; k < 9; k ++) fd[1 + k] = (u64)__double_as_longlong(Jfrag[k]);
          }
        }
      }
      __builtin_amdgcn_wave_barrier();
      const unsigned p0 = p - (unsigned)lane;
      const unsigned nvalid = nk - p0 >= 64u ? 64u : (p0 < nk ? nk - p0 : 0u);
      // (consecutive positions of a block are consecutive records of the output: consecutive lanes write consecutive words wherever the
      // wavefront's 64 records do not cross into another block -- 512-byte pieces over PCIe, like series_record_kernel)
      for (unsigned w8 = (unsigned)lane; w8 < nvalid * 9u; w8 += 64u) {
        const unsigned ri = w8 / 9u, kk = w8 - ri * 9u;
        __hip_atomic_store(reinterpret_cast<u64 *>(a.out + s_pos[p0 + ri]) + kk, s_rec[wv][w8], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      }
      __builtin_amdgcn_wave_barrier();
    }
  }
  ONE_STAMP(6);
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
  if (tid == 4) { a.scratch[ONE_NFRAG] = 0ull; a.scratch[ONE_TESTED] = 0ull; a.scratch[ONE_CELLS] = 0ull; a.scratch[ONE_OVER] = 0ull; }
  __threadfence_system();
  __syncthreads();
  ONE_STAMP(7);
  if (tid == 0) __hip_atomic_store(a.flag, a.seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

void launch_series_one(const Mesh &m, const OneArgs &a, int nwg, hipStream_t st)
{
  if (m.nd == 2) hipLaunchKernelGGL(series_one_kernel<2>, dim3((unsigned)nwg), dim3(kThreads), 0, st, m, a);
  else hipLaunchKernelGGL(series_one_kernel<3>, dim3((unsigned)nwg), dim3(kThreads), 0, st, m, a);
}

}  // namespace ftkx
