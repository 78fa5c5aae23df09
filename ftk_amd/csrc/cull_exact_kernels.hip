// FAST PATH 2/3 and 3/3 (gfx950): the cull over the sign masks (summaries, then mask words), the exact integer test of the surviving
// corners, and the FP64 half that turns a simplex that passed into a record.  Reference: check_simplex, critical_point_tracker_2d_regular.hh:584-685,
// ..._3d_regular.hh:425-514.  (Split from sweep_kernels.hip in round 6.)
#include "internal.hpp"
#include "sweep_device.hpp"
#include "series_device.hpp"

namespace ftkx {

// ---------------------------------------------------------------------------------------------------------------
// FAST PATH 2/3: corner cull on the mask bytes, 8 corners per lane (SWAR), survivors -> work list
// ---------------------------------------------------------------------------------------------------------------
// list entry: bits 0..39 corner index inside core (x fastest), bits 40..41 scope flags (1 ordinal, 2 interval), bits 44.. step
__device__ inline u64 load_row_pair_and(const unsigned char *__restrict__ M, size_t row_off, int g)
{
  const u64 *w = reinterpret_cast<const u64 *>(M + row_off) + g;
  const u64 w0 = w[0], w1 = w[1];                  // the pitch has 8 spare bytes: w[1] always exists
  return w0 & ((w0 >> 8) | (w1 << 56));            // byte b = mask(x = 8g + b) & mask(x + 1)
}

// Marching form of the cull: a lane keeps, for its 8 corners and a short run of z planes, the AND over each slice's 2^d
// spatial cube vertices in registers and walks through the consecutive timesteps of the batch, so that every mask byte is
// read from HBM once per batch instead of once per (step, role).  The x+1 word of a lane is its upper neighbour's word (DPP).
__device__ inline u64 dpp_u64_from_upper_lane(u64 v)
{
  int lo = (int)v, hi = (int)(v >> 32);
  lo = __builtin_amdgcn_update_dpp(lo, lo, 0x130 /* wave_shl:1 */, 0xf, 0xf, false);
  hi = __builtin_amdgcn_update_dpp(hi, hi, 0x130, 0xf, 0xf, false);
  return ((u64)(unsigned)hi << 32) | (u64)(unsigned)lo;
}

// COARSE = true runs the very same cull one level up: the "mask array" is the per-word summary U (one byte = 8 vertices), a
// "corner" is an aligned group of 8 corners, and what survives is appended to the refine list instead of the final list.
template <int ND, int ZC, bool COARSE>
__global__ __launch_bounds__(kThreads) void cull_march_kernel(const Mesh m, const Fields *__restrict__ steps, int nsteps, int step_chunk,
                                                              int gx_log2, u64 *__restrict__ list, u64 list_capacity, const FactorJob fj)
{
  if (fj.enabled && blockIdx.z == gridDim.z - 1) {             // the extra layer of the grid: one of its workgroups forms the factors
    if (blockIdx.x == 0 && blockIdx.y == 0)
      series_factors_body<kThreads, kFoldMaxSlices>(fj.steps, fj.nsteps, fj.slices, fj.nslices, fj.sinfo, fj.red, fj.running_in, fj.running_from, fj.safe_m, fj.results, fj.counters);
    return;
  }
  constexpr int kListCounter = COARSE ? CNT_REFINE_LIST : CNT_SURVIVOR_LIST;
  __shared__ unsigned s_wave_total[4];
  __shared__ u64 s_block_base;
  // 2D: the survivors of a workgroup's steps are parked in LDS and appended with ONE atomic at the end (or when the buffer could
  // overflow): on hit-dense data every workgroup has survivors in every step, and the list counter is a single address -- 4 096
  // returning atomics on it were half of this kernel's 40 us on 64 steps of 1024^2
  constexpr unsigned STAGE_CAP = (ND == 2) ? 2048u : 1u;             // (a step's worst case: 256 lanes x 8 corners)
  __shared__ u64 s_stage[STAGE_CAP];
  __shared__ unsigned s_staged, s_run;
  if (ND == 2) { if (threadIdx.x == 0) s_staged = 0; __syncthreads(); }
  auto flush_stage = [&]() {                                   // called by the whole workgroup, after a barrier that made s_staged final
    const unsigned n = s_staged;
    if (threadIdx.x == 0) s_block_base = n ? atomicAdd(&m.counters[kListCounter], (u64)n) : 0ull;
    __syncthreads();
    const u64 base = s_block_base;
    for (unsigned h = threadIdx.x; h < n; h += kThreads) if (base + h < list_capacity) list[base + h] = s_stage[h];
    __syncthreads();
    if (threadIdx.x == 0) s_staged = 0;
    __syncthreads();
  };
  constexpr u64 kAll = 0x3f3f3f3f3f3f3f3full, k7f = 0x7f7f7f7f7f7f7f7full, k80 = 0x8080808080808080ull;
  const int DW = m.ext_sz[0], DH = m.ext_sz[1], DD = (ND == 3) ? m.ext_sz[2] : 1, P = m.mask_pitch;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  // a wavefront covers GX 8-corner groups along x times 64/GX rows (GX = 64 for rows of 512+ vertices)
  const int GX = 1 << gx_log2, rows_per_wave = 64 >> gx_log2;
  const int gl = lane & (GX - 1);
  const int g = blockIdx.x * GX + gl;
  const int j = (blockIdx.y * 4 + wv) * rows_per_wave + (lane >> gx_log2);
  const int nzc = (ND == 3) ? (DD + ZC - 1) / ZC : 1;
  const int z0 = (ND == 3) ? (int)(blockIdx.z % nzc) * ZC : 0;
  const int s0 = (int)(blockIdx.z / nzc) * step_chunk;
  const int s1 = s0 + step_chunk < nsteps ? s0 + step_chunk : nsteps;
  const int ngroups = (DW + 7) / 8;
  const bool g_ok = g < ngroups;
  const int gc = g_ok ? g : ngroups - 1;                       // clamped: every lane issues valid loads
  const int jc = j < DH ? j : DH - 1;
  const int cy = j + m.ext_st[1];
  const bool row_ok = j < DH && cy >= m.core_st[1] && cy < m.core_st[1] + m.core_sz[1];
  u64 in_core = 0;
  if (g_ok && row_ok)
    for (int b = 0; b < 8; b ++) {
      const int cx = g * 8 + b + m.ext_st[0];
      if (cx >= m.core_st[0] && cx < m.core_st[0] + m.core_sz[0]) in_core |= 0x80ull << (8 * b);
    }
  unsigned zmask = 0;                                          // planes of the chunk whose corners are in core
  for (int zi = 0; zi < ZC; zi ++) {
    const int k = z0 + zi, cz = k + m.ext_st[2];
    if (k < DD && (ND == 2 || (cz >= m.core_st[2] && cz < m.core_st[2] + m.core_sz[2]))) zmask |= 1u << zi;
  }
  if (zmask == 0) return;                                      // wave-uniform
  const bool have_row1 = jc + 1 < DH;
  const bool seg_end = gl == GX - 1;                           // the x+1 word is not in the next lane

  // raw words of one slice for this lane: planes z0 .. z0+ZC (ND == 3) x rows (y, y+1) x (own word, x+1 word).
  // The lane at the end of an x segment reads its x+1 word from memory (the pitch has 8 spare bytes: always addressable);
  // the other lanes take it from the next lane by DPP, their second load just re-reads their own word from L1.
  constexpr int NP = (ND == 3) ? ZC + 1 : 1;
  struct Raw { u64 a0[NP], b0[NP], an[NP], bn[NP]; };
  const int gn = seg_end ? gc + 1 : gc;
  auto load_raw = [&](const unsigned char *__restrict__ M, Raw &r) {
    for (int p = 0; p < NP; p ++) {
      const int k = z0 + p < DD ? z0 + p : DD - 1;             // clamped: a plane beyond the array is replaced by neutral below
      const size_t off0 = (size_t)P * ((size_t)jc + (size_t)DH * (size_t)k);
      const u64 *row0 = reinterpret_cast<const u64 *>(M + off0);
      const u64 *row1 = reinterpret_cast<const u64 *>(M + off0 + (have_row1 ? (size_t)P : 0));
      r.a0[p] = row0[gc]; r.b0[p] = row1[gc]; r.an[p] = row0[gn]; r.bn[p] = row1[gn];
    }
  };
  // AND over the 2^d spatial cube vertices, per plane pair
  auto combine = [&](const Raw &r, u64 cube[ZC]) {
    u64 pl[NP];
    for (int p = 0; p < NP; p ++) {
      // DPP reads need every source lane active: shift first, under the full exec mask, select afterwards
      const u64 da = dpp_u64_from_upper_lane(r.a0[p]), db = dpp_u64_from_upper_lane(r.b0[p]);
      const u64 a1 = seg_end ? r.an[p] : da, b1 = seg_end ? r.bn[p] : db;
      const u64 v = (r.a0[p] & ((r.a0[p] >> 8) | (a1 << 56))) & (r.b0[p] & ((r.b0[p] >> 8) | (b1 << 56)));
      pl[p] = (z0 + p < DD) ? v : kAll;
    }
    for (int zi = 0; zi < ZC; zi ++) cube[zi] = (ND == 3) ? (pl[zi] & pl[zi + 1]) : pl[0];
  };

  const u64 row_lin = (u64)(cy - m.core_st[1]) * (u64)m.core_sz[0];
  const u64 plane_sz = (u64)m.core_sz[0] * (u64)m.core_sz[1];
  u64 cur[ZC], nxt[ZC];
  const unsigned char *have_cur = nullptr, *pending_ptr = nullptr;
  Raw pending;                                                 // software prefetch: the slice the NEXT step will need first
  auto fetch = [&](const unsigned char *ptr, u64 cube[ZC]) {
    if (ptr == pending_ptr) combine(pending, cube);
    else { Raw r; load_raw(ptr, r); combine(r, cube); }
  };
#pragma unroll 1
  for (int s = s0; s < s1; s ++) {
    const Fields f = steps[s];
    const bool need_next = (f.scope_mask & FTKX_SCOPE_INTERVAL) != 0;
    const unsigned char *fm0 = COARSE ? f.U[0] : f.M[0], *fm1 = COARSE ? f.U[1] : f.M[1];
    if (have_cur != fm0) fetch(fm0, cur);                      // otherwise slice t is last step's slice t+1: already in registers
    if (need_next) fetch(fm1, nxt);
    have_cur = need_next ? fm1 : fm0;
    pending_ptr = nullptr;
    if (s + 1 < s1) {                                          // issue the next step's loads now; they land while this step is scanned
      const Fields g = steps[s + 1];
      const unsigned char *gm0 = COARSE ? g.U[0] : g.M[0], *gm1 = COARSE ? g.U[1] : g.M[1];
      const unsigned char *want = (gm0 != have_cur) ? gm0 : ((g.scope_mask & FTKX_SCOPE_INTERVAL) ? gm1 : nullptr);
      if (want) { load_raw(want, pending); pending_ptr = want; }
    }
    for (int zi = 0; zi < ZC; zi ++) {
      u64 surv_o = 0, surv_i = 0;
      if ((zmask >> zi) & 1) {
        // bytes are <= 0x3f: adding 0x7f sets bit 7 exactly in the non-zero bytes, without carries between bytes
        if (f.scope_mask & FTKX_SCOPE_ORDINAL) surv_o = ~(cur[zi] + k7f) & k80 & in_core;
        if (need_next) surv_i = ~((cur[zi] & nxt[zi]) + k7f) & k80 & in_core;
      }
      const u64 any = surv_o | surv_i;
      u64 pos;
      unsigned cnt;
      if constexpr (ND == 2) {
        // hit-dense 2D data: most wavefronts have survivors.  Wave totals through LDS give every lane its place in the workgroup's
        // staging buffer; the list counter is touched once per workgroup (flush_stage).
        if (__syncthreads_or(any != 0) == 0) continue;         // (block-uniform: no wavefront left the kernel, see the early exits above)
        cnt = (unsigned)__popcll(any);
        unsigned incl = cnt;
        for (int o = 1; o < 64; o <<= 1) { const unsigned up = __shfl_up(incl, o); if (lane >= o) incl += up; }
        if (lane == 63) s_wave_total[wv] = incl;
        __syncthreads();
        if (threadIdx.x == 0) {
          unsigned run = 0;
          for (int q = 0; q < 4; q ++) { const unsigned t = s_wave_total[q]; s_wave_total[q] = run; run += t; }
          s_run = run;
        }
        __syncthreads();
        if (s_staged + s_run > STAGE_CAP) flush_stage();       // (block-uniform)
        unsigned at = s_staged + s_wave_total[wv] + (incl - cnt);
        if (cnt) {
          const u64 lin0 = row_lin;
          for (int b = 0; b < 8; b ++) {
            const unsigned fl = (unsigned)((surv_o >> (8 * b + 7)) & 1) | ((unsigned)((surv_i >> (8 * b + 7)) & 1) << 1);
            if (!fl) continue;
            s_stage[at ++] = (lin0 + (u64)(g * 8 + b + m.ext_st[0] - m.core_st[0])) | ((u64)fl << 40) | ((u64)s << 44);
          }
        }
        __syncthreads();
        if (threadIdx.x == 0) s_staged += s_run;
        continue;                                              // (the staged entries go out at the end of the kernel)
      } else {
        if (__ballot(any != 0) == 0) continue;                 // the common case: nothing survives in this wavefront
        cnt = (unsigned)__popcll(any);
        unsigned incl = cnt;
        for (int o = 1; o < 64; o <<= 1) { const unsigned up = __shfl_up(incl, o); if (lane >= o) incl += up; }
        const unsigned total = __shfl(incl, 63);
        u64 base = 0;
        if (lane == 63) base = atomicAdd(&m.counters[kListCounter], (u64)total);
        base = __shfl(base, 63);
        pos = base + (incl - cnt);
      }
      if (cnt) {
        const u64 lin0 = row_lin + (ND == 3 ? (u64)(z0 + zi + m.ext_st[2] - m.core_st[2]) * plane_sz : 0ull);
        for (int b = 0; b < 8; b ++) {
          const unsigned fl = (unsigned)((surv_o >> (8 * b + 7)) & 1) | ((unsigned)((surv_i >> (8 * b + 7)) & 1) << 1);
          if (!fl) continue;
          const u64 lin = lin0 + (u64)(g * 8 + b + m.ext_st[0] - m.core_st[0]);
          if (pos < list_capacity) list[pos] = lin | ((u64)fl << 40) | ((u64)s << 44);
          pos ++;
        }
      }
    }
    if (need_next) for (int zi = 0; zi < ZC; zi ++) cur[zi] = nxt[zi];
  }
  if constexpr (ND == 2) { __syncthreads(); if (s_staged) flush_stage(); }   // (block-uniform)
}

// Second level of the two-level cull: one lane per refine-list entry (an aligned word of 8 corners whose summaries could not
// rule it out) repeats the test on the vertex mask bytes and appends the corners that still survive to the work list.
// `mc` is the coarse view the first level ran on (its core / ext describe words), `m` the real mesh.
template <int ND>
__global__ __launch_bounds__(kThreads) void refine_kernel(const Mesh m, const Mesh mc, const Fields *__restrict__ steps,
                                                          const u64 *__restrict__ refine, u64 refine_capacity,
                                                          u64 *__restrict__ list, u64 list_capacity)
{
  const int DH = m.ext_sz[1], DD = (ND == 3) ? m.ext_sz[2] : 1, P = m.mask_pitch;
  const int lane = threadIdx.x & 63;
  __shared__ unsigned s_wave_total[kThreads / 64];
  __shared__ u64 s_block_base;
  if (m.counters[CNT_SERIES_DONE]) return;              // (series pass: finished early; block-uniform)
  u64 count = m.counters[CNT_REFINE_LIST];
  if (blockIdx.x == 0 && threadIdx.x == 0) { atomicMax(&m.counters[CNT_REFINE_PEAK], count); atomicAdd(&m.counters[CNT_WORDS_REFINED], count); }
  if (count > refine_capacity) count = refine_capacity;
  const u64 k7f = 0x7f7f7f7f7f7f7f7full, k80 = 0x8080808080808080ull;
  // an entry of the refine list is a coarse cell: 8 corners along x times u_rows rows -- one lane per row of it
  const u64 UR = (u64)m.u_rows, urows = (u64)((DH + m.u_rows - 1) / m.u_rows);
  const u64 work = count * UR;
  for (u64 base = (u64)blockIdx.x * kThreads; base < work; base += (u64)gridDim.x * kThreads) {   // block-uniform trip count
    const u64 idx = base + threadIdx.x;
    u64 surv_o = 0, surv_i = 0, row_lin = 0;
    int g = 0, step = 0;
    bool mine = idx < work;
    int j = 0, k = 0, cy = 0, cz = 0;
    unsigned want = 0;
    if (mine) {
      const u64 e = refine[idx / UR];
      step = (int)(e >> 44);
      want = (unsigned)((e >> 40) & 3);
      u64 lin = e & 0xffffffffffull;
      g = mc.core_st[0] + (int)(lin % (u64)mc.core_sz[0]); lin /= (u64)mc.core_sz[0];
      const int cyc = mc.core_st[1] + (int)(lin % (u64)mc.core_sz[1]); lin /= (u64)mc.core_sz[1];   // coarse row, relative to the array
      cz = (ND == 3) ? mc.core_st[2] + (int)lin : 0;
      j = cyc * m.u_rows + (int)(idx % UR); k = cz - m.ext_st[2];
      cy = j + m.ext_st[1];
      mine = cy >= m.core_st[1] && cy < m.core_st[1] + m.core_sz[1];      // (a block at the edge of the core: not all of its rows are corners)
    }
    if (mine) {
      const Fields f = steps[step];
      const bool need_next = (f.scope_mask & FTKX_SCOPE_INTERVAL) != 0 && (want & 2);
      u64 a0 = ~0ull, a1 = ~0ull;
      // words whose summary is non-zero were not written to M (mask_march2_kernel): their summary, replicated, stands in
      auto row_pair_and = [&](const unsigned char *__restrict__ Mp, const unsigned char *__restrict__ Up, int jj, int kk) -> u64 {
        const size_t row = (size_t)jj + (size_t)DH * (size_t)kk;
        const unsigned char *u = Up + (size_t)m.u_pitch * ((size_t)(jj / m.u_rows) + (size_t)urows * (size_t)kk) + g;
        const u64 *w = reinterpret_cast<const u64 *>(Mp + (size_t)P * row) + g;
        const unsigned u0 = u[0], u1 = u[1];                       // the summary pitch has spare bytes too
        const u64 w0 = u0 ? (u64)u0 * 0x0101010101010101ull : w[0];
        const u64 w1 = u1 ? (u64)u1 * 0x0101010101010101ull : w[1];
        return w0 & ((w0 >> 8) | (w1 << 56));
      };
      for (int dz = 0; dz < (ND == 3 ? 2 : 1); dz ++)
        for (int dy = 0; dy < 2; dy ++) {
          if (j + dy >= DH || k + dz >= DD) continue;                   // row outside the array: invalid vertices, neutral
          a0 &= row_pair_and(f.M[0], f.U[0], j + dy, k + dz);
          if (need_next) a1 &= row_pair_and(f.M[1], f.U[1], j + dy, k + dz);
        }
      u64 in_core = 0;
      for (int b = 0; b < 8; b ++) {
        const int cx = g * 8 + b + m.ext_st[0];
        if (cx >= m.core_st[0] && cx < m.core_st[0] + m.core_sz[0]) in_core |= 0x80ull << (8 * b);
      }
      if ((f.scope_mask & FTKX_SCOPE_ORDINAL) && (want & 1)) surv_o = ~(a0 + k7f) & k80 & in_core;
      if (need_next) surv_i = ~((a0 & a1) + k7f) & k80 & in_core;
      row_lin = (u64)(cy - m.core_st[1]) * (u64)m.core_sz[0] + (ND == 3 ? (u64)(cz - m.core_st[2]) * (u64)m.core_sz[0] * (u64)m.core_sz[1] : 0ull);
    }
    const u64 any = surv_o | surv_i;
    // one list atomic per workgroup and iteration (wave totals through LDS): on hit-dense data nearly every wavefront has survivors
    if (__syncthreads_or(any != 0) == 0) continue;             // block-uniform trip count, see the loop header
    const unsigned cnt = (unsigned)__popcll(any);
    unsigned incl = cnt;
    for (int o = 1; o < 64; o <<= 1) { const unsigned up = __shfl_up(incl, o); if (lane >= o) incl += up; }
    if (lane == 63) s_wave_total[threadIdx.x >> 6] = incl;
    __syncthreads();
    if (threadIdx.x == 0) {
      unsigned run = 0;
      for (int q = 0; q < kThreads / 64; q ++) { const unsigned t = s_wave_total[q]; s_wave_total[q] = run; run += t; }
      s_block_base = run ? atomicAdd(&m.counters[CNT_SURVIVOR_LIST], (u64)run) : 0ull;
    }
    __syncthreads();
    u64 pos = s_block_base + s_wave_total[threadIdx.x >> 6] + (incl - cnt);
    for (int b = 0; b < 8 && cnt; b ++) {
      const unsigned fl = (unsigned)((surv_o >> (8 * b + 7)) & 1) | ((unsigned)((surv_i >> (8 * b + 7)) & 1) << 1);
      if (!fl) continue;
      const u64 lin = row_lin + (u64)(g * 8 + b + m.ext_st[0] - m.core_st[0]);
      if (pos < list_capacity) list[pos] = lin | ((u64)fl << 40) | ((u64)step << 44);
      pos ++;
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------
// FAST PATH 3/3: exact test of the surviving corners
// ---------------------------------------------------------------------------------------------------------------
template <int ND>
__global__ __launch_bounds__(kThreads) void exact_kernel(const Mesh m, const Fields *__restrict__ steps, int step_base, const u64 *__restrict__ list, u64 list_capacity)
{
  constexpr int N = ND + 1;
  constexpr int NVC = 1 << N;                 // vertices of a corner's space-time hypercube
  constexpr int G = kThreads / NVC;           // corners per chunk: one lane per hypercube vertex while staging
  constexpr int NTYPES = fan_table<N>::NTYPES;
  constexpr int SUB = 4;                      // rounds of G corners staged together
  __shared__ i64 s_vf[SUB * G][NVC][ND];
  __shared__ unsigned char s_flag[SUB * G][NVC];
  __shared__ u64 s_entry[SUB * G];
  __shared__ unsigned s_tab[NTYPES];
  // descriptors of the (corner, type) pairs that passed the predicate, parked in LDS across chunks: the counter behind m.pass is ONE
  // address for the whole device (a same-address atomic costs ~5 ns of serialised L2 time: one per chunk was a quarter of this
  // kernel on hit-dense 2D data), so a workgroup takes a range of it only when its buffer could overflow, and once at the end
  constexpr unsigned OUT_CAP = 2048;
  static_assert(G * NTYPES <= OUT_CAP / 2, "a chunk's worst case must fit twice");
  __shared__ u64 s_out[OUT_CAP];
  __shared__ unsigned s_nout, s_tested;
  __shared__ u64 s_base;

  const int tid = threadIdx.x;
  const fan_table<N> &fan = dev_fan<ND>();
  if (m.counters[CNT_SERIES_DONE]) return;              // (series pass: the single-workgroup tail has finished this pass already)
  u64 count = m.counters[CNT_SURVIVOR_LIST];
  if (blockIdx.x == 0 && tid == 0) {   // the host checks the peak against the capacity; the statistic: cells that survived the cull
    atomicMax(&m.counters[CNT_LIST_PEAK], count);
    atomicAdd(&m.counters[CNT_CELLS_SURVIVED], count);
  }
  if (count > list_capacity) count = list_capacity;     // overflow: the host grows the list and replays the batch
  if ((u64)blockIdx.x * (SUB * G) >= count) return;     // nothing for this workgroup: leave before touching LDS or scratch
  if (tid < NTYPES) {
    unsigned w = 0;
    for (int i = 0; i < N; i ++) w |= (unsigned)fan.vert[tid][i] << (8 * i);
    s_tab[tid] = w;
  }
  if (tid == 0) { s_nout = 0; s_tested = 0; }
  unsigned tested = 0;

  auto flush = [&]() {                                  // called by the whole workgroup, after a barrier that made s_nout final
    const unsigned n = s_nout;
    if (tid == 0) s_base = atomicAdd(&m.counters[CNT_PASS], (u64)n);
    __syncthreads();
    const u64 base = s_base;
    for (unsigned h = tid; h < n; h += kThreads)
      if (base + h < m.capacity) {
        m.pass[base + h] = s_out[h];
        // series pass: how many simplices passed per bucket of the order key (the records are put in order without a sort, series.hip)
        if (m.hist) atomicAdd(&m.hist[order_key(s_out[h], m.core_cells) >> m.hist_shift], 1u);
      }
    __syncthreads();
    if (tid == 0) s_nout = 0;
  };

  // Staging is a chain of dependent memory round trips (list entry, the step's descriptor, the field values) with barriers in between:
  // SUB x G corners are fetched per chain instead of G (woven 1024^2 x 64: 5.5 chains per workgroup -> 1.4).  The test itself goes G
  // corners at a time, so that s_out can be emptied in between.
  bool narrow = false;
  for (u64 chunk = blockIdx.x; chunk * (SUB * G) < count; chunk += gridDim.x) {
    __syncthreads();                                    // previous chunk's LDS readers are done
    if (tid < SUB * G) s_entry[tid] = (chunk * (SUB * G) + tid < count) ? list[chunk * (SUB * G) + tid] : ~0ull;
    __syncthreads();
    {
      const int vtx = tid % NVC, sl = (vtx >> ND) & 1;
      u64 ent[SUB];
      const double *pS[SUB], *pV[SUB];
      double factor[SUB];
      bool live[SUB];
#pragma unroll
      for (int r = 0; r < SUB; r ++) {                  // descriptors
        ent[r] = s_entry[r * G + tid / NVC];
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
      for (int r = 0; r < SUB; r ++) {                  // field values: every load of the round in flight before the first is used
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
          } else if constexpr (ND == 2) {               // gradient2D (grad.hh:17-28): clamped indices
            const int ip = clampi(i + 1, 0, DW - 1), im = clampi(i - 1, 0, DW - 1), jp = clampi(j + 1, 0, DH - 1), jm = clampi(j - 1, 0, DH - 1);
            const int ic = clampi(i, 0, DW - 1), jc = clampi(j, 0, DH - 1);
            raw[r][0] = pS[r][(size_t)ip + (size_t)DW * jc]; raw[r][1] = pS[r][(size_t)im + (size_t)DW * jc];
            raw[r][2] = pS[r][(size_t)ic + (size_t)DW * jp]; raw[r][3] = pS[r][(size_t)ic + (size_t)DW * jm];
          } else {                                      // gradient3D (grad.hh:138-146): interior vertices only
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
      for (int r = 0; r < SUB; r ++) {                  // the same operations as vector_at / gradient_at on the same values, then classify_vertex's
        const int gi = r * G + tid / NVC;
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
      narrow = __syncthreads_and(mine_narrow) != 0;       // (the barrier between staging and testing, with the chunk's "fits in 32 bits" on it)
    }
    // (corner, type) pairs over all lanes; the few that pass go to record_kernel, whose expensive FP64 record construction then
    // runs on densely packed lanes instead of one or two lanes per wavefront
    for (int sub = 0; sub < SUB && (chunk * SUB + (u64)sub) * G < count; sub ++) {
      if (sub) __syncthreads();
      if (s_nout > OUT_CAP - G * NTYPES) flush();       // (workgroup-uniform: s_nout was final at the barrier above)
      for (int base = 0; base < G * NTYPES; base += kThreads) {
        const int w = base + tid;
        if (w < G * NTYPES) {
          const int gi = sub * G + w / NTYPES, type = w % NTYPES;
          const u64 e = s_entry[gi];
          const unsigned scope_flags = (e == ~0ull) ? 0u : (unsigned)((e >> 40) & 3);
          const bool wanted = fan.ordinal[type] ? (scope_flags & 1) : (scope_flags & 2);
          if (wanted) {
            const Fields &f = steps[e >> 44];
            int corner[N];
            core_corner<ND>(m, e & 0xffffffffffull, corner);
            corner[ND] = f.t;
            const unsigned tab = s_tab[type];
            unsigned char flags[N];
            u64 X[N][ND];
            for (int i = 0; i < N; i ++) {
              const unsigned vm = (tab >> (8 * i)) & 0xffu;   // the axis bitmask IS the hypercube vertex index
              flags[i] = s_flag[gi][vm];
              for (int c = 0; c < ND; c ++) X[i][c] = (u64)s_vf[gi][vm][c];
            }
            int ids[N]; double mu[N]; bool presolved;
            if (simplex_inside<ND>(m, f, 1, corner, tab, flags, X, tested, ids, mu, &presolved, narrow))
              s_out[atomicAdd(&s_nout, 1u)] = (e & kPassLinMask) | ((u64)type << kPassTypeShift) | ((u64)(step_base + (int)(e >> 44)) << kPassStepShift);
          }
        }
      }
    }
  }
  __syncthreads();
  if (s_nout) flush();
  {
    unsigned t_sum = tested;
    for (int o = 32; o > 0; o >>= 1) t_sum += __shfl_down(t_sum, o);
    if ((tid & 63) == 0 && t_sum) atomicAdd(&s_tested, t_sum);
    __syncthreads();
    if (tid == 0 && s_tested) atomicAdd(&m.counters[CNT_SIMPLICES_TESTED], (u64)s_tested);
  }
}

// ---------------------------------------------------------------------------------------------------------------
// The FP64 half of the sweep: one lane per simplex that passed the integer test (CNT_PASS descriptors written by exact_kernel /
// tile_kernel).  Re-quantises the simplex's d+1 vertices (a handful of loads), then inverse interpolation, lerps, Jacobian and
// classification exactly as check_simplex does after its test (2d:624-684, 3d:468-512), and the ballot-compacted append.
// Keeping this out of the integer kernels takes their scratch from 800-944 bytes per lane to none.
// ---------------------------------------------------------------------------------------------------------------
template <int ND>
__global__ __launch_bounds__(kThreads) void record_kernel(const Mesh m, const Fields *__restrict__ fields)
{
  constexpr int N = ND + 1;
  const fan_table<N> &fan = dev_fan<ND>();
  u64 count = m.counters[CNT_PASS];
  if (count > m.capacity) count = m.capacity;               // overflow: the host grows the buffers and replays the batch
  const u64 padded = (count + 63) / 64 * 64;                // wave-uniform trip count: emit_hits ballots
  for (u64 i = (u64)blockIdx.x * kThreads + threadIdx.x; i < padded; i += (u64)gridDim.x * kThreads) {
    bool hit = false, fragile = false;
    double Jfrag[9];
    ftkx_cp_t rec;
    if (i < count) {
      const u64 d = m.pass[i];
      const Fields &f = fields[d >> kPassStepShift];
      const int type = (int)((d >> kPassTypeShift) & 63u);
      u64 lin = d & kPassLinMask;
      int corner[N];
      core_corner<ND>(m, lin, corner);
      corner[ND] = f.t;
      u64 X[N][ND];
      int ids[N];
      // (the quantised vectors and SoS ids feed only the 2D degree computation: nobody else pays for re-deriving them)
      if (ND == 2 && m.compute_degrees)
      for (int v = 0; v < N; v ++) {
        const unsigned vm = fan.vert[type][v];
        int vx[3] = {0, 0, 0};
        for (int a = 0; a < ND; a ++) vx[a] = corner[a] + (int)((vm >> a) & 1u);
        const int sl = (int)((vm >> ND) & 1u);
        i64 q[ND];
        classify_vertex<ND>(m, f.S[sl], f.V[sl], f.factor, vx, q);
        for (int c = 0; c < ND; c ++) X[v][c] = (u64)q[c];
        ids[v] = vertex_id<ND>(m, corner, vm);
      }
      // (per lane: records next to the array border, given J, vector input, degrees take the general path)
      hit = record_is_fast<ND>(m, f, corner) ? make_record_impl<ND, true>(m, f, corner, type, X, ids, false, nullptr, &rec, &fragile, Jfrag)
                                             : make_record_general<ND>(m, f, corner, type, X, ids, false, nullptr, &rec, &fragile, Jfrag);
    }
    const u64 slot = emit_hits(m, hit, rec);
    if (ND == 3 && hit && fragile && slot != ~0ull) {          // (rare) handed to the host for classification with ITS libm
      const u64 e = atomicAdd(&m.counters[CNT_FRAGILE], 1ull);
      if (e < m.fragile_capacity) {
        u64 *dst = m.fragile + e * 10;
        dst[0] = slot;
        for (int q = 0; q < 9; q ++) dst[1 + q] = (u64)__double_as_longlong(Jfrag[q]);
      }
    }
  }
}

template <bool COARSE>
static void launch_cull_level(const Mesh &m, const Fields *d_steps, int nsteps, u64 *d_list, u64 cap, hipStream_t stream, const FactorJob *job = nullptr)
{
  FactorJob fj = FactorJob();
  if (job) fj = *job;
  int ZC = m.nd == 3 ? 4 : 1;
  const int groups = (m.ext_sz[0] + 7) / 8;
  int gx_log2 = 3;
  while (gx_log2 < 6 && (1 << gx_log2) < groups) gx_log2 ++;
  const int GX = 1 << gx_log2, rows_per_block = 4 * (64 >> gx_log2);
  const int nzc = m.nd == 3 ? (m.ext_sz[2] + ZC - 1) / ZC : 1;
  // steps per lane: consecutive steps reuse the shared slice from registers; more chunks = more parallelism
  // (3D: 4 -- a chunk re-reads one slice, a quarter more bytes of arrays that are 1/256 of the input, and gives four times the
  // wavefronts: the coarse cull of 256^3 x 16 0.052 -> 0.026 ms, of 512^3 x 32 0.089 -> 0.081 ms)
  int step_chunk = 4;   // (2D: the survivors of a workgroup's four steps leave with one atomic on the list counter)
  const int nsc = (nsteps + step_chunk - 1) / step_chunk;
  const dim3 grid((unsigned)((groups + GX - 1) / GX), (unsigned)((m.ext_sz[1] + rows_per_block - 1) / rows_per_block), (unsigned)(nzc * nsc) + (fj.enabled ? 1u : 0u));
#define FTKX_CULL_LAUNCH(ND_, ZC_) hipLaunchKernelGGL((cull_march_kernel<ND_, ZC_, COARSE>), grid, dim3(kThreads), 0, stream, m, d_steps, nsteps, step_chunk, gx_log2, d_list, cap, fj)
  if (m.nd == 2) FTKX_CULL_LAUNCH(2, 1);
  else if (ZC == 2) FTKX_CULL_LAUNCH(3, 2);
  else if (ZC == 8) FTKX_CULL_LAUNCH(3, 8);
  else FTKX_CULL_LAUNCH(3, 4);
#undef FTKX_CULL_LAUNCH
}

void launch_cull(const Mesh &m, const Fields *d_steps, int nsteps, u64 *d_list, u64 cap, hipStream_t stream, const FactorJob *job)
{
  if (nsteps <= 0) return;
  launch_cull_level<false>(m, d_steps, nsteps, d_list, cap, stream, job);
}

// two-level form: summaries first (1/8 of the bytes), vertex masks only for the words the summaries could not rule out
Mesh coarse_view(const Mesh &m)
{
  Mesh mc = m;                                   // the coarse view: one "vertex" per aligned word of 8 (x) and u_rows rows (y)
  const int w0 = (m.core_st[0] - m.ext_st[0]) / 8, w1 = (m.core_st[0] + m.core_sz[0] - 1 - m.ext_st[0]) / 8;
  mc.ext_st[0] = 0; mc.ext_sz[0] = (m.ext_sz[0] + 7) / 8;
  mc.core_st[0] = w0; mc.core_sz[0] = w1 - w0 + 1;
  const int r0 = (m.core_st[1] - m.ext_st[1]) / m.u_rows, r1 = (m.core_st[1] + m.core_sz[1] - 1 - m.ext_st[1]) / m.u_rows;
  mc.ext_st[1] = 0; mc.ext_sz[1] = (m.ext_sz[1] + m.u_rows - 1) / m.u_rows;
  mc.core_st[1] = r0; mc.core_sz[1] = r1 - r0 + 1;
  mc.mask_pitch = m.u_pitch;
  return mc;
}

// the two levels as separate launches (the series pass puts its factor kernel between them)
void launch_cull_coarse(const Mesh &m, const Fields *d_steps, int nsteps, u64 *d_refine, u64 refine_cap, hipStream_t stream, const FactorJob *job)
{
  if (nsteps <= 0) return;
  launch_cull_level<true>(coarse_view(m), d_steps, nsteps, d_refine, refine_cap, stream, job);
}

void launch_refine(const Mesh &m, const Fields *d_steps, const u64 *d_refine, u64 refine_cap, u64 *d_list, u64 cap, hipStream_t stream, int few_wgs)
{
  const Mesh mc = coarse_view(m);
  const dim3 grid(few_wgs > 0 ? (unsigned)few_wgs : 256u * 4u);      // (few: the tail of a split pass, next to a mask kernel -- sparse data, every workgroup waits for a slot)
  if (m.nd == 2) hipLaunchKernelGGL(refine_kernel<2>, grid, dim3(kThreads), 0, stream, m, mc, d_steps, d_refine, refine_cap, d_list, cap);
  else hipLaunchKernelGGL(refine_kernel<3>, grid, dim3(kThreads), 0, stream, m, mc, d_steps, d_refine, refine_cap, d_list, cap);
}

void launch_cull_two_level(const Mesh &m, const Fields *d_steps, int nsteps, u64 *d_refine, u64 refine_cap, u64 *d_list, u64 cap, hipStream_t stream)
{
  if (nsteps <= 0) return;
  launch_cull_coarse(m, d_steps, nsteps, d_refine, refine_cap, stream, nullptr);
  launch_refine(m, d_steps, d_refine, refine_cap, d_list, cap, stream, 0);
}

void launch_records(const Mesh &m, const Fields *d_fields, hipStream_t stream)
{
  // grid-stride over a device-side count: a few workgroups per CU are plenty (hits are rare; hit-dense 2D data: 1e4-1e5 per batch)
  const dim3 grid(256u * 2u);
  if (m.nd == 2) hipLaunchKernelGGL(record_kernel<2>, grid, dim3(kThreads), 0, stream, m, d_fields);
  else hipLaunchKernelGGL(record_kernel<3>, grid, dim3(kThreads), 0, stream, m, d_fields);
}

void launch_exact(const Mesh &m, const Fields *d_steps, int step_base, const u64 *d_list, u64 cap, hipStream_t stream, int few_wgs)
{
  // persistent-style: workgroups stride over the list, every wave exits when it is drained (no scratch; 21-23 KB of LDS).  Four per
  // CU: woven 1024^2 x 64 (181 853 cells) 0.084 ms with the record kernel, double_gyre 2048 x 1024 x 128 0.078 (0.097 with two)
  int per_cu = 4;
  const dim3 grid(few_wgs > 0 ? (unsigned)few_wgs : 256u * (unsigned)per_cu);
  if (m.nd == 2) hipLaunchKernelGGL(exact_kernel<2>, grid, dim3(kThreads), 0, stream, m, d_steps, step_base, d_list, cap);
  else hipLaunchKernelGGL(exact_kernel<3>, grid, dim3(kThreads), 0, stream, m, d_steps, step_base, d_list, cap);
}

}  // namespace ftkx
