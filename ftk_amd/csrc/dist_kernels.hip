// Kernels of the device-driven pass over a timestep SLAB (several ranks, one per GPU; host side: series.hip, ftkx_series_dist_*).
//
// What links the ranks of a t-slab partition is small: the sticky running minimum of update_vector_field_scaling_factor
// (include/ftk/filters/critical_point_tracker.hh:850-864) runs over the slices in time order, so rank r needs the minimum over the slabs
// before it and the reduction of the one slice it shares with rank r + 1; and its last interval sweep reads that slice -- as sign masks
// plus the input values around the cells that survive the cull (the compact halo, halo_kernels.hip).  Every one of these numbers is
// produced and consumed by a kernel here, so that the collectives between them (an all_gather of four doubles per rank, three
// neighbour messages) are queued on the stream by the caller and the host waits once per pass, as with one rank.
#include "sweep_device.hpp"
#include "series_device.hpp"

namespace ftkx {

// ---- what this rank contributes: {min, max} of its slab's fused reductions, and those of its first slice ----------------------------
// `slices`: the pass's slice table (series.hip); the first `nown` entries are this rank's own slices in time order.  One workgroup, a
// wavefront per slice at a time.  Also zeroes the per-pass dist counters.
__global__ __launch_bounds__(256) void dist_contrib_kernel(const SeriesSlice *__restrict__ slices, int nown, const u64 *__restrict__ red, u64 *__restrict__ contrib,
                                                           u64 *__restrict__ block)
{
  __shared__ u64 s_min[4], s_max[4], s_first[2];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  u64 wmin = 0x7fefffffffffffffull, wmax = 0ull;
  for (int j = wv; j < nown; j += 4) {
    const SeriesSlice sl = slices[j];
    u64 mn = 0x7fefffffffffffffull, mx = 0ull;
    if (sl.red_index >= 0) { const u64 *r = red + (size_t)sl.red_index * 128; mn = r[2 * lane]; mx = r[2 * lane + 1]; }
    for (int o = 32; o > 0; o >>= 1) { const u64 a = __shfl_xor(mn, o), b = __shfl_xor(mx, o); mn = a < mn ? a : mn; mx = b > mx ? b : mx; }
    // (bit patterns of non-negative doubles order like the values; what is known from before this pass counts too)
    u64 kr = (u64)__double_as_longlong(sl.known_res), kx = (u64)__double_as_longlong(sl.known_max);
    if (sl.from_res) { kr = *sl.from_res; kx = *sl.from_max; }
    mn = kr < mn ? kr : mn;                               // (DBL_MAX / 0 where nothing is known: neutral, as in series_factors_body)
    mx = kx > mx ? kx : mx;
    if (j == 0 && lane == 0) { s_first[0] = mn; s_first[1] = mx; }
    wmin = mn < wmin ? mn : wmin; wmax = mx > wmax ? mx : wmax;
  }
  if (lane == 0) { s_min[wv] = wmin; s_max[wv] = wmax; }
  __syncthreads();
  if (threadIdx.x == 0) {
    u64 mn = s_min[0], mx = s_max[0];
    for (int q = 1; q < 4; q ++) { mn = s_min[q] < mn ? s_min[q] : mn; mx = s_max[q] > mx ? s_max[q] : mx; }
    contrib[0] = mn; contrib[1] = mx;
    contrib[2] = nown > 0 ? s_first[0] : 0x7fefffffffffffffull; contrib[3] = nown > 0 ? s_first[1] : 0ull;
  }
  (void)block;
}

// ---- the first slice's sign masks as ONE message for the lower neighbour: header | summary array | word indices | words --------------
// One pass over the summary array, eight summary bytes per lane: the bytes are copied into the message as they are, and wherever one of
// them is 0 -- the only blocks whose mask words the mask kernel wrote -- the block's u_rows words go on the list (index of the 8-byte
// word in M, its 8 bytes; any order).  A wavefront reserves its run of the list with ONE atomic.  The workgroup that finishes last
// writes the header (the count lives on the device; the receiver reads it there).  block[DB_WORDS], block[DB_DONE] must be 0 on entry
// (the import kernel of the pass that used this block before leaves them so).
constexpr u64 kPackedMagicD = 0x66746b786d61736bull;      // "ftkxmask" (halo_kernels.hip: kPackedMagic)
__global__ __launch_bounds__(256) void dist_export_kernel(const Mesh m, const u64 *__restrict__ U, const unsigned char *__restrict__ M, u64 u_bytes, u64 *__restrict__ hdr,
                                                          unsigned *__restrict__ idx, u64 *__restrict__ words, u64 capacity, unsigned factor_log2, u64 *__restrict__ block)
{
  const int UP = m.u_pitch, P = m.mask_pitch, DH = m.ext_sz[1], UR = m.u_rows;
  const int ngroups = (m.ext_sz[0] + 7) / 8;
  const u64 urows = (u64)((DH + UR - 1) / UR);
  const u64 nw = u_bytes / 8;
  u64 *dst = hdr + 4;
  const int lane = threadIdx.x & 63;
  const u64 nw_pad = (nw + 63) / 64 * 64;
  for (u64 i = (u64)blockIdx.x * 256 + threadIdx.x; i < nw_pad; i += (u64)gridDim.x * 256) {
    u64 w = ~0ull;
    if (i < nw) { w = U[i]; dst[i] = w; }
    // zero bytes of w (exact: the classic (w - 0x01..) & ~w & 0x80.. test has no false positives when taken byte by byte from the low end;
    // evaluated per byte here, it only runs where the quick test fires)
    unsigned zero = 0;
    if ((w - 0x0101010101010101ull) & ~w & 0x8080808080808080ull)
      for (int b = 0; b < 8; b ++) if (((w >> (8 * b)) & 0xffull) == 0) zero |= 1u << b;
    // (bytes of a summary row beyond its last group are padding: skipped by position)
    unsigned nemit = 0;
    unsigned take = 0;
    for (int b = 0; b < 8; b ++) {
      if (!((zero >> b) & 1u)) continue;
      const u64 pos = i * 8 + (u64)b, urow = pos / (u64)UP;
      const int g = (int)(pos - urow * (u64)UP);
      if (g >= ngroups) continue;
      const u64 yb = urow % urows;
      int rows = DH - (int)(yb * (u64)UR); rows = rows < UR ? rows : UR;
      take |= 1u << b; nemit += (unsigned)rows;
    }
    // the wavefront's run of the list: exclusive prefix over the lanes, one atomic
    unsigned incl = nemit;
    for (int o = 1; o < 64; o <<= 1) { const unsigned up = __shfl_up(incl, o); if (lane >= o) incl += up; }
    const unsigned total = __shfl(incl, 63);
    if (!total) continue;
    u64 base = 0;
    if (lane == 63) base = atomicAdd(&block[DB_WORDS], (u64)total);
    base = __shfl(base, 63) + (u64)(incl - nemit);
    for (int b = 0; b < 8; b ++) {
      if (!((take >> b) & 1u)) continue;
      const u64 pos = i * 8 + (u64)b, urow = pos / (u64)UP;
      const int g = (int)(pos - urow * (u64)UP);
      const u64 z = urow / urows, yb = urow % urows;
      int rows = DH - (int)(yb * (u64)UR); rows = rows < UR ? rows : UR;
      for (int r = 0; r < rows; r ++) {
        const u64 row = z * (u64)DH + yb * (u64)UR + (u64)r;
        const u64 wi = (row * (u64)P) / 8 + (u64)g;
        if (base < capacity) { idx[base] = (unsigned)wi; words[base] = reinterpret_cast<const u64 *>(M)[wi]; }
        base ++;
      }
    }
  }
  // The last workgroup to finish writes the header.  No fence: what it needs of the others is their share of the COUNT, which is an
  // atomic like the arrival counter -- a wavefront has its reservation's return value (it computes addresses from it) before it reaches
  // the barrier below, so every reservation of a workgroup is performed before its arrival is.  (A __threadfence here is an L2
  // write-back per workgroup on this part: 47 us for the 512 of them.)  The list and the summary bytes only have to be there when the
  // kernel ends.
  __shared__ unsigned s_last;
  __syncthreads();
  if (threadIdx.x == 0) s_last = atomicAdd(&block[DB_DONE], 1ull) == (u64)gridDim.x - 1 ? 1u : 0u;
  __syncthreads();
  if (s_last && threadIdx.x == 0) {
    hdr[0] = __hip_atomic_load(&block[DB_WORDS], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    hdr[1] = u_bytes; hdr[2] = capacity | ((u64)UR << 48) | ((u64)factor_log2 << 56); hdr[3] = kPackedMagicD;
    block[DB_WORDS] = 0ull; block[DB_DONE] = 0ull;         // (as found)
  }
}

// ---- the halo's masks imported (header checked on the device: geometry, rows per summary byte, and the sender's factor must not exceed
// ours -- masks serve their own factor and larger ones), with the prefix job riding in workgroup 0: the running minimum BEFORE this rank's slab (min over the lower ranks' contributions) into the stub the
// factor job reads, and the gathered block into the results (for the host) ---------------------------------
__global__ __launch_bounds__(256) void dist_import_kernel(const u64 *__restrict__ gathered, int rank, int nranks, double running_in, u64 *__restrict__ block, u64 *__restrict__ results_tail,
                                                          const u64 *__restrict__ hdr /* nullptr: no halo */, const unsigned *__restrict__ idx, const u64 *__restrict__ words, u64 u_bytes,
                                                          u64 capacity, unsigned u_rows, unsigned max_factor_log2, u64 *__restrict__ U, unsigned char *__restrict__ M, u64 mask_words)
{
  if (blockIdx.x == 0 && threadIdx.x < 64) {
    const int lane = threadIdx.x;
    u64 mn = (u64)__double_as_longlong(running_in);
    for (int r = lane; r < rank; r += 64) { const u64 v = gathered[kDistContrib * r]; mn = v < mn ? v : mn; }
    for (int o = 32; o > 0; o >>= 1) { const u64 a = __shfl_xor(mn, o); mn = a < mn ? a : mn; }
    if (lane == 0) { block[DB_PSEUDO + SR_RUNNING] = mn; block[DB_BAD] = 0ull; }
    for (int i = lane; i < kDistContrib * nranks; i += 64) results_tail[i] = gathered[i];
  }
  if (!hdr) return;
  const u64 n = hdr[0], geo = hdr[2];
  if (hdr[1] != u_bytes || (geo & ((1ull << 48) - 1ull)) != capacity || ((geo >> 48) & 0xffull) != (u64)u_rows || (geo >> 56) > (u64)max_factor_log2 ||
      hdr[3] != kPackedMagicD || n > capacity) {
    __syncthreads();                                      // (workgroup 0: behind the reset of DB_BAD above)
    if (blockIdx.x == 0 && threadIdx.x == 0) block[DB_BAD] = 1ull;
    return;
  }
  const u64 *src = hdr + 4;
  const u64 nw = u_bytes / 8;
  for (u64 i = (u64)blockIdx.x * 256 + threadIdx.x; i < nw; i += (u64)gridDim.x * 256) U[i] = src[i];
  bool bad = false;
  for (u64 i = (u64)blockIdx.x * 256 + threadIdx.x; i < n; i += (u64)gridDim.x * 256) {
    const u64 w = idx[i];
    if (w < mask_words) reinterpret_cast<u64 *>(M)[w] = words[i];
    else bad = true;
  }
  // (an index outside the mask array: flagged -- by workgroups other than 0 too, whose store may race with workgroup 0's reset; the
  // request kernel reads DB_BAD two kernels later, and a lost flag needs a sender that breaks the protocol in the first place.  Keep it
  // exact anyway: such workgroups raise bit 1, which workgroup 0 never writes)
  if (bad) atomicOr((unsigned long long *)&block[DB_BAD2], 1ull);
}

// ---- the request: the cells listed (as sparse_cells_kernel does: survivors of the cull whose interval sweep reads the halo slice,
// a wavefront's run reserved with one atomic) and the request's header written by the workgroup that arrives last -- the count is an atomic,
// read back where atomics are performed; the cells only have to be there when the kernel ends ---------------------------------------------
__global__ __launch_bounds__(256) void dist_cells_kernel(const Mesh m, const Fields *__restrict__ steps, const u64 *__restrict__ list, u64 list_capacity, u64 refine_capacity,
                                                         const double *halo_field, u64 *__restrict__ request, u64 cap, u64 *__restrict__ block, u64 *__restrict__ results)
{
  u64 count = m.counters[CNT_SURVIVOR_LIST];
  if (count > list_capacity) count = list_capacity;
  u64 *cells = request + 1;
  const u64 padded = (count + 63) / 64 * 64;
  for (u64 i = (u64)blockIdx.x * 256 + threadIdx.x; i < padded; i += (u64)gridDim.x * 256) {
    bool take = false;
    u64 lin = 0;
    if (i < count) {
      const u64 e = list[i];
      const Fields &f = steps[e >> 44];
      lin = e & 0xffffffffffull;
      take = ((e >> 40) & 2) && (f.S[1] == halo_field || f.V[1] == halo_field);
    }
    const unsigned long long b = __ballot(take);
    if (!b) continue;
    const int lane = threadIdx.x & 63, leader = __ffsll((long long)b) - 1;
    u64 base = 0;
    if (lane == leader) base = atomicAdd(&m.counters[CNT_SPARSE], (u64)__popcll(b));
    base = __shfl(base, leader);
    if (take) { const u64 slot = base + (u64)__popcll(b & ((1ull << lane) - 1ull)); if (slot < cap) cells[slot] = lin; }
  }
  __shared__ unsigned s_last;
  __syncthreads();                                        // (every wavefront has its reservation's return value: the count is complete when all have arrived)
  if (threadIdx.x == 0) s_last = atomicAdd(&block[DB_DONE], 1ull) == (u64)gridDim.x - 1 ? 1u : 0u;
  __syncthreads();
  if (!s_last || threadIdx.x != 0) return;
  block[DB_DONE] = 0ull;
  const u64 n = __hip_atomic_load(&m.counters[CNT_SPARSE], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  const bool overflow = m.counters[CNT_SURVIVOR_LIST] > list_capacity || m.counters[CNT_REFINE_LIST] > refine_capacity || m.counters[CNT_REFINE_PEAK] > refine_capacity;
  const bool remask = (results[SR_STATUS] & (u64)(SERIES_MASKS_INVALID | SERIES_INF)) != 0;
  const bool full = n > cap || block[DB_BAD] != 0 || block[DB_BAD2] != 0 || overflow || remask;
  const long long asked = full ? -1ll : (long long)n;
  request[0] = (u64)asked;
  results[SR_HALO_ASKED] = (u64)asked;
  if (full) {
    atomicOr((unsigned long long *)&results[SR_STATUS], (unsigned long long)SERIES_HALO_FULL);
    m.counters[CNT_SERIES_DONE] = 2ull;                  // (the rest of the chain leaves at once; the finish kernel reports)
  }
  block[DB_BAD2] = 0ull;                                 // (as found, for the next pass that uses this block)
}

// ---- patches: the input values a cell's exact test and record can touch (corner - 2 .. corner + 3 on every axis, clamped to the array),
// ncomp values per vertex.  The cell count is the request's first word, read HERE: the owner gathers into a reply of fixed size without
// ever seeing the count on the host; the asker scatters the reply into its masks-only slice. ------------------------------------------
template <bool SCATTER>
__global__ __launch_bounds__(kThreads) void dist_patches_kernel(const Mesh m, const u64 *__restrict__ request, u64 cap, int ncomp, double *field, double *patches, u64 *served /* nullable */)
{
  const long long asked = (long long)request[0];
  if (served && blockIdx.x == 0 && threadIdx.x == 0) *served = (u64)asked;
  if (asked <= 0 || (u64)asked > cap) return;
  const u64 *cells = request + 1;
  const int nd = m.nd, pe = nd == 3 ? 216 : 36;
  const size_t total = (size_t)asked * (size_t)pe;
  for (size_t i = (size_t)blockIdx.x * kThreads + threadIdx.x; i < total; i += (size_t)gridDim.x * kThreads) {
    const size_t cell = i / pe;
    int p = (int)(i - cell * pe);
    u64 lin = cells[cell];
    size_t at = 0, stride = 1;
    bool ok = true;
    for (int a = 0; a < nd; a ++) {
      const u64 c = lin % (u64)m.core_sz[a]; lin /= (u64)m.core_sz[a];
      const int corner = m.core_st[a] + (int)c - m.ext_st[a];
      const int x = clampi(corner - 2 + p % 6, 0, m.ext_sz[a] - 1); p /= 6;
      at += (size_t)x * stride; stride *= (size_t)m.ext_sz[a];
    }
    ok = lin == 0;                                        // (a cell index from another rank: outside the core = ignored, never an address)
    if (!ok) { if (!SCATTER) for (int c = 0; c < ncomp; c ++) patches[i * ncomp + c] = 0.0; continue; }
    for (int c = 0; c < ncomp; c ++) {
      if (SCATTER) field[at * ncomp + c] = patches[i * ncomp + c];
      else patches[i * ncomp + c] = field[at * ncomp + c];
    }
  }
}

void launch_dist_contrib(const SeriesSlice *slices, int nown, const u64 *red, u64 *contrib, u64 *block, hipStream_t st)
{ hipLaunchKernelGGL(dist_contrib_kernel, dim3(1), dim3(256), 0, st, slices, nown, red, contrib, block); }
void launch_dist_export(const Mesh &m, const unsigned char *U, const unsigned char *M, u64 u_bytes, u64 *hdr, unsigned *idx, u64 *words, u64 capacity, int factor_log2, u64 *block, hipStream_t st)
{ hipLaunchKernelGGL(dist_export_kernel, dim3(512), dim3(256), 0, st, m, reinterpret_cast<const u64 *>(U), M, u_bytes, hdr, idx, words, capacity, (unsigned)factor_log2, block); }
void launch_dist_import(const u64 *gathered, int rank, int nranks, double running_in, u64 *block, u64 *results_tail, const u64 *hdr, const unsigned *idx, const u64 *words, u64 u_bytes,
                        u64 capacity, int u_rows, int max_factor_log2, unsigned char *U, unsigned char *M, u64 mask_words, hipStream_t st)
{
  hipLaunchKernelGGL(dist_import_kernel, dim3(hdr ? 128 : 1), dim3(256), 0, st, gathered, rank, nranks, running_in, block, results_tail, hdr, idx, words, u_bytes, capacity, (unsigned)u_rows,
                     (unsigned)max_factor_log2, reinterpret_cast<u64 *>(U), M, mask_words);
}
void launch_dist_cells(const Mesh &m, const Fields *d_steps, const u64 *d_list, u64 list_capacity, u64 refine_capacity, const double *halo_field, u64 *request, u64 cap, u64 *block, u64 *results, hipStream_t st)
{ hipLaunchKernelGGL(dist_cells_kernel, dim3(256), dim3(256), 0, st, m, d_steps, d_list, list_capacity, refine_capacity, halo_field, request, cap, block, results); }
void launch_dist_patches(const Mesh &m, bool scatter, const u64 *request, u64 cap, int ncomp, double *field, double *patches, u64 *served, hipStream_t st)
{
  size_t b = (cap * (size_t)(m.nd == 3 ? 216 : 36) + kThreads - 1) / kThreads;
  if (b > 1024) b = 1024;
  if (b < 1) b = 1;
  if (scatter) hipLaunchKernelGGL(dist_patches_kernel<true>, dim3((unsigned)b), dim3(kThreads), 0, st, m, request, cap, ncomp, field, patches, served);
  else hipLaunchKernelGGL(dist_patches_kernel<false>, dim3((unsigned)b), dim3(kThreads), 0, st, m, request, cap, ncomp, field, patches, served);
}

}  // namespace ftkx
