// Kernels of the compact t-slab halo: mask words the summaries do not describe, packed mask messages, the cells a
// neighbour asks values for, and the patches around them.  (Split from sweep_kernels.hip in round 6.)
#include "internal.hpp"
#include "sweep_device.hpp"
#include "series_device.hpp"

namespace ftkx {

// ---------------------------------------------------------------------------------------------------------------
// Compact t-slab halo.  The rank that owns a boundary slice hands its neighbour the slice's sign masks -- the summary
// array U as it is and the mask words the summaries do not describe, compacted here -- instead of the slice; the neighbour culls
// with them, asks for the input values around the few cells that survive (sparse_cells_kernel -> gather_patches_kernel on the
// owner -> scatter_patches_kernel into an otherwise empty array on the neighbour) and runs the exact test on those.
// ---------------------------------------------------------------------------------------------------------------
// words of M whose summary byte is 0 (the only ones the mask kernels write): (index of the 8-byte word in M, its 8 bytes)
__global__ __launch_bounds__(kThreads) void compact_words_kernel(const Mesh m, const unsigned char *__restrict__ U, const unsigned char *__restrict__ M,
                                                                 unsigned *__restrict__ idx, u64 *__restrict__ words, u64 capacity, u64 *counter)
{
  const int UP = m.u_pitch, P = m.mask_pitch, DH = m.ext_sz[1], DD = m.nd == 3 ? m.ext_sz[2] : 1;
  const int ngroups = (m.ext_sz[0] + 7) / 8;
  const size_t total = (size_t)ngroups * DH * DD, padded = (total + 63) / 64 * 64;
  const size_t urows = (size_t)((DH + m.u_rows - 1) / m.u_rows);
  for (size_t i = (size_t)blockIdx.x * kThreads + threadIdx.x; i < padded; i += (size_t)gridDim.x * kThreads) {
    bool take = false;
    size_t row = 0; int g = 0;
    if (i < total) {
      row = i / ngroups; g = (int)(i - row * ngroups);
      const size_t k = row / (size_t)DH, j = row - k * (size_t)DH;       // the summary of the word's block: row j / u_rows of plane k
      take = U[(j / (size_t)m.u_rows + urows * k) * (size_t)UP + g] == 0;
    }
    const unsigned long long b = __ballot(take);
    if (!b) continue;
    const int lane = threadIdx.x & 63, leader = __ffsll((long long)b) - 1;
    u64 base = 0;
    if (lane == leader) base = atomicAdd(counter, (u64)__popcll(b));
    base = __shfl(base, leader);
    if (take) {
      const u64 slot = base + (u64)__popcll(b & ((1ull << lane) - 1ull));
      const size_t w = (row * (size_t)P) / 8 + (size_t)g;                 // P is a multiple of 8: whole words
      if (slot < capacity) { idx[slot] = (unsigned)w; words[slot] = reinterpret_cast<const u64 *>(M)[w]; }
    }
  }
}

// (word indices come from another rank: anything outside the mask array is dropped and flagged, never written)
__global__ __launch_bounds__(kThreads) void scatter_words_kernel(const unsigned *__restrict__ idx, const u64 *__restrict__ words, size_t n, unsigned char *__restrict__ M,
                                                                 size_t mask_words, u64 *bad)
{
  const size_t i = (size_t)blockIdx.x * kThreads + threadIdx.x;
  if (i >= n) return;
  const size_t w = idx[i];
  if (w < mask_words) reinterpret_cast<u64 *>(M)[w] = words[i];
  else if (bad) atomicOr((unsigned long long *)bad, 1ull);
}

// Packed form of a slice's masks -- ONE message for the compact halo: u64 header {words, summary bytes, word capacity | rows per summary
// byte << 48 | log2(factor the masks were built under) << 56, magic}, the summary array, the word indices (capacity entries), the words
// (capacity entries).  The header is written on the device (the count of compacted words lives there) and read on the device: neither
// side waits for the other's numbers on the host.  The summary array travels through these kernels too (a copy queued through the
// runtime behind a running kernel holds the host until that kernel has finished: DESIGN.md 4, cull-ahead).
constexpr u64 kPackedMagic = 0x66746b786d61736bull;       // "ftkxmask"
__global__ __launch_bounds__(kThreads) void pack_masks_kernel(u64 *__restrict__ hdr, const u64 *__restrict__ counter, const u64 *__restrict__ U, u64 u_bytes, u64 capacity,
                                                              unsigned u_rows, unsigned factor_log2)
{
  u64 *dst = hdr + 4;
  const u64 nw = (u_bytes + 7) / 8;                        // (U is allocated in whole words: u_pitch is a multiple of 8)
  for (u64 i = (u64)blockIdx.x * kThreads + threadIdx.x; i < nw; i += (u64)gridDim.x * kThreads) dst[i] = U[i];
  if (blockIdx.x == 0 && threadIdx.x == 0) { hdr[0] = *counter; hdr[1] = u_bytes; hdr[2] = capacity | ((u64)u_rows << 48) | ((u64)factor_log2 << 56); hdr[3] = kPackedMagic; }
}

// import: header checked (geometry, rows per summary byte, and the factor the sender built the masks under must not exceed max_factor_log2:
// masks only serve factors at least as large as their own), summary array and words into the slice's arrays
__global__ __launch_bounds__(kThreads) void scatter_packed_kernel(const u64 *__restrict__ hdr, const unsigned *__restrict__ idx, const u64 *__restrict__ words,
                                                                  u64 u_bytes, u64 capacity, unsigned u_rows, unsigned max_factor_log2, u64 *__restrict__ U,
                                                                  unsigned char *__restrict__ M, size_t mask_words, u64 *bad)
{
  const u64 n = hdr[0], geo = hdr[2];
  if (hdr[1] != u_bytes || (geo & ((1ull << 48) - 1ull)) != capacity || ((geo >> 48) & 0xffull) != (u64)u_rows || (geo >> 56) > (u64)max_factor_log2 ||
      hdr[3] != kPackedMagic || n > capacity) {            // another geometry or mask setting, a larger factor, or more words than the message holds
    if (blockIdx.x == 0 && threadIdx.x == 0) atomicOr((unsigned long long *)bad, 1ull);
    return;
  }
  const u64 *src = hdr + 4;
  const u64 nw = (u_bytes + 7) / 8;
  for (u64 i = (u64)blockIdx.x * kThreads + threadIdx.x; i < nw; i += (u64)gridDim.x * kThreads) U[i] = src[i];
  for (u64 i = (u64)blockIdx.x * kThreads + threadIdx.x; i < n; i += (u64)gridDim.x * kThreads) {
    const size_t w = idx[i];
    if (w < mask_words) reinterpret_cast<u64 *>(M)[w] = words[i];
    else atomicOr((unsigned long long *)bad, 1ull);
  }
}

// survivors of the cull whose interval sweep reads the slice `sparse` (by its S or V pointer): their corner index inside core
__global__ __launch_bounds__(kThreads) void sparse_cells_kernel(const Mesh m, const Fields *__restrict__ steps, const u64 *__restrict__ list, u64 list_capacity,
                                                                const double *sparse, u64 *__restrict__ cells, u64 cells_capacity)
{
  u64 count = m.counters[CNT_SURVIVOR_LIST];
  if (count > list_capacity) count = list_capacity;
  const u64 padded = (count + 63) / 64 * 64;
  for (u64 i = (u64)blockIdx.x * kThreads + threadIdx.x; i < padded; i += (u64)gridDim.x * kThreads) {
    bool take = false;
    u64 lin = 0;
    if (i < count) {
      const u64 e = list[i];
      const Fields &f = steps[e >> 44];
      lin = e & 0xffffffffffull;
      take = ((e >> 40) & 2) && (f.S[1] == sparse || f.V[1] == sparse);
    }
    const unsigned long long b = __ballot(take);
    if (!b) continue;
    const int lane = threadIdx.x & 63, leader = __ffsll((long long)b) - 1;
    u64 base = 0;
    if (lane == leader) base = atomicAdd(&m.counters[CNT_SPARSE], (u64)__popcll(b));
    base = __shfl(base, leader);
    if (take) { const u64 slot = base + (u64)__popcll(b & ((1ull << lane) - 1ull)); if (slot < cells_capacity) cells[slot] = lin; }
  }
}

// the input values a cell's exact test and record can touch: array coordinates corner - 2 .. corner + 3 on every axis (vertices
// 0/1, +-1 for the gradient, +-1 more for the Jacobian of the gradient), clamped to the array; ncomp values per vertex
template <bool SCATTER>
__global__ __launch_bounds__(kThreads) void patches_kernel(const Mesh m, const u64 *__restrict__ cells, size_t n, int ncomp, double *field, double *patches)
{
  const int nd = m.nd, pe = nd == 3 ? 216 : 36;
  const size_t total = n * (size_t)pe;
  for (size_t i = (size_t)blockIdx.x * kThreads + threadIdx.x; i < total; i += (size_t)gridDim.x * kThreads) {
    const size_t cell = i / pe;
    int p = (int)(i - cell * pe);
    u64 lin = cells[cell];
    size_t at = 0, stride = 1;
    for (int a = 0; a < nd; a ++) {
      const int corner = m.core_st[a] + (int)(lin % (u64)m.core_sz[a]) - m.ext_st[a]; lin /= (u64)m.core_sz[a];
      const int x = clampi(corner - 2 + p % 6, 0, m.ext_sz[a] - 1); p /= 6;
      at += (size_t)x * stride; stride *= (size_t)m.ext_sz[a];
    }
    for (int c = 0; c < ncomp; c ++) {
      if (SCATTER) field[at * ncomp + c] = patches[i * ncomp + c];
      else patches[i * ncomp + c] = field[at * ncomp + c];
    }
  }
}

void launch_compact_words(const Mesh &m, const unsigned char *U, const unsigned char *M, unsigned *idx, u64 *words, u64 capacity, u64 *counter, hipStream_t st)
{ hipLaunchKernelGGL(compact_words_kernel, dim3(256 * 8), dim3(kThreads), 0, st, m, U, M, idx, words, capacity, counter); }
void launch_scatter_words(const unsigned *idx, const u64 *words, size_t n, unsigned char *M, size_t mask_words, u64 *bad, hipStream_t st)
{ if (n) hipLaunchKernelGGL(scatter_words_kernel, dim3((unsigned)((n + kThreads - 1) / kThreads)), dim3(kThreads), 0, st, idx, words, n, M, mask_words, bad); }
void launch_pack_masks(u64 *hdr, const u64 *counter, const unsigned char *U, u64 u_bytes, u64 capacity, int u_rows, int factor_log2, hipStream_t st)
{ hipLaunchKernelGGL(pack_masks_kernel, dim3(128), dim3(kThreads), 0, st, hdr, counter, reinterpret_cast<const u64 *>(U), u_bytes, capacity, (unsigned)u_rows, (unsigned)factor_log2); }
void launch_scatter_packed(const u64 *hdr, const unsigned *idx, const u64 *words, u64 u_bytes, u64 capacity, int u_rows, int max_factor_log2, unsigned char *U, unsigned char *M,
                           size_t mask_words, u64 *bad, hipStream_t st)
{ hipLaunchKernelGGL(scatter_packed_kernel, dim3(128), dim3(kThreads), 0, st, hdr, idx, words, u_bytes, capacity, (unsigned)u_rows, (unsigned)max_factor_log2, reinterpret_cast<u64 *>(U), M, mask_words, bad); }
void launch_sparse_cells(const Mesh &m, const Fields *d_steps, const u64 *d_list, u64 cap, const double *sparse, u64 *cells, u64 cells_cap, hipStream_t st)
{ hipLaunchKernelGGL(sparse_cells_kernel, dim3(256 * 2), dim3(kThreads), 0, st, m, d_steps, d_list, cap, sparse, cells, cells_cap); }
void launch_patches(const Mesh &m, bool scatter, const u64 *cells, size_t n, int ncomp, double *field, double *patches, hipStream_t st)
{
  if (!n) return;
  size_t b = (n * (m.nd == 3 ? 216 : 36) + kThreads - 1) / kThreads;
  if (b > 4096) b = 4096;
  if (scatter) hipLaunchKernelGGL(patches_kernel<true>, dim3((unsigned)b), dim3(kThreads), 0, st, m, cells, n, ncomp, field, patches);
  else hipLaunchKernelGGL(patches_kernel<false>, dim3((unsigned)b), dim3(kThreads), 0, st, m, cells, n, ncomp, field, patches);
}

}  // namespace ftkx
