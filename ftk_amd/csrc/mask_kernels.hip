// The MASK kernels of the fast path (gfx950): one pass over a slice -- S, 8 bytes per vertex, the gradient evaluated in flight, or V for
// vector input -- writes a sign-mask byte per vertex, the block summaries and the reduction update_vector_field_scaling_factor needs
// (critical_point_tracker.hh:850-864, ndarray.hh:770-778).  HBM-bound; which kernel takes which mesh: launch_masks_impl at the end.
// (Split from sweep_kernels.hip in round 6.)
#include "internal.hpp"
#include "sweep_device.hpp"
#include "series_device.hpp"

namespace ftkx {

// ---------------------------------------------------------------------------------------------------------------
// FAST PATH 1/3: vertex sign masks.  One lane per mask byte (row pitch padded to whole 8-byte words, see Mesh::mask_pitch).
// ---------------------------------------------------------------------------------------------------------------
// wavefront-level fold of a kernel's {min, max} pair into one of the job's 64 slots (non-negative doubles order like their bits)
__device__ inline void red_commit(u64 *red, double mn, double mx, unsigned slot)
{
  for (int o = 32; o > 0; o >>= 1) { mn = fmin(mn, __shfl_down(mn, o)); mx = fmax(mx, __shfl_down(mx, o)); }
  if ((threadIdx.x & 63) == 0) {
    atomicMin(&red[2 * (slot & 63u)], (u64)__double_as_longlong(mn));
    atomicMax(&red[2 * (slot & 63u) + 1], (u64)__double_as_longlong(mx));
  }
}

template <int ND>
__global__ __launch_bounds__(kThreads) void mask_kernel(const Mesh m, const MaskJob *__restrict__ jobs)
{
  const MaskJob job = jobs[blockIdx.y];
  const int P = m.mask_pitch, DH = m.ext_sz[1], DD = (ND == 3) ? m.ext_sz[2] : 1;
  const size_t n = (size_t)P * DH * DD;
  double red_mn = DBL_MAX, red_mx = 0.0;
  for (size_t idx = (size_t)blockIdx.x * kThreads + threadIdx.x; idx < n; idx += (size_t)gridDim.x * kThreads) {
    const int i = (int)(idx % P), j = (int)((idx / P) % DH), k = (int)(idx / ((size_t)P * DH));
    unsigned char mk = kNeutral;   // padding, vertices outside the domain and non-finite vertices never block a cull
    if (i < m.ext_sz[0]) {
      const int vx[3] = {i + m.ext_st[0], j + m.ext_st[1], k + m.ext_st[2]};
      bool in_dom = true;
      for (int d = 0; d < ND; d ++) in_dom = in_dom && vx[d] >= m.dom_lb[d] && vx[d] <= m.dom_ub[d];
      double v[ND];
      vector_at<ND>(m, job.S, job.V, i, j, k, v);
      unsigned bits = 0;
      bool finite = true, big = false;
      for (int c = 0; c < ND; c ++) {
        const double a = fabs(v[c]);
        finite = finite && !(isnan(v[c]) || isinf(v[c]));
        if (v[c] >= job.threshold) bits |= 1u << c;          // trunc(v * factor) >= 1
        if (v[c] <= -job.threshold) bits |= 8u << c;         // trunc(v * factor) <= -1
        big = big || a >= job.big;
        // the reduction covers the WHOLE array, like ndarray::resolution() (ndarray.hh:770-778)
        red_mn = fmin(red_mn, (a == 0.0 || !(a < job.threshold)) ? DBL_MAX : a);
        red_mx = fmax(red_mx, a);
      }
      if (in_dom) mk = finite ? (unsigned char)(big ? 0u : bits) : kNeutral;
    }
    job.M[idx] = mk;
  }
  if (job.red) red_commit(job.red, red_mn, red_mx, blockIdx.x * 5u + (threadIdx.x >> 6));
}

// Vector input, row length a multiple of 8: four consecutive vertices per lane (ND x 32 contiguous bytes, 16-byte loads), their
// four mask bytes stored as one word, the summary of an aligned 8-vertex word formed with the neighbouring lane (two-level cull
// as for scalar input), and the slice's reduction (MaskJob::red) fused in -- V is read once for the whole sweep.
template <int ND>
__global__ __launch_bounds__(kThreads) void mask_vec_kernel(const Mesh m, const MaskJob *__restrict__ jobs)
{
  const MaskJob job = jobs[blockIdx.y];
  const int DW = m.ext_sz[0], DH = m.ext_sz[1], DD = (ND == 3) ? m.ext_sz[2] : 1, P = m.mask_pitch, UP = m.u_pitch;
  const size_t ngroups = (size_t)(DW / 4), total = ngroups * (size_t)DH * (size_t)DD;   // DW % 8 == 0: total is even
  const size_t padded = (total + 63) / 64 * 64;                                        // wave-uniform trip count (DPP, ballot)
  const double thr = job.threshold, nthr = -job.threshold;
  const bool have_u = job.U != nullptr;
  double red_mn = DBL_MAX, red_mx = 0.0;
  // a lane's group of 4 vertices -> (row, group in the row): 32-bit arithmetic where the slice allows it (a 64-bit division by a
  // run-time value is ~100 instructions, per 64 bytes of input)
  const bool small = total < (1ull << 32);                     // (wave-uniform)
  auto locate = [&](size_t gc, size_t &row, int &g) {
    if (small) { const unsigned r = (unsigned)gc / (unsigned)ngroups; row = r; g = (int)((unsigned)gc - r * (unsigned)ngroups); }
    else { row = gc / ngroups; g = (int)(gc - row * ngroups); }
  };
  auto fetch = [&](size_t gi, double2 (&t)[2 * ND]) {
    const size_t gc = gi < total ? gi : total - 1;
    size_t row; int g;
    locate(gc, row, g);
    const double2 *src = reinterpret_cast<const double2 *>(job.V + (row * (size_t)DW + (size_t)(4 * g)) * ND);
#pragma unroll
    for (int q = 0; q < 2 * ND; q ++) t[q] = src[q];
  };
  const size_t stride = (size_t)gridDim.x * kThreads;
  size_t gi = (size_t)blockIdx.x * kThreads + threadIdx.x;
  double2 nxt[2 * ND];
  if (gi < padded) fetch(gi, nxt);
  for (; gi < padded; gi += stride) {
    const bool live = gi < total;
    const size_t gc = live ? gi : total - 1;
    size_t row; int g;
    locate(gc, row, g);
    int j, k;
    if (ND == 2) { j = (int)row; k = 0; }
    else if (small) { k = (int)((unsigned)row / (unsigned)DH); j = (int)((unsigned)row - (unsigned)k * (unsigned)DH); }
    else { j = (int)(row % (size_t)DH); k = (int)(row / (size_t)DH); }
    double v[4 * ND];
#pragma unroll
    for (int q = 0; q < 2 * ND; q ++) { v[2 * q] = nxt[q].x; v[2 * q + 1] = nxt[q].y; }
    if (gi + stride < padded) fetch(gi + stride, nxt);         // the next group's loads are on their way while this one is classified
    bool row_dom = j + m.ext_st[1] >= m.dom_lb[1] && j + m.ext_st[1] <= m.dom_ub[1];
    if (ND == 3) row_dom = row_dom && k + m.ext_st[2] >= m.dom_lb[2] && k + m.ext_st[2] <= m.dom_ub[2];
    unsigned word = 0;
    bool cand = false;
    for (int q = 0; q < 4; q ++) {
      unsigned bits = 0;
      double mx = 0.0;
      bool fin = true;
      for (int c = 0; c < ND; c ++) {
        const double x = v[q * ND + c], a = fabs(x);
        if (x >= thr) bits |= 1u << c;
        if (x <= nthr) bits |= 8u << c;
        fin = fin && a < HUGE_VAL;                             // false for NaN and Inf
        mx = fmax(mx, a);
      }
      red_mx = live ? fmax(red_mx, mx) : red_mx;
      cand = cand || ((bits | (bits >> 3)) & ((1u << ND) - 1u)) != ((1u << ND) - 1u);   // a component without a strict sign
      const int x = 4 * g + q + m.ext_st[0];
      const bool dom = row_dom && x >= m.dom_lb[0] && x <= m.dom_ub[0];
      if (mx >= job.big) bits = 0u;                            // may overflow a determinant: supports no cull
      if (!fin || !dom) bits = kNeutral;                       // simplices with this vertex are rejected (2d:611, 3d:457) / never formed
      word |= bits << (8 * q);
    }
    if (__builtin_amdgcn_ballot_w64(cand && live)) {           // rare on smooth data: candidates for the slice's resolution
      if (live)
        for (int q = 0; q < 4 * ND; q ++) { const double a = fabs(v[q]); red_mn = fmin(red_mn, (a == 0.0 || !(a < thr)) ? DBL_MAX : a); }
    }
    bool word_uniform = false;
    if (have_u) {
      int q8 = (int)(word & (word >> 16)); q8 &= q8 >> 8; q8 &= 0x3f;
      q8 &= __builtin_amdgcn_update_dpp(q8, q8, 0xb1 /* quad_perm:[1,0,3,2] */, 0xf, 0xf, false);   // the other half of the 8-vertex word
      if (live && (g & 1) == 0) job.U[row * (size_t)UP + (size_t)(g >> 1)] = (unsigned char)q8;
      word_uniform = q8 != 0;                                  // the refine kernel substitutes the summary: mask bytes not needed
    }
    if (live && !word_uniform) *reinterpret_cast<unsigned *>(job.M + row * (size_t)P + (size_t)(4 * g)) = word;
  }
  if (job.red) red_commit(job.red, red_mn, red_mx, blockIdx.x * 5u + (threadIdx.x >> 6));
}

// ---------------------------------------------------------------------------------------------------------------
// The same walk as mask_march2_kernel<ND, EDGE = true, REDUCE> on a VALU diet.  rocprofv3 showed the kernel above issue-bound
// (VALUBusy 72 %, 2.96e9 VALU instructions per 512^3 x 32 launch) rather than HBM-bound, so this version removes instructions
// that do no arithmetic for the result:
//   * the four plane buffers rotate by NAME (the z loop is unrolled four times) instead of being copied every plane;
//   * row addresses are wave-uniform (SGPR base, readfirstlane'd wavefront id) + one constant per-lane byte offset, loaded
//     and stored through global (not flat) address space: no 64-bit VALU address arithmetic;
//   * 3D: 0.5 * (a - b) >= thr is tested as (a - b) >= 2 thr -- exact, scaling by a power of two -- so the multiply goes;
//   * lanes 0 / 63 take the outside neighbour through the DPP shift's `old` operand (an invalid source lane keeps `old`)
//     instead of two selects per value;
//   * the 6 sign bits of a vertex are shifted into an accumulator by add-with-carry straight from the compare masks.
// Same mask / summary bytes as the kernel above (tests/test_gpu_properties.py compares the two).
// ---------------------------------------------------------------------------------------------------------------
typedef double v2d __attribute__((ext_vector_type(2)));
#define FTKX_GLOBAL __attribute__((address_space(1)))

__device__ inline double dpp_lower_or(double v, double edge)   // lane n <- lane n-1; lane 0 <- its own `edge`
{
  const long long b = __double_as_longlong(v), e = __double_as_longlong(edge);
  const int lo = __builtin_amdgcn_update_dpp((int)e, (int)b, 0x138 /* wave_shr:1 */, 0xf, 0xf, false);
  const int hi = __builtin_amdgcn_update_dpp((int)(e >> 32), (int)(b >> 32), 0x138, 0xf, 0xf, false);
  return __longlong_as_double(((long long)hi << 32) | (unsigned)lo);
}
__device__ inline double dpp_upper_or(double v, double edge)   // lane n <- lane n+1; lane 63 <- its own `edge`
{
  const long long b = __double_as_longlong(v), e = __double_as_longlong(edge);
  const int lo = __builtin_amdgcn_update_dpp((int)e, (int)b, 0x130 /* wave_shl:1 */, 0xf, 0xf, false);
  const int hi = __builtin_amdgcn_update_dpp((int)(e >> 32), (int)(b >> 32), 0x130, 0xf, 0xf, false);
  return __longlong_as_double(((long long)hi << 32) | (unsigned)lo);
}

__device__ inline void remap_block(int swizzle, unsigned &bx, unsigned &by, unsigned &bz)
{
  bx = blockIdx.x; by = blockIdx.y; bz = blockIdx.z;
  if (swizzle & 1) {
    const unsigned nb = gridDim.x * gridDim.y * gridDim.z;
    unsigned b = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);
    const unsigned per = nb / 8, rem = nb % 8, xcd = b % 8, k = b / 8;
    b = xcd * per + (xcd < rem ? xcd : rem) + k;
    bx = b % gridDim.x; by = (b / gridDim.x) % gridDim.y; bz = b / (gridDim.x * gridDim.y);
  }
  if (swizzle & 8) {   // grouped placement, see mask_march2_kernel
    const unsigned YG = ((unsigned)swizzle >> 8) & 0xffu;
    const unsigned nb = gridDim.x * gridDim.y * gridDim.z, G = gridDim.x * YG;
    const unsigned b = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);
    if (b < (nb / (8u * G)) * (8u * G)) {
      const unsigned q = b % 8u, r = b / 8u, grp = (r / G) * 8u + q, in = r % G, ngy = gridDim.y / YG;
      bx = in % gridDim.x; by = (grp % ngy) * YG + in / gridDim.x; bz = grp / ngy;
    }
  }
}

typedef unsigned v4u __attribute__((ext_vector_type(4)));
typedef unsigned v2u __attribute__((ext_vector_type(2)));

// Shifts the sign bits of one row's vertex pair into two accumulators (overwritten: the first add-with-carry is 0 + 0 + carry),
// most significant first:
//   a = [neg_z neg_y neg_x pos_z pos_y pos_x]   (2D: [neg_y neg_x pos_y pos_x], spread by the caller)
// one compare + one add-with-carry (a = a + a + carry) per bit.  gfx950 needs two wait states between a VALU writing an
// SGPR and a VALU reading it; the compares run three ahead of the adds, so no s_nop is needed.
template <int ND>
__device__ inline void shift_in_signs(unsigned &a0, unsigned &a1, double dx0, double dx1, double dy0, double dy1, double dz0, double dz1, double tn, double tp)
{
  unsigned long long m0, m1, m2;
  if constexpr (ND == 3)
    asm("v_cmp_le_f64_e64 %[m0], %[dz0], %[tn]\n\tv_cmp_le_f64_e64 %[m1], %[dz1], %[tn]\n\tv_cmp_le_f64_e64 %[m2], %[dy0], %[tn]\n\t"
        "v_addc_co_u32_e64 %[a0], vcc, 0, 0, %[m0]\n\tv_cmp_le_f64_e64 %[m0], %[dy1], %[tn]\n\t"
        "v_addc_co_u32_e64 %[a1], vcc, 0, 0, %[m1]\n\tv_cmp_le_f64_e64 %[m1], %[dx0], %[tn]\n\t"
        "v_addc_co_u32_e64 %[a0], vcc, %[a0], %[a0], %[m2]\n\tv_cmp_le_f64_e64 %[m2], %[dx1], %[tn]\n\t"
        "v_addc_co_u32_e64 %[a1], vcc, %[a1], %[a1], %[m0]\n\tv_cmp_ge_f64_e64 %[m0], %[dz0], %[tp]\n\t"
        "v_addc_co_u32_e64 %[a0], vcc, %[a0], %[a0], %[m1]\n\tv_cmp_ge_f64_e64 %[m1], %[dz1], %[tp]\n\t"
        "v_addc_co_u32_e64 %[a1], vcc, %[a1], %[a1], %[m2]\n\tv_cmp_ge_f64_e64 %[m2], %[dy0], %[tp]\n\t"
        "v_addc_co_u32_e64 %[a0], vcc, %[a0], %[a0], %[m0]\n\tv_cmp_ge_f64_e64 %[m0], %[dy1], %[tp]\n\t"
        "v_addc_co_u32_e64 %[a1], vcc, %[a1], %[a1], %[m1]\n\tv_cmp_ge_f64_e64 %[m1], %[dx0], %[tp]\n\t"
        "v_addc_co_u32_e64 %[a0], vcc, %[a0], %[a0], %[m2]\n\tv_cmp_ge_f64_e64 %[m2], %[dx1], %[tp]\n\t"
        "v_addc_co_u32_e64 %[a1], vcc, %[a1], %[a1], %[m0]\n\tv_addc_co_u32_e64 %[a0], vcc, %[a0], %[a0], %[m1]\n\t"
        "v_addc_co_u32_e64 %[a1], vcc, %[a1], %[a1], %[m2]"
        : [a0] "=&v"(a0), [a1] "=&v"(a1), [m0] "=&s"(m0), [m1] "=&s"(m1), [m2] "=&s"(m2)
        : [dx0] "v"(dx0), [dx1] "v"(dx1), [dy0] "v"(dy0), [dy1] "v"(dy1), [dz0] "v"(dz0), [dz1] "v"(dz1), [tn] "s"(tn), [tp] "s"(tp)
        : "vcc");
  else
    asm("v_cmp_le_f64_e64 %[m0], %[dy0], %[tn]\n\tv_cmp_le_f64_e64 %[m1], %[dy1], %[tn]\n\tv_cmp_le_f64_e64 %[m2], %[dx0], %[tn]\n\t"
        "v_addc_co_u32_e64 %[a0], vcc, 0, 0, %[m0]\n\tv_cmp_le_f64_e64 %[m0], %[dx1], %[tn]\n\t"
        "v_addc_co_u32_e64 %[a1], vcc, 0, 0, %[m1]\n\tv_cmp_ge_f64_e64 %[m1], %[dy0], %[tp]\n\t"
        "v_addc_co_u32_e64 %[a0], vcc, %[a0], %[a0], %[m2]\n\tv_cmp_ge_f64_e64 %[m2], %[dy1], %[tp]\n\t"
        "v_addc_co_u32_e64 %[a1], vcc, %[a1], %[a1], %[m0]\n\tv_cmp_ge_f64_e64 %[m0], %[dx0], %[tp]\n\t"
        "v_addc_co_u32_e64 %[a0], vcc, %[a0], %[a0], %[m1]\n\tv_cmp_ge_f64_e64 %[m1], %[dx1], %[tp]\n\t"
        "v_addc_co_u32_e64 %[a1], vcc, %[a1], %[a1], %[m2]\n\tv_addc_co_u32_e64 %[a0], vcc, %[a0], %[a0], %[m0]\n\t"
        "v_addc_co_u32_e64 %[a1], vcc, %[a1], %[a1], %[m1]"
        : [a0] "=&v"(a0), [a1] "=&v"(a1), [m0] "=&s"(m0), [m1] "=&s"(m1), [m2] "=&s"(m2)
        : [dx0] "v"(dx0), [dx1] "v"(dx1), [dy0] "v"(dy0), [dy1] "v"(dy1), [tn] "s"(tn), [tp] "s"(tp)
        : "vcc");
}

// 2D, the bits at their final places in the 3D layout [0 neg_y neg_x 0 pos_y pos_x] (one shift between the two halves instead of a
// mask / shift / or per accumulator afterwards) and the compares against thresholds of the UNSCALED differences (see exact_threshold)
__device__ inline void shift_in_signs2(unsigned &a0, unsigned &a1, double dx0, double dx1, double dy0, double dy1, double tnx, double tpx, double tny, double tpy)
{
  unsigned long long m0, m1, m2;
  asm("v_cmp_le_f64_e64 %[m0], %[dy0], %[tny]\n\tv_cmp_le_f64_e64 %[m1], %[dy1], %[tny]\n\tv_cmp_le_f64_e64 %[m2], %[dx0], %[tnx]\n\t"
      "v_addc_co_u32_e64 %[a0], vcc, 0, 0, %[m0]\n\tv_cmp_le_f64_e64 %[m0], %[dx1], %[tnx]\n\t"
      "v_addc_co_u32_e64 %[a1], vcc, 0, 0, %[m1]\n\tv_cmp_ge_f64_e64 %[m1], %[dy0], %[tpy]\n\t"
      "v_addc_co_u32_e64 %[a0], vcc, %[a0], %[a0], %[m2]\n\tv_cmp_ge_f64_e64 %[m2], %[dy1], %[tpy]\n\t"
      "v_addc_co_u32_e64 %[a1], vcc, %[a1], %[a1], %[m0]\n\tv_cmp_ge_f64_e64 %[m0], %[dx0], %[tpx]\n\t"
      "v_lshlrev_b32_e32 %[a0], 1, %[a0]\n\tv_lshlrev_b32_e32 %[a1], 1, %[a1]\n\t"
      "v_addc_co_u32_e64 %[a0], vcc, %[a0], %[a0], %[m1]\n\tv_cmp_ge_f64_e64 %[m1], %[dx1], %[tpx]\n\t"
      "v_addc_co_u32_e64 %[a1], vcc, %[a1], %[a1], %[m2]\n\tv_addc_co_u32_e64 %[a0], vcc, %[a0], %[a0], %[m0]\n\t"
      "v_addc_co_u32_e64 %[a1], vcc, %[a1], %[a1], %[m1]"
      : [a0] "=&v"(a0), [a1] "=&v"(a1), [m0] "=&s"(m0), [m1] "=&s"(m1), [m2] "=&s"(m2)
      : [dx0] "v"(dx0), [dx1] "v"(dx1), [dy0] "v"(dy0), [dy1] "v"(dy1), [tnx] "s"(tnx), [tpx] "s"(tpx), [tny] "s"(tny), [tpy] "s"(tpy)
      : "vcc");
}

// compile-time loop: DPP controls must be integer constant expressions, so the row index has to be one
template <class F, int... I> __device__ inline void static_for_impl(F &&f, std::integer_sequence<int, I...>) { (f(std::integral_constant<int, I>{}), ...); }
template <int N, class F> __device__ inline void static_for(F &&f) { static_for_impl(f, std::make_integer_sequence<int, N>{}); }

// The outside-neighbour values of a plane live in ONE register: lane r holds the left neighbour of row r's first column, lane
// 64 - RY + r the right neighbour of its last column (r = 0 .. RY-1; one buffer load per plane, 2 RY lanes active).  Row r's
// pair is moved to lanes 0 and 63 -- where the wavefront shifts below pick it up as their `old` operand -- by two row shifts
// (row_shl:r restricted to lanes 0-15, row_shr:(RY-1-r) restricted to lanes 48-63).
template <int R_, int RY_>
__device__ inline double edge_for_row(double xe)
{
  const long long b = __double_as_longlong(xe);
  const int lo0 = (int)b, hi0 = (int)(b >> 32);
  int lo = lo0, hi = hi0;
  if constexpr (R_ > 0) {
    lo = __builtin_amdgcn_update_dpp(lo0, lo0, 0x100 + R_ /* row_shl:R_ */, 0x1, 0xf, false);
    hi = __builtin_amdgcn_update_dpp(hi0, hi0, 0x100 + R_, 0x1, 0xf, false);
  }
  constexpr int S_ = RY_ - 1 - R_;
  if constexpr (S_ > 0) {
    lo = __builtin_amdgcn_update_dpp(lo, lo0, 0x110 + S_ /* row_shr:S_ */, 0x8, 0xf, false);
    hi = __builtin_amdgcn_update_dpp(hi, hi0, 0x110 + S_, 0x8, 0xf, false);
  }
  return __longlong_as_double(((long long)hi << 32) | (unsigned)lo);
}

// summary byte of an aligned 8-vertex word from the pair of mask bytes each of the quad's four lanes holds: the sign bits ALL
// eight vertices share.  Three VALU instructions (SDWA byte select, then the AND folded into the DPP quad permutes); written by
// hand because the compiler emits mov_dpp + and pairs (7 instructions).  s_nop 1 = the two wait states a DPP read needs after a
// VALU write of its source.
__device__ inline unsigned word_summary(unsigned bits)
{
  unsigned q;
  asm("v_and_b32_sdwa %0, %1, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_0 src1_sel:BYTE_1\n\t"
      "s_nop 1\n\t"
      "v_and_b32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
      "s_nop 1\n\t"
      "v_and_b32_dpp %0, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf"
      : "=&v"(q) : "v"(bits));
  return q;
}

// What the marching kernels do per row beside the sign bits, so that ONE pass over S serves the whole sweep (MaskJob):
//   * running max of |d| per column (a0 / a1 = this lane's two vertices): the slice's max |v|;
//   * candidates for the slice's resolution: only components WITHOUT a strict sign (|v| < threshold) can lower the scaling
//     factor's running minimum below 1 / F, and on smooth data almost no row has one -- one ballot per row, the min-non-zero
//     arithmetic runs only where it fires;
//   * vertices that may overflow a determinant (|v| >= big) lose their sign bits -- only when the job asks for it (big finite):
//     the host first builds masks WITHOUT the rule and keeps them only if the fused max shows that no vertex is big.
// d* are the raw central differences the sign bits were taken from (3D: twice the gradient, h = 0.5; 2D: the gradient, h = 1);
// `rv` (wave-uniform): this row holds real entries of gradient(S); cmask: which of the lane's two columns do, as the bits
// 0x07 (0x03 in 2D) of their byte.  Returns the pair of mask bytes.
// v_max_f64 with |x| input modifiers, as ONE instruction: fmax() makes the compiler canonicalise both operands first (two more
// v_max_f64 each) although arithmetic results are never signalling NaNs.  A NaN operand is ignored (IEEE maxNum).
__device__ inline double max_abs2(double a, double b) { double r; asm("v_max_f64 %0, |%1|, |%2|" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ inline double max_with_abs(double acc, double b) { asm("v_max_f64 %0, %0, |%1|" : "+v"(acc) : "v"(b)); return acc; }
__device__ inline double max_plain(double acc, double b) { asm("v_max_f64 %0, %0, %1" : "+v"(acc) : "v"(b)); return acc; }

template <int ND>
__device__ inline unsigned guard_and_reduce(unsigned a0, unsigned a1, double dx0, double dx1, double dy0, double dy1, double dz0, double dz1,
                                            bool rv, unsigned cmask, bool per_vertex_rule, double tbig, double h, double &acc0, double &acc1, double &red_mn)
{
  double m0 = max_abs2(dx0, dy0), m1 = max_abs2(dx1, dy1);
  if constexpr (ND == 3) { m0 = max_with_abs(m0, dz0); m1 = max_with_abs(m1, dz1); }
  unsigned bits = a0 | (a1 << 8);
  if (rv) {                                                    // wave-uniform
    acc0 = max_plain(acc0, m0); acc1 = max_plain(acc1, m1);
    const unsigned u = bits | (bits >> 3);                     // bit c of a byte: component c is strictly signed
    if (__builtin_amdgcn_ballot_w64((~u & cmask) != 0u)) {
      auto take = [&](double d) { const double a = fabs(h * d); red_mn = fmin(red_mn, a == 0.0 ? DBL_MAX : a); };
      if (cmask & 0x00ffu) { take(dx0); take(dy0); if (ND == 3) take(dz0); }
      if (cmask & 0xff00u) { take(dx1); take(dy1); if (ND == 3) take(dz1); }
    }
  }
  if (per_vertex_rule) {                                       // wave-uniform: only slices known to hold such vertices pay for it
    asm volatile("" ::: "memory");                             // (keeps this a branch: if-converted it costs every row 8 instructions)
    bits = (m0 >= tbig ? 0u : (bits & 0x00ffu)) | (m1 >= tbig ? 0u : (bits & 0xff00u));
  }
  return bits;
}

// ---------------------------------------------------------------------------------------------------------------
// mask_vec_kernel with ONE summary byte per 8 x 4 block of vertices (Mesh::u_rows == 4) and on a VALU diet (round 4).
// Why blocks: every byte of summary a streaming kernel stores costs far more than its share of the traffic (on
// double_gyre 2048 x 1024 x 128 the 32 summary bytes per 4 KB of input were 39 us of the kernel's 726, measured with the stores aimed out of range -- tools/probe/bw_probe.hip
// `v` shows the same on a bare read walk); a quarter of the bytes is three quarters of that gone, and the coarse cull reads a
// quarter too.  A wavefront therefore takes a unit of 4 rows x 64 groups (16 KB): the block's summary is an AND across its own
// registers and one neighbouring lane.
// Why the diet: rocprofv3 had the kernel above at VALUBusy 68 %, 279 VALU instructions per 4 KB of input
// (profiles/r04_c5_valu_summary.json) -- two 32-bit divisions per step to find (row, group), 64-bit address arithmetic for every
// load and store, a compare + select + or per sign bit, per-vertex domain / finiteness / magnitude tests whose outcome is the same
// for all but a few wavefronts.  Here
//   * a unit's coordinates are wave-uniform and advance by additions on the scalar unit; loads and stores go through buffer
//     resources (SGPR offset + one constant per-lane offset);
//   * a component is strictly signed iff |x| >= thr -- ONE compare, shifted into an accumulator by add-with-carry -- and its sign is
//     the top bit of the double, shifted into a second accumulator by v_alignbit: pos = strict & ~sign, neg = strict & sign
//     (thr = 1 / F > 0: neither a zero nor a NaN is strict);
//   * the border of the domain, non-finite values and values past `big` are detected per WAVEFRONT and row (scalar comparisons of
//     the chunk's ends with the domain; one v_max chain, which the reduction needs anyway, and one unordered-compare per pair of
//     values); only then are the row's vertices walked one by one as the kernel above does.
// (The diet alone, on the kernel above's linear walk: 0.744 -> 0.742 ms -- the kernel was not waiting for its arithmetic.)
// Same fused reduction and records as mask_vec_kernel; the same mask words wherever both write them (tests/test_gpu_properties.py:
// fields with NaNs, infinities, big values and plateaus).  Needs rows of at least 64 groups and byte offsets that fit 31 bits:
// vec_lean() below.
// ---------------------------------------------------------------------------------------------------------------
template <int ND>
__device__ inline void push_vertex_signs(unsigned &S, unsigned &G, const double *x, double thr)
{
  // S = S << ND | strict bits, G = G << ND | sign bits, most significant component first; the compares run ahead of the
  // add-with-carry that consumes them (two wait states between a VALU writing an SGPR and a VALU reading it)
  unsigned long long m0, m1, m2;
  if constexpr (ND == 2)
    asm("v_cmp_ge_f64_e64 %[m0], |%[x1]|, %[thr]\n\tv_cmp_ge_f64_e64 %[m1], |%[x0]|, %[thr]\n\t"
        "v_alignbit_b32 %[G], %[G], %[h1], 31\n\tv_alignbit_b32 %[G], %[G], %[h0], 31\n\t"
        "v_addc_co_u32_e64 %[S], vcc, %[S], %[S], %[m0]\n\tv_addc_co_u32_e64 %[S], vcc, %[S], %[S], %[m1]"
        : [S] "+v"(S), [G] "+v"(G), [m0] "=&s"(m0), [m1] "=&s"(m1)
        : [x0] "v"(x[0]), [x1] "v"(x[1]), [h0] "v"((int)(__double_as_longlong(x[0]) >> 32)), [h1] "v"((int)(__double_as_longlong(x[1]) >> 32)), [thr] "s"(thr)
        : "vcc");
  else
    asm("v_cmp_ge_f64_e64 %[m0], |%[x2]|, %[thr]\n\tv_cmp_ge_f64_e64 %[m1], |%[x1]|, %[thr]\n\tv_cmp_ge_f64_e64 %[m2], |%[x0]|, %[thr]\n\t"
        "v_alignbit_b32 %[G], %[G], %[h2], 31\n\tv_addc_co_u32_e64 %[S], vcc, %[S], %[S], %[m0]\n\t"
        "v_alignbit_b32 %[G], %[G], %[h1], 31\n\tv_addc_co_u32_e64 %[S], vcc, %[S], %[S], %[m1]\n\t"
        "v_alignbit_b32 %[G], %[G], %[h0], 31\n\tv_addc_co_u32_e64 %[S], vcc, %[S], %[S], %[m2]"
        : [S] "+v"(S), [G] "+v"(G), [m0] "=&s"(m0), [m1] "=&s"(m1), [m2] "=&s"(m2)
        : [x0] "v"(x[0]), [x1] "v"(x[1]), [x2] "v"(x[ND - 1]), [h0] "v"((int)(__double_as_longlong(x[0]) >> 32)), [h1] "v"((int)(__double_as_longlong(x[1]) >> 32)),
          [h2] "v"((int)(__double_as_longlong(x[ND - 1]) >> 32)), [thr] "s"(thr)
        : "vcc");
}

// AND of the four mask bytes of a word, then with the neighbouring lane's: the summary of the aligned 8-vertex word (mask bytes never
// carry bits 6 and 7)
__device__ inline unsigned word_summary4(unsigned word)
{
  unsigned q;
  asm("v_and_b32_sdwa %0, %1, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0 src1_sel:WORD_1\n\t"
      "s_nop 0\n\t"
      "v_and_b32_sdwa %0, %0, %0 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_0 src1_sel:BYTE_1\n\t"
      "s_nop 1\n\t"
      "v_and_b32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf"
      : "=&v"(q) : "v"(word));
  return q;
}

template <int ND>
__global__ __launch_bounds__(kThreads) void mask_vec2_kernel(const Mesh m, const MaskJob *__restrict__ jobs)
{
  constexpr int NV = 4 * ND, NL = 2 * ND;                      // values and 16-byte loads per lane and row
  constexpr unsigned GB = 32u * ND;                            // bytes of a group of 4 vertices
  constexpr unsigned FULL = (ND == 2) ? 0x03030303u : 0x07070707u;
  const MaskJob job = jobs[blockIdx.y];
  const unsigned DW = (unsigned)m.ext_sz[0], DH = (unsigned)m.ext_sz[1], DD = (ND == 3) ? (unsigned)m.ext_sz[2] : 1u;
  const unsigned P = (unsigned)m.mask_pitch, UP = (unsigned)m.u_pitch;
  const unsigned ngroups = DW / 4u, nch = (ngroups + 63u) / 64u, nby = (DH + 3u) / 4u, nunits = nch * nby * DD;
  const unsigned lane = threadIdx.x & 63u;
  const unsigned wv = (unsigned)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const unsigned W = gridDim.x * (unsigned)(kThreads / 64);    // units between a wavefront's units
  const double thr = job.threshold;
  const bool have_u = job.U != nullptr;
  const __amdgpu_buffer_rsrc_t rV = __builtin_amdgcn_make_buffer_rsrc((void *)job.V, 0, (int)(ngroups * DH * DD * GB), 0x00020000);
  const __amdgpu_buffer_rsrc_t rM = __builtin_amdgcn_make_buffer_rsrc((void *)job.M, 0, (int)(P * DH * DD), 0x00020000);
  const __amdgpu_buffer_rsrc_t rU = __builtin_amdgcn_make_buffer_rsrc((void *)job.U, 0, have_u ? (int)(UP * nby * DD) : 0, 0x00020000);

  // unit u = (chunk ch of 64 groups, block jb of 4 rows, plane k), chunks fastest: found once by division, then advanced by additions
  unsigned u = blockIdx.x * (unsigned)(kThreads / 64) + wv;
  unsigned ch, jb, k;
  { const unsigned t = u / nch; ch = u - t * nch; k = t / nby; jb = t - k * nby; }
  unsigned dch, djb, dk;
  { const unsigned t = W / nch; dch = W - t * nch; dk = t / nby; djb = t - dk * nby; }
  // groups whose four vertices all lie inside the domain along x: glo .. ghi
  const int xl = m.dom_lb[0] - m.ext_st[0], xh = m.dom_ub[0] - m.ext_st[0];
  const int glo = xl <= 0 ? 0 : (xl + 3) / 4, ghi = xh < 3 ? -1 : (xh - 3) / 4;

  const unsigned lv = lane * GB, lm = lane * 4u, lu = lane >> 1;
  const bool u_lane = (lane & 1u) == 0;
  double red_mn = DBL_MAX, red_mx = 0.0;
  while (u < nunits) {
    // (wave-uniform by construction; said again, the compiler keeps the unit's coordinates and everything derived from them on the scalar
    // unit -- without it the loads' offsets end up in a VGPR and every load in a waterfall loop)
    u = (unsigned)__builtin_amdgcn_readfirstlane((int)u); ch = (unsigned)__builtin_amdgcn_readfirstlane((int)ch);
    jb = (unsigned)__builtin_amdgcn_readfirstlane((int)jb); k = (unsigned)__builtin_amdgcn_readfirstlane((int)k);
    const unsigned g0 = ch * 64u, j0 = jb * 4u;
    const bool lane_live = g0 + lane < ngroups;                // (only a row's last chunk has lanes past its end)
    v4u D[4][NL];
#pragma unroll
    for (int r = 0; r < 4; r ++) {                             // all four rows' loads, then the rows as they arrive (rows past the slice: the last row again)
      const unsigned j = j0 + (unsigned)r < DH ? j0 + (unsigned)r : DH - 1u;
      const unsigned so = ((k * DH + j) * ngroups + g0) * GB;
#pragma unroll
      for (int q = 0; q < NL; q ++) D[r][q] = __builtin_amdgcn_raw_buffer_load_b128(rV, lv + 16u * (unsigned)q, so, 0);
    }
    const bool x_edge = (int)g0 < glo || (int)(g0 + 63u) > ghi;  // (a chunk with lanes past the row's end: g0 + 63 > ngroups - 1 >= ghi)
    const bool k_dom = ND == 2 || ((int)k + m.ext_st[2] >= m.dom_lb[2] && (int)k + m.ext_st[2] <= m.dom_ub[2]);
    unsigned wr[4];
#pragma unroll
    for (int r = 0; r < 4; r ++) {
      const unsigned j = j0 + (unsigned)r;
      if (!(j < DH)) { wr[r] = 0x3f3f3f3fu; continue; }        // (wave-uniform) no such row: neutral for the block's summary
      double v[NV];
#pragma unroll
      for (int q = 0; q < NL; q ++) { const v2d t = __builtin_bit_cast(v2d, D[r][q]); v[2 * q] = t.x; v[2 * q + 1] = t.y; }
      // strict / sign bits of the lane's four vertices, vertex 3 first: byte q of S and G = vertex q
      unsigned S = 0, G = 0;
#pragma unroll
      for (int q = 3; q >= 0; q --) {
        if (q < 3) { S <<= 8 - ND; G <<= 8 - ND; }
        push_vertex_signs<ND>(S, G, &v[q * ND], thr);
      }
      const unsigned neg = S & G;
      unsigned word = (S ^ neg) | (neg << 3);
      double mxl = max_abs2(v[0], v[1]);
#pragma unroll
      for (int q = 2; q < NV; q ++) mxl = max_with_abs(mxl, v[q]);
      bool odd = mxl >= job.big;                               // (a NaN is ignored by the maximum: the unordered compares find it)
#pragma unroll
      for (int q = 0; q < NV; q += 2) odd = odd || __builtin_isunordered(v[q], v[q + 1]);
      const bool row_dom = k_dom && (int)j + m.ext_st[1] >= m.dom_lb[1] && (int)j + m.ext_st[1] <= m.dom_ub[1];
      const bool cand = (S & FULL) != FULL;                    // a component without a strict sign: a candidate for the slice's resolution (rare on smooth data)
      if (x_edge || !row_dom || __builtin_amdgcn_ballot_w64(odd)) {   // the vertices one by one, as mask_vec_kernel does
        asm volatile("" ::: "memory");
        const unsigned g = g0 + lane;
        word = 0;
        for (int q = 0; q < 4; q ++) {
          unsigned bits = 0;
          double mx = 0.0;
          bool fin = true;
          for (int c = 0; c < ND; c ++) {
            const double x = v[q * ND + c], a = fabs(x);
            if (x >= thr) bits |= 1u << c;
            if (x <= -thr) bits |= 8u << c;
            fin = fin && a < HUGE_VAL;
            mx = fmax(mx, a);
          }
          red_mx = lane_live ? fmax(red_mx, mx) : red_mx;
          const int x = 4 * (int)g + q + m.ext_st[0];
          const bool dom = row_dom && x >= m.dom_lb[0] && x <= m.dom_ub[0];
          if (mx >= job.big) bits = 0u;
          if (!fin || !dom) bits = kNeutral;
          word |= bits << (8 * q);
        }
        if (!lane_live) word = 0x3f3f3f3fu;
        if (__builtin_amdgcn_ballot_w64(cand && lane_live)) {
          if (lane_live)
            for (int q = 0; q < NV; q ++) { const double a = fabs(v[q]); red_mn = fmin(red_mn, (a == 0.0 || !(a < thr)) ? DBL_MAX : a); }
        }
      } else {
        red_mx = max_plain(red_mx, mxl);
        if (__builtin_amdgcn_ballot_w64(cand))
          for (int q = 0; q < NV; q ++) { const double a = fabs(v[q]); red_mn = fmin(red_mn, (a == 0.0 || !(a < thr)) ? DBL_MAX : a); }
      }
      wr[r] = word;
    }
    // the block's summary: the sign bits all 8 x 4 vertices share; its mask words are stored where there is none
    bool uniform = false;
    if (have_u) {
      const unsigned q8 = word_summary4((wr[0] & wr[1]) & (wr[2] & wr[3]));
      if (lane_live && u_lane) __builtin_amdgcn_raw_buffer_store_b8((unsigned char)q8, rU, lu, (k * nby + jb) * UP + (g0 >> 1), 0);
      uniform = q8 != 0;
    }
    if (lane_live && !uniform) {
#pragma unroll
      for (int r = 0; r < 4; r ++)
        if (j0 + (unsigned)r < DH) __builtin_amdgcn_raw_buffer_store_b32(wr[r], rM, lm, (k * DH + j0 + (unsigned)r) * P + g0 * 4u, 0);
    }
    // the next unit (scalar unit)
    u += W; ch += dch; jb += djb; k += dk;
    if (ch >= nch) { ch -= nch; jb ++; }
    if (jb >= nby) { jb -= nby; k ++; }
  }
  if (job.red) red_commit(job.red, red_mn, red_mx, blockIdx.x * 5u + (threadIdx.x >> 6));
}

// PD = prefetch distance: at the step for plane k the loads of plane k + 1 + PD are issued (PD = 1: three planes of registers
// plus one in flight).  RY = rows per wavefront: a plane costs (RY + 2) * 4 + 2 VGPRs; 3D runs RY = 4 at three wavefronts per
// SIMD or RY = 8 at two (fewer halo rows per useful row, and 6 of 10 row loads are private to the wavefront).
// LEAN (the 2D mask kernel, round 4): the kernel issues instructions 72 % of its time -- the (D - 1) scaling of gradient2D is folded
// into exact thresholds of the unscaled differences (4 multiplications per row less; the fused maximum is kept per component and scaled once at
// the end: max fl(|d| f) = fl(max |d| f), rounding is monotone) and the sign bits are shifted in at their final places (4 instructions per row
// less): woven 1024^2 x 64 0.099 -> 0.097 ms, bit-identical masks, reductions and records
template <int ND, bool REDUCE, int PD, int RY>
__global__ __launch_bounds__((ND == 2 || (PD == 1 && RY <= 4)) ? 768 : 512) void mask_march4_kernel(const Mesh m, const MaskJob *__restrict__ jobs, int zchunk, int swizzle)
{
  static_assert(RY >= 2 && RY <= 8, "row flags are 8-bit sets; the edge register holds 2 RY <= 16 values");
  constexpr bool LEAN = ND == 2 && !REDUCE;                    // the 2D mask kernel (the exact stand-alone reduction keeps the scaled values)
  constexpr int NB = 3 + PD;                                   // plane buffers: k-1, k, k+1, and PD planes on their way
  const int DW = m.ext_sz[0], DH = m.ext_sz[1], DD = (ND == 3) ? m.ext_sz[2] : 1, P = m.mask_pitch;
  const int nzc = (ND == 3) ? (DD + zchunk - 1) / zchunk : 1;
  unsigned bx, by, bz;
  remap_block(swizzle, bx, by, bz);
  const MaskJob job = jobs[bz / nzc];
  const int z0 = (ND == 3) ? (int)(bz % nzc) * zchunk : 0;
  const int z1 = (ND == 3) ? (z0 + zchunk < DD ? z0 + zchunk : DD) : 1;
  const int lane = threadIdx.x & 63;
  const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));   // wave-uniform by construction; tell the compiler
  const int wpb = blockDim.x >> 6;
  const int i0 = (int)bx * 128 + 2 * lane;                     // this lane's columns i0, i0 + 1
  const int j0 = ((int)by * wpb + wv) * RY;
  if (j0 >= DH) return;
  // Buffer resources: address = wave-uniform SGPR offset + one constant per-lane VGPR offset, no VALU address arithmetic.
  // (march2_supported guarantees slices below 4 GiB; the mask and summary arrays are smaller still.)
  const unsigned sy = (unsigned)DW * 8u, sz = (unsigned)DW * (unsigned)DH * 8u;   // byte strides
  const __amdgpu_buffer_rsrc_t rS = __builtin_amdgcn_make_buffer_rsrc((void *)job.S, 0, (int)(sz * (unsigned)DD), 0x00020000);
  const __amdgpu_buffer_rsrc_t rM = __builtin_amdgcn_make_buffer_rsrc((void *)job.M, 0, (int)((unsigned)P * (unsigned)DH * (unsigned)DD), 0x00020000);
  const __amdgpu_buffer_rsrc_t rU = __builtin_amdgcn_make_buffer_rsrc((void *)job.U, 0, (int)((unsigned)m.u_pitch * (unsigned)DH * (unsigned)DD), 0x00020000);
  const bool have_u = job.U != nullptr;
  const bool block4 = ND == 2 && RY % 4 == 0 && have_u && m.u_rows == 4;   // (wave-uniform) one summary byte per 8 x 4 block
  const double thr = job.threshold;
  // what the raw central difference (a - b) is compared with: 3D g = 0.5 (a - b); 2D g = (a - b) (D - 1) keeps its multiply
  const double tpos = (ND == 3) ? 2.0 * thr : thr, tneg = -tpos;

  const int ic = i0 < DW ? i0 : DW - 2;                        // clamped (even) column pair: lanes beyond the row load valid memory
  const unsigned cb = (unsigned)ic * 8u;                       // the only per-lane part of a row load's address
  // edge register: which (row, side) this lane fetches, as a byte offset inside a plane; every other lane carries an offset
  // beyond num_records, which a buffer load answers with 0 without touching memory (no branch around the load: see step)
  unsigned xoff = 0xfffffff0u;
  {
    const int t0c = (int)bx * 128;                             // the tile's first column
    const int xr = lane < RY ? lane : (lane >= 64 - RY ? lane - (64 - RY) : -1);
    if (xr >= 0) {
      const int col = lane < RY ? (t0c > 0 ? t0c - 1 : 0) : (t0c + 128 < DW ? t0c + 128 : DW - 1);
      xoff = sy * (unsigned)clampi(j0 + xr, 0, DH - 1) + (unsigned)col * 8u;
    }
  }
  unsigned xkeep = 0, xneutral = 0;                            // per column: byte c of the pair
  for (int c = 0; c < 2; c ++) {
    const int i = i0 + c;
    const bool x_dom = i < DW && i + m.ext_st[0] >= m.dom_lb[0] && i + m.ext_st[0] <= m.dom_ub[0];
    const bool x_int = (ND == 2) || (i >= 1 && i < DW - 1);
    if (x_int) xkeep |= 0x3fu << (8 * c);
    if (!x_dom) xneutral |= 0x3fu << (8 * c);
  }
  unsigned roff[RY + 2];                                       // wave-uniform row offsets, rows -1 .. RY (clamped)
  unsigned row_dom = 0, row_int = 0, row_ok = 0;
  for (int r = 0; r < RY + 2; r ++) {
    const int j = j0 + r - 1;
    roff[r] = sy * (unsigned)clampi(j, 0, DH - 1);
    if (r >= 1 && r <= RY) {
      if (j < DH) row_ok |= 1u << (r - 1);
      if (j + m.ext_st[1] >= m.dom_lb[1] && j + m.ext_st[1] <= m.dom_ub[1]) row_dom |= 1u << (r - 1);
      if (j >= 1 && j < DH - 1) row_int |= 1u << (r - 1);
    }
  }
  const int nt_mode = (swizzle & 4) ? 1 : ((swizzle & 16) ? 2 : 0);   // 1: all loads nontemporal, 2: only rows no other wavefront reads
  auto load_plane = [&](v2d (&B)[RY + 2], double &X, int k) {
    const unsigned zo = sz * (unsigned)clampi(k, 0, DD - 1);
    for (int r = 0; r < RY + 2; r ++) {
      const bool nt = nt_mode == 1 || (nt_mode == 2 && r >= 2 && r <= RY - 1);
      const v4u raw = nt ? __builtin_amdgcn_raw_buffer_load_b128(rS, cb, zo + roff[r], 2) : __builtin_amdgcn_raw_buffer_load_b128(rS, cb, zo + roff[r], 0);
      B[r] = __builtin_bit_cast(v2d, raw);
    }
    X = __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(rS, xoff, zo, 0));
  };

  const bool in_row = i0 < DW;
  const bool store_ok = in_row;
  const unsigned mcol = (unsigned)i0, ucol = (unsigned)(i0 >> 3);
  const bool u_lane = (lane & 3) == 0 && in_row;
  double red_mn = DBL_MAX, red_mx = 0.0;
  auto red_take = [&](double g) {
    const double a = fabs(g);
    red_mn = fmin(red_mn, a == 0.0 ? DBL_MAX : a);
    red_mx = fmax(red_mx, a < HUGE_VAL ? a : 0.0);
  };
  // fused pre-pass (mask instantiations): see guard_and_reduce
  double acc0 = 0.0, acc1 = 0.0;
  double accy0 = 0.0, accy1 = 0.0;                             // LEAN: acc* hold max |dx|, accy* max |dy|, unscaled
  const unsigned cmask = store_ok ? (xkeep & (ND == 3 ? 0x0707u : 0x0303u)) : 0u;
  const double tbig = (ND == 3) ? 2.0 * job.big : job.big;
  const bool per_vertex_rule = job.big < HUGE_VAL;
  const double fx2 = (double)(DW - 1), fy2 = (double)(DH - 1);
  // (the job carries them where the host set them -- set_lean_thresholds --: no division per wavefront)
  const double tpx = LEAN ? (job.tx > 0.0 ? job.tx : exact_threshold(thr, fx2)) : 0.0, tpy = LEAN ? (job.ty > 0.0 ? job.ty : exact_threshold(thr, fy2)) : 0.0;

  // one plane: prefetch plane k + 1 + PD into NN / XNN, classify plane k from (PR = k-1, CU = k, NX = k+1) and CU's edge register XC
  auto step = [&](const v2d (&PR)[RY + 2], const v2d (&CU)[RY + 2], const v2d (&NX)[RY + 2], v2d (&NN)[RY + 2], const double XC, double &XNN, int k) {
    // Unconditional, and no branch around any load in this loop (the last steps of a chunk re-request plane z1, an L2 hit):
    // on a path that skips loads the compiler's s_waitcnt bookkeeping has to assume the fewest loads in flight, and it then
    // waits for the prefetch itself (s_waitcnt vmcnt(0) per plane) -- measured, not hypothetical.
    if (ND == 3) load_plane(NN, XNN, k + 1 + PD < z1 ? k + 1 + PD : z1);
    const bool z_dom = ND == 2 || (k + m.ext_st[2] >= m.dom_lb[2] && k + m.ext_st[2] <= m.dom_ub[2]);
    const bool z_int = ND == 2 || (k >= 1 && k < DD - 1);
    const unsigned mplane = (unsigned)P * ((unsigned)j0 + (unsigned)DH * (unsigned)k);
    const unsigned uplane = (unsigned)m.u_pitch * ((unsigned)j0 + (unsigned)DH * (unsigned)k);
    unsigned bw[RY];                                           // (block summaries) the rows' pairs of mask bytes
    static_for<RY>([&](auto rc) {
      constexpr int r = decltype(rc)::value;
      const v2d c = CU[r + 1];
      const double xe = edge_for_row<r, RY>(XC);               // lane 0: left neighbour of row r, lane 63: right neighbour
      double xm = dpp_lower_or(c.y, xe);                       // left neighbour of column i0
      double xp = dpp_upper_or(c.x, xe);                       // right neighbour of column i0 + 1
      if (ND == 2 && i0 + 1 == DW - 1) xp = c.y;              // 2D clamp: the right neighbour of the last column is itself
      double dx0 = c.y - xm, dx1 = xp - c.x;
      double dy0 = CU[r + 2].x - CU[r].x, dy1 = CU[r + 2].y - CU[r].y;
      double dz0 = 0.0, dz1 = 0.0;
      if constexpr (ND == 3) { dz0 = NX[r + 1].x - PR[r + 1].x; dz1 = NX[r + 1].y - PR[r + 1].y; }
      else if constexpr (!LEAN) { const double fx = (double)(DW - 1), fy = (double)(DH - 1); dx0 *= fx; dx1 *= fx; dy0 *= fy; dy1 *= fy; }
      const bool u_int = ND == 2 || (((row_int >> r) & 1) && z_int), u_dom = ((row_dom >> r) & 1) && z_dom;
      if constexpr (REDUCE) {
        if (u_int && ((row_ok >> r) & 1) && store_ok && k < z1) {
          const double h = (ND == 3) ? 0.5 : 1.0;
          if (xkeep & 0x3fu) { red_take(h * dx0); red_take(h * dy0); if (ND == 3) red_take(h * dz0); }
          if (xkeep & 0x3f00u) { red_take(h * dx1); red_take(h * dy1); if (ND == 3) red_take(h * dz1); }
        }
        return;
      }
      unsigned a0 = 0, a1 = 0;
      const bool rok = ((row_ok >> r) & 1) && k < z1;
      unsigned bits;
      if constexpr (LEAN) {
        shift_in_signs2(a0, a1, dx0, dx1, dy0, dy1, -tpx, tpx, -tpy, tpy);
        bits = a0 | (a1 << 8);
        if (u_int && rok) {                                    // wave-uniform (guard_and_reduce, on the unscaled differences)
          acc0 = max_with_abs(acc0, dx0); acc1 = max_with_abs(acc1, dx1);
          accy0 = max_with_abs(accy0, dy0); accy1 = max_with_abs(accy1, dy1);
          const unsigned u = bits | (bits >> 3);
          if (__builtin_amdgcn_ballot_w64((~u & cmask) != 0u)) {
            auto take = [&](double g) { const double a = fabs(g); red_mn = fmin(red_mn, a == 0.0 ? DBL_MAX : a); };
            if (cmask & 0x00ffu) { take(dx0 * fx2); take(dy0 * fy2); }
            if (cmask & 0xff00u) { take(dx1 * fx2); take(dy1 * fy2); }
          }
        }
        if (per_vertex_rule) {                                 // wave-uniform, rare: the scaled magnitudes after all
          asm volatile("" ::: "memory");
          const double m0 = max_abs2(dx0 * fx2, dy0 * fy2), m1 = max_abs2(dx1 * fx2, dy1 * fy2);
          bits = (m0 >= tbig ? 0u : (bits & 0x00ffu)) | (m1 >= tbig ? 0u : (bits & 0xff00u));
        }
      } else {
      shift_in_signs<ND>(a0, a1, dx0, dx1, dy0, dy1, dz0, dz1, tneg, tpos);
      if (ND == 2) { a0 = ((a0 & 0xcu) << 1) | (a0 & 3u); a1 = ((a1 & 0xcu) << 1) | (a1 & 3u); }   // leave the two z bits empty
      bits = guard_and_reduce<ND>(a0, a1, dx0, dx1, dy0, dy1, dz0, dz1, u_int && rok, cmask, per_vertex_rule, tbig, ND == 3 ? 0.5 : 1.0, acc0, acc1, red_mn);
      }
      // wave-uniform row / plane conditions, per-lane column conditions
      const unsigned keep = u_int ? xkeep : 0u;                // gradient3D leaves the array border at 0
      const unsigned neut = u_dom ? xneutral : 0x3f3fu;        // outside the domain / row padding: never blocks a cull
      bits = (bits & keep) | neut;
      if (block4) { bw[r] = rok ? bits : 0x3f3fu; return; }   // (wave-uniform) summaries per 8 x 4 block: below, once the block's rows are known
      bool word_uniform = false;
      // summary byte of the aligned 8-vertex word this quad of lanes covers: the sign bits ALL eight vertices share
      if (have_u) {
        int q = (int)((bits & (bits >> 8)) & 0x3fu);
        q &= __builtin_amdgcn_update_dpp(q, q, 0xb1 /* quad_perm:[1,0,3,2] */, 0xf, 0xf, false);
        q &= __builtin_amdgcn_update_dpp(q, q, 0x4e /* quad_perm:[2,3,0,1] */, 0xf, 0xf, false);
        if (rok && u_lane) __builtin_amdgcn_raw_buffer_store_b8((unsigned char)q, rU, ucol, uplane + (unsigned)m.u_pitch * (unsigned)r, 0);
        word_uniform = q != 0;                                 // then the refine kernel substitutes the summary: mask bytes not needed
      }
      if (rok && store_ok && !word_uniform)
        __builtin_amdgcn_raw_buffer_store_b16((unsigned short)bits, rM, mcol, mplane + (unsigned)P * (unsigned)r, 0);
    });
    // One summary byte per 8 x 4 block (Mesh::u_rows == 4; 2D): the sign bits all 32 vertices share -- a quarter of the summary bytes, and
    // every byte of them costs (the summary stores).  The mask words of a block are stored where it has no such bit.
    if constexpr (ND == 2 && RY % 4 == 0 && !REDUCE) if (block4) {
      static_for<RY / 4>([&](auto bc) {
        constexpr int b = decltype(bc)::value;
        const unsigned all = (bw[4 * b] & bw[4 * b + 1]) & (bw[4 * b + 2] & bw[4 * b + 3]);
        int q = (int)((all & (all >> 8)) & 0x3fu);
        q &= __builtin_amdgcn_update_dpp(q, q, 0xb1 /* quad_perm:[1,0,3,2] */, 0xf, 0xf, false);
        q &= __builtin_amdgcn_update_dpp(q, q, 0x4e /* quad_perm:[2,3,0,1] */, 0xf, 0xf, false);
        const bool first_ok = ((row_ok >> (4 * b)) & 1) && k < z1;
        if (first_ok && u_lane) __builtin_amdgcn_raw_buffer_store_b8((unsigned char)q, rU, ucol, (unsigned)m.u_pitch * ((unsigned)(j0 / 4 + b) + (unsigned)((DH + 3) / 4) * (unsigned)k), 0);
        if (q == 0 && store_ok) {
#pragma unroll
          for (int r = 4 * b; r < 4 * b + 4; r ++)
            if (((row_ok >> r) & 1) && k < z1) __builtin_amdgcn_raw_buffer_store_b16((unsigned short)bw[r], rM, mcol, mplane + (unsigned)P * (unsigned)r, 0);
        }
      });
    }
  };

  // the plane buffers rotate by NAME: the z loop is unrolled NB times, every index below is a compile-time constant
  v2d B[NB][RY + 2];
  double X[NB];
  for (int b = 0; b < NB; b ++) X[b] = 0.0;
  load_plane(B[0], X[0], z0 - 1);
  load_plane(B[1], X[1], z0);
  if (ND == 3) { for (int b = 2; b < 2 + PD; b ++) load_plane(B[b], X[b], z0 + b - 1 < z1 ? z0 + b - 1 : z1); }
  else for (int b = 2; b < NB; b ++) for (int r = 0; r < RY + 2; r ++) B[b][r] = B[1][r];
  for (int k = z0; k < z1; k += NB) {
#pragma unroll
    for (int i = 0; i < NB; i ++)   // no `if (k + i < z1)` around a step: planes past the chunk are walked without stores
      step(B[i], B[(i + 1) % NB], B[(i + 2) % NB], B[(i + 2 + PD) % NB], X[(i + 1) % NB], X[(i + 2 + PD) % NB], k + i);
  }
  const unsigned slot = blockIdx.x + blockIdx.y * 7u + blockIdx.z * 13u + (unsigned)wv;
  if constexpr (REDUCE) red_commit(job.red, red_mn, red_mx, slot);
  else if (job.red) {
    const double h = (ND == 3) ? 0.5 : 1.0;
    double mx = fmax((cmask & 0x00ffu) ? acc0 : 0.0, (cmask & 0xff00u) ? acc1 : 0.0) * h;
    if constexpr (LEAN) {                                      // the maxima were kept per component, unscaled: fl(max |d| f) = max fl(|d| f)
      const double mxx = fmax((cmask & 0x00ffu) ? acc0 : 0.0, (cmask & 0xff00u) ? acc1 : 0.0) * fx2;
      const double mxy = fmax((cmask & 0x00ffu) ? accy0 : 0.0, (cmask & 0xff00u) ? accy1 : 0.0) * fy2;
      mx = max_plain(mxx, mxy);
    }
    red_commit(job.red, red_mn < job.threshold ? red_mn : DBL_MAX, mx, slot);
  }
}

// ---------------------------------------------------------------------------------------------------------------
// 2D scalar input, round 5: the arithmetic of mask_march4_kernel<2, false, 1, 8> by a wavefront that MARCHES DOWN y.
// In 2D that kernel's wavefront takes 8 rows x 128 columns -- 8 KB of input -- and leaves: ten row loads issued at once and all of them
// waited for before the first row is classified, one set-up and one pair of reduction atomics per 8 KB.  Here a wavefront takes `groups`
// consecutive groups of 8 rows of its 128 columns: the set-up once, the loads of group g + 1 on their way while group g is classified (two
// register sets that rotate by name).  Faster on WIDE slices only (launch_masks_impl has the numbers): most of the scalar instructions of
// either kernel are the per-row uniform tests inside the step, not the set-up.
// Same mask words, summaries and reductions as the kernel above, bit for bit (tests/test_gpu_properties.py::test_mask_kernel_generations_agree
// runs them against each other: FTKX_MASK_PLAN rows=0 takes the kernel above).
// ---------------------------------------------------------------------------------------------------------------
template <int RY>
__global__ __launch_bounds__(256) void mask_rows2_kernel(const Mesh m, const MaskJob *__restrict__ jobs, int groups, int swizzle)
{
  static_assert(RY == 8, "two 8 x 4 blocks per group; the edge register holds 2 RY <= 16 values");
  const int DW = m.ext_sz[0], DH = m.ext_sz[1], P = m.mask_pitch;
  unsigned bx, by, bz;
  remap_block(swizzle, bx, by, bz);
  const MaskJob job = jobs[bz];
  const int lane = threadIdx.x & 63;
  const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int wpb = blockDim.x >> 6;
  const int i0 = (int)bx * 128 + 2 * lane;                     // this lane's columns i0, i0 + 1
  const int jbase = ((int)by * wpb + wv) * RY * groups;        // this wavefront's rows: jbase .. jend - 1
  if (jbase >= DH) return;
  const int jend = jbase + RY * groups < DH ? jbase + RY * groups : DH;
  const unsigned sy = (unsigned)DW * 8u;
  const __amdgpu_buffer_rsrc_t rS = __builtin_amdgcn_make_buffer_rsrc((void *)job.S, 0, (int)(sy * (unsigned)DH), 0x00020000);
  const __amdgpu_buffer_rsrc_t rM = __builtin_amdgcn_make_buffer_rsrc((void *)job.M, 0, (int)((unsigned)P * (unsigned)DH), 0x00020000);
  const __amdgpu_buffer_rsrc_t rU = __builtin_amdgcn_make_buffer_rsrc((void *)job.U, 0, (int)((unsigned)m.u_pitch * (unsigned)DH), 0x00020000);
  const bool have_u = job.U != nullptr;
  const bool block4 = have_u && m.u_rows == 4;                 // (wave-uniform) one summary byte per 8 x 4 block
  const double thr = job.threshold;
  const int ic = i0 < DW ? i0 : DW - 2;                        // clamped (even) column pair: lanes beyond the row load valid memory
  const unsigned cb = (unsigned)ic * 8u;
  // edge register: lane r < RY fetches the left neighbour of row r's first column, lane 64 - RY + r the right neighbour of its last
  const int t0c = (int)bx * 128;
  const int xr = lane < RY ? lane : (lane >= 64 - RY ? lane - (64 - RY) : -1);
  const unsigned xcol8 = (unsigned)(lane < RY ? (t0c > 0 ? t0c - 1 : 0) : (t0c + 128 < DW ? t0c + 128 : DW - 1)) * 8u;
  unsigned xkeep = 0, xneutral = 0;                            // per column: byte c of the pair
  for (int c = 0; c < 2; c ++) {
    const int i = i0 + c;
    const bool x_dom = i < DW && i + m.ext_st[0] >= m.dom_lb[0] && i + m.ext_st[0] <= m.dom_ub[0];
    xkeep |= 0x3fu << (8 * c);
    if (!x_dom) xneutral |= 0x3fu << (8 * c);
  }
  const bool in_row = i0 < DW;
  const unsigned mcol = (unsigned)i0, ucol = (unsigned)(i0 >> 3);
  const bool u_lane = (lane & 3) == 0 && in_row;
  double red_mn = DBL_MAX;
  double acc0 = 0.0, acc1 = 0.0, accy0 = 0.0, accy1 = 0.0;     // max |dx|, max |dy| per column, unscaled (mask_march4_kernel: LEAN)
  const unsigned cmask = in_row ? (xkeep & 0x0303u) : 0u;
  const double tbig = job.big;
  const bool per_vertex_rule = job.big < HUGE_VAL;
  const double fx2 = (double)(DW - 1), fy2 = (double)(DH - 1);
  const double tpx = job.tx > 0.0 ? job.tx : exact_threshold(thr, fx2), tpy = job.ty > 0.0 ? job.ty : exact_threshold(thr, fy2);

  struct Rows { unsigned roff[RY + 2]; unsigned row_dom, row_ok, xoff, cbv; int j0; };
  // a group's wave-uniform row offsets and flags; a group past the wavefront's rows loads nothing (offsets beyond num_records: answered with 0
  // without touching memory -- no branch around a load) and stores nothing
  auto setup = [&](Rows &R, int j0) {
    R.j0 = j0;
    const bool live = j0 < jend;
    R.row_dom = 0; R.row_ok = 0;
    for (int r = 0; r < RY + 2; r ++) {
      const int j = j0 + r - 1;
      R.roff[r] = sy * (unsigned)clampi(j, 0, DH - 1);
      if (r >= 1 && r <= RY) {
        if (j < jend) R.row_ok |= 1u << (r - 1);
        if (j + m.ext_st[1] >= m.dom_lb[1] && j + m.ext_st[1] <= m.dom_ub[1]) R.row_dom |= 1u << (r - 1);
      }
    }
    R.cbv = live ? cb : 0xfffffff0u;
    R.xoff = (live && xr >= 0) ? sy * (unsigned)clampi(j0 + xr, 0, DH - 1) + xcol8 : 0xfffffff0u;
  };
  auto load = [&](v2d (&B)[RY + 2], double &X, const Rows &R) {
    for (int r = 0; r < RY + 2; r ++) B[r] = __builtin_bit_cast(v2d, __builtin_amdgcn_raw_buffer_load_b128(rS, R.cbv, R.roff[r], 0));
    X = __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(rS, R.xoff, 0, 0));
  };
  auto step = [&](const v2d (&CU)[RY + 2], const double XC, const Rows &R) {
    const unsigned mplane = (unsigned)P * (unsigned)R.j0, uplane = (unsigned)m.u_pitch * (unsigned)R.j0;
    unsigned bw[RY];
    static_for<RY>([&](auto rc) {
      constexpr int r = decltype(rc)::value;
      const v2d c = CU[r + 1];
      const double xe = edge_for_row<r, RY>(XC);               // lane 0: left neighbour of row r, lane 63: right neighbour
      const double xm = dpp_lower_or(c.y, xe);                 // left neighbour of column i0
      double xp = dpp_upper_or(c.x, xe);                       // right neighbour of column i0 + 1
      if (i0 + 1 == DW - 1) xp = c.y;                          // 2D clamp: the right neighbour of the last column is itself
      const double dx0 = c.y - xm, dx1 = xp - c.x;
      const double dy0 = CU[r + 2].x - CU[r].x, dy1 = CU[r + 2].y - CU[r].y;
      const bool rok = (R.row_ok >> r) & 1, u_dom = (R.row_dom >> r) & 1;
      unsigned a0 = 0, a1 = 0;
      shift_in_signs2(a0, a1, dx0, dx1, dy0, dy1, -tpx, tpx, -tpy, tpy);
      unsigned bits = a0 | (a1 << 8);
      if (rok) {                                               // wave-uniform
        acc0 = max_with_abs(acc0, dx0); acc1 = max_with_abs(acc1, dx1);
        accy0 = max_with_abs(accy0, dy0); accy1 = max_with_abs(accy1, dy1);
        const unsigned u = bits | (bits >> 3);
        if (__builtin_amdgcn_ballot_w64((~u & cmask) != 0u)) {
          auto take = [&](double g) { const double a = fabs(g); red_mn = fmin(red_mn, a == 0.0 ? DBL_MAX : a); };
          if (cmask & 0x00ffu) { take(dx0 * fx2); take(dy0 * fy2); }
          if (cmask & 0xff00u) { take(dx1 * fx2); take(dy1 * fy2); }
        }
      }
      if (per_vertex_rule) {                                   // wave-uniform, rare: the scaled magnitudes after all
        asm volatile("" ::: "memory");
        const double m0 = max_abs2(dx0 * fx2, dy0 * fy2), m1 = max_abs2(dx1 * fx2, dy1 * fy2);
        bits = (m0 >= tbig ? 0u : (bits & 0x00ffu)) | (m1 >= tbig ? 0u : (bits & 0xff00u));
      }
      const unsigned neut = u_dom ? xneutral : 0x3f3fu;        // outside the domain / row padding: never blocks a cull
      bits = (bits & xkeep) | neut;
      if (block4) { bw[r] = rok ? bits : 0x3f3fu; return; }
      bool word_uniform = false;
      if (have_u) {                                            // one summary byte per word of 8 (FTKX_U_ROWS=1)
        int q = (int)((bits & (bits >> 8)) & 0x3fu);
        q &= __builtin_amdgcn_update_dpp(q, q, 0xb1 /* quad_perm:[1,0,3,2] */, 0xf, 0xf, false);
        q &= __builtin_amdgcn_update_dpp(q, q, 0x4e /* quad_perm:[2,3,0,1] */, 0xf, 0xf, false);
        if (rok && u_lane) __builtin_amdgcn_raw_buffer_store_b8((unsigned char)q, rU, ucol, uplane + (unsigned)m.u_pitch * (unsigned)r, 0);
        word_uniform = q != 0;
      }
      if (rok && in_row && !word_uniform) __builtin_amdgcn_raw_buffer_store_b16((unsigned short)bits, rM, mcol, mplane + (unsigned)P * (unsigned)r, 0);
    });
    if (block4) {
      static_for<RY / 4>([&](auto bc) {
        constexpr int b = decltype(bc)::value;
        const unsigned all = (bw[4 * b] & bw[4 * b + 1]) & (bw[4 * b + 2] & bw[4 * b + 3]);
        int q = (int)((all & (all >> 8)) & 0x3fu);
        q &= __builtin_amdgcn_update_dpp(q, q, 0xb1 /* quad_perm:[1,0,3,2] */, 0xf, 0xf, false);
        q &= __builtin_amdgcn_update_dpp(q, q, 0x4e /* quad_perm:[2,3,0,1] */, 0xf, 0xf, false);
        const bool first_ok = (R.row_ok >> (4 * b)) & 1;
        if (first_ok && u_lane) __builtin_amdgcn_raw_buffer_store_b8((unsigned char)q, rU, ucol, (unsigned)m.u_pitch * (unsigned)(R.j0 / 4 + b), 0);
        if (q == 0 && in_row) {
#pragma unroll
          for (int r = 4 * b; r < 4 * b + 4; r ++)
            if ((R.row_ok >> r) & 1) __builtin_amdgcn_raw_buffer_store_b16((unsigned short)bw[r], rM, mcol, mplane + (unsigned)P * (unsigned)r, 0);
        }
      });
    }
  };

  Rows Ra, Rb;
  v2d A[RY + 2], B[RY + 2];
  double XA, XB;
  setup(Ra, jbase);
  load(A, XA, Ra);
  for (int g = 0; g < groups; g += 2) {
    setup(Rb, jbase + (g + 1) * RY);
    load(B, XB, Rb);
    step(A, XA, Ra);
    setup(Ra, jbase + (g + 2) * RY);
    load(A, XA, Ra);
    if (g + 1 < groups) step(B, XB, Rb);
  }
  if (job.red) {
    const double mxx = fmax((cmask & 0x00ffu) ? acc0 : 0.0, (cmask & 0xff00u) ? acc1 : 0.0) * fx2;   // fl(max |d| f) = max fl(|d| f)
    const double mxy = fmax((cmask & 0x00ffu) ? accy0 : 0.0, (cmask & 0xff00u) ? accy1 : 0.0) * fy2;
    red_commit(job.red, red_mn < job.threshold ? red_mn : DBL_MAX, max_plain(mxx, mxy), blockIdx.x + blockIdx.y * 7u + blockIdx.z * 13u + (unsigned)wv);
  }
}

// ---------------------------------------------------------------------------------------------------------------
// The same walk WITHOUT a dedicated producer: every wavefront of the workgroup is a consumer and issues the LDS-DMA loads of its
// own rows (plus one of the three odd jobs: the two halo rows and the edge values).  Why: a workgroup's wavefronts land on the
// CU's four SIMDs round-robin from a random start (tools/probe/simd_map.hip), so with 1 + 3 wavefronts per workgroup and three
// workgroups per CU some SIMD always carries three consumers -- a third of the CU's sign arithmetic instead of a quarter -- and
// the plane step waits for it.  Four symmetric wavefronts x three workgroups put exactly three equal shares on every SIMD.
// What that costs:
//   * a wavefront that issues LDS-DMA and also reads LDS gets an s_waitcnt vmcnt(0) from the compiler before every LDS read
//     (it cannot tell the slots apart), so the LDS reads are inline assembly with their own lgkmcnt wait;
//   * vmcnt counts this wavefront's mask stores too (gfx9: one in-order counter for loads and stores), so the stores are issued
//     unconditionally -- lanes / rows with nothing to store carry an out-of-range offset, which a buffer store drops -- and the
//     wait before the barrier names exactly how many younger operations may still be in flight.
// ---------------------------------------------------------------------------------------------------------------
// byte R of T = byte 0 & byte 1 of `bits` (R = 0 starts a new T)
template <int R> __device__ inline void pack_pair_and(unsigned &T, unsigned bits)
{
  if constexpr (R == 0) asm("v_and_b32_sdwa %0, %1, %1 dst_sel:BYTE_0 dst_unused:UNUSED_PAD src0_sel:BYTE_0 src1_sel:BYTE_1" : "=v"(T) : "v"(bits));
  else if constexpr (R == 1) asm("v_and_b32_sdwa %0, %1, %1 dst_sel:BYTE_1 dst_unused:UNUSED_PRESERVE src0_sel:BYTE_0 src1_sel:BYTE_1" : "+v"(T) : "v"(bits));
  else if constexpr (R == 2) asm("v_and_b32_sdwa %0, %1, %1 dst_sel:BYTE_2 dst_unused:UNUSED_PRESERVE src0_sel:BYTE_0 src1_sel:BYTE_1" : "+v"(T) : "v"(bits));
  else asm("v_and_b32_sdwa %0, %1, %1 dst_sel:BYTE_3 dst_unused:UNUSED_PRESERVE src0_sel:BYTE_0 src1_sel:BYTE_1" : "+v"(T) : "v"(bits));
}
// AND over the four lanes of every quad (s_nop 1: the two wait states a DPP read needs after a VALU write of its source)
__device__ inline void quad_and(unsigned &T)
{
  asm("s_nop 1\n\t"
      "v_and_b32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
      "s_nop 1\n\t"
      "v_and_b32_dpp %0, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf"
      : "+v"(T));
}
// OR over the four quads of every 16-lane row (lanes i, i+4, i+8, i+12): two rotate-and-OR steps
__device__ inline void row_quads_or(unsigned &x)
{
  asm("s_nop 1\n\t"
      "v_or_b32_dpp %0, %0, %0 row_ror:4 row_mask:0xf bank_mask:0xf\n\t"
      "s_nop 1\n\t"
      "v_or_b32_dpp %0, %0, %0 row_ror:8 row_mask:0xf bank_mask:0xf"
      : "+v"(x));
}
// b in the lanes of `mask`, a elsewhere
__device__ inline unsigned select_lanes(unsigned a, unsigned b, unsigned long long mask)
{
  unsigned r;
  asm("v_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "s"(mask));
  return r;
}
// lane mask: byte R of T is not zero
template <int R> __device__ inline void byte_nonzero(unsigned long long &mask, unsigned T, unsigned zero)
{
  if constexpr (R == 0) asm volatile("v_cmp_ne_u32_sdwa %0, %1, %2 src0_sel:BYTE_0 src1_sel:DWORD" : "=s"(mask) : "v"(T), "v"(zero));
  else if constexpr (R == 1) asm volatile("v_cmp_ne_u32_sdwa %0, %1, %2 src0_sel:BYTE_1 src1_sel:DWORD" : "=s"(mask) : "v"(T), "v"(zero));
  else if constexpr (R == 2) asm volatile("v_cmp_ne_u32_sdwa %0, %1, %2 src0_sel:BYTE_2 src1_sel:DWORD" : "=s"(mask) : "v"(T), "v"(zero));
  else asm volatile("v_cmp_ne_u32_sdwa %0, %1, %2 src0_sel:BYTE_3 src1_sel:DWORD" : "=s"(mask) : "v"(T), "v"(zero));
}
// off, or an offset beyond any buffer (-16 = 0xfffffff0) in the lanes of `mask`
__device__ inline unsigned out_of_range_where(unsigned off, unsigned long long mask)
{
  unsigned r;
  asm volatile("v_cndmask_b32_e64 %0, %1, -16, %2" : "=v"(r) : "v"(off), "s"(mask));
  return r;
}
template <int I> __device__ inline void lds_read128(v2d &d, unsigned a) { asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(d) : "v"(a), "n"(I * 1024)); }
template <int I> __device__ inline void lds_read64(double &d, unsigned a) { asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(d) : "v"(a), "n"(I)); }
__device__ inline void lds_tie(v2d &d) { asm volatile("" : "+v"(d)); }
__device__ inline void lds_tie(double &d) { asm volatile("" : "+v"(d)); }
template <int N, int... I> __device__ inline void lds_rows(v2d (&B)[N], unsigned a, std::integer_sequence<int, I...>) { (lds_read128<I>(B[I], a), ...); }
template <int N, int... I> __device__ inline void lds_tie_rows(v2d (&B)[N], std::integer_sequence<int, I...>) { (lds_tie(B[I]), ...); }
template <int RIGHT, int N, int... I> __device__ inline void lds_edges(double (&XL)[N], double (&XR)[N], unsigned a, std::integer_sequence<int, I...>)
{
  (lds_read64<8 * I>(XL[I], a), ...);
  (lds_read64<RIGHT + 8 * I>(XR[I], a), ...);
}
template <int N, int... I> __device__ inline void lds_tie_edges(double (&XL)[N], double (&XR)[N], std::integer_sequence<int, I...>) { ((lds_tie(XL[I]), lds_tie(XR[I])), ...); }

// mask_march6_kernel, step s of a chunk: wait until this wavefront's loads of the plane the step needs have landed.  vmcnt counts
// loads and stores in issue order (gfx9), so the wait names how many YOUNGER operations may still be in flight: the loads of that
// plane were issued NS steps earlier; since then the wavefront issued the S stores of that step and, per later step, at least RY
// loads and exactly S stores.  The first NS steps wait for loads the prologue issued back to back, with no stores in between.
template <int NS, int RY, int S>
__device__ inline void wait_plane_landed(int s)
{
  static_assert((NS - 1) * RY + NS * S < 64 && NS <= 4, "vmcnt is a 6-bit counter; four early steps are spelled out");
#define FTKX_WAIT_VM(N) __builtin_amdgcn_s_waitcnt(((N) & 15) | (7 << 4) | (15 << 8) | (((N) >> 4) << 14))
  if (s >= NS) FTKX_WAIT_VM((NS - 1) * RY + NS * S);
  else if (s == 1) FTKX_WAIT_VM((NS - 1) * RY + S);
  else if (s == 2) FTKX_WAIT_VM((NS - 1) * RY + 2 * S);
  else if (s == 3) FTKX_WAIT_VM((NS - 1) * RY + 3 * S);
  else FTKX_WAIT_VM((NS - 1) * RY);
#undef FTKX_WAIT_VM
}

// The z extent of a slice in PIECES of unequal length, the same for every tile column: blockIdx.z = piece * njobs + slice, pieces in
// order of decreasing length.  The hardware hands workgroups out in order of their index as slots free up -- a queue --, so the long
// pieces (up to half a column) go out first and the launch ends on pieces of a few planes: the device stays full until a few
// microseconds before the end whatever the size of the series (equal chunks of 32 planes: 256^3 x 16 is 5.3 rounds of workgroups, a
// 512^3 slice on its own 2.7), and the two start-up planes of a march are paid per piece -- 7 pieces per 512 planes on 512^3 x 32
// instead of 16 chunks.  (Persistent workgroups pulling such pieces from a queue of their own were built and measured: the loop state
// costs the kernel 36 more SGPR spills and 12 VGPRs of scratch at three wavefronts per SIMD -- 7.6 ms against 5.7 on 512^3 x 32.)
// Equal chunks (launches of a dozen rounds and more) go out slice by slice instead: neighbouring chunks of a slice then run at the same
// time and find each other's start-up planes in the caches.
struct ZPlan {
  unsigned npieces;
  unsigned z0[47], len[47];
};

template <int NS, int CY, int RY, bool TWOB>
__device__ __forceinline__ void march6_body(const Mesh &m, const MaskJob *__restrict__ jobs, int swizzle, int njobs, const ZPlan &plan)
{
#if defined(__HIP_DEVICE_COMPILE__)
  constexpr int ROWS = RY * CY, TROWS = ROWS + 2, NE = NS + 1;             // NS row slots, NS + 1 edge entries
  // edge values of a plane: up to 16 rows -> one wavefront instruction (left neighbours at 8 e, right ones at 128 + 8 e of the entry);
  // up to 32 rows -> the left ones by wavefront 1, the right ones by wavefront 2 (at 256 + 8 e)
  constexpr bool EDGE2 = ROWS > 16;
  static_assert(ROWS <= 32 && CY >= (EDGE2 ? 4 : 3), "the odd jobs (two halo rows, edge values) go to different wavefronts");
  constexpr unsigned ROWB = 1024u, SLOT = TROWS * ROWB, EDGEB = EDGE2 ? 512u : 256u, ERIGHT = EDGE2 ? 256u : 128u, ERING = NS * SLOT;
  // TWOB: a row slot is refilled in the very step that reads it (second barrier below) -- NS planes are on their way or waiting at
  // any time; otherwise one step later, after the next step's only barrier (NS - 1 planes).  The wait before the first barrier:
  // wait_plane_landed.
  constexpr int DEPTH = TWOB ? NS : NS - 1;
  static_assert(DEPTH >= 1, "at least one plane on its way");
  extern __shared__ __attribute__((aligned(16))) char lds[];
  const int DW = m.ext_sz[0], DH = m.ext_sz[1], DD = m.ext_sz[2], P = m.mask_pitch;
  unsigned bx, by, bz;
  remap_block(swizzle, bx, by, bz);
  // (njobs < 0: slice-major -- blockIdx.z = slice * npieces + piece, the pieces of a slice back to back)
  const unsigned nj = (unsigned)(njobs < 0 ? -njobs : njobs);
  const unsigned piece = njobs < 0 ? bz % plan.npieces : bz / nj;
  const MaskJob job = jobs[njobs < 0 ? bz / plan.npieces : bz - piece * nj];
  const int z0 = (int)plan.z0[piece];
  const int z1 = z0 + (int)plan.len[piece];
  const int lane = threadIdx.x & 63;
  const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int jb = (int)by * ROWS, t0c = (int)bx * 128;
  if (jb >= DH) return;
  const unsigned sy = (unsigned)DW * 8u, sz = (unsigned)DW * (unsigned)DH * 8u;
  const __amdgpu_buffer_rsrc_t rS = __builtin_amdgcn_make_buffer_rsrc((void *)job.S, 0, (int)(sz * (unsigned)DD), 0x00020000);
  const int nsteps = ((z1 - z0 + 2) / 3) * 3;
  auto slot_of = [&](int q) -> unsigned { return (unsigned)(((q - (z0 - 1)) % NS + NS) % NS) * SLOT; };
  auto edge_of = [&](int q) -> unsigned { return ERING + (unsigned)(((q - (z0 - 1)) % NE + NE) % NE) * EDGEB; };

  // ---- this wavefront's share of a plane's loads ----
  const int ip = t0c + 2 * lane;
  const unsigned cb = (unsigned)(ip < DW ? ip : DW - 2) * 8u;
  unsigned eoff = 0xfffffff0u;
  if (wv == 1 || (EDGE2 && wv == 2)) {
    const int e = lane >> 1, row = EDGE2 ? e : (e & 15);
    const bool left = EDGE2 ? wv == 1 : e < 16;
    if (row < ROWS) {
      const int col = left ? (t0c > 0 ? t0c - 1 : 0) : (t0c + 128 < DW ? t0c + 128 : DW - 1);
      eoff = sy * (unsigned)clampi(jb + row, 0, DH - 1) + (unsigned)col * 8u + 4u * (unsigned)(lane & 1);
    }
  }
  unsigned roff[RY];                                            // tile rows 1 + wv RY + r: this wavefront's own rows
  for (int r = 0; r < RY; r ++) roff[r] = sy * (unsigned)clampi(jb + wv * RY + r, 0, DH - 1);
  const int xt = wv == 0 ? 0 : TROWS - 1;                       // the halo row wavefront 0 / CY-1 also fetches
  const unsigned xoff = sy * (unsigned)clampi(jb + xt - 1, 0, DH - 1);
  auto issue = [&](int q) {
    const unsigned zo = sz * (unsigned)clampi(q, 0, DD - 1);
    const unsigned base = slot_of(q);
    for (int r = 0; r < RY; r ++) {
      __attribute__((address_space(3))) void *dst = (__attribute__((address_space(3))) void *)(lds + base + (unsigned)(1 + wv * RY + r) * ROWB);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rS, dst, 16, cb, zo + roff[r], 0, 0);
    }
    if (wv == 0 || wv == CY - 1) {
      __attribute__((address_space(3))) void *dst = (__attribute__((address_space(3))) void *)(lds + base + (unsigned)xt * ROWB);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rS, dst, 16, cb, zo + xoff, 0, 0);
    } else if (wv == 1 || (EDGE2 && wv == 2)) {
      __attribute__((address_space(3))) void *edst = (__attribute__((address_space(3))) void *)(lds + edge_of(q) + ((EDGE2 && wv == 2) ? ERIGHT : 0u));
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rS, edst, 4, eoff, zo, 0, 0);
    }
  };
#define FTKX_WAIT_VM(N) __builtin_amdgcn_s_waitcnt(((N) & 15) | (7 << 4) | (15 << 8) | (((N) >> 4) << 14))

  // ---- what the classification of a plane needs ----
  const int wy = wv;
  const int i0 = t0c + 2 * lane;
  const __amdgpu_buffer_rsrc_t rM = __builtin_amdgcn_make_buffer_rsrc((void *)job.M, 0, (int)((unsigned)P * (unsigned)DH * (unsigned)DD), 0x00020000);
  const __amdgpu_buffer_rsrc_t rU = __builtin_amdgcn_make_buffer_rsrc((void *)job.U, 0, job.U ? (int)((unsigned)m.u_pitch * (unsigned)DH * (unsigned)DD) : 0, 0x00020000);
  const double thr = job.threshold;
  const double tpos = 2.0 * thr, tneg = -tpos;
  const int j0 = jb + wy * RY;
  unsigned xkeep = 0, xneutral = 0;
  for (int c = 0; c < 2; c ++) {
    const int i = i0 + c;
    const bool x_dom = i < DW && i + m.ext_st[0] >= m.dom_lb[0] && i + m.ext_st[0] <= m.dom_ub[0];
    const bool x_int = i >= 1 && i < DW - 1;
    if (x_int) xkeep |= 0x3fu << (8 * c);
    if (!x_dom) xneutral |= 0x3fu << (8 * c);
  }
  unsigned row_dom = 0, row_int = 0, row_ok = 0;
  for (int r = 0; r < RY; r ++) {
    const int j = j0 + r;
    if (j < DH) row_ok |= 1u << r;
    if (j + m.ext_st[1] >= m.dom_lb[1] && j + m.ext_st[1] <= m.dom_ub[1]) row_dom |= 1u << r;
    if (j >= 1 && j < DH - 1) row_int |= 1u << r;
  }
  const bool in_row = i0 < DW;
  constexpr unsigned OOB = 0xfffffff0u;                          // beyond num_records: the store is dropped
  const bool have_u = job.U != nullptr;
  const unsigned mcol = in_row ? (unsigned)i0 : OOB;
  const unsigned ucol = ((lane & 3) == 0 && in_row && have_u) ? (unsigned)(i0 >> 3) : OOB;
  double acc0 = 0.0, acc1 = 0.0, red_mn = DBL_MAX;
  const unsigned cmask = in_row ? (xkeep & 0x0707u) : 0u;
  const double tbig = 2.0 * job.big;
  const bool per_vertex_rule = job.big < HUGE_VAL;
  const unsigned lds0 = (unsigned)(__UINTPTR_TYPE__)(__attribute__((address_space(3))) char *)lds;
  const unsigned lrow = lds0 + (unsigned)(wy * RY) * ROWB + (unsigned)(lane * 16);
  const unsigned eown = lds0 + (unsigned)(wy * RY) * 8u;
  // per-row store offsets (the plane's offset travels as the scalar operand); out of range where there is nothing to store
  unsigned uoff[RY], moff[RY];
  for (int r = 0; r < RY; r ++) {
    const bool rok = (row_ok >> r) & 1;
    uoff[r] = (rok && ucol != OOB) ? ucol + (unsigned)m.u_pitch * (unsigned)r : OOB;
    moff[r] = (rok && in_row) ? mcol + (unsigned)P * (unsigned)r : OOB;
  }
  const bool wave_interior = (row_ok & row_int & row_dom) == (1u << RY) - 1u;
  // Summaries, RY == 4: ONE dword store per FOUR planes instead of four byte stores per plane (every real store of a wavefront that
  // also streams loads costs time: see DESIGN.md).  Lane (g, a, r) = (lane / 16, lane / 4 % 4, lane % 4) keeps, for plane a of the
  // batch, the summaries of row r at the four word columns 4 g .. 4 g + 3 -- one dword of U as it lies in memory.
  const bool batched = RY == 4 && have_u;
  const bool block4 = batched && m.u_rows == 4;                // one summary byte per 8 x 4 block (this wavefront's four rows)
  // ... or per 8 x 16 block: the workgroup's sixteen rows (u_rows == 16).  Every byte of summary stored costs (with the
  // stores of three of the four wavefronts aimed out of range 512^3 x 32 ran in 5.45 instead of 5.61 ms, with all of them 5.39), so the
  // four wavefronts' block bytes are ANDed through LDS: a wavefront publishes its byte at the end of step k and everybody reads the four
  // of them behind the barrier of step k + 1 (no second barrier per plane).  The mask words are still stored by the wavefront's own
  // rule, at once (its 8 x 4 block has no common bit); where the sixteen rows have none but a wavefront's four do, its summary goes out
  // as mask bytes one step late (rare: a wave-uniform branch) -- a stand-in with fewer bits than the real bytes, as everywhere.
  const bool block16 = batched && CY == 4 && m.u_rows == 16;
  const bool blk = block4 || block16;
  const unsigned a_l = ((unsigned)lane >> 2) & 3u, r_l = (unsigned)lane & 3u, g_l = (unsigned)lane >> 4;
  const unsigned urows = (unsigned)((DH + m.u_rows - 1) / m.u_rows);
  const unsigned uplane_stride = (unsigned)m.u_pitch * urows;
  // (block summaries: the quad's four lanes hold the same dword; lane r = 0 stores it -- if the block's first row exists)
  const unsigned uo4 = (blk ? ((block4 || wy == 0) && r_l == 0 && (row_ok & 1u)) : ((row_ok >> r_l) & 1u)) ? a_l * uplane_stride + (blk ? 0u : r_l * (unsigned)m.u_pitch) + (unsigned)(t0c >> 3) + 4u * g_l : OOB;
  const unsigned psel = (0x0c0c0c0cu & ~(0xffu << (8u * a_l))) | (r_l << (8u * a_l));   // v_perm: byte a <- byte r of the source, 0 elsewhere
  unsigned uacc = 0u;
  const unsigned vzero = 0u;
  // (8 x 16 blocks) the exchange: two slots of 16 quads x 4 wavefronts' bytes behind the edge ring; what is pending from the previous step
  const unsigned xch = lds0 + ERING + (unsigned)NE * EDGEB + (unsigned)(lane >> 2) * 4u;
  unsigned t_prev = 0u;
  int k_prev = 0;
  bool pend = false;
  auto finish_prev = [&](unsigned g4) {                       // g4: the four wavefronts' block bytes of plane k_prev at this quad
    unsigned t = g4 & (g4 >> 16);
    t &= t >> 8;
    const unsigned G = t & 0xffu;                              // the sign bits all 8 x 16 vertices share
    const unsigned uplane_prev = (unsigned)m.u_pitch * ((unsigned)(j0 / m.u_rows) + urows * (unsigned)k_prev);
    if (wy == 0) {                                             // (wave-uniform) the first wavefront gathers four planes' summaries and stores them
      const int j = (k_prev - z0) & 3;
      unsigned c = G << (8u * a_l);
      row_quads_or(c);
      uacc = select_lanes(uacc, c, 0x000f000f000f000full << (4 * j));
      if (j == 3 || k_prev == z1 - 1) {
        const unsigned off = j == 3 ? uo4 : (a_l <= (unsigned)j ? uo4 : OOB);
        __builtin_amdgcn_raw_buffer_store_b32(uacc, rU, off, uplane_prev - (unsigned)j * uplane_stride, 0);
      }
    }
    const bool need = G == 0u && t_prev != 0u;                 // no common bit in sixteen rows, one in this wavefront's four: its words were not stored
    if (__builtin_amdgcn_ballot_w64(need)) {
      asm volatile("" ::: "memory");
      const unsigned v = t_prev | (t_prev << 8);
      const unsigned mp = (unsigned)P * ((unsigned)j0 + (unsigned)DH * (unsigned)k_prev);
      static_for<RY>([&](auto rc) { constexpr int r = decltype(rc)::value; __builtin_amdgcn_raw_buffer_store_b16((unsigned short)v, rM, need ? moff[r] : OOB, mp, 0); });
    }
    pend = false;
  };

  // LDS reads by hand (see the header): issue them all, one lgkmcnt wait, then tie the registers to the wait
  auto take_plane = [&](v2d (&B)[RY + 2], int q) { lds_rows(B, lrow + slot_of(q), std::make_integer_sequence<int, RY + 2>{}); };
  auto landed = [&](v2d (&B)[RY + 2]) {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    lds_tie_rows(B, std::make_integer_sequence<int, RY + 2>{});
  };

  auto step = [&](const v2d (&PR)[RY + 2], const v2d (&CU)[RY + 2], v2d (&NX)[RY + 2], int k) {
    // this wavefront's share of plane k+1 has landed: all but the operations issued after it may still be in flight
    if (batched) wait_plane_landed<DEPTH, RY, RY>(k - z0);    // (the summaries' one store per four planes is not counted: the wait is one operation stricter then)
    else wait_plane_landed<DEPTH, RY, 2 * RY>(k - z0);
    if (block16) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // (the block byte this wavefront published at the end of the previous step is in LDS)
    __builtin_amdgcn_s_barrier();                              // ... and so has everybody else's
    const bool live = k < z1;                                  // (wave-uniform) padding steps of the chunk have nothing to classify or store
    if constexpr (!TWOB) { const int q = k + NS; issue(q < z1 ? q : z1); }   // the slot read during the previous step, the edge entry of plane k-1
    double XL[RY], XR[RY];
    const bool fin = block16 && pend;                          // (wave-uniform) the previous plane's 8 x 16 summaries: the four wavefronts' bytes
    unsigned g4 = 0u;
    if (fin) asm volatile("ds_read_b32 %0, %1" : "=v"(g4) : "v"(xch + (unsigned)(k_prev & 1) * 64u));
    if (live) {
      take_plane(NX, k + 1 < z1 ? k + 1 : z1);
      lds_edges<(int)ERIGHT>(XL, XR, eown + edge_of(k), std::make_integer_sequence<int, RY>{});
      landed(NX);
      lds_tie_edges(XL, XR, std::make_integer_sequence<int, RY>{});
    } else if (fin) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if (fin) { asm volatile("" : "+v"(g4)); finish_prev(g4); }
    if constexpr (TWOB) {
      __builtin_amdgcn_s_barrier();                            // everybody has plane k+1 (and plane k's edge values) in registers:
      const int q = k + 1 + NS; issue(q < z1 ? q : z1);        // its row slot and that edge entry take the plane NS steps ahead
    }
    if (!live) return;
    const bool z_dom = k + m.ext_st[2] >= m.dom_lb[2] && k + m.ext_st[2] <= m.dom_ub[2];
    const bool z_int = k >= 1 && k < DD - 1;
    const unsigned mplane = (unsigned)P * ((unsigned)j0 + (unsigned)DH * (unsigned)k);
    const unsigned uplane = (unsigned)m.u_pitch * ((unsigned)(j0 / m.u_rows) + urows * (unsigned)k);
    // A wavefront issues ONE instruction per four cycles, scalar or vector: per-row tests and branches cost as much as the sign
    // arithmetic.  So the rows come in four compile-time flavours -- INTERIOR: every row of this wavefront and this plane lies inside
    // the domain (no per-row boundary logic at all; 15 of 16 tiles, 507 of 512 planes on a 512^3 slice); RULE: the per-vertex
    // overflow rule is on -- and what can wait is done once per step on all rows together: the summaries (packed, one pair of quad
    // permutes), the test for components without a strict sign (on the AND of the rows), the stores.
    unsigned bw[RY];                                            // the rows' pairs of mask bytes
    unsigned raw_and = 0xffffffffu;                             // AND of the rows' sign bits as classified (before rule / boundary logic)
    auto rows = [&](auto interior_c, auto rule_c) {
      constexpr bool INTERIOR = decltype(interior_c)::value, RULE = decltype(rule_c)::value;
      static_for<RY>([&](auto rc) {
        constexpr int r = decltype(rc)::value;
        const v2d c = CU[r + 1];
        const double xm = dpp_lower_or(c.y, XL[r]), xp = dpp_upper_or(c.x, XR[r]);
        const double dx0 = c.y - xm, dx1 = xp - c.x;
        const double dy0 = CU[r + 2].x - CU[r].x, dy1 = CU[r + 2].y - CU[r].y;
        const double dz0 = NX[r + 1].x - PR[r + 1].x, dz1 = NX[r + 1].y - PR[r + 1].y;
        unsigned a0 = 0, a1 = 0;
        shift_in_signs<3>(a0, a1, dx0, dx1, dy0, dy1, dz0, dz1, tneg, tpos);
        double m0 = max_abs2(dx0, dy0), m1 = max_abs2(dx1, dy1);
        m0 = max_with_abs(m0, dz0); m1 = max_with_abs(m1, dz1);
        unsigned bits = a0 | (a1 << 8);
        const bool u_int = INTERIOR || (((row_int >> r) & 1) && z_int), u_dom = INTERIOR || (((row_dom >> r) & 1) && z_dom);
        if (INTERIOR || (u_int && ((row_ok >> r) & 1))) {       // this row holds real entries of gradient(S)
          acc0 = max_plain(acc0, m0); acc1 = max_plain(acc1, m1);
          raw_and &= bits;
        }
        if constexpr (RULE) bits = (m0 >= tbig ? 0u : (bits & 0x00ffu)) | (m1 >= tbig ? 0u : (bits & 0xff00u));
        if constexpr (INTERIOR) bits = (bits & xkeep) | xneutral;
        else bits = (bits & (u_int ? xkeep : 0u)) | (u_dom ? xneutral : 0x3f3fu);
        bw[r] = bits;
      });
    };
    const bool interior = z_int && z_dom && wave_interior;
    if (interior) { if (per_vertex_rule) rows(std::true_type{}, std::true_type{}); else rows(std::true_type{}, std::false_type{}); }
    else { if (per_vertex_rule) rows(std::false_type{}, std::true_type{}); else rows(std::false_type{}, std::false_type{}); }
    // candidates for the slice's resolution (see guard_and_reduce): components without a strict sign.  A component that is strict
    // in every row with one sign survives the AND; anything else sends the wavefront through the exact per-row arithmetic, which
    // takes the minimum over ALL entries of the rows (entries at or above the threshold never lower it)
    {
      const unsigned u = raw_and | (raw_and >> 3);
      if (__builtin_amdgcn_ballot_w64((~u & cmask) != 0u)) {
        static_for<RY>([&](auto rc) {
          constexpr int r = decltype(rc)::value;
          if (((row_int >> r) & 1) && z_int && ((row_ok >> r) & 1)) {
            const v2d c = CU[r + 1];
            const double xm = dpp_lower_or(c.y, XL[r]), xp = dpp_upper_or(c.x, XR[r]);
            auto take = [&](double d) { const double a = fabs(0.5 * d); red_mn = fmin(red_mn, a == 0.0 ? DBL_MAX : a); };
            if (cmask & 0x00ffu) { take(c.y - xm); take(CU[r + 2].x - CU[r].x); take(NX[r + 1].x - PR[r + 1].x); }
            if (cmask & 0xff00u) { take(xp - c.x); take(CU[r + 2].y - CU[r].y); take(NX[r + 1].y - PR[r + 1].y); }
          }
        });
      }
    }
    // summaries: byte r of T = AND of row r's two mask bytes, then of the quad's four lanes -- all rows in one pair of permutes.  A
    // word (block) that has no common sign bit gets its mask words stored; the others aim beyond the buffer
    unsigned T[(RY + 3) / 4];
    unsigned long long wu[RY];
    unsigned mo[RY];
    if (blk) {
      // ONE byte for the wavefront's four rows: the sign bits all 8 x 4 vertices share
      if constexpr (RY == 4) {
        unsigned all = bw[0] & bw[1];
        all &= bw[2] & bw[3];
        pack_pair_and<0>(T[0], all);
        quad_and(T[0]);
        byte_nonzero<0>(wu[0], T[0], vzero);
        asm volatile("s_nop 1");
        static_for<RY>([&](auto rc) { constexpr int r = decltype(rc)::value; mo[r] = out_of_range_where(moff[r], wu[0]); });
      }
    } else {
      static_for<RY>([&](auto rc) { constexpr int r = decltype(rc)::value; pack_pair_and<r % 4>(T[r / 4], bw[r]); });
      static_for<(RY + 3) / 4>([&](auto gc) { quad_and(T[decltype(gc)::value]); });
      if (!have_u) static_for<(RY + 3) / 4>([&](auto gc) { T[decltype(gc)::value] = 0u; });   // (wave-uniform) no summaries: every mask word is stored
      static_for<RY>([&](auto rc) { constexpr int r = decltype(rc)::value; byte_nonzero<r % 4>(wu[r], T[r / 4], vzero); });
      if constexpr (RY < 3) asm volatile("s_nop 1");
      static_for<RY>([&](auto rc) { constexpr int r = decltype(rc)::value; mo[r] = out_of_range_where(moff[r], wu[r]); });
    }
    if (block16) {
      // the wavefront's block byte goes to the exchange (every lane of a quad writes the same byte to the same place); the sixteen rows'
      // summary and whatever follows from it: finish_prev, behind the next barrier
      asm volatile("ds_write_b8 %0, %1" :: "v"(xch + (unsigned)(k & 1) * 64u + (unsigned)wy), "v"(T[0]) : "memory");
      t_prev = T[0]; k_prev = k; pend = true;
      static_for<RY>([&](auto rc) { constexpr int r = decltype(rc)::value; __builtin_amdgcn_raw_buffer_store_b16((unsigned short)bw[r], rM, mo[r], mplane, 0); });
      return;
    }
    if (batched) {
      // every lane of a quad holds the quad's T (byte r = row r at the quad's word column).  Lane (a, r) moves ITS row's byte to byte a;
      // the OR over the row's four quads then is the dword [row r at word columns 4 g .. 4 g + 3], in all four quads; quad j keeps it
      const int j = (k - z0) & 3;
      unsigned c = block4 ? (T[0] << (8u * a_l)) : __builtin_amdgcn_perm(0u, T[0], psel);
      row_quads_or(c);
      uacc = select_lanes(uacc, c, 0x000f000f000f000full << (4 * j));
      if (j == 3 || k == z1 - 1) {                             // (wave-uniform) four planes gathered, or the chunk ends
        const unsigned off = j == 3 ? uo4 : (a_l <= (unsigned)j ? uo4 : OOB);
        __builtin_amdgcn_raw_buffer_store_b32(uacc, rU, off, uplane - (unsigned)j * uplane_stride, 0);
      }
      static_for<RY>([&](auto rc) { constexpr int r = decltype(rc)::value; __builtin_amdgcn_raw_buffer_store_b16((unsigned short)bw[r], rM, mo[r], mplane, 0); });
      return;
    }
    static_for<RY>([&](auto rc) {
      constexpr int r = decltype(rc)::value;
      __builtin_amdgcn_raw_buffer_store_b8((unsigned char)(T[r / 4] >> (8 * (r % 4))), rU, uoff[r], uplane, 0);
      __builtin_amdgcn_raw_buffer_store_b16((unsigned short)bw[r], rM, mo[r], mplane, 0);
    });
  };

  v2d B[3][RY + 2];
  issue(z0 - 1); issue(z0);
  FTKX_WAIT_VM(0);
  __builtin_amdgcn_s_barrier();                                // planes z0-1 and z0 are in LDS
  take_plane(B[0], z0 - 1);
  take_plane(B[1], z0);
  landed(B[0]); landed(B[1]);
  __builtin_amdgcn_s_barrier();                                // everybody has taken them: their slots may be refilled
  for (int q = z0 + 1; q <= z0 + DEPTH; q ++) issue(q < z1 ? q : z1);
  for (int s = 0; s < nsteps; s += 3) {
    step(B[0], B[1], B[2], z0 + s);
    step(B[1], B[2], B[0], z0 + s + 1);
    step(B[2], B[0], B[1], z0 + s + 2);
  }
  if (block16 && pend) {                                       // (workgroup-uniform) the last plane's summaries: no step follows it
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    unsigned g4;
    asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(g4) : "v"(xch + (unsigned)(k_prev & 1) * 64u) : "memory");
    finish_prev(g4);
  }
  FTKX_WAIT_VM(0);                                             // nothing may still be writing LDS when the workgroup retires
#undef FTKX_WAIT_VM
  if (job.red) {
    const double mx = fmax((cmask & 0x00ffu) ? acc0 : 0.0, (cmask & 0xff00u) ? acc1 : 0.0) * 0.5;
    red_commit(job.red, red_mn < job.threshold ? red_mn : DBL_MAX, mx, blockIdx.x + blockIdx.y * 7u + blockIdx.z * 13u + (unsigned)wv);
  }
#endif
}

template <int NS, int CY, int RY, bool TWOB>
__global__ __launch_bounds__(64 * CY) void mask_march6_kernel(const Mesh m, const MaskJob *__restrict__ jobs, int swizzle, int njobs, const ZPlan plan)
{
  march6_body<NS, CY, RY, TWOB>(m, jobs, swizzle, njobs, plan);
}
// ---------------------------------------------------------------------------------------------------------------
// ndarray::resolution() of V = gradient(S) without materialising V (include/ftk/ndarray.hh:770-778 over grad.hh's output):
// min over non-zero finite |v| and max finite |v| as raw IEEE bit patterns (they order like unsigned integers for v >= 0)
// ---------------------------------------------------------------------------------------------------------------
template <int ND>
__global__ __launch_bounds__(kThreads) void resolution_scalar_kernel(const Mesh m, const double *__restrict__ S, u64 *out)
{
  u64 mn = 0x7fefffffffffffffull, mx = 0ull;
  const int DW = m.ext_sz[0], DH = m.ext_sz[1], DD = (ND == 3) ? m.ext_sz[2] : 1;
  const size_t n = (size_t)DW * DH * DD;
  for (size_t idx = (size_t)blockIdx.x * kThreads + threadIdx.x; idx < n; idx += (size_t)gridDim.x * kThreads) {
    const int i = (int)(idx % DW), j = (int)((idx / DW) % DH), k = (int)(idx / ((size_t)DW * DH));
    double g[ND];
    gradient_at<ND>(m, S, i, j, k, g);
    for (int c = 0; c < ND; c ++) {
      const double a = fabs(g[c]);
      const u64 bits = (u64)__double_as_longlong(a);
      if (a != 0.0 && bits < 0x7ff0000000000000ull) { mn = bits < mn ? bits : mn; mx = bits > mx ? bits : mx; }
    }
  }
  for (int o = 32; o > 0; o >>= 1) {
    const u64 omn = __shfl_down(mn, o), omx = __shfl_down(mx, o);
    mn = omn < mn ? omn : mn; mx = omx > mx ? omx : mx;
  }
  const unsigned slot = (blockIdx.x * 5u + (threadIdx.x >> 6)) & 63u;   // 64 result slots, folded by the host
  if ((threadIdx.x & 63) == 0) { atomicMin(&out[2 * slot], mn); atomicMax(&out[2 * slot + 1], mx); }
}

// name of the mask-kernel instantiation the last (non pre-pass) launch used, as rocprofv3 prints it: bench.py reports it next to the
// kernel family so that its roofline line can be matched with the profiler's summary
static const char *g_last_mask_kernel = "";
const char *last_mask_kernel() { return g_last_mask_kernel; }
// Which mask kernel took how many launches in this process: launch_masks_impl picks one of six by the mesh (3D scalar: march6; 2D scalar: march4,
// rows2 for rows of 32 KB and more; vector input: vec2 with block summaries, vec without; everything else: the generic kernel), and the GPU
// suite is meant to reach every one of them (tests/conftest.py writes the table; profiles/r06_mask_kernel_coverage.json).
static unsigned long long g_mask_launches[kMaskKernels] = {0, 0, 0, 0, 0, 0, 0};
static const char *const g_mask_names[kMaskKernels] = {"mask_march6_kernel (3D scalar)", "mask_march4_kernel (2D scalar)", "mask_rows2_kernel (2D scalar, long rows)",
                                                       "mask_march4_kernel<reduce> (stand-alone reduction)", "mask_vec2_kernel (vector input, block summaries)",
                                                       "mask_vec_kernel (vector input)", "mask_kernel (generic)"};
void mask_kernel_launches(unsigned long long *out, const char **names) { for (int i = 0; i < kMaskKernels; i ++) { if (out) out[i] = g_mask_launches[i]; if (names) names[i] = g_mask_names[i]; } }

void launch_masks_impl(const Mesh &m, const MaskJob *d_jobs, int njobs, bool reduce, hipStream_t stream);
bool masks_have_summary(const Mesh &m);
int mask_summary_rows(const Mesh &m);

// can the 128-column marching kernels (which also carry the exact pre-pass reduction) walk this mesh? (which carries the fused reduction) walk this mesh?
bool march2_supported(const Mesh &m)
{
  const int DD = m.nd == 3 ? m.ext_sz[2] : 1;
  const size_t slice_bytes = (size_t)m.ext_sz[0] * m.ext_sz[1] * DD * 8;
  return m.scalar_mode && (m.ext_sz[0] % 2) == 0 && m.ext_sz[0] >= 2 && slice_bytes < (1ull << 32);
}

void launch_resolution_scalar(const Mesh &m, const double *S, u64 *out2, hipStream_t stream)
{
  const size_t n = (size_t)m.ext_sz[0] * m.ext_sz[1] * (m.nd == 3 ? m.ext_sz[2] : 1);
  size_t bx = (n + kThreads - 1) / kThreads;
  if (bx > 4096) bx = 4096;
  if (m.nd == 2) hipLaunchKernelGGL(resolution_scalar_kernel<2>, dim3((unsigned)bx), dim3(kThreads), 0, stream, m, S, out2);
  else hipLaunchKernelGGL(resolution_scalar_kernel<3>, dim3((unsigned)bx), dim3(kThreads), 0, stream, m, S, out2);
}

// Calibration of the HBM counters (MI355X_MICROARCH.md, HBM: "calibrate on a known byte count in your own access pattern"):
// streams `n16` 16-byte words with the mask kernel's load shape (16 B per lane, 1 KiB per wavefront instruction) and nothing else.
__global__ __launch_bounds__(kThreads) void calib_read_kernel(const double2 *__restrict__ p, size_t n16, double *out)
{
  double acc = 0.0;
  for (size_t i = (size_t)blockIdx.x * kThreads + threadIdx.x; i < n16; i += (size_t)gridDim.x * kThreads) { const double2 v = p[i]; acc += v.x + v.y; }
  if (acc == 1.2345e300) out[0] = acc;    // never true for finite data: keeps the loads alive without a store
}

void launch_calib_read(const void *p, size_t bytes, double *scratch, hipStream_t stream)
{
  hipLaunchKernelGGL(calib_read_kernel, dim3(256 * 16), dim3(kThreads), 0, stream, (const double2 *)p, bytes / 16, scratch);
}

// does this mesh take the fast vector-input kernel?
static bool vec_fast(const Mesh &m) { return !m.scalar_mode && m.ext_sz[0] >= 8 && (m.ext_sz[0] % 8) == 0; }
// ... its form with block summaries (mask_vec2_kernel)?  Rows of at least 64 groups, byte offsets that fit 31 bits.  FTKX_MASK_PLAN lean=0: never
static bool vec_lean(const Mesh &m)
{
  if (!vec_fast(m) || m.ext_sz[0] < 256) return false;
  if (env_hook("FTKX_MASK_PLAN", "lean", 1) == 0) return false;
  const size_t n = (size_t)m.ext_sz[1] * (m.nd == 3 ? (size_t)m.ext_sz[2] : 1);
  return (size_t)m.ext_sz[0] * n * 8 * (size_t)m.nd < (1ull << 31) && (size_t)m.mask_pitch * n < (1ull << 31);
}

void launch_masks_impl(const Mesh &m, const MaskJob *d_jobs, int njobs, bool reduce, hipStream_t stream)
{
  if (njobs <= 0) return;
  if (m.scalar_mode && march2_supported(m)) {
    // scalar slices with an even row length below 4 GiB: the marching kernels on 128-column, line-aligned tiles
    const int DW = m.ext_sz[0], DD = m.nd == 3 ? m.ext_sz[2] : 1;
    int swizzle = 8;   // grouped placement -- the x tiles of some row groups on one XCD -- cuts the fabric reads from 47.7 to 41.4 GB per 512^3 x 32 launch
    // 2D: the rows of a wavefront's block that no other wavefront reads (all but its first and last two) are loaded non-temporally -- they are
    // read once, and keeping them out of the caches leaves the halo rows there for the neighbours: woven 1024^2 x 64 0.115 -> 0.102 ms (all
    // loads non-temporal: 0.106; 3D, where the planes are re-read by the z march: 256^3 x 16 -1 %, 512^3 x 32 +1.3 %: left as it is.  The
    // vector-input kernel, whose every value is read once, does NOT like non-temporal loads: double_gyre 0.73 -> 1.30 ms)
    if (m.nd == 2) swizzle |= 16;
    swizzle = (int)env_hook("FTKX_MASK_PLAN", "swizzle", swizzle);
    int zchunk = 32;
    bool zforced = false;
    if (env_hook("FTKX_MASK_PLAN", "zchunk", 0) > 0) { zchunk = (int)env_hook("FTKX_MASK_PLAN", "zchunk", 0); zforced = true; }
    if (m.nd == 3 && !reduce) {
      // 3D: mask_march6_kernel -- 128 x 16 tiles as four wavefronts of 4 rows that all load (LDS-DMA) and classify, TWO row slots in
      // LDS (37 KB: three workgroups = twelve wavefronts per CU, which its 164 VGPRs allow), one barrier per plane; grouped placement:
      // 16 row groups (all of a 256^2 plane's, half of a 512^2 plane's tiles) of one piece of planes share an XCD's L2 (4: +3.5 %, 8: +0.5 %)
      int yg_want = 16;
      if (env_hook_set("FTKX_MASK_PLAN", "yg")) yg_want = env_hook("FTKX_MASK_PLAN", "yg", 16) > 0 ? (int)env_hook("FTKX_MASK_PLAN", "yg", 16) : 1;
      if (yg_want > 255) yg_want = 255;
      // The pieces a tile column is marched in (ZPlan): at most 24 planes, at most half of what is left of the column, at least 6,
      // multiples of 3 (the march is unrolled three planes deep), handed out longest first.  Measured, not derived (tools/ab_mask.py,
      // interleaved on one box): against equal chunks of 32 planes 256^3 x 16 0.418 -> 0.405 ms, one 512^3 slice 0.206 -> 0.197, four
      // 0.756 -> 0.748, 512^3 x 32 5.72 -> 5.69.  LONGER marches are slower although they pay fewer start-up planes (caps of 28 / 32 / 48:
      // +9 / +5 / +1..2 % on 512^3 x 32; half columns +3.7 %: the tiles of a group drift apart and stop sharing their halo rows in the L2),
      // equal chunks swing by +-3 % with their length (24: 5.89, 27: 5.68, 30: 5.98, 32: 5.72, 33: 5.84 ms -- what is left over at a
      // column's top decides).  Order of the workgroups: slice by slice where a slice alone fills the device (neighbouring pieces of a
      // slice then run together and find each other's start-up planes in the caches: 512^3 x 32 5.59 against 5.85 ms piece by piece),
      // piece by piece over all slices otherwise (256^3 x 16: 0.379 against 0.399).
      // FTKX_MASK_PLAN (test hooks): zchunk=n: equal chunks of n planes; lcap / lmin: the two bounds; order=0 / 1
      ZPlan plan = ZPlan();
      bool planned = false;
      {
        int lmin = 6, lcap = 24;
        planned = !zforced;
        if (env_hook("FTKX_MASK_PLAN", "lmin", 0) >= 1) lmin = (int)env_hook("FTKX_MASK_PLAN", "lmin", 0);
        if (env_hook("FTKX_MASK_PLAN", "lcap", 0) >= 1) lcap = (int)env_hook("FTKX_MASK_PLAN", "lcap", 0);
        if (lcap < lmin) lcap = lmin;
        std::vector<int> lens;
        int rem = DD;
        if (!planned) while (rem > 0) { const int l = rem < zchunk ? rem : zchunk; lens.push_back(l); rem -= l; }
        while (rem > 0) {
          int l = (rem + 1) / 2;
          if (l > lcap) l = lcap;
          if (l < lmin) l = lmin;
          if (l >= 3) l -= l % 3;      // (the march is unrolled three planes deep: a length that is no multiple of 3 pays for up to two empty steps)
          if (l > rem || rem - l < (lmin + 1) / 2) l = rem;
          lens.push_back(l); rem -= l;
        }
        while (lens.size() > 47) { const int l = lens.back(); lens.pop_back(); lens.back() += l; }      // (more pieces than the table holds: merged from the end)
        std::stable_sort(lens.begin(), lens.end(), [](int a, int b) { return a > b; });
        plan.npieces = (unsigned)lens.size();
        int z = 0;
        for (size_t i = 0; i < lens.size(); i ++) { plan.z0[i] = (unsigned)z; plan.len[i] = (unsigned)lens[i]; z += lens[i]; }
      }
      constexpr int NS6 = 2, CY6 = 4, RY6 = 4, rows = CY6 * RY6;
      g_last_mask_kernel = "ftkx::mask_march6_kernel<2, 4, 4, false>"; g_mask_launches[0] ++;
      dim3 grid6((unsigned)((DW + 127) / 128), (unsigned)((m.ext_sz[1] + rows - 1) / rows), plan.npieces * (unsigned)njobs);
      int sw = swizzle;
      // grouped placement needs a y extent that is a multiple of the group height: pad it (workgroups past the last row leave at once)
      if (sw & 8) { int yg = yg_want; if (yg > (int)grid6.y) yg = (int)grid6.y; grid6.y = (grid6.y + (unsigned)yg - 1) / (unsigned)yg * (unsigned)yg; sw = (sw & 0xff) | (yg << 8); }
      const unsigned bytes = (unsigned)NS6 * (unsigned)(rows + 2) * 1024u + (unsigned)(NS6 + 1) * 256u + 128u;   // row slots, edge ring, the summaries' exchange
      (void)hipFuncSetAttribute((const void *)mask_march6_kernel<NS6, CY6, RY6, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
      bool slice_major = (size_t)grid6.x * ((m.ext_sz[1] + rows - 1) / rows) * plan.npieces >= 768;      // a slice alone fills the device (three workgroups per CU)
      if (env_hook_set("FTKX_MASK_PLAN", "order")) slice_major = env_hook("FTKX_MASK_PLAN", "order", 0) == 0;
      hipLaunchKernelGGL((mask_march6_kernel<NS6, CY6, RY6, false>), grid6, dim3(64u * CY6), bytes, stream, m, d_jobs, sw, slice_major ? -njobs : njobs, plan);
      return;
    }
    // 2D, and the exact stand-alone reduction (ftkx_slice_resolution) of either dimension: mask_march4_kernel -- every wavefront loads
    // its rows into registers (4 wavefronts of 8 rows in 2D, of 4 rows marching along z in 3D)
    const int RY = (m.nd == 3) ? 4 : 8, wpb = 4;
    if (m.nd == 2 && !reduce) {
      // mask_rows2_kernel: a wavefront marches down `groups` groups of 8 rows (FTKX_MASK_PLAN rows=n, test hook; rows=0: the kernel below).
      // Measured (tools/ab_mask.py, interleaved on one box, 4 groups against the kernel below): 4096^2 x 16 0.428 -> 0.384 ms, 2048^2 x 64
      // 0.412 -> 0.397 (8 groups: 0.390), but 1024^2 x 64 0.096 -> 0.109 and 1024^2 x 256 0.379 -> 0.399: with rows of 8 KB the short
      // wavefronts of the kernel below, whose neighbours in x run together, read whole rows; taken for rows of 32 KB and more
      int groups = (DW >= 4096 && m.ext_sz[1] >= 512) ? 4 : 0;
      if (env_hook_set("FTKX_MASK_PLAN", "rows")) groups = (int)env_hook("FTKX_MASK_PLAN", "rows", groups);
      if (groups > 64) groups = 64;
      if (groups >= 1) {
        dim3 gridr((unsigned)((DW + 127) / 128), (unsigned)((m.ext_sz[1] + wpb * RY * groups - 1) / (wpb * RY * groups)), (unsigned)njobs);
        int sw = swizzle;
        if (sw & 8) {
          int yg = 4;
          if (env_hook_set("FTKX_MASK_PLAN", "yg")) { const long v = env_hook("FTKX_MASK_PLAN", "yg", 4); yg = v > 0 ? (v > 255 ? 255 : (int)v) : 1; }
          while (yg > 1 && gridr.y % (unsigned)yg) yg --;
          sw = (sw & 0xff) | (yg << 8);
        }
        g_last_mask_kernel = "ftkx::mask_rows2_kernel<8>"; g_mask_launches[2] ++;
        hipLaunchKernelGGL(mask_rows2_kernel<8>, gridr, dim3((unsigned)(64 * wpb)), 0, stream, m, d_jobs, groups, sw);
        return;
      }
    }
    const int nzc = m.nd == 3 ? (DD + zchunk - 1) / zchunk : 1;
    const dim3 grid4((unsigned)((DW + 127) / 128), (unsigned)((m.ext_sz[1] + wpb * RY - 1) / (wpb * RY)), (unsigned)(nzc * njobs));
    if (swizzle & 8) {     // (the group height must divide the grid's y extent)
      int yg = 4;
      if (env_hook_set("FTKX_MASK_PLAN", "yg")) { const long v = env_hook("FTKX_MASK_PLAN", "yg", 4); yg = v > 0 ? (v > 255 ? 255 : (int)v) : 1; }
      while (yg > 1 && grid4.y % (unsigned)yg) yg --;
      swizzle = (swizzle & 0xff) | (yg << 8);
    }
    const dim3 blk((unsigned)(64 * wpb));
#define FTKX_M4(ND_, R_, PD_, RY_) do { if (!reduce) g_last_mask_kernel = "ftkx::mask_march4_kernel<" #ND_ ", " #R_ ", " #PD_ ", " #RY_ ">"; g_mask_launches[reduce ? 3 : 1] ++; \
      hipLaunchKernelGGL((mask_march4_kernel<ND_, R_, PD_, RY_>), grid4, blk, 0, stream, m, d_jobs, zchunk, swizzle); } while (0)
    if (reduce) { if (m.nd == 2) FTKX_M4(2, true, 1, 8); else FTKX_M4(3, true, 1, 4); }
    else FTKX_M4(2, false, 1, 8);
#undef FTKX_M4
    return;
  }
  const size_t DDv = m.nd == 3 ? (size_t)m.ext_sz[2] : 1;
  if (vec_fast(m)) {
    const size_t groups = (size_t)(m.ext_sz[0] / 4) * m.ext_sz[1] * DDv;
    size_t bx = (groups + kThreads - 1) / kThreads;
    if (bx > 2048) bx = 2048;               // grid-stride the rest: 8 workgroups per CU per job
    // ... and at least four groups per lane where the slice has them: the per-wavefront fixed costs (index arithmetic, the two
    // reduction atomics) are paid per 16 KB instead of per 4 KB (double_gyre 2048 x 1024 x 128: 0.795 -> 0.746 ms)
    while (bx > 256 && bx * kThreads * 4 > groups) bx /= 2;
    const dim3 grid((unsigned)bx, (unsigned)njobs);
    // mask_vec2_kernel (units of 4 rows x 64 groups, one summary byte per 8 x 4 block): where the mesh carries block summaries
    if (m.u_rows == 4 && vec_lean(m)) {
      const size_t units = (size_t)((m.ext_sz[0] / 4 + 63) / 64) * ((m.ext_sz[1] + 3) / 4) * DDv;
      size_t bx2 = (units + 7) / 8;             // two units (32 KB) per wavefront where the slice has them (four: +1.3 %, one: +0.3 % on double_gyre 2048 x 1024 x 128)
      if (bx2 > 2048) bx2 = 2048;
      const dim3 grid((unsigned)bx2, (unsigned)njobs);
      g_last_mask_kernel = m.nd == 2 ? "ftkx::mask_vec2_kernel<2>" : "ftkx::mask_vec2_kernel<3>"; g_mask_launches[4] ++;
      if (m.nd == 2) hipLaunchKernelGGL(mask_vec2_kernel<2>, grid, dim3(kThreads), 0, stream, m, d_jobs);
      else hipLaunchKernelGGL(mask_vec2_kernel<3>, grid, dim3(kThreads), 0, stream, m, d_jobs);
      return;
    }
    g_last_mask_kernel = m.nd == 2 ? "ftkx::mask_vec_kernel<2>" : "ftkx::mask_vec_kernel<3>"; g_mask_launches[5] ++;
    if (m.nd == 2) hipLaunchKernelGGL(mask_vec_kernel<2>, grid, dim3(kThreads), 0, stream, m, d_jobs);
    else hipLaunchKernelGGL(mask_vec_kernel<3>, grid, dim3(kThreads), 0, stream, m, d_jobs);
    return;
  }
  // the one generic form (odd row lengths, slices of 4 GiB and more, vector rows that are not a multiple of 8): one lane per mask byte
  const size_t n = (size_t)m.mask_pitch * m.ext_sz[1] * DDv;
  size_t bx = (n + kThreads - 1) / kThreads;
  if (bx > 4096) bx = 4096;                 // grid-stride the rest
  const dim3 grid((unsigned)bx, (unsigned)njobs);
  g_last_mask_kernel = m.nd == 2 ? "ftkx::mask_kernel<2>" : "ftkx::mask_kernel<3>"; g_mask_launches[6] ++;
  if (m.nd == 2) hipLaunchKernelGGL(mask_kernel<2>, grid, dim3(kThreads), 0, stream, m, d_jobs);
  else hipLaunchKernelGGL(mask_kernel<3>, grid, dim3(kThreads), 0, stream, m, d_jobs);
}

// Rows a summary byte stands for.  mask_march6_kernel with four rows per wavefront writes ONE byte per 8 x 4 block of vertices
// (aligned in y): a quarter of the summary bytes to write (what they cost: DESIGN.md) and for the coarse cull to read.  Everything
// else writes one byte per word of 8.  The same decision as launch_masks_impl's choice of kernel (same environment knobs).
int mask_summary_rows(const Mesh &m)
{
  if (!masks_have_summary(m)) return 1;
  int want = 0;
  if (const char *e = getenv("FTKX_U_ROWS")) want = atoi(e);
  if (want == 1) return 1;
  if (!m.scalar_mode) return vec_lean(m) ? 4 : 1;              // mask_vec2_kernel / mask_vec_kernel
  if (m.nd == 3) return want == 4 ? 4 : 16;                    // mask_march6_kernel: the workgroup's sixteen rows (FTKX_U_ROWS=4: a wavefront's four)
  return 4;                                                    // mask_march4_kernel<2, ...>: a wavefront's rows in blocks of four
}

// does launch_masks produce the per-word summaries for this mesh?  (the 128-column marching kernels and the fast vector kernel do)
bool masks_have_summary(const Mesh &m)
{
  if (const char *e = getenv("FTKX_U_ROWS")) if (atoi(e) < 0) return false;      // (FTKX_U_ROWS=-1: no summaries at all, the one-level cull)
  if (!m.scalar_mode) return vec_fast(m);
  return march2_supported(m) && (m.ext_sz[0] % 8) == 0;
}

// are the reduction slots of MaskJob::red filled by launch_masks (the fused one-pass form)?  All mask kernels do.
bool masks_fuse_reduction(const Mesh &) { return true; }

void launch_masks(const Mesh &m, const MaskJob *d_jobs, int njobs, hipStream_t stream) { launch_masks_impl(m, d_jobs, njobs, false, stream); }
// pre-pass: only valid when march2_supported(m)
void launch_reduce_march(const Mesh &m, const MaskJob *d_jobs, int njobs, hipStream_t stream) { launch_masks_impl(m, d_jobs, njobs, true, stream); }

}  // namespace ftkx
