// HIP kernels of the critical-point space-time simplex sweep for gfx950 (MI355X, CDNA4).
//
// Replaces, behind the reference's accelerator boundary, what the reference does per timestep in
//   critical_point_tracker_{2d,3d}_regular::update_timestep() -> element_for_{ordinal,interval} -> check_simplex
//   (include/ftk/filters/critical_point_tracker_2d_regular.hh:263-433, 584-685; ..._3d_regular.hh:150-308, 425-514)
// and what its CUDA back-end does with one thread per simplex and a global atomic per hit
//   (src/filters/critical_point_tracer_{2d,3d}_regular.cu).
//
// Structure of one launch (one workgroup of 256 lanes = 4 wavefronts per tile of lattice corners):
//   1. stage   the (TX+1)x(TY+1)x(TZ+1) vertex block of slice t (and t+1) is read from HBM once, quantised to int64
//              and parked in LDS together with one classification byte per vertex;
//   2. cull    each lane owns one corner: if some vector component has the same strict sign on every vertex of the
//              corner's space-time hypercube, none of its 12/60 simplices can contain the origin (legal only while no
//              determinant can overflow int64 -- the host decides, SweepParams::cull); surviving corners are
//              ballot-compacted into an LDS work list;
//   3. test    the (surviving corner x simplex type) pairs are spread over all 256 lanes; each pair reads its d+1
//              vertices from LDS, repeats the cull per simplex, then runs the exact integer predicate (cp_device.hpp);
//   4. emit    hits -- a fraction of a percent -- gather FP64 inputs from HBM, solve, classify, and are appended to the
//              device hit buffer with one wave-aggregated atomic per wavefront.
// No MFMA: this is a stencil of 64-bit integer VALU work on 8-24 bytes per vertex, bounded by HBM once the cull applies.
#include <hip/hip_runtime.h>

#include "cp_device.hpp"
#include "fan_tables.hpp"
#include "sweep_params.hpp"

namespace ftkx {

__constant__ fan_table<3> c_fan3 = make_fan<3>();
__constant__ fan_table<4> c_fan4 = make_fan<4>();

template <int ND> struct tile_cfg;
template <> struct tile_cfg<2> { static constexpr int TX = 32, TY = 8, TZ = 1; };
template <> struct tile_cfg<3> { static constexpr int TX = 16, TY = 4, TZ = 4; };

constexpr int kThreads = 256;
constexpr unsigned char kInvalid = 0x80;    // vertex outside the domain / array
constexpr unsigned char kNonFinite = 0x40;  // NaN or Inf component: the reference rejects the simplex (2d:611, 3d:457)

template <int ND> __device__ inline const fan_table<ND + 1> &dev_fan();
template <> __device__ inline const fan_table<3> &dev_fan<2>() { return c_fan3; }
template <> __device__ inline const fan_table<4> &dev_fan<3>() { return c_fan4; }

// ---------------------------------------------------------------------------------------------------------------
// hit path (rare): everything in FP64 from HBM
// ---------------------------------------------------------------------------------------------------------------
template <int ND>
__device__ inline size_t ext_index(const SweepParams &p, const int *vx)
{
  size_t idx = (size_t)(vx[0] - p.ext_st[0]);
  size_t stride = (size_t)p.ext_sz[0];
  for (int d = 1; d < ND; d ++) { idx += (size_t)(vx[d] - p.ext_st[d]) * stride; stride *= (size_t)p.ext_sz[d]; }
  return idx;
}

__device__ inline int clampi(int i, int lo, int hi) { return i < lo ? lo : (i > hi ? hi : i); }

// J at one vertex, derived from V exactly like ndarray/grad.hh (jacobian2D 54-86 incl. its operator precedence, jacobian3D
// 175-212 incl. its interior-only support); Js[j][k] = J(k, j, vertex) as the trackers read it (2d:566-582, 3d:405-422)
template <int ND>
__device__ inline void derive_jacobian_at(const SweepParams &p, const double *V, const int *vx, double Js[ND][ND])
{
  const int DW = p.ext_sz[0], DH = p.ext_sz[1];
  const int i = vx[0] - p.ext_st[0], j = vx[1] - p.ext_st[1];
  if constexpr (ND == 2) {
    auto f = [&](int c, int a, int b) { return V[(size_t)c + 2 * ((size_t)clampi(a, 0, DW - 1) + (size_t)DW * (size_t)clampi(b, 0, DH - 1))]; };
    const double H00 = f(0, i + 1, j) - f(0, i - 1, j) * (DW - 1),
                 H01 = f(0, i, j + 1) - f(0, i, j - 1) * (DH - 1),
                 H10 = f(1, i + 1, j) - f(1, i - 1, j) * (DW - 1),
                 H11 = f(1, i, j + 1) - f(1, i, j - 1) * (DH - 1);
    Js[0][0] = H00;
    Js[1][1] = H11;
    // symmetric instantiation stores the mean in both off-diagonals; the other one leaves them 0 (grad.hh:79-82)
    Js[0][1] = Js[1][0] = p.jac_symmetric_derive ? (H01 + H10) * 0.5 : 0.0;
  } else {
    const int DD = p.ext_sz[2];
    const int k = vx[2] - p.ext_st[2];
    const bool interior = i >= 2 && i < DW - 2 && j >= 2 && j < DH - 2 && k >= 2 && k < DD - 2;
    auto f = [&](int c, int a, int b, int d) { return V[(size_t)c + 3 * ((size_t)a + (size_t)DW * ((size_t)b + (size_t)DH * (size_t)d))]; };
    for (int a = 0; a < 3; a ++) {
      // J(a, b) = 0.5 * (V_a(x + e_b) - V_a(x - e_b));  Js[j][k] = J(k, j)
      Js[0][a] = interior ? 0.5 * (f(a, i + 1, j, k) - f(a, i - 1, j, k)) : 0.0;
      Js[1][a] = interior ? 0.5 * (f(a, i, j + 1, k) - f(a, i, j - 1, k)) : 0.0;
      Js[2][a] = interior ? 0.5 * (f(a, i, j, k + 1) - f(a, i, j, k - 1)) : 0.0;
    }
  }
}

// e.to_integer(m), mesh/simplicial_regular_mesh.hh:496-502
template <int ND>
__device__ inline u64 element_tag(const SweepParams &p, const int *corner /*ND spatial + time*/, int type, u64 work_index)
{
  constexpr int N = ND + 1;
  constexpr int ntypes_all = fan_table<N>::NTYPES;
  if (p.tag_mode == FTKX_TAG_WORK_INDEX) return work_index;
  u64 ci = 0;
  for (int i = 0; i < N; i ++) {
    const int rel = corner[i] - (i < ND ? p.dom_lb[i] : 0);
    if (p.tag_mode == FTKX_TAG_REFERENCE) ci += (u64)(i64)(int)((unsigned)rel * (unsigned)p.dimprod[i]);   // int * int, wraps
    else ci += (u64)(i64)rel * p.exact_prod[i];
  }
  return ci * (u64)ntypes_all + (u64)type;
}

// returns false when the 2D type filter drops the record
template <int ND>
__device__ __noinline__ bool make_record(const SweepParams &p, const int *corner, int type, u64 work_index,
                                         const u64 (*X)[ND], const int *ids, bool presolved, const double *mu_in, ftkx_cp_t *out)
{
  constexpr int N = ND + 1;
  const fan_table<N> &fan = dev_fan<ND>();
  int vx[N][N];
  size_t at[N];
  int slice[N];
  double v[N][ND];
  for (int i = 0; i < N; i ++) {
    const unsigned m = fan.vert[type][i];
    for (int d = 0; d < N; d ++) vx[i][d] = corner[d] + ((m >> d) & 1u);
    slice[i] = (m >> ND) & 1u;
    at[i] = ext_index<ND>(p, vx[i]);
    for (int j = 0; j < ND; j ++) v[i][j] = p.V[slice[i]][at[i] * ND + j];
  }
  double mu[N];
  if (presolved) { for (int i = 0; i < N; i ++) mu[i] = mu_in[i]; }
  if constexpr (ND == 2) {
    if (!solve_barycentric2(v, mu)) clamp_barycentric<3>(mu);      // 2d:626-631
  } else {
    if (!presolved) solve_barycentric3(v, mu);
    clamp_barycentric<4>(mu);                                      // 3d:470, unconditional
  }
  ftkx_cp_t r;
  r.scalar[0] = r.scalar[1] = r.scalar[2] = 0.0;
  {
    // lerp of the lattice coordinates, left to right (linear_interpolation.hh:83-101, 129-139)
    double x[4] = {0, 0, 0, 0};
    for (int d = 0; d < N; d ++) {
      double acc = (double)vx[0][d] * mu[0];
      for (int i = 1; i < N; i ++) acc = acc + (double)vx[i][d] * mu[i];
      x[d] = acc;
    }
    if constexpr (ND == 2) {
      r.x[0] = x[0]; r.x[1] = x[1];
      // z: the reference lerps three zeros: 0*mu0 + 0*mu1 + 0*mu2 (NaN if a mu is not finite, as there)
      r.x[2] = 0.0 * mu[0] + 0.0 * mu[1] + 0.0 * mu[2];
      r.t = x[2];
    } else { r.x[0] = x[0]; r.x[1] = x[1]; r.x[2] = x[2]; r.t = x[3]; }
  }
  if (p.S[0]) {
    double acc = p.S[slice[0]][at[0]] * mu[0];
    for (int i = 1; i < N; i ++) acc = acc + p.S[slice[i]][at[i]] * mu[i];
    r.scalar[0] = acc;
  }
  const bool have_j = p.J[0] != nullptr || p.derive_jacobian;
  if constexpr (ND == 2) {
    if (p.compute_degrees) {                                       // 2d:653-662
      if (fan.ordinal[type]) {
        int deg = orientation2(X, ids);
        deg *= (type == 4) ? 1 : -1;
        r.type = deg == 1 ? 1u : 2u;
      } else r.type = 0u;
    } else {
      double J[2][2] = {{0, 0}, {0, 0}};
      if (have_j) {
        double Js[3][2][2];
        for (int i = 0; i < 3; i ++) {
          if (p.J[0]) { for (int j = 0; j < 2; j ++) for (int k = 0; k < 2; k ++) Js[i][j][k] = p.J[slice[i]][at[i] * 4 + (size_t)j * 2 + k]; }
          else derive_jacobian_at<2>(p, p.V[slice[i]], vx[i], Js[i]);
        }
        for (int j = 0; j < 2; j ++) for (int k = 0; k < 2; k ++)
          J[j][k] = Js[0][j][k] * mu[0] + Js[1][j][k] * mu[1] + Js[2][j][k] * mu[2];
        const double s = 0.5 * (J[0][1] + J[1][0]);                // make_symmetric2x2, always (2d:669)
        J[0][1] = J[1][0] = s;
      }
      r.type = classify2(J[0][0], J[0][1], J[1][0], J[1][1], p.jacobian_symmetric != 0);
    }
    if (p.use_type_filter && !(p.type_filter & r.type)) return false;   // 2d:280
  } else {
    double J[3][3];
    double Js[4][3][3];
    for (int i = 0; i < 4; i ++) {
      if (p.J[0]) { for (int j = 0; j < 3; j ++) for (int k = 0; k < 3; k ++) Js[i][j][k] = p.J[slice[i]][at[i] * 9 + (size_t)j * 3 + k]; }
      else if (p.derive_jacobian) derive_jacobian_at<3>(p, p.V[slice[i]], vx[i], Js[i]);
      else { for (int j = 0; j < 3; j ++) for (int k = 0; k < 3; k ++) Js[i][j][k] = 0.0; }
    }
    for (int j = 0; j < 3; j ++) for (int k = 0; k < 3; k ++) {   // lerp_s3m3x3 accumulates from 0 (linear_interpolation.hh:141-151)
      double acc = 0.0;
      for (int i = 0; i < 4; i ++) acc += Js[i][j][k] * mu[i];
      J[j][k] = acc;
    }
    r.type = classify3(J, p.jacobian_symmetric != 0);
  }
  r.tag = element_tag<ND>(p, corner, type, work_index);
  *out = r;
  // aux word in the struct's padding (include/ftkx.h): bit 0 = ordinal, bits 1.. = emitting timestep
  reinterpret_cast<unsigned int *>(out)[15] = (unsigned)fan.ordinal[type] | ((unsigned)p.t << 1);
  return true;
}

// ---------------------------------------------------------------------------------------------------------------
// the sweep kernel
// ---------------------------------------------------------------------------------------------------------------
template <int ND>
__global__ __launch_bounds__(kThreads) void sweep_kernel(const SweepParams p)
{
  using cfg = tile_cfg<ND>;
  constexpr int N = ND + 1;
  constexpr int HX = cfg::TX + 1, HY = cfg::TY + 1, HZ = (ND == 3) ? cfg::TZ + 1 : 1;
  constexpr int NH = HX * HY * HZ;
  constexpr int NORD = fan_table<N>::NORD, NINT = fan_table<N>::NINT;
  static_assert(cfg::TX * cfg::TY * cfg::TZ == kThreads, "one corner per lane");

  __shared__ i64 s_vf[2][NH][ND];
  __shared__ unsigned char s_mask[2][NH];
  __shared__ unsigned s_tab[fan_table<N>::NTYPES];     // four vertex masks of a type packed in one word
  __shared__ unsigned short s_list[2][kThreads];       // surviving corners: [0] ordinal sweep, [1] interval sweep
  __shared__ unsigned s_cnt[2];

  const int tid = threadIdx.x;
  const fan_table<N> &fan = dev_fan<ND>();

  // workgroup -> tile.  Workgroups are dealt round-robin over the 8 XCDs (b and b+8 share an L2): give each XCD a
  // contiguous run of tiles so that neighbouring tiles' shared halo vertices hit the same L2.
  const unsigned nblocks = gridDim.x;
  unsigned b = blockIdx.x;
  {
    const unsigned per = nblocks / 8, rem = nblocks % 8, xcd = b % 8, k = b / 8;
    // XCD x owns per + (x < rem) tiles
    const unsigned start = xcd * per + (xcd < rem ? xcd : rem);
    b = start + k;
  }
  int tile[3];
  tile[0] = b % p.ntiles[0];
  tile[1] = (b / p.ntiles[0]) % p.ntiles[1];
  tile[2] = b / (p.ntiles[0] * p.ntiles[1]);
  int origin[3] = {p.core_st[0] + tile[0] * cfg::TX, p.core_st[1] + tile[1] * cfg::TY, (ND == 3) ? p.core_st[2] + tile[2] * cfg::TZ : 0};

  const bool need_next = (p.scope_mask & FTKX_SCOPE_INTERVAL) != 0;

  if (tid < fan_table<N>::NTYPES) {
    unsigned w = 0;
    for (int i = 0; i < N; i ++) w |= (unsigned)fan.vert[tid][i] << (8 * i);
    s_tab[tid] = w;
  }
  if (tid < 2) s_cnt[tid] = 0;

  // ---- 1. stage ----
  for (int h = tid; h < 2 * NH; h += kThreads) {
    const int sl = h / NH, hv = h - sl * NH;
    if (sl == 1 && !need_next) break;
    const int hx = hv % HX, hy = (hv / HX) % HY, hz = hv / (HX * HY);
    int vx[3] = {origin[0] + hx, origin[1] + hy, origin[2] + hz};
    bool ok = true;
    for (int d = 0; d < ND; d ++)
      ok = ok && vx[d] >= p.dom_lb[d] && vx[d] <= p.dom_ub[d] && vx[d] >= p.ext_st[d] && vx[d] < p.ext_st[d] + p.ext_sz[d];
    unsigned char m = 0;
    i64 q[ND];
    for (int j = 0; j < ND; j ++) q[j] = 0;
    if (ok) {
      const size_t at = ext_index<ND>(p, vx);
      const double *V = p.V[sl];
      for (int j = 0; j < ND; j ++) {
        const double v = V[at * ND + j];
        if (isnan(v) || isinf(v)) m |= kNonFinite;
        q[j] = quantize(v, p.factor);
        if (q[j] > 0) m |= (unsigned char)(1u << j);
        if (q[j] < 0) m |= (unsigned char)(8u << j);
      }
    } else m = kInvalid;
    for (int j = 0; j < ND; j ++) s_vf[sl][hv][j] = q[j];
    s_mask[sl][hv] = m;
  }
  __syncthreads();

  // ---- 2. cull: one corner per lane ----
  const int cx = tid % cfg::TX, cy = (tid / cfg::TX) % cfg::TY, cz = tid / (cfg::TX * cfg::TY);
  const int corner_sp[3] = {origin[0] + cx, origin[1] + cy, origin[2] + cz};
  bool in_core = true;
  for (int d = 0; d < ND; d ++) in_core = in_core && corner_sp[d] < p.core_st[d] + p.core_sz[d];
  const int hbase = cx + HX * (cy + HY * cz);
  {
    unsigned and0 = 0x3f, and1 = 0x3f;
    for (int c = 0; c < (1 << ND); c ++) {
      const int off = (c & 1) + HX * (((c >> 1) & 1) + HY * ((c >> 2) & 1));
      and0 &= s_mask[0][hbase + off];
      if (need_next) and1 &= s_mask[1][hbase + off];
    }
    // a corner is dropped only by the strict-sign argument; invalid / non-finite vertices are handled per simplex
    const bool keep_o = in_core && (p.scope_mask & FTKX_SCOPE_ORDINAL) && !(p.cull && (and0 & 0x3f));
    const bool keep_i = in_core && need_next && !(p.cull && (and0 & and1 & 0x3f));
    const unsigned long long bo = __ballot(keep_o), bi = __ballot(keep_i);
    const int lane = tid & 63;
    const unsigned long long below = (1ull << lane) - 1ull;
    unsigned base_o = 0, base_i = 0;
    if (lane == 0) {
      if (bo) base_o = atomicAdd(&s_cnt[0], (unsigned)__popcll(bo));
      if (bi) base_i = atomicAdd(&s_cnt[1], (unsigned)__popcll(bi));
    }
    base_o = __shfl(base_o, 0);
    base_i = __shfl(base_i, 0);
    if (keep_o) s_list[0][base_o + __popcll(bo & below)] = (unsigned short)tid;
    if (keep_i) s_list[1][base_i + __popcll(bi & below)] = (unsigned short)tid;
  }
  __syncthreads();

  // ---- 3. test: (corner, type) pairs over all lanes ----
  const unsigned n_o = s_cnt[0], n_i = s_cnt[1];
  const unsigned items_o = n_o * NORD, total = items_o + n_i * NINT;
  unsigned tested = 0, slow = 0;
  for (unsigned base = 0; base < total; base += kThreads) {   // wave-uniform trip count: ballots below stay convergent
    const unsigned w = base + tid;
    bool hit = false;
    ftkx_cp_t rec;
    if (w < total) {
      const bool ordinal = w < items_o;
      const unsigned wl = ordinal ? w : w - items_o;
      const unsigned ci = ordinal ? wl / NORD : wl / NINT;
      const unsigned it = ordinal ? wl % NORD : wl % NINT;
      const int type = ordinal ? fan.ord_types[it] : fan.int_types[it];
      const int ct = s_list[ordinal ? 0 : 1][ci];
      const int ccx = ct % cfg::TX, ccy = (ct / cfg::TX) % cfg::TY, ccz = ct / (cfg::TX * cfg::TY);
      const int hb = ccx + HX * (ccy + HY * ccz);
      const unsigned tab = s_tab[type];
      int hidx[N], hsl[N];
      unsigned m_and = 0x3f, m_or = 0;
      for (int i = 0; i < N; i ++) {
        const unsigned m = (tab >> (8 * i)) & 0xffu;
        hidx[i] = hb + (m & 1) + HX * (((m >> 1) & 1) + ((ND == 3) ? HY * ((m >> 2) & 1) : 0));
        hsl[i] = (m >> ND) & 1;
        const unsigned mk = s_mask[hsl[i]][hidx[i]];
        m_and &= mk; m_or |= mk;
      }
      const bool skip = (m_or & (kInvalid | kNonFinite)) || (p.cull && (m_and & 0x3f));
      if (!skip) {
        tested ++;
        int corner[N];
        corner[0] = origin[0] + ccx; corner[1] = origin[1] + ccy;
        if (ND == 3) corner[2] = origin[2] + ccz;
        corner[ND] = p.t;
        u64 X[N][ND];
        int ids[N];
        for (int i = 0; i < N; i ++) {
          for (int j = 0; j < ND; j ++) X[i][j] = (u64)s_vf[hsl[i]][hidx[i]][j];
          const unsigned m = (tab >> (8 * i)) & 0xffu;
          u64 id = (u64)(i64)(corner[0] + (int)(m & 1) - p.dom_lb[0]);
          for (int d = 1; d < N; d ++) {
            const int rel = corner[d] + (int)((m >> d) & 1) - (d < ND ? p.dom_lb[d] : 0);
            id += (u64)(i64)rel * p.mesh_prod[d];
          }
          ids[i] = (int)id;                                          // regular_tracker.hh:188-194: truncated to int
        }
        // work index inside `core` for this scope (simplicial_regular_mesh.hh:480-493), x fastest
        u64 lin = (u64)(corner[0] - p.core_st[0]);
        {
          u64 stride = (u64)p.core_sz[0];
          for (int d = 1; d < ND; d ++) { lin += (u64)(corner[d] - p.core_st[d]) * stride; stride *= (u64)p.core_sz[d]; }
        }
        const u64 work_index = lin * (u64)(ordinal ? NORD : NINT) + it;
        bool inside;
        double mu[N];
        bool presolved = false;
        if (ND == 3 && !p.robust) {
          // enable_robust_detection == false (3d:465-467): the FP64 solve decides
          double v[N][ND];
          for (int i = 0; i < N; i ++) {
            int vxx[N];
            const unsigned m = (tab >> (8 * i)) & 0xffu;
            for (int d = 0; d < N; d ++) vxx[d] = corner[d] + (int)((m >> d) & 1);
            const size_t at = ext_index<ND>(p, vxx);
            for (int j = 0; j < ND; j ++) v[i][j] = p.V[hsl[i]][at * ND + j];
          }
          if constexpr (ND == 3) inside = solve_barycentric3(v, mu); else inside = false;
          presolved = true;
        } else if constexpr (ND == 2) inside = origin_in_simplex2(X, ids);
        else inside = origin_in_simplex3(X, ids);
        if (inside) hit = make_record<ND>(p, corner, type, work_index, X, ids, presolved, mu, &rec);
      }
    }
    // ---- 4. emit: one atomic per wavefront ----
    const unsigned long long hb = __ballot(hit);
    if (hb) {
      const int lane = tid & 63;
      u64 slot0 = 0;
      if (lane == __ffsll((long long)hb) - 1) slot0 = atomicAdd(&p.counters[CNT_HITS], (u64)__popcll(hb));
      slot0 = __shfl(slot0, __ffsll((long long)hb) - 1);
      if (hit) {
        const u64 slot = slot0 + (u64)__popcll(hb & ((1ull << lane) - 1ull));
        if (slot < p.capacity) p.hits[slot] = rec;
      }
    }
  }
  (void)slow;
  // statistics: one atomic per workgroup per counter
  {
    unsigned t_sum = tested;
    for (int o = 32; o > 0; o >>= 1) t_sum += __shfl_down(t_sum, o);
    if ((tid & 63) == 0 && t_sum) atomicAdd(&p.counters[CNT_SIMPLICES_TESTED], (u64)t_sum);
    if (tid == 0 && (n_o + n_i)) atomicAdd(&p.counters[CNT_CELLS_SURVIVED], (u64)(n_o > n_i ? n_o : n_i));
  }
}

void launch_sweep(const SweepParams &p, hipStream_t stream)
{
  const unsigned nblocks = (unsigned)p.ntiles[0] * p.ntiles[1] * p.ntiles[2];
  if (nblocks == 0) return;
  if (p.nd == 2) hipLaunchKernelGGL(sweep_kernel<2>, dim3(nblocks), dim3(kThreads), 0, stream, p);
  else hipLaunchKernelGGL(sweep_kernel<3>, dim3(nblocks), dim3(kThreads), 0, stream, p);
}

void sweep_tile_dims(int nd, int tile[3])
{
  if (nd == 2) { tile[0] = tile_cfg<2>::TX; tile[1] = tile_cfg<2>::TY; tile[2] = 1; }
  else { tile[0] = tile_cfg<3>::TX; tile[1] = tile_cfg<3>::TY; tile[2] = tile_cfg<3>::TZ; }
}

}  // namespace ftkx
