// Device functions shared by the kernels of the sweep and of the device-driven series pass (series_kernels.hip): field access with the
// reference's derived-field arithmetic (ndarray/grad.hh), vertex classification, element tags, the per-simplex test and the FP64 record
// construction (check_simplex, critical_point_tracker_2d_regular.hh:584-685, ..._3d_regular.hh:425-514).
//
// The kernels replace, behind the reference's accelerator boundary, what update_timestep() -> element_for_{ordinal,interval} ->
// check_simplex does per timestep (critical_point_tracker_2d_regular.hh:263-433, ..._3d_regular.hh:150-308), and what its CUDA back-end
// does with one thread per simplex and a global atomic per hit (src/filters/critical_point_tracer_{2d,3d}_regular.cu).  Where they live:
//   mask_kernels.hip        FAST PATH 1/3 -- a slice is streamed once (S, 8 B/vertex, gradient in flight; V for vector input): one sign
//                           byte per vertex, block summaries, the reduction the scaling factor needs.  HBM-bound, the dominant kernel.
//   cull_exact_kernels.hip  FAST PATH 2/3, 3/3 -- SWAR AND over the 2^(d+1) hypercube on the summaries / mask words; the surviving corners'
//                           simplices through the exact integer predicate (cp_device.hpp); the FP64 half of a record.
//   tile_kernels.hip        TILE PATH -- exact_only, non-robust 3D, odd factors, the overflow regime: every simplex of a tile of corners
//                           staged in LDS.  Integer / FP64 VALU-bound.
//   halo_kernels.hip, dist_kernels.hip   the compact t-slab halo (several ranks);  series_kernels.hip, one_kernel.hip   the device-driven pass.
// No MFMA anywhere: nothing here is a contraction.
#pragma once
#include <hip/hip_runtime.h>
#include <type_traits>
#include <utility>
#include <cstdlib>

#include "cp_device.hpp"
#include "fan_tables.hpp"
#include "sweep_params.hpp"

namespace ftkx {

static __constant__ fan_table<3> c_fan3 = make_fan<3>();
static __constant__ fan_table<4> c_fan4 = make_fan<4>();

template <int ND> struct tile_cfg;
template <> struct tile_cfg<2> { static constexpr int TX = 32, TY = 8, TZ = 1; };
template <> struct tile_cfg<3> { static constexpr int TX = 16, TY = 4, TZ = 4; };

constexpr int kThreads = 256;
constexpr unsigned char kInvalid = 0x80;    // vertex outside the domain / array
constexpr unsigned char kNonFinite = 0x40;  // NaN or Inf component: the reference rejects the simplex (2d:611, 3d:457)
constexpr unsigned char kNeutral = 0x3f;    // cull-neutral mask byte (all six sign bits set)

template <int ND> __device__ inline const fan_table<ND + 1> &dev_fan();
template <> __device__ inline const fan_table<3> &dev_fan<2>() { return c_fan3; }
template <> __device__ inline const fan_table<4> &dev_fan<3>() { return c_fan4; }

// ---------------------------------------------------------------------------------------------------------------
// field access.  Array coordinates (i, j, k) are relative to ext_st.
// ---------------------------------------------------------------------------------------------------------------
__device__ inline int clampi(int i, int lo, int hi) { return i < lo ? lo : (i > hi ? hi : i); }

template <int ND>
__device__ inline size_t arr_index(const Mesh &m, int i, int j, int k)
{
  size_t idx = (size_t)i + (size_t)m.ext_sz[0] * (size_t)j;
  if (ND == 3) idx += (size_t)m.ext_sz[0] * (size_t)m.ext_sz[1] * (size_t)k;
  return idx;
}

// gradient2D / gradient3D of ndarray/grad.hh at one vertex, the same FP64 operations in the same order:
//   2D (grad.hh:17-28): indices clamped to the array, no 0.5, scaled by (D-1);  3D (grad.hh:138-146): 0.5 * central difference
//   on interior vertices only, the array border stays 0.
template <int ND>
__device__ inline void gradient_at(const Mesh &m, const double *__restrict__ S, int i, int j, int k, double g[ND])
{
  const int DW = m.ext_sz[0], DH = m.ext_sz[1];
  if constexpr (ND == 2) {
    const int ip = clampi(i + 1, 0, DW - 1), im = clampi(i - 1, 0, DW - 1), jp = clampi(j + 1, 0, DH - 1), jm = clampi(j - 1, 0, DH - 1);
    const int ic = clampi(i, 0, DW - 1), jc = clampi(j, 0, DH - 1);
    g[0] = (S[(size_t)ip + (size_t)DW * jc] - S[(size_t)im + (size_t)DW * jc]) * (double)(DW - 1);
    g[1] = (S[(size_t)ic + (size_t)DW * jp] - S[(size_t)ic + (size_t)DW * jm]) * (double)(DH - 1);
  } else {
    const int DD = m.ext_sz[2];
    if (i >= 1 && i < DW - 1 && j >= 1 && j < DH - 1 && k >= 1 && k < DD - 1) {
      const size_t sy = (size_t)DW, sz = (size_t)DW * DH, c = (size_t)i + sy * j + sz * k;
      g[0] = 0.5 * (S[c + 1] - S[c - 1]);
      g[1] = 0.5 * (S[c + sy] - S[c - sy]);
      g[2] = 0.5 * (S[c + sz] - S[c - sz]);
    } else { g[0] = 0.0; g[1] = 0.0; g[2] = 0.0; }
  }
}

// the vector field at array coordinates: stored V, or gradient(S) when the vector field is derived
template <int ND>
__device__ inline void vector_at(const Mesh &m, const double *S, const double *V, int i, int j, int k, double v[ND])
{
  if (m.scalar_mode) gradient_at<ND>(m, S, i, j, k, v);
  else {
    const size_t at = arr_index<ND>(m, i, j, k) * ND;
    for (int c = 0; c < ND; c ++) v[c] = V[at + c];
  }
}

// J at one vertex, derived from V exactly like ndarray/grad.hh (jacobian2D 54-86 incl. its operator precedence and clamped
// indices, jacobian3D 175-212 incl. its interior-only support); Js[j][k] = J(k, j, vertex) as the trackers read it.
template <int ND>
__device__ inline void derive_jacobian_at(const Mesh &m, const double *S, const double *V, int i, int j, int k, double Js[ND][ND])
{
  const int DW = m.ext_sz[0], DH = m.ext_sz[1];
  if constexpr (ND == 2) {
    double xp[2], xm[2], yp[2], ym[2];
    const int ic = clampi(i, 0, DW - 1), jc = clampi(j, 0, DH - 1);
    vector_at<2>(m, S, V, clampi(i + 1, 0, DW - 1), jc, 0, xp);
    vector_at<2>(m, S, V, clampi(i - 1, 0, DW - 1), jc, 0, xm);
    vector_at<2>(m, S, V, ic, clampi(j + 1, 0, DH - 1), 0, yp);
    vector_at<2>(m, S, V, ic, clampi(j - 1, 0, DH - 1), 0, ym);
    const double H00 = xp[0] - xm[0] * (double)(DW - 1), H01 = yp[0] - ym[0] * (double)(DH - 1),
                 H10 = xp[1] - xm[1] * (double)(DW - 1), H11 = yp[1] - ym[1] * (double)(DH - 1);
    Js[0][0] = H00;
    Js[1][1] = H11;
    // jacobian2D<T, true> (scalar input) stores the mean in both off-diagonals; <T, false> leaves them 0 (grad.hh:79-82)
    Js[0][1] = Js[1][0] = m.scalar_mode ? (H01 + H10) * 0.5 : 0.0;
  } else {
    const int DD = m.ext_sz[2];
    const bool interior = i >= 2 && i < DW - 2 && j >= 2 && j < DH - 2 && k >= 2 && k < DD - 2;
    if (!interior) { for (int a = 0; a < 3; a ++) for (int b = 0; b < 3; b ++) Js[a][b] = 0.0; return; }
    double p[3], q[3];
    // J(a, b) = 0.5 * (V_a(x + e_b) - V_a(x - e_b));  Js[b][a] = J(a, b)
    vector_at<3>(m, S, V, i + 1, j, k, p); vector_at<3>(m, S, V, i - 1, j, k, q);
    for (int a = 0; a < 3; a ++) Js[0][a] = 0.5 * (p[a] - q[a]);
    vector_at<3>(m, S, V, i, j + 1, k, p); vector_at<3>(m, S, V, i, j - 1, k, q);
    for (int a = 0; a < 3; a ++) Js[1][a] = 0.5 * (p[a] - q[a]);
    vector_at<3>(m, S, V, i, j, k + 1, p); vector_at<3>(m, S, V, i, j, k - 1, q);
    for (int a = 0; a < 3; a ++) Js[2][a] = 0.5 * (p[a] - q[a]);
  }
}

template <int ND>
__device__ inline bool vertex_usable(const Mesh &m, const int *vx)
{
  bool ok = true;
  for (int d = 0; d < ND; d ++)
    ok = ok && vx[d] >= m.dom_lb[d] && vx[d] <= m.dom_ub[d] && vx[d] >= m.ext_st[d] && vx[d] < m.ext_st[d] + m.ext_sz[d];
  return ok;
}

// a linear index over the core's corners (x fastest) -> the corner.  Cores below 2^32 corners: 32-bit division (a 64-bit one by a run-time
// value is ~100 instructions, three per corner)
template <int ND>
__device__ inline void core_corner(const Mesh &m, u64 lin, int *corner)
{
  u64 cells = 1;
  for (int d = 0; d < ND; d ++) cells *= (u64)m.core_sz[d];
  if (lin < (1ull << 32) && cells < (1ull << 32)) {
    unsigned l = (unsigned)lin;
    for (int d = 0; d < ND; d ++) { const unsigned sz = (unsigned)m.core_sz[d], q = l / sz; corner[d] = m.core_st[d] + (int)(l - q * sz); l = q; }
  } else
    for (int d = 0; d < ND; d ++) { corner[d] = m.core_st[d] + (int)(lin % (u64)m.core_sz[d]); lin /= (u64)m.core_sz[d]; }
}

// quantised vertex + classification byte (bits 0..2 strictly positive, 3..5 strictly negative, kNonFinite, kInvalid)
template <int ND>
__device__ inline unsigned char classify_value(const double *v, double factor, i64 q[ND])
{
  unsigned char mk = 0;
  bool big = false;
  for (int j = 0; j < ND; j ++) {
    if (isnan(v[j]) || isinf(v[j])) mk |= kNonFinite;
    q[j] = quantize(v[j], factor);
    if (q[j] > 0) mk |= (unsigned char)(1u << j);
    if (q[j] < 0) mk |= (unsigned char)(8u << j);
    big = big || q[j] >= safe_m<ND>() || q[j] <= -safe_m<ND>();
  }
  // |q| + 1 > safe_m: a determinant with this vertex may wrap, and the wrapped sign is what the reference reports.
  // Without sign bits the vertex never supports a cull -- the same rule the mask kernels apply in the double domain
  // (|v| >= safe_m / factor  <=>  |trunc(v * factor)| >= safe_m, the factor being a power of two).
  if (big) mk &= (unsigned char)~0x3fu;
  return mk;
}

template <int ND>
__device__ inline unsigned char classify_vertex(const Mesh &m, const double *S, const double *V, double factor, const int *vx, i64 q[ND])
{
  for (int j = 0; j < ND; j ++) q[j] = 0;
  if (!vertex_usable<ND>(m, vx)) return kInvalid;
  double v[ND];
  vector_at<ND>(m, S, V, vx[0] - m.ext_st[0], vx[1] - m.ext_st[1], ND == 3 ? vx[2] - m.ext_st[2] : 0, v);
  return classify_value<ND>(v, factor, q);
}

// e.to_integer(m), mesh/simplicial_regular_mesh.hh:496-502
template <int ND>
__device__ inline u64 element_tag(const Mesh &m, const int *corner /*ND spatial + time*/, int type, u64 work_index)
{
  constexpr int N = ND + 1;
  constexpr int ntypes_all = fan_table<N>::NTYPES;
  if (m.tag_mode == FTKX_TAG_WORK_INDEX) return work_index;
  u64 ci = 0;
  for (int i = 0; i < N; i ++) {
    const int rel = corner[i] - (i < ND ? m.dom_lb[i] : 0);
    if (m.tag_mode == FTKX_TAG_REFERENCE) ci += (u64)(i64)(int)((unsigned)rel * (unsigned)m.dimprod[i]);   // int * int, wraps
    else ci += (u64)(i64)rel * m.exact_prod[i];
  }
  return ci * (u64)ntypes_all + (u64)type;
}

// SoS vertex id: m.get_lattice().to_integer(vertex) truncated to int (regular_tracker.hh:188-194, lattice.hh:196-207)
template <int ND>
__device__ inline int vertex_id(const Mesh &m, const int *corner, unsigned vmask)
{
  u64 id = (u64)(i64)(corner[0] + (int)(vmask & 1) - m.dom_lb[0]);
  for (int d = 1; d <= ND; d ++) {
    const int rel = corner[d] + (int)((vmask >> d) & 1) - (d < ND ? m.dom_lb[d] : 0);
    id += (u64)(i64)rel * m.mesh_prod[d];
  }
  return (int)id;
}

// work index inside `core` for one scope (simplicial_regular_mesh.hh:480-493), x fastest
template <int ND>
__device__ inline u64 core_linear(const Mesh &m, const int *corner)
{
  u64 lin = (u64)(corner[0] - m.core_st[0]);
  u64 stride = (u64)m.core_sz[0];
  for (int d = 1; d < ND; d ++) { lin += (u64)(corner[d] - m.core_st[d]) * stride; stride *= (u64)m.core_sz[d]; }
  return lin;
}

// ---------------------------------------------------------------------------------------------------------------
// hit path (rare): everything in FP64 from HBM.  Returns false when the 2D type filter drops the record.
// ---------------------------------------------------------------------------------------------------------------
// FAST = the common case, straight-line: scalar input, J derived in flight, every vertex at least two vertices away from the array
// border (what a domain of [2, D-3] guarantees).  All the index clamps and interior tests of gradient_at / derive_jacobian_at are
// then identities, and without them the ~170 loads of a 3D record (a radius-2 star of S around each of the 4 vertices) are issued
// back to back instead of behind ~90 dependent waits -- the record kernel is one memory-latency chain per record, and that chain was
// 40-47 us long for ANY number of records.  Same FP64 operations in the same order as the general path.
template <int ND>
__device__ inline void gather_fast(const Mesh &m, const double *__restrict__ S, int i, int j, int k, double v[ND], double Js[ND][ND], double &sc)
{
  const int DW = m.ext_sz[0], DH = m.ext_sz[1];
  if constexpr (ND == 2) {
    const double wx = (double)(DW - 1), wy = (double)(DH - 1);
    const size_t sy = (size_t)DW, c = (size_t)i + sy * (size_t)j;
    auto grad = [&](size_t at, double g[2]) {
      g[0] = (S[at + 1] - S[at - 1]) * wx;
      g[1] = (S[at + sy] - S[at - sy]) * wy;
    };
    double xp[2], xm[2], yp[2], ym[2];
    grad(c, v); grad(c + 1, xp); grad(c - 1, xm); grad(c + sy, yp); grad(c - sy, ym);
    sc = S[c];
    const double H00 = xp[0] - xm[0] * wx, H01 = yp[0] - ym[0] * wy, H10 = xp[1] - xm[1] * wx, H11 = yp[1] - ym[1] * wy;
    Js[0][0] = H00;
    Js[1][1] = H11;
    Js[0][1] = Js[1][0] = (H01 + H10) * 0.5;
  } else {
    const size_t sy = (size_t)DW, sz = (size_t)DW * (size_t)DH, c = (size_t)i + sy * (size_t)j + sz * (size_t)k;
    auto grad = [&](size_t at, double g[3]) {
      g[0] = 0.5 * (S[at + 1] - S[at - 1]);
      g[1] = 0.5 * (S[at + sy] - S[at - sy]);
      g[2] = 0.5 * (S[at + sz] - S[at - sz]);
    };
    double p[3][3], q[3][3];
    grad(c, v);
    grad(c + 1, p[0]); grad(c - 1, q[0]); grad(c + sy, p[1]); grad(c - sy, q[1]); grad(c + sz, p[2]); grad(c - sz, q[2]);
    sc = S[c];
#pragma unroll
    for (int b = 0; b < 3; b ++)
#pragma unroll
      for (int a = 0; a < 3; a ++) Js[b][a] = 0.5 * (p[b][a] - q[b][a]);
  }
}

// may this record take the straight-line gather?
template <int ND>
__device__ inline bool record_is_fast(const Mesh &m, const Fields &f, const int *corner)
{
  bool ok = m.scalar_mode && f.J[0] == nullptr && m.derive_jacobian && f.S[0] != nullptr && !(ND == 2 && m.compute_degrees) && !m.record_general;
  // the simplex's vertices are corner + {0, 1} per axis: corner - 2 .. corner + 3 must lie inside the array
  for (int d = 0; d < ND; d ++) { const int a = corner[d] - m.ext_st[d]; ok = ok && a >= 2 && a + 3 < m.ext_sz[d]; }
  return ok;
}

template <int ND, bool FAST>
__device__ inline bool make_record_impl(const Mesh &m, const Fields &f, const int *corner, int type,
                                         const u64 (*X)[ND], const int *ids, bool presolved, const double *mu_in, ftkx_cp_t *out,
                                         bool *fragile, double *Jfrag /* [9], written when *fragile */)
{
  constexpr int N = ND + 1;
  const fan_table<N> &fan = dev_fan<ND>();
  int vx[N][N], ai[N][3], slice[N];
  double v[N][ND], Js[N][ND][ND], sc[N];
  const bool have_j = FAST || f.J[0] != nullptr || m.derive_jacobian;
  const bool want_j = ND == 3 || (have_j && !m.compute_degrees);
  // (both slice pointers and the simplex's vertex masks up front: a pointer picked by index, f.S[slice], is one more dependent load
  // per vertex in a kernel that is nothing but a chain of memory latencies)
  const double *const S0 = f.S[0], *const S1 = f.S[1];
  unsigned vms[N];
#pragma unroll
  for (int i = 0; i < N; i ++) vms[i] = fan.vert[type][i];
#pragma unroll
  for (int i = 0; i < N; i ++) {
    const unsigned vm = vms[i];
#pragma unroll
    for (int d = 0; d < N; d ++) vx[i][d] = corner[d] + ((vm >> d) & 1u);
    slice[i] = (vm >> ND) & 1u;
    ai[i][0] = vx[i][0] - m.ext_st[0]; ai[i][1] = vx[i][1] - m.ext_st[1]; ai[i][2] = ND == 3 ? vx[i][2] - m.ext_st[2] : 0;
    if constexpr (FAST) gather_fast<ND>(m, slice[i] ? S1 : S0, ai[i][0], ai[i][1], ai[i][2], v[i], Js[i], sc[i]);
    else {
      vector_at<ND>(m, f.S[slice[i]], f.V[slice[i]], ai[i][0], ai[i][1], ai[i][2], v[i]);
      sc[i] = f.S[0] ? f.S[slice[i]][arr_index<ND>(m, ai[i][0], ai[i][1], ai[i][2])] : 0.0;
      if (want_j) {
        if (f.J[0]) {
          const size_t at = arr_index<ND>(m, ai[i][0], ai[i][1], ai[i][2]) * (size_t)(ND * ND);
#pragma unroll
          for (int j = 0; j < ND; j ++) for (int k = 0; k < ND; k ++) Js[i][j][k] = f.J[slice[i]][at + (size_t)j * ND + k];
        } else if (m.derive_jacobian) derive_jacobian_at<ND>(m, f.S[slice[i]], f.V[slice[i]], ai[i][0], ai[i][1], ai[i][2], Js[i]);
        else { for (int j = 0; j < ND; j ++) for (int k = 0; k < ND; k ++) Js[i][j][k] = 0.0; }
      }
    }
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
    // lerp of the vertex coordinates, left to right (linear_interpolation.hh:83-101, 129-139).  simplex_coordinates (2d:494-527,
    // 3d:342-378): lattice integers; image bounds ((v - array_lb) / double(array_size - 1)) * (b1 - b0) + b0; rectilinear
    // coords[axis][v]; explicit coords(c, x, y) -- which the 3D tracker also reads with three indices (the z = 0 plane) while
    // reporting the vertex's z index as its time (3d:371-376): reproduced as written.
    double X[N][4];
#pragma unroll
    for (int i = 0; i < N; i ++) {
      X[i][2] = 0.0;
      X[i][3] = (double)vx[i][ND];
      if (m.coords_mode == 1) {
#pragma unroll
        for (int d = 0; d < ND; d ++)
          X[i][d] = ((double)(unsigned long long)(vx[i][d] - m.ext_st[d]) / (double)(m.ext_sz[d] - 1)) * (m.coords_bounds[2 * d + 1] - m.coords_bounds[2 * d]) + m.coords_bounds[2 * d];
      } else if (m.coords_mode == 2) {
#pragma unroll
        for (int d = 0; d < ND; d ++) X[i][d] = m.coords_rect[d][vx[i][d]];
      } else if (m.coords_mode == 3) {
        const size_t at = (size_t)m.coords_expl_ncomp * ((size_t)vx[i][0] + (size_t)m.coords_expl_n0 * (size_t)vx[i][1]);
        X[i][0] = m.coords_expl[at]; X[i][1] = m.coords_expl[at + 1];
        if constexpr (ND == 2) X[i][2] = m.coords_expl_ncomp > 2 ? m.coords_expl[at + 2] : 0.0;
        else { X[i][2] = m.coords_expl[at + 2]; X[i][3] = (double)vx[i][2]; }
      } else {
#pragma unroll
        for (int d = 0; d < ND; d ++) X[i][d] = (double)vx[i][d];
      }
    }
    double x[4];
#pragma unroll
    for (int d = 0; d < 4; d ++) {
      double acc = X[0][d] * mu[0];
#pragma unroll
      for (int i = 1; i < N; i ++) acc = acc + X[i][d] * mu[i];
      x[d] = acc;
    }
    r.x[0] = x[0]; r.x[1] = x[1]; r.x[2] = x[2]; r.t = x[3];      // 2D: x[2] lerps three zeros unless explicit coordinates carry a z
  }
  if (FAST || f.S[0]) {
    double acc = sc[0] * mu[0];
#pragma unroll
    for (int i = 1; i < N; i ++) acc = acc + sc[i] * mu[i];
    r.scalar[0] = acc;
  }
  if constexpr (ND == 2) {
    if (!FAST && m.compute_degrees) {                              // 2d:653-662 (on the quantised vectors X)
      if (fan.ordinal[type]) {
        int deg = orientation2(X, ids);
        deg *= (type == 4) ? 1 : -1;
        r.type = deg == 1 ? 1u : 2u;
      } else r.type = 0u;
    } else {
      double J[2][2] = {{0, 0}, {0, 0}};
      if (have_j) {
#pragma unroll
        for (int j = 0; j < 2; j ++) for (int k = 0; k < 2; k ++)
          J[j][k] = Js[0][j][k] * mu[0] + Js[1][j][k] * mu[1] + Js[2][j][k] * mu[2];
        const double s = 0.5 * (J[0][1] + J[1][0]);                // make_symmetric2x2, always (2d:669)
        J[0][1] = J[1][0] = s;
      }
      r.type = classify2(J[0][0], J[0][1], J[1][0], J[1][1], m.jacobian_symmetric != 0);
    }
    if (m.use_type_filter && !(m.type_filter & r.type)) return false;   // 2d:280
  } else {
    double J[3][3];
#pragma unroll
    for (int j = 0; j < 3; j ++) for (int k = 0; k < 3; k ++) {   // lerp_s3m3x3 accumulates from 0 (linear_interpolation.hh:141-151)
      double acc = 0.0;
#pragma unroll
      for (int i = 0; i < 4; i ++) acc += Js[i][j][k] * mu[i];
      J[j][k] = acc;
    }
    bool frag = false;
    r.type = classify3(J, m.jacobian_symmetric != 0, &frag);
    if (frag) {
      *fragile = true;
#pragma unroll
      for (int j = 0; j < 3; j ++)
#pragma unroll
        for (int k = 0; k < 3; k ++) Jfrag[3 * j + k] = J[j][k];
    }
  }
  const bool ordinal = fan.ordinal[type] != 0;
  const u64 work_index = core_linear<ND>(m, corner) * (u64)(ordinal ? fan_table<N>::NORD : fan_table<N>::NINT) + fan.local_index[type];
  r.tag = element_tag<ND>(m, corner, type, work_index);
  *out = r;
  // aux word in the struct's padding (include/ftkx.h): bit 0 = ordinal, bits 1.. = emitting timestep
  reinterpret_cast<unsigned int *>(out)[15] = (unsigned)ordinal | ((unsigned)f.t << 1);
  return true;
}

// the general path stays a call (rare, and large); the straight-line one is inlined into the record kernel, where the mesh lives in
// scalar registers instead of behind a reference
template <int ND>
__device__ __forceinline__ bool make_record_general(const Mesh &m, const Fields &f, const int *corner, int type,
                                                 const u64 (*X)[ND], const int *ids, bool presolved, const double *mu_in, ftkx_cp_t *out,
                                                 bool *fragile, double *Jfrag)
{
  return make_record_impl<ND, false>(m, f, corner, type, X, ids, presolved, mu_in, out, fragile, Jfrag);
}

// one simplex: vertices already classified/quantised (flags[i], X[i]).  Returns whether the origin is inside (robust integer
// test; or the FP64 solve when enable_robust_detection is off, in which case mu is filled and *presolved set).
// narrow: every quantised component the caller staged fits in 32 bits (checked while staging: fits_s32) -- the cheaper multiplies apply
template <int ND>
__device__ inline bool simplex_inside(const Mesh &m, const Fields &f, int cull, const int *corner, unsigned tab,
                                      const unsigned char *flags, const u64 (*X)[ND], unsigned &tested, int *ids, double *mu, bool *presolved, bool narrow = false,
                                      bool *degenerate = nullptr /* non-null: a degenerate value is reported instead of taken through the cascade here */)
{
  constexpr int N = ND + 1;
  unsigned m_and = 0x3f, m_or = 0;
  for (int i = 0; i < N; i ++) { m_and &= flags[i]; m_or |= flags[i]; }
  *presolved = false;
  if ((m_or & (kInvalid | kNonFinite)) || (cull && (m_and & 0x3f))) return false;
  tested ++;
  if (ND == 3 && !m.robust) {
    for (int i = 0; i < N; i ++) ids[i] = vertex_id<ND>(m, corner, (tab >> (8 * i)) & 0xffu);
    // enable_robust_detection == false (3d:465-467): the FP64 solve decides
    double v[N][ND];
    for (int i = 0; i < N; i ++) {
      const unsigned vm = (tab >> (8 * i)) & 0xffu;
      const int sl = (vm >> ND) & 1;
      vector_at<ND>(m, f.S[sl], f.V[sl], corner[0] + (int)(vm & 1) - m.ext_st[0], corner[1] + (int)((vm >> 1) & 1) - m.ext_st[1],
                    ND == 3 ? corner[2] + (int)((vm >> 2) & 1) - m.ext_st[2] : 0, v[i]);
    }
    *presolved = true;
    if constexpr (ND == 3) return solve_barycentric3(v, mu); else return false;
  }
  // the sort-free test on cofactors; a degenerate value (a zero, INT64_MIN) takes the literal cascade, and only that needs the SoS vertex
  // ids (regular_tracker.hh:188-194: four 64-bit multiply-adds per vertex -- computed for every simplex they cost as much as the test)
  int r;
  if constexpr (ND == 2) r = origin_in_simplex2_try(X, narrow); else r = origin_in_simplex3_try(X, narrow);
  if (r >= 0) return r != 0;
  if (degenerate) { *degenerate = true; return false; }
  for (int i = 0; i < N; i ++) ids[i] = vertex_id<ND>(m, corner, (tab >> (8 * i)) & 0xffu);
  return sos_origin_in_simplex_resolved<ND>(X, ids);
}

// hits of one wavefront appended with a single atomic (must be reached by all 64 lanes)
// returns the record's slot in the hit buffer (~0 if it was not stored)
__device__ inline u64 emit_hits(const Mesh &m, bool hit, const ftkx_cp_t &rec)
{
  const unsigned long long hb = __ballot(hit);
  if (!hb) return ~0ull;
  const int lane = threadIdx.x & 63;
  const int leader = __ffsll((long long)hb) - 1;
  u64 slot0 = 0;
  if (lane == leader) slot0 = atomicAdd(&m.counters[CNT_HITS], (u64)__popcll(hb));
  slot0 = __shfl(slot0, leader);
  if (hit) {
    const u64 slot = slot0 + (u64)__popcll(hb & ((1ull << lane) - 1ull));
    if (slot < m.capacity) { m.hits[slot] = rec; return slot; }
  }
  return ~0ull;
}

// simplices that passed the test: appended for record_kernel with a single atomic per wavefront (must be reached by all 64 lanes)
__device__ inline void emit_pass(const Mesh &m, bool hit, u64 desc)
{
  const unsigned long long hb = __ballot(hit);
  if (!hb) return;
  const int lane = threadIdx.x & 63;
  const int leader = __ffsll((long long)hb) - 1;
  u64 slot0 = 0;
  if (lane == leader) slot0 = atomicAdd(&m.counters[CNT_PASS], (u64)__popcll(hb));
  slot0 = __shfl(slot0, leader);
  if (hit) {
    const u64 slot = slot0 + (u64)__popcll(hb & ((1ull << lane) - 1ull));
    if (slot < m.capacity) m.pass[slot] = desc;
  }
}

}  // namespace ftkx
