// Per-simplex arithmetic of the critical-point sweep, written for the CDNA4 VALU (64-bit integer ring arithmetic,
// FP64 without contraction).  Every function is FTKX_HD so that tests/hostcheck can run the very same code on the
// host against the oracle; the product only ever runs it inside the HIP kernels.
//
// Reference semantics being reproduced (paths under /root/reference/include/ftk/):
//   integer predicate   numeric/critical_point_test.hh:20-34, numeric/sign_det.hh:44-289, 360-414, numeric/det.hh:9-55
//   inverse lerp        numeric/inverse_linear_interpolation_solver.hh:32-54, 143-167; linear_solver.hh:12-34;
//                       matrix_inverse.hh:25-45; matrix_multiplication.hh:55-60
//   clamp               numeric/clamp.hh:15-37
//   classification      numeric/critical_point_type.hh:40-93; eigen_solver2.hh:20-41, 61-67; eigen_solver3.hh:20-47;
//                       characteristic_polynomial.hh:12-17, 40-47; quadratic_solver.hh:14-25; symmetric_matrix.hh:10-15
//
// Formulation (not the reference's): the reference evaluates five (d+1)x(d+1) homogeneous determinants per simplex, each
// behind a bubble sort of the vertex ids.  Here the d+1 "vertex replaced by the origin" determinants are computed once as
// signed dxd cofactors C_i, the full determinant is their sum, and NO sorting happens unless one of those d+2 values is
// zero: a determinant is antisymmetric in its rows, so sign(det(sorted rows)) * (-1)^swaps == sign(det(rows)) whenever it is
// non-zero.  All integer arithmetic is modulo 2^64, where these are polynomial identities, so the wrapped values -- and hence
// the signs the reference reads off its overflowing int64 -- are reproduced exactly ("int64 wrap").
// Only degenerate simplices (a zero, or the one value whose negation wraps onto itself) take the literal cascade.
#pragma once

#include <math.h>
#include <float.h>

#if defined(__HIPCC__)
#define FTKX_HD __host__ __device__
#else
#define FTKX_HD
#endif

namespace ftkx {

typedef unsigned long long u64;
typedef long long i64;

constexpr u64 kMinI64 = 0x8000000000000000ull;

FTKX_HD inline int sgn_wrapped(u64 x) { const i64 s = (i64)x; return (s > 0) - (s < 0); }
FTKX_HD inline bool degenerate_value(u64 x) { return x == 0 || x == kMinI64; }

// (int64_t)(v * factor) with x86-64's out-of-range result (0x8000...0), critical_point_tracker_2d_regular.hh:605-616
FTKX_HD inline i64 quantize(double v, double factor)
{
  const double p = v * factor;
  if (!(p > -9223372036854775808.0 && p < 9223372036854775808.0)) return (i64)kMinI64;
  return (i64)p;
}

// ---------------------------------------------------------------------------------------------------------------
// literal Simulation-of-Simplicity cascades (slow path; rows already sorted by ascending vertex id)
// ---------------------------------------------------------------------------------------------------------------
FTKX_HD inline u64 h2(u64 a, u64 b) { return a - b; }                       // | a 1 ; b 1 |
FTKX_HD inline u64 h3(u64 a0, u64 a1, u64 b0, u64 b1, u64 c0, u64 c1)       // | a0 a1 1 ; b0 b1 1 ; c0 c1 1 |
{
  return a0 * (b1 - c1) - a1 * (b0 - c0) + (b0 * c1 - b1 * c0);
}

FTKX_HD inline int sos_sign3(const u64 X[3][2])   // sign_det.hh:44-90
{
  int s;
  if ((s = sgn_wrapped(h3(X[0][0], X[0][1], X[1][0], X[1][1], X[2][0], X[2][1])))) return s;
  if ((s = -sgn_wrapped(h2(X[1][0], X[2][0])))) return s;
  if ((s = sgn_wrapped(h2(X[1][1], X[2][1])))) return s;
  if ((s = sgn_wrapped(h2(X[0][0], X[2][0])))) return s;
  return 1;
}

FTKX_HD inline u64 det3x3(const u64 a[3], const u64 b[3], const u64 c[3])
{
  return a[0] * (b[1] * c[2] - b[2] * c[1]) - a[1] * (b[0] * c[2] - b[2] * c[0]) + a[2] * (b[0] * c[1] - b[1] * c[0]);
}

FTKX_HD inline int sos_sign4(const u64 X[4][3])   // sign_det.hh:92-200
{
  int s;
  {
    // | X 1 | expanded along the column of ones: sum of the four signed 3x3 minors
    const u64 d = det3x3(X[0], X[1], X[2]) - det3x3(X[0], X[1], X[3]) + det3x3(X[0], X[2], X[3]) - det3x3(X[1], X[2], X[3]);
    if ((s = sgn_wrapped(d))) return s;
  }
#define FTKX_H3(r0, r1, r2, c0, c1) h3(X[r0][c0], X[r0][c1], X[r1][c0], X[r1][c1], X[r2][c0], X[r2][c1])
  if ((s =  sgn_wrapped(FTKX_H3(1, 2, 3, 0, 1)))) return s;
  if ((s = -sgn_wrapped(FTKX_H3(1, 2, 3, 0, 2)))) return s;
  if ((s =  sgn_wrapped(FTKX_H3(1, 2, 3, 1, 2)))) return s;
  if ((s = -sgn_wrapped(FTKX_H3(0, 2, 3, 0, 1)))) return s;
  if ((s =  sgn_wrapped(h2(X[2][0], X[3][0])))) return s;
  if ((s = -sgn_wrapped(h2(X[2][1], X[3][1])))) return s;
  if ((s =  sgn_wrapped(FTKX_H3(0, 2, 3, 0, 2)))) return s;
  if ((s =  sgn_wrapped(h2(X[2][2], X[3][2])))) return s;
  if ((s = -sgn_wrapped(FTKX_H3(0, 2, 3, 1, 2)))) return s;
  if ((s =  sgn_wrapped(FTKX_H3(0, 1, 3, 0, 1)))) return s;
  if ((s = -sgn_wrapped(h2(X[1][0], X[3][0])))) return s;
  if ((s =  sgn_wrapped(h2(X[1][1], X[3][1])))) return s;
  if ((s =  sgn_wrapped(h2(X[0][0], X[3][0])))) return s;
#undef FTKX_H3
  return 1;
}

// orientation with the reference's sort-by-id + swap-parity convention (positive2/positive3, sign_det.hh:243-289).
// ND+1 rows; a row with id -1 is the origin.
template <int ND>
FTKX_HD inline int sos_orientation(const u64 X[ND + 1][ND], const int id[ND + 1])
{
  constexpr int n = ND + 1;
  // the reference's bubble sort by vertex id (its swap count's parity is what matters), on the ROWS themselves with compile-time indices:
  // sorting an index array and gathering the rows through it puts all of this into scratch memory on the device -- one cascade was 16 us
  int key[n];
  u64 R[n][ND];
#pragma unroll
  for (int i = 0; i < n; i ++) {
    key[i] = id[i];
#pragma unroll
    for (int j = 0; j < ND; j ++) R[i][j] = X[i][j];
  }
  int swaps = 0;
#pragma unroll
  for (int i = 0; i < n - 1; i ++)
#pragma unroll
    for (int j = 0; j < n - i - 1; j ++) {
      const bool sw = key[j] > key[j + 1];
      const int ka = key[j], kb = key[j + 1];
      key[j] = sw ? kb : ka; key[j + 1] = sw ? ka : kb;
#pragma unroll
      for (int c = 0; c < ND; c ++) { const u64 a = R[j][c], b = R[j + 1][c]; R[j][c] = sw ? b : a; R[j + 1][c] = sw ? a : b; }
      swaps += sw ? 1 : 0;
    }
  int d;
  if constexpr (ND == 2) d = sos_sign3(R); else d = sos_sign4(R);
  return (swaps & 1) ? -d : d;
}

// literal robust_critical_point_in_simplex{2,3}: used when the fast path meets a degenerate value
template <int ND>
FTKX_HD inline bool sos_origin_in_simplex(const u64 X[ND + 1][ND], const int id[ND + 1])
{
  constexpr int n = ND + 1;
  const int s = sos_orientation<ND>(X, id);
#pragma unroll
  for (int i = 0; i < n; i ++) {
    u64 Y[n][ND]; int yid[n];
#pragma unroll
    for (int j = 0; j < n; j ++) {
      yid[j] = (j == i) ? -1 : id[j];
#pragma unroll
      for (int k = 0; k < ND; k ++) Y[j][k] = (j == i) ? 0ull : X[j][k];
    }
    if (sos_orientation<ND>(Y, yid) != s) return false;
  }
  return true;
}

// The cascade only where it is needed.  A value that is neither 0 nor INT64_MIN has its own sign under the reference's convention too:
// sorting the rows by vertex id negates the determinant once per swap -- exactly, in Z / 2^64 --, and sos_sign3 / sos_sign4 return the
// sign of their first determinant when it is not zero.  So of the d + 2 determinants of a simplex (the full one, and one per vertex
// replaced by the origin) only the degenerate ones go through sos_orientation: typically one, where the literal form above sorts and
// expands all of them again.  Same result (tests/test_host_numerics.py: against the literal cascade from all-degenerate to all-wrapped).
template <int ND>
FTKX_HD inline bool sos_origin_in_simplex_resolved(const u64 X[ND + 1][ND], const int id[ND + 1])
{
  constexpr int n = ND + 1;
  u64 c[n], d;
  if constexpr (ND == 2) {
    c[0] = X[1][0] * X[2][1] - X[1][1] * X[2][0];
    c[1] = X[2][0] * X[0][1] - X[2][1] * X[0][0];
    c[2] = X[0][0] * X[1][1] - X[0][1] * X[1][0];
    d = c[0] + c[1] + c[2];
  } else {
    const u64 p_yz = X[2][1] * X[3][2] - X[2][2] * X[3][1], p_zx = X[2][2] * X[3][0] - X[2][0] * X[3][2], p_xy = X[2][0] * X[3][1] - X[2][1] * X[3][0];
    const u64 q_yz = X[0][1] * X[1][2] - X[0][2] * X[1][1], q_zx = X[0][2] * X[1][0] - X[0][0] * X[1][2], q_xy = X[0][0] * X[1][1] - X[0][1] * X[1][0];
    const u64 m0 = X[1][0] * p_yz + X[1][1] * p_zx + X[1][2] * p_xy, m1 = X[0][0] * p_yz + X[0][1] * p_zx + X[0][2] * p_xy;
    const u64 m2 = X[3][0] * q_yz + X[3][1] * q_zx + X[3][2] * q_xy, m3 = X[2][0] * q_yz + X[2][1] * q_zx + X[2][2] * q_xy;
    c[0] = 0ull - m0; c[1] = m1; c[2] = 0ull - m2; c[3] = m3;
    d = c[0] + c[1] + c[2] + c[3];
  }
  const int s = degenerate_value(d) ? sos_orientation<ND>(X, id) : sgn_wrapped(d);
#pragma unroll
  for (int i = 0; i < n; i ++) {
    int si;
    if (degenerate_value(c[i])) {
      u64 Y[n][ND]; int yid[n];
#pragma unroll
      for (int j = 0; j < n; j ++) {
        yid[j] = (j == i) ? -1 : id[j];
#pragma unroll
        for (int k = 0; k < ND; k ++) Y[j][k] = (j == i) ? 0ull : X[j][k];
      }
      si = sos_orientation<ND>(Y, yid);
    } else si = sgn_wrapped(c[i]);
    if (si != s) return false;
  }
  return true;
}

// ---------------------------------------------------------------------------------------------------------------
// fast path
// ---------------------------------------------------------------------------------------------------------------
// 2D: C_i = det3h(X with row i := origin) = (-1)^i * det2(other two rows in order);  D = C_0 + C_1 + C_2
FTKX_HD inline bool origin_in_simplex2(const u64 X[3][2], const int id[3])
{
  const u64 c0 = X[1][0] * X[2][1] - X[1][1] * X[2][0];
  const u64 c1 = X[2][0] * X[0][1] - X[2][1] * X[0][0];     // = -(X0 x X2)
  const u64 c2 = X[0][0] * X[1][1] - X[0][1] * X[1][0];
  const u64 d = c0 + c1 + c2;
  if (degenerate_value(d) || degenerate_value(c0) || degenerate_value(c1) || degenerate_value(c2))
    return sos_origin_in_simplex<2>(X, id);
  const int s = sgn_wrapped(d);
  return sgn_wrapped(c0) == s && sgn_wrapped(c1) == s && sgn_wrapped(c2) == s;
}

// 3D: C_i = det4h(X with row i := origin) = (-1)^(i+1) * det3(other three rows in order);  D = sum C_i
FTKX_HD inline bool origin_in_simplex3(const u64 X[4][3], const int id[4])
{
  // 2x2 minors of rows (2,3) and rows (0,1), shared by two cofactors each
  const u64 p_yz = X[2][1] * X[3][2] - X[2][2] * X[3][1], p_zx = X[2][2] * X[3][0] - X[2][0] * X[3][2], p_xy = X[2][0] * X[3][1] - X[2][1] * X[3][0];
  const u64 q_yz = X[0][1] * X[1][2] - X[0][2] * X[1][1], q_zx = X[0][2] * X[1][0] - X[0][0] * X[1][2], q_xy = X[0][0] * X[1][1] - X[0][1] * X[1][0];
  const u64 m0 = X[1][0] * p_yz + X[1][1] * p_zx + X[1][2] * p_xy;   // det3(X1, X2, X3)
  const u64 m1 = X[0][0] * p_yz + X[0][1] * p_zx + X[0][2] * p_xy;   // det3(X0, X2, X3)
  const u64 m2 = X[3][0] * q_yz + X[3][1] * q_zx + X[3][2] * q_xy;   // det3(X0, X1, X3)
  const u64 m3 = X[2][0] * q_yz + X[2][1] * q_zx + X[2][2] * q_xy;   // det3(X0, X1, X2)
  const u64 c0 = 0ull - m0, c1 = m1, c2 = 0ull - m2, c3 = m3;
  const u64 d = c0 + c1 + c2 + c3;
  if (degenerate_value(d) || degenerate_value(c0) || degenerate_value(c1) || degenerate_value(c2) || degenerate_value(c3))
    return sos_origin_in_simplex<3>(X, id);
  const int s = sgn_wrapped(d);
  return sgn_wrapped(c0) == s && sgn_wrapped(c1) == s && sgn_wrapped(c2) == s && sgn_wrapped(c3) == s;
}

// The same two tests for quantised components that fit in 32 bits -- every |q| < 2^31, which is the case whenever the cull is legal
// (|q| < 727 041) and far into the overflow regime (nbits 21 on values below 1 024).  Same ring arithmetic, cheaper instructions: gfx950's
// 32-bit integer multiply runs at a quarter of the VALU rate, and a 64 x 64 -> 64-bit product takes four of them.  A product of two
// sign-extended 32-bit values is ONE v_mad_i64_i32, a 64-bit value times a sign-extended 32-bit one two multiplies and a correction: 36
// multiply instructions per 3-simplex instead of 96.  (Written on sign-extended ints so that the compiler sees it; the values are the
// very same elements of Z / 2^64.)
FTKX_HD inline u64 mul_s32(u64 a, u64 b) { return (u64)((i64)(int)a * (i64)(int)b); }            // both operands fit in int32
FTKX_HD inline u64 mul_64_s32(u64 p, u64 b) { return p * (u64)(i64)(int)b; }                    // b fits in int32

FTKX_HD inline bool origin_in_simplex2_s32(const u64 X[3][2], const int id[3])
{
  const u64 c0 = mul_s32(X[1][0], X[2][1]) - mul_s32(X[1][1], X[2][0]);
  const u64 c1 = mul_s32(X[2][0], X[0][1]) - mul_s32(X[2][1], X[0][0]);
  const u64 c2 = mul_s32(X[0][0], X[1][1]) - mul_s32(X[0][1], X[1][0]);
  const u64 d = c0 + c1 + c2;
  if (degenerate_value(d) || degenerate_value(c0) || degenerate_value(c1) || degenerate_value(c2))
    return sos_origin_in_simplex<2>(X, id);
  const int s = sgn_wrapped(d);
  return sgn_wrapped(c0) == s && sgn_wrapped(c1) == s && sgn_wrapped(c2) == s;
}

FTKX_HD inline bool origin_in_simplex3_s32(const u64 X[4][3], const int id[4])
{
  const u64 p_yz = mul_s32(X[2][1], X[3][2]) - mul_s32(X[2][2], X[3][1]), p_zx = mul_s32(X[2][2], X[3][0]) - mul_s32(X[2][0], X[3][2]),
            p_xy = mul_s32(X[2][0], X[3][1]) - mul_s32(X[2][1], X[3][0]);
  const u64 q_yz = mul_s32(X[0][1], X[1][2]) - mul_s32(X[0][2], X[1][1]), q_zx = mul_s32(X[0][2], X[1][0]) - mul_s32(X[0][0], X[1][2]),
            q_xy = mul_s32(X[0][0], X[1][1]) - mul_s32(X[0][1], X[1][0]);
  const u64 m0 = mul_64_s32(p_yz, X[1][0]) + mul_64_s32(p_zx, X[1][1]) + mul_64_s32(p_xy, X[1][2]);   // det3(X1, X2, X3)
  const u64 m1 = mul_64_s32(p_yz, X[0][0]) + mul_64_s32(p_zx, X[0][1]) + mul_64_s32(p_xy, X[0][2]);   // det3(X0, X2, X3)
  const u64 m2 = mul_64_s32(q_yz, X[3][0]) + mul_64_s32(q_zx, X[3][1]) + mul_64_s32(q_xy, X[3][2]);   // det3(X0, X1, X3)
  const u64 m3 = mul_64_s32(q_yz, X[2][0]) + mul_64_s32(q_zx, X[2][1]) + mul_64_s32(q_xy, X[2][2]);   // det3(X0, X1, X2)
  const u64 c0 = 0ull - m0, c1 = m1, c2 = 0ull - m2, c3 = m3;
  const u64 d = c0 + c1 + c2 + c3;
  if (degenerate_value(d) || degenerate_value(c0) || degenerate_value(c1) || degenerate_value(c2) || degenerate_value(c3))
    return sos_origin_in_simplex<3>(X, id);
  const int s = sgn_wrapped(d);
  return sgn_wrapped(c0) == s && sgn_wrapped(c1) == s && sgn_wrapped(c2) == s && sgn_wrapped(c3) == s;
}

FTKX_HD inline bool fits_s32(i64 q) { return q == (i64)(int)q; }

// The fast tests again, WITHOUT the vertex ids: 1 inside, 0 outside, -1 a degenerate value was met.  Only the literal cascade needs the
// SoS vertex ids (four 64-bit multiply-adds each): the kernels compute them when this says -1, not for every simplex.
FTKX_HD inline int origin_in_simplex2_try(const u64 X[3][2], bool narrow)
{
  u64 c0, c1, c2;
  if (narrow) {
    c0 = mul_s32(X[1][0], X[2][1]) - mul_s32(X[1][1], X[2][0]);
    c1 = mul_s32(X[2][0], X[0][1]) - mul_s32(X[2][1], X[0][0]);
    c2 = mul_s32(X[0][0], X[1][1]) - mul_s32(X[0][1], X[1][0]);
  } else {
    c0 = X[1][0] * X[2][1] - X[1][1] * X[2][0];
    c1 = X[2][0] * X[0][1] - X[2][1] * X[0][0];
    c2 = X[0][0] * X[1][1] - X[0][1] * X[1][0];
  }
  const u64 d = c0 + c1 + c2;
  if (degenerate_value(d) || degenerate_value(c0) || degenerate_value(c1) || degenerate_value(c2)) return -1;
  const int s = sgn_wrapped(d);
  return (sgn_wrapped(c0) == s && sgn_wrapped(c1) == s && sgn_wrapped(c2) == s) ? 1 : 0;
}

FTKX_HD inline int origin_in_simplex3_try(const u64 X[4][3], bool narrow)
{
  u64 m0, m1, m2, m3;
  if (narrow) {
    const u64 p_yz = mul_s32(X[2][1], X[3][2]) - mul_s32(X[2][2], X[3][1]), p_zx = mul_s32(X[2][2], X[3][0]) - mul_s32(X[2][0], X[3][2]),
              p_xy = mul_s32(X[2][0], X[3][1]) - mul_s32(X[2][1], X[3][0]);
    const u64 q_yz = mul_s32(X[0][1], X[1][2]) - mul_s32(X[0][2], X[1][1]), q_zx = mul_s32(X[0][2], X[1][0]) - mul_s32(X[0][0], X[1][2]),
              q_xy = mul_s32(X[0][0], X[1][1]) - mul_s32(X[0][1], X[1][0]);
    m0 = mul_64_s32(p_yz, X[1][0]) + mul_64_s32(p_zx, X[1][1]) + mul_64_s32(p_xy, X[1][2]);
    m1 = mul_64_s32(p_yz, X[0][0]) + mul_64_s32(p_zx, X[0][1]) + mul_64_s32(p_xy, X[0][2]);
    m2 = mul_64_s32(q_yz, X[3][0]) + mul_64_s32(q_zx, X[3][1]) + mul_64_s32(q_xy, X[3][2]);
    m3 = mul_64_s32(q_yz, X[2][0]) + mul_64_s32(q_zx, X[2][1]) + mul_64_s32(q_xy, X[2][2]);
  } else {
    const u64 p_yz = X[2][1] * X[3][2] - X[2][2] * X[3][1], p_zx = X[2][2] * X[3][0] - X[2][0] * X[3][2], p_xy = X[2][0] * X[3][1] - X[2][1] * X[3][0];
    const u64 q_yz = X[0][1] * X[1][2] - X[0][2] * X[1][1], q_zx = X[0][2] * X[1][0] - X[0][0] * X[1][2], q_xy = X[0][0] * X[1][1] - X[0][1] * X[1][0];
    m0 = X[1][0] * p_yz + X[1][1] * p_zx + X[1][2] * p_xy;
    m1 = X[0][0] * p_yz + X[0][1] * p_zx + X[0][2] * p_xy;
    m2 = X[3][0] * q_yz + X[3][1] * q_zx + X[3][2] * q_xy;
    m3 = X[2][0] * q_yz + X[2][1] * q_zx + X[2][2] * q_xy;
  }
  const u64 c0 = 0ull - m0, c1 = m1, c2 = 0ull - m2, c3 = m3;
  const u64 d = c0 + c1 + c2 + c3;
  if (degenerate_value(d) || degenerate_value(c0) || degenerate_value(c1) || degenerate_value(c2) || degenerate_value(c3)) return -1;
  const int s = sgn_wrapped(d);
  return (sgn_wrapped(c0) == s && sgn_wrapped(c1) == s && sgn_wrapped(c2) == s && sgn_wrapped(c3) == s) ? 1 : 0;
}

// positive2 (orientation only), for enable_computing_degrees
FTKX_HD inline int orientation2(const u64 X[3][2], const int id[3]) { return sos_orientation<2>(X, id); }

// ---------------------------------------------------------------------------------------------------------------
// FP64 part: evaluated for hits only.  Compile with -ffp-contract=off; the single fused operation is the explicit fma.
// ---------------------------------------------------------------------------------------------------------------
FTKX_HD inline bool solve_barycentric2(const double V[3][2], double mu[3])
{
  const double a00 = V[0][0] - V[2][0], a01 = V[1][0] - V[2][0], a10 = V[0][1] - V[2][1], a11 = V[1][1] - V[2][1];
  const double b0 = -V[2][0], b1 = -V[2][1];
  const double D = a00 * a11 - a10 * a01, Dx = b0 * a11 - a01 * b1, Dy = a00 * b1 - b0 * a10;
  mu[0] = Dx / D;
  mu[1] = Dy / D;
  mu[2] = 1.0 - mu[0] - mu[1];
  const double e = DBL_EPSILON;
  return mu[0] >= -e && mu[0] <= 1.0 + e && mu[1] >= -e && mu[1] <= 1.0 + e && mu[2] >= -e && mu[2] <= 1.0 + e;
}

FTKX_HD inline bool solve_barycentric3(const double V[4][3], double l[4])
{
  double m[3][3], b[3], c[3][3];
  for (int r = 0; r < 3; r ++) {
    for (int k = 0; k < 3; k ++) m[r][k] = V[k][r] - V[3][r];
    b[r] = -V[3][r];
  }
  c[0][0] =  m[1][1] * m[2][2] - m[1][2] * m[2][1];
  c[0][1] = -m[0][1] * m[2][2] + m[0][2] * m[2][1];
  c[0][2] =  m[0][1] * m[1][2] - m[0][2] * m[1][1];
  c[1][0] = -m[1][0] * m[2][2] + m[1][2] * m[2][0];
  c[1][1] =  m[0][0] * m[2][2] - m[0][2] * m[2][0];
  c[1][2] = -m[0][0] * m[1][2] + m[0][2] * m[1][0];
  c[2][0] =  m[1][0] * m[2][1] - m[1][1] * m[2][0];
  c[2][1] = -m[0][0] * m[2][1] + m[0][1] * m[2][0];
  c[2][2] =  m[0][0] * m[1][1] - m[0][1] * m[1][0];
  const double det = m[0][0] * c[0][0] + m[0][1] * c[1][0] + m[0][2] * c[2][0];
  const double inv = 1.0 / det;
  for (int r = 0; r < 3; r ++) for (int k = 0; k < 3; k ++) c[r][k] = c[r][k] * inv;
  for (int r = 0; r < 3; r ++) l[r] = c[r][0] * b[0] + c[r][1] * b[1] + c[r][2] * b[2];
  l[3] = 1.0 - l[0] - l[1] - l[2];
  const double e = DBL_EPSILON;
  return l[0] >= -e && l[0] < 1.0 + e && l[1] >= -e && l[1] < 1.0 + e && l[2] >= -e && l[2] < 1.0 + e && l[3] >= -e && l[3] < 1.0 + e;
}

// clamp.hh:15-37 -- NaN clamps to 0 because std::max(0.0, x) is (0.0 < x) ? x : 0.0
template <int n>
FTKX_HD inline void clamp_barycentric(double x[n])
{
  double sum = 0.0;
  for (int i = 0; i < n; i ++) {
    const double lo = (0.0 < x[i]) ? x[i] : 0.0;
    x[i] = (1.0 < lo) ? 1.0 : lo;
    sum += x[i];
  }
  for (int i = 0; i < n; i ++) x[i] /= sum;
  if (isnan(x[0]) || isinf(x[0]))
    for (int i = 0; i < n; i ++) x[i] = 1.0 / n;
}

FTKX_HD inline unsigned classify2(double j00, double j01, double j10, double j11, bool symmetric)
{
  if (symmetric) {
    const double b = -(j00 + j11), c = j00 * j11 - j10 * j10;
    const double delta = fma(b, b, -4 * c);
    const double sq = delta < 0 ? 0 : sqrt(delta);
    const double e0 = 0.5 * (-b + sq), e1 = 0.5 * (-b - sq);   // the reference also swaps by |.|: irrelevant to the signs
    if (e0 > 0 && e1 > 0) return 2u;
    if (e0 < 0 && e1 < 0) return 8u;
    if (e0 * e1 < 0) return 4u;
    return 1u;
  }
  const double p1 = -(j00 + j11), p0 = j00 * j11 - j10 * j01;
  const double delta = p1 * p1 - 4 * 1.0 * p0;
  if (delta >= 0) {
    const double r0 = (-p1 + sqrt(delta)) / 2.0, r1 = (-p1 - sqrt(delta)) / 2.0;
    if (r0 * r1 < 0) return 4u;
    if (r0 > 0 && r1 > 0) return 2u;
    if (r0 < 0 && r1 < 0) return 8u;
    return 1u;
  }
  // the reference takes a complex square root via pow(z, 1/2) = polar(.., pi/2), whose real part is rho*cos(pi/2) =
  // rho*6.1e-17 rather than 0 (numeric/sqrt.hh:9-15): keep that term, only the sign of the sum is used
  const double re = (-p1 + sqrt(-delta) * 6.123233995736766e-17) / 2.0;
  if (re < 0) return 16u;
  if (re > 0) return 32u;
  return 64u;
}

// `fragile` (optional): set when an eigenvalue is so close to zero, relative to the others, that the last bits of pow / acos / cos
// decide the class.  Those functions are not correctly rounded, and the device library's results differ from the host libm's (the
// one the reference runs on) in the last place now and then -- enough to turn an exactly singular Hessian (plateaus, lattice-aligned
// data: eigenvalue 0.0 on the host, -1e-17 on the device) from "degenerate" into a maximum.  The sweep re-classifies such records on
// the host, with the host's libm (ftkx_api.hip); everything else in the record path is +, -, *, /, sqrt and fma: identical.
FTKX_HD inline unsigned classify3(const double A[3][3], bool symmetric, bool *fragile = nullptr)
{
  if (fragile) *fragile = false;
  if (!symmetric) return 0u;   // critical_point_type.hh:87-91
  const double b = -(A[0][0] + A[1][1] + A[2][2]);
  const double c = A[1][1] * A[2][2] + A[0][0] * A[2][2] + A[0][0] * A[1][1] - A[0][1] * A[1][0] - A[1][2] * A[2][1] - A[0][2] * A[2][0];
  const double d = -(A[0][0] * (A[1][1] * A[2][2] - A[1][2] * A[2][1]) - A[0][1] * (A[1][0] * A[2][2] - A[1][2] * A[2][0])
                   + A[0][2] * (A[1][0] * A[2][1] - A[1][1] * A[2][0]));
  double q = (3.0 * c - (b * b)) / 9.0;
  const double r = (-(27.0 * d) + b * (9.0 * c - 2.0 * (b * b))) / 54.0;
  const double disc = q * q * q + r * r;
  const double term1 = b / 3.0;
  double x0, x1, x2;
  if (disc >= 0) {
    const double r13 = (r < 0) ? -pow(-r, 1.0 / 3.0) : pow(r, 1.0 / 3.0);
    x0 = -term1 + 2.0 * r13;
    x1 = -(r13 + term1);
    x2 = x1;
  } else {
    q = -q;
    const double th = acos(r / sqrt(q * q * q));
    const double r13 = 2.0 * sqrt(q);
    x0 = -term1 + r13 * cos(th / 3.0);
    x1 = -term1 + r13 * cos((th + 2.0 * M_PI) / 3.0);
    x2 = -term1 + r13 * cos((th + 4.0 * M_PI) / 3.0);
  }
  if (fragile) {
    const double a0 = fabs(x0), a1 = fabs(x1), a2 = fabs(x2);
    const double big = fmax(a0, fmax(a1, a2)), small = fmin(a0, fmin(a1, a2));
    // (NaN compares false: a NaN anywhere is fragile too.  1e-9 is many orders of magnitude above any libm discrepancy and still
    // as good as never true on data that is not exactly degenerate)
    *fragile = !(small > 1e-9 * big);
  }
  if (x0 * x1 * x2 == 0.0) return 1u;
  if (x0 < 0 && x1 < 0 && x2 < 0) return 8u;
  if (x0 > 0 && x1 > 0 && x2 > 0) return 2u;
  return 4u;
}

}  // namespace ftkx
