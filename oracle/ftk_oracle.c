/* TEST INFRASTRUCTURE -- NOT PART OF THE PRODUCT.  See ftk_oracle.h.
 *
 * Plain-C restatement of the reference CPU sweep of hguo/ftk:
 *   critical_point_tracker_2d_regular::update_timestep / check_simplex
 *   critical_point_tracker_3d_regular::update_timestep / check_simplex
 * Build: oracle/Makefile (gcc -O2 -ffp-contract=off -fwrapv).  -fwrapv makes signed overflow wrap, which is
 * what the reference's int64 determinants do in practice; all determinant arithmetic here is
 * done on uint64_t anyway so the result does not depend on that flag.
 *
 * Citations are relative to /root/reference/include/ftk/.
 */
#define _GNU_SOURCE
#include "ftk_oracle.h"

#include <float.h>
#include <limits.h>
#include <math.h>
#include <pthread.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

typedef uint64_t u64;
typedef int64_t i64;

/* ------------------------------------------------------------------------------------------------
 * The simplex fan.  mesh/simplicial_regular_mesh.hh:620-715 builds it by subdividing the unit n-cube
 * into n! Kuhn simplices, taking all faces, translating each face so that its smallest vertex is the
 * origin ("reduce") and sorting.  Equivalent closed form: a reduced k-simplex is a chain
 * 0 = m_0 < m_1 < ... < m_k of nested axis bitmasks; types are numbered in lexicographic order of the
 * vertex lists, vertices compared as 0/1 vectors with axis 0 (x) most significant.
 * ---------------------------------------------------------------------------------------------- */
#define MAXN 4
#define MAXT 64

typedef struct {
  int n;                         /* mesh dimension */
  int ntypes[MAXN + 1];          /* per simplex dimension k */
  int verts[MAXN + 1][MAXT][MAXN + 1][MAXN]; /* [k][type][vertex][axis] */
  int ordinal[MAXN + 1][MAXT];
  int n_ord[MAXN + 1], n_int[MAXN + 1];
  int ord_types[MAXN + 1][MAXT], int_types[MAXN + 1][MAXT];
} fan_t;

static fan_t g_fan[MAXN + 1];
static pthread_once_t g_fan_once = PTHREAD_ONCE_INIT;

/* a vertex is held as an integer "key" whose bit (n-1-axis) is the coordinate: integer order == lexicographic order
 * of the 0/1 vector with x most significant */
static void enum_chains(fan_t *f, int k, int depth, int *chain)
{
  const int n = f->n;
  if (depth == k + 1) {
    const int t = f->ntypes[k] ++;
    for (int i = 0; i <= k; i ++)
      for (int a = 0; a < n; a ++)
        f->verts[k][t][i][a] = (chain[i] >> (n - 1 - a)) & 1;  /* chain[] holds keys */
    return;
  }
  /* next vertex: a strict superset (as axis set) of chain[depth-1]; iterate keys in increasing order so that the
   * enumeration comes out already in lexicographic order of the vertex list */
  for (int key = 0; key < (1 << n); key ++) {
    const int prev = chain[depth - 1];
    if (key != prev && (key & prev) == prev) {
      chain[depth] = key;
      enum_chains(f, k, depth + 1, chain);
    }
  }
}

static void build_fan_n(int n)
{
  fan_t *f = &g_fan[n];
  memset(f, 0, sizeof(*f));
  f->n = n;
  for (int k = 0; k <= n; k ++) {
    int chain[MAXN + 1] = {0};
    enum_chains(f, k, 1, chain);
    /* derive_ordinal_and_interval_simplices, simplicial_regular_mesh.hh:799-831: ordinal iff no vertex has the time bit */
    for (int t = 0; t < f->ntypes[k]; t ++) {
      int time = 0;
      for (int i = 0; i <= k; i ++) time += f->verts[k][t][i][n - 1];
      f->ordinal[k][t] = (k == 0) ? 1 : (time == 0);
      if (f->ordinal[k][t]) f->ord_types[k][f->n_ord[k] ++] = t;
      else f->int_types[k][f->n_int[k] ++] = t;
    }
  }
}

static void build_fans(void) { build_fan_n(3); build_fan_n(4); }
static const fan_t *fan(int n) { pthread_once(&g_fan_once, build_fans); return &g_fan[n]; }

int ftko_unit_simplices(int n, int verts[60][4][4], int is_ordinal[60])
{
  const fan_t *f = fan(n);
  const int k = n - 1;
  for (int t = 0; t < f->ntypes[k]; t ++) {
    for (int i = 0; i <= k; i ++)
      for (int a = 0; a < 4; a ++) verts[t][i][a] = a < n ? f->verts[k][t][i][a] : 0;
    is_ordinal[t] = f->ordinal[k][t];
  }
  return f->ntypes[k];
}

static int find_type(const fan_t *f, int k, int (*v)[MAXN])
{
  for (int t = 0; t < f->ntypes[k]; t ++) {
    int same = 1;
    for (int i = 0; i <= k && same; i ++)
      for (int a = 0; a < f->n; a ++)
        if (f->verts[k][t][i][a] != v[i][a]) { same = 0; break; }
    if (same) return t;
  }
  return -1;
}

/* enumerate_unit_simplex_sides, simplicial_regular_mesh.hh:760-797: drop one vertex, reduce_unit_simplex (655-683), look the
 * reduced face up; result sorted by (type, offset) because the reference collects into a std::set of tuples. */
static int sides_of_cell(const fan_t *f, int k, int type, int face_type[MAXN + 1], int face_offset[MAXN + 1][MAXN])
{
  const int n = f->n;
  int cnt = 0;
  for (int drop = 0; drop <= k; drop ++) {
    int v[MAXN + 1][MAXN], m = 0;
    for (int i = 0; i <= k; i ++) if (i != drop) { memcpy(v[m], f->verts[k][type][i], sizeof(int) * MAXN); m ++; }
    int off[MAXN] = {0};
    for (int a = 0; a < n; a ++) {
      int all_one = 1;
      for (int i = 0; i < m; i ++) if (v[i][a] == 0) all_one = 0;
      if (all_one) { off[a] = 1; for (int i = 0; i < m; i ++) v[i][a] = 0; }
    }
    /* chain vertices stay sorted after reduction (nested sets) */
    const int ft = find_type(f, k - 1, v);
    face_type[cnt] = ft;
    memcpy(face_offset[cnt], off, sizeof(off));
    cnt ++;
  }
  /* sort by (type, offset lexicographic) */
  for (int i = 0; i < cnt; i ++)
    for (int j = i + 1; j < cnt; j ++) {
      int less = 0;
      if (face_type[j] != face_type[i]) less = face_type[j] < face_type[i];
      else less = memcmp(face_offset[j], face_offset[i], sizeof(int) * n) < 0; /* 0/1 ints: memcmp order == lexicographic on little endian */
      if (less) {
        int t = face_type[i]; face_type[i] = face_type[j]; face_type[j] = t;
        int o[MAXN]; memcpy(o, face_offset[i], sizeof(o)); memcpy(face_offset[i], face_offset[j], sizeof(o)); memcpy(face_offset[j], o, sizeof(o));
      }
    }
  return cnt;
}

int ftko_sides(int n, int cell_type, int face_type[5], int face_offset[5][4])
{
  const fan_t *f = fan(n);
  int ft[MAXN + 1], fo[MAXN + 1][MAXN];
  const int cnt = sides_of_cell(f, n, cell_type, ft, fo);
  for (int i = 0; i < cnt; i ++) {
    face_type[i] = ft[i];
    for (int a = 0; a < 4; a ++) face_offset[i][a] = a < n ? fo[i][a] : 0;
  }
  return cnt;
}

/* enumerate_unit_simplex_side_of, simplicial_regular_mesh.hh:717-758: all (k+1)-cells (type, corner in {-1,0,1}^n) whose
 * vertex set contains the face; sorted by (type, corner). */
int ftko_side_of(int n, int type, int cell_type[2], int cell_offset[2][4])
{
  const fan_t *f = fan(n);
  const int k = n - 1;
  int cnt = 0;
  for (int ct = 0; ct < f->ntypes[k + 1]; ct ++) {
    int ft[MAXN + 1], fo[MAXN + 1][MAXN];
    const int ns = sides_of_cell(f, k + 1, ct, ft, fo);
    for (int s = 0; s < ns; s ++)
      if (ft[s] == type && cnt < 2) {
        cell_type[cnt] = ct;
        for (int a = 0; a < 4; a ++) cell_offset[cnt][a] = a < n ? -fo[s][a] : 0;
        cnt ++;
      }
  }
  if (cnt == 2) {
    int swap = 0;
    if (cell_type[1] < cell_type[0]) swap = 1;
    else if (cell_type[1] == cell_type[0])
      for (int a = 0; a < n; a ++) {
        if (cell_offset[1][a] != cell_offset[0][a]) { swap = cell_offset[1][a] < cell_offset[0][a]; break; }
      }
    if (swap) {
      int t = cell_type[0]; cell_type[0] = cell_type[1]; cell_type[1] = t;
      for (int a = 0; a < 4; a ++) { int o = cell_offset[0][a]; cell_offset[0][a] = cell_offset[1][a]; cell_offset[1][a] = o; }
    }
  }
  return cnt;
}

/* ------------------------------------------------------------------------------------------------
 * Integer predicates.  numeric/det.hh, numeric/sign.hh, numeric/sign_det.hh, numeric/critical_point_test.hh.
 * All determinant arithmetic is modulo 2^64 (u64) and the sign is read from the two's-complement value: this is
 * exactly what the reference's int64_t code computes on x86-64/gcc when it overflows.
 * ---------------------------------------------------------------------------------------------- */
static inline int sgn64(u64 x) { const i64 s = (i64)x; return (0 < s) - (s < 0); }        /* sign.hh:10-14 */

static inline u64 det2u(u64 a00, u64 a01, u64 a10, u64 a11) { return a00 * a11 - a10 * a01; }   /* det.hh:9-14 */

static inline u64 det3u(const u64 m[3][3])                                                       /* det.hh:16-24 */
{
  return m[0][0] * (m[1][1] * m[2][2] - m[1][2] * m[2][1])
       - m[0][1] * (m[1][0] * m[2][2] - m[1][2] * m[2][0])
       + m[0][2] * (m[1][0] * m[2][1] - m[1][1] * m[2][0]);
}

static inline u64 det4u(const u64 m[4][4])                                                       /* det.hh:26-55 */
{
  const u64 d2233 = m[2][2] * m[3][3] - m[2][3] * m[3][2],
            d2133 = m[2][1] * m[3][3] - m[2][3] * m[3][1],
            d2132 = m[2][1] * m[3][2] - m[2][2] * m[3][1],
            d2033 = m[2][0] * m[3][3] - m[2][3] * m[3][0],
            d2032 = m[2][0] * m[3][2] - m[2][2] * m[3][0],
            d2031 = m[2][0] * m[3][1] - m[2][1] * m[3][0];
  return m[0][0] * (m[1][1] * d2233 - m[1][2] * d2133 + m[1][3] * d2132)
       - m[0][1] * (m[1][0] * d2233 - m[1][2] * d2033 + m[1][3] * d2032)
       + m[0][2] * (m[1][0] * d2133 - m[1][1] * d2033 + m[1][3] * d2031)
       - m[0][3] * (m[1][0] * d2132 - m[1][1] * d2032 + m[1][2] * d2031);
}

/* determinant of rows (given as pointers) restricted to the listed columns, with a trailing column of ones */
static u64 hdet2(const i64 *r0, const i64 *r1, int c)
{ return det2u((u64)r0[c], 1, (u64)r1[c], 1); }

static u64 hdet3(const i64 *r0, const i64 *r1, const i64 *r2, int c0, int c1)
{
  const u64 m[3][3] = {{(u64)r0[c0], (u64)r0[c1], 1}, {(u64)r1[c0], (u64)r1[c1], 1}, {(u64)r2[c0], (u64)r2[c1], 1}};
  return det3u(m);
}

/* sign_det.hh:44-90 */
static int robust_sign_det3(const i64 X[3][2])
{
  int s;
  if ((s = sgn64(hdet3(X[0], X[1], X[2], 0, 1)))) return s;
  if ((s = -sgn64(hdet2(X[1], X[2], 0)))) return s;
  if ((s = sgn64(hdet2(X[1], X[2], 1)))) return s;
  if ((s = sgn64(hdet2(X[0], X[2], 0)))) return s;
  return 1;
}

/* sign_det.hh:92-200 */
static int robust_sign_det4(const i64 X[4][3])
{
  int s;
  {
    const u64 m[4][4] = {
      {(u64)X[0][0], (u64)X[0][1], (u64)X[0][2], 1}, {(u64)X[1][0], (u64)X[1][1], (u64)X[1][2], 1},
      {(u64)X[2][0], (u64)X[2][1], (u64)X[2][2], 1}, {(u64)X[3][0], (u64)X[3][1], (u64)X[3][2], 1}};
    if ((s = sgn64(det4u(m)))) return s;                                  /* t = 0  */
  }
  if ((s =  sgn64(hdet3(X[1], X[2], X[3], 0, 1)))) return s;              /* t = 1  */
  if ((s = -sgn64(hdet3(X[1], X[2], X[3], 0, 2)))) return s;              /* t = 2  */
  if ((s =  sgn64(hdet3(X[1], X[2], X[3], 1, 2)))) return s;              /* t = 3  */
  if ((s = -sgn64(hdet3(X[0], X[2], X[3], 0, 1)))) return s;              /* t = 4  */
  if ((s =  sgn64(hdet2(X[2], X[3], 0)))) return s;                       /* t = 5  */
  if ((s = -sgn64(hdet2(X[2], X[3], 1)))) return s;                       /* t = 6  */
  if ((s =  sgn64(hdet3(X[0], X[2], X[3], 0, 2)))) return s;              /* t = 7  */
  if ((s =  sgn64(hdet2(X[2], X[3], 2)))) return s;                       /* t = 8  */
  if ((s = -sgn64(hdet3(X[0], X[2], X[3], 1, 2)))) return s;              /* t = 9  */
  if ((s =  sgn64(hdet3(X[0], X[1], X[3], 0, 1)))) return s;              /* t = 10 */
  if ((s = -sgn64(hdet2(X[1], X[3], 0)))) return s;                       /* t = 11 */
  if ((s =  sgn64(hdet2(X[1], X[3], 1)))) return s;                       /* t = 12 */
  if ((s =  sgn64(hdet2(X[0], X[3], 0)))) return s;                       /* t = 13 */
  return 1;
}

/* nswaps_bubble_sort, sign_det.hh:203-220 */
static int bubble(int n, int *arr, int *order)
{
  for (int i = 0; i < n; i ++) order[i] = i;
  int nswaps = 0;
  for (int i = 0; i < n - 1; i ++)
    for (int j = 0; j < n - i - 1; j ++)
      if (arr[j] > arr[j + 1]) {
        int t = arr[j]; arr[j] = arr[j + 1]; arr[j + 1] = t;
        t = order[j]; order[j] = order[j + 1]; order[j + 1] = t;
        nswaps ++;
      }
  return nswaps;
}

/* positive2, sign_det.hh:243-267 */
static int positive2(const i64 X1[3][2], const int idx1[3])
{
  int idx[3] = {idx1[0], idx1[1], idx1[2]}, ord[3];
  const int s = bubble(3, idx, ord);
  i64 X[3][2];
  for (int i = 0; i < 3; i ++) for (int j = 0; j < 2; j ++) X[i][j] = X1[ord[i]][j];
  int d = robust_sign_det3(X);
  if (s % 2 != 0) d = -d;
  return d;
}

/* positive3, sign_det.hh:269-289 */
static int positive3(const i64 X1[4][3], const int idx1[4])
{
  int idx[4] = {idx1[0], idx1[1], idx1[2], idx1[3]}, ord[4];
  const int s = bubble(4, idx, ord);
  i64 X[4][3];
  for (int i = 0; i < 4; i ++) for (int j = 0; j < 3; j ++) X[i][j] = X1[ord[i]][j];
  int d = robust_sign_det4(X);
  if (s % 2 != 0) d = -d;
  return d;
}

/* robust_critical_point_in_simplex2 -> robust_point_in_simplex2 with x = 0, ix = -1
 * critical_point_test.hh:20-26, sign_det.hh:360-389 */
static int robust_cp_in_simplex2(const i64 X[3][2], const int idx[3])
{
  const int s = positive2(X, idx);
  for (int i = 0; i < 3; i ++) {
    i64 Y[3][2]; int my[3];
    for (int j = 0; j < 3; j ++)
      if (i == j) { my[j] = -1; Y[j][0] = Y[j][1] = 0; }
      else { my[j] = idx[j]; Y[j][0] = X[j][0]; Y[j][1] = X[j][1]; }
    if (positive2(Y, my) != s) return 0;
  }
  return 1;
}

/* critical_point_test.hh:28-34, sign_det.hh:391-414 */
static int robust_cp_in_simplex3(const i64 X[4][3], const int idx[4])
{
  const int s = positive3(X, idx);
  for (int i = 0; i < 4; i ++) {
    i64 Y[4][3]; int my[4];
    for (int j = 0; j < 4; j ++)
      if (i == j) { my[j] = -1; Y[j][0] = Y[j][1] = Y[j][2] = 0; }
      else { my[j] = idx[j]; for (int k = 0; k < 3; k ++) Y[j][k] = X[j][k]; }
    if (positive3(Y, my) != s) return 0;
  }
  return 1;
}

/* (int64_t)(v * factor), filters/critical_point_tracker_2d_regular.hh:605-616, ..._3d_regular.hh:453-460.
 * factor (uint64_t) is converted to double; out-of-range casts give x86's "integer indefinite" 0x8000000000000000. */
static inline i64 quantize(double v, u64 factor)
{
  const double p = v * (double)factor;
  if (!(p > -9223372036854775808.0 && p < 9223372036854775808.0)) return INT64_MIN;
  return (i64)p;
}

/* ------------------------------------------------------------------------------------------------
 * Floating-point part (evaluated only for simplices that pass, except where the reference's control
 * flow makes the result observable).
 * ---------------------------------------------------------------------------------------------- */
/* inverse_lerp_s2v2, numeric/inverse_linear_interpolation_solver.hh:32-54; solve_linear2x2 linear_solver.hh:22-34 */
static int inverse_lerp_s2v2(const double V[3][2], double mu[3])
{
  const double A[2][2] = {{V[0][0] - V[2][0], V[1][0] - V[2][0]}, {V[0][1] - V[2][1], V[1][1] - V[2][1]}};
  const double b[2] = {-V[2][0], -V[2][1]};
  const double D  = A[0][0] * A[1][1] - A[1][0] * A[0][1],
               Dx = b[0] * A[1][1] - A[0][1] * b[1],
               Dy = A[0][0] * b[1] - b[0] * A[1][0];
  mu[0] = Dx / D;
  mu[1] = Dy / D;
  mu[2] = 1.0 - mu[0] - mu[1];
  const double eps = DBL_EPSILON;
  return mu[0] >= -eps && mu[0] <= 1.0 + eps && mu[1] >= -eps && mu[1] <= 1.0 + eps && mu[2] >= -eps && mu[2] <= 1.0 + eps;
}

/* inverse_lerp_s3v3, inverse_linear_interpolation_solver.hh:143-167; solve_linear3x3 linear_solver.hh:12-20;
 * matrix_inverse3x3 matrix_inverse.hh:25-45; matrix3x3_vector3_multiplication matrix_multiplication.hh:55-60 */
static int inverse_lerp_s3v3(const double V[4][3], double l[4])
{
  const double m[3][3] = {
    {V[0][0] - V[3][0], V[1][0] - V[3][0], V[2][0] - V[3][0]},
    {V[0][1] - V[3][1], V[1][1] - V[3][1], V[2][1] - V[3][1]},
    {V[0][2] - V[3][2], V[1][2] - V[3][2], V[2][2] - V[3][2]}};
  const double b[3] = {-V[3][0], -V[3][1], -V[3][2]};
  double inv[3][3];
  inv[0][0] =   m[1][1] * m[2][2] - m[1][2] * m[2][1];
  inv[0][1] = - m[0][1] * m[2][2] + m[0][2] * m[2][1];
  inv[0][2] =   m[0][1] * m[1][2] - m[0][2] * m[1][1];
  inv[1][0] = - m[1][0] * m[2][2] + m[1][2] * m[2][0];
  inv[1][1] =   m[0][0] * m[2][2] - m[0][2] * m[2][0];
  inv[1][2] = - m[0][0] * m[1][2] + m[0][2] * m[1][0];
  inv[2][0] =   m[1][0] * m[2][1] - m[1][1] * m[2][0];
  inv[2][1] = - m[0][0] * m[2][1] + m[0][1] * m[2][0];
  inv[2][2] =   m[0][0] * m[1][1] - m[0][1] * m[1][0];
  const double det = m[0][0] * inv[0][0] + m[0][1] * inv[1][0] + m[0][2] * inv[2][0];
  const double invdet = 1.0 / det;
  for (int i = 0; i < 3; i ++) for (int j = 0; j < 3; j ++) inv[i][j] = inv[i][j] * invdet;
  l[0] = inv[0][0] * b[0] + inv[0][1] * b[1] + inv[0][2] * b[2];
  l[1] = inv[1][0] * b[0] + inv[1][1] * b[1] + inv[1][2] * b[2];
  l[2] = inv[2][0] * b[0] + inv[2][1] * b[1] + inv[2][2] * b[2];
  l[3] = 1.0 - l[0] - l[1] - l[2];
  const double eps = DBL_EPSILON;
  return l[0] >= -eps && l[0] < 1.0 + eps && l[1] >= -eps && l[1] < 1.0 + eps &&
         l[2] >= -eps && l[2] < 1.0 + eps && l[3] >= -eps && l[3] < 1.0 + eps;
}

/* clamp_barycentric<n>, numeric/clamp.hh:15-37.  std::max(a,x) = (a<x)?x:a, std::min(a,b) = (b<a)?b:a: NaN -> 0 */
static void clamp_barycentric(int n, double *x)
{
  double sum = 0.0;
  for (int i = 0; i < n; i ++) {
    const double mx = (0.0 < x[i]) ? x[i] : 0.0;
    x[i] = (1.0 < mx) ? 1.0 : mx;
    sum += x[i];
  }
  for (int i = 0; i < n; i ++) x[i] /= sum;
  if (isnan(x[0]) || isinf(x[0]))
    for (int i = 0; i < n; i ++) x[i] = 1.0 / n;
}

/* critical_point_type_2d, numeric/critical_point_type.hh:40-72 */
static unsigned cp_type_2d(const double J[2][2], int symmetric)
{
  if (symmetric) {
    /* solve_eigenvalues_symmetric2x2, eigen_solver2.hh:20-41 */
    const double m00 = J[0][0], m10 = J[1][0], m11 = J[1][1];
    const double b = -(m00 + m11), c = m00 * m11 - m10 * m10;
    const double delta = fma(b, b, -4 * c);
    const double sqrt_delta = delta < 0 ? 0 : sqrt(delta);
    double e0 = 0.5 * (-b + sqrt_delta), e1 = 0.5 * (-b - sqrt_delta);
    if (fabs(e0) < fabs(e1)) { const double t = e0; e0 = e1; e1 = t; }
    if (e0 > 0 && e1 > 0) return 2;
    else if (e0 < 0 && e1 < 0) return 8;
    else if (e0 * e1 < 0) return 4;
    else return 1;
  } else {
    /* solve_eigenvalues2x2 (complex), eigen_solver2.hh:61-67; characteristic_polynomial_2x2
     * characteristic_polynomial.hh:12-17; solve_quadratic quadratic_solver.hh:14-25 */
    const double P2 = 1.0, P1 = -(J[0][0] + J[1][1]), P0 = J[0][0] * J[1][1] - J[1][0] * J[0][1];
    const double delta = P1 * P1 - 4 * P2 * P0;
    if (delta >= 0) {
      const double r0 = (-P1 + sqrt(delta)) / (2 * P2), r1 = (-P1 - sqrt(delta)) / (2 * P2);
      if (r0 * r1 < 0) return 4;
      else if (r0 > 0 && r1 > 0) return 2;
      else if (r0 < 0 && r1 < 0) return 8;
      else return 1;
    } else {
      /* complex_sqrt(delta) = std::pow(std::complex(delta, 0), 0.5) (numeric/sqrt.hh:9-15); libstdc++ evaluates it as
       * polar(exp(0.5*log|delta|), 0.5*arg) with arg = pi, whose real part is rho*cos(pi/2) = rho*6.1e-17, not 0.
       * (a NaN delta also lands here: both comparisons are false -> CENTER) */
      const double rho = exp(0.5 * log(fabs(delta)));
      const double re = (-P1 + rho * cos(0.5 * atan2(0.0, delta))) / (2 * P2);
      if (re < 0) return 16;
      else if (re > 0) return 32;
      else return 64;
    }
  }
}

/* critical_point_type_3d, critical_point_type.hh:76-93; solve_eigenvalues_symmetric3x3 eigen_solver3.hh:20-47;
 * characteristic_polynomial_3x3 characteristic_polynomial.hh:40-47 */
static unsigned cp_type_3d(const double A[3][3], int symmetric)
{
  if (!symmetric) return 0;
  const double b = -(A[0][0] + A[1][1] + A[2][2]);
  const double c = A[1][1] * A[2][2] + A[0][0] * A[2][2] + A[0][0] * A[1][1]
                 - A[0][1] * A[1][0] - A[1][2] * A[2][1] - A[0][2] * A[2][0];
  const double det3 = A[0][0] * (A[1][1] * A[2][2] - A[1][2] * A[2][1])
                    - A[0][1] * (A[1][0] * A[2][2] - A[1][2] * A[2][0])
                    + A[0][2] * (A[1][0] * A[2][1] - A[1][1] * A[2][0]);
  const double d = -det3;
  double q, r, disc, dum1, term1, r13, x[3];
  q = (3.0 * c - (b * b)) / 9.0;
  r = (-(27.0 * d) + b * (9.0 * c - 2.0 * (b * b))) / 54.0;
  disc = q * q * q + r * r;
  term1 = (b / 3.0);
  if (disc >= 0) {
    r13 = ((r < 0) ? -pow(-r, (1.0 / 3.0)) : pow(r, (1.0 / 3.0)));
    x[0] = -term1 + 2.0 * r13;
    x[1] = -(r13 + term1);
    x[2] = -(r13 + term1);
  } else {
    q = -q;
    dum1 = q * q * q;
    dum1 = acos(r / sqrt(dum1));
    r13 = 2.0 * sqrt(q);
    x[0] = -term1 + r13 * cos(dum1 / 3.0);
    x[1] = -term1 + r13 * cos((dum1 + 2.0 * M_PI) / 3.0);
    x[2] = -term1 + r13 * cos((dum1 + 4.0 * M_PI) / 3.0);
  }
  if (x[0] * x[1] * x[2] == 0.0) return 1;
  if (x[0] < 0 && x[1] < 0 && x[2] < 0) return 8;
  else if (x[0] > 0 && x[1] > 0 && x[2] > 0) return 2;
  else return 4;
}

/* ------------------------------------------------------------------------------------------------
 * check_simplex
 * ---------------------------------------------------------------------------------------------- */
typedef struct {
  const ftko_sweep_args *a;
  const fan_t *f;
  int n;                 /* nd + 1 */
  int ntypes_scope, ntypes_all;
  const int *scope_types;
  i64 lb[3], ub[3];      /* spatial vertex validity (inclusive) */
  u64 mesh_prod[4];      /* lattice prod_ of the mesh lattice, lattice.hh:156-167 */
  int dimprod[4];        /* simplicial_regular_mesh::dimprod_ (int!), simplicial_regular_mesh.hh:930-947 */
} sweep_ctx;

static inline size_t ext_index(const sweep_ctx *c, const int *vx)
{
  const ftko_sweep_args *a = c->a;
  size_t idx = (size_t)(vx[0] - a->ext_st[0]);
  size_t stride = (size_t)a->ext_sz[0];
  for (int d = 1; d < a->nd; d ++) { idx += (size_t)(vx[d] - a->ext_st[d]) * stride; stride *= (size_t)a->ext_sz[d]; }
  return idx;
}

/* e.to_integer(m), simplicial_regular_mesh.hh:496-502: int*int products, accumulated into uint64 */
static u64 element_tag(const sweep_ctx *c, const int *corner, int type, int mode)
{
  const int n = c->n;
  u64 ci = 0;
  for (int i = 0; i < n; i ++) {
    const i64 lb = i < n - 1 ? c->lb[i] : 0;
    if (mode == FTKO_TAG_REFERENCE) {
      const int prod = (int)((unsigned)(corner[i] - (int)lb) * (unsigned)c->dimprod[i]); /* wrapping int multiply */
      ci += (u64)(i64)prod;
    } else {
      u64 dp = 1;
      for (int j = 0; j < i; j ++) dp *= (u64)(c->ub[j] - c->lb[j] + 1);
      ci += (u64)(corner[i] - lb) * dp;
    }
  }
  return ci * (u64)c->ntypes_all + (u64)type;
}

/* simplex_coordinates (2d:494-527, 3d:342-378), one vertex -> X[0..3]:
 *   SIMPLE       the lattice integers
 *   BOUNDS       ((v - array_lb) / double(array_size - 1)) * (b1 - b0) + b0  per axis (array_domain == ext)
 *   RECTILINEAR  rectilinear_coords[axis][v]
 *   EXPLICIT     explicit_coords(c, x, y) for c = 0, 1 (2D: and 2 if the array has a third component, else 0);
 *                3D reads the SAME three-index expression -- i.e. the z = 0 plane -- and reports vertices[i][2], the z
 *                index, as the time (3d:371-376): reproduced as written */
static inline void phys_coords(const ftko_sweep_args *a, int nd, const int *v, double X[4])
{
  X[2] = 0.0;
  X[3] = (double)v[nd];
  if (a->coords_mode == 1) {
    for (int d = 0; d < nd; d ++)
      X[d] = ((double)(unsigned long long)(v[d] - a->ext_st[d]) / (double)(a->ext_sz[d] - 1)) * (a->bounds[2 * d + 1] - a->bounds[2 * d]) + a->bounds[2 * d];
  } else if (a->coords_mode == 2) {
    for (int d = 0; d < nd; d ++) X[d] = a->rect[d][v[d]];
  } else if (a->coords_mode == 3) {
    const size_t at = (size_t)a->expl_ncomp * ((size_t)v[0] + (size_t)a->expl_n0 * (size_t)v[1]);
    X[0] = a->expl[at]; X[1] = a->expl[at + 1];
    if (nd == 2) X[2] = a->expl_ncomp > 2 ? a->expl[at + 2] : 0.0;
    else { X[2] = a->expl[at + 2]; X[3] = (double)v[2]; }
  } else {
    for (int d = 0; d < nd; d ++) X[d] = (double)v[d];
  }
}

static int check_simplex(const sweep_ctx *c, u64 work_index, ftko_rec_t *rec)
{
  const ftko_sweep_args *a = c->a;
  const int nd = a->nd, n = c->n, nv = n; /* vertices per simplex = nd + 1 */
  /* from_work_index, simplicial_regular_mesh.hh:480-493; lattice::from_integer lattice.hh:209-223 */
  const int itype = (int)(work_index % (u64)c->ntypes_scope);
  u64 ii = work_index / (u64)c->ntypes_scope;
  const int type = c->scope_types[itype];
  int corner[4] = {0, 0, 0, 0};
  for (int d = 0; d < nd; d ++) { corner[d] = (int)(a->core_st[d] + (i64)(ii % (u64)a->core_sz[d])); ii /= (u64)a->core_sz[d]; }
  corner[nd] = a->current_timestep;

  /* valid()/vertices(), simplicial_regular_mesh.hh:356-386 */
  int vx[4][4];
  for (int i = 0; i < nv; i ++)
    for (int d = 0; d < n; d ++) {
      vx[i][d] = corner[d] + c->f->verts[n - 1][type][i][d];
      if (d < nd) { if (vx[i][d] < c->lb[d] || vx[i][d] > c->ub[d]) return 0; }
      else if (vx[i][d] < 0) return 0;
    }

  /* simplex_vectors, 2d:494-582 / 3d:342-422 */
  double v[4][3];
  size_t at[4]; int iv[4];
  for (int i = 0; i < nv; i ++) {
    iv[i] = vx[i][nd] == a->current_timestep ? 0 : 1;
    at[i] = ext_index(c, vx[i]);
    for (int j = 0; j < nd; j ++) v[i][j] = a->V[iv[i]][at[i] * nd + j];
  }

  /* simplex_indices, regular_tracker.hh:188-194 -> lattice::to_integer lattice.hh:196-207, truncated to int */
  int indices[4];
  for (int i = 0; i < nv; i ++) {
    u64 id = (u64)(i64)(int)(vx[i][0] - (int)c->lb[0]);
    for (int d = 1; d < n; d ++) {
      const int rel = d < nd ? (int)(vx[i][d] - (int)c->lb[d]) : vx[i][d];
      id += (u64)(i64)rel * c->mesh_prod[d];
    }
    indices[i] = (int)id;
  }

  double mu[4];
  i64 vf[4][3];
  if (nd == 2) {
    /* critical_point_tracker_2d_regular.hh:584-685 */
    for (int i = 0; i < 3; i ++) for (int j = 0; j < 2; j ++) {
      if (isnan(v[i][j]) || isinf(v[i][j])) return 0;
      vf[i][j] = quantize(v[i][j], a->factor);
    }
    i64 vf2[3][2]; double v2[3][2];
    for (int i = 0; i < 3; i ++) for (int j = 0; j < 2; j ++) { vf2[i][j] = vf[i][j]; v2[i][j] = v[i][j]; }
    if (!robust_cp_in_simplex2(vf2, indices)) return 0;
    const int succ2 = inverse_lerp_s2v2(v2, mu);
    if (!succ2) clamp_barycentric(3, mu);

    memset(rec, 0, sizeof(*rec));
    /* simplex_coordinates (REGULAR_COORDS_SIMPLE) + lerp_s2v4, linear_interpolation.hh:83-101 */
    double X[3][4];
    for (int i = 0; i < 3; i ++) phys_coords(a, 2, vx[i], X[i]);
    rec->cp.x[0] = X[0][0] * mu[0] + X[1][0] * mu[1] + X[2][0] * mu[2];
    rec->cp.x[1] = X[0][1] * mu[0] + X[1][1] * mu[1] + X[2][1] * mu[2];
    rec->cp.x[2] = X[0][2] * mu[0] + X[1][2] * mu[1] + X[2][2] * mu[2];
    rec->cp.t    = X[0][3] * mu[0] + X[1][3] * mu[1] + X[2][3] * mu[2];
    if (a->S[0]) {
      double val[3];
      for (int i = 0; i < 3; i ++) val[i] = a->S[iv[i]][at[i]];
      rec->cp.scalar[0] = val[0] * mu[0] + val[1] * mu[1] + val[2] * mu[2];            /* lerp_s2, :53-57 */
    }
    const int ordinal = c->f->ordinal[2][type];
    if (a->compute_degrees) {                                                           /* 2d:653-662 */
      if (ordinal) {
        int deg = positive2(vf2, indices);
        const int chi = type == 4 ? 1 : -1;
        deg *= chi;
        rec->cp.type = deg == 1 ? 1 : 2;
      } else rec->cp.type = 0;
    } else {
      double J[2][2] = {{0, 0}, {0, 0}};
      if (a->J[0]) {
        double Js[3][2][2];                                                             /* simplex_jacobians 2d:566-582 */
        for (int i = 0; i < 3; i ++) for (int j = 0; j < 2; j ++) for (int k = 0; k < 2; k ++)
          Js[i][j][k] = a->J[iv[i]][at[i] * 4 + (size_t)j * 2 + k];
        /* lerp_s2m2x2 :103-108 */
        J[0][0] = Js[0][0][0] * mu[0] + Js[1][0][0] * mu[1] + Js[2][0][0] * mu[2];
        J[0][1] = Js[0][0][1] * mu[0] + Js[1][0][1] * mu[1] + Js[2][0][1] * mu[2];
        J[1][0] = Js[0][1][0] * mu[0] + Js[1][1][0] * mu[1] + Js[2][1][0] * mu[2];
        J[1][1] = Js[0][1][1] * mu[0] + Js[1][1][1] * mu[1] + Js[2][1][1] * mu[2];
        const double s = 0.5 * (J[0][1] + J[1][0]);                                     /* make_symmetric2x2 symmetric_matrix.hh:10-15 */
        J[0][1] = J[1][0] = s;
      }
      rec->cp.type = cp_type_2d(J, a->jacobian_symmetric);
    }
    if (a->use_type_filter && !(a->type_filter & rec->cp.type)) return 0;               /* filter_critical_point_type */
  } else {
    /* critical_point_tracker_3d_regular.hh:425-514 */
    const int succ2 = inverse_lerp_s3v3((const double (*)[3])v, mu);
    if (a->robust) {
      for (int i = 0; i < 4; i ++) for (int j = 0; j < 3; j ++) {
        if (isnan(v[i][j]) || isinf(v[i][j])) return 0;
        vf[i][j] = quantize(v[i][j], a->factor);
      }
      if (!robust_cp_in_simplex3((const i64 (*)[3])vf, indices)) return 0;
    } else if (!succ2) return 0;
    clamp_barycentric(4, mu);

    memset(rec, 0, sizeof(*rec));
    double X[4][4];
    for (int i = 0; i < 4; i ++) phys_coords(a, 3, vx[i], X[i]);
    double x[4];
    for (int d = 0; d < 4; d ++) x[d] = X[0][d] * mu[0] + X[1][d] * mu[1] + X[2][d] * mu[2] + X[3][d] * mu[3]; /* lerp_s3v4 :129-139 */
    rec->cp.x[0] = x[0]; rec->cp.x[1] = x[1]; rec->cp.x[2] = x[2]; rec->cp.t = x[3];
    if (a->S[0]) {
      double val[4];
      for (int i = 0; i < 4; i ++) val[i] = a->S[iv[i]][at[i]];
      rec->cp.scalar[0] = val[0] * mu[0] + val[1] * mu[1] + val[2] * mu[2] + val[3] * mu[3];   /* lerp_s3 :110-115 */
    }
    double J[3][3];
    for (int j = 0; j < 3; j ++) for (int k = 0; k < 3; k ++) {                          /* lerp_s3m3x3 :141-151 */
      J[j][k] = 0.0;
      for (int i = 0; i < 4; i ++) {
        const double Jijk = a->J[0] ? a->J[iv[i]][at[i] * 9 + (size_t)j * 3 + k] : 0.0;
        J[j][k] += Jijk * mu[i];
      }
    }
    rec->cp.type = cp_type_3d(J, a->jacobian_symmetric);
  }

  rec->ordinal = c->f->ordinal[n - 1][type];
  rec->timestep = a->current_timestep;
  rec->etype = type;
  for (int d = 0; d < 4; d ++) rec->corner[d] = d < n ? corner[d] : 0;
  if (nd == 2) { rec->corner[3] = 0; }
  rec->cp.tag = a->tag_mode == FTKO_TAG_WORK_INDEX ? work_index : element_tag(c, corner, type, a->tag_mode);
  return 1;
}

/* ------------------------------------------------------------------------------------------------
 * sweep driver (element_for -> parallel_for, simplicial_regular_mesh.hh:1030-1045; object.hh:61-82)
 * ---------------------------------------------------------------------------------------------- */
typedef struct { ftko_rec_t *p; size_t n, cap; } recvec;
static void rv_push(recvec *v, const ftko_rec_t *r)
{
  if (v->n == v->cap) { v->cap = v->cap ? v->cap * 2 : 256; v->p = (ftko_rec_t *)realloc(v->p, v->cap * sizeof(ftko_rec_t)); }
  v->p[v->n ++] = *r;
}

typedef struct { const sweep_ctx *c; u64 begin, end; recvec out; } job_t;

static void *job_main(void *arg)
{
  job_t *j = (job_t *)arg;
  ftko_rec_t rec;
  for (u64 w = j->begin; w < j->end; w ++)
    if (check_simplex(j->c, w, &rec)) rv_push(&j->out, &rec);
  return NULL;
}

static int rec_cmp(const void *pa, const void *pb)
{
  const ftko_rec_t *a = (const ftko_rec_t *)pa, *b = (const ftko_rec_t *)pb;
  /* order by (t, z, y, x, type): corner[] is x,y,(z,)t -- compare from the slowest axis */
  const int order[4] = {3, 2, 1, 0};
  int ta = a->corner[3], tb = b->corner[3];
  (void)ta; (void)tb;
  for (int k = 0; k < 4; k ++) { const int d = order[k]; if (a->corner[d] != b->corner[d]) return a->corner[d] < b->corner[d] ? -1 : 1; }
  if (a->etype != b->etype) return a->etype < b->etype ? -1 : 1;
  return 0;
}

static void make_ctx(const ftko_sweep_args *a, sweep_ctx *c)
{
  memset(c, 0, sizeof(*c));
  c->a = a;
  c->n = a->nd + 1;
  c->f = fan(c->n);
  const int k = c->n - 1;
  c->ntypes_all = c->f->ntypes[k];
  if (a->scope == FTKO_SCOPE_ORDINAL) { c->ntypes_scope = c->f->n_ord[k]; c->scope_types = c->f->ord_types[k]; }
  else { c->ntypes_scope = c->f->n_int[k]; c->scope_types = c->f->int_types[k]; }
  for (int d = 0; d < a->nd; d ++) { c->lb[d] = a->domain_st[d]; c->ub[d] = a->domain_st[d] + a->domain_sz[d] - 1; }
  c->mesh_prod[0] = 1;
  c->dimprod[0] = 1;
  for (int d = 1; d < c->n; d ++) {
    c->mesh_prod[d] = c->mesh_prod[d - 1] * (u64)a->domain_sz[d - 1];
    c->dimprod[d] = (int)((u64)a->domain_sz[d - 1] * (u64)(i64)c->dimprod[d - 1]);
  }
}

unsigned long long ftko_num_work_items(const ftko_sweep_args *a)
{
  sweep_ctx c; make_ctx(a, &c);
  u64 n = (u64)c.ntypes_scope;
  for (int d = 0; d < a->nd; d ++) n *= (u64)a->core_sz[d];
  return n;
}

size_t ftko_sweep(const ftko_sweep_args *a, ftko_rec_t **out)
{
  sweep_ctx c; make_ctx(a, &c);
  const u64 ntasks = ftko_num_work_items(a);
  int nt = a->nthreads > 1 ? a->nthreads : 1;
  if ((u64)nt > ntasks) nt = ntasks ? (int)ntasks : 1;
  job_t *jobs = (job_t *)calloc((size_t)nt, sizeof(job_t));
  pthread_t *th = (pthread_t *)calloc((size_t)nt, sizeof(pthread_t));
  for (int i = 0; i < nt; i ++) {
    jobs[i].c = &c;
    jobs[i].begin = ntasks * (u64)i / (u64)nt;
    jobs[i].end = ntasks * (u64)(i + 1) / (u64)nt;
  }
  if (nt == 1) job_main(&jobs[0]);
  else {
    for (int i = 0; i < nt; i ++) pthread_create(&th[i], NULL, job_main, &jobs[i]);
    for (int i = 0; i < nt; i ++) pthread_join(th[i], NULL);
  }
  size_t total = 0;
  for (int i = 0; i < nt; i ++) total += jobs[i].out.n;
  ftko_rec_t *res = (ftko_rec_t *)malloc((total ? total : 1) * sizeof(ftko_rec_t));
  size_t off = 0;
  for (int i = 0; i < nt; i ++) {
    if (jobs[i].out.n) memcpy(res + off, jobs[i].out.p, jobs[i].out.n * sizeof(ftko_rec_t));
    off += jobs[i].out.n;
    free(jobs[i].out.p);
  }
  free(jobs); free(th);
  qsort(res, total, sizeof(ftko_rec_t), rec_cmp);
  *out = res;
  return total;
}

void ftko_free(void *p) { free(p); }

/* ------------------------------------------------------------------------------------------------
 * Derived fields, ndarray/grad.hh -- including the reference's quirks
 * ---------------------------------------------------------------------------------------------- */
static inline int clampi(int i, int lo, int hi) { return i < lo ? lo : (i > hi ? hi : i); }

void ftko_gradient2D(const double *S, int DW, int DH, double *V)                        /* grad.hh:10-31 */
{
#define F2(i, j) S[(size_t)clampi((i), 0, DW - 1) + (size_t)DW * (size_t)clampi((j), 0, DH - 1)]
  for (int j = 0; j < DH; j ++)
    for (int i = 0; i < DW; i ++) {
      const size_t o = 2 * ((size_t)i + (size_t)DW * (size_t)j);
      V[o + 0] = (F2(i + 1, j) - F2(i - 1, j)) * (DW - 1);
      V[o + 1] = (F2(i, j + 1) - F2(i, j - 1)) * (DH - 1);
    }
#undef F2
}

void ftko_jacobian2D(const double *V, int DW, int DH, int symmetric, double *J)         /* grad.hh:54-86 */
{
#define FV(c, i, j) V[(size_t)(c) + 2 * ((size_t)clampi((i), 0, DW - 1) + (size_t)DW * (size_t)clampi((j), 0, DH - 1))]
  memset(J, 0, sizeof(double) * 4 * (size_t)DW * (size_t)DH);
  for (int j = 0; j < DH; j ++)
    for (int i = 0; i < DW; i ++) {
      /* operator precedence as written in the reference: a - b * (D-1) */
      const double H00 = FV(0, i + 1, j) - FV(0, i - 1, j) * (DW - 1),
                   H01 = FV(0, i, j + 1) - FV(0, i, j - 1) * (DH - 1),
                   H10 = FV(1, i + 1, j) - FV(1, i - 1, j) * (DW - 1),
                   H11 = FV(1, i, j + 1) - FV(1, i, j - 1) * (DH - 1);
      const size_t o = 4 * ((size_t)i + (size_t)DW * (size_t)j);
      J[o + 0] = H00;           /* grad(0,0,i,j) */
      J[o + 3] = H11;           /* grad(1,1,i,j) */
      if (symmetric) J[o + 2] = J[o + 1] = (H01 + H10) * 0.5;
      else { J[2] = H01; J[1] = H10; }   /* grad(0,1) / grad(1,0): two-index accessors -> fixed flat offsets 2 and 1 (grad.hh:79-82) */
    }
#undef FV
}

void ftko_gradient3D(const double *S, int DW, int DH, int DD, double *V)                /* grad.hh:130-149 */
{
  memset(V, 0, sizeof(double) * 3 * (size_t)DW * DH * DD);
#define S3(i, j, k) S[(size_t)(i) + (size_t)DW * ((size_t)(j) + (size_t)DH * (size_t)(k))]
  for (int k = 1; k < DD - 1; k ++)
    for (int j = 1; j < DH - 1; j ++)
      for (int i = 1; i < DW - 1; i ++) {
        const size_t o = 3 * ((size_t)i + (size_t)DW * ((size_t)j + (size_t)DH * (size_t)k));
        V[o + 0] = 0.5 * (S3(i + 1, j, k) - S3(i - 1, j, k));
        V[o + 1] = 0.5 * (S3(i, j + 1, k) - S3(i, j - 1, k));
        V[o + 2] = 0.5 * (S3(i, j, k + 1) - S3(i, j, k - 1));
      }
#undef S3
}

void ftko_jacobian3D(const double *V, int DW, int DH, int DD, double *J)                /* grad.hh:175-212, b = 2 */
{
  memset(J, 0, sizeof(double) * 9 * (size_t)DW * DH * DD);
#define V3(c, i, j, k) V[(size_t)(c) + 3 * ((size_t)(i) + (size_t)DW * ((size_t)(j) + (size_t)DH * (size_t)(k)))]
  for (int k = 2; k < DD - 2; k ++)
    for (int j = 2; j < DH - 2; j ++)
      for (int i = 2; i < DW - 2; i ++) {
        const size_t o = 9 * ((size_t)i + (size_t)DW * ((size_t)j + (size_t)DH * (size_t)k));
        for (int a = 0; a < 3; a ++) {
          /* J(a, b, i, j, k) at flat offset a + 3*b */
          J[o + a + 0] = 0.5 * (V3(a, i + 1, j, k) - V3(a, i - 1, j, k));
          J[o + a + 3] = 0.5 * (V3(a, i, j + 1, k) - V3(a, i, j - 1, k));
          J[o + a + 6] = 0.5 * (V3(a, i, j, k + 1) - V3(a, i, j, k - 1));
        }
      }
#undef V3
}

double ftko_resolution(const double *p, size_t n)                                       /* ndarray.hh:770-778 */
{
  double r = DBL_MAX;
  for (size_t i = 0; i < n; i ++)
    if (p[i] != 0.0) { const double ab = fabs(p[i]); r = (ab < r) ? ab : r; }  /* std::min(r, |p|) = (|p| < r) ? |p| : r */
  return r;
}

unsigned long long ftko_scaling_factor(double resolution, int *nbits_out)               /* critical_point_tracker.hh:850-864 */
{
  int nbits = (int)ceil(log2(1.0 / resolution));
  const int mn = nbits < 21 ? nbits : 21;
  nbits = 8 > mn ? 8 : mn;
  if (nbits_out) *nbits_out = nbits;
  return (unsigned long long)(1 << nbits);
}

/* ------------------------------------------------------------------------------------------------
 * Synthetic inputs, ndarray/synthetic.hh
 * ---------------------------------------------------------------------------------------------- */
void ftko_synthetic_woven_2D(int DW, int DH, double t, double *S)                       /* synthetic.hh:11-14, 29-45 */
{
  const double scaling_factor = 15;
  for (int j = 0; j < DH; j ++)
    for (int i = 0; i < DW; i ++) {
      const double x = (((double)i / (DW - 1)) - 0.5) * scaling_factor,
                   y = (((double)j / (DH - 1)) - 0.5) * scaling_factor;
      S[(size_t)i + (size_t)DW * j] = cos(x * cos(t) - y * sin(t)) * sin(x * sin(t) + y * cos(t));
    }
}

void ftko_synthetic_merger_2D(int DW, int DH, double t, double *S)                      /* synthetic.hh:263-298 */
{
  for (int j = 0; j < DH; j ++)
    for (int i = 0; i < DW; i ++) {
      double x = (((double)i / (DW - 1)) - 0.5) * 4, y = (((double)j / (DH - 1)) - 0.5) * 4;
      const double xp = x * cos(t) - y * sin(t), yp = x * sin(t) + y * cos(t);
      x = xp; y = yp;
      const double cx0 = sin(t - M_PI_2), cx1 = sin(t + M_PI_2), cy0 = 1e-4, cy1 = 1e-4;
      const double f0 = exp(-((x - cx0) * (x - cx0) + (y - cy0) * (y - cy0)));
      const double f1 = exp(-((x - cx1) * (x - cx1) + (y - cy1) * (y - cy1)));
      S[(size_t)i + (size_t)DW * j] = (f0 < f1) ? f1 : f0;   /* std::max(f0, f1) */
    }
}

void ftko_synthetic_moving_extremum(int nd, const int *D, const double *x0, const double *dir, double t, double *S)
{                                                                                       /* synthetic.hh:332-354 */
  double xc[3] = {0, 0, 0};
  for (int j = 0; j < nd; j ++) xc[j] = x0[j] + dir[j] * t;
  const int D2 = nd > 2 ? D[2] : 1;
  for (int k = 0; k < D2; k ++)
    for (int j = 0; j < D[1]; j ++)
      for (int i = 0; i < D[0]; i ++) {
        const int xi[3] = {i, j, k};
        double d = 0;
        for (int a = 0; a < nd; a ++) d += pow(xi[a] - xc[a], 2.0);
        S[(size_t)i + (size_t)D[0] * ((size_t)j + (size_t)D[1] * (size_t)k)] = d;
      }
}

void ftko_synthetic_double_gyre(int DW, int DH, double time, double A, double omega, double eps, double *V)
{                                                                                       /* synthetic.hh:130-150, 193-217 */
  for (int j = 0; j < DH; j ++)
    for (int i = 0; i < DW; i ++) {
      const double x = ((double)i / (DW - 1)) * 2, y = ((double)j / (DH - 1));
      const double a = eps * sin(omega * time);
      const double b = 1 - 2 * eps * sin(omega * time);
      const double f = a * x * x + b * x;
      const double dfdx = 2 * a * x + b;
      const size_t o = 2 * ((size_t)i + (size_t)DW * j);
      V[o + 0] = -M_PI * A * sin(M_PI * f) * cos(M_PI * y);
      V[o + 1] =  M_PI * A * cos(M_PI * f) * sin(M_PI * y) * dfdx;
    }
}

/* ------------------------------------------------------------------------------------------------
 * Whole tracker loop (json_interface.hh:606-725; critical_point_tracker.hh:841-864;
 * critical_point_tracker_{2d,3d}_regular::push_*_field_snapshot / update_timestep)
 * ---------------------------------------------------------------------------------------------- */
typedef struct { double *S, *V, *J; } snapshot_t;

static double now_s(void) { struct timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return ts.tv_sec + 1e-9 * ts.tv_nsec; }

size_t ftko_track(const ftko_track_args *a, ftko_rec_t **out, unsigned long long *factors, double *sweep_seconds)
{
  const int nd = a->nd;
  const size_t N = (size_t)a->D[0] * a->D[1] * (nd > 2 ? a->D[2] : 1);
  const int scalar_in = a->nv == 1;
  snapshot_t q[2]; int nq = 0;
  double resolution = DBL_MAX;       /* vector_field_resolution, critical_point_tracker.hh:162 (sticky) */
  int current = a->t0;
  recvec all = {0, 0, 0};
  double tsweep = 0;

  ftko_sweep_args sa;
  memset(&sa, 0, sizeof(sa));
  sa.nd = nd;
  for (int d = 0; d < nd; d ++) {
    sa.domain_st[d] = scalar_in ? 2 : 1;                         /* json_interface.hh:640-653 */
    sa.domain_sz[d] = scalar_in ? a->D[d] - 3 : a->D[d] - 2;
    sa.core_st[d] = sa.domain_st[d]; sa.core_sz[d] = sa.domain_sz[d];
    sa.ext_st[d] = 0; sa.ext_sz[d] = a->D[d];
  }
  sa.jacobian_symmetric = scalar_in;
  sa.robust = a->robust; sa.use_type_filter = a->use_type_filter; sa.type_filter = a->type_filter;
  sa.compute_degrees = a->compute_degrees; sa.tag_mode = a->tag_mode; sa.nthreads = a->nthreads;
  sa.coords_mode = a->coords_mode; for (int i = 0; i < 6; i ++) sa.bounds[i] = a->bounds[i];
  for (int i = 0; i < 3; i ++) sa.rect[i] = a->rect[i];
  sa.expl = a->expl; sa.expl_ncomp = a->expl_ncomp; sa.expl_n0 = a->expl_n0;
  int degenerate = 0;
  for (int d = 0; d < nd; d ++) if (sa.domain_sz[d] <= 0) degenerate = 1;

  for (int k = 0; k < a->DT; k ++) {
    /* push_{scalar,vector}_field_snapshot */
    snapshot_t s = {NULL, NULL, NULL};
    s.V = (double *)malloc(sizeof(double) * N * nd);
    s.J = (double *)malloc(sizeof(double) * N * nd * nd);
    if (scalar_in) {
      s.S = (double *)malloc(sizeof(double) * N);
      memcpy(s.S, a->steps[k], sizeof(double) * N);
      if (nd == 2) { ftko_gradient2D(s.S, a->D[0], a->D[1], s.V); ftko_jacobian2D(s.V, a->D[0], a->D[1], 1, s.J); }
      else { ftko_gradient3D(s.S, a->D[0], a->D[1], a->D[2], s.V); ftko_jacobian3D(s.V, a->D[0], a->D[1], a->D[2], s.J); }
    } else {
      memcpy(s.V, a->steps[k], sizeof(double) * N * nd);
      if (nd == 2) ftko_jacobian2D(s.V, a->D[0], a->D[1], 0, s.J);
      else ftko_jacobian3D(s.V, a->D[0], a->D[1], a->D[2], s.J);
    }
    q[nq ++] = s;

    const int n_updates = (k != 0) + (k == a->DT - 1);   /* advance_timestep(); then the final update_timestep() */
    for (int u = 0; u < n_updates; u ++) {
      if (u == 1 && k == 0) break;
      /* update_timestep */
      for (int i = 0; i < nq; i ++) { const double r = ftko_resolution(q[i].V, N * nd); resolution = (r < resolution) ? r : resolution; }
      const u64 factor = ftko_scaling_factor(resolution, NULL);
      if (factors) factors[current - a->t0] = factor;
      sa.factor = factor;
      sa.current_timestep = current;
      sa.V[0] = q[0].V; sa.J[0] = q[0].J; sa.S[0] = q[0].S;
      sa.V[1] = nq > 1 ? q[1].V : NULL; sa.J[1] = nq > 1 ? q[1].J : NULL; sa.S[1] = nq > 1 ? q[1].S : NULL;
      const double t0 = now_s();
      for (int scope = FTKO_SCOPE_ORDINAL; scope <= (nq >= 2 ? FTKO_SCOPE_INTERVAL : FTKO_SCOPE_ORDINAL); scope ++) {
        if (degenerate) break;
        sa.scope = scope;
        ftko_rec_t *r = NULL;
        const size_t n = ftko_sweep(&sa, &r);
        for (size_t i = 0; i < n; i ++) rv_push(&all, &r[i]);
        free(r);
      }
      tsweep += now_s() - t0;
      if (k != 0 && u == 0) {
        /* advance_timestep: pop + current_timestep ++ */
        free(q[0].S); free(q[0].V); free(q[0].J);
        q[0] = q[1]; nq --;
        current ++;
      }
    }
  }
  for (int i = 0; i < nq; i ++) { free(q[i].S); free(q[i].V); free(q[i].J); }
  if (sweep_seconds) *sweep_seconds = tsweep;
  qsort(all.p, all.n, sizeof(ftko_rec_t), rec_cmp);
  *out = all.p ? all.p : (ftko_rec_t *)malloc(sizeof(ftko_rec_t));
  return all.n;
}

/* ------------------------------------------------------------------------------------------------
 * hooks for unit tests of the per-simplex arithmetic (tests/test_host_numerics.py)
 * ---------------------------------------------------------------------------------------------- */
int ftko_hook_cp_in_simplex2(const long long *X, const int *idx) { return robust_cp_in_simplex2((const i64 (*)[2])X, idx); }
int ftko_hook_cp_in_simplex3(const long long *X, const int *idx) { return robust_cp_in_simplex3((const i64 (*)[3])X, idx); }
int ftko_hook_positive2(const long long *X, const int *idx) { return positive2((const i64 (*)[2])X, idx); }
int ftko_hook_inverse_lerp2(const double *V, double *mu) { return inverse_lerp_s2v2((const double (*)[2])V, mu); }
int ftko_hook_inverse_lerp3(const double *V, double *mu) { return inverse_lerp_s3v3((const double (*)[3])V, mu); }
void ftko_hook_clamp(int n, double *x) { clamp_barycentric(n, x); }
unsigned ftko_hook_type2(const double *J, int symmetric) { return cp_type_2d((const double (*)[2])J, symmetric); }
unsigned ftko_hook_type3(const double *J, int symmetric) { return cp_type_3d((const double (*)[3])J, symmetric); }
long long ftko_hook_quantize(double v, unsigned long long factor) { return quantize(v, factor); }
