// TEST INFRASTRUCTURE.  Compiles the product's per-simplex arithmetic (ftk_amd/csrc/cp_device.hpp, fan_tables.hpp) for the
// HOST with g++ so that tests/test_host_numerics.py can compare it with the oracle on millions of random and degenerate
// inputs without a GPU.  Nothing in the product loads this library; the same headers run on the device in the HIP kernels.
#include "../../ftk_amd/csrc/cp_device.hpp"
#include "../../ftk_amd/csrc/fan_tables.hpp"
#include "../../ftk_amd/csrc/split_policy.hpp"

using namespace ftkx;

extern "C" {

int hc_origin_in_simplex2(const long long *X, const int *ids) { return origin_in_simplex2((const u64 (*)[2])X, ids); }
int hc_origin_in_simplex3(const long long *X, const int *ids) { return origin_in_simplex3((const u64 (*)[3])X, ids); }
int hc_sos_origin_in_simplex2(const long long *X, const int *ids) { return sos_origin_in_simplex<2>((const u64 (*)[2])X, ids); }
int hc_sos_origin_in_simplex3(const long long *X, const int *ids) { return sos_origin_in_simplex<3>((const u64 (*)[3])X, ids); }
int hc_orientation2(const long long *X, const int *ids) { return orientation2((const u64 (*)[2])X, ids); }
int hc_solve2(const double *V, double *mu) { return solve_barycentric2((const double (*)[2])V, mu); }
int hc_solve3(const double *V, double *mu) { return solve_barycentric3((const double (*)[3])V, mu); }
void hc_clamp3(double *x) { clamp_barycentric<3>(x); }
void hc_clamp4(double *x) { clamp_barycentric<4>(x); }
unsigned hc_classify2(const double *J, int symmetric) { return classify2(J[0], J[1], J[2], J[3], symmetric != 0); }
unsigned hc_classify3(const double *J, int symmetric) { return classify3((const double (*)[3])J, symmetric != 0); }
long long hc_quantize(double v, double factor) { return quantize(v, factor); }

// batch versions (python loops are slow)
void hc_batch_in_simplex2(int n, const long long *X, const int *ids, int *out_fast, int *out_sos)
{ for (int i = 0; i < n; i ++) { out_fast[i] = hc_origin_in_simplex2(X + 6 * i, ids + 3 * i); out_sos[i] = hc_sos_origin_in_simplex2(X + 6 * i, ids + 3 * i); } }
void hc_batch_in_simplex3(int n, const long long *X, const int *ids, int *out_fast, int *out_sos)
{ for (int i = 0; i < n; i ++) { out_fast[i] = hc_origin_in_simplex3(X + 12 * i, ids + 4 * i); out_sos[i] = hc_sos_origin_in_simplex3(X + 12 * i, ids + 4 * i); } }

// the 32-bit-operand forms of the fast path (every component must fit in int32)
void hc_batch_in_simplex2_s32(int n, const long long *X, const int *ids, int *out)
{ for (int i = 0; i < n; i ++) out[i] = origin_in_simplex2_s32((const u64 (*)[2])(X + 6 * i), ids + 3 * i); }
void hc_batch_in_simplex3_s32(int n, const long long *X, const int *ids, int *out)
{ for (int i = 0; i < n; i ++) out[i] = origin_in_simplex3_s32((const u64 (*)[3])(X + 12 * i), ids + 4 * i); }

// the id-free forms the kernels use: 1 / 0, or -1 = degenerate value (then the literal cascade decides)
void hc_batch_in_simplex_try(int nd, int n, const long long *X, const int *ids, int narrow, int *out)
{
  for (int i = 0; i < n; i ++) {
    int r = nd == 2 ? origin_in_simplex2_try((const u64 (*)[2])(X + 6 * i), narrow != 0) : origin_in_simplex3_try((const u64 (*)[3])(X + 12 * i), narrow != 0);
    if (r < 0) r = nd == 2 ? (int)sos_origin_in_simplex<2>((const u64 (*)[2])(X + 6 * i), ids + 3 * i) : (int)sos_origin_in_simplex<3>((const u64 (*)[3])(X + 12 * i), ids + 4 * i);
    out[i] = r;
  }
}

// the cascade only for the degenerate ones of a simplex's determinants (what the kernels call)
void hc_batch_in_simplex_resolved(int nd, int n, const long long *X, const int *ids, int *out)
{
  for (int i = 0; i < n; i ++)
    out[i] = nd == 2 ? (int)sos_origin_in_simplex_resolved<2>((const u64 (*)[2])(X + 6 * i), ids + 3 * i) : (int)sos_origin_in_simplex_resolved<3>((const u64 (*)[3])(X + 12 * i), ids + 4 * i);
}

int hc_fan(int n, int *verts /* [ntypes][n][n] */, int *ordinal, int *ord_types, int *int_types)
{
  if (n == 3) {
    for (int t = 0; t < 12; t ++) { ordinal[t] = k_fan3.ordinal[t]; for (int i = 0; i < 3; i ++) for (int a = 0; a < 3; a ++) verts[(t * 3 + i) * 3 + a] = (k_fan3.vert[t][i] >> a) & 1; }
    for (int i = 0; i < 2; i ++) ord_types[i] = k_fan3.ord_types[i];
    for (int i = 0; i < 10; i ++) int_types[i] = k_fan3.int_types[i];
    return 12;
  }
  for (int t = 0; t < 60; t ++) { ordinal[t] = k_fan4.ordinal[t]; for (int i = 0; i < 4; i ++) for (int a = 0; a < 4; a ++) verts[(t * 4 + i) * 4 + a] = (k_fan4.vert[t][i] >> a) & 1; }
  for (int i = 0; i < 6; i ++) ord_types[i] = k_fan4.ord_types[i];
  for (int i = 0; i < 54; i ++) int_types[i] = k_fan4.int_types[i];
  return 60;
}

// ---- the split pass's policy (ftk_amd/csrc/split_policy.hpp), driven without a device: tests/test_split_policy.py ----
void *hc_split_new() { return new ftkxh::split_cal(); }
void hc_split_delete(void *k) { delete (ftkxh::split_cal *)k; }
// -> split | cal_kind << 1 | forced << 3
int hc_split_decide(void *k, long mode, int pipelined, int dist, int profiling_ok, int sparse_now, unsigned long long ntodo, unsigned long long mask_bytes, unsigned long long signature)
{
  ftkxh::split_inputs in{mode, pipelined != 0, dist != 0, profiling_ok != 0, sparse_now != 0, ntodo, mask_bytes, signature};
  const ftkxh::split_verdict v = ftkxh::split_decide(*(ftkxh::split_cal *)k, in);
  return (v.split ? 1 : 0) | (v.cal_kind << 1) | (v.forced << 3);
}
void hc_split_sample(void *k, int cal_kind, double now, double last_s, int last_kind, int chained) { ftkxh::split_sample(*(ftkxh::split_cal *)k, cal_kind, now, last_s, last_kind, chained != 0); }
int hc_split_state(void *k, int forced, double *median_order, double *median_split, int *phase, unsigned *countdown)
{
  const ftkxh::split_cal &K = *(const ftkxh::split_cal *)k;
  *median_order = K.median_order; *median_split = K.median_split; *phase = K.phase; *countdown = K.countdown;
  return ftkxh::split_state(K, forced);
}
}
