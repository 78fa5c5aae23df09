// Derived fields and the quantisation-factor reduction on the device.
//
// Reference (host loops, single-threaded unless OpenMP): include/ftk/ndarray/grad.hh
//   gradient2D 10-31, jacobian2D 54-86, gradient3D 130-149, jacobian3D 175-212
// and ndarray<T>::resolution() include/ftk/ndarray.hh:770-778, called per snapshot from
// critical_point_tracker::update_vector_field_scaling_factor (include/ftk/filters/critical_point_tracker.hh:850-864).
// The kernels reproduce those loops bit for bit, quirks included: each output element is produced by the
// same one or two FP64 operations in the same order; the file is compiled with -ffp-contract=off.
// All of them are pure streaming kernels: one lane per vertex, x fastest, so loads and stores coalesce.
#include <hip/hip_runtime.h>
#include <float.h>

#include "sweep_params.hpp"

namespace ftkx {

__device__ inline int clamp_idx(int i, int hi) { return i < 0 ? 0 : (i > hi ? hi : i); }

__global__ __launch_bounds__(256) void gradient2d_kernel(const double *__restrict__ S, int DW, int DH, double *__restrict__ V)
{
  const size_t n = (size_t)DW * DH;
  for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < n; idx += (size_t)gridDim.x * blockDim.x) {
    const int i = (int)(idx % DW), j = (int)(idx / DW);
    const double xp = S[(size_t)clamp_idx(i + 1, DW - 1) + (size_t)DW * j], xm = S[(size_t)clamp_idx(i - 1, DW - 1) + (size_t)DW * j];
    const double yp = S[(size_t)i + (size_t)DW * clamp_idx(j + 1, DH - 1)], ym = S[(size_t)i + (size_t)DW * clamp_idx(j - 1, DH - 1)];
    double2 g;
    g.x = (xp - xm) * (double)(DW - 1);    // no 0.5, scaled by (D-1): grad.hh:26-27
    g.y = (yp - ym) * (double)(DH - 1);
    reinterpret_cast<double2 *>(V)[idx] = g;
  }
}

__global__ __launch_bounds__(256) void jacobian2d_kernel(const double *__restrict__ V, int DW, int DH, int symmetric, double *__restrict__ J)
{
  const size_t n = (size_t)DW * DH;
  auto f = [&](int c, int i, int j) { return V[(size_t)c + 2 * ((size_t)clamp_idx(i, DW - 1) + (size_t)DW * (size_t)clamp_idx(j, DH - 1))]; };
  for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < n; idx += (size_t)gridDim.x * blockDim.x) {
    const int i = (int)(idx % DW), j = (int)(idx / DW);
    // precedence as written in the reference: a - b * (D-1)   (grad.hh:70-73)
    const double H00 = f(0, i + 1, j) - f(0, i - 1, j) * (double)(DW - 1),
                 H01 = f(0, i, j + 1) - f(0, i, j - 1) * (double)(DH - 1),
                 H10 = f(1, i + 1, j) - f(1, i - 1, j) * (double)(DW - 1),
                 H11 = f(1, i, j + 1) - f(1, i, j - 1) * (double)(DH - 1);
    double *o = J + 4 * idx;
    const double off = symmetric ? (H01 + H10) * 0.5 : 0.0;
    o[0] = H00; o[3] = H11;
    // non-symmetric instantiation: grad(0,1) / grad(1,0) are two-index accessors, i.e. the fixed flat elements 2 and 1 of
    // vertex 0, overwritten by every iteration; the value that survives is the one of the last vertex (grad.hh:79-82).
    // Vertex 0's own lane therefore leaves those two elements to the last lane.
    if (symmetric || idx != 0) { o[1] = off; o[2] = off; }
    if (!symmetric && idx == n - 1) { J[2] = H01; J[1] = H10; }
  }
}

__global__ __launch_bounds__(256) void gradient3d_kernel(const double *__restrict__ S, int DW, int DH, int DD, double *__restrict__ V)
{
  const size_t n = (size_t)DW * DH * DD;
  for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < n; idx += (size_t)gridDim.x * blockDim.x) {
    const int i = (int)(idx % DW), j = (int)((idx / DW) % DH), k = (int)(idx / ((size_t)DW * DH));
    double gx = 0.0, gy = 0.0, gz = 0.0;
    if (i >= 1 && i < DW - 1 && j >= 1 && j < DH - 1 && k >= 1 && k < DD - 1) {   // interior only, borders stay 0 (grad.hh:138-146)
      const size_t sy = (size_t)DW, sz = (size_t)DW * DH;
      gx = 0.5 * (S[idx + 1] - S[idx - 1]);
      gy = 0.5 * (S[idx + sy] - S[idx - sy]);
      gz = 0.5 * (S[idx + sz] - S[idx - sz]);
    }
    double *o = V + 3 * idx;
    o[0] = gx; o[1] = gy; o[2] = gz;
  }
}

__global__ __launch_bounds__(256) void jacobian3d_kernel(const double *__restrict__ V, int DW, int DH, int DD, double *__restrict__ J)
{
  const size_t n = (size_t)DW * DH * DD;
  for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < n; idx += (size_t)gridDim.x * blockDim.x) {
    const int i = (int)(idx % DW), j = (int)((idx / DW) % DH), k = (int)(idx / ((size_t)DW * DH));
    double o[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    if (i >= 2 && i < DW - 2 && j >= 2 && j < DH - 2 && k >= 2 && k < DD - 2) {   // b = 2 (grad.hh:184-186)
      const size_t sy = (size_t)DW, sz = (size_t)DW * DH;
      for (int a = 0; a < 3; a ++) {
        o[a + 0] = 0.5 * (V[3 * (idx + 1) + a] - V[3 * (idx - 1) + a]);      // J(a, 0)
        o[a + 3] = 0.5 * (V[3 * (idx + sy) + a] - V[3 * (idx - sy) + a]);    // J(a, 1)
        o[a + 6] = 0.5 * (V[3 * (idx + sz) + a] - V[3 * (idx - sz) + a]);    // J(a, 2)
      }
    }
    double *dst = J + 9 * idx;
    for (int q = 0; q < 9; q ++) dst[q] = o[q];
  }
}

// min over non-zero |v| (NaN and Inf never win, as with std::min in ndarray::resolution) and max over finite |v|.
// |v| >= 0 so the IEEE bit patterns order like unsigned integers: one atomicMin/atomicMax per wavefront on the raw bits.
__global__ __launch_bounds__(256) void resolution_kernel(const double *__restrict__ p, size_t n, u64 *out /* [0] = min bits, [1] = max bits */)
{
  u64 mn = 0x7fefffffffffffffull /* DBL_MAX */, mx = 0ull;
  const size_t n2 = n / 2;
  const double2 *p2 = reinterpret_cast<const double2 *>(p);
  auto take = [&](double v) {
    const double a = fabs(v);
    const u64 bits = (u64)__double_as_longlong(a);
    if (a != 0.0 && bits < 0x7ff0000000000000ull) { mn = bits < mn ? bits : mn; mx = bits > mx ? bits : mx; }
  };
  for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < n2; idx += (size_t)gridDim.x * blockDim.x) {
    const double2 v = p2[idx];
    take(v.x); take(v.y);
  }
  if ((n & 1) && blockIdx.x == 0 && threadIdx.x == 0) take(p[n - 1]);
  for (int o = 32; o > 0; o >>= 1) {
    const u64 omn = __shfl_down(mn, o), omx = __shfl_down(mx, o);
    mn = omn < mn ? omn : mn; mx = omx > mx ? omx : mx;
  }
  // 64 {min, max} slots folded by the host: thousands of wavefronts on one address serialise at the memory side (0.19 ms per 34 MB slice)
  const unsigned slot = (blockIdx.x * 5u + (threadIdx.x >> 6)) & 63u;
  if ((threadIdx.x & 63) == 0) { atomicMin(&out[2 * slot], mn); atomicMax(&out[2 * slot + 1], mx); }
}

static inline unsigned stream_grid(size_t n)
{
  const size_t want = (n + 255) / 256;
  return (unsigned)(want < 2048 ? (want ? want : 1) : 2048);   // 256 CUs x 8 workgroups, grid-stride the rest
}

void launch_gradient2d(const double *S, int DW, int DH, double *V, hipStream_t st)
{ hipLaunchKernelGGL(gradient2d_kernel, dim3(stream_grid((size_t)DW * DH)), dim3(256), 0, st, S, DW, DH, V); }
void launch_jacobian2d(const double *V, int DW, int DH, int symmetric, double *J, hipStream_t st)
{ hipLaunchKernelGGL(jacobian2d_kernel, dim3(stream_grid((size_t)DW * DH)), dim3(256), 0, st, V, DW, DH, symmetric, J); }
void launch_gradient3d(const double *S, int DW, int DH, int DD, double *V, hipStream_t st)
{ hipLaunchKernelGGL(gradient3d_kernel, dim3(stream_grid((size_t)DW * DH * DD)), dim3(256), 0, st, S, DW, DH, DD, V); }
void launch_jacobian3d(const double *V, int DW, int DH, int DD, double *J, hipStream_t st)
{ hipLaunchKernelGGL(jacobian3d_kernel, dim3(stream_grid((size_t)DW * DH * DD)), dim3(256), 0, st, V, DW, DH, DD, J); }
void launch_resolution(const double *p, size_t n, u64 *out2, hipStream_t st)
{ hipLaunchKernelGGL(resolution_kernel, dim3(stream_grid(n / 2 + 1)), dim3(256), 0, st, p, n, out2); }

}  // namespace ftkx
