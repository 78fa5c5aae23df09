// Does a kernel launched with hipExtAnyOrderLaunch start next to the kernel queued before it on the SAME stream (gfx950)?  And: two streams.
//   hipcc --offload-arch=gfx950 -O2 -o tools/probe/anyorder tools/probe/anyorder.hip && tools/probe/anyorder
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <cstdio>
#include <vector>

__global__ void spin_kernel(unsigned long long *stamps, unsigned long long ticks)
{
  const unsigned long long t0 = wall_clock64();
  if (threadIdx.x == 0) { atomicMin(&stamps[0], t0); }
  while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(8);
  if (threadIdx.x == 0) atomicMax(&stamps[1], wall_clock64());
}

__global__ __launch_bounds__(256) void big_kernel(unsigned long long *stamps, const double *in, double *out, size_t n)
{
  const unsigned long long t0 = wall_clock64();
  if (threadIdx.x == 0) atomicMin(&stamps[2], t0);
  double acc = 0;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) acc += in[i];
  if (acc == 1.2345) out[0] = acc;
  if (threadIdx.x == 0) atomicMax(&stamps[3], wall_clock64());
}

int main()
{
  unsigned long long *d_st, h[4];
  hipMalloc(&d_st, 32);
  const size_t n = (size_t)1 << 28;      // 2 GiB
  double *in, *out;
  hipMalloc(&in, n * 8); hipMalloc(&out, 8);
  hipMemset(in, 0, n * 8);
  hipStream_t s, s2;
  hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
  int lo, hi; hipDeviceGetStreamPriorityRange(&lo, &hi);
  hipStreamCreateWithPriority(&s2, hipStreamNonBlocking, hi);
  auto reset = [&] { unsigned long long z[4] = {~0ull, 0, ~0ull, 0}; hipMemcpy(d_st, z, 32, hipMemcpyHostToDevice); };
  auto show = [&](const char *what) {
    hipDeviceSynchronize();
    hipMemcpy(h, d_st, 32, hipMemcpyDeviceToHost);
    printf("%-60s spin [0, %6.1f] us   big [%6.1f, %6.1f] us\n", what, (h[1] - h[0]) / 100.0, ((long long)h[2] - (long long)h[0]) / 100.0, ((long long)h[3] - (long long)h[0]) / 100.0);
  };
  const unsigned long long ticks = 30000;      // 300 us at 100 MHz
  for (int rep = 0; rep < 2; rep ++) {
    reset();
    hipLaunchKernelGGL(spin_kernel, dim3(8), dim3(64), 0, s, d_st, ticks);
    hipLaunchKernelGGL(big_kernel, dim3(4096), dim3(256), 0, s, d_st, in, out, n);
    show("same stream, ordinary launches");
    reset();
    hipLaunchKernelGGL(spin_kernel, dim3(8), dim3(64), 0, s, d_st, ticks);
    hipExtLaunchKernelGGL(big_kernel, dim3(4096), dim3(256), 0, s, nullptr, nullptr, hipExtAnyOrderLaunch, d_st, in, out, n);
    show("same stream, big kernel with hipExtAnyOrderLaunch");
    reset();
    hipLaunchKernelGGL(spin_kernel, dim3(8), dim3(64), 0, s2, d_st, ticks);
    hipLaunchKernelGGL(big_kernel, dim3(4096), dim3(256), 0, s, d_st, in, out, n);
    show("two streams: spin queued first on its own stream");
    reset();
    hipLaunchKernelGGL(big_kernel, dim3(4096), dim3(256), 0, s, d_st, in, out, n);
    hipLaunchKernelGGL(spin_kernel, dim3(8), dim3(64), 0, s2, d_st, ticks);
    show("two streams: big queued first, spin (8 x 64 threads) behind it");
    reset();
    hipLaunchKernelGGL(big_kernel, dim3(4096), dim3(256), 0, s, d_st, in, out, n);
    hipLaunchKernelGGL(spin_kernel, dim3(256), dim3(256), 0, s2, d_st, ticks);
    show("two streams: big queued first, spin (256 x 256 threads) behind");
  }
  return 0;
}
