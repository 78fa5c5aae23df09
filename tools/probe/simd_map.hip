// Which SIMD does wave w of a 256-thread (4-wave) workgroup land on?  HW_ID (gfx9 encoding): wave 3:0, simd 5:4, pipe 7:6, cu 11:8, sh 12, se 15:13
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void probe(unsigned *out, int spin)
{
  unsigned id;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(id));
  unsigned xcc;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
  // stay resident for a while so that several workgroups share a CU
  long long t0 = clock64();
  while (clock64() - t0 < spin) { }
  if ((threadIdx.x & 63) == 0) { out[(blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64) * 2] = id; out[(blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64) * 2 + 1] = xcc; }
}
int main(int argc, char **argv)
{
  const int waves = argc > 1 ? atoi(argv[1]) : 4, nwg = 768;
  unsigned *d; hipMalloc(&d, nwg * waves * 8);
  hipLaunchKernelGGL(probe, dim3(nwg), dim3(64 * waves), 0, 0, d, 200000);
  hipDeviceSynchronize();
  std::vector<unsigned> h(nwg * waves * 2);
  hipMemcpy(h.data(), d, h.size() * 4, hipMemcpyDeviceToHost);
  int hist[16][4] = {};
  for (int g = 0; g < nwg; g ++) for (int w = 0; w < waves; w ++) hist[w][(h[(g * waves + w) * 2] >> 4) & 3] ++;
  for (int w = 0; w < waves; w ++) printf("wave %d of the workgroup: SIMD0 %d SIMD1 %d SIMD2 %d SIMD3 %d\n", w, hist[w][0], hist[w][1], hist[w][2], hist[w][3]);
  for (int g = 0; g < 12; g ++) { printf("wg %d:", g); for (int w = 0; w < waves; w ++) { unsigned id = h[(g * waves + w) * 2]; printf("  [xcc %u se %u cu %u simd %u slot %u]", h[(g * waves + w) * 2 + 1] & 15, (id >> 13) & 7, (id >> 8) & 15, (id >> 4) & 3, id & 15); } printf("\n"); }
  return 0;
}
