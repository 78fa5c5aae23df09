// Issue rate of the VALU instructions the mask kernel is made of (gfx950): cycles per wave64 instruction per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

#define BODY8(INS) INS INS INS INS INS INS INS INS
template <int KIND>
__global__ __launch_bounds__(256) void k(double *out, int iters, double seed)
{
  double a = seed + threadIdx.x, b = seed * 0.5, c = 0.25;
  unsigned u = threadIdx.x, w = 7;
  unsigned long long m;
  for (int i = 0; i < iters; i ++) {
    if (KIND == 0) { BODY8(asm volatile("v_add_f64 %0, %1, %2" : "=v"(c) : "v"(a), "v"(b));) }
    if (KIND == 1) { BODY8(asm volatile("v_cmp_ge_f64_e64 %0, %1, %2" : "=s"(m) : "v"(a), "v"(b));) }
    if (KIND == 2) { BODY8(asm volatile("v_cmp_ge_i32_e64 %0, %1, %2" : "=s"(m) : "v"(u), "v"(w));) }
    if (KIND == 3) { BODY8(asm volatile("v_addc_co_u32_e64 %0, vcc, %1, %1, %2" : "=v"(w) : "v"(u), "s"(m) : "vcc");) }
    if (KIND == 4) { BODY8(asm volatile("v_mov_b32_dpp %0, %1 wave_shr:1 row_mask:0xf bank_mask:0xf" : "=v"(w) : "v"(u));) }
    if (KIND == 5) { BODY8(asm volatile("v_mul_f64 %0, %1, %2" : "=v"(c) : "v"(a), "v"(b));) }
    if (KIND == 6) { BODY8(asm volatile("v_and_b32 %0, %1, %2" : "=v"(w) : "v"(u), "v"(w));) }
    if (KIND == 7) { BODY8(asm volatile("v_cmp_ge_f64_e32 vcc, %0, %1" : : "v"(a), "v"(b) : "vcc");) }
    if (KIND == 8) { BODY8(asm volatile("v_cmp_ge_u32_e64 %0, %1, %2" : "=s"(m) : "v"(u), "v"(w));) }
    if (KIND == 9) { BODY8(asm volatile("v_fma_f64 %0, %1, %2, %0" : "+v"(c) : "v"(a), "v"(b));) }
  }
  if (c == 1.2345 && w == 77 && m == 5) out[0] = c;
}

template <int KIND> void run(const char *name, double *out)
{
  const int iters = 20000, blocks = 256 * 8;       // 8 workgroups of 4 waves per CU: 8 waves per SIMD
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(256), 0, 0, out, 100, 1.5); CK(hipDeviceSynchronize());
  CK(hipEventRecord(a));
  hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(256), 0, 0, out, iters, 1.5);
  CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
  float ms; CK(hipEventElapsedTime(&ms, a, b));
  // per SIMD: 8 waves x iters x 8 instructions
  const double inst_per_simd = 8.0 * iters * 8;
  printf("%-28s %8.3f ms  -> %.2f ns per wave-instruction per SIMD (= %.1f cycles at 2.4 GHz)\n", name, ms, ms * 1e6 / inst_per_simd, ms * 1e6 / inst_per_simd * 2.4);
}

int main()
{
  double *out; CK(hipMalloc(&out, 8));
  run<0>("v_add_f64", out); run<5>("v_mul_f64", out); run<9>("v_fma_f64", out); run<1>("v_cmp_ge_f64_e64 (sgpr dst)", out); run<7>("v_cmp_ge_f64_e32 (vcc)", out);
  run<2>("v_cmp_ge_i32_e64", out); run<8>("v_cmp_ge_u32_e64", out); run<3>("v_addc_co_u32_e64", out); run<4>("v_mov_b32_dpp wave_shr", out); run<6>("v_and_b32", out);
  return 0;
}
