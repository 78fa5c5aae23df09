// Stand-alone probe: what read bandwidth can a pure streaming kernel reach on this GPU, as a function of load shape?
// hipcc -O3 --offload-arch=gfx950 -o bw_probe bw_probe.hip && ./bw_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

// block-contiguous segments: one segment = blockDim * U * 16 bytes, loads of a thread are blockDim*16 apart (each wave
// instruction reads 1 KiB contiguous); segments dealt round-robin to blocks
template <int U, bool NT>
__global__ __launch_bounds__(1024) void seg_read(const double2 *__restrict__ p, size_t nseg, double *out)
{
  double acc = 0.0;
  const size_t seg16 = (size_t)blockDim.x * U;
  for (size_t s = blockIdx.x; s < nseg; s += gridDim.x) {
    const double2 *q = p + s * seg16 + threadIdx.x;
    double2 v[U];
#pragma unroll
    for (int u = 0; u < U; u ++) {
      if (NT) { v[u].x = __builtin_nontemporal_load(&q[(size_t)u * blockDim.x].x); v[u].y = __builtin_nontemporal_load(&q[(size_t)u * blockDim.x].y); }
      else v[u] = q[(size_t)u * blockDim.x];
    }
#pragma unroll
    for (int u = 0; u < U; u ++) acc += v[u].x + v[u].y;
  }
  if (acc == 1.2345e300) out[0] = acc;
}

// software-pipelined variant: keep the next segment's loads in flight while summing the current one
template <int U>
__global__ __launch_bounds__(1024) void seg_read_pipe(const double2 *__restrict__ p, size_t nseg, double *out)
{
  double acc = 0.0;
  const size_t seg16 = (size_t)blockDim.x * U;
  size_t s = blockIdx.x;
  double2 cur[U], nxt[U];
  if (s < nseg) { const double2 *q = p + s * seg16 + threadIdx.x;
#pragma unroll
    for (int u = 0; u < U; u ++) cur[u] = q[(size_t)u * blockDim.x]; }
  for (; s < nseg; s += gridDim.x) {
    const size_t sn = s + gridDim.x;
    if (sn < nseg) { const double2 *q = p + sn * seg16 + threadIdx.x;
#pragma unroll
      for (int u = 0; u < U; u ++) nxt[u] = q[(size_t)u * blockDim.x]; }
#pragma unroll
    for (int u = 0; u < U; u ++) { acc += cur[u].x + cur[u].y; cur[u] = nxt[u]; }
  }
  if (acc == 1.2345e300) out[0] = acc;
}

// the mask kernel's walk without its arithmetic: wave = 128 columns x R rows (+2 halo rows), marching along z
// NTMODE: 0 plain loads, 1 all nontemporal, 2 nontemporal for the rows no other wavefront reads (r = 2 .. R-1)
// XW: wavefronts of a workgroup side by side along x (1: all stacked in y, as the mask kernel does; 4: a whole 512-column row)
// EDGELD: lanes 0 / 63 also fetch one 8-byte neighbour per own row (the mask kernel's x edges); USTORE: one byte per 4 lanes per
// own row is written (the mask kernel's summary stream, 1/8 byte per vertex)
template <int R, int NTMODE = 0, int XW = 1, int EDGELD = 0, int USTORE = 0, int YG = 0, int WORK = 0>
__global__ __launch_bounds__(1024) void march_read(const char *__restrict__ S, int DW, int DH, int DD, int zchunk, double *out, unsigned char *U = nullptr)
{
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int nzc = (DD + zchunk - 1) / zchunk;
  unsigned bx = blockIdx.x, by = blockIdx.y, bz = blockIdx.z;
  if (YG > 0) {
    const unsigned nb = gridDim.x * gridDim.y * gridDim.z, G = gridDim.x * YG;
    const unsigned b = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);
    if (b < (nb / (8u * G)) * (8u * G)) {
      const unsigned q = b % 8u, r = b / 8u, grp = (r / G) * 8u + q, in = r % G, ngy = gridDim.y / YG;
      bx = in % gridDim.x; by = (grp % ngy) * YG + in / gridDim.x; bz = grp / ngy;
    }
  }
  const unsigned slice = bz / nzc;
  const int z0 = (bz % nzc) * zchunk, z1 = min(z0 + zchunk, DD);
  const int wpb = blockDim.x >> 6;
  const int j0 = (by * (wpb / XW) + wv / XW) * R;
  const unsigned sy = DW * 8u, sz = (unsigned)DW * DH * 8u;
  const char *base = S + (size_t)slice * sz * DD + (size_t)((bx * XW + wv % XW) * 128 + 2 * lane) * 8;
  double acc = 0.0;
  double2 nn[R + 2];
  for (int k = z0 - 1; k <= z1; k ++) {
    const int kc = k < 0 ? 0 : (k >= DD ? DD - 1 : k);
#pragma unroll
    for (int r = 0; r < R + 2; r ++) {
      if (NTMODE == 3 && ((r == 0 && wv != 0) || (r == R + 1 && wv != wpb - 1))) { nn[r] = double2{0.0, 0.0}; continue; }
      int j = j0 + r - 1; j = j < 0 ? 0 : (j >= DH ? DH - 1 : j);
      if (WORK == 1000) {
        typedef unsigned u4 __attribute__((ext_vector_type(4)));
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void *)(S + (size_t)slice * sz * DD), 0, (int)(sz * (unsigned)DD), 0x00020000);
        const unsigned vo = (unsigned)((bx * XW + wv % XW) * 128 + 2 * lane) * 8u, so = sz * (unsigned)kc + sy * (unsigned)j;
        const bool ntl = NTMODE == 1 || (NTMODE == 2 && r >= 2 && r <= R - 1) || (NTMODE == 3 && r >= 1 && r <= R);
        const u4 raw = ntl ? __builtin_amdgcn_raw_buffer_load_b128(rs, vo, so, 2) : __builtin_amdgcn_raw_buffer_load_b128(rs, vo, so, 0);
        nn[r] = __builtin_bit_cast(double2, raw);
        continue;
      }
      const double2 *q = reinterpret_cast<const double2 *>(base + (size_t)sz * kc + (size_t)sy * j);
      if (NTMODE == 1 || (NTMODE == 2 && r >= 2 && r <= R - 1) || (NTMODE == 3 && r >= 1 && r <= R)) { nn[r].x = __builtin_nontemporal_load(&q->x); nn[r].y = __builtin_nontemporal_load(&q->y); }
      else nn[r] = *q;
    }
#pragma unroll
    for (int r = 0; r < R + 2; r ++) acc += nn[r].x + nn[r].y;
    if (WORK > 0 && WORK < 1000) {   // register-only arithmetic next to the stream: WORK x 8 independent FP64 FMAs (WORK < 0: 32-bit integer ops instead)
      double w0 = acc, w1 = acc + 1, w2 = acc + 2, w3 = acc + 3, w4 = acc + 4, w5 = acc + 5, w6 = acc + 6, w7 = acc + 7;
#pragma unroll
      for (int t = 0; t < WORK; t ++) { w0 = w0 * 1.000001 + 0.5; w1 = w1 * 1.000001 + 0.5; w2 = w2 * 1.000001 + 0.5; w3 = w3 * 1.000001 + 0.5;
                                        w4 = w4 * 1.000001 + 0.5; w5 = w5 * 1.000001 + 0.5; w6 = w6 * 1.000001 + 0.5; w7 = w7 * 1.000001 + 0.5; }
      acc = ((w0 + w1) + (w2 + w3)) + ((w4 + w5) + (w6 + w7));
    }
    if (WORK < 0) {
      unsigned u0 = (unsigned)acc, u1 = u0 + 1, u2 = u0 + 2, u3 = u0 + 3, u4 = u0 + 4, u5 = u0 + 5, u6 = u0 + 6, u7 = u0 + 7;
#pragma unroll
      for (int t = 0; t < -WORK; t ++) { u0 = u0 * 1664525u + 1013904223u; u1 = u1 * 1664525u + 1013904223u; u2 = u2 * 1664525u + 1013904223u; u3 = u3 * 1664525u + 1013904223u;
                                         u4 = u4 * 1664525u + 1013904223u; u5 = u5 * 1664525u + 1013904223u; u6 = u6 * 1664525u + 1013904223u; u7 = u7 * 1664525u + 1013904223u; }
      acc += (double)(u0 ^ u1 ^ u2 ^ u3 ^ u4 ^ u5 ^ u6 ^ u7);
    }
    if (EDGELD == 2 && (lane < R || lane >= 64 - R)) {
      const int rr = lane < R ? lane : lane - (64 - R);
      const int ic = bx * 128, ih = lane < R ? (ic > 0 ? ic - 1 : 0) : (ic + 128 < DW ? ic + 128 : DW - 1);
      int j = j0 + rr; j = j >= DH ? DH - 1 : j;
      acc += *reinterpret_cast<const double *>(S + (size_t)slice * sz * DD + (size_t)sz * kc + (size_t)sy * j + (size_t)ih * 8);
    }
    if (EDGELD == 1 && (lane == 0 || lane == 63)) {
      const int ic = bx * 128 + 2 * lane, ih = lane == 0 ? (ic > 0 ? ic - 1 : 0) : (ic + 2 < DW ? ic + 2 : DW - 1);
#pragma unroll
      for (int r = 1; r <= R; r ++) {
        int j = j0 + r - 1; j = j >= DH ? DH - 1 : j;
        acc += *reinterpret_cast<const double *>(S + (size_t)slice * sz * DD + (size_t)sz * kc + (size_t)sy * j + (size_t)ih * 8);
      }
    }
    if (USTORE == 1 && k >= z0 && k < z1 && (lane & 3) == 0) {
#pragma unroll
      for (int r = 0; r < R; r ++)
        U[(size_t)slice * (DW / 8) * DH * DD + (size_t)(DW / 8) * ((size_t)(j0 + r) + (size_t)DH * k) + ((bx * 128 + 2 * lane) >> 3)] = (unsigned char)(acc > 0.5);
    }
    if (USTORE == 2 && k >= z0 && k < z1 && lane < 4) {       // same bytes, 4 lanes x 4 B per row
#pragma unroll
      for (int r = 0; r < R; r ++)
        *reinterpret_cast<unsigned *>(U + (size_t)slice * (DW / 8) * DH * DD + (size_t)(DW / 8) * ((size_t)(j0 + r) + (size_t)DH * k) + bx * 16 + lane * 4) = (unsigned)(acc > 0.5);
    }
    if (USTORE >= 4 && k >= z0 && k < z1 && lane < R) {
      typedef unsigned u4 __attribute__((ext_vector_type(4)));
      unsigned char *q = U + (size_t)slice * (DW / 8) * DH * DD + (((size_t)bx * DD + k) * DH + j0 + lane) * 16;
      if (USTORE == 4) { u4 v = {(unsigned)k, 1u, 2u, 3u}; *reinterpret_cast<u4 *>(q) = v; }                       // value independent of the loads
      if (USTORE == 5 && ((k - z0) & 3) == 3) { u4 v = {(unsigned)(acc > 0.5), 1u, 2u, 3u};                         // every 4th plane, 4x the bytes
        for (int t = 0; t < 4; t ++) *reinterpret_cast<u4 *>(q - (size_t)t * DH * 16) = v; }
      if (USTORE == 6) { u4 v = {(unsigned)(acc > 0.5), 1u, 2u, 3u}; __builtin_nontemporal_store(v, reinterpret_cast<u4 *>(q)); }
      if (USTORE == 7 && ((k - z0) & 7) == 7) { u4 v = {(unsigned)(acc > 0.5), 1u, 2u, 3u};
        for (int t = 0; t < 8; t ++) *reinterpret_cast<u4 *>(q - (size_t)t * DH * 16) = v; }
    }
    if (USTORE == 3 && k >= z0 && k < z1 && lane < R) {       // tiled layout: a wavefront's R rows x 16 B are contiguous, lane r stores row r
      typedef unsigned u4 __attribute__((ext_vector_type(4)));
      u4 v = {(unsigned)(acc > 0.5), 1u, 2u, 3u};
      *reinterpret_cast<u4 *>(U + (size_t)slice * (DW / 8) * DH * DD + (((size_t)bx * DD + k) * DH + j0 + lane) * 16) = v;
    }
  }
  if (acc == 1.2345e300) out[0] = acc;
}

template <class F> float time_it(F f, int reps = 5)
{
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  f(); CK(hipDeviceSynchronize());
  CK(hipEventRecord(a));
  for (int i = 0; i < reps; i ++) f();
  CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
  float ms; CK(hipEventElapsedTime(&ms, a, b));
  return ms / reps;
}

// the vector mask kernel's walk without its arithmetic: a wavefront reads runs of 4 KB (64 groups of 4 vertices x 16 bytes), grid-stride
// over one slice per blockIdx.y, the next run's loads in flight while the current one is consumed.
// PAT 0: lane l reads its group's 64 contiguous bytes (4 loads 16 B apart: every wave instruction touches all 32 lines of the run);
// PAT 1: load q of lane l reads bytes q * 1024 + l * 16 (every wave instruction reads 1 KB contiguous: 8 whole lines)
// ST 1: the even lanes store one summary byte per run and lane (32 contiguous bytes per wave instruction); ST 2: every lane its mask word too;
// ST 3: the summary bytes gathered into one 8-byte store per 8 lanes' worth (4 lanes store a dword each... no: lanes 0..3 store 8 bytes each)
template <int PAT, int ST = 0>
__global__ __launch_bounds__(256) void vec_read(const char *__restrict__ p, unsigned groups_per_slice, double *out, unsigned char *U = nullptr, unsigned *M = nullptr)
{
  const unsigned lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const char *base = p + (size_t)blockIdx.y * groups_per_slice * 64u;
  const unsigned step = gridDim.x * 256u;
  const unsigned lo = PAT == 0 ? lane * 64u : lane * 16u, qs = PAT == 0 ? 16u : 1024u;
  double acc = 0.0;
  double2 cur[4], nxt[4];
  unsigned gw = (blockIdx.x * 4u + wv) * 64u;
  if (gw < groups_per_slice) for (int q = 0; q < 4; q ++) cur[q] = *(const double2 *)(base + (size_t)gw * 64u + lo + q * qs);
  unsigned char held = 0; size_t held_at = ~(size_t)0;      // ST 5: the summary byte of the previous run, stored right behind the next run's loads
  for (; gw < groups_per_slice; gw += step) {
    const unsigned gn = gw + step < groups_per_slice ? gw + step : gw;
#pragma unroll
    for (int q = 0; q < 4; q ++) nxt[q] = *(const double2 *)(base + (size_t)gn * 64u + lo + q * qs);
    if (ST == 5 && held_at != ~(size_t)0 && (lane & 1) == 0) U[held_at] = held;
#pragma unroll
    for (int q = 0; q < 4; q ++) { acc += cur[q].x + cur[q].y; cur[q] = nxt[q]; }
    if (ST == 1 || ST == 2) { if ((lane & 1) == 0) U[(size_t)blockIdx.y * (groups_per_slice / 2) + (gw + lane) / 2] = (unsigned char)(acc != 0.0); }
    if (ST == 2) M[(size_t)blockIdx.y * groups_per_slice + gw + lane] = (unsigned)(acc != 0.0);
    if (ST == 6) { if ((lane & 1) == 0) __builtin_nontemporal_store((unsigned char)(acc != 0.0), &U[(size_t)blockIdx.y * (groups_per_slice / 2) + (gw + lane) / 2]); }
    if (ST == 7) {      // the workgroup's four runs are side by side: their 4 x 32 summary bytes through LDS, ONE 128-byte store (wave 0, 32 lanes x 4 bytes)
      __shared__ unsigned char s_u[128];
      if ((lane & 1) == 0) s_u[wv * 32u + lane / 2] = (unsigned char)(acc != 0.0);
      __syncthreads();
      const unsigned g_wg = gw - wv * 64u;      // the workgroup's first group of this round
      if (wv == 0 && lane < 32) ((unsigned *)(U + (size_t)blockIdx.y * (groups_per_slice / 2) + g_wg / 2))[lane] = ((const unsigned *)s_u)[lane];
      __syncthreads();
    }
    if (ST == 5) { held = (unsigned char)(acc != 0.0); held_at = (size_t)blockIdx.y * (groups_per_slice / 2) + (gw + lane) / 2; }
    if (ST == 3) {      // the run's 32 summary bytes as four 8-byte stores (lanes 0 .. 3)
      const unsigned long long b = __builtin_amdgcn_ballot_w64(acc != 0.0);
      if (lane < 4) ((unsigned long long *)U)[((size_t)blockIdx.y * (groups_per_slice / 2) + gw / 2) / 8 + lane] = b >> lane;
    }
  }
  if (ST == 5 && held_at != ~(size_t)0 && (lane & 1) == 0) U[held_at] = held;
  if (acc == 1.2345e300) out[0] = acc;
}

// the same walk with CONSECUTIVE runs per wavefront: NC runs of 4 KB side by side (the 32 summary bytes of each land in the same 128-byte
// line, written from one CU), then on by gridDim.x * 4 * NC runs
template <int NC, int ST>
__global__ __launch_bounds__(256) void vec_read_consec(const char *__restrict__ p, unsigned groups_per_slice, double *out, unsigned char *U)
{
  const unsigned lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const char *base = p + (size_t)blockIdx.y * groups_per_slice * 64u;
  const unsigned step = gridDim.x * 256u * NC;
  const unsigned lo = lane * 64u;
  double acc = 0.0;
  double2 cur[4], nxt[4];
  unsigned g0 = (blockIdx.x * 4u + wv) * 64u * NC;
  unsigned i = 0;
  if (g0 < groups_per_slice) for (int q = 0; q < 4; q ++) cur[q] = *(const double2 *)(base + (size_t)g0 * 64u + lo + q * 16u);
  while (g0 < groups_per_slice) {
    const unsigned gw = g0 + i * 64u;
    unsigned ni = i + 1, ng0 = g0;
    if (ni == NC) { ni = 0; ng0 = g0 + step; }
    const unsigned gn = ng0 < groups_per_slice ? ng0 + ni * 64u : gw;
#pragma unroll
    for (int q = 0; q < 4; q ++) nxt[q] = *(const double2 *)(base + (size_t)gn * 64u + lo + q * 16u);
#pragma unroll
    for (int q = 0; q < 4; q ++) { acc += cur[q].x + cur[q].y; cur[q] = nxt[q]; }
    if (ST == 1) { if ((lane & 1) == 0) U[(size_t)blockIdx.y * (groups_per_slice / 2) + (gw + lane) / 2] = (unsigned char)(acc != 0.0); }
    i = ni; g0 = ng0;
  }
  if (acc == 1.2345e300) out[0] = acc;
}

int main(int argc, char **argv)
{
  const bool only_march = argc > 1;
  const int NSARG = argc > 2 ? atoi(argv[2]) : 8;
  const size_t bytes = (size_t)NSARG << 30;
  double2 *p; double *out;
  CK(hipMalloc(&p, bytes)); CK(hipMalloc(&out, 8));
  CK(hipMemset(p, 0, bytes));
  printf("%-44s %8s %8s\n", "variant", "ms", "TB/s");
  auto rep = [&](const char *name, float ms, double b) { printf("%-44s %8.3f %8.2f\n", name, ms, b / ms / 1e9); fflush(stdout); };
  if (argc > 1 && argv[1][0] == 'v') {      // double_gyre 2048 x 1024 x 128: 128 slices of 32 MiB
    const unsigned gps = 2048u * 1024u / 4u;
    for (int bx : {256, 512, 1024}) for (int rep_ = 0; rep_ < 2; rep_ ++) {
      char nm[96];
      snprintf(nm, 96, "vec PAT=0 (64 B per lane) grid=%dx128", bx);
      rep(nm, time_it([&] { hipLaunchKernelGGL((vec_read<0>), dim3(bx, 128), dim3(256), 0, 0, (const char *)p, gps, out); }), 128.0 * gps * 64.0);
      snprintf(nm, 96, "vec PAT=1 (1 KB per instruction) grid=%dx128", bx);
      rep(nm, time_it([&] { hipLaunchKernelGGL((vec_read<1>), dim3(bx, 128), dim3(256), 0, 0, (const char *)p, gps, out); }), 128.0 * gps * 64.0);
      unsigned char *Uv = (unsigned char *)p + (5ull << 30); unsigned *Mv = (unsigned *)((char *)p + (6ull << 30));      // (beyond the 4 GiB that are read)
      snprintf(nm, 96, "vec PAT=0 + summary byte stores grid=%dx128", bx);
      rep(nm, time_it([&] { hipLaunchKernelGGL((vec_read<0, 1>), dim3(bx, 128), dim3(256), 0, 0, (const char *)p, gps, out, Uv, Mv); }), 128.0 * gps * 64.0);
      snprintf(nm, 96, "vec PAT=0 + summary bytes + mask words grid=%dx128", bx);
      rep(nm, time_it([&] { hipLaunchKernelGGL((vec_read<0, 2>), dim3(bx, 128), dim3(256), 0, 0, (const char *)p, gps, out, Uv, Mv); }), 128.0 * gps * 64.0);
      snprintf(nm, 96, "vec PAT=0 + summary bytes one run late grid=%dx128", bx);
      rep(nm, time_it([&] { hipLaunchKernelGGL((vec_read<0, 5>), dim3(bx, 128), dim3(256), 0, 0, (const char *)p, gps, out, Uv, Mv); }), 128.0 * gps * 64.0);
      snprintf(nm, 96, "vec PAT=0 + summary bytes nontemporal grid=%dx128", bx);
      rep(nm, time_it([&] { hipLaunchKernelGGL((vec_read<0, 6>), dim3(bx, 128), dim3(256), 0, 0, (const char *)p, gps, out, Uv, Mv); }), 128.0 * gps * 64.0);
      snprintf(nm, 96, "vec PAT=0 + summaries 128 B per workgroup grid=%dx128", bx);
      rep(nm, time_it([&] { hipLaunchKernelGGL((vec_read<0, 7>), dim3(bx, 128), dim3(256), 0, 0, (const char *)p, gps, out, Uv, Mv); }), 128.0 * gps * 64.0);
      snprintf(nm, 96, "vec 4 consecutive runs, no stores grid=%dx128", bx / 4);
      rep(nm, time_it([&] { hipLaunchKernelGGL((vec_read_consec<4, 0>), dim3(bx / 4, 128), dim3(256), 0, 0, (const char *)p, gps, out, Uv); }), 128.0 * gps * 64.0);
      snprintf(nm, 96, "vec 4 consecutive runs + summary bytes grid=%dx128", bx / 4);
      rep(nm, time_it([&] { hipLaunchKernelGGL((vec_read_consec<4, 1>), dim3(bx / 4, 128), dim3(256), 0, 0, (const char *)p, gps, out, Uv); }), 128.0 * gps * 64.0);
      snprintf(nm, 96, "vec 4 consecutive runs + summary bytes grid=%dx128", bx);
      rep(nm, time_it([&] { hipLaunchKernelGGL((vec_read_consec<4, 1>), dim3(bx, 128), dim3(256), 0, 0, (const char *)p, gps, out, Uv); }), 128.0 * gps * 64.0);
      snprintf(nm, 96, "vec 8 consecutive runs + summary bytes grid=%dx128", bx / 4);
      rep(nm, time_it([&] { hipLaunchKernelGGL((vec_read_consec<8, 1>), dim3(bx / 4, 128), dim3(256), 0, 0, (const char *)p, gps, out, Uv); }), 128.0 * gps * 64.0);
      snprintf(nm, 96, "vec PAT=0 + summaries as 4 x 8 B grid=%dx128", bx);
      rep(nm, time_it([&] { hipLaunchKernelGGL((vec_read<0, 3>), dim3(bx, 128), dim3(256), 0, 0, (const char *)p, gps, out, Uv, Mv); }), 128.0 * gps * 64.0);
    }
    return 0;
  }
#define SEG(U, NT, BD, GRID) { const size_t nseg = bytes / ((size_t)BD * U * 16); char nm[96]; snprintf(nm, 96, "seg U=%d nt=%d block=%d grid=%d", U, NT, BD, GRID); \
    rep(nm, time_it([&] { hipLaunchKernelGGL((seg_read<U, NT>), dim3(GRID), dim3(BD), 0, 0, p, nseg, out); }), (double)bytes); }
  if (!only_march) {
  SEG(1, false, 256, 4096) SEG(2, false, 256, 4096) SEG(4, false, 256, 4096) SEG(8, false, 256, 4096)
  SEG(4, false, 256, 2048) SEG(4, false, 256, 8192) SEG(4, false, 256, 16384) SEG(4, false, 256, 65536)
  SEG(4, false, 512, 2048) SEG(4, false, 1024, 1024) SEG(8, false, 512, 2048) SEG(4, false, 64, 16384) SEG(8, false, 64, 16384)
  SEG(4, true, 256, 4096) SEG(8, true, 256, 4096) SEG(4, true, 256, 16384)
#define PIPE(U, BD, GRID) { const size_t nseg = bytes / ((size_t)BD * U * 16); char nm[96]; snprintf(nm, 96, "pipe U=%d block=%d grid=%d", U, BD, GRID); \
    rep(nm, time_it([&] { hipLaunchKernelGGL((seg_read_pipe<U>), dim3(GRID), dim3(BD), 0, 0, p, nseg, out); }), (double)bytes); }
  PIPE(2, 256, 4096) PIPE(4, 256, 4096) PIPE(4, 256, 2048) PIPE(8, 256, 2048) PIPE(4, 256, 1024) PIPE(6, 256, 3072) PIPE(6, 256, 768)
  }
  // marching walk over 8 slices of 512^3 (8 GiB)
  const int DW = 512, DH = 512, DD = 512, NS = NSARG;
#define MARCH(R, NTM, XW, WPB, ZC) { const int nzc = DD / ZC; char nm[96]; snprintf(nm, 96, "march R=%d nt=%d xw=%d wpb=%d zchunk=%d", R, NTM, XW, WPB, ZC); \
    rep(nm, time_it([&] { hipLaunchKernelGGL((march_read<R, NTM, XW>), dim3(DW / (128 * XW), DH / (R * (WPB / XW)), nzc * NS), dim3(64 * WPB), 0, 0, (const char *)p, DW, DH, DD, ZC, out); }), (double)bytes); }
  unsigned char *U; CK(hipMalloc(&U, bytes / 64));
#define MARCHG(R, NTM, E, US, WPB, ZC, YG) { const int nzc = DD / ZC; char nm[96]; snprintf(nm, 96, "march R=%d nt=%d edge=%d ust=%d wpb=%d zc=%d yg=%d", R, NTM, E, US, WPB, ZC, YG); \
    rep(nm, time_it([&] { hipLaunchKernelGGL((march_read<R, NTM, 1, E, US, YG>), dim3(DW / 128, DH / (R * WPB), nzc * NS), dim3(64 * WPB), 0, 0, (const char *)p, DW, DH, DD, ZC, out, U); }), (double)bytes); }
#define MARCHW(R, NTM, YG, W) { const int nzc = DD / 32; char nm[96]; snprintf(nm, 96, "march R=%d nt=%d yg=%d work=%d", R, NTM, YG, W); \
    rep(nm, time_it([&] { hipLaunchKernelGGL((march_read<R, NTM, 1, 0, 0, YG, W>), dim3(DW / 128, DH / (R * 4), nzc * NS), dim3(256), 0, 0, (const char *)p, DW, DH, DD, 32, out, U); }), (double)bytes); }
  MARCHW(4, 0, 4, 0) MARCHW(4, 0, 4, 1000) MARCHW(8, 2, 4, 0) MARCHW(8, 2, 4, 1000) MARCHW(8, 0, 4, 0) MARCHW(8, 0, 4, 1000) MARCHW(8, 1, 4, 0) MARCHW(8, 1, 4, 1000)
  return 0;
}
