// How fast does a FRESH pageable host array (what ndarray<double> is: malloc'ed per timestep by the caller) reach HBM?  Round 6: the patched
// reference tracker's per-step cost is this upload (134 MB at 256^3).  Variants: the runtime's own pageable copy in one call, in chunks, from
// several host threads on streams of their own, hipHostRegister around the copy, and a pinned source for reference.
//   hipcc --offload-arch=gfx950 -O2 -o h2d_probe h2d_probe.hip -lpthread && ./h2d_probe [MiB]
#include <hip/hip_runtime.h>
#include <sys/mman.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
using clk = std::chrono::steady_clock;
static double ms(clk::time_point a, clk::time_point b) { return std::chrono::duration<double, std::milli>(b - a).count(); }

static char *fresh(size_t bytes) { char *p = (char *)malloc(bytes); memset(p, 1, bytes); return p; }   // (touched: the caller wrote its data)

int main(int argc, char **argv)
{
  const size_t bytes = (size_t)(argc > 1 ? atoi(argv[1]) : 128) << 20;
  char *d; CK(hipMalloc(&d, bytes));
  hipStream_t st[8]; for (auto &s : st) CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  { char *h = fresh(bytes); CK(hipMemcpyAsync(d, h, bytes, hipMemcpyHostToDevice, st[0])); CK(hipStreamSynchronize(st[0])); free(h); }   // warm-up
  auto report = [&](const char *name, std::vector<double> &v) {
    double best = 1e9, sum = 0; for (double x : v) { best = x < best ? x : best; sum += x; }
    printf("%-44s mean %7.3f ms  best %7.3f ms  = %5.1f GB/s (mean)\n", name, sum / v.size(), best, bytes / (sum / v.size()) / 1e6);
  };
  const int reps = 8;
  std::vector<double> v;
  v.clear();
  for (int r = 0; r < reps; r ++) { char *h = fresh(bytes); auto t0 = clk::now(); CK(hipMemcpyAsync(d, h, bytes, hipMemcpyHostToDevice, st[0])); CK(hipStreamSynchronize(st[0])); v.push_back(ms(t0, clk::now())); free(h); }
  report("fresh pageable, one hipMemcpyAsync", v);
  v.clear();
  for (int r = 0; r < reps; r ++) { char *h = fresh(bytes); auto t0 = clk::now(); CK(hipMemcpy(d, h, bytes, hipMemcpyHostToDevice)); v.push_back(ms(t0, clk::now())); free(h); }
  report("fresh pageable, one synchronous hipMemcpy", v);
  v.clear();
  for (int r = 0; r < reps; r ++) { char *h = fresh(bytes); auto t0 = clk::now(); CK(hipMemcpyAsync(d, h, bytes, hipMemcpyHostToDevice, 0)); CK(hipStreamSynchronize(0)); v.push_back(ms(t0, clk::now())); free(h); }
  report("fresh pageable, hipMemcpyAsync on the null stream", v);
  v.clear();
  for (int r = 0; r < reps; r ++) {      // the same, but the array's pages touched by the copy's own thread right before (a caller that has just written it)
    char *h = (char *)malloc(bytes); for (size_t i = 0; i < bytes; i += 4096) h[i] = 1;
    auto t0 = clk::now(); CK(hipMemcpyAsync(d, h, bytes, hipMemcpyHostToDevice, st[0])); CK(hipStreamSynchronize(st[0])); v.push_back(ms(t0, clk::now())); free(h);
  }
  report("fresh pageable (one write per page), hipMemcpyAsync", v);
  v.clear();
  for (int r = 0; r < reps; r ++) {      // 2 MiB-aligned and advised huge: what the kernel has to pin is 64 pages instead of 32 768
    char *h = (char *)aligned_alloc(2u << 20, bytes); madvise(h, bytes, MADV_HUGEPAGE); memset(h, 1, bytes);
    auto t0 = clk::now(); CK(hipMemcpyAsync(d, h, bytes, hipMemcpyHostToDevice, st[0])); CK(hipStreamSynchronize(st[0])); v.push_back(ms(t0, clk::now())); free(h);
  }
  report("fresh pageable, 2 MiB-aligned + MADV_HUGEPAGE", v);
  for (int chunks : {2, 4, 8, 16}) {
    v.clear();
    for (int r = 0; r < reps; r ++) {
      char *h = fresh(bytes); auto t0 = clk::now();
      const size_t cb = bytes / chunks;
      for (int c = 0; c < chunks; c ++) CK(hipMemcpyAsync(d + c * cb, h + c * cb, cb, hipMemcpyHostToDevice, st[0]));
      CK(hipStreamSynchronize(st[0])); v.push_back(ms(t0, clk::now())); free(h);
    }
    char nm[64]; snprintf(nm, sizeof nm, "fresh pageable, %d chunks, one stream", chunks); report(nm, v);
  }
  for (int nthreads : {2, 4, 8}) {
    v.clear();
    for (int r = 0; r < reps; r ++) {
      char *h = fresh(bytes); auto t0 = clk::now();
      const size_t cb = bytes / nthreads;
      std::vector<std::thread> th;
      for (int c = 0; c < nthreads; c ++) th.emplace_back([&, c] { CK(hipMemcpyAsync(d + c * cb, h + c * cb, cb, hipMemcpyHostToDevice, st[c])); CK(hipStreamSynchronize(st[c])); });
      for (auto &t : th) t.join();
      v.push_back(ms(t0, clk::now())); free(h);
    }
    char nm[64]; snprintf(nm, sizeof nm, "fresh pageable, %d threads x own stream", nthreads); report(nm, v);
  }
  v.clear();
  for (int r = 0; r < reps; r ++) {
    char *h = fresh(bytes); auto t0 = clk::now();
    CK(hipHostRegister(h, bytes, hipHostRegisterDefault)); CK(hipMemcpyAsync(d, h, bytes, hipMemcpyHostToDevice, st[0])); CK(hipStreamSynchronize(st[0])); CK(hipHostUnregister(h));
    v.push_back(ms(t0, clk::now())); free(h);
  }
  report("fresh pageable, hipHostRegister + copy + unreg", v);
  {
    v.clear();
    char *h = fresh(bytes);
    for (int r = 0; r < reps; r ++) { auto t0 = clk::now(); CK(hipMemcpyAsync(d, h, bytes, hipMemcpyHostToDevice, st[0])); CK(hipStreamSynchronize(st[0])); v.push_back(ms(t0, clk::now())); }
    report("the SAME pageable array again and again", v); free(h);
  }
  {
    // staging: host threads memcpy pieces into a ring of pinned buffers, each piece DMA'd as soon as it is there
    const size_t piece = 4u << 20; const int ring = 8;
    char *pin; CK(hipHostMalloc(&pin, piece * ring, hipHostMallocDefault));
    hipEvent_t ev[ring]; for (auto &e : ev) CK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    for (int nthreads : {1, 2, 4}) {
      v.clear();
      for (int r = 0; r < reps; r ++) {
        char *h = fresh(bytes); auto t0 = clk::now();
        const size_t np = bytes / piece;
        for (size_t i = 0; i < np; i ++) {
          const int slot = (int)(i % ring);
          if (i >= (size_t)ring) CK(hipEventSynchronize(ev[slot]));
          if (nthreads == 1) memcpy(pin + slot * piece, h + i * piece, piece);
          else {
            std::vector<std::thread> th; const size_t sub = piece / nthreads;
            for (int c = 0; c < nthreads; c ++) th.emplace_back([&, c] { memcpy(pin + slot * piece + c * sub, h + i * piece + c * sub, sub); });
            for (auto &t : th) t.join();
          }
          CK(hipMemcpyAsync(d + i * piece, pin + slot * piece, piece, hipMemcpyHostToDevice, st[0]));
          CK(hipEventRecord(ev[slot], st[0]));
        }
        CK(hipStreamSynchronize(st[0])); v.push_back(ms(t0, clk::now())); free(h);
      }
      char nm[64]; snprintf(nm, sizeof nm, "staged through a pinned ring, %d copy threads", nthreads); report(nm, v);
    }
  }
  {
    v.clear();
    char *h; CK(hipHostMalloc(&h, bytes, hipHostMallocDefault)); memset(h, 1, bytes);
    for (int r = 0; r < reps; r ++) { auto t0 = clk::now(); CK(hipMemcpyAsync(d, h, bytes, hipMemcpyHostToDevice, st[0])); CK(hipStreamSynchronize(st[0])); v.push_back(ms(t0, clk::now())); }
    report("pinned source (hipHostMalloc)", v);
  }
  return 0;
}
