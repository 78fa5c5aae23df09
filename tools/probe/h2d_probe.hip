// How fast does a FRESH pageable host array (what ndarray<double> is: malloc'ed per timestep by the caller) reach HBM?  Round 6: the patched
// reference tracker's per-step cost is this upload (134 MB at 256^3).  Variants: the runtime's own pageable copy in one call, in chunks, from
// several host threads on streams of their own, hipHostRegister around the copy, and a pinned source for reference.
//   hipcc --offload-arch=gfx950 -O2 -o h2d_probe h2d_probe.hip -lpthread && ./h2d_probe [MiB]
#include <hip/hip_runtime.h>
#include <sys/mman.h>
#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <mutex>
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
    // staging, every thread for itself: thread c takes pieces c, c + T, ... through its own ring of pinned buffers and its own stream (the
    // threads are started once per array: what a library without a thread pool would do)
    const int ring = 4;
    for (size_t piece : {(size_t)1 << 20, (size_t)2 << 20, (size_t)4 << 20})
      for (int nthreads : {2, 4, 6, 8}) {
        char *pin; CK(hipHostMalloc(&pin, piece * ring * nthreads, hipHostMallocDefault));
        std::vector<hipEvent_t> ev((size_t)ring * nthreads); for (auto &e : ev) CK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
        v.clear();
        for (int r = 0; r < reps; r ++) {
          char *h = fresh(bytes); auto t0 = clk::now();
          const size_t np = (bytes + piece - 1) / piece;
          std::vector<std::thread> th;
          for (int c = 0; c < nthreads; c ++) th.emplace_back([&, c] {
            char *mine = pin + (size_t)c * ring * piece; size_t mycount = 0;
            for (size_t i = c; i < np; i += nthreads, mycount ++) {
              const int slot = (int)(mycount % ring);
              if (mycount >= (size_t)ring) CK(hipEventSynchronize(ev[(size_t)c * ring + slot]));
              const size_t len = std::min(piece, bytes - i * piece);
              memcpy(mine + slot * piece, h + i * piece, len);
              CK(hipMemcpyAsync(d + i * piece, mine + slot * piece, len, hipMemcpyHostToDevice, st[c]));
              CK(hipEventRecord(ev[(size_t)c * ring + slot], st[c]));
            }
            CK(hipStreamSynchronize(st[c]));
          });
          for (auto &t : th) t.join();
          v.push_back(ms(t0, clk::now())); free(h);
        }
        char nm[96]; snprintf(nm, sizeof nm, "staged, %d threads x own ring+stream, %zu MiB pieces", nthreads, piece >> 20); report(nm, v);
        for (auto &e : ev) CK(hipEventDestroy(e));
        CK(hipHostFree(pin));
      }
  }
  {
    // the same with threads that are kept: T - 1 workers wait on a condition variable, the caller is worker 0; pieces are handed out by an
    // atomic counter; every worker has its own ring of pinned buffers, its own stream
    struct Pool {
      int T; size_t piece; int ring; char *pin; std::vector<hipEvent_t> ev; hipStream_t *st;
      std::mutex mu; std::condition_variable cv; unsigned long long gen = 0; bool stop = false;
      const char *src = nullptr; char *dst = nullptr; size_t bytes = 0; std::atomic<size_t> next{0}; std::atomic<int> done{0};
      std::vector<std::thread> th;
      void work(int c) {
        char *mine = pin + (size_t)c * ring * piece; size_t mycount = 0; const size_t np = (bytes + piece - 1) / piece;
        for (;;) {
          const size_t i = next.fetch_add(1); if (i >= np) break;
          const int slot = (int)(mycount % ring);
          if (mycount >= (size_t)ring) CK(hipEventSynchronize(ev[(size_t)c * ring + slot]));
          const size_t len = std::min(piece, bytes - i * piece);
          memcpy(mine + slot * piece, src + i * piece, len);
          CK(hipMemcpyAsync(dst + i * piece, mine + slot * piece, len, hipMemcpyHostToDevice, st[c]));
          CK(hipEventRecord(ev[(size_t)c * ring + slot], st[c]));
          mycount ++;
        }
        CK(hipStreamSynchronize(st[c]));
      }
      void loop(int c) {
        unsigned long long seen = 0;
        for (;;) {
          { std::unique_lock<std::mutex> l(mu); cv.wait(l, [&] { return stop || gen != seen; }); if (stop) return; seen = gen; }
          work(c); done.fetch_add(1);
        }
      }
      void run(char *d_, const char *s_, size_t b_) {
        { std::lock_guard<std::mutex> l(mu); src = s_; dst = d_; bytes = b_; next = 0; done = 0; gen ++; }
        cv.notify_all();
        work(0);
        while (done.load() < T - 1) std::this_thread::yield();
      }
    };
    for (size_t piece : {(size_t)2 << 20, (size_t)4 << 20})
      for (int T : {3, 4, 6}) {
        Pool P; P.T = T; P.piece = piece; P.ring = 4; P.st = st;
        CK(hipHostMalloc(&P.pin, piece * P.ring * T, hipHostMallocDefault));
        P.ev.resize((size_t)P.ring * T); for (auto &e : P.ev) CK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
        for (int c = 1; c < T; c ++) P.th.emplace_back([&P, c] { P.loop(c); });
        { char *h = fresh(bytes); P.run(d, h, bytes); free(h); }
        v.clear();
        for (int r = 0; r < reps; r ++) { char *h = fresh(bytes); auto t0 = clk::now(); P.run(d, h, bytes); v.push_back(ms(t0, clk::now())); free(h); }
        char nm[96]; snprintf(nm, sizeof nm, "staged, kept pool of %d, %zu MiB pieces", T, piece >> 20); report(nm, v);
        { v.clear(); char *h = fresh(bytes); for (int r = 0; r < reps; r ++) { auto t0 = clk::now(); P.run(d, h, bytes); v.push_back(ms(t0, clk::now())); } free(h);
          snprintf(nm, sizeof nm, "   ... the SAME array again and again"); report(nm, v); }
        { std::lock_guard<std::mutex> l(P.mu); P.stop = true; } P.cv.notify_all();
        for (auto &t : P.th) t.join();
        for (auto &e : P.ev) CK(hipEventDestroy(e));
        CK(hipHostFree(P.pin));
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
