// Host memory -> HBM for the slices a caller hands over as HOST arrays (ftkx_push_scalar_slice / ftkx_push_slice with on_device = 0).
//
// Reference boundary: the reference hands host arrays over on every call (critical_point_tracker_2d_regular.hh:369-384; its CUDA back-end
// cudaMemcpy's them from pageable memory, src/filters/critical_point_tracer_2d_regular.cu:194-232), and the patched tracker's resident mode
// (patches/ftk-xl-hip.patch, critical_point_tracker_regular.hh: hip_push_snapshot) pushes ONE fresh ndarray<double> per timestep: at 256^3
// that upload (134 MB) IS the step -- the sweep behind it takes 0.17 ms.
//
// What was measured (tools/probe/h2d_probe.hip, profiles/r06_h2d_probe.txt, NOTES.md round 6):
//  * the runtime's own copy from pageable memory is fast when the array's pages are known to it (the SAME array again and again: 2.4 ms for
//    128 MiB, 55 GB/s -- 96 % of a pinned source) and anything between 2.6 and 6 ms for an array that was malloc'ed for this timestep (its
//    32 768 pages are pinned one by one in front of the DMA); which of the two a box delivers changed from box to box and from call to call
//    (the patched tracker's push: 2.65 ms on one box, 4.2 on the next, alternating 2.7 / 4.5 within one series on a third);
//  * staging does not depend on that: host threads copy pieces into pinned buffers of their own and each piece goes to the device by DMA as
//    soon as it is there -- 2.7-2.9 ms, fresh array or not.  One thread filling one ring: 27 GB/s, a memcpy's rate; four: the DMA is the limit
//    (their memcpy's are done 1 ms before the last DMA lands).  Pieces of 1 MiB and less were slower than the runtime (small copies take
//    another path in it).  Threads that are kept save 0.04-0.15 ms against threads started per array: started per array, then -- nothing of the
//    library outlives a call;
//  * a DMA stream per thread (four of them) is 0.05-0.1 ms faster than two streams shared by the four threads, but one or two uploads among a
//    process's first ten then take 9-10 ms: a hipMemcpyAsync blocks for 6.5-8 ms where the runtime brings up another DMA queue the first time
//    it finds the ones it has busy (more of them with six streams; a warm-up of concurrent copies moved some of them into the first call,
//    not all).  Two streams: none in 60 uploads.  One stream: 3.3 ms, the gaps between its copies show.
//  * 1 GiB (a 512^3 snapshot): 8 MiB pieces 19.5-20.0 ms = 0.95-0.97 of a pinned source, fresh array or not; 4 MiB pieces 22.1, 16 MiB 20.9;
//    the runtime's copy 19.2 for an array it knows, 21.1 (calls of 25-28) for a fresh one.
// The staged path is taken for pageable sources of kStagedMinBytes and more; pinned or registered sources and small arrays go through the
// runtime's copy.  FTKX_UPLOAD_THREADS: the number of copy threads (default 4, the caller's thread is one of them); 0 or 1: the runtime's
// copy always.
#include "ctx.hpp"

#include <sched.h>

#include <atomic>
#include <mutex>
#include <thread>

using namespace ftkxh;

namespace {

constexpr size_t kPieceSmall = 4u << 20;       // one DMA ...
constexpr size_t kPieceBig = 8u << 20;         // ... of arrays of kBigBytes and more: 1 GiB in 19.5 ms (55 GB/s, 0.97 of a pinned source) against 22.1 with
constexpr size_t kBigBytes = 256u << 20;       // 4 MiB pieces (~10 us are lost per DMA); 128 MiB in 2.85 ms with 4 MiB pieces against 2.98 with 8 (the first piece's memcpy)
constexpr int kRing = 4;                       // pinned pieces per thread
constexpr int kMaxLanes = 8;
constexpr int kDmaStreams = 2;                 // lane l queues its DMAs on stream l % kDmaStreams
constexpr size_t kStagedMinBytes = 32u << 20;

struct Lane { char *pin = nullptr; size_t pin_bytes = 0; hipEvent_t ev[kRing] = {}; };
struct Engine { Lane lane[kMaxLanes]; int lanes = 0; hipStream_t dma[kDmaStreams] = {}; hipEvent_t gate = nullptr; };

std::mutex g_mutex;                            // one staged upload at a time per process: the pinned rings are the process's
std::map<int, Engine> g_engine;                // per device; kept for the process (64 MiB of pinned memory at four lanes, 128 once a big array has come)

// (default: four, or as many CPUs as the process may run on if those are fewer -- copy threads that share a CPU are slower than the runtime's copy)
int upload_threads()
{
  const char *e = getenv("FTKX_UPLOAD_THREADS");
  int t = 4;
  if (e) t = atoi(e);
  else {
    cpu_set_t set;
    CPU_ZERO(&set);
    if (sched_getaffinity(0, sizeof(set), &set) == 0) t = std::min(t, CPU_COUNT(&set));
  }
  return t < 0 ? 0 : (t > kMaxLanes ? kMaxLanes : t);
}

// a source the runtime can DMA from as it is (hipHostMalloc, hipHostRegister, managed memory) does not need the detour
bool source_is_pageable(const void *src)
{
  hipPointerAttribute_t a;
  memset(&a, 0, sizeof(a));
  const hipError_t e = hipPointerGetAttributes(&a, src);
  if (e != hipSuccess) { (void)hipGetLastError(); return true; }       // (older runtimes: "invalid value" for memory they have never seen)
  return a.type == hipMemoryTypeUnregistered;
}

// (what the staging needs and cannot get -- pinned memory is a limited resource of the process -- is not an error of the push: +1, and the
// array goes up by the runtime's copy)
int ensure_lanes(Engine &E, int want, size_t piece)
{
#define LANES_TRY(call) do { if ((call) != hipSuccess) { (void)hipGetLastError(); return 1; } } while (0)
  if (!E.gate) LANES_TRY(hipEventCreateWithFlags(&E.gate, hipEventDisableTiming));
  for (hipStream_t &s : E.dma) if (!s) LANES_TRY(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  if (E.lanes < want) E.lanes = want;
  for (int l = 0; l < want; l ++) {
    Lane &L = E.lane[l];
    if (L.pin_bytes < piece * kRing) {           // (no upload is in flight: every one of them has waited for its DMAs)
      if (L.pin) { char *old = L.pin; L.pin = nullptr; L.pin_bytes = 0; LANES_TRY(hipHostFree(old)); }
      if (hipHostMalloc((void **)&L.pin, piece * kRing, hipHostMallocDefault) != hipSuccess) { L.pin = nullptr; (void)hipGetLastError(); return 1; }
      L.pin_bytes = piece * kRing;
    }
    for (hipEvent_t &e : L.ev) if (!e) LANES_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming));
  }
#undef LANES_TRY
  return FTKX_OK;
}

struct Job {
  int device; char *dst; const char *src; size_t bytes, piece, npieces;
  std::atomic<size_t> next{0};
  std::atomic<int> err{0};                     // the first hipError_t any lane met; the others stop at their next piece
  int line = 0;
};

#define LANE_TRY(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) { int z_ = 0; if (J.err.compare_exchange_strong(z_, (int)e_)) J.line = __LINE__; return; } } while (0)

// one lane: pieces as the counter hands them out, through this lane's ring, queued on `st`; returns when its DMAs have landed
void lane_work(Job &J, Lane &L, hipStream_t st)
{
  LANE_TRY(hipSetDevice(J.device));
  size_t mine = 0;
  for (;;) {
    if (J.err.load(std::memory_order_relaxed)) break;
    const size_t i = J.next.fetch_add(1);
    if (i >= J.npieces) break;
    const int slot = (int)(mine % kRing);
    if (mine >= (size_t)kRing) LANE_TRY(hipEventSynchronize(L.ev[slot]));      // (the DMA that last read this pinned piece)
    const size_t off = i * J.piece, len = std::min(J.piece, J.bytes - off);
    memcpy(L.pin + (size_t)slot * J.piece, J.src + off, len);
    LANE_TRY(hipMemcpyAsync(J.dst + off, L.pin + (size_t)slot * J.piece, len, hipMemcpyHostToDevice, st));
    LANE_TRY(hipEventRecord(L.ev[slot], st));
    mine ++;
  }
  for (size_t r = 0; r < std::min<size_t>(mine, (size_t)kRing); r ++) LANE_TRY(hipEventSynchronize(L.ev[r]));     // (this lane's own pieces, not the stream's)
}

// FTKX_OK, an error, or +1: not taken (the caller falls back to the runtime's copy)
int staged_upload(ftkx_ctx *c, void *dst, const void *src, size_t bytes, int threads)
{
  std::unique_lock<std::mutex> lock(g_mutex, std::try_to_lock);
  if (!lock.owns_lock()) return 1;             // (another context of the process is uploading: the rings are taken)
  Engine &E = g_engine[c->device];
  const size_t piece = bytes >= kBigBytes ? kPieceBig : kPieceSmall;
  if (ensure_lanes(E, threads, piece)) return 1;
  // behind whatever the context's stream still does with the destination (an array recycled from a dropped slice: free_slice has made the
  // stream wait for the passes that read it)
  // (nothing to be ordered behind when that stream is idle -- the tracker's push after a collected step --: 35 us)
  if (hipStreamQuery(c->stream) != hipSuccess) {
    (void)hipGetLastError();
    HIP_TRY(c, hipEventRecord(E.gate, c->stream));
    for (hipStream_t s : E.dma) HIP_TRY(c, hipStreamWaitEvent(s, E.gate, 0));
  }
  Job J;
  J.device = c->device; J.dst = (char *)dst; J.src = (const char *)src; J.bytes = bytes; J.piece = piece; J.npieces = (bytes + piece - 1) / piece;
  // Starting a thread costs 40 us in a process that has the runtime loaded: the caller starts ONE and goes to work, that one starts the others
  // before its own lane (the first DMA is on its way 0.1 ms earlier).  th[0] is joined first: what it wrote into th[1..] is visible then.
  std::thread th[kMaxLanes];                   // (th[l - 1]: lane l)
  auto start = [&](int l) {                    // lane l on a thread of its own; false: no thread to be had (the lanes that run take all the pieces)
    try { th[(size_t)l - 1] = std::thread(lane_work, std::ref(J), std::ref(E.lane[l]), E.dma[l % kDmaStreams]); return true; }
    catch (...) { return false; }
  };
  // (the tracker's push, four runs each on one box: medians 2.70 / 2.71 / 2.79 against 2.78 / 2.80 / 2.87 / 2.88 ms with all three started by the caller)
  if (threads > 1) {
    try {
      th[0] = std::thread([&] {
        for (int l = 2; l < threads; l ++) if (!start(l)) break;
        lane_work(J, E.lane[1], E.dma[1 % kDmaStreams]);
      });
    } catch (...) {}
  }
  lane_work(J, E.lane[0], E.dma[0]);
  for (std::thread &t : th) if (t.joinable()) t.join();         // (in order: th[0] first)
  if (const int e = J.err.load()) {
    for (hipStream_t s : E.dma) (void)hipStreamSynchronize(s);          // (nothing of this upload is in flight when the error is reported)
    return fail(c, e == (int)hipErrorOutOfMemory ? FTKX_E_NOMEM : FTKX_E_DEVICE, "upload: %s (%s:%d)", hipGetErrorString((hipError_t)e), __FILE__, J.line);
  }
  return FTKX_OK;
}

}  // namespace

namespace ftkxh {

int upload_from_host(ftkx_ctx *c, void *dst, const void *src, size_t bytes)
{
  const int threads = upload_threads();
  if (threads >= 2 && bytes >= kStagedMinBytes && source_is_pageable(src)) {
    const int rc = staged_upload(c, dst, src, bytes, threads);
    if (rc <= 0) { c->uploads_staged += rc == FTKX_OK ? 1 : 0; return rc; }
  }
  HIP_TRY(c, hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, c->stream));
  HIP_TRY(c, hipStreamSynchronize(c->stream));            // (the caller may reuse its array on return)
  c->uploads_direct ++;
  return FTKX_OK;
}

}  // namespace ftkxh

extern "C" int ftkx_debug_upload_counts(const ftkx_ctx *c, unsigned long long *staged, unsigned long long *direct)
{
  if (!c) return FTKX_E_INVALID;
  if (staged) *staged = c->uploads_staged;
  if (direct) *direct = c->uploads_direct;
  return FTKX_OK;
}
