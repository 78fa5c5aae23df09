// The split pass's policy as one small state machine, free of the device: whether a pass's tail goes to a stream of its own (series.hip,
// path 5) is decided from what the plan knows -- the setting, the size of the mask launch, whether the data is sparse -- and, in the "auto"
// setting, from the context's own measurement.  series.hip feeds it (split_decide when a pass is planned, split_sample when one completes);
// tests/test_split_policy.py drives it without a GPU through tests/hostcheck.
//
// Does the split pass pay HERE?  How the hardware arbitrates between the context's queue and the tail's is not something the library sees
// (NOTES.md: the same binary runs 256^3 x 16 at 0.41 or at 0.51 ms per pass, against 0.44 in order, with the runtime's number of hardware
// queues).  So in "auto" the first passes of a shape that qualify are measured: five in order, five split -- the host's time between two
// completions while the pipeline is full -- and the split pass is taken for the passes of that shape unless it was clearly slower.
#pragma once
#include <algorithm>
#include <vector>

namespace ftkxh {

constexpr unsigned long long kSplitMinBytes = 1000000000ull;   // a sparse pass's tail hides behind a mask launch of 1 GB; a hit-dense one's behind 4 GB

struct split_cal {
  unsigned long long signature = 0;   // (steps, slices to mask, cells): what the samples are about
  int phase = 0;                      // 0: in-order samples, 1: split samples, 2: decided
  int skip = 0;                       // samples to discard (a change of form: buffers, streams, mask arrays of its first passes)
  std::vector<double> t_order, t_split;
  bool good = true;
  double median_order = 0, median_split = 0;   // seconds per pass, what the decision was taken on
  unsigned countdown = 0;             // decided "not here": passes until it is measured again
};

struct split_inputs {
  long mode;                          // FTKX_SERIES_HOOKS split: 0 never | 1 auto | 4 (= 3) on: the size rule alone | 2 whatever the size (tests)
  bool pipelined, dist, profiling_ok; // submitted with others in flight; a slab pass; no kernel events between the tail and anything
  bool sparse_now;                    // the last pass was sparse and its records do not go by way of the copy kernel
  unsigned long long ntodo, mask_bytes, signature;
};
struct split_verdict { bool split; int cal_kind; int forced; };      // cal_kind: 0 none, 1 an in-order sample, 2 a split sample; forced: 0 auto, 1 on, 2 off

inline split_verdict split_decide(split_cal &K, const split_inputs &in)
{
  split_verdict v{false, 0, in.mode == 0 ? 2 : in.mode != 1 ? 1 : 0};
  v.split = in.mode != 0 && in.pipelined && !in.dist && in.profiling_ok && in.ntodo > 0 &&
            (in.mask_bytes >= (in.sparse_now ? kSplitMinBytes : 4 * kSplitMinBytes) || in.mode == 2);
  if (!(v.split && in.mode == 1)) return v;
  if (K.signature != in.signature) { K = split_cal(); K.signature = in.signature; K.skip = 2; }
  if (K.phase == 0) { v.split = false; v.cal_kind = 1; }
  else if (K.phase == 1) v.cal_kind = 2;
  else if (!K.good) {
    v.split = false;
    if (K.countdown > 0 && -- K.countdown == 0) { const unsigned long long keep = K.signature; K = split_cal(); K.signature = keep; K.skip = 2; }
  }
  return v;
}

// a pass of a measuring phase has completed at host time `now`; the completion before it was at last_s (0: the pipeline ran empty since) and
// of kind last_kind; chained: another pass is open now (the pipeline is full)
inline void split_sample(split_cal &K, int cal_kind, double now, double last_s, int last_kind, bool chained)
{
  if (!cal_kind || K.phase >= 2) return;
  if (last_s > 0 && last_kind == cal_kind && chained) {
    if (K.skip > 0) K.skip --;
    else (cal_kind == 1 ? K.t_order : K.t_split).push_back(now - last_s);
  }
  auto median = [](std::vector<double> v) { std::sort(v.begin(), v.end()); return v[v.size() / 2]; };
  if (K.phase == 0 && K.t_order.size() >= 5) { K.phase = 1; K.skip = 4; }      // (the first split passes allocate: their stream, the mask arrays they swap in)
  else if (K.phase == 1 && K.t_split.size() >= 5) {
    K.phase = 2;
    K.median_order = median(K.t_order); K.median_split = median(K.t_split);
    K.good = K.median_split <= 1.02 * K.median_order;      // (kept unless clearly slower: the bad state is +13 %, a good one between -1 and -10 %)
    K.countdown = K.good ? 0u : 4096u;
  }
}

// 0 auto, still measuring; 1 auto, split; 2 auto, in order; 3 forced on; 4 forced off
inline int split_state(const split_cal &K, int forced) { return forced == 1 ? 3 : forced == 2 ? 4 : K.phase < 2 ? 0 : K.good ? 1 : 2; }

}  // namespace ftkxh
