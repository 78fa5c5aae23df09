// std::sort on a few host threads: sorted runs, then pairwise merges (62 181 order keys: 4.5 ms on one thread).  Shared by the tracker
// (element order of the discrete points) and pass 2 (tag index of a record set that does not come sorted).
#pragma once
#include <algorithm>
#include <thread>
#include <vector>

namespace ftkx {
template <class T>
void sort_on_threads(std::vector<T> &v)
{
  const size_t n = v.size();
  const unsigned hw = std::max(1u, std::thread::hardware_concurrency());
  unsigned parts = n < 16384 ? 1u : std::min(8u, hw);
  while (parts & (parts - 1)) parts --;                    // a power of two
  if (parts <= 1) { std::sort(v.begin(), v.end()); return; }
  std::vector<size_t> cut(parts + 1);
  for (unsigned i = 0; i <= parts; i ++) cut[i] = n * i / parts;
  {
    std::vector<std::thread> th;
    for (unsigned i = 1; i < parts; i ++) th.emplace_back([&, i] { std::sort(v.begin() + (long)cut[i], v.begin() + (long)cut[i + 1]); });
    std::sort(v.begin(), v.begin() + (long)cut[1]);
    for (auto &t : th) t.join();
  }
  for (unsigned w = 1; w < parts; w *= 2) {
    std::vector<std::thread> th;
    for (unsigned i = 0; i + w < parts; i += 2 * w) {
      const size_t a = cut[i], m = cut[i + w], b = cut[std::min(parts, i + 2 * w)];
      if (i + 2 * w < parts) th.emplace_back([&, a, m, b] { std::inplace_merge(v.begin() + (long)a, v.begin() + (long)m, v.begin() + (long)b); });
      else std::inplace_merge(v.begin() + (long)a, v.begin() + (long)m, v.begin() + (long)b);
    }
    for (auto &t : th) t.join();
  }
}


// f(begin, end) over [0, n) on a few host threads (plain std::thread: for loops of ~10^5 elements that copy or convert)
template <class F>
void for_ranges_on_threads(size_t n, F f)
{
  const unsigned hw = std::max(1u, std::thread::hardware_concurrency());
  const unsigned parts = n < 16384 ? 1u : std::min(8u, hw);
  if (parts <= 1) { f((size_t)0, n); return; }
  std::vector<std::thread> th;
  for (unsigned i = 1; i < parts; i ++) th.emplace_back([&, i] { f(n * i / parts, n * (i + 1) / parts); });
  f((size_t)0, n / parts);
  for (auto &t : th) t.join();
}
}  // namespace ftkx
