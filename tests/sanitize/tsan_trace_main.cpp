// ThreadSanitizer driver for pass 2 (tests/test_sanitizers.py): the worker pool, the parallel hash build, two callers at once.
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <thread>
#include "ftkx.h"
int main(int argc, char **argv) {
  FILE *f = fopen(argv[1], "rb"); fseek(f, 0, SEEK_END); long sz = ftell(f); fseek(f, 0, SEEK_SET);
  std::vector<ftkx_cp_t> recs(sz / sizeof(ftkx_cp_t)); if (fread(recs.data(), 1, sz, f) != (size_t)sz) return 2; fclose(f);
  // usage: tsan_trace <records.raw> <domain start x> <start y> <size x> <size y>   (2D)
  long long st[3] = {atoll(argv[2]), atoll(argv[3]), 0}, dsz[3] = {atoll(argv[4]), atoll(argv[5]), 1};
  for (int rep = 0; rep < 3; rep ++) {
    ftkx_curves c; int rc = ftkx_trace_curves(2, st, dsz, recs.data(), recs.size(), &c);
    printf("rc %d curves %zu points %zu\n", rc, c.n_curves, c.n_points); ftkx_free_curves(&c);
  }
  // two callers at once (the pool serialises jobs)
  std::thread a([&]{ ftkx_curves c; ftkx_trace_curves(2, st, dsz, recs.data(), recs.size(), &c); ftkx_free_curves(&c); });
  std::thread b([&]{ ftkx_curves c; ftkx_trace_curves(2, st, dsz, recs.data(), recs.size(), &c); ftkx_free_curves(&c); });
  a.join(); b.join();
  printf("tsan run complete\n");
  return 0;
}
