// Shared between the translation units of libftkx.so; not part of the ABI.
#ifndef FTKX_INTERNAL_HPP
#define FTKX_INTERNAL_HPP
namespace ftkx {
// message returned by ftkx_last_error(NULL, ...) on this thread (entry points that have no context)
void set_global_error(const char *msg);
}
#include <cstdlib>
#include <cstring>
#include <vector>
#include "../../include/ftkx.h"
namespace ftkx {
// The library's test hooks come in two families, one environment variable each, "name=value,name=value" (DESIGN.md section 8):
// FTKX_SERIES_HOOKS (small, short, fold, split, one, rank_max: which forms of the device-driven pass are taken) and FTKX_MASK_PLAN (swizzle,
// yg, zchunk, lmin, lcap, order, rows, lean: launch geometry of the mask kernels).  Read at every use: tests switch them inside one process.
inline long env_hook(const char *var, const char *name, long dflt)
{
  const char *e = getenv(var);
  if (!e) return dflt;
  const size_t n = strlen(name);
  for (const char *p = e; *p;) {
    while (*p == ',' || *p == ' ') p ++;
    if (!strncmp(p, name, n) && p[n] == '=') return atol(p + n + 1);
    while (*p && *p != ',') p ++;
  }
  return dflt;
}
inline bool env_hook_set(const char *var, const char *name) { return env_hook(var, name, -0x7fffffffL) != -0x7fffffffL; }
}
namespace ftkx {
// pass 2 with the neighbour search and the component labelling done on the device (trace.cpp <-> trace_device.hip)
int trace_candidates(int nd, std::vector<int> &cand_off, std::vector<int> &cand_flat);
int trace_curves_with(int nd, const long long *dst, const long long *dsz, const unsigned long long *tags, size_t n, ftkx_curves *out,
                      const int *nbr, const unsigned char *deg, const int *root, int maxnb);
int trace_curves_tags(int nd, const long long *dst, const long long *dsz, const unsigned long long *tags, size_t n, ftkx_curves *out);   // host only
}
#endif
