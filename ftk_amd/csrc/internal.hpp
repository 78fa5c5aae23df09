// Shared between the translation units of libftkx.so; not part of the ABI.
#ifndef FTKX_INTERNAL_HPP
#define FTKX_INTERNAL_HPP
namespace ftkx {
// message returned by ftkx_last_error(NULL, ...) on this thread (entry points that have no context)
void set_global_error(const char *msg);
}
#include <vector>
#include "../../include/ftkx.h"
namespace ftkx {
// pass 2 with the neighbour search and the component labelling done on the device (trace.cpp <-> trace_device.hip)
int trace_candidates(int nd, std::vector<int> &cand_off, std::vector<int> &cand_flat);
int trace_curves_with(int nd, const long long *dst, const long long *dsz, const unsigned long long *tags, size_t n, ftkx_curves *out,
                      const int *nbr, const unsigned char *deg, const int *root, int maxnb);
int trace_curves_tags(int nd, const long long *dst, const long long *dsz, const unsigned long long *tags, size_t n, ftkx_curves *out);   // host only
}
#endif
