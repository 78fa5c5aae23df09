// Shared between the translation units of libftkx.so; not part of the ABI.
#ifndef FTKX_INTERNAL_HPP
#define FTKX_INTERNAL_HPP
namespace ftkx {
// message returned by ftkx_last_error(NULL, ...) on this thread (entry points that have no context)
void set_global_error(const char *msg);
}
#endif
