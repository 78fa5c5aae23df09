#include <string>
#include <cstring>
#include <cstdlib>
#include "ftkx.h"
namespace ftkx { static thread_local std::string g; void set_global_error(const char *m) { g = m ? m : ""; } }
extern "C" {
int ftkx_last_error(const ftkx_ctx *, char *buf, size_t n) { if (buf && n) { strncpy(buf, ftkx::g.c_str(), n - 1); buf[n - 1] = 0; } return (int)ftkx::g.size(); }
void ftkx_free(void *p) { free(p); }
}
