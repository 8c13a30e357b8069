// Library-level entry points: version and last-error string.
#include <stdio.h>
#include <string.h>

#include "mf_common.h"

namespace {
thread_local char g_err[512] = "";
}

namespace mf {
void set_last_error(hipError_t e, const char *where) {
  snprintf(g_err, sizeof(g_err), "%s: %s (%d)", where, hipGetErrorString(e), (int)e);
}
}  // namespace mf

extern "C" int mf_version(void) { return 100; }  // 0.1.0

extern "C" const char *mf_last_error_string(void) { return g_err; }
