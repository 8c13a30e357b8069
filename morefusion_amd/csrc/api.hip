// Library-level entry points: version and last-error string.
#include <stdio.h>
#include <string.h>

#include <mutex>
#include <set>
#include <utility>

#include "mf_common.h"

namespace {
thread_local char g_err[512] = "";
std::mutex g_attr_mu;
std::set<std::pair<int, const void *>> g_attr_done;
}  // namespace

namespace mf {
void set_last_error(hipError_t e, const char *where) {
  snprintf(g_err, sizeof(g_err), "%s: %s (%d)", where, hipGetErrorString(e), (int)e);
}
int allow_big_lds(const void *kernel, int bytes) {
  int dev = 0;
  MF_TRY(hipGetDevice(&dev));
  std::lock_guard<std::mutex> lock(g_attr_mu);
  if (g_attr_done.count({dev, kernel})) return 0;
  MF_TRY(hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
  g_attr_done.insert({dev, kernel});
  return 0;
}
}  // namespace mf

extern "C" int mf_version(void) { return 200; }  // 0.2.0

extern "C" const char *mf_last_error_string(void) { return g_err; }
