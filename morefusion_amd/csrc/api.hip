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

namespace {
__global__ __launch_bounds__(256) void k_fill_words(uint32_t *__restrict__ p, uint32_t word, int64_t n) {
  // 16-byte stores where the address allows it; head / tail words singly
  const int64_t i = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 4;
  if (i + 4 <= n && (((uintptr_t)(p + i)) & 15) == 0) {
    *reinterpret_cast<uint4 *>(p + i) = make_uint4(word, word, word, word);
  } else {
    for (int64_t j = i; j < n && j < i + 4; ++j) p[j] = word;
  }
}
}  // namespace

namespace mf {
int fill_bytes(void *dst, int value, int64_t nbytes, hipStream_t stream) {
  if (nbytes <= 0) return 0;
  if ((nbytes & 3) || (((uintptr_t)dst) & 3)) {
    set_last_error(hipErrorInvalidValue, "fill_bytes: 4-byte aligned address and size");
    return -(int)hipErrorInvalidValue;
  }
  const int64_t n = nbytes / 4;
  const uint32_t word = (uint32_t)(value & 0xff) * 0x01010101u;
  hipLaunchKernelGGL(k_fill_words, dim3((unsigned)((n + 1023) / 1024)), dim3(256), 0, stream, (uint32_t *)dst, word, n);
  return check_launch("fill_bytes");
}
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
