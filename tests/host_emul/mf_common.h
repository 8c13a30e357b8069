/* TEST INFRASTRUCTURE ONLY -- shadows morefusion_amd/csrc/mf_common.h so that a barrier-free HIP
 * kernel source (csrc/preprocess.hip: k_pre_crops) can be compiled with g++ and executed thread by
 * thread on the host: tests/test_preprocess_oracle.py checks the very kernel text against the
 * oracle without a GPU.  Kernels that need __syncthreads()/LDS cooperation are NOT emulated
 * faithfully (threads run one after another) and are not called by the tests. */
#pragma once
#include <math.h>
#include <stdint.h>

#include <algorithm>
#include <cmath>

#include "mfhip.h"

typedef int hipError_t;
typedef void *hipStream_t;
enum { hipSuccess = 0, hipErrorInvalidValue = 1 };
struct dim3 {
  unsigned x, y, z;
  dim3(unsigned a = 1, unsigned b = 1, unsigned c = 1) : x(a), y(b), z(c) {}
};
static dim3 blockIdx, threadIdx, blockDim, gridDim;
#define __global__
#define __device__
#define __forceinline__ inline
#define __shared__ static
#define __launch_bounds__(...)
#define __restrict__
static inline void __syncthreads() {}
using std::isnan;
using std::max;
using std::min;
template <class T> T atomicMin(T *p, T v) { T o = *p; *p = std::min(o, v); return o; }
template <class T> T atomicMax(T *p, T v) { T o = *p; *p = std::max(o, v); return o; }
template <class T> T atomicAdd(T *p, T v) { T o = *p; *p = o + v; return o; }
namespace mf {
inline void set_last_error(int, const char *) {}
inline int check_launch(const char *) { return 0; }
}  // namespace mf
#define hipLaunchKernelGGL(k, g, b, sh, st, ...)                                   \
  do {                                                                             \
    dim3 _g = g, _b = b;                                                           \
    gridDim = _g;                                                                  \
    blockDim = _b;                                                                 \
    for (unsigned by = 0; by < _g.y; ++by)                                         \
      for (unsigned bx = 0; bx < _g.x; ++bx)                                       \
        for (unsigned tx = 0; tx < _b.x; ++tx) {                                   \
          blockIdx = dim3(bx, by);                                                 \
          threadIdx = dim3(tx);                                                    \
          k(__VA_ARGS__);                                                          \
        }                                                                          \
  } while (0)
