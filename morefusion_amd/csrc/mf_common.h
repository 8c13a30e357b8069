// Shared device/host helpers for libmfhip (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/mfhip.h"

namespace mf {

void set_last_error(hipError_t e, const char *where);

inline int check_launch(const char *where) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    set_last_error(e, where);
    return -(int)e;
  }
  return 0;
}

#define MF_TRY(expr)                                  \
  do {                                                \
    hipError_t _e = (expr);                           \
    if (_e != hipSuccess) {                           \
      ::mf::set_last_error(_e, #expr);                \
      return -(int)_e;                                \
    }                                                 \
  } while (0)

// Opt a kernel in to more than 64 KB of dynamic LDS.  The attribute is per DEVICE (one process
// may drive several GPUs) and this may be called from any thread: api.hip keeps a mutex-guarded
// (device, function) set.
int allow_big_lds(const void *kernel, int bytes);

constexpr int kWave = 64;

// dynamic LDS of the workgroup (tests/host_emul/mf_common.h gives the host-emulation form)
#define MF_DYN_LDS(type, name) extern __shared__ __attribute__((aligned(16))) type name[]

__device__ __forceinline__ int cdiv(int a, int b) { return (a + b - 1) / b; }

// round((p - o) / pitch), CUDA round() == roundf(): half away from zero.
// Division stays a correctly-rounded IEEE divide (no reciprocal tricks): voxel
// indices must be bit-identical to the oracle.
__device__ __forceinline__ float voxel_coord(float p, float o, float pitch) {
  return (p - o) / pitch;
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
  return v;
}

__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v = fmaxf(v, __shfl_down(v, off, 64));
  return v;
}

}  // namespace mf
