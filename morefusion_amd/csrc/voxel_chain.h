// Per-voxel point chains of average_voxelization_3d, shared by the dense op (voxelize.hip) and
// the sparse conv3 front end that consumes the compact rows directly (sparseconv.hip).
#pragma once
#include "mf_common.h"

namespace mf {

__device__ __forceinline__ bool voxel_of(const float *__restrict__ points, int64_t i, float ox,
                                         float oy, float oz, float pitch, int X, int Y, int Z,
                                         int &v, bool &has_nan) {
  float x = points[3 * i], y = points[3 * i + 1], z = points[3 * i + 2];
  has_nan = (x != x) || (y != y) || (z != z);
  float rx = roundf(mf::voxel_coord(x, ox, pitch));
  float ry = roundf(mf::voxel_coord(y, oy, pitch));
  float rz = roundf(mf::voxel_coord(z, oz, pitch));
  bool ok = rx >= 0.0f && rx < (float)X && ry >= 0.0f && ry < (float)Y && rz >= 0.0f &&
            rz < (float)Z;
  v = ok ? ((int)rx * Y + (int)ry) * Z + (int)rz : -1;
  return ok;
}

// link pass, one thread per point: integer atomics only (count += 1, head = exch(point id))
// build a per-voxel chain.  counts zero / head -1 on entry.
__device__ __forceinline__ void chain_link(const float *__restrict__ points,
                                           const int32_t *__restrict__ batch_indices, int64_t i,
                                           int B, int X, int Y, int Z, float ox, float oy, float oz,
                                           float pitch, int32_t *__restrict__ counts,
                                           int32_t *__restrict__ head, int32_t *__restrict__ link,
                                           int32_t *__restrict__ nan_flag) {
  int v;
  bool has_nan;
  bool ok = voxel_of(points, i, ox, oy, oz, pitch, X, Y, Z, v, has_nan);
  if (has_nan && nan_flag) atomicOr(nan_flag, 1);
  int b = batch_indices[i];
  ok = ok && b >= 0 && b < B;
  int32_t l = -2;
  if (ok) {
    int64_t key = (int64_t)b * X * Y * Z + v;
    atomicAdd(&counts[key], 1);
    l = atomicExch(&head[key], (int32_t)i);
  }
  link[i] = l;
}

// One WAVE per point i; only the wave of a voxel's chain head (exactly one per occupied voxel)
// does work: lane 0 walks the chain once, the ids are rank-sorted by the lanes, then lanes run
// over channels: coalesced reads of the value rows, sum in increasing point index (== the CPU
// loop order: bit-equal to forward_cpu, run-to-run deterministic), divide, store(ch, mean).
// s_ids / s_sorted: 64 ints of LDS per wave.  Returns the voxel key (b*V + v) or -1.
template <class Store>
__device__ __forceinline__ int64_t chain_mean(const float *__restrict__ values,
                                              const float *__restrict__ points,
                                              const int32_t *__restrict__ batch_indices,
                                              const int32_t *__restrict__ counts,
                                              const int32_t *__restrict__ head,
                                              const int32_t *__restrict__ link, int64_t i, int C,
                                              int B, int X, int Y, int Z, float ox, float oy,
                                              float oz, float pitch, int *s_ids, int *s_sorted,
                                              int lane, Store &&store, int64_t ldv = 0) {
  if (ldv == 0) ldv = C;  // row pitch of ``values`` (floats); 0 = dense rows
  int v;
  bool has_nan;
  const bool ok = voxel_of(points, i, ox, oy, oz, pitch, X, Y, Z, v, has_nan);
  const int b = batch_indices[i];
  if (!(ok && b >= 0 && b < B)) return -1;
  const int64_t V = (int64_t)X * Y * Z;
  const int64_t key = (int64_t)b * V + v;
  if (head[key] != (int32_t)i) return -1;  // not this voxel's chain head (wave-uniform)
  const int cnt = counts[key];
  if (cnt <= 64) {
    if (lane == 0) {
      int m = (int)i;
      for (int k = 0; k < cnt; ++k) { s_ids[k] = m; m = link[m]; }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    if (lane < cnt) {
      const int mine = s_ids[lane];
      int rank = 0;
      for (int k = 0; k < cnt; ++k) rank += s_ids[k] < mine ? 1 : 0;
      s_sorted[rank] = mine;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    for (int ch = lane; ch < C; ch += 64) {
      float s = 0.0f;
      for (int k = 0; k < cnt; ++k) s += values[(int64_t)s_sorted[k] * ldv + ch];
      store(ch, s / (float)cnt);
    }
  } else {  // pathological pile-up in one voxel: repeated selection, still in index order
    for (int ch = lane; ch < C; ch += 64) {
      float s = 0.0f;
      int last = -1;
      for (int k = 0; k < cnt; ++k) {
        int best = 0x7fffffff;
        for (int m = (int)i; m >= 0; m = link[m])
          if (m > last && m < best) best = m;
        s += values[(int64_t)best * ldv + ch];
        last = best;
      }
      store(ch, s / (float)cnt);
    }
  }
  return key;
}

}  // namespace mf
