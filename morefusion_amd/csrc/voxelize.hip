// average_voxelization_3d / max_voxelization_3d for gfx950.
//
// Reference: morefusion/functions/geometry/average_voxelization_3d.py:42-118 (K1),
// :146-220 (K2); max_voxelization_3d.py:58-138 (K3), :140-185 (K4).
//
// MI355X design (not the reference's thread-per-(point,channel) float atomics into
// a zero-filled [B,C,X,Y,Z] tensor followed by nonzero/divide launches):
//   1. link pass    -- one thread per point: integer atomics only (count += 1,
//                      head = exch(point id)) build a per-voxel chain.  128 KB of
//                      int32 per 32^3 object instead of 2 x 18.9 MB of atomic targets.
//   2. dense part   -- the [B,C,X,Y,Z] result is zero-filled at memset speed
//                      (the compulsory C*V*4 bytes per object; ~97 % of it stays zero).
//   3. sparse part  -- one wave per occupied voxel (the wave of its chain head): walk
//                      the chain once, rank-sort the ids, lanes over channels read the
//                      value rows coalesced, sum in increasing point index (== the CPU
//                      loop order, so bit-equal to forward_cpu and run-to-run
//                      deterministic), divide, store.  No float atomics, no nonzero().
#include <algorithm>

#include "mf_common.h"
#include "voxel_chain.h"

namespace {

using mf::voxel_of;

__global__ __launch_bounds__(256) void k_avgvox_link(const float *__restrict__ points,
                                                     const int32_t *__restrict__ batch_indices,
                                                     int64_t n, int B, int X, int Y, int Z,
                                                     float ox, float oy, float oz, float pitch,
                                                     int32_t *__restrict__ counts,
                                                     int32_t *__restrict__ head,
                                                     int32_t *__restrict__ link,
                                                     int32_t *__restrict__ nan_flag) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  mf::chain_link(points, batch_indices, i, B, X, Y, Z, ox, oy, oz, pitch, counts, head, link, nan_flag);
}

// Sparse half of the forward: one WAVE per point; only the wave of a voxel's chain head
// (exactly one per occupied voxel) continues (mf::chain_mean, voxel_chain.h).  The dense
// [B,C,X,Y,Z] tensor was zero-filled beforehand at memset speed, so only ~3 % of its lines
// are touched twice.
__global__ __launch_bounds__(256) void k_avgvox_scatter(const float *__restrict__ values,
                                                        const float *__restrict__ points,
                                                        const int32_t *__restrict__ batch_indices,
                                                        const int32_t *__restrict__ counts,
                                                        const int32_t *__restrict__ head,
                                                        const int32_t *__restrict__ link, int64_t n,
                                                        int C, int B, int X, int Y, int Z, float ox,
                                                        float oy, float oz, float pitch,
                                                        float *__restrict__ matrix) {
  __shared__ int s_ids[4][64];
  __shared__ int s_sorted[4][64];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int64_t i = (int64_t)blockIdx.x * 4 + wave;
  if (i >= n) return;
  const int64_t V = (int64_t)X * Y * Z;
  // the destination depends on the voxel: recompute its key for the store
  int v;
  bool has_nan;
  voxel_of(points, i, ox, oy, oz, pitch, X, Y, Z, v, has_nan);
  const int b = batch_indices[i];
  float *out = matrix + (int64_t)b * C * V + v;
  mf::chain_mean(values, points, batch_indices, counts, head, link, i, C, B, X, Y, Z, ox, oy, oz, pitch,
                 s_ids[wave], s_sorted[wave], lane, [&](int ch, float mean) { out[(int64_t)ch * V] = mean; });
}

__global__ __launch_bounds__(256) void k_avgvox_bwd(const float *__restrict__ gmatrix,
                                                    const float *__restrict__ points,
                                                    const int32_t *__restrict__ batch_indices,
                                                    const int32_t *__restrict__ counts, int64_t n,
                                                    int C, int B, int X, int Y, int Z, float ox,
                                                    float oy, float oz, float pitch,
                                                    float *__restrict__ gvalues) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;  // over n*C
  if (i >= n * C) return;
  int64_t p = i / C;
  int c = (int)(i - p * C);
  int v;
  bool has_nan;
  bool ok = voxel_of(points, p, ox, oy, oz, pitch, X, Y, Z, v, has_nan);
  int b = batch_indices[p];
  float g = 0.0f;
  if (ok && b >= 0 && b < B) {
    int64_t V = (int64_t)X * Y * Z;
    g = gmatrix[((int64_t)b * C + c) * V + v] / (float)counts[(int64_t)b * V + v];
  }
  gvalues[i] = g;
}

// ---- max voxelization -----------------------------------------------------
__device__ __forceinline__ uint32_t orderable(float f) {
  uint32_t u = __float_as_uint(f);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}

__global__ __launch_bounds__(256) void k_maxvox_key(const float *__restrict__ points,
                                                    const int32_t *__restrict__ batch_indices,
                                                    const float *__restrict__ intensities,
                                                    int64_t n, int B, int X, int Y, int Z, float ox,
                                                    float oy, float oz, float pitch,
                                                    unsigned long long *__restrict__ key,
                                                    int32_t *__restrict__ nan_flag) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  int v;
  bool has_nan;
  bool ok = voxel_of(points, i, ox, oy, oz, pitch, X, Y, Z, v, has_nan);
  if (has_nan && nan_flag) atomicOr(nan_flag, 1);
  int b = batch_indices[i];
  if (!(ok && b >= 0 && b < B)) return;
  // max intensity first, then LOWEST point index: (orderable(intensity), ~i)
  unsigned long long k =
      ((unsigned long long)orderable(intensities[i]) << 32) | (0xffffffffu - (uint32_t)i);
  atomicMax(&key[(int64_t)b * X * Y * Z + v], k);
}

template <int VEC>
__global__ __launch_bounds__(256) void k_maxvox_write(const float *__restrict__ values,
                                                      const unsigned long long *__restrict__ key,
                                                      int C, int V, int cpw,
                                                      float *__restrict__ matrix,
                                                      int32_t *__restrict__ indices) {
  const int b = blockIdx.z;
  const int v0 = (blockIdx.x * 256 + threadIdx.x) * VEC;
  if (v0 >= V) return;
  const int c0 = blockIdx.y * cpw;
  const int c1 = min(C, c0 + cpw);
  const int64_t kb = (int64_t)b * V + v0;
  int idx[VEC];
#pragma unroll
  for (int j = 0; j < VEC; ++j) {
    unsigned long long k = key[kb + j];
    idx[j] = k ? (int)(0xffffffffu - (uint32_t)(k & 0xffffffffu)) : -1;
    if (blockIdx.y == 0) indices[kb + j] = idx[j];
  }
  float *out = matrix + ((int64_t)b * C + c0) * V + v0;
  for (int c = c0; c < c1; ++c, out += V) {
    float r[VEC];
#pragma unroll
    for (int j = 0; j < VEC; ++j) r[j] = idx[j] >= 0 ? values[(int64_t)idx[j] * C + c] : 0.0f;
    if (VEC == 4) {
      *reinterpret_cast<float4 *>(out) = make_float4(r[0], r[1 % VEC], r[2 % VEC], r[3 % VEC]);
    } else {
      out[0] = r[0];
    }
  }
}

// A point wins at most one voxel (its own), so the backward scatter needs no atomics.
__global__ __launch_bounds__(256) void k_maxvox_bwd(const float *__restrict__ gmatrix,
                                                    const int32_t *__restrict__ indices, int C,
                                                    int64_t V, int64_t total,
                                                    float *__restrict__ gvalues) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;  // over B*C*V
  if (i >= total) return;
  int64_t v = i % V;
  int64_t bc = i / V;
  int c = (int)(bc % C);
  int64_t b = bc / C;
  int n = indices[b * V + v];
  if (n >= 0) gvalues[(int64_t)n * C + c] = gmatrix[i];
}


// ---- channels-LAST bf16 average voxelization (round 4: the conv3 input of the bf16 training path) ------------
// values rows [n][ldv] bf16 -> x [B][V][ldx] bf16, columns [0, C): the mean of the rows of a voxel's points (fp32
// sum in increasing point index, divided, rounded to bf16), zeros elsewhere.  A voxel's C channels are one
// contiguous row: the scatter is a coalesced row store per occupied voxel (the channels-first op writes C strided
// planes and its result has to be transposed and cast for the channels-last convolution: 302 MB + 151 MB at B = 16).
// Backward: every point reads its voxel's gradient row and divides by the voxel's count.
__global__ __launch_bounds__(256) void k_avgvox_cl_fwd(const uint16_t *__restrict__ values, int64_t ldv,
                                                       const float *__restrict__ points,
                                                       const int32_t *__restrict__ batch_indices,
                                                       const int32_t *__restrict__ counts,
                                                       const int32_t *__restrict__ head,
                                                       const int32_t *__restrict__ link, int64_t n, int C, int B, int D,
                                                       uint16_t *__restrict__ x, int64_t ldx,
                                                       const int32_t *__restrict__ rowmap) {
  __shared__ int s_ids[4][64];
  __shared__ int s_sorted[4][64];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int64_t i = (int64_t)blockIdx.x * 4 + wave;
  if (i >= n) return;
  int v;
  bool has_nan;
  const bool ok = mf::voxel_of(points, i, 0.0f, 0.0f, 0.0f, 1.0f, D, D, D, v, has_nan);
  const int b = batch_indices[i];
  if (!(ok && b >= 0 && b < B)) return;
  const int64_t key = (int64_t)b * D * D * D + v;
  if (head[key] != (int32_t)i) return;  // one wave per occupied voxel: its chain head's
  const int cnt = counts[key];
  // rowmap (the sparse conv3 of the training path, csrc/sparseconv_bf16.hip): the mean row goes to the voxel's
  // COMPACT row instead of its place in a dense grid
  const int64_t row = rowmap ? (int64_t)rowmap[key] : key;
  if (row < 0) return;
  uint16_t *dst = x + row * ldx;
  const float inv = (float)cnt;
  if (cnt <= 64) {
    if (lane == 0) {
      int m = (int)i;
      for (int k = 0; k < cnt; ++k) { s_ids[wave][k] = m; m = link[m]; }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    if (lane < cnt) {
      const int mine = s_ids[wave][lane];
      int rank = 0;
      for (int k = 0; k < cnt; ++k) rank += s_ids[wave][k] < mine ? 1 : 0;
      s_sorted[wave][rank] = mine;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    for (int c2 = lane; 2 * c2 < C; c2 += 64) {  // two channels (one dword) per lane
      float s0 = 0.0f, s1 = 0.0f;
      for (int k = 0; k < cnt; ++k) {
        const uint32_t w = *reinterpret_cast<const uint32_t *>(values + (int64_t)s_sorted[wave][k] * ldv + 2 * c2);
        s0 += mf::bf16_lo(w);
        s1 += mf::bf16_hi(w);
      }
      *reinterpret_cast<uint32_t *>(dst + 2 * c2) = mf::pack_bf16x2(s0 / inv, s1 / inv);
    }
  } else {  // pathological pile-up in one voxel: repeated selection, still in index order
    for (int c2 = lane; 2 * c2 < C; c2 += 64) {
      float s0 = 0.0f, s1 = 0.0f;
      int last = -1;
      for (int k = 0; k < cnt; ++k) {
        int best = 0x7fffffff;
        for (int m = (int)i; m >= 0; m = link[m])
          if (m > last && m < best) best = m;
        const uint32_t w = *reinterpret_cast<const uint32_t *>(values + (int64_t)best * ldv + 2 * c2);
        s0 += mf::bf16_lo(w);
        s1 += mf::bf16_hi(w);
        last = best;
      }
      *reinterpret_cast<uint32_t *>(dst + 2 * c2) = mf::pack_bf16x2(s0 / inv, s1 / inv);
    }
  }
}

__global__ __launch_bounds__(256) void k_avgvox_cl_bwd(const uint16_t *__restrict__ gx, int64_t ldx,
                                                       const float *__restrict__ points,
                                                       const int32_t *__restrict__ batch_indices,
                                                       const int32_t *__restrict__ counts, int64_t n, int C, int B,
                                                       int D, uint16_t *__restrict__ gvalues, int64_t ldg,
                                                       const int32_t *__restrict__ rowmap) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int64_t i = (int64_t)blockIdx.x * 4 + wave;
  if (i >= n) return;
  int v;
  bool has_nan;
  const bool ok = mf::voxel_of(points, i, 0.0f, 0.0f, 0.0f, 1.0f, D, D, D, v, has_nan);
  const int b = batch_indices[i];
  bool in = ok && b >= 0 && b < B;
  const int64_t key = in ? (int64_t)b * D * D * D + v : 0;
  const float cnt = in ? (float)counts[key] : 1.0f;
  const int64_t row = (in && rowmap) ? (int64_t)rowmap[key] : key;  // (gradient rows of the compact layout)
  in = in && row >= 0;
  for (int c2 = lane; 2 * c2 < C; c2 += 64) {
    uint32_t w = 0u;
    if (in) {
      const uint32_t g = *reinterpret_cast<const uint32_t *>(gx + row * ldx + 2 * c2);
      w = mf::pack_bf16x2(mf::bf16_lo(g) / cnt, mf::bf16_hi(g) / cnt);
    }
    *reinterpret_cast<uint32_t *>(gvalues + i * ldg + 2 * c2) = w;
  }
}


__global__ __launch_bounds__(256) void k_zero_cols_bf16(uint16_t *__restrict__ x, int64_t ldx, int64_t rows, int c2) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= rows * c2) return;
  const int64_t r = i / c2;
  *reinterpret_cast<uint32_t *>(x + r * ldx + 2 * (i - r * c2)) = 0u;
}

// channels per workgroup: enough workgroups to cover 256 CUs several times, but
// keep >= 2 channels so the per-thread count/head loads amortise.
int pick_cpw(int C, int B, int tiles) {
  int cpw = 16;
  while (cpw > 2 && (int64_t)tiles * B * ((C + cpw - 1) / cpw) < 2048) cpw >>= 1;
  return cpw;
}

}  // namespace

extern "C" int mf_average_voxelization_3d_fwd(const float *values, const float *points,
                                              const int32_t *batch_indices, int64_t n, int C,
                                              int B, int X, int Y, int Z, float ox, float oy,
                                              float oz, float pitch, float *matrix,
                                              int32_t *counts, int32_t *head, int32_t *link,
                                              int32_t *nan_flag, mfStream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  const int64_t V = (int64_t)X * Y * Z;
  if (B <= 0 || C <= 0 || V <= 0) return 0;
  if (int e_ = mf::fill_bytes(counts, 0, sizeof(int32_t) * B * V, stream)) return e_;
  if (int e_ = mf::fill_bytes(head, 0xff, sizeof(int32_t) * B * V, stream)) return e_;
  if (nan_flag) if (int e_ = mf::fill_bytes(nan_flag, 0, sizeof(int32_t), stream)) return e_;
  if (n > 0) {
    hipLaunchKernelGGL(k_avgvox_link, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream,
                       points, batch_indices, n, B, X, Y, Z, ox, oy, oz, pitch, counts, head, link,
                       nan_flag);
  }
  // dense part at memset speed, then the sparse part (one wave per occupied voxel)
  if (int e_ = mf::fill_bytes(matrix, 0, sizeof(float) * B * C * V, stream)) return e_;
  if (n > 0)
    hipLaunchKernelGGL(k_avgvox_scatter, dim3((unsigned)((n + 3) / 4)), dim3(256), 0, stream,
                       values, points, batch_indices, counts, head, link, n, C, B, X, Y, Z, ox,
                       oy, oz, pitch, matrix);
  return mf::check_launch("mf_average_voxelization_3d_fwd");
}

extern "C" int mf_average_voxelization_3d_bwd(const float *gmatrix, const float *points,
                                              const int32_t *batch_indices,
                                              const int32_t *counts, int64_t n, int C, int B,
                                              int X, int Y, int Z, float ox, float oy, float oz,
                                              float pitch, float *gvalues, mfStream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (n * C == 0) return 0;
  hipLaunchKernelGGL(k_avgvox_bwd, dim3((unsigned)((n * C + 255) / 256)), dim3(256), 0, stream,
                     gmatrix, points, batch_indices, counts, n, C, B, X, Y, Z, ox, oy, oz, pitch,
                     gvalues);
  return mf::check_launch("mf_average_voxelization_3d_bwd");
}


/* Channels-last bf16 average voxelization of the pose network's point features (origin 0, pitch 1, cubic grid of D):
 * values bf16 [n, ldv >= C], points fp32 [n,3], batch_indices [n] -> x bf16 [B, D^3, ldx >= C] columns [0, C)
 * (zero where no point falls; columns >= C are left untouched) + counts / head [B*D^3], link [n] (int32 scratch the
 * backward reuses).  C even. */
extern "C" int mf_average_voxelization_cl_bf16_fwd(const void *values, int64_t ldv, const float *points,
                                                   const int32_t *batch_indices, int64_t n, int32_t C, int32_t B,
                                                   int32_t D, void *x, int64_t ldx, int32_t *counts, int32_t *head,
                                                   int32_t *link, mfStream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  const int64_t V = (int64_t)D * D * D;
  if (B <= 0 || C <= 0 || V <= 0) return 0;
  if (C % 2 || ldv % 2 || ldx % 2 || ldx < C || ldv < C) {
    mf::set_last_error(hipErrorInvalidValue, "average_voxelization_cl_bf16: even C, ldv, ldx");
    return -(int)hipErrorInvalidValue;
  }
  if (int e_ = mf::fill_bytes(counts, 0, sizeof(int32_t) * B * V, stream)) return e_;
  if (int e_ = mf::fill_bytes(head, 0xff, sizeof(int32_t) * B * V, stream)) return e_;
  if (n > 0)
    hipLaunchKernelGGL(k_avgvox_link, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, points, batch_indices,
                       n, B, D, D, D, 0.0f, 0.0f, 0.0f, 1.0f, counts, head, link, (int32_t *)nullptr);
  if (ldx == C) {
    if (int e_ = mf::fill_bytes(x, 0, (int64_t)B * V * ldx * 2, stream)) return e_;
  } else {  // only the C columns: a strided zero fill
    hipLaunchKernelGGL(k_zero_cols_bf16, dim3((unsigned)(((int64_t)B * V * (C / 2) + 255) / 256)), dim3(256), 0, stream,
                       (uint16_t *)x, ldx, (int64_t)B * V, C / 2);
  }
  if (n > 0)
    hipLaunchKernelGGL(k_avgvox_cl_fwd, dim3((unsigned)((n + 3) / 4)), dim3(256), 0, stream, (const uint16_t *)values,
                       ldv, points, batch_indices, counts, head, link, n, C, B, D, (uint16_t *)x, ldx,
                       (const int32_t *)nullptr);
  return mf::check_launch("mf_average_voxelization_cl_bf16_fwd");
}

/* gvalues bf16 [n, ldg >= C] = gx[b, voxel(point), :C] / count(voxel)  (zero for points outside the grid) */
extern "C" int mf_average_voxelization_cl_bf16_bwd(const void *gx, int64_t ldx, const float *points,
                                                   const int32_t *batch_indices, const int32_t *counts, int64_t n,
                                                   int32_t C, int32_t B, int32_t D, void *gvalues, int64_t ldg,
                                                   mfStream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (n <= 0 || C <= 0) return 0;
  hipLaunchKernelGGL(k_avgvox_cl_bwd, dim3((unsigned)((n + 3) / 4)), dim3(256), 0, stream, (const uint16_t *)gx, ldx,
                     points, batch_indices, counts, n, C, B, D, (uint16_t *)gvalues, ldg, (const int32_t *)nullptr);
  return mf::check_launch("mf_average_voxelization_cl_bf16_bwd");
}

/* The same pair for COMPACT rows (csrc/sparseconv_bf16.hip: the sparse conv3 of the bf16 training path): the mean
 * row of an occupied voxel goes to A[rowmap[voxel]] (rows the map does not name are left as they are: the caller
 * zero-fills A), the backward reads the gradient rows through the same map.  counts / head / link / rowmap: the
 * tables of mf_sparse_conv3_bf16_index (mf_sparse_conv3_bf16_tables). */
extern "C" int mf_average_voxelization_rows_bf16_fwd(const void *values, int64_t ldv, const float *points,
                                                     const int32_t *batch_indices, int64_t n, int32_t C, int32_t B,
                                                     int32_t D, const int32_t *counts, const int32_t *head,
                                                     const int32_t *link, const int32_t *rowmap, void *A, int64_t lda,
                                                     mfStream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (n <= 0 || C <= 0) return 0;
  if (C % 2 || ldv % 2 || lda % 2 || lda < C || ldv < C || !rowmap) {
    mf::set_last_error(hipErrorInvalidValue, "average_voxelization_rows_bf16: even C, ldv, lda; a row map");
    return -(int)hipErrorInvalidValue;
  }
  hipLaunchKernelGGL(k_avgvox_cl_fwd, dim3((unsigned)((n + 3) / 4)), dim3(256), 0, stream, (const uint16_t *)values, ldv,
                     points, batch_indices, counts, head, link, n, C, B, D, (uint16_t *)A, lda, rowmap);
  return mf::check_launch("mf_average_voxelization_rows_bf16_fwd");
}

extern "C" int mf_average_voxelization_rows_bf16_bwd(const void *dA, int64_t lda, const float *points,
                                                     const int32_t *batch_indices, const int32_t *counts,
                                                     const int32_t *rowmap, int64_t n, int32_t C, int32_t B, int32_t D,
                                                     void *gvalues, int64_t ldg, mfStream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (n <= 0 || C <= 0) return 0;
  hipLaunchKernelGGL(k_avgvox_cl_bwd, dim3((unsigned)((n + 3) / 4)), dim3(256), 0, stream, (const uint16_t *)dA, lda, points,
                     batch_indices, counts, n, C, B, D, (uint16_t *)gvalues, ldg, rowmap);
  return mf::check_launch("mf_average_voxelization_rows_bf16_bwd");
}

extern "C" int mf_max_voxelization_3d_fwd(const float *values, const float *points,
                                          const int32_t *batch_indices, const float *intensities,
                                          int64_t n, int C, int B, int X, int Y, int Z, float ox,
                                          float oy, float oz, float pitch, float *matrix,
                                          int32_t *indices, uint64_t *key, int32_t *nan_flag,
                                          mfStream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  const int64_t V = (int64_t)X * Y * Z;
  if (B <= 0 || C <= 0 || V <= 0) return 0;
  if (int e_ = mf::fill_bytes(key, 0, sizeof(uint64_t) * B * V, stream)) return e_;
  if (nan_flag) if (int e_ = mf::fill_bytes(nan_flag, 0, sizeof(int32_t), stream)) return e_;
  if (n > 0)
    hipLaunchKernelGGL(k_maxvox_key, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream,
                       points, batch_indices, intensities, n, B, X, Y, Z, ox, oy, oz, pitch,
                       (unsigned long long *)key, nan_flag);
  const bool vec4 = (V % 4 == 0);
  const int per = 256 * (vec4 ? 4 : 1);
  const int tiles = (int)((V + per - 1) / per);
  const int cpw = pick_cpw(C, B, tiles);
  dim3 grid(tiles, (C + cpw - 1) / cpw, B);
  if (vec4)
    hipLaunchKernelGGL(k_maxvox_write<4>, grid, dim3(256), 0, stream, values,
                       (const unsigned long long *)key, C, (int)V, cpw, matrix, indices);
  else
    hipLaunchKernelGGL(k_maxvox_write<1>, grid, dim3(256), 0, stream, values,
                       (const unsigned long long *)key, C, (int)V, cpw, matrix, indices);
  return mf::check_launch("mf_max_voxelization_3d_fwd");
}

extern "C" int mf_max_voxelization_3d_bwd(const float *gmatrix, const int32_t *indices, int64_t n,
                                          int C, int B, int X, int Y, int Z, float *gvalues,
                                          mfStream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  const int64_t V = (int64_t)X * Y * Z;
  const int64_t total = (int64_t)B * C * V;
  if (total == 0) return 0;
  hipLaunchKernelGGL(k_maxvox_bwd, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream,
                     gmatrix, indices, C, V, total, gvalues);
  return mf::check_launch("mf_max_voxelization_3d_bwd");
}
