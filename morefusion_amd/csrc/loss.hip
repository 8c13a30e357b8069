// Fused ADD / ADD-S pose loss (functions.average_distance) for gfx950, batched over objects.
//
// Reference: morefusion/functions/loss/average_distance.py:64-85 -- transform_points twice
// (materialises [P, M, 3]), for symmetric objects geometry.nn over P*M queries (an R x Q distance
// matrix in the original, knn/nn.py:18-49), fancy-index gather, sub, square, sum, sqrt, mean;
// called once per object from a Python loop (contrib/singleview_3d/models/model.py:406-434).
//
// Here: one workgroup per (object b, predicted pose p).  The M model points are transformed by
// the true pose once per workgroup into LDS (the ADD-S search set), every lane owns model
// points m, m + 256, ...: predicted point, (ADD-S) running nearest true point with the
// reference's tie rule (lowest index among equal squared distances, knn/nn.py:48 argmin),
// distance, fixed-order block sum.  Backward = the same walk with the 3 x 4 moments of the
// unit residuals (gradient to the predicted transforms only; the true pose is data).
// Nothing of size P*M ever reaches HBM except the optional arg-min indices kept for backward.
#include "mf_common.h"

namespace {

constexpr int kLossThreads = 256;
constexpr int kLossTile = 1024;  // true points staged per LDS tile

struct Pose {
  float r[12];  // rows 0..2 of a row-major 4x4: R (3x3) | t
};

__device__ __forceinline__ Pose load_pose(const float *T) {
  Pose p;
#pragma unroll
  for (int i = 0; i < 12; ++i) p.r[i] = T[i];
  return p;
}

// transform_points.py:18-24: ((R0 x + R1 y) + R2 z) + t, un-fused
__device__ __forceinline__ float3 apply(const Pose &p, float x, float y, float z) {
  float3 o;
  o.x = ((p.r[0] * x + p.r[1] * y) + p.r[2] * z) + p.r[3];
  o.y = ((p.r[4] * x + p.r[5] * y) + p.r[6] * z) + p.r[7];
  o.z = ((p.r[8] * x + p.r[9] * y) + p.r[10] * z) + p.r[11];
  return o;
}

// residual of model point m under predicted pose `pp`: true' - pred, with true' the point's own
// true image (ADD) or the nearest true image of ANY model point (ADD-S).
template <bool BWD>
__device__ __forceinline__ void loss_walk(const float *__restrict__ points, const float *__restrict__ T_true,
                                          const float *__restrict__ T_pred, const uint8_t *__restrict__ symmetric,
                                          int M, int P, const float *__restrict__ gout,
                                          float *__restrict__ out, int32_t *__restrict__ nn_idx,
                                          float *__restrict__ gT) {
  __shared__ float4 s_true[kLossTile];
  __shared__ float s_red[kLossThreads / 64][12];
  const int p = blockIdx.x, b = blockIdx.y;
  const float *pts = points + (int64_t)b * M * 3;
  const Pose pt = load_pose(T_true + (int64_t)b * 16);
  const Pose pp = load_pose(T_pred + ((int64_t)b * P + p) * 16);
  const bool sym = symmetric != nullptr && symmetric[b] != 0;
  int32_t *idx_row = nn_idx ? nn_idx + ((int64_t)b * P + p) * M : nullptr;
  float acc[BWD ? 12 : 1];
#pragma unroll
  for (int i = 0; i < (BWD ? 12 : 1); ++i) acc[i] = 0.0f;
  const float g = BWD ? gout[(int64_t)b * P + p] / (float)M : 0.0f;

  for (int m0 = 0; m0 < M; m0 += kLossThreads) {  // every lane walks the same number of rounds
    const int m = m0 + (int)threadIdx.x;
    const bool live = m < M;
    float x = 0, y = 0, z = 0;
    if (live) { x = pts[3 * m]; y = pts[3 * m + 1]; z = pts[3 * m + 2]; }
    const float3 q = apply(pp, x, y, z);
    float3 tr = apply(pt, x, y, z);
    if (sym) {
      int bi = 0;
      if (BWD && idx_row) {  // indices saved by the forward pass
        if (live) {
          bi = idx_row[m];
          tr = apply(pt, pts[3 * bi], pts[3 * bi + 1], pts[3 * bi + 2]);
        }
      } else {
        float best = INFINITY;
        float3 bt = tr;
        for (int base = 0; base < M; base += kLossTile) {
          const int nt = min(kLossTile, M - base);
          __syncthreads();
          for (int i = threadIdx.x; i < nt; i += kLossThreads) {
            const int k = base + i;
            const float3 t3 = apply(pt, pts[3 * k], pts[3 * k + 1], pts[3 * k + 2]);
            s_true[i] = make_float4(t3.x, t3.y, t3.z, 0.0f);
          }
          __syncthreads();
          if (live) {
#pragma unroll 4
            for (int i = 0; i < nt; ++i) {
              const float4 r = s_true[i];
              const float dx = r.x - q.x, dy = r.y - q.y, dz = r.z - q.z;
              const float ssd = (dx * dx + dy * dy) + dz * dz;  // cuComputeDistanceGlobal.cu:64-67
              if (ssd < best) { best = ssd; bi = base + i; bt = make_float3(r.x, r.y, r.z); }
            }
          }
        }
        tr = bt;
        if (!BWD && idx_row && live) idx_row[m] = bi;
      }
    }
    if (!live) continue;
    const float dx = tr.x - q.x, dy = tr.y - q.y, dz = tr.z - q.z;
    const float n = sqrtf((dx * dx + dy * dy) + dz * dz);
    if (!BWD) {
      acc[0] += n;
    } else if (n > 0.0f) {
      // d n / d pred = -(true' - pred) / n ; pred_i = R_i . x + t_i
      const float u[3] = {-g * (dx / n), -g * (dy / n), -g * (dz / n)};
#pragma unroll
      for (int i = 0; i < 3; ++i) {
        acc[4 * i + 0] += u[i] * x;
        acc[4 * i + 1] += u[i] * y;
        acc[4 * i + 2] += u[i] * z;
        acc[4 * i + 3] += u[i];
      }
    }
  }
  // fixed-order block sums: DPP wave sums, then the 4 waves in order
  constexpr int kN = BWD ? 12 : 1;
#pragma unroll
  for (int i = 0; i < kN; ++i) {
    const float w = mf::wave_sum(acc[i]);
    if ((threadIdx.x & 63) == 0) s_red[threadIdx.x >> 6][i] = w;
  }
  __syncthreads();
  if (threadIdx.x < kN) {
    const int i = threadIdx.x;
    const float s = ((s_red[0][i] + s_red[1][i]) + s_red[2][i]) + s_red[3][i];
    if (!BWD) {
      out[(int64_t)b * P + p] = s / (float)M;
    } else {
      gT[((int64_t)b * P + p) * 16 + i] = s;
    }
  }
  if (BWD && threadIdx.x >= 12 && threadIdx.x < 16) gT[((int64_t)b * P + p) * 16 + threadIdx.x] = 0.0f;
}

__global__ __launch_bounds__(kLossThreads) void k_add_fwd(const float *points, const float *T_true,
                                                          const float *T_pred, const uint8_t *symmetric,
                                                          int M, int P, float *out, int32_t *nn_idx) {
  loss_walk<false>(points, T_true, T_pred, symmetric, M, P, nullptr, out, nn_idx, nullptr);
}

__global__ __launch_bounds__(kLossThreads) void k_add_bwd(const float *points, const float *T_true,
                                                          const float *T_pred, const uint8_t *symmetric,
                                                          int M, int P, const float *gout,
                                                          int32_t *nn_idx, float *gT) {
  loss_walk<true>(points, T_true, T_pred, symmetric, M, P, gout, nullptr, nn_idx, gT);
}

}  // namespace

extern "C" int mf_average_distance_fwd(const float *points, const float *T_true, const float *T_pred,
                                       const uint8_t *symmetric, int32_t B, int32_t M, int32_t P,
                                       float *out, int32_t *nn_idx, mfStream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (B <= 0 || P <= 0) return 0;
  if (M <= 0) {
    mf::set_last_error(hipErrorInvalidValue, "mf_average_distance_fwd: no model points");
    return -(int)hipErrorInvalidValue;
  }
  hipLaunchKernelGGL(k_add_fwd, dim3(P, B), dim3(kLossThreads), 0, stream, points, T_true, T_pred,
                     symmetric, M, P, out, nn_idx);
  return mf::check_launch("mf_average_distance_fwd");
}

extern "C" int mf_average_distance_bwd(const float *points, const float *T_true, const float *T_pred,
                                       const uint8_t *symmetric, const float *gout, int32_t B,
                                       int32_t M, int32_t P, const int32_t *nn_idx, float *gT_pred,
                                       mfStream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (B <= 0 || P <= 0) return 0;
  if (M <= 0) {
    mf::set_last_error(hipErrorInvalidValue, "mf_average_distance_bwd: no model points");
    return -(int)hipErrorInvalidValue;
  }
  hipLaunchKernelGGL(k_add_bwd, dim3(P, B), dim3(kLossThreads), 0, stream, points, T_true, T_pred,
                     symmetric, M, P, gout, const_cast<int32_t *>(nn_idx), gT_pred);
  return mf::check_launch("mf_average_distance_bwd");
}
