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

constexpr int kPosesPerWg = 1;

// true images of model points [base, base + nt) into the SoA tile, padded to a multiple of 4 with +inf (never the
// minimum); barriers on both sides
__device__ __forceinline__ void stage_true_images(const Pose &pt, const float *__restrict__ pts, int base, int nt,
                                                  float *s_tx, float *s_ty, float *s_tz) {
  const int nt4 = (nt + 3) & ~3;
  __syncthreads();
  for (int i = threadIdx.x; i < nt4; i += kLossThreads) {
    float3 t3 = make_float3(INFINITY, INFINITY, INFINITY);
    if (i < nt) {
      const int k = base + i;
      t3 = apply(pt, pts[3 * k], pts[3 * k + 1], pts[3 * k + 2]);
    }
    s_tx[i] = t3.x; s_ty[i] = t3.y; s_tz[i] = t3.z;
  }
  __syncthreads();
}

// residual of model point m under predicted pose `pp`: true' - pred, with true' the point's own
// true image (ADD) or the nearest true image of ANY model point (ADD-S).
template <bool BWD>
__device__ __forceinline__ void loss_walk(const float *__restrict__ points, const float *__restrict__ T_true,
                                          const float *__restrict__ T_pred, const uint8_t *__restrict__ symmetric,
                                          int M, int P, const float *__restrict__ gout,
                                          float *__restrict__ out, int32_t *__restrict__ nn_idx,
                                          float *__restrict__ gT) {
  __shared__ __attribute__((aligned(16))) float s_tx[kLossTile], s_ty[kLossTile], s_tz[kLossTile];  // true images, SoA
  __shared__ float s_red[kLossThreads / 64][12];
  const int b = blockIdx.y;
  const float *pts = points + (int64_t)b * M * 3;
  const Pose pt = load_pose(T_true + (int64_t)b * 16);
  const bool sym = symmetric != nullptr && symmetric[b] != 0;
  // The true images are the same for every predicted pose of the object: a workgroup takes kPosesPerWg poses and,
  // when the model fits one tile (M <= 1024: always, for the 500-point YCB clouds), stages the images once for all
  // of them.  kPosesPerWg = 1 is what the training step wants (measured, 16 objects x 1000 poses, the symmetric
  // objects' searches dominate: 129 us with one pose per workgroup, 161 us with two, 175 us with four -- fewer,
  // longer workgroups lose more latency hiding than the shared staging saves).
  const bool search = sym && !(BWD && nn_idx);
  const bool staged = search && M <= kLossTile;
  if (staged) stage_true_images(pt, pts, 0, M, s_tx, s_ty, s_tz);
  for (int p = blockIdx.x * kPosesPerWg; p < min(P, (int)(blockIdx.x + 1) * kPosesPerWg); ++p) {  // block-uniform
  const Pose pp = load_pose(T_pred + ((int64_t)b * P + p) * 16);
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
        // The search walks the true images FOUR at a time in packed float math (v_pk_add_f32 / v_pk_mul_f32: two
        // candidates per instruction, the same IEEE operations in the same order as the scalar form) and keeps only
        // the running minimum and the group it came from -- 5.3 VALU instructions per candidate instead of 13 (three
        // subtractions, five products / sums, a compare and five selects); the winner inside the group -- the FIRST
        // member that attains the minimum, the reference's tie rule -- is resolved once per tile.
        typedef float f2 __attribute__((ext_vector_type(2)));
        float best = INFINITY;
        float3 bt = tr;
        const f2 qx2 = {q.x, q.x}, qy2 = {q.y, q.y}, qz2 = {q.z, q.z};
        for (int base = 0; base < M; base += kLossTile) {
          const int nt = min(kLossTile, M - base), nt4 = (nt + 3) & ~3;
          if (!staged) stage_true_images(pt, pts, base, nt, s_tx, s_ty, s_tz);
          if (live) {
            float tb = best;
            int grp = -1;
#pragma unroll 2
            for (int i = 0; i < nt4; i += 4) {
              const float4 X = *reinterpret_cast<const float4 *>(&s_tx[i]);
              const float4 Y = *reinterpret_cast<const float4 *>(&s_ty[i]);
              const float4 Z = *reinterpret_cast<const float4 *>(&s_tz[i]);
              const f2 dxa = (f2){X.x, X.y} - qx2, dxb = (f2){X.z, X.w} - qx2;
              const f2 dya = (f2){Y.x, Y.y} - qy2, dyb = (f2){Y.z, Y.w} - qy2;
              const f2 dza = (f2){Z.x, Z.y} - qz2, dzb = (f2){Z.z, Z.w} - qz2;
              const f2 sa = (dxa * dxa + dya * dya) + dza * dza;  // cuComputeDistanceGlobal.cu:64-67
              const f2 sb = (dxb * dxb + dyb * dyb) + dzb * dzb;
              const float mn = fminf(fminf(sa[0], sa[1]), fminf(sb[0], sb[1]));
              if (mn < tb) { tb = mn; grp = i; }
            }
            if (grp >= 0) {
              best = tb;
#pragma unroll
              for (int j = 3; j >= 0; --j) {  // downwards: the lowest member that attains the minimum stays
                const float rx = s_tx[grp + j], ry = s_ty[grp + j], rz = s_tz[grp + j];
                const float dx = rx - q.x, dy = ry - q.y, dz = rz - q.z;
                const float ssd = (dx * dx + dy * dy) + dz * dz;
                if (ssd == best) { bi = base + grp + j; bt = make_float3(rx, ry, rz); }
              }
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
  __syncthreads();  // s_red is reused by the next pose
  }
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

// ---- the confidence terms of the pose loss and their reduction (round 5) --------------------------------------
// contrib/singleview_3d/models/model.py:417-434: per object the mean over the confident points (conf > 0) of
// add * conf - lambda * log(conf); the loss is the mean over objects.  ~28 torch launches forward and ~20 backward
// -> one each.  One workgroup: wave w owns objects w, w + 16, ...; per-lane partial sums in point order, a fixed
// butterfly over the wave, the objects added in index order -> the value does not depend on the launch.
constexpr int kConfThreads = 1024;

__global__ __launch_bounds__(kConfThreads) void k_conf_loss_fwd(const float *__restrict__ add,
                                                                const float *__restrict__ conf, int B, int P,
                                                                float lambda, float *__restrict__ loss,
                                                                int32_t *__restrict__ cnt) {
  MF_DYN_LDS(float, s_obj);  // [B]
  const int w = threadIdx.x >> 6, l = threadIdx.x & 63;
  for (int b = w; b < B; b += kConfThreads / 64) {
    float acc = 0.0f;
    int n = 0;
    for (int p = l; p < P; p += 64) {
      const float c = conf[(int64_t)b * P + p];
      if (c > 0.0f) {
        acc += add[(int64_t)b * P + p] * c - lambda * logf(c);
        ++n;
      }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      acc += __shfl_xor(acc, o, 64);
      n += __shfl_xor(n, o, 64);
    }
    if (l == 0) {
      s_obj[b] = acc / (float)n;  // no confident point: 0 / 0 = NaN, like .mean() of nothing
      cnt[b] = n;
    }
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    float t = 0.0f;
    for (int b = 0; b < B; ++b) t += s_obj[b];
    loss[0] = t / (float)B;
  }
}

__global__ __launch_bounds__(256) void k_conf_loss_bwd(const float *__restrict__ add, const float *__restrict__ conf,
                                                       const int32_t *__restrict__ cnt, const float *__restrict__ gloss,
                                                       int B, int P, float lambda, float *__restrict__ dadd,
                                                       float *__restrict__ dconf) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (int64_t)B * P) return;
  const int b = (int)(i / P);
  const float c = conf[i];
  float ga = 0.0f, gc = 0.0f;
  if (c > 0.0f) {
    const float g = gloss[0] / (float)B / (float)cnt[b];
    ga = g * c;
    gc = g * (add[i] - lambda / c);
  }
  dadd[i] = ga;
  dconf[i] = gc;
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
  hipLaunchKernelGGL(k_add_fwd, dim3((P + kPosesPerWg - 1) / kPosesPerWg, B), dim3(kLossThreads), 0, stream, points, T_true, T_pred,
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
  hipLaunchKernelGGL(k_add_bwd, dim3((P + kPosesPerWg - 1) / kPosesPerWg, B), dim3(kLossThreads), 0, stream, points, T_true, T_pred,
                     symmetric, M, P, gout, const_cast<int32_t *>(nn_idx), gT_pred);
  return mf::check_launch("mf_average_distance_bwd");
}

/* The confidence terms of the pose loss (contrib/singleview_3d/models/model.py:417-434): loss[0] = mean over objects
 * of the mean over the confident points (conf > 0) of add * conf - lambda * log(conf), add / conf [B, P]; cnt [B]
 * (confident points per object) is kept for the backward: dadd, dconf [B, P] from the scalar gradient gloss[0]. */
extern "C" int mf_confidence_loss_fwd(const float *add, const float *conf, int32_t B, int32_t P, float lambda,
                                      float *loss, int32_t *cnt, mfStream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (B <= 0 || P <= 0 || B > 8192) {
    mf::set_last_error(hipErrorInvalidValue, "mf_confidence_loss_fwd: 1 <= B <= 8192, P >= 1");
    return -(int)hipErrorInvalidValue;
  }
  hipLaunchKernelGGL(k_conf_loss_fwd, dim3(1), dim3(kConfThreads), (size_t)B * sizeof(float), stream, add, conf, B, P,
                     lambda, loss, cnt);
  return mf::check_launch("mf_confidence_loss_fwd");
}

extern "C" int mf_confidence_loss_bwd(const float *add, const float *conf, const int32_t *cnt, const float *gloss,
                                      int32_t B, int32_t P, float lambda, float *dadd, float *dconf,
                                      mfStream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (B <= 0 || P <= 0) return 0;
  const int64_t n = (int64_t)B * P;
  hipLaunchKernelGGL(k_conf_loss_bwd, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, add, conf, cnt, gloss, B, P,
                     lambda, dadd, dconf);
  return mf::check_launch("mf_confidence_loss_bwd");
}
