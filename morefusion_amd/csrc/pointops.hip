// Point-wise prologue / epilogue of the pose network's volumetric part (inference), for gfx950.
//
// Reference: contrib/singleview_3d/models/model.py:232-275 -- the camera -> voxel frame change of the sampled
// points (:236), `to_center` (:101), and after the heads the class selection, the quaternion
// normalisation (`F.normalize`, chainer: x / (|x| + 1e-5)), `cls_trans * pitch + points` (:264-266) and
// the confidence sigmoid (:262).  In torch these are ~25 launches of a few microseconds; at batch 1
// (BASELINE config 2) that is ~5 % of the frame.  Two kernels, one lane per point:
//   k_point_prep     [B,3,P] camera points, [B,32,P] image features -> voxel-frame points [n,3],
//                    to_center [n,4] (4th column zero: the K-padded input of conv1_pcd), image features as
//                    rows [n,32] (the point MLP's GEMM input), batch indices [n]
//   k_pose_epilogue  heads' output rows [n, 3*np4] -> (quaternion [B,P,4], translation [B,P,3],
//                    confidence [B,P]) of each object's class
// Arithmetic is the torch expression order ((p - origin) / pitch with an IEEE divide; p * pitch + origin).
#include "mf_common.h"
#include "quat.h"

namespace {

__global__ __launch_bounds__(256) void k_point_prep(const float *__restrict__ points_cam,  // [B,3,P]
                                                    const float *__restrict__ values,      // [B,Cv,P]
                                                    const float *__restrict__ origin,      // [B,3]
                                                    const float *__restrict__ pitch,       // [B]
                                                    int B, int P, int Cv, float center,
                                                    float *__restrict__ pts,               // [n,3]
                                                    float *__restrict__ tc4,               // [n,4]
                                                    float *__restrict__ x_rows,            // [n,Cv]
                                                    int32_t *__restrict__ batch_indices) { // [n]
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (int64_t)B * P) return;
  const int b = (int)(i / P), p = (int)(i - (int64_t)b * P);
  const float pit = pitch[b];
  float v[3];
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    v[a] = (points_cam[((int64_t)b * 3 + a) * P + p] - origin[3 * b + a]) / pit;
    pts[3 * i + a] = v[a];
  }
  *reinterpret_cast<float4 *>(tc4 + 4 * i) = make_float4(center - v[0], center - v[1], center - v[2], 0.0f);
  batch_indices[i] = b;
  const float *src = values + (int64_t)b * Cv * P + p;
  float *dst = x_rows + i * Cv;
  for (int c = 0; c < Cv; c += 4) {  // Cv % 4 == 0
    *reinterpret_cast<float4 *>(dst + c) = make_float4(src[(int64_t)c * P], src[(int64_t)(c + 1) * P],
                                                       src[(int64_t)(c + 2) * P], src[(int64_t)(c + 3) * P]);
  }
}

__global__ __launch_bounds__(256) void k_pose_epilogue(const float *__restrict__ o, int64_t ldo, int np4,
                                                       const int64_t *__restrict__ class_id,  // [B], 1-based
                                                       const float *__restrict__ pts,          // [n,3] voxel frame
                                                       const float *__restrict__ origin, const float *__restrict__ pitch,
                                                       int B, int P, int n_fg, float *__restrict__ rot,  // [n,4]
                                                       float *__restrict__ trans,              // [n,3]
                                                       float *__restrict__ conf) {             // [n]
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (int64_t)B * P) return;
  const int b = (int)(i / P);
  const int fg = (int)class_id[b] - 1;
  const float pit = pitch[b];
  const float *row = o + i * ldo;
  if (fg < 0 || fg >= n_fg) {  // background (0) or an id beyond the heads' classes: no pose -- NaN, never a read
    const float nan = __uint_as_float(0x7fc00000u);  // outside the row (the torch indexing it replaces raises / wraps)
    *reinterpret_cast<float4 *>(rot + 4 * i) = make_float4(nan, nan, nan, nan);
    trans[3 * i] = trans[3 * i + 1] = trans[3 * i + 2] = nan;
    conf[i] = nan;
    return;
  }
  const float q0 = row[4 * fg], q1 = row[4 * fg + 1], q2 = row[4 * fg + 2], q3 = row[4 * fg + 3];
  const float nrm = sqrtf(((q0 * q0 + q1 * q1) + q2 * q2) + q3 * q3) + 1e-5f;
  *reinterpret_cast<float4 *>(rot + 4 * i) = make_float4(q0 / nrm, q1 / nrm, q2 / nrm, q3 / nrm);
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    const float pc = pts[3 * i + a] * pit + origin[3 * b + a];  // voxel -> camera frame (model.py:264)
    trans[3 * i + a] = pc + row[np4 + 3 * fg + a] * pit;
  }
  conf[i] = 1.0f / (1.0f + expf(-row[2 * np4 + fg]));
}

// ---- the pose epilogue of the TRAINING path, forward and backward (round 5) -------------------------------------
// model.py:262-273 on the three heads' separate outputs (fp32 rows [n, n_fg * 4 | n_fg * 3 | n_fg]): class selection,
// F.normalize (chainer: x / (|x| + 1e-5)), translation = voxel point * pitch + origin + raw * pitch, sigmoid.  The
// torch form is ~27 launches forward and ~47 backward -- three advanced-indexing backward passes, each an
// index_put(accumulate) with a radix sort.  One lane per point; the backward writes the three gradient rows
// completely (zeros outside the object's class).
__global__ __launch_bounds__(256) void k_pose_epi3_fwd(const float *__restrict__ orot, const float *__restrict__ otrn,
                                                       const float *__restrict__ ocnf, const int64_t *__restrict__ class_id,
                                                       const float *__restrict__ pts, const float *__restrict__ origin,
                                                       const float *__restrict__ pitch, int B, int P, int n_fg,
                                                       float *__restrict__ rot, float *__restrict__ trans,
                                                       float *__restrict__ conf) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (int64_t)B * P) return;
  const int b = (int)(i / P);
  const int fg = (int)class_id[b] - 1;
  const float pit = pitch[b];
  if (fg < 0 || fg >= n_fg) {  // no head for this id: NaN (k_pose_epilogue's rule)
    const float nan = __uint_as_float(0x7fc00000u);
    *reinterpret_cast<float4 *>(rot + 4 * i) = make_float4(nan, nan, nan, nan);
    trans[3 * i] = trans[3 * i + 1] = trans[3 * i + 2] = nan;
    conf[i] = nan;
    return;
  }
  const float *r = orot + i * (int64_t)(4 * n_fg) + 4 * fg;
  const float q0 = r[0], q1 = r[1], q2 = r[2], q3 = r[3];
  const float nrm = sqrtf(((q0 * q0 + q1 * q1) + q2 * q2) + q3 * q3) + 1e-5f;
  *reinterpret_cast<float4 *>(rot + 4 * i) = make_float4(q0 / nrm, q1 / nrm, q2 / nrm, q3 / nrm);
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    const float pc = pts[3 * i + a] * pit + origin[3 * b + a];
    trans[3 * i + a] = pc + otrn[i * (int64_t)(3 * n_fg) + 3 * fg + a] * pit;
  }
  conf[i] = 1.0f / (1.0f + expf(-ocnf[i * (int64_t)n_fg + fg]));
}

// One lane per ELEMENT of the concatenated gradient row [4 n_fg | 3 n_fg | n_fg] of a point (coalesced stores; the
// first version -- a lane per point writing its 168 floats -- took 17.8 us for 16 000 points): zero outside the
// object's class, the 8 lanes inside it recompute what they need of the point (4 + 1 floats).
__global__ __launch_bounds__(256) void k_pose_epi3_bwd(const float *__restrict__ orot, const float *__restrict__ ocnf,
                                                       const int64_t *__restrict__ class_id, const float *__restrict__ pitch,
                                                       const float *__restrict__ grot, const float *__restrict__ gtrans,
                                                       const float *__restrict__ gconf, int B, int P, int n_fg,
                                                       float *__restrict__ drot, float *__restrict__ dtrn,
                                                       float *__restrict__ dcnf) {
  const int W = 8 * n_fg;
  const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= (int64_t)B * P * W) return;
  const int64_t i = e / W;
  const int col = (int)(e - i * W);
  const int b = (int)(i / P);
  const int fg = (int)class_id[b] - 1;
  const bool live = fg >= 0 && fg < n_fg;
  if (col < 4 * n_fg) {
    float v = 0.0f;
    const int k = col - 4 * fg;
    if (live && k >= 0 && k < 4) {
      const float *r = orot + i * (int64_t)(4 * n_fg) + 4 * fg;
      const float q[4] = {r[0], r[1], r[2], r[3]};
      const float g[4] = {grot[4 * i], grot[4 * i + 1], grot[4 * i + 2], grot[4 * i + 3]};
      // y = x / (s + eps), s = |x|:  dx = g / (s + eps) - x (x . g) / (s (s + eps)^2)
      const float s = sqrtf(((q[0] * q[0] + q[1] * q[1]) + q[2] * q[2]) + q[3] * q[3]);
      const float d = s + 1e-5f;
      const float dot = ((q[0] * g[0] + q[1] * g[1]) + q[2] * g[2]) + q[3] * g[3];
      const float c2 = s > 0.0f ? dot / (s * d * d) : 0.0f;
      v = g[k] / d - q[k] * c2;
    }
    drot[i * (int64_t)(4 * n_fg) + col] = v;
  } else if (col < 7 * n_fg) {
    const int ct = col - 4 * n_fg, a = ct - 3 * fg;
    dtrn[i * (int64_t)(3 * n_fg) + ct] = (live && a >= 0 && a < 3) ? gtrans[3 * i + a] * pitch[b] : 0.0f;
  } else {
    const int cc = col - 7 * n_fg;
    float v = 0.0f;
    if (live && cc == fg) {
      const float c = 1.0f / (1.0f + expf(-ocnf[i * (int64_t)n_fg + fg]));
      v = gconf[i] * c * (1.0f - c);
    }
    dcnf[i * (int64_t)n_fg + cc] = v;
  }
}

// ---- transformation_matrix of a batch of poses, forward and backward (round 5: the training loss) ---------------
// functions/geometry/transformation_matrix.py:5-18 = quaternion_matrix.py:65-78 (wxyz, any norm: scaled by
// sqrt(2 / |q|^2)) + compose_transform.py:5-48.  The torch composite is ~25 launches forward and ~60 backward for the
// B * P = 16000 predicted poses of a training step; here one lane per pose (quat.h: the expressions of the
// refinement loops).  T [n][4][4] row-major; backward: gq, gt from gT (only its top three rows matter).
__global__ __launch_bounds__(256) void k_tfm_fwd(const float *__restrict__ q, const float *__restrict__ t, int64_t n,
                                                 float *__restrict__ T) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float4 qq = *reinterpret_cast<const float4 *>(q + 4 * i);
  const float qa[4] = {qq.x, qq.y, qq.z, qq.w};
  float R[9];
  mf::quat_to_R(qa, R);
  float4 *o = reinterpret_cast<float4 *>(T + 16 * i);
  o[0] = make_float4(R[0], R[1], R[2], t[3 * i]);
  o[1] = make_float4(R[3], R[4], R[5], t[3 * i + 1]);
  o[2] = make_float4(R[6], R[7], R[8], t[3 * i + 2]);
  o[3] = make_float4(0.0f, 0.0f, 0.0f, 1.0f);
}

__global__ __launch_bounds__(256) void k_tfm_bwd(const float *__restrict__ q, const float *__restrict__ gT, int64_t n,
                                                 float *__restrict__ gq, float *__restrict__ gt) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float4 qq = *reinterpret_cast<const float4 *>(q + 4 * i);
  const float qa[4] = {qq.x, qq.y, qq.z, qq.w};
  const float4 *g = reinterpret_cast<const float4 *>(gT + 16 * i);
  const float4 g0 = g[0], g1 = g[1], g2 = g[2];
  const float gR[9] = {g0.x, g0.y, g0.z, g1.x, g1.y, g1.z, g2.x, g2.y, g2.z};
  float go[4];
  mf::quat_backward(qa, gR, go);
  *reinterpret_cast<float4 *>(gq + 4 * i) = make_float4(go[0], go[1], go[2], go[3]);
  gt[3 * i] = g0.w;
  gt[3 * i + 1] = g1.w;
  gt[3 * i + 2] = g2.w;
}

}  // namespace

extern "C" int mf_point_prep(const float *points_cam, const float *values, const float *origin, const float *pitch,
                             int32_t B, int32_t P, int32_t Cv, float center, float *pts, float *tc4, float *x_rows,
                             int32_t *batch_indices, mfStream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (B <= 0 || P <= 0) return 0;
  if (Cv % 4 || (((uintptr_t)tc4 | (uintptr_t)x_rows) & 15)) {
    mf::set_last_error(hipErrorInvalidValue, "point_prep: need Cv % 4 == 0 and 16-byte aligned outputs");
    return -(int)hipErrorInvalidValue;
  }
  const int64_t n = (int64_t)B * P;
  hipLaunchKernelGGL(k_point_prep, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, points_cam, values, origin,
                     pitch, B, P, Cv, center, pts, tc4, x_rows, batch_indices);
  return mf::check_launch("mf_point_prep");
}

extern "C" int mf_pose_epilogue(const float *heads_out, int64_t ldo, int32_t np4, const int64_t *class_id,
                                const float *pts, const float *origin, const float *pitch, int32_t B, int32_t P,
                                int32_t n_fg, float *rot, float *trans, float *conf, mfStream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (B <= 0 || P <= 0) return 0;
  if (((uintptr_t)rot & 15) || ldo < 3 * (int64_t)np4 || n_fg < 1 || 4 * n_fg > np4) {
    mf::set_last_error(hipErrorInvalidValue, "pose_epilogue: need a 16-byte aligned rot, ldo >= 3 * np4, 4 * n_fg <= np4");
    return -(int)hipErrorInvalidValue;
  }
  const int64_t n = (int64_t)B * P;
  hipLaunchKernelGGL(k_pose_epilogue, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, heads_out, ldo, np4,
                     class_id, pts, origin, pitch, B, P, n_fg, rot, trans, conf);
  return mf::check_launch("mf_pose_epilogue");
}

/* T [n,4,4] = transformation_matrix(q [n,4] wxyz, t [n,3]) and its backward (gq [n,4], gt [n,3] from gT [n,4,4]):
 * functions/geometry/transformation_matrix.py:5-18 (quaternion_matrix.py:36-78 + compose_transform.py), one launch
 * each.  q, T, gT, gq 16-byte aligned. */
extern "C" int mf_transformation_matrix_fwd(const float *q, const float *t, int64_t n, float *T, mfStream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (n <= 0) return 0;
  hipLaunchKernelGGL(k_tfm_fwd, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, q, t, n, T);
  return mf::check_launch("mf_transformation_matrix_fwd");
}

extern "C" int mf_transformation_matrix_bwd(const float *q, const float *gT, int64_t n, float *gq, float *gt,
                                            mfStream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (n <= 0) return 0;
  hipLaunchKernelGGL(k_tfm_bwd, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, q, gT, n, gq, gt);
  return mf::check_launch("mf_transformation_matrix_bwd");
}

/* The pose epilogue of the training path (contrib/singleview_3d/models/model.py:262-273) on the three heads' fp32
 * outputs orot [n, 4 n_fg], otrn [n, 3 n_fg], ocnf [n, n_fg] (n = B * P rows, object b = row / P, class_id 1-based):
 * rot [n,4] normalised (x / (|x| + 1e-5)), trans [n,3] = (pts * pitch + origin) + raw * pitch, conf [n] = sigmoid --
 * and the backward: the three gradient row blocks, written completely (zeros outside the object's class). */
extern "C" int mf_pose_epilogue_train_fwd(const float *orot, const float *otrn, const float *ocnf, const int64_t *class_id,
                                          const float *pts, const float *origin, const float *pitch, int32_t B, int32_t P,
                                          int32_t n_fg, float *rot, float *trans, float *conf, mfStream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (B <= 0 || P <= 0) return 0;
  if (((uintptr_t)rot & 15) || n_fg < 1) {
    mf::set_last_error(hipErrorInvalidValue, "pose_epilogue_train: 16-byte aligned rot, n_fg >= 1");
    return -(int)hipErrorInvalidValue;
  }
  const int64_t n = (int64_t)B * P;
  hipLaunchKernelGGL(k_pose_epi3_fwd, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, orot, otrn, ocnf, class_id,
                     pts, origin, pitch, B, P, n_fg, rot, trans, conf);
  return mf::check_launch("mf_pose_epilogue_train_fwd");
}

extern "C" int mf_pose_epilogue_train_bwd(const float *orot, const float *ocnf, const int64_t *class_id,
                                          const float *pitch, const float *grot, const float *gtrans, const float *gconf,
                                          int32_t B, int32_t P, int32_t n_fg, float *drot, float *dtrn, float *dcnf,
                                          mfStream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (B <= 0 || P <= 0 || n_fg < 1) return 0;
  const int64_t n = (int64_t)B * P * 8 * n_fg;  // one lane per gradient element
  hipLaunchKernelGGL(k_pose_epi3_bwd, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, orot, ocnf, class_id, pitch,
                     grot, gtrans, gconf, B, P, n_fg, drot, dtrn, dcnf);
  return mf::check_launch("mf_pose_epilogue_train_bwd");
}
