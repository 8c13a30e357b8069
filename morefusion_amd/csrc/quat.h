// Quaternion (wxyz) <-> rotation matrix pieces shared by the refinement loops (icc.hip, the
// fused ICP loop in occgrid_knn.hip).
#pragma once
#include "mf_common.h"

namespace mf {

__device__ __forceinline__ void quat_to_R(const float *q, float *R) {
  // morefusion/functions/geometry/quaternion_matrix.py:65-78, :14-34
  const float n = ((q[0] * q[0] + q[1] * q[1]) + q[2] * q[2]) + q[3] * q[3];
  const float s = sqrtf(2.0f / n);
  const float qs[4] = {q[0] * s, q[1] * s, q[2] * s, q[3] * s};
  float Q[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) Q[i][j] = qs[i] * qs[j];
  R[0] = 1.0f - Q[2][2] - Q[3][3];
  R[1] = Q[1][2] - Q[3][0];
  R[2] = Q[1][3] + Q[2][0];
  R[3] = Q[1][2] + Q[3][0];
  R[4] = 1.0f - Q[1][1] - Q[3][3];
  R[5] = Q[2][3] - Q[1][0];
  R[6] = Q[1][3] - Q[2][0];
  R[7] = Q[2][3] + Q[1][0];
  R[8] = 1.0f - Q[1][1] - Q[2][2];
}

__device__ __forceinline__ void quat_backward(const float *q, const float *gR, float *gq) {
  // quaternion_matrix.py:36-51 (dR/dQ), outer product :54-62, scaling :71-72
  float gQ[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) gQ[i][j] = 0.0f;
  gQ[1][0] = -gR[5] + gR[7];
  gQ[1][1] = -gR[4] - gR[8];
  gQ[1][2] = gR[1] + gR[3];
  gQ[1][3] = gR[2] + gR[6];
  gQ[2][0] = gR[2] - gR[6];
  gQ[2][2] = -gR[0] - gR[8];
  gQ[2][3] = gR[5] + gR[7];
  gQ[3][0] = -gR[1] + gR[3];
  gQ[3][3] = -gR[0] - gR[4];
  const float n = ((q[0] * q[0] + q[1] * q[1]) + q[2] * q[2]) + q[3] * q[3];
  const float s = sqrtf(2.0f / n);
  const float qs[4] = {q[0] * s, q[1] * s, q[2] * s, q[3] * s};
  float gqs[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    float a = 0.0f, b = 0.0f;
#pragma unroll
    for (int j = 0; j < 4; ++j) { a += gQ[i][j] * qs[j]; b += gQ[j][i] * qs[j]; }
    gqs[i] = a + b;
  }
  const float dot = ((gqs[0] * q[0] + gqs[1] * q[1]) + gqs[2] * q[2]) + gqs[3] * q[3];
#pragma unroll
  for (int i = 0; i < 4; ++i) gq[i] = s * gqs[i] - (s / n) * dot * q[i];
}

// chainer.optimizers.Adam (v7) in float32 on the 7 pose parameters (q then t): m += (1-b1)(g-m);
// v += (1-b2)(g^2-v); theta -= alpha_t * m / (sqrt(v) + eps), alpha_t evaluated by the host.
__device__ __forceinline__ void adam_pose_step(const float *gq, const float *gt, float aq, float at,
                                               float *qq, float *tt, float *mm7, float *vv7) {
  const float omb1 = (float)(1.0 - 0.9), omb2 = (float)(1.0 - 0.999), eps = 1e-8f;
#pragma unroll
  for (int i = 0; i < 7; ++i) {
    const float gi = i < 4 ? gq[i] : gt[i - 4];
    float m = mm7[i], v = vv7[i];
    m += omb1 * (gi - m);
    v += omb2 * (gi * gi - v);
    mm7[i] = m;
    vv7[i] = v;
    const float upd = (i < 4 ? aq : at) * m / (sqrtf(v) + eps);
    if (i < 4) qq[i] -= upd; else tt[i - 4] -= upd;
  }
}

}  // namespace mf
