// occupancy_grid_3d (fused min-over-points), geometry.nn (fused distance + arg-min),
// IterativeClosestPointLink loss/gradient -- gfx950.
//
// Reference:
//   occupancy_grid_3d.py:31-85  materialises three [X,Y,Z,P] float tensors (393 MB at
//     32^3 x 1000) and reduces them with five more launches;
//   knn/cuComputeDistanceGlobal.cu:20-86 + knn/nn.py:18-49 writes the full R x Q
//     distance matrix (1 GB at 500 x 500 000) and re-reads it for cupy.argmin;
//   contrib/iterative_closest_point_link.py:26-44 materialises [T,S,3].
// Here each output element keeps a running (min, arg-min) in registers while the
// point set streams through LDS in tiles that every lane reads at the same address
// (LDS broadcast, conflict-free): HBM traffic drops to the compulsory (R+Q)*12 B.
#include <math.h>

#include "mf_common.h"
#include "quat.h"

namespace {

constexpr int kTile = 1024;  // points per LDS tile (16 KB as float4)

// ---- A5 occupancy_grid_3d ---------------------------------------------------
__global__ __launch_bounds__(256) void k_occgrid_fwd(const float *__restrict__ points, int P,
                                                     float pitch, float ox, float oy, float oz,
                                                     int X, int Y, int Z, float threshold,
                                                     float *__restrict__ grid,
                                                     float *__restrict__ dmin_out) {
  __shared__ float4 s_p[kTile];
  const int v = blockIdx.x * blockDim.x + threadIdx.x;
  const int V = X * Y * Z;
  const float vi = (float)(v / (Y * Z)), vj = (float)((v / Z) % Y), vk = (float)(v % Z);
  float dmin = INFINITY;
  for (int base = 0; base < P; base += kTile) {
    const int nt = min(kTile, P - base);
    __syncthreads();
    for (int i = threadIdx.x; i < nt; i += blockDim.x) {
      const int p = base + i;
      // points = (points - origin) / pitch   (occupancy_grid_3d.py:43)
      s_p[i] = make_float4((points[3 * p] - ox) / pitch, (points[3 * p + 1] - oy) / pitch,
                           (points[3 * p + 2] - oz) / pitch, 0.0f);
    }
    __syncthreads();
    if (v < V) {
#pragma unroll 4
      for (int i = 0; i < nt; ++i) {
        const float4 q = s_p[i];
        const float a = vi - q.x, b = vj - q.y, c = vk - q.z;
        const float d = sqrtf((a * a + b * b) + c * c);
        dmin = fminf(dmin, d);
      }
    }
  }
  if (v < V) {
    float m = threshold - dmin;
    m = m > 0.0f ? m : 0.0f;          // relu
    grid[v] = m < 1.0f ? m : 1.0f;    // minimum(., 1)
    dmin_out[v] = dmin;
  }
}

// chainer rules: minimum(a,1) -> a where a <= 1; relu where > 0; F.min -> EVERY
// element equal to the minimum; sqrt -> gy/(2y); x**2 -> 2x gy; then
// OccupancyGrid3D.backward (occupancy_grid_3d.py:56-74): -g/pitch summed over voxels.
__global__ __launch_bounds__(256) void k_occgrid_bwd(const float *__restrict__ ggrid,
                                                     const float *__restrict__ points, int P,
                                                     float pitch, float ox, float oy, float oz,
                                                     int X, int Y, int Z, float threshold,
                                                     const float *__restrict__ dmin_in,
                                                     float *__restrict__ gpoints) {
  __shared__ float4 s_p[kTile];
  const int v = blockIdx.x * blockDim.x + threadIdx.x;
  const int V = X * Y * Z;
  const float vi = (float)(v / (Y * Z)), vj = (float)((v / Z) % Y), vk = (float)(v % Z);
  float dmin = 0.0f, g_d = 0.0f;
  if (v < V) {
    dmin = dmin_in[v];
    const float r = threshold - dmin;
    const float rr = r > 0.0f ? r : 0.0f;
    g_d = (r > 0.0f && rr <= 1.0f) ? -ggrid[v] : 0.0f;
  }
  for (int base = 0; base < P; base += kTile) {
    const int nt = min(kTile, P - base);
    __syncthreads();
    for (int i = threadIdx.x; i < nt; i += blockDim.x) {
      const int p = base + i;
      s_p[i] = make_float4((points[3 * p] - ox) / pitch, (points[3 * p + 1] - oy) / pitch,
                           (points[3 * p + 2] - oz) / pitch, 0.0f);
    }
    __syncthreads();
    if (v < V && g_d != 0.0f) {
      for (int i = 0; i < nt; ++i) {
        const float4 q = s_p[i];
        const float a = vi - q.x, b = vj - q.y, c = vk - q.z;
        const float d = sqrtf((a * a + b * b) + c * c);
        if (d == dmin) {
          const float g_dd = g_d / (2.0f * d);
          const int p = base + i;
          atomicAdd(&gpoints[3 * p], -(2.0f * a * g_dd) / pitch);
          atomicAdd(&gpoints[3 * p + 1], -(2.0f * b * g_dd) / pitch);
          atomicAdd(&gpoints[3 * p + 2], -(2.0f * c * g_dd) / pitch);
        }
      }
    }
  }
}

// ---- A11 geometry.nn ----------------------------------------------------------
// ssd accumulated x,y,z un-fused (cuComputeDistanceGlobal.cu:64-67), strict '<' while
// scanning refs in increasing index == argmin's first minimum (knn/nn.py:48).
__global__ __launch_bounds__(256) void k_nn(const float *__restrict__ ref, int R,
                                            const float *__restrict__ query, int64_t Q,
                                            int64_t *__restrict__ out,
                                            float *__restrict__ out_dist) {
  __shared__ float4 s_r[kTile];
  const int64_t qi = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  float qx = 0, qy = 0, qz = 0;
  if (qi < Q) { qx = query[3 * qi]; qy = query[3 * qi + 1]; qz = query[3 * qi + 2]; }
  float best = INFINITY;
  int bi = 0;
  for (int base = 0; base < R; base += kTile) {
    const int nt = min(kTile, R - base);
    __syncthreads();
    for (int i = threadIdx.x; i < nt; i += blockDim.x) {
      const int r = base + i;
      s_r[i] = make_float4(ref[3 * r], ref[3 * r + 1], ref[3 * r + 2], 0.0f);
    }
    __syncthreads();
#pragma unroll 4
    for (int i = 0; i < nt; ++i) {
      const float4 r = s_r[i];
      const float dx = r.x - qx, dy = r.y - qy, dz = r.z - qz;
      const float ssd = (dx * dx + dy * dy) + dz * dz;
      if (ssd < best) { best = ssd; bi = base + i; }
    }
  }
  if (qi < Q) {
    out[qi] = bi;
    if (out_dist) out_dist[qi] = best;
  }
}

// ---- A10 ICP link ---------------------------------------------------------------
// A workgroup = kIcpTargets target points x kIcpSlices lanes each: the transformed source streams
// through LDS in tiles, lane `sl` of a target scans sources sl, sl + 32, ... (adjacent lanes read
// adjacent float4: conflict-free), then the 32 partial (min, arg-min) meet in a 5-step butterfly
// with argmin's rule (lowest index among equal squared distances).  T x S work spread over
// T / 8 workgroups: one lane per target (round 1) left a 891-point target on 4 workgroups that
// each walked all 4740 sources serially -- 400 us per call, measured.
// out[0] += loss, out[1] += matches, out[4..15] += d loss / d [R|t] (row-major 3x4).
// Link l = blockIdx.y of a batch (src_off / tgt_off: [L+1] row offsets, NULL for a single link
// whose arrays are S / T rows long); Rt [L][12], out [L][16].
constexpr int kIcpSlices = 32;
constexpr int kIcpTargets = 256 / kIcpSlices;

__global__ __launch_bounds__(256) void k_icp(const float *__restrict__ source_all, int S_single,
                                             const float *__restrict__ target_all, int T_single,
                                             const int32_t *__restrict__ src_off,
                                             const int32_t *__restrict__ tgt_off,
                                             const float *__restrict__ Rt_all, float thresh,
                                             float *__restrict__ out_all) {
  const int l = blockIdx.y;
  const int s0 = src_off ? src_off[l] : 0, t0 = tgt_off ? tgt_off[l] : 0;
  const int S = src_off ? src_off[l + 1] - s0 : S_single;
  const int T = tgt_off ? tgt_off[l + 1] - t0 : T_single;
  if ((int)(blockIdx.x * kIcpTargets) >= T) return;  // block-uniform: grid.x covers the longest link
  const float *source = source_all + 3 * (int64_t)s0;
  const float *target = target_all + 3 * (int64_t)t0;
  const float *Rt = Rt_all + 12 * l;
  float *out = out_all + 16 * l;
  __shared__ float4 s_s[kTile];
  __shared__ float s_red[kIcpTargets][14];
  const int sl = threadIdx.x % kIcpSlices, tl = threadIdx.x / kIcpSlices;
  const int ti = blockIdx.x * kIcpTargets + tl;
  float R[12];
#pragma unroll
  for (int i = 0; i < 12; ++i) R[i] = Rt[i];
  float tx = 0, ty = 0, tz = 0;
  if (ti < T) { tx = target[3 * ti]; ty = target[3 * ti + 1]; tz = target[3 * ti + 2]; }
  float best = INFINITY;
  int bi = 0x7fffffff;
  float bx = 0, by = 0, bz = 0;
  for (int base = 0; base < S; base += kTile) {
    const int nt = min(kTile, S - base);
    __syncthreads();
    for (int i = threadIdx.x; i < nt; i += blockDim.x) {
      const int s = base + i;
      const float x = source[3 * s], y = source[3 * s + 1], z = source[3 * s + 2];
      s_s[i] = make_float4(((R[0] * x + R[1] * y) + R[2] * z) + R[9],
                           ((R[3] * x + R[4] * y) + R[5] * z) + R[10],
                           ((R[6] * x + R[7] * y) + R[8] * z) + R[11], 0.0f);
    }
    __syncthreads();
#pragma unroll 4
    for (int i = sl; i < nt; i += kIcpSlices) {
      const float4 s = s_s[i];
      const float dx = s.x - tx, dy = s.y - ty, dz = s.z - tz;
      const float ssd = (dx * dx + dy * dy) + dz * dz;
      if (ssd < best) { best = ssd; bi = base + i; bx = dx; by = dy; bz = dz; }
    }
  }
  // (min, lowest index) over the 32 slices of this target; every lane ends with the winner
#pragma unroll
  for (int off = kIcpSlices / 2; off > 0; off >>= 1) {
    const float ob = __shfl_xor(best, off);
    const int oi = __shfl_xor(bi, off);
    const float ox = __shfl_xor(bx, off), oy = __shfl_xor(by, off), oz = __shfl_xor(bz, off);
    if (ob < best || (ob == best && oi < bi)) { best = ob; bi = oi; bx = ox; by = oy; bz = oz; }
  }
  if (sl == 0) {
    float acc[14];
#pragma unroll
    for (int i = 0; i < 14; ++i) acc[i] = 0.0f;
    if (ti < T && best < thresh) {  // keep = dists < 0.02 (squared distance, :38)
      const float mx = source[3 * bi], my = source[3 * bi + 1], mz = source[3 * bi + 2];
      acc[0] = (bx * bx + by * by) + bz * bz;
      acc[1] = 1.0f;
      const float g[3] = {2.0f * bx, 2.0f * by, 2.0f * bz};
#pragma unroll
      for (int a = 0; a < 3; ++a) {
        acc[2 + 3 * a] = g[a] * mx;
        acc[3 + 3 * a] = g[a] * my;
        acc[4 + 3 * a] = g[a] * mz;
        acc[11 + a] = g[a];
      }
    }
#pragma unroll
    for (int i = 0; i < 14; ++i) s_red[tl][i] = acc[i];
  }
  __syncthreads();
  if (threadIdx.x < 14) {
    float s = 0.0f;
#pragma unroll
    for (int k = 0; k < kIcpTargets; ++k) s += s_red[k][threadIdx.x];  // targets in order
    const int dst = threadIdx.x < 2 ? threadIdx.x : threadIdx.x + 2;  // gR at 4..12, gt at 13..15
    if (s != 0.0f) atomicAdd(&out[dst], s);
  }
}


// The optimiser step of the fused ICP loop: one lane per link.  out[l] = {loss, matches, -, -,
// d loss / d R (9), d loss / d t (3)} of k_icp -> chain rule through the quaternion -> chainer-Adam
// -> next R|t; out[l] is cleared for the next iteration.
__global__ __launch_bounds__(64) void k_icp_step(int L, float *__restrict__ q, float *__restrict__ t,
                                                 float *__restrict__ adam_m, float *__restrict__ adam_v,
                                                 float aq, float at, float *__restrict__ Rt,
                                                 float *__restrict__ out, float *__restrict__ losses,
                                                 int apply) {
  const int l = blockIdx.x * blockDim.x + threadIdx.x;
  if (l >= L) return;
  float qq[4], tt[3];
#pragma unroll
  for (int i = 0; i < 4; ++i) qq[i] = q[4 * l + i];
#pragma unroll
  for (int i = 0; i < 3; ++i) tt[i] = t[3 * l + i];
  if (apply) {
    float gR[9], gt[3], gq[4], mm[7], vv[7];
#pragma unroll
    for (int i = 0; i < 9; ++i) gR[i] = out[16 * l + 4 + i];
#pragma unroll
    for (int i = 0; i < 3; ++i) gt[i] = out[16 * l + 13 + i];
    if (losses) losses[l] = out[16 * l];
    mf::quat_backward(qq, gR, gq);
#pragma unroll
    for (int i = 0; i < 7; ++i) { mm[i] = adam_m[7 * l + i]; vv[i] = adam_v[7 * l + i]; }
    mf::adam_pose_step(gq, gt, aq, at, qq, tt, mm, vv);
#pragma unroll
    for (int i = 0; i < 7; ++i) { adam_m[7 * l + i] = mm[i]; adam_v[7 * l + i] = vv[i]; }
#pragma unroll
    for (int i = 0; i < 4; ++i) q[4 * l + i] = qq[i];
#pragma unroll
    for (int i = 0; i < 3; ++i) t[3 * l + i] = tt[i];
  }
  float R[9];
  mf::quat_to_R(qq, R);
#pragma unroll
  for (int i = 0; i < 9; ++i) Rt[12 * l + i] = R[i];
#pragma unroll
  for (int i = 0; i < 3; ++i) Rt[12 * l + 9 + i] = tt[i];
#pragma unroll
  for (int i = 0; i < 16; ++i) out[16 * l + i] = 0.0f;
}

}  // namespace

extern "C" int mf_occupancy_grid_3d_fwd(const float *points, int64_t P, float pitch, float ox,
                                        float oy, float oz, int X, int Y, int Z, float threshold,
                                        float *grid, float *dmin, mfStream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  const int V = X * Y * Z;
  if (V == 0) return 0;
  hipLaunchKernelGGL(k_occgrid_fwd, dim3((V + 255) / 256), dim3(256), 0, stream, points, (int)P,
                     pitch, ox, oy, oz, X, Y, Z, threshold, grid, dmin);
  return mf::check_launch("mf_occupancy_grid_3d_fwd");
}

extern "C" int mf_occupancy_grid_3d_bwd(const float *ggrid, const float *points, int64_t P,
                                        float pitch, float ox, float oy, float oz, int X, int Y,
                                        int Z, float threshold, const float *dmin,
                                        float *gpoints, mfStream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  const int V = X * Y * Z;
  if (V == 0 || P == 0) return 0;
  hipLaunchKernelGGL(k_occgrid_bwd, dim3((V + 255) / 256), dim3(256), 0, stream, ggrid, points,
                     (int)P, pitch, ox, oy, oz, X, Y, Z, threshold, dmin, gpoints);
  return mf::check_launch("mf_occupancy_grid_3d_bwd");
}

extern "C" int mf_nn(const float *ref, int64_t R, const float *query, int64_t Q, int64_t *out,
                     float *out_dist, mfStream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (Q == 0) return 0;
  hipLaunchKernelGGL(k_nn, dim3((unsigned)((Q + 255) / 256)), dim3(256), 0, stream, ref, (int)R,
                     query, Q, out, out_dist);
  return mf::check_launch("mf_nn");
}

extern "C" int mf_icp_loss_grad(const float *source, int64_t S, const float *target, int64_t T,
                                const float *Rt, float thresh, float *out, mfStream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (T == 0 || S == 0) return 0;
  hipLaunchKernelGGL(k_icp, dim3((unsigned)((T + kIcpTargets - 1) / kIcpTargets)), dim3(256), 0, stream, source,
                     (int)S, target, (int)T, (const int32_t *)nullptr, (const int32_t *)nullptr, Rt, thresh, out);
  return mf::check_launch("mf_icp_loss_grad");
}

extern "C" int mf_icp_refine(const float *source, const int32_t *src_off, const float *target,
                             const int32_t *tgt_off, int32_t L, int32_t max_T, float thresh, float *q,
                             float *t, float *adam_m, float *adam_v, int32_t n_iter, int32_t step0,
                             float alpha_q, float alpha_t, float *losses, float *ws,
                             mfStream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (L <= 0 || n_iter <= 0) return 0;
  float *Rt = ws, *out = ws + 12 * (int64_t)L;  // ws: 28 floats per link
  const dim3 gs((L + 63) / 64), gk((unsigned)((max_T + kIcpTargets - 1) / kIcpTargets), L);
  // R|t of the initial poses, cleared sums; then n_iter x {loss + gradient, step}
  hipLaunchKernelGGL(k_icp_step, gs, dim3(64), 0, stream, (int)L, q, t, adam_m, adam_v, 0.0f, 0.0f, Rt, out,
                     (float *)nullptr, 0);
  for (int k = 0; k < n_iter; ++k) {
    if (max_T > 0)
      hipLaunchKernelGGL(k_icp, gk, dim3(256), 0, stream, source, 0, target, 0, src_off, tgt_off,
                         (const float *)Rt, thresh, out);
    // chainer Adam: alpha_t = alpha * sqrt(1 - b2^t) / (1 - b1^t), in double, cast once
    const int st = step0 + k + 1;
    const double fix1 = 1.0 - pow(0.9, (double)st), fix2 = 1.0 - pow(0.999, (double)st);
    hipLaunchKernelGGL(k_icp_step, gs, dim3(64), 0, stream, (int)L, q, t, adam_m, adam_v,
                       (float)((double)alpha_q * sqrt(fix2) / fix1), (float)((double)alpha_t * sqrt(fix2) / fix1),
                       Rt, out, losses ? losses + (int64_t)k * L : (float *)nullptr, 1);
  }
  return mf::check_launch("mf_icp_refine");
}
