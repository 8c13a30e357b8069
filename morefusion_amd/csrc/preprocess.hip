// Pre-processing in front of the pose network -- gfx950.
//
// Reference: a host loop per instance, NumPy + imgviz (cv2 backend) on [H,W] images
//   ros/src/morefusion_ros/nodes/singleview_3d_pose_estimation.py:116-176
//   morefusion/datasets/rgbd_pose_estimation/base.py:112-137
//   geometry/pointcloud_from_depth.py:4-26, geometry/masks_to_bboxes.py:4-38
// i.e. back-project the whole depth image, then per instance: mask = label == id, skip if
// fewer than 50 valid points, tight bounding box, crop rgb (masked to 0) and the point
// cloud (masked to NaN), imgviz.centerize both to S x S (aspect-preserving resize +
// centred padding; rgb bilinear, points nearest).
//
// Here: two launches for ALL instances, nothing on the host.
//   k_pre_stats  one pass over the label/depth images: per instance bounding box, mask
//                area and valid-depth count (LDS atomics per workgroup, one global atomic
//                per touched instance and workgroup).
//   k_pre_crops  one thread per output pixel and instance: inverts the centerize geometry,
//                samples rgb with OpenCV's fixed-point bilinear arithmetic and the depth
//                image with its nearest-neighbour rule, back-projects only the sampled
//                pixel (float64 like the NumPy expression, stored as float32).
// HBM-bound and tiny: reads H*W*(4+4+3) bytes once + taps, writes n*S*S*15 bytes.
#include <limits.h>
#include <math.h>

#include <algorithm>

#include "mf_common.h"

namespace {

constexpr int kMaxInst = 256;
constexpr int kStatsThreads = 256;

__global__ void k_pre_init(int32_t *stats, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) {
    stats[6 * i + 0] = INT_MAX;  // y1
    stats[6 * i + 1] = INT_MAX;  // x1
    stats[6 * i + 2] = 0;        // y2 (max + 1)
    stats[6 * i + 3] = 0;        // x2
    stats[6 * i + 4] = 0;        // mask pixels
    stats[6 * i + 5] = 0;        // mask pixels with a valid depth
  }
}

__global__ __launch_bounds__(kStatsThreads) void k_pre_stats(
    const int32_t *__restrict__ label, const float *__restrict__ depth, int H, int W,
    const int32_t *__restrict__ ids, int n, int32_t *__restrict__ stats) {
  __shared__ int32_t s_id[kMaxInst];
  __shared__ int32_t s_st[kMaxInst][6];
  for (int i = threadIdx.x; i < n; i += blockDim.x) {
    s_id[i] = ids[i];
    s_st[i][0] = INT_MAX; s_st[i][1] = INT_MAX;
    s_st[i][2] = 0; s_st[i][3] = 0; s_st[i][4] = 0; s_st[i][5] = 0;
  }
  __syncthreads();
  const int64_t npix = (int64_t)H * W;
  for (int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; p < npix;
       p += (int64_t)gridDim.x * blockDim.x) {
    const int32_t l = label[p];
    const bool valid = !isnan(depth[p]);
    const int y = (int)(p / W), x = (int)(p % W);
    for (int i = 0; i < n; ++i) {  // duplicate ids each get their own statistics
      if (s_id[i] != l) continue;
      atomicMin(&s_st[i][0], y);
      atomicMin(&s_st[i][1], x);
      atomicMax(&s_st[i][2], y + 1);
      atomicMax(&s_st[i][3], x + 1);
      atomicAdd(&s_st[i][4], 1);
      if (valid) atomicAdd(&s_st[i][5], 1);
    }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < n; i += blockDim.x) {
    if (s_st[i][4] == 0) continue;
    atomicMin(&stats[6 * i + 0], s_st[i][0]);
    atomicMin(&stats[6 * i + 1], s_st[i][1]);
    atomicMax(&stats[6 * i + 2], s_st[i][2]);
    atomicMax(&stats[6 * i + 3], s_st[i][3]);
    atomicAdd(&stats[6 * i + 4], s_st[i][4]);
    atomicAdd(&stats[6 * i + 5], s_st[i][5]);
  }
}

// cv::resize INTER_LINEAR source index + fixed-point weights for one destination index
// (modules/imgproc/src/resize.cpp, resizeGeneric_ set-up): f = (float)((d+0.5)*scale-0.5),
// s = floor(f), f -= s; clamped at both borders; weights = short(rint(w * 2048)).
__device__ __forceinline__ void linear_tap(int d, double scale, int ssize, int &s0, int &s1,
                                           int &w0, int &w1, const bool zero_frac_at_border) {
  float f = (float)(((double)d + 0.5) * scale - 0.5);
  int s = (int)floorf(f);
  f -= (float)s;
  if (zero_frac_at_border) {  // the x direction resets the fraction, y only clamps the rows
    if (s < 0) { f = 0.0f; s = 0; }
    if (s >= ssize - 1) { f = 0.0f; s = ssize - 1; }
  }
  w0 = (int)(short)rintf((1.0f - f) * 2048.0f);
  w1 = (int)(short)rintf(f * 2048.0f);
  s0 = min(max(s, 0), ssize - 1);
  s1 = min(max(s + 1, 0), ssize - 1);
}

__global__ __launch_bounds__(256) void k_pre_crops(
    const uint8_t *__restrict__ rgb, const float *__restrict__ depth,
    const int32_t *__restrict__ label, int H, int W, double fx, double fy, double cx, double cy,
    const int32_t *__restrict__ ids, const int32_t *__restrict__ stats, int S, int min_valid,
    uint8_t *__restrict__ rgb_out, float *__restrict__ pcd_out, uint8_t *__restrict__ keep) {
  const int i = blockIdx.y;
  const int o = blockIdx.x * blockDim.x + threadIdx.x;
  if (o >= S * S) return;
  const int32_t id = ids[i];
  const int y1 = stats[6 * i + 0], x1 = stats[6 * i + 1];
  const int sh = stats[6 * i + 2] - y1, sw = stats[6 * i + 3] - x1;
  const bool ok = stats[6 * i + 4] > 0 && stats[6 * i + 5] >= min_valid;
  if (o == 0) keep[i] = ok ? 1 : 0;
  const int oy = o / S, ox = o % S;
  uint8_t *ro = rgb_out + ((int64_t)i * S * S + o) * 3;
  float *po = pcd_out + ((int64_t)i * S * S + o) * 3;
  const float nanv = __builtin_nanf("");
  ro[0] = 0; ro[1] = 0; ro[2] = 0;
  po[0] = nanv; po[1] = nanv; po[2] = nanv;
  if (!ok) return;
  // imgviz.centerize: scale = min(S/sh, S/sw); resized size = round(size*scale) (half-even)
  int dh = S, dw = S, ph = 0, pw = 0;
  const bool identity = (sh == S && sw == S);
  if (!identity) {
    const double scale_h = 1.0 * S / sh, scale_w = 1.0 * S / sw;
    const double scale = scale_h < scale_w ? scale_h : scale_w;
    dh = (int)rint(sh * scale);
    dw = (int)rint(sw * scale);
    if (dh < S) ph = (S - dh) / 2;
    if (dw < S) pw = (S - dw) / 2;
  }
  const int dy = oy - ph, dx = ox - pw;
  if (dy < 0 || dy >= dh || dx < 0 || dx >= dw || dh <= 0 || dw <= 0) return;  // padding
  auto masked = [&](int yy, int xx) { return label[(int64_t)(y1 + yy) * W + (x1 + xx)] == id; };

  // ---- points: cv::resize INTER_NEAREST: s = min(floor(d * (1/(dsize/ssize))), ssize-1)
  {
    int sy = dy, sx = dx;
    if (!identity) {
      const double ify = 1.0 / ((double)dh / sh), ifx = 1.0 / ((double)dw / sw);
      sy = min((int)floor(dy * ify), sh - 1);
      sx = min((int)floor(dx * ifx), sw - 1);
    }
    if (masked(sy, sx)) {
      const int r = y1 + sy, c = x1 + sx;
      const float z = depth[(int64_t)r * W + c];
      if (!isnan(z)) {  // pointcloud_from_depth.py:17-21, float64 arithmetic
        po[0] = (float)(((double)z * ((double)c - cx)) / fx);
        po[1] = (float)(((double)z * ((double)r - cy)) / fy);
        po[2] = z;
      }
    }
  }
  // ---- rgb: masked crop, cv::resize INTER_LINEAR for 8-bit (2x2 box for an exact 2:1)
  auto pix = [&](int yy, int xx, int c) -> int {
    return masked(yy, xx) ? (int)rgb[((int64_t)(y1 + yy) * W + (x1 + xx)) * 3 + c] : 0;
  };
  if (identity) {
#pragma unroll
    for (int c = 0; c < 3; ++c) ro[c] = (uint8_t)pix(dy, dx, c);
    return;
  }
  const double scale_x = 1.0 / ((double)dw / sw), scale_y = 1.0 / ((double)dh / sh);
  const bool area2 = fabs(scale_x - 2.0) < 2.220446049250313e-16 &&
                     fabs(scale_y - 2.0) < 2.220446049250313e-16;
  if (area2) {
#pragma unroll
    for (int c = 0; c < 3; ++c)
      ro[c] = (uint8_t)((pix(2 * dy, 2 * dx, c) + pix(2 * dy, 2 * dx + 1, c) +
                         pix(2 * dy + 1, 2 * dx, c) + pix(2 * dy + 1, 2 * dx + 1, c) + 2) >> 2);
    return;
  }
  int xa, xb, a0, a1, ya, yb, b0, b1;
  linear_tap(dx, scale_x, sw, xa, xb, a0, a1, true);
  linear_tap(dy, scale_y, sh, ya, yb, b0, b1, false);
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    const int r0 = pix(ya, xa, c) * a0 + pix(ya, xb, c) * a1;  // horizontal pass, row ya
    const int r1 = pix(yb, xa, c) * a0 + pix(yb, xb, c) * a1;  // row yb
    const int v = (((b0 * (r0 >> 4)) >> 16) + ((b1 * (r1 >> 4)) >> 16) + 2) >> 2;
    ro[c] = (uint8_t)min(max(v, 0), 255);
  }
}


// ---- valid-pixel compaction in front of Model.predict's point selection -----------------
// contrib/singleview_3d/models/model.py:195-196,206 -- `iy, ix = xp.where(mask[i])`,
// `n_point = int(mask[i].sum())` per object, with mask = ~isnan(pcd).any(axis): row-major list
// of the pixels whose three coordinates are all finite-or-inf (not NaN), and their count.
// A workgroup = (chunk of 4096 pixels, image): 4 consecutive pixels per lane (three 16-byte
// loads).  It first counts the valid pixels of the chunks BEFORE its own (all loads in flight, no
// barrier: the prefix is re-read from L2 rather than handed over between workgroups -- 16 chunks
// per 256 x 256 image, so no second launch and no inter-workgroup protocol), then scans its own
// chunk (wave prefix by shuffles, wave totals through LDS) and writes the indices.
// (As torch ops this was isnan / any / sum / cumsum / where x2 / scatter: the int64 cumsum over
// 8 x 65536 alone measured 149 us per predict; one workgroup per image walking all 16 chunks
// behind barriers measured 42 us.)
constexpr int kCompactThreads = 1024;
constexpr int kCompactChunk = 4 * kCompactThreads;

__device__ __forceinline__ unsigned valid4(const float *__restrict__ src, int p, int HW, int vec) {
  unsigned v = 0u;
  if (vec && p + 3 < HW) {
    const float4 *q = reinterpret_cast<const float4 *>(src + 3 * (int64_t)p);
    const float4 a = q[0], c = q[1], d = q[2];
    v = ((a.x == a.x && a.y == a.y && a.z == a.z) ? 1u : 0u) |
        ((a.w == a.w && c.x == c.x && c.y == c.y) ? 2u : 0u) |
        ((c.z == c.z && c.w == c.w && d.x == d.x) ? 4u : 0u) |
        ((d.y == d.y && d.z == d.z && d.w == d.w) ? 8u : 0u);
  } else {
    for (int k = 0; k < 4; ++k)
      if (p + k < HW) {
        const float x = src[3 * (int64_t)(p + k)], y = src[3 * (int64_t)(p + k) + 1], z = src[3 * (int64_t)(p + k) + 2];
        if (x == x && y == y && z == z) v |= 1u << k;
      }
  }
  return v;
}

__global__ __launch_bounds__(kCompactThreads) void k_valid_order(const float *__restrict__ pcd, int HW,
                                                                  int vec, int32_t *__restrict__ order,
                                                                  int32_t *__restrict__ counts) {
  __shared__ int s_wave[kCompactThreads / 64], s_pre[kCompactThreads / 64];
  const int b = blockIdx.y, chunk = blockIdx.x;
  const float *src = pcd + (int64_t)b * HW * 3;
  int32_t *dst = order + (int64_t)b * HW;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  // valid pixels of the chunks before this one
  int pre = 0;
  if (vec) {  // earlier chunks are full: no bounds test, 4 chunks (12 loads) in flight per lane
    constexpr int kU = 4;
    for (int c0 = 0; c0 < chunk; c0 += kU) {
      float4 r[kU][3];
#pragma unroll
      for (int u = 0; u < kU; ++u) {
        const int c = min(c0 + u, chunk - 1);
        const float4 *q = reinterpret_cast<const float4 *>(src + 3 * ((int64_t)c * kCompactChunk + 4 * (int)threadIdx.x));
        r[u][0] = q[0]; r[u][1] = q[1]; r[u][2] = q[2];
      }
#pragma unroll
      for (int u = 0; u < kU; ++u) {
        if (c0 + u >= chunk) continue;
        const float4 a = r[u][0], c = r[u][1], d = r[u][2];
        pre += ((a.x == a.x && a.y == a.y && a.z == a.z) ? 1 : 0) + ((a.w == a.w && c.x == c.x && c.y == c.y) ? 1 : 0) +
               ((c.z == c.z && c.w == c.w && d.x == d.x) ? 1 : 0) + ((d.y == d.y && d.z == d.z && d.w == d.w) ? 1 : 0);
      }
    }
  } else {
    for (int c = 0; c < chunk; ++c) pre += __popc(valid4(src, c * kCompactChunk + 4 * (int)threadIdx.x, HW, 0));
  }
  const int p = chunk * kCompactChunk + 4 * (int)threadIdx.x;
  const unsigned v = valid4(src, p, HW, vec);
  const int n = __popc(v);
  int incl = n;  // inclusive prefix over the wave
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const int t = __shfl_up(incl, off);
    if (lane >= off) incl += t;
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) pre += __shfl_xor(pre, off);
  if (lane == 63) s_wave[wave] = incl;
  if (lane == 0) s_pre[wave] = pre;
  __syncthreads();
  int before = 0, total = 0;
#pragma unroll
  for (int w = 0; w < kCompactThreads / 64; ++w) {
    const int cw = s_wave[w];
    before += s_pre[w] + (w < wave ? cw : 0);
    total += s_pre[w] + cw;
  }
  int at = before + incl - n;
#pragma unroll
  for (int k = 0; k < 4; ++k)
    if ((v >> k) & 1u) dst[at++] = p + k;
  if (threadIdx.x == 0 && chunk == (int)gridDim.x - 1) counts[b] = total;
}

}  // namespace

extern "C" int mf_instance_stats(const int32_t *label, const float *depth, int H, int W,
                                 const int32_t *instance_ids, int n_inst, int32_t *stats,
                                 mfStream_t stream) {
  if (n_inst <= 0) return 0;
  if (n_inst > kMaxInst || H <= 0 || W <= 0) {
    mf::set_last_error(hipErrorInvalidValue, "mf_instance_stats: 1..256 instances, H,W > 0");
    return -(int)hipErrorInvalidValue;
  }
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(k_pre_init, dim3((n_inst + 63) / 64), dim3(64), 0, s, stats, n_inst);
  const int64_t npix = (int64_t)H * W;
  const int blocks = (int)std::min<int64_t>((npix + kStatsThreads * 4 - 1) / (kStatsThreads * 4), 1024);
  hipLaunchKernelGGL(k_pre_stats, dim3(blocks), dim3(kStatsThreads), 0, s, label, depth, H, W,
                     instance_ids, n_inst, stats);
  return mf::check_launch("mf_instance_stats");
}

extern "C" int mf_instance_crops(const uint8_t *rgb, const float *depth, const int32_t *label,
                                 int H, int W, double fx, double fy, double cx, double cy,
                                 const int32_t *instance_ids, const int32_t *stats, int n_inst,
                                 int S, int min_valid, uint8_t *rgb_out, float *pcd_out,
                                 uint8_t *keep, mfStream_t stream) {
  if (n_inst <= 0) return 0;
  if (S <= 0 || H <= 0 || W <= 0) {
    mf::set_last_error(hipErrorInvalidValue, "mf_instance_crops: S, H, W > 0");
    return -(int)hipErrorInvalidValue;
  }
  hipLaunchKernelGGL(k_pre_crops, dim3((S * S + 255) / 256, n_inst), dim3(256), 0,
                     (hipStream_t)stream, rgb, depth, label, H, W, fx, fy, cx, cy, instance_ids,
                     stats, S, min_valid, rgb_out, pcd_out, keep);
  return mf::check_launch("mf_instance_crops");
}

extern "C" int mf_valid_pixel_order(const float *pcd, int32_t B, int32_t HW, int32_t *order,
                                    int32_t *counts, mfStream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (B <= 0) return 0;
  if (HW < 0) {
    mf::set_last_error(hipErrorInvalidValue, "mf_valid_pixel_order: negative image size");
    return -(int)hipErrorInvalidValue;
  }
  // 16-byte loads need every image to start 16-byte aligned: HW * 3 floats per image
  const int vec = (reinterpret_cast<uintptr_t>(pcd) & 15u) == 0 && HW % 4 == 0;
  if (HW == 0) return mf::fill_bytes(counts, 0, sizeof(int32_t) * B, stream);
  hipLaunchKernelGGL(k_valid_order, dim3((HW + kCompactChunk - 1) / kCompactChunk, B), dim3(kCompactThreads), 0, stream,
                     pcd, (int)HW, vec, order, counts);
  return mf::check_launch("mf_valid_pixel_order");
}
