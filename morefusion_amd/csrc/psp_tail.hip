// The last PSPNet level at the SAMPLED pixels only, fused, for gfx950.
//
// Reference: models/dense_fusion/pspnet.py:10-35,57-73 -- `up3` (bilinear x2 up-sampling, align_corners, of the
// [B,64,128,128] map + Convolution2D 3x3 64 -> 64 + PReLU), `conv1` (1x1, 64 -> 32) and log-softmax over the 32
// channels, evaluated over the whole 256^2 image; contrib/singleview_3d/models/model.py:222 then reads 1000
// pixels per object.  Round 1 restricted that level to the sampled pixels (4.8 GFLOP and > 400 MB of
// activations per 8 objects less) as ~55 torch launches: index arithmetic of the 9 x 4 bilinear taps, four
// gathers of [B,64,9000] elements, an einsum, a conv1d, a log-softmax.  This kernel is that level in ONE launch:
//   one WAVE per group of 4 sampled pixels;
//   lane c (a channel of the 64-channel map) forms the pixel's 3 x 3 window of the virtually up-sampled map:
//     9 window positions x 4 bilinear source pixels, same source-index arithmetic as F.interpolate
//     (src = dst * (H-1)/(Ho-1), floor, +1 clamped), zero outside the image (the 3x3 convolution's padding);
//     with a channels-last map the 64 lanes read one 256-byte row per source pixel;
//   the window goes to LDS as [9][64][4 pixels]; lane o then accumulates output channel o of the 3x3
//     convolution for the 4 pixels: per (tap, c) ONE coalesced weight load ([9][64][64], output channel
//     innermost), ONE broadcast ds_read_b128 and 4 FMAs; + bias, PReLU;
//   through LDS again for the 1x1 convolution (lanes 0..31), then log-softmax with wave shuffles;
//   out rows [n, 32]: the point MLP's GEMM input, no transpose.
#include <algorithm>

#include "mf_common.h"

namespace {

constexpr int kC = 64, kCo = 32, kPts = 4;  // channels of the map, output channels, pixels per wave iteration

struct TailArgs {
  const float *u2;          // [B,64,H,W] with element strides sb, sc, sy, sx
  int64_t sb, sc, sy, sx;
  const int64_t *pix;       // [B*P] flat index into the [2H, 2W] full-resolution map
  const float *w3t;         // [9][64 c][64 o]
  const float *b3;          // [64]
  const float *w1t;         // [64 c][32 o]
  const float *b1;          // [32]
  const float *slope;       // PReLU slope (one shared parameter), device pointer
  int B, P, H, W;
  float ry, rx;             // (H-1)/(2H-1), (W-1)/(2W-1) as float32 (F.interpolate, align_corners=True)
  float *out;               // [B*P][32]
};

__global__ __launch_bounds__(256) void k_psp_tail(TailArgs a) {
  __shared__ __attribute__((aligned(16))) float s_up[4][9][kC][kPts];
  __shared__ __attribute__((aligned(16))) float s_h[4][kC][kPts];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int64_t n = (int64_t)a.B * a.P;
  const int Ho = 2 * a.H, Wo = 2 * a.W;
  const int64_t groups = (n + kPts - 1) / kPts;
  for (int64_t g = (int64_t)blockIdx.x * 4 + wave; g < groups; g += (int64_t)gridDim.x * 4) {
    // ---- the 3x3 windows of 4 pixels, channel `lane`
#pragma unroll
    for (int t = 0; t < kPts; ++t) {
      const int64_t i = g * kPts + t;
      const bool live = i < n;
      const int64_t ii = live ? i : n - 1;
      const int b = (int)(ii / a.P);
      const int64_t pixel = a.pix[ii];
      const int py = (int)(pixel / Wo), px = (int)(pixel - (int64_t)py * Wo);
      const float *base = a.u2 + b * a.sb + lane * a.sc;
#pragma unroll
      for (int k = 0; k < 9; ++k) {
        const int yy = py + k / 3 - 1, xx = px + k % 3 - 1;
        const bool valid = live && yy >= 0 && yy < Ho && xx >= 0 && xx < Wo;
        const int yc = min(max(yy, 0), Ho - 1), xc = min(max(xx, 0), Wo - 1);
        const float sy = (float)yc * a.ry, sx = (float)xc * a.rx;
        const int y0 = (int)floorf(sy), x0 = (int)floorf(sx);
        const int y1 = min(y0 + 1, a.H - 1), x1 = min(x0 + 1, a.W - 1);
        const float ly = sy - (float)y0, lx = sx - (float)x0;
        const float t00 = base[y0 * a.sy + x0 * a.sx], t01 = base[y0 * a.sy + x1 * a.sx];
        const float t10 = base[y1 * a.sy + x0 * a.sx], t11 = base[y1 * a.sy + x1 * a.sx];
        const float v = (1.0f - ly) * ((1.0f - lx) * t00 + lx * t01) + ly * ((1.0f - lx) * t10 + lx * t11);
        s_up[wave][k][lane][t] = valid ? v : 0.0f;
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    // ---- 3x3 convolution, output channel `lane`, 4 pixels
    float acc[kPts];
    const float bias3 = a.b3[lane], slope = a.slope[0];
#pragma unroll
    for (int t = 0; t < kPts; ++t) acc[t] = bias3;
    for (int k = 0; k < 9; ++k) {
      const float *wk = a.w3t + (int64_t)k * kC * kC + lane;
#pragma unroll 8
      for (int c = 0; c < kC; ++c) {
        const float w = wk[c * kC];
        const float4 u = *reinterpret_cast<const float4 *>(&s_up[wave][k][c][0]);
        acc[0] = fmaf(u.x, w, acc[0]);
        acc[1] = fmaf(u.y, w, acc[1]);
        acc[2] = fmaf(u.z, w, acc[2]);
        acc[3] = fmaf(u.w, w, acc[3]);
      }
    }
#pragma unroll
    for (int t = 0; t < kPts; ++t) s_h[wave][lane][t] = acc[t] > 0.0f ? acc[t] : slope * acc[t];  // PReLU
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    // ---- 1x1 convolution 64 -> 32 (lanes 32..63 mirror lanes 0..31: the shuffles below stay full-wave)
    const int o = lane & (kCo - 1);
    float z[kPts];
    const float bias1 = a.b1[o];
#pragma unroll
    for (int t = 0; t < kPts; ++t) z[t] = bias1;
#pragma unroll 8
    for (int c = 0; c < kC; ++c) {
      const float w = a.w1t[c * kCo + o];
      const float4 h = *reinterpret_cast<const float4 *>(&s_h[wave][c][0]);
      z[0] = fmaf(h.x, w, z[0]);
      z[1] = fmaf(h.y, w, z[1]);
      z[2] = fmaf(h.z, w, z[2]);
      z[3] = fmaf(h.w, w, z[3]);
    }
    // ---- log-softmax over the 32 channels (each half-wave holds them once)
#pragma unroll
    for (int t = 0; t < kPts; ++t) {
      float m = z[t];
#pragma unroll
      for (int d = 16; d >= 1; d >>= 1) m = fmaxf(m, __shfl_xor(m, d, 64));
      float e = expf(z[t] - m);
#pragma unroll
      for (int d = 16; d >= 1; d >>= 1) e += __shfl_xor(e, d, 64);
      const int64_t i = g * kPts + t;
      if (i < n && lane < kCo) a.out[i * kCo + o] = (z[t] - m) - logf(e);
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();  // the next group overwrites s_up / s_h
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  }
}

// ---- the tail under bf16 training (round 5): window rows for the GEMM engines, and their backward -------------------
// Training keeps up3's 3x3 convolution, the 1x1 head and their weight gradients on the bf16 GEMM engines
// (csrc/gemm_bf16.hip through bf16_ops.Linear); what is left of the tail is index arithmetic + gathers forward
// (~75 torch launches incl. the taps) and four scatter-adds backward (~46).  Forward here: ONE launch builds the
// GEMM's input rows [n, 64 * 9] bf16 (column c * 9 + k: Convolution2D's own weight layout [o][c][ky][kx] flattened)
// from the channels-last bf16 map.  Backward: ONE launch folds every point's 9 x 4 tap gradients into its 3 x 3 patch
// of source pixels in LDS (a x2 up-sampling window touches at most 3 source rows / columns) and adds the patch to an
// fp32 image with <= 9 coalesced 256-byte atomics per point; a second launch rounds the image to bf16.
struct TapGeom {
  int y0[3], y1[3], x0[3], x1[3];
  float ly[3], lx[3];
  bool vy[3], vx[3];
};

__device__ __forceinline__ TapGeom tap_geom(int py, int px, int H, int W, float ry, float rx) {
  TapGeom g;
  const int Ho = 2 * H, Wo = 2 * W;
#pragma unroll
  for (int d = 0; d < 3; ++d) {
    const int yy = py + d - 1, xx = px + d - 1;
    g.vy[d] = yy >= 0 && yy < Ho;
    g.vx[d] = xx >= 0 && xx < Wo;
    const int yc = min(max(yy, 0), Ho - 1), xc = min(max(xx, 0), Wo - 1);
    const float sy = (float)yc * ry, sx = (float)xc * rx;
    g.y0[d] = (int)floorf(sy);
    g.x0[d] = (int)floorf(sx);
    g.y1[d] = min(g.y0[d] + 1, H - 1);
    g.x1[d] = min(g.x0[d] + 1, W - 1);
    g.ly[d] = sy - (float)g.y0[d];
    g.lx[d] = sx - (float)g.x0[d];
  }
  return g;
}

struct TailRowsArgs {
  const uint16_t *u2;   // [B, H, W, 64] bf16 (channels-last)
  const int64_t *pix;   // [B * P]
  int B, P, H, W;
  float ry, rx;
  uint16_t *rows;       // [B * P][576] bf16 (forward: out; backward: the rows' gradient, in)
  float *acc;           // backward: [B, H, W, 64] fp32, zero on entry
};

__global__ __launch_bounds__(256) void k_tail_rows_fwd(TailRowsArgs a) {
  __shared__ __attribute__((aligned(16))) uint16_t s_row[4][kC * 9];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int64_t n = (int64_t)a.B * a.P;
  const int Wo = 2 * a.W;
  for (int64_t i = (int64_t)blockIdx.x * 4 + wave; i < n; i += (int64_t)gridDim.x * 4) {
    const int b = (int)(i / a.P);
    const int64_t pixel = a.pix[i];
    const int py = (int)(pixel / Wo), px = (int)(pixel - (int64_t)py * Wo);
    const TapGeom g = tap_geom(py, px, a.H, a.W, a.ry, a.rx);
    const uint16_t *base = a.u2 + (int64_t)b * a.H * a.W * kC + lane;
#pragma unroll
    for (int k = 0; k < 9; ++k) {
      const int r = k / 3, q = k % 3;
      float v = 0.0f;
      if (g.vy[r] && g.vx[q]) {
        const float t00 = __uint_as_float((uint32_t)base[((int64_t)g.y0[r] * a.W + g.x0[q]) * kC] << 16);
        const float t01 = __uint_as_float((uint32_t)base[((int64_t)g.y0[r] * a.W + g.x1[q]) * kC] << 16);
        const float t10 = __uint_as_float((uint32_t)base[((int64_t)g.y1[r] * a.W + g.x0[q]) * kC] << 16);
        const float t11 = __uint_as_float((uint32_t)base[((int64_t)g.y1[r] * a.W + g.x1[q]) * kC] << 16);
        v = (1.0f - g.ly[r]) * ((1.0f - g.lx[q]) * t00 + g.lx[q] * t01) +
            g.ly[r] * ((1.0f - g.lx[q]) * t10 + g.lx[q] * t11);
      }
      s_row[wave][lane * 9 + k] = (uint16_t)mf::bf16_bits(v);
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    const uint32_t *src = reinterpret_cast<const uint32_t *>(s_row[wave]);
    uint32_t *dst = reinterpret_cast<uint32_t *>(a.rows + i * (kC * 9));
    for (int w = lane; w < kC * 9 / 2; w += 64) dst[w] = src[w];
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  }
}

__global__ __launch_bounds__(256) void k_tail_rows_bwd(TailRowsArgs a) {
  __shared__ __attribute__((aligned(16))) uint16_t s_row[4][kC * 9];
  __shared__ float s_patch[4][9][kC];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int64_t n = (int64_t)a.B * a.P;
  const int Wo = 2 * a.W;
  for (int64_t i = (int64_t)blockIdx.x * 4 + wave; i < n; i += (int64_t)gridDim.x * 4) {
    const int b = (int)(i / a.P);
    const int64_t pixel = a.pix[i];
    const int py = (int)(pixel / Wo), px = (int)(pixel - (int64_t)py * Wo);
    const TapGeom g = tap_geom(py, px, a.H, a.W, a.ry, a.rx);
    const uint32_t *src = reinterpret_cast<const uint32_t *>(a.rows + i * (kC * 9));
    uint32_t *dst = reinterpret_cast<uint32_t *>(s_row[wave]);
    for (int w = lane; w < kC * 9 / 2; w += 64) dst[w] = src[w];
#pragma unroll
    for (int k = 0; k < 9; ++k) s_patch[wave][k][lane] = 0.0f;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    // source rows / columns of the window: y0[0] .. y0[0] + 2 (x2 up-sampling: three consecutive output rows map
    // into at most two source rows + their lower neighbours); the lane owns column `lane` of the patch
    const int by = g.y0[0], bx = g.x0[0];
#pragma unroll
    for (int k = 0; k < 9; ++k) {
      const int r = k / 3, q = k % 3;
      if (!(g.vy[r] && g.vx[q])) continue;
      const float gv = __uint_as_float((uint32_t)s_row[wave][lane * 9 + k] << 16);
      const float wy1 = g.ly[r], wy0 = 1.0f - wy1, wx1 = g.lx[q], wx0 = 1.0f - wx1;
      const int dy0 = g.y0[r] - by, dy1 = g.y1[r] - by, dx0 = g.x0[q] - bx, dx1 = g.x1[q] - bx;
      s_patch[wave][dy0 * 3 + dx0][lane] += (wy0 * wx0) * gv;
      s_patch[wave][dy0 * 3 + dx1][lane] += (wy0 * wx1) * gv;
      s_patch[wave][dy1 * 3 + dx0][lane] += (wy1 * wx0) * gv;
      s_patch[wave][dy1 * 3 + dx1][lane] += (wy1 * wx1) * gv;
    }
    float *img = a.acc + (int64_t)b * a.H * a.W * kC + lane;
#pragma unroll
    for (int k = 0; k < 9; ++k) {
      const int y = by + k / 3, x = bx + k % 3;
      const float v = s_patch[wave][k][lane];
      if (y < a.H && x < a.W && __ballot(v != 0.0f) != 0)
        mf::atomic_add_f32(img + ((int64_t)y * a.W + x) * kC, v);
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  }
}

__global__ __launch_bounds__(256) void k_tail_zero(float4 *p, int64_t n4) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256)
    p[i] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
}

__global__ __launch_bounds__(256) void k_tail_round(const float4 *__restrict__ acc, uint2 *__restrict__ out, int64_t n4) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
    const float4 v = acc[i];
    out[i] = make_uint2(mf::pack_bf16x2(v.x, v.y), mf::pack_bf16x2(v.z, v.w));
  }
}

}  // namespace

extern "C" int mf_psp_tail_fwd(const float *u2, int64_t sb, int64_t sc, int64_t sy, int64_t sx, const int64_t *pix,
                               const float *w3t, const float *b3, const float *prelu_slope, const float *w1t,
                               const float *b1, int32_t B, int32_t P, int32_t H, int32_t W, float *out,
                               mfStream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (B <= 0 || P <= 0) return 0;
  if (H < 2 || W < 2) {
    mf::set_last_error(hipErrorInvalidValue, "psp_tail: the source map must be at least 2 x 2");
    return -(int)hipErrorInvalidValue;
  }
  TailArgs a;
  a.u2 = u2; a.sb = sb; a.sc = sc; a.sy = sy; a.sx = sx; a.pix = pix; a.w3t = w3t; a.b3 = b3; a.w1t = w1t; a.b1 = b1;
  a.slope = prelu_slope; a.B = B; a.P = P; a.H = H; a.W = W;
  // python: yy.to(float32) * ((H - 1) / (Ho - 1)) -- the double quotient rounded to float32 once
  a.ry = (float)((double)(H - 1) / (double)(2 * H - 1));
  a.rx = (float)((double)(W - 1) / (double)(2 * W - 1));
  a.out = out;
  const int64_t groups = ((int64_t)B * P + kPts - 1) / kPts;
  const unsigned nb = (unsigned)std::min<int64_t>((groups + 3) / 4, 2048);
  hipLaunchKernelGGL(k_psp_tail, dim3(nb), dim3(256), 0, stream, a);
  return mf::check_launch("mf_psp_tail_fwd");
}

/* The sampled tail under bf16 training: the 3 x 3 windows of the (virtually) up-sampled [B, 2H, 2W, 64] map at the
 * sampled pixels as GEMM rows [B * P, 576] bf16 (column c * 9 + ky * 3 + kx), from the channels-last bf16 map
 * u2 [B, H, W, 64] (models/dense_fusion/pspnet.py:18-22,50-56 restricted to model.py:222's pixels). */
extern "C" int mf_psp_tail_rows_bf16_fwd(const void *u2, const int64_t *pix, int32_t B, int32_t P, int32_t H, int32_t W,
                                         void *rows, mfStream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (B <= 0 || P <= 0) return 0;
  if (H < 2 || W < 2 || ((uintptr_t)rows & 3)) {
    mf::set_last_error(hipErrorInvalidValue, "psp_tail_rows: a source map of at least 2 x 2, 4-byte aligned rows");
    return -(int)hipErrorInvalidValue;
  }
  TailRowsArgs a;
  a.u2 = (const uint16_t *)u2; a.pix = pix; a.B = B; a.P = P; a.H = H; a.W = W;
  a.ry = (float)((double)(H - 1) / (double)(2 * H - 1));
  a.rx = (float)((double)(W - 1) / (double)(2 * W - 1));
  a.rows = (uint16_t *)rows; a.acc = nullptr;
  const int64_t n = (int64_t)B * P;
  hipLaunchKernelGGL(k_tail_rows_fwd, dim3((unsigned)std::min<int64_t>((n + 3) / 4, 4096)), dim3(256), 0, stream, a);
  return mf::check_launch("mf_psp_tail_rows_bf16_fwd");
}

/* Its backward: grows [B * P, 576] bf16 -> gu2 [B, H, W, 64] bf16 through acc [B, H, W, 64] fp32 (workspace; zeroed
 * here).  The fp32 atomics make the sum order (not the set of addends) depend on the launch. */
extern "C" int mf_psp_tail_rows_bf16_bwd(const void *grows, const int64_t *pix, int32_t B, int32_t P, int32_t H, int32_t W,
                                         float *acc, void *gu2, mfStream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (B <= 0 || H < 2 || W < 2) {
    mf::set_last_error(hipErrorInvalidValue, "psp_tail_rows_bwd: B >= 1 and a source map of at least 2 x 2");
    return -(int)hipErrorInvalidValue;
  }
  TailRowsArgs a;
  a.u2 = nullptr; a.pix = pix; a.B = B; a.P = P; a.H = H; a.W = W;
  a.ry = (float)((double)(H - 1) / (double)(2 * H - 1));
  a.rx = (float)((double)(W - 1) / (double)(2 * W - 1));
  a.rows = (uint16_t *)const_cast<void *>(grows); a.acc = acc;
  const int64_t n4 = (int64_t)B * H * W * kC / 4, n = (int64_t)B * P;
  const unsigned nz = (unsigned)std::min<int64_t>((n4 + 255) / 256, 8192);
  hipLaunchKernelGGL(k_tail_zero, dim3(nz), dim3(256), 0, stream, reinterpret_cast<float4 *>(acc), n4);
  if (n > 0)
    hipLaunchKernelGGL(k_tail_rows_bwd, dim3((unsigned)std::min<int64_t>((n + 3) / 4, 4096)), dim3(256), 0, stream, a);
  hipLaunchKernelGGL(k_tail_round, dim3(nz), dim3(256), 0, stream, reinterpret_cast<const float4 *>(acc),
                     reinterpret_cast<uint2 *>(gu2), n4);
  return mf::check_launch("mf_psp_tail_rows_bf16_bwd");
}
