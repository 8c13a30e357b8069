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
