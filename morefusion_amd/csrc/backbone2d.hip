// Element-wise pieces of the 2-D backbone's decoder, forward and backward, for gfx950 (round 4).
//
// Reference: morefusion/models/dense_fusion/pspnet.py:10-35,40-73 -- PSPNet's decoder is three times
// `F.resize_images(x, 2x)` (bilinear, align_corners) -> Convolution2D 3x3 -> PReLU, and the pooling pyramid
// up-samples four pooled maps back to the feature size.  The convolutions stay MIOpen's; the bilinear resize and
// the single-parameter PReLU are memory-bound maps that the stock kernels run far below the HBM roofline:
// measured on the bf16 training step (profiles/r04_train_bf16_steady_step_kernel_stats.csv, 16 objects):
// upsample_bilinear2d_backward 2.8 ms for ~600 MB (0.2 TB/s: a float-atomic scatter), prelu_backward 1.3 ms for
// ~400 MB (0.3 TB/s: the gradient of the ONE slope is a whole-tensor reduction), the forward resize 0.7 ms.
//
// MI355X design: channels-LAST tensors ([B][H][W][C]: what MIOpen's NHWC convolutions on either side consume), a lane
// takes one pixel x 8 channels (bf16: one 16-byte access; fp32: two) -> every load and store is a full coalesced row
// segment.  The resize backward is a GATHER: an input pixel sums the <= ~4 x 4 output pixels whose bilinear footprint
// contains it, in increasing (oy, ox) -- no atomics, deterministic (torch's scatter is neither).  The PReLU
// backward writes dx and block-partial sums of dy * x over the negative side; a second launch adds the partials in
// block order.
#include "mf_common.h"

namespace {

// torch's area_pixel_compute_scale / source index for align_corners = true, in float
__device__ __forceinline__ float up_scale(int in, int out) { return out > 1 ? (float)(in - 1) / (float)(out - 1) : 0.0f; }

template <bool BF16>
__device__ __forceinline__ void load8(const void *p, int64_t idx, float v[8]) {
  if (BF16) {
    const uint4 w = *reinterpret_cast<const uint4 *>(reinterpret_cast<const uint16_t *>(p) + idx);
    const uint32_t d[4] = {w.x, w.y, w.z, w.w};
#pragma unroll
    for (int i = 0; i < 4; ++i) { v[2 * i] = mf::bf16_lo(d[i]); v[2 * i + 1] = mf::bf16_hi(d[i]); }
  } else {
    const float4 a = *reinterpret_cast<const float4 *>(reinterpret_cast<const float *>(p) + idx);
    const float4 b = *reinterpret_cast<const float4 *>(reinterpret_cast<const float *>(p) + idx + 4);
    v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
  }
}

template <bool BF16>
__device__ __forceinline__ void store8(void *p, int64_t idx, const float v[8]) {
  if (BF16) {
    *reinterpret_cast<uint4 *>(reinterpret_cast<uint16_t *>(p) + idx) =
        make_uint4(mf::pack_bf16x2(v[0], v[1]), mf::pack_bf16x2(v[2], v[3]), mf::pack_bf16x2(v[4], v[5]),
                   mf::pack_bf16x2(v[6], v[7]));
  } else {
    *reinterpret_cast<float4 *>(reinterpret_cast<float *>(p) + idx) = make_float4(v[0], v[1], v[2], v[3]);
    *reinterpret_cast<float4 *>(reinterpret_cast<float *>(p) + idx + 4) = make_float4(v[4], v[5], v[6], v[7]);
  }
}

// y [B][Ho][Wo][C] = bilinear(x [B][H][W][C]), align_corners = true; the sum order of torch's kernel:
// h0 (w0 x00 + w1 x01) + h1 (w0 x10 + w1 x11)
template <bool BF16>
__global__ __launch_bounds__(256) void k_up_fwd(const void *__restrict__ x, void *__restrict__ y, int B, int H, int W,
                                                int Ho, int Wo, int C8) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t total = (int64_t)B * Ho * Wo * C8;
  if (i >= total) return;
  const int c8 = (int)(i % C8);
  const int64_t pix = i / C8;
  const int ox = (int)(pix % Wo), oy = (int)((pix / Wo) % Ho), b = (int)(pix / ((int64_t)Wo * Ho));
  const float sy = up_scale(H, Ho) * (float)oy, sx = up_scale(W, Wo) * (float)ox;
  const int y0 = (int)sy, x0 = (int)sx;
  const int yp = y0 < H - 1 ? 1 : 0, xp = x0 < W - 1 ? 1 : 0;
  const float h1 = sy - (float)y0, h0 = 1.0f - h1, w1 = sx - (float)x0, w0 = 1.0f - w1;
  const int C = 8 * C8;
  const int64_t base = (((int64_t)b * H + y0) * W + x0) * C + 8 * c8;
  float a00[8], a01[8], a10[8], a11[8], o[8];
  load8<BF16>(x, base, a00);
  load8<BF16>(x, base + (int64_t)xp * C, a01);
  load8<BF16>(x, base + (int64_t)yp * W * C, a10);
  load8<BF16>(x, base + ((int64_t)yp * W + xp) * C, a11);
#pragma unroll
  for (int k = 0; k < 8; ++k) o[k] = h0 * (w0 * a00[k] + w1 * a01[k]) + h1 * (w0 * a10[k] + w1 * a11[k]);
  store8<BF16>(y, pix * C + 8 * c8, o);
}

// gx [B][H][W][C] = sum over the output pixels whose footprint contains (iy, ix), increasing (oy, ox)
template <bool BF16>
__global__ __launch_bounds__(256) void k_up_bwd(const void *__restrict__ gy, void *__restrict__ gx, int B, int H, int W,
                                                int Ho, int Wo, int C8) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t total = (int64_t)B * H * W * C8;
  if (i >= total) return;
  const int c8 = (int)(i % C8);
  const int64_t pix = i / C8;
  const int ix = (int)(pix % W), iy = (int)((pix / W) % H), b = (int)(pix / ((int64_t)W * H));
  const float ssy = up_scale(H, Ho), ssx = up_scale(W, Wo);
  // candidate output rows / columns: those with y0 in {iy - 1, iy} (generous bounds, exact test inside)
  int oy_lo = 0, oy_hi = Ho - 1, ox_lo = 0, ox_hi = Wo - 1;
  if (ssy > 0.0f) {
    oy_lo = max(0, (int)floorf((float)(iy - 1) / ssy));   // one row of slack on each side for the float quotient
    oy_hi = min(Ho - 1, (int)ceilf((float)(iy + 1) / ssy));
  }
  if (ssx > 0.0f) {
    ox_lo = max(0, (int)floorf((float)(ix - 1) / ssx));
    ox_hi = min(Wo - 1, (int)ceilf((float)(ix + 1) / ssx));
  }
  const int C = 8 * C8;
  float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  for (int oy = oy_lo; oy <= oy_hi; ++oy) {
    const float sy = ssy * (float)oy;
    const int y0 = (int)sy;
    const int yp = y0 < H - 1 ? 1 : 0;
    const float h1 = sy - (float)y0, h0 = 1.0f - h1;
    if (!(y0 == iy || y0 + yp == iy)) continue;
    float wy = 0.0f;
    if (y0 == iy) wy += h0;
    if (y0 + yp == iy) wy += h1;
    for (int ox = ox_lo; ox <= ox_hi; ++ox) {
      const float sx = ssx * (float)ox;
      const int x0 = (int)sx;
      const int xp = x0 < W - 1 ? 1 : 0;
      const float w1 = sx - (float)x0, w0 = 1.0f - w1;
      if (!(x0 == ix || x0 + xp == ix)) continue;
      float wx = 0.0f;
      if (x0 == ix) wx += w0;
      if (x0 + xp == ix) wx += w1;
      float g[8];
      load8<BF16>(gy, (((int64_t)b * Ho + oy) * Wo + ox) * C + 8 * c8, g);
      const float w = wy * wx;
#pragma unroll
      for (int k = 0; k < 8; ++k) acc[k] += w * g[k];
    }
  }
  store8<BF16>(gx, pix * C + 8 * c8, acc);
}


// The channels-last bf16 backward through an LDS tile: a workgroup owns 8 x 8 input pixels x 64 channels, stages
// the output-gradient patch their footprints cover (~20 x 20 pixels for a x2 resize: 51 KB) with coalesced 16-byte
// loads, and the 512 lanes (pixel, 8-channel chunk) gather from LDS -- the direct kernel re-reads every output
// pixel ~6x from L2 (measured 0.37 ms for [16,256,64,64] <- [16,256,128,128]; the patch is read 1.6x).
constexpr int kUpT = 8;
__device__ __forceinline__ int up_lo(int i, float s) { return s > 0.0f ? max(0, (int)floorf((float)(i - 1) / s)) : 0; }
__device__ __forceinline__ int up_hi(int i, float s, int n_out) {
  return s > 0.0f ? min(n_out - 1, (int)ceilf((float)(i + 1) / s)) : n_out - 1;
}

__global__ __launch_bounds__(512) void k_up_bwd_tile_bf16(const uint16_t *__restrict__ gy, uint16_t *__restrict__ gx,
                                                          int H, int W, int Ho, int Wo, int C, int tiles_x) {
  MF_DYN_LDS(uint4, s_patch);  // [rows][cols][8 chunks]
  const int ty0 = (blockIdx.x / tiles_x) * kUpT, tx0 = (blockIdx.x % tiles_x) * kUpT;
  const int c0 = blockIdx.y * 64, b = blockIdx.z;
  const float ssy = up_scale(H, Ho), ssx = up_scale(W, Wo);
  const int r_lo = up_lo(ty0, ssy), r_hi = up_hi(min(ty0 + kUpT - 1, H - 1), ssy, Ho);
  const int q_lo = up_lo(tx0, ssx), q_hi = up_hi(min(tx0 + kUpT - 1, W - 1), ssx, Wo);
  const int R = r_hi - r_lo + 1, S = q_hi - q_lo + 1;
  for (int i = threadIdx.x; i < R * S * 8; i += 512) {
    const int ch = i & 7, pix = i >> 3, r = pix / S, q = pix - r * S;
    s_patch[i] = *reinterpret_cast<const uint4 *>(gy + (((int64_t)b * Ho + r_lo + r) * Wo + q_lo + q) * C + c0 + 8 * ch);
  }
  __syncthreads();
  const int ch = threadIdx.x & 7, p = threadIdx.x >> 3;
  const int iy = ty0 + p / kUpT, ix = tx0 + p % kUpT;
  if (iy >= H || ix >= W) return;
  float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  const int oy_lo = up_lo(iy, ssy), oy_hi = up_hi(iy, ssy, Ho), ox_lo = up_lo(ix, ssx), ox_hi = up_hi(ix, ssx, Wo);
  for (int oy = oy_lo; oy <= oy_hi; ++oy) {
    const float sy = ssy * (float)oy;
    const int y0 = (int)sy;
    const int yp = y0 < H - 1 ? 1 : 0;
    if (!(y0 == iy || y0 + yp == iy)) continue;
    const float h1 = sy - (float)y0, h0 = 1.0f - h1;
    float wy = 0.0f;
    if (y0 == iy) wy += h0;
    if (y0 + yp == iy) wy += h1;
    for (int ox = ox_lo; ox <= ox_hi; ++ox) {
      const float sx = ssx * (float)ox;
      const int x0 = (int)sx;
      const int xp = x0 < W - 1 ? 1 : 0;
      if (!(x0 == ix || x0 + xp == ix)) continue;
      const float w1 = sx - (float)x0, w0 = 1.0f - w1;
      float wx = 0.0f;
      if (x0 == ix) wx += w0;
      if (x0 + xp == ix) wx += w1;
      const uint4 g = s_patch[((oy - r_lo) * S + (ox - q_lo)) * 8 + ch];
      const uint32_t d[4] = {g.x, g.y, g.z, g.w};
      const float w = wy * wx;
#pragma unroll
      for (int k = 0; k < 4; ++k) { acc[2 * k] += w * mf::bf16_lo(d[k]); acc[2 * k + 1] += w * mf::bf16_hi(d[k]); }
    }
  }
  store8<true>(gx, (((int64_t)b * H + iy) * W + ix) * C + c0 + 8 * ch, acc);
}

// Channels-last backward when the INPUT map is small (PSPNet's pooled 1x1 .. 6x6 maps resized to 32 x 32): an input
// pixel's footprint is hundreds of output pixels -- walking it with one lane (the direct kernel) measured 0.37 ms for
// 17 MB.  One workgroup per (input pixel, 64 channels, image): 64 lanes share the footprint's output pixels, 8 lanes
// the channel chunks; partial sums meet in LDS in a fixed order.
template <bool BF16>
__global__ __launch_bounds__(512) void k_up_bwd_small(const void *__restrict__ gy, void *__restrict__ gx, int H, int W,
                                                      int Ho, int Wo, int C) {
  __shared__ float s_part[64][8][9];
  const int ix = blockIdx.x % W, iy = blockIdx.x / W;
  const int c0 = blockIdx.y * 64, b = blockIdx.z;
  const int ch = threadIdx.x & 7, pl = threadIdx.x >> 3;
  const float ssy = up_scale(H, Ho), ssx = up_scale(W, Wo);
  const int oy_lo = up_lo(iy, ssy), oy_hi = up_hi(iy, ssy, Ho), ox_lo = up_lo(ix, ssx), ox_hi = up_hi(ix, ssx, Wo);
  const int nx = ox_hi - ox_lo + 1, n = (oy_hi - oy_lo + 1) * nx;
  float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  for (int k = pl; k < n; k += 64) {
    const int oy = oy_lo + k / nx, ox = ox_lo + k % nx;
    const float sy = ssy * (float)oy, sx = ssx * (float)ox;
    const int y0 = (int)sy, x0 = (int)sx;
    const int yp = y0 < H - 1 ? 1 : 0, xp = x0 < W - 1 ? 1 : 0;
    if (!(y0 == iy || y0 + yp == iy) || !(x0 == ix || x0 + xp == ix)) continue;
    const float h1 = sy - (float)y0, h0 = 1.0f - h1, w1 = sx - (float)x0, w0 = 1.0f - w1;
    float wy = 0.0f, wx = 0.0f;
    if (y0 == iy) wy += h0;
    if (y0 + yp == iy) wy += h1;
    if (x0 == ix) wx += w0;
    if (x0 + xp == ix) wx += w1;
    float g[8];
    load8<BF16>(gy, (((int64_t)b * Ho + oy) * Wo + ox) * C + c0 + 8 * ch, g);
    const float w = wy * wx;
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[j] += w * g[j];
  }
#pragma unroll
  for (int j = 0; j < 8; ++j) s_part[pl][ch][j] = acc[j];
  __syncthreads();
  if (threadIdx.x < 64) {  // lane = (chunk, element): add the 64 partials in lane order
    const int cc = threadIdx.x >> 3, j = threadIdx.x & 7;
    float t = 0.0f;
    for (int q = 0; q < 64; ++q) t += s_part[q][cc][j];
    const int64_t o = (((int64_t)b * H + iy) * W + ix) * C + c0 + threadIdx.x;
    if (BF16) reinterpret_cast<uint16_t *>(gx)[o] = (uint16_t)mf::bf16_bits(t);
    else reinterpret_cast<float *>(gx)[o] = t;
  }
}

// The same maps for channels-FIRST tensors [B][C][H][W] (what MIOpen's fp32 NCHW solvers leave behind at inference):
// one lane per element, coalesced along x.
template <bool BF16>
__device__ __forceinline__ float load1(const void *p, int64_t i) {
  return BF16 ? __uint_as_float((uint32_t)reinterpret_cast<const uint16_t *>(p)[i] << 16) : reinterpret_cast<const float *>(p)[i];
}
template <bool BF16>
__device__ __forceinline__ void store1(void *p, int64_t i, float v) {
  if (BF16) reinterpret_cast<uint16_t *>(p)[i] = (uint16_t)mf::bf16_bits(v);
  else reinterpret_cast<float *>(p)[i] = v;
}

template <bool BF16>
__global__ __launch_bounds__(256) void k_up_fwd_cf(const void *__restrict__ x, void *__restrict__ y, int64_t BC, int H,
                                                   int W, int Ho, int Wo) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= BC * Ho * Wo) return;
  const int ox = (int)(i % Wo), oy = (int)((i / Wo) % Ho);
  const int64_t bc = i / ((int64_t)Wo * Ho);
  const float sy = up_scale(H, Ho) * (float)oy, sx = up_scale(W, Wo) * (float)ox;
  const int y0 = (int)sy, x0 = (int)sx;
  const int yp = y0 < H - 1 ? 1 : 0, xp = x0 < W - 1 ? 1 : 0;
  const float h1 = sy - (float)y0, h0 = 1.0f - h1, w1 = sx - (float)x0, w0 = 1.0f - w1;
  const int64_t base = (bc * H + y0) * W + x0;
  const float a00 = load1<BF16>(x, base), a01 = load1<BF16>(x, base + xp), a10 = load1<BF16>(x, base + (int64_t)yp * W),
              a11 = load1<BF16>(x, base + (int64_t)yp * W + xp);
  store1<BF16>(y, i, h0 * (w0 * a00 + w1 * a01) + h1 * (w0 * a10 + w1 * a11));
}

template <bool BF16>
__global__ __launch_bounds__(256) void k_up_bwd_cf(const void *__restrict__ gy, void *__restrict__ gx, int64_t BC, int H,
                                                   int W, int Ho, int Wo) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= BC * H * W) return;
  const int ix = (int)(i % W), iy = (int)((i / W) % H);
  const int64_t bc = i / ((int64_t)W * H);
  const float ssy = up_scale(H, Ho), ssx = up_scale(W, Wo);
  int oy_lo = 0, oy_hi = Ho - 1, ox_lo = 0, ox_hi = Wo - 1;
  if (ssy > 0.0f) { oy_lo = max(0, (int)floorf((float)(iy - 1) / ssy)); oy_hi = min(Ho - 1, (int)ceilf((float)(iy + 1) / ssy)); }
  if (ssx > 0.0f) { ox_lo = max(0, (int)floorf((float)(ix - 1) / ssx)); ox_hi = min(Wo - 1, (int)ceilf((float)(ix + 1) / ssx)); }
  float acc = 0.0f;
  for (int oy = oy_lo; oy <= oy_hi; ++oy) {
    const float sy = ssy * (float)oy;
    const int y0 = (int)sy;
    const int yp = y0 < H - 1 ? 1 : 0;
    if (!(y0 == iy || y0 + yp == iy)) continue;
    const float h1 = sy - (float)y0, h0 = 1.0f - h1;
    float wy = 0.0f;
    if (y0 == iy) wy += h0;
    if (y0 + yp == iy) wy += h1;
    for (int ox = ox_lo; ox <= ox_hi; ++ox) {
      const float sx = ssx * (float)ox;
      const int x0 = (int)sx;
      const int xp = x0 < W - 1 ? 1 : 0;
      if (!(x0 == ix || x0 + xp == ix)) continue;
      const float w1 = sx - (float)x0, w0 = 1.0f - w1;
      float wx = 0.0f;
      if (x0 == ix) wx += w0;
      if (x0 + xp == ix) wx += w1;
      acc += (wy * wx) * load1<BF16>(gy, (bc * Ho + oy) * Wo + ox);
    }
  }
  store1<BF16>(gx, i, acc);
}

// PReLU with ONE slope: y = x > 0 ? x : a x
template <bool BF16>
__global__ __launch_bounds__(256) void k_prelu_fwd(const void *__restrict__ x, const float *__restrict__ slope,
                                                   void *__restrict__ y, int64_t n8) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n8) return;
  const float a = slope[0];
  float v[8];
  load8<BF16>(x, 8 * i, v);
#pragma unroll
  for (int k = 0; k < 8; ++k) v[k] = v[k] > 0.0f ? v[k] : a * v[k];
  store8<BF16>(y, 8 * i, v);
}

// BatchNorm in inference mode (+ residual add) (+ ReLU) of ResNet18Extractor (models/resnet.py:44: the statistics are
// never updated): y = relu?((x - mean) * (weight * invstd) + bias (+ identity)), invstd = 1 / sqrt(var + eps) -- torch's
// own operation order.  torch: one elementwise launch for the normalisation, one for the add, one for the ReLU.
// CL: channels-last (8 consecutive elements = 8 consecutive channels); otherwise NCHW with HW % 8 == 0 (8 consecutive
// elements = one channel).
template <bool BF16, bool CL>
__global__ __launch_bounds__(256) void k_bn_act(const void *__restrict__ x, const void *__restrict__ identity,
                                                const float *__restrict__ mean, const float *__restrict__ var,
                                                const float *__restrict__ weight, const float *__restrict__ bias,
                                                float eps, void *__restrict__ y, int64_t n8, int C, int64_t HW, int relu) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n8) return;
  float v[8], r[8];
  load8<BF16>(x, 8 * i, v);
  if (identity) load8<BF16>(identity, 8 * i, r);
  const int c0 = CL ? (int)((8 * i) % C) : (int)(((8 * i) / HW) % C);
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    const int c = CL ? c0 + k : c0;
    const float invstd = 1.0f / sqrtf(var[c] + eps);
    float o = (v[k] - mean[c]) * (weight[c] * invstd) + bias[c];
    if (identity) o += r[k];
    v[k] = relu ? (o > 0.0f ? o : 0.0f) : o;
  }
  store8<BF16>(y, 8 * i, v);
}

// The channels-last form at ResNet widths (C / 8 = G in {8, 16, 32, 64} channel groups): a lane keeps ONE channel group
// (its 4 x 8 parameters arrive as eight 16-byte loads -- read per element they would be 32 dword loads for 8 elements
// of data, and the kernel would be bound by its own parameter traffic) and walks `ppt` pixels with it.
template <bool BF16>
__global__ __launch_bounds__(256) void k_bn_act_cl(const void *__restrict__ x, const void *__restrict__ identity,
                                                   const float *__restrict__ mean, const float *__restrict__ var,
                                                   const float *__restrict__ weight, const float *__restrict__ bias,
                                                   float eps, void *__restrict__ y, int64_t npix, int G, int ppt, int relu) {
  const int cg = threadIdx.x % G, lp = threadIdx.x / G, ppb = 256 / G;  // pixels per block pass
  float mu[8], sc[8], sh[8], vr[8];
  load8<false>(mean, 8 * cg, mu);
  load8<false>(var, 8 * cg, vr);
  load8<false>(weight, 8 * cg, sc);
  load8<false>(bias, 8 * cg, sh);
#pragma unroll
  for (int k = 0; k < 8; ++k) sc[k] = sc[k] * (1.0f / sqrtf(vr[k] + eps));
  const int64_t p0 = (int64_t)blockIdx.x * ppb * ppt + lp;
  for (int u = 0; u < ppt; ++u) {
    const int64_t pix = p0 + (int64_t)u * ppb;
    if (pix >= npix) break;
    const int64_t e = (pix * G + cg) * 8;
    float v[8], r[8];
    load8<BF16>(x, e, v);
    if (identity) load8<BF16>(identity, e, r);
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      float o = (v[k] - mu[k]) * sc[k] + sh[k];
      if (identity) o += r[k];
      v[k] = relu ? (o > 0.0f ? o : 0.0f) : o;
    }
    store8<BF16>(y, e, v);
  }
}

// ResNet18Extractor's input normalisation (models/resnet.py:33-36): (rgb / 255 - mean) / std per channel, torch's
// operation order, on the [B, H, W, 3] image as it arrives (uint8 or float32) -> float32 in the same (channels-last)
// memory order.  torch: cast, divide, subtract, divide = four launches.
template <bool U8>
__global__ __launch_bounds__(256) void k_rgb_norm(const void *__restrict__ rgb, float m0, float m1, float m2, float s0,
                                                  float s1, float s2, float *__restrict__ out, int64_t npix) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= npix) return;
  float r, g, b;
  if (U8) {
    const uint8_t *p = reinterpret_cast<const uint8_t *>(rgb) + 3 * i;
    r = (float)p[0]; g = (float)p[1]; b = (float)p[2];
  } else {
    const float *p = reinterpret_cast<const float *>(rgb) + 3 * i;
    r = p[0]; g = p[1]; b = p[2];
  }
  out[3 * i] = (r / 255.0f - m0) / s0;
  out[3 * i + 1] = (g / 255.0f - m1) / s1;
  out[3 * i + 2] = (b / 255.0f - m2) / s2;
}

// dx = dy (x > 0 ? 1 : a); partial[block] = sum over the block's elements with x <= 0 of dy x
constexpr int kPreluPerThread = 4;  // 8-element chunks per lane: 8192 elements per workgroup

template <bool BF16>
__global__ __launch_bounds__(256) void k_prelu_bwd(const void *__restrict__ x, const void *__restrict__ dy,
                                                   const float *__restrict__ slope, void *__restrict__ dx,
                                                   float *__restrict__ partial, int64_t n8) {
  __shared__ float s_red[4];
  const float a = slope[0];
  float part = 0.0f;
#pragma unroll
  for (int u = 0; u < kPreluPerThread; ++u) {
    const int64_t i = ((int64_t)blockIdx.x * kPreluPerThread + u) * blockDim.x + threadIdx.x;
    if (i >= n8) continue;
    float v[8], g[8];
    load8<BF16>(x, 8 * i, v);
    load8<BF16>(dy, 8 * i, g);
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const bool pos = v[k] > 0.0f;
      part += pos ? 0.0f : g[k] * v[k];
      g[k] = pos ? g[k] : a * g[k];
    }
    store8<BF16>(dx, 8 * i, g);
  }
  part = mf::wave_sum(part);
  if ((threadIdx.x & 63) == 0) s_red[threadIdx.x >> 6] = part;
  __syncthreads();
  if (threadIdx.x == 0) partial[blockIdx.x] = (s_red[0] + s_red[1]) + (s_red[2] + s_red[3]);
}

__global__ __launch_bounds__(256) void k_prelu_finish(const float *__restrict__ partial, int n, float *__restrict__ da) {
  __shared__ float s_red[4];
  float s = 0.0f;
  for (int i = threadIdx.x; i < n; i += 256) s += partial[i];  // fixed assignment of partials to lanes
  s = mf::wave_sum(s);
  if ((threadIdx.x & 63) == 0) s_red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) da[0] = (s_red[0] + s_red[1]) + (s_red[2] + s_red[3]);
}

int bad2d(const char *msg) {
  mf::set_last_error(hipErrorInvalidValue, msg);
  return -(int)hipErrorInvalidValue;
}

}  // namespace

/* y [B, Ho, Wo, C] = bilinear resize (align_corners) of x [B, H, W, C], channels-last; bf16 != 0: bfloat16 tensors,
 * else float32.  C % 8 == 0, 16-byte aligned. */
extern "C" int mf_upsample_bilinear_cl_fwd(const void *x, void *y, int32_t B, int32_t H, int32_t W, int32_t Ho,
                                           int32_t Wo, int32_t C, int32_t bf16, mfStream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if ((int64_t)B * Ho * Wo * C == 0) return 0;
  if (C % 8 || H < 1 || W < 1 || (((uintptr_t)x | (uintptr_t)y) & 15)) return bad2d("upsample_bilinear_cl: C % 8 == 0, aligned");
  const int64_t total = (int64_t)B * Ho * Wo * (C / 8);
  const unsigned nb = (unsigned)((total + 255) / 256);
  if (bf16) hipLaunchKernelGGL(k_up_fwd<true>, dim3(nb), dim3(256), 0, stream, x, y, B, H, W, Ho, Wo, C / 8);
  else hipLaunchKernelGGL(k_up_fwd<false>, dim3(nb), dim3(256), 0, stream, x, y, B, H, W, Ho, Wo, C / 8);
  return mf::check_launch("mf_upsample_bilinear_cl_fwd");
}

extern "C" int mf_upsample_bilinear_cl_bwd(const void *gy, void *gx, int32_t B, int32_t H, int32_t W, int32_t Ho,
                                           int32_t Wo, int32_t C, int32_t bf16, mfStream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if ((int64_t)B * H * W * C == 0) return 0;
  if (C % 8 || (((uintptr_t)gx | (uintptr_t)gy) & 15)) return bad2d("upsample_bilinear_cl: C % 8 == 0, aligned");
  const int64_t total = (int64_t)B * H * W * (C / 8);
  const unsigned nb = (unsigned)((total + 255) / 256);
  if (C % 64 == 0 && (int64_t)H * W <= 64 && (int64_t)Ho * Wo >= 16 * (int64_t)H * W) {  // small input, large footprints
    const dim3 grid((unsigned)(H * W), (unsigned)(C / 64), (unsigned)B);
    if (bf16) hipLaunchKernelGGL(k_up_bwd_small<true>, grid, dim3(512), 0, stream, gy, gx, H, W, Ho, Wo, C);
    else hipLaunchKernelGGL(k_up_bwd_small<false>, grid, dim3(512), 0, stream, gy, gx, H, W, Ho, Wo, C);
    return mf::check_launch("mf_upsample_bilinear_cl_bwd");
  }
  if (bf16 && C % 64 == 0 && H >= kUpT && W >= kUpT) {
    // the LDS-tiled gather when the worst tile's output patch fits 64 KB (a x2 resize: 20 x 20 x 128 B = 51 KB)
    const float sy = Ho > 1 ? (float)(H - 1) / (float)(Ho - 1) : 0.0f, sx = Wo > 1 ? (float)(W - 1) / (float)(Wo - 1) : 0.0f;
    const int rows = sy > 0.0f ? (int)ceilf((float)(kUpT + 1) / sy) + 3 : Ho, cols = sx > 0.0f ? (int)ceilf((float)(kUpT + 1) / sx) + 3 : Wo;
    const size_t lds = (size_t)rows * cols * 128;
    if (lds <= 64 * 1024) {
      const int tx = (W + kUpT - 1) / kUpT, ty = (H + kUpT - 1) / kUpT;
      hipLaunchKernelGGL(k_up_bwd_tile_bf16, dim3((unsigned)(tx * ty), (unsigned)(C / 64), (unsigned)B), dim3(512), lds, stream,
                         (const uint16_t *)gy, (uint16_t *)gx, H, W, Ho, Wo, C, tx);
      return mf::check_launch("mf_upsample_bilinear_cl_bwd");
    }
  }
  if (bf16) hipLaunchKernelGGL(k_up_bwd<true>, dim3(nb), dim3(256), 0, stream, gy, gx, B, H, W, Ho, Wo, C / 8);
  else hipLaunchKernelGGL(k_up_bwd<false>, dim3(nb), dim3(256), 0, stream, gy, gx, B, H, W, Ho, Wo, C / 8);
  return mf::check_launch("mf_upsample_bilinear_cl_bwd");
}


/* the same for channels-first tensors x [B*C, H, W] -> y [B*C, Ho, Wo] (no alignment or channel-count requirement) */
extern "C" int mf_upsample_bilinear_cf_fwd(const void *x, void *y, int64_t BC, int32_t H, int32_t W, int32_t Ho,
                                           int32_t Wo, int32_t bf16, mfStream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  const int64_t total = BC * Ho * Wo;
  if (total == 0) return 0;
  const unsigned nb = (unsigned)((total + 255) / 256);
  if (bf16) hipLaunchKernelGGL(k_up_fwd_cf<true>, dim3(nb), dim3(256), 0, stream, x, y, BC, H, W, Ho, Wo);
  else hipLaunchKernelGGL(k_up_fwd_cf<false>, dim3(nb), dim3(256), 0, stream, x, y, BC, H, W, Ho, Wo);
  return mf::check_launch("mf_upsample_bilinear_cf_fwd");
}

extern "C" int mf_upsample_bilinear_cf_bwd(const void *gy, void *gx, int64_t BC, int32_t H, int32_t W, int32_t Ho,
                                           int32_t Wo, int32_t bf16, mfStream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  const int64_t total = BC * H * W;
  if (total == 0) return 0;
  const unsigned nb = (unsigned)((total + 255) / 256);
  if (bf16) hipLaunchKernelGGL(k_up_bwd_cf<true>, dim3(nb), dim3(256), 0, stream, gy, gx, BC, H, W, Ho, Wo);
  else hipLaunchKernelGGL(k_up_bwd_cf<false>, dim3(nb), dim3(256), 0, stream, gy, gx, BC, H, W, Ho, Wo);
  return mf::check_launch("mf_upsample_bilinear_cf_bwd");
}

/* PReLU with one slope (device pointer, fp32) over n elements (n % 8 == 0) */
extern "C" int mf_prelu_fwd(const void *x, const float *slope, void *y, int64_t n, int32_t bf16, mfStream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (n <= 0) return 0;
  if (n % 8 || (((uintptr_t)x | (uintptr_t)y) & 15)) return bad2d("prelu: n % 8 == 0, aligned");
  const unsigned nb = (unsigned)((n / 8 + 255) / 256);
  if (bf16) hipLaunchKernelGGL(k_prelu_fwd<true>, dim3(nb), dim3(256), 0, stream, x, slope, y, n / 8);
  else hipLaunchKernelGGL(k_prelu_fwd<false>, dim3(nb), dim3(256), 0, stream, x, slope, y, n / 8);
  return mf::check_launch("mf_prelu_fwd");
}

extern "C" int64_t mf_prelu_bwd_workspace_floats(int64_t n) { return (n / 8 + 256 * kPreluPerThread - 1) / (256 * kPreluPerThread); }

/* dx = dy * (x > 0 ? 1 : slope); dslope[0] = sum_{x <= 0} dy x (fp32; block partials in ws, added in block order) */
extern "C" int mf_prelu_bwd(const void *x, const void *dy, const float *slope, void *dx, float *dslope, float *ws,
                            int64_t n, int32_t bf16, mfStream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (n <= 0) return 0;
  if (n % 8 || !ws || (((uintptr_t)x | (uintptr_t)dy | (uintptr_t)dx) & 15)) return bad2d("prelu backward: n % 8 == 0, aligned, workspace");
  const int64_t nblk = mf_prelu_bwd_workspace_floats(n);
  if (bf16) hipLaunchKernelGGL(k_prelu_bwd<true>, dim3((unsigned)nblk), dim3(256), 0, stream, x, dy, slope, dx, ws, n / 8);
  else hipLaunchKernelGGL(k_prelu_bwd<false>, dim3((unsigned)nblk), dim3(256), 0, stream, x, dy, slope, dx, ws, n / 8);
  hipLaunchKernelGGL(k_prelu_finish, dim3(1), dim3(256), 0, stream, (const float *)ws, (int)nblk, dslope);
  return mf::check_launch("mf_prelu_bwd");
}

/* BatchNorm (inference statistics) + optional residual add + optional ReLU over a dense [B, C, H, W] tensor, NCHW
 * (channels_last = 0; H * W % 8 == 0) or channels-last (C % 8 == 0); fp32 or bf16 activations, fp32 parameters. */
extern "C" int mf_bn_act_fwd(const void *x, const void *identity, const float *mean, const float *var, const float *weight,
                             const float *bias, float eps, void *y, int64_t n, int32_t C, int64_t HW, int32_t channels_last,
                             int32_t relu, int32_t bf16, mfStream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (n <= 0) return 0;
  if (n % 8 || C <= 0 || HW <= 0 || (channels_last ? C % 8 : HW % 8) ||
      (((uintptr_t)x | (uintptr_t)y | (uintptr_t)identity) & 15))
    return bad2d("bn_act: n % 8 == 0, 8 | C (channels-last) or 8 | H W (NCHW), aligned");
  const int G = C / 8;
  if (channels_last && G <= 256 && 256 % G == 0 &&
      ((((uintptr_t)mean | (uintptr_t)var | (uintptr_t)weight | (uintptr_t)bias) & 15) == 0)) {
    const int64_t npix = n / C, ppb = 256 / G;
    int ppt = 1;  // pixels per lane: as many as still leave >= 2048 workgroups (8 per CU), at most 4
    while (ppt < 4 && npix / (ppb * ppt * 2) >= 2048) ppt *= 2;
    const unsigned nbp = (unsigned)((npix + ppb * ppt - 1) / (ppb * ppt));
    if (bf16) hipLaunchKernelGGL(k_bn_act_cl<true>, dim3(nbp), dim3(256), 0, stream, x, identity, mean, var, weight, bias,
                                 eps, y, npix, G, ppt, relu);
    else hipLaunchKernelGGL(k_bn_act_cl<false>, dim3(nbp), dim3(256), 0, stream, x, identity, mean, var, weight, bias, eps,
                            y, npix, G, ppt, relu);
    return mf::check_launch("mf_bn_act_fwd");
  }
  const unsigned nb = (unsigned)((n / 8 + 255) / 256);
#define MF_BN_LAUNCH(BF, CL_)                                                                                   \
  hipLaunchKernelGGL((k_bn_act<BF, CL_>), dim3(nb), dim3(256), 0, stream, x, identity, mean, var, weight, bias, eps, y, \
                     n / 8, C, HW, relu)
  if (bf16) { if (channels_last) MF_BN_LAUNCH(true, true); else MF_BN_LAUNCH(true, false); }
  else { if (channels_last) MF_BN_LAUNCH(false, true); else MF_BN_LAUNCH(false, false); }
#undef MF_BN_LAUNCH
  return mf::check_launch("mf_bn_act_fwd");
}

/* (rgb / 255 - mean) / std of a [B, H, W, 3] image, uint8 (u8 = 1) or float32, -> float32 [B, H, W, 3]. */
extern "C" int mf_rgb_normalize(const void *rgb, int32_t u8, const float *mean3, const float *std3, float *out,
                                int64_t npix, mfStream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (npix <= 0) return 0;
  const unsigned nb = (unsigned)((npix + 255) / 256);
  if (u8) hipLaunchKernelGGL(k_rgb_norm<true>, dim3(nb), dim3(256), 0, stream, rgb, mean3[0], mean3[1], mean3[2], std3[0],
                             std3[1], std3[2], out, npix);
  else hipLaunchKernelGGL(k_rgb_norm<false>, dim3(nb), dim3(256), 0, stream, rgb, mean3[0], mean3[1], mean3[2], std3[0],
                          std3[1], std3[2], out, npix);
  return mf::check_launch("mf_rgb_normalize");
}
