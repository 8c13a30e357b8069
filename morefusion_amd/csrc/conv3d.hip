// Dense 3-D convolution, kernel 4 / stride 2 / pad 1, as an fp32-MFMA implicit GEMM for gfx950.
//
// Reference: conv4 of the pose network, `L.Convolution3D(256, 512, 4, 2, pad=1)` + ReLU
// (contrib/singleview_3d/models/model.py:74,139: cuDNN, 8.6 GFLOP per object = half of the
// volumetric part's arithmetic), and the dense 16 occupancy channels of conv3
// (`L.Convolution3D(None, 256, 4, 2, pad=1)`, model.py:73,128: 2.1 GFLOP per object).
//
// MI355X design
//   * channels-last tensors: x [B][D^3][Cin], out [B][(D/2)^3][Cout].  An im2col row of one tap
//     is then Cin CONTIGUOUS floats, the GEMM's K index is (tap, cin) and nothing is ever
//     gathered element-wise; the trilinear sampler downstream reads whole voxels (Cout floats).
//   * GEMM  out[m][n] = sum_k A[m][k] * Wt[n][k],  m = (b, ox, oy, oz), k = tap * Cin + cin;
//     weights pre-packed k-contiguous ([Cout][64][Cin]).  Padding taps load zeros.
//   * 128 x 128 x 32 tile per 256-lane workgroup, each wave a 64 x 64 corner as 2 x 2
//     v_mfma_f32_32x32x2_f32 accumulators (exact fp32, 64 cycles each: 4096 MFMA cycles per
//     K-tile and wave against ~150 cycles of LDS traffic).  Both operands sit in LDS k-contiguous
//     with a row pitch of 36 floats: every fragment read is ONE conflict-free ds_read_b128 (lane
//     half h takes k = 8 kk + 4 h .. + 3 -- the MFMA's k pair (0, 1) becomes (j, 4 + j), the
//     same permutation for A and B).
//   * register-staged double buffering: the global loads of K-tile t + 1 are in flight during
//     the 64 MFMAs of tile t; two LDS buffers -> one barrier per K-tile; 2 workgroups per CU
//     (2 x 72 KB LDS) fill each other's barrier bubbles.
//   * split-K over taps when B is small (grid >= 512 workgroups): slab s holds the partial sum of
//     its taps; k_conv_finish adds the slabs in slab order + bias + ReLU (deterministic, no
//     atomics).  The workgroup order is XCD-aware: an XCD's 64 resident workgroups share one K
//     range and two N tiles, so the weight slice they stream is read from HBM once per XCD.
#include "mf_common.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int kBM = 128, kBN = 128, kBK = 32;
constexpr int kPitch = kBK + 4;                         // LDS row pitch (floats)
constexpr int kTileFloats = (kBM + kBN) * kPitch;       // one buffer: A rows then B rows
constexpr int kConvLds = 2 * kTileFloats * (int)sizeof(float);  // 73,728 B

struct ConvArgs {
  const float *x;    // [B][D^3][Cin]
  const float *wt;   // [Cout][64][Cin]
  const float *bias; // [Cout] or null       (used when S == 1)
  const float *add;  // [M][Cout] or null    (used when S == 1)
  float *out;        // S == 1: [M][Cout];  S > 1: slabs [S][M][Cout]
  int B, D, Cin, cin_log2, Cout, S, relu;
};

// One float4 of an A row: x[b][voxel(o, tap)][c .. c + 3], zeros for a padding tap.  ``base`` is the
// element offset of tap (0,0,0), channel 0 of this row (may be negative: pad = 1); ``mask`` has bit
// kx | 4 + ky | 8 + kz set when that tap coordinate is inside the grid (bit 12: the row exists).
__device__ __forceinline__ float4 load_a(const float *__restrict__ x, int base, int mask, int tapoff, int tapbits) {
  const bool ok = (mask & tapbits) == tapbits;
  // a padding tap reads x[0..3] (always mapped, cache-resident) and selects zeros
  const float4 r = *reinterpret_cast<const float4 *>(x + (ok ? base + tapoff : 0));
  return ok ? r : make_float4(0.0f, 0.0f, 0.0f, 0.0f);
}

__global__ __launch_bounds__(256, 2) void k_conv3d_k4s2_mfma(ConvArgs a) {
  MF_DYN_LDS(float, s_mem);
  const int Do = a.D / 2, Vo = Do * Do * Do;
  const int M = a.B * Vo, N = a.Cout, K = 64 * a.Cin;
  const int tiles_m = (M + kBM - 1) / kBM, tiles_n = N / kBN;
  // XCD-aware order: hardware block b runs on XCD b % 8; give every XCD a CONTIGUOUS range of the
  // logical order (split slowest, then N tile, M tile fastest)
  const int G = gridDim.x;
  int L = blockIdx.x;
  if ((G & 7) == 0) L = (blockIdx.x & 7) * (G >> 3) + (blockIdx.x >> 3);
  const int split = L / (tiles_m * tiles_n);
  const int rem = L - split * (tiles_m * tiles_n);
  const int m0 = (rem % tiles_m) * kBM, n0 = (rem / tiles_m) * kBN;
  const int T = (K / kBK) / a.S;           // K-tiles of this split
  const int kt0 = split * T;

  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int wm = wave & 1, wn = wave >> 1;
  const int lrow = lane & 31, lhalf = lane >> 5;
  const int chunk = tid & 7, r0 = tid >> 3;  // this lane stages rows r0 + 32 i, floats 4 chunk ..

  // per staged A row: element offset of tap (0,0,0) and the in-grid masks of the 4 + 4 + 4 tap coordinates
  int base[4], mask[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int m = m0 + r0 + 32 * i;
    const bool row_ok = m < M;
    const int mm = row_ok ? m : 0;
    const int b = mm / Vo, o = mm - b * Vo;
    const int ox = o / (Do * Do), oy = (o / Do) % Do, oz = o % Do;
    const int x0 = 2 * ox - 1, y0 = 2 * oy - 1, z0 = 2 * oz - 1;
    base[i] = (((b * a.D + x0) * a.D + y0) * a.D + z0) * a.Cin;
    int mk = row_ok ? 1 << 12 : 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      mk |= ((unsigned)(x0 + k) < (unsigned)a.D ? 1 : 0) << k;
      mk |= ((unsigned)(y0 + k) < (unsigned)a.D ? 1 : 0) << (4 + k);
      mk |= ((unsigned)(z0 + k) < (unsigned)a.D ? 1 : 0) << (8 + k);
    }
    mask[i] = mk;
  }
  const float *wrow[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) wrow[i] = a.wt + (n0 + r0 + 32 * i) * K + 4 * chunk;

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.0f;

  // K-tile kt -> registers (A: 4 rows x 16 B, B: 4 rows x 16 B per lane)
#define MF_CONV_FETCH(kt_)                                                                        \
  {                                                                                               \
    const int kt__ = (kt_);                                                                       \
    const int kglob = kt__ * kBK + 4 * chunk;                                                     \
    const int tap = kglob >> a.cin_log2, kx = tap >> 4, ky = (tap >> 2) & 3, kz = tap & 3;        \
    const int tapoff = ((kx * a.D + ky) * a.D + kz) * a.Cin + (kglob & (a.Cin - 1));              \
    const int tapbits = (1 << kx) | (16 << ky) | (256 << kz) | (1 << 12);                         \
    ra0 = load_a(a.x, base[0], mask[0], tapoff, tapbits);                                         \
    ra1 = load_a(a.x, base[1], mask[1], tapoff, tapbits);                                         \
    ra2 = load_a(a.x, base[2], mask[2], tapoff, tapbits);                                         \
    ra3 = load_a(a.x, base[3], mask[3], tapoff, tapbits);                                         \
    rb0 = *reinterpret_cast<const float4 *>(wrow[0] + kt__ * kBK);                       \
    rb1 = *reinterpret_cast<const float4 *>(wrow[1] + kt__ * kBK);                       \
    rb2 = *reinterpret_cast<const float4 *>(wrow[2] + kt__ * kBK);                       \
    rb3 = *reinterpret_cast<const float4 *>(wrow[3] + kt__ * kBK);                       \
  }
#define MF_CONV_STASH(buf_)                                                                       \
  {                                                                                               \
    float *As_ = s_mem + (buf_) * kTileFloats + r0 * kPitch + 4 * chunk;                          \
    float *Bs_ = As_ + kBM * kPitch;                                                              \
    *reinterpret_cast<float4 *>(As_) = ra0;                                                       \
    *reinterpret_cast<float4 *>(As_ + 32 * kPitch) = ra1;                                         \
    *reinterpret_cast<float4 *>(As_ + 64 * kPitch) = ra2;                                         \
    *reinterpret_cast<float4 *>(As_ + 96 * kPitch) = ra3;                                         \
    *reinterpret_cast<float4 *>(Bs_) = rb0;                                                       \
    *reinterpret_cast<float4 *>(Bs_ + 32 * kPitch) = rb1;                                         \
    *reinterpret_cast<float4 *>(Bs_ + 64 * kPitch) = rb2;                                         \
    *reinterpret_cast<float4 *>(Bs_ + 96 * kPitch) = rb3;                                         \
  }
  float4 ra0, ra1, ra2, ra3, rb0, rb1, rb2, rb3;
  MF_CONV_FETCH(kt0);
  MF_CONV_STASH(0);
  __syncthreads();
  for (int t = 0; t < T; ++t) {
    // tile t + 1 in flight during this tile's MFMAs (the last iteration re-fetches its own tile:
    // branch-free loop body, the extra stash lands in the buffer nobody reads again)
    MF_CONV_FETCH(kt0 + (t + 1 < T ? t + 1 : t));
    // keep the eight loads HERE (hipcc otherwise sinks the weight loads down to their ds_write and
    // exposes their latency in front of the barrier)
    asm volatile("" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    const float *As = s_mem + (t & 1) * kTileFloats + (wm * 64 + lrow) * kPitch + 4 * lhalf;
    const float *Bs = s_mem + (t & 1) * kTileFloats + (kBM + wn * 64 + lrow) * kPitch + 4 * lhalf;
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      const float4 a0 = *reinterpret_cast<const float4 *>(As + 8 * kk);
      const float4 a1 = *reinterpret_cast<const float4 *>(As + 32 * kPitch + 8 * kk);
      const float4 b0 = *reinterpret_cast<const float4 *>(Bs + 8 * kk);
      const float4 b1 = *reinterpret_cast<const float4 *>(Bs + 32 * kPitch + 8 * kk);
#define MF_CONV_STEP(c_)                                                                          \
      acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.c_, b0.c_, acc[0][0], 0, 0, 0);         \
      acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.c_, b1.c_, acc[0][1], 0, 0, 0);         \
      acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.c_, b0.c_, acc[1][0], 0, 0, 0);         \
      acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.c_, b1.c_, acc[1][1], 0, 0, 0);
      MF_CONV_STEP(x)
      MF_CONV_STEP(y)
      MF_CONV_STEP(z)
      MF_CONV_STEP(w)
#undef MF_CONV_STEP
    }
    MF_CONV_STASH((t + 1) & 1);
    __syncthreads();
  }
#undef MF_CONV_FETCH
#undef MF_CONV_STASH

  // epilogue.  C fragment of a 32x32 block: col = lane & 31, row = (e & 3) + 8 (e >> 2) + 4 (lane >> 5)
  float *dst = a.out + (a.S > 1 ? (int64_t)split * M * N : 0);
#pragma unroll
  for (int mi = 0; mi < 2; ++mi)
#pragma unroll
    for (int ni = 0; ni < 2; ++ni) {
      const int n = n0 + wn * 64 + ni * 32 + lrow;
      const float bn = (a.S == 1 && a.bias) ? a.bias[n] : 0.0f;
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int m = m0 + wm * 64 + mi * 32 + (e & 3) + 8 * (e >> 2) + 4 * lhalf;
        if (m >= M) continue;
        float v = acc[mi][ni][e];
        if (a.S == 1) {
          v += bn;
          if (a.add) v += a.add[(int64_t)m * N + n];
          if (a.relu) v = v > 0.0f ? v : 0.0f;
        }
        dst[(int64_t)m * N + n] = v;
      }
    }
}

// out[m][n] = act( sum_s slab[s][m][n] (increasing s) + bias[n] + add[m][n] )
__global__ __launch_bounds__(256) void k_conv_finish(const float *__restrict__ slabs, const float *__restrict__ bias,
                                                     const float *__restrict__ add, float *__restrict__ out,
                                                     int64_t MN, int N, int S, int relu) {
  const int64_t i4 = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i4 * 4 >= MN) return;
  const float4 *s4 = reinterpret_cast<const float4 *>(slabs);
  float4 v = s4[i4];
  for (int s = 1; s < S; ++s) {
    const float4 w = s4[(int64_t)s * (MN / 4) + i4];
    v.x += w.x; v.y += w.y; v.z += w.z; v.w += w.w;
  }
  const int n = (int)((i4 * 4) % N);
  if (bias) {
    const float4 b = *reinterpret_cast<const float4 *>(bias + n);
    v.x += b.x; v.y += b.y; v.z += b.z; v.w += b.w;
  }
  if (add) {
    const float4 w = reinterpret_cast<const float4 *>(add)[i4];
    v.x += w.x; v.y += w.y; v.z += w.z; v.w += w.w;
  }
  if (relu) {
    v.x = v.x > 0.0f ? v.x : 0.0f; v.y = v.y > 0.0f ? v.y : 0.0f;
    v.z = v.z > 0.0f ? v.z : 0.0f; v.w = v.w > 0.0f ? v.w : 0.0f;
  }
  reinterpret_cast<float4 *>(out)[i4] = v;
}

// wt[co][tap][ci] = W[co][ci][kx][ky][kz]   (torch / Chainer ConvolutionND layout -> k-contiguous)
__global__ void k_conv_pack(const float *__restrict__ W, int Cout, int Cin, int w_cin, int c_off,
                            float *__restrict__ wt) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t total = (int64_t)Cout * 64 * Cin;
  if (i >= total) return;
  const int ci = (int)(i % Cin), tap = (int)((i / Cin) % 64), co = (int)(i / ((int64_t)64 * Cin));
  wt[i] = W[((int64_t)co * w_cin + c_off + ci) * 64 + tap];
}

// channels-first [B][C][V] -> channels-last [B][V][C] through a 32 x 32 LDS tile
__global__ __launch_bounds__(256) void k_to_channels_last(const float *__restrict__ src, float *__restrict__ dst,
                                                          int C, int V) {
  __shared__ float tile[32][33];
  const int b = blockIdx.z, v0 = blockIdx.x * 32, c0 = blockIdx.y * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  for (int j = ty; j < 32; j += 8)
    if (c0 + j < C && v0 + tx < V) tile[j][tx] = src[((int64_t)b * C + c0 + j) * V + v0 + tx];
  __syncthreads();
  for (int j = ty; j < 32; j += 8)
    if (v0 + j < V && c0 + tx < C) dst[((int64_t)b * V + v0 + j) * C + c0 + tx] = tile[tx][j];
}

// ---- occupancy branch: conv1_occ (1 -> 8, k3 p1) and conv2_occ (8 -> 16, k3 dilation 2 p2), + ReLU ----
// Reference: model.py:69-72,120-124 (two cuDNN Convolution3D on the 32^3 no-entry grid).  0.24 GFLOP per
// object: VALU work.  One lane per output voxel, all output channels in registers, channels-last
// in / out (the 16-channel result feeds conv3's implicit GEMM as it is); the weights are uniform ->
// scalar loads, every v_fma takes its weight from an SGPR.  Packed weights: w[tap][ci][co].
template <int CI, int CO, int DIL>
__global__ __launch_bounds__(256) void k_occ_conv3(const float *__restrict__ x, const float *__restrict__ w,
                                                   const float *__restrict__ bias, float *__restrict__ out,
                                                   int B, int D) {
  const int64_t total = (int64_t)B * D * D * D;
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const int iz = (int)(i % D), iy = (int)((i / D) % D), ix = (int)((i / ((int64_t)D * D)) % D);
  const int64_t b = i / ((int64_t)D * D * D);
  float acc[CO];
#pragma unroll
  for (int co = 0; co < CO; ++co) acc[co] = bias[co];
#pragma unroll 1
  for (int kx = 0; kx < 3; ++kx) {
    const int jx = ix + (kx - 1) * DIL;
#pragma unroll 1
    for (int ky = 0; ky < 3; ++ky) {
      const int jy = iy + (ky - 1) * DIL;
#pragma unroll
      for (int kz = 0; kz < 3; ++kz) {
        const int jz = iz + (kz - 1) * DIL;
        const bool ok = (unsigned)jx < (unsigned)D && (unsigned)jy < (unsigned)D && (unsigned)jz < (unsigned)D;
        const float *src = x + (ok ? (((b * D + jx) * D + jy) * D + jz) * CI : 0);
        float in[CI];
        if (CI % 4 == 0) {
#pragma unroll
          for (int q = 0; q < CI / 4; ++q) {
            const float4 v = reinterpret_cast<const float4 *>(src)[q];
            in[4 * q] = v.x; in[4 * q + 1] = v.y; in[4 * q + 2] = v.z; in[4 * q + 3] = v.w;
          }
        } else {
#pragma unroll
          for (int ci = 0; ci < CI; ++ci) in[ci] = src[ci];
        }
        const float *wt = w + ((kx * 3 + ky) * 3 + kz) * CI * CO;
#pragma unroll
        for (int ci = 0; ci < CI; ++ci) {
          const float v = ok ? in[ci] : 0.0f;
#pragma unroll
          for (int co = 0; co < CO; ++co) acc[co] = fmaf(v, wt[ci * CO + co], acc[co]);
        }
      }
    }
  }
  float *dst = out + i * CO;
#pragma unroll
  for (int q = 0; q < CO / 4; ++q)
    reinterpret_cast<float4 *>(dst)[q] =
        make_float4(fmaxf(acc[4 * q], 0.0f), fmaxf(acc[4 * q + 1], 0.0f), fmaxf(acc[4 * q + 2], 0.0f),
                    fmaxf(acc[4 * q + 3], 0.0f));
}

int ilog2(int x) {
  int l = 0;
  while ((1 << l) < x) ++l;
  return l;
}

}  // namespace

extern "C" int mf_conv3d_k4s2_pack_weights(const float *W, int32_t Cout, int32_t Cin, int32_t w_cin,
                                           int32_t c_off, float *wt, mfStream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  const int64_t total = (int64_t)Cout * 64 * Cin;
  hipLaunchKernelGGL(k_conv_pack, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, W, Cout, Cin,
                     w_cin, c_off, wt);
  return mf::check_launch("mf_conv3d_k4s2_pack_weights");
}

extern "C" int32_t mf_conv3d_k4s2_default_split(int32_t B, int32_t Cin, int32_t Cout, int32_t D) {
  const int Do = D / 2;
  const int64_t M = (int64_t)B * Do * Do * Do;
  const int64_t tiles = ((M + kBM - 1) / kBM) * (Cout / kBN);
  const int max_split = 2 * Cin >= 64 ? 64 : 2 * Cin;  // K-tiles = 2 Cin; a split keeps >= 1 of them
  int S = 1;
  while (tiles * S < 512 && S * 2 <= max_split && (2 * Cin) / (S * 2) >= 8) S *= 2;
  return S;
}

extern "C" int64_t mf_conv3d_k4s2_workspace_bytes(int32_t B, int32_t Cout, int32_t D, int32_t split) {
  const int Do = D / 2;
  return split > 1 ? (int64_t)split * B * Do * Do * Do * Cout * 4 : 0;
}

extern "C" int mf_conv3d_k4s2_fwd(const float *x, const float *wt, const float *bias, const float *add,
                                  float *out, void *ws, int32_t B, int32_t Cin, int32_t Cout, int32_t D,
                                  int32_t split, int32_t relu, mfStream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (B <= 0) return 0;
  const int lg = ilog2(Cin);
  if ((1 << lg) != Cin || Cin < 4 || Cout % kBN || D % 2 || D < 2 || split < 1 || (2 * Cin) % split ||
      (split > 1 && !ws) || (int64_t)B * D * D * D * Cin >= (1ll << 31) || (int64_t)Cout * 64 * Cin >= (1ll << 31)) {
    mf::set_last_error(hipErrorInvalidValue,
                       "conv3d_k4s2: need Cin a power of two >= 4, Cout % 128 == 0, even D, split | 2 Cin, "
                       "input and weights < 2^31 elements");
    return -(int)hipErrorInvalidValue;
  }
  if (int e = mf::allow_big_lds((const void *)k_conv3d_k4s2_mfma, kConvLds)) return e;
  const int Do = D / 2;
  const int64_t M = (int64_t)B * Do * Do * Do;
  ConvArgs a;
  a.x = x; a.wt = wt; a.bias = bias; a.add = add;
  a.out = split > 1 ? (float *)ws : out;
  a.B = B; a.D = D; a.Cin = Cin; a.cin_log2 = lg; a.Cout = Cout; a.S = split; a.relu = relu;
  const int64_t grid = ((M + kBM - 1) / kBM) * (Cout / kBN) * split;
  hipLaunchKernelGGL(k_conv3d_k4s2_mfma, dim3((unsigned)grid), dim3(256), kConvLds, stream, a);
  if (split > 1) {
    const int64_t MN = M * Cout;
    hipLaunchKernelGGL(k_conv_finish, dim3((unsigned)((MN / 4 + 255) / 256)), dim3(256), 0, stream,
                       (const float *)ws, bias, add, out, MN, Cout, split, relu);
  }
  return mf::check_launch("mf_conv3d_k4s2_fwd");
}

extern "C" int mf_to_channels_last(const float *src, float *dst, int32_t B, int32_t C, int64_t V,
                                   mfStream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (B <= 0 || C <= 0 || V <= 0) return 0;
  hipLaunchKernelGGL(k_to_channels_last, dim3((unsigned)((V + 31) / 32), (unsigned)((C + 31) / 32), (unsigned)B),
                     dim3(256), 0, stream, src, dst, C, (int)V);
  return mf::check_launch("mf_to_channels_last");
}

/* w1 [27][1][8], w2 [27][8][16] (tap-major, output channel innermost), biases [8], [16];
 * grid [B,D,D,D] -> h1 [B,D^3,8] (scratch) -> h2 [B,D^3,16], both ReLU-ed, channels-last. */
extern "C" int mf_occupancy_convs_fwd(const float *grid, const float *w1, const float *b1, const float *w2,
                                      const float *b2, float *h1, float *h2, int32_t B, int32_t D,
                                      mfStream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (B <= 0) return 0;
  const int64_t total = (int64_t)B * D * D * D;
  if (total * 16 >= (1ll << 31)) {
    mf::set_last_error(hipErrorInvalidValue, "occupancy_convs: B * D^3 * 16 must stay below 2^31");
    return -(int)hipErrorInvalidValue;
  }
  const unsigned nb = (unsigned)((total + 255) / 256);
  hipLaunchKernelGGL((k_occ_conv3<1, 8, 1>), dim3(nb), dim3(256), 0, stream, grid, w1, b1, h1, B, D);
  hipLaunchKernelGGL((k_occ_conv3<8, 16, 2>), dim3(nb), dim3(256), 0, stream, (const float *)h1, w2, b2, h2, B, D);
  return mf::check_launch("mf_occupancy_convs_fwd");
}
