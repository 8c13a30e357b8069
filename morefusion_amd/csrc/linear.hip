// Row-major fp32 GEMM with a fused bias + ReLU epilogue on v_mfma_f32_32x32x2_f32, for gfx950.
//
// Reference: the per-point 1x1 convolutions of the pose network -- `L.Convolution1D(c_in, c_out, 1)` x 4
// in each of the three heads (contrib/singleview_3d/models/model.py:76-91,245-262: 984 -> 640 -> 256 -> 128
// -> n_fg * {4,3,1}, ReLU between) and the point MLP in front of the voxelization (:59-66,101-111).  On
// points-major activations ([n points, channels], n = B * 1000) a 1x1 convolution is
//     out[m][n] = act( sum_k A[m][k] * W[n][k] + bias[n] ),
// 5.0 GFLOP per object for the heads.
//
// Same tile machinery as csrc/conv3d.hip (128 x 128 x 32 per 256-lane workgroup, 2 x 2 accumulators of
// 32 x 32 per wave, k-contiguous LDS rows of pitch 36 read with one conflict-free ds_read_b128 per
// fragment, register-staged double buffering); the A operand is a plain matrix with a row pitch, so a
// layer reads its input in place from a column block of a wider activation matrix and writes its output
// into one (`groups` independent GEMMs per launch: the three heads' layers 2-4 run side by side).
// K need not be a multiple of 32 (984): chunks of 4 beyond K load zeros.  W rows beyond N (the caller
// pads W to a multiple of 128 rows) are computed and not stored.
#include "mf_common.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int kBN = 128, kBK = 32;
constexpr int kPitch = kBK + 4;
// MI = 32-row accumulator blocks per wave along M: 2 -> 128 x 128 tiles, 1 -> 64 x 128 tiles (small M: batch 1
// has 1000 rows = 8 tiles of 128; the half-height tile doubles the workgroups that share the chip)
template <int MI> constexpr int tile_floats() { return (64 * MI + kBN) * kPitch; }
template <int MI> constexpr int lin_lds() { return 2 * tile_floats<MI>() * (int)sizeof(float); }

struct LinArgs {
  const float *A;     // [M][lda], group g at A + g * a_gs
  const float *W;     // [Npad][ldw] (k-contiguous rows), group g at W + g * w_gs
  const float *bias;  // [N] or null, group g at bias + g * b_gs
  float *out;         // [M][ldo], group g at out + g * o_gs
  int64_t a_gs, w_gs, b_gs, o_gs;
  int M, N, Npad, K, lda, ldw, ldo, groups, relu;
};


template <int MI>
__global__ __launch_bounds__(256, 2) void k_linear_mfma(LinArgs a) {
  MF_DYN_LDS(float, s_mem);
  constexpr int kBM = 64 * MI, kTileFloats = tile_floats<MI>();
  const int tiles_m = (a.M + kBM - 1) / kBM, tiles_n = a.Npad / kBN;
  const int per_group = tiles_m * tiles_n;
  // XCD-aware order as in conv3d.hip: an XCD gets a contiguous range of (group, N tile, M tile)
  const int G = gridDim.x;
  int L = blockIdx.x;
  if ((G & 7) == 0) L = (blockIdx.x & 7) * (G >> 3) + (blockIdx.x >> 3);
  const int grp = L / per_group;
  const int rem = L - grp * per_group;
  // N tile fastest: an XCD's contiguous range is a band of M tiles x all N tiles, so its 4 MB L2 holds the band's
  // rows of A (the wide operand: 31 MB for the heads' first layer) while W streams from the Infinity Cache
  // (M tile fastest: 562 MB of HBM-side fetches per launch for 39 MB of operands, profiles/r03_kernels_pmc.json)
  const int m0 = (rem / tiles_n) * kBM, n0 = (rem % tiles_n) * kBN;
  const float *A = a.A + grp * a.a_gs, *W = a.W + grp * a.w_gs;
  const int T = (a.K + kBK - 1) / kBK;

  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int wm = wave & 1, wn = wave >> 1;
  const int lrow = lane & 31, lhalf = lane >> 5;
  const int chunk = tid & 7, r0 = tid >> 3;

  const float *arow[4], *wrow[4];
  bool aok[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int m = m0 + r0 + 32 * i;
    aok[i] = m < a.M && i < 2 * MI;
    arow[i] = A + (int64_t)(aok[i] ? m : 0) * a.lda;
    wrow[i] = W + (int64_t)(n0 + r0 + 32 * i) * a.ldw;
  }

  f32x16 acc[MI][2];
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.0f;

  // a chunk past K reads the row's FIRST chunk (always inside the matrix) and selects zeros.
  // Two register sets (P, Q): the loads of K-tile t + 2 are issued while tile t is multiplied and tile t + 1 waits in
  // the other set -- a TWO-tile prefetch distance.  The wide operand (the heads' [8000, 984] input) streams from HBM /
  // Infinity Cache with no reuse inside a workgroup: one tile ahead (1.7-3.4 us of MFMAs) did not cover a loaded
  // miss (MFMA pipe 0.69 busy, profiles/r03_mfma_kernels_pmc.json; conv4, whose taps re-hit L2, runs at 0.84).
#define MF_LIN_FETCH(S, kt_)                                                                      \
  {                                                                                               \
    const int ko = (kt_) * kBK;                                                                   \
    kin##S = ko + 4 * chunk + 4 <= a.K;                                                           \
    const int kq = kin##S ? ko + 4 * chunk : 0;                                                   \
    ra0##S = *reinterpret_cast<const float4 *>(arow[0] + kq);                                     \
    ra1##S = *reinterpret_cast<const float4 *>(arow[1] + kq);                                     \
    if constexpr (MI == 2) {                                                                      \
      ra2##S = *reinterpret_cast<const float4 *>(arow[2] + kq);                                   \
      ra3##S = *reinterpret_cast<const float4 *>(arow[3] + kq);                                   \
    }                                                                                             \
    rb0##S = *reinterpret_cast<const float4 *>(wrow[0] + kq);                                     \
    rb1##S = *reinterpret_cast<const float4 *>(wrow[1] + kq);                                     \
    rb2##S = *reinterpret_cast<const float4 *>(wrow[2] + kq);                                     \
    rb3##S = *reinterpret_cast<const float4 *>(wrow[3] + kq);                                     \
  }
  // Rows past M read row 0 and compute values nobody stores (the epilogue skips them).  Chunks past K are zeroed in
  // the STASH, behind a branch that is only ever taken in the last K-tile: as selects right behind the loads they made
  // every fetch wait for its own data (s_waitcnt vmcnt(0) immediately after issue) -- the prefetch hid nothing.
#define MF_LIN_STASH(S, buf_)                                                                     \
  {                                                                                               \
    float *As_ = s_mem + (buf_) * kTileFloats + r0 * kPitch + 4 * chunk;                          \
    float *Bs_ = As_ + kBM * kPitch;                                                              \
    if (!kin##S) { /* only in the last K-tile of a K that is no multiple of 32 */                 \
      ra0##S = ra1##S = ra2##S = ra3##S = z4;                                                     \
      rb0##S = rb1##S = rb2##S = rb3##S = z4;                                                     \
    }                                                                                             \
    *reinterpret_cast<float4 *>(As_) = ra0##S;                                                    \
    *reinterpret_cast<float4 *>(As_ + 32 * kPitch) = ra1##S;                                      \
    if constexpr (MI == 2) {                                                                      \
      *reinterpret_cast<float4 *>(As_ + 64 * kPitch) = ra2##S;                                    \
      *reinterpret_cast<float4 *>(As_ + 96 * kPitch) = ra3##S;                                    \
    }                                                                                             \
    *reinterpret_cast<float4 *>(Bs_) = rb0##S;                                                    \
    *reinterpret_cast<float4 *>(Bs_ + 32 * kPitch) = rb1##S;                                      \
    *reinterpret_cast<float4 *>(Bs_ + 64 * kPitch) = rb2##S;                                      \
    *reinterpret_cast<float4 *>(Bs_ + 96 * kPitch) = rb3##S;                                      \
  }
#define MF_LIN_STEP(c_)                                                                           \
  acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.c_, b0.c_, acc[0][0], 0, 0, 0);             \
  acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.c_, b1.c_, acc[0][1], 0, 0, 0);             \
  if constexpr (MI == 2) {                                                                        \
    acc[MI - 1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.c_, b0.c_, acc[MI - 1][0], 0, 0, 0); \
    acc[MI - 1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.c_, b1.c_, acc[MI - 1][1], 0, 0, 0); \
  }
  // the 16 (MI = 2) MFMAs x 4 k-steps of K-tile `buf_`'s LDS buffer
#define MF_LIN_COMPUTE(buf_)                                                                      \
  {                                                                                               \
    const float *As = s_mem + (buf_) * kTileFloats + (wm * 32 * MI + lrow) * kPitch + 4 * lhalf;  \
    const float *Bs = s_mem + (buf_) * kTileFloats + (kBM + wn * 64 + lrow) * kPitch + 4 * lhalf; \
    _Pragma("unroll") for (int kk = 0; kk < 4; ++kk) {                                            \
      const float4 a0 = *reinterpret_cast<const float4 *>(As + 8 * kk);                           \
      const float4 a1 = MI == 2 ? *reinterpret_cast<const float4 *>(As + 32 * kPitch + 8 * kk) : a0; \
      const float4 b0 = *reinterpret_cast<const float4 *>(Bs + 8 * kk);                           \
      const float4 b1 = *reinterpret_cast<const float4 *>(Bs + 32 * kPitch + 8 * kk);             \
      MF_LIN_STEP(x) MF_LIN_STEP(y) MF_LIN_STEP(z) MF_LIN_STEP(w)                                 \
    }                                                                                             \
  }
#define MF_LIN_PIN()                                                                              \
  asm volatile("" ::: "memory"); /* keep the eight loads in front of the MFMAs (see conv3d.hip) */ \
  __builtin_amdgcn_sched_barrier(0);
  const float4 z4 = make_float4(0, 0, 0, 0);
  float4 ra0P, ra1P, ra2P = z4, ra3P = z4, rb0P, rb1P, rb2P, rb3P;
  float4 ra0Q, ra1Q, ra2Q = z4, ra3Q = z4, rb0Q, rb1Q, rb2Q, rb3Q;
  bool kinP = false, kinQ = false;
  const int Tl = T - 1;  // (tile indices past the end re-fetch the last tile: branch-free loads)
  MF_LIN_FETCH(P, 0);
  MF_LIN_STASH(P, 0);
  MF_LIN_FETCH(P, min(1, Tl));  // set P <- tile 1
  __syncthreads();
  for (int t = 0; t < T; t += 2) {
    // even tile t: LDS buffer 0; set Q <- tile t + 2; set P (tile t + 1) -> buffer 1
    MF_LIN_FETCH(Q, min(t + 2, Tl));
    MF_LIN_PIN();
    MF_LIN_COMPUTE(0);
    MF_LIN_STASH(P, 1);
    __syncthreads();
    if (t + 1 >= T) break;  // block-uniform
    // odd tile t + 1: LDS buffer 1; set P <- tile t + 3; set Q (tile t + 2) -> buffer 0
    MF_LIN_FETCH(P, min(t + 3, Tl));
    MF_LIN_PIN();
    MF_LIN_COMPUTE(1);
    MF_LIN_STASH(Q, 0);
    __syncthreads();
  }
#undef MF_LIN_PIN
#undef MF_LIN_COMPUTE
#undef MF_LIN_STEP
#undef MF_LIN_FETCH
#undef MF_LIN_STASH

  // Epilogue through LDS (the operand buffers are free: the loop ended on a barrier): a lane holds one COLUMN of
  // sixteen rows per accumulator, i.e. 64 four-byte stores scattered over 64 output rows -- store-issue bound, 8 us of
  // a 60 us workgroup at K = 984.  Bias + activation are applied on the way into a [rows][128 + 4] tile, each lane then
  // writes 16-byte row segments: 16 coalesced stores per lane instead of 64 scalar ones.
  float *dst = a.out + grp * a.o_gs;
  const float *bias = a.bias ? a.bias + grp * a.b_gs : nullptr;
  constexpr int kEp = kBN + 4;
  float *s_out = s_mem;  // [kBM][kEp] floats: 67.6 KB (MI = 2) / 33.8 KB (MI = 1) <= the two operand buffers
#pragma unroll
  for (int mi = 0; mi < MI; ++mi)
#pragma unroll
    for (int ni = 0; ni < 2; ++ni) {
      const int nl = wn * 64 + ni * 32 + lrow;
      const float bn = (bias && n0 + nl < a.N) ? bias[n0 + nl] : 0.0f;
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int ml = wm * 32 * MI + mi * 32 + (e & 3) + 8 * (e >> 2) + 4 * lhalf;
        float v = acc[mi][ni][e] + bn;
        if (a.relu) v = v > 0.0f ? v : 0.0f;
        s_out[ml * kEp + nl] = v;
      }
    }
  __syncthreads();
  const bool vec_ok = (a.ldo & 3) == 0 && (((uintptr_t)dst) & 15) == 0;  // 16-byte stores need aligned rows
  for (int i = tid; i < kBM * (kBN / 4); i += 256) {
    const int ml = i / (kBN / 4), c4 = i - ml * (kBN / 4);
    const int m = m0 + ml, n = n0 + 4 * c4;
    if (m >= a.M || n >= a.N) continue;
    const float4 v = *reinterpret_cast<const float4 *>(s_out + ml * kEp + 4 * c4);
    float *o = dst + (int64_t)m * a.ldo + n;
    if (vec_ok && n + 4 <= a.N) {
      *reinterpret_cast<float4 *>(o) = v;
    } else {
      const float vv[4] = {v.x, v.y, v.z, v.w};
      for (int j = 0; j < 4 && n + j < a.N; ++j) o[j] = vv[j];
    }
  }
}

}  // namespace

extern "C" int mf_linear_fwd(const float *A, int64_t a_group_stride, int32_t lda, const float *W,
                             int64_t w_group_stride, int32_t ldw, const float *bias, int64_t b_group_stride,
                             float *out, int64_t o_group_stride, int32_t ldo, int32_t M, int32_t N, int32_t Npad,
                             int32_t K, int32_t groups, int32_t relu, mfStream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (M <= 0 || N <= 0 || groups <= 0) return 0;
  if (K <= 0 || K % 4 || lda % 4 || ldw % 4 || a_group_stride % 4 || w_group_stride % 4 || Npad % kBN || Npad < N ||
      lda < K || ldw < K || ldo < N || (((uintptr_t)A | (uintptr_t)W) & 15)) {
    mf::set_last_error(hipErrorInvalidValue,
                       "linear: need K, lda, ldw, group strides % 4 == 0, 16-byte aligned A / W, W padded to Npad % 128 == 0 rows");
    return -(int)hipErrorInvalidValue;
  }
  // half-height tiles when full-height ones leave the chip under-filled (< 1 workgroup per CU)
  const int64_t full = (int64_t)((M + 127) / 128) * (Npad / kBN) * groups;
  const bool half = full < 256;
  if (int e = mf::allow_big_lds(half ? (const void *)k_linear_mfma<1> : (const void *)k_linear_mfma<2>,
                                half ? lin_lds<1>() : lin_lds<2>()))
    return e;
  LinArgs a;
  a.A = A; a.W = W; a.bias = bias; a.out = out;
  a.a_gs = a_group_stride; a.w_gs = w_group_stride; a.b_gs = b_group_stride; a.o_gs = o_group_stride;
  a.M = M; a.N = N; a.Npad = Npad; a.K = K; a.lda = lda; a.ldw = ldw; a.ldo = ldo; a.groups = groups; a.relu = relu;
  if (half) {
    const int64_t grid = (int64_t)((M + 63) / 64) * (Npad / kBN) * groups;
    hipLaunchKernelGGL(k_linear_mfma<1>, dim3((unsigned)grid), dim3(256), lin_lds<1>(), stream, a);
  } else {
    hipLaunchKernelGGL(k_linear_mfma<2>, dim3((unsigned)full), dim3(256), lin_lds<2>(), stream, a);
  }
  return mf::check_launch("mf_linear_fwd");
}
