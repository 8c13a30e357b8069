// bf16 MFMA GEMM engines (fp32 accumulate) for the 3-D CNN and the per-point 1x1 convolutions, forward AND
// backward, for gfx950 -- the kernels BASELINE config 5 (bf16 training) and `--dtype bf16` inference run on.
//
// Reference: contrib/singleview_3d/models/model.py:114-139 (conv3 / conv4: `L.Convolution3D(.., 4, 2, pad=1)`),
// :239-258 (the three heads' Convolution1D chains), :59-66,101-111 (point MLP), trained by
// examples/ycb_video/singleview_3d/train.py:342-369 (cuDNN forward + backward-data + backward-filter).
//
// Two engines on v_mfma_f32_32x32x16_bf16 (16x the fp32-MFMA rate of csrc/conv3d.hip / linear.hip):
//
//   NT   C[m][n] = sum_k A(m, k) * W[n][k]        both operands k-contiguous in memory
//        A loaders:  rows          a row-major activation matrix with a row pitch   (linear forward / dgrad)
//                    conv forward  im2col rows of a channels-last grid, k = (tap, cin)
//                    conv dgrad    rows = INPUT voxels grouped by parity class, k = (slot, cout): an input
//                                  voxel x receives from the 2 x 2 x 2 output voxels o = h + p - s with tap
//                                  (1 - p) + 2 s per axis (x = 2 h + p); a tile of 128 rows is class-homogeneous
//                                  and multiplies that class's [Cin][8 Cout] weight slice
//   TN   C[i][j] = sum_m P[m][i] * Q(m, j)        the reduction index is the ROW index of both operands
//        (weight gradients: P = dY, Q = the layer's input rows / im2col rows).  Rows land in LDS in the global order
//        (plain ds_write_b128); the MFMA fragments come out through gfx950's transposing LDS read
//        ds_read_b64_tr_b16.  Split over m into fp32 slabs (mf_wgrad_split: a cost model over rounds of workgroup
//        slots), summed in slab order by k_wgrad_finish / k_wgrad_finish_conv (deterministic, no float atomics).
//
// Tile: 128 x 128 x 64 per 256-lane workgroup (4 waves, each a 64 x 64 corner = 2 x 2 accumulators of 32 x 32).
// NT: LDS rows of 64 bf16 at a pitch of 144 bytes (36 dwords: the sixteen rows a ds_read_b128 phase touches start
// 4 dwords apart modulo 64 banks -- conflict-free).  Register-staged double buffering with the next tile's global
// loads issued in front of this tile's 16 MFMAs and kept opaque until behind them, one barrier per K-tile,
// 2 workgroups per CU.  Masked operand chunks (padding taps, tails) are buffer loads at an out-of-range offset: the
// hardware returns zeros, nothing touches the loaded data (mf_common.h).  The epilogue goes through LDS: bias +
// ReLU on the way in, 16-byte row segments out (bf16 or fp32, optionally accumulating into an fp32 tensor).
#include <cstdlib>

#include "mf_common.h"

namespace {

using mf::mf_f32x16;

constexpr int kBN = 128, kBK = 64;
constexpr int kPitch = 144;  // bytes per LDS row (64 bf16 + 16 bytes)
template <int MI> constexpr int nt_buf_bytes() { return (64 * MI + kBN) * kPitch; }
template <int MI> constexpr int nt_lds() {
  return 2 * nt_buf_bytes<MI>() > 64 * MI * (kBN + 4) * 4 ? 2 * nt_buf_bytes<MI>() : 64 * MI * (kBN + 4) * 4;
}

enum { kRows = 0, kConvFwd = 1, kConvDgrad = 2 };

struct NtArgs {
  const uint16_t *A;   // bf16 operand (rows / channels-last grid / channels-last output gradient)
  const uint16_t *W;   // bf16 [N][ldw] k-contiguous; group g at W + g * w_gs; dgrad: class p at W + p * N * ldw
  const float *bias;   // fp32 [N] or null; group g at bias + g * b_gs
  void *out;           // bf16 or fp32 rows, pitch ldo (elements); group g at out + g * o_gs
  int64_t a_gs, w_gs, b_gs, o_gs;
  int M, N, K, lda, ldw, ldo, groups;
  int relu, out_f32, accumulate;
  // rows mode only: weight group of every 64-row block of A (device array, -1 = no rows: the tile exits).  Row
  // tiles must not straddle groups (the producer pads each group to a multiple of 128 rows): the compact
  // parity-class rows of the sparse conv3 (csrc/sparseconv_bf16.hip) multiply their class's weight slice.
  const int32_t *tile_group;
  // conv geometry: D = INPUT grid size, Do = OUTPUT grid size = 1 << olog; forward taps ks^3 at x = stride * o - pad
  // + dil * k per axis (dgrad: the k4 / s2 / p1 parity-class form only)
  int B, D, Do, olog, Cin, Cout, ks, stride, pad, dil;
  // k_gemm_nt_bf16_pp only: S > 1 splits the K-tiles into S contiguous ranges; split s writes its fp32 partial
  // sums (no bias / ReLU) to slab + s * M * N (row pitch N), k_splitk_finish adds them in order
  int S;
  float *slab;
  int dbg;  // k_gemm_nt_bf16_pp ablations (MF_PP_DBG; timing experiments only, results are wrong): see launch_nt
};


template <int MODE, int MI>
__global__ __launch_bounds__(256, 2) void k_gemm_nt_bf16(NtArgs a) {
  MF_DYN_LDS(unsigned char, s_raw);
  constexpr int kBM = 64 * MI, kBuf = nt_buf_bytes<MI>();
  const int tiles_m = (a.M + kBM - 1) / kBM, tiles_n = (a.N + kBN - 1) / kBN;
  const int per_group = tiles_m * tiles_n;
  const int G = gridDim.x;
  int L = blockIdx.x;
  if ((G & 7) == 0) L = (blockIdx.x & 7) * (G >> 3) + (blockIdx.x >> 3);  // XCD-contiguous logical order
  const int grp = L / per_group;
  const int rem = L - grp * per_group;
  const int m0 = (rem / tiles_n) * kBM, n0 = (rem % tiles_n) * kBN;  // N tile fastest (csrc/linear.hip)
  const int T = (a.K + kBK - 1) / kBK;

  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int wm = wave & 1, wn = wave >> 1;
  const int lrow = lane & 31, lhalf = lane >> 5;
  const int chunk = tid & 7, r0 = tid >> 3;  // this lane stages rows r0 + 32 i, bf16 8 chunk .. + 7 of the K-tile

  const int Do = a.Do, dol = a.olog;
  const uint16_t *A = a.A + grp * a.a_gs;
  const uint16_t *W = a.W + grp * a.w_gs;
  if (MODE == kRows && a.tile_group) {
    const int g = a.tile_group[m0 >> 6];  // (block-uniform)
    if (g < 0) return;
    W += (int64_t)g * a.w_gs;
  }
  int cls = 0;
  if (MODE == kConvDgrad) {  // tile-uniform parity class: its weight slice
    cls = (m0 >> (3 * dol)) & 7;
    W += (int64_t)cls * a.N * a.ldw;
  }
  // per staged row: element offset of its k = 0 chunk and validity bits
  //   rows:        bit 12 = row exists
  //   conv fwd:    bits kx | 4 + ky | 8 + kz = tap coordinate inside the grid (csrc/conv3d.hip)
  //   conv dgrad:  bits sx | 4 + sy | 8 + sz = contributing output voxel h + p - s inside the output grid
  int base[2 * MI], mask[2 * MI];
#pragma unroll
  for (int i = 0; i < 2 * MI; ++i) {
    const int m = m0 + r0 + 32 * i;
    const bool row_ok = m < a.M;
    const int mm = row_ok ? m : 0;
    int mk = row_ok ? 1 << 12 : 0;
    if (MODE == kRows) {
      base[i] = mm * a.lda;
    } else if (MODE == kConvFwd) {
      const int b = mm >> (3 * dol), o = mm & ((1 << (3 * dol)) - 1);
      const int ox = o >> (2 * dol), oy = (o >> dol) & (Do - 1), oz = o & (Do - 1);
      const int x0 = a.stride * ox - a.pad, y0 = a.stride * oy - a.pad, z0 = a.stride * oz - a.pad;
      base[i] = (((b * a.D + x0) * a.D + y0) * a.D + z0) * a.Cin;
#pragma unroll
      for (int k = 0; k < 4; ++k) {  // (k >= ks: never asked for)
        mk |= ((unsigned)(x0 + a.dil * k) < (unsigned)a.D ? 1 : 0) << k;
        mk |= ((unsigned)(y0 + a.dil * k) < (unsigned)a.D ? 1 : 0) << (4 + k);
        mk |= ((unsigned)(z0 + a.dil * k) < (unsigned)a.D ? 1 : 0) << (8 + k);
      }
    } else {
      // m = ((b * 8 + p) * Do^3 + h): input voxel x = 2 h + p per axis
      const int h = mm & ((1 << (3 * dol)) - 1), b = mm >> (3 * dol + 3);
      const int hx = h >> (2 * dol), hy = (h >> dol) & (Do - 1), hz = h & (Do - 1);
      const int ux = hx + (cls & 1), uy = hy + ((cls >> 1) & 1), uz = hz + ((cls >> 2) & 1);  // slot (0,0,0)
      base[i] = (((b * Do + ux) * Do + uy) * Do + uz) * a.Cout;
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        mk |= ((unsigned)(ux - s) < (unsigned)Do ? 1 : 0) << s;
        mk |= ((unsigned)(uy - s) < (unsigned)Do ? 1 : 0) << (4 + s);
        mk |= ((unsigned)(uz - s) < (unsigned)Do ? 1 : 0) << (8 + s);
      }
    }
    mask[i] = mk;
  }
  uint32_t wrow[4];  // byte offsets into W (weights: far below 2^32 bytes)
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int n = n0 + r0 + 32 * i;
    wrow[i] = 2u * (uint32_t)((int64_t)(n < a.N ? n : 0) * a.ldw);
  }

  mf_f32x16 acc[MI][2];
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.0f;

  // K-tile kt -> registers.  Nothing touches the loaded data before the stash: a select right behind a load would
  // make the wave wait for its own data at once (s_waitcnt vmcnt(0) in front of the MFMAs) and the prefetch would
  // hide nothing.
  // (Scalars and macros, not arrays in lambdas: behind the "memory" clobber that pins the loads in front of the MFMAs,
  // arrays captured by reference were kept in scratch memory -- every load waited for and stored.)
  // ONE register set, one tile ahead.  (Two sets -- tile t + 2 in flight while t + 1 waits -- were measured twice in
  // round 4, before and after the VALU diet: no gain on any shape, 90 more registers.)
  uint4 ra0P, ra1P, ra2P = make_uint4(0u, 0u, 0u, 0u), ra3P = ra2P, rb0P, rb1P, rb2P, rb3P;
  // This lane's position in K, advanced by one K-tile per fetch (the fetches run over kt = 0, 1, 2, ... in order): the
  // chunk's k offset and, for the convolutions, its (tap, channel) -- tracked incrementally (round 4, first version:
  // two integer divisions per fetch and 64-bit address arithmetic per load, 12 VALU instructions per MFMA by
  // SQ_INSTS_VALU; the MFMA pipe at 0.37).
  int kg = 8 * chunk, tc = 0, tx = 0, ty = 0, tz = 0;  // conv fwd: tap (tx, ty, tz), channel tc; dgrad: slot tx, cout tc
  if (MODE == kConvFwd) {
    const int tap = kg / a.Cin;
    tc = kg - tap * a.Cin;
    const int kxy = tap / a.ks;
    tz = tap - kxy * a.ks; tx = kxy / a.ks; ty = kxy - tx * a.ks;
  } else if (MODE == kConvDgrad) {
    tx = kg / a.Cout;
    tc = kg - tx * a.Cout;
  }
  // A masked chunk (padding tap, row past the edge, K tail) is a buffer load at an OUT-OF-RANGE offset: the hardware
  // returns zeros (mf_common.h).  No select or AND on the loaded data (that was 44 VALU instructions per K-tile in
  // every wave that touches a border -- nearly all of them in a 16^3 grid), and the stash is eight plain
  // ds_write_b128.  The weight operand needs no mask at all: behind the K tail it re-reads k = 0 (finite; the A chunk
  // there is zero), and a column past N re-reads row 0 into an accumulator column the epilogue never stores.
  const mf::BufRsrc Ars = mf::make_rsrc(A), Wrs = mf::make_rsrc(W);
#define MF_NT_LOAD_A(S, i_, reg_)                                                                     \
  reg_ = mf::buf_load16(Ars, (mask[i_] & bits_) == bits_ ? 2u * (uint32_t)(base[i_] + off_) : mf::kBufMasked);
#define MF_NT_LOAD_B(S, i_, reg_) reg_ = mf::buf_load16(Wrs, wrow[i_] + kofs_);
#define MF_NT_FETCH(S)                                                                                \
  {                                                                                                   \
    const bool kin_ = kg + 8 <= a.K;                                                                  \
    const uint32_t kofs_ = kin_ ? 2u * (uint32_t)kg : 0u;                                             \
    int off_ = kg, bits_ = 1 << 12;                                                                   \
    if (MODE == kConvFwd) {                                                                           \
      off_ = ((tx * a.D + ty) * a.D + tz) * a.dil * a.Cin + tc;                                       \
      bits_ = tx < a.ks ? (1 << tx) | (16 << ty) | (256 << tz) | (1 << 12) : 1 << 13;                 \
    } else if (MODE == kConvDgrad) {                                                                  \
      const int sx = tx & 1, sy = (tx >> 1) & 1, sz = tx >> 2;                                        \
      off_ = tc - ((sx * Do + sy) * Do + sz) * a.Cout;                                                \
      bits_ = (1 << sx) | (16 << sy) | (256 << sz) | (1 << 12);                                       \
    }                                                                                                 \
    if (!kin_) bits_ = 1 << 13; /* (no row has bit 13) */                                             \
    MF_NT_LOAD_A(S, 0, ra0##S) MF_NT_LOAD_A(S, 1, ra1##S)                                             \
    if constexpr (MI == 2) { MF_NT_LOAD_A(S, 2 * MI - 2, ra2##S) MF_NT_LOAD_A(S, 2 * MI - 1, ra3##S) } \
    MF_NT_LOAD_B(S, 0, rb0##S) MF_NT_LOAD_B(S, 1, rb1##S) MF_NT_LOAD_B(S, 2, rb2##S) MF_NT_LOAD_B(S, 3, rb3##S) \
    kg += kBK;                                                                                        \
    if (MODE == kConvFwd) {                                                                           \
      tc += kBK;                                                                                      \
      while (tc >= a.Cin) {                                                                           \
        tc -= a.Cin;                                                                                  \
        if (++tz == a.ks) { tz = 0; if (++ty == a.ks) { ty = 0; ++tx; } }                             \
      }                                                                                               \
    } else if (MODE == kConvDgrad) {                                                                  \
      tc += kBK;                                                                                      \
      while (tc >= a.Cout) { tc -= a.Cout; ++tx; }                                                    \
    }                                                                                                 \
  }
  // (MF_HOLD: the staged registers stay opaque until here, BEHIND the MFMAs -- and with them the wait for the loads.)
#define MF_NT_STASH(S, buf_)                                                                          \
  {                                                                                                   \
    MF_HOLD(ra0##S); MF_HOLD(ra1##S); MF_HOLD(rb0##S); MF_HOLD(rb1##S); MF_HOLD(rb2##S); MF_HOLD(rb3##S); \
    if constexpr (MI == 2) { MF_HOLD(ra2##S); MF_HOLD(ra3##S); }                                      \
    unsigned char *As_ = s_raw + (buf_) * kBuf + r0 * kPitch + 16 * chunk;                            \
    unsigned char *Bs_ = As_ + kBM * kPitch;                                                          \
    *reinterpret_cast<uint4 *>(As_) = ra0##S; *reinterpret_cast<uint4 *>(As_ + 32 * kPitch) = ra1##S; \
    if constexpr (MI == 2) {                                                                          \
      *reinterpret_cast<uint4 *>(As_ + 64 * kPitch) = ra2##S; *reinterpret_cast<uint4 *>(As_ + 96 * kPitch) = ra3##S; \
    }                                                                                                 \
    *reinterpret_cast<uint4 *>(Bs_) = rb0##S; *reinterpret_cast<uint4 *>(Bs_ + 32 * kPitch) = rb1##S; \
    *reinterpret_cast<uint4 *>(Bs_ + 64 * kPitch) = rb2##S; *reinterpret_cast<uint4 *>(Bs_ + 96 * kPitch) = rb3##S; \
  }
  // NJ_ = 2: both 32-column blocks of the wave's 64 columns; NJ_ = 1: the first only (the second lies past N).
  // The fragments of k-step s + 1 are read while the MFMAs of step s run (round 5: two
  // register sets, the order pinned with sched_group_barrier; MF_NT_PIPE=0: the scheduler's own order).
#ifndef MF_NT_PIPE
#define MF_NT_PIPE 1
#endif
#define MF_NT_FRAGS(set_, s_, NJ_)                                                                    \
  {                                                                                                   \
    fa[set_][0] = *reinterpret_cast<const uint4 *>(As + 32 * (s_));                                   \
    if constexpr (MI == 2) fa[set_][1] = *reinterpret_cast<const uint4 *>(As + 32 * kPitch + 32 * (s_)); \
    fb[set_][0] = *reinterpret_cast<const uint4 *>(Bs + 32 * (s_));                                   \
    if constexpr (NJ_ == 2) fb[set_][1] = *reinterpret_cast<const uint4 *>(Bs + 32 * kPitch + 32 * (s_)); \
  }
#define MF_NT_COMPUTE(buf_, NJ_)                                                                      \
  {                                                                                                   \
    asm volatile("" ::: "memory");                                                                    \
    __builtin_amdgcn_sched_barrier(0);                                                                \
    const unsigned char *As = s_raw + (buf_) * kBuf + (wm * 32 * MI + lrow) * kPitch + 16 * lhalf;    \
    const unsigned char *Bs = s_raw + (buf_) * kBuf + (kBM + wn * 64 + lrow) * kPitch + 16 * lhalf;   \
    if constexpr (MF_NT_PIPE) {                                                                       \
      uint4 fa[2][2], fb[2][2];                                                                       \
      MF_NT_FRAGS(0, 0, NJ_)                                                                          \
      __builtin_amdgcn_sched_group_barrier(0x100, MI + NJ_, 0);                                       \
      _Pragma("unroll") for (int s = 0; s < 4; ++s) {                                                 \
        const int c = s & 1;                                                                          \
        if (s < 3) MF_NT_FRAGS(c ^ 1, s + 1, NJ_)                                                     \
        _Pragma("unroll") for (int mi = 0; mi < MI; ++mi) {                                           \
          acc[mi][0] = mf::mfma_bf16_32x32x16(fa[c][mi], fb[c][0], acc[mi][0]);                       \
          if constexpr (NJ_ == 2) acc[mi][1] = mf::mfma_bf16_32x32x16(fa[c][mi], fb[c][1], acc[mi][1]); \
        }                                                                                             \
        if (s < 3) __builtin_amdgcn_sched_group_barrier(0x100, MI + NJ_, 0);                          \
        __builtin_amdgcn_sched_group_barrier(0x008, MI * NJ_, 0);                                     \
      }                                                                                               \
    } else {                                                                                          \
      _Pragma("unroll") for (int s = 0; s < 4; ++s) {                                                 \
        const uint4 a0 = *reinterpret_cast<const uint4 *>(As + 32 * s);                               \
        const uint4 b0 = *reinterpret_cast<const uint4 *>(Bs + 32 * s);                               \
        acc[0][0] = mf::mfma_bf16_32x32x16(a0, b0, acc[0][0]);                                        \
        if constexpr (NJ_ == 2) {                                                                     \
          const uint4 b1 = *reinterpret_cast<const uint4 *>(Bs + 32 * kPitch + 32 * s);               \
          acc[0][1] = mf::mfma_bf16_32x32x16(a0, b1, acc[0][1]);                                      \
          if constexpr (MI == 2) {                                                                    \
            const uint4 a1 = *reinterpret_cast<const uint4 *>(As + 32 * kPitch + 32 * s);             \
            acc[MI - 1][0] = mf::mfma_bf16_32x32x16(a1, b0, acc[MI - 1][0]);                          \
            acc[MI - 1][1] = mf::mfma_bf16_32x32x16(a1, b1, acc[MI - 1][1]);                          \
          }                                                                                           \
        } else if constexpr (MI == 2) {                                                               \
          const uint4 a1 = *reinterpret_cast<const uint4 *>(As + 32 * kPitch + 32 * s);               \
          acc[MI - 1][0] = mf::mfma_bf16_32x32x16(a1, b0, acc[MI - 1][0]);                            \
        }                                                                                             \
      }                                                                                               \
    }                                                                                                 \
    __builtin_amdgcn_sched_barrier(0);                                                                \
  }
  // Column blocks of this wave that lie past N (the last N-tile of a layer whose width is not a multiple of 128:
  // conv3's data gradient has N = 160 -- its second tile holds 32 columns) are not multiplied: wave-uniform.
  const int ncols = a.N - (n0 + wn * 64);  // columns of this wave's 64 that exist
  // Per K-tile t: the loads of tile t + 1 are issued, tile t is multiplied, the registers go into the other buffer,
  // barrier.  (Stash AFTER the barrier and the next fetch right behind it -- the order the 256^2-tile GEMMs of the
  // programming guide prefer -- measured 3 - 5 % slower here, at 2 workgroups per CU.)  The fetches run over the
  // K-tiles in order, one past the last (k beyond K: every chunk masked, zeros into the buffer nobody reads again) --
  // NOT under "if (t + 1 < T)": behind a branch the compiler copies the loaded registers at the join and waits for
  // the loads right where they are issued (measured: 2x slower).
  MF_NT_FETCH(P);  // tile 0
  MF_NT_STASH(P, 0);
  __syncthreads();
  for (int t = 0; t < T; ++t) {
    MF_NT_FETCH(P);  // tile t + 1 in flight under the MFMAs of tile t
    if (ncols > 32) MF_NT_COMPUTE(t & 1, 2) else if (ncols > 0) MF_NT_COMPUTE(t & 1, 1)
    MF_NT_STASH(P, (t + 1) & 1);
    __syncthreads();
  }
#undef MF_NT_FRAGS
#undef MF_NT_COMPUTE
#undef MF_NT_STASH
#undef MF_NT_FETCH
#undef MF_NT_LOAD_B
#undef MF_NT_LOAD_A

  // epilogue through LDS (the loop ended on a barrier: the operand buffers are free)
  constexpr int kEp = kBN + 4;
  float *s_out = reinterpret_cast<float *>(s_raw);  // [kBM][kEp]
  const float *bias = a.bias ? a.bias + grp * a.b_gs : nullptr;
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
  for (int i = tid; i < kBM * (kBN / 8); i += 256) {
    const int ml = i / (kBN / 8), c8 = i - ml * (kBN / 8);
    const int m = m0 + ml, n = n0 + 8 * c8;
    if (m >= a.M || n >= a.N) continue;
    int64_t orow = m;
    if (MODE == kConvDgrad) {  // class-ordered row -> channels-last voxel row of the input gradient
      const int h = m & ((1 << (3 * dol)) - 1), p = (m >> (3 * dol)) & 7, b = m >> (3 * dol + 3);
      const int x = 2 * (h >> (2 * dol)) + (p & 1), y = 2 * ((h >> dol) & (Do - 1)) + ((p >> 1) & 1),
                z = 2 * (h & (Do - 1)) + (p >> 2);
      orow = (((int64_t)b * a.D + x) * a.D + y) * a.D + z;
    }
    const float4 v0 = *reinterpret_cast<const float4 *>(s_out + ml * kEp + 8 * c8);
    const float4 v1 = *reinterpret_cast<const float4 *>(s_out + ml * kEp + 8 * c8 + 4);
    float v[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
    const int nv = a.N - n < 8 ? a.N - n : 8;
    if (a.out_f32) {
      float *o = reinterpret_cast<float *>(a.out) + grp * a.o_gs + orow * a.ldo + n;
      if (nv == 8 && (a.ldo & 3) == 0 && ((uintptr_t)o & 15) == 0) {
        float4 *o4 = reinterpret_cast<float4 *>(o);
        if (a.accumulate) {
          const float4 p0 = o4[0], p1 = o4[1];
          v[0] += p0.x; v[1] += p0.y; v[2] += p0.z; v[3] += p0.w;
          v[4] += p1.x; v[5] += p1.y; v[6] += p1.z; v[7] += p1.w;
        }
        o4[0] = make_float4(v[0], v[1], v[2], v[3]);
        o4[1] = make_float4(v[4], v[5], v[6], v[7]);
      } else {
        for (int j = 0; j < nv; ++j) o[j] = a.accumulate ? o[j] + v[j] : v[j];
      }
    } else {
      uint16_t *o = reinterpret_cast<uint16_t *>(a.out) + grp * a.o_gs + orow * a.ldo + n;
      if (nv == 8 && (a.ldo & 7) == 0 && ((uintptr_t)o & 15) == 0) {
        *reinterpret_cast<uint4 *>(o) = make_uint4(mf::pack_bf16x2(v[0], v[1]), mf::pack_bf16x2(v[2], v[3]),
                                                   mf::pack_bf16x2(v[4], v[5]), mf::pack_bf16x2(v[6], v[7]));
      } else {
        for (int j = 0; j < nv; ++j) o[j] = (uint16_t)mf::bf16_bits(v[j]);
      }
    }
  }
}

// ---- the 256 x 256 x 64 tile (round 5: eight waves, each a 128 x 64 corner = 4 x 2 accumulators) --------------------
// A wave of the 128 x 128 tile reads one LDS fragment (ds_read_b128) per MFMA: at the MFMA rate of gfx950 that alone
// keeps the LDS pipe busy all the time (1 KB per wave per 32-cycle MFMA, four SIMDs).  With a 128 x 64 wave tile a
// k-step reads 4 + 2 fragments for 8 MFMAs -- 0.75 per MFMA.  Round 5's form of this tile staged its operands through
// registers (global -> VGPR -> ds_write_b128, one barrier per K-tile: 852 / 961 TFLOP/s on conv3 forward / conv4 data
// gradient, MFMA pipe 0.43 busy); round 6's k_gemm_nt_bf16_pp below replaced it (1019-1059 / 1202-1225 in the same
// measurement) and the register-staged kernel is gone from the source.
constexpr int kBigM = 256, kBigN = 256;

// ---- the 256 x 256 x 64 tile with LDS-DMA operands and two wave groups in ping-pong (round 6) ------------------------
// Round 5's form of this tile moved every operand chunk global -> VGPR -> ds_write_b128 -> barrier, once per K-tile: all
// eight waves met at that barrier, waited out their loads, stored, and started reading fragments at the same moment --
// the MFMA pipe idled through every one of these episodes (0.42-0.43 busy).  Here
//   * operands go global -> LDS directly (buffer_load_dwordx4 ... lds, mf::glds16: 1 KiB = 8 tile rows per wave
//     instruction; masked chunks are out-of-range offsets and land as zeros): no staging registers, no store pass;
//   * the LDS image of an operand is [256 rows][128 bytes] with the 16-byte chunk index XORed with (row >> 1) & 7.  The
//     DMA writes a wave's 1 KiB lane-linear, so the swizzle is applied to the SOURCE: the lane whose slot is chunk
//     position q of row r fetches global chunk q ^ ((r >> 1) & 7); a fragment read (row = lane % 32, chunk
//     2 s + lane / 32) XORs the same value: every 16-lane group of a ds_read_b128 covers all 64 banks once;
//   * a K-tile is two phases of two k-steps: 12 fragment reads and 4 DMA requests in the phase's load section, 16
//     MFMAs in its MFMA section, a barrier behind each.  The waves with the upper and the lower 128 rows of the tile
//     (waves 0-3 / 4-7: one of each per SIMD) run ONE BARRIER APART: while one group multiplies (s_setprio 1) the
//     other reads the fragments of its next phase and issues DMA -- the SIMD's MFMA pipe always has a wave with
//     operands in registers, and the LDS round trip and the DMA issue time (60-180 cycles of the issuing wave per
//     request) are paid beside the other group's MFMAs;
//   * the 160 KiB of LDS are THREE A stages + TWO W stages (the weight panel is shared by every workgroup and comes
//     from L2; the activation rows come from HBM): in tile t, phase 0 requests W(t + 1) into the stage W(t - 1) was
//     read from, phase 1 requests A(t + 2) into the stage of A(t - 1) and then waits with a COUNTED vmcnt(4) --
//     everything but A(t + 2), which stays in flight across the barriers.
// Ordering, in barrier intervals (group 0's load section of phase (t, p) is interval 4 t + 2 p, its MFMA section
// 4 t + 2 p + 1; group 1 one interval later):
//   WAR  the last reads of tile t - 1 are group 1's phase (t - 1, 1) in interval 4 t - 1, retired by lgkmcnt(0) BEFORE
//        the barrier that ends it; the earliest request into a stage of tile t - 1 is group 0's in interval 4 t.
//   RAW  every wave waits for its own requests of A(t + 1) and W(t + 1) in the load section of its phase (t, 1)
//        (intervals 4 t + 2 / 4 t + 3) in front of a barrier; the first read of tile t + 1 is group 0's in 4 t + 4.
// The fragment reads are inline asm (mf::lds_read16_async): the compiler puts s_waitcnt vmcnt(0) in front of any LDS
// read it can see while a DMA is pending.  Same loaders, masks, tile order and epilogue as the kernel above.
constexpr int kPpOp = 256 * 128;             // bytes of one operand stage: [256 rows][128]
constexpr int kPpW0 = 3 * kPpOp;             // A stages at 0, 1, 2 x kPpOp; W stages behind them
constexpr int nt_pp_lds() { return 5 * kPpOp; }  // 160 KiB: all of a CU's LDS (the epilogue's 64 x 260 floats fit inside)

template <int MODE>
__global__ __launch_bounds__(512, 2) void k_gemm_nt_bf16_pp(NtArgs a) {
  MF_DYN_LDS(unsigned char, s_raw);
  constexpr int kBM = kBigM, kBNb = kBigN;
  const int tiles_m = (a.M + kBM - 1) / kBM, tiles_n = (a.N + kBNb - 1) / kBNb;
  const int per_group = tiles_m * tiles_n;
  const int G = gridDim.x;
  int L = blockIdx.x;
  if ((G & 7) == 0) L = (blockIdx.x & 7) * (G >> 3) + (blockIdx.x >> 3);  // XCD-contiguous logical order
  const int split = L / (per_group * a.groups);  // (0 unless a.S > 1: the splits of a tile are S whole rounds apart)
  L -= split * per_group * a.groups;
  const int grp = L / per_group;
  const int rem = L - grp * per_group;
  const int m0 = (rem / tiles_n) * kBM, n0 = (rem % tiles_n) * kBNb;  // N tile fastest (csrc/linear.hip)
  const int Tall = (a.K + kBK - 1) / kBK, Tper = (Tall + a.S - 1) / a.S;
  const int t0 = split * Tper;
  const int T = max(0, min(Tall, t0 + Tper) - t0);  // this workgroup's K-tiles: t0 .. t0 + T - 1

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = mf::wave_uniform(tid >> 6);
  const int wm = wave >> 2, wn = wave & 3;  // wm = the ping-pong group: waves w and w + 4 share a SIMD
  const int lrow = lane & 31, lhalf = lane >> 5;
  // DMA slot of this lane: tile rows r0 + 64 i, chunk position tid & 7 of the row -> global chunk ``chunk``
  const int r0 = tid >> 3;
  const int chunk = (tid & 7) ^ ((r0 >> 1) & 7);

  const int Do = a.Do, dol = a.olog;
  const uint16_t *A = a.A + grp * a.a_gs;
  const uint16_t *W = a.W + grp * a.w_gs;
  int cls = 0;
  if (MODE == kConvDgrad) {  // tile-uniform parity class: its weight slice
    cls = (m0 >> (3 * dol)) & 7;
    W += (int64_t)cls * a.N * a.ldw;
  }
  // per staged row: element offset of its k = 0 chunk and validity bits (as in k_gemm_nt_bf16)
  int base[4], mask[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int m = m0 + r0 + 64 * i;
    const bool row_ok = m < a.M;
    const int mm = row_ok ? m : 0;
    int mk = row_ok ? 1 << 12 : 0;
    if (MODE == kRows) {
      base[i] = mm * a.lda;
    } else if (MODE == kConvFwd) {
      const int b = mm >> (3 * dol), o = mm & ((1 << (3 * dol)) - 1);
      const int ox = o >> (2 * dol), oy = (o >> dol) & (Do - 1), oz = o & (Do - 1);
      const int x0 = a.stride * ox - a.pad, y0 = a.stride * oy - a.pad, z0 = a.stride * oz - a.pad;
      base[i] = (((b * a.D + x0) * a.D + y0) * a.D + z0) * a.Cin;
#pragma unroll
      for (int k = 0; k < 4; ++k) {  // (k >= ks: never asked for)
        mk |= ((unsigned)(x0 + a.dil * k) < (unsigned)a.D ? 1 : 0) << k;
        mk |= ((unsigned)(y0 + a.dil * k) < (unsigned)a.D ? 1 : 0) << (4 + k);
        mk |= ((unsigned)(z0 + a.dil * k) < (unsigned)a.D ? 1 : 0) << (8 + k);
      }
    } else {
      const int h = mm & ((1 << (3 * dol)) - 1), b = mm >> (3 * dol + 3);
      const int hx = h >> (2 * dol), hy = (h >> dol) & (Do - 1), hz = h & (Do - 1);
      const int ux = hx + (cls & 1), uy = hy + ((cls >> 1) & 1), uz = hz + ((cls >> 2) & 1);  // slot (0,0,0)
      base[i] = (((b * Do + ux) * Do + uy) * Do + uz) * a.Cout;
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        mk |= ((unsigned)(ux - s) < (unsigned)Do ? 1 : 0) << s;
        mk |= ((unsigned)(uy - s) < (unsigned)Do ? 1 : 0) << (4 + s);
        mk |= ((unsigned)(uz - s) < (unsigned)Do ? 1 : 0) << (8 + s);
      }
    }
    mask[i] = mk;
  }
  uint32_t wrow[4];  // byte offsets into W (weights: far below 2^32 bytes)
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int n = n0 + r0 + 64 * i;
    wrow[i] = 2u * (uint32_t)((int64_t)(n < a.N ? n : 0) * a.ldw);
  }

  mf_f32x16 acc[4][2];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.0f;

  // this lane's position in K, advanced by one K-tile per request (see k_gemm_nt_bf16); the A requests run one tile
  // ahead of the W requests
  int kg = 8 * chunk + kBK * t0, tc = 0, tx = 0, ty = 0, tz = 0, kgw = kg;
  if (MODE == kConvFwd) {
    const int tap = kg / a.Cin;
    tc = kg - tap * a.Cin;
    const int kxy = tap / a.ks;
    tz = tap - kxy * a.ks; tx = kxy / a.ks; ty = kxy - tx * a.ks;
  } else if (MODE == kConvDgrad) {
    tx = kg / a.Cout;
    tc = kg - tx * a.Cout;
  }
  const mf::BufRsrc Ars = mf::make_rsrc(A), Wrs = mf::make_rsrc(W);
  // the 1 KiB of LDS a DMA instruction of this wave fills: rows 64 i + 8 wave .. + 7 of an operand stage
  unsigned char *const dma0 = s_raw + 8 * wave * 128;
  // requests i0_ .. i1_ - 1 (rows 64 i .. + 63) of A's next K-tile -> A stage sa_; the position advances behind the last
#define MF_PP_REQ_A(sa_, i0_, i1_)                                                                    \
  {                                                                                                   \
    const bool kin_ = kg + 8 <= a.K;                                                                  \
    int off_ = kg, bits_ = 1 << 12;                                                                   \
    if (MODE == kConvFwd) {                                                                           \
      off_ = ((tx * a.D + ty) * a.D + tz) * a.dil * a.Cin + tc;                                       \
      bits_ = tx < a.ks ? (1 << tx) | (16 << ty) | (256 << tz) | (1 << 12) : 1 << 13;                 \
    } else if (MODE == kConvDgrad) {                                                                  \
      const int sx = tx & 1, sy = (tx >> 1) & 1, sz = tx >> 2;                                        \
      off_ = tc - ((sx * Do + sy) * Do + sz) * a.Cout;                                                \
      bits_ = (1 << sx) | (16 << sy) | (256 << sz) | (1 << 12);                                       \
    }                                                                                                 \
    if (!kin_) bits_ = 1 << 13; /* (no row has bit 13) */                                             \
    _Pragma("unroll") for (int i = (i0_); i < (i1_); ++i)                                             \
      mf::glds16(Ars, (mask[i] & bits_) == bits_ ? 2u * (uint32_t)(base[i] + off_) : mf::kBufMasked,  \
                 dma0 + (sa_) * kPpOp + i * 64 * 128);                                                \
    if ((i1_) == 4) {                                                                                 \
      kg += kBK;                                                                                      \
      if (MODE == kConvFwd) {                                                                         \
        tc += kBK;                                                                                    \
        while (tc >= a.Cin) {                                                                         \
          tc -= a.Cin;                                                                                \
          if (++tz == a.ks) { tz = 0; if (++ty == a.ks) { ty = 0; ++tx; } }                           \
        }                                                                                             \
      } else if (MODE == kConvDgrad) {                                                                \
        tc += kBK;                                                                                    \
        while (tc >= a.Cout) { tc -= a.Cout; ++tx; }                                                  \
      }                                                                                               \
    }                                                                                                 \
  }
  // (the weight operand needs no mask: behind the K tail it re-reads k = 0 -- finite, and the A chunk there is zero --
  // and a column past N re-reads row 0 into an accumulator column the epilogue never stores)
#define MF_PP_REQ_W(sw_, i0_, i1_)                                                                    \
  {                                                                                                   \
    const uint32_t kofs_ = kgw + 8 <= a.K ? 2u * (uint32_t)kgw : 0u;                                  \
    _Pragma("unroll") for (int i = (i0_); i < (i1_); ++i)                                             \
      mf::glds16(Wrs, wrow[i] + kofs_, dma0 + kPpW0 + (sw_) * kPpOp + i * 64 * 128);                  \
    if ((i1_) == 4) kgw += kBK;                                                                       \
  }
  // fragment addresses inside a stage: row R = 128 wm + 32 mi + lrow of A (64 wn + 32 ni + lrow of W), chunk
  // (2 s + lhalf) ^ ((lrow >> 1) & 7) = ((s ^ (lrow >> 2 & 3)) << 1) | ((lhalf ^ (lrow >> 1)) & 1)
  const int gh = (lrow >> 2) & 3, c0 = ((lhalf ^ (lrow >> 1)) & 1) << 4;
  const mf::lds_addr_t fragA = mf::lds_addr(s_raw) + (128 * wm + lrow) * 128 + c0;
  const mf::lds_addr_t fragW = mf::lds_addr(s_raw) + kPpW0 + (64 * wn + lrow) * 128 + c0;
  const int ncols = a.N - (n0 + wn * 64);  // columns of this wave's 64 that exist (wave-uniform)
  // A phase = two k-steps of 16: twelve fragment reads and four DMA requests in its load section, sixteen MFMAs
  // (every accumulator twice, eight MFMAs apart) in its MFMA section.  (One k-step per phase -- eight barriers per
  // K-tile -- left the MFMA pipe at 0.55 of its peak even with NO DMA at all, whether the load section waited for its
  // own six reads or they were issued between the previous phase's MFMAs: the barrier hand-over itself, ~100 cycles
  // per 256 cycles of MFMAs.  MF_PP_DBG ablations, tools/ab_gemm.sh.)
  uint4 fa[2][4], fb[2][2];
#define MF_PP_READS(NJ_, kk_, sa_, sw_, s_, r0_, r1_)                                                  \
  if ((NJ_) > 0) {                                                                                    \
    const mf::lds_addr_t va_ = fragA + (sa_) * kPpOp + (((s_) ^ gh) << 5);                            \
    const mf::lds_addr_t vb_ = fragW + (sw_) * kPpOp + (((s_) ^ gh) << 5);                            \
    if ((r0_) <= 0 && 0 < (r1_)) fa[kk_][0] = mf::lds_read16_async<0>(va_);                           \
    if ((r0_) <= 1 && 1 < (r1_)) fb[kk_][0] = mf::lds_read16_async<0>(vb_);                           \
    if ((r0_) <= 2 && 2 < (r1_)) fa[kk_][1] = mf::lds_read16_async<4096>(va_);                        \
    if ((r0_) <= 3 && 3 < (r1_) && (NJ_) > 1) fb[kk_][1] = mf::lds_read16_async<4096>(vb_);           \
    if ((r0_) <= 4 && 4 < (r1_)) fa[kk_][2] = mf::lds_read16_async<8192>(va_);                        \
    if ((r0_) <= 5 && 5 < (r1_)) fa[kk_][3] = mf::lds_read16_async<12288>(va_);                       \
  }
  // NJ_ = the wave's 32-column blocks that exist (2, 1 or 0: wave-uniform, one loop per value).  A fragment that no
  // MFMA uses is NOT read: the compiler takes the asm's result register as written when the statement ends and hands
  // a dead one out again at once -- the data then lands on top of whatever lives there (seen: the offset of the next
  // DMA request, a memory fault).  MF_HOLD behind the wait keeps every fragment register reserved up to there.
  // (the load section alternating three reads and one request, and one or two of a phase's four requests issued
  // between its MFMAs instead, both measured slower: DESIGN.md 4)
#define MF_PP_PHASE(NJ_, p_, REQ_, WAIT_)                                                             \
  {                                                                                                   \
    MF_PP_READS(NJ_, 0, sa, sw, 2 * (p_), 0, 6)                                                       \
    MF_PP_READS(NJ_, 1, sa, sw, 2 * (p_) + 1, 0, 6)                                                   \
    REQ_(0, 4)                                                                                        \
    WAIT_                                                                                             \
    mf::wait_lds_reads();                                                                             \
    if ((NJ_) > 0) {                                                                                  \
      _Pragma("unroll") for (int kk = 0; kk < 2; ++kk) {                                              \
        _Pragma("unroll") for (int mi = 0; mi < 4; ++mi) MF_HOLD(fa[kk][mi]);                         \
        MF_HOLD(fb[kk][0]);                                                                           \
        if ((NJ_) > 1) MF_HOLD(fb[kk][1]);                                                            \
      }                                                                                               \
    }                                                                                                 \
    mf::raw_barrier();                                                                                \
    __builtin_amdgcn_s_setprio(1);                                                                    \
    _Pragma("unroll") for (int q = 0; q < 16; ++q) {                                                  \
      const int kk = q >> 3, mi = q & 3, nj = (q >> 2) & 1;                                           \
      if (nj < (NJ_)) acc[mi][nj] = mf::mfma_bf16_32x32x16(fa[kk][mi], fb[kk][nj], acc[mi][nj]);      \
    }                                                                                                 \
    __builtin_amdgcn_s_setprio(0);                                                                    \
    mf::raw_barrier();                                                                                \
  }
#define MF_PP_RW(i0_, i1_) if (more1) MF_PP_REQ_W(sw ^ 1, i0_, i1_)
#define MF_PP_RA(i0_, i1_) if (more2) MF_PP_REQ_A(sa2, i0_, i1_)
#define MF_PP_LOOP(NJ_)                                                                               \
  for (int t = 0; t < T; ++t) {                                                                       \
    const int sw = t & 1;                                                                             \
    const bool more1 = t + 1 < T && !(a.dbg & 1), more2 = t + 2 < T && !(a.dbg & 1);                  \
    if (a.dbg & 2) { kg = kgw = 8 * chunk; tc = kg; tx = ty = tz = 0; }                               \
    MF_PP_PHASE(NJ_, 0, MF_PP_RW, )                                                                   \
    MF_PP_PHASE(NJ_, 1, MF_PP_RA, if (more2) mf::wait_dma<4>(); else mf::wait_dma<0>();)              \
    sa = sa == 2 ? 0 : sa + 1;                                                                        \
    sa2 = sa2 == 2 ? 0 : sa2 + 1;                                                                     \
  }
  // tiles 0 (A, W) and 1 (A) before the loop; the requests of A(1) stay in flight
  MF_PP_REQ_A(0, 0, 4) MF_PP_REQ_W(0, 0, 4)
  if (T > 1 && !(a.dbg & 1)) {
    MF_PP_REQ_A(1, 0, 4)
    mf::wait_dma<4>();
  } else {
    mf::wait_dma<0>();
  }
  mf::raw_barrier();
  int sa = 0, sa2 = 2;  // A stages of tiles t and t + 2
  if (wm == 1 && !(a.dbg & 4)) mf::raw_barrier();  // the lower half runs one barrier behind from here on
  if (ncols > 32) {
    MF_PP_LOOP(2)
  } else if (ncols > 0) {
    MF_PP_LOOP(1)
  } else {
    MF_PP_LOOP(0)
  }
  if (wm == 0 && !(a.dbg & 4)) mf::raw_barrier();  // the groups meet again: every fragment read is retired, no DMA is pending
#undef MF_PP_LOOP
#undef MF_PP_RW
#undef MF_PP_RA
#undef MF_PP_READS
#undef MF_PP_PHASE
#undef MF_PP_REQ_W
#undef MF_PP_REQ_A

  // epilogue through LDS in four passes of 64 rows (64 x 260 floats)
  constexpr int kEp = kBNb + 4;
  float *s_out = reinterpret_cast<float *>(s_raw);  // [64][kEp]
  const float *bias = a.bias && a.S == 1 ? a.bias + grp * a.b_gs : nullptr;
  const bool relu = a.relu && a.S == 1, out_f32 = a.out_f32 || a.S > 1;
  const int ldo = a.S > 1 ? a.N : a.ldo;
  void *const outp = a.S > 1 ? (void *)(a.slab + (int64_t)split * a.M * a.N) : a.out;
#pragma unroll
  for (int pass = 0; pass < 4; ++pass) {
    if (wm == (pass >> 1)) {
#pragma unroll
      for (int mh = 0; mh < 2; ++mh)
#pragma unroll
        for (int ni = 0; ni < 2; ++ni) {
          const int nl = wn * 64 + ni * 32 + lrow;
          const float bn = (bias && n0 + nl < a.N) ? bias[n0 + nl] : 0.0f;
          const mf_f32x16 &c = (pass & 1) ? acc[2 + mh][ni] : acc[mh][ni];
#pragma unroll
          for (int e = 0; e < 16; ++e) {
            const int ml = mh * 32 + (e & 3) + 8 * (e >> 2) + 4 * lhalf;
            float v = c[e] + bn;
            if (relu) v = v > 0.0f ? v : 0.0f;
            s_out[ml * kEp + nl] = v;
          }
        }
    }
    __syncthreads();
    for (int i = tid; i < 64 * (kBNb / 8); i += 512) {
      const int ml = i / (kBNb / 8), c8 = i - ml * (kBNb / 8);
      const int m = m0 + 128 * (pass >> 1) + 64 * (pass & 1) + ml, n = n0 + 8 * c8;
      if (m >= a.M || n >= a.N) continue;
      int64_t orow = m;
      if (MODE == kConvDgrad) {  // class-ordered row -> channels-last voxel row of the input gradient
        const int h = m & ((1 << (3 * dol)) - 1), p = (m >> (3 * dol)) & 7, b = m >> (3 * dol + 3);
        const int x = 2 * (h >> (2 * dol)) + (p & 1), y = 2 * ((h >> dol) & (Do - 1)) + ((p >> 1) & 1),
                  z = 2 * (h & (Do - 1)) + (p >> 2);
        orow = (((int64_t)b * a.D + x) * a.D + y) * a.D + z;
      }
      const float4 v0 = *reinterpret_cast<const float4 *>(s_out + ml * kEp + 8 * c8);
      const float4 v1 = *reinterpret_cast<const float4 *>(s_out + ml * kEp + 8 * c8 + 4);
      float v[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
      const int nv = a.N - n < 8 ? a.N - n : 8;
      if (out_f32) {
        float *o = reinterpret_cast<float *>(outp) + grp * a.o_gs + orow * ldo + n;
        if (nv == 8 && (ldo & 3) == 0 && ((uintptr_t)o & 15) == 0) {
          float4 *o4 = reinterpret_cast<float4 *>(o);
          if (a.accumulate && a.S == 1) {
            const float4 p0 = o4[0], p1 = o4[1];
            v[0] += p0.x; v[1] += p0.y; v[2] += p0.z; v[3] += p0.w;
            v[4] += p1.x; v[5] += p1.y; v[6] += p1.z; v[7] += p1.w;
          }
          o4[0] = make_float4(v[0], v[1], v[2], v[3]);
          o4[1] = make_float4(v[4], v[5], v[6], v[7]);
        } else {
          for (int j = 0; j < nv; ++j) o[j] = a.accumulate && a.S == 1 ? o[j] + v[j] : v[j];
        }
      } else {
        uint16_t *o = reinterpret_cast<uint16_t *>(outp) + grp * a.o_gs + orow * ldo + n;
        if (nv == 8 && (ldo & 7) == 0 && ((uintptr_t)o & 15) == 0) {
          *reinterpret_cast<uint4 *>(o) = make_uint4(mf::pack_bf16x2(v[0], v[1]), mf::pack_bf16x2(v[2], v[3]),
                                                     mf::pack_bf16x2(v[4], v[5]), mf::pack_bf16x2(v[6], v[7]));
        } else {
          for (int j = 0; j < nv; ++j) o[j] = (uint16_t)mf::bf16_bits(v[j]);
        }
      }
    }
    __syncthreads();
  }
}

// ---- TN engine: C[i][j] = sum_m P[m][i] * Q(m, j) -------------------------------------------------------------
// Both operands arrive with the reduction index m as the SLOW dimension (rows of dY, rows of x / of the im2col view),
// the MFMA wants 8 consecutive m per lane.  The LDS image keeps the global order -- 16-byte chunks land with a plain
// ds_write_b128 -- and the fragments come out through gfx950's transposing LDS read (ds_read_b64_tr_b16: a 16-lane
// group reads a [4 m][16 columns] block, lane c receives column c).  Round 4's first version transposed 4 x 8 blocks
// in registers on the way in: 17 VALU instructions per MFMA and LDS bank conflicts on half of the LDS cycles.
//   image of one operand: 8 subtiles of 16 columns, each [64 m][16] bf16 (32 bytes per m) + 128 bytes, so that two
//   neighbouring subtiles -- the two 16-lane groups of a half-wave -- sit 32 banks apart
constexpr int kTnSub = 64 * 32 + 128;
constexpr int kTnOperand = 8 * kTnSub;
constexpr int kTnBuf = 2 * kTnOperand;
constexpr int kTnLds = 2 * kTnBuf > 128 * (128 + 4) * 4 ? 2 * kTnBuf : 128 * (128 + 4) * 4;

struct TnArgs {
  const uint16_t *P;  // bf16 [M][ldp]  (dY), group g at P + g * p_gs
  const uint16_t *Q;  // bf16 rows [M][ldq] (group g at Q + g * q_gs) or a channels-last grid [B][D^3][Cin] (conv)
  float *out;         // S == 1: C [Ni][ldc] (group g at out + g * c_gs); S > 1: slabs [S][groups][Ni][ldc]
  int64_t p_gs, q_gs, c_gs;
  int M, Ni, Nj, ldp, ldq, ldc, groups, S;
  const int32_t *m_range;  // rows mode, S == 1: group g reduces rows [m_range[g], m_range[g + 1]) of P / Q (device array)
  int conv, B, D, Do, olog, Cin, ks, stride, pad, dil;  // conv: Q(m, j = tap * Cin + cin) = x[b][stride o - pad + dil tap][cin], m = (b, o)
};

template <bool CONV>
__global__ __launch_bounds__(256, 2) void k_gemm_tn_bf16(TnArgs a) {
  MF_DYN_LDS(unsigned char, s_raw);
  const int tiles_i = (a.Ni + 127) / 128, tiles_j = (a.Nj + 127) / 128;
  const int per_group = tiles_i * tiles_j;
  const int G = gridDim.x;
  int L = blockIdx.x;
  if ((G & 7) == 0) L = (blockIdx.x & 7) * (G >> 3) + (blockIdx.x >> 3);
  const int split = L / (per_group * a.groups);
  const int rem0 = L - split * per_group * a.groups;
  const int grp = rem0 / per_group;
  const int rem = rem0 - grp * per_group;
  const int i0 = (rem % tiles_i) * 128, j0 = (rem / tiles_i) * 128;  // i tile fastest: neighbours share Q columns
  // this split's rows: K-tiles of 64 rows, contiguous ranges
  int m_lo = 0, M = a.M;
  if (!CONV && a.m_range) {  // (block-uniform)
    m_lo = a.m_range[grp];
    M = a.m_range[grp + 1] - m_lo;
  }
  const int Tall = (M + 63) / 64;
  const int Tper = (Tall + a.S - 1) / a.S;
  const int t0 = split * Tper, t1 = min(Tall, t0 + Tper);

  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int wm = wave & 1, wn = wave >> 1;
  const int lrow = lane & 31, lhalf = lane >> 5;
  // staging: load r (0..3) of this lane goes to LDS row 16 r + 4 wave + kr of the K-tile, columns 16 sub + 8 half .. + 7.  Eight
  // consecutive lanes (the unit a ds_write_b128 is served in) fill 4 rows x 32 bytes = 128 contiguous bytes of one
  // subtile: 32 distinct banks; a wave's load covers 4 rows x 256 contiguous bytes.
  const int half = lane & 1, kr = (lane >> 1) & 3, sub = lane >> 3;
  const int col = 16 * sub + 8 * half;
  const int st_off = sub * kTnSub + (4 * wave + kr) * 32 + 16 * half;  // + 512 r
  const int mrow = 16 * kr + 4 * wave;  // this lane's rows of a K-tile: mrow .. mrow + 3 (see MF_TN_LOAD)

  const uint16_t *P = a.P + grp * a.p_gs + (int64_t)m_lo * a.ldp;
  const uint16_t *Q = a.Q + grp * a.q_gs + (CONV ? 0 : (int64_t)m_lo * a.ldq);
  const int Do = a.Do, dol = a.olog;
  const bool pcol_ok = i0 + col + 8 <= a.Ni;
  // conv: this lane's column chunk is one (tap, cin .. cin + 7) for the whole loop, so per row only the output voxel
  // (b, ox, oy, oz) is decoded: the address is linear in it, and "the tap lies inside the grid" is one range test per
  // axis on the output coordinate (lo <= o <= lo + span, as one unsigned compare); a chunk past the last tap, or a
  // tap no output voxel can reach, is never valid
  const int q_off = j0 + col;
  int tap_const = 0, lo_x = 0, lo_y = 0, lo_z = 0;
  unsigned span_x = 0, span_y = 0, span_z = 0;
  bool qcol_ok = j0 + col + 8 <= a.Nj;
  const int cxs = a.stride * a.D * a.D * a.Cin, cys = a.stride * a.D * a.Cin, czs = a.stride * a.Cin;
  const int cb = a.D * a.D * a.D * a.Cin;
  if (CONV) {
    const int jj = qcol_ok ? j0 + col : 0;
    const int tap = jj / a.Cin, tap_c = jj - tap * a.Cin;
    const int kxy = tap / a.ks, kz = tap - kxy * a.ks, kx = kxy / a.ks, ky = kxy - kx * a.ks;
    const int tx = a.dil * kx - a.pad, ty = a.dil * ky - a.pad, tz = a.dil * kz - a.pad;
    tap_const = ((tx * a.D + ty) * a.D + tz) * a.Cin + tap_c;
    qcol_ok = qcol_ok && kx < a.ks;
    const int t3[3] = {tx, ty, tz};
    int lo3[3];
    unsigned sp3[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {  // 0 <= stride * o + t < D
      const int lo = t3[k] >= 0 ? 0 : (-t3[k] + a.stride - 1) / a.stride;
      const int hi = a.D - 1 - t3[k] >= 0 ? min((a.D - 1 - t3[k]) / a.stride, Do - 1) : -1;
      qcol_ok = qcol_ok && hi >= lo;
      lo3[k] = lo;
      sp3[k] = (unsigned)max(hi - lo, 0);
    }
    lo_x = lo3[0]; lo_y = lo3[1]; lo_z = lo3[2];
    span_x = sp3[0]; span_y = sp3[1]; span_z = sp3[2];
  }

  mf_f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.0f;

  // one register set, one tile ahead; a masked chunk (row past M, column past the edge, padding tap) is a buffer
  // load at an out-of-range offset and comes back as zeros (see the NT kernel)
  const mf::BufRsrc Prs = mf::make_rsrc(P), Qrs = mf::make_rsrc(Q);
  uint4 rp0, rp1, rp2, rp3, rq0, rq1, rq2, rq3;
  // Load r of this lane is row 16 kr + 4 wave + r of the K-tile (it lands in LDS row 16 r + 4 wave + kr: any
  // permutation of the reduction index is fine as long as both operands use it): the lane's four rows are consecutive,
  // so with an output size that is a multiple of 4 they share (b, ox, oy) and differ in oz only -- one voxel decode,
  // two range tests and one address per K-tile, then a z test and an add per load (the per-load decode was 20 of the
  // kernel's 29 VALU instructions per load: 7.3 per MFMA).
#define MF_TN_LOAD(r_, rp_, rq_)                                                                      \
  {                                                                                                   \
    const bool ok_ = mb_ + (r_) < M;                                                                \
    rp_ = mf::buf_load16(Prs, ok_ && pcol_ok ? 2u * (uint32_t)(pb_ + (r_) * a.ldp) : mf::kBufMasked); \
    bool qok_ = ok_ && qrow_ok_;                                                                      \
    if (CONV) qok_ = qok_ && (unsigned)(zrel_ + (r_)) <= span_z;                                      \
    rq_ = mf::buf_load16(Qrs, qok_ ? 2u * (uint32_t)(qb_ + (r_) * (CONV ? czs : a.ldq)) : mf::kBufMasked); \
  }
#define MF_TN_FETCH(tt_)                                                                              \
  {                                                                                                   \
    const int mb_ = (tt_) * 64 + mrow;                                                                \
    const int pb_ = mb_ * a.ldp + i0 + col;                                                           \
    int qb_ = mb_ * a.ldq + q_off, zrel_ = 0;                                                         \
    bool qrow_ok_ = qcol_ok;                                                                          \
    if (CONV) {                                                                                       \
      const int b_ = mb_ >> (3 * dol), ox_ = (mb_ >> (2 * dol)) & (Do - 1), oy_ = (mb_ >> dol) & (Do - 1), \
                oz_ = mb_ & (Do - 1);                                                                 \
      qrow_ok_ = qcol_ok && (unsigned)(ox_ - lo_x) <= span_x && (unsigned)(oy_ - lo_y) <= span_y;     \
      qb_ = tap_const + b_ * cb + ox_ * cxs + oy_ * cys + oz_ * czs;                                  \
      zrel_ = oz_ - lo_z;                                                                             \
    }                                                                                                 \
    MF_TN_LOAD(0, rp0, rq0) MF_TN_LOAD(1, rp1, rq1) MF_TN_LOAD(2, rp2, rq2) MF_TN_LOAD(3, rp3, rq3)   \
  }
#define MF_TN_STASH(buf_)                                                                             \
  {                                                                                                   \
    MF_HOLD(rp0); MF_HOLD(rp1); MF_HOLD(rp2); MF_HOLD(rp3);                                           \
    MF_HOLD(rq0); MF_HOLD(rq1); MF_HOLD(rq2); MF_HOLD(rq3);                                           \
    unsigned char *Ps_ = s_raw + (buf_) * kTnBuf + st_off;                                            \
    *reinterpret_cast<uint4 *>(Ps_) = rp0; *reinterpret_cast<uint4 *>(Ps_ + 512) = rp1;               \
    *reinterpret_cast<uint4 *>(Ps_ + 1024) = rp2; *reinterpret_cast<uint4 *>(Ps_ + 1536) = rp3;       \
    unsigned char *Qs_ = Ps_ + kTnOperand;                                                            \
    *reinterpret_cast<uint4 *>(Qs_) = rq0; *reinterpret_cast<uint4 *>(Qs_ + 512) = rq1;               \
    *reinterpret_cast<uint4 *>(Qs_ + 1024) = rq2; *reinterpret_cast<uint4 *>(Qs_ + 1536) = rq3;       \
  }
  // fragment of a 32-column block at subtile pair (2 n, 2 n + 1), k-step s: lane l = 16 g + c takes column c of
  // subtile 2 n + (g & 1), m = 16 s + 8 (g >> 1) + 0..3 (first read) and + 4..7 (second): the operand layout of
  // v_mfma_f32_32x32x16_bf16 (row l % 32, k = 8 (l / 32) .. + 7)
  const int frag = ((lane >> 4) & 1) * kTnSub + (8 * (lane >> 5) + ((lane & 15) >> 2)) * 32 + 8 * (lane & 3);
#define MF_TN_FRAG(ptr_, s_) mf::lds_read_tr16_b64x2((ptr_) + 512 * (s_), 128)
#define MF_TN_COMPUTE(buf_)                                                                           \
  {                                                                                                   \
    asm volatile("" ::: "memory");                                                                    \
    __builtin_amdgcn_sched_barrier(0);                                                                \
    const unsigned char *Ps = s_raw + (buf_) * kTnBuf + 4 * wm * kTnSub + frag;                       \
    const unsigned char *Qs = s_raw + (buf_) * kTnBuf + kTnOperand + 4 * wn * kTnSub + frag;          \
    _Pragma("unroll") for (int s = 0; s < 4; ++s) {                                                   \
      const uint4 a0 = MF_TN_FRAG(Ps, s), a1 = MF_TN_FRAG(Ps + 2 * kTnSub, s);                        \
      const uint4 b0 = MF_TN_FRAG(Qs, s), b1 = MF_TN_FRAG(Qs + 2 * kTnSub, s);                        \
      acc[0][0] = mf::mfma_bf16_32x32x16(a0, b0, acc[0][0]);                                          \
      acc[0][1] = mf::mfma_bf16_32x32x16(a0, b1, acc[0][1]);                                          \
      acc[1][0] = mf::mfma_bf16_32x32x16(a1, b0, acc[1][0]);                                          \
      acc[1][1] = mf::mfma_bf16_32x32x16(a1, b1, acc[1][1]);                                          \
    }                                                                                                 \
    __builtin_amdgcn_sched_barrier(0);                                                                \
  }
  // (the fetch past this split's last tile reads the next split's rows, or rows past M as zeros, into a buffer nobody
  // reads: unconditional on purpose, see the NT kernel)
  if (t0 < t1) {
    MF_TN_FETCH(t0);
    MF_TN_STASH(0);
  }
  __syncthreads();
  for (int t = t0; t < t1; ++t) {
    MF_TN_FETCH(t + 1);
    MF_TN_COMPUTE((t - t0) & 1);
    MF_TN_STASH((t - t0 + 1) & 1);
    __syncthreads();
  }
#undef MF_TN_COMPUTE
#undef MF_TN_FRAG
#undef MF_TN_STASH
#undef MF_TN_FETCH
#undef MF_TN_LOAD

  constexpr int kEp = 128 + 4;
  float *s_out = reinterpret_cast<float *>(s_raw);
#pragma unroll
  for (int mi = 0; mi < 2; ++mi)
#pragma unroll
    for (int ni = 0; ni < 2; ++ni) {
      const int nl = wn * 64 + ni * 32 + lrow;
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int ml = wm * 64 + mi * 32 + (e & 3) + 8 * (e >> 2) + 4 * lhalf;
        s_out[ml * kEp + nl] = acc[mi][ni][e];
      }
    }
  __syncthreads();
  float *dst = a.out + ((int64_t)split * a.groups + grp) * (a.S > 1 ? (int64_t)a.Ni * a.ldc : 0) +
               (a.S > 1 ? 0 : grp * a.c_gs);
  for (int i = tid; i < 128 * 32; i += 256) {
    const int il = i >> 5, c4 = i & 31;
    const int ii = i0 + il, jj = j0 + 4 * c4;
    if (ii >= a.Ni || jj >= a.Nj) continue;
    const float4 v = *reinterpret_cast<const float4 *>(s_out + il * kEp + 4 * c4);
    float *o = dst + (int64_t)ii * a.ldc + jj;
    if (jj + 4 <= a.Nj && (a.ldc & 3) == 0 && ((uintptr_t)o & 15) == 0) {
      *reinterpret_cast<float4 *>(o) = v;
    } else {
      const float vv[4] = {v.x, v.y, v.z, v.w};
      for (int j = 0; j < 4 && jj + j < a.Nj; ++j) o[j] = vv[j];
    }
  }
}

// ---- the TN engine on the 256 x 256 tile with LDS-DMA operands and two wave groups in ping-pong (round 6) -----------
// The structure of k_gemm_nt_bf16_pp (see there: phases of two k-steps, the groups one barrier apart, counted vmcnt,
// inline-asm fragment reads) for C[i][j] = sum_m P[m][i] Q(m, j): a K-tile is 64 rows m of both operands, 256 columns
// each -- [64][512 bytes] per operand and stage, THREE Q stages (the im2col rows: re-read from far away) + TWO P stages
// (dY: shared by every workgroup of a column of tiles).
//   * DMA: one request = two LDS rows (2 x 512 bytes: whole contiguous row segments of the tile).  LDS row
//     rho = 16 i + 2 wave + h (request i of the wave, h = lane / 32) holds global row 64 t + 8 wave + 4 h + i: any
//     permutation of the reduction index is fine as long as both operands use it, and this one gives a lane four
//     CONSECUTIVE rows per K-tile -- for a convolution one voxel decode, an oz test and an add per request.
//   * fragments: ds_read_b64_tr_b16 (a 16-lane group reads a [4 rows][16 columns] block, lane c receives column c):
//     the four rows of a group are 512 bytes apart -- the same banks -- so the 16-byte chunk index of LDS row rho is
//     XORed with 4 (rho & 3) (applied to the SOURCE column of the DMA lane, as in the NT kernel): the four rows of a
//     group land in the four 64-byte quarters of the 256-byte bank row.  rho & 3 is the lane's r = (lane & 15) / 4 in
//     every read, so a lane's address of column block n is (n ^ r) * 64 + const: one address register per block.

template <bool CONV>
__global__ __launch_bounds__(512, 2) void k_gemm_tn_bf16_pp(TnArgs a) {
  MF_DYN_LDS(unsigned char, s_raw);
  const int tiles_i = (a.Ni + 255) / 256, tiles_j = (a.Nj + 255) / 256;
  const int per_group = tiles_i * tiles_j;
  const int G = gridDim.x;
  int L = blockIdx.x;
  if ((G & 7) == 0) L = (blockIdx.x & 7) * (G >> 3) + (blockIdx.x >> 3);
  const int split = L / (per_group * a.groups);
  const int rem0 = L - split * per_group * a.groups;
  const int grp = rem0 / per_group;
  const int rem = rem0 - grp * per_group;
  const int i0 = (rem % tiles_i) * 256, j0 = (rem / tiles_i) * 256;  // i tile fastest: neighbours share Q columns
  int m_lo = 0, M = a.M;
  if (!CONV && a.m_range) {  // (block-uniform)
    m_lo = a.m_range[grp];
    M = a.m_range[grp + 1] - m_lo;
  }
  const int Tall = (M + 63) / 64;
  const int Tper = (Tall + a.S - 1) / a.S;
  const int t0 = split * Tper;
  const int T = max(0, min(Tall, t0 + Tper) - t0);

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = mf::wave_uniform(tid >> 6);
  const int wm = wave >> 2, wn = wave & 3;
  const int lrow = lane & 31, lhalf = lane >> 5;
  // DMA slot: LDS rows 16 i + 2 wave + h, chunk position lane & 31 -> this lane's column chunk (the same for every row)
  const int h = lane >> 5;
  const int cq = (lane & 31) ^ (4 * ((2 * wave + h) & 3));
  const int col = 8 * cq;
  const int mrow = 8 * wave + 4 * h;  // this lane's rows of a K-tile: mrow + i

  const uint16_t *P = a.P + grp * a.p_gs + (int64_t)m_lo * a.ldp;
  const uint16_t *Q = a.Q + grp * a.q_gs + (CONV ? 0 : (int64_t)m_lo * a.ldq);
  const int Do = a.Do, dol = a.olog;
  const bool pcol_ok = i0 + col + 8 <= a.Ni;
  // conv: the lane's column chunk is one (tap, cin .. cin + 7) for the whole loop (see k_gemm_tn_bf16)
  const int q_off = j0 + col;
  int tap_const = 0, lo_x = 0, lo_y = 0, lo_z = 0;
  unsigned span_x = 0, span_y = 0, span_z = 0;
  bool qcol_ok = j0 + col + 8 <= a.Nj;
  const int cxs = a.stride * a.D * a.D * a.Cin, cys = a.stride * a.D * a.Cin, czs = a.stride * a.Cin;
  const int cb = a.D * a.D * a.D * a.Cin;
  if (CONV) {
    const int jj = qcol_ok ? j0 + col : 0;
    const int tap = jj / a.Cin, tap_c = jj - tap * a.Cin;
    const int kxy = tap / a.ks, kz = tap - kxy * a.ks, kx = kxy / a.ks, ky = kxy - kx * a.ks;
    const int tx = a.dil * kx - a.pad, ty = a.dil * ky - a.pad, tz = a.dil * kz - a.pad;
    tap_const = ((tx * a.D + ty) * a.D + tz) * a.Cin + tap_c;
    qcol_ok = qcol_ok && kx < a.ks;
    const int t3[3] = {tx, ty, tz};
    int lo3[3];
    unsigned sp3[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {  // 0 <= stride * o + t < D
      const int lo = t3[k] >= 0 ? 0 : (-t3[k] + a.stride - 1) / a.stride;
      const int hi = a.D - 1 - t3[k] >= 0 ? min((a.D - 1 - t3[k]) / a.stride, Do - 1) : -1;
      qcol_ok = qcol_ok && hi >= lo;
      lo3[k] = lo;
      sp3[k] = (unsigned)max(hi - lo, 0);
    }
    lo_x = lo3[0]; lo_y = lo3[1]; lo_z = lo3[2];
    span_x = sp3[0]; span_y = sp3[1]; span_z = sp3[2];
  }

  mf_f32x16 acc[4][2];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.0f;

  const mf::BufRsrc Prs = mf::make_rsrc(P), Qrs = mf::make_rsrc(Q);
  unsigned char *const dma0 = s_raw + wave * 1024;  // request i of this wave fills LDS rows 16 i + 2 wave, + 1
  constexpr int kQ0 = 0, kP0 = 3 * kPpOp;            // Q stages at 0, 1, 2 x kPpOp; P stages behind them
  int tq = t0, tp = t0;                              // K-tiles the next Q / P requests fetch (Q runs one ahead)
  // requests i0_ .. i1_ - 1 of Q's next K-tile -> Q stage sq_ (rows 64 tq + mrow + i); the tile advances behind the last
#define MF_TP_REQ_Q(sq_, i0_, i1_)                                                                    \
  {                                                                                                   \
    const int mb_ = tq * 64 + mrow;                                                                   \
    int qb_ = mb_ * a.ldq + q_off, zrel_ = 0;                                                         \
    bool qrow_ok_ = qcol_ok;                                                                          \
    if (CONV) {                                                                                       \
      const int b_ = mb_ >> (3 * dol), ox_ = (mb_ >> (2 * dol)) & (Do - 1), oy_ = (mb_ >> dol) & (Do - 1), \
                oz_ = mb_ & (Do - 1);                                                                 \
      qrow_ok_ = qcol_ok && (unsigned)(ox_ - lo_x) <= span_x && (unsigned)(oy_ - lo_y) <= span_y;     \
      qb_ = tap_const + b_ * cb + ox_ * cxs + oy_ * cys + oz_ * czs;                                  \
      zrel_ = oz_ - lo_z;                                                                             \
    }                                                                                                 \
    _Pragma("unroll") for (int i = (i0_); i < (i1_); ++i) {                                           \
      bool ok_ = mb_ + i < M && qrow_ok_;                                                             \
      if (CONV) ok_ = ok_ && (unsigned)(zrel_ + i) <= span_z;                                         \
      mf::glds16(Qrs, ok_ ? 2u * (uint32_t)(qb_ + i * (CONV ? czs : a.ldq)) : mf::kBufMasked,         \
                 dma0 + kQ0 + (sq_) * kPpOp + i * 8192);                                              \
    }                                                                                                 \
    if ((i1_) == 4) ++tq;                                                                             \
  }
#define MF_TP_REQ_P(sp_, i0_, i1_)                                                                    \
  {                                                                                                   \
    const int mb_ = tp * 64 + mrow;                                                                   \
    const int pb_ = mb_ * a.ldp + i0 + col;                                                           \
    _Pragma("unroll") for (int i = (i0_); i < (i1_); ++i)                                             \
      mf::glds16(Prs, mb_ + i < M && pcol_ok ? 2u * (uint32_t)(pb_ + i * a.ldp) : mf::kBufMasked,     \
                 dma0 + kP0 + (sp_) * kPpOp + i * 8192);                                              \
    if ((i1_) == 4) ++tp;                                                                             \
  }
  // fragment addresses inside a stage (k-step s: + s * 8192, second half of a fragment: + 2048)
  const int r = (lane & 15) >> 2, q4 = lane & 3, g = lane >> 4;
  const int rowpart = (8 * (g >> 1) + r) * 512 + (2 * (g & 1) + (q4 >> 1)) * 16 + 8 * (q4 & 1);
  mf::lds_addr_t fragP[4], fragQ[2];
#pragma unroll
  for (int mi = 0; mi < 4; ++mi) fragP[mi] = mf::lds_addr(s_raw) + kP0 + rowpart + ((4 * wm + mi) ^ r) * 64;
#pragma unroll
  for (int ni = 0; ni < 2; ++ni) fragQ[ni] = mf::lds_addr(s_raw) + kQ0 + rowpart + ((2 * wn + ni) ^ r) * 64;
  const int ncols = a.Nj - (j0 + wn * 64);  // columns of this wave's 64 that exist (wave-uniform)
  uint4 fa[2][4], fb[2][2];
  // (a fragment is two 8-byte transposing reads into the halves of one 16-byte register: mf::lds_read_tr16_x2_async)
#define MF_TP_READS(NJ_, kk_, sq_, sp_, s_)                                                           \
  if ((NJ_) > 0) {                                                                                    \
    _Pragma("unroll") for (int mi = 0; mi < 4; ++mi)                                                  \
      fa[kk_][mi] = mf::lds_read_tr16_x2_async<(s_) * 8192>(fragP[mi] + (sp_) * kPpOp);               \
    fb[kk_][0] = mf::lds_read_tr16_x2_async<(s_) * 8192>(fragQ[0] + (sq_) * kPpOp);                   \
    if ((NJ_) > 1) fb[kk_][1] = mf::lds_read_tr16_x2_async<(s_) * 8192>(fragQ[1] + (sq_) * kPpOp);    \
  }
#define MF_TP_PHASE(NJ_, p_, REQ_, WAIT_)                                                             \
  {                                                                                                   \
    MF_TP_READS(NJ_, 0, sq, sp, 2 * (p_))                                                             \
    MF_TP_READS(NJ_, 1, sq, sp, 2 * (p_) + 1)                                                         \
    REQ_(0, 4)                                                                                        \
    WAIT_                                                                                             \
    mf::wait_lds_reads();                                                                             \
    if ((NJ_) > 0) {                                                                                  \
      _Pragma("unroll") for (int kk = 0; kk < 2; ++kk) {                                              \
        _Pragma("unroll") for (int mi = 0; mi < 4; ++mi) MF_HOLD(fa[kk][mi]);                         \
        MF_HOLD(fb[kk][0]);                                                                           \
        if ((NJ_) > 1) MF_HOLD(fb[kk][1]);                                                            \
      }                                                                                               \
    }                                                                                                 \
    mf::raw_barrier();                                                                                \
    __builtin_amdgcn_s_setprio(1);                                                                    \
    _Pragma("unroll") for (int qq = 0; qq < 16; ++qq) {                                               \
      const int kk = qq >> 3, mi = qq & 3, nj = (qq >> 2) & 1;                                        \
      if (nj < (NJ_)) acc[mi][nj] = mf::mfma_bf16_32x32x16(fa[kk][mi], fb[kk][nj], acc[mi][nj]);      \
    }                                                                                                 \
    __builtin_amdgcn_s_setprio(0);                                                                    \
    mf::raw_barrier();                                                                                \
  }
#define MF_TP_RP(i0_, i1_) if (more1) MF_TP_REQ_P(sp ^ 1, i0_, i1_)
#define MF_TP_RQ(i0_, i1_) if (more2) MF_TP_REQ_Q(sq2, i0_, i1_)
#define MF_TP_LOOP(NJ_)                                                                               \
  for (int t = 0; t < T; ++t) {                                                                       \
    const int sp = t & 1;                                                                             \
    const bool more1 = t + 1 < T, more2 = t + 2 < T;                                                  \
    MF_TP_PHASE(NJ_, 0, MF_TP_RP, )                                                                   \
    MF_TP_PHASE(NJ_, 1, MF_TP_RQ, if (more2) mf::wait_dma<4>(); else mf::wait_dma<0>();)              \
    sq = sq == 2 ? 0 : sq + 1;                                                                        \
    sq2 = sq2 == 2 ? 0 : sq2 + 1;                                                                     \
  }
  // tiles 0 (Q, P) and 1 (Q) before the loop; the requests of Q(1) stay in flight
  MF_TP_REQ_Q(0, 0, 4) MF_TP_REQ_P(0, 0, 4)
  if (T > 1) {
    MF_TP_REQ_Q(1, 0, 4)
    mf::wait_dma<4>();
  } else {
    mf::wait_dma<0>();
  }
  mf::raw_barrier();
  int sq = 0, sq2 = 2;  // Q stages of tiles t and t + 2
  if (wm == 1) mf::raw_barrier();  // the lower half runs one barrier behind from here on
  if (ncols > 32) {
    MF_TP_LOOP(2)
  } else if (ncols > 0) {
    MF_TP_LOOP(1)
  } else {
    MF_TP_LOOP(0)
  }
  if (wm == 0) mf::raw_barrier();  // the groups meet again
#undef MF_TP_LOOP
#undef MF_TP_RQ
#undef MF_TP_RP
#undef MF_TP_PHASE
#undef MF_TP_READS
#undef MF_TP_REQ_P
#undef MF_TP_REQ_Q

  // epilogue through LDS in four passes of 64 rows (fp32 tile rows i, columns j)
  constexpr int kEp = 256 + 4;
  float *s_out = reinterpret_cast<float *>(s_raw);  // [64][kEp]
  float *dst = a.out + ((int64_t)split * a.groups + grp) * (a.S > 1 ? (int64_t)a.Ni * a.ldc : 0) +
               (a.S > 1 ? 0 : grp * a.c_gs);
#pragma unroll
  for (int pass = 0; pass < 4; ++pass) {
    if (wm == (pass >> 1)) {
#pragma unroll
      for (int mh = 0; mh < 2; ++mh)
#pragma unroll
        for (int ni = 0; ni < 2; ++ni) {
          const int nl = wn * 64 + ni * 32 + lrow;
          const mf_f32x16 &c = (pass & 1) ? acc[2 + mh][ni] : acc[mh][ni];
#pragma unroll
          for (int e = 0; e < 16; ++e) {
            const int ml = mh * 32 + (e & 3) + 8 * (e >> 2) + 4 * lhalf;
            s_out[ml * kEp + nl] = c[e];
          }
        }
    }
    __syncthreads();
    for (int i = tid; i < 64 * 64; i += 512) {
      const int il = i >> 6, c4 = i & 63;
      const int ii = i0 + 64 * pass + il, jj = j0 + 4 * c4;
      if (ii >= a.Ni || jj >= a.Nj) continue;
      const float4 v = *reinterpret_cast<const float4 *>(s_out + il * kEp + 4 * c4);
      float *o = dst + (int64_t)ii * a.ldc + jj;
      if (jj + 4 <= a.Nj && (a.ldc & 3) == 0 && ((uintptr_t)o & 15) == 0) {
        *reinterpret_cast<float4 *>(o) = v;
      } else {
        const float vv[4] = {v.x, v.y, v.z, v.w};
        for (int j = 0; j < 4 && jj + j < a.Nj; ++j) o[j] = vv[j];
      }
    }
    __syncthreads();
  }
}

// out[g][i][f(j)] = sum_s slab[s][g][i][j] (increasing s); conv: j = tap * Cin + cin -> f(j) = cin * taps + tap
// (the torch / Chainer ConvolutionND weight layout [Cout][w_cin][ks][ks][ks]); channels cin >= cin_keep (the zero
// padding of a narrow layer's input up to 8 channels) are dropped.
__global__ __launch_bounds__(256) void k_wgrad_finish(const float *__restrict__ slabs, float *__restrict__ out,
                                                      int64_t per_slab, int Nj, int ldc, int S, int conv_cin,
                                                      int64_t c_gs, int64_t per_group, int64_t out_row_pitch,
                                                      int taps, int cin_keep) {
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= per_slab) return;
  const int64_t g = idx / per_group, in_g = idx - g * per_group;
  const int64_t i = in_g / ldc;
  const int j = (int)(in_g - i * ldc);
  if (j >= Nj) return;
  float v = slabs[idx];
  for (int s = 1; s < S; ++s) v += slabs[(int64_t)s * per_slab + idx];
  int64_t o = j;
  if (conv_cin) {
    const int tap = j / conv_cin, ci = j - tap * conv_cin;
    if (ci >= cin_keep) return;
    o = (int64_t)ci * taps + tap;
  }
  out[g * c_gs + i * out_row_pitch + o] = v;
}

// The same sum for MANY slabs of a SMALL result (the occupancy convolutions: a few thousand weights in up to 256
// slabs -- one thread per weight walked 256 dependent-latency loads: 62 us): one WAVE per weight, lane l adds slabs
// l, l + 64, ... in increasing order, the 64 partial sums meet in a fixed butterfly (deterministic).
__global__ __launch_bounds__(256) void k_wgrad_finish_deep(const float *__restrict__ slabs, float *__restrict__ out,
                                                           int64_t per_slab, int Nj, int ldc, int S, int conv_cin,
                                                           int64_t c_gs, int64_t per_group, int64_t out_row_pitch,
                                                           int taps, int cin_keep) {
  const int64_t idx = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (idx >= per_slab) return;  // wave-uniform
  const int lane = threadIdx.x & 63;
  float v = 0.0f;
  for (int s = lane; s < S; s += 64) v += slabs[(int64_t)s * per_slab + idx];
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d);
  if (lane != 0) return;
  const int64_t g = idx / per_group, in_g = idx - g * per_group;
  const int64_t i = in_g / ldc;
  const int j = (int)(in_g - i * ldc);
  if (j >= Nj) return;
  int64_t o = j;
  if (conv_cin) {
    const int tap = j / conv_cin, ci = j - tap * conv_cin;
    if (ci >= cin_keep) return;
    o = (int64_t)ci * taps + tap;
  }
  out[g * c_gs + i * out_row_pitch + o] = v;
}

// The convolution form of the finish pass as a tiled transpose: workgroup (ci block of 64, co) sums the slabs'
// [tap][cin] tile -- rows of 64 consecutive cin, coalesced -- into LDS and writes it out as [cin][tap], the
// framework's order, contiguous again.  (The element-wise form above would write 4-byte values ``taps`` floats apart:
// 0.3 ms per training step over the six convolution layers.)
constexpr int kPackTile = 64;  // channels per tile; taps <= 64 (kernel 3 or 4)

__global__ __launch_bounds__(256) void k_wgrad_finish_conv(const float *__restrict__ slabs, float *__restrict__ out,
                                                           int64_t per_slab, int S, int Cin, int taps, int w_cin,
                                                           int keep) {
  __shared__ float s_t[kPackTile][kPackTile + 1];
  const int ci0 = blockIdx.x * kPackTile, co = blockIdx.y;
  const float *src = slabs + (int64_t)co * taps * Cin;
  for (int i = threadIdx.x; i < taps * kPackTile; i += 256) {
    const int tap = i >> 6, cl = i & 63;
    float v = 0.0f;
    if (ci0 + cl < Cin) {
      v = src[(int64_t)tap * Cin + ci0 + cl];
      for (int s = 1; s < S; ++s) v += src[(int64_t)s * per_slab + (int64_t)tap * Cin + ci0 + cl];
    }
    s_t[cl][tap] = v;
  }
  __syncthreads();
  float *dst = out + ((int64_t)co * w_cin + ci0) * taps;  // (out already points at channel c_off)
  const int nci = min(kPackTile, keep - ci0);             // channels >= keep: the zero padding of a narrow input
  for (int i = threadIdx.x; i < nci * taps; i += 256) {
    const int cl = i / taps, tap = i - cl * taps;
    dst[i] = s_t[cl][tap];
  }
}

// ---- operand preparation ---------------------------------------------------------------------------------
// fp32 [rows][src_ld] -> bf16 [rows][dst_ld] (zero columns beyond ``cols``), 8 elements per lane
__global__ __launch_bounds__(256) void k_cast_rows_bf16(const float *__restrict__ src, int64_t src_ld,
                                                        uint16_t *__restrict__ dst, int64_t dst_ld, int64_t rows,
                                                        int cols) {
  const int per_row = (int)(dst_ld / 8);
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= rows * per_row) return;
  const int64_t r = i / per_row;
  const int c0 = (int)(i - r * per_row) * 8;
  float v[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) v[j] = c0 + j < cols ? src[r * src_ld + c0 + j] : 0.0f;
  *reinterpret_cast<uint4 *>(dst + r * dst_ld + c0) =
      make_uint4(mf::pack_bf16x2(v[0], v[1]), mf::pack_bf16x2(v[2], v[3]), mf::pack_bf16x2(v[4], v[5]),
                 mf::pack_bf16x2(v[6], v[7]));
}

// W [Cout][w_cin][ks][ks][ks] fp32 (input channels c_off .. c_off + Cin; channels past w_cin read as zero) ->
//   fwd   [Cout][tap][Cin]            (k = tap * Cin + cin)                                  taps = ks^3
//   dgrad [class p][Cin][slot][Cout]  (k4 / s2 / p1 only: k = slot * Cout + cout; tap = (1 - p) + 2 s per axis)
//   flipT [Cin][tap][Cout]            the forward operand of the DATA-GRADIENT convolution of a stride-1 layer:
//                                     dx = conv(dy, flipT), flipT[ci][tap][co] = W[co][ci][ks^3 - 1 - tap]
// Tiled transposes through LDS (round 4's first, element-wise pack read 4-byte values ``taps`` floats apart: 0.1 ms
// for conv4's 8.4 M weights, every training step):
//   k_conv_pack_fwd_tile   workgroup (ci block, co): W[co][c_off + ci ..][taps] -> fwd[co][tap][ci ..]
//   k_conv_pack_cof_tile   workgroup (co block, ci): W[co ..][c_off + ci][taps] -> flipT[ci][tap][co ..] and / or
//                          dgrad[p][ci][slot][co ..]   (the layouts with the OUTPUT channel fastest)
__global__ __launch_bounds__(256) void k_conv_pack_fwd_tile(const float *__restrict__ W, int Cin, int w_cin, int c_off,
                                                            int taps, uint16_t *__restrict__ fwd) {
  __shared__ float s_t[kPackTile][kPackTile + 1];
  const int ci0 = blockIdx.x * kPackTile, co = blockIdx.y;
  const int nci = min(kPackTile, Cin - ci0), live = max(0, min(nci, w_cin - c_off - ci0));
  const float *src = W + ((int64_t)co * w_cin + c_off + ci0) * taps;  // [ci][tap], contiguous
  for (int i = threadIdx.x; i < nci * taps; i += 256) {
    const int cl = i / taps, tap = i - cl * taps;
    s_t[cl][tap] = cl < live ? src[i] : 0.0f;
  }
  __syncthreads();
  uint16_t *dst = fwd + (int64_t)co * taps * Cin + ci0;
  for (int i = threadIdx.x; i < taps * kPackTile; i += 256) {
    const int tap = i >> 6, cl = i & 63;
    if (cl < nci) dst[(int64_t)tap * Cin + cl] = (uint16_t)mf::bf16_bits(s_t[cl][tap]);
  }
}

__global__ __launch_bounds__(256) void k_conv_pack_cof_tile(const float *__restrict__ W, int Cout, int Cin, int w_cin,
                                                            int c_off, int ks, uint16_t *__restrict__ dgrad,
                                                            uint16_t *__restrict__ flipT) {
  __shared__ float s_t[kPackTile][kPackTile + 1];
  const int taps = ks * ks * ks;
  const int co0 = blockIdx.x * kPackTile, ci = blockIdx.y;
  const int nco = min(kPackTile, Cout - co0);
  const bool live = c_off + ci < w_cin;
  for (int i = threadIdx.x; i < nco * taps; i += 256) {
    const int cl = i / taps, tap = i - cl * taps;
    s_t[cl][tap] = live ? W[((int64_t)(co0 + cl) * w_cin + c_off + ci) * taps + tap] : 0.0f;
  }
  __syncthreads();
  for (int i = threadIdx.x; i < taps * kPackTile; i += 256) {
    const int t = i >> 6, cl = i & 63;
    if (cl >= nco) continue;
    if (flipT) flipT[((int64_t)ci * taps + t) * Cout + co0 + cl] = (uint16_t)mf::bf16_bits(s_t[cl][taps - 1 - t]);
    if (dgrad) {  // t = 8 p + slot (k4 / s2 / p1 only: 64 taps)
      const int p = t >> 3, slot = t & 7;
      const int kx = (1 - (p & 1)) + 2 * (slot & 1), ky = (1 - ((p >> 1) & 1)) + 2 * ((slot >> 1) & 1),
                kz = (1 - (p >> 2)) + 2 * (slot >> 2);
      dgrad[(((int64_t)p * Cin + ci) * 8 + slot) * Cout + co0 + cl] = (uint16_t)mf::bf16_bits(s_t[cl][kx * 16 + ky * 4 + kz]);
    }
  }
}

// dz = dy where y > 0 else 0 (the ReLU behind a fused GEMM epilogue), bf16 in / out, 8 elements per lane.
// ``dy32``: an fp32 gradient instead (the accumulated gradient of a sampled grid).
__global__ __launch_bounds__(256) void k_relu_mask_bf16(const uint16_t *__restrict__ y, const uint16_t *__restrict__ dy,
                                                        const float *__restrict__ dy32, uint16_t *__restrict__ dz,
                                                        int64_t n8) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n8) return;
  const uint4 yv = reinterpret_cast<const uint4 *>(y)[i];
  const uint32_t yw[4] = {yv.x, yv.y, yv.z, yv.w};
  uint32_t ow[4];
  if (dy32) {
    const float4 g0 = reinterpret_cast<const float4 *>(dy32)[2 * i], g1 = reinterpret_cast<const float4 *>(dy32)[2 * i + 1];
    const float g[8] = {g0.x, g0.y, g0.z, g0.w, g1.x, g1.y, g1.z, g1.w};
#pragma unroll
    for (int d = 0; d < 4; ++d)
      ow[d] = mf::pack_bf16x2(mf::bf16_lo(yw[d]) > 0.0f ? g[2 * d] : 0.0f, mf::bf16_hi(yw[d]) > 0.0f ? g[2 * d + 1] : 0.0f);
  } else {
    const uint4 gv = reinterpret_cast<const uint4 *>(dy)[i];
    const uint32_t gw[4] = {gv.x, gv.y, gv.z, gv.w};
#pragma unroll
    for (int d = 0; d < 4; ++d)
      ow[d] = (mf::bf16_lo(yw[d]) > 0.0f ? gw[d] & 0xffffu : 0u) | (mf::bf16_hi(yw[d]) > 0.0f ? gw[d] & 0xffff0000u : 0u);
  }
  reinterpret_cast<uint4 *>(dz)[i] = make_uint4(ow[0], ow[1], ow[2], ow[3]);
}

// out[m][n] = act(sum_s slab[s][m][n] + bias[n]) (increasing s: deterministic), bf16 or fp32 rows of pitch ldo: the
// second half of a split-K launch of k_gemm_nt_bf16_pp.  Eight columns per lane (N % 8 == 0).
__global__ __launch_bounds__(256) void k_splitk_finish(const float *__restrict__ slab, const float *__restrict__ bias,
                                                       void *__restrict__ out, int64_t M, int N, int S, int ldo,
                                                       int relu, int out_f32) {
  const int n8 = N >> 3;
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= M * n8) return;
  const int64_t m = i / n8;
  const int n = (int)(i - m * n8) * 8;
  const float4 *src = reinterpret_cast<const float4 *>(slab + m * N + n);
  float4 a0 = src[0], a1 = src[1];
  for (int s = 1; s < S; ++s) {
    const float4 *p = reinterpret_cast<const float4 *>(slab + (int64_t)s * M * N + m * N + n);
    const float4 b0 = p[0], b1 = p[1];
    a0.x += b0.x; a0.y += b0.y; a0.z += b0.z; a0.w += b0.w;
    a1.x += b1.x; a1.y += b1.y; a1.z += b1.z; a1.w += b1.w;
  }
  float v[8] = {a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w};
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    if (bias) v[j] += bias[n + j];
    if (relu) v[j] = v[j] > 0.0f ? v[j] : 0.0f;
  }
  if (out_f32) {
    float *o = reinterpret_cast<float *>(out) + m * ldo + n;
    if ((ldo & 3) == 0 && ((uintptr_t)o & 15) == 0) {
      reinterpret_cast<float4 *>(o)[0] = make_float4(v[0], v[1], v[2], v[3]);
      reinterpret_cast<float4 *>(o)[1] = make_float4(v[4], v[5], v[6], v[7]);
    } else {
      for (int j = 0; j < 8; ++j) o[j] = v[j];
    }
  } else {
    uint16_t *o = reinterpret_cast<uint16_t *>(out) + m * ldo + n;
    if ((ldo & 7) == 0 && ((uintptr_t)o & 15) == 0) {
      *reinterpret_cast<uint4 *>(o) = make_uint4(mf::pack_bf16x2(v[0], v[1]), mf::pack_bf16x2(v[2], v[3]),
                                                 mf::pack_bf16x2(v[4], v[5]), mf::pack_bf16x2(v[6], v[7]));
    } else {
      for (int j = 0; j < 8; ++j) o[j] = (uint16_t)mf::bf16_bits(v[j]);
    }
  }
}

// ---- 3 x 3 x 3 convolutions between NARROW layers (round 6): the occupancy branch ----------------------------------
// conv1_occ (1 -> 8, fed as 8 channels), conv2_occ (8 -> 16, dilation 2) and conv2_occ's data gradient (16 -> 8)
// (model.py:69-72,120-124) went through k_gemm_nt_bf16<conv forward>: 8 or 16 valid columns of a 128-column tile,
// 92-99 us each at 16 objects for 8-16 MB of operands.  Here the convolution is out^T = W (x) im2col with the VOXELS as
// the MFMA's columns: a wave owns 32 voxels, its B operand of k-step s is one 16-byte global load per lane -- the 8
// channels [c0, c0 + 8) of tap (16 s + 8 (lane / 32)) / CI of the voxel lane % 32, a masked (out-of-range) buffer load
// for padding taps -- with no LDS stage at all (neighbouring voxels re-read the same 16 bytes from L1 / L2); the A
// operand, the weights [n][k = tap * CI + ci] of <= 32 output channels, stays in registers for all the tiles a wave
// walks (KS x 16 bytes per lane).  The accumulator's rows are channels: lane (voxel v, half h) ends up with channels
// {0..3, 8..11} + 4 h of its voxel -> two 8-byte stores.
template <int CI, int KS>  // KS = ceil(27 CI / 16) k-steps
__global__ __launch_bounds__(256) void k_conv_k3_narrow_bf16(const uint16_t *__restrict__ x, const uint16_t *__restrict__ wp,
                                                            const float *__restrict__ bias, uint16_t *__restrict__ out,
                                                            int B, int D, int dlog, int CO, int dil, int relu,
                                                            int tiles_per_wave) {
  const int lane = threadIdx.x & 63, wave_g = (int)(((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6);
  const int n = lane & 31, h = lane >> 5;
  // the weights: row n of wp [32][KS * 16] (rows >= CO are zero), k = 16 s + 8 h .. + 7
  uint4 wf[KS];
#pragma unroll
  for (int s = 0; s < KS; ++s) wf[s] = *reinterpret_cast<const uint4 *>(wp + (size_t)n * (KS * 16) + 16 * s + 8 * h);
  float bn[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const int row = (e & 3) + 8 * (e >> 2) + 4 * h;
    bn[e] = bias && row < CO ? bias[row] : 0.0f;
  }
  const mf::BufRsrc xrs = mf::make_rsrc(x);
  const int64_t total = (int64_t)B << (3 * dlog);
  for (int it = 0; it < tiles_per_wave; ++it) {
    const int64_t v = ((int64_t)wave_g * tiles_per_wave + it) * 32 + n;  // this lane's voxel (columns of the MFMA)
    if (v - n >= total) break;  // wave-uniform
    const bool vok = v < total;
    const int iz = (int)(v & (D - 1)), iy = (int)((v >> dlog) & (D - 1)), ix = (int)((v >> (2 * dlog)) & (D - 1));
    const int64_t vb = v - (((int64_t)ix << (2 * dlog)) + ((int64_t)iy << dlog) + iz);  // b * D^3
    mf_f32x16 acc;
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[e] = 0.0f;
    uint4 xf[KS];
#pragma unroll
    for (int s = 0; s < KS; ++s) {
      const int k0 = 16 * s + 8 * h;
      const int tap = k0 / CI, c0 = k0 - tap * CI;  // (CI = 8: tap = 2 s + h; CI = 16: tap = s, c0 = 8 h)
      const int kx = tap / 9, ky = (tap - 9 * kx) / 3, kz = tap - 9 * kx - 3 * ky;
      const int jx = ix + (kx - 1) * dil, jy = iy + (ky - 1) * dil, jz = iz + (kz - 1) * dil;
      const bool ok = vok && tap < 27 && (unsigned)jx < (unsigned)D && (unsigned)jy < (unsigned)D && (unsigned)jz < (unsigned)D;
      const int64_t src = (vb + (((int64_t)jx << (2 * dlog)) + ((int64_t)jy << dlog) + jz)) * CI + c0;
      xf[s] = mf::buf_load16(xrs, ok ? 2u * (uint32_t)src : mf::kBufMasked);
    }
#pragma unroll
    for (int s = 0; s < KS; ++s) acc = mf::mfma_bf16_32x32x16(wf[s], xf[s], acc);
    if (!vok) continue;
    // rows of the accumulator = output channels (e & 3) + 8 (e >> 2) + 4 h; columns = this lane's voxel
#pragma unroll
    for (int g = 0; g < 2; ++g) {
      if (8 * g + 4 * h >= CO) continue;
      float o4[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        float val = acc[4 * g + j] + bn[4 * g + j];
        if (relu) val = val > 0.0f ? val : 0.0f;
        o4[j] = val;
      }
      *reinterpret_cast<uint2 *>(out + v * CO + 8 * g + 4 * h) = make_uint2(mf::pack_bf16x2(o4[0], o4[1]), mf::pack_bf16x2(o4[2], o4[3]));
    }
  }
}

// W [Cout][w_cin][3][3][3] fp32 (framework layout) -> wp bf16 [32 rows][KS * 16]:
//   forward        row n = output channel, k = tap * CI + ci:   W[n][c_off + ci][tap]          (CI = the layer's Cin)
//   data gradient  row n = INPUT channel of the layer, k = tap * CI + co:  W[co][c_off + n][26 - tap]   (CI = Cout)
// rows >= the valid count, k beyond 27 CI and channels at or beyond w_cin are zero.
__global__ __launch_bounds__(256) void k_conv_k3_narrow_pack(const float *__restrict__ W, int Cout, int Cin, int w_cin,
                                                            int c_off, int transpose, int CI, int Kp,
                                                            uint16_t *__restrict__ wp) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= 32 * Kp) return;
  const int nrow = i / Kp, k = i - nrow * Kp;
  const int tap = k / CI, c = k - tap * CI;
  float v = 0.0f;
  if (tap < 27) {
    if (!transpose) {
      if (nrow < Cout && c < Cin && c_off + c < w_cin) v = W[((int64_t)nrow * w_cin + c_off + c) * 27 + tap];
    } else {
      if (nrow < Cin && c < Cout && c_off + nrow < w_cin) v = W[((int64_t)c * w_cin + c_off + nrow) * 27 + (26 - tap)];
    }
  }
  wp[i] = (uint16_t)mf::bf16_bits(v);
}

int ilog2_exact(int x) {
  int l = 0;
  while ((1 << l) < x) ++l;
  return (1 << l) == x ? l : -1;
}

// MF_NT_BIG in the environment: 0 = never the 256 x 256 tile, 1 = by problem size (the default), 2 = wherever its
// structure allows (tests of small problems)
int nt_big_override() {
  const char *e = getenv("MF_NT_BIG");
  return e ? atoi(e) : -1;
}

int g_nt_last_tile = 0;  // rows of the tile the last NT launch used (64 / 128 / 256): mf_gemm_bf16_last_tile

template <int MODE>
int launch_nt(const NtArgs &a, hipStream_t stream) {
  const int64_t full = (int64_t)((a.M + 127) / 128) * ((a.N + kBN - 1) / kBN) * a.groups;
  const int64_t big = (int64_t)((a.M + kBigM - 1) / kBigM) * ((a.N + kBigN - 1) / kBigN) * a.groups;
  bool use_big = big >= 224 && a.N >= 160 && !(MODE == kRows && a.tile_group) &&
                 !(MODE == kConvDgrad && ((a.Do * a.Do * a.Do) & (kBigM - 1)));
  // (N >= 160: dense conv3's data gradient, N = 160 -- one tile of 256 columns, the waves of its last 96 columns idle --
  // measured 551 -> 778 TFLOP/s on the ping-pong form against the 128 x 128 tile's two column tiles)
  if (a.S > 1) use_big = true;  // (a split-K launch: the caller checked the structure, nt_splitk)
  if (nt_big_override() == 2)  // (tests: the big tile wherever its structure allows, whatever the tile count)
    use_big = !(MODE == kRows && a.tile_group) && !(MODE == kConvDgrad && ((a.Do * a.Do * a.Do) & (kBigM - 1)));
  else if (nt_big_override() >= 0)
    use_big = use_big && nt_big_override() == 1;
  if (use_big) {
    // MF_PP_DBG (timing ablations, WRONG results): 1 = no operand requests after tile 0, 2 = every request re-reads
    // K-tile 0 (cache hits), 4 = the two wave groups in lockstep instead of one barrier apart
    static const int dbg = getenv("MF_PP_DBG") ? atoi(getenv("MF_PP_DBG")) : 0;
    NtArgs b = a;
    b.dbg = dbg;
    if (b.S < 1) b.S = 1;
    if (int e = mf::allow_big_lds((const void *)k_gemm_nt_bf16_pp<MODE>, nt_pp_lds())) return e;
    hipLaunchKernelGGL((k_gemm_nt_bf16_pp<MODE>), dim3((unsigned)(big * b.S)), dim3(512), nt_pp_lds(), stream, b);
    if (b.S > 1)
      hipLaunchKernelGGL(k_splitk_finish, dim3((unsigned)(((int64_t)a.M * (a.N / 8) + 255) / 256)), dim3(256), 0,
                         stream, (const float *)b.slab, a.bias, a.out, (int64_t)a.M, a.N, b.S, a.ldo, a.relu,
                         a.out_f32);
    g_nt_last_tile = kBigM;
    return 0;
  }
  // 64-row tiles when the 128-row ones would leave CUs with fewer than two workgroups (MF_NT_HALF_MAX: the largest
  // count of 128-row tiles that still takes the half-height tile; tuning knob)
  static const int half_max = getenv("MF_NT_HALF_MAX") ? atoi(getenv("MF_NT_HALF_MAX")) : 255;
  const bool half = full <= half_max && MODE != kConvDgrad;  // (dgrad tiles must stay class-homogeneous: 128 | Do^3)
  g_nt_last_tile = half ? 64 : 128;
  if (half) {
    if (int e = mf::allow_big_lds((const void *)k_gemm_nt_bf16<MODE, 1>, nt_lds<1>())) return e;
    const int64_t grid = (int64_t)((a.M + 63) / 64) * ((a.N + kBN - 1) / kBN) * a.groups;
    hipLaunchKernelGGL((k_gemm_nt_bf16<MODE, 1>), dim3((unsigned)grid), dim3(256), nt_lds<1>(), stream, a);
  } else {
    if (int e = mf::allow_big_lds((const void *)k_gemm_nt_bf16<MODE, 2>, nt_lds<2>())) return e;
    hipLaunchKernelGGL((k_gemm_nt_bf16<MODE, 2>), dim3((unsigned)full), dim3(256), nt_lds<2>(), stream, a);
  }
  return 0;
}

constexpr int64_t kMaxBf16Elems = 1ll << 30;  // 2^31 bytes: the span of a buffer resource (mf_common.h kBufSpan)
int bad(const char *msg) {
  mf::set_last_error(hipErrorInvalidValue, msg);
  return -(int)hipErrorInvalidValue;
}

}  // namespace

namespace {
int wgrad_split_model(int64_t tiles, int64_t ktiles, int64_t slab_bytes, int slots) {
  if (tiles <= 0 || ktiles <= 0) return 1;
  const double per_slab = slab_bytes / 3.0e6 > 0.05 ? slab_bytes / 3.0e6 : 0.05;  // us at ~3 TB/s, launch floor
  int best = 1;
  double best_cost = 1e30;
  for (int S = 1; S <= 512; ++S) {
    if (S > 1 && ktiles / S < 8) break;
    const int64_t rounds = (tiles * S + slots - 1) / slots;
    const double cost = (double)rounds * (double)((ktiles + S - 1) / S + 8) + (S > 1 ? S * per_slab : 0.0);
    if (cost < best_cost) { best_cost = cost; best = S; }
  }
  return best;
}
// The 256 x 256 ping-pong form of the TN engine (k_gemm_tn_bf16_pp) takes the weight gradients whose result has at
// least 192 rows and columns AND whose reduction is long enough that a split filling the 256 CUs still leaves every
// workgroup >= 48 K-tiles (its prologue / epilogue -- 256 KB of fp32 tile through LDS -- weigh on shorter ones: the
// heads' first layer, 32 tiles x 250 K-tiles, measured 570 TFLOP/s on it against 670 on the 128 x 128 form).
// MF_TN_PP=0: never; MF_NT_BIG=2, the tests' switch: wherever the shape allows.
bool tn_use_pp(int Ni, int Nj, int64_t ktiles, int groups, bool ranges) {
  if (getenv("MF_TN_PP") && atoi(getenv("MF_TN_PP")) == 0) return false;
  if (nt_big_override() == 0 || ranges) return false;
  if (nt_big_override() == 2) return true;
  if (Ni < 192 || Nj < 192) return false;
  const int64_t tiles = (int64_t)((Ni + 255) / 256) * ((Nj + 255) / 256) * groups;
  const int64_t fill = tiles >= 256 ? 1 : (256 + tiles - 1) / tiles;  // splits that fill the chip
  return ktiles >= 48 * fill;
}
// slabs of a weight gradient [Ni][Nj] reduced over ``ktiles`` row tiles of 64, ``groups`` results side by side: the
// cost model below on the tile / workgroup-slot counts of the form that will run it (the pp form: one 256 x 256 tile
// per CU and K-tiles of twice the work)
int wgrad_split_for(int Ni, int Nj, int64_t ktiles, int groups) {
  const int64_t slab = (int64_t)Ni * Nj * 4 * groups;
  if (tn_use_pp(Ni, Nj, ktiles, groups, false))
    return wgrad_split_model((int64_t)((Ni + 255) / 256) * ((Nj + 255) / 256) * groups, 2 * ktiles, slab, 256);
  return wgrad_split_model((int64_t)((Ni + 127) / 128) * ((Nj + 127) / 128) * groups, ktiles, slab, 512);
}
template <bool CONV>
int launch_tn(const TnArgs &a, bool ranges, hipStream_t stream) {
  if (tn_use_pp(a.Ni, a.Nj, ((int64_t)a.M + 63) / 64, a.groups, ranges)) {
    if (int e = mf::allow_big_lds((const void *)k_gemm_tn_bf16_pp<CONV>, 5 * kPpOp)) return e;
    const int64_t grid = (int64_t)((a.Ni + 255) / 256) * ((a.Nj + 255) / 256) * a.groups * a.S;
    hipLaunchKernelGGL(k_gemm_tn_bf16_pp<CONV>, dim3((unsigned)grid), dim3(512), 5 * kPpOp, stream, a);
    return 0;
  }
  if (int e = mf::allow_big_lds((const void *)k_gemm_tn_bf16<CONV>, kTnLds)) return e;
  const int64_t grid = (int64_t)((a.Ni + 127) / 128) * ((a.Nj + 127) / 128) * a.groups * a.S;
  hipLaunchKernelGGL(k_gemm_tn_bf16<CONV>, dim3((unsigned)grid), dim3(256), kTnLds, stream, a);
  return 0;
}
}  // namespace

extern "C" int mf_cast_rows_bf16(const float *src, int64_t src_ld, void *dst, int64_t dst_ld, int64_t rows,
                                 int32_t cols, mfStream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (rows <= 0) return 0;
  if (dst_ld % 8 || cols > dst_ld || src_ld < cols || ((uintptr_t)dst & 15)) return bad("cast_rows_bf16: dst_ld % 8 == 0, cols <= dst_ld");
  const int64_t n = rows * (dst_ld / 8);
  hipLaunchKernelGGL(k_cast_rows_bf16, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, src, src_ld,
                     (uint16_t *)dst, dst_ld, rows, cols);
  return mf::check_launch("mf_cast_rows_bf16");
}

extern "C" int mf_relu_mask_bf16(const void *y, const void *dy, const float *dy32, void *dz, int64_t n,
                                 mfStream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (n <= 0) return 0;
  if (n % 8 || (!dy == !dy32)) return bad("relu_mask_bf16: n % 8 == 0 and exactly one of dy / dy32");
  hipLaunchKernelGGL(k_relu_mask_bf16, dim3((unsigned)((n / 8 + 255) / 256)), dim3(256), 0, stream,
                     (const uint16_t *)y, (const uint16_t *)dy, dy32, (uint16_t *)dz, n / 8);
  return mf::check_launch("mf_relu_mask_bf16");
}

/* out = act(A W^T + bias): A bf16 [M][lda], W bf16 [N][ldw] (k-contiguous rows), bias fp32; out bf16 or fp32
 * [M][ldo] (accumulate: out += , fp32 only); ``groups`` independent problems at the given element strides. */
extern "C" int mf_linear_bf16(const void *A, int64_t a_gs, int32_t lda, const void *W, int64_t w_gs, int32_t ldw,
                              const float *bias, int64_t b_gs, void *out, int64_t o_gs, int32_t ldo, int32_t M,
                              int32_t N, int32_t K, int32_t groups, int32_t relu, int32_t out_f32,
                              int32_t accumulate, mfStream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (M <= 0 || N <= 0 || groups <= 0) return 0;
  if (K <= 0 || K % 8 || lda % 8 || ldw % 8 || a_gs % 8 || w_gs % 8 || lda < K || ldw < K || ldo < N ||
      (((uintptr_t)A | (uintptr_t)W) & 15) || (accumulate && !out_f32))
    return bad("linear_bf16: K, lda, ldw, group strides % 8 == 0, 16-byte aligned A / W, accumulate needs fp32 out");
  if ((int64_t)M * lda >= kMaxBf16Elems || (int64_t)N * ldw >= kMaxBf16Elems)
    return bad("linear_bf16: an operand of one group spans >= 2^31 bytes (32-bit byte offsets: split the rows)");
  NtArgs a = {};
  a.A = (const uint16_t *)A; a.W = (const uint16_t *)W; a.bias = bias; a.out = out;
  a.a_gs = a_gs; a.w_gs = w_gs; a.b_gs = b_gs; a.o_gs = o_gs;
  a.M = M; a.N = N; a.K = K; a.lda = lda; a.ldw = ldw; a.ldo = ldo; a.groups = groups;
  a.relu = relu; a.out_f32 = out_f32; a.accumulate = accumulate;
  if (int e = launch_nt<kRows>(a, stream)) return e;
  return mf::check_launch("mf_linear_bf16");
}

/* Weight gradient of out = A W^T: dW[n][k] (fp32, row pitch ldc) = sum_m dY[m][n] A[m][k]; ``split`` > 1: partial
 * sums over row ranges in ws (split * groups * N * ldc floats), added in order. */
extern "C" int mf_linear_wgrad_bf16(const void *dY, int64_t y_gs, int32_t ldy, const void *A, int64_t a_gs, int32_t lda,
                                    float *dW, int64_t w_gs, int32_t ldc, void *ws, int32_t M, int32_t N, int32_t K,
                                    int32_t groups, int32_t split, mfStream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (M <= 0 || N <= 0 || K <= 0 || groups <= 0) return 0;
  if (N % 8 || K % 8 || ldy % 8 || lda % 8 || y_gs % 8 || a_gs % 8 || split < 1 || (split > 1 && !ws) || ldc < K ||
      (((uintptr_t)dY | (uintptr_t)A) & 15))
    return bad("linear_wgrad_bf16: N, K, ldy, lda, group strides % 8 == 0, 16-byte aligned operands");
  if ((int64_t)M * ldy >= kMaxBf16Elems || (int64_t)M * lda >= kMaxBf16Elems)
    return bad("linear_wgrad_bf16: an operand of one group spans >= 2^31 bytes (32-bit byte offsets: split the rows)");
  TnArgs a = {};
  a.P = (const uint16_t *)dY; a.Q = (const uint16_t *)A; a.out = split > 1 ? (float *)ws : dW;
  a.p_gs = y_gs; a.q_gs = a_gs; a.c_gs = w_gs;
  a.M = M; a.Ni = N; a.Nj = K; a.ldp = ldy; a.ldq = lda; a.ldc = ldc; a.groups = groups; a.S = split;
  if (int e = launch_tn<false>(a, false, stream)) return e;
  if (split > 1) {
    const int64_t per_group = (int64_t)N * ldc, per_slab = per_group * groups;
    if (split >= 32 && per_slab <= (1 << 16))
      hipLaunchKernelGGL(k_wgrad_finish_deep, dim3((unsigned)((per_slab + 3) / 4)), dim3(256), 0, stream,
                         (const float *)ws, dW, per_slab, K, ldc, split, 0, w_gs, per_group, (int64_t)ldc, 0, 0);
    else
      hipLaunchKernelGGL(k_wgrad_finish, dim3((unsigned)((per_slab + 255) / 256)), dim3(256), 0, stream,
                         (const float *)ws, dW, per_slab, K, ldc, split, 0, w_gs, per_group, (int64_t)ldc, 0, 0);
  }
  return mf::check_launch("mf_linear_wgrad_bf16");
}

/* mf_linear_bf16 for rows that come in GROUPS with their own weights, the group of every 64-row block read from a
 * device table (``tile_group`` [ceil(M / 64)], -1 = empty block; groups are padded to multiples of 128 rows by the
 * producer): out[m][:] = A[m][:] W[group(m)]^T.  W: [n_groups][N][ldw] at group stride ``w_gs``.  No bias / ReLU. */
extern "C" int mf_linear_bf16_tiles(const void *A, int32_t lda, const void *W, int64_t w_gs, int32_t ldw,
                                    const int32_t *tile_group, void *out, int32_t ldo, int32_t M, int32_t N, int32_t K,
                                    int32_t out_f32, mfStream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (M <= 0 || N <= 0) return 0;
  if (K <= 0 || K % 8 || lda % 8 || ldw % 8 || w_gs % 8 || lda < K || ldw < K || ldo < N || M % 128 || !tile_group ||
      (((uintptr_t)A | (uintptr_t)W) & 15))
    return bad("linear_bf16_tiles: K, lda, ldw, w_gs % 8 == 0, M % 128 == 0, 16-byte aligned A / W, a group table");
  if ((int64_t)M * lda >= kMaxBf16Elems || (int64_t)N * ldw >= kMaxBf16Elems)
    return bad("linear_bf16_tiles: an operand spans >= 2^31 bytes");
  NtArgs a = {};
  a.A = (const uint16_t *)A; a.W = (const uint16_t *)W; a.out = out;
  a.w_gs = w_gs;
  a.M = M; a.N = N; a.K = K; a.lda = lda; a.ldw = ldw; a.ldo = ldo; a.groups = 1;
  a.out_f32 = out_f32;
  a.tile_group = tile_group;
  if (int e = launch_nt<kRows>(a, stream)) return e;
  return mf::check_launch("mf_linear_bf16_tiles");
}

/* mf_linear_wgrad_bf16 over row RANGES read from the device: dW[g][n][k] = sum over rows m in
 * [m_range[g], m_range[g + 1]) of dY[m][n] A[m][k] for g < groups (ranges are multiples of 64 rows; rows of a range
 * that hold no data must be zero in one operand).  dW fp32 [groups][N][ldc] at group stride ``w_gs``. */
extern "C" int mf_linear_wgrad_bf16_ranges(const void *dY, int32_t ldy, const void *A, int32_t lda, float *dW,
                                           int64_t w_gs, int32_t ldc, const int32_t *m_range, int32_t groups, int32_t N,
                                           int32_t K, mfStream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (N <= 0 || K <= 0 || groups <= 0) return 0;
  if (N % 8 || K % 8 || ldy % 8 || lda % 8 || ldc < K || !m_range || (((uintptr_t)dY | (uintptr_t)A) & 15))
    return bad("linear_wgrad_bf16_ranges: N, K, ldy, lda % 8 == 0, 16-byte aligned operands, a range table");
  TnArgs a = {};
  a.P = (const uint16_t *)dY; a.Q = (const uint16_t *)A; a.out = dW;
  a.c_gs = w_gs;
  a.M = 0; a.Ni = N; a.Nj = K; a.ldp = ldy; a.ldq = lda; a.ldc = ldc; a.groups = groups; a.S = 1;
  a.m_range = m_range;
  if (int e = launch_tn<false>(a, true, stream)) return e;
  return mf::check_launch("mf_linear_wgrad_bf16_ranges");
}

/* Convolution3D on channels-last bf16 grids: kernel ks in {3, 4}, stride in {1, 2}, any pad / dilation whose output
 * size Do = (D + 2 pad - dil (ks - 1) - 1) / stride + 1 is a power of two.  W: fp32 in the framework layout
 * [Cout][w_cin][ks][ks][ks]. */
namespace {
struct Geom { int Do, olog, taps; };
int conv_geom(int32_t B, int32_t Cin, int32_t Cout, int32_t D, int32_t ks, int32_t stride, int32_t pad, int32_t dil,
              Geom *g) {
  const int span = dil * (ks - 1) + 1;
  if ((ks != 3 && ks != 4) || (stride != 1 && stride != 2) || dil < 1 || pad < 0 || D + 2 * pad < span)
    return bad("conv3d (bf16): kernel 3 or 4, stride 1 or 2");
  g->Do = (D + 2 * pad - span) / stride + 1;
  g->olog = ilog2_exact(g->Do);
  g->taps = ks * ks * ks;
  // operands are addressed with 32-bit BYTE offsets from a buffer resource of 2^31 bytes (mf_common.h: an offset
  // >= 2^31 is the masked value and reads zeros): every bf16 tensor must stay below 2^30 ELEMENTS
  if (g->olog < 1 || Cin % 8 || Cout % 8 || (int64_t)B * D * D * D * Cin >= kMaxBf16Elems ||
      (int64_t)B * g->Do * g->Do * g->Do * Cout >= kMaxBf16Elems || (int64_t)Cout * g->taps * Cin >= kMaxBf16Elems)
    return bad("conv3d (bf16): output size a power of two, Cin % 8 == 0, Cout % 8 == 0, tensors < 2^30 elements (2^31 bytes)");
  return 0;
}
}  // namespace

namespace {
// Split of K for a forward / data-gradient GEMM with too few 256 x 256 tiles to fill the chip (conv4's forward at 16
// objects: 32 x 2 tiles for 256 CUs): S workgroups per tile, each over a contiguous range of >= 16 K-tiles, fp32
// partial sums in S slabs, added in order by k_splitk_finish (deterministic).  1 = no split.
int nt_splitk(int64_t M, int N, int K) {
  const int forced = getenv("MF_NT_SPLITK") ? atoi(getenv("MF_NT_SPLITK")) : 0;  // (tests; 0 = by problem size)
  if (N < 192 || N % 8 || nt_big_override() == 0) return 1;
  const int64_t big = ((M + kBigM - 1) / kBigM) * ((N + kBigN - 1) / kBigN);
  const int T = (K + kBK - 1) / kBK;
  if (forced > 0) return forced <= T ? forced : 1;
  if (big >= 160 || big < 16) return 1;
  int S = (int)(256 / big);
  while (S > 1 && T / S < 16) --S;
  return S;
}
}  // namespace

extern "C" int64_t mf_conv3d_bf16_fwd_workspace_bytes(int32_t B, int32_t Cin, int32_t Cout, int32_t D, int32_t ks,
                                                      int32_t stride, int32_t pad, int32_t dil) {
  Geom g;
  if (B <= 0 || conv_geom(B, Cin, Cout, D, ks, stride, pad, dil, &g)) return 0;
  const int64_t M = (int64_t)B * g.Do * g.Do * g.Do;
  const int S = nt_splitk(M, Cout, g.taps * Cin);
  return S > 1 ? (int64_t)S * M * Cout * 4 : 0;
}

extern "C" int mf_conv3d_bf16_pack(const float *W, int32_t Cout, int32_t Cin, int32_t w_cin, int32_t c_off, int32_t ks,
                                   void *fwd, void *dgrad_k4s2, void *flipT, mfStream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if ((ks != 3 && ks != 4) || (dgrad_k4s2 && ks != 4)) return bad("conv3d_bf16_pack: kernel 3 or 4 (parity-class dgrad: 4)");
  const int taps = ks * ks * ks;
  if (fwd)
    hipLaunchKernelGGL(k_conv_pack_fwd_tile, dim3((Cin + kPackTile - 1) / kPackTile, Cout), dim3(256), 0, stream, W, Cin,
                       w_cin, c_off, taps, (uint16_t *)fwd);
  if (dgrad_k4s2 || flipT)
    hipLaunchKernelGGL(k_conv_pack_cof_tile, dim3((Cout + kPackTile - 1) / kPackTile, Cin), dim3(256), 0, stream, W,
                       Cout, Cin, w_cin, c_off, ks, (uint16_t *)dgrad_k4s2, (uint16_t *)flipT);
  return mf::check_launch("mf_conv3d_bf16_pack");
}

/* out [B][Do^3][ldo >= Cout] = act(conv(x [B][D^3][Cin]) + bias): x, wt ([Cout][ks^3][Cin]) bf16; out bf16 / fp32 */
extern "C" int mf_conv3d_bf16_fwd(const void *x, const void *wt, const float *bias, void *out, int32_t B, int32_t Cin,
                                  int32_t Cout, int32_t D, int32_t ks, int32_t stride, int32_t pad, int32_t dil,
                                  int32_t relu, int32_t out_f32, int32_t ldo, mfStream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (B <= 0) return 0;
  Geom g;
  if (int e = conv_geom(B, Cin, Cout, D, ks, stride, pad, dil, &g)) return e;
  if (ldo < Cout) return bad("conv3d_bf16_fwd: ldo >= Cout");
  NtArgs a = {};
  a.A = (const uint16_t *)x; a.W = (const uint16_t *)wt; a.bias = bias; a.out = out;
  a.M = B * g.Do * g.Do * g.Do; a.N = Cout; a.K = g.taps * Cin; a.ldw = g.taps * Cin; a.ldo = ldo; a.groups = 1;
  a.relu = relu; a.out_f32 = out_f32;
  a.B = B; a.D = D; a.Do = g.Do; a.olog = g.olog; a.Cin = Cin; a.Cout = Cout;
  a.ks = ks; a.stride = stride; a.pad = pad; a.dil = dil;
  if (int e = launch_nt<kConvFwd>(a, stream)) return e;
  return mf::check_launch("mf_conv3d_bf16_fwd");
}

/* mf_conv3d_bf16_fwd with a workspace of mf_conv3d_bf16_fwd_workspace_bytes(...) bytes (0: none needed, ws may be
 * null): a layer with too few output tiles for the chip splits its reduction over the workspace's fp32 slabs. */
extern "C" int mf_conv3d_bf16_fwd_ws(const void *x, const void *wt, const float *bias, void *out, void *ws,
                                     int64_t ws_bytes, int32_t B, int32_t Cin, int32_t Cout, int32_t D, int32_t ks,
                                     int32_t stride, int32_t pad, int32_t dil, int32_t relu, int32_t out_f32,
                                     int32_t ldo, mfStream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (B <= 0) return 0;
  Geom g;
  if (int e = conv_geom(B, Cin, Cout, D, ks, stride, pad, dil, &g)) return e;
  if (ldo < Cout) return bad("conv3d_bf16_fwd_ws: ldo >= Cout");
  const int64_t M = (int64_t)B * g.Do * g.Do * g.Do;
  const int S = nt_splitk(M, Cout, g.taps * Cin);
  if (S <= 1 || !ws) return mf_conv3d_bf16_fwd(x, wt, bias, out, B, Cin, Cout, D, ks, stride, pad, dil, relu, out_f32, ldo, stream_);
  if (ws_bytes < (int64_t)S * M * Cout * 4 || ((uintptr_t)ws & 15)) return bad("conv3d_bf16_fwd_ws: workspace too small / unaligned");
  NtArgs a = {};
  a.A = (const uint16_t *)x; a.W = (const uint16_t *)wt; a.bias = bias; a.out = out;
  a.M = (int)M; a.N = Cout; a.K = g.taps * Cin; a.ldw = g.taps * Cin; a.ldo = ldo; a.groups = 1;
  a.relu = relu; a.out_f32 = out_f32;
  a.B = B; a.D = D; a.Do = g.Do; a.olog = g.olog; a.Cin = Cin; a.Cout = Cout;
  a.ks = ks; a.stride = stride; a.pad = pad; a.dil = dil;
  a.S = S; a.slab = (float *)ws;
  if (int e = launch_nt<kConvFwd>(a, stream)) return e;
  return mf::check_launch("mf_conv3d_bf16_fwd_ws");
}

/* dx [B][D^3][Cin] (+)= conv^T(dy [B][(D/2)^3][Cout]) of the k4 / s2 / p1 layers: dy, wd (packed parity-class layout)
 * bf16; dx bf16, or fp32 with ``accumulate`` (dx already holds another consumer's gradient).  (Stride-1 layers take
 * their data gradient through mf_conv3d_bf16_fwd on the flipT operand.) */
extern "C" int mf_conv3d_k4s2_bf16_dgrad(const void *dy, const void *wd, void *dx, int32_t B, int32_t Cin,
                                         int32_t Cout, int32_t D, int32_t out_f32, int32_t accumulate,
                                         mfStream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (B <= 0) return 0;
  Geom g;
  if (int e = conv_geom(B, Cin, Cout, D, 4, 2, 1, 1, &g)) return e;
  if ((g.Do * g.Do * g.Do) % 128 || (accumulate && !out_f32)) return bad("conv3d_k4s2 dgrad: (D/2)^3 % 128 == 0; accumulate needs fp32");
  NtArgs a = {};
  a.A = (const uint16_t *)dy; a.W = (const uint16_t *)wd; a.out = dx;
  a.M = B * D * D * D; a.N = Cin; a.K = 8 * Cout; a.ldw = 8 * Cout; a.ldo = Cin; a.groups = 1;
  a.out_f32 = out_f32; a.accumulate = accumulate;
  a.B = B; a.D = D; a.Do = g.Do; a.olog = g.olog; a.Cin = Cin; a.Cout = Cout; a.ks = 4; a.stride = 2; a.pad = 1; a.dil = 1;
  if (int e = launch_nt<kConvDgrad>(a, stream)) return e;
  return mf::check_launch("mf_conv3d_k4s2_bf16_dgrad");
}

extern "C" int64_t mf_conv3d_bf16_wgrad_workspace_bytes(int32_t Cin, int32_t Cout, int32_t ks, int32_t split) {
  return (int64_t)(split > 1 ? split : 1) * Cout * ks * ks * ks * Cin * 4;
}

/* Split of the reduction (rows) of a weight-gradient GEMM over S workgroups per output tile, S fp32 slabs summed by
 * the finish pass.  ``tiles`` output tiles of 128 x 128, ``ktiles`` row tiles of 64, ``slab_bytes`` = size of one
 * slab.  Cost model in units of one K-tile of one workgroup (~1 us at two workgroups per CU, 512 slots on the chip):
 *   ceil(tiles S / 512) rounds x (ceil(ktiles / S) + 8 K-tiles of prologue / epilogue)  +  S slabs read by the finish
 * -- the first version doubled S until tiles * S >= 512, which put conv3 (160 tiles) at S = 4: 640 workgroups, a
 * second round a quarter full; S = 3 fills one round. */
/* Split of the reduction (rows) of a weight-gradient GEMM over S workgroups per output tile, S fp32 slabs summed by
 * the finish pass.  ``tiles`` output tiles of 128 x 128, ``ktiles`` row tiles of 64, ``slab_bytes`` = size of one
 * slab.  Cost model in units of one K-tile of one workgroup (~1 us at two workgroups per CU, 512 slots on the chip):
 *   ceil(tiles S / 512) rounds x (ceil(ktiles / S) + 8 K-tiles of prologue / epilogue)  +  S slabs read by the finish
 * -- the first version doubled S until tiles * S >= 512, which put conv3 (160 tiles) at S = 4: 640 workgroups, a
 * second round a quarter full; S = 3 fills one round.  (The 128 x 128 form's model; mf_linear_wgrad_bf16_default_split
 * / mf_conv3d_bf16_wgrad_default_split answer for the form that will actually run.) */
extern "C" int32_t mf_wgrad_split(int64_t tiles, int64_t ktiles, int64_t slab_bytes) {
  return wgrad_split_model(tiles, ktiles, slab_bytes, 512);
}

extern "C" int32_t mf_linear_wgrad_bf16_default_split(int64_t M, int32_t N, int32_t K, int32_t groups) {
  return wgrad_split_for(N, K, (M + 63) / 64, groups);
}

extern "C" int32_t mf_conv3d_bf16_wgrad_default_split(int32_t B, int32_t Cin, int32_t Cout, int32_t Do, int32_t ks) {
  return wgrad_split_for(Cout, ks * ks * ks * Cin, ((int64_t)B * Do * Do * Do + 63) / 64, 1);
}

/* dW [Cout][w_cin][ks][ks][ks] (input channels c_off .., those below w_cin: fp32, the framework layout) = sum over
 * output voxels of dy (x) im2col(x); ws: mf_conv3d_bf16_wgrad_workspace_bytes. */
extern "C" int mf_conv3d_bf16_wgrad(const void *dy, const void *x, float *dW, void *ws, int32_t B, int32_t Cin,
                                    int32_t Cout, int32_t D, int32_t ks, int32_t stride, int32_t pad, int32_t dil,
                                    int32_t w_cin, int32_t c_off, int32_t split, mfStream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (B <= 0) return 0;
  Geom g;
  if (int e = conv_geom(B, Cin, Cout, D, ks, stride, pad, dil, &g)) return e;
  if (split < 1 || !ws) return bad("conv3d wgrad: workspace required");
  if (g.olog < 2) return bad("conv3d wgrad: output size >= 4 per axis");
  TnArgs a = {};
  a.P = (const uint16_t *)dy; a.Q = (const uint16_t *)x; a.out = (float *)ws;
  a.M = B * g.Do * g.Do * g.Do; a.Ni = Cout; a.Nj = g.taps * Cin; a.ldp = Cout; a.ldc = g.taps * Cin; a.groups = 1;
  a.S = split;
  a.conv = 1; a.B = B; a.D = D; a.Do = g.Do; a.olog = g.olog; a.Cin = Cin;
  a.ks = ks; a.stride = stride; a.pad = pad; a.dil = dil;
  // (S == 1 also goes through the workspace: the finish pass permutes (tap, cin) -> (cin, tap))
  if (int e = launch_tn<true>(a, false, stream)) return e;
  const int64_t per_slab = (int64_t)Cout * g.taps * Cin;
  const int keep = w_cin - c_off < Cin ? w_cin - c_off : Cin;
  // big layers: the tiled transpose; small ones (the occupancy convolutions: a few thousand weights in up to 256
  // slabs) stay element-wise -- one thread per weight walks the slabs, where a tile's 256 threads would walk them
  // seven elements each (measured: 0.42 ms for conv2_occ's 3456 weights)
  if (keep > 0 && per_slab >= (1 << 20))
    hipLaunchKernelGGL(k_wgrad_finish_conv, dim3((keep + kPackTile - 1) / kPackTile, Cout), dim3(256), 0, stream,
                       (const float *)ws, dW + (int64_t)c_off * g.taps, per_slab, a.S, Cin, g.taps, w_cin, keep);
  else if (keep > 0 && a.S >= 32 && per_slab <= (1 << 16))
    hipLaunchKernelGGL(k_wgrad_finish_deep, dim3((unsigned)((per_slab + 3) / 4)), dim3(256), 0, stream,
                       (const float *)ws, dW + (int64_t)c_off * g.taps, per_slab, g.taps * Cin, g.taps * Cin, a.S, Cin,
                       (int64_t)0, per_slab, (int64_t)w_cin * g.taps, g.taps, keep);
  else if (keep > 0)
    hipLaunchKernelGGL(k_wgrad_finish, dim3((unsigned)((per_slab + 255) / 256)), dim3(256), 0, stream,
                       (const float *)ws, dW + (int64_t)c_off * g.taps, per_slab, g.taps * Cin, g.taps * Cin, a.S, Cin,
                       (int64_t)0, per_slab, (int64_t)w_cin * g.taps, g.taps, keep);
  return mf::check_launch("mf_conv3d_bf16_wgrad");
}

/* 3 x 3 x 3 convolutions (stride 1, pad = dil) between narrow layers -- Cin in {8, 16}, Cout <= 16 -- on channels-last
 * bf16 grids whose size D is a power of two: the occupancy branch's conv1_occ / conv2_occ and conv2_occ's data
 * gradient (``transpose`` at pack time).  wp: 32 x ceil(27 CI / 16) x 16 bf16 from mf_conv3d_k3_narrow_bf16_pack
 * (CI = the channels of the tensor the convolution READS: Cin forward, Cout for the data gradient). */
extern "C" int64_t mf_conv3d_k3_narrow_bf16_pack_elems(int32_t CI) { return (int64_t)32 * ((27 * CI + 15) / 16) * 16; }

extern "C" int mf_conv3d_k3_narrow_bf16_pack(const float *W, int32_t Cout, int32_t Cin, int32_t w_cin, int32_t c_off,
                                             int32_t transpose, void *wp, mfStream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  const int CI = transpose ? Cout : Cin, NO = transpose ? Cin : Cout;
  if ((CI != 8 && CI != 16) || NO < 1 || NO > 16 || (NO & 3)) return bad("conv3d_k3_narrow pack: read channels 8 or 16, written channels 4 .. 16 (% 4)");
  const int Kp = ((27 * CI + 15) / 16) * 16;
  hipLaunchKernelGGL(k_conv_k3_narrow_pack, dim3((32 * Kp + 255) / 256), dim3(256), 0, stream, W, Cout, Cin, w_cin, c_off,
                     transpose, CI, Kp, (uint16_t *)wp);
  return mf::check_launch("mf_conv3d_k3_narrow_bf16_pack");
}

extern "C" int mf_conv3d_k3_narrow_bf16(const void *x, const void *wp, const float *bias, void *out, int32_t B,
                                        int32_t CI, int32_t CO, int32_t D, int32_t dil, int32_t relu,
                                        mfStream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (B <= 0) return 0;
  const int dlog = ilog2_exact(D);
  if ((CI != 8 && CI != 16) || CO < 1 || CO > 16 || (CO & 3) || dlog < 1 || dil < 1 ||
      (int64_t)B * D * D * D * 16 >= kMaxBf16Elems || (((uintptr_t)x | (uintptr_t)wp | (uintptr_t)out) & 15))
    return bad("conv3d_k3_narrow: read channels 8 or 16, written channels 4 .. 16 (% 4), D a power of two, 16-byte aligned");
  const int64_t tiles = ((int64_t)B * D * D * D + 31) / 32;
  // a wave walks a few tiles with its weights in registers; enough workgroups (4 waves) for every CU
  int tpw = 1;
  while (tpw < 8 && tiles / (4 * tpw * 2) >= 2048) tpw *= 2;
  const unsigned blocks = (unsigned)((tiles + 4 * tpw - 1) / (4 * tpw));
  if (CI == 8)
    hipLaunchKernelGGL((k_conv_k3_narrow_bf16<8, 14>), dim3(blocks), dim3(256), 0, stream, (const uint16_t *)x,
                       (const uint16_t *)wp, bias, (uint16_t *)out, B, D, dlog, CO, dil, relu, tpw);
  else
    hipLaunchKernelGGL((k_conv_k3_narrow_bf16<16, 27>), dim3(blocks), dim3(256), 0, stream, (const uint16_t *)x,
                       (const uint16_t *)wp, bias, (uint16_t *)out, B, D, dlog, CO, dil, relu, tpw);
  return mf::check_launch("mf_conv3d_k3_narrow_bf16");
}

/* the k4 / s2 / p1 forms (conv3, conv4) under their round-4 names */
extern "C" int mf_conv3d_k4s2_pack_bf16(const float *W, int32_t Cout, int32_t Cin, int32_t w_cin, int32_t c_off,
                                        void *fwd, void *dgrad, mfStream_t stream) {
  return mf_conv3d_bf16_pack(W, Cout, Cin, w_cin, c_off, 4, fwd, dgrad, nullptr, stream);
}
extern "C" int mf_conv3d_k4s2_bf16_fwd(const void *x, const void *wt, const float *bias, void *out, int32_t B,
                                       int32_t Cin, int32_t Cout, int32_t D, int32_t relu, int32_t out_f32,
                                       mfStream_t stream) {
  return mf_conv3d_bf16_fwd(x, wt, bias, out, B, Cin, Cout, D, 4, 2, 1, 1, relu, out_f32, Cout, stream);
}
extern "C" int64_t mf_conv3d_k4s2_bf16_wgrad_workspace_bytes(int32_t Cin, int32_t Cout, int32_t split) {
  return mf_conv3d_bf16_wgrad_workspace_bytes(Cin, Cout, 4, split);
}
extern "C" int32_t mf_conv3d_k4s2_bf16_wgrad_default_split(int32_t B, int32_t Cin, int32_t Cout, int32_t D) {
  return mf_conv3d_bf16_wgrad_default_split(B, Cin, Cout, D / 2, 4);
}
extern "C" int mf_conv3d_k4s2_bf16_wgrad(const void *dy, const void *x, float *dW, void *ws, int32_t B, int32_t Cin,
                                         int32_t Cout, int32_t D, int32_t w_cin, int32_t c_off, int32_t split,
                                         mfStream_t stream) {
  return mf_conv3d_bf16_wgrad(dy, x, dW, ws, B, Cin, Cout, D, 4, 2, 1, 1, w_cin, c_off, split, stream);
}

/* Tile height (64 / 128 / 256 rows) of the NT engine's most recent launch in this process: lets tests and the
 * timing tools see which form of the engine a problem was given to (launch_nt's choice, MF_NT_BIG). */
extern "C" int mf_gemm_bf16_last_tile(void) { return g_nt_last_tile; }
