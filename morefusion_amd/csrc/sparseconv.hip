// Sparse-input 3-D convolution (kernel 4, stride 2, pad 1) on fp32 MFMA for gfx950.
//
// Reference: conv3 of the pose network, `L.Convolution3D(None, 256, 4, 2, pad=1)` applied to
// the voxelized point features (contrib/singleview_3d/models/model.py:73,128): a dense cuDNN
// convolution over [B,144,32^3] = 18.9 GFLOP per object -- although average_voxelization_3d
// leaves at most P = 1000 of the 32768 voxels non-zero (<= 3 %).
//
// MI355X design: only occupied voxels do work.
//   * k4/s2/p1 geometry: input voxel v feeds output o = ((v+1)>>1) - a with kernel tap
//     k = p + 2a per axis, p = (v+1)&1, a in {0,1}.  So voxels fall into 8 parity classes,
//     and all voxels of a class use the same 8 kernel taps.
//   * per class ONE GEMM  C[n_class x (8*Cout)] = A[n_class x Cs] . Wp[Cs x (8*Cout)]
//     (A = gathered voxel features, Wp = the 8 taps' weight slices side by side) on
//     v_mfma_f32_16x16x4_f32: exact fp32, k-ordered fma chain.  0.6 GFLOP per object.
//   * an output-stationary reduce then sums, for every output voxel, its <= 64 contributing
//     (voxel, tap) rows in fixed tap order (deterministic, no atomics), adds the dense part
//     (the 16 occupancy channels, computed by a stock dense convolution) and the bias,
//     applies ReLU, and stores coalesced through an LDS transpose.
#include <algorithm>

#include "mf_common.h"
#include "voxel_chain.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int kTM = 64, kTN = 64;  // GEMM tile per 256-lane workgroup

struct ScArgs {
  const float *x;         // [B,Cs,D,D,D] dense, mostly zero
  const int32_t *counts;  // [B,D,D,D] occupancy counts of average_voxelization_3d
  int B, Cs, Cout, D, max_rows;
  int32_t *rowmap;     // [B*V] compact row of an occupied voxel, else -1
  int32_t *rowvox;     // [max_rows] b*V+v of a row
  int32_t *class_cnt;  // [8] rows per parity class
  int32_t *class_fill; // [8]
  float *A;            // [max_rows][Cs]
  float *C;            // [max_rows][8*Cout]
};

__device__ __forceinline__ int parity_class(int ix, int iy, int iz) {
  return ((ix + 1) & 1) | (((iy + 1) & 1) << 1) | (((iz + 1) & 1) << 2);
}

// Index build.  Same-address global atomics cost ~12 ns each, so both passes first count in
// LDS (32-bit LDS atomics are cheap) and touch the 8 global class counters once per
// workgroup of 2048 voxels.
constexpr int kIdxPerThread = 8;

__device__ __forceinline__ int voxel_class(const ScArgs &a, int64_t i, int64_t total) {
  if (i >= total || a.counts[i] <= 0) return -1;
  const int V = a.D * a.D * a.D;
  const int v = (int)(i % V);
  return parity_class(v / (a.D * a.D), (v / a.D) % a.D, v % a.D);
}

// pass 1: rows per class
__global__ __launch_bounds__(256) void k_sc_count(ScArgs a) {
  __shared__ int s_cnt[8];
  if (threadIdx.x < 8) s_cnt[threadIdx.x] = 0;
  __syncthreads();
  const int64_t total = (int64_t)a.B * a.D * a.D * a.D;
  const int64_t base = (int64_t)blockIdx.x * 256 * kIdxPerThread;
#pragma unroll
  for (int j = 0; j < kIdxPerThread; ++j) {
    const int cls = voxel_class(a, base + j * 256 + threadIdx.x, total);
    if (cls >= 0) atomicAdd(&s_cnt[cls], 1);
  }
  __syncthreads();
  if (threadIdx.x < 8 && s_cnt[threadIdx.x] > 0) atomicAdd(&a.class_cnt[threadIdx.x], s_cnt[threadIdx.x]);
}

// pass 2: compact row ids (class-major), row -> voxel map.  Row order inside a class is
// arbitrary (it only names rows); every sum that depends on order is taken by tap index.
__global__ __launch_bounds__(256) void k_sc_assign(ScArgs a) {
  __shared__ int s_cnt[8], s_base[8], s_fill[8];
  if (threadIdx.x < 8) { s_cnt[threadIdx.x] = 0; s_fill[threadIdx.x] = 0; }
  __syncthreads();
  const int64_t total = (int64_t)a.B * a.D * a.D * a.D;
  const int64_t base = (int64_t)blockIdx.x * 256 * kIdxPerThread;
  int cls[kIdxPerThread];
#pragma unroll
  for (int j = 0; j < kIdxPerThread; ++j) {
    cls[j] = voxel_class(a, base + j * 256 + threadIdx.x, total);
    if (cls[j] >= 0) atomicAdd(&s_cnt[cls[j]], 1);
  }
  __syncthreads();
  if (threadIdx.x < 8) {
    int off = 0;
    for (int k = 0; k < (int)threadIdx.x; ++k) off += a.class_cnt[k];
    s_base[threadIdx.x] =
        off + (s_cnt[threadIdx.x] > 0 ? atomicAdd(&a.class_fill[threadIdx.x], s_cnt[threadIdx.x]) : 0);
  }
  __syncthreads();
#pragma unroll
  for (int j = 0; j < kIdxPerThread; ++j) {
    const int64_t i = base + j * 256 + threadIdx.x;
    if (i >= total) continue;
    int row = -1;
    if (cls[j] >= 0) row = s_base[cls[j]] + atomicAdd(&s_fill[cls[j]], 1);
    if (row >= a.max_rows) row = -1;
    a.rowmap[i] = row;
    if (row >= 0) a.rowvox[row] = (int32_t)i;
  }
}

// A[row][c] = x[b][c][v]  (lanes over channels: coalesced stores, strided gathers)
__global__ __launch_bounds__(256) void k_sc_gather(ScArgs a) {
  const int V = a.D * a.D * a.D;
  int n_rows = 0;
#pragma unroll
  for (int c = 0; c < 8; ++c) n_rows += a.class_cnt[c];
  n_rows = min(n_rows, a.max_rows);
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int row = (int)(i / a.Cs), ch = (int)(i % a.Cs);
  if (row >= n_rows) return;
  const int bv = a.rowvox[row];
  const int b = bv / V, v = bv % V;
  a.A[i] = a.x[((int64_t)b * a.Cs + ch) * V + v];
}

// ---- front end WITHOUT the dense tensor (inference) ------------------------------------
// average_voxelization_3d leaves <= 3 % of the voxels occupied; handing that through a dense
// [B,Cs,D,D,D] tensor costs a 151 MB zero-fill (B = 8) and a strided re-gather.  Here the
// per-voxel point chains (voxel_chain.h) feed the compact A rows directly: link -> count/assign
// -> one wave per occupied voxel writes its mean row into A[rowmap[voxel]].
__global__ __launch_bounds__(256) void k_sc_link(const float *__restrict__ points,
                                                 const int32_t *__restrict__ batch_indices, int64_t n,
                                                 int B, int D, float ox, float oy, float oz, float pitch,
                                                 int32_t *__restrict__ counts, int32_t *__restrict__ head,
                                                 int32_t *__restrict__ link) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  mf::chain_link(points, batch_indices, i, B, D, D, D, ox, oy, oz, pitch, counts, head, link, nullptr);
}

__global__ __launch_bounds__(256) void k_sc_rows_from_chains(ScArgs a, const float *__restrict__ values,
                                                            const float *__restrict__ points,
                                                            const int32_t *__restrict__ batch_indices,
                                                            int64_t n, float ox, float oy, float oz,
                                                            float pitch, const int32_t *__restrict__ head,
                                                            const int32_t *__restrict__ link, int64_t ldv) {
  __shared__ int s_ids[4][64];
  __shared__ int s_sorted[4][64];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int64_t i = (int64_t)blockIdx.x * 4 + wave;
  if (i >= n) return;
  // the row of this point's voxel (only used by the chain-head wave)
  int v;
  bool has_nan;
  const bool ok = mf::voxel_of(points, i, ox, oy, oz, pitch, a.D, a.D, a.D, v, has_nan);
  const int b = batch_indices[i];
  int row = -1;
  if (ok && b >= 0 && b < a.B) row = a.rowmap[(int64_t)b * a.D * a.D * a.D + v];
  if (row < 0) return;  // outside the grid, or beyond max_rows (wave-uniform)
  float *dst = a.A + (int64_t)row * a.Cs;
  mf::chain_mean(values, points, batch_indices, a.counts, head, link, i, a.Cs, a.B, a.D, a.D, a.D, ox, oy, oz,
                 pitch, s_ids[wave], s_sorted[wave], lane, [&](int ch, float mean) { dst[ch] = mean; }, ldv);
}

// C[row][n] = sum_k A[row][k] * Wp[class(row)][k][n],  n in [0, 8*Cout)
// Work list = (class, row tile, column tile).  Both operands of a tile sit in LDS
// ([k][row] and [k][col]: conflict-free fragment reads); K = Cs is small enough to
// stage completely.  Each wave owns 16 rows x 64 columns = 4 accumulators.
__global__ __launch_bounds__(256) void k_sc_gemm(ScArgs a, const float *__restrict__ Wp) {
  MF_DYN_LDS(float, s_mem);
  const int K = a.Cs, N = 8 * a.Cout;
  const int n_ntiles = N / kTN;
  float *As = s_mem;                  // [K][kTM + 4]
  float *Bs = s_mem + K * (kTM + 4);  // [K][kTN + 4]
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int lr = lane & 15, lk = lane >> 4;
  // class tile tables (8 classes): every workgroup walks the REAL tiles only, grid-strided
  int cls_rows[8], cls_row0[8], cls_tile0[9];
  {
    int r0 = 0, t0 = 0;
#pragma unroll
    for (int c = 0; c < 8; ++c) {
      const int n = min(a.class_cnt[c], max(0, a.max_rows - r0));
      cls_rows[c] = n;
      cls_row0[c] = r0;
      cls_tile0[c] = t0;
      r0 += a.class_cnt[c];
      t0 += ((n + kTM - 1) / kTM) * n_ntiles;
    }
    cls_tile0[8] = t0;
  }
  for (int tile = blockIdx.x; tile < cls_tile0[8]; tile += gridDim.x) {
    int cls = 0;
#pragma unroll
    for (int c = 1; c < 8; ++c) cls += tile >= cls_tile0[c] ? 1 : 0;
    const int local = tile - cls_tile0[cls];
    const int m0 = (local / n_ntiles) * kTM, n0 = (local % n_ntiles) * kTN;
    const int n_cls = cls_rows[cls], row0 = cls_row0[cls];
    __syncthreads();  // LDS tiles of the previous iteration are no longer read
    // stage A transposed (rows beyond the class are zero) and B
    for (int i = threadIdx.x; i < kTM * (K / 4); i += 256) {
      const int r = i / (K / 4), k4 = i % (K / 4);
      float4 v = make_float4(0, 0, 0, 0);
      if (m0 + r < n_cls)
        v = *reinterpret_cast<const float4 *>(a.A + (int64_t)(row0 + m0 + r) * K + 4 * k4);
      As[(4 * k4 + 0) * (kTM + 4) + r] = v.x;
      As[(4 * k4 + 1) * (kTM + 4) + r] = v.y;
      As[(4 * k4 + 2) * (kTM + 4) + r] = v.z;
      As[(4 * k4 + 3) * (kTM + 4) + r] = v.w;
    }
    const float *Wc = Wp + (int64_t)cls * K * N;
    for (int i = threadIdx.x; i < K * (kTN / 4); i += 256) {
      const int k = i / (kTN / 4), c4 = i % (kTN / 4);
      *reinterpret_cast<float4 *>(Bs + k * (kTN + 4) + 4 * c4) =
          *reinterpret_cast<const float4 *>(Wc + (int64_t)k * N + n0 + 4 * c4);
    }
    __syncthreads();
    f32x4 acc[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[j] = (f32x4){0.0f, 0.0f, 0.0f, 0.0f};
    for (int k0 = 0; k0 < K; k0 += 4) {
      // A fragment: lane l holds A[row = l&15][k = l>>4];  B: B[k = l>>4][col = l&15]
      const float af = As[(k0 + lk) * (kTM + 4) + wave * 16 + lr];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float bf = Bs[(k0 + lk) * (kTN + 4) + j * 16 + lr];
        acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(af, bf, acc[j], 0, 0, 0);
      }
    }
    // C/D fragment: col = lane&15, row = (lane>>4)*4 + i
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int r = m0 + wave * 16 + lk * 4 + i;
        if (r < n_cls) a.C[(int64_t)(row0 + r) * N + n0 + j * 16 + lr] = acc[j][i];
      }
  }
}

// out[b][co][o] = relu( sum_taps C[row(v)][slot*Cout + co] + dense[b][co][o] + bias[co] )
// One workgroup = 64 consecutive output voxels x all Cout channels, staged through LDS so
// that the final stores are coalesced along o.  Each wave walks 4 output voxels; for one
// voxel the 64 lanes look up its 64 taps at once (one rowmap gather), then every lane
// accumulates Cout/64 channels over the (few) taps that hit an occupied voxel, in tap order.
constexpr int kRedThreads = 1024;  // 16 waves x 4 output voxels: short dependent-load chains

__global__ __launch_bounds__(kRedThreads) void k_sc_reduce(ScArgs a, const float *__restrict__ dense,
                                                          const float *__restrict__ bias, int relu,
                                                          float *__restrict__ out) {
  MF_DYN_LDS(float, s_tile);  // [Cout][65]
  const int D = a.D, Do = D / 2, V = D * D * D, Vo = Do * Do * Do;
  const int b = blockIdx.y;
  const int o0 = blockIdx.x * 64;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  constexpr int kWaves = kRedThreads / 64, kPerWave = 64 / kWaves;
  const int N = 8 * a.Cout;
  const int nj = a.Cout / 64;
  // this lane's tap: k = (kx,ky,kz) in [0,4)^3, fixed for the whole kernel
  const int kx = lane >> 4, ky = (lane >> 2) & 3, kz = lane & 3;
  const int slot = (kx >> 1) | ((ky >> 1) << 1) | ((kz >> 1) << 2);
  int rows[kPerWave];
#pragma unroll
  for (int t = 0; t < kPerWave; ++t) {  // all row-map lookups of this wave in flight at once
    const int o = o0 + wave * kPerWave + t;
    rows[t] = -1;
    if (o < Vo) {
      const int oz = o % Do, oy = (o / Do) % Do, oxx = o / (Do * Do);
      const int vx = 2 * oxx - 1 + kx, vy = 2 * oy - 1 + ky, vz = 2 * oz - 1 + kz;
      if (vx >= 0 && vx < D && vy >= 0 && vy < D && vz >= 0 && vz < D)
        rows[t] = a.rowmap[(int64_t)b * V + (vx * D + vy) * D + vz];
    }
  }
#pragma unroll
  for (int t = 0; t < kPerWave; ++t) {
    const int ol = wave * kPerWave + t;
    float acc[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[j] = 0.0f;
    unsigned long long hits = __ballot(rows[t] >= 0);
    while (hits) {  // taps in increasing k: fixed summation order
      const int src = __ffsll((long long)hits) - 1;
      hits &= hits - 1;
      const int r = __shfl(rows[t], src, 64);
      const int sl = __shfl(slot, src, 64);
      const float *crow = a.C + (int64_t)r * N + sl * a.Cout;
      for (int j = 0; j < nj; ++j) acc[j] += crow[lane + 64 * j];
    }
    for (int j = 0; j < nj; ++j) s_tile[(lane + 64 * j) * 65 + ol] = acc[j];
  }
  __syncthreads();
  for (int co = wave; co < a.Cout; co += kWaves) {
    const int o = o0 + lane;
    if (o < Vo) {
      const int64_t idx = ((int64_t)b * a.Cout + co) * Vo + o;
      float v = s_tile[co * 65 + lane] + (dense ? dense[idx] : 0.0f) + (bias ? bias[co] : 0.0f);
      if (relu) v = v > 0.0f ? v : 0.0f;
      out[idx] = v;
    }
  }
}

// The same reduce for a CHANNELS-LAST output [B][Vo][Cout] (what conv4's implicit GEMM and the
// channels-last trilinear sampler consume): lanes run over channels, 4 each -- a contributing
// (row, slot) is ONE 16-byte load per lane (a whole 1 KB row segment per wave), the store is a
// 16-byte row segment, no LDS transpose.  Same taps in the same order, then + dense + bias:
// the same bits as k_sc_reduce.  Cout % 256 == 0, Cout <= 512.
__global__ __launch_bounds__(kRedThreads) void k_sc_reduce_cl(ScArgs a, const float *__restrict__ dense,
                                                             const float *__restrict__ bias, int relu,
                                                             float *__restrict__ out) {
  const int D = a.D, Do = D / 2, V = D * D * D, Vo = Do * Do * Do;
  const int b = blockIdx.y;
  const int o0 = blockIdx.x * 64;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  constexpr int kWaves = kRedThreads / 64, kPerWave = 64 / kWaves;
  const int N = 8 * a.Cout;
  const int nj = a.Cout / 256;
  const int kx = lane >> 4, ky = (lane >> 2) & 3, kz = lane & 3;
  const int slot = (kx >> 1) | ((ky >> 1) << 1) | ((kz >> 1) << 2);
  int rows[kPerWave];
#pragma unroll
  for (int t = 0; t < kPerWave; ++t) {
    const int o = o0 + wave * kPerWave + t;
    rows[t] = -1;
    if (o < Vo) {
      const int oz = o % Do, oy = (o / Do) % Do, oxx = o / (Do * Do);
      const int vx = 2 * oxx - 1 + kx, vy = 2 * oy - 1 + ky, vz = 2 * oz - 1 + kz;
      if (vx >= 0 && vx < D && vy >= 0 && vy < D && vz >= 0 && vz < D)
        rows[t] = a.rowmap[(int64_t)b * V + (vx * D + vy) * D + vz];
    }
  }
#pragma unroll
  for (int t = 0; t < kPerWave; ++t) {
    const int o = o0 + wave * kPerWave + t;
    if (o >= Vo) continue;  // wave-uniform
    const int64_t obase = ((int64_t)b * Vo + o) * a.Cout;
    float4 dn[2], acc[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {  // the dense addend's load overlaps the tap walk
      dn[j] = acc[j] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
      if (j >= nj) continue;
      dn[j] = dense ? *reinterpret_cast<const float4 *>(dense + obase + 256 * j + 4 * lane)
                    : make_float4(0.0f, 0.0f, 0.0f, 0.0f);
      acc[j] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    }
    unsigned long long hits = __ballot(rows[t] >= 0);
    while (hits) {  // taps in increasing k: fixed summation order
      const int src = __ffsll((long long)hits) - 1;
      hits &= hits - 1;
      const int r = __shfl(rows[t], src, 64);
      const int sl = __shfl(slot, src, 64);
      const float *crow = a.C + (int64_t)r * N + sl * a.Cout + 4 * lane;
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        if (j >= nj) continue;
        const float4 c = *reinterpret_cast<const float4 *>(crow + 256 * j);
        acc[j].x += c.x; acc[j].y += c.y; acc[j].z += c.z; acc[j].w += c.w;
      }
    }
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      if (j >= nj) continue;
      float4 bs = bias ? *reinterpret_cast<const float4 *>(bias + 256 * j + 4 * lane)
                       : make_float4(0.0f, 0.0f, 0.0f, 0.0f);
      float4 v;
      v.x = acc[j].x + dn[j].x + bs.x; v.y = acc[j].y + dn[j].y + bs.y;
      v.z = acc[j].z + dn[j].z + bs.z; v.w = acc[j].w + dn[j].w + bs.w;
      if (relu) {
        v.x = v.x > 0.0f ? v.x : 0.0f; v.y = v.y > 0.0f ? v.y : 0.0f;
        v.z = v.z > 0.0f ? v.z : 0.0f; v.w = v.w > 0.0f ? v.w : 0.0f;
      }
      *reinterpret_cast<float4 *>(out + obase + 256 * j + 4 * lane) = v;
    }
  }
}

// Wp[class][c][slot*Cout + co] = W[co][c][kx][ky][kz],  k = parity + 2*slot bit per axis
__global__ void k_sc_pack(const float *__restrict__ W, int Cout, int Cs, int w_cin, int c_off,
                          float *__restrict__ Wp) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t total = (int64_t)8 * Cs * 8 * Cout;
  if (i >= total) return;
  const int co = (int)(i % Cout);
  const int slot = (int)((i / Cout) % 8);
  const int c = (int)((i / ((int64_t)8 * Cout)) % Cs);
  const int cls = (int)(i / ((int64_t)8 * Cout * Cs));
  const int kx = (cls & 1) + 2 * (slot & 1), ky = ((cls >> 1) & 1) + 2 * ((slot >> 1) & 1),
            kz = ((cls >> 2) & 1) + 2 * ((slot >> 2) & 1);
  Wp[i] = W[((((int64_t)co * w_cin + c_off + c) * 4 + kx) * 4 + ky) * 4 + kz];
}

int64_t align256(int64_t x) { return (x + 255) & ~(int64_t)255; }

}  // namespace

extern "C" int64_t mf_sparse_conv3d_workspace_bytes(int32_t B, int32_t Cs, int32_t Cout, int32_t D,
                                                    int32_t max_rows, int64_t n_points) {
  const int64_t V = (int64_t)D * D * D;
  return align256(B * V * 4) + align256((int64_t)max_rows * 4) + align256(64) + align256(64) +
         align256((int64_t)max_rows * Cs * 4) + align256((int64_t)max_rows * 8 * Cout * 4) +
         2 * align256(B * V * 4) + align256(n_points * 4);  // counts, chain heads, chain links
}

extern "C" int mf_sparse_conv3d_pack_weights(const float *W, int32_t Cout, int32_t Cs,
                                             int32_t w_cin, int32_t c_off, float *Wp,
                                             mfStream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  const int64_t total = (int64_t)8 * Cs * 8 * Cout;
  hipLaunchKernelGGL(k_sc_pack, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, W,
                     Cout, Cs, w_cin, c_off, Wp);
  return mf::check_launch("mf_sparse_conv3d_pack_weights");
}

namespace {

struct ScWs {
  ScArgs a;
  int32_t *head, *link, *counts_own;
};

// workspace carve-up shared by both entry points (n_points == 0: no chain arrays)
ScWs sc_carve(void *ws, int B, int Cs, int Cout, int D, int max_rows, int64_t n_points) {
  const int64_t V = (int64_t)D * D * D;
  ScWs w;
  w.a.B = B; w.a.Cs = Cs; w.a.Cout = Cout; w.a.D = D; w.a.max_rows = max_rows;
  w.a.x = nullptr; w.a.counts = nullptr;
  char *p = (char *)ws;
  w.a.rowmap = (int32_t *)p; p += align256(B * V * 4);
  w.a.rowvox = (int32_t *)p; p += align256((int64_t)max_rows * 4);
  w.a.class_cnt = (int32_t *)p; p += align256(64);
  w.a.class_fill = (int32_t *)p; p += align256(64);
  w.a.A = (float *)p; p += align256((int64_t)max_rows * Cs * 4);
  w.a.C = (float *)p; p += align256((int64_t)max_rows * 8 * Cout * 4);
  w.counts_own = (int32_t *)p; p += align256(B * V * 4);
  w.head = (int32_t *)p; p += align256(B * V * 4);
  w.link = (int32_t *)p;
  return w;
}

int sc_check(int B, int Cs, int Cout, int D) {
  if (Cs % 4 || Cout % 64 || Cout > 512 || D % 2 || D > 64) {
    mf::set_last_error(hipErrorInvalidValue, "sparse_conv3d: need Cs%4==0, Cout%64==0 (<=512), even D");
    return -(int)hipErrorInvalidValue;
  }
  const size_t lds_g = (size_t)Cs * (kTM + 4 + kTN + 4) * sizeof(float);
  if (lds_g > 150 * 1024) {
    mf::set_last_error(hipErrorInvalidValue, "sparse_conv3d: Cs too large for the LDS-resident K");
    return -(int)hipErrorInvalidValue;
  }
  if (int e = mf::allow_big_lds((const void *)k_sc_gemm, 150 * 1024)) return e;
  return mf::allow_big_lds((const void *)k_sc_reduce, 150 * 1024);
}

// rows per class + compact row ids from a.counts
void sc_index(const ScArgs &a, hipStream_t stream) {
  const int64_t V = (int64_t)a.D * a.D * a.D;
  const unsigned nb = (unsigned)((a.B * V + 256 * kIdxPerThread - 1) / (256 * kIdxPerThread));
  hipLaunchKernelGGL(k_sc_count, dim3(nb), dim3(256), 0, stream, a);
  hipLaunchKernelGGL(k_sc_assign, dim3(nb), dim3(256), 0, stream, a);
}

// 8 parity-class GEMMs + output-stationary reduce
void sc_gemm_reduce(const ScArgs &a, const float *Wp, const float *dense, const float *bias, int relu,
                    float *out, hipStream_t stream, int channels_last = 0) {
  const size_t lds_g = (size_t)a.Cs * (kTM + 4 + kTN + 4) * sizeof(float);
  // persistent-style: 2 workgroups per CU walk the (class, row tile, column tile) list
  hipLaunchKernelGGL(k_sc_gemm, dim3(512), dim3(256), lds_g, stream, a, Wp);
  const int Vo = (a.D / 2) * (a.D / 2) * (a.D / 2);
  if (channels_last) {
    hipLaunchKernelGGL(k_sc_reduce_cl, dim3((Vo + 63) / 64, a.B), dim3(kRedThreads), 0, stream, a, dense, bias,
                       relu, out);
    return;
  }
  hipLaunchKernelGGL(k_sc_reduce, dim3((Vo + 63) / 64, a.B), dim3(kRedThreads),
                     (size_t)a.Cout * 65 * sizeof(float), stream, a, dense, bias, relu, out);
}

}  // namespace

extern "C" int mf_sparse_conv3d_k4s2_fwd(const float *x, const int32_t *counts, const float *Wp,
                                         const float *dense, const float *bias, float *out,
                                         void *ws, int32_t B, int32_t Cs, int32_t Cout, int32_t D,
                                         int32_t max_rows, int32_t relu, mfStream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (B <= 0 || max_rows <= 0) return 0;
  if (int e = sc_check(B, Cs, Cout, D)) return e;
  ScWs w = sc_carve(ws, B, Cs, Cout, D, max_rows, 0);
  w.a.x = x;
  w.a.counts = counts;
  if (int e_ = mf::fill_bytes(w.a.class_cnt, 0, 512, stream)) return e_;  // class_cnt and class_fill
  sc_index(w.a, stream);
  hipLaunchKernelGGL(k_sc_gather, dim3((unsigned)(((int64_t)max_rows * Cs + 255) / 256)),
                     dim3(256), 0, stream, w.a);
  sc_gemm_reduce(w.a, Wp, dense, bias, relu, out, stream);
  return mf::check_launch("mf_sparse_conv3d_k4s2_fwd");
}

extern "C" int mf_sparse_conv3d_k4s2_points_fwd(const float *values, const float *points,
                                                const int32_t *batch_indices, int64_t n, float ox,
                                                float oy, float oz, float pitch, const float *Wp,
                                                const float *dense, const float *bias, float *out,
                                                void *ws, int32_t B, int32_t Cs, int32_t Cout,
                                                int32_t D, int32_t max_rows, int32_t relu,
                                                mfStream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (B <= 0 || max_rows <= 0) return 0;
  if (int e = sc_check(B, Cs, Cout, D)) return e;
  const int64_t V = (int64_t)D * D * D;
  ScWs w = sc_carve(ws, B, Cs, Cout, D, max_rows, n);
  w.a.counts = w.counts_own;
  if (int e_ = mf::fill_bytes(w.a.class_cnt, 0, 512, stream)) return e_;
  if (int e_ = mf::fill_bytes(w.counts_own, 0, sizeof(int32_t) * B * V, stream)) return e_;
  if (int e_ = mf::fill_bytes(w.head, 0xff, sizeof(int32_t) * B * V, stream)) return e_;
  if (n > 0)
    hipLaunchKernelGGL(k_sc_link, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, points,
                       batch_indices, n, B, D, ox, oy, oz, pitch, w.counts_own, w.head, w.link);
  sc_index(w.a, stream);
  if (n > 0)
    hipLaunchKernelGGL(k_sc_rows_from_chains, dim3((unsigned)((n + 3) / 4)), dim3(256), 0, stream, w.a,
                       values, points, batch_indices, n, ox, oy, oz, pitch, w.head, w.link, (int64_t)Cs);
  sc_gemm_reduce(w.a, Wp, dense, bias, relu, out, stream);
  return mf::check_launch("mf_sparse_conv3d_k4s2_points_fwd");
}

extern "C" int mf_sparse_conv3d_k4s2_points_cl_fwd(const float *values, int64_t ldv, const float *points,
                                                const int32_t *batch_indices, int64_t n, float ox,
                                                float oy, float oz, float pitch, const float *Wp,
                                                const float *dense, const float *bias, float *out,
                                                void *ws, int32_t B, int32_t Cs, int32_t Cout,
                                                int32_t D, int32_t max_rows, int32_t relu,
                                                mfStream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (B <= 0 || max_rows <= 0) return 0;
  if (int e = sc_check(B, Cs, Cout, D)) return e;
  if (Cout % 256 || ldv < Cs) {
    mf::set_last_error(hipErrorInvalidValue, "sparse_conv3d (channels-last): need Cout % 256 == 0, ldv >= Cs");
    return -(int)hipErrorInvalidValue;
  }
  const int64_t V = (int64_t)D * D * D;
  ScWs w = sc_carve(ws, B, Cs, Cout, D, max_rows, n);
  w.a.counts = w.counts_own;
  if (int e_ = mf::fill_bytes(w.a.class_cnt, 0, 512, stream)) return e_;
  if (int e_ = mf::fill_bytes(w.counts_own, 0, sizeof(int32_t) * B * V, stream)) return e_;
  if (int e_ = mf::fill_bytes(w.head, 0xff, sizeof(int32_t) * B * V, stream)) return e_;
  if (n > 0)
    hipLaunchKernelGGL(k_sc_link, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, points,
                       batch_indices, n, B, D, ox, oy, oz, pitch, w.counts_own, w.head, w.link);
  sc_index(w.a, stream);
  if (n > 0)
    hipLaunchKernelGGL(k_sc_rows_from_chains, dim3((unsigned)((n + 3) / 4)), dim3(256), 0, stream, w.a,
                       values, points, batch_indices, n, ox, oy, oz, pitch, w.head, w.link, ldv);
  sc_gemm_reduce(w.a, Wp, dense, bias, relu, out, stream, 1);
  return mf::check_launch("mf_sparse_conv3d_k4s2_points_cl_fwd");
}
