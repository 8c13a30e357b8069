// conv3 of the pose network in the bf16 TRAINING path, on occupied voxels only: forward, data gradient and weight
// gradient (round 5).
//
// Reference: `L.Convolution3D(None, 256, 4, 2, pad=1)` on the voxelized point features + the 16 occupancy channels
// (contrib/singleview_3d/models/model.py:73,114-128), trained by examples/ycb_video/singleview_3d/train.py:342-369
// (cuDNN forward / backward-data / backward-filter over the dense [B,160,32^3] tensor).  average_voxelization_3d
// leaves <= P = 1000 of the 32768 voxels of an object occupied in 144 of those 160 channels; round 4 still
// zero-filled the dense 168 MB input (B = 16) and ran three dense implicit GEMMs over it (0.44 + 0.64 + 0.41 ms).
//
// Here the 144 voxelized channels never exist as a grid:
//   index     per-voxel point chains (voxel_chain.h) -> occupied voxels -> compact rows, CLASS-MAJOR (k4 / s2 / p1: a
//             voxel feeds output o = ((v + 1) >> 1) - a through tap k = p + 2 a per axis, p = (v + 1) & 1, a in {0, 1}:
//             8 parity classes, all voxels of a class use the same 8 taps = "slots"), every class padded to a
//             multiple of 128 rows so that a GEMM row tile never straddles classes; tables: the class of every
//             64-row block, the row range of every class.
//   forward   A [rows][144] = mean point rows (k_avgvox_cl_fwd through the row map)
//             C [rows][8 slots x Cout] = A Wp[class]^T            (k_gemm_nt_bf16, rows mode + tile_group table)
//             out[b][o][:] = relu(dense[b][o][:] + bias + sum over the <= 64 (voxel, tap) pairs of o of its C segment)
//             (k_scb_reduce; `dense` = the 16 occupancy channels through the dense bf16 engine, fp32 pre-activation)
//   backward  dz = relu mask (k_relu_mask_bf16), dYg [rows][8 x Cout] = dz gathered at the 8 outputs of every row
//             dA = dYg Wq[class]^T (the same engine) -> point gradients (k_avgvox_cl_bwd through the row map)
//             dWp[class] = dYg^T A over the class's row range (k_gemm_tn_bf16 + m_range table) -> scattered into the
//             framework layout [Cout][160][4][4][4]
//   the 16 occupancy channels keep the dense engines with Cin = 16 (1 / 10 of the dense arithmetic).
// Everything is deterministic: fixed summation orders; the compact rows are handed out by a prefix scan over the
// voxel index (integer atomics only build the per-voxel point chains, whose sums are taken in point order).
#include "mf_common.h"

namespace {

constexpr int kPadRows = 128;

struct ScbWs {  // workspace carve-up (device pointers)
  int32_t *counts, *head, *link, *rowmap, *rowvox, *class_cnt, *blk_cnt, *class_off, *tile_group;
  int64_t max_rows;
};

inline int64_t align256(int64_t x) { return (x + 255) & ~(int64_t)255; }

constexpr int kIdxPerThread = 8;  // voxels per thread of the index kernels
inline int64_t scb_blocks(int64_t BV) { return (BV + 256 * kIdxPerThread - 1) / (256 * kIdxPerThread); }

inline int64_t scb_max_rows(int64_t n) { return ((n + 8 * (kPadRows - 1) + kPadRows - 1) / kPadRows) * kPadRows; }

inline int64_t scb_carve(void *ws, int64_t n, int B, int D, ScbWs *w) {
  const int64_t BV = (int64_t)B * D * D * D, Mp = scb_max_rows(n);
  char *p = (char *)ws;
  int64_t off = 0;
  auto take = [&](int64_t bytes) { char *q = p ? p + off : nullptr; off = align256(off + bytes); return q; };
  w->counts = (int32_t *)take(BV * 4);
  w->head = (int32_t *)take(BV * 4);
  w->link = (int32_t *)take((n > 0 ? n : 1) * 4);
  w->rowmap = (int32_t *)take(BV * 4);
  w->rowvox = (int32_t *)take(Mp * 4);
  w->class_cnt = (int32_t *)take(8 * 4);
  w->blk_cnt = (int32_t *)take(scb_blocks(BV) * 8 * 4);  // per (index workgroup, class): count, then exclusive prefix
  w->class_off = (int32_t *)take(9 * 4);
  w->tile_group = (int32_t *)take((Mp / 64) * 4);
  w->max_rows = Mp;
  return off;
}

__device__ __forceinline__ int parity_class(int ix, int iy, int iz) {
  return ((ix + 1) & 1) | (((iy + 1) & 1) << 1) | (((iz + 1) & 1) << 2);
}

__device__ __forceinline__ int voxel_class(const int32_t *counts, int D, int64_t i, int64_t total) {
  if (i >= total || counts[i] <= 0) return -1;
  const int V = D * D * D;
  const int v = (int)(i % V);
  return parity_class(v / (D * D), (v / D) % D, v % D);
}

// Row assignment is a DETERMINISTIC prefix scan over the voxel index (round 6; it handed rows out with LDS / global
// atomicAdd before: the order of the voxels inside a class then changed from launch to launch, and with it the
// fp32 summation order of conv3's weight gradient -- dW was not bitwise reproducible):
//   k_scb_count    per (workgroup of 2048 voxels, class): occupied voxels            -> blk_cnt
//   k_scb_offsets  per class: total, padded class offsets; blk_cnt -> exclusive prefix over the workgroups
//   k_scb_assign   row = class offset + workgroup prefix + rank of the voxel among the workgroup's voxels of its class
//                  in increasing voxel index (ballot / popcount)
__global__ __launch_bounds__(256) void k_scb_count(ScbWs w, int B, int D) {
  __shared__ int s_cnt[8];
  if (threadIdx.x < 8) s_cnt[threadIdx.x] = 0;
  __syncthreads();
  const int64_t total = (int64_t)B * D * D * D;
  const int64_t base = (int64_t)blockIdx.x * 256 * kIdxPerThread;
#pragma unroll
  for (int j = 0; j < kIdxPerThread; ++j) {
    const int cls = voxel_class(w.counts, D, base + j * 256 + threadIdx.x, total);
    if (cls >= 0) atomicAdd(&s_cnt[cls], 1);  // (integer count: the order does not matter)
  }
  __syncthreads();
  if (threadIdx.x < 8) w.blk_cnt[(int64_t)blockIdx.x * 8 + threadIdx.x] = s_cnt[threadIdx.x];
}

// class totals + exclusive prefixes over the index workgroups (32 lanes per class: chunks of 32 workgroups, a
// shuffle scan inside the chunk), padded class offsets (the row ranges of the weight-gradient GEMM) and the class of
// every 64-row block
__global__ __launch_bounds__(256) void k_scb_offsets(ScbWs w, int nblocks) {
  __shared__ int s_off[9], s_tot[8];
  {
    const int c = threadIdx.x >> 5, l = threadIdx.x & 31;
    int carry = 0;
    for (int b0 = 0; b0 < nblocks; b0 += 32) {
      const int b = b0 + l;
      const int v = b < nblocks ? w.blk_cnt[(int64_t)b * 8 + c] : 0;
      int inc = v;
#pragma unroll
      for (int d = 1; d < 32; d <<= 1) {
        const int up = __shfl_up(inc, d);  // (lanes below d of a half read across the half's edge: unused)
        if (l >= d) inc += up;
      }
      if (b < nblocks) w.blk_cnt[(int64_t)b * 8 + c] = carry + inc - v;
      carry += __shfl(inc, (int)(threadIdx.x & 32) | 31);  // the last lane of this class's 32-lane half
    }
    if (l == 0) { s_tot[c] = carry; w.class_cnt[c] = carry; }
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    int off = 0;
    for (int c = 0; c < 8; ++c) {
      s_off[c] = off;
      off += (s_tot[c] + kPadRows - 1) / kPadRows * kPadRows;
    }
    s_off[8] = off;
    for (int c = 0; c < 9; ++c) w.class_off[c] = s_off[c];
  }
  __syncthreads();
  const int nblk = (int)(w.max_rows / 64);
  for (int t = threadIdx.x; t < nblk; t += blockDim.x) {
    const int r = t * 64;
    int g = -1;
    for (int c = 0; c < 8; ++c)
      if (r >= s_off[c] && r < s_off[c + 1]) g = c;
    w.tile_group[t] = g;
  }
}

// compact row ids (class-major, padded class starts), row -> voxel map; pad rows keep rowvox = -1
__global__ __launch_bounds__(256) void k_scb_assign(ScbWs w, int B, int D) {
  __shared__ int s_run[8], s_wave[4][8];
  if (threadIdx.x < 8)
    s_run[threadIdx.x] = w.class_off[threadIdx.x] + w.blk_cnt[(int64_t)blockIdx.x * 8 + threadIdx.x];
  __syncthreads();
  const int64_t total = (int64_t)B * D * D * D;
  const int64_t base = (int64_t)blockIdx.x * 256 * kIdxPerThread;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  for (int j = 0; j < kIdxPerThread; ++j) {  // voxel index order: j, then the thread
    const int64_t i = base + j * 256 + threadIdx.x;
    const int cls = voxel_class(w.counts, D, i, total);
    int before = 0;  // voxels of my class in lower lanes of my wave
#pragma unroll
    for (int c = 0; c < 8; ++c) {
      const unsigned long long m = __ballot(cls == c);
      if (cls == c) before = __popcll(m & ((1ull << lane) - 1ull));
      if (lane == 0) s_wave[wave][c] = __popcll(m);
    }
    __syncthreads();
    int row = -1;
    if (cls >= 0) {
      row = s_run[cls] + before;
      for (int w2 = 0; w2 < wave; ++w2) row += s_wave[w2][cls];
      if (row >= w.max_rows) row = -1;
    }
    if (i < total) {
      w.rowmap[i] = row;
      if (row >= 0) w.rowvox[row] = (int32_t)i;
    }
    __syncthreads();
    if (threadIdx.x < 8)
      s_run[threadIdx.x] += s_wave[0][threadIdx.x] + s_wave[1][threadIdx.x] + s_wave[2][threadIdx.x] + s_wave[3][threadIdx.x];
    __syncthreads();
  }
}

__global__ __launch_bounds__(256) void k_scb_link(const float *__restrict__ points, const int32_t *__restrict__ batch_indices,
                                                  int64_t n, int B, int D, ScbWs w) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  // (voxel_chain.h's chain_link with origin 0, pitch 1: the network's voxel frame, model.py:154-162)
  const float x = points[3 * i], y = points[3 * i + 1], z = points[3 * i + 2];
  const float rx = roundf(x), ry = roundf(y), rz = roundf(z);
  const bool ok = rx >= 0.0f && rx < (float)D && ry >= 0.0f && ry < (float)D && rz >= 0.0f && rz < (float)D;
  const int b = batch_indices[i];
  int32_t l = -2;
  if (ok && b >= 0 && b < B) {
    const int64_t key = (int64_t)b * D * D * D + ((int)rx * D + (int)ry) * D + (int)rz;
    atomicAdd(&w.counts[key], 1);
    l = atomicExch(&w.head[key], (int32_t)i);
  }
  w.link[i] = l;
}

// Wp[class][n = slot * Cout + co][k = c] and Wq[class][n = c][k = slot * Cout + co] (both k-contiguous bf16) from
// W fp32 [Cout][w_cin][4][4][4], input channels [c_off, c_off + Cs); tap = parity + 2 * slot bit per axis
__global__ void k_scb_pack(const float *__restrict__ W, int Cout, int Cs, int w_cin, int c_off, uint16_t *__restrict__ Wp,
                           uint16_t *__restrict__ Wq) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t total = (int64_t)8 * 8 * Cout * Cs;
  if (i >= total) return;
  const int c = (int)(i % Cs);
  const int co = (int)((i / Cs) % Cout);
  const int slot = (int)((i / ((int64_t)Cs * Cout)) % 8);
  const int cls = (int)(i / ((int64_t)Cs * Cout * 8));
  const int kx = (cls & 1) + 2 * (slot & 1), ky = ((cls >> 1) & 1) + 2 * ((slot >> 1) & 1),
            kz = ((cls >> 2) & 1) + 2 * ((slot >> 2) & 1);
  const float v = W[((((int64_t)co * w_cin + c_off + c) * 4 + kx) * 4 + ky) * 4 + kz];
  const uint16_t h = (uint16_t)mf::bf16_bits(v);
  const int64_t N8 = (int64_t)8 * Cout;
  Wp[((int64_t)cls * N8 + slot * Cout + co) * Cs + c] = h;
  if (Wq) Wq[((int64_t)cls * Cs + c) * N8 + slot * Cout + co] = h;
}

// dW[co][c_off + c][kx][ky][kz] = dWp[class][slot * Cout + co][c] (fp32), the inverse of the pack's index map
__global__ void k_scb_unpack_dw(const float *__restrict__ dWp, int Cout, int Cs, int w_cin, int c_off,
                                float *__restrict__ dW) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t total = (int64_t)Cout * Cs * 64;
  if (i >= total) return;
  const int tap = (int)(i % 64);
  const int c = (int)((i / 64) % Cs);
  const int co = (int)(i / ((int64_t)64 * Cs));
  const int kx = tap >> 4, ky = (tap >> 2) & 3, kz = tap & 3;
  const int cls = (kx & 1) | ((ky & 1) << 1) | ((kz & 1) << 2);
  const int slot = (kx >> 1) | ((ky >> 1) << 1) | ((kz >> 1) << 2);
  dW[(((int64_t)co * w_cin + c_off + c) * 64) + tap] = dWp[((int64_t)cls * 8 * Cout + slot * Cout + co) * Cs + c];
}

// out[b][o][:] = relu(dense[b][o][:] + bias + sum_taps C[row(v(o, tap))][slot(tap) * Cout ..]): one workgroup = 64
// consecutive output voxels, a wave walks 16 of them; for one voxel the 64 lanes look up its 64 taps at once (one
// rowmap gather), then every lane accumulates Cout / 64 channels (4 at Cout = 256: one 8-byte bf16 load per
// contributing row) over the taps that hit an occupied voxel, in tap order (fp32 sums).
constexpr int kRedThreads = 256;
__global__ __launch_bounds__(kRedThreads) void k_scb_reduce(const uint16_t *__restrict__ C, const float *__restrict__ dense,
                                                           const float *__restrict__ bias, const int32_t *__restrict__ rowmap,
                                                           int D, int Cout, int relu, uint16_t *__restrict__ out) {
  const int Do = D / 2, V = D * D * D, Vo = Do * Do * Do;
  const int b = blockIdx.y;
  const int o0 = blockIdx.x * 64;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  constexpr int kWaves = kRedThreads / 64, kPerWave = 64 / kWaves;
  const int64_t N = (int64_t)8 * Cout;
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
        rows[t] = rowmap[(int64_t)b * V + (vx * D + vy) * D + vz];
    }
  }
#pragma unroll
  for (int t = 0; t < kPerWave; ++t) {
    const int o = o0 + wave * kPerWave + t;
    if (o >= Vo) continue;  // wave-uniform
    const int64_t obase = ((int64_t)b * Vo + o) * Cout;
    for (int c0 = 4 * lane; c0 < Cout; c0 += 256) {  // (one trip at Cout = 256)
      float4 acc = dense ? *reinterpret_cast<const float4 *>(dense + obase + c0) : make_float4(0.0f, 0.0f, 0.0f, 0.0f);
      unsigned long long hits = __ballot(rows[t] >= 0);
      float4 s = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
      while (hits) {  // taps in increasing index: fixed summation order
        const int src = __ffsll((long long)hits) - 1;
        hits &= hits - 1;
        const int r = __shfl(rows[t], src, 64);
        const int sl = __shfl(slot, src, 64);
        const uint2 cw = *reinterpret_cast<const uint2 *>(C + (int64_t)r * N + sl * Cout + c0);
        s.x += mf::bf16_lo(cw.x); s.y += mf::bf16_hi(cw.x); s.z += mf::bf16_lo(cw.y); s.w += mf::bf16_hi(cw.y);
      }
      const float4 bs = bias ? *reinterpret_cast<const float4 *>(bias + c0) : make_float4(0.0f, 0.0f, 0.0f, 0.0f);
      float4 v;
      v.x = s.x + acc.x + bs.x; v.y = s.y + acc.y + bs.y; v.z = s.z + acc.z + bs.z; v.w = s.w + acc.w + bs.w;
      if (relu) {
        v.x = v.x > 0.0f ? v.x : 0.0f; v.y = v.y > 0.0f ? v.y : 0.0f;
        v.z = v.z > 0.0f ? v.z : 0.0f; v.w = v.w > 0.0f ? v.w : 0.0f;
      }
      *reinterpret_cast<uint2 *>(out + obase + c0) = make_uint2(mf::pack_bf16x2(v.x, v.y), mf::pack_bf16x2(v.z, v.w));
    }
  }
}

// dYg[row][slot * Cout + co] = dz[b][o(v_row, slot)][co] (zeros where that output lies outside the grid, and in pad
// rows): one wave per (row, slot), 16-byte chunks
__global__ __launch_bounds__(256) void k_scb_gather_dy(const uint16_t *__restrict__ dz, const int32_t *__restrict__ rowvox,
                                                       const int32_t *__restrict__ class_off, int D, int Cout,
                                                       uint16_t *__restrict__ dYg) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int64_t item = (int64_t)blockIdx.x * 4 + wave;
  const int64_t row = item >> 3;
  const int slot = (int)(item & 7);
  if (row >= class_off[8]) return;  // beyond the last class (wave-uniform)
  const int Do = D / 2, V = D * D * D, Vo = Do * Do * Do;
  const int bv = rowvox[row];
  const uint16_t *src = nullptr;
  if (bv >= 0) {
    const int b = bv / V, v = bv % V;
    const int vx = v / (D * D), vy = (v / D) % D, vz = v % D;
    const int ox = ((vx + 1) >> 1) - (slot & 1), oy = ((vy + 1) >> 1) - ((slot >> 1) & 1),
              oz = ((vz + 1) >> 1) - ((slot >> 2) & 1);
    if (ox >= 0 && ox < Do && oy >= 0 && oy < Do && oz >= 0 && oz < Do)
      src = dz + ((int64_t)b * Vo + (ox * Do + oy) * Do + oz) * Cout;
  }
  uint16_t *dst = dYg + row * (int64_t)8 * Cout + slot * Cout;
  for (int c0 = 8 * lane; c0 < Cout; c0 += 512) {
    uint4 v = make_uint4(0u, 0u, 0u, 0u);
    if (src) v = *reinterpret_cast<const uint4 *>(src + c0);
    *reinterpret_cast<uint4 *>(dst + c0) = v;
  }
}

// ---- data gradient of the k4 / s2 / p1 convolution for a NARROW input (the 16 occupancy channels) ---------------
// The parity-class engine (k_gemm_nt_bf16<conv dgrad>) gathers 8 x Cout gradient values per input voxel and pays a
// 128-column tile for Cin = 16 columns: 268 us at B = 16, bound by the 2.1 GB of gathered operand reads.  Turned
// around -- "columns first": T[o][tap * Cin + c] = sum_co dz[o][co] W[co][c][tap] is ONE plain GEMM over the output
// voxels (dz read once: 33 MB), then every input voxel sums its 8 contributions (col2im as a gather: deterministic).
// W2[n = tap * Cin + c][k = co] bf16 from W fp32 [Cout][w_cin][64], channels c_off ..
__global__ void k_scb_pack_cols(const float *__restrict__ W, int Cout, int Cin, int w_cin, int c_off,
                                uint16_t *__restrict__ W2) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t total = (int64_t)64 * Cin * Cout;
  if (i >= total) return;
  const int co = (int)(i % Cout);
  const int c = (int)((i / Cout) % Cin);
  const int tap = (int)(i / ((int64_t)Cout * Cin));
  W2[i] = (uint16_t)mf::bf16_bits(W[((int64_t)co * w_cin + c_off + c) * 64 + tap]);
}

// dx[b][v][c0 .. c0 + 7] = sum over the 8 slots of T[b][o(v, slot)][tap(v, slot) * Cin + c0 ..] (fp32 sum in slot
// order, outputs outside the grid skipped): one lane per (input voxel, 8-channel chunk)
__global__ __launch_bounds__(256) void k_scb_col2im(const uint16_t *__restrict__ T, int B, int D, int Cin,
                                                    uint16_t *__restrict__ dx) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int chunks = Cin / 8;
  const int64_t V = (int64_t)D * D * D, total = (int64_t)B * V * chunks;
  if (i >= total) return;
  const int ch = (int)(i % chunks);
  const int64_t bv = i / chunks;
  const int b = (int)(bv / V), v = (int)(bv % V);
  const int vx = v / (D * D), vy = (v / D) % D, vz = v % D;
  const int Do = D / 2;
  const int64_t Vo = (int64_t)Do * Do * Do, ldt = (int64_t)64 * Cin;
  const int px = (vx + 1) & 1, py = (vy + 1) & 1, pz = (vz + 1) & 1;
  float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
  for (int slot = 0; slot < 8; ++slot) {
    const int ax = slot & 1, ay = (slot >> 1) & 1, az = (slot >> 2) & 1;
    const int ox = ((vx + 1) >> 1) - ax, oy = ((vy + 1) >> 1) - ay, oz = ((vz + 1) >> 1) - az;
    if (ox < 0 || ox >= Do || oy < 0 || oy >= Do || oz < 0 || oz >= Do) continue;
    const int tap = ((px + 2 * ax) * 4 + (py + 2 * ay)) * 4 + (pz + 2 * az);
    const uint4 w = *reinterpret_cast<const uint4 *>(T + ((int64_t)b * Vo + (ox * Do + oy) * Do + oz) * ldt + tap * Cin + 8 * ch);
    acc[0] += mf::bf16_lo(w.x); acc[1] += mf::bf16_hi(w.x); acc[2] += mf::bf16_lo(w.y); acc[3] += mf::bf16_hi(w.y);
    acc[4] += mf::bf16_lo(w.z); acc[5] += mf::bf16_hi(w.z); acc[6] += mf::bf16_lo(w.w); acc[7] += mf::bf16_hi(w.w);
  }
  *reinterpret_cast<uint4 *>(dx + bv * Cin + 8 * ch) =
      make_uint4(mf::pack_bf16x2(acc[0], acc[1]), mf::pack_bf16x2(acc[2], acc[3]), mf::pack_bf16x2(acc[4], acc[5]),
                 mf::pack_bf16x2(acc[6], acc[7]));
}

int bad(const char *msg) {
  mf::set_last_error(hipErrorInvalidValue, msg);
  return -(int)hipErrorInvalidValue;
}

}  // namespace

extern "C" int64_t mf_sparse_conv3_bf16_max_rows(int64_t n_points) { return scb_max_rows(n_points); }

extern "C" int64_t mf_sparse_conv3_bf16_workspace_bytes(int64_t n_points, int32_t B, int32_t D) {
  if (n_points < 0 || B <= 0 || D <= 0 || D % 2) return -1;
  ScbWs w;
  return scb_carve(nullptr, n_points, B, D, &w);
}

/* Device addresses of the tables the GEMM engines read: out[0] = tile_group (class of every 64-row block, -1 =
 * empty), out[1] = class_off (9 padded row offsets = the 8 row ranges), out[2] = rowmap [B * D^3] (compact row of a
 * voxel or -1), out[3] = counts [B * D^3], out[4] = rowvox [max_rows], out[5] = head [B * D^3], out[6] = link [n]
 * (the per-voxel point chains: mf_average_voxelization_rows_bf16_fwd / _bwd read them). */
extern "C" int mf_sparse_conv3_bf16_tables(void *ws, int64_t n_points, int32_t B, int32_t D, int64_t *out) {
  ScbWs w;
  scb_carve(ws, n_points, B, D, &w);
  out[0] = (int64_t)(uintptr_t)w.tile_group;
  out[1] = (int64_t)(uintptr_t)w.class_off;
  out[2] = (int64_t)(uintptr_t)w.rowmap;
  out[3] = (int64_t)(uintptr_t)w.counts;
  out[4] = (int64_t)(uintptr_t)w.rowvox;
  out[5] = (int64_t)(uintptr_t)w.head;
  out[6] = (int64_t)(uintptr_t)w.link;
  return 0;
}

/* points [n,3] in the network's voxel frame (origin 0, pitch 1), batch_indices [n] -> chains, compact class-major
 * padded rows and the engine tables in ``ws``. */
extern "C" int mf_sparse_conv3_bf16_index(const float *points, const int32_t *batch_indices, int64_t n, int32_t B,
                                          int32_t D, void *ws, mfStream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (B <= 0 || D <= 0 || D % 2) return bad("sparse_conv3_bf16_index: B > 0, even D");
  ScbWs w;
  scb_carve(ws, n, B, D, &w);
  const int64_t BV = (int64_t)B * D * D * D;
  if (int e = mf::fill_bytes(w.counts, 0, BV * 4, stream)) return e;
  if (int e = mf::fill_bytes(w.head, 0xff, BV * 4, stream)) return e;
  if (int e = mf::fill_bytes(w.rowvox, 0xff, w.max_rows * 4, stream)) return e;
  if (n > 0)
    hipLaunchKernelGGL(k_scb_link, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, points, batch_indices, n, B,
                       D, w);
  const unsigned blocks = (unsigned)scb_blocks(BV);
  hipLaunchKernelGGL(k_scb_count, dim3(blocks), dim3(256), 0, stream, w, B, D);
  hipLaunchKernelGGL(k_scb_offsets, dim3(1), dim3(256), 0, stream, w, (int)blocks);
  hipLaunchKernelGGL(k_scb_assign, dim3(blocks), dim3(256), 0, stream, w, B, D);
  return mf::check_launch("mf_sparse_conv3_bf16_index");
}

extern "C" int mf_sparse_conv3_bf16_pack(const float *W, int32_t Cout, int32_t Cs, int32_t w_cin, int32_t c_off,
                                         void *Wp, void *Wq, mfStream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (Cout <= 0 || Cs <= 0 || Cs % 8 || Cout % 8 || c_off < 0 || c_off + Cs > w_cin)
    return bad("sparse_conv3_bf16_pack: Cs % 8 == 0, Cout % 8 == 0, c_off + Cs <= w_cin");
  const int64_t total = (int64_t)64 * Cout * Cs;
  hipLaunchKernelGGL(k_scb_pack, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, W, Cout, Cs, w_cin, c_off,
                     (uint16_t *)Wp, (uint16_t *)Wq);
  return mf::check_launch("mf_sparse_conv3_bf16_pack");
}

extern "C" int mf_sparse_conv3_bf16_unpack_dw(const float *dWp, int32_t Cout, int32_t Cs, int32_t w_cin, int32_t c_off,
                                              float *dW, mfStream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (Cout <= 0 || Cs <= 0 || c_off < 0 || c_off + Cs > w_cin) return bad("sparse_conv3_bf16_unpack_dw: c_off + Cs <= w_cin");
  const int64_t total = (int64_t)64 * Cout * Cs;
  hipLaunchKernelGGL(k_scb_unpack_dw, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, dWp, Cout, Cs, w_cin,
                     c_off, dW);
  return mf::check_launch("mf_sparse_conv3_bf16_unpack_dw");
}

/* out [B][(D/2)^3][Cout] bf16 = relu?(dense + bias + the sparse contributions C [max_rows][8 * Cout] bf16). */
extern "C" int mf_sparse_conv3_bf16_reduce(const void *C, const float *dense, const float *bias, void *ws,
                                           int64_t n_points, int32_t B, int32_t D, int32_t Cout, int32_t relu, void *out,
                                           mfStream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (B <= 0) return 0;
  if (Cout % 256) return bad("sparse_conv3_bf16_reduce: Cout % 256 == 0");
  ScbWs w;
  scb_carve(ws, n_points, B, D, &w);
  const int Vo = (D / 2) * (D / 2) * (D / 2);
  hipLaunchKernelGGL(k_scb_reduce, dim3((Vo + 63) / 64, B), dim3(kRedThreads), 0, stream, (const uint16_t *)C, dense, bias,
                     w.rowmap, D, Cout, relu, (uint16_t *)out);
  return mf::check_launch("mf_sparse_conv3_bf16_reduce");
}

/* dYg [max_rows][8 * Cout] bf16 from dz [B][(D/2)^3][Cout] bf16 (every row of the padded range is written). */
extern "C" int mf_sparse_conv3_bf16_gather_dy(const void *dz, void *ws, int64_t n_points, int32_t B, int32_t D,
                                              int32_t Cout, void *dYg, mfStream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (B <= 0) return 0;
  if (Cout % 8) return bad("sparse_conv3_bf16_gather_dy: Cout % 8 == 0");
  ScbWs w;
  scb_carve(ws, n_points, B, D, &w);
  const int64_t items = w.max_rows * 8;
  hipLaunchKernelGGL(k_scb_gather_dy, dim3((unsigned)((items + 3) / 4)), dim3(256), 0, stream, (const uint16_t *)dz, w.rowvox,
                     w.class_off, D, Cout, (uint16_t *)dYg);
  return mf::check_launch("mf_sparse_conv3_bf16_gather_dy");
}

/* Narrow-input data gradient of Convolution3D(.., 4, 2, pad=1), "columns first" (see k_scb_pack_cols):
 *   mf_conv3d_k4s2_bf16_pack_cols   W fp32 [Cout, w_cin, 4,4,4] channels c_off .. c_off + Cin -> W2 bf16 [64 Cin][Cout]
 *   T [B (D/2)^3][64 Cin] = dz W2^T is a plain mf_linear_bf16 call (the caller's)
 *   mf_conv3d_k4s2_bf16_col2im      dx [B, D^3, Cin] bf16 = the 8 contributions of every input voxel, summed */
extern "C" int mf_conv3d_k4s2_bf16_pack_cols(const float *W, int32_t Cout, int32_t Cin, int32_t w_cin, int32_t c_off,
                                             void *W2, mfStream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (Cout <= 0 || Cin <= 0 || Cout % 8 || c_off < 0 || c_off + Cin > w_cin) return bad("conv3d_k4s2_bf16_pack_cols: Cout % 8 == 0, c_off + Cin <= w_cin");
  const int64_t total = (int64_t)64 * Cin * Cout;
  hipLaunchKernelGGL(k_scb_pack_cols, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, W, Cout, Cin, w_cin, c_off,
                     (uint16_t *)W2);
  return mf::check_launch("mf_conv3d_k4s2_bf16_pack_cols");
}

extern "C" int mf_conv3d_k4s2_bf16_col2im(const void *T, int32_t B, int32_t D, int32_t Cin, void *dx, mfStream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (B <= 0) return 0;
  if (Cin % 8 || D % 2) return bad("conv3d_k4s2_bf16_col2im: Cin % 8 == 0, even D");
  const int64_t total = (int64_t)B * D * D * D * (Cin / 8);
  hipLaunchKernelGGL(k_scb_col2im, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, (const uint16_t *)T, B, D, Cin,
                     (uint16_t *)dx);
  return mf::check_launch("mf_conv3d_k4s2_bf16_col2im");
}
