// truncated_distance_function forward / backward, pseudo-occupancy weights (gfx950).
//
// Reference: morefusion/functions/geometry/truncated_distance_function.py:21-103
// (K7: one thread per (point, kernel offset); float atomicMin on a global grid, then
// a racy atomicExch of the candidate id), :105-166 (K8), :181-213 (weights).
//
// MI355X design (round 4): a workgroup owns ONE x-plane x a y-range of the voxel grid in LDS as two 32-bit words
// per voxel (distance bits, candidate id).  Distances are >= 0, so unsigned integer order == float order: pass 0
// takes a 32-bit atomicMin of the distance bits, pass 1 a 32-bit atomicMin of the flat id among the candidates that
// equal the minimum -- exact minimum AND deterministic arg-min (lowest flat id among equal distances); the
// reference's atomicMin + atomicExch pair can record a non-minimal writer.  (A packed 64-bit key with one ds_min_u64
// was the first version: 64-bit LDS atomics measured ~10x slower than 32-bit ones.)
// Every workgroup streams the whole point list (12 B/point, L2-resident) and keeps the points whose kernel
// neighbourhood meets its plane: 3 of 32 planes for the default kernel size, i.e. ~10 % of the points reach the
// candidate loop, and only with the ONE kernel offset b that lands on the plane.  Round 1-3 used 8 x-slabs of 4
// planes (8 workgroups on a 256-CU chip, each lane walking ~12 points x 27 candidates twice: measured 148 us for
// 3000 points into 32^3); plane x y-range tiles give 128-256 workgroups whose lanes see ~1 in-range point per pass.
// Same float expressions in the same order -> the same bits.  No global atomics, no pre-filled global grids, one
// coalesced write.
#include <algorithm>

#include "mf_common.h"

namespace mf {

__device__ __forceinline__ int tdf_ksize(float pitch, float trunc) {
  int ks = (int)ceilf(trunc / pitch);
  return (ks & 1) ? ks : ks + 1;
}

}  // namespace mf

namespace {

constexpr int kTdfThreads = 256;
constexpr int kTdfChunk = 4096;  // points per filter round = capacity of the in-range list (64 KB of LDS)

// Kernel offsets follow numpy.meshgrid's default 'xy' indexing used at
// truncated_distance_function.py:39-41: flat k = (a*ks + b)*ks + c  ->  (b, a, c) - ks/2.
//
// Two phases per chunk of 4096 points.  FILTER: a lane takes 16 points (all loads issued before the first use),
// one divide + round decides whether the point's kernel neighbourhood meets this x-plane (3 planes of 32 for the
// default kernel), the survivors' other two coordinates decide the y-range; survivors go to an LDS list
// {fx, fy, fz, point id} -- ~50 of 3000 points for a 4-row tile.  SCAN: the lanes run over (list entry, a, c)
// triples, i.e. over the CANDIDATE voxels (b is fixed by the plane), so the work is balanced whatever the points'
// order and the code is a short rolled loop (round 4, second version: 16 points x 9 candidates x 2 passes fully
// unrolled per lane -- a 90 KB instruction stream, 47 us).
template <int KS>
__global__ __launch_bounds__(kTdfThreads) void k_tdf_fwd(const float *__restrict__ points,
                                                         int64_t P, float pitch, float ox,
                                                         float oy, float oz, int X, int Y, int Z,
                                                         float trunc, int ks_rt, int SY,
                                                         float *__restrict__ tdf,
                                                         int32_t *__restrict__ flat) {
  MF_DYN_LDS(uint32_t, s_w);  // dist bits [SY * Z], id [SY * Z], list [kTdfChunk] x float4
  __shared__ int s_n;
  const int ks = KS > 0 ? KS : ks_rt;
  const int h = ks / 2, K = ks * ks * ks;
  const int x0 = blockIdx.x, y0 = blockIdx.y * SY;
  const int sy = min(SY, Y - y0);
  const int nvox = sy * Z;
  uint32_t *s_dist = s_w, *s_id = s_w + SY * Z;
  float4 *s_list = reinterpret_cast<float4 *>(s_w + 2 * ((SY * Z + 3) & ~3));
  const uint32_t tbits = __float_as_uint(trunc);
  for (int i = threadIdx.x; i < nvox; i += kTdfThreads) { s_dist[i] = tbits; s_id[i] = 0xffffffffu; }
  const float fh = (float)h;
  auto filter = [&](int64_t base) {
    constexpr int R = kTdfChunk / kTdfThreads;
    float px[R], py[R], pz[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const int64_t p = base + (int64_t)r * kTdfThreads + threadIdx.x;
      const int64_t q = p < P ? p : 0;
      px[r] = p < P ? points[3 * q] : __uint_as_float(0x7fc00000u);
      py[r] = points[3 * q + 1];
      pz[r] = points[3 * q + 2];
    }
#pragma unroll 1
    for (int r = 0; r < R; ++r) {
      const float fx = (px[r] - ox) / pitch;
      const float rx = roundf(fx);
      if (!(rx + fh >= (float)x0 && rx - fh < (float)(x0 + 1))) continue;  // (false for NaN)
      const float fy = (py[r] - oy) / pitch, fz = (pz[r] - oz) / pitch;
      const float ry = roundf(fy), rz = roundf(fz);
      if (!(ry + fh >= (float)y0 && ry - fh < (float)(y0 + sy) && rz + fh >= 0.0f && rz - fh < (float)Z)) continue;
      const int slot = atomicAdd(&s_n, 1);  // < kTdfChunk: a chunk holds that many points
      s_list[slot] = make_float4(fx, fy, fz, __int_as_float((int)(r * kTdfThreads + threadIdx.x)));
    }
  };
  // pass 0: minimum distance per voxel (32-bit atomicMin on the float bits; distances >= 0);
  // pass 1: lowest flat id among the candidates whose distance equals that minimum.
  auto scan = [&](int pass, int64_t base) {
    const int n = s_n * ks * ks;
    for (int w = threadIdx.x; w < n; w += kTdfThreads) {
      const int e = w / (ks * ks), ac = w - e * ks * ks;
      const int a = ac / ks, c = ac - a * ks;
      const float4 rec = s_list[e];
      const int irx = (int)roundf(rec.x), iry = (int)roundf(rec.y), irz = (int)roundf(rec.z);
      const int b = x0 - irx + h;  // the one x offset of the kernel that lands on this plane: 0 <= b < ks
      const int iy = iry + a - h, iz = irz + c - h;
      if (iy < y0 || iy >= y0 + sy || iz < 0 || iz >= Z) continue;
      const float dx = rec.x - (float)x0, dy = rec.y - (float)iy, dz = rec.z - (float)iz;
      const float dxy = dx * dx + dy * dy;
      const float dist = pitch * sqrtf(dxy + dz * dz);
      if (dist < trunc) {
        const int li = (iy - y0) * Z + iz;
        const uint32_t db = __float_as_uint(dist);
        if (pass == 0) {
          if (db < s_dist[li]) atomicMin(&s_dist[li], db);
        } else if (db == s_dist[li]) {
          const int64_t p = base + __float_as_int(rec.w);
          atomicMin(&s_id[li], (uint32_t)(p * K + (a * ks + b) * ks + c));
        }
      }
    }
  };
  if (threadIdx.x == 0) s_n = 0;
  __syncthreads();
  if (P <= kTdfChunk) {  // the usual case: one filter round serves both passes
    if (P > 0) filter(0);  // (an empty point list may come with a null pointer)
    __syncthreads();
    scan(0, 0);
    __syncthreads();
    scan(1, 0);
  } else {
    for (int pass = 0; pass < 2; ++pass)
      for (int64_t base = 0; base < P; base += kTdfChunk) {
        filter(base);
        __syncthreads();
        scan(pass, base);
        __syncthreads();
        if (threadIdx.x == 0) s_n = 0;
        __syncthreads();
      }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < nvox; i += kTdfThreads) {
    const int64_t g = ((int64_t)x0 * Y + y0) * Z + i;  // (iy - y0) * Z + iz == i: the tile is contiguous in the grid
    tdf[g] = __uint_as_float(s_dist[i]);
    const uint32_t lo = s_id[i];
    flat[g] = lo == 0xffffffffu ? -1 : (int32_t)lo;
  }
}

// One thread per voxel (truncated_distance_function.py:121-146).  The voxel IS
// round(p_f)+kernel[k], so k is not needed to rebuild the unit vector.
__global__ __launch_bounds__(256) void k_tdf_bwd(const float *__restrict__ gtdf,
                                                 const float *__restrict__ points,
                                                 const int32_t *__restrict__ flat, float pitch,
                                                 float ox, float oy, float oz, int X, int Y,
                                                 int Z, int K, float *__restrict__ gpoints) {
  const int v = blockIdx.x * blockDim.x + threadIdx.x;
  if (v >= X * Y * Z) return;
  const int f = flat[v];
  if (f < 0) return;
  const int p = f / K;
  const int iz = v % Z, iy = (v / Z) % Y, ix = v / (Z * Y);
  const float dx = (points[3 * p] - ox) / pitch - (float)ix;
  const float dy = (points[3 * p + 1] - oy) / pitch - (float)iy;
  const float dz = (points[3 * p + 2] - oz) / pitch - (float)iz;
  const float n = sqrtf((dx * dx + dy * dy) + dz * dz);
  if (n > 0.0f) {
    const float g = gtdf[v];
    atomicAdd(&gpoints[3 * p], dx / n * g);
    atomicAdd(&gpoints[3 * p + 1], dy / n * g);
    atomicAdd(&gpoints[3 * p + 2], dz / n * g);
  }
}

// truncated_distance_function.py:198-204, pass 1: raw inside weight + its maximum.
__global__ __launch_bounds__(256) void k_pocc_wraw(const int32_t *__restrict__ flat,
                                                   const float *__restrict__ sdf, int V, int K,
                                                   float sdf_offset, float *__restrict__ win,
                                                   float *__restrict__ wsurf,
                                                   uint32_t *__restrict__ wmax_bits) {
  const int v = blockIdx.x * blockDim.x + threadIdx.x;
  float w = 0.0f;
  if (v < V) {
    const int f = flat[v];
    w = (f >= 0 ? sdf[f / K] : -1.0f) + sdf_offset;
    const bool neg = w < 0.0f;
    if (neg) w = 0.0f;
    win[v] = w;
    wsurf[v] = neg ? 0.0f : 1.0f;  // marker, finished in pass 2
  }
  float m = mf::wave_max(w);
  if ((threadIdx.x & 63) == 0) atomicMax(wmax_bits, __float_as_uint(m));  // w >= 0
}

// :204-213, pass 2: normalise, surface weight, the three weighted grids.
__global__ __launch_bounds__(256) void k_pocc_grids(const float *__restrict__ tdf, int V,
                                                    float trunc,
                                                    const uint32_t *__restrict__ wmax_bits,
                                                    float *__restrict__ win,
                                                    float *__restrict__ wsurf,
                                                    float *__restrict__ grids) {
  const int v = blockIdx.x * blockDim.x + threadIdx.x;
  if (v >= V) return;
  const float M = __uint_as_float(*wmax_bits);
  const float wi = win[v] / M;  // 0/0 -> NaN exactly like the reference
  const float ws = wsurf[v] != 0.0f ? 1.0f - wi : wi;
  const float g = 1.0f - tdf[v] / trunc;
  win[v] = wi;
  wsurf[v] = ws;
  grids[v] = g;
  grids[V + v] = g * ws;
  grids[2 * V + v] = g * wi;
}

}  // namespace

extern "C" int mf_truncated_distance_function_fwd(const float *points, int64_t P, float pitch,
                                                  float ox, float oy, float oz, int X, int Y,
                                                  int Z, float truncation, float *tdf,
                                                  int32_t *flat, mfStream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if ((int64_t)X * Y * Z == 0) return 0;
  int ks = (int)ceilf(truncation / pitch);
  if (ks % 2 == 0) ks += 1;
  if ((double)P * ks * ks * ks >= 4294967295.0) {
    mf::set_last_error(hipErrorInvalidValue, "tdf: P*K exceeds 32-bit candidate ids");
    return -(int)hipErrorInvalidValue;
  }
  // tile = one x-plane x SY rows: <= 4096 voxels (32 KB of keys + the 64 KB list), split further until ~256 workgroups exist
  const int cap = 4096;
  if (Z > cap) {
    mf::set_last_error(hipErrorInvalidValue, "tdf: Z dimension too large for an LDS tile");
    return -(int)hipErrorInvalidValue;
  }
  int SY = std::max(1, std::min(Y, cap / Z));
  while (SY > 4 && (int64_t)X * ((Y + SY - 1) / SY) < 256) SY = (SY + 1) / 2;
  dim3 grid(X, (Y + SY - 1) / SY);
  const size_t lds = (size_t)((SY * Z + 3) & ~3) * 2 * sizeof(uint32_t) + (size_t)kTdfChunk * 16;
  if (int e = mf::allow_big_lds(ks == 3 ? (const void *)k_tdf_fwd<3> : (const void *)k_tdf_fwd<0>, (int)lds)) return e;
  if (ks == 3)
    hipLaunchKernelGGL(k_tdf_fwd<3>, grid, dim3(kTdfThreads), lds, stream, points, P, pitch, ox,
                       oy, oz, X, Y, Z, truncation, ks, SY, tdf, flat);
  else
    hipLaunchKernelGGL(k_tdf_fwd<0>, grid, dim3(kTdfThreads), lds, stream, points, P, pitch, ox,
                       oy, oz, X, Y, Z, truncation, ks, SY, tdf, flat);
  return mf::check_launch("mf_truncated_distance_function_fwd");
}

extern "C" int mf_truncated_distance_function_bwd(const float *gtdf, const float *points,
                                                  const int32_t *flat, int64_t P, float pitch,
                                                  float ox, float oy, float oz, int X, int Y,
                                                  int Z, float truncation, float *gpoints,
                                                  mfStream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  const int V = X * Y * Z;
  if (V == 0 || P == 0) return 0;
  int ks = (int)ceilf(truncation / pitch);
  if (ks % 2 == 0) ks += 1;
  hipLaunchKernelGGL(k_tdf_bwd, dim3((V + 255) / 256), dim3(256), 0, stream, gtdf, points, flat,
                     pitch, ox, oy, oz, X, Y, Z, ks * ks * ks, gpoints);
  return mf::check_launch("mf_truncated_distance_function_bwd");
}

extern "C" int mf_pseudo_occupancy_weights(const float *tdf, const int32_t *flat,
                                           const float *sdf, int X, int Y, int Z, int K,
                                           float truncation, float sdf_offset, float *grids,
                                           float *wsurf, float *win, float *wmax,
                                           mfStream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  const int V = X * Y * Z;
  if (V == 0) return 0;
  if (int e_ = mf::fill_bytes(wmax, 0, sizeof(float), stream)) return e_;
  hipLaunchKernelGGL(k_pocc_wraw, dim3((V + 255) / 256), dim3(256), 0, stream, flat, sdf, V, K,
                     sdf_offset, win, wsurf, (uint32_t *)wmax);
  hipLaunchKernelGGL(k_pocc_grids, dim3((V + 255) / 256), dim3(256), 0, stream, tdf, V, truncation,
                     (const uint32_t *)wmax, win, wsurf, grids);
  return mf::check_launch("mf_pseudo_occupancy_weights");
}
