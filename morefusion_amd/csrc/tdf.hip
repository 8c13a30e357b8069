// truncated_distance_function forward / backward, pseudo-occupancy weights (gfx950).
//
// Reference: morefusion/functions/geometry/truncated_distance_function.py:21-103
// (K7: one thread per (point, kernel offset); float atomicMin on a global grid, then
// a racy atomicExch of the candidate id), :105-166 (K8), :181-213 (weights).
//
// MI355X design: a workgroup owns a tile of the voxel grid in LDS as two 32-bit words per
// voxel (distance bits, candidate id).  Distances are >= 0, so unsigned integer order ==
// float order: pass 0 takes a 32-bit atomicMin of the distance bits, pass 1 a 32-bit
// atomicMin of the flat id among the candidates that equal the minimum -- exact minimum AND
// deterministic arg-min (lowest flat id among equal distances); the reference's
// atomicMin + atomicExch pair can record a non-minimal writer.  (A packed 64-bit key with one
// ds_min_u64 was the first version: 64-bit LDS atomics measured ~10x slower than 32-bit ones.)
// Tiles are x-slabs (optionally split in y) of <= 64 KB, so two workgroups share a CU's 160 KB
// LDS; every workgroup streams the whole point list (12 B/point, L2-resident) and keeps only
// candidates that land in its tile.  No global atomics, no pre-filled global grids, one
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

// Kernel offsets follow numpy.meshgrid's default 'xy' indexing used at
// truncated_distance_function.py:39-41: flat k = (a*ks + b)*ks + c  ->  (b, a, c) - ks/2.
template <int KS>
__global__ __launch_bounds__(kTdfThreads) void k_tdf_fwd(const float *__restrict__ points,
                                                         int64_t P, float pitch, float ox,
                                                         float oy, float oz, int X, int Y, int Z,
                                                         float trunc, int ks_rt, int SX, int SY,
                                                         float *__restrict__ tdf,
                                                         int32_t *__restrict__ flat) {
  MF_DYN_LDS(uint32_t, s_w);  // dist bits [nvox], id [nvox]
  const int ks = KS > 0 ? KS : ks_rt;
  const int h = ks / 2, K = ks * ks * ks;
  const int x0 = blockIdx.x * SX, y0 = blockIdx.y * SY;
  const int sx = min(SX, X - x0), sy = min(SY, Y - y0);
  const int nvox = sx * sy * Z;
  uint32_t *s_dist = s_w, *s_id = s_w + SX * SY * Z;
  const uint32_t tbits = __float_as_uint(trunc);
  for (int i = threadIdx.x; i < nvox; i += kTdfThreads) { s_dist[i] = tbits; s_id[i] = 0xffffffffu; }
  __syncthreads();
  const float fh = (float)h;
  // pass 0: minimum distance per voxel (32-bit atomicMin on the float bits; distances >= 0);
  // pass 1: lowest flat id among the candidates whose distance equals that minimum.
  for (int pass = 0; pass < 2; ++pass) {
    for (int64_t p = threadIdx.x; p < P; p += kTdfThreads) {
      const float fx = (points[3 * p] - ox) / pitch;
      const float fy = (points[3 * p + 1] - oy) / pitch;
      const float fz = (points[3 * p + 2] - oz) / pitch;
      const float rx = roundf(fx), ry = roundf(fy), rz = roundf(fz);
      // neighbourhood vs tile (false for NaN)
      if (!(rx + fh >= (float)x0 && rx - fh < (float)(x0 + sx) && ry + fh >= (float)y0 &&
            ry - fh < (float)(y0 + sy) && rz + fh >= 0.0f && rz - fh < (float)Z))
        continue;
      const int irx = (int)rx, iry = (int)ry, irz = (int)rz;
#pragma unroll
      for (int a = 0; a < ks; ++a) {
        const int iy = iry + a - h;
        if (iy < y0 || iy >= y0 + sy) continue;
        const float dy = fy - (float)iy;
#pragma unroll
        for (int b = 0; b < ks; ++b) {
          const int ix = irx + b - h;
          if (ix < x0 || ix >= x0 + sx) continue;
          const float dx = fx - (float)ix;
          const float dxy = dx * dx + dy * dy;
#pragma unroll
          for (int c = 0; c < ks; ++c) {
            const int iz = irz + c - h;
            if (iz < 0 || iz >= Z) continue;
            const float dz = fz - (float)iz;
            const float dist = pitch * sqrtf(dxy + dz * dz);
            if (dist < trunc) {
              const int li = ((ix - x0) * sy + (iy - y0)) * Z + iz;
              const uint32_t db = __float_as_uint(dist);
              if (pass == 0) {
                if (db < s_dist[li]) atomicMin(&s_dist[li], db);
              } else if (db == s_dist[li]) {
                atomicMin(&s_id[li], (uint32_t)(p * K + (a * ks + b) * ks + c));
              }
            }
          }
        }
      }
    }
    __syncthreads();
  }
  for (int i = threadIdx.x; i < nvox; i += kTdfThreads) {
    const int iz = i % Z, iy = (i / Z) % sy, ix = i / (Z * sy);
    const int64_t g = ((int64_t)(x0 + ix) * Y + (y0 + iy)) * Z + iz;
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
  // tile: <= 8192 voxels (64 KB of keys); split x first, then y.
  const int cap = 8192;
  int SX, SY;
  if (Z > cap) {
    mf::set_last_error(hipErrorInvalidValue, "tdf: Z dimension too large for an LDS tile");
    return -(int)hipErrorInvalidValue;
  }
  if ((int64_t)Y * Z <= cap) {
    SY = Y;
    SX = std::max(1, std::min(X, cap / (Y * Z)));
    // keep at least ~8 workgroups when the grid allows it
    while (SX > 1 && (X + SX - 1) / SX < 8) SX = (SX + 1) / 2;
  } else {
    SX = 1;
    SY = std::max(1, cap / Z);
  }
  dim3 grid((X + SX - 1) / SX, (Y + SY - 1) / SY);
  const size_t lds = (size_t)SX * SY * Z * 2 * sizeof(uint32_t);
  if (ks == 3)
    hipLaunchKernelGGL(k_tdf_fwd<3>, grid, dim3(kTdfThreads), lds, stream, points, P, pitch, ox,
                       oy, oz, X, Y, Z, truncation, ks, SX, SY, tdf, flat);
  else
    hipLaunchKernelGGL(k_tdf_fwd<0>, grid, dim3(kTdfThreads), lds, stream, points, P, pitch, ox,
                       oy, oz, X, Y, Z, truncation, ks, SX, SY, tdf, flat);
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
