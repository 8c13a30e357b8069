// interpolate_voxel_grid forward / backward for gfx950.
//
// Reference: morefusion/functions/geometry/interpolate_voxel_grid.py:159-214 (K5),
// :216-268 (K6): one thread per (point, channel), 8 corner gathers each from a
// channel-major [B,C,X,Y,Z] grid -> every corner read of neighbouring threads is
// X*Y*Z*4 bytes apart (uncoalesced), and the backward is 8 float atomics per thread
// into a zero-filled global tensor.
//
// MI355X design: a workgroup owns (batch item b, chunk of `cpw` channels) and stages
// that chunk of the grid in LDS with coalesced 16 B loads (the grid is read from HBM
// exactly once); the points of item b then gather their 8 corners from LDS.
// Backward mirrors it: ds_add_f32 into an LDS-resident chunk, one coalesced write-out
// (gvox is written completely -- no memset, no global atomics).
// 16^3 x 4 ch = 64 KB, 8^3 x 32 ch = 64 KB -> two workgroups per CU (160 KB LDS).
#include <algorithm>

#include "mf_common.h"

namespace {

constexpr int kInterpThreads = 256;
constexpr int kInterpPre = 4;  // points per lane loaded ahead of the grid staging (row-range mode)

struct Corner {
  int off[8];
  float w[8];
};

// Weights/offsets in the reference's order w000,w100,w010,w001,w110,w011,w101,w111
// (interpolate_voxel_grid.py:27-58); low = (int)coord (trunc toward zero, :11-13);
// out-of-grid corners get offset -1.
__device__ __forceinline__ void corners(float px, float py, float pz, int X, int Y, int Z,
                                        Corner &k) {
  int lx = (int)px, ly = (int)py, lz = (int)pz;
  float fx = px - (float)lx, fy = py - (float)ly, fz = pz - (float)lz;
  float hx = 1.0f - fx, hy = 1.0f - fy, hz = 1.0f - fz;
  const int dx[8] = {0, 1, 0, 0, 1, 0, 1, 1};
  const int dy[8] = {0, 0, 1, 0, 1, 1, 0, 1};
  const int dz[8] = {0, 0, 0, 1, 0, 1, 1, 1};
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    int ix = lx + dx[j], iy = ly + dy[j], iz = lz + dz[j];
    bool ok = ix >= 0 && ix < X && iy >= 0 && iy < Y && iz >= 0 && iz < Z;
    k.off[j] = ok ? (ix * Y + iy) * Z + iz : -1;
    k.w[j] = ((dx[j] ? fx : hx) * (dy[j] ? fy : hy)) * (dz[j] ? fz : hz);
  }
}

// Finite coordinates far outside the grid would overflow (int)coord; clamp the test.
__device__ __forceinline__ bool plausible(float px, float py, float pz) {
  const float L = 1.0e9f;
  return fabsf(px) < L && fabsf(py) < L && fabsf(pz) < L;  // false for NaN too
}

// Candidate rows of batch item b.  With ``batch_start`` (B+1 row offsets: the rows of item b
// are [batch_start[b], batch_start[b+1]), what the pose network passes) a workgroup visits
// only its own rows; without it every workgroup filters all n rows by batch_indices, with
// kScanU index loads in flight per lane (round 1 walked them one dependent load at a time:
// 15 us of the 31 us kernel at B = 8).  Rows that belong to no item (index outside [0, B) /
// outside every range) are owned by the workgroups of item 0, which write zeros for them.
constexpr int kScanU = 8;

// Precondition of ``batch_start``: points sorted by batch item and offsets consistent with
// batch_indices (the wrapper documents it).  Offsets are clamped to [0, n] so that a stale or
// inconsistent table can give wrong VALUES but never an out-of-bounds read or write.
__device__ __forceinline__ int64_t row_clamp(int32_t r, int64_t n) {
  return r < 0 ? 0 : (r > n ? n : (int64_t)r);
}

template <class F>
__device__ __forceinline__ void for_rows_of_item(const int32_t *__restrict__ batch_indices,
                                                 const int32_t *__restrict__ batch_start,
                                                 int64_t n, int b, int B, F &&visit) {
  if (batch_start) {
    const int64_t p0 = row_clamp(batch_start[b], n), p1 = row_clamp(batch_start[b + 1], n);
    for (int64_t p = p0 + threadIdx.x; p < p1; p += kInterpThreads) visit(p, true);
    if (b == 0) {
      const int64_t lo = row_clamp(batch_start[0], n), hi = row_clamp(batch_start[B], n);
      for (int64_t p = threadIdx.x; p < n; p += kInterpThreads)
        if (p < lo || p >= hi) visit(p, false);
    }
    return;
  }
  for (int64_t base = 0; base < n; base += (int64_t)kInterpThreads * kScanU) {
    int bi[kScanU];
#pragma unroll
    for (int u = 0; u < kScanU; ++u) {
      const int64_t p = base + (int64_t)u * kInterpThreads + threadIdx.x;
      bi[u] = p < n ? batch_indices[p] : -1;
    }
#pragma unroll
    for (int u = 0; u < kScanU; ++u) {
      const int64_t p = base + (int64_t)u * kInterpThreads + threadIdx.x;
      if (p >= n) continue;
      if (bi[u] == b)
        visit(p, true);
      else if (b == 0 && (bi[u] < 0 || bi[u] >= B))
        visit(p, false);
    }
  }
}

// coalesced copy global -> LDS (or back), kStageU 16 B loads in flight per lane
constexpr int kStageU = 8;

__device__ __forceinline__ void stage_copy(float *__restrict__ dst, const float *__restrict__ src,
                                           int total) {
  // 16-byte path only when both ends are 16-byte aligned (odd V with a channel offset is not)
  if ((total & 3) == 0 && (((uintptr_t)src | (uintptr_t)dst) & 15) == 0) {
    const float4 *s4 = reinterpret_cast<const float4 *>(src);
    float4 *d4 = reinterpret_cast<float4 *>(dst);
    const int n4 = total / 4;
    for (int i0 = threadIdx.x; i0 < n4; i0 += kInterpThreads * kStageU) {
      float4 v[kStageU];
#pragma unroll
      for (int u = 0; u < kStageU; ++u) {
        const int i = i0 + u * kInterpThreads;
        if (i < n4) v[u] = s4[i];
      }
#pragma unroll
      for (int u = 0; u < kStageU; ++u) {
        const int i = i0 + u * kInterpThreads;
        if (i < n4) d4[i] = v[u];
      }
    }
  } else {
    for (int i = threadIdx.x; i < total; i += kInterpThreads) dst[i] = src[i];
  }
}

// grid: x = channel chunk, y = batch item.  LDS: cpw * V floats.
__global__ __launch_bounds__(kInterpThreads) void k_interp_fwd_lds(
    const float *__restrict__ vox, const float *__restrict__ points,
    const int32_t *__restrict__ batch_indices, const int32_t *__restrict__ batch_start, int64_t n,
    int B, int C, int X, int Y, int Z, int cpw, float *__restrict__ values, int channels_first) {
  MF_DYN_LDS(float, s_grid);
  const int b = blockIdx.y;
  const int c0 = blockIdx.x * cpw;
  const int nc = min(cpw, C - c0);
  const int V = X * Y * Z;
  // one point: corners once, then nc channels of 8 LDS gathers each
  auto sample = [&](int64_t p, bool mine, float px, float py, float pz) {
    Corner k;
    bool ok = mine && plausible(px, py, pz);
    if (ok) corners(px, py, pz, X, Y, Z, k);
    if (!ok) {
#pragma unroll
      for (int j = 0; j < 8; ++j) { k.off[j] = -1; k.w[j] = 0.0f; }
    }
    for (int c = 0; c < nc; ++c) {
      const float *g = s_grid + c * V;
      float acc = 0.0f;
#pragma unroll
      for (int j = 0; j < 8; ++j)
        if (k.off[j] >= 0) acc += k.w[j] * g[k.off[j]];
      if (channels_first)
        values[(int64_t)(c0 + c) * n + p] = acc;
      else
        values[p * C + c0 + c] = acc;
    }
  };
  // Row ranges known and the item's rows fit kInterpPre per lane (the pose network: 1000 rows,
  // 4 per lane): the point loads are issued BEFORE the grid chunk is staged, so their latency
  // hides behind the 64 KB copy instead of costing one dependent round trip per loop iteration.
  if (batch_start) {
    const int64_t p0 = row_clamp(batch_start[b], n), p1 = row_clamp(batch_start[b + 1], n);
    if (p1 - p0 <= (int64_t)kInterpThreads * kInterpPre) {
      float px[kInterpPre], py[kInterpPre], pz[kInterpPre];
#pragma unroll
      for (int u = 0; u < kInterpPre; ++u) {
        const int64_t p = p0 + (int64_t)u * kInterpThreads + threadIdx.x;
        px[u] = py[u] = pz[u] = 0.0f;
        if (p < p1) { px[u] = points[3 * p]; py[u] = points[3 * p + 1]; pz[u] = points[3 * p + 2]; }
      }
      stage_copy(s_grid, vox + ((int64_t)b * C + c0) * V, nc * V);
      __syncthreads();
#pragma unroll
      for (int u = 0; u < kInterpPre; ++u) {
        const int64_t p = p0 + (int64_t)u * kInterpThreads + threadIdx.x;
        if (p < p1) sample(p, true, px[u], py[u], pz[u]);
      }
      if (b == 0) {  // rows outside every item: zeros (same contract as the scan path)
        const int64_t lo = row_clamp(batch_start[0], n), hi = row_clamp(batch_start[B], n);
        for (int64_t p = threadIdx.x; p < n; p += kInterpThreads)
          if (p < lo || p >= hi) sample(p, false, 0.0f, 0.0f, 0.0f);
      }
      return;
    }
  }
  stage_copy(s_grid, vox + ((int64_t)b * C + c0) * V, nc * V);
  __syncthreads();
  for_rows_of_item(batch_indices, batch_start, n, b, B, [&](int64_t p, bool mine) {
    float px = 0.0f, py = 0.0f, pz = 0.0f;
    if (mine) { px = points[3 * p]; py = points[3 * p + 1]; pz = points[3 * p + 2]; }
    sample(p, mine, px, py, pz);
  });
}

// The same workgroup shape with the chunk stored VOXEL-major in LDS, channels innermost
// ([V][cpw], cpw a multiple of 4): a corner gather is one 16-byte LDS read per 4 channels instead
// of 4 scalar reads at 4 addresses, and the corner offset is computed once for all of them --
// the gather phase (8 x cpw random LDS reads per point, the part that made the kernel LDS-issue
// bound at 0.2 of the HBM roofline) shrinks 4x.  Staging transposes on the way in: a lane takes a
// voxel, loads its cpw channels from the cpw channel planes (each load coalesced across the
// wave) and stores them as 16-byte rows (consecutive lanes -> consecutive rows).
// Same sums in the same order per channel as k_interp_fwd_lds: identical bits.
template <int CPW>
__global__ __launch_bounds__(kInterpThreads) void k_interp_fwd_vm(
    const float *__restrict__ vox, const float *__restrict__ points,
    const int32_t *__restrict__ batch_indices, const int32_t *__restrict__ batch_start, int64_t n,
    int B, int C, int X, int Y, int Z, float *__restrict__ values, int channels_first) {
  MF_DYN_LDS(float, s_grid);
  constexpr int Q = CPW / 4;
  const int b = blockIdx.y;
  const int c0 = blockIdx.x * CPW;
  const int V = X * Y * Z;
  float4 *s4 = reinterpret_cast<float4 *>(s_grid);
  auto sample = [&](int64_t p, bool mine, float px, float py, float pz) {
    Corner k;
    bool ok = mine && plausible(px, py, pz);
    if (ok) corners(px, py, pz, X, Y, Z, k);
    float acc[CPW];
#pragma unroll
    for (int c = 0; c < CPW; ++c) acc[c] = 0.0f;
    if (ok) {
      float4 g[8][Q];
#pragma unroll
      for (int j = 0; j < 8; ++j)
#pragma unroll
        for (int q = 0; q < Q; ++q) g[j][q] = s4[(k.off[j] < 0 ? 0 : k.off[j]) * Q + q];
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        if (k.off[j] < 0) continue;
#pragma unroll
        for (int q = 0; q < Q; ++q) {
          acc[4 * q + 0] += k.w[j] * g[j][q].x;
          acc[4 * q + 1] += k.w[j] * g[j][q].y;
          acc[4 * q + 2] += k.w[j] * g[j][q].z;
          acc[4 * q + 3] += k.w[j] * g[j][q].w;
        }
      }
    }
#pragma unroll
    for (int c = 0; c < CPW; ++c) {
      if (channels_first)
        values[(int64_t)(c0 + c) * n + p] = acc[c];
      else
        values[p * C + c0 + c] = acc[c];
    }
  };
  auto stage = [&]() {
    const float *src = vox + ((int64_t)b * C + c0) * V;
    constexpr int kVox = 16 / CPW > 0 ? 16 / CPW : 1;  // voxels per lane and round: 16 loads in flight
    for (int v0 = threadIdx.x; v0 < V; v0 += kInterpThreads * kVox) {
      float r[kVox][CPW];
#pragma unroll
      for (int u = 0; u < kVox; ++u) {
        const int v = v0 + u * kInterpThreads;
#pragma unroll
        for (int c = 0; c < CPW; ++c) r[u][c] = v < V ? src[(int64_t)c * V + v] : 0.0f;
      }
#pragma unroll
      for (int u = 0; u < kVox; ++u) {
        const int v = v0 + u * kInterpThreads;
        if (v >= V) continue;
#pragma unroll
        for (int q = 0; q < Q; ++q)
          s4[v * Q + q] = make_float4(r[u][4 * q], r[u][4 * q + 1], r[u][4 * q + 2], r[u][4 * q + 3]);
      }
    }
  };
  if (batch_start) {
    const int64_t p0 = row_clamp(batch_start[b], n), p1 = row_clamp(batch_start[b + 1], n);
    if (p1 - p0 <= (int64_t)kInterpThreads * kInterpPre) {
      float px[kInterpPre], py[kInterpPre], pz[kInterpPre];
#pragma unroll
      for (int u = 0; u < kInterpPre; ++u) {
        const int64_t p = p0 + (int64_t)u * kInterpThreads + threadIdx.x;
        px[u] = py[u] = pz[u] = 0.0f;
        if (p < p1) { px[u] = points[3 * p]; py[u] = points[3 * p + 1]; pz[u] = points[3 * p + 2]; }
      }
      stage();
      __syncthreads();
#pragma unroll
      for (int u = 0; u < kInterpPre; ++u) {
        const int64_t p = p0 + (int64_t)u * kInterpThreads + threadIdx.x;
        if (p < p1) sample(p, true, px[u], py[u], pz[u]);
      }
      if (b == 0) {  // rows outside every item: zeros (same contract as the scan path)
        const int64_t lo = row_clamp(batch_start[0], n), hi = row_clamp(batch_start[B], n);
        for (int64_t p = threadIdx.x; p < n; p += kInterpThreads)
          if (p < lo || p >= hi) sample(p, false, 0.0f, 0.0f, 0.0f);
      }
      return;
    }
  }
  stage();
  __syncthreads();
  for_rows_of_item(batch_indices, batch_start, n, b, B, [&](int64_t p, bool mine) {
    float px = 0.0f, py = 0.0f, pz = 0.0f;
    if (mine) { px = points[3 * p]; py = points[3 * p + 1]; pz = points[3 * p + 2]; }
    sample(p, mine, px, py, pz);
  });
}

// Fallback for grids too large for LDS: direct gathers (thread per point x channel chunk).
__global__ __launch_bounds__(kInterpThreads) void k_interp_fwd_direct(
    const float *__restrict__ vox, const float *__restrict__ points,
    const int32_t *__restrict__ batch_indices, int64_t n, int B, int C, int X, int Y, int Z,
    float *__restrict__ values, int channels_first) {
  int64_t p = (int64_t)blockIdx.x * kInterpThreads + threadIdx.x;
  if (p >= n) return;
  const int b = batch_indices[p];
  const int V = X * Y * Z;
  float px = points[3 * p], py = points[3 * p + 1], pz = points[3 * p + 2];
  Corner k;
  bool ok = plausible(px, py, pz) && b >= 0 && b < B;
  if (ok) corners(px, py, pz, X, Y, Z, k);
  for (int c = blockIdx.y; c < C; c += gridDim.y) {
    float acc = 0.0f;
    if (ok) {
      const float *g = vox + ((int64_t)b * C + c) * V;
#pragma unroll
      for (int j = 0; j < 8; ++j)
        if (k.off[j] >= 0) acc += k.w[j] * g[k.off[j]];
    }
    if (channels_first)
      values[(int64_t)c * n + p] = acc;
    else
      values[p * C + c] = acc;
  }
}

__global__ __launch_bounds__(kInterpThreads) void k_interp_bwd_lds(
    const float *__restrict__ gvalues, const float *__restrict__ points,
    const int32_t *__restrict__ batch_indices, const int32_t *__restrict__ batch_start, int64_t n,
    int B, int C, int X, int Y, int Z, int cpw, float *__restrict__ gvox, int channels_first) {
  MF_DYN_LDS(float, s_grid);
  const int b = blockIdx.y;
  const int c0 = blockIdx.x * cpw;
  const int nc = min(cpw, C - c0);
  const int V = X * Y * Z;
  const int total = nc * V;
  for (int i = threadIdx.x; i < total; i += kInterpThreads) s_grid[i] = 0.0f;
  __syncthreads();
  for_rows_of_item(batch_indices, batch_start, n, b, B, [&](int64_t p, bool mine) {
    if (!mine) return;
    const float px = points[3 * p], py = points[3 * p + 1], pz = points[3 * p + 2];
    if (!plausible(px, py, pz)) return;
    Corner k;
    corners(px, py, pz, X, Y, Z, k);
    for (int c = 0; c < nc; ++c) {
      const float g = channels_first ? gvalues[(int64_t)(c0 + c) * n + p] : gvalues[p * C + c0 + c];
      float *dst = s_grid + c * V;
#pragma unroll
      for (int j = 0; j < 8; ++j)
        if (k.off[j] >= 0) atomicAdd(&dst[k.off[j]], k.w[j] * g);
    }
  });
  __syncthreads();
  stage_copy(gvox + ((int64_t)b * C + c0) * V, s_grid, total);
}

__global__ __launch_bounds__(kInterpThreads) void k_interp_bwd_direct(
    const float *__restrict__ gvalues, const float *__restrict__ points,
    const int32_t *__restrict__ batch_indices, int64_t n, int B, int C, int X, int Y, int Z,
    float *__restrict__ gvox, int channels_first) {
  int64_t p = (int64_t)blockIdx.x * kInterpThreads + threadIdx.x;
  if (p >= n) return;
  const int b = batch_indices[p];
  const int V = X * Y * Z;
  float px = points[3 * p], py = points[3 * p + 1], pz = points[3 * p + 2];
  if (!(plausible(px, py, pz) && b >= 0 && b < B)) return;
  Corner k;
  corners(px, py, pz, X, Y, Z, k);
  for (int c = blockIdx.y; c < C; c += gridDim.y) {
    float g = channels_first ? gvalues[(int64_t)c * n + p] : gvalues[p * C + c];
    float *dst = gvox + ((int64_t)b * C + c) * V;
#pragma unroll
    for (int j = 0; j < 8; ++j)
      if (k.off[j] >= 0) atomicAdd(&dst[k.off[j]], k.w[j] * g);
  }
}

constexpr int kMaxLds = 160 * 1024;

// channels per workgroup: fill <= 64 KB of LDS (2 WGs/CU) but keep the grid large.
int pick_cpw(int C, int B, int64_t V) {
  int64_t bytes_per_ch = V * 4;
  int cpw = (int)std::max<int64_t>(1, (64 * 1024) / bytes_per_ch);
  cpw = std::min(cpw, C);
  while (cpw > 1 && (int64_t)B * ((C + cpw - 1) / cpw) < 512) cpw >>= 1;
  return cpw;
}

// ---- channels-last forward: vox [B][X*Y*Z][C] -> out rows of pitch ``ldo`` ---------------------
// With the channel index innermost a corner is C contiguous floats: one WAVE per point reads its 8
// corners as whole rows (16 bytes per lane, 1 KB per wave instruction, straight from L2 -- no LDS
// staging, no transpose) and writes C contiguous outputs at out[p * ldo ..], which lets the caller
// sample straight into a column block of the heads' [n, 984] feature matrix.  Weights, corner order
// and the per-channel sum order are those of ``corners`` / k_interp_fwd_vm: identical bits.
constexpr int kClPointsPerWave = 2;

__global__ __launch_bounds__(kInterpThreads) void k_interp_fwd_cl(
    const float *__restrict__ vox, const float *__restrict__ points,
    const int32_t *__restrict__ batch_indices, int64_t n, int B, int C, int X, int Y, int Z,
    float *__restrict__ out, int64_t ldo) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int64_t V = (int64_t)X * Y * Z;
  const int64_t p0 = ((int64_t)blockIdx.x * (kInterpThreads / 64) + wave) * kClPointsPerWave;
#pragma unroll
  for (int u = 0; u < kClPointsPerWave; ++u) {
    const int64_t p = p0 + u;
    if (p >= n) break;  // wave-uniform
    const float px = points[3 * p], py = points[3 * p + 1], pz = points[3 * p + 2];
    const int b = batch_indices[p];
    const bool ok = b >= 0 && b < B && plausible(px, py, pz);
    Corner k;
    if (ok) corners(px, py, pz, X, Y, Z, k);
    const float *grid = vox + (int64_t)(ok ? b : 0) * V * C;
    for (int c = 4 * lane; c < C; c += 256) {
      float4 acc = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
      if (ok) {
        float4 g[8];
#pragma unroll
        for (int j = 0; j < 8; ++j)
          g[j] = *reinterpret_cast<const float4 *>(grid + (int64_t)(k.off[j] < 0 ? 0 : k.off[j]) * C + c);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          if (k.off[j] < 0) continue;
          acc.x += k.w[j] * g[j].x;
          acc.y += k.w[j] * g[j].y;
          acc.z += k.w[j] * g[j].z;
          acc.w += k.w[j] * g[j].w;
        }
      }
      *reinterpret_cast<float4 *>(out + p * ldo + c) = acc;
    }
  }
}


// ---- channels-last bf16 sampler, forward and backward (round 4: the bf16 training path) ------------------------
// vox bf16 [B][V][C] -> out bf16 rows [n][ldo]; one wave per point, a lane takes 8 channels (one 16-byte load per
// corner), fp32 accumulation in the corner order of ``corners`` (the weights are interpolate_voxel_grid.py:27-58's).
// Backward: the point's gradient row is scattered to its 8 corner rows with fp32 row atomics into a zero-filled
// gvox fp32 [B][V][C] (a corner row is C contiguous floats: one coalesced atomic instruction per 64 channels).
__global__ __launch_bounds__(kInterpThreads) void k_interp_cl_bf16_fwd(
    const uint16_t *__restrict__ vox, const float *__restrict__ points, const int32_t *__restrict__ batch_indices,
    int64_t n, int B, int C, int X, int Y, int Z, uint16_t *__restrict__ out, int64_t ldo) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int64_t V = (int64_t)X * Y * Z;
  const int64_t p0 = ((int64_t)blockIdx.x * (kInterpThreads / 64) + wave) * kClPointsPerWave;
#pragma unroll
  for (int u = 0; u < kClPointsPerWave; ++u) {
    const int64_t p = p0 + u;
    if (p >= n) break;  // wave-uniform
    const float px = points[3 * p], py = points[3 * p + 1], pz = points[3 * p + 2];
    const int b = batch_indices[p];
    const bool ok = b >= 0 && b < B && plausible(px, py, pz);
    Corner k;
    if (ok) corners(px, py, pz, X, Y, Z, k);
    const uint16_t *grid = vox + (int64_t)(ok ? b : 0) * V * C;
    for (int c = 8 * lane; c < C; c += 512) {
      float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
      if (ok) {
        uint4 g[8];
#pragma unroll
        for (int j = 0; j < 8; ++j)
          g[j] = *reinterpret_cast<const uint4 *>(grid + (int64_t)(k.off[j] < 0 ? 0 : k.off[j]) * C + c);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          if (k.off[j] < 0) continue;
          const uint32_t w[4] = {g[j].x, g[j].y, g[j].z, g[j].w};
#pragma unroll
          for (int d = 0; d < 4; ++d) {
            acc[2 * d] += k.w[j] * mf::bf16_lo(w[d]);
            acc[2 * d + 1] += k.w[j] * mf::bf16_hi(w[d]);
          }
        }
      }
      *reinterpret_cast<uint4 *>(out + p * ldo + c) =
          make_uint4(mf::pack_bf16x2(acc[0], acc[1]), mf::pack_bf16x2(acc[2], acc[3]), mf::pack_bf16x2(acc[4], acc[5]),
                     mf::pack_bf16x2(acc[6], acc[7]));
    }
  }
}

// Backward of the channels-last sampler WITHOUT float atomics: workgroup (cc, r, b) owns CPW channels of the voxels
// [r Vr, (r + 1) Vr) of item b, Vr = 512: every thread keeps the sums of two voxels in registers.  The item's points
// come 256 at a time; per pass
//   1. every lane computes its point's 8 corners, takes a slot in the per-voxel counter of each corner inside the
//      range (integer LDS atomics: 8 per point) and stages the point's CPW gradient values (32 bytes) in LDS;
//   2. an exclusive scan of the 512 counters turns the slots into a list sorted by voxel: (point, weight) pairs;
//   3. every thread walks the lists of its two voxels and accumulates weight x value from the staged rows.
// At the end each thread writes its voxels' CPW channels (bf16 or fp32): every element, zeros included.
// Earlier versions of round 4, measured in the training step (16000 points, 16^3 x 256 and 8^3 x 512 grids):
//  * one wave per point, fp32 atomics into a zero-filled global grid: 0.42 ms per call (78 G atomics/s);
//  * voxel-range owner with all channels, one wave per (point, corner) and LDS float atomics: 0.66 ms (the object
//    fills an eighth of the grid: a few workgroups held all the pairs and walked them one 1 KB row read at a time);
//  * this decomposition with LDS float atomics (CPW x Vr accumulators, 128 ds_add_f32 per point): 0.28 ms -- the LDS
//    atomic unit retires about one lane per clock, whatever the bank pattern.
constexpr int kBwdVr = 2 * kInterpThreads;

template <int CPW>
__global__ __launch_bounds__(kInterpThreads) void k_interp_cl_bf16_bwd_sorted(
    const uint16_t *__restrict__ gout, int64_t ldg, const float *__restrict__ points,
    const int32_t *__restrict__ batch_indices, const int32_t *__restrict__ batch_start, int64_t n, int B, int C,
    int X, int Y, int Z, void *__restrict__ gvox, int out_bf16) {
  __shared__ __attribute__((aligned(16))) uint16_t s_g[kInterpThreads][CPW];  // the pass's gradient rows (this chunk)
  __shared__ int s_cnt[kBwdVr];        // entries per voxel, then (after the scan) the voxel's first entry
  __shared__ int s_wave[kInterpThreads / 64];
  __shared__ uint2 s_ent[kInterpThreads * 8];  // sorted by voxel: (point of the pass, weight bits)
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int c0 = blockIdx.x * CPW, v0 = blockIdx.y * kBwdVr, b = blockIdx.z, V = X * Y * Z;
  float acc[2][CPW];
#pragma unroll
  for (int u = 0; u < 2; ++u)
#pragma unroll
    for (int c = 0; c < CPW; ++c) acc[u][c] = 0.0f;
  // this item's rows: [batch_start[b], batch_start[b + 1]) when the caller has the offsets (points sorted by item),
  // else all n rows filtered by batch_indices (correct for any order)
  const int64_t p0 = batch_start ? row_clamp(batch_start[b], n) : 0;
  const int64_t p1 = batch_start ? row_clamp(batch_start[b + 1], n) : n;
  for (int64_t base = p0; base < p1; base += kInterpThreads) {  // block-uniform
    s_cnt[tid] = 0;
    s_cnt[tid + kInterpThreads] = 0;
    __syncthreads();
    const int64_t p = base + tid;
    int vox[8], slot[8];
    float wgt[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) vox[j] = -1;
    if (p < p1 && (batch_start || batch_indices[p] == b)) {
      const float px = points[3 * p], py = points[3 * p + 1], pz = points[3 * p + 2];
      if (plausible(px, py, pz)) {
        Corner k;
        corners(px, py, pz, X, Y, Z, k);
        bool any = false;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const int v = k.off[j] - v0;
          if (k.off[j] >= 0 && v >= 0 && v < kBwdVr) {
            vox[j] = v;
            wgt[j] = k.w[j];
            slot[j] = atomicAdd(&s_cnt[v], 1);
            any = true;
          }
        }
        if (any) {
          const uint16_t *row = gout + p * ldg + c0;
#pragma unroll
          for (int q = 0; q < CPW / 4; ++q)
            *reinterpret_cast<uint2 *>(&s_g[tid][4 * q]) = *reinterpret_cast<const uint2 *>(row + 4 * q);
        }
      }
    }
    __syncthreads();
    // exclusive scan of the 512 counters (thread t: counters 2 t, 2 t + 1)
    const int c_lo = s_cnt[2 * tid], c_hi = s_cnt[2 * tid + 1];
    int incl = c_lo + c_hi;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      const int up = __shfl_up(incl, d);
      if (lane >= d) incl += up;
    }
    if (lane == 63) s_wave[wave] = incl;
    __syncthreads();
    int before = incl - (c_lo + c_hi);
    int total = 0;
#pragma unroll
    for (int w = 0; w < kInterpThreads / 64; ++w) {
      if (w < wave) before += s_wave[w];
      total += s_wave[w];
    }
    if (total == 0) continue;  // block-uniform: no corner of this pass inside the range (no barrier pending)
    s_cnt[2 * tid] = before;
    s_cnt[2 * tid + 1] = before + c_lo;
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 8; ++j)
      if (vox[j] >= 0) s_ent[s_cnt[vox[j]] + slot[j]] = make_uint2((unsigned)tid, __float_as_uint(wgt[j]));
    __syncthreads();
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int v = tid + u * kInterpThreads;
      const int e0 = s_cnt[v], e1 = v + 1 < kBwdVr ? s_cnt[v + 1] : total;
      for (int e = e0; e < e1; ++e) {
        const uint2 ent = s_ent[e];
        const float w = __uint_as_float(ent.y);
#pragma unroll
        for (int q = 0; q < CPW / 4; ++q) {
          const uint2 g = *reinterpret_cast<const uint2 *>(&s_g[ent.x][4 * q]);
          acc[u][4 * q] += w * mf::bf16_lo(g.x);
          acc[u][4 * q + 1] += w * mf::bf16_hi(g.x);
          acc[u][4 * q + 2] += w * mf::bf16_lo(g.y);
          acc[u][4 * q + 3] += w * mf::bf16_hi(g.y);
        }
      }
    }
    __syncthreads();
  }
#pragma unroll
  for (int u = 0; u < 2; ++u) {
    const int v = v0 + tid + u * kInterpThreads;
    if (v >= V) continue;
    const int64_t o = ((int64_t)b * V + v) * C + c0;
#pragma unroll
    for (int q = 0; q < CPW / 4; ++q) {
      if (out_bf16)
        *reinterpret_cast<uint2 *>(static_cast<uint16_t *>(gvox) + o + 4 * q) = make_uint2(
            mf::pack_bf16x2(acc[u][4 * q], acc[u][4 * q + 1]), mf::pack_bf16x2(acc[u][4 * q + 2], acc[u][4 * q + 3]));
      else
        *reinterpret_cast<float4 *>(static_cast<float *>(gvox) + o + 4 * q) =
            make_float4(acc[u][4 * q], acc[u][4 * q + 1], acc[u][4 * q + 2], acc[u][4 * q + 3]);
    }
  }
}

}  // namespace

extern "C" int mf_interpolate_voxel_grid_fwd(const float *vox, const float *points,
                                             const int32_t *batch_indices,
                                             const int32_t *batch_start, int64_t n, int B,
                                             int C, int X, int Y, int Z, float *values,
                                             int channels_first, mfStream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (n == 0 || C == 0) return 0;
  const int64_t V = (int64_t)X * Y * Z;
  if (V * 4 <= kMaxLds) {
    const int cpw = pick_cpw(C, B, V);
    const size_t lds = (size_t)cpw * V * 4;
    if ((cpw == 4 || cpw == 8) && C % cpw == 0 && lds <= 64 * 1024) {  // voxel-major chunk, 16 B gathers
      if (cpw == 4)
        hipLaunchKernelGGL(k_interp_fwd_vm<4>, dim3(C / cpw, B), dim3(kInterpThreads), lds, stream, vox, points,
                           batch_indices, batch_start, n, B, C, X, Y, Z, values, channels_first);
      else
        hipLaunchKernelGGL(k_interp_fwd_vm<8>, dim3(C / cpw, B), dim3(kInterpThreads), lds, stream, vox, points,
                           batch_indices, batch_start, n, B, C, X, Y, Z, values, channels_first);
      return mf::check_launch("mf_interpolate_voxel_grid_fwd");
    }
    if (int e = mf::allow_big_lds((const void *)k_interp_fwd_lds, kMaxLds)) return e;
    hipLaunchKernelGGL(k_interp_fwd_lds, dim3((C + cpw - 1) / cpw, B), dim3(kInterpThreads), lds,
                       stream, vox, points, batch_indices, batch_start, n, B, C, X, Y, Z, cpw,
                       values, channels_first);
  } else {
    hipLaunchKernelGGL(k_interp_fwd_direct,
                       dim3((unsigned)((n + kInterpThreads - 1) / kInterpThreads),
                            std::min(C, 64)),
                       dim3(kInterpThreads), 0, stream, vox, points, batch_indices, n, B, C, X, Y,
                       Z, values, channels_first);
  }
  return mf::check_launch("mf_interpolate_voxel_grid_fwd");
}

extern "C" int mf_interpolate_voxel_grid_bwd(const float *gvalues, const float *points,
                                             const int32_t *batch_indices,
                                             const int32_t *batch_start, int64_t n, int B,
                                             int C, int X, int Y, int Z, float *gvox,
                                             int channels_first, mfStream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  const int64_t V = (int64_t)X * Y * Z;
  if (C == 0 || B == 0 || V == 0) return 0;
  if (V * 4 <= kMaxLds) {
    const int cpw = pick_cpw(C, B, V);
    const size_t lds = (size_t)cpw * V * 4;
    if (int e = mf::allow_big_lds((const void *)k_interp_bwd_lds, kMaxLds)) return e;
    hipLaunchKernelGGL(k_interp_bwd_lds, dim3((C + cpw - 1) / cpw, B), dim3(kInterpThreads), lds,
                       stream, gvalues, points, batch_indices, batch_start, n, B, C, X, Y, Z, cpw,
                       gvox, channels_first);
  } else {
    if (int e_ = mf::fill_bytes(gvox, 0, sizeof(float) * B * C * V, stream)) return e_;
    if (n > 0)
      hipLaunchKernelGGL(k_interp_bwd_direct,
                         dim3((unsigned)((n + kInterpThreads - 1) / kInterpThreads),
                              std::min(C, 64)),
                         dim3(kInterpThreads), 0, stream, gvalues, points, batch_indices, n, B, C,
                         X, Y, Z, gvox, channels_first);
  }
  return mf::check_launch("mf_interpolate_voxel_grid_bwd");
}

extern "C" int mf_interpolate_voxel_grid_cl_fwd(const float *vox, const float *points,
                                                const int32_t *batch_indices, int64_t n, int B, int C,
                                                int X, int Y, int Z, float *out, int64_t ldo,
                                                mfStream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (n <= 0) return 0;
  if (C % 4 || ldo % 4 || ldo < C || ((uintptr_t)out & 15) || ((uintptr_t)vox & 15)) {
    mf::set_last_error(hipErrorInvalidValue,
                       "interpolate_voxel_grid (channels-last): need C % 4 == 0, ldo % 4 == 0, 16-byte aligned out");
    return -(int)hipErrorInvalidValue;
  }
  const int64_t per_block = (kInterpThreads / 64) * kClPointsPerWave;
  hipLaunchKernelGGL(k_interp_fwd_cl, dim3((unsigned)((n + per_block - 1) / per_block)), dim3(kInterpThreads), 0,
                     stream, vox, points, batch_indices, n, B, C, X, Y, Z, out, ldo);
  return mf::check_launch("mf_interpolate_voxel_grid_cl_fwd");
}

/* Channels-last bf16 sampler of the training path: vox bf16 [B, X*Y*Z, C] -> out bf16 [n, ldo >= C] (C % 8 == 0,
 * 16-byte aligned rows); backward: gout bf16 [n, ldg] -> gvox [B, X*Y*Z, C], bf16 (out_bf16 = 1) or fp32, every element
 * written (no zero fill needed); batch_start: B + 1 row offsets when the points are sorted by item, or NULL. */
extern "C" int mf_interpolate_voxel_grid_cl_bf16_fwd(const void *vox, const float *points,
                                                     const int32_t *batch_indices, int64_t n, int B, int C, int X,
                                                     int Y, int Z, void *out, int64_t ldo, mfStream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (n <= 0) return 0;
  if (C % 8 || ldo % 8 || ldo < C || ((uintptr_t)out & 15) || ((uintptr_t)vox & 15)) {
    mf::set_last_error(hipErrorInvalidValue, "interpolate_voxel_grid_cl_bf16: C % 8 == 0, ldo % 8 == 0, aligned");
    return -(int)hipErrorInvalidValue;
  }
  const int64_t per_block = (kInterpThreads / 64) * kClPointsPerWave;
  hipLaunchKernelGGL(k_interp_cl_bf16_fwd, dim3((unsigned)((n + per_block - 1) / per_block)), dim3(kInterpThreads), 0,
                     stream, (const uint16_t *)vox, points, batch_indices, n, B, C, X, Y, Z, (uint16_t *)out, ldo);
  return mf::check_launch("mf_interpolate_voxel_grid_cl_bf16_fwd");
}

extern "C" int mf_interpolate_voxel_grid_cl_bf16_bwd(const void *gout, int64_t ldg, const float *points,
                                                     const int32_t *batch_indices, const int32_t *batch_start,
                                                     int64_t n, int B, int C, int X, int Y, int Z, void *gvox,
                                                     int32_t out_bf16, mfStream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  const int64_t V = (int64_t)X * Y * Z;
  if ((int64_t)B * V * C == 0) return 0;
  if (C % 4 || ldg % 4 || B > 65535 || V > (1 << 24) || (((uintptr_t)gout | (uintptr_t)gvox) & 15)) {
    mf::set_last_error(hipErrorInvalidValue,
                       "interpolate_voxel_grid_cl_bf16 backward: C % 4 == 0, ldg % 4 == 0, 16-byte aligned, B < 65536");
    return -(int)hipErrorInvalidValue;
  }
  const int cpw = C % 16 == 0 ? 16 : (C % 8 == 0 ? 8 : 4);
  const dim3 grid(C / cpw, (unsigned)((V + kBwdVr - 1) / kBwdVr), B);
  const int64_t nn = n > 0 ? n : 0;
  if (cpw == 16)
    hipLaunchKernelGGL(k_interp_cl_bf16_bwd_sorted<16>, grid, dim3(kInterpThreads), 0, stream, (const uint16_t *)gout,
                       ldg, points, batch_indices, batch_start, nn, B, C, X, Y, Z, gvox, out_bf16);
  else if (cpw == 8)
    hipLaunchKernelGGL(k_interp_cl_bf16_bwd_sorted<8>, grid, dim3(kInterpThreads), 0, stream, (const uint16_t *)gout,
                       ldg, points, batch_indices, batch_start, nn, B, C, X, Y, Z, gvox, out_bf16);
  else
    hipLaunchKernelGGL(k_interp_cl_bf16_bwd_sorted<4>, grid, dim3(kInterpThreads), 0, stream, (const uint16_t *)gout,
                       ldg, points, batch_indices, batch_start, nn, B, C, X, Y, Z, gvox, out_bf16);
  return mf::check_launch("mf_interpolate_voxel_grid_cl_bf16_bwd");
}
