// Fused Iterative Collision Check (ICC) for gfx950: forward, backward and the
// chainer-Adam step of IterativeCollisionCheckLink, batched over independent scenes.
//
// Reference call graph (one iteration, N objects):
//   contrib/iterative_collision_check_link.py:31-99   transformation_matrix, N x
//   transform_points, 2N x pseudo_occupancy_voxelization (each: TDF kernel K7 with
//   global float atomics + ~15 elementwise launches), N x isnan().any() D2H syncs,
//   stack/maximum/sum; backward = 2N x K8 (truncated_distance_function.py:105-166)
//   + matmul/quaternion backward; optimizer.update().  ~300 launches and N host
//   syncs per iteration, x100 iterations
//   (examples/ycb_video/pose_refinement/check_iterative_collision_check_link.py:52-79).
//
// Here one iteration is THREE launches and the whole n_iter loop is one hipGraph:
//   k_icc_tdf    grid (x-plane tile, 2*O): pose -> world point -> TDF of the "own" / "other"
//                point set of object o; the tile's (min distance, arg-min id) live in LDS as
//                two 32-bit words per voxel and are resolved with two passes of 32-bit LDS
//                atomics (64-bit LDS atomics measured ~10x slower); epilogue stores the
//                winners and the per-grid max of the raw inside weight (integer atomicMax).
//   k_icc_accum  grid (block, O): per voxel pseudo-occupancy weights, max() with the
//                no-entry grid, partial sums of reward / penalty AND the pose-gradient
//                moments.  The loss gradient is linear in {1/S_t, 1/S_in, PN/S_in^2},
//                so moments are accumulated per coefficient and combined later --
//                no second pass over the grids once the global sums are known.
//   k_icc_step   grid (scene): fixed-order reduction of the partials, loss, chain rule
//                to (q, t), chainer-Adam update, next iteration's rotation matrices.
// Every reduction has a fixed order (ordered partials, wave-sliced block sums, integer
// fixed-point limbs for the cross-object collision terms): bitwise reproducible run to
// run.  No host synchronisation anywhere.  MF_ICC_DEBUG / MF_ICC_SX are tuning aids.
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <map>
#include <mutex>
#include <tuple>
#include <vector>

#include "mf_common.h"

namespace {

__device__ unsigned long long g_dbg_stamps[4096 * 8];  // tuning aid (MF_ICC_DEBUG & 32)

constexpr int kTdfThreads = 1024;
constexpr int kAccThreads = 512;
constexpr int kVoxPerBlock = 1024;  // k_icc_accum: voxels per workgroup
constexpr int kNumOwn = 39;         // RN, S_in, PN + 3 x 12 gradient moments
constexpr uint32_t kNoCand = 0xffffffffu;
constexpr double kFix = 17592186044416.0;  // 2^44 fixed point for the collision moments
constexpr int kMaxSceneObjects = 32;

struct IccArgs {
  const float4 *pts4;
  const int32_t *obj_off;
  const int32_t *scene_off;
  const int32_t *obj_scene;
  const float *pitch;
  const float *origin;
  const float *grid_target;
  const float *grid_ne;
  int O, S, D;
  float thr, sdf_offset;
  // workspace
  unsigned long long *W;  // [2*O][V]
  uint32_t *Mbits;        // [2*O]
  float *Rt;              // [O][12]  R row-major, then t
  float *bound;           // [O][4]   model-frame bounding sphere
  float *St;              // [S]
  float *part;            // [O][NB][kNumOwn]
  float *oth;             // [O][NB][max_ns][12] collision moments per block
  int max_ns;
  int32_t *step;          // [S] (unused scratch)
  int4 *meta;             // [O] {scene first object, scene end object, point begin, point end}
  int dbg;                // tuning aid: MF_ICC_DEBUG bit mask (0 in production)
};

__device__ __forceinline__ void quat_to_R(const float *q, float *R) {
  // morefusion/functions/geometry/quaternion_matrix.py:65-78, :14-34
  const float n = ((q[0] * q[0] + q[1] * q[1]) + q[2] * q[2]) + q[3] * q[3];
  const float s = sqrtf(2.0f / n);
  const float qs[4] = {q[0] * s, q[1] * s, q[2] * s, q[3] * s};
  float Q[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) Q[i][j] = qs[i] * qs[j];
  R[0] = 1.0f - Q[2][2] - Q[3][3];
  R[1] = Q[1][2] - Q[3][0];
  R[2] = Q[1][3] + Q[2][0];
  R[3] = Q[1][2] + Q[3][0];
  R[4] = 1.0f - Q[1][1] - Q[3][3];
  R[5] = Q[2][3] - Q[1][0];
  R[6] = Q[1][3] - Q[2][0];
  R[7] = Q[2][3] + Q[1][0];
  R[8] = 1.0f - Q[1][1] - Q[2][2];
}

__device__ __forceinline__ void quat_backward(const float *q, const float *gR, float *gq) {
  // quaternion_matrix.py:36-51 (dR/dQ), outer product :54-62, scaling :71-72
  float gQ[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) gQ[i][j] = 0.0f;
  gQ[1][0] = -gR[5] + gR[7];
  gQ[1][1] = -gR[4] - gR[8];
  gQ[1][2] = gR[1] + gR[3];
  gQ[1][3] = gR[2] + gR[6];
  gQ[2][0] = gR[2] - gR[6];
  gQ[2][2] = -gR[0] - gR[8];
  gQ[2][3] = gR[5] + gR[7];
  gQ[3][0] = -gR[1] + gR[3];
  gQ[3][3] = -gR[0] - gR[4];
  const float n = ((q[0] * q[0] + q[1] * q[1]) + q[2] * q[2]) + q[3] * q[3];
  const float s = sqrtf(2.0f / n);
  const float qs[4] = {q[0] * s, q[1] * s, q[2] * s, q[3] * s};
  float gqs[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    float a = 0.0f, b = 0.0f;
#pragma unroll
    for (int j = 0; j < 4; ++j) { a += gQ[i][j] * qs[j]; b += gQ[j][i] * qs[j]; }
    gqs[i] = a + b;
  }
  const float dot = ((gqs[0] * q[0] + gqs[1] * q[1]) + gqs[2] * q[2]) + gqs[3] * q[3];
#pragma unroll
  for (int i = 0; i < 4; ++i) gq[i] = s * gqs[i] - (s / n) * dot * q[i];
}

// ---- setup: bounding spheres, sum(grid_target) per scene, R|t from (q,t) -----------
__global__ __launch_bounds__(256) void k_icc_bound(IccArgs a) {
  __shared__ float s_red[4][4];
  const int o = blockIdx.x;
  const int p0 = a.obj_off[o], p1 = a.obj_off[o + 1];
  float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
  for (int p = p0 + threadIdx.x; p < p1; p += blockDim.x) {
    const float4 m = a.pts4[p];
    lo[0] = fminf(lo[0], m.x); hi[0] = fmaxf(hi[0], m.x);
    lo[1] = fminf(lo[1], m.y); hi[1] = fmaxf(hi[1], m.y);
    lo[2] = fminf(lo[2], m.z); hi[2] = fmaxf(hi[2], m.z);
  }
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  float c[3];
#pragma unroll
  for (int d = 0; d < 3; ++d) {
    const float l = -mf::wave_max(-lo[d]), h = mf::wave_max(hi[d]);
    __syncthreads();
    if (lane == 0) { s_red[wave][0] = l; s_red[wave][1] = h; }
    __syncthreads();
    const float L = fminf(fminf(s_red[0][0], s_red[1][0]), fminf(s_red[2][0], s_red[3][0]));
    const float H = fmaxf(fmaxf(s_red[0][1], s_red[1][1]), fmaxf(s_red[2][1], s_red[3][1]));
    c[d] = 0.5f * (L + H);
  }
  float r2 = 0.0f;
  for (int p = p0 + threadIdx.x; p < p1; p += blockDim.x) {
    const float4 m = a.pts4[p];
    const float dx = m.x - c[0], dy = m.y - c[1], dz = m.z - c[2];
    r2 = fmaxf(r2, dx * dx + dy * dy + dz * dz);
  }
  r2 = mf::wave_max(r2);
  __syncthreads();
  if (lane == 0) s_red[wave][0] = r2;
  __syncthreads();
  if (threadIdx.x == 0) {
    r2 = fmaxf(fmaxf(s_red[0][0], s_red[1][0]), fmaxf(s_red[2][0], s_red[3][0]));
    const bool empty = p1 <= p0;
    a.bound[4 * o + 0] = empty ? 0.0f : c[0];
    a.bound[4 * o + 1] = empty ? 0.0f : c[1];
    a.bound[4 * o + 2] = empty ? 0.0f : c[2];
    a.bound[4 * o + 3] = empty ? -1.0f : sqrtf(r2) * 1.0001f + 1e-6f;
    const int sc = a.obj_scene[o];
    a.meta[o] = make_int4(a.scene_off[sc], a.scene_off[sc + 1], p0, p1);
  }
}

__global__ __launch_bounds__(256) void k_icc_scene_setup(IccArgs a, int32_t step0) {
  __shared__ float s_red[4];
  const int s = blockIdx.x;
  if (threadIdx.x == 0) a.step[s] = step0;
  const int V = a.D * a.D * a.D;
  const int64_t b0 = (int64_t)a.scene_off[s] * V, b1 = (int64_t)a.scene_off[s + 1] * V;
  float acc = 0.0f;
  for (int64_t i = b0 + threadIdx.x; i < b1; i += blockDim.x) acc += a.grid_target[i];
  acc = mf::wave_sum(acc);
  if ((threadIdx.x & 63) == 0) s_red[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) a.St[s] = (s_red[0] + s_red[1]) + (s_red[2] + s_red[3]);
}

__global__ __launch_bounds__(64) void k_icc_pose(IccArgs a, const float *__restrict__ q,
                                                 const float *__restrict__ t) {
  const int o = blockIdx.x * blockDim.x + threadIdx.x;
  if (o >= a.O) return;
  float R[9];
  quat_to_R(q + 4 * o, R);
#pragma unroll
  for (int i = 0; i < 9; ++i) a.Rt[12 * o + i] = R[i];
#pragma unroll
  for (int i = 0; i < 3; ++i) a.Rt[12 * o + 9 + i] = t[3 * o + i];
  a.Mbits[2 * o] = 0;
  a.Mbits[2 * o + 1] = 0;
}

// ---- launch 1: TDF tiles in LDS -------------------------------------------------
// What measurement taught (profiles/): a dependent global load costs ~0.3-0.7 us, an
// empty launch ~5 us, ds_min_u64 is ~an order of magnitude slower than 32-bit LDS
// atomics, and every slab workgroup re-scanning every point is instruction-bound.  So:
//  (1) the <= 32 source objects' R|t, bounding spheres and point ranges are fetched by
//      one lane each, in parallel; whole objects are culled by their bounding sphere;
//  (2) objects are walked wave-uniformly (R|t in scalar registers, ~20 instructions per
//      rejected point), U loads in flight per lane;
//  (3) points whose 3^3 neighbourhood touches this tile are appended to an LDS list
//      (wave-aggregated), then (survivor, offset) work items are spread over all lanes;
//  (4) (min, arg-min) is resolved with two passes of 32-bit LDS atomics: pass 1
//      atomicMin(distance bits), pass 2 atomicMin(candidate id) among the candidates
//      that equal the minimum -- exact, deterministic (lowest id among ties).
// KS = kernel size of truncated_distance_function.py:36-38 (3 for voxel_threshold 2).
constexpr int kSurvCap = 12288;  // LDS survivor list of packed (object slot, point id): 48 KB

template <int KS>
__global__ __launch_bounds__(kTdfThreads, 8) void k_icc_tdf(IccArgs a, int ks_rt, int SX) {
  MF_DYN_LDS(uint32_t, s_dyn1);  // dist[nvox], id[nvox]
  __shared__ float s_Rt[kMaxSceneObjects][12];
  __shared__ int s_p0[kMaxSceneObjects], s_p1[kMaxSceneObjects];
  __shared__ uint32_t s_surv[kSurvCap];  // (object slot << 27) | point id
  __shared__ int s_nsurv;
  __shared__ float s_max[kTdfThreads / 64];
  const int ks = KS > 0 ? KS : ks_rt;
  const int h = ks / 2, K = ks * ks * ks;
  const int D = a.D;
  const int g = blockIdx.y, o = g >> 1, other = g & 1;
  const int4 meta = a.meta[o];  // {ja, jb, p0, p1}
  const int ja = meta.x, jb = meta.y;
  const int Ns = jb - ja;
  const int x0 = blockIdx.x * SX;
  const int sx = min(SX, D - x0);
  const int nvox = sx * D * D;
  uint32_t *s_dist = s_dyn1, *s_id = s_dyn1 + SX * D * D;
  const float pitch = a.pitch[o];
  const float ox = a.origin[3 * o], oy = a.origin[3 * o + 1], oz = a.origin[3 * o + 2];
  const float trunc = a.thr * pitch;
  const float fh = (float)h, inv_pitch = 1.0f / pitch;
  // conservative (approximate-arithmetic) rejection bounds, in voxel units
  const float xlo = (float)x0 - fh - 0.51f, xhi = (float)(x0 + sx - 1) + fh + 0.51f;
  const float glo = -fh - 0.51f, ghi = (float)(D - 1) + fh + 0.51f;
  for (int i = threadIdx.x; i < nvox; i += kTdfThreads) { s_dist[i] = 0x7f800000u; s_id[i] = kNoCand; }
  if (threadIdx.x == 0) s_nsurv = 0;
  // (1) per-object metadata, one lane per object
  if (threadIdx.x < Ns) {
    const int j = ja + threadIdx.x;
    int p0 = 0, p1 = 0;
    if (other ? (j != o) : (j == o)) {
      const float4 r0 = *reinterpret_cast<const float4 *>(a.Rt + 12 * j);
      const float4 r1 = *reinterpret_cast<const float4 *>(a.Rt + 12 * j + 4);
      const float4 r2 = *reinterpret_cast<const float4 *>(a.Rt + 12 * j + 8);
      const float4 b = *reinterpret_cast<const float4 *>(a.bound + 4 * j);
      const int4 mj = a.meta[j];
      float *R = s_Rt[threadIdx.x];
      R[0] = r0.x; R[1] = r0.y; R[2] = r0.z; R[3] = r0.w; R[4] = r1.x; R[5] = r1.y;
      R[6] = r1.z; R[7] = r1.w; R[8] = r2.x; R[9] = r2.y; R[10] = r2.z; R[11] = r2.w;
      // whole-object rejection with the model's bounding sphere
      const float cx = (((R[0] * b.x + R[1] * b.y) + R[2] * b.z) + R[9] - ox) * inv_pitch;
      const float cy = (((R[3] * b.x + R[4] * b.y) + R[5] * b.z) + R[10] - oy) * inv_pitch;
      const float cz = (((R[6] * b.x + R[7] * b.y) + R[8] * b.z) + R[11] - oz) * inv_pitch;
      const float r = b.w * inv_pitch + 0.05f + 1e-4f * (fabsf(cx) + fabsf(cy) + fabsf(cz));
      const bool hit = b.w >= 0.0f && !(cx + r < xlo || cx - r > xhi || cy + r < glo ||
                                        cy - r > ghi || cz + r < glo || cz - r > ghi);
      if (hit) { p0 = mj.z; p1 = mj.w; }
    }
    s_p0[threadIdx.x] = p0;
    s_p1[threadIdx.x] = p1;
  }
  __syncthreads();
  const int lane = threadIdx.x & 63;

  // Work item = one survivor: its ks*ks (y, z) columns times the x offsets inside this
  // tile.  Instruction count per lane is what bounds this kernel (1024 lanes share 4
  // SIMDs: ~8 cycles per instruction), so the inner loop is kept lean:
  //  pass 1 works on SQUARED distances in voxel units -- no sqrt, no pitch: ~15
  //         instructions per candidate; 32-bit atomicMin of the d2 bits behind a peek.
  //         dist = pitch*sqrt(d2) is monotone in d2, so the minimum is the same voxel.
  //  pass 2 re-derives, only for survivors that touched a minimum, the EXACT float
  //         distance of near-minimal candidates (d2 within a few ulp) and, where it equals
  //         the exact minimum and is < truncation, takes atomicMin of the candidate id:
  //         identical winners to the oracle (lowest id among equal ROUNDED distances).
  const float d2_hi = a.thr * a.thr * 1.00002f;  // conservative inclusion; exact test later
  auto items = [&](const int ns, const int pass, unsigned long long &maybe, const bool marks) {
    int item_no = 0;
    for (int si = threadIdx.x; si < ns; si += kTdfThreads, ++item_no) {
      const int mbit = item_no < 63 ? item_no : 63;
      if (pass == 2 && marks && !((maybe >> mbit) & 1ull)) continue;
      const uint32_t packed = s_surv[si];
      const uint32_t pid = packed & 0x07ffffffu;
      const float *R = s_Rt[packed >> 27];
      const float4 m = a.pts4[pid];
      // same expressions as the scan -> bit-identical coordinates
      float4 sv;
      sv.x = ((((R[0] * m.x + R[1] * m.y) + R[2] * m.z) + R[9]) - ox) / pitch;
      sv.y = ((((R[3] * m.x + R[4] * m.y) + R[5] * m.z) + R[10]) - oy) / pitch;
      sv.z = ((((R[6] * m.x + R[7] * m.y) + R[8] * m.z) + R[11]) - oz) / pitch;
      const int irx = (int)roundf(sv.x), iry = (int)roundf(sv.y), irz = (int)roundf(sv.z);
      const uint32_t idb = pid * (uint32_t)K;
      const int bb0 = max(0, x0 - irx + h), bb1 = min(ks - 1, x0 + sx - 1 - irx + h);
      bool cand = false;
      for (int bb = bb0; bb <= bb1; ++bb) {
        const int ix = irx + bb - h;
        const float dx = sv.x - (float)ix;
        const float dx2 = dx * dx;
#pragma unroll
        for (int aa = 0; aa < ks; ++aa) {
          const int iy = iry + aa - h;
          if (iy < 0 || iy >= D) continue;
          const float dy = sv.y - (float)iy;
          const float dxy = dx2 + dy * dy;  // (dx^2 + dy^2) + dz^2: the oracle's order
          const int lrow = ((ix - x0) * D + iy) * D;
#pragma unroll
          for (int cc = 0; cc < ks; ++cc) {
            const int iz = irz + cc - h;
            if (iz < 0 || iz >= D) continue;
            const float dz = sv.z - (float)iz;
            const float d2 = dxy + dz * dz;
            if (!(d2 < d2_hi)) continue;
            const uint32_t db = __float_as_uint(d2);
            const uint32_t cur = s_dist[lrow + iz];
            if (pass == 1) {
              if (db <= cur) { atomicMin(&s_dist[lrow + iz], db); cand = true; }
            } else if (db <= cur + 8u) {  // within a few ulp of the minimal d2
              const float dist = pitch * sqrtf(d2);
              const float dmin = pitch * sqrtf(__uint_as_float(cur));
              if (dist == dmin && dist < trunc)
                atomicMin(&s_id[lrow + iz], idb + (uint32_t)((aa * ks + bb) * ks + cc));
            }
          }
        }
      }
      if (pass == 1 && cand) maybe |= 1ull << mbit;
    }
  };

  // Stream every accepted object (wave-uniform R|t), appending tile survivors to the LDS
  // list; whenever the list could overflow during the next super-chunk it is drained
  // through items(pass) -- block-uniform decision behind a barrier.  Returns whether it
  // drained (then pass 2 must re-stream, because the list no longer holds everything).
  constexpr int U = 2;
  auto scan = [&](const int pass) -> bool {
    bool drained = false;
    unsigned long long unused = 0ull;
    for (int e = 0; e < Ns; ++e) {
      const int p0 = s_p0[e], p1 = s_p1[e];  // block-uniform
      if (p1 <= p0) continue;
      const float R0 = s_Rt[e][0], R1 = s_Rt[e][1], R2 = s_Rt[e][2], R3 = s_Rt[e][3],
                  R4 = s_Rt[e][4], R5 = s_Rt[e][5], R6 = s_Rt[e][6], R7 = s_Rt[e][7],
                  R8 = s_Rt[e][8], T0 = s_Rt[e][9], T1 = s_Rt[e][10], T2 = s_Rt[e][11];
      for (int c0 = p0; c0 < p1; c0 += kTdfThreads * U) {
        float4 mm[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
          const int p = c0 + u * kTdfThreads + threadIdx.x;
          mm[u] = p < p1 ? a.pts4[p] : make_float4(0, 0, 0, 0);
        }
        __syncthreads();  // s_nsurv below is the value every lane agrees on
        if (s_nsurv + kTdfThreads * U > kSurvCap) {
          items(s_nsurv, pass, unused, false);
          __syncthreads();
          if (threadIdx.x == 0) s_nsurv = 0;
          __syncthreads();
          drained = true;
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
          const int p = c0 + u * kTdfThreads + threadIdx.x;
          bool surv = false;
          float fx = 0, fy = 0, fz = 0;
          if (p < p1) {
            const float4 m = mm[u];
            // transform_points: ((R0 x + R1 y) + R2 z) + t, un-fused (oracle order)
            const float wx = ((R0 * m.x + R1 * m.y) + R2 * m.z) + T0;
            const float ax = (wx - ox) * inv_pitch;  // cheap reject before the IEEE divides
            const float ex = 0.01f + 1e-5f * fabsf(ax);
            if (ax >= xlo - ex && ax <= xhi + ex) {
              const float wy = ((R3 * m.x + R4 * m.y) + R5 * m.z) + T1;
              const float wz = ((R6 * m.x + R7 * m.y) + R8 * m.z) + T2;
              fx = (wx - ox) / pitch; fy = (wy - oy) / pitch; fz = (wz - oz) / pitch;
              const float rx = roundf(fx), ry = roundf(fy), rz = roundf(fz);
              surv = rx + fh >= (float)x0 && rx - fh < (float)(x0 + sx) && ry + fh >= 0.0f &&
                     ry - fh < (float)D && rz + fh >= 0.0f && rz - fh < (float)D;
            }
          }
          const unsigned long long mask = (a.dbg & 2) ? 0ull : __ballot(surv);
          if (mask == 0ull) continue;  // wave-uniform
          int base = 0;
          if (lane == 0) base = atomicAdd(&s_nsurv, __popcll(mask));
          base = __shfl(base, 0, 64);
          if (surv)
            s_surv[base + __popcll(mask & ((1ull << lane) - 1ull))] =
                ((uint32_t)e << 27) | (uint32_t)p;
        }
      }
    }
    return drained;
  };

  const int wg = blockIdx.y * gridDim.x + blockIdx.x;
  auto stamp = [&](int i) {
    if ((a.dbg & 32) && threadIdx.x == 0 && wg < 4096) g_dbg_stamps[wg * 8 + i] = wall_clock64();
  };
  stamp(0);
  unsigned long long maybe = 0ull;
  bool drained = false;
  if (!(a.dbg & 1)) drained = scan(1);
  __syncthreads();
  stamp(1);
  if ((a.dbg & 32) && threadIdx.x == 0 && wg < 4096) g_dbg_stamps[wg * 8 + 6] = (unsigned long long)s_nsurv | ((unsigned long long)drained << 32);
  if (!drained) {  // the common case: the whole tile's survivors are in LDS
    const int ns = (a.dbg & 4) ? 0 : s_nsurv;
    items(ns, 1, maybe, true);
    __syncthreads();
    stamp(2);
    items(ns, 2, maybe, true);
  } else {  // crowded tile: finish pass 1, then stream everything again for the ids
    items(s_nsurv, 1, maybe, false);
    __syncthreads();
    if (threadIdx.x == 0) s_nsurv = 0;
    __syncthreads();
    scan(2);
    __syncthreads();
    items(s_nsurv, 2, maybe, false);
  }
  __syncthreads();
  stamp(3);
  // epilogue: winners out (coalesced 8 B/lane) + max raw inside weight of this tile
  // (truncated_distance_function.py:198-204: -1 where no winner, + offset, clamp at 0)
  const float offset = other ? 0.0f : a.sdf_offset;
  unsigned long long *Wg = a.W + (int64_t)g * D * D * D + (int64_t)x0 * D * D;
  float wmax = 0.0f;
  for (int i = threadIdx.x; i < nvox; i += kTdfThreads) {
    const uint32_t lo = s_id[i];  // set only where pitch*sqrt(min d2) < trunc
    const float dist = lo != kNoCand ? pitch * sqrtf(__uint_as_float(s_dist[i])) : trunc;
    Wg[i] = ((unsigned long long)__float_as_uint(dist) << 32) | lo;
    float w = (lo != kNoCand ? a.pts4[lo / (uint32_t)K].w : -1.0f) + offset;
    w = w < 0.0f ? 0.0f : w;
    wmax = fmaxf(wmax, w);
  }
  wmax = mf::wave_max(wmax);
  if ((threadIdx.x & 63) == 0) s_max[threadIdx.x >> 6] = wmax;
  __syncthreads();
  if (threadIdx.x == 0) {
    float m = s_max[0];
#pragma unroll
    for (int i = 1; i < kTdfThreads / 64; ++i) m = fmaxf(m, s_max[i]);
    atomicMax(&a.Mbits[g], __float_as_uint(m));  // m >= 0: uint order == float order
  }
  stamp(4);
}

// ---- launch 2: weights, sums, gradient moments ------------------------------------
__device__ __forceinline__ void world_frac(const float *Rt, const float4 m, float ox, float oy,
                                           float oz, float pitch, int ix, int iy, int iz,
                                           float &ux, float &uy, float &uz, bool &ok) {
  const float wx = ((Rt[0] * m.x + Rt[1] * m.y) + Rt[2] * m.z) + Rt[9];
  const float wy = ((Rt[3] * m.x + Rt[4] * m.y) + Rt[5] * m.z) + Rt[10];
  const float wz = ((Rt[6] * m.x + Rt[7] * m.y) + Rt[8] * m.z) + Rt[11];
  const float dx = (wx - ox) / pitch - (float)ix;
  const float dy = (wy - oy) / pitch - (float)iy;
  const float dz = (wz - oz) / pitch - (float)iz;
  const float n = sqrtf((dx * dx + dy * dy) + dz * dz);
  ok = n > 0.0f;  // truncated_distance_function.py:141
  ux = dx / n; uy = dy / n; uz = dz / n;
}

constexpr int kVPT = kVoxPerBlock / kAccThreads;  // voxels per thread

__global__ __launch_bounds__(kAccThreads) void k_icc_accum(IccArgs a, int K) {
  __shared__ float s_tr[kAccThreads * kNumOwn];
  // collision moments as 2^44 fixed point split in three 20-bit limbs held in 32-bit LDS
  // words: <= 1024 adds per block can never overflow a limb, so plain NON-returning
  // ds_add_u32 suffice (no carries).  Integer addition is associative: the result is
  // independent of the order of the atomics (bitwise reproducible), and 32-bit LDS atomics
  // are ~10x cheaper than the 64-bit ones.
  __shared__ uint32_t s_l0[kMaxSceneObjects * 12], s_l1[kMaxSceneObjects * 12];
  __shared__ int32_t s_l2[kMaxSceneObjects * 12];
  __shared__ float s_Rt[kMaxSceneObjects][12];
  __shared__ int s_off[kMaxSceneObjects + 1];
  const int o = blockIdx.y;
  const int wg2 = 2048 + blockIdx.y * gridDim.x + blockIdx.x;
  auto stamp = [&](int i) {
    if ((a.dbg & 32) && threadIdx.x == 0 && wg2 < 4096) g_dbg_stamps[wg2 * 8 + i] = wall_clock64();
  };
  stamp(0);
  const int D = a.D, V = D * D * D;
  const int4 meta = a.meta[o];
  const int ja = meta.x, jb = meta.y;
  const int Ns = jb - ja;
  // all independent loads first: scene tables, scalars, and this thread's voxels
  if (threadIdx.x < Ns * 12) s_Rt[threadIdx.x / 12][threadIdx.x % 12] = a.Rt[12 * ja + threadIdx.x];
  if (threadIdx.x <= Ns) s_off[threadIdx.x] = a.obj_off[ja + threadIdx.x];
  for (int i = threadIdx.x; i < a.max_ns * 12; i += kAccThreads) { s_l0[i] = 0u; s_l1[i] = 0u; s_l2[i] = 0; }
  const float pitch = a.pitch[o];
  const float ox = a.origin[3 * o], oy = a.origin[3 * o + 1], oz = a.origin[3 * o + 2];
  const float M_own = __uint_as_float(a.Mbits[2 * o]);
  const float M_oth = __uint_as_float(a.Mbits[2 * o + 1]);
  const float trunc = a.thr * pitch;
  // iterative_collision_check_link.py:82: skip the max() when grid_other has NaN,
  // which happens iff its normaliser max(weight) is 0 (0/0 everywhere).
  const bool use_oth = (Ns > 1) && (M_oth != 0.0f);
  const unsigned long long *W_own = a.W + (int64_t)(2 * o) * V;
  const unsigned long long *W_oth = a.W + (int64_t)(2 * o + 1) * V;
  const float *tgt = a.grid_target + (int64_t)o * V;
  const float *gne = a.grid_ne + (int64_t)o * V;

  unsigned long long ko[kVPT], kk[kVPT];
  float ne_[kVPT], tg_[kVPT];
  float4 m_own[kVPT], m_oth[kVPT];
#pragma unroll
  for (int it = 0; it < kVPT; ++it) {
    const int v = blockIdx.x * kVoxPerBlock + it * kAccThreads + threadIdx.x;
    const bool in = v < V;
    ko[it] = in ? W_own[v] : (((unsigned long long)__float_as_uint(trunc) << 32) | kNoCand);
    kk[it] = (in && use_oth) ? W_oth[v] : (unsigned long long)kNoCand;
    ne_[it] = in ? gne[v] : 0.0f;
    tg_[it] = in ? tgt[v] : 0.0f;
  }
#pragma unroll
  for (int it = 0; it < kVPT; ++it) {  // second level: winner gathers
    const uint32_t lo = (uint32_t)ko[it], lo_o = (uint32_t)kk[it];
    m_own[it] = lo != kNoCand ? a.pts4[lo / (uint32_t)K] : make_float4(0, 0, 0, -1.0f);
    m_oth[it] = lo_o != kNoCand ? a.pts4[lo_o / (uint32_t)K] : make_float4(0, 0, 0, -1.0f);
  }
  __syncthreads();
  stamp(1);
  const float *Rt_o = s_Rt[o - ja];

  float acc[kNumOwn];
#pragma unroll
  for (int i = 0; i < kNumOwn; ++i) acc[i] = 0.0f;

#pragma unroll
  for (int it = 0; it < kVPT; ++it) {
    const int v = blockIdx.x * kVoxPerBlock + it * kAccThreads + threadIdx.x;
    if (v >= V) continue;
    const int iz = v % D, iy = (v / D) % D, ix = v / (D * D);
    const uint32_t lo = (uint32_t)ko[it];
    const bool has = lo != kNoCand;
    const float g = 1.0f - __uint_as_float((uint32_t)(ko[it] >> 32)) / trunc;  // 1 - tdf/trunc
    float w = m_own[it].w + a.sdf_offset;
    const bool neg = w < 0.0f;
    if (neg) w = 0.0f;
    const float win = w / M_own;
    const float wsurf = neg ? win : 1.0f - win;
    const float surf = g * wsurf, ins = g * win;
    const float ne = ne_[it], tg = tg_[it];
    float ne_eff = ne;
    bool oth_wins = false;
    float wo_in = 0.0f;
    const uint32_t lo_o = (uint32_t)kk[it];
    if (use_oth) {
      const float go = 1.0f - __uint_as_float((uint32_t)(kk[it] >> 32)) / trunc;
      float wo = m_oth[it].w + 0.0f;
      if (wo < 0.0f) wo = 0.0f;
      wo_in = wo / M_oth;
      const float oth = go * wo_in;
      // F.maximum(grid_nontarget_empty, grid_other): gradient to `other` only if larger
      oth_wins = !(ne >= oth);
      if (oth_wins) ne_eff = oth;
    }
    acc[0] += surf * tg;
    acc[1] += ins;
    acc[2] += ins * ne_eff;
    if (has) {
      float ux, uy, uz;
      bool ok;
      world_frac(Rt_o, m_own[it], ox, oy, oz, pitch, ix, iy, iz, ux, uy, uz, ok);
      if (ok) {
        const float A[3] = {wsurf * tg / trunc, win * ne_eff / trunc, win / trunc};
        const float u[3] = {ux, uy, uz};
#pragma unroll
        for (int k = 0; k < 3; ++k)
#pragma unroll
          for (int d = 0; d < 3; ++d) {
            const float s = u[d] * A[k];
            acc[3 + 12 * k + 4 * d + 0] += s * m_own[it].x;
            acc[3 + 12 * k + 4 * d + 1] += s * m_own[it].y;
            acc[3 + 12 * k + 4 * d + 2] += s * m_own[it].z;
            acc[3 + 12 * k + 4 * d + 3] += s;
          }
      }
    }
    if (oth_wins && lo_o != kNoCand && ins != 0.0f) {
      // collision term: gradient flows to the OTHER object's pose
      const uint32_t p = lo_o / (uint32_t)K;
      int e = 0;
      while (e + 1 < Ns && (int)p >= s_off[e + 1]) ++e;
      const float4 m = m_oth[it];  // fetched with the second-level gathers above
      float ux, uy, uz;
      bool ok;
      world_frac(s_Rt[e], m, ox, oy, oz, pitch, ix, iy, iz, ux, uy, uz, ok);
      const float B = wo_in * ins / trunc;
      if (ok && isfinite(B)) {
        const float u[3] = {ux, uy, uz};
#pragma unroll
        for (int d = 0; d < 3; ++d) {
          const float sB = u[d] * B;
          const float val[4] = {sB * m.x, sB * m.y, sB * m.z, sB};
#pragma unroll
          for (int c = 0; c < 4; ++c) {
            const long long x = __double2ll_rn((double)val[c] * kFix);
            const int idx = 12 * e + 4 * d + c;
            atomicAdd(&s_l0[idx], (uint32_t)(x & 0xfffff));
            atomicAdd(&s_l1[idx], (uint32_t)((x >> 20) & 0xfffff));
            atomicAdd(&s_l2[idx], (int32_t)(x >> 40));
          }
        }
      }
    }
  }
  // fixed-order block reduction: component-major LDS layout (conflict-free stores), then
  // each wave owns components {wave, wave+8, ...}: 8 strided LDS reads per lane + one
  // 6-step wave reduction per component -- ~30 cross-lane steps per wave instead of 234.
  stamp(2);
  constexpr int kWaves = kAccThreads / 64;
#pragma unroll
  for (int i = 0; i < kNumOwn; ++i) s_tr[i * kAccThreads + threadIdx.x] = acc[i];
  __syncthreads();
  {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    for (int c = wave; c < kNumOwn; c += kWaves) {
      float sacc = 0.0f;
#pragma unroll
      for (int k = 0; k < kWaves; ++k) sacc += s_tr[c * kAccThreads + lane + 64 * k];
      sacc = mf::wave_sum(sacc);
      if (lane == 0) a.part[((int64_t)o * gridDim.x + blockIdx.x) * kNumOwn + c] = sacc;
    }
  }
  stamp(3);
  // collision partials of this block (the barriers above order the LDS atomics)
  float *po = a.oth + ((int64_t)o * gridDim.x + blockIdx.x) * a.max_ns * 12;
  for (int i = threadIdx.x; i < a.max_ns * 12; i += kAccThreads) {
    const long long x = ((long long)s_l2[i] << 40) + ((long long)s_l1[i] << 20) + (long long)s_l0[i];
    po[i] = (float)((double)x / kFix);
  }
}

// ---- launch 3: reduce, loss, chain rule, chainer-Adam -----------------------------
// One 1024-lane workgroup per scene.  Every partial is fetched with independent,
// coalesced loads (one memory latency), reduced in LDS in a fixed order.
// mode 0: write loss/gq/gt only.  mode 1: Adam update in place + refresh R|t.
// aq/at: alpha_t of chainer's Adam for this step (evaluated in double on the host).
constexpr int kStepThreads = 1024;

__global__ __launch_bounds__(kStepThreads) void k_icc_step(IccArgs a, int NB, int mode, float *q,
                                                           float *t, float *adam_m, float *adam_v,
                                                           float aq, float at, float *loss_out,
                                                           float *gq_out, float *gt_out,
                                                           float *traj, int it) {
  MF_DYN_LDS(float, s_dyn);
  __shared__ float s_o[kMaxSceneObjects * 12];
  __shared__ float s_coef[4];
  const int sc = blockIdx.x;
  const int ja = a.scene_off[sc], jb = a.scene_off[sc + 1];
  const int Ns = jb - ja;
  float *s_part = s_dyn;                      // [Ns*NB][kNumOwn] raw copy
  float *s_tot = s_part + Ns * NB * kNumOwn;  // [Ns][kNumOwn]
  float *s_G = s_tot + Ns * kNumOwn;          // [Ns][12]
  float *s_orow = s_G + Ns * 12;              // [8][Ns*12] partial collision sums
  const float S_t = a.St[sc];
  // independent, coalesced loads: own partials -> LDS
  const float *src = a.part + (int64_t)ja * NB * kNumOwn;
  const int n_own = Ns * NB * kNumOwn;
  for (int i = threadIdx.x; i < n_own; i += kStepThreads) s_part[i] = src[i];
  {
    // collision partials [Ns*NB rows][max_ns*12]: 8 lane groups x (Ns*12) columns, each
    // lane sums its rows in increasing order with 8 loads in flight; then the 8 groups
    // are added in order -> fixed summation order, hence reproducible
    const int row = a.max_ns * 12, ncol = Ns * 12, n_rows = Ns * NB;
    const float *po = a.oth + (int64_t)ja * NB * row;
    if (threadIdx.x < 8 * ncol) {
      const int grp = threadIdx.x / ncol, c = threadIdx.x - grp * ncol;
      float sacc = 0.0f;
      for (int r0 = grp; r0 < n_rows; r0 += 64) {
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const int r = r0 + 8 * u;
          v[u] = r < n_rows ? po[(int64_t)r * row + c] : 0.0f;
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) sacc += v[u];
      }
      s_orow[threadIdx.x] = sacc;
    }
  }
  __syncthreads();
  if (threadIdx.x < Ns * 12) {
    float sacc = 0.0f;
#pragma unroll
    for (int grp = 0; grp < 8; ++grp) sacc += s_orow[grp * Ns * 12 + threadIdx.x];
    s_o[threadIdx.x] = sacc;
  }
  // fixed-order reduction over blocks: 4 lanes per (object, component), 2 shuffle steps
  for (int i = threadIdx.x; i < Ns * kNumOwn * 4; i += kStepThreads) {
    const int oc = i >> 2, sub = i & 3;
    const int jo = oc / kNumOwn, c = oc - jo * kNumOwn;
    const float *p = s_part + (int64_t)jo * NB * kNumOwn + c;
    float s = 0.0f;
    for (int b = sub; b < NB; b += 4) s += p[b * kNumOwn];
    s += __shfl_xor(s, 1, 64);
    s += __shfl_xor(s, 2, 64);
    if (sub == 0) s_tot[oc] = s;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    float RN = 0.0f, S_in = 0.0f, PN = 0.0f;
    for (int jo = 0; jo < Ns; ++jo) {
      RN += s_tot[jo * kNumOwn + 0];
      S_in += s_tot[jo * kNumOwn + 1];
      PN += s_tot[jo * kNumOwn + 2];
    }
    // iterative_collision_check_link.py:91-98
    const float reward = RN / S_t, penalty = PN / S_in;
    if (loss_out) loss_out[sc] = penalty - reward;
    s_coef[0] = 1.0f / S_t;
    s_coef[1] = 1.0f / S_in;
    s_coef[2] = PN / (S_in * S_in);
  }
  __syncthreads();
  if (threadIdx.x < Ns * 12) {
    const int jo = threadIdx.x / 12, c = threadIdx.x % 12;
    const float *U = s_tot + jo * kNumOwn + 3;
    const float oth = s_o[threadIdx.x];
    s_G[threadIdx.x] = ((s_coef[0] * U[c] - s_coef[1] * U[12 + c]) + s_coef[2] * U[24 + c]) -
                       s_coef[1] * oth;
  }
  __syncthreads();
  if (threadIdx.x < Ns) {
    const int o = ja + threadIdx.x;
    const float *G = s_G + threadIdx.x * 12;
    float gR[9], gt[3], gq[4], qq[4], tt[3];
#pragma unroll
    for (int d = 0; d < 3; ++d) {
      gR[3 * d + 0] = G[4 * d + 0];
      gR[3 * d + 1] = G[4 * d + 1];
      gR[3 * d + 2] = G[4 * d + 2];
      gt[d] = G[4 * d + 3];
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) qq[i] = q[4 * o + i];
#pragma unroll
    for (int i = 0; i < 3; ++i) tt[i] = t[3 * o + i];
    quat_backward(qq, gR, gq);
    if (gq_out) {
#pragma unroll
      for (int i = 0; i < 4; ++i) gq_out[4 * o + i] = gq[i];
#pragma unroll
      for (int i = 0; i < 3; ++i) gt_out[3 * o + i] = gt[i];
    }
    if (mode == 1) {
      if (traj) {
        float *tr = traj + ((int64_t)it * a.O + o) * 7;
#pragma unroll
        for (int i = 0; i < 4; ++i) tr[i] = qq[i];
#pragma unroll
        for (int i = 0; i < 3; ++i) tr[4 + i] = tt[i];
      }
      // chainer.optimizers.Adam (v7) update rule in float32
      const float omb1 = (float)(1.0 - 0.9), omb2 = (float)(1.0 - 0.999), eps = 1e-8f;
#pragma unroll
      for (int i = 0; i < 7; ++i) {
        const float gi = i < 4 ? gq[i] : gt[i - 4];
        float mm = adam_m[7 * o + i], vv = adam_v[7 * o + i];
        mm += omb1 * (gi - mm);
        vv += omb2 * (gi * gi - vv);
        adam_m[7 * o + i] = mm;
        adam_v[7 * o + i] = vv;
        const float upd = (i < 4 ? aq : at) * mm / (sqrtf(vv) + eps);
        if (i < 4) qq[i] -= upd; else tt[i - 4] -= upd;
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) q[4 * o + i] = qq[i];
#pragma unroll
      for (int i = 0; i < 3; ++i) t[3 * o + i] = tt[i];
      float R[9];
      quat_to_R(qq, R);
#pragma unroll
      for (int i = 0; i < 9; ++i) a.Rt[12 * o + i] = R[i];
#pragma unroll
      for (int i = 0; i < 3; ++i) a.Rt[12 * o + 9 + i] = tt[i];
    }
    // reset the per-iteration accumulators for the next launch 1
    a.Mbits[2 * o] = 0;
    a.Mbits[2 * o + 1] = 0;
  }
}

__global__ void k_pack(const float *__restrict__ points, const float *__restrict__ sdf, int64_t n,
                       float4 *__restrict__ out) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = make_float4(points[3 * i], points[3 * i + 1], points[3 * i + 2], sdf[i]);
}

// ---- host side ---------------------------------------------------------------------
inline int64_t align256(int64_t x) { return (x + 255) & ~(int64_t)255; }

struct WsLayout {
  int64_t W, M, Rt, bound, St, part, oth, step, meta, total;
  int NB;
};

WsLayout ws_layout(int O, int S, int D, int max_ns = kMaxSceneObjects) {
  WsLayout l;
  const int64_t V = (int64_t)D * D * D;
  l.NB = (int)((V + kVoxPerBlock - 1) / kVoxPerBlock);
  int64_t off = 0;
  l.W = off; off = align256(off + 2 * O * V * 8);
  l.M = off; off = align256(off + 2 * O * 4);
  l.Rt = off; off = align256(off + O * 12 * 4);
  l.bound = off; off = align256(off + O * 4 * 4);
  l.St = off; off = align256(off + S * 4);
  l.part = off; off = align256(off + (int64_t)O * l.NB * kNumOwn * 4);
  l.oth = off; off = align256(off + (int64_t)O * l.NB * max_ns * 12 * 4);
  l.step = off; off = align256(off + S * 4);
  l.meta = off; off = align256(off + (int64_t)O * 16);
  l.total = off;
  return l;
}

IccArgs make_args(const mfIccBatch *b, void *ws, int max_ns) {
  IccArgs a;
  a.pts4 = (const float4 *)b->pts4;
  a.obj_off = b->obj_off;
  a.scene_off = b->scene_off;
  a.obj_scene = b->obj_scene;
  a.pitch = b->pitch;
  a.origin = b->origin;
  a.grid_target = b->grid_target;
  a.grid_ne = b->grid_ne;
  a.O = b->n_objects;
  a.S = b->n_scenes;
  a.D = b->dim;
  a.thr = b->voxel_threshold;
  a.sdf_offset = b->sdf_offset;
  a.max_ns = max_ns;
  a.dbg = getenv("MF_ICC_DEBUG") ? atoi(getenv("MF_ICC_DEBUG")) : 0;
  const WsLayout l = ws_layout(a.O, a.S, a.D);
  char *p = (char *)ws;
  a.W = (unsigned long long *)(p + l.W);
  a.Mbits = (uint32_t *)(p + l.M);
  a.Rt = (float *)(p + l.Rt);
  a.bound = (float *)(p + l.bound);
  a.St = (float *)(p + l.St);
  a.part = (float *)(p + l.part);
  a.oth = (float *)(p + l.oth);
  a.step = (int32_t *)(p + l.step);
  a.meta = (int4 *)(p + l.meta);
  return a;
}

int ksize_host(float thr) {
  // truncation / pitch == thr exactly (truncation = thr * pitch in float32 is exact for
  // thr = 2; for other thresholds the float32 quotient is what the reference evaluates)
  int ks = (int)ceilf(thr);
  if (ks % 2 == 0) ks += 1;
  return ks;
}

void launch_iteration(const IccArgs &a, int ks, int SX, int NB, int max_ns, int mode, float *q,
                      float *t, float *adam_m, float *adam_v, float alpha_q, float alpha_t,
                      int adam_step, float *loss, float *gq, float *gt, float *traj, int it,
                      hipStream_t stream) {
  const int D = a.D;
  const dim3 g1((D + SX - 1) / SX, 2 * a.O);
  const size_t lds1 = (size_t)SX * D * D * 2 * sizeof(uint32_t);
  if (ks == 3)
    hipLaunchKernelGGL(k_icc_tdf<3>, g1, dim3(kTdfThreads), lds1, stream, a, ks, SX);
  else
    hipLaunchKernelGGL(k_icc_tdf<0>, g1, dim3(kTdfThreads), lds1, stream, a, ks, SX);
  hipLaunchKernelGGL(k_icc_accum, dim3(NB, a.O), dim3(kAccThreads), 0, stream, a, ks * ks * ks);
  // chainer Adam: alpha_t = alpha * sqrt(1 - b2^t) / (1 - b1^t), in double, cast once
  const double fix1 = 1.0 - pow(0.9, (double)adam_step), fix2 = 1.0 - pow(0.999, (double)adam_step);
  const float aq = (float)((double)alpha_q * sqrt(fix2) / fix1);
  const float at = (float)((double)alpha_t * sqrt(fix2) / fix1);
  const size_t lds3 = (size_t)max_ns * (NB * kNumOwn + kNumOwn + 12 + 8 * 12) * sizeof(float);
  hipLaunchKernelGGL(k_icc_step, dim3(a.S), dim3(kStepThreads), lds3, stream, a, NB, mode, q, t,
                     adam_m, adam_v, aq, at, loss, gq, gt, traj, it);
}

struct GraphKey {
  std::vector<uint64_t> v;
  bool operator<(const GraphKey &o) const { return v < o.v; }
};
std::map<GraphKey, hipGraphExec_t> g_graphs;
std::mutex g_graph_mu;

}  // namespace

extern "C" int64_t mf_icc_workspace_bytes(int32_t n_objects, int32_t n_scenes, int32_t dim) {
  return ws_layout(n_objects, n_scenes, dim).total;
}

extern "C" int mf_pack_points_sdf(const float *points, const float *sdf, int64_t n, void *pts4,
                                  mfStream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (n == 0) return 0;
  hipLaunchKernelGGL(k_pack, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, points, sdf,
                     n, (float4 *)pts4);
  return mf::check_launch("mf_pack_points_sdf");
}

static int icc_prepare_kernels() {
  static bool done = false;
  if (done) return 0;
  // static + dynamic LDS above 64 KB is opt-in
  MF_TRY(hipFuncSetAttribute((const void *)k_icc_tdf<3>,
                             hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024));
  MF_TRY(hipFuncSetAttribute((const void *)k_icc_tdf<0>,
                             hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024));
  MF_TRY(hipFuncSetAttribute((const void *)k_icc_step,
                             hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024));
  done = true;
  return 0;
}

static int icc_validate(const mfIccBatch *b) {
  if (int e = icc_prepare_kernels()) return e;
  if (!b || b->n_objects <= 0 || b->n_scenes <= 0 || b->dim <= 0 || b->dim > 64 ||
      b->max_scene_objects <= 0 || b->max_scene_objects > kMaxSceneObjects ||
      (size_t)b->max_scene_objects * (ws_layout(1, 1, b->dim).NB * kNumOwn + kNumOwn + 12 + 96) * 4 >
          150 * 1024 ||
      (double)b->n_points * 343.0 >= 4294967295.0 || b->n_points >= (1 << 27)) {
    mf::set_last_error(hipErrorInvalidValue, "mf_icc: invalid batch descriptor");
    return -(int)hipErrorInvalidValue;
  }
  return 0;
}

static int slab_planes(int D, int n_grids) {
  if (const char *e = getenv("MF_ICC_SX")) {
    const int v = atoi(e);
    if (v >= 1 && v * D * D <= 8192) return std::min(v, D);
  }
  // tile <= 64 KB of (dist, id) words; two 1024-lane workgroups per CU -> aim for >= 512
  int SX = std::max(1, std::min(D, 8192 / (D * D)));
  while (SX > 1 && (int64_t)((D + SX - 1) / SX) * n_grids < 512) SX = (SX + 1) / 2;
  return SX;
}

extern "C" int mf_icc_debug_stamps(unsigned long long *host_out, int n) {
  return -(int)hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_dbg_stamps), sizeof(unsigned long long) * n);
}

extern "C" int mf_icc_launch_tdf(const mfIccBatch *batch, const float *q, const float *t, void *ws,
                                 mfStream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (int e = icc_validate(batch)) return e;
  IccArgs a = make_args(batch, ws, batch->max_scene_objects);
  const int ks = ksize_host(a.thr);
  const int SX = slab_planes(a.D, 2 * a.O);
  const int D = a.D;
  if (q && t) hipLaunchKernelGGL(k_icc_pose, dim3((a.O + 63) / 64), dim3(64), 0, stream, a, q, t);
  const dim3 g1((D + SX - 1) / SX, 2 * a.O);
  const size_t lds1 = (size_t)SX * D * D * 2 * sizeof(uint32_t);
  if (ks == 3)
    hipLaunchKernelGGL(k_icc_tdf<3>, g1, dim3(kTdfThreads), lds1, stream, a, ks, SX);
  else
    hipLaunchKernelGGL(k_icc_tdf<0>, g1, dim3(kTdfThreads), lds1, stream, a, ks, SX);
  return mf::check_launch("mf_icc_launch_tdf");
}

extern "C" int mf_icc_prepare(const mfIccBatch *batch, void *ws, mfStream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (int e = icc_validate(batch)) return e;
  IccArgs a = make_args(batch, ws, batch->max_scene_objects);
  hipLaunchKernelGGL(k_icc_bound, dim3(a.O), dim3(256), 0, stream, a);
  hipLaunchKernelGGL(k_icc_scene_setup, dim3(a.S), dim3(256), 0, stream, a, 0);
  return mf::check_launch("mf_icc_prepare");
}

extern "C" int mf_icc_loss_grad(const mfIccBatch *batch, const float *q, const float *t,
                                float *loss, float *gq, float *gt, void *ws,
                                mfStream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (int e = icc_validate(batch)) return e;
  const int max_ns = batch->max_scene_objects;
  IccArgs a = make_args(batch, ws, max_ns);
  const WsLayout l = ws_layout(a.O, a.S, a.D);
  const int ks = ksize_host(a.thr);
  const int SX = slab_planes(a.D, 2 * a.O);
  hipLaunchKernelGGL(k_icc_pose, dim3((a.O + 63) / 64), dim3(64), 0, stream, a, q, t);
  launch_iteration(a, ks, SX, l.NB, max_ns, 0, const_cast<float *>(q), const_cast<float *>(t),
                   nullptr, nullptr, 0.0f, 0.0f, 1, loss, gq, gt, nullptr, 0, stream);
  return mf::check_launch("mf_icc_loss_grad");
}

extern "C" int mf_icc_refine(const mfIccBatch *batch, float *q, float *t, float *adam_m,
                             float *adam_v, int32_t n_iter, int32_t step0, float alpha_q,
                             float alpha_t, float *losses, float *traj, void *ws,
                             mfStream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (int e = icc_validate(batch)) return e;
  if (n_iter <= 0) return 0;
  const int max_ns = batch->max_scene_objects;
  IccArgs a = make_args(batch, ws, max_ns);
  const WsLayout l = ws_layout(a.O, a.S, a.D);
  const int ks = ksize_host(a.thr);
  const int SX = slab_planes(a.D, 2 * a.O);

  GraphKey key;
  auto push = [&](const void *p) { key.v.push_back((uint64_t)(uintptr_t)p); };
  push(batch->pts4); push(batch->obj_off); push(batch->scene_off); push(batch->obj_scene);
  push(batch->pitch); push(batch->origin); push(batch->grid_target); push(batch->grid_ne);
  push(q); push(t); push(adam_m); push(adam_v); push(losses); push(traj); push(ws);
  uint32_t fb[4];
  memcpy(&fb[0], &alpha_q, 4); memcpy(&fb[1], &alpha_t, 4);
  memcpy(&fb[2], &a.thr, 4); memcpy(&fb[3], &a.sdf_offset, 4);
  key.v.push_back(((uint64_t)fb[0] << 32) | fb[1]);
  key.v.push_back(((uint64_t)fb[2] << 32) | fb[3]);
  key.v.push_back(((uint64_t)(uint32_t)a.O << 32) | (uint32_t)a.S);
  key.v.push_back(((uint64_t)(uint32_t)a.D << 32) | (uint32_t)batch->n_points);
  key.v.push_back(((uint64_t)(uint32_t)n_iter << 32) | (uint32_t)step0);
  key.v.push_back((uint64_t)max_ns);

  std::lock_guard<std::mutex> lock(g_graph_mu);
  auto itg = g_graphs.find(key);
  if (itg == g_graphs.end()) {
    hipGraph_t graph = nullptr;
    // The caller's stream may be the legacy NULL stream (torch's default), which cannot be
    // captured: record the graph on a private stream, replay it on the caller's.
    static hipStream_t cap = nullptr;
    if (!cap) MF_TRY(hipStreamCreateWithFlags(&cap, hipStreamNonBlocking));
    MF_TRY(hipStreamBeginCapture(cap, hipStreamCaptureModeThreadLocal));
    hipLaunchKernelGGL(k_icc_pose, dim3((a.O + 63) / 64), dim3(64), 0, cap, a, q, t);
    for (int it = 0; it < n_iter; ++it)
      launch_iteration(a, ks, SX, l.NB, max_ns, 1, q, t, adam_m, adam_v, alpha_q, alpha_t,
                       step0 + it + 1, losses ? losses + (int64_t)it * a.S : nullptr, nullptr,
                       nullptr, traj, it, cap);
    hipError_t ce = hipStreamEndCapture(cap, &graph);
    if (ce != hipSuccess) {
      mf::set_last_error(ce, "hipStreamEndCapture(icc)");
      return -(int)ce;
    }
    hipGraphExec_t exec = nullptr;
    hipError_t ie = hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0);
    (void)hipGraphDestroy(graph);
    if (ie != hipSuccess) {
      mf::set_last_error(ie, "hipGraphInstantiate(icc)");
      return -(int)ie;
    }
    if (g_graphs.size() >= 64) {  // bounded cache
      for (auto &kv : g_graphs) (void)hipGraphExecDestroy(kv.second);
      g_graphs.clear();
    }
    itg = g_graphs.emplace(key, exec).first;
  }
  MF_TRY(hipGraphLaunch(itg->second, stream));
  return 0;
}
