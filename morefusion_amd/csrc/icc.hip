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
  // x-plane bins of the per-iteration point binning (k_icc_bin -> k_icc_tile)
  int4 *tab;              // [n_tab] {target object o, source object j, point begin, point end}; o < 0: unused
  int n_tab;
  int nbins;              // D + 2h planes: rounded x in [-h, D-1+h]
  uint32_t *bin_cnt;      // [2*O][nbins] records in each bin (zero between iterations)
  int32_t *bin_cap;       // [2*O] capacity of each bin of grid g = number of its source points
  int64_t *bin_base;      // [2*O] first record of grid g's bins; bin b starts at base + b*cap
  float4 *rec;            // records {fx, fy, fz, point id bits}: voxel-frame coordinates
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

// Kernel size of one grid: truncated_distance_function.py:36-38 evaluates
// ceil(truncation / pitch) in float32 with truncation = threshold * pitch
// (:184), made odd.  For threshold 2 (the link's default) the quotient is exactly 2 -> 3;
// for other thresholds it depends on the rounding of the two float32 operations, i.e. on the
// grid's pitch -- so it is evaluated per grid, like the reference does.
__device__ __forceinline__ int ksize_of(float thr, float pitch) {
  int ks = (int)ceilf((thr * pitch) / pitch);
  if (ks % 2 == 0) ks += 1;
  return ks;
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

// ---- v2 front end: per-iteration x-plane binning + bin-fed TDF tiles -----------------
// v1's k_icc_tdf lets every one of the 32 plane workgroups of a grid re-scan all source points of
// the scene (32x read amplification, a dependent global load per work item).  v2 transforms every
// (source point, target grid) pair ONCE (k_icc_bin), appends the survivors' voxel-frame
// coordinates to the bin of their rounded x-plane, and the tile of plane x reads only bins
// x-h..x+h (k_icc_tile): records arrive as coalesced 16 B loads, both LDS passes run on
// registers + LDS only.  Coordinates are computed with the oracle's expressions, the candidate
// set of a tile is exactly v1's survivor set, (min, arg-min) are exact -> bit-identical winners.
constexpr int kBinThreads = 256;
constexpr int kBinPPT = 4;                          // points per thread
constexpr int kBinChunk = kBinThreads * kBinPPT;    // points per workgroup
constexpr int kMaxBins = 64 + 6;                    // D <= 64, ks <= 7

// Once per batch: bin capacities/offsets per grid and the (target, source, point chunk) table.
__global__ __launch_bounds__(256) void k_icc_tables(IccArgs a) {
  __shared__ int s_tab_base[1];
  if (threadIdx.x == 0) {
    int64_t rec_off = 0;
    int tab_off = 0;
    for (int o = 0; o < a.O; ++o) {
      const int sc = a.obj_scene[o];
      const int ja = a.scene_off[sc], jb = a.scene_off[sc + 1];
      const int p_own = a.obj_off[o + 1] - a.obj_off[o];
      const int p_all = a.obj_off[jb] - a.obj_off[ja];
      a.bin_cap[2 * o] = p_own;
      a.bin_base[2 * o] = rec_off;
      rec_off += (int64_t)a.nbins * p_own;
      a.bin_cap[2 * o + 1] = p_all - p_own;
      a.bin_base[2 * o + 1] = rec_off;
      rec_off += (int64_t)a.nbins * (p_all - p_own);
      for (int j = ja; j < jb; ++j) {
        const int p0 = a.obj_off[j], p1 = a.obj_off[j + 1];
        for (int c = p0; c < p1; c += kBinChunk)
          if (tab_off < a.n_tab) a.tab[tab_off++] = make_int4(o, j, c, min(c + kBinChunk, p1));
      }
    }
    s_tab_base[0] = tab_off;
  }
  __syncthreads();
  for (int i = s_tab_base[0] + threadIdx.x; i < a.n_tab; i += blockDim.x) a.tab[i] = make_int4(-1, -1, 0, 0);
  for (int i = threadIdx.x; i < 2 * a.O * a.nbins; i += blockDim.x) a.bin_cnt[i] = 0u;
}

// launch 1: one workgroup per (target grid, source object, chunk of <= 1024 points)
__global__ __launch_bounds__(kBinThreads) void k_icc_bin(IccArgs a, int hmax) {
  __shared__ int s_cnt[kMaxBins], s_base[kMaxBins];
  auto stamp = [&](int i) {  // tuning aid (MF_ICC_DEBUG & 32)
    if ((a.dbg & 32) && threadIdx.x == 0 && blockIdx.x < 1024)
      g_dbg_stamps[(3072 + blockIdx.x) * 8 + i] = wall_clock64();
  };
  stamp(0);
  const int4 e = a.tab[blockIdx.x];
  const int o = e.x, j = e.y;
  if (o < 0) return;  // block-uniform
  const int D = a.D, nb = a.nbins;
  const int g = 2 * o + (j != o ? 1 : 0);
  // everything below depends on the table entry only: one memory round trip
  const float4 r0 = *reinterpret_cast<const float4 *>(a.Rt + 12 * j);
  const float4 r1 = *reinterpret_cast<const float4 *>(a.Rt + 12 * j + 4);
  const float4 r2 = *reinterpret_cast<const float4 *>(a.Rt + 12 * j + 8);
  const float4 bnd = *reinterpret_cast<const float4 *>(a.bound + 4 * j);
  const float pitch = a.pitch[o];
  const float ox = a.origin[3 * o], oy = a.origin[3 * o + 1], oz = a.origin[3 * o + 2];
  const int cap = a.bin_cap[g];
  const int64_t base_g = a.bin_base[g];
  float4 m[kBinPPT];
#pragma unroll
  for (int u = 0; u < kBinPPT; ++u) {
    const int p = e.z + u * kBinThreads + (int)threadIdx.x;
    m[u] = p < e.w ? a.pts4[p] : make_float4(0, 0, 0, 0);
  }
  for (int i = threadIdx.x; i < nb; i += kBinThreads) s_cnt[i] = 0;
  const float R0 = r0.x, R1 = r0.y, R2 = r0.z, R3 = r0.w, R4 = r1.x, R5 = r1.y, R6 = r1.z,
              R7 = r1.w, R8 = r2.x, T0 = r2.y, T1 = r2.z, T2 = r2.w;
  const int h = min(ksize_of(a.thr, pitch) / 2, hmax);
  const float fh = (float)h, inv_pitch = 1.0f / pitch;
  {
    // whole-object rejection with the model's bounding sphere (conservative, block-uniform)
    const float glo = -fh - 0.51f, ghi = (float)(D - 1) + fh + 0.51f;
    const float cx = (((R0 * bnd.x + R1 * bnd.y) + R2 * bnd.z) + T0 - ox) * inv_pitch;
    const float cy = (((R3 * bnd.x + R4 * bnd.y) + R5 * bnd.z) + T1 - oy) * inv_pitch;
    const float cz = (((R6 * bnd.x + R7 * bnd.y) + R8 * bnd.z) + T2 - oz) * inv_pitch;
    const float r = bnd.w * inv_pitch + 0.05f + 1e-4f * (fabsf(cx) + fabsf(cy) + fabsf(cz));
    const bool hit = bnd.w >= 0.0f && !(cx + r < glo || cx - r > ghi || cy + r < glo ||
                                         cy - r > ghi || cz + r < glo || cz - r > ghi);
    if (!hit) return;
  }
  __syncthreads();
  float fx[kBinPPT], fy[kBinPPT], fz[kBinPPT];
  int bin[kBinPPT], slot[kBinPPT];
#pragma unroll
  for (int u = 0; u < kBinPPT; ++u) {
    const int p = e.z + u * kBinThreads + (int)threadIdx.x;
    bin[u] = -1;
    slot[u] = 0;
    if (p < e.w) {
      // transform_points: ((R0 x + R1 y) + R2 z) + t, un-fused (oracle order), then
      // (p - origin) / pitch with a correctly rounded divide (voxelization_3d index rule)
      const float wx = ((R0 * m[u].x + R1 * m[u].y) + R2 * m[u].z) + T0;
      const float wy = ((R3 * m[u].x + R4 * m[u].y) + R5 * m[u].z) + T1;
      const float wz = ((R6 * m[u].x + R7 * m[u].y) + R8 * m[u].z) + T2;
      fx[u] = (wx - ox) / pitch; fy[u] = (wy - oy) / pitch; fz[u] = (wz - oz) / pitch;
      const float rx = roundf(fx[u]), ry = roundf(fy[u]), rz = roundf(fz[u]);
      const bool surv = rx + fh >= 0.0f && rx - fh < (float)D && ry + fh >= 0.0f &&
                        ry - fh < (float)D && rz + fh >= 0.0f && rz - fh < (float)D;
      if (surv) {
        bin[u] = (int)rx + hmax;  // in [0, D + 2 hmax)
        slot[u] = atomicAdd(&s_cnt[bin[u]], 1);
      }
    }
  }
  __syncthreads();
  stamp(1);
  for (int i = threadIdx.x; i < nb; i += kBinThreads) {
    const int c = s_cnt[i];
    s_base[i] = c > 0 ? (int)atomicAdd(&a.bin_cnt[(int64_t)g * nb + i], (uint32_t)c) : 0;
  }
  __syncthreads();
  stamp(2);
#pragma unroll
  for (int u = 0; u < kBinPPT; ++u) {
    if (bin[u] < 0) continue;
    const int idx = s_base[bin[u]] + slot[u];
    if (idx >= cap) continue;  // cannot happen while bin_cnt starts at zero; never write out of bounds
    const int p = e.z + u * kBinThreads + (int)threadIdx.x;
    a.rec[base_g + (int64_t)bin[u] * cap + idx] = make_float4(fx[u], fy[u], fz[u], __uint_as_float((uint32_t)p));
  }
  stamp(3);
}

// launch 2: TDF of one x-plane of one grid, fed from bins x-h..x+h.
//  pass 1 works on SQUARED distances in voxel units (no sqrt, no pitch): 32-bit atomicMin of
//         the d2 bits behind a peek.  dist = pitch*sqrt(d2) is monotone in d2.
//  pass 2 re-derives, only for records that touched a minimum, the EXACT float distance of
//         near-minimal candidates (d2 within a few ulp) and, where it equals the exact minimum
//         and is < truncation, takes atomicMin of the candidate id: the same winners as
//         the oracle (lowest id among equal ROUNDED distances).
constexpr int kTileThreads = 256;
constexpr int kTileStripes = 4;  // y-stripes per plane: a workgroup owns D/4 rows of one x-plane
constexpr int kTileKeep = 10;    // records per lane kept in registers over both passes
constexpr int kTileR = 4;        // records in flight per lane beyond those

// A crowded plane (2400 records at 8 objects) is bound by instruction issue and same-address LDS
// atomics of ONE workgroup while the other 500 idle: each plane is split in kTileStripes
// y-stripes.  Every stripe workgroup streams all records of the plane's bins (16 B each, L2
// hits) and keeps those whose ks rows touch its stripe; the (min, arg-min) of a voxel only
// depends on the set of candidates, so the winners do not change.
template <int KS>
__device__ __forceinline__ void icc_tile_body(const IccArgs &a, const int ks_rt, const int hmax) {
  MF_DYN_LDS(uint32_t, s_tile);  // dist[rows*D], id[rows*D]
  __shared__ float s_max[kTileThreads / 64];
  const int ks = KS > 0 ? KS : ks_rt;
  const int h = ks / 2, K = ks * ks * ks;
  const int D = a.D, nb = a.nbins;
  const int g = blockIdx.y, o = g >> 1, other = g & 1;
  const int x = blockIdx.x / kTileStripes, stripe = blockIdx.x % kTileStripes;
  const int rows_max = (D + kTileStripes - 1) / kTileStripes;
  const int y0 = stripe * rows_max, y1 = min(D, y0 + rows_max);
  const int nvox = max(0, y1 - y0) * D;
  uint32_t *s_dist = s_tile, *s_id = s_tile + rows_max * D;
  // independent loads: the <= 7 bin counts of this tile, capacity, offset
  int c[8];
  c[0] = 0;
  const int cap = a.bin_cap[g];
  const int64_t base_g = a.bin_base[g];
  const float pitch = a.pitch[o];
  const int bin0 = x + hmax - h;  // bin of plane x - h
#pragma unroll
  for (int b = 0; b < 7; ++b) {
    int n = 0;
    if (b < ks) n = min((int)a.bin_cnt[(int64_t)g * nb + bin0 + b], cap);
    c[b + 1] = c[b] + n;
  }
  const int T = c[7];
  const int wg = blockIdx.y * gridDim.x + blockIdx.x;
  auto stamp = [&](int i) {  // tuning aid (MF_ICC_DEBUG & 32)
    if ((a.dbg & 32) && threadIdx.x == 0 && wg < 2048) g_dbg_stamps[wg * 8 + i] = wall_clock64();
  };
  stamp(0);
  if ((a.dbg & 32) && threadIdx.x == 0 && wg < 2048) g_dbg_stamps[wg * 8 + 6] = (unsigned long long)T;
  const float trunc = a.thr * pitch;
  for (int i = threadIdx.x; i < nvox; i += kTileThreads) { s_dist[i] = 0x7f800000u; s_id[i] = kNoCand; }
  __syncthreads();
  const float d2_hi = a.thr * a.thr * 1.00002f;  // conservative inclusion; exact test in pass 2
  const float d2_in = a.thr * a.thr * 0.999f;    // certainly inside the truncation radius
  const float4 *recs = a.rec + base_g + (int64_t)bin0 * cap;
  const float fxp = (float)x;

  // record i of this tile's concatenated bins -> (bin b, record); rb < 0: none / not in my stripe
  auto fetch = [&](const int i, float4 &rv, int &rb) {
    rb = -1;
    if (i >= T) return;
    int b = 0;
#pragma unroll
    for (int k = 1; k < 7; ++k) b += (k < ks && i >= c[k]) ? 1 : 0;
    int cb = 0;
#pragma unroll
    for (int k = 1; k < 7; ++k) cb = (k == b) ? c[k] : cb;
    rb = b;
    rv = recs[(int64_t)b * cap + (i - cb)];
  };
  auto mine = [&](const float4 &rv) {  // do the ks rows around round(y) touch rows [y0, y1)?
    const int iry = (int)roundf(rv.y);
    return iry + h >= y0 && iry - h < y1;
  };
  // One record against its ks x ks (y, z) candidates in plane x.  All peeks of a record are
  // issued together (KS == 3: nine independent ds_read), then the non-returning atomics.  A
  // peek may be stale (another lane lowered the voxel meanwhile): values only decrease, so a
  // stale peek only lets MORE candidates through -- the atomicMin / the exact test of pass 2
  // decide.
  auto visit = [&](const int pass, const float4 sv, const int rb) -> bool {
    const int iry = (int)roundf(sv.y), irz = (int)roundf(sv.z);
    const uint32_t idb = __float_as_uint(sv.w) * (uint32_t)K;
    const int bb = ks - 1 - rb;  // x offset of plane x inside this point's neighbourhood
    const float dx = sv.x - fxp;
    const float dx2 = dx * dx;
    bool cand = false;
    if constexpr (KS == 3) {
      uint32_t db[9], cur[9];
      int ad[9];
#pragma unroll
      for (int aa = 0; aa < 3; ++aa) {
        const int iy = iry + aa - 1;
        const float dy = sv.y - (float)iy;
        const float dxy = dx2 + dy * dy;  // (dx^2 + dy^2) + dz^2: the oracle's order
#pragma unroll
        for (int cc = 0; cc < 3; ++cc) {
          const int iz = irz + cc - 1;
          const float dz = sv.z - (float)iz;
          const float d2 = dxy + dz * dz;
          const bool ok = iy >= y0 && iy < y1 && iz >= 0 && iz < D && d2 < d2_hi;
          db[aa * 3 + cc] = __float_as_uint(d2);
          ad[aa * 3 + cc] = ok ? (iy - y0) * D + iz : -1;
        }
      }
#pragma unroll
      for (int k = 0; k < 9; ++k) cur[k] = s_dist[ad[k] < 0 ? 0 : ad[k]];
#pragma unroll
      for (int k = 0; k < 9; ++k) {
        if (ad[k] < 0) continue;
        if (pass == 1) {
          if (db[k] <= cur[k]) { atomicMin(&s_dist[ad[k]], db[k]); cand = true; }
        } else if (db[k] <= cur[k] + 8u) {  // within a few ulp of the minimal d2
          // dist == dmin is certain for equal bits; dist < trunc is certain well inside the
          // truncation radius (pitch*sqrt(d2) <= 0.9995 thr pitch (1 + 2^-22) < trunc)
          bool win = db[k] == cur[k] && __uint_as_float(db[k]) < d2_in;
          if (!win) {
            const float dist = pitch * sqrtf(__uint_as_float(db[k]));
            const float dmin = pitch * sqrtf(__uint_as_float(cur[k]));
            win = dist == dmin && dist < trunc;
          }
          if (win) atomicMin(&s_id[ad[k]], idb + (uint32_t)(((k / 3) * 3 + bb) * 3 + (k % 3)));
        }
      }
    } else {
      for (int aa = 0; aa < ks; ++aa) {
        const int iy = iry + aa - h;
        if (iy < y0 || iy >= y1) continue;
        const float dy = sv.y - (float)iy;
        const float dxy = dx2 + dy * dy;
        const int lrow = (iy - y0) * D;
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
          } else if (db <= cur + 8u) {
            const float dist = pitch * sqrtf(d2);
            const float dmin = pitch * sqrtf(__uint_as_float(cur));
            if (dist == dmin && dist < trunc)
              atomicMin(&s_id[lrow + iz], idb + (uint32_t)((aa * ks + bb) * ks + cc));
          }
        }
      }
    }
    return cand;
  };

  // The first kTileThreads * kTileKeep records stay in registers over both passes (all loads
  // in flight at once: ONE memory round trip); a more crowded tile streams the rest again.
  float4 rv[kTileKeep];
  int rb[kTileKeep];
  unsigned keep_cand = 0u;  // bit u: record u touched a minimum in pass 1
#pragma unroll
  for (int u = 0; u < kTileKeep; ++u) fetch(u * kTileThreads + (int)threadIdx.x, rv[u], rb[u]);
  stamp(1);
#pragma unroll
  for (int u = 0; u < kTileKeep; ++u) {
    if (rb[u] >= 0 && !mine(rv[u])) rb[u] = -1;
    if (rb[u] >= 0 && visit(1, rv[u], rb[u])) keep_cand |= 1u << u;
  }
  for (int base = kTileThreads * kTileKeep; base < T; base += kTileThreads * kTileR) {
    float4 xv[kTileR];
    int xb[kTileR];
#pragma unroll
    for (int u = 0; u < kTileR; ++u) fetch(base + u * kTileThreads + (int)threadIdx.x, xv[u], xb[u]);
#pragma unroll
    for (int u = 0; u < kTileR; ++u)
      if (xb[u] >= 0 && mine(xv[u])) visit(1, xv[u], xb[u]);
  }
  __syncthreads();
  stamp(2);
#pragma unroll
  for (int u = 0; u < kTileKeep; ++u)
    if ((keep_cand >> u) & 1u) visit(2, rv[u], rb[u]);
  for (int base = kTileThreads * kTileKeep; base < T; base += kTileThreads * kTileR) {
    float4 xv[kTileR];
    int xb[kTileR];
#pragma unroll
    for (int u = 0; u < kTileR; ++u) fetch(base + u * kTileThreads + (int)threadIdx.x, xv[u], xb[u]);
#pragma unroll
    for (int u = 0; u < kTileR; ++u)
      if (xb[u] >= 0 && mine(xv[u])) visit(2, xv[u], xb[u]);
  }
  __syncthreads();
  stamp(3);
  // epilogue: winners out (coalesced 8 B/lane) + max raw inside weight of this tile
  // (truncated_distance_function.py:198-204: -1 where no winner, + offset, clamp at 0)
  const float offset = other ? 0.0f : a.sdf_offset;
  unsigned long long *Wg = a.W + (int64_t)g * D * D * D + ((int64_t)x * D + y0) * D;
  float wmax = 0.0f;
  for (int i0 = threadIdx.x; i0 < nvox; i0 += kTileThreads * 4) {
    uint32_t lo[4];
    float sd[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int i = i0 + u * kTileThreads;
      lo[u] = i < nvox ? s_id[i] : kNoCand;  // set only where pitch*sqrt(min d2) < trunc
      sd[u] = lo[u] != kNoCand ? a.pts4[lo[u] / (uint32_t)K].w : -1.0f;
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int i = i0 + u * kTileThreads;
      if (i >= nvox) continue;
      const float dist = lo[u] != kNoCand ? pitch * sqrtf(__uint_as_float(s_dist[i])) : trunc;
      Wg[i] = ((unsigned long long)__float_as_uint(dist) << 32) | lo[u];
      float w = sd[u] + offset;
      w = w < 0.0f ? 0.0f : w;
      wmax = fmaxf(wmax, w);
    }
  }
  wmax = mf::wave_max(wmax);
  if ((threadIdx.x & 63) == 0) s_max[threadIdx.x >> 6] = wmax;
  __syncthreads();
  if (threadIdx.x == 0) {
    float m = s_max[0];
#pragma unroll
    for (int i = 1; i < kTileThreads / 64; ++i) m = fmaxf(m, s_max[i]);
    if (m > 0.0f) atomicMax(&a.Mbits[g], __float_as_uint(m));  // m >= 0: uint order == float order
  }
  stamp(4);
}

__global__ __launch_bounds__(kTileThreads) void k_icc_tile(IccArgs a, int hmax) {
  const int ks = min(ksize_of(a.thr, a.pitch[blockIdx.y >> 1]), 2 * hmax + 1);  // block-uniform
  if (ks == 3)
    icc_tile_body<3>(a, 3, hmax);
  else
    icc_tile_body<0>(a, ks, hmax);
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
constexpr int kAccRep = 16;                       // replicas of the collision-limb accumulators

__global__ __launch_bounds__(kAccThreads) void k_icc_accum(IccArgs a, int K_v1) {
  __shared__ float s_rows[kAccThreads / 16][kNumOwn + 1];  // 16-lane row sums (+1: bank spread)
  // collision moments as 2^44 fixed point split in three 20-bit limbs held in 32-bit LDS
  // words: <= 1024 adds per block can never overflow a limb, so plain NON-returning
  // ds_add_u32 suffice (no carries).  Integer addition is associative: the result is
  // independent of the order of the atomics (bitwise reproducible), and 32-bit LDS atomics
  // are ~10x cheaper than the 64-bit ones.  All colliding voxels of a block add into the same
  // 36 words per other object, and same-address LDS atomics serialise (measured: the crowded
  // blocks spent 3-5 us here) -> kAccRep replicas selected by lane, in different banks (odd
  // stride), summed at the end: integer sums, still order-independent.
  MF_DYN_LDS(uint32_t, s_lim);  // [3 limbs][kAccRep][lim_stride]
  const int lim_stride = a.max_ns * 12 + 1;
  const int lim_words = kAccRep * lim_stride;
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
  for (int i = threadIdx.x; i < 3 * lim_words; i += kAccThreads) s_lim[i] = 0u;
  // this object's two grids have been consumed by k_icc_tile: empty their bins for the next k_icc_bin
  if (blockIdx.x == 0)
    for (int i = threadIdx.x; i < 2 * a.nbins; i += kAccThreads) a.bin_cnt[(int64_t)2 * o * a.nbins + i] = 0u;
  const float pitch = a.pitch[o];
  // candidate ids are point * K + offset with this grid's own kernel size (K_v1: round-1 front end)
  const int ks_o = ksize_of(a.thr, pitch);
  const int K = K_v1 > 0 ? K_v1 : ks_o * ks_o * ks_o;
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
            const int idx = (int)(threadIdx.x & (kAccRep - 1)) * lim_stride + 12 * e + 4 * d + c;
            atomicAdd(&s_lim[idx], (uint32_t)(x & 0xfffff));
            atomicAdd(&s_lim[lim_words + idx], (uint32_t)((x >> 20) & 0xfffff));
            atomicAdd(&s_lim[2 * lim_words + idx], (uint32_t)(int32_t)(x >> 40));  // two's complement
          }
        }
      }
    }
  }
  // fixed-order block reduction: every component is summed over each 16-lane row on DPP (4 VALU
  // steps, no LDS), the 32 row sums go through LDS, one lane per component adds them in order.
  stamp(2);
#pragma unroll
  for (int i = 0; i < kNumOwn; ++i) {
    const float r = mf::row16_sum(acc[i]);
    if ((threadIdx.x & 15) == 0) s_rows[threadIdx.x >> 4][i] = r;
  }
  __syncthreads();
  if (threadIdx.x < kNumOwn) {
    float sacc = 0.0f;
#pragma unroll
    for (int r = 0; r < kAccThreads / 16; ++r) sacc += s_rows[r][threadIdx.x];
    a.part[((int64_t)o * gridDim.x + blockIdx.x) * kNumOwn + threadIdx.x] = sacc;
  }
  stamp(3);
  // collision partials of this block (the barrier above orders the LDS atomics)
  float *po = a.oth + ((int64_t)o * gridDim.x + blockIdx.x) * a.max_ns * 12;
  for (int i = threadIdx.x; i < a.max_ns * 12; i += kAccThreads) {
    long long l0 = 0, l1 = 0, l2 = 0;
#pragma unroll
    for (int r = 0; r < kAccRep; ++r) {
      l0 += (long long)s_lim[r * lim_stride + i];
      l1 += (long long)s_lim[lim_words + r * lim_stride + i];
      l2 += (long long)(int32_t)s_lim[2 * lim_words + r * lim_stride + i];
    }
    const long long x = (l2 << 40) + (l1 << 20) + l0;
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
  int64_t W, M, Rt, bound, St, part, oth, step, meta, tab, bin_cnt, bin_cap, bin_base, rec, total;
  int NB, n_tab, nbins;
};

int ksize_host(float thr) {
  // Upper bound of the per-grid kernel size ksize_of(thr, pitch): the float32 quotient
  // (thr * pitch) / pitch is within 2 ulp of thr (exactly thr for thr = 2, the link's default).
  int ks = (int)ceilf(thr * 1.000001f);
  if (ks % 2 == 0) ks += 1;
  return ks;
}

WsLayout ws_layout(const mfIccBatch *b) {
  WsLayout l;
  const int O = b->n_objects, S = b->n_scenes, D = b->dim, max_ns = b->max_scene_objects;
  const int64_t V = (int64_t)D * D * D;
  l.NB = (int)((V + kVoxPerBlock - 1) / kVoxPerBlock);
  l.nbins = D + 2 * (ksize_host(b->voxel_threshold) / 2);
  // every (target, source) pair of a scene in chunks of kBinChunk points:
  // sum_pairs ceil(P_j / chunk) <= max_ns * n_points / chunk + O * max_ns
  l.n_tab = (int)(((int64_t)max_ns * b->n_points + kBinChunk - 1) / kBinChunk) + O * max_ns;
  int64_t off = 0;
  l.W = off; off = align256(off + 2 * O * V * 8);
  l.M = off; off = align256(off + 2 * O * 4);
  l.Rt = off; off = align256(off + O * 12 * 4);
  l.bound = off; off = align256(off + O * 4 * 4);
  l.St = off; off = align256(off + S * 4);
  l.part = off; off = align256(off + (int64_t)O * l.NB * kNumOwn * 4);
  l.oth = off; off = align256(off + (int64_t)O * l.NB * kMaxSceneObjects * 12 * 4);
  l.step = off; off = align256(off + S * 4);
  l.meta = off; off = align256(off + (int64_t)O * 16);
  l.tab = off; off = align256(off + (int64_t)l.n_tab * 16);
  l.bin_cnt = off; off = align256(off + (int64_t)2 * O * l.nbins * 4);
  l.bin_cap = off; off = align256(off + (int64_t)2 * O * 4);
  l.bin_base = off; off = align256(off + (int64_t)2 * O * 8);
  // a grid's bins hold <= (its source points) records each: sum over grids of a scene
  // = Ns * P_scene <= max_ns * n_points, times nbins planes
  l.rec = off; off = align256(off + (int64_t)l.nbins * max_ns * b->n_points * 16);
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
  const WsLayout l = ws_layout(b);
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
  a.tab = (int4 *)(p + l.tab);
  a.n_tab = l.n_tab;
  a.nbins = l.nbins;
  a.bin_cnt = (uint32_t *)(p + l.bin_cnt);
  a.bin_cap = (int32_t *)(p + l.bin_cap);
  a.bin_base = (int64_t *)(p + l.bin_base);
  a.rec = (float4 *)(p + l.rec);
  return a;
}

// MF_ICC_IMPL=1 keeps round 1's scan-everything front end (k_icc_tdf) for A/B measurements
int icc_impl() {
  static const int impl = getenv("MF_ICC_IMPL") ? atoi(getenv("MF_ICC_IMPL")) : 2;
  return impl;
}

void launch_front(const IccArgs &a, int ks, int SX, hipStream_t stream) {
  const int D = a.D;
  if (icc_impl() == 1) {
    const dim3 g1((D + SX - 1) / SX, 2 * a.O);
    const size_t lds1 = (size_t)SX * D * D * 2 * sizeof(uint32_t);
    if (ks == 3)
      hipLaunchKernelGGL(k_icc_tdf<3>, g1, dim3(kTdfThreads), lds1, stream, a, ks, SX);
    else
      hipLaunchKernelGGL(k_icc_tdf<0>, g1, dim3(kTdfThreads), lds1, stream, a, ks, SX);
    return;
  }
  const int hmax = (a.nbins - D) / 2;
  hipLaunchKernelGGL(k_icc_bin, dim3(a.n_tab), dim3(kBinThreads), 0, stream, a, hmax);
  const size_t lds = (size_t)((D + kTileStripes - 1) / kTileStripes) * D * 2 * sizeof(uint32_t);
  hipLaunchKernelGGL(k_icc_tile, dim3(D * kTileStripes, 2 * a.O), dim3(kTileThreads), lds, stream, a,
                     hmax);
}

void launch_iteration(const IccArgs &a, int ks, int SX, int NB, int max_ns, int mode, float *q,
                      float *t, float *adam_m, float *adam_v, float alpha_q, float alpha_t,
                      int adam_step, float *loss, float *gq, float *gt, float *traj, int it,
                      hipStream_t stream) {
  launch_front(a, ks, SX, stream);
  const size_t lds2 = (size_t)3 * kAccRep * (max_ns * 12 + 1) * sizeof(uint32_t);  // <= 74 KB
  hipLaunchKernelGGL(k_icc_accum, dim3(NB, a.O), dim3(kAccThreads), lds2, stream, a,
                     icc_impl() == 1 ? ks * ks * ks : 0);
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

static bool icc_batch_ok(const mfIccBatch *b) {
  return b && b->n_objects > 0 && b->n_scenes > 0 && b->dim > 0 && b->dim <= 64 &&
         b->n_points >= 0 && b->max_scene_objects > 0 && b->max_scene_objects <= kMaxSceneObjects &&
         b->voxel_threshold > 0.0f && ksize_host(b->voxel_threshold) <= 7 &&
         (double)b->n_points * 343.0 < 4294967295.0 && b->n_points < (1 << 27);
}

extern "C" int64_t mf_icc_workspace_bytes(const mfIccBatch *batch) {
  if (!icc_batch_ok(batch)) return -1;
  return ws_layout(batch).total;
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
  // static + dynamic LDS above 64 KB is opt-in (per device, thread-safe: mf::allow_big_lds)
  if (int e = mf::allow_big_lds((const void *)k_icc_tdf<3>, 64 * 1024)) return e;
  if (int e = mf::allow_big_lds((const void *)k_icc_tdf<0>, 64 * 1024)) return e;
  if (int e = mf::allow_big_lds((const void *)k_icc_accum, 80 * 1024)) return e;
  return mf::allow_big_lds((const void *)k_icc_step, 150 * 1024);
}

static int icc_validate(const mfIccBatch *b) {
  if (int e = icc_prepare_kernels()) return e;
  if (!icc_batch_ok(b) ||
      (size_t)b->max_scene_objects * (ws_layout(b).NB * kNumOwn + kNumOwn + 12 + 96) * 4 > 150 * 1024) {
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
  if (q && t) hipLaunchKernelGGL(k_icc_pose, dim3((a.O + 63) / 64), dim3(64), 0, stream, a, q, t);
  // inside an iteration k_icc_accum empties the bins; this hook has no accum launch
  MF_TRY(hipMemsetAsync(a.bin_cnt, 0, sizeof(uint32_t) * 2 * a.O * a.nbins, stream));
  launch_front(a, ks, SX, stream);
  return mf::check_launch("mf_icc_launch_tdf");
}

extern "C" int mf_icc_prepare(const mfIccBatch *batch, void *ws, mfStream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (int e = icc_validate(batch)) return e;
  IccArgs a = make_args(batch, ws, batch->max_scene_objects);
  hipLaunchKernelGGL(k_icc_bound, dim3(a.O), dim3(256), 0, stream, a);
  hipLaunchKernelGGL(k_icc_scene_setup, dim3(a.S), dim3(256), 0, stream, a, 0);
  hipLaunchKernelGGL(k_icc_tables, dim3(1), dim3(256), 0, stream, a);
  return mf::check_launch("mf_icc_prepare");
}

extern "C" int mf_icc_loss_grad(const mfIccBatch *batch, const float *q, const float *t,
                                float *loss, float *gq, float *gt, void *ws,
                                mfStream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (int e = icc_validate(batch)) return e;
  const int max_ns = batch->max_scene_objects;
  IccArgs a = make_args(batch, ws, max_ns);
  const WsLayout l = ws_layout(batch);
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
  const WsLayout l = ws_layout(batch);
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
  int dev = 0;
  MF_TRY(hipGetDevice(&dev));
  key.v.push_back(((uint64_t)(uint32_t)dev << 32) | (uint32_t)max_ns);

  std::lock_guard<std::mutex> lock(g_graph_mu);
  auto itg = g_graphs.find(key);
  if (itg == g_graphs.end()) {
    hipGraph_t graph = nullptr;
    // The caller's stream may be the legacy NULL stream (torch's default), which cannot be
    // captured: record the graph on a private stream, replay it on the caller's.
    static std::map<int, hipStream_t> caps;  // one capture stream per device (under g_graph_mu)
    hipStream_t &cap = caps[dev];
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
