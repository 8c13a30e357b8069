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
//   k_icc_bin    one workgroup per (target grid, source object, chunk of 1024 points).  Inside
//                the loop it first applies the PREVIOUS iteration's optimiser step for its source
//                object (the reduced gradient is ~200 fixed-point words: every workgroup
//                recomputes the same bits, one designated workgroup per object stores them),
//                then transforms its points once and appends the survivors' voxel-frame
//                coordinates to the bin of their rounded x-plane.
//   k_icc_tile   grid (x-plane * y-stripe, 2*O): TDF of the "own" / "other" point set of object
//                o from bins x-h..x+h; (min distance, arg-min id) live in LDS as two 32-bit words
//                per voxel, resolved with two passes of 32-bit LDS atomics (64-bit LDS atomics
//                measured ~10x slower); epilogue stores the winners and the per-grid max of
//                the raw inside weight (integer atomicMax).
//   k_icc_accum  grid (block, O): per voxel pseudo-occupancy weights, max() with the
//                no-entry grid, sums of reward / penalty AND the pose-gradient moments.  The
//                loss gradient is linear in {1/S_t, 1/S_in, PN/S_in^2}, so moments are
//                accumulated per coefficient and combined later -- no second pass over the
//                grids once the global sums are known.  Block sums are added as 64-bit fixed
//                point with global integer atomics: exact, order-independent.
//   k_icc_step   (once, after the last iteration; and for mf_icc_loss_grad) the same per-object
//                step as a kernel of its own: loss, chain rule to (q, t), chainer-Adam.
// Every reduction has a fixed order or is an integer sum: bitwise reproducible run to run.
// No host synchronisation anywhere.  MF_ICC_DEBUG is a tuning aid.
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <map>
#include <mutex>
#include <tuple>
#include <vector>

#include "mf_common.h"
#include "quat.h"

namespace {

// Tuning aids (per-phase time stamps, phase skipping: MF_ICC_DEBUG bit mask at run time) exist only in a
// build with -DMF_ICC_DEBUG_BUILD=1 (`make ICC_DEBUG=1`, tools/stamps_*.py); the production kernels carry
// none of their branches.
#ifndef MF_ICC_DEBUG_BUILD
#define MF_ICC_DEBUG_BUILD 0
#endif
#define MF_DBG(a_, bits_) (MF_ICC_DEBUG_BUILD != 0 && ((a_).dbg & (bits_)) != 0)
#if MF_ICC_DEBUG_BUILD
__device__ unsigned long long g_dbg_stamps[4096 * 8];  // (MF_ICC_DEBUG & 32)
#else
__device__ unsigned long long g_dbg_stamps[8];
#endif

constexpr int kAccThreads = 512;
constexpr int kVoxPerBlock = 1024;  // k_icc_accum: voxels per workgroup
constexpr int kNumOwn = 39;         // RN, S_in, PN + 3 x 12 gradient moments
constexpr uint32_t kNoCand = 0xffffffffu;
constexpr double kFixOth = 1099511627776.0;  // 2^40 fixed point: collision moments summed over blocks
constexpr double kFixOwn = 4294967296.0;     // 2^32: reward / penalty sums and own-gradient moments
constexpr int kNumF = 65;                    // single-pass path: 5 scene sums + 5 x 12 moments (below)
constexpr int kOwnSlots = kNumF + 1;         // accumulator words per object; the slot after the sums
                                             // counts non-finite block sums (-> NaN loss)
constexpr int kStateFloats = 21;             // q[4] t[3] m[7] v[7] of one object
// Objects per scene.  The single-pass path (k_icc_fused, {0,1} no-entry grids: what every caller of the reference
// passes) takes up to 128 (round 6): its LDS tables of a scene (R|t, offsets) are sized for that, and the collision
// moments -- 1664 bytes of LDS rows per other object and workgroup -- are reduced in chunks of kRows2Chunk objects (one
// chunk up to 64: the round-3 code path; > 64: the voxels' collision terms stay in registers and a second chunk reuses
// the rows).  The two-kernel path (k_icc_accum: any no-entry grid values) carries the objects a block meets as a
// 64-bit mask and stays at 64.  Scenes beyond 32 objects take > 64 KB of dynamic LDS (one workgroup per CU).
constexpr int kMaxSceneObjects = 128;
constexpr int kMaxSceneObjectsGeneral = 64;
constexpr int kRows2Chunk = 64;

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
  uint32_t *Mbits;        // [3 parities][2*O] per-grid max of the raw inside weight (float bits)
                          // (the two-launch path uses parities 0 and 1 of every three-parity array)
  int ne_binary;          // every grid_ne value is exactly 0 or 1 -> single-pass path (see k_icc_fused)
  float *Rt;              // [2][O][12]  R row-major, then t (the two-launch path uses copy 0)
  float *bound;           // [O][4]   model-frame bounding sphere
  float *St;              // [S]
  // reduced sums of one iteration, 64-bit fixed point, two parities (iteration k adds into
  // k & 1 while the step folded into k_icc_bin still reads (k - 1) & 1)
  long long *acc_own;     // [2][O][kOwnSlots]
  long long *acc_oth;     // [2][O][max_ns][12]  collision moments of grid o onto scene object e
  float *state_alt;       // [O][kStateFloats] second copy of (q, t, m, v): odd iterates
  int max_ns;
  int4 *meta;             // [O] {scene first object, scene end object, point begin, point end}
  // x-plane bins of the per-iteration point binning (k_icc_bin -> k_icc_tile)
  int4 *tab;              // [n_tab] {target object o, source object j, point begin, point end}; o < 0: unused
  int4 *tab2;             // [n_tab] {scene first object, objects in scene, scene, 1 = designated entry of j}
  int n_tab;
  int hmax;               // largest TDF half-kernel of the batch
  int nbins;              // COUNTER STRIDE of a grid: kHalves * (D + 2 hmax) real bins ((x-plane of the rounded x in
                          // [-hmax, D-1+hmax], y-half)) + 1: the last word counts the grid's OVERFLOW records
  uint32_t *bin_cnt;      // [2 parities][2*O][nbins] records in each bin: iteration k fills parity k & 1,
                          // the step side of k_icc_bin empties the other one for iteration k + 1
  // A (point, grid) pair lands in ONE plane (two bins when its rows straddle the halves), so a bin can
  // hold all P_g source points of its grid in the worst case -- but reserving that for every bin is
  // nbins x the records that can exist (3.3 GB for 32 objects x 3000 points).  A bin therefore gets
  // cap_g = max(kBinMinCap, P_g / kBinShare) slots; records beyond it go to the grid's overflow list
  // (2 P_g slots behind its bins), which EVERY tile of the grid scans with the bin-membership test
  // when its counter is non-zero.  Winners are exact minima with lowest-id ties and the sums are
  // fixed point: where a record is stored cannot change a bit of the result.
  int32_t *bin_cap;       // [2*O] capacity cap_g of each bin of grid g
  int32_t *bin_pts;       // [2*O] P_g = source points of grid g (overflow capacity = 2 P_g)
  int64_t *bin_base;      // [2*O] first record of grid g; bin b starts at base + b*cap_g, overflow at base + nreal*cap_g
  int bin_cap_force;      // > 0: every cap_g = this (MF_ICC_BIN_CAP: exercises the overflow path in tests)
  float4 *rec;            // records {fx, fy, fz, point id bits}: voxel-frame coordinates
  int dbg;                // tuning aid: MF_ICC_DEBUG bit mask (0 in production)
  int uniform_ns;         // > 0: every scene holds exactly this many objects (scene tables need no load)
  int xcd_order;          // k_icc_fused: XCD-contiguous logical workgroup order (see there)
};

using mf::quat_backward;
using mf::quat_to_R;

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
  const int V = a.D * a.D * a.D;
  const int64_t b0 = (int64_t)a.scene_off[s] * V, b1 = (int64_t)a.scene_off[s + 1] * V;
  float acc = 0.0f;
  for (int64_t i = b0 + threadIdx.x; i < b1; i += blockDim.x) acc += a.grid_target[i];
  acc = mf::wave_sum(acc);
  if ((threadIdx.x & 63) == 0) s_red[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) a.St[s] = (s_red[0] + s_red[1]) + (s_red[2] + s_red[3]);
}

// Start of a loss evaluation / refinement: R|t from (q, t) (both copies); empty accumulators, per-grid
// maxima and bin counters of every parity; traj[0] = the initial pose.  One workgroup per object.
constexpr int kParities = 2;  // iteration k fills parity k & 1 while the folded step reads (k - 1) & 1 and empties it
__global__ __launch_bounds__(256) void k_icc_pose(IccArgs a, const float *__restrict__ q,
                                                  const float *__restrict__ t, float *traj) {
  const int o = blockIdx.x;
  const int tid = threadIdx.x;
  if (tid == 0) {
    float R[9];
    quat_to_R(q + 4 * o, R);
    for (int cp = 0; cp < 2; ++cp) {
      float *Rt = a.Rt + ((int64_t)cp * a.O + o) * 12;
#pragma unroll
      for (int i = 0; i < 9; ++i) Rt[i] = R[i];
#pragma unroll
      for (int i = 0; i < 3; ++i) Rt[9 + i] = t[3 * o + i];
    }
    if (traj) {
#pragma unroll
      for (int i = 0; i < 4; ++i) traj[7 * o + i] = q[4 * o + i];
#pragma unroll
      for (int i = 0; i < 3; ++i) traj[7 * o + 4 + i] = t[3 * o + i];
    }
  }
  for (int par = 0; par < kParities; ++par) {
    if (tid < 2) a.Mbits[(int64_t)par * 2 * a.O + 2 * o + tid] = 0;
    for (int i = tid; i < 2 * a.nbins; i += blockDim.x) a.bin_cnt[((int64_t)par * 2 * a.O + 2 * o) * a.nbins + i] = 0u;
    for (int i = tid; i < kOwnSlots; i += blockDim.x) a.acc_own[((int64_t)par * a.O + o) * kOwnSlots + i] = 0;
    for (int i = tid; i < a.max_ns * 12; i += blockDim.x) a.acc_oth[((int64_t)par * a.O + o) * a.max_ns * 12 + i] = 0;
  }
}

// ---- front end: per-iteration x-plane binning + bin-fed TDF tiles ----------------------
// Round 1 let every one of the 32 plane workgroups of a grid re-scan all source points of the
// scene (32x read amplification, a dependent global load per work item).  Now every (source
// point, target grid) pair is transformed ONCE (k_icc_bin), the survivors' voxel-frame
// coordinates are appended to the bin of their rounded x-plane, and the tile of plane x reads
// only bins x-h..x+h (k_icc_tile): records arrive as coalesced 16 B loads, both LDS passes run
// on registers + LDS only.  Coordinates are computed with the oracle's expressions and (min,
// arg-min) are exact -> the same winners (verified bit-identical against round 1 on the GPU).
constexpr int kBinThreads = 256;
constexpr int kBinPPT = 4;                          // points per thread (2: 23.0 vs 23.1 us/iteration, twice the redundant steps)
constexpr int kBinChunk = kBinThreads * kBinPPT;    // points per workgroup
constexpr int kHalves = 2;                          // y-halves of a plane: rows [0, D/2), [D/2, D)
constexpr int kMaxBins = kHalves * (64 + 8);        // D <= 64, ks <= 7, + one margin plane per side 
constexpr int kBinShare = 8;                        // a bin holds 1/8 of its grid's source points ...
constexpr int kBinMinCap = 64;                      // ... at least this many, the rest overflows

__host__ __device__ inline int bin_cap_of(int P, int force) {
  int c = force > 0 ? force : max(kBinMinCap, (P + kBinShare - 1) / kBinShare);
  return min(max(c, 1), max(P, 1));
}

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
      const int nreal = a.nbins - 1;
      a.bin_cap[2 * o] = bin_cap_of(p_own, a.bin_cap_force);
      a.bin_pts[2 * o] = p_own;
      a.bin_base[2 * o] = rec_off;
      rec_off += (int64_t)nreal * a.bin_cap[2 * o] + 2 * (int64_t)p_own;
      a.bin_cap[2 * o + 1] = bin_cap_of(p_all - p_own, a.bin_cap_force);
      a.bin_pts[2 * o + 1] = p_all - p_own;
      a.bin_base[2 * o + 1] = rec_off;
      rec_off += (int64_t)nreal * a.bin_cap[2 * o + 1] + 2 * (int64_t)(p_all - p_own);
      for (int j = ja; j < jb; ++j) {
        const int p0 = a.obj_off[j], p1 = a.obj_off[j + 1];
        // the first chunk of the pair (j, j) is the designated entry of object j: it stores the
        // optimiser step folded into k_icc_bin (exists even for an object without points)
        for (int c = p0; c < p1 || (c == p0 && j == o); c += kBinChunk)
          if (tab_off < a.n_tab) {
            a.tab[tab_off] = make_int4(o, j, c, min(c + kBinChunk, p1));
            a.tab2[tab_off] = make_int4(ja, jb - ja, sc, (j == o && c == p0) ? 1 : 0);
            ++tab_off;
          }
      }
    }
    s_tab_base[0] = tab_off;
  }
  __syncthreads();
  for (int i = s_tab_base[0] + threadIdx.x; i < a.n_tab; i += blockDim.x) a.tab[i] = make_int4(-1, -1, 0, 0);
  for (int i = threadIdx.x; i < kParities * 2 * a.O * a.nbins; i += blockDim.x) a.bin_cnt[i] = 0u;
}

// ---- the optimiser step of ONE object from the reduced sums of an iteration ------------
// (iterative_collision_check_link.py:91-98 loss; chain rule through transformation_matrix /
// quaternion_matrix.py:36-78; chainer.optimizers.Adam v7 in float32).  A pure function of global
// memory: every workgroup that needs object j's next pose evaluates it and gets the same bits.
// `sv` = 52 sums gathered by the caller (LDS or registers): [0..2] RN, S_in, PN of the scene,
// [3..38] the 3 x 12 own-gradient moments of j, [39..50] collision moments onto j, [51] != 0 if
// any block sum of the scene was not finite.
constexpr int kStepSums = 52;

struct IccStepArgs {
  int mode;        // 0: none (bin reads a.Rt), 1: Adam step + outputs, 2: gradients only (k_icc_step)
  int fused;       // the sums come from k_icc_fused (monomials in 1/M_own, 1/M_oth) instead of k_icc_accum
  int par;         // parity of the accumulators / per-grid maxima to read
  int cpar;        // parity of the bin counters this launch fills
  int it;          // iteration whose pose is produced (traj row; its loss goes to losses[it - 1])
  float aq, at;    // alpha_t of chainer's Adam for this step (evaluated in double on the host)
  const float *q_in, *t_in, *m_in, *v_in;  // state before the step
  float *q_out, *t_out, *m_out, *v_out;    // state after it (may alias the inputs)
  float *loss_out;                         // [S] or NULL
  float *gq_out, *gt_out;                  // mode 2
  float *traj;                             // [n_iter][O][7] or NULL
};

// The calling workgroup (NT lanes) gathers the kStepSums sums of object j into s_sum.  Every
// accumulator word is fetched by its own lane -- ONE memory round trip (a lane walking the
// scene's objects serially costs a dependent load per object: measured 9 us at 8 objects) --
// staged in LDS, then summed in object order.  s_raw: >= (16 * max_ns + kNumOwn) 64-bit words.
// Contains two barriers: call it from uniform control flow.
constexpr int kStepRawWords = 20 * 64 + 60;  // 64-bit words: the single-pass path stages 20 Ns + 60 FLOATS in them
static_assert(20 * kMaxSceneObjects + 60 <= 2 * kStepRawWords, "staged floats of the single-pass step");
static_assert(16 * kMaxSceneObjectsGeneral + 36 <= kStepRawWords, "staged words of the two-kernel path's step");

template <int NT>
__device__ __forceinline__ void icc_step_gather(const IccArgs &a, int par, int j, int ja, int Ns,
                                                long long *s_raw, float *s_sum) {
  const long long *own = a.acc_own + (int64_t)par * a.O * kOwnSlots;
  const long long *oth = a.acc_oth + (int64_t)par * a.O * a.max_ns * 12;
  // items: [0, 4 Ns): own slots {RN, S_in, PN, non-finite count} of every scene object;
  // [4 Ns, 16 Ns): the 12 collision moments onto j from every scene object's grid;
  // [16 Ns, 16 Ns + 36): the own-gradient moments of j
  const int n_items = 16 * Ns + (kNumOwn - 3);
  for (int i = threadIdx.x; i < n_items; i += NT) {
    long long x;
    if (i < 4 * Ns) {
      const int jo = i >> 2, l = i & 3;
      x = own[(int64_t)(ja + jo) * kOwnSlots + (l < 3 ? l : kNumOwn)];
    } else if (i < 16 * Ns) {
      const int k = i - 4 * Ns, jo = k / 12, c = k - 12 * jo;
      x = oth[((int64_t)(ja + jo) * a.max_ns + (j - ja)) * 12 + c];
    } else {
      x = own[(int64_t)j * kOwnSlots + 3 + (i - 16 * Ns)];
    }
    s_raw[i] = x;
  }
  __syncthreads();
  if (threadIdx.x < kStepSums) {
    const int l = threadIdx.x;
    float r;
    if (l < 3) {  // scene sums, objects in order
      r = 0.0f;
      for (int jo = 0; jo < Ns; ++jo) r += (float)((double)s_raw[4 * jo + l] * (1.0 / kFixOwn));
    } else if (l < kNumOwn) {
      r = (float)((double)s_raw[16 * Ns + (l - 3)] * (1.0 / kFixOwn));
    } else if (l < kNumOwn + 12) {  // exact integer sum over the scene's grids
      long long x = 0;
      for (int jo = 0; jo < Ns; ++jo) x += s_raw[4 * Ns + 12 * jo + (l - kNumOwn)];
      r = (float)((double)x * (1.0 / kFixOth));
    } else {
      long long bad = 0;
      for (int jo = 0; jo < Ns; ++jo) bad |= s_raw[4 * jo + 3];
      r = bad != 0 ? 1.0f : 0.0f;
    }
    s_sum[l] = r;
  }
  __syncthreads();
}

// The same for the single-pass path (k_icc_fused): the accumulators hold the monomial sums, the
// per-grid maxima M_own / M_oth give a = 1/M_own, b = 1/M_oth (b = 0 where the "other" grid is
// empty or absent: iterative_collision_check_link.py:62-63,82), and the lanes form the sums the
// step expects (see the table above k_icc_fused).  Staged words, all converted to float by the lane that
// fetched them (fixed point -> float, M -> 1/M: the conversions and the IEEE reciprocals run in parallel):
//   sA[8 jo + l]   per scene object jo: {5 scene sums, non-finite flag, a = 1/M_own, b = 1/M_oth}
//   sB[12 jo + c]  the 12 collision moments onto object j from the grid of scene object jo
//   sC[i]          the 5 x 12 own-gradient moments of j
__device__ __forceinline__ float fused_item_scene(const long long *own, const uint32_t *Mb, int obj, int l, int Ns) {
  if (l < 5) return (float)((double)own[(int64_t)obj * kOwnSlots + l] * (1.0 / kFixOwn));
  if (l == 5) return own[(int64_t)obj * kOwnSlots + kNumF] != 0 ? 1.0f : 0.0f;
  const float M = __uint_as_float(Mb[2 * obj + (l - 6)]);  // a = 1/M_own, b = 1/M_oth (b = 0 where the "other" grid is empty)
  return l == 6 ? 1.0f / M : ((Ns > 1 && M != 0.0f) ? 1.0f / M : 0.0f);
}
__device__ __forceinline__ float fused_item_oth(const long long *oth, int grid_obj, int max_ns, int jj, int c) {
  return (float)((double)oth[((int64_t)grid_obj * max_ns + jj) * 12 + c] * (1.0 / kFixOth));
}
__device__ __forceinline__ float fused_item_own(const long long *own, int obj, int i) {
  return (float)((double)own[(int64_t)obj * kOwnSlots + 5 + i] * (1.0 / kFixOwn));
}
// sum l (< kStepSums) of scene-local object jj from the staged words
__device__ __forceinline__ float fused_sum(const int l, const int Ns, const int jj, const float *sA, const float *sB,
                                           const float *sC) {
  auto a_of = [&](int jo) { return sA[8 * jo + 6]; };
  auto b_of = [&](int jo) { return sA[8 * jo + 7]; };
  float r = 0.0f;
  if (l == 0) {  // RN
#pragma unroll 8
    for (int jo = 0; jo < Ns; ++jo) r += sA[8 * jo + 0] - a_of(jo) * sA[8 * jo + 1];
  } else if (l == 1) {  // S_in
#pragma unroll 8
    for (int jo = 0; jo < Ns; ++jo) r += a_of(jo) * sA[8 * jo + 2];
  } else if (l == 2) {  // PN
#pragma unroll 8
    for (int jo = 0; jo < Ns; ++jo) r += a_of(jo) * (sA[8 * jo + 3] + b_of(jo) * sA[8 * jo + 4]);
  } else if (l < 15) {  // reward moments
    const int c = l - 3;
    r = sC[c] - a_of(jj) * sC[12 + c];
  } else if (l < 27) {  // penalty numerator moments
    const int c = l - 15;
    r = a_of(jj) * (sC[24 + c] + b_of(jj) * sC[36 + c]);
  } else if (l < 39) {  // penalty denominator moments
    r = a_of(jj) * sC[48 + (l - 27)];
  } else if (l < 51) {  // collision moments of every grid of the scene onto j
#pragma unroll 8
    for (int jo = 0; jo < Ns; ++jo) r += (a_of(jo) * b_of(jo)) * sB[12 * jo + (l - 39)];
  } else {
#pragma unroll 8
    for (int jo = 0; jo < Ns; ++jo) r = sA[8 * jo + 5] != 0.0f ? 1.0f : r;
  }
  return r;
}

template <int NT>
__device__ __forceinline__ void icc_step_gather_fused(const IccArgs &a, int par, int j, int ja, int Ns,
                                                      long long *s_raw, float *s_sum) {
  const long long *own = a.acc_own + (int64_t)par * a.O * kOwnSlots;
  const long long *oth = a.acc_oth + (int64_t)par * a.O * a.max_ns * 12;
  const uint32_t *Mb = a.Mbits + (int64_t)par * 2 * a.O;
  // items: [0, 8 Ns) sA; [8 Ns, 20 Ns) sB; [20 Ns, 20 Ns + 60) sC
  const int n_items = 20 * Ns + 60;
  float *s_f = reinterpret_cast<float *>(s_raw);
  for (int i0 = 0; i0 < n_items; i0 += NT) {
    const int i = i0 + (int)threadIdx.x;
    float fv = 0.0f;
    if (i < n_items) {
      if (i < 8 * Ns) {
        fv = fused_item_scene(own, Mb, ja + (i >> 3), i & 7, Ns);
      } else if (i < 20 * Ns) {
        const int k = i - 8 * Ns, jo = k / 12, c = k - 12 * jo;
        fv = fused_item_oth(oth, ja + jo, a.max_ns, j - ja, c);
      } else {
        fv = fused_item_own(own, j, i - 20 * Ns);
      }
    }
    if (i < n_items) s_f[i] = fv;
  }
  __syncthreads();
  if constexpr (NT >= 256) {
    // the three scene sums are loops over the scene's objects: one wave each, the other 49 sums on a fourth (as 52
    // lanes of one wave the loops ran one after the other)
    const int w = threadIdx.x >> 6, ln = threadIdx.x & 63;
    const int l = w < 3 ? (ln == 0 ? w : -1) : (w == 3 && ln < kStepSums - 3 ? 3 + ln : -1);
    if (l >= 0) s_sum[l] = fused_sum(l, Ns, j - ja, s_f, s_f + 8 * Ns, s_f + 20 * Ns);
  } else {
    if (threadIdx.x < kStepSums) s_sum[threadIdx.x] = fused_sum((int)threadIdx.x, Ns, j - ja, s_f, s_f + 8 * Ns, s_f + 20 * Ns);
  }
  __syncthreads();
}

// The optimiser step of one object spread over the 16 lanes `c` of a lane group (sv: the gathered sums; st: (q, t,
// m, v) before the step): the twelve gradient components, the seven Adam updates (chainer.optimizers.Adam v7 rule in
// float32) and the rotation are evaluated by different lanes -- a third of the dependent instruction chain of one
// lane doing all of it (that chain was 1.5 us of every iteration; the one-lane form is gone, the bits are its).
// xg: kStepLaneWords floats of LDS scratch owned by the group; the state after the step is left
// in xg[12 ..] (q, t, m, v); every lane returns R|t and the loss, and the gradients in (gq, gt).
// Call from wave-uniform control flow (contains wave-level LDS hand-overs).
constexpr int kStepLaneWords = 12 + kStateFloats;
__device__ __forceinline__ void icc_step_lanes(const float *sv, float S_t, const float *st, const IccStepArgs &sp,
                                               const int c, float *xg, float *Rt_out, float &loss, float *gq,
                                               float *gt) {
  const float RN = sv[0], S_in = sv[1], PN = sv[2];
  const float reward = RN / S_t, penalty = PN / S_in;
  loss = sv[51] != 0.0f ? __builtin_nanf("") : penalty - reward;
  const float c0 = 1.0f / S_t, c1 = 1.0f / S_in, c2 = PN / (S_in * S_in);
  if (c < 12) xg[c] = ((c0 * sv[3 + c] - c1 * sv[15 + c]) + c2 * sv[27 + c]) - c1 * sv[39 + c];
  __builtin_amdgcn_wave_barrier();
  float gR[9];
#pragma unroll
  for (int d = 0; d < 3; ++d)
#pragma unroll
    for (int cc = 0; cc < 4; ++cc) {
      const float G = xg[4 * d + cc];
      if (cc < 3) gR[3 * d + cc] = G; else gt[d] = G;
    }
  float qq[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) qq[i] = st[i];
  quat_backward(qq, gR, gq);
  if (sv[51] != 0.0f) {
#pragma unroll
    for (int i = 0; i < 4; ++i) gq[i] = loss;
#pragma unroll
    for (int i = 0; i < 3; ++i) gt[i] = loss;
  }
  float *so = xg + 12;
  if (c < 7) {
    float th = st[c];
    if (sp.mode == 1) {
      // chainer.optimizers.Adam (v7) update rule in float32, parameter c
      const float omb1 = (float)(1.0 - 0.9), omb2 = (float)(1.0 - 0.999), eps = 1e-8f;
      const float gi = c == 0 ? gq[0] : c == 1 ? gq[1] : c == 2 ? gq[2] : c == 3 ? gq[3] : c == 4 ? gt[0] : c == 5 ? gt[1] : gt[2];
      float mm = st[7 + c], vv = st[14 + c];
      mm += omb1 * (gi - mm);
      vv += omb2 * (gi * gi - vv);
      so[7 + c] = mm;
      so[14 + c] = vv;
      const float upd = (c < 4 ? sp.aq : sp.at) * mm / (sqrtf(vv) + eps);
      th -= upd;
    }
    so[c] = th;
  }
  __builtin_amdgcn_wave_barrier();
  float qn[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) qn[i] = so[i];
  quat_to_R(qn, Rt_out);
#pragma unroll
  for (int i = 0; i < 3; ++i) Rt_out[9 + i] = so[4 + i];
}

// launch 1: one workgroup per (target grid, source object, chunk of <= 1024 points)
__global__ __launch_bounds__(kBinThreads) void k_icc_bin(IccArgs a, IccStepArgs sp) {
  __shared__ int s_cnt[kMaxBins], s_base[kMaxBins];
  __shared__ float s_sum[kStepSums], s_state[kStateFloats];
  __shared__ __attribute__((aligned(16))) float s_Rt12[16];
  __shared__ float s_x[kStepLaneWords];
  __shared__ long long s_raw[kStepRawWords];
  auto stamp = [&](int i) {  // tuning aid (MF_ICC_DEBUG & 32)
    if (MF_DBG(a, 32) && threadIdx.x == 0 && blockIdx.x < 1024)
      g_dbg_stamps[(3072 + blockIdx.x) * 8 + i] = wall_clock64();
  };
  stamp(0);
  // (batches of >= 32 objects: the same XCD-contiguous logical order as k_icc_fused -- the workgroups that bin for a
  // grid run on the XCD whose L2 its tiles will read the records from)
  int bi = blockIdx.x;
  if (a.xcd_order && (gridDim.x & 7) == 0) bi = (bi & 7) * (int)(gridDim.x >> 3) + (bi >> 3);
  const int4 e = a.tab[bi];
  const int o = e.x, j = e.y;
  if (o < 0) return;  // block-uniform
  const int D = a.D, nb = a.nbins, hmax = a.hmax;
  const int g = 2 * o + (j != o ? 1 : 0);
  // everything below depends on the table entries only: one memory round trip
  const int4 e2 = a.tab2[bi];  // {scene first object, objects in scene, scene, designated}
  float4 r0, r1, r2;
  float S_t = 1.0f;
  if (sp.mode == 0) {
    r0 = *reinterpret_cast<const float4 *>(a.Rt + 12 * j);
    r1 = *reinterpret_cast<const float4 *>(a.Rt + 12 * j + 4);
    r2 = *reinterpret_cast<const float4 *>(a.Rt + 12 * j + 8);
  } else {
    // its optimiser state; the reduced sums are gathered below, in the same round trip
    if (threadIdx.x >= 224 && threadIdx.x < 224 + kStateFloats) {
      const int i = threadIdx.x - 224;
      s_state[i] = i < 4 ? sp.q_in[4 * j + i] : i < 7 ? sp.t_in[3 * j + i - 4]
                   : i < 14 ? sp.m_in[7 * j + i - 7] : sp.v_in[7 * j + i - 14];
    }
    S_t = a.St[e2.z];
  }
  const float4 bnd = *reinterpret_cast<const float4 *>(a.bound + 4 * j);
  const float pitch = a.pitch[o];
  const float ox = a.origin[3 * o], oy = a.origin[3 * o + 1], oz = a.origin[3 * o + 2];
  const int cap = a.bin_cap[g];
  const int ovf_cap = 2 * a.bin_pts[g];
  const int64_t base_g = a.bin_base[g];
  const int nbr = nb - 1;  // real bins; counter nbr = the grid's overflow records
  float4 m[kBinPPT];
#pragma unroll
  for (int u = 0; u < kBinPPT; ++u) {
    const int p = e.z + u * kBinThreads + (int)threadIdx.x;
    m[u] = p < e.w ? a.pts4[p] : make_float4(0, 0, 0, 0);
  }
  for (int i = threadIdx.x; i < nbr; i += kBinThreads) s_cnt[i] = 0;
  if (sp.mode != 0) {
    // the previous iteration's reduced sums of object j (fixed point)
    if (sp.fused)
      icc_step_gather_fused<kBinThreads>(a, sp.par, j, e2.x, e2.y, s_raw, s_sum);
    else
      icc_step_gather<kBinThreads>(a, sp.par, j, e2.x, e2.y, s_raw, s_sum);
    stamp(4);
    // The step on the first 16 lanes (icc_step_lanes: gradient components, Adam updates and rotation on different
    // lanes), R|t to the others through LDS.  (Rounds 2-4: every lane of every wave evaluated the serial step --
    // 850 dependent instructions, 1.5 us of the critical path and of every SIMD's issue time.)
    if (threadIdx.x < 16) {
      float Rt[12], loss, gq[4], gt[3];
      icc_step_lanes(s_sum, S_t, s_state, sp, (int)threadIdx.x, s_x, Rt, loss, gq, gt);
      if (threadIdx.x < 12) {
        float rv = Rt[0];
#pragma unroll
        for (int i = 1; i < 12; ++i) rv = (int)threadIdx.x == i ? Rt[i] : rv;
        s_Rt12[threadIdx.x] = rv;
      }
      if (threadIdx.x == 0) s_Rt12[12] = loss;
    }
    __syncthreads();
    stamp(5);
    r0 = *reinterpret_cast<const float4 *>(&s_Rt12[0]);
    r1 = *reinterpret_cast<const float4 *>(&s_Rt12[4]);
    r2 = *reinterpret_cast<const float4 *>(&s_Rt12[8]);
    const float *st_new = s_x + 12;
    const float loss = s_Rt12[12];
    const float Rt[12] = {r0.x, r0.y, r0.z, r0.w, r1.x, r1.y, r1.z, r1.w, r2.x, r2.y, r2.z, r2.w};
    if (e2.w != 0 && threadIdx.x == 0) {  // the designated workgroup of object j stores the step
#pragma unroll
      for (int i = 0; i < 4; ++i) sp.q_out[4 * j + i] = st_new[i];
#pragma unroll
      for (int i = 0; i < 3; ++i) sp.t_out[3 * j + i] = st_new[4 + i];
#pragma unroll
      for (int i = 0; i < 7; ++i) { sp.m_out[7 * j + i] = st_new[7 + i]; sp.v_out[7 * j + i] = st_new[14 + i]; }
#pragma unroll
      for (int i = 0; i < 12; ++i) a.Rt[12 * j + i] = Rt[i];
      if (sp.traj) {
        float *tr = sp.traj + ((int64_t)sp.it * a.O + j) * 7;
#pragma unroll
        for (int i = 0; i < 7; ++i) tr[i] = st_new[i];
      }
      if (sp.loss_out && j == e2.x) sp.loss_out[e2.z] = loss;
    }
    if (e2.w != 0) {
      // ... and empties, all lanes together (one lane storing ~300 words in a row measured 4 us):
      // this object's accumulators and maxima of the parity the coming iteration adds into, and
      // the bins of its two grids that the NEXT iteration fills
      if (threadIdx.x < 2) a.Mbits[(int64_t)(sp.par ^ 1) * 2 * a.O + 2 * j + threadIdx.x] = 0;
      long long *own = a.acc_own + ((int64_t)(sp.par ^ 1) * a.O + j) * kOwnSlots;
      for (int i = threadIdx.x; i < kOwnSlots; i += kBinThreads) own[i] = 0;
      long long *oth = a.acc_oth + ((int64_t)(sp.par ^ 1) * a.O + j) * a.max_ns * 12;
      for (int i = threadIdx.x; i < a.max_ns * 12; i += kBinThreads) oth[i] = 0;
      for (int i = threadIdx.x; i < 2 * nb; i += kBinThreads)
        a.bin_cnt[((int64_t)(sp.cpar ^ 1) * 2 * a.O + 2 * j) * nb + i] = 0u;
    }
  }
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
  // A survivor goes to the bin of its rounded x-plane, in the y-half (or both halves) its
  // ks rows touch: the tile of a half then finds exactly its own records, dense.
  float fx[kBinPPT], fy[kBinPPT], fz[kBinPPT];
  int bin[kBinPPT][kHalves], slot[kBinPPT][kHalves];
  const int Dh = (D + 1) / 2;
#pragma unroll
  for (int u = 0; u < kBinPPT; ++u) {
    const int p = e.z + u * kBinThreads + (int)threadIdx.x;
#pragma unroll
    for (int hf = 0; hf < kHalves; ++hf) { bin[u][hf] = -1; slot[u][hf] = 0; }
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
        const int plane = (int)rx + hmax;  // in [0, D + 2 hmax)
        const int iry = (int)ry;
        if (iry - h < Dh) {
          bin[u][0] = plane * kHalves;
          slot[u][0] = atomicAdd(&s_cnt[bin[u][0]], 1);
        }
        if (iry + h >= Dh) {
          bin[u][1] = plane * kHalves + 1;
          slot[u][1] = atomicAdd(&s_cnt[bin[u][1]], 1);
        }
      }
    }
  }
  __syncthreads();
  stamp(1);
  for (int i = threadIdx.x; i < nbr; i += kBinThreads) {
    const int c = s_cnt[i];
    s_base[i] = c > 0 ? (int)atomicAdd(&a.bin_cnt[((int64_t)sp.cpar * 2 * a.O + g) * nb + i], (uint32_t)c) : 0;
  }
  __syncthreads();
  stamp(2);
#pragma unroll
  for (int u = 0; u < kBinPPT; ++u) {
    const int p = e.z + u * kBinThreads + (int)threadIdx.x;
#pragma unroll
    for (int hf = 0; hf < kHalves; ++hf) {
      if (bin[u][hf] < 0) continue;
      const int idx = s_base[bin[u][hf]] + slot[u][hf];
      const float4 r = make_float4(fx[u], fy[u], fz[u], __uint_as_float((uint32_t)p));
      if (idx < cap) {
        a.rec[base_g + (int64_t)bin[u][hf] * cap + idx] = r;
      } else {  // bin full: the grid's overflow list (its tiles find the record by the membership test)
        const uint32_t k = atomicAdd(&a.bin_cnt[((int64_t)sp.cpar * 2 * a.O + g) * nb + nbr], 1u);
        if ((int)k < ovf_cap) a.rec[base_g + (int64_t)nbr * cap + k] = r;  // (k < 2 P_g always: a point adds <= 2 records)
      }
    }
  }
  stamp(3);
}

// launch 2: TDF of one half of an x-plane (rows [y0, y1)) of one grid, fed from the bins of
// planes x-h..x+h of that half.
//  pass 1 works on SQUARED distances in voxel units (no sqrt, no pitch): 32-bit atomicMin of
//         the d2 bits behind a batched peek.  dist = pitch*sqrt(d2) is monotone in d2.  A lane
//         remembers, per record, WHICH of its candidates were within a few ulp of the minimum
//         it saw (9-bit mask): minima only decrease, so no other candidate can end up minimal.
//  pass 2 re-derives, only for those candidates (~ln n of the n candidates of a voxel), the EXACT
//         float distance and, where it equals the exact minimum and is < truncation, takes
//         atomicMin of the candidate id: the same winners as the oracle (lowest id among
//         equal ROUNDED distances).
// Measured alternatives (profiles/): a single pass with a 64-bit (d2, id) LDS
// atomicMin per improving candidate is slower (ds_min_u64 processes lanes serially); splitting
// a crowded plane over 4 workgroups that each scan all its records is slower (every stripe
// pays for every record, and 2048 workgroups no longer fit the chip at once) -- hence the
// halves are made by the binning kernel, where it costs one extra append for 1 point in 8.
constexpr int kTileThreads = 512;
constexpr int kTileKeep = 4;  // records per lane kept in registers over both passes
constexpr int kTileR = 4;     // records in flight per lane beyond those
constexpr int kFusedKeepOwn = 2, kFusedKeepOth = 4;  // k_icc_fused: kept records per lane and grid
constexpr int kPad = 2;  // margin cells of its LDS tile on every side (ks = 3: candidates reach 2 cells out)
// LDS words of one (dist | id) array of the single-pass kernel's padded half-plane tile
__host__ __device__ constexpr int fused_tile_words(int D) { return ((D + 1) / 2 + 2 * kPad) * (D + 2 * kPad); }

template <int KS>
__device__ __forceinline__ void icc_tile_body(const IccArgs &a, const int ks_rt, const int par) {
  MF_DYN_LDS(uint32_t, s_tile);  // dist[rows*D], id[rows*D]
  __shared__ float s_max[kTileThreads / 64];
  const int ks = KS > 0 ? KS : ks_rt;
  const int h = ks / 2, K = ks * ks * ks;
  const int D = a.D, nb = a.nbins, hmax = a.hmax;
  const int g = blockIdx.y, o = g >> 1, other = g & 1;
  const int x = blockIdx.x / kHalves, half = blockIdx.x % kHalves;
  const int Dh = (D + 1) / 2;
  const int y0 = half * Dh, y1 = half == 0 ? Dh : D;
  const int nvox = (y1 - y0) * D;
  uint32_t *s_dist = s_tile, *s_id = s_tile + Dh * D;
  // independent loads: the <= 7 bin counts of this tile, capacity, offset
  int c[8];
  c[0] = 0;
  const int cap = a.bin_cap[g];
  const int64_t base_g = a.bin_base[g];
  const float pitch = a.pitch[o];
  const int bin0 = x + hmax - h;  // plane x - h
  const int nbr = nb - 1;
  const int nov = min((int)a.bin_cnt[((int64_t)par * 2 * a.O + g) * nb + nbr], 2 * a.bin_pts[g]);
#pragma unroll
  for (int b = 0; b < 7; ++b) {
    int n = 0;
    if (b < ks) n = min((int)a.bin_cnt[((int64_t)par * 2 * a.O + g) * nb + (bin0 + b) * kHalves + half], cap);
    c[b + 1] = c[b] + n;
  }
  const int T = c[7] + nov;  // the tile's bins, then the grid's overflow list (filtered by fetch)
  const int wg = blockIdx.y * gridDim.x + blockIdx.x;
  auto stamp = [&](int i) {  // tuning aid (MF_ICC_DEBUG & 32)
    if (MF_DBG(a, 32) && threadIdx.x == 0 && wg < 2048) g_dbg_stamps[wg * 8 + i] = wall_clock64();
  };
  stamp(0);
  if (MF_DBG(a, 32) && threadIdx.x == 0 && wg < 2048) g_dbg_stamps[wg * 8 + 6] = (unsigned long long)T;
  const float trunc = a.thr * pitch;
  for (int i = threadIdx.x; i < nvox; i += kTileThreads) { s_dist[i] = 0x7f800000u; s_id[i] = kNoCand; }
  __syncthreads();
  const float d2_hi = a.thr * a.thr * 1.00002f;  // conservative inclusion; exact test in pass 2
  const float d2_in = a.thr * a.thr * 0.999f;    // certainly inside the truncation radius
  const float4 *recs = a.rec + base_g;
  const float fxp = (float)x;

  // record i of this tile's concatenated bins -> (plane offset b, record); rb < 0: none
  auto fetch = [&](const int i, float4 &rv, int &rb) {
    rb = -1;
    if (i >= T) return;
    if (i >= c[7]) {  // overflow record: belongs to this tile iff its plane is in x-h..x+h and its rows touch the half
      rv = recs[(int64_t)nbr * cap + (i - c[7])];
      const int pl = (int)roundf(rv.x) - (x - h), iry_ = (int)roundf(rv.y);
      const bool in_half = half == 0 ? (iry_ - h < Dh) : (iry_ + h >= Dh);
      rb = (pl >= 0 && pl < ks && in_half) ? pl : -1;
      return;
    }
    int b = 0;
#pragma unroll
    for (int k = 1; k < 7; ++k) b += (k < ks && i >= c[k]) ? 1 : 0;
    int cb = 0;
#pragma unroll
    for (int k = 1; k < 7; ++k) cb = (k == b) ? c[k] : cb;
    rb = b;
    rv = recs[(int64_t)((bin0 + b) * kHalves + half) * cap + (i - cb)];
  };
  // exact tie-break of ONE candidate against the final minimum of its voxel
  auto settle = [&](const int ad, const uint32_t db, const uint32_t cid) {
    const uint32_t cur = s_dist[ad];
    if (db <= cur + 8u) {  // within a few ulp of the minimal d2
      // dist == dmin is certain for equal bits; dist < trunc is certain well inside the
      // truncation radius (pitch*sqrt(d2) <= 0.9995 thr pitch (1 + 2^-22) < trunc)
      bool win = db == cur && __uint_as_float(db) < d2_in;
      if (!win) {
        const float dist = pitch * sqrtf(__uint_as_float(db));
        const float dmin = pitch * sqrtf(__uint_as_float(cur));
        win = dist == dmin && dist < trunc;
      }
      if (win) atomicMin(&s_id[ad], cid);
    }
  };
  // One record against its ks x ks (y, z) candidates in plane x.  pass 1 returns the mask of
  // candidates that may still win (KS == 3: one bit per candidate; else bit 0 = "any"); pass 2
  // visits the candidates of `mask`.  All peeks of a record are issued together, then the
  // non-returning atomics.  A peek may be stale (another lane lowered the voxel meanwhile):
  // values only decrease, so a stale peek only lets MORE candidates through.
  auto visit = [&](const int pass, const float4 sv, const int rb, const unsigned mask) -> unsigned {
    const int iry = (int)roundf(sv.y), irz = (int)roundf(sv.z);
    const uint32_t idb = __float_as_uint(sv.w) * (uint32_t)K;
    const int bb = ks - 1 - rb;  // x offset of plane x inside this point's neighbourhood
    const float dx = sv.x - fxp;
    const float dx2 = dx * dx;
    unsigned out = 0u;
    if constexpr (KS == 3) {
      if (pass == 1) {
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
          if (db[k] <= cur[k]) atomicMin(&s_dist[ad[k]], db[k]);
          if (db[k] <= cur[k] + 8u) out |= 1u << k;
        }
      } else {
#pragma unroll
        for (int k = 0; k < 9; ++k) {
          if (!((mask >> k) & 1u)) continue;  // in range and near-minimal when pass 1 saw it
          const int aa = k / 3, cc = k % 3;
          const int iy = iry + aa - 1, iz = irz + cc - 1;
          const float dy = sv.y - (float)iy, dz = sv.z - (float)iz;
          const float d2 = (dx2 + dy * dy) + dz * dz;
          settle((iy - y0) * D + iz, __float_as_uint(d2), idb + (uint32_t)((aa * 3 + bb) * 3 + cc));
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
          if (pass == 1) {
            const uint32_t cur = s_dist[lrow + iz];
            if (db <= cur) atomicMin(&s_dist[lrow + iz], db);
            if (db <= cur + 8u) out = 1u;
          } else {
            settle(lrow + iz, db, idb + (uint32_t)((aa * ks + bb) * ks + cc));
          }
        }
      }
    }
    return out;
  };

  // The first kTileThreads * kTileKeep records stay in registers over both passes (all loads
  // in flight at once: ONE memory round trip); a more crowded tile streams the rest again.
  float4 rv[kTileKeep];
  int rb[kTileKeep];
  unsigned long long keep = 0ull;  // 9 bits per kept record: candidates that may still win
#pragma unroll
  for (int u = 0; u < kTileKeep; ++u) fetch(u * kTileThreads + (int)threadIdx.x, rv[u], rb[u]);
  stamp(1);
#pragma unroll
  for (int u = 0; u < kTileKeep; ++u)
    if (rb[u] >= 0) keep |= (unsigned long long)visit(1, rv[u], rb[u], 0u) << (9 * u);
  for (int base = kTileThreads * kTileKeep; base < T; base += kTileThreads * kTileR) {
    float4 xv[kTileR];
    int xb[kTileR];
#pragma unroll
    for (int u = 0; u < kTileR; ++u) fetch(base + u * kTileThreads + (int)threadIdx.x, xv[u], xb[u]);
#pragma unroll
    for (int u = 0; u < kTileR; ++u)
      if (xb[u] >= 0) visit(1, xv[u], xb[u], 0u);
  }
  __syncthreads();
  stamp(2);
#pragma unroll
  for (int u = 0; u < kTileKeep; ++u) {
    const unsigned m9 = (unsigned)(keep >> (9 * u)) & 0x1ffu;
    if (m9 != 0u) visit(2, rv[u], rb[u], m9);
  }
  for (int base = kTileThreads * kTileKeep; base < T; base += kTileThreads * kTileR) {
    float4 xv[kTileR];
    int xb[kTileR];
#pragma unroll
    for (int u = 0; u < kTileR; ++u) fetch(base + u * kTileThreads + (int)threadIdx.x, xv[u], xb[u]);
#pragma unroll
    for (int u = 0; u < kTileR; ++u) {
      if (xb[u] < 0) continue;
      // streamed records carry no mask: every in-range candidate within the window is examined
      if constexpr (KS == 3) {
        const int iry = (int)roundf(xv[u].y), irz = (int)roundf(xv[u].z);
        unsigned m9 = 0u;
#pragma unroll
        for (int k = 0; k < 9; ++k) {
          const int iy = iry + k / 3 - 1, iz = irz + k % 3 - 1;
          const float dxs = xv[u].x - fxp, dy = xv[u].y - (float)iy, dz = xv[u].z - (float)iz;
          const float d2 = (dxs * dxs + dy * dy) + dz * dz;
          if (iy >= y0 && iy < y1 && iz >= 0 && iz < D && d2 < d2_hi) m9 |= 1u << k;
        }
        if (m9 != 0u) visit(2, xv[u], xb[u], m9);
      } else {
        visit(2, xv[u], xb[u], 1u);
      }
    }
  }
  __syncthreads();
  stamp(3);
  // epilogue: winners out (coalesced 8 B/lane) + max raw inside weight of this tile
  // (truncated_distance_function.py:198-204: -1 where no winner, + offset, clamp at 0)
  const float offset = other ? 0.0f : a.sdf_offset;
  unsigned long long *Wg = a.W + (int64_t)g * D * D * D + ((int64_t)x * D + y0) * D;
  float wmax = 0.0f;
  for (int i0 = threadIdx.x; i0 < nvox; i0 += kTileThreads * 2) {
    uint32_t lo[2];
    float sd[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int i = i0 + u * kTileThreads;
      lo[u] = i < nvox ? s_id[i] : kNoCand;  // set only where pitch*sqrt(min d2) < trunc
      sd[u] = lo[u] != kNoCand ? a.pts4[lo[u] / (uint32_t)K].w : -1.0f;
    }
#pragma unroll
    for (int u = 0; u < 2; ++u) {
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
    if (m > 0.0f) atomicMax(&a.Mbits[(int64_t)par * 2 * a.O + g], __float_as_uint(m));  // m >= 0: uint order == float order
  }
  stamp(4);
}

__global__ __launch_bounds__(kTileThreads) void k_icc_tile(IccArgs a, int par) {
  const int ks = min(ksize_of(a.thr, a.pitch[blockIdx.y >> 1]), 2 * a.hmax + 1);  // block-uniform
  if (ks == 3)
    icc_tile_body<3>(a, 3, par);
  else
    icc_tile_body<0>(a, ks, par);
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

// the same with reciprocal multiplies (k_icc_fused's voxel phase; see there)
__device__ __forceinline__ void world_frac_r(const float *Rt, const float4 m, float ox, float oy,
                                             float oz, float inv_pitch, int ix, int iy, int iz,
                                             float &ux, float &uy, float &uz, bool &ok) {
  const float wx = ((Rt[0] * m.x + Rt[1] * m.y) + Rt[2] * m.z) + Rt[9];
  const float wy = ((Rt[3] * m.x + Rt[4] * m.y) + Rt[5] * m.z) + Rt[10];
  const float wz = ((Rt[6] * m.x + Rt[7] * m.y) + Rt[8] * m.z) + Rt[11];
  const float dx = (wx - ox) * inv_pitch - (float)ix;
  const float dy = (wy - oy) * inv_pitch - (float)iy;
  const float dz = (wz - oz) * inv_pitch - (float)iz;
  const float n2 = (dx * dx + dy * dy) + dz * dz;
  ok = n2 > 0.0f;  // truncated_distance_function.py:141
  const float rn = __frsqrt_rn(n2);
  ux = dx * rn; uy = dy * rn; uz = dz * rn;
}

constexpr int kVPT = kVoxPerBlock / kAccThreads;  // voxels per thread

__global__ __launch_bounds__(kAccThreads) void k_icc_accum(IccArgs a, int par) {
  __shared__ float s_rows[kAccThreads / 16][kNumOwn + 1];  // 16-lane row sums (+1: bank spread)
  // Collision moments (gradient of this grid's penalty onto ANOTHER object's pose): each lane
  // keeps the 12 moments of its colliding voxels in registers; after the voxel loop the block
  // reduces them per other object in a fixed order (DPP row sums + ordered row adds), exactly
  // like its own moments.  (Round 1 / early round 2 pushed every colliding voxel through 36
  // fixed-point LDS atomics behind float64 conversions: ~600 instructions per colliding voxel,
  // 3-5 us in the crowded blocks.)
  MF_DYN_LDS(float, s_rows2);   // [max_ns][kAccThreads / 16][12 + 1] row sums per other object
  __shared__ unsigned long long s_emask;  // scene objects some voxel of this block collides with (<= 64 per scene)
  __shared__ float s_Rt[kMaxSceneObjectsGeneral][12];
  __shared__ int s_off[kMaxSceneObjectsGeneral + 1];
  const int o = blockIdx.y;
  const int wg2 = 2048 + blockIdx.y * gridDim.x + blockIdx.x;
  auto stamp = [&](int i) {
    if (MF_DBG(a, 32) && threadIdx.x == 0 && wg2 < 4096) g_dbg_stamps[wg2 * 8 + i] = wall_clock64();
  };
  stamp(0);
  const int D = a.D, V = D * D * D;
  const int4 meta = a.meta[o];
  const int ja = meta.x, jb = meta.y;
  const int Ns = jb - ja;
  // all independent loads first: scene tables, scalars, and this thread's voxels
  for (int i = threadIdx.x; i < Ns * 12; i += blockDim.x) s_Rt[i / 12][i % 12] = a.Rt[12 * ja + i];  // Ns up to 64: 768 words
  if (threadIdx.x <= Ns) s_off[threadIdx.x] = a.obj_off[ja + threadIdx.x];
  if (threadIdx.x == 0) s_emask = 0ull;
  const float pitch = a.pitch[o];
  // candidate ids are point * K + offset with this grid's own kernel size
  const int ks_o = ksize_of(a.thr, pitch);
  const int K = ks_o * ks_o * ks_o;
  const float ox = a.origin[3 * o], oy = a.origin[3 * o + 1], oz = a.origin[3 * o + 2];
  const float M_own = __uint_as_float(a.Mbits[(int64_t)par * 2 * a.O + 2 * o]);
  const float M_oth = __uint_as_float(a.Mbits[(int64_t)par * 2 * a.O + 2 * o + 1]);
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
  int ecol[kVPT];
  float cv[kVPT][12];
#pragma unroll
  for (int it = 0; it < kVPT; ++it) ecol[it] = -1;

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
        ecol[it] = e;
#pragma unroll
        for (int d = 0; d < 3; ++d) {
          const float sB = u[d] * B;
          cv[it][4 * d + 0] = sB * m.x;
          cv[it][4 * d + 1] = sB * m.y;
          cv[it][4 * d + 2] = sB * m.z;
          cv[it][4 * d + 3] = sB;
        }
      }
    }
  }
#pragma unroll
  for (int it = 0; it < kVPT; ++it)
    if (ecol[it] >= 0) atomicOr(&s_emask, 1ull << ecol[it]);
  // fixed-order block reduction: every component is summed over each 16-lane row on DPP (4 VALU
  // steps, no LDS), the 32 row sums go through LDS, one lane per component adds them in order.
  stamp(2);
#pragma unroll
  for (int i = 0; i < kNumOwn; ++i) {
    const float r = mf::row16_sum(acc[i]);
    if ((threadIdx.x & 15) == 0) s_rows[threadIdx.x >> 4][i] = r;
  }
  __syncthreads();
  // The block sums join the object's accumulators as 64-bit fixed point: integer atomics are
  // exact and order-independent, so the iteration's reduced sums (~200 words per scene) are
  // bitwise reproducible and the optimiser step needs no reduction pass of its own.
  long long *own = a.acc_own + ((int64_t)par * a.O + o) * kOwnSlots;
  if (threadIdx.x < kNumOwn) {
    float sacc = 0.0f;
#pragma unroll
    for (int r = 0; r < kAccThreads / 16; ++r) sacc += s_rows[r][threadIdx.x];
    if (isfinite(sacc)) {
      const long long x = __double2ll_rn((double)sacc * kFixOwn);
      if (x != 0) atomicAdd(reinterpret_cast<unsigned long long *>(own + threadIdx.x), (unsigned long long)x);
    } else {
      atomicAdd(reinterpret_cast<unsigned long long *>(own + kNumOwn), 1ull);  // -> NaN loss
    }
  }
  stamp(3);
  // collision moments: row sums of every other object this block collides with (block-uniform
  // loop over the set bits, no barrier inside), ONE barrier, then 12 lanes per object add the
  // rows in order
  long long *po = a.acc_oth + ((int64_t)par * a.O + o) * a.max_ns * 12;
  const unsigned long long em0 = s_emask;  // complete: every atomicOr precedes the barrier above
  constexpr int kRows = kAccThreads / 16;
  for (unsigned long long em = em0; em != 0ull; em &= em - 1ull) {
    const int e = __ffsll((long long)em) - 1;
#pragma unroll
    for (int c = 0; c < 12; ++c) {
      float v = 0.0f;
#pragma unroll
      for (int it = 0; it < kVPT; ++it) v += ecol[it] == e ? cv[it][c] : 0.0f;
      const float r = mf::row16_sum(v);
      if ((threadIdx.x & 15) == 0) s_rows2[(e * kRows + (threadIdx.x >> 4)) * 13 + c] = r;
    }
  }
  if (em0 == 0ull) return;  // block-uniform
  __syncthreads();
  for (int i = threadIdx.x; i < a.max_ns * 12; i += kAccThreads) {
    const int e = i / 12, c = i - 12 * e;
    if (!((em0 >> e) & 1ull)) continue;
    float sacc = 0.0f;
#pragma unroll
    for (int r = 0; r < kRows; ++r) sacc += s_rows2[(e * kRows + r) * 13 + c];
    const long long x = isfinite(sacc) ? __double2ll_rn((double)sacc * kFixOth) : 0;
    if (x != 0) atomicAdd(reinterpret_cast<unsigned long long *>(po + i), (unsigned long long)x);
  }
}

// ---- single-pass path: TDF tiles + weights / sums / moments in ONE kernel -------------
// k_icc_tile -> W -> k_icc_accum exists only because the weights are normalised by the per-grid
// maximum M = max(inside weight), known once every tile of the grid is done.  Every caller of
// the reference passes {0,1} no-entry grids (bool cast to float32:
// check_iterative_collision_check_link.py:36-38, collision_based_pose_refinement.py:162), and
// for those maximum(no-entry, other) is a selection, so the loss and its gradient are POLYNOMIAL
// in a = 1/M_own and b = 1/M_oth.  With gw = g*w (g = 1 - tdf/trunc, w = clamped inside weight),
// go, wo the same of the "other" grid, nb = [sdf + offset >= 0], ne in {0,1}:
//   RN   = sum nb*g*tg          - a   sum gw*tg                  (sums 0, 1)
//   S_in =                        a   sum gw                      (sum 2)
//   PN   =                        a   sum gw*ne + a b sum gw*(1-ne)*go*wo        (sums 3, 4)
//   own gradient moments (u = unit residual of the winner, m its model point; 12 each):
//     U0a: nb*tg/trunc, U0b: w*tg/trunc (coeff -a), U1a: w*ne/trunc (a), U1b: w*(1-ne)*go*wo/trunc
//     (a b), U2: w/trunc (a)
//   collision moments onto the other object e: u_o (x) {m_o,1} * wo*gw/trunc     (coeff a b)
// A workgroup = (object, x-plane, y-half) runs both TDFs of its voxels in LDS (own + other
// records), then one lane per voxel accumulates the 65 monomial sums; the step (icc_step_gather)
// applies a, b from the per-grid maxima.  The winners never leave LDS: no W round trip, no
// second launch, no dependent re-load of what the tile just computed.  Same arithmetic per
// voxel as k_icc_accum up to the association of the normaliser (tests: loss within 2e-5,
// step within 1e-5 of the oracle's).  Grids with other values take the two-kernel path.
// LDS of the voxel phase (k_icc_fused)
struct VoxLds {
  float rows[kTileThreads / 16][kNumF + 1];
  float max[2][kTileThreads / 64];
  // voxels with an own winner, compacted in voxel order: index, (no-entry, target), winner points
  uint16_t list[kTileThreads];
  float2 netg[kTileThreads];
  float4 mown[kTileThreads], moth[kTileThreads];
  int wcnt[kTileThreads / 64];
};
// static LDS of k_icc_fused (declared once in the kernel: the body is instantiated per kernel size)
// (MAXNS = 64: the kernel every scene of <= 64 objects runs, unchanged since round 3; 128: k_icc_fused_big)
template <int MAXNS>
struct FusedLds {
  VoxLds v;
  float Rt[MAXNS][12];
  int off[MAXNS + 1];
};

// Constants of one padded half-plane tile (k_icc_fused, kernel size 3).
struct Tile3 {
  int Wp, rows_p, y0;
  float fxp, pitch, trunc, d2_in;
  uint32_t in_bits, hi_bits;
};

// The two-pass (min, arg-min) of k_icc_tile on the LDS arrays of one grid, kernel size 3.  The tile
// carries a margin of kPad cells on every side: all nine (y, z) candidates of a record of this half's
// bins (rounded y in [y0 - 1, y1], z in [-1, D]) address cells of the padded tile, the ones
// outside the half land in margin cells nobody reads.  ks = 3 therefore needs NO predicate:
// pass 1 = nine fire-and-forget ds_min at constant offsets from one base address (a peek at
// the current minimum first, or range / radius tests per candidate, cost more instructions
// than the atomics they save: 19.9 -> 18.4 us without the peek alone), pass 2 = the nine
// FINAL minima in one batch of reads, the exact test only where this record is within a few
// ulp.  A minimum beyond the truncation radius simply finds no winner in pass 2.
// (sx, sy, sz) = voxel-frame coordinates of the point, pid its id, rb = its plane's offset in x-1 .. x+1.
__device__ __forceinline__ void icc_visit3(const int pass, uint32_t *dist, uint32_t *id, const Tile3 &tl,
                                           const float sx, const float sy, const float sz, const uint32_t pid,
                                           const int rb) {
  const int Wp = tl.Wp;
  const int iry = (int)roundf(sy), irz = (int)roundf(sz);
  const uint32_t idb = pid * 27u;
  const int bb = 2 - rb;
  const float dx = sx - tl.fxp;
  const float dx2 = dx * dx;
  // cell of candidate (aa, cc) = (0, 0): row iry - 1, column irz - 1; clamped so that a
  // corrupt record cannot leave the tile
  const int r0 = min(max(iry - 1 - tl.y0 + kPad, 0), tl.rows_p - 3);
  const int c0 = min(max(irz - 1 + kPad, 0), Wp - 3);
  const int cbase = r0 * Wp + c0;
  uint32_t db[9];
#pragma unroll
  for (int aa = 0; aa < 3; ++aa) {
    const float dy = sy - (float)(iry + aa - 1);
    const float dxy = dx2 + dy * dy;
#pragma unroll
    for (int cc = 0; cc < 3; ++cc) {
      const float dz = sz - (float)(irz + cc - 1);
      db[aa * 3 + cc] = __float_as_uint(dxy + dz * dz);
    }
  }
  if (pass == 1) {
#pragma unroll
    for (int k = 0; k < 9; ++k) atomicMin(&dist[cbase + (k / 3) * Wp + (k % 3)], db[k]);
  } else {
    uint32_t cur[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) cur[k] = dist[cbase + (k / 3) * Wp + (k % 3)];
    // fast: this record IS the minimum, certainly inside the truncation radius -> candidate
    // for the arg-min.  slow (rare): within a few ulp of the minimum or near the radius ->
    // the exact float test, in a ROLLED loop behind one branch that recomputes what it
    // needs.  (Inlined next to the fast path the compiler speculated both square roots into
    // every candidate: pass 2 took 4-5 us in every tile; unrolled behind the branch it was
    // still 900 instructions of code per record.)
    const uint32_t cid0 = idb + (uint32_t)(bb * 3);
    bool any_slow = false;
#pragma unroll
    for (int k = 0; k < 9; ++k) {
      const bool f = db[k] == cur[k] && db[k] < tl.in_bits;
      // (issuing it unconditionally with a neutral value instead: measured slower, 2.3 vs 1.6 us)
      if (f) atomicMin(&id[cbase + (k / 3) * Wp + (k % 3)], cid0 + (uint32_t)((k / 3) * 9 + (k % 3)));
      any_slow |= !f && db[k] <= min(cur[k] + 8u, tl.hi_bits);
    }
    if (any_slow) {
#pragma nounroll
      for (int k = 0; k < 9; ++k) {
        const int aa = k / 3, cc = k - 3 * aa;
        const float dy = sy - (float)(iry + aa - 1), dz = sz - (float)(irz + cc - 1);
        const uint32_t dbk = __float_as_uint((dx2 + dy * dy) + dz * dz);
        const int ad = cbase + aa * Wp + cc;
        const uint32_t curk = dist[ad];
        const bool f = dbk == curk && dbk < tl.in_bits;
        if (!f && dbk <= min(curk + 8u, tl.hi_bits)) {
          // candidate at squared distance bits dbk against the final minimum curk of its voxel
          bool win = dbk == curk && __uint_as_float(dbk) < tl.d2_in;
          if (!win) {
            const float dd = tl.pitch * sqrtf(__uint_as_float(dbk));
            const float dmin = tl.pitch * sqrtf(__uint_as_float(curk));
            win = dd == dmin && dd < tl.trunc;
          }
          if (win) atomicMin(&id[ad], cid0 + (uint32_t)(aa * 9 + cc));
        }
      }
    }
  }
}

// ---- voxel phase of a half-plane tile whose (min distance, arg-min) arrays are final.  Only a voxel
// WITH an own winner adds to any sum (without one g = 0 and w = 0), and those are the few voxels of
// the surface shell, scattered over most waves of the tile: compact them, so that ceil(n / 64) waves
// pay the arithmetic and the 65 row reductions instead of every wave the shell touches.  The maximum
// of the OTHER grid's weights needs every voxel with an other-winner: taken here in the
// voxel-per-lane layout, its gather is in flight during the compaction.
// V.rows and s_rows2 must be zero on entry (a wave writes only the sets / objects it meets).
struct TileGeom {
  int o, ja, Ns, x, y0, nvox, nvh, Wp, D, K;
  float pitch, trunc, ox, oy, oz;
};

template <bool BIG, class Stamp>
__device__ __forceinline__ void icc_voxel_phase(const IccArgs &a, const int par, const TileGeom &tg_, const float ne0,
                                                const float tg0, uint32_t *s_dist, uint32_t *s_id, float *s_rows2,
                                                VoxLds &Vx, const float (*s_Rt)[12], const int *s_off, Stamp stamp) {
  auto &s_rows = Vx.rows;
  auto &s_max = Vx.max;
  auto &s_list = Vx.list;
  auto &s_netg = Vx.netg;
  auto &s_mown = Vx.mown;
  auto &s_moth = Vx.moth;
  auto &s_wcnt = Vx.wcnt;
  const int o = tg_.o, ja = tg_.ja, Ns = tg_.Ns, x = tg_.x, y0 = tg_.y0, nvox = tg_.nvox, nvh = tg_.nvh, Wp = tg_.Wp,
            D = tg_.D, K = tg_.K;
  const float pitch = tg_.pitch, trunc = tg_.trunc, ox = tg_.ox, oy = tg_.oy, oz = tg_.oz;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  // voxel -> (row, column) without an integer divide: exact for vi < 1024, D <= 64
  const uint32_t rcpD = (65536u + (uint32_t)D - 1u) / (uint32_t)D;  // (scalar)
  const int my_r = (int)(((uint32_t)tid * rcpD) >> 16), my_c = tid - my_r * D;
  const int my_cell = tid < nvox ? (my_r + kPad) * Wp + (my_c + kPad) : 0;
  const uint32_t my_id = tid < nvox ? s_id[my_cell] : kNoCand;
  const uint32_t my_ido = tid < nvox ? s_id[nvh + my_cell] : kNoCand;
  // both winner gathers of this voxel in flight during the compaction; the lane that takes the
  // voxel reads them from LDS (no second dependent global round trip)
  const bool act = my_id != kNoCand;
  const float4 g_own = act ? a.pts4[my_id / (uint32_t)K] : make_float4(0, 0, 0, -1.0f);
  const float4 g_oth = my_ido != kNoCand ? a.pts4[my_ido / (uint32_t)K] : make_float4(0, 0, 0, -1.0f);
  const unsigned long long bal = __ballot(act);
  if (lane == 0) s_wcnt[wave] = __popcll(bal);
  __syncthreads();
  int before = 0, total = 0;
#pragma unroll
  for (int w = 0; w < kTileThreads / 64; ++w) {
    const int cw = s_wcnt[w];
    before += w < wave ? cw : 0;
    total += cw;
  }
  if (act) {
    const int slot = before + __popcll(bal & ((1ull << lane) - 1ull));
    s_list[slot] = (uint16_t)((my_r << 8) | my_c);
    s_mown[slot] = g_own;
    s_moth[slot] = g_oth;
    s_netg[slot] = make_float2(ne0, tg0);
  }
  __syncthreads();
  stamp(5);
  const float *Rt_o = s_Rt[o - ja];
  float wmax_own = 0.0f;
  float wmax_oth = fmaxf(g_oth.w + 0.0f, 0.0f);
  constexpr int kRows = kTileThreads / 16;
  const int n_rows = (total + 15) / 16;
  int ecol_keep = -1;  // (BIG only: the collision terms of the later chunks of a scene of > kRows2Chunk objects)
  float cv_keep[12];
  if constexpr (BIG) {
#pragma unroll
    for (int cc = 0; cc < 12; ++cc) cv_keep[cc] = 0.0f;
  }
  if ((tid & ~63) < total) {  // wave-uniform
    const bool live = tid < total;
    const int rc = live ? (int)s_list[tid] : 0;
    const int vr = rc >> 8, vc = rc & 255;
    const int pc = (vr + kPad) * Wp + (vc + kPad);
    const uint32_t lo = live ? s_id[pc] : kNoCand;
    const uint32_t lo_o = live ? s_id[nvh + pc] : kNoCand;
    const float4 m_own = live ? s_mown[tid] : make_float4(0, 0, 0, -1.0f);
    const float4 m_oth = live ? s_moth[tid] : make_float4(0, 0, 0, -1.0f);
    const float2 netg = s_netg[tid];
    const float ne = live ? netg.x : 0.0f, tg = live ? netg.y : 0.0f;
    const bool has = lo != kNoCand, has_o = lo_o != kNoCand;
    // Winners (arg-min) are exact; from here on the weights use reciprocal multiplies
    // (x * (1/trunc), x * (1/pitch), d * rsq(|d|^2)) instead of IEEE divides: <= 2 ulp per factor,
    // far inside the tolerance of the sums (which are re-associated anyway), and ~200 fewer
    // instructions on the one wave whose issue time is this phase.
    const float inv_trunc = 1.0f / trunc, inv_pitch = 1.0f / pitch;
    const float dist_o = has ? pitch * sqrtf(__uint_as_float(s_dist[pc])) : trunc;
    const float dist_k = has_o ? pitch * sqrtf(__uint_as_float(s_dist[nvh + pc])) : trunc;
    const int iy = y0 + vr, iz = vc;
    const float g = has ? fmaxf(1.0f - dist_o * inv_trunc, 0.0f) : 0.0f;  // 1 - tdf/trunc
    float w = m_own.w + a.sdf_offset;
    const bool neg = w < 0.0f;
    if (neg) w = 0.0f;
    const float go = has_o ? fmaxf(1.0f - dist_k * inv_trunc, 0.0f) : 0.0f;
    float wo = m_oth.w + 0.0f;
    if (wo < 0.0f) wo = 0.0f;
    if (live) wmax_own = w;
    const float gw = g * w;
    const float gwo = (1.0f - ne) * (go * wo);  // (1 - ne) * go * wo: the part that needs b
    const int row = tid >> 4;
    const bool row_lead = (tid & 15) == 0;
    {
      const float v5[5] = {live && !neg ? g * tg : 0.0f, live ? gw * tg : 0.0f, live ? gw : 0.0f,
                           live ? gw * ne : 0.0f, live ? gw * gwo : 0.0f};
      float r5[5];
#pragma unroll
      for (int k = 0; k < 5; ++k) r5[k] = mf::row16_sum(v5[k]);
      if (row_lead) {
#pragma unroll
        for (int k = 0; k < 5; ++k) s_rows[row][k] = r5[k];
      }
    }
    // own-gradient moments, set by set; a set no lane of the wave contributes to is skipped
    // (s_rows starts zeroed): target-free or no-entry-free regions drop 24 of the 60 reductions
    {
      float uu[3] = {0.0f, 0.0f, 0.0f};
      bool ok = false;
      if (live && has) {
        world_frac_r(Rt_o, m_own, ox, oy, oz, inv_pitch, x, iy, iz, uu[0], uu[1], uu[2], ok);
        if (!ok) uu[0] = uu[1] = uu[2] = 0.0f;
      }
      const float wt = ok ? w * inv_trunc : 0.0f;
      const float kk[5] = {ok && !neg ? tg * inv_trunc : 0.0f, wt * tg, wt * ne, wt * gwo, wt};
      const float mc[4] = {m_own.x, m_own.y, m_own.z, 1.0f};
#pragma unroll
      for (int sset = 0; sset < 5; ++sset) {
        if (__ballot(kk[sset] != 0.0f) == 0ull) continue;  // wave-uniform
        // the 12 chains in one block (independent DPP chains interleave: no wait-state nops),
        // one predicated burst of stores
        float r12[12];
#pragma unroll
        for (int d = 0; d < 3; ++d) {
          const float sc = uu[d] * kk[sset];
#pragma unroll
          for (int cc = 0; cc < 4; ++cc) r12[4 * d + cc] = mf::row16_sum(sc * mc[cc]);
        }
        if (row_lead) {
#pragma unroll
          for (int i = 0; i < 12; ++i) s_rows[row][5 + 12 * sset + i] = r12[i];
        }
      }
    }
    // collision term: gradient flows to the OTHER object's pose
    int ecol = -1;
    float cv[12];
    if (live && ne == 0.0f && has_o && go * wo > 0.0f && gw != 0.0f) {
      const int pp = (int)(lo_o / (uint32_t)K);
      int e = 0;  // scene object of the point: independent LDS reads, no dependent search loop
      for (int k = 1; k < Ns; ++k) e += pp >= s_off[k] ? 1 : 0;
      float ux, uy, uz;
      bool ok;
      world_frac_r(s_Rt[e], m_oth, ox, oy, oz, inv_pitch, x, iy, iz, ux, uy, uz, ok);
      const float B = wo * gw * inv_trunc;
      if (ok && isfinite(B)) {
        const float uu[3] = {ux, uy, uz};
        ecol = e;
#pragma unroll
        for (int d = 0; d < 3; ++d) {
          const float sB = uu[d] * B;
          cv[4 * d + 0] = sB * m_oth.x;
          cv[4 * d + 1] = sB * m_oth.y;
          cv[4 * d + 2] = sB * m_oth.z;
          cv[4 * d + 3] = sB;
        }
      }
    }
    // the 12 collision moments per other object some lane of this wave collides with (rows2
    // starts zeroed: a wave writes only the objects it meets); objects beyond the first chunk: below
    if (__ballot(ecol >= 0) != 0ull) {
      const int e1 = BIG ? min(Ns, kRows2Chunk) : Ns;
      for (int e = 0; e < e1; ++e) {
        if (__ballot(ecol == e) == 0ull) continue;  // wave-uniform
        float r12[12];
#pragma unroll
        for (int cc = 0; cc < 12; ++cc) r12[cc] = mf::row16_sum(ecol == e ? cv[cc] : 0.0f);
        if (row_lead) {
#pragma unroll
          for (int cc = 0; cc < 12; ++cc) s_rows2[(e * kRows + row) * 13 + cc] = r12[cc];
        }
      }
    }
    if constexpr (BIG) {  // kept for the later chunks of a scene of more than kRows2Chunk objects
      ecol_keep = ecol;
#pragma unroll
      for (int cc = 0; cc < 12; ++cc) cv_keep[cc] = ecol >= 0 ? cv[cc] : 0.0f;
    }
  }
  stamp(7);
  // per-grid maxima of the raw inside weights (the normalisers a, b of the step)
  wmax_own = mf::wave_max(wmax_own);
  wmax_oth = mf::wave_max(wmax_oth);
  if (lane == 0) { s_max[0][wave] = wmax_own; s_max[1][wave] = wmax_oth; }
  __syncthreads();
  if (tid >= kTileThreads - 2) {  // (lanes away from the ones that reduce the sums below)
    const int kd = tid - (kTileThreads - 2);
    float m = s_max[kd][0];
#pragma unroll
    for (int i = 1; i < kTileThreads / 64; ++i) m = fmaxf(m, s_max[kd][i]);
    if (m > 0.0f) atomicMax(&a.Mbits[(int64_t)par * 2 * a.O + 2 * o + kd], __float_as_uint(m));
  }
  if (total == 0) return;  // block-uniform: no own winner here, nothing to add
  // ONE reduction phase: lane k < 65 adds the rows of own sum k, the next 12 Ns lanes the rows of
  // a collision moment (zero rows where no wave met that object); fixed order, fixed point
  if (tid < kNumF) {
    long long *own = a.acc_own + ((int64_t)par * a.O + o) * kOwnSlots;
    float sacc = 0.0f;
    for (int r = 0; r < n_rows; ++r) sacc += s_rows[r][tid];
    if (isfinite(sacc)) {
      const long long xq = __double2ll_rn((double)sacc * kFixOwn);
      if (xq != 0) atomicAdd(reinterpret_cast<unsigned long long *>(own + tid), (unsigned long long)xq);
    } else {
      atomicAdd(reinterpret_cast<unsigned long long *>(own + kNumF), 1ull);  // -> NaN loss
    }
  } else {  // (one trip up to 37 scene objects; a 64-object scene takes two)
    long long *po = a.acc_oth + ((int64_t)par * a.O + o) * a.max_ns * 12;
    for (int i = tid - kNumF; i < 12 * (BIG ? min(Ns, kRows2Chunk) : Ns); i += kTileThreads - kNumF) {
      const int e = i / 12, cc = i - 12 * e;
      float sacc = 0.0f;
      for (int r = 0; r < n_rows; ++r) sacc += s_rows2[(e * kRows + r) * 13 + cc];
      const long long xq = isfinite(sacc) ? __double2ll_rn((double)sacc * kFixOth) : 0;
      if (xq != 0) atomicAdd(reinterpret_cast<unsigned long long *>(po + i), (unsigned long long)xq);
    }
  }
  // Scene objects kRows2Chunk .. Ns - 1 (a scene of more than 64 objects, block-uniform): the same row sums and the
  // same reduction on the SAME LDS rows, chunk by chunk -- zero the rows, the waves write the objects of the chunk
  // they met (their collision terms waited in registers), all lanes add the rows.  Fixed order, fixed point: what a
  // single pass over 1664 Ns bytes of rows would give, in 106 KB.
  if constexpr (BIG)
  for (int eb = kRows2Chunk; eb < Ns; eb += kRows2Chunk) {
    const int ne_ = min(Ns - eb, kRows2Chunk);
    __syncthreads();
    for (int i = tid; i < ne_ * kRows * 13; i += kTileThreads) s_rows2[i] = 0.0f;
    __syncthreads();
    if ((tid & ~63) < total && __ballot(ecol_keep >= eb && ecol_keep < eb + ne_) != 0ull) {  // wave-uniform
      const int row = tid >> 4;
      const bool row_lead = (tid & 15) == 0;
      for (int e = eb; e < eb + ne_; ++e) {
        if (__ballot(ecol_keep == e) == 0ull) continue;  // wave-uniform
        float r12[12];
#pragma unroll
        for (int cc = 0; cc < 12; ++cc) r12[cc] = mf::row16_sum(ecol_keep == e ? cv_keep[cc] : 0.0f);
        if (row_lead) {
#pragma unroll
          for (int cc = 0; cc < 12; ++cc) s_rows2[((e - eb) * kRows + row) * 13 + cc] = r12[cc];
        }
      }
    }
    __syncthreads();
    long long *po = a.acc_oth + ((int64_t)par * a.O + o) * a.max_ns * 12 + 12 * eb;
    for (int i = tid; i < 12 * ne_; i += kTileThreads) {
      const int e = i / 12, cc = i - 12 * e;
      float sacc = 0.0f;
      for (int r = 0; r < n_rows; ++r) sacc += s_rows2[(e * kRows + r) * 13 + cc];
      const long long xq = isfinite(sacc) ? __double2ll_rn((double)sacc * kFixOth) : 0;
      if (xq != 0) atomicAdd(reinterpret_cast<unsigned long long *>(po + i), (unsigned long long)xq);
    }
  }
}

template <int KS, int MAXNS>
__device__ __forceinline__ void icc_fused_body(const IccArgs &a, const int ks_rt, const int par, FusedLds<MAXNS> &L,
                                               const int o, const int tile_) {
  MF_DYN_LDS(uint32_t, s_tile);  // dist[2][nvh] | id[2][nvh] | rows2[max_ns][32][13] floats
  auto &s_rows = L.v.rows;
  auto &s_Rt = L.Rt;
  auto &s_off = L.off;
  const int ks = KS > 0 ? KS : ks_rt;
  const int h = ks / 2, K = ks * ks * ks;
  const int D = a.D, nb = a.nbins, hmax = a.hmax, V = D * D * D;
  const int x = tile_ / kHalves, half = tile_ % kHalves;
  const int Dh = (D + 1) / 2;
  const int y0 = half * Dh, y1 = half == 0 ? Dh : D;
  const int Wp = D + 2 * kPad, rows_p = Dh + 2 * kPad;  // padded tile (see icc_visit3)
  const int nvh = rows_p * Wp;          // LDS stride of one (dist | id) array
  const int nvox = (y1 - y0) * D;
  uint32_t *s_dist = s_tile, *s_id = s_tile + 2 * nvh;
  float *s_rows2 = reinterpret_cast<float *>(s_tile + 4 * nvh);
  const int4 meta = a.meta[o];
  const int ja = meta.x, Ns = meta.y - meta.x;
  // independent loads: bin counts of both grids, capacities, offsets, scalars, scene tables
  int c[2][8];
  int cap[2], nov[2], tot[2];
  int64_t base_g[2];
  const float pitch = a.pitch[o];
  const int bin0 = x + hmax - h;  // plane x - h
  const int nbr = nb - 1;
#pragma unroll
  for (int kd = 0; kd < 2; ++kd) {
    const int g = 2 * o + kd;
    cap[kd] = a.bin_cap[g];
    base_g[kd] = a.bin_base[g];
    nov[kd] = (kd == 0 || Ns > 1)
                  ? min((int)a.bin_cnt[((int64_t)par * 2 * a.O + g) * nb + nbr], 2 * a.bin_pts[g]) : 0;
    c[kd][0] = 0;
#pragma unroll
    for (int b = 0; b < 7; ++b) {
      int n = 0;
      if (b < ks && (kd == 0 || Ns > 1))
        n = min((int)a.bin_cnt[((int64_t)par * 2 * a.O + g) * nb + (bin0 + b) * kHalves + half], cap[kd]);
      c[kd][b + 1] = c[kd][b] + n;
    }
    tot[kd] = c[kd][7] + nov[kd];  // the tile's bins, then the grid's overflow list (filtered by fetch)
  }
  if (tot[0] + tot[1] == 0) return;  // block-uniform: no record of either grid reaches this tile
  const float ox = a.origin[3 * o], oy = a.origin[3 * o + 1], oz = a.origin[3 * o + 2];
  for (int i = threadIdx.x; i < Ns * 12; i += blockDim.x) s_Rt[i / 12][i % 12] = a.Rt[12 * ja + i];  // Ns up to 64: 768 words
  if (threadIdx.x <= Ns) s_off[threadIdx.x] = a.obj_off[ja + threadIdx.x];
  const float trunc = a.thr * pitch;
  for (int i = threadIdx.x; i < 2 * nvh; i += kTileThreads) { s_dist[i] = 0x7f800000u; s_id[i] = kNoCand; }
  for (int i = threadIdx.x; i < (MAXNS > kRows2Chunk ? min(Ns, kRows2Chunk) : Ns) * (kTileThreads / 16) * 13; i += kTileThreads)
    s_rows2[i] = 0.0f;  // (the rows of the first chunk; MAXNS = 64: Ns <= 64)
  for (int i = threadIdx.x; i < (kTileThreads / 16) * (kNumF + 1); i += kTileThreads) (&s_rows[0][0])[i] = 0.0f;
  const int wg = blockIdx.y * gridDim.x + blockIdx.x;
  auto stamp = [&](int i) {  // tuning aid (MF_ICC_DEBUG & 32)
    if (MF_DBG(a, 32) && threadIdx.x == 0 && wg < 2048) g_dbg_stamps[wg * 8 + i] = wall_clock64();
  };
  stamp(0);
  if (MF_DBG(a, 32) && threadIdx.x == 0 && wg < 2048) g_dbg_stamps[wg * 8 + 6] = (unsigned long long)(c[0][7] + c[1][7]);
#if MF_ICC_DEBUG_BUILD
  if (MF_DBG(a, 32) && threadIdx.x == 0 && wg < 2048) {  // where did this workgroup run? (HW_ID: CU / SH / SE; XCC_ID)
    unsigned hw, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    g_dbg_stamps[(2048 + wg) * 8 + 0] = hw;
    g_dbg_stamps[(2048 + wg) * 8 + 1] = xcc;
  }
#endif
  // the voxel phase's first-level loads, issued now: this lane's voxel of the two input grids
  float ne0 = 0.0f, tg0 = 0.0f;
  if ((int)threadIdx.x < nvox) {
    const int64_t gv = (int64_t)o * V + ((int64_t)x * D + y0) * D + (int)threadIdx.x;
    ne0 = a.grid_ne[gv];
    tg0 = a.grid_target[gv];
  }
  __syncthreads();
  const float d2_hi = a.thr * a.thr * 1.00002f;  // conservative inclusion; exact test in pass 2
  const float d2_in = a.thr * a.thr * 0.999f;    // certainly inside the truncation radius
  const float fxp = (float)x;
  Tile3 tl;
  tl.Wp = Wp; tl.rows_p = rows_p; tl.y0 = y0; tl.fxp = fxp; tl.pitch = pitch; tl.trunc = trunc; tl.d2_in = d2_in;
  tl.hi_bits = __float_as_uint(d2_hi) - 1u;  // d2 < d2_hi on the bit patterns (d2 >= 0)
  tl.in_bits = __float_as_uint(d2_in);       // d2 < d2_in  <=>  bits < in_bits

  // record i of grid kd's concatenated bins -> (plane offset b, record); rb < 0: none
  auto fetch = [&](const int kd, const int i, float4 &rv, int &rb) {
    rb = -1;
    if (i >= tot[kd]) return;
    if (i >= c[kd][7]) {  // overflow record: this tile's iff its plane is in x-h..x+h and its rows touch the half
      rv = a.rec[base_g[kd] + (int64_t)nbr * cap[kd] + (i - c[kd][7])];
      const int pl = (int)roundf(rv.x) - (x - h), iry_ = (int)roundf(rv.y);
      const bool in_half = half == 0 ? (iry_ - h < Dh) : (iry_ + h >= Dh);
      rb = (pl >= 0 && pl < ks && in_half) ? pl : -1;
      return;
    }
    int b = 0;
#pragma unroll
    for (int k = 1; k < 7; ++k) b += (k < ks && i >= c[kd][k]) ? 1 : 0;
    int cb = 0;
#pragma unroll
    for (int k = 1; k < 7; ++k) cb = (k == b) ? c[kd][k] : cb;
    rb = b;
    rv = a.rec[base_g[kd] + (int64_t)((bin0 + b) * kHalves + half) * cap[kd] + (i - cb)];
  };
  // candidate `cid` at squared distance bits `db` against the final minimum `cur` of its voxel
  auto settle_at = [&](uint32_t *id, const int ad, const uint32_t db, const uint32_t cur, const uint32_t cid) {
    bool win = db == cur && __uint_as_float(db) < d2_in;
    if (!win) {
      const float dd = pitch * sqrtf(__uint_as_float(db));
      const float dmin = pitch * sqrtf(__uint_as_float(cur));
      win = dd == dmin && dd < trunc;
    }
    if (win) atomicMin(&id[ad], cid);
  };
  auto visit = [&](const int pass, const int kd, const float4 sv, const int rb) {
    uint32_t *dist = s_dist + kd * nvh, *id = s_id + kd * nvh;
    if constexpr (KS == 3) {
      icc_visit3(pass, dist, id, tl, sv.x, sv.y, sv.z, __float_as_uint(sv.w), rb);
    } else {
      const int iry = (int)roundf(sv.y), irz = (int)roundf(sv.z);
      const uint32_t idb = __float_as_uint(sv.w) * (uint32_t)K;
      const int bb = ks - 1 - rb;
      const float dx = sv.x - fxp;
      const float dx2 = dx * dx;
      for (int aa = 0; aa < ks; ++aa) {
        const int iy = iry + aa - h;
        if (iy < y0 || iy >= y1) continue;
        const float dy = sv.y - (float)iy;
        const float dxy = dx2 + dy * dy;
        const int lrow = (iy - y0 + kPad) * Wp + kPad;
        for (int cc = 0; cc < ks; ++cc) {
          const int iz = irz + cc - h;
          if (iz < 0 || iz >= D) continue;
          const float dz = sv.z - (float)iz;
          const float d2 = dxy + dz * dz;
          if (!(d2 < d2_hi)) continue;
          const uint32_t db = __float_as_uint(d2);
          if (pass == 1) {
            atomicMin(&dist[lrow + iz], db);
          } else {
            const uint32_t cur = dist[lrow + iz];
            if (db <= cur + 8u) settle_at(id, lrow + iz, db, cur, idb + (uint32_t)((aa * ks + bb) * ks + cc));
          }
        }
      }
    }
  };

  // kept records: kFusedKeepOwn per lane of the own grid, kFusedKeepOth of the other grid, all
  // loads in flight at once (ONE memory round trip); more crowded tiles stream the rest twice
  float4 rvo[kFusedKeepOwn], rvk[kFusedKeepOth];
  int rbo[kFusedKeepOwn], rbk[kFusedKeepOth];
#pragma unroll
  for (int u = 0; u < kFusedKeepOwn; ++u) fetch(0, u * kTileThreads + (int)threadIdx.x, rvo[u], rbo[u]);
#pragma unroll
  for (int u = 0; u < kFusedKeepOth; ++u) fetch(1, u * kTileThreads + (int)threadIdx.x, rvk[u], rbk[u]);
  stamp(1);
  auto pass_over = [&](const int pass) {
#pragma unroll
    for (int u = 0; u < kFusedKeepOwn; ++u)
      if (rbo[u] >= 0) visit(pass, 0, rvo[u], rbo[u]);
#pragma unroll
    for (int u = 0; u < kFusedKeepOth; ++u)
      if (rbk[u] >= 0) visit(pass, 1, rvk[u], rbk[u]);
#pragma unroll
    for (int kd = 0; kd < 2; ++kd) {
      const int first = kTileThreads * (kd == 0 ? kFusedKeepOwn : kFusedKeepOth);
      for (int base = first; base < tot[kd]; base += kTileThreads * kTileR) {
        float4 xv[kTileR];
        int xb[kTileR];
#pragma unroll
        for (int u = 0; u < kTileR; ++u) fetch(kd, base + u * kTileThreads + (int)threadIdx.x, xv[u], xb[u]);
#pragma unroll
        for (int u = 0; u < kTileR; ++u)
          if (xb[u] >= 0) visit(pass, kd, xv[u], xb[u]);
      }
    }
  };
  if (!MF_DBG(a, 128)) pass_over(1);  // (MF_ICC_DEBUG & 128 / 256 / 512: skip a phase to time the others; results invalid)
  __syncthreads();
  stamp(2);
  if (!MF_DBG(a, 256)) pass_over(2);
  __syncthreads();
  stamp(3);
  if (MF_DBG(a, 512)) return;

  TileGeom tg_;
  tg_.o = o; tg_.ja = ja; tg_.Ns = Ns; tg_.x = x; tg_.y0 = y0; tg_.nvox = nvox; tg_.nvh = nvh; tg_.Wp = Wp; tg_.D = D;
  tg_.K = K; tg_.pitch = pitch; tg_.trunc = trunc; tg_.ox = ox; tg_.oy = oy; tg_.oz = oz;
  icc_voxel_phase<(MAXNS > kRows2Chunk)>(a, par, tg_, ne0, tg0, s_dist, s_id, s_rows2, L.v, s_Rt, s_off, stamp);
  stamp(4);
}

#ifndef MF_ICC_FUSED_WPE
#define MF_ICC_FUSED_WPE 4  // waves per SIMD the register budget is cut for (4: 128 VGPRs; 5: 96; 6: 80; 8: 64)
#endif
// (a macro, not a wrapper function: through a wrapper the standard kernel compiled to seven more SGPR spills)
  // Workgroup b runs on XCD b % 8 (observed dispatch order, MI355X_MICROARCH.md): with the plain (tile, object)
  // numbering the 64 tiles of a grid are spread over all eight L2s and each of them fetches the grid's records,
  // points and voxels over the fabric.  XCD-contiguous logical order (a.dbg bit 2048 for now): XCD k takes the logical
  // workgroups [k G/8, (k + 1) G/8) -- whole objects -- and inside an XCD workgroups i and i + 32 share a CU: planes x
  // and x + 16, a central with an outer one.  *Measured* (round 5): 8 scenes x 8 objects 96.6 -> 89.5 us per
  // iteration (the working set of a grid stays in one L2), but ONE scene 23.0 -> 24.2 us: eight objects of different
  // size on eight XCDs, the largest one's XCD is the straggler, while the plain order spreads every object over all
  // of them.  Hence by batch size: a.xcd_order is set for >= 32 objects (MF_ICC_DEBUG bit 2048 forces it on, 4096
  // off).  Two other placements for ONE scene, both measured slower than the plain order (22.6-22.9 us) and removed:
  // planes rotated by D / 2 in every other block of 256 workgroups (a central next to an outer plane on a CU:
  // 23.9-24.1), centre-out dispatch with the objects rotating over the XCDs (23.3); profiles/r05_icc_xcd_order_ab.log.
#define MF_ICC_FUSED_KERNEL_BODY(MAXNS_) \
  __shared__ FusedLds<MAXNS_> L; \
  int lin = blockIdx.y * gridDim.x + blockIdx.x; \
  const int G_ = gridDim.x * gridDim.y; \
  if (a.xcd_order && (G_ & 7) == 0) lin = (lin & 7) * (G_ >> 3) + (lin >> 3); \
  const int o = lin / (int)gridDim.x; \
  const int tile_ = lin - o * (int)gridDim.x; \
  const int ks = min(ksize_of(a.thr, a.pitch[o]), 2 * a.hmax + 1); \
  if (ks == 3) \
    icc_fused_body<3>(a, 3, par, L, o, tile_); \
  else \
    icc_fused_body<0>(a, ks, par, L, o, tile_);
__global__ __launch_bounds__(kTileThreads, MF_ICC_FUSED_WPE) void k_icc_fused(IccArgs a, int par) {  // 2 workgroups per CU
  MF_ICC_FUSED_KERNEL_BODY(kRows2Chunk)
}
// scenes of 65 .. 128 objects: the scene tables for 128, the collision rows re-used chunk by chunk (round 6)
__global__ __launch_bounds__(kTileThreads, MF_ICC_FUSED_WPE) void k_icc_fused_big(IccArgs a, int par) {
  MF_ICC_FUSED_KERNEL_BODY(kMaxSceneObjects)
}

// ---- the step as a kernel of its own: one 64-lane workgroup per object ----------------
// mode 1: after the last iteration of mf_icc_refine.  mode 2: mf_icc_loss_grad (loss, gq, gt).
__global__ __launch_bounds__(64) void k_icc_step(IccArgs a, IccStepArgs sp) {
  __shared__ float s_sum[kStepSums], s_state[kStateFloats];
  __shared__ long long s_raw[kStepRawWords];
  const int j = blockIdx.x;
  const int4 meta = a.meta[j];
  const int ja = meta.x, Ns = meta.y - meta.x;
  const int sc = a.obj_scene[j];
  if (threadIdx.x < kStateFloats) {
    const int i = threadIdx.x;
    s_state[i] = i < 4 ? sp.q_in[4 * j + i] : i < 7 ? sp.t_in[3 * j + i - 4]
                 : (sp.mode == 1 ? (i < 14 ? sp.m_in[7 * j + i - 7] : sp.v_in[7 * j + i - 14]) : 0.0f);
  }
  const float S_t = a.St[sc];
  if (sp.fused)
    icc_step_gather_fused<64>(a, sp.par, j, ja, Ns, s_raw, s_sum);
  else
    icc_step_gather<64>(a, sp.par, j, ja, Ns, s_raw, s_sum);
  if (threadIdx.x >= 16) return;
  __shared__ float s_x[kStepLaneWords];
  float Rt[12], loss, gq[4], gt[3];
  icc_step_lanes(s_sum, S_t, s_state, sp, (int)threadIdx.x, s_x, Rt, loss, gq, gt);
  __builtin_amdgcn_wave_barrier();
  if (threadIdx.x != 0) return;
  const float *st_new = s_x + 12;
  if (sp.loss_out && j == ja) sp.loss_out[sc] = loss;
  if (sp.mode == 1) {
#pragma unroll
    for (int i = 0; i < 4; ++i) sp.q_out[4 * j + i] = st_new[i];
#pragma unroll
    for (int i = 0; i < 3; ++i) sp.t_out[3 * j + i] = st_new[4 + i];
#pragma unroll
    for (int i = 0; i < 7; ++i) { sp.m_out[7 * j + i] = st_new[7 + i]; sp.v_out[7 * j + i] = st_new[14 + i]; }
#pragma unroll
    for (int i = 0; i < 12; ++i) a.Rt[(int64_t)j * 12 + i] = Rt[i];
    if (sp.traj) {
      float *tr = sp.traj + ((int64_t)sp.it * a.O + j) * 7;
#pragma unroll
      for (int i = 0; i < 7; ++i) tr[i] = st_new[i];
    }
  } else if (sp.gq_out) {
#pragma unroll
    for (int i = 0; i < 4; ++i) sp.gq_out[4 * j + i] = gq[i];
#pragma unroll
    for (int i = 0; i < 3; ++i) sp.gt_out[3 * j + i] = gt[i];
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
  int64_t W, M, Rt, bound, St, acc_own, acc_oth, state_alt, meta, tab, tab2, bin_cnt, bin_cap, bin_pts, bin_base,
      rec, rec_n, total;
  int NB, n_tab, nbins;
};

int ksize_host(float thr) {
  // Upper bound of the per-grid kernel size ksize_of(thr, pitch): the float32 quotient
  // (thr * pitch) / pitch is within 2 ulp of thr (exactly thr for thr = 2, the link's default).
  int ks = (int)ceilf(thr * 1.000001f);
  if (ks % 2 == 0) ks += 1;
  return ks;
}

int icc_bin_cap_force() {  // testing knob: MF_ICC_BIN_CAP=<n> forces every bin's capacity (overflow path)
  const char *e = getenv("MF_ICC_BIN_CAP");
  return e ? atoi(e) : 0;
}

WsLayout ws_layout(const mfIccBatch *b) {
  WsLayout l;
  const int O = b->n_objects, S = b->n_scenes, D = b->dim, max_ns = b->max_scene_objects;
  const int64_t V = (int64_t)D * D * D;
  l.NB = (int)((V + kVoxPerBlock - 1) / kVoxPerBlock);
  l.nbins = kHalves * (D + 2 * (ksize_host(b->voxel_threshold) / 2)) + 1;  // + the overflow counter
  // every (target, source) pair of a scene in chunks of kBinChunk points:
  // sum_pairs ceil(P_j / chunk) <= max_ns * n_points / chunk + O * max_ns (+ O designated entries)
  l.n_tab = (int)(((int64_t)max_ns * b->n_points + kBinChunk - 1) / kBinChunk) + O * max_ns + O;
  l.n_tab = (l.n_tab + 7) & ~7;  // (a multiple of 8: the XCD-contiguous order of k_icc_bin)
  int64_t off = 0;
  l.W = off; off = align256(off + 2 * O * V * 8);
  l.M = off; off = align256(off + kParities * 2 * O * 4);
  l.Rt = off; off = align256(off + 2 * O * 12 * 4);
  l.bound = off; off = align256(off + O * 4 * 4);
  l.St = off; off = align256(off + S * 4);
  l.acc_own = off; off = align256(off + (int64_t)kParities * O * kOwnSlots * 8);
  l.acc_oth = off; off = align256(off + (int64_t)kParities * O * max_ns * 12 * 8);
  l.state_alt = off; off = align256(off + (int64_t)O * kStateFloats * 4);
  l.meta = off; off = align256(off + (int64_t)O * 16);
  l.tab = off; off = align256(off + (int64_t)l.n_tab * 16);
  l.tab2 = off; off = align256(off + (int64_t)l.n_tab * 16);
  l.bin_cnt = off; off = align256(off + (int64_t)kParities * 2 * O * l.nbins * 4);
  l.bin_cap = off; off = align256(off + (int64_t)2 * O * 4);
  l.bin_pts = off; off = align256(off + (int64_t)2 * O * 4);
  l.bin_base = off; off = align256(off + (int64_t)2 * O * 8);
  // records: per grid (nbins - 1) bins of cap_g <= P_g / kBinShare + kBinMinCap + 1 slots + an overflow list of
  // 2 P_g; sum_g P_g = sum over scenes of Ns * P_scene <= max_ns * n_points  ->  O(N * sum P), not nbins x that
  {
    const int64_t sumP = (int64_t)max_ns * b->n_points;
    const int force = icc_bin_cap_force();
    const int64_t per_grid_extra = (force > 0 ? force : kBinMinCap) + 1;
    const int64_t binned = force > 0 ? 0 : sumP / kBinShare;
    l.rec_n = (int64_t)(l.nbins - 1) * (binned + 2 * O * per_grid_extra) + 2 * sumP;
    l.rec = off; off = align256(off + l.rec_n * 16);
  }
  l.total = off;
  return l;
}

// {0,1} no-entry grids on a tile that fits the workgroup take the single-pass kernel; MF_ICC_GENERAL=1 asks for the
// two-kernel path (A/B measurements)
bool icc_single_pass(const mfIccBatch *b) {
  return b->grid_ne_binary != 0 && ((b->dim + 1) / 2) * b->dim <= kTileThreads &&
         !(getenv("MF_ICC_GENERAL") && atoi(getenv("MF_ICC_GENERAL")) != 0);
}

IccArgs make_args(const mfIccBatch *b, void *ws) {
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
  a.max_ns = b->max_scene_objects;
  // single pass only for {0,1} no-entry grids, one voxel of a half-plane per lane, and unless
  a.ne_binary = icc_single_pass(b);
  a.dbg = getenv("MF_ICC_DEBUG") ? atoi(getenv("MF_ICC_DEBUG")) : 0;
  const WsLayout l = ws_layout(b);
  char *p = (char *)ws;
  a.W = (unsigned long long *)(p + l.W);
  a.Mbits = (uint32_t *)(p + l.M);
  a.Rt = (float *)(p + l.Rt);
  a.bound = (float *)(p + l.bound);
  a.St = (float *)(p + l.St);
  a.acc_own = (long long *)(p + l.acc_own);
  a.acc_oth = (long long *)(p + l.acc_oth);
  a.state_alt = (float *)(p + l.state_alt);
  a.meta = (int4 *)(p + l.meta);
  a.tab = (int4 *)(p + l.tab);
  a.tab2 = (int4 *)(p + l.tab2);
  a.n_tab = l.n_tab;
  a.nbins = l.nbins;
  a.hmax = ksize_host(b->voxel_threshold) / 2;
  a.bin_cnt = (uint32_t *)(p + l.bin_cnt);
  a.bin_cap = (int32_t *)(p + l.bin_cap);
  a.bin_pts = (int32_t *)(p + l.bin_pts);
  a.bin_cap_force = icc_bin_cap_force();
  a.bin_base = (int64_t *)(p + l.bin_base);
  a.rec = (float4 *)(p + l.rec);
  a.uniform_ns = (int64_t)b->n_scenes * b->max_scene_objects == b->n_objects ? b->max_scene_objects : 0;
  a.xcd_order = ((a.O >= 32) || (a.dbg & 2048)) && !(a.dbg & 4096);
  return a;
}

// One iteration k (counters / accumulators / maxima of parity k & 1): bin (+ the previous
// iteration's step when sp.mode == 1), then either the single-pass kernel or tile -> accum.
void launch_iteration(const IccArgs &a, IccStepArgs sp, int NB, int k, hipStream_t stream) {
  const int D = a.D, par = k & 1;
  sp.cpar = par;
  sp.fused = a.ne_binary;
  hipLaunchKernelGGL(k_icc_bin, dim3(a.n_tab), dim3(kBinThreads), 0, stream, a, sp);
  const size_t lds_tile = (size_t)((D + 1) / 2) * D * 2 * sizeof(uint32_t);  // 4 KB at D = 32
  // 53 KB at 32, 106 KB at 64 objects; beyond that the single-pass kernel re-uses the rows chunk by chunk
  const size_t lds_rows2 = (size_t)min(a.max_ns, kRows2Chunk) * (kAccThreads / 16) * 13 * sizeof(float);
  if (a.ne_binary) {
    // MF_ICC_LDS_PAD (bytes, tuning): unused dynamic LDS on top -- from ~48 KB on only ONE workgroup of k_icc_fused
    // fits a CU (half the resident waves: the experiment of leaving wave slots to a network running beside it)
    static const size_t pad = getenv("MF_ICC_LDS_PAD") ? (size_t)atoi(getenv("MF_ICC_LDS_PAD")) : 0;
    const size_t lds = 4 * fused_tile_words(D) * sizeof(uint32_t) + lds_rows2 + pad;
    if (pad) mf::allow_big_lds((const void *)k_icc_fused, (int)lds);
    if (a.max_ns > kRows2Chunk)
      hipLaunchKernelGGL(k_icc_fused_big, dim3(D * kHalves, a.O), dim3(kTileThreads), lds, stream, a, par);
    else
      hipLaunchKernelGGL(k_icc_fused, dim3(D * kHalves, a.O), dim3(kTileThreads), lds, stream, a, par);
    return;
  }
  hipLaunchKernelGGL(k_icc_tile, dim3(D * kHalves, 2 * a.O), dim3(kTileThreads), lds_tile, stream, a, par);
  hipLaunchKernelGGL(k_icc_accum, dim3(NB, a.O), dim3(kAccThreads), lds_rows2, stream, a, par);
}

// chainer Adam: alpha_t = alpha * sqrt(1 - b2^t) / (1 - b1^t), in double, cast once
float adam_alpha_t(float alpha, int step) {
  const double fix1 = 1.0 - pow(0.9, (double)step), fix2 = 1.0 - pow(0.999, (double)step);
  return (float)((double)alpha * sqrt(fix2) / fix1);
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
         b->n_points >= 0 && b->max_scene_objects > 0 &&
         b->max_scene_objects <= (icc_single_pass(b) ? kMaxSceneObjects : kMaxSceneObjectsGeneral) &&
         b->voxel_threshold > 0.0f && ksize_host(b->voxel_threshold) <= 7 &&
         (double)b->n_points * 343.0 < 4294967295.0 && b->n_points < (1 << 27) && b->flags == 0;
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

extern "C" int mf_icc_iteration_launches(const mfIccBatch *batch) {
  if (!icc_batch_ok(batch)) return -1;
  char dummy[8];
  const IccArgs a = make_args(batch, dummy);  // (pointers are offsets from a dummy base: not dereferenced)
  return a.ne_binary ? 2 : 3;
}

static int icc_validate(const mfIccBatch *b) {
  // collision-moment rows: max_scene_objects x 1664 B of dynamic LDS (106 KB at 64 objects)
  if (int e = mf::allow_big_lds((const void *)k_icc_fused, 124 * 1024)) return e;
  if (int e = mf::allow_big_lds((const void *)k_icc_fused_big, 124 * 1024)) return e;
  if (int e = mf::allow_big_lds((const void *)k_icc_accum, 124 * 1024)) return e;
  if (!icc_batch_ok(b)) {
    mf::set_last_error(hipErrorInvalidValue, "mf_icc: invalid batch descriptor");
    return -(int)hipErrorInvalidValue;
  }
  return 0;
}

extern "C" int mf_icc_debug_stamps(unsigned long long *host_out, int n) {
  if (!MF_ICC_DEBUG_BUILD) {
    mf::set_last_error(hipErrorInvalidValue, "mf_icc_debug_stamps: build with `make ICC_DEBUG=1`");
    return -(int)hipErrorInvalidValue;
  }
  return -(int)hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_dbg_stamps), sizeof(unsigned long long) * n);
}

extern "C" int mf_icc_launch_stage(const mfIccBatch *batch, const float *q, const float *t, void *ws,
                                   int32_t stage, mfStream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (int e = icc_validate(batch)) return e;
  IccArgs a = make_args(batch, ws);
  const int D = a.D;
  if (stage == 0) {
    if (q && t) hipLaunchKernelGGL(k_icc_pose, dim3(a.O), dim3(256), 0, stream, a, q, t, (float *)nullptr);
    // inside an iteration k_icc_accum empties the bins; this hook has no accum launch
    if (int e_ = mf::fill_bytes(a.bin_cnt, 0, sizeof(uint32_t) * 2 * a.O * a.nbins, stream)) return e_;
    IccStepArgs sp = {};
    hipLaunchKernelGGL(k_icc_bin, dim3(a.n_tab), dim3(kBinThreads), 0, stream, a, sp);
  } else if (stage == 1) {
    const size_t lds = (size_t)((D + 1) / 2) * D * 2 * sizeof(uint32_t);
    hipLaunchKernelGGL(k_icc_tile, dim3(D * kHalves, 2 * a.O), dim3(kTileThreads), lds, stream, a, 0);
  } else if (stage == 2) {
    if (!a.ne_binary) {
      mf::set_last_error(hipErrorInvalidValue, "mf_icc_launch_stage: stage 2 needs {0,1} no-entry grids");
      return -(int)hipErrorInvalidValue;
    }
    const size_t lds = 4 * fused_tile_words(D) * sizeof(uint32_t) +
                       (size_t)min(a.max_ns, kRows2Chunk) * (kAccThreads / 16) * 13 * sizeof(float);
    if (a.max_ns > kRows2Chunk)
      hipLaunchKernelGGL(k_icc_fused_big, dim3(D * kHalves, a.O), dim3(kTileThreads), lds, stream, a, 0);
    else
      hipLaunchKernelGGL(k_icc_fused, dim3(D * kHalves, a.O), dim3(kTileThreads), lds, stream, a, 0);
  } else {
    mf::set_last_error(hipErrorInvalidValue, "mf_icc_launch_stage: stage must be 0, 1 or 2");
    return -(int)hipErrorInvalidValue;
  }
  return mf::check_launch("mf_icc_launch_stage");
}

extern "C" int mf_icc_prepare(const mfIccBatch *batch, void *ws, mfStream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (int e = icc_validate(batch)) return e;
  IccArgs a = make_args(batch, ws);
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
  IccArgs a = make_args(batch, ws);
  const WsLayout l = ws_layout(batch);
  hipLaunchKernelGGL(k_icc_pose, dim3(a.O), dim3(256), 0, stream, a, q, t, (float *)nullptr);
  IccStepArgs none = {};
  launch_iteration(a, none, l.NB, 0, stream);
  IccStepArgs sp = {};
  sp.mode = 2;
  sp.fused = a.ne_binary;
  sp.par = 0;
  sp.q_in = q;
  sp.t_in = t;
  sp.loss_out = loss;
  sp.gq_out = gq;
  sp.gt_out = gt;
  hipLaunchKernelGGL(k_icc_step, dim3(a.O), dim3(64), 0, stream, a, sp);
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
  IccArgs a = make_args(batch, ws);
  const WsLayout l = ws_layout(batch);

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
  key.v.push_back(((uint64_t)(uint32_t)dev << 32) | ((uint32_t)max_ns << 1) | (uint32_t)a.ne_binary);
  key.v.push_back((uint64_t)(uint32_t)a.bin_cap_force);
  key.v.push_back((uint64_t)(uint32_t)a.dbg);

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
    // State after i steps lives in the caller's arrays for even i and in the workspace copy for
    // odd i: the step folded into k_icc_bin reads one while its designated workgroups write the
    // other.  Iteration k: [bin: step k-1 (k > 0), binning] -> tile -> accum; then one last step.
    float *alt = a.state_alt;
    float *sq[2] = {q, alt}, *st[2] = {t, alt + 4 * a.O}, *sm[2] = {adam_m, alt + 7 * a.O},
          *sv[2] = {adam_v, alt + 14 * a.O};
    hipLaunchKernelGGL(k_icc_pose, dim3(a.O), dim3(256), 0, cap, a, (const float *)q,
                       (const float *)t, traj);
    for (int k = 0; k <= n_iter; ++k) {
      IccStepArgs sp = {};
      if (k > 0) {
        const int in = (k - 1) & 1, out = k == n_iter ? 0 : (k & 1);
        sp.mode = 1;
        sp.par = (k - 1) & 1;
        sp.it = k;
        sp.aq = adam_alpha_t(alpha_q, step0 + k);
        sp.at = adam_alpha_t(alpha_t, step0 + k);
        sp.q_in = sq[in]; sp.t_in = st[in]; sp.m_in = sm[in]; sp.v_in = sv[in];
        sp.q_out = sq[out]; sp.t_out = st[out]; sp.m_out = sm[out]; sp.v_out = sv[out];
        sp.loss_out = losses ? losses + (int64_t)(k - 1) * a.S : nullptr;
        sp.traj = k < n_iter ? traj : nullptr;
      }
      if (k == n_iter) {  // the step of the last iteration, as a kernel of its own
        sp.fused = a.ne_binary;
        hipLaunchKernelGGL(k_icc_step, dim3(a.O), dim3(64), 0, cap, a, sp);
        break;
      }
      launch_iteration(a, sp, l.NB, k, cap);
    }
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
