/*
 * mfhip.h -- C ABI of libmfhip.so: MI355X (gfx950) kernels for MoreFusion's
 * volumetric pose hot path (voxelize -> trilinear sample -> TDF / pseudo-occupancy
 * -> ICC / ICP refinement -> KNN).
 *
 * The reference has NO binary FFI for this path: its CUDA code is source text
 * handed to cupy.ElementwiseKernel / cupy.RawKernel at run time.
 * Each entry point below therefore replaces one such JIT-kernel call site (cited
 * as file:line under /root/reference/) and is what a maintainer's ctypes stub
 * binds instead.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer unless marked host; buffers are borrowed,
 *     never freed or reallocated here; outputs and workspaces are caller-allocated;
 *   - all calls are asynchronous on `stream` (a hipStream_t passed as void*), never
 *     synchronise, never allocate device memory (mf_icc_* keep a small host-side
 *     hipGraph cache);
 *   - return 0 on success, a negative hipError_t otherwise (mf_last_error_string());
 *   - float = IEEE binary32, arithmetic un-fused (no FMA contraction) so that voxel
 *     indices and arg-min decisions are bit-identical to the CPU oracle;
 *   - voxel index = round-half-away((p - origin) / pitch)  (CUDA round());
 *   - layouts: points [n,3] row-major; voxel grids [B,C,X,Y,Z] (z fastest).
 */
#ifndef MFHIP_H_
#define MFHIP_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void *mfStream_t; /* hipStream_t */

int mf_version(void);
const char *mf_last_error_string(void);

/* ---- A1/A2 average_voxelization_3d -------------------------------------
 * replaces ElementwiseKernel "average_voxelization_3d_fwd" + the nonzero/divide
 * epilogue   morefusion/functions/geometry/average_voxelization_3d.py:42-118
 * and "voxelize_bwd"                                              :146-220.
 * matrix [B,C,X,Y,Z] and counts [B,X,Y,Z] are written completely (no pre-zero).
 * head [B*X*Y*Z] int32 and link [n] int32 are scratch.  nan_flag (1 int32, may
 * be NULL) is set to 1 if any point coordinate is NaN (reference raises
 * ValueError("points include nan"); the host wrapper does after reading it).
 * Sums are taken in increasing point index (== the reference's CPU loop order),
 * so results are run-to-run deterministic and bit-equal to forward_cpu. */
int mf_average_voxelization_3d_fwd(const float *values, const float *points,
                                   const int32_t *batch_indices, int64_t n, int C,
                                   int B, int X, int Y, int Z, float ox, float oy,
                                   float oz, float pitch, float *matrix,
                                   int32_t *counts, int32_t *head, int32_t *link,
                                   int32_t *nan_flag, mfStream_t stream);
int mf_average_voxelization_3d_bwd(const float *gmatrix, const float *points,
                                   const int32_t *batch_indices,
                                   const int32_t *counts, int64_t n, int C, int B,
                                   int X, int Y, int Z, float ox, float oy, float oz,
                                   float pitch, float *gvalues, mfStream_t stream);

/* ---- A3 max_voxelization_3d ---------------------------------------------
 * replaces "max_voxelization_3d_fwd" + gather, "max_voxelization_3d_bwd"
 *   morefusion/functions/geometry/max_voxelization_3d.py:58-138, :140-185.
 * Winner = max intensity, lowest point index among ties (the CPU rule :33-38;
 * the CUDA CAS/Max/Exch sequence is racy).  indices [B,X,Y,Z] (-1 = empty),
 * matrix [B,C,X,Y,Z] both written completely; key [B*X*Y*Z] uint64 scratch. */
int mf_max_voxelization_3d_fwd(const float *values, const float *points,
                               const int32_t *batch_indices, const float *intensities,
                               int64_t n, int C, int B, int X, int Y, int Z, float ox,
                               float oy, float oz, float pitch, float *matrix,
                               int32_t *indices, uint64_t *key, int32_t *nan_flag,
                               mfStream_t stream);
/* gvalues [n,C] must be zeroed by the caller. */
int mf_max_voxelization_3d_bwd(const float *gmatrix, const int32_t *indices, int64_t n,
                               int C, int B, int X, int Y, int Z, float *gvalues,
                               mfStream_t stream);

/* ---- A4 interpolate_voxel_grid ------------------------------------------
 * replaces "interpolate_voxel_grid_fwd" / "_bwd"
 *   morefusion/functions/geometry/interpolate_voxel_grid.py:159-214, :216-268.
 * points are in voxel-index units; low corner = (int)coord (trunc toward zero).
 * values: [n,C] if channels_first == 0, else [C,n] (coalesced layout the pose
 * network consumes); every row is written (zeros for rows whose batch index is outside
 * [0, B)).  gvox [B,C,X,Y,Z] is written completely by _bwd.
 * batch_start (may be NULL): B+1 row offsets, rows of item b = [batch_start[b],
 * batch_start[b+1]) -- what a caller with batch-sorted points knows for free (the pose network:
 * b * P).  With it a workgroup touches only its item's rows; without it every workgroup
 * filters all n batch_indices. */
int mf_interpolate_voxel_grid_fwd(const float *vox, const float *points,
                                  const int32_t *batch_indices, const int32_t *batch_start,
                                  int64_t n, int B, int C, int X, int Y, int Z, float *values,
                                  int channels_first, mfStream_t stream);
int mf_interpolate_voxel_grid_bwd(const float *gvalues, const float *points,
                                  const int32_t *batch_indices, const int32_t *batch_start,
                                  int64_t n, int B, int C, int X, int Y, int Z, float *gvox,
                                  int channels_first, mfStream_t stream);

/* ---- A5 occupancy_grid_3d -------------------------------------------------
 * replaces the dense [X,Y,Z,P] composite
 *   morefusion/functions/geometry/occupancy_grid_3d.py:31-85
 * with one fused min-over-points kernel.  grid [X,Y,Z]; dmin [X,Y,Z] (saved for
 * backward).  _bwd: gradient to EVERY point at the minimum distance (chainer
 * F.min rule); gpoints [P,3] must be zeroed by the caller. */
int mf_occupancy_grid_3d_fwd(const float *points, int64_t P, float pitch, float ox,
                             float oy, float oz, int X, int Y, int Z, float threshold,
                             float *grid, float *dmin, mfStream_t stream);
int mf_occupancy_grid_3d_bwd(const float *ggrid, const float *points, int64_t P,
                             float pitch, float ox, float oy, float oz, int X, int Y,
                             int Z, float threshold, const float *dmin, float *gpoints,
                             mfStream_t stream);

/* ---- A6/A7 truncated_distance_function, pseudo_occupancy_voxelization ----
 * replaces "truncated_distance_function_fwd" / "_bwd"
 *   morefusion/functions/geometry/truncated_distance_function.py:21-103, :105-166
 * tdf [X,Y,Z]; flat [X,Y,Z] int32 = p*K+k of the arg-min candidate (-1 none;
 * exact arg-min, lowest flat index among equal distances -- the reference's
 * atomicMin/atomicExch pair is racy).  K = ksize^3, ksize = ceil(trunc/pitch)
 * made odd (:36-38).  _bwd: gpoints [P,3] must be zeroed by the caller. */
int mf_truncated_distance_function_fwd(const float *points, int64_t P, float pitch,
                                       float ox, float oy, float oz, int X, int Y, int Z,
                                       float truncation, float *tdf, int32_t *flat,
                                       mfStream_t stream);
int mf_truncated_distance_function_bwd(const float *gtdf, const float *points,
                                       const int32_t *flat, int64_t P, float pitch,
                                       float ox, float oy, float oz, int X, int Y, int Z,
                                       float truncation, float *gpoints,
                                       mfStream_t stream);
/* pseudo_occupancy_voxelization epilogue (:181-213) on the TDF result:
 * grids [3,X,Y,Z] = uniform, surface, inside; wmax: 1 float scratch (zeroed here). */
int mf_pseudo_occupancy_weights(const float *tdf, const int32_t *flat, const float *sdf,
                                int X, int Y, int Z, int K, float truncation,
                                float sdf_offset, float *grids, float *wsurf, float *win,
                                float *wmax, mfStream_t stream);

/* ---- A11 geometry.nn ------------------------------------------------------
 * replaces RawKernel cuComputeDistanceGlobal + cupy.argmin
 *   morefusion/geometry/knn/nn.py:18-49, knn/cuComputeDistanceGlobal.cu:20-86
 * without materialising the R x Q matrix.  dim = 3.  out [Q] int64; optional
 * out_dist [Q] float (squared distance to the match; may be NULL). */
int mf_nn(const float *ref, int64_t R, const float *query, int64_t Q, int64_t *out,
          float *out_dist, mfStream_t stream);

/* ---- A10 IterativeClosestPointLink ----------------------------------------
 *   morefusion/contrib/iterative_closest_point_link.py:26-44
 * source [S,3] model frame, target [T,3]; Rt [12] = row-major R(3x3) then t.
 * out [16]: loss, n_matched, pad, pad, gRt[12] (d loss / d R, d loss / d t).
 * out must be zeroed by the caller. */
int mf_icp_loss_grad(const float *source, int64_t S, const float *target, int64_t T,
                     const float *Rt, float thresh, float *out, mfStream_t stream);

/* The ICP driver's whole loop on the device
 *   examples/ycb_video/pose_refinement/check_iterative_closest_point_link.py:40-70
 * (one link per instance, loss = sum of the links' losses, chainer Adam with the translations'
 * alpha scaled): n_iter x {k_icp over every link, quaternion chain rule + Adam step}, 2 launches
 * per iteration, no host synchronisation and no autograd round trip.
 *   source [sum S_l,3], src_off [L+1]; target [sum T_l,3], tgt_off [L+1]; max_T >= max_l T_l;
 *   q [L,4], t [L,3], adam_m / adam_v [L,7] updated in place; losses [n_iter,L] may be NULL;
 *   ws: 28 * L floats. */
int mf_icp_refine(const float *source, const int32_t *src_off, const float *target,
                  const int32_t *tgt_off, int32_t L, int32_t max_T, float thresh, float *q, float *t,
                  float *adam_m, float *adam_v, int32_t n_iter, int32_t step0, float alpha_q,
                  float alpha_t, float *losses, float *ws, mfStream_t stream);

/* ---- A9 IterativeCollisionCheckLink (fused) --------------------------------
 *   morefusion/contrib/iterative_collision_check_link.py:31-99 (forward),
 *   truncated_distance_function.py:105-166 (backward), driver loop
 *   examples/ycb_video/pose_refinement/check_iterative_collision_check_link.py:44-79
 *
 * A batch holds S independent scenes with O objects in total.  Device arrays:
 *   pts4      [Ptot] float4  model-frame point (x,y,z) + its sdf value (w)
 *   obj_off   [O+1]  int32   point range of object o
 *   scene_off [S+1]  int32   object range of scene s
 *   obj_scene [O]    int32
 *   pitch [O], origin [O,3], grid_target [O,D,D,D], grid_ne [O,D,D,D]
 *   q [O,4] (wxyz), t [O,3]          -- parameters (updated in place by refine)
 *   adam_m, adam_v [O,7]             -- chainer-Adam moments (q then t)
 * ws: workspace of mf_icc_workspace_bytes() bytes.
 */
typedef struct {
  const void *pts4;
  const int32_t *obj_off;
  const int32_t *scene_off;
  const int32_t *obj_scene;
  const float *pitch;
  const float *origin;
  const float *grid_target;
  const float *grid_ne;
  int32_t n_objects; /* O */
  int32_t n_scenes;  /* S */
  int32_t n_points;  /* Ptot */
  int32_t dim;       /* D (32) */
  int32_t max_scene_objects; /* max objects in one scene (<= 128 with grid_ne_binary, else <= 64) */
  float voxel_threshold;
  float sdf_offset;
  int32_t grid_ne_binary; /* != 0: the caller guarantees that every grid_ne value is exactly 0 or 1
                             (what the reference's callers pass: bool grids cast to float32) ->
                             single-pass iteration (TDF tiles and weights/sums in one kernel, two
                             launches per iteration); 0: any values, two-kernel path */
  int32_t flags;          /* reserved, must be 0 (round 5: bit 0 selected the experimental one-launch iteration
                             k_icc_iter -- same bits, measured slower, removed in round 6) */
} mfIccBatch;

/* Bytes of workspace for this batch (negative: invalid descriptor).  Holds the winners of every
 * grid, partial sums, and the per-iteration x-plane bins of voxel-frame point records:
 * (dim + 2h) * max_scene_objects * n_points * 16 B of address space, of which one iteration
 * touches only the points that fall inside a grid. */
int64_t mf_icc_workspace_bytes(const mfIccBatch *batch);

/* Once per batch (and again whenever its point / grid_target arrays change): model-frame
 * bounding spheres, scene tables, sum(grid_target) per scene, the (target grid, source object,
 * point chunk) table and the bin offsets into the workspace. */
int mf_icc_prepare(const mfIccBatch *batch, void *ws, mfStream_t stream);

/* One forward+backward: loss [S], gq [O,4], gt [O,3]; q,t not modified.
 * Requires mf_icc_prepare on the same (batch, ws). */
int mf_icc_loss_grad(const mfIccBatch *batch, const float *q, const float *t,
                     float *loss, float *gq, float *gt, void *ws, mfStream_t stream);

/* n_iter x {forward, backward, chainer-Adam step} entirely on device (one
 * hipGraph launch; no host sync).  losses [n_iter,S] and traj [n_iter,O,7]
 * (pose BEFORE each step) may be NULL.  step0 = number of Adam steps already
 * taken (bias correction). */
int mf_icc_refine(const mfIccBatch *batch, float *q, float *t, float *adam_m,
                  float *adam_v, int32_t n_iter, int32_t step0, float alpha_q,
                  float alpha_t, float *losses, float *traj, void *ws,
                  mfStream_t stream);

/* Launches per iteration mf_icc_refine will use for this batch: 2 = k_icc_bin + k_icc_fused ({0,1} no-entry grids:
 * the default), 3 = k_icc_bin + k_icc_tile + k_icc_accum (any no-entry grid values).  Negative: invalid descriptor. */
int mf_icc_iteration_launches(const mfIccBatch *batch);

/* Measurement hook so that bench.py can time ONE kernel of an ICC iteration with HIP events:
 * stage 0 = (optional pose refresh from q, t if non-NULL) + empty the bins + k_icc_bin (pose ->
 * world points -> x-plane bins of voxel-frame records); stage 1 = k_icc_tile alone (bins ->
 * per-grid (min distance, arg-min) winners); stage 2 = k_icc_fused alone (bins -> winners ->
 * weights / sums / moments; needs grid_ne_binary).  Stages 1, 2 are repeatable: they do not
 * consume the bins (their sums simply keep adding up). */
int mf_icc_launch_stage(const mfIccBatch *batch, const float *q, const float *t, void *ws,
                        int32_t stage, mfStream_t stream);

/* Tuning aid: with MF_ICC_DEBUG=32 in the environment k_icc_bin / k_icc_tile / k_icc_accum record
 * wall_clock64() phase stamps per workgroup; this copies the first n 64-bit words of that
 * table to host memory (synchronous).  Not used by the product path. */
int mf_icc_debug_stamps(unsigned long long *host_out, int n);

/* ---- A13 conv3 of the pose network on sparse voxelized features -----------------
 * replaces the dense cuDNN Convolution3D(Cs+16 -> Cout, k=4, s=2, pad=1) at
 *   morefusion/contrib/singleview_3d/models/model.py:73,128
 * for the Cs voxelized channels, whose input is <= 3 % occupied (counts > 0): 8 parity
 * classes x one fp32-MFMA GEMM [n x Cs].[Cs x 8*Cout] + an output-stationary reduce.
 *   x      [B,Cs,D,D,D]  dense voxelized features (zeros where counts == 0)
 *   counts [B,D,D,D]     from mf_average_voxelization_3d_fwd
 *   Wp     packed weights from mf_sparse_conv3d_pack_weights (8*Cs*8*Cout floats)
 *   dense  [B,Cout,D/2..] optional addend (the dense occupancy-channel part), bias [Cout]
 *   out    [B,Cout,D/2,D/2,D/2] written completely; relu != 0 applies max(.,0)
 *   max_rows >= number of occupied voxels (e.g. the number of points); ws from
 *   mf_sparse_conv3d_workspace_bytes (n_points = 0 for this entry).  Deterministic (fixed tap order, no atomics). */
int64_t mf_sparse_conv3d_workspace_bytes(int32_t B, int32_t Cs, int32_t Cout, int32_t D,
                                         int32_t max_rows, int64_t n_points);
/* W [Cout, w_cin, 4,4,4]; packs input channels [c_off, c_off+Cs). */
int mf_sparse_conv3d_pack_weights(const float *W, int32_t Cout, int32_t Cs, int32_t w_cin,
                                  int32_t c_off, float *Wp, mfStream_t stream);
int mf_sparse_conv3d_k4s2_fwd(const float *x, const int32_t *counts, const float *Wp,
                              const float *dense, const float *bias, float *out, void *ws,
                              int32_t B, int32_t Cs, int32_t Cout, int32_t D, int32_t max_rows,
                              int32_t relu, mfStream_t stream);

/* ---- A12 functions.average_distance (ADD / ADD-S pose loss), batched ----------------
 * replaces the transform_points x2 + geometry.nn + gather + sub/square/sum/sqrt/mean composite
 *   morefusion/functions/loss/average_distance.py:64-85
 * and the per-object Python loop around it, contrib/singleview_3d/models/model.py:406-434.
 *   points [B,M,3] model points, T_true [B,4,4], T_pred [B,P,4,4] row-major, symmetric [B]
 *   uint8 (NULL = none): 0 -> ADD, else ADD-S (nearest true point, lowest index among equal
 *   squared distances).  out [B,P] = mean_m || true'_m - pred_m ||.
 *   nn_idx [B,P,M] int32 (may be NULL): arg-min indices of the ADD-S rows, written by _fwd and
 *   read by _bwd (NULL: _bwd searches again).
 * _bwd: gT_pred [B,P,4,4] = d(sum gout * out) / d T_pred (rows 0..2; row 3 = 0).  The true
 * pose and the model points receive no gradient (they are data in the reference's callers). */
int mf_average_distance_fwd(const float *points, const float *T_true, const float *T_pred,
                            const uint8_t *symmetric, int32_t B, int32_t M, int32_t P,
                            float *out, int32_t *nn_idx, mfStream_t stream);
int mf_average_distance_bwd(const float *points, const float *T_true, const float *T_pred,
                            const uint8_t *symmetric, const float *gout, int32_t B, int32_t M,
                            int32_t P, const int32_t *nn_idx, float *gT_pred, mfStream_t stream);

/* The same convolution fed by the POINTS (inference): average_voxelization_3d's per-voxel chains
 * write the compact rows the GEMM consumes -- the dense [B,Cs,D,D,D] tensor of
 *   morefusion/contrib/singleview_3d/models/model.py:114-128 (151 MB at B = 8) is never
 * materialised.  values [n,Cs], points [n,3] voxel-frame (origin o, pitch), batch_indices [n];
 * same means (increasing point index), same output bits as voxelize + mf_sparse_conv3d_k4s2_fwd.
 * ws from mf_sparse_conv3d_workspace_bytes(..., n_points = n). */
int mf_sparse_conv3d_k4s2_points_fwd(const float *values, const float *points,
                                     const int32_t *batch_indices, int64_t n, float ox, float oy,
                                     float oz, float pitch, const float *Wp, const float *dense,
                                     const float *bias, float *out, void *ws, int32_t B,
                                     int32_t Cs, int32_t Cout, int32_t D, int32_t max_rows,
                                     int32_t relu, mfStream_t stream);

/* ---- A13 conv4 (and the dense occupancy channels of conv3): Convolution3D k=4 s=2 pad=1 ----
 * replaces the cuDNN convolutions at
 *   morefusion/contrib/singleview_3d/models/model.py:74,139  (conv4: 256 -> 512 on 16^3, + ReLU)
 *   morefusion/contrib/singleview_3d/models/model.py:73,128  (conv3's 16 dense occupancy channels)
 * with an fp32-MFMA (v_mfma_f32_32x32x2_f32, exact fp32) implicit GEMM over CHANNELS-LAST tensors:
 *   x    [B, D,D,D, Cin]          Cin a power of two >= 4
 *   wt   [Cout, 64, Cin]          from mf_conv3d_k4s2_pack_weights (tap = (kx*4 + ky)*4 + kz)
 *   bias [Cout] or NULL, add [B, (D/2)^3, Cout] or NULL (added before the activation)
 *   out  [B, D/2,D/2,D/2, Cout]   Cout % 128 == 0; relu != 0 applies max(., 0)
 *   split: K (taps) is cut into `split` slabs (mf_conv3d_k4s2_default_split picks >= 512 workgroups),
 *   partial sums go to ws (mf_conv3d_k4s2_workspace_bytes) and are added in slab order: deterministic. */
int mf_conv3d_k4s2_pack_weights(const float *W, int32_t Cout, int32_t Cin, int32_t w_cin, int32_t c_off,
                                float *wt, mfStream_t stream);
int32_t mf_conv3d_k4s2_default_split(int32_t B, int32_t Cin, int32_t Cout, int32_t D);
int64_t mf_conv3d_k4s2_workspace_bytes(int32_t B, int32_t Cout, int32_t D, int32_t split);
int mf_conv3d_k4s2_fwd(const float *x, const float *wt, const float *bias, const float *add, float *out,
                       void *ws, int32_t B, int32_t Cin, int32_t Cout, int32_t D, int32_t split,
                       int32_t relu, mfStream_t stream);
/* [B, C, V] -> [B, V, C] (channels-first grid -> channels-last) */
int mf_to_channels_last(const float *src, float *dst, int32_t B, int32_t C, int64_t V, mfStream_t stream);

/* Channels-last forms of the inference path between conv3 and the heads (same results as the
 * channels-first entries above, other memory layout):
 *   mf_sparse_conv3d_k4s2_points_cl_fwd: as mf_sparse_conv3d_k4s2_points_fwd, but values rows have pitch
 *     ldv floats (a column block of a wider matrix), dense / out are [B, (D/2)^3, Cout]; Cout % 256 == 0.
 *   mf_interpolate_voxel_grid_cl_fwd (replaces K5, interpolate_voxel_grid.py:170-212, for vox
 *     [B, X*Y*Z, C]): out[p*ldo + c], rows of pitch ldo floats; rows with a batch index outside [0,B) = 0.
 *   mf_occupancy_convs_fwd (replaces the cuDNN Convolution3D pair model.py:69-72,120-124: 1 -> 8 k3 pad 1,
 *     8 -> 16 k3 dilation 2 pad 2, ReLU after each): grid [B,D,D,D] -> h2 [B, D^3, 16]; h1 [B, D^3, 8] scratch;
 *     w1 [27,1,8], w2 [27,8,16] = W.permute(2,3,4,1,0) of the Chainer/torch weight. */
int mf_sparse_conv3d_k4s2_points_cl_fwd(const float *values, int64_t ldv, const float *points,
                                        const int32_t *batch_indices, int64_t n, float ox, float oy,
                                        float oz, float pitch, const float *Wp, const float *dense,
                                        const float *bias, float *out, void *ws, int32_t B, int32_t Cs,
                                        int32_t Cout, int32_t D, int32_t max_rows, int32_t relu,
                                        mfStream_t stream);
int mf_interpolate_voxel_grid_cl_fwd(const float *vox, const float *points, const int32_t *batch_indices,
                                     int64_t n, int B, int C, int X, int Y, int Z, float *out, int64_t ldo,
                                     mfStream_t stream);
int mf_occupancy_convs_fwd(const float *grid, const float *w1, const float *b1, const float *w2,
                           const float *b2, float *h1, float *h2, int32_t B, int32_t D, mfStream_t stream);

/* ---- per-point 1x1 convolutions (heads, point MLP) as row-major fp32-MFMA GEMMs ---------------
 * replaces the cuDNN Convolution1D(k = 1) chains at
 *   morefusion/contrib/singleview_3d/models/model.py:76-91,245-262 (three heads: 984-640-256-128-n_fg*c)
 * on points-major activations:  out[m][n] = act( sum_k A[m][k] * W[n][k] + bias[n] ),  m < M, n < N.
 *   A [M, lda], W [Npad, ldw] (Convolution1D's W[:, :, 0], zero rows up to Npad % 128 == 0), out [M, ldo];
 *   `groups` independent problems per launch at the given element strides (heads side by side, each
 *   reading / writing its own column block); K % 4 == 0 (no multiple of 32 needed); relu != 0: max(., 0). */
int mf_linear_fwd(const float *A, int64_t a_group_stride, int32_t lda, const float *W, int64_t w_group_stride,
                  int32_t ldw, const float *bias, int64_t b_group_stride, float *out, int64_t o_group_stride,
                  int32_t ldo, int32_t M, int32_t N, int32_t Npad, int32_t K, int32_t groups, int32_t relu,
                  mfStream_t stream);

/* ---- bf16 training / inference path of the 3-D CNN and the 1x1 convolution chains (round 4) --------------------
 * replaces cuDNN's forward, backward-data and backward-filter of
 *   morefusion/contrib/singleview_3d/models/model.py:73-74,125-139 (conv3, conv4: Convolution3D(.., 4, 2, pad = 1))
 *   morefusion/contrib/singleview_3d/models/model.py:59-66,76-91,101-111,239-258 (Convolution1D chains)
 * as trained by examples/ycb_video/singleview_3d/train.py:342-369 (BASELINE config 5: bf16, data-parallel).
 * All activations / gradients are bf16 (the framework's bfloat16 bit pattern, passed as void *), accumulation is
 * fp32 on v_mfma_f32_32x32x16_bf16, parameters and their gradients stay fp32 (csrc/gemm_bf16.hip).
 *
 * mf_cast_rows_bf16     fp32 [rows, src_ld] -> bf16 [rows, dst_ld] (columns >= cols zero; dst_ld % 8 == 0)
 * mf_relu_mask_bf16     dz = (y > 0) ? dy : 0 over n elements (n % 8 == 0); dy bf16, or dy32 fp32 (exactly one)
 * mf_linear_bf16        out = act(A W^T + bias): A [M, lda], W [N, ldw] bf16 (rows k-contiguous), bias fp32 [N],
 *                       out [M, ldo] bf16 or fp32 (out_f32), `accumulate`: out += (fp32 only); `groups` problems
 *                       per launch at the given element strides.  The data gradient of a layer is the same call
 *                       with the transposed weight: dA = dY W  ->  mf_linear_bf16(dY, W^T [K, N]).
 * mf_linear_wgrad_bf16  dW [N, ldc] fp32 = sum_m dY[m][n] A[m][k]; split > 1: row ranges into ws
 *                       (split * groups * N * ldc floats), summed in range order.
 * mf_conv3d_k4s2_pack_bf16   W fp32 [Cout, w_cin, 4, 4, 4] (channels c_off .. c_off + Cin) -> fwd bf16
 *                       [Cout, 64, Cin] and / or dgrad bf16 [8 parity classes, Cin, 8 slots, Cout] (either may be null)
 * mf_conv3d_k4s2_bf16_fwd    out [B, (D/2)^3, Cout] = act(conv(x [B, D^3, Cin]) + bias)      (channels-last)
 * mf_conv3d_k4s2_bf16_dgrad  dx [B, D^3, Cin] (+)= conv^T(dy [B, (D/2)^3, Cout]); bf16, or fp32 (+ accumulate)
 * mf_conv3d_k4s2_bf16_wgrad  dW fp32 in the framework layout (channels c_off ..) = sum_voxels dy (x) im2col(x);
 *                       ws: mf_conv3d_k4s2_bf16_wgrad_workspace_bytes(Cin, Cout, split)
 * D is a power of two, Cin % 8 == 0, Cout % 8 == 0.  Asynchronous, never allocate or synchronise.
 *
 * The same engines with a general geometry -- kernel ks in {3, 4}, stride in {1, 2}, pad, dilation; output size
 * Do = (D + 2 pad - dil (ks - 1) - 1) / stride + 1 a power of two -- run the occupancy branch
 *   morefusion/contrib/singleview_3d/models/model.py:69-72,120-124 (conv1_occ 1 -> 8 k3 p1, conv2_occ 8 -> 16 k3
 *   dilation 2 p2; a 1-channel input is fed as 8 channels, 7 of them zero):
 * mf_conv3d_bf16_pack   fwd [Cout, ks^3, Cin]; dgrad_k4s2 (ks = 4 only); flipT [Cin, ks^3, Cout] = the forward operand
 *                       of a stride-1 layer's data-gradient convolution (dx = conv(dy, flipT), pad' = dil (ks-1) - pad);
 *                       input channels at or beyond w_cin pack as zeros
 * mf_conv3d_bf16_fwd    out rows have pitch ldo >= Cout (a column block of a wider channels-last grid)
 * mf_conv3d_bf16_fwd_ws the same with a workspace of mf_conv3d_bf16_fwd_workspace_bytes(...) bytes (0: not needed): a
 *                       layer with too few 256 x 256 output tiles for the chip (conv4 at 16 objects: 64) splits its
 *                       reduction over fp32 slabs in the workspace, added in order (deterministic) with bias / ReLU
 * mf_conv3d_bf16_wgrad  as above; only the channels below w_cin are written */
int mf_cast_rows_bf16(const float *src, int64_t src_ld, void *dst, int64_t dst_ld, int64_t rows, int32_t cols,
                      mfStream_t stream);
int mf_relu_mask_bf16(const void *y, const void *dy, const float *dy32, void *dz, int64_t n, mfStream_t stream);
int mf_linear_bf16(const void *A, int64_t a_group_stride, int32_t lda, const void *W, int64_t w_group_stride,
                   int32_t ldw, const float *bias, int64_t b_group_stride, void *out, int64_t o_group_stride,
                   int32_t ldo, int32_t M, int32_t N, int32_t K, int32_t groups, int32_t relu, int32_t out_f32,
                   int32_t accumulate, mfStream_t stream);
int mf_linear_wgrad_bf16(const void *dY, int64_t y_group_stride, int32_t ldy, const void *A, int64_t a_group_stride,
                         int32_t lda, float *dW, int64_t w_group_stride, int32_t ldc, void *ws, int32_t M, int32_t N,
                         int32_t K, int32_t groups, int32_t split, mfStream_t stream);
int mf_conv3d_k4s2_pack_bf16(const float *W, int32_t Cout, int32_t Cin, int32_t w_cin, int32_t c_off, void *fwd,
                             void *dgrad, mfStream_t stream);
int mf_conv3d_k4s2_bf16_fwd(const void *x, const void *wt, const float *bias, void *out, int32_t B, int32_t Cin,
                            int32_t Cout, int32_t D, int32_t relu, int32_t out_f32, mfStream_t stream);
int mf_conv3d_k4s2_bf16_dgrad(const void *dy, const void *wd, void *dx, int32_t B, int32_t Cin, int32_t Cout,
                              int32_t D, int32_t out_f32, int32_t accumulate, mfStream_t stream);
int64_t mf_conv3d_k4s2_bf16_wgrad_workspace_bytes(int32_t Cin, int32_t Cout, int32_t split);
int32_t mf_conv3d_k4s2_bf16_wgrad_default_split(int32_t B, int32_t Cin, int32_t Cout, int32_t D);
int mf_conv3d_k4s2_bf16_wgrad(const void *dy, const void *x, float *dW, void *ws, int32_t B, int32_t Cin,
                              int32_t Cout, int32_t D, int32_t w_cin, int32_t c_off, int32_t split,
                              mfStream_t stream);

int mf_conv3d_bf16_pack(const float *W, int32_t Cout, int32_t Cin, int32_t w_cin, int32_t c_off, int32_t ks, void *fwd,
                        void *dgrad_k4s2, void *flipT, mfStream_t stream);
int mf_conv3d_bf16_fwd(const void *x, const void *wt, const float *bias, void *out, int32_t B, int32_t Cin,
                       int32_t Cout, int32_t D, int32_t ks, int32_t stride, int32_t pad, int32_t dil, int32_t relu,
                       int32_t out_f32, int32_t ldo, mfStream_t stream);
int64_t mf_conv3d_bf16_fwd_workspace_bytes(int32_t B, int32_t Cin, int32_t Cout, int32_t D, int32_t ks, int32_t stride,
                                           int32_t pad, int32_t dil);
int mf_conv3d_bf16_fwd_ws(const void *x, const void *wt, const float *bias, void *out, void *ws, int64_t ws_bytes,
                          int32_t B, int32_t Cin, int32_t Cout, int32_t D, int32_t ks, int32_t stride, int32_t pad,
                          int32_t dil, int32_t relu, int32_t out_f32, int32_t ldo, mfStream_t stream);
/* 3 x 3 x 3, stride 1, pad = dilation convolutions between NARROW layers (read channels 8 or 16, written channels <= 16)
 * on channels-last bf16 grids -- the occupancy branch conv1_occ / conv2_occ (model.py:69-72,120-124) and conv2_occ's
 * data gradient (pack with transpose = 1): voxels are the MFMA's columns, operands straight from global memory.
 * wp: mf_conv3d_k3_narrow_bf16_pack_elems(CI) bf16, CI = channels of the tensor the convolution reads. */
int64_t mf_conv3d_k3_narrow_bf16_pack_elems(int32_t CI);
int mf_conv3d_k3_narrow_bf16_pack(const float *W, int32_t Cout, int32_t Cin, int32_t w_cin, int32_t c_off,
                                  int32_t transpose, void *wp, mfStream_t stream);
int mf_conv3d_k3_narrow_bf16(const void *x, const void *wp, const float *bias, void *out, int32_t B, int32_t CI,
                             int32_t CO, int32_t D, int32_t dil, int32_t relu, mfStream_t stream);
int64_t mf_conv3d_bf16_wgrad_workspace_bytes(int32_t Cin, int32_t Cout, int32_t ks, int32_t split);
int32_t mf_wgrad_split(int64_t tiles, int64_t ktiles, int64_t slab_bytes);
/* slabs mf_linear_wgrad_bf16 should be given for dW [N][K] over M rows (answers for the tile form that will run) */
int32_t mf_linear_wgrad_bf16_default_split(int64_t M, int32_t N, int32_t K, int32_t groups);
int32_t mf_conv3d_bf16_wgrad_default_split(int32_t B, int32_t Cin, int32_t Cout, int32_t Do, int32_t ks);
int mf_conv3d_bf16_wgrad(const void *dy, const void *x, float *dW, void *ws, int32_t B, int32_t Cin, int32_t Cout,
                         int32_t D, int32_t ks, int32_t stride, int32_t pad, int32_t dil, int32_t w_cin, int32_t c_off,
                         int32_t split, mfStream_t stream);

/* ---- conv3 of the bf16 training path on OCCUPIED VOXELS only (round 5; csrc/sparseconv_bf16.hip) -----------------
 * replaces the dense forward / backward-data / backward-filter of `L.Convolution3D(None, 256, 4, 2, pad=1)` over the
 * voxelized point features (contrib/singleview_3d/models/model.py:73,114-128; train.py:342-369): average_voxelization_3d
 * leaves <= 1000 of an object's 32768 voxels occupied in 144 of the 160 input channels.  Compact CLASS-MAJOR rows
 * (8 parity classes of the k4 / s2 / p1 geometry, each padded to a multiple of 128 rows):
 *   mf_sparse_conv3_bf16_index      points [n,3] (voxel frame: origin 0, pitch 1), batch_indices -> chains, row map,
 *                                   row -> voxel map, class row ranges, class of every 64-row block (all in ws)
 *   mf_sparse_conv3_bf16_tables     device addresses of those tables (see csrc/sparseconv_bf16.hip)
 *   mf_average_voxelization_rows_bf16_fwd / _bwd   voxel means into / gradients out of the compact rows A [rows, lda]
 *   mf_sparse_conv3_bf16_pack       W fp32 [Cout, w_cin, 4,4,4] channels c_off.. -> Wp bf16 [8][8 Cout][Cs] (forward /
 *                                   weight-gradient operand) and Wq bf16 [8][Cs][8 Cout] (data-gradient operand, or NULL)
 *   mf_linear_bf16_tiles            C = A Wp[class(row)]^T, the class of every 64-row block from the device table
 *   mf_sparse_conv3_bf16_reduce     out [B, (D/2)^3, Cout] bf16 = relu?(dense fp32 (or NULL) + bias + the <= 64
 *                                   (voxel, tap) contributions of every output voxel, in tap order)
 *   mf_sparse_conv3_bf16_gather_dy  dYg [rows, 8 Cout] = dz at the 8 output voxels each row feeds (zeros outside)
 *   mf_linear_wgrad_bf16_ranges     dWp[class] = dYg^T A over the class's row range (device table)
 *   mf_sparse_conv3_bf16_unpack_dw  dWp fp32 [8][8 Cout][Cs] -> dW fp32 [Cout, w_cin, 4,4,4] channels c_off..
 * rows = mf_sparse_conv3_bf16_max_rows(n) = n + 8 * 127 rounded up to 128.  Asynchronous, never allocate. */
int64_t mf_sparse_conv3_bf16_max_rows(int64_t n_points);
int64_t mf_sparse_conv3_bf16_workspace_bytes(int64_t n_points, int32_t B, int32_t D);
int mf_sparse_conv3_bf16_tables(void *ws, int64_t n_points, int32_t B, int32_t D, int64_t *out7);
int mf_sparse_conv3_bf16_index(const float *points, const int32_t *batch_indices, int64_t n, int32_t B, int32_t D,
                               void *ws, mfStream_t stream);
int mf_sparse_conv3_bf16_pack(const float *W, int32_t Cout, int32_t Cs, int32_t w_cin, int32_t c_off, void *Wp, void *Wq,
                              mfStream_t stream);
int mf_sparse_conv3_bf16_unpack_dw(const float *dWp, int32_t Cout, int32_t Cs, int32_t w_cin, int32_t c_off, float *dW,
                                   mfStream_t stream);
int mf_sparse_conv3_bf16_reduce(const void *C, const float *dense, const float *bias, void *ws, int64_t n_points,
                                int32_t B, int32_t D, int32_t Cout, int32_t relu, void *out, mfStream_t stream);
int mf_sparse_conv3_bf16_gather_dy(const void *dz, void *ws, int64_t n_points, int32_t B, int32_t D, int32_t Cout,
                                   void *dYg, mfStream_t stream);
int mf_linear_bf16_tiles(const void *A, int32_t lda, const void *W, int64_t w_group_stride, int32_t ldw,
                         const int32_t *tile_group, void *out, int32_t ldo, int32_t M, int32_t N, int32_t K,
                         int32_t out_f32, mfStream_t stream);
int mf_linear_wgrad_bf16_ranges(const void *dY, int32_t ldy, const void *A, int32_t lda, float *dW,
                                int64_t w_group_stride, int32_t ldc, const int32_t *m_range, int32_t groups, int32_t N,
                                int32_t K, mfStream_t stream);
/* Narrow-input data gradient of Convolution3D(.., 4, 2, pad=1) "columns first" (the 16 occupancy channels of conv3):
 * W2 bf16 [64 Cin][Cout]; T [B (D/2)^3][64 Cin] = dz W2^T through mf_linear_bf16; dx [B, D^3, Cin] = col2im gather. */
int mf_conv3d_k4s2_bf16_pack_cols(const float *W, int32_t Cout, int32_t Cin, int32_t w_cin, int32_t c_off, void *W2,
                                  mfStream_t stream);
int mf_conv3d_k4s2_bf16_col2im(const void *T, int32_t B, int32_t D, int32_t Cin, void *dx, mfStream_t stream);
int mf_average_voxelization_rows_bf16_fwd(const void *values, int64_t ldv, const float *points,
                                          const int32_t *batch_indices, int64_t n, int32_t C, int32_t B, int32_t D,
                                          const int32_t *counts, const int32_t *head, const int32_t *link,
                                          const int32_t *rowmap, void *A, int64_t lda, mfStream_t stream);
int mf_average_voxelization_rows_bf16_bwd(const void *dA, int64_t lda, const float *points,
                                          const int32_t *batch_indices, const int32_t *counts, const int32_t *rowmap,
                                          int64_t n, int32_t C, int32_t B, int32_t D, void *gvalues, int64_t ldg,
                                          mfStream_t stream);

/* Channels-last bf16 voxelization and trilinear sampling of the bf16 training path (K1/K2 and K5/K6 of SURVEY 2.1 --
 * functions/geometry/average_voxelization_3d.py:8-113, interpolate_voxel_grid.py:61-215 -- on the layout the bf16
 * convolutions consume; origin 0, pitch 1, cubic grids, as contrib/singleview_3d/models/model.py:113,131,141 call them):
 *   mf_average_voxelization_cl_bf16_fwd  values bf16 [n, ldv] -> x bf16 [B, D^3, ldx] columns [0, C): voxel means
 *     (fp32 sum in increasing point index), zeros elsewhere; counts / head [B*D^3], link [n]: int32 scratch
 *   mf_average_voxelization_cl_bf16_bwd  gvalues[p] = gx[b, voxel(p)] / count
 *   mf_interpolate_voxel_grid_cl_bf16_fwd / _bwd   vox bf16 [B, X*Y*Z, C] <-> rows bf16 [n, ld]; the backward gathers
 *     per voxel range (fp32 sums in LDS) and writes every element of gvox [B, X*Y*Z, C] once, as bf16 (out_bf16 = 1)
 *     or fp32; batch_start = B + 1 row offsets of points sorted by item, or NULL (any order, slower) */
int mf_average_voxelization_cl_bf16_fwd(const void *values, int64_t ldv, const float *points,
                                        const int32_t *batch_indices, int64_t n, int32_t C, int32_t B, int32_t D,
                                        void *x, int64_t ldx, int32_t *counts, int32_t *head, int32_t *link,
                                        mfStream_t stream);
int mf_average_voxelization_cl_bf16_bwd(const void *gx, int64_t ldx, const float *points, const int32_t *batch_indices,
                                        const int32_t *counts, int64_t n, int32_t C, int32_t B, int32_t D,
                                        void *gvalues, int64_t ldg, mfStream_t stream);
int mf_interpolate_voxel_grid_cl_bf16_fwd(const void *vox, const float *points, const int32_t *batch_indices,
                                          int64_t n, int B, int C, int X, int Y, int Z, void *out, int64_t ldo,
                                          mfStream_t stream);
int mf_interpolate_voxel_grid_cl_bf16_bwd(const void *gout, int64_t ldg, const float *points,
                                          const int32_t *batch_indices, const int32_t *batch_start, int64_t n, int B,
                                          int C, int X, int Y, int Z, void *gvox, int32_t out_bf16, mfStream_t stream);

/* Element-wise pieces of the 2-D backbone's decoder (morefusion/models/dense_fusion/pspnet.py:10-35,40-73:
 * F.resize_images bilinear align_corners, L.PReLU with one slope), forward and backward, channels-last tensors
 * [B, H, W, C], float32 (bf16 = 0) or bfloat16 (bf16 = 1), C % 8 == 0:
 *   mf_upsample_bilinear_cl_fwd  y [B, Ho, Wo, C] from x [B, H, W, C]
 *   mf_upsample_bilinear_cl_bwd  gx from gy: a gather in increasing (oy, ox) -- deterministic, no atomics
 *   mf_prelu_fwd / mf_prelu_bwd  y = x > 0 ? x : a x;  dx and dslope[0] = sum_{x <= 0} dy x (ws: mf_prelu_bwd_workspace_floats) */
int mf_upsample_bilinear_cl_fwd(const void *x, void *y, int32_t B, int32_t H, int32_t W, int32_t Ho, int32_t Wo,
                                int32_t C, int32_t bf16, mfStream_t stream);
int mf_upsample_bilinear_cl_bwd(const void *gy, void *gx, int32_t B, int32_t H, int32_t W, int32_t Ho, int32_t Wo,
                                int32_t C, int32_t bf16, mfStream_t stream);
/* ... and for channels-first tensors [B*C, H, W] (one lane per element) */
int mf_upsample_bilinear_cf_fwd(const void *x, void *y, int64_t BC, int32_t H, int32_t W, int32_t Ho, int32_t Wo,
                                int32_t bf16, mfStream_t stream);
int mf_upsample_bilinear_cf_bwd(const void *gy, void *gx, int64_t BC, int32_t H, int32_t W, int32_t Ho, int32_t Wo,
                                int32_t bf16, mfStream_t stream);
int mf_prelu_fwd(const void *x, const float *slope, void *y, int64_t n, int32_t bf16, mfStream_t stream);
/* ResNet18Extractor's input normalisation (models/resnet.py:33-36): (rgb / 255 - mean) / std on a [B, H, W, 3] image,
 * uint8 (u8 = 1) or float32 -> float32 [B, H, W, 3]; mean3 / std3 are HOST arrays of three floats. */
int mf_rgb_normalize(const void *rgb, int32_t u8, const float *mean3, const float *std3, float *out, int64_t npix,
                     mfStream_t stream);
/* BatchNorm with inference statistics (+ residual add) (+ ReLU) in one launch (models/resnet.py:44: ResNet18Extractor's
 * BatchNorm never updates): y = relu?((x - mean) * (weight / sqrt(var + eps)) + bias (+ identity)) over a dense
 * [B, C, H, W] tensor, NCHW (8 | H W) or channels-last (8 | C); fp32 / bf16 activations, fp32 parameters. */
int mf_bn_act_fwd(const void *x, const void *identity, const float *mean, const float *var, const float *weight,
                  const float *bias, float eps, void *y, int64_t n, int32_t C, int64_t HW, int32_t channels_last,
                  int32_t relu, int32_t bf16, mfStream_t stream);
int64_t mf_prelu_bwd_workspace_floats(int64_t n);
int mf_prelu_bwd(const void *x, const void *dy, const float *slope, void *dx, float *dslope, float *ws, int64_t n,
                 int32_t bf16, mfStream_t stream);

/* Point-wise prologue / epilogue of the volumetric part (inference), one launch each instead of ~25 torch launches:
 *   mf_point_prep: camera-frame points [B,3,P] + image features [B,Cv,P] -> voxel-frame points [n,3]
 *     ((p - origin) / pitch, model.py:236), to_center [n,4] = (center - p | 0) (:101), feature rows [n,Cv],
 *     batch indices [n];  n = B*P, Cv % 4 == 0.
 *   mf_pose_epilogue: heads' output rows [n, ldo] (rot at column 0, trans at np4, conf at 2*np4; class c at
 *     4c / 3c / c) -> rot [n,4] = q / (|q| + 1e-5) (chainer F.normalize), trans [n,3] = (p*pitch + origin) +
 *     t*pitch (:264-266), conf [n] = sigmoid (:262) of each object's class (class_id int64 [B], 1-based; an id
 *     outside 1 .. n_fg gives NaN outputs, never a read outside the row). */
/* Tile height (64 / 128 / 256 rows) the bf16 NT engine used for its most recent launch in this process (the
 * 256 x 256 form -- eight waves of 128 x 64 -- takes problems with >= 224 such tiles; MF_NT_BIG = 0 / 2 in the
 * environment forces never / wherever possible). */
int mf_gemm_bf16_last_tile(void);

/* PSPNet's sampled tail under bf16 training (pspnet.py:18-22,50-56 at model.py:222's pixels): the 3 x 3 windows of
 * the virtually x2 up-sampled map as GEMM rows [B * P, 576] bf16 (column c * 9 + ky * 3 + kx) from the channels-last
 * bf16 map u2 [B, H, W, 64], pix [B * P] flat indices into [2H, 2W]; and the backward: grows -> gu2 [B, H, W, 64] bf16
 * through the fp32 workspace acc [B, H, W, 64] (zeroed by the call; fp32 atomics). */
int mf_psp_tail_rows_bf16_fwd(const void *u2, const int64_t *pix, int32_t B, int32_t P, int32_t H, int32_t W, void *rows,
                              mfStream_t stream);
int mf_psp_tail_rows_bf16_bwd(const void *grows, const int64_t *pix, int32_t B, int32_t P, int32_t H, int32_t W,
                              float *acc, void *gu2, mfStream_t stream);

/* The confidence terms of the pose loss and their reduction (contrib/singleview_3d/models/model.py:417-434):
 * loss[0] = mean over objects of the mean over confident points (conf > 0) of add * conf - lambda * log(conf);
 * cnt [B] is kept for the backward (dadd, dconf [B, P] from the scalar gradient gloss[0]). */
int mf_confidence_loss_fwd(const float *add, const float *conf, int32_t B, int32_t P, float lambda, float *loss,
                           int32_t *cnt, mfStream_t stream);
int mf_confidence_loss_bwd(const float *add, const float *conf, const int32_t *cnt, const float *gloss, int32_t B,
                           int32_t P, float lambda, float *dadd, float *dconf, mfStream_t stream);

/* Pose epilogue of the TRAINING path (model.py:262-273: class selection, F.normalize, translation, sigmoid) on the
 * three heads' fp32 outputs orot [n, 4 n_fg], otrn [n, 3 n_fg], ocnf [n, n_fg]; the backward writes the three gradient
 * row blocks completely.  One launch each (torch: ~27 / ~47 incl. three index_put sorts). */
int mf_pose_epilogue_train_fwd(const float *orot, const float *otrn, const float *ocnf, const int64_t *class_id,
                               const float *pts, const float *origin, const float *pitch, int32_t B, int32_t P,
                               int32_t n_fg, float *rot, float *trans, float *conf, mfStream_t stream);
int mf_pose_epilogue_train_bwd(const float *orot, const float *ocnf, const int64_t *class_id, const float *pitch,
                               const float *grot, const float *gtrans, const float *gconf, int32_t B, int32_t P,
                               int32_t n_fg, float *drot, float *dtrn, float *dcnf, mfStream_t stream);
/* transformation_matrix of a batch of poses and its backward (functions/geometry/transformation_matrix.py:5-18 =
 * quaternion_matrix.py:36-78 + compose_transform.py:5-48): T [n,4,4] row-major from q [n,4] (wxyz, any norm), t [n,3];
 * gq [n,4], gt [n,3] from gT [n,4,4].  One launch each (the torch composite: ~25 / ~60). */
int mf_transformation_matrix_fwd(const float *q, const float *t, int64_t n, float *T, mfStream_t stream);
int mf_transformation_matrix_bwd(const float *q, const float *gT, int64_t n, float *gq, float *gt, mfStream_t stream);
int mf_point_prep(const float *points_cam, const float *values, const float *origin, const float *pitch,
                  int32_t B, int32_t P, int32_t Cv, float center, float *pts, float *tc4, float *x_rows,
                  int32_t *batch_indices, mfStream_t stream);
int mf_pose_epilogue(const float *heads_out, int64_t ldo, int32_t np4, const int64_t *class_id,
                     const float *pts, const float *origin, const float *pitch, int32_t B, int32_t P, int32_t n_fg,
                     float *rot, float *trans, float *conf, mfStream_t stream);

/* The last PSPNet level (up3: bilinear x2 + Convolution2D 3x3 64->64 + PReLU; conv1 1x1 64->32; log-softmax,
 *   morefusion/models/dense_fusion/pspnet.py:10-35,57-73) evaluated ONLY at the sampled pixels the pose network
 *   reads (contrib/singleview_3d/models/model.py:222), one launch:
 *   u2 [B,64,H,W] with element strides (sb, sc, sy, sx) (NCHW or channels-last), pix [B*P] int64 flat indices into
 *   the [2H,2W] full-resolution map, w3t [9,64,64] = W3.permute(2,3,1,0), w1t [64,32] = W1[:, :, 0, 0].T,
 *   prelu_slope: device pointer to the single PReLU parameter -> out [B*P, 32] (rows). */
int mf_psp_tail_fwd(const float *u2, int64_t sb, int64_t sc, int64_t sy, int64_t sx, const int64_t *pix,
                    const float *w3t, const float *b3, const float *prelu_slope, const float *w1t,
                    const float *b1, int32_t B, int32_t P, int32_t H, int32_t W, float *out, mfStream_t stream);

/* small fused helpers of the same path */
/* pack [Ptot,3] points + [Ptot] sdf into float4 */
int mf_pack_points_sdf(const float *points, const float *sdf, int64_t n, void *pts4,
                       mfStream_t stream);

/* ---- pre-processing in front of the network ------------------
 * replaces the per-instance host loop (NumPy + imgviz/cv2) at
 *   ros/src/morefusion_ros/nodes/singleview_3d_pose_estimation.py:116-176
 *   morefusion/datasets/rgbd_pose_estimation/base.py:112-137
 * incl. geometry/pointcloud_from_depth.py:4-26 and geometry/masks_to_bboxes.py:4-38.
 *   label [H,W] int32 instance image, depth [H,W] float32 metres (NaN = invalid),
 *   rgb [H,W,3] uint8, instance_ids [n] (n <= 256).
 * mf_instance_stats: stats [n,6] = {y1, x1, y2, x2 (end-exclusive box of label == id),
 *   mask pixels, mask pixels with valid depth}; a mask without pixels leaves
 *   {INT_MAX, INT_MAX, 0, 0, 0, 0}.
 * mf_instance_crops: per instance the masked crop centerized to S x S:
 *   rgb_out [n,S,S,3] uint8 (0 outside the mask / padding; 8-bit bilinear resize),
 *   pcd_out [n,S,S,3] float32 camera-frame points (NaN outside the mask / padding / invalid
 *   depth; nearest-neighbour resize; back-projection in float64), keep [n] uint8 = mask is
 *   non-empty and has >= min_valid valid points (the reference skips the others; here
 *   their outputs are all padding).  Both calls are asynchronous and never synchronise. */
int mf_instance_stats(const int32_t *label, const float *depth, int H, int W,
                      const int32_t *instance_ids, int n_inst, int32_t *stats,
                      mfStream_t stream);
int mf_instance_crops(const uint8_t *rgb, const float *depth, const int32_t *label, int H,
                      int W, double fx, double fy, double cx, double cy,
                      const int32_t *instance_ids, const int32_t *stats, int n_inst, int S,
                      int min_valid, uint8_t *rgb_out, float *pcd_out, uint8_t *keep,
                      mfStream_t stream);

/* valid-pixel list in front of Model.predict's point selection: replaces, per object,
 *   contrib/singleview_3d/models/model.py:195-196 `iy, ix = xp.where(mask[i])` and :206
 *   `n_point = int(mask[i].sum())` with mask = ~isnan(pcd).any(channel).
 * pcd [B,HW,3] float32 -> order [B,HW] int32: the row-major
 * pixel indices h*W+w of the pixels without a NaN coordinate, in increasing order (entries beyond
 * counts[b] are left untouched), counts [B] int32.  The reference's NumPy-RNG subsample
 * (`keep`, :207-219) then indexes this list.  Asynchronous, never synchronises. */
int mf_valid_pixel_order(const float *pcd, int32_t B, int32_t HW, int32_t *order, int32_t *counts,
                         mfStream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* MFHIP_H_ */
