/*
 * mf_oracle.c -- plain-C restatement of the reference algorithms on the hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Not part of the product: only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load the library
 * built from this file (oracle/_build/libmforacle.so), and only as the checker
 * or as the timed CPU baseline ("kind": "port").
 *
 * Second, independent restatement (the first is oracle/oracle_np.py; the two are
 * required to agree bit-for-bit in tests/test_oracle_c.py).  Scalar loops in the
 * order of the reference's CUDA thread index so that "lowest flat index wins" is
 * simply "first writer with a strictly smaller distance wins".
 *
 * Citations are relative to /root/reference/.
 * Build: see oracle/Makefile (gcc -O2 -ffp-contract=off: IEEE float32, no FMA).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#ifdef _OPENMP
#include <omp.h>
#endif

/* CUDA round(): half away from zero.  roundf() has exactly this semantic. */
static inline float round_away(float x) { return roundf(x); }

int mfo_version(void) { return 1; }

void mfo_set_threads(int n) {
#ifdef _OPENMP
  omp_set_num_threads(n > 0 ? n : 1);
#else
  (void)n;
#endif
}

int mfo_max_threads(void) {
#ifdef _OPENMP
  return omp_get_max_threads();
#else
  return 1;
#endif
}

/* ------------------------------------------------------------------------
 * A1 average_voxelization_3d, GPU fork (float32, half-away), CPU loop order.
 * morefusion/functions/geometry/average_voxelization_3d.py:8-40 (loop),
 * :80-98 (index arithmetic).  matrix [B,C,X,Y,Z], counts [B,X,Y,Z]; both
 * zero-initialised here.
 * ---------------------------------------------------------------------- */
void mfo_average_voxelization_3d(const float *values, const float *points,
                                 const int32_t *batch_indices, int64_t n, int C,
                                 int B, int X, int Y, int Z, const float *origin,
                                 float pitch, float *matrix, int32_t *counts) {
  const int64_t V = (int64_t)X * Y * Z;
  memset(matrix, 0, sizeof(float) * B * C * V);
  memset(counts, 0, sizeof(int32_t) * B * V);
  for (int64_t i = 0; i < n; ++i) {
    int ix = (int)round_away((points[3 * i + 0] - origin[0]) / pitch);
    int iy = (int)round_away((points[3 * i + 1] - origin[1]) / pitch);
    int iz = (int)round_away((points[3 * i + 2] - origin[2]) / pitch);
    if (ix < 0 || ix >= X || iy < 0 || iy >= Y || iz < 0 || iz >= Z) continue;
    int b = batch_indices[i];
    int64_t v = ((int64_t)ix * Y + iy) * Z + iz;
    for (int c = 0; c < C; ++c) matrix[((int64_t)b * C + c) * V + v] += values[i * C + c];
    counts[b * V + v] += 1;
  }
  for (int b = 0; b < B; ++b)
    for (int64_t v = 0; v < V; ++v) {
      int32_t k = counts[b * V + v];
      if (k > 0)
        for (int c = 0; c < C; ++c) matrix[((int64_t)b * C + c) * V + v] /= (float)k;
    }
}

/* ------------------------------------------------------------------------
 * A4 interpolate_voxel_grid forward, GPU fork (trunc toward zero).
 * morefusion/functions/geometry/interpolate_voxel_grid.py:6-59, :170-212.
 * ---------------------------------------------------------------------- */
void mfo_interpolate_voxel_grid(const float *vox, const float *points,
                                const int32_t *batch_indices, int64_t n, int C, int X,
                                int Y, int Z, float *values) {
  static const int CO[8][3] = {{0, 0, 0}, {1, 0, 0}, {0, 1, 0}, {0, 0, 1},
                               {1, 1, 0}, {0, 1, 1}, {1, 0, 1}, {1, 1, 1}};
  const int64_t V = (int64_t)X * Y * Z;
  for (int64_t i = 0; i < n; ++i) {
    float px = points[3 * i], py = points[3 * i + 1], pz = points[3 * i + 2];
    int lx = (int)px, ly = (int)py, lz = (int)pz;
    float fx = px - (float)lx, fy = py - (float)ly, fz = pz - (float)lz;
    float hx = 1.0f - fx, hy = 1.0f - fy, hz = 1.0f - fz;
    int b = batch_indices[i];
    for (int c = 0; c < C; ++c) values[i * C + c] = 0.0f;
    for (int j = 0; j < 8; ++j) {
      int ix = lx + CO[j][0], iy = ly + CO[j][1], iz = lz + CO[j][2];
      if (ix < 0 || ix >= X || iy < 0 || iy >= Y || iz < 0 || iz >= Z) continue;
      float w = ((CO[j][0] ? fx : hx) * (CO[j][1] ? fy : hy)) * (CO[j][2] ? fz : hz);
      int64_t v = ((int64_t)ix * Y + iy) * Z + iz;
      for (int c = 0; c < C; ++c)
        values[i * C + c] += w * vox[((int64_t)b * C + c) * V + v];
    }
  }
}

/* ------------------------------------------------------------------------
 * A5 occupancy_grid_3d forward.
 * morefusion/functions/geometry/occupancy_grid_3d.py:31-54, :77-85.
 * ---------------------------------------------------------------------- */
void mfo_occupancy_grid_3d(const float *points, int64_t P, float pitch,
                           const float *origin, int X, int Y, int Z, float threshold,
                           float *grid) {
  float *pf = (float *)malloc(sizeof(float) * 3 * P);
  for (int64_t p = 0; p < P; ++p)
    for (int d = 0; d < 3; ++d) pf[3 * p + d] = (points[3 * p + d] - origin[d]) / pitch;
#pragma omp parallel for collapse(2)
  for (int i = 0; i < X; ++i)
    for (int j = 0; j < Y; ++j)
      for (int k = 0; k < Z; ++k) {
        float dmin = INFINITY;
        for (int64_t p = 0; p < P; ++p) {
          float a = (float)i - pf[3 * p], b = (float)j - pf[3 * p + 1],
                c = (float)k - pf[3 * p + 2];
          float d = sqrtf((a * a + b * b) + c * c);
          if (d < dmin) dmin = d;
        }
        float m = threshold - dmin;
        m = m > 0.0f ? m : 0.0f;
        grid[((int64_t)i * Y + j) * Z + k] = m < 1.0f ? m : 1.0f;
      }
  free(pf);
}

/* ------------------------------------------------------------------------
 * A11 nn: brute-force 1-NN, squared distance accumulated x,y,z, first minimum.
 * morefusion/geometry/knn/cuComputeDistanceGlobal.cu:62-72 + knn/nn.py:48.
 * ---------------------------------------------------------------------- */
void mfo_nn(const float *ref, int64_t R, const float *query, int64_t Q, int64_t *out) {
#pragma omp parallel for
  for (int64_t q = 0; q < Q; ++q) {
    float best = INFINITY;
    int64_t bi = 0;
    for (int64_t r = 0; r < R; ++r) {
      float ssd = 0.0f;
      for (int d = 0; d < 3; ++d) {
        float tmp = ref[3 * r + d] - query[3 * q + d];
        ssd += tmp * tmp;
      }
      if (ssd < best) { best = ssd; bi = r; }
    }
    out[q] = bi;
  }
}

/* ------------------------------------------------------------------------
 * A6 truncated_distance_function forward, ksize = 3 generalised to any odd ks.
 * morefusion/functions/geometry/truncated_distance_function.py:33-41 (kernel
 * offsets: flat k=(a*ks+b)*ks+c -> (b,a,c)-ks/2), :51-79 (CUDA body).
 * tdf pre-filled with truncation, flat with -1 by this function.
 * Sequential in flat index i = p*K+k with strict '<' == lowest index among ties.
 * ---------------------------------------------------------------------- */
static int ksize_of(float pitch, float trunc) {
  int ks = (int)ceilf(trunc / pitch);
  if (ks % 2 == 0) ks += 1;
  return ks;
}

void mfo_tdf(const float *points, int64_t P, float pitch, const float *origin, int X,
             int Y, int Z, float trunc, float *tdf, int64_t *flat) {
  const int64_t V = (int64_t)X * Y * Z;
  const int ks = ksize_of(pitch, trunc), h = ks / 2, K = ks * ks * ks;
  for (int64_t v = 0; v < V; ++v) { tdf[v] = trunc; flat[v] = -1; }
  for (int64_t p = 0; p < P; ++p) {
    float fx = (points[3 * p] - origin[0]) / pitch;
    float fy = (points[3 * p + 1] - origin[1]) / pitch;
    float fz = (points[3 * p + 2] - origin[2]) / pitch;
    float rx = round_away(fx), ry = round_away(fy), rz = round_away(fz);
    for (int k = 0; k < K; ++k) {
      int a = k / (ks * ks), b = (k / ks) % ks, c = k % ks;
      int ix = (int)(rx + (float)(b - h));
      int iy = (int)(ry + (float)(a - h));
      int iz = (int)(rz + (float)(c - h));
      if (ix < 0 || ix >= X || iy < 0 || iy >= Y || iz < 0 || iz >= Z) continue;
      float dx = fx - (float)ix, dy = fy - (float)iy, dz = fz - (float)iz;
      float dist = pitch * sqrtf((dx * dx + dy * dy) + dz * dz);
      if (dist < trunc) {
        int64_t v = ((int64_t)ix * Y + iy) * Z + iz;
        if (dist < tdf[v]) { tdf[v] = dist; flat[v] = p * K + k; }
      }
    }
  }
}

/* ------------------------------------------------------------------------
 * quaternion (wxyz, any norm) -> R ; and the chain-rule backward.
 * morefusion/functions/geometry/quaternion_matrix.py:14-34, :36-51, :65-78.
 * ---------------------------------------------------------------------- */
static void quat_to_R(const float *q, float R[9]) {
  float n = ((q[0] * q[0] + q[1] * q[1]) + q[2] * q[2]) + q[3] * q[3];
  float s = sqrtf(2.0f / n);
  float qs[4] = {q[0] * s, q[1] * s, q[2] * s, q[3] * s};
  float Q[4][4];
  for (int i = 0; i < 4; ++i)
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

static void quat_backward(const float *q, const float gR[9], float gq[4]) {
  float gQ[4][4];
  memset(gQ, 0, sizeof(gQ));
  gQ[1][0] = -gR[5] + gR[7];
  gQ[1][1] = -gR[4] - gR[8];
  gQ[1][2] = gR[1] + gR[3];
  gQ[1][3] = gR[2] + gR[6];
  gQ[2][0] = gR[2] - gR[6];
  gQ[2][2] = -gR[0] - gR[8];
  gQ[2][3] = gR[5] + gR[7];
  gQ[3][0] = -gR[1] + gR[3];
  gQ[3][3] = -gR[0] - gR[4];
  float n = ((q[0] * q[0] + q[1] * q[1]) + q[2] * q[2]) + q[3] * q[3];
  float s = sqrtf(2.0f / n);
  float qs[4] = {q[0] * s, q[1] * s, q[2] * s, q[3] * s};
  float gqs[4];
  for (int i = 0; i < 4; ++i) {
    float a = 0.0f, b = 0.0f;
    for (int j = 0; j < 4; ++j) { a += gQ[i][j] * qs[j]; b += gQ[j][i] * qs[j]; }
    gqs[i] = a + b;
  }
  float dot = ((gqs[0] * q[0] + gqs[1] * q[1]) + gqs[2] * q[2]) + gqs[3] * q[3];
  for (int i = 0; i < 4; ++i) gq[i] = s * gqs[i] - (s / n) * dot * q[i];
}

/* ------------------------------------------------------------------------
 * A9 IterativeCollisionCheckLink: loss and d loss / d (q, t) for one scene.
 * morefusion/contrib/iterative_collision_check_link.py:31-99 with
 * pseudo_occupancy_voxelization (truncated_distance_function.py:181-213) and the
 * TDF backward (:121-146) inlined.
 *
 * points: concatenated model points [Ptot,3]; sdf [Ptot]; offs [N+1] segment
 * offsets; pitch [N]; origin [N,3]; grid_target, grid_nontarget_empty [N,D,D,D];
 * q [N,4]; t [N,3].  Outputs gq [N,4], gt [N,3], sums[4] = {S_t, RN, S_in, PN}.
 * Returns the loss.  Sums are accumulated in double and rounded once (the
 * reference's reduction order is unspecified; tests compare with tolerance).
 * ---------------------------------------------------------------------- */
typedef struct {
  float *tdf;     /* [V] */
  int64_t *flat;  /* [V] winner flat index (global point id * 27 + k) */
  float *w_in, *w_surf;
  float M;
} grid_t;

static void pseudo_weights(grid_t *g, int64_t V, const float *sdf, float offset, int K) {
  float M = -INFINITY;
  for (int64_t v = 0; v < V; ++v) {
    float w = g->flat[v] >= 0 ? sdf[g->flat[v] / K] : -1.0f;
    w += offset;
    int neg = w < 0.0f;
    if (neg) w = 0.0f;
    g->w_in[v] = w;
    g->w_surf[v] = neg ? 0.0f : 1.0f; /* marker, finished below */
    if (w > M) M = w;
  }
  g->M = M;
  for (int64_t v = 0; v < V; ++v) {
    float wi = g->w_in[v] / M; /* 0/0 -> NaN like the reference */
    g->w_surf[v] = g->w_surf[v] != 0.0f ? 1.0f - wi : wi;
    g->w_in[v] = wi;
  }
}

/* TDF over a subset of the scene's world points: all points with object id == only
 * (only >= 0) or != skip (skip >= 0), scanned in global order. */
static void tdf_subset(const float *pw, const int32_t *obj, int64_t Ptot, int only,
                       int skip, float pitch, const float *origin, int D, float trunc,
                       float *tdf, int64_t *flat) {
  const int64_t V = (int64_t)D * D * D;
  const int ks = ksize_of(pitch, trunc), h = ks / 2, K = ks * ks * ks;
  for (int64_t v = 0; v < V; ++v) { tdf[v] = trunc; flat[v] = -1; }
  for (int64_t p = 0; p < Ptot; ++p) {
    if (only >= 0 && obj[p] != only) continue;
    if (skip >= 0 && obj[p] == skip) continue;
    float fx = (pw[3 * p] - origin[0]) / pitch;
    float fy = (pw[3 * p + 1] - origin[1]) / pitch;
    float fz = (pw[3 * p + 2] - origin[2]) / pitch;
    float rx = round_away(fx), ry = round_away(fy), rz = round_away(fz);
    if (rx < -(float)h || rx > (float)(D - 1 + h) || ry < -(float)h ||
        ry > (float)(D - 1 + h) || rz < -(float)h || rz > (float)(D - 1 + h))
      continue; /* whole neighbourhood out of the grid */
    for (int k = 0; k < K; ++k) {
      int a = k / (ks * ks), b = (k / ks) % ks, c = k % ks;
      int ix = (int)(rx + (float)(b - h));
      int iy = (int)(ry + (float)(a - h));
      int iz = (int)(rz + (float)(c - h));
      if (ix < 0 || ix >= D || iy < 0 || iy >= D || iz < 0 || iz >= D) continue;
      float dx = fx - (float)ix, dy = fy - (float)iy, dz = fz - (float)iz;
      float dist = pitch * sqrtf((dx * dx + dy * dy) + dz * dz);
      if (dist < trunc) {
        int64_t v = ((int64_t)ix * D + iy) * D + iz;
        if (dist < tdf[v]) { tdf[v] = dist; flat[v] = p * K + k; }
      }
    }
  }
}

float mfo_icc_loss_grad(const float *points, const float *sdf, const int64_t *offs,
                        int N, const float *pitch, const float *origin,
                        const float *grid_target, const float *grid_ne, const float *q,
                        const float *t, int D, float voxel_threshold, float sdf_offset,
                        float *gq, float *gt, float *sums) {
  const int64_t V = (int64_t)D * D * D;
  const int64_t Ptot = offs[N];
  float *pw = (float *)malloc(sizeof(float) * 3 * Ptot);
  float *gpw = (float *)calloc(3 * Ptot, sizeof(float));
  int32_t *obj = (int32_t *)malloc(sizeof(int32_t) * Ptot);
  float *R = (float *)malloc(sizeof(float) * 9 * N);
  for (int i = 0; i < N; ++i) {
    quat_to_R(q + 4 * i, R + 9 * i);
    const float *Ri = R + 9 * i, *ti = t + 3 * i;
    for (int64_t p = offs[i]; p < offs[i + 1]; ++p) {
      float x = points[3 * p], y = points[3 * p + 1], z = points[3 * p + 2];
      for (int d = 0; d < 3; ++d)
        pw[3 * p + d] = ((Ri[3 * d] * x + Ri[3 * d + 1] * y) + Ri[3 * d + 2] * z) + ti[d];
      obj[p] = i;
    }
  }
  grid_t *own = (grid_t *)malloc(sizeof(grid_t) * N);
  grid_t *oth = (grid_t *)malloc(sizeof(grid_t) * N);
  float *trunc = (float *)malloc(sizeof(float) * N);
  int *K = (int *)malloc(sizeof(int) * N);
#pragma omp parallel for schedule(dynamic, 1)
  for (int gi = 0; gi < 2 * N; ++gi) {
    int i = gi >> 1, other = gi & 1;
    grid_t *g = other ? &oth[i] : &own[i];
    g->tdf = (float *)malloc(sizeof(float) * V);
    g->flat = (int64_t *)malloc(sizeof(int64_t) * V);
    g->w_in = (float *)malloc(sizeof(float) * V);
    g->w_surf = (float *)malloc(sizeof(float) * V);
    float tr = voxel_threshold * pitch[i];
    int ks = ksize_of(pitch[i], tr);
    if (!other) { trunc[i] = tr; K[i] = ks * ks * ks; }
    if (other && N <= 1) { g->M = 0.0f; continue; }
    tdf_subset(pw, obj, Ptot, other ? -1 : i, other ? i : -1, pitch[i], origin + 3 * i, D,
               tr, g->tdf, g->flat);
    pseudo_weights(g, V, sdf, other ? 0.0f : sdf_offset, ks * ks * ks);
  }
  /* sums (iterative_collision_check_link.py:91-98) */
  double S_t = 0, RN = 0, S_in = 0, PN = 0;
  int *use_oth = (int *)calloc(N, sizeof(int));
  for (int i = 0; i < N; ++i) {
    if (N > 1) {
      /* NaN guard (:82): grid_other is NaN everywhere iff M is 0 (0/0) */
      int has_nan = 0;
      for (int64_t v = 0; v < V && !has_nan; ++v) {
        float go = (1.0f - oth[i].tdf[v] / trunc[i]) * oth[i].w_in[v];
        if (go != go) has_nan = 1;
      }
      use_oth[i] = !has_nan;
    }
    for (int64_t v = 0; v < V; ++v) {
      float g = 1.0f - own[i].tdf[v] / trunc[i];
      float surf = g * own[i].w_surf[v], ins = g * own[i].w_in[v];
      float ne = grid_ne[i * V + v];
      if (use_oth[i]) {
        float go = (1.0f - oth[i].tdf[v] / trunc[i]) * oth[i].w_in[v];
        if (!(ne >= go)) ne = go; /* maximum(ne, go) */
      }
      float tg = grid_target[i * V + v];
      S_t += tg;
      RN += (double)(surf * tg);
      S_in += ins;
      PN += (double)(ins * ne);
    }
  }
  float fS_t = (float)S_t, fRN = (float)RN, fS_in = (float)S_in, fPN = (float)PN;
  sums[0] = fS_t; sums[1] = fRN; sums[2] = fS_in; sums[3] = fPN;
  float loss = fPN / fS_in - fRN / fS_t;
  /* backward */
  for (int i = 0; i < N; ++i) {
    const float *o = origin + 3 * i;
    for (int pass = 0; pass < 2; ++pass) {
      if (pass == 1 && !use_oth[i]) continue;
      grid_t *g = pass ? &oth[i] : &own[i];
      for (int64_t v = 0; v < V; ++v) {
        if (g->flat[v] < 0) continue;
        float gg = 1.0f - own[i].tdf[v] / trunc[i];
        float ins = gg * own[i].w_in[v];
        float ne = grid_ne[i * V + v];
        float g_grid;
        if (!pass) {
          float ne_eff = ne;
          if (use_oth[i]) {
            float go = (1.0f - oth[i].tdf[v] / trunc[i]) * oth[i].w_in[v];
            if (!(ne >= go)) ne_eff = go;
          }
          float g_surf = -grid_target[i * V + v] / fS_t;
          float g_ins = ne_eff / fS_in - fPN / (fS_in * fS_in);
          g_grid = own[i].w_surf[v] * g_surf + own[i].w_in[v] * g_ins;
        } else {
          float go = (1.0f - oth[i].tdf[v] / trunc[i]) * oth[i].w_in[v];
          float g_oth = (ne >= go) ? 0.0f : ins / fS_in;
          g_grid = oth[i].w_in[v] * g_oth;
        }
        float g_tdf = -g_grid / trunc[i];
        int64_t p = g->flat[v] / K[i];
        int ix = (int)(v / ((int64_t)D * D)), iy = (int)((v / D) % D), iz = (int)(v % D);
        float dx = (pw[3 * p] - o[0]) / pitch[i] - (float)ix;
        float dy = (pw[3 * p + 1] - o[1]) / pitch[i] - (float)iy;
        float dz = (pw[3 * p + 2] - o[2]) / pitch[i] - (float)iz;
        float nrm = sqrtf((dx * dx + dy * dy) + dz * dz);
        if (nrm > 0.0f) {
          gpw[3 * p] += dx / nrm * g_tdf;
          gpw[3 * p + 1] += dy / nrm * g_tdf;
          gpw[3 * p + 2] += dz / nrm * g_tdf;
        }
      }
    }
  }
  for (int i = 0; i < N; ++i) {
    double gRd[9] = {0}, gtd[3] = {0};
    for (int64_t p = offs[i]; p < offs[i + 1]; ++p)
      for (int a = 0; a < 3; ++a) {
        gtd[a] += gpw[3 * p + a];
        for (int b = 0; b < 3; ++b) gRd[3 * a + b] += (double)(gpw[3 * p + a] * points[3 * p + b]);
      }
    float gR[9];
    for (int k = 0; k < 9; ++k) gR[k] = (float)gRd[k];
    for (int a = 0; a < 3; ++a) gt[3 * i + a] = (float)gtd[a];
    quat_backward(q + 4 * i, gR, gq + 4 * i);
  }
  for (int i = 0; i < N; ++i) {
    free(own[i].tdf); free(own[i].flat); free(own[i].w_in); free(own[i].w_surf);
    free(oth[i].tdf); free(oth[i].flat); free(oth[i].w_in); free(oth[i].w_surf);
  }
  free(own); free(oth); free(trunc); free(K); free(use_oth);
  free(pw); free(gpw); free(obj); free(R);
  return loss;
}

/* ------------------------------------------------------------------------
 * Chainer Adam step (third party, chainer/optimizers/adam.py v7; parity
 * unpinned): m += (1-b1)(g-m); v += (1-b2)(g*g-v); p -= alpha_t*m/(sqrt(v)+eps)
 * with alpha_t = alpha*sqrt(1-b2^t)/(1-b1^t) evaluated in double.
 * ---------------------------------------------------------------------- */
void mfo_adam_step(float *p, const float *g, float *m, float *v, int64_t n, double alpha,
                   int t) {
  const double b1 = 0.9, b2 = 0.999;
  float alpha_t = (float)(alpha * sqrt(1.0 - pow(b2, t)) / (1.0 - pow(b1, t)));
  float omb1 = (float)(1 - b1), omb2 = (float)(1 - b2), eps = 1e-8f;
  for (int64_t i = 0; i < n; ++i) {
    m[i] += omb1 * (g[i] - m[i]);
    v[i] += omb2 * (g[i] * g[i] - v[i]);
    p[i] -= alpha_t * m[i] / (sqrtf(v[i]) + eps);
  }
}

/* The ICC driver loop: examples/ycb_video/pose_refinement/
 * check_iterative_collision_check_link.py:44-79 (Adam alpha, translation alpha*0.1).
 * q,t updated in place; losses[n_iter]; traj [n_iter,N,7] = pose BEFORE each step. */
void mfo_icc_refine(const float *points, const float *sdf, const int64_t *offs, int N,
                    const float *pitch, const float *origin, const float *grid_target,
                    const float *grid_ne, float *q, float *t, int D, float voxel_threshold,
                    float sdf_offset, int n_iter, double alpha, float *losses, float *traj,
                    float *adam_hist /* [n_iter,2,N,7] m,v BEFORE each step, or NULL */) {
  float *gq = (float *)malloc(sizeof(float) * 4 * N), *gt = (float *)malloc(sizeof(float) * 3 * N);
  float *mq = (float *)calloc(4 * N, sizeof(float)), *vq = (float *)calloc(4 * N, sizeof(float));
  float *mt = (float *)calloc(3 * N, sizeof(float)), *vt = (float *)calloc(3 * N, sizeof(float));
  float sums[4];
  for (int it = 0; it < n_iter; ++it) {
    if (traj)
      for (int i = 0; i < N; ++i) {
        memcpy(traj + ((int64_t)it * N + i) * 7, q + 4 * i, 4 * sizeof(float));
        memcpy(traj + ((int64_t)it * N + i) * 7 + 4, t + 3 * i, 3 * sizeof(float));
      }
    float loss = mfo_icc_loss_grad(points, sdf, offs, N, pitch, origin, grid_target, grid_ne,
                                   q, t, D, voxel_threshold, sdf_offset, gq, gt, sums);
    if (losses) losses[it] = loss;
    if (adam_hist)
      for (int i = 0; i < N; ++i) {
        float *hm = adam_hist + (((int64_t)it * 2 + 0) * N + i) * 7;
        float *hv = adam_hist + (((int64_t)it * 2 + 1) * N + i) * 7;
        memcpy(hm, mq + 4 * i, 4 * sizeof(float)); memcpy(hm + 4, mt + 3 * i, 3 * sizeof(float));
        memcpy(hv, vq + 4 * i, 4 * sizeof(float)); memcpy(hv + 4, vt + 3 * i, 3 * sizeof(float));
      }
    mfo_adam_step(q, gq, mq, vq, 4 * N, alpha, it + 1);
    mfo_adam_step(t, gt, mt, vt, 3 * N, alpha * 0.1, it + 1);
  }
  free(gq); free(gt); free(mq); free(vq); free(mt); free(vt);
}

/* ------------------------------------------------------------------------
 * A10 IterativeClosestPointLink loss + gradient.
 * morefusion/contrib/iterative_closest_point_link.py:26-44.
 * ---------------------------------------------------------------------- */
float mfo_icp_loss_grad(const float *source, int64_t S, const float *target, int64_t T,
                        const float *q, const float *t, float *gq, float *gt) {
  float R[9];
  quat_to_R(q, R);
  float *src = (float *)malloc(sizeof(float) * 3 * S);
  float *gs = (float *)calloc(3 * S, sizeof(float));
  for (int64_t s = 0; s < S; ++s)
    for (int d = 0; d < 3; ++d)
      src[3 * s + d] = ((R[3 * d] * source[3 * s] + R[3 * d + 1] * source[3 * s + 1]) +
                        R[3 * d + 2] * source[3 * s + 2]) + t[d];
  double loss = 0;
  for (int64_t k = 0; k < T; ++k) {
    float best = INFINITY;
    int64_t bi = 0;
    for (int64_t s = 0; s < S; ++s) {
      float ssd = 0.0f;
      for (int d = 0; d < 3; ++d) {
        float tmp = src[3 * s + d] - target[3 * k + d];
        ssd += tmp * tmp;
      }
      if (ssd < best) { best = ssd; bi = s; }
    }
    if (best < 0.02f) {
      float l = 0.0f;
      for (int d = 0; d < 3; ++d) {
        float diff = src[3 * bi + d] - target[3 * k + d];
        l += diff * diff;
        gs[3 * bi + d] += 2.0f * diff;
      }
      loss += l;
    }
  }
  double gRd[9] = {0}, gtd[3] = {0};
  for (int64_t s = 0; s < S; ++s)
    for (int a = 0; a < 3; ++a) {
      gtd[a] += gs[3 * s + a];
      for (int b = 0; b < 3; ++b) gRd[3 * a + b] += (double)(gs[3 * s + a] * source[3 * s + b]);
    }
  float gR[9];
  for (int k = 0; k < 9; ++k) gR[k] = (float)gRd[k];
  for (int a = 0; a < 3; ++a) gt[a] = (float)gtd[a];
  quat_backward(q, gR, gq);
  free(src); free(gs);
  return (float)loss;
}
