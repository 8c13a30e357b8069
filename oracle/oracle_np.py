"""NumPy restatement of the reference algorithms on the hot path.

TEST INFRASTRUCTURE ONLY -- never imported by ``morefusion_amd`` (the product).
Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg may import this module, and only as the checker.

Every function cites the reference file:line it follows (paths relative to
``/root/reference/``).  Where the reference has a CPU twin (``forward_cpu``) the
restatement is pinned against golden vectors produced by running that twin
itself (``oracle/gen_golden.py`` -> ``tests/golden/ref_*.npz``,
``tests/test_oracle_golden.py``).  Where the reference is GPU-only CUDA text
(truncated_distance_function fwd/bwd, pseudo_occupancy_voxelization, the GPU forms of
interpolate_voxel_grid incl. its only backward, knn, the ICC / ICP links' forward) the
restatement is pinned against golden vectors produced by EXECUTING THAT TEXT: the kernel
strings compiled by g++ and run sequentially, the Python around them run on NumPy
(``oracle/cuda_text.py`` + ``oracle/gen_golden_cuda.py`` -> ``tests/golden/ref_cuda_*.npz``,
``tests/test_oracle_vs_reference_cuda_text.py``): distances, winner indices and
pseudo-occupancy grids bit-exact, losses to float32 rounding; the links' GRADIENTS against the
reference's own ``Function.backward`` methods run under a reverse-mode tape that restates only
chainer's elementary rules (``oracle/chainer_tape.py``).  Still **parity unpinned** (third party,
absent here): chainer's Adam, trimesh's quaternion_from_matrix, cv2 / imgviz resizing (DESIGN.md
section 3).

``mode`` selects between the reference's two semantic forks:
  "cpu": what ``forward_cpu`` does (NumPy promotion, round-half-even, floor)
  "gpu": what the CUDA text does (all-float32, round-half-away, trunc-to-zero)
The HIP kernels implement "gpu"; the two agree except on exact .5 ties /
negative coordinates, which dedicated tests cover.
"""
import math

import numpy as np

f32 = np.float32


# --------------------------------------------------------------------------
# helpers
# --------------------------------------------------------------------------
def round_half_away(x):
    """CUDA/HIP ``round()``: half away from zero (NumPy ``round`` is half-even)."""
    x = np.asarray(x)
    return np.copysign(np.floor(np.abs(x) + x.dtype.type(0.5)), x)


def _exact_half_fix(x, r):
    # floor(|x|+0.5) can be off for |x| = 0.49999997 (x+0.5 rounds up to 1.0).
    # CUDA round() returns 0 there.  Patch that single binade.
    ax = np.abs(x)
    bad = (ax < x.dtype.type(0.5)) & (np.abs(r) >= 1)
    r = np.where(bad, np.copysign(x.dtype.type(0), x), r)
    return r


def cuda_round(x):
    x = np.asarray(x)
    return _exact_half_fix(x, round_half_away(x))


def voxel_index(points, origin, pitch, mode="gpu"):
    """idx = round((p - origin) / pitch).

    cpu: functions/geometry/average_voxelization_3d.py:29 (NumPy promotion + .round())
    gpu: functions/geometry/average_voxelization_3d.py:80-86 (float32, CUDA round)
    """
    if mode == "cpu":
        return ((points - origin) / pitch).round().astype(int)
    p = np.asarray(points, dtype=f32)
    o = np.asarray(origin, dtype=f32)
    h = f32(pitch)
    return cuda_round((p - o) / h).astype(np.int64)


# --------------------------------------------------------------------------
# A1/A2 average_voxelization_3d
# --------------------------------------------------------------------------
def average_voxelization_3d(
    values, points, batch_indices, *, batch_size, origin, pitch, dimensions, mode="gpu"
):
    """functions/geometry/average_voxelization_3d.py:8-40 (cpu), :42-118 (gpu).

    Sum in increasing point index (the CPU loop order; the GPU's atomic order is
    unspecified), then divide by the per-voxel count.  Returns (matrix, counts).
    """
    if np.isnan(points).sum():
        raise ValueError("points include nan")
    B, C = batch_size, values.shape[1]
    X, Y, Z = dimensions
    idx = voxel_index(points, origin, pitch, mode)
    valid = ((idx >= 0) & (idx < np.array(dimensions))).all(axis=1)
    matrix = np.zeros((B, X, Y, Z, C), dtype=f32)
    counts = np.zeros((B, X, Y, Z), dtype=np.int32)
    n = np.flatnonzero(valid)
    key = (batch_indices[n].astype(np.int64), idx[n, 0], idx[n, 1], idx[n, 2])
    np.add.at(matrix, key, values[n])  # unbuffered, in index order, float32
    np.add.at(counts, key, 1)
    nz = counts > 0
    matrix[nz] = (matrix[nz] / counts[nz][:, None]).astype(f32)
    return np.ascontiguousarray(matrix.transpose(0, 4, 1, 2, 3)), counts


def average_voxelization_3d_backward(
    gmatrix, points, batch_indices, counts, *, origin, pitch, dimensions, mode="gpu"
):
    """functions/geometry/average_voxelization_3d.py:120-144 (cpu), :146-220 (gpu)."""
    P, C = points.shape[0], gmatrix.shape[1]
    idx = voxel_index(points, origin, pitch, mode)
    valid = ((idx >= 0) & (idx < np.array(dimensions))).all(axis=1)
    gvalues = np.zeros((P, C), dtype=f32)
    n = np.flatnonzero(valid)
    b = batch_indices[n].astype(np.int64)
    g = gmatrix[b, :, idx[n, 0], idx[n, 1], idx[n, 2]]
    c = counts[b, idx[n, 0], idx[n, 1], idx[n, 2]]
    gvalues[n] = (g / c[:, None]).astype(f32)
    return gvalues


# --------------------------------------------------------------------------
# A3 max_voxelization_3d
# --------------------------------------------------------------------------
def max_voxelization_3d(
    values, points, batch_indices, intensities, *, batch_size, origin, pitch,
    dimensions, mode="gpu",
):
    """functions/geometry/max_voxelization_3d.py:8-45.

    Winner per voxel = first point, replaced only by a strictly greater
    intensity (== arg-max, lowest index among ties).  The CUDA version
    (:75-134) is a racy approximation of the same rule; the HIP kernel is
    deterministic and implements exactly this.  Returns (matrix, indices).
    """
    if np.isnan(points).sum():
        raise ValueError("points include nan")
    B, C = batch_size, values.shape[1]
    X, Y, Z = dimensions
    idx = voxel_index(points, origin, pitch, mode)
    valid = ((idx >= 0) & (idx < np.array(dimensions))).all(axis=1)
    matrix = np.zeros((B, C, X, Y, Z), dtype=f32)
    indices = np.full((B, X, Y, Z), -1, dtype=np.int32)
    maxi = np.zeros((B, X, Y, Z), dtype=f32)
    for i in np.flatnonzero(valid):
        k = (int(batch_indices[i]), idx[i, 0], idx[i, 1], idx[i, 2])
        if indices[k] < 0 or intensities[i] > maxi[k]:
            indices[k] = i
            maxi[k] = intensities[i]
    ib, ix, iy, iz = np.nonzero(indices >= 0)
    matrix[ib, :, ix, iy, iz] = values[indices[ib, ix, iy, iz]]
    return matrix, indices


def max_voxelization_3d_backward(gmatrix, indices, n_points):
    """functions/geometry/max_voxelization_3d.py:47-56 (cpu), :140-185 (gpu)."""
    C = gmatrix.shape[1]
    gvalues = np.zeros((n_points, C), dtype=f32)
    ib, ix, iy, iz = np.nonzero(indices >= 0)
    np.add.at(gvalues, indices[ib, ix, iy, iz], gmatrix[ib, :, ix, iy, iz])
    return gvalues


# --------------------------------------------------------------------------
# A4 interpolate_voxel_grid
# --------------------------------------------------------------------------
_CORNERS = np.array(  # (dx, dy, dz) in the reference's weight order w000,w100,w010,w001,w110,w011,w101,w111
    [[0, 0, 0], [1, 0, 0], [0, 1, 0], [0, 0, 1], [1, 1, 0], [0, 1, 1], [1, 0, 1], [1, 1, 1]]
)


def _trilinear(points, mode):
    """functions/geometry/interpolate_voxel_grid.py:6-59 (gpu: static_cast<int>),
    :62-113 (cpu: np.floor)."""
    p = np.asarray(points, dtype=f32)
    low = np.floor(p) if mode == "cpu" else np.trunc(p)
    low_i = low.astype(np.int32)
    if mode == "cpu":
        # np.float32 scalar - np.int32 scalar promotes to float64 (:67-72): the CPU
        # twin forms the weights in double and rounds once on the store (:74-82)
        lo = p.astype(np.float64) - low_i
        hi = 1.0 - lo
    else:
        lo = (p - low_i.astype(f32)).astype(f32)
        hi = (f32(1.0) - lo).astype(f32)
    w = np.empty((p.shape[0], 8), dtype=f32)
    for j, (dx, dy, dz) in enumerate(_CORNERS):
        wx = lo[:, 0] if dx else hi[:, 0]
        wy = lo[:, 1] if dy else hi[:, 1]
        wz = lo[:, 2] if dz else hi[:, 2]
        w[:, j] = (wx * wy) * wz
    ixyz = low_i[:, None, :] + _CORNERS[None, :, :]
    return w, ixyz


def interpolate_voxel_grid(voxelized, points, batch_indices, mode="gpu"):
    """functions/geometry/interpolate_voxel_grid.py:132-154 (cpu), :159-214 (gpu).

    Accumulates the 8 corners in the reference's j order.  (For non-cubic
    grids the CUDA forward indexes with strides X*Y,Y (:203-204), a bug its own
    backward and CPU twin do not share; the restatement uses the correct Y*Z,Z.)
    """
    B, C, X, Y, Z = voxelized.shape
    w, ixyz = _trilinear(points, mode)
    P = points.shape[0]
    out = np.zeros((P, C), dtype=f32)
    b = batch_indices.astype(np.int64)
    for j in range(8):
        ix, iy, iz = ixyz[:, j, 0], ixyz[:, j, 1], ixyz[:, j, 2]
        ok = (ix >= 0) & (ix < X) & (iy >= 0) & (iy < Y) & (iz >= 0) & (iz < Z)
        n = np.flatnonzero(ok)
        out[n] += w[n, j, None] * voxelized[b[n], :, ix[n], iy[n], iz[n]]
    return out


def interpolate_voxel_grid_backward(gvalues, points, batch_indices, shape, mode="gpu"):
    """functions/geometry/interpolate_voxel_grid.py:216-268 (GPU only; CPU raises)."""
    B, C, X, Y, Z = shape
    w, ixyz = _trilinear(points, mode)
    g = np.zeros((B, X, Y, Z, C), dtype=f32)
    b = batch_indices.astype(np.int64)
    for j in range(8):
        ix, iy, iz = ixyz[:, j, 0], ixyz[:, j, 1], ixyz[:, j, 2]
        ok = (ix >= 0) & (ix < X) & (iy >= 0) & (iy < Y) & (iz >= 0) & (iz < Z)
        n = np.flatnonzero(ok)
        np.add.at(g, (b[n], ix[n], iy[n], iz[n]), w[n, j, None] * gvalues[n])
    return np.ascontiguousarray(g.transpose(0, 4, 1, 2, 3))


# --------------------------------------------------------------------------
# A5 occupancy_grid_3d
# --------------------------------------------------------------------------
def occupancy_grid_3d(points, *, pitch, origin, dims, threshold=1, return_aux=False):
    """functions/geometry/occupancy_grid_3d.py:31-54 (distances) + :77-85
    (sqrt -> min over points -> relu(thr - d) -> min(., 1)).  Chunked over
    voxels so the [X,Y,Z,P] temporaries stay small; arithmetic identical."""
    dtype = points.dtype
    X, Y, Z = (int(d) for d in dims)
    o = np.asarray(origin, dtype=dtype)
    h = np.asarray(pitch, dtype=dtype)
    pf = (points - o) / h
    I, J, K = np.meshgrid(np.arange(X), np.arange(Y), np.arange(Z), indexing="ij")
    vox = np.stack([I, J, K], -1).reshape(-1, 3).astype(dtype)
    dmin = np.empty(vox.shape[0], dtype=dtype)
    for s in range(0, vox.shape[0], 4096):
        d = vox[s : s + 4096, None, :] - pf[None, :, :]
        dd = np.sqrt((d[..., 0] ** 2 + d[..., 1] ** 2) + d[..., 2] ** 2)
        dmin[s : s + 4096] = dd.min(axis=1)
    dmin = dmin.reshape(X, Y, Z)
    m = np.maximum(dtype.type(threshold) - dmin, 0)
    m = np.minimum(m, dtype.type(1))
    if return_aux:
        return m, dmin
    return m


def occupancy_grid_3d_backward(gm, points, *, pitch, origin, dims, threshold=1):
    """Backward of the composite (occupancy_grid_3d.py:56-74 + chainer's
    minimum / relu / min / sqrt / pow backward rules).

    chainer ``F.min`` routes the gradient to EVERY element equal to the
    minimum; ``F.minimum(a,b)`` to ``a`` where ``a <= b``; ``relu`` where > 0."""
    dtype = points.dtype
    X, Y, Z = (int(d) for d in dims)
    o = np.asarray(origin, dtype=dtype)
    h = np.asarray(pitch, dtype=dtype)
    pf = (points - o) / h
    I, J, K = np.meshgrid(np.arange(X), np.arange(Y), np.arange(Z), indexing="ij")
    vox = np.stack([I, J, K], -1).reshape(-1, 3).astype(dtype)
    gm = gm.reshape(-1)
    gpf = np.zeros_like(pf)
    thr = dtype.type(threshold)
    for s in range(0, vox.shape[0], 4096):
        d = vox[s : s + 4096, None, :] - pf[None, :, :]  # [v, P, 3]
        dd = np.sqrt((d[..., 0] ** 2 + d[..., 1] ** 2) + d[..., 2] ** 2)
        dmin = dd.min(axis=1)
        r = thr - dmin
        g_d = np.where((r > 0) & (np.maximum(r, 0) <= 1), -gm[s : s + 4096], 0)
        sel = dd == dmin[:, None]
        with np.errstate(divide="ignore", invalid="ignore"):
            g_dd = np.where(sel, g_d[:, None], 0) / (2 * dd)  # sqrt backward gy/(2y)
            g_comp = 2 * d * g_dd[..., None]  # x**2 backward
        g_comp = np.where(sel[..., None], g_comp, 0)
        gpf += (-g_comp).sum(axis=0)  # d = vox - pf
    return (gpf / h).astype(dtype)


# --------------------------------------------------------------------------
# A6/A7 truncated_distance_function, pseudo_occupancy_voxelization (GPU-only)
# --------------------------------------------------------------------------
def tdf_kernel_offsets(pitch, truncation):
    """functions/geometry/truncated_distance_function.py:36-41.
    ksize = ceil(trunc/pitch) made odd; offsets from meshgrid('xy'):
    flat k=(a*ks+b)*ks+c -> (b, a, c) - ks//2."""
    ksize = int(np.ceil(f32(truncation) / f32(pitch)))
    if ksize % 2 == 0:
        ksize += 1
    kern = np.meshgrid(*(np.arange(ksize),) * 3)
    kern = np.stack(kern, -1).reshape(-1, 3).astype(f32)
    kern -= ksize // 2
    return ksize, kern


def truncated_distance_function(points, *, pitch, origin, dims, truncation, dtype=f32):
    """functions/geometry/truncated_distance_function.py:21-103 (CUDA text :51-79).

    Returns (tdf [X,Y,Z], flat_index [X,Y,Z] int64 = p*K+k of the winner or -1,
    ksize).  The reference's (atomicMin, then atomicExch if smaller) pair is racy
    for the index; the restatement -- like the HIP kernel -- records the exact
    arg-min with the lowest flat index among equal distances."""
    X, Y, Z = dims
    p = np.asarray(points, dtype=dtype)
    h = dtype(pitch)
    o = np.asarray(origin, dtype=dtype)
    trunc = dtype(truncation)
    ksize, kern = tdf_kernel_offsets(pitch, truncation)
    kern = kern.astype(dtype)
    K = ksize ** 3
    pf = (p - o) / h  # [P,3]
    r = cuda_round(pf)
    # int ix = round(ix_f) + kernel[...]  (float add, then conversion to int)
    v = (r[:, None, :] + kern[None, :, :]).astype(np.int64)  # [P,K,3]
    inb = ((v >= 0) & (v < np.array([X, Y, Z]))).all(axis=2)
    d = pf[:, None, :] - v.astype(dtype)
    dist = h * np.sqrt((d[..., 0] * d[..., 0] + d[..., 1] * d[..., 1]) + d[..., 2] * d[..., 2])
    cand = inb & (dist < trunc)
    pi, ki = np.nonzero(cand)
    flat = pi.astype(np.int64) * K + ki
    vox = (v[pi, ki, 0] * Y + v[pi, ki, 1]) * Z + v[pi, ki, 2]
    dd = dist[pi, ki]
    order = np.lexsort((flat, dd, vox))  # by voxel, then distance, then flat index
    vox_s = vox[order]
    first = np.ones(len(order), dtype=bool)
    first[1:] = vox_s[1:] != vox_s[:-1]
    win = order[first]
    tdf = np.full(X * Y * Z, trunc, dtype=dtype)
    idx = np.full(X * Y * Z, -1, dtype=np.int64)
    tdf[vox[win]] = dd[win]
    idx[vox[win]] = flat[win]
    return tdf.reshape(X, Y, Z), idx.reshape(X, Y, Z), ksize


def truncated_distance_function_backward(
    gmatrix, points, flat_index, ksize, *, pitch, origin, dtype=f32
):
    """functions/geometry/truncated_distance_function.py:105-166 (CUDA :121-146):
    unit vector from the winning voxel to the point, scaled by the voxel's
    gradient, accumulated per point."""
    p = np.asarray(points, dtype=dtype)
    h = dtype(pitch)
    o = np.asarray(origin, dtype=dtype)
    X, Y, Z = flat_index.shape
    K = ksize ** 3
    gp = np.zeros_like(p)
    vx, vy, vz = np.nonzero(flat_index >= 0)
    fi = flat_index[vx, vy, vz]
    pi = fi // K
    pf = (p[pi] - o) / h
    v = np.stack([vx, vy, vz], 1).astype(dtype)  # == round(pf) + kernel[k]
    d = pf - v
    n = np.sqrt((d[:, 0] * d[:, 0] + d[:, 1] * d[:, 1]) + d[:, 2] * d[:, 2])
    ok = n > 0
    g = gmatrix[vx, vy, vz]
    contrib = np.zeros_like(d)
    contrib[ok] = d[ok] / n[ok, None] * g[ok, None]
    np.add.at(gp, pi, contrib)
    return gp


def pseudo_occupancy_weights(tdf_index_point, sdf, sdf_offset, dtype=f32):
    """functions/geometry/truncated_distance_function.py:198-207.
    ``tdf_index_point``: winning POINT id per voxel (-1 none).
    Returns (weight_surface, weight_inside, max)."""
    w = np.full(tdf_index_point.shape, -1, dtype=dtype)
    mask = tdf_index_point != -1
    w[mask] = sdf[tdf_index_point[mask]]
    w = w + dtype(sdf_offset)
    neg = w < 0
    w[neg] = 0
    M = w.max()
    with np.errstate(invalid="ignore", divide="ignore"):
        w_in = w / M
    w_surf = w_in.copy()
    w_surf[~neg] = 1 - w_surf[~neg]
    return w_surf, w_in, M


def pseudo_occupancy_voxelization(
    points, sdf, *, pitch, origin, dims, threshold=1, sdf_offset=0, dtype=f32,
    return_aux=False,
):
    """functions/geometry/truncated_distance_function.py:181-213."""
    trunc = dtype(threshold) * dtype(pitch)
    tdf, flat, ksize = truncated_distance_function(
        points, pitch=pitch, origin=origin, dims=dims, truncation=trunc, dtype=dtype
    )
    K = ksize ** 3
    pidx = np.where(flat >= 0, flat // K, -1)
    grid = 1 - tdf / trunc
    w_surf, w_in, M = pseudo_occupancy_weights(pidx, np.asarray(sdf, dtype=dtype), sdf_offset, dtype)
    out = (grid, grid * w_surf, grid * w_in)
    if return_aux:
        return out, dict(tdf=tdf, flat=flat, ksize=ksize, pidx=pidx, w_surf=w_surf,
                         w_in=w_in, M=M, trunc=trunc)
    return out


# --------------------------------------------------------------------------
# A8 rigid transforms
# --------------------------------------------------------------------------
def quaternion_matrix(q):
    """functions/geometry/quaternion_matrix.py:65-78 + :14-34.  q = wxyz, any norm."""
    q = np.asarray(q)
    squeeze = q.ndim == 1
    if squeeze:
        q = q[None]
    dt = q.dtype.type
    norm = (q ** 2).sum(axis=1, keepdims=True)
    qs = q * np.sqrt(dt(2.0) / norm)
    Q = qs[:, :, None] * qs[:, None, :]
    R = np.eye(4, dtype=q.dtype)[None].repeat(q.shape[0], axis=0)
    R[:, 0, 0] = 1 - Q[:, 2, 2] - Q[:, 3, 3]
    R[:, 0, 1] = Q[:, 1, 2] - Q[:, 3, 0]
    R[:, 0, 2] = Q[:, 1, 3] + Q[:, 2, 0]
    R[:, 1, 0] = Q[:, 1, 2] + Q[:, 3, 0]
    R[:, 1, 1] = 1 - Q[:, 1, 1] - Q[:, 3, 3]
    R[:, 1, 2] = Q[:, 2, 3] - Q[:, 1, 0]
    R[:, 2, 0] = Q[:, 1, 3] - Q[:, 2, 0]
    R[:, 2, 1] = Q[:, 2, 3] + Q[:, 1, 0]
    R[:, 2, 2] = 1 - Q[:, 1, 1] - Q[:, 2, 2]
    return R[0] if squeeze else R


def quaternion_matrix_backward(q, gR):
    """Chain rule through quaternion_matrix.py:36-51 (hand-written dR/dQ), the
    outer product (:54-62) and the sqrt(2/|q|^2) scaling (:71-72).  q [N,4], gR [N,4,4]."""
    dt = q.dtype.type
    gQ = np.zeros((q.shape[0], 4, 4), dtype=q.dtype)
    gQ[:, 1, 0] = -gR[:, 1, 2] + gR[:, 2, 1]
    gQ[:, 1, 1] = -gR[:, 1, 1] - gR[:, 2, 2]
    gQ[:, 1, 2] = gR[:, 0, 1] + gR[:, 1, 0]
    gQ[:, 1, 3] = gR[:, 0, 2] + gR[:, 2, 0]
    gQ[:, 2, 0] = gR[:, 0, 2] - gR[:, 2, 0]
    gQ[:, 2, 2] = -gR[:, 0, 0] - gR[:, 2, 2]
    gQ[:, 2, 3] = gR[:, 1, 2] + gR[:, 2, 1]
    gQ[:, 3, 0] = -gR[:, 0, 1] + gR[:, 1, 0]
    gQ[:, 3, 3] = -gR[:, 0, 0] - gR[:, 1, 1]
    n = (q ** 2).sum(axis=1, keepdims=True)
    s = np.sqrt(dt(2.0) / n)
    qs = q * s
    gqs = np.einsum("nij,nj->ni", gQ, qs) + np.einsum("nji,nj->ni", gQ, qs)
    dot = (gqs * q).sum(axis=1, keepdims=True)
    return s * gqs - (s / n) * dot * q


def compose_transform(R, t):
    """functions/geometry/compose_transform.py:18-28."""
    squeeze = R.ndim == 2
    if squeeze:
        R, t = R[None], t[None]
    T = np.eye(4, dtype=R.dtype)[None].repeat(R.shape[0], axis=0)
    T[:, :3, :3] = R
    T[:, :3, 3] = t
    return T[0] if squeeze else T


def translation_matrix(t):
    """functions/geometry/translation_matrix.py:15-22."""
    squeeze = t.ndim == 1
    if squeeze:
        t = t[None]
    T = np.eye(4, dtype=t.dtype)[None].repeat(t.shape[0], axis=0)
    T[:, :3, 3] = t
    return T[0] if squeeze else T


def transformation_matrix(q, t):
    """functions/geometry/transformation_matrix.py:5-18."""
    if q.ndim == 2:
        return compose_transform(quaternion_matrix(q)[:, :3, :3], t)
    return compose_transform(quaternion_matrix(q[None])[:, :3, :3], t[None])[0]


def transform_points(points, transform):
    """functions/geometry/transform_points.py:6-30.  The reference goes through a
    BLAS/cuBLAS matmul whose accumulation order is unspecified; the restatement
    fixes ((R0*x + R1*y) + R2*z) + t, un-fused -- exactly what the HIP kernels do."""
    squeeze = transform.ndim == 2
    if squeeze:
        transform = transform[None]
    R = transform[:, :3, :3]
    t = transform[:, :3, 3]
    x, y, z = points[:, 0], points[:, 1], points[:, 2]
    out = (
        (R[:, :, 0, None] * x[None, None] + R[:, :, 1, None] * y[None, None])
        + R[:, :, 2, None] * z[None, None]
    ) + t[:, :, None]
    out = out.transpose(0, 2, 1)
    return out[0] if squeeze else out


def quaternion_from_matrix(matrix):
    """trimesh.transformations.quaternion_from_matrix(isprecise=False) as called at
    contrib/iterative_collision_check_link.py:22 (third party; restated from the
    published Gohlke algorithm: largest eigenvector of the symmetric K matrix)."""
    M = np.asarray(matrix, dtype=np.float64)[:4, :4]
    m00, m01, m02 = M[0, 0], M[0, 1], M[0, 2]
    m10, m11, m12 = M[1, 0], M[1, 1], M[1, 2]
    m20, m21, m22 = M[2, 0], M[2, 1], M[2, 2]
    K = np.array(
        [
            [m00 - m11 - m22, 0.0, 0.0, 0.0],
            [m01 + m10, m11 - m00 - m22, 0.0, 0.0],
            [m02 + m20, m12 + m21, m22 - m00 - m11, 0.0],
            [m21 - m12, m02 - m20, m10 - m01, m00 + m11 + m22],
        ]
    )
    K /= 3.0
    w, V = np.linalg.eigh(K)
    q = V[[3, 0, 1, 2], np.argmax(w)]
    if q[0] < 0.0:
        np.negative(q, q)
    return q


# --------------------------------------------------------------------------
# A11 nn, A10 ICP link, A12 average_distance
# --------------------------------------------------------------------------
def nn(ref, query, chunk=4096):
    """geometry/knn/nn.py:18-49 + knn/cuComputeDistanceGlobal.cu:20-86:
    all-pairs ssd accumulated x,y,z in float32, argmin over ref (first minimum)."""
    ref = np.asarray(ref, dtype=f32)
    query = np.asarray(query, dtype=f32)
    out = np.empty(query.shape[0], dtype=np.int64)
    for s in range(0, query.shape[0], chunk):
        q = query[s : s + chunk]
        ssd = np.zeros((ref.shape[0], q.shape[0]), dtype=f32)
        for d in range(ref.shape[1]):
            tmp = ref[:, d, None] - q[None, :, d]
            ssd += tmp * tmp
        out[s : s + chunk] = ssd.argmin(axis=0)
    return out


def icp_loss(source, target, q, t, grad=True):
    """contrib/iterative_closest_point_link.py:26-44.  Returns loss, (gq, gt)."""
    dt = source.dtype
    T = transformation_matrix(q, t)
    src = transform_points(source, T)
    dists = np.zeros((target.shape[0], src.shape[0]), dtype=dt)
    for d in range(3):
        tmp = src[None, :, d] - target[:, None, d]
        dists += tmp * tmp
    corr = dists.argmin(axis=1)
    dmin = dists[np.arange(dists.shape[0]), corr]
    keep = dmin < 0.02
    diff = src[corr[keep]] - target[keep]
    loss = ((diff ** 2).sum(axis=1)).sum(axis=0)
    if not grad:
        return loss
    gsrc = np.zeros_like(src)
    np.add.at(gsrc, corr[keep], 2 * diff)
    gR = np.zeros((1, 4, 4), dtype=dt)
    gR[0, :3, :3] = gsrc.T @ source
    gt = gsrc.sum(axis=0)
    gq = quaternion_matrix_backward(q[None], gR)[0]
    return loss, (gq, gt)


def average_distance(points, transform_true, transforms_pred, symmetric=False):
    """functions/loss/average_distance.py:40-85."""
    pt = transform_points(points, transform_true)
    pp = transform_points(points, transforms_pred)
    n_pred, n_points = pp.shape[:2]
    if symmetric:
        idx = nn(pt, pp.reshape(-1, 3))
        ptr = pt[idx].reshape(n_pred, n_points, 3)
    else:
        ptr = np.repeat(pt[None], n_pred, axis=0)
    return np.sqrt(((ptr - pp) ** 2).sum(axis=2)).mean(axis=1)


# --------------------------------------------------------------------------
# A15 metrics
# --------------------------------------------------------------------------
def metrics_average_distance(points, transform1, transform2):
    """metrics/average_distance.py:6-19 (one instance): returns (add, add_s) float64.
    KDTree replaced by exact brute force (same arg-min up to exact ties)."""
    def tp(p, T):
        return p @ T[:3, :3].T + T[:3, 3]
    p1 = tp(np.asarray(points, dtype=np.float64), np.asarray(transform1, dtype=np.float64))
    p2 = tp(np.asarray(points, dtype=np.float64), np.asarray(transform2, dtype=np.float64))
    add = np.linalg.norm(p1 - p2, axis=1).mean()
    d2 = ((p1[:, None, :] - p2[None, :, :]) ** 2).sum(axis=2)
    idx = d2.argmin(axis=1)
    add_s = np.linalg.norm(p1 - p2[idx], axis=1).mean()
    return add, add_s


def voc_ap(rec, prec, max_value=0.1):
    """metrics/ycb_video_add_auc.py:36-51."""
    mrec = np.r_[0, rec, max_value]
    mpre = np.r_[0, prec, prec[-1]]
    for i in range(1, len(mpre)):
        mpre[i] = max(mpre[i], mpre[i - 1])
    i = np.argwhere(mrec[1:] != mrec[:-1]) + 1
    return np.sum((mrec[i] - mrec[i - 1]) * mpre[i]) / max_value


def ycb_video_add_auc(adds, max_value=0.1):
    """metrics/ycb_video_add_auc.py:5-32."""
    adds = np.asarray(adds)
    D = adds.astype(float).copy()
    D[D > max_value] = np.inf
    d = np.sort(D)
    n = len(d)
    acc = np.cumsum(np.ones((1, n))) / n
    keep = np.isfinite(d)
    if keep.any():
        return voc_ap(d[keep], acc[keep], max_value=max_value)
    return 0


def median(x, axis=None):
    """extra/_cupy.py:47-62: mean of the two middle values for even n."""
    if axis is None:
        x = x.flatten()
        axis = 0
    n = x.shape[axis]
    s = np.sort(x, axis)
    m_odd = np.take(s, n // 2, axis)
    if n % 2 == 1:
        return m_odd
    return (m_odd + np.take(s, n // 2 - 1, axis)) / 2


# --------------------------------------------------------------------------
# A9 IterativeCollisionCheckLink forward + analytic backward
# --------------------------------------------------------------------------
def icc_loss(points, sdf, pitch, origin, grid_target, grid_nontarget_empty, q, t,
             voxel_dim=32, voxel_threshold=2, sdf_offset=0.0, dtype=f32, grad=True):
    """contrib/iterative_collision_check_link.py:31-99.

    points/sdf: lists (per object, model frame); q [N,4] wxyz, t [N,3].
    Returns loss, and if grad: (gq [N,4], gt [N,3], aux dict).
    Backward = chainer's rules for every node of the graph (maximum: gradient
    to the first argument when equal; TDF backward :105-166)."""
    N = len(points)
    D = voxel_dim
    dims = (D, D, D)
    q = np.asarray(q, dtype=dtype)
    t = np.asarray(t, dtype=dtype)
    T = transformation_matrix(q, t)
    pw = [transform_points(np.asarray(points[i], dtype=dtype), T[i]) for i in range(N)]
    sdf = [np.asarray(s, dtype=dtype) for s in sdf]
    tgt = np.asarray(grid_target, dtype=dtype)
    ne = np.asarray(grid_nontarget_empty, dtype=dtype).copy()

    own, oth = [], []
    surf = np.zeros((N, D, D, D), dtype=dtype)
    ins = np.zeros((N, D, D, D), dtype=dtype)
    ne_eff = ne.copy()
    oth_used = np.zeros(N, dtype=bool)
    for i in range(N):
        (_, s_i, in_i), aux = pseudo_occupancy_voxelization(
            pw[i], sdf[i], pitch=pitch[i], origin=origin[i], dims=dims,
            threshold=voxel_threshold, sdf_offset=sdf_offset, dtype=dtype, return_aux=True)
        own.append(aux)
        surf[i], ins[i] = s_i, in_i
        if N <= 1:
            oth.append(None)
            continue
        po = np.concatenate([pw[j] for j in range(N) if j != i], axis=0)
        so = np.concatenate([sdf[j] for j in range(N) if j != i], axis=0)
        (_, _, g_o), aux_o = pseudo_occupancy_voxelization(
            po, so, pitch=pitch[i], origin=origin[i], dims=dims,
            threshold=voxel_threshold, dtype=dtype, return_aux=True)
        aux_o["grid_inside"] = g_o
        oth.append(aux_o)
        if not np.isnan(g_o).any():
            oth_used[i] = True
            ne_eff[i] = np.maximum(ne[i], g_o)

    S_t = tgt.sum(dtype=dtype)
    RN = (surf * tgt).sum(dtype=dtype)
    S_in = ins.sum(dtype=dtype)
    PN = (ins * ne_eff).sum(dtype=dtype)
    with np.errstate(invalid="ignore", divide="ignore"):
        reward = RN / S_t
        penalty = PN / S_in
    loss = penalty - reward
    if not grad:
        return loss

    g_surf = -tgt / S_t
    g_ins = ne_eff / S_in - PN / (S_in * S_in)
    gpw = [np.zeros_like(p) for p in pw]
    for i in range(N):
        a = own[i]
        g_grid = a["w_surf"] * g_surf[i] + a["w_in"] * g_ins[i]
        g_tdf = -g_grid / a["trunc"]
        gpw[i] += truncated_distance_function_backward(
            g_tdf, pw[i], a["flat"], a["ksize"], pitch=pitch[i], origin=origin[i], dtype=dtype)
        if oth[i] is not None and oth_used[i]:
            b = oth[i]
            g_oth = np.where(ne[i] >= b["grid_inside"], 0, ins[i] / S_in).astype(dtype)
            g_grid_o = b["w_in"] * g_oth
            g_tdf_o = -g_grid_o / b["trunc"]
            po = np.concatenate([pw[j] for j in range(N) if j != i], axis=0)
            gpo = truncated_distance_function_backward(
                g_tdf_o, po, b["flat"], b["ksize"], pitch=pitch[i], origin=origin[i], dtype=dtype)
            s = 0
            for j in range(N):
                if j == i:
                    continue
                n_j = pw[j].shape[0]
                gpw[j] += gpo[s : s + n_j]
                s += n_j
    gT = np.zeros((N, 4, 4), dtype=dtype)
    for i in range(N):
        gT[i, :3, :3] = gpw[i].T @ np.asarray(points[i], dtype=dtype)
        gT[i, :3, 3] = gpw[i].sum(axis=0)
    gt = gT[:, :3, 3].copy()
    gR = gT.copy()
    gR[:, :3, 3] = 0
    gq = quaternion_matrix_backward(q, gR)
    aux = dict(S_t=S_t, RN=RN, S_in=S_in, PN=PN, own=own, oth=oth, surf=surf, ins=ins,
               ne_eff=ne_eff, gpw=gpw)
    return loss, (gq, gt, aux)


# --------------------------------------------------------------------------
# Chainer's Adam (third party; chainer/optimizers/adam.py, v7) -- parity unpinned
# --------------------------------------------------------------------------
class ChainerAdam:
    """m += (1-b1)(g-m); v += (1-b2)(g*g-v); p -= alpha_t * m / (sqrt(v)+eps),
    alpha_t = alpha*sqrt(1-b2^t)/(1-b1^t) evaluated in double then cast.
    Call sites: examples/ycb_video/pose_refinement/check_iterative_collision_check_link.py:48-50."""

    def __init__(self, params, alphas, beta1=0.9, beta2=0.999, eps=1e-8):
        self.params = params
        self.alphas = alphas
        self.b1, self.b2, self.eps = beta1, beta2, eps
        self.m = [np.zeros_like(p) for p in params]
        self.v = [np.zeros_like(p) for p in params]
        self.t = 0

    def update(self, grads):
        self.t += 1
        fix1 = 1.0 - math.pow(self.b1, self.t)
        fix2 = 1.0 - math.pow(self.b2, self.t)
        for p, g, m, v, alpha in zip(self.params, grads, self.m, self.v, self.alphas):
            dt = p.dtype.type
            alpha_t = dt(alpha * math.sqrt(fix2) / fix1)
            m += dt(1 - self.b1) * (g - m)
            v += dt(1 - self.b2) * (g * g - v)
            p -= alpha_t * m / (np.sqrt(v) + dt(self.eps))


def icc_refine(points, sdf, pitch, origin, grid_target, grid_nontarget_empty, transform_init,
               n_iter=100, sdf_offset=0.02, alpha=0.01, dtype=f32, **kw):
    """The ICC driver loop: check_iterative_collision_check_link.py:44-79."""
    q = np.stack([quaternion_from_matrix(T) for T in transform_init]).astype(dtype)
    t = np.stack([np.asarray(T)[:3, 3] for T in transform_init]).astype(dtype)
    opt = ChainerAdam([q, t], [alpha, alpha * 0.1])
    losses, traj = [], []
    for _ in range(n_iter):
        traj.append(np.concatenate([q, t], axis=1).copy())
        loss, (gq, gt, _) = icc_loss(points, sdf, pitch, origin, grid_target,
                                     grid_nontarget_empty, q, t, sdf_offset=sdf_offset,
                                     dtype=dtype, **kw)
        losses.append(loss)
        opt.update([gq, gt])
    return q, t, np.array(losses), np.stack(traj)


# --------------------------------------------------------------------------
# pre-processing in front of the network
# --------------------------------------------------------------------------
def pointcloud_from_depth(depth, fx, fy, cx, cy):
    """geometry/pointcloud_from_depth.py:4-21 (depth_type "z"); float64 result for a
    float32 depth because ``c - cx`` is float64.  Pinned by tests/golden/ref_preprocess.npz."""
    rows, cols = depth.shape
    c, r = np.meshgrid(np.arange(cols), np.arange(rows), sparse=True)
    valid = ~np.isnan(depth)
    z = np.where(valid, depth, np.nan)
    x = np.where(valid, z * (c - cx) / fx, np.nan)
    y = np.where(valid, z * (r - cy) / fy, np.nan)
    return np.dstack((x, y, z))


def mask_to_bbox(mask):
    """geometry/masks_to_bboxes.py:28-34 for one mask: (y1, x1, y2, x2), end-exclusive."""
    where = np.argwhere(mask)
    if len(where) == 0:
        return np.zeros(4, dtype=np.int64)
    (y1, x1), (y2, x2) = where.min(0), where.max(0) + 1
    return np.array([y1, x1, y2, x2])


def cv2_resize_nearest(src, height, width):
    """OpenCV ``cv::resize(..., INTER_NEAREST)`` (opencv-python, unpinned in
    requirements.txt:13; absent here -> restated from imgproc/resize.cpp ``resizeNN``:
    ``sx = min(floor(x * (1 / (dst_w / src_w))), src_w - 1)``).  **parity unpinned**."""
    sh, sw = src.shape[:2]
    ify, ifx = 1.0 / (float(height) / sh), 1.0 / (float(width) / sw)
    sy = np.minimum(np.floor(np.arange(height) * ify).astype(np.int64), sh - 1)
    sx = np.minimum(np.floor(np.arange(width) * ifx).astype(np.int64), sw - 1)
    return src[sy][:, sx]


def _cv2_linear_taps(n_dst, scale, n_src, zero_frac_at_border):
    d = np.arange(n_dst, dtype=np.float64)
    f = ((d + 0.5) * scale - 0.5).astype(np.float32)
    s = np.floor(f).astype(np.int64)
    f = (f - s.astype(np.float32)).astype(np.float32)
    if zero_frac_at_border:
        lo, hi = s < 0, s >= n_src - 1
        f = np.where(lo | hi, f32(0), f)
        s = np.where(lo, 0, np.where(hi, n_src - 1, s))
    w0 = np.rint((f32(1) - f) * f32(2048)).astype(np.int64)
    w1 = np.rint(f * f32(2048)).astype(np.int64)
    return np.clip(s, 0, n_src - 1), np.clip(s + 1, 0, n_src - 1), w0, w1


def cv2_resize_linear_u8(src, height, width):
    """OpenCV ``cv::resize(..., INTER_LINEAR)`` for 8-bit images, native (non-IPP) path of
    imgproc/resize.cpp: 11-bit fixed-point taps (``resizeGeneric_`` set-up, HResizeLinear,
    VResizeLinear ``((b0*(S0>>4))>>16) + ((b1*(S1>>4))>>16) + 2 >> 2``) and the 2x2 box
    average for an exact 2:1 reduction.  **parity unpinned** (cv2 absent; IPP builds of
    opencv-python may differ by one grey level)."""
    sh, sw = src.shape[:2]
    scale_x, scale_y = 1.0 / (float(width) / sw), 1.0 / (float(height) / sh)
    img = src.astype(np.int64)
    eps = np.finfo(np.float64).eps
    if abs(scale_x - 2.0) < eps and abs(scale_y - 2.0) < eps:
        out = (img[0::2, 0::2] + img[0::2, 1::2] + img[1::2, 0::2] + img[1::2, 1::2] + 2) >> 2
        return out[:height, :width].astype(np.uint8)
    xa, xb, a0, a1 = _cv2_linear_taps(width, scale_x, sw, True)
    ya, yb, b0, b1 = _cv2_linear_taps(height, scale_y, sh, False)
    sel = (slice(None), slice(None)) + (None,) * (img.ndim - 2)
    rows = img[:, xa] * a0[None, :][sel] + img[:, xb] * a1[None, :][sel]   # [sh, width, ...]
    r0, r1 = rows[ya], rows[yb]
    b0e, b1e = b0[:, None][sel], b1[:, None][sel]
    out = (((b0e * (r0 >> 4)) >> 16) + ((b1e * (r1 >> 4)) >> 16) + 2) >> 2
    return np.clip(out, 0, 255).astype(np.uint8)


def centerize(src, shape, cval=None, interpolation="linear"):
    """imgviz.centerize (imgviz>=0.10, requirements.txt:8; absent here): aspect-preserving
    resize to fit ``shape`` then centred padding with ``cval``.  **parity unpinned**."""
    if src.shape[:2] == tuple(shape[:2]):
        return src
    dst = np.zeros(tuple(shape[:2]) + src.shape[2:], dtype=src.dtype)
    if cval is not None:
        dst[:, :] = cval
    sh, sw = src.shape[:2]
    scale = min(1.0 * shape[0] / sh, 1.0 * shape[1] / sw)
    dh, dw = int(round(sh * scale)), int(round(sw * scale))
    if interpolation == "nearest":
        res = cv2_resize_nearest(src, dh, dw)
    else:
        res = cv2_resize_linear_u8(src, dh, dw)
    ph = (shape[0] - dh) // 2 if dh < shape[0] else 0
    pw = (shape[1] - dw) // 2 if dw < shape[1] else 0
    dst[ph:ph + dh, pw:pw + dw] = res
    return dst


def instance_crops(rgb, depth, K, label, instance_ids, image_size=256, min_valid=50):
    """ros/src/morefusion_ros/nodes/singleview_3d_pose_estimation.py:116-126,158-176 (same
    steps as datasets/rgbd_pose_estimation/base.py:112-137): per instance masked rgb / point
    cloud crops centerized to ``image_size``.  Instances the reference skips (< ``min_valid``
    valid points) come back as pure padding with ``keep`` False."""
    pcd = pointcloud_from_depth(depth, K[0, 0], K[1, 1], K[0, 2], K[1, 2])
    nanmask = np.isnan(pcd).any(axis=2)
    S = image_size
    n = len(instance_ids)
    rgb_out = np.zeros((n, S, S, 3), np.uint8)
    pcd_out = np.full((n, S, S, 3), np.nan, np.float64)
    keep = np.zeros(n, bool)
    bboxes = np.zeros((n, 4), np.int64)
    for i, ins_id in enumerate(instance_ids):
        mask = label == ins_id
        bboxes[i] = mask_to_bbox(mask)
        if (~nanmask & mask).sum() < min_valid:
            continue
        y1, x1, y2, x2 = bboxes[i]
        rgb_ins = rgb[y1:y2, x1:x2].copy()
        rgb_ins[~mask[y1:y2, x1:x2]] = 0
        rgb_out[i] = centerize(rgb_ins, (S, S), cval=0)
        pcd_ins = pcd[y1:y2, x1:x2].copy()
        pcd_ins[~mask[y1:y2, x1:x2]] = np.nan
        pcd_out[i] = centerize(pcd_ins, (S, S), cval=np.nan, interpolation="nearest")
        keep[i] = True
    return rgb_out, pcd_out, keep, bboxes


def valid_pixel_order(pcd):
    """Row-major list of the pixels without a NaN coordinate, per image, and their counts:
    ``iy, ix = xp.where(mask[i])`` / ``int(mask[i].sum())`` with ``mask = ~isnan(pcd).any(axis)``
    (contrib/singleview_3d/models/model.py:195-196,206) as flat indices iy * W + ix.
    pcd [B,H,W,3] or [B,HW,3] -> (order [B,HW] int32, valid entries first; counts [B] int64)."""
    pcd = np.asarray(pcd)
    valid = ~np.isnan(pcd).any(axis=-1).reshape(pcd.shape[0], -1)
    order = np.zeros(valid.shape, np.int32)
    for i, v in enumerate(valid):
        idx = np.flatnonzero(v)
        order[i, :len(idx)] = idx
    return order, valid.sum(axis=1)
