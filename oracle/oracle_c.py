"""ctypes binding of oracle/mf_oracle.c (TEST INFRASTRUCTURE ONLY -- see the
header of that file).  ``build()`` compiles it with gcc if the .so is missing."""
import ctypes
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
SO = os.path.join(HERE, "_build", "libmforacle.so")
_lib = None

c_f = ctypes.POINTER(ctypes.c_float)
c_i32 = ctypes.POINTER(ctypes.c_int32)
c_i64 = ctypes.POINTER(ctypes.c_int64)


def build(force=False):
    src = os.path.join(HERE, "mf_oracle.c")
    if force or not os.path.exists(SO) or os.path.getmtime(SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", HERE, "-s", "-B" if force else "-s"])
    return SO


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = ctypes.CDLL(SO)
        _lib.mfo_icc_loss_grad.restype = ctypes.c_float
        _lib.mfo_icp_loss_grad.restype = ctypes.c_float
    return _lib


def _f(a):
    a = np.ascontiguousarray(a, dtype=np.float32)
    return a, a.ctypes.data_as(c_f)


def set_threads(n):
    os.environ["OMP_NUM_THREADS"] = str(n)  # for a not-yet-loaded runtime
    lib().mfo_set_threads(int(n))


def max_threads():
    return lib().mfo_max_threads()


def average_voxelization_3d(values, points, batch_indices, *, batch_size, origin, pitch, dimensions):
    v, pv = _f(values)
    p, pp = _f(points)
    b = np.ascontiguousarray(batch_indices, dtype=np.int32)
    o, po = _f(origin)
    X, Y, Z = dimensions
    C = v.shape[1]
    m = np.empty((batch_size, C, X, Y, Z), np.float32)
    c = np.empty((batch_size, X, Y, Z), np.int32)
    lib().mfo_average_voxelization_3d(
        pv, pp, b.ctypes.data_as(c_i32), ctypes.c_int64(p.shape[0]), C, batch_size, X, Y, Z, po,
        ctypes.c_float(pitch), m.ctypes.data_as(c_f), c.ctypes.data_as(c_i32))
    return m, c


def interpolate_voxel_grid(vox, points, batch_indices):
    v, pv = _f(vox)
    p, pp = _f(points)
    b = np.ascontiguousarray(batch_indices, dtype=np.int32)
    B, C, X, Y, Z = v.shape
    out = np.empty((p.shape[0], C), np.float32)
    lib().mfo_interpolate_voxel_grid(pv, pp, b.ctypes.data_as(c_i32), ctypes.c_int64(p.shape[0]),
                                     C, X, Y, Z, out.ctypes.data_as(c_f))
    return out


def occupancy_grid_3d(points, *, pitch, origin, dims, threshold=1):
    p, pp = _f(points)
    o, po = _f(origin)
    X, Y, Z = dims
    g = np.empty((X, Y, Z), np.float32)
    lib().mfo_occupancy_grid_3d(pp, ctypes.c_int64(p.shape[0]), ctypes.c_float(pitch), po, X, Y, Z,
                                ctypes.c_float(threshold), g.ctypes.data_as(c_f))
    return g


def nn(ref, query):
    r, pr = _f(ref)
    q, pq = _f(query)
    out = np.empty(q.shape[0], np.int64)
    lib().mfo_nn(pr, ctypes.c_int64(r.shape[0]), pq, ctypes.c_int64(q.shape[0]), out.ctypes.data_as(c_i64))
    return out


def truncated_distance_function(points, *, pitch, origin, dims, truncation):
    p, pp = _f(points)
    o, po = _f(origin)
    X, Y, Z = dims
    tdf = np.empty((X, Y, Z), np.float32)
    flat = np.empty((X, Y, Z), np.int64)
    lib().mfo_tdf(pp, ctypes.c_int64(p.shape[0]), ctypes.c_float(pitch), po, X, Y, Z,
                  ctypes.c_float(truncation), tdf.ctypes.data_as(c_f), flat.ctypes.data_as(c_i64))
    return tdf, flat


def _scene(points, sdf):
    offs = np.zeros(len(points) + 1, np.int64)
    offs[1:] = np.cumsum([len(p) for p in points])
    P = np.ascontiguousarray(np.concatenate(points, 0), dtype=np.float32)
    S = np.ascontiguousarray(np.concatenate(sdf, 0), dtype=np.float32)
    return P, S, offs


def icc_loss_grad(points, sdf, pitch, origin, grid_target, grid_ne, q, t, voxel_dim=32,
                  voxel_threshold=2, sdf_offset=0.0):
    P, S, offs = _scene(points, sdf)
    N = len(points)
    pi, ppi = _f(pitch)
    o, po = _f(origin)
    gt_, pgt = _f(grid_target)
    gn, pgn = _f(grid_ne)
    q_, pq = _f(q)
    t_, pt = _f(t)
    gq = np.empty((N, 4), np.float32)
    gtr = np.empty((N, 3), np.float32)
    sums = np.empty(4, np.float32)
    loss = lib().mfo_icc_loss_grad(
        P.ctypes.data_as(c_f), S.ctypes.data_as(c_f), offs.ctypes.data_as(c_i64), N, ppi, po, pgt,
        pgn, pq, pt, voxel_dim, ctypes.c_float(voxel_threshold), ctypes.c_float(sdf_offset),
        gq.ctypes.data_as(c_f), gtr.ctypes.data_as(c_f), sums.ctypes.data_as(c_f))
    return np.float32(loss), gq, gtr, sums


def icc_refine(points, sdf, pitch, origin, grid_target, grid_ne, q, t, n_iter=100, voxel_dim=32,
               voxel_threshold=2, sdf_offset=0.02, alpha=0.01, return_adam=False):
    """q,t: initial float32 arrays (copied).  Returns q, t, losses, traj[n_iter,N,7]
    (+ adam_hist[n_iter,2,N,7] = (m, v) before each step if return_adam)."""
    P, S, offs = _scene(points, sdf)
    N = len(points)
    pi, ppi = _f(pitch)
    o, po = _f(origin)
    gt_, pgt = _f(grid_target)
    gn, pgn = _f(grid_ne)
    q = np.array(q, dtype=np.float32, copy=True)
    t = np.array(t, dtype=np.float32, copy=True)
    losses = np.empty(n_iter, np.float32)
    traj = np.empty((n_iter, N, 7), np.float32)
    hist = np.empty((n_iter, 2, N, 7), np.float32) if return_adam else None
    lib().mfo_icc_refine(
        P.ctypes.data_as(c_f), S.ctypes.data_as(c_f), offs.ctypes.data_as(c_i64), N, ppi, po, pgt,
        pgn, q.ctypes.data_as(c_f), t.ctypes.data_as(c_f), voxel_dim,
        ctypes.c_float(voxel_threshold), ctypes.c_float(sdf_offset), n_iter,
        ctypes.c_double(alpha), losses.ctypes.data_as(c_f), traj.ctypes.data_as(c_f),
        hist.ctypes.data_as(c_f) if return_adam else None)
    if return_adam:
        return q, t, losses, traj, hist
    return q, t, losses, traj


def icp_loss_grad(source, target, q, t):
    s, ps = _f(source)
    tg, ptg = _f(target)
    q_, pq = _f(q)
    t_, pt = _f(t)
    gq = np.empty(4, np.float32)
    gt = np.empty(3, np.float32)
    loss = lib().mfo_icp_loss_grad(ps, ctypes.c_int64(s.shape[0]), ptg, ctypes.c_int64(tg.shape[0]),
                                   pq, pt, gq.ctypes.data_as(c_f), gt.ctypes.data_as(c_f))
    return np.float32(loss), gq, gt
