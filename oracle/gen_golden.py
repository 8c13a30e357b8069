#!/usr/bin/env python
"""Generate golden vectors by executing the REFERENCE's own NumPy code paths.

TEST INFRASTRUCTURE ONLY.  Runs only in the build container, where
``/root/reference`` is mounted; the GPU box never sees the reference, it only
sees the small ``tests/golden/*.npz`` files this script writes.

How: the reference package cannot be imported (chainer / cupy / trimesh / ...
are absent), but the pure-NumPy bodies of its ``chainer.Function`` subclasses
can be, once a ~100-line stand-in ``chainer`` namespace is injected whose
``Function.__call__`` dispatches to ``forward_cpu``/``forward`` and whose
``functions`` module maps the handful of ``F.*`` wrappers onto NumPy.  The stub
holds NO arithmetic of the path beyond those one-line NumPy forwards: every
number written below comes out of reference source files loaded by path from
``/root/reference`` (never copied into this repository).

Reference files executed (all under /root/reference/morefusion):
  functions/geometry/voxelization_3d.py, average_voxelization_3d.py,
  max_voxelization_3d.py, interpolate_voxel_grid.py, occupancy_grid_3d.py,
  quaternion_matrix.py, compose_transform.py, translation_matrix.py,
  transformation_matrix.py, transform_points.py,
  functions/loss/average_distance.py (ADD branch),
  metrics/ycb_video_add_auc.py, metrics/auc_for_errors.py,
  geometry/pointcloud_from_depth.py, geometry/masks_to_bboxes.py,
  extra/_cupy.py (median)

Usage:  python oracle/gen_golden.py   (writes tests/golden/ref_*.npz)
"""
import importlib.util
import os
import sys
import types

import numpy as np

sys.dont_write_bytecode = True  # /root/reference is read-only

REF = "/root/reference/morefusion"
HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(os.path.dirname(HERE), "tests", "golden")


# --------------------------------------------------------------------------
# stand-in chainer
# --------------------------------------------------------------------------
class Variable:
    """Array holder with the operator surface the reference wrappers use."""

    __array_ufunc__ = None  # ndarray <op> Variable defers to Variable.__r<op>__

    def __init__(self, array):
        self.array = np.asarray(array)

    data = property(lambda self: self.array)
    shape = property(lambda self: self.array.shape)
    ndim = property(lambda self: self.array.ndim)
    dtype = property(lambda self: self.array.dtype)
    T = property(lambda self: Variable(self.array.T))

    def __getitem__(self, k):
        return Variable(self.array[_unwrap(k)])

    def transpose(self, *a):
        return Variable(self.array.transpose(*a))

    def reshape(self, *a):
        return Variable(self.array.reshape(*a))

    def _bin(op):
        def f(self, other):
            return Variable(op(self.array, _unwrap(other)))

        return f

    def _rbin(op):
        def f(self, other):
            return Variable(op(_unwrap(other), self.array))

        return f

    __add__ = _bin(np.add)
    __radd__ = _rbin(np.add)
    __sub__ = _bin(np.subtract)
    __rsub__ = _rbin(np.subtract)
    __mul__ = _bin(np.multiply)
    __rmul__ = _rbin(np.multiply)
    __truediv__ = _bin(np.divide)
    __rtruediv__ = _rbin(np.divide)
    __pow__ = _bin(np.power)

    def __neg__(self):
        return Variable(-self.array)


def _unwrap(x):
    if isinstance(x, Variable):
        return x.array
    if isinstance(x, tuple):
        return tuple(_unwrap(v) for v in x)
    return x


class Function:
    def __call__(self, *inputs):
        arrays = tuple(_unwrap(x) for x in inputs)
        self._inputs = arrays
        fwd = getattr(self, "forward_cpu", None)
        if fwd is None or not _overridden(self, "forward_cpu"):
            fwd = self.forward
        outs = fwd(arrays)
        outs = tuple(Variable(o) for o in outs)
        return outs[0] if len(outs) == 1 else outs

    def retain_inputs(self, idx):
        pass

    def forward(self, inputs):  # chainer's own default dispatch
        return self.forward_cpu(inputs)


def _overridden(obj, name):
    return name in {k for c in type(obj).__mro__[:-2] for k in vars(c)}


def _install_stub():
    chainer = types.ModuleType("chainer")
    chainer.Function = Function
    chainer.Variable = Variable
    chainer.Link = object

    backends = types.ModuleType("chainer.backends")
    cuda = types.ModuleType("chainer.backends.cuda")
    cuda.get_array_module = lambda *a: np
    cuda.to_cpu = lambda x: _unwrap(x)
    backends.cuda = cuda
    chainer.backends = backends
    chainer.cuda = cuda

    utils = types.ModuleType("chainer.utils")
    type_check = types.ModuleType("chainer.utils.type_check")
    type_check.expect = lambda *a, **k: None
    utils.type_check = type_check
    chainer.utils = utils

    F = types.ModuleType("chainer.functions")
    F.sqrt = lambda x: Variable(np.sqrt(_unwrap(x)))
    F.min = lambda x, axis=None: Variable(np.min(_unwrap(x), axis=axis))
    F.relu = lambda x: Variable(np.maximum(_unwrap(x), 0))
    F.minimum = lambda a, b: Variable(np.minimum(_unwrap(a), _unwrap(b)))
    F.absolute = lambda x: Variable(np.absolute(_unwrap(x)))
    F.max = lambda x, axis=None: Variable(np.max(_unwrap(x), axis=axis))
    F.sum = lambda x, axis=None, keepdims=False: Variable(
        np.sum(_unwrap(x), axis=axis, keepdims=keepdims)
    )
    F.mean = lambda x, axis=None: Variable(np.mean(_unwrap(x), axis=axis))
    F.repeat = lambda x, n, axis: Variable(np.repeat(_unwrap(x), n, axis=axis))
    F.matmul = lambda a, b: Variable(np.matmul(_unwrap(a), _unwrap(b)))
    F.concat = lambda xs, axis=1: Variable(
        np.concatenate([_unwrap(x) for x in xs], axis=axis)
    )
    chainer.functions = F

    for name, mod in {
        "chainer": chainer,
        "chainer.backends": backends,
        "chainer.backends.cuda": cuda,
        "chainer.utils": utils,
        "chainer.utils.type_check": type_check,
        "chainer.functions": F,
    }.items():
        sys.modules[name] = mod

    # empty parent packages so the reference's relative imports resolve
    for pkg, sub in [
        ("morefusion", ""),
        ("morefusion.functions", "functions"),
        ("morefusion.functions.geometry", "functions/geometry"),
        ("morefusion.functions.loss", "functions/loss"),
        ("morefusion.geometry", "geometry"),
        ("morefusion.metrics", "metrics"),
        ("morefusion.extra", "extra"),
    ]:
        m = types.ModuleType(pkg)
        m.__path__ = [os.path.join(REF, sub)]
        sys.modules[pkg] = m


def _load(modname, relpath):
    spec = importlib.util.spec_from_file_location(
        modname, os.path.join(REF, relpath)
    )
    mod = importlib.util.module_from_spec(spec)
    sys.modules[modname] = mod
    spec.loader.exec_module(mod)
    return mod


def main():
    _install_stub()
    os.makedirs(OUT, exist_ok=True)
    g = "morefusion.functions.geometry"
    _load(g + ".voxelization_3d", "functions/geometry/voxelization_3d.py")
    avg = _load(
        g + ".average_voxelization_3d",
        "functions/geometry/average_voxelization_3d.py",
    )
    mx = _load(
        g + ".max_voxelization_3d", "functions/geometry/max_voxelization_3d.py"
    )
    itp = _load(
        g + ".interpolate_voxel_grid",
        "functions/geometry/interpolate_voxel_grid.py",
    )
    occ = _load(g + ".occupancy_grid_3d", "functions/geometry/occupancy_grid_3d.py")
    qm = _load(g + ".quaternion_matrix", "functions/geometry/quaternion_matrix.py")
    ct = _load(g + ".compose_transform", "functions/geometry/compose_transform.py")
    tm = _load(g + ".translation_matrix", "functions/geometry/translation_matrix.py")
    tfm = _load(
        g + ".transformation_matrix", "functions/geometry/transformation_matrix.py"
    )
    tp = _load(g + ".transform_points", "functions/geometry/transform_points.py")
    sys.modules[g].transform_points = tp.transform_points
    # functions/loss/average_distance.py does `from ... import geometry` (for
    # nn, ADD-S only) and `from ..geometry import transform_points`
    sys.modules["morefusion"].geometry = sys.modules["morefusion.geometry"]
    ad = _load(
        "morefusion.functions.loss.average_distance",
        "functions/loss/average_distance.py",
    )
    auc = _load(
        "morefusion.metrics.ycb_video_add_auc", "metrics/ycb_video_add_auc.py"
    )
    auce = _load("morefusion.metrics.auc_for_errors", "metrics/auc_for_errors.py")
    pfd = _load(
        "morefusion.geometry.pointcloud_from_depth",
        "geometry/pointcloud_from_depth.py",
    )
    m2b = _load(
        "morefusion.geometry.masks_to_bboxes", "geometry/masks_to_bboxes.py"
    )
    xcupy = _load("morefusion.extra._cupy", "extra/_cupy.py")

    rs = np.random.RandomState(0)

    # ---- A1/A2 average_voxelization_3d: the reference test's own setup
    # (tests/functions_tests/geometry_tests/test_average_voxelization_3d.py:20-42)
    D, C, B, P = 32, 4, 3, 128
    origin = np.array([-1, -1, -1], dtype=np.float32)
    pitch = np.float32(2.0 / D)
    points = rs.uniform(-1, 1, (P, 3)).astype(np.float32)
    values = rs.uniform(-1, 1, (P, C)).astype(np.float32)
    bidx = rs.randint(0, B, P).astype(np.int32)
    gy = rs.uniform(-1, 1, (B, C, D, D, D)).astype(np.float32)
    f = avg.AverageVoxelization3D(
        batch_size=B, pitch=pitch, origin=origin, dimensions=(D, D, D)
    )
    y = f(values, points, bidx).array
    gvalues = f.backward_cpu((values, points, bidx), (gy,))[0]
    nz = np.flatnonzero(gy.reshape(-1))  # keep golden small: store gy sparse
    np.savez_compressed(
        os.path.join(OUT, "ref_average_voxelization_3d.npz"),
        origin=origin, pitch=pitch, points=points, values=values,
        batch_indices=bidx, batch_size=B, dims=np.array([D, D, D]),
        y_nonzero_index=np.flatnonzero(y.reshape(-1)).astype(np.int64),
        y_nonzero_value=y.reshape(-1)[np.flatnonzero(y.reshape(-1))],
        counts_nonzero_index=np.flatnonzero(f.counts.reshape(-1)).astype(np.int64),
        counts_nonzero_value=f.counts.reshape(-1)[np.flatnonzero(f.counts.reshape(-1))],
        gy_seed=np.int64(1), gvalues_note="gy = RandomState(1).uniform(-1,1,shape).astype(f4)",
    )
    # backward golden with a reproducible gy (regenerated in the test from seed)
    gy = np.random.RandomState(1).uniform(-1, 1, (B, C, D, D, D)).astype(np.float32)
    gvalues = f.backward_cpu((values, points, bidx), (gy,))[0]
    d = dict(np.load(os.path.join(OUT, "ref_average_voxelization_3d.npz")))
    d["gvalues"] = gvalues
    np.savez_compressed(os.path.join(OUT, "ref_average_voxelization_3d.npz"), **d)

    # model-shaped case: 2 objects x 1000 points, C=8, origin (0,0,0), pitch 1
    # (contrib/singleview_3d/models/model.py:154-162), many voxel collisions
    P2, C2, B2 = 1000, 8, 2
    pts2 = rs.uniform(8, 24, (B2 * P2, 3)).astype(np.float32)
    pts2[:50] = rs.uniform(-3, 35, (50, 3)).astype(np.float32)  # some out of range
    val2 = rs.uniform(-1, 1, (B2 * P2, C2)).astype(np.float32)
    bi2 = np.arange(B2, dtype=np.int32).repeat(P2)
    f2 = avg.AverageVoxelization3D(
        batch_size=B2, pitch=1.0, origin=(0, 0, 0), dimensions=(D, D, D)
    )
    y2 = f2(val2, pts2, bi2).array
    np.savez_compressed(
        os.path.join(OUT, "ref_average_voxelization_3d_model.npz"),
        points=pts2, values=val2, batch_indices=bi2, batch_size=B2,
        y_nonzero_index=np.flatnonzero(y2.reshape(-1)).astype(np.int64),
        y_nonzero_value=y2.reshape(-1)[np.flatnonzero(y2.reshape(-1))],
        counts_nonzero_index=np.flatnonzero(f2.counts.reshape(-1)).astype(np.int64),
        counts_nonzero_value=f2.counts.reshape(-1)[np.flatnonzero(f2.counts.reshape(-1))],
    )

    # ---- A3 max_voxelization_3d (test_max_voxelization_3d.py:18-45)
    inten = np.sqrt((values ** 2).sum(axis=1)).astype(np.float32)
    fm = mx.MaxVoxelization3D(
        batch_size=B, pitch=pitch, origin=origin, dimensions=(D, D, D)
    )
    ym = fm(values, points, bidx, inten).array
    gvm = fm.backward_cpu((values, points, bidx, inten), (gy,))[0]
    np.savez_compressed(
        os.path.join(OUT, "ref_max_voxelization_3d.npz"),
        origin=origin, pitch=pitch, points=points, values=values,
        batch_indices=bidx, intensities=inten, batch_size=B,
        y_nonzero_index=np.flatnonzero(ym.reshape(-1)).astype(np.int64),
        y_nonzero_value=ym.reshape(-1)[np.flatnonzero(ym.reshape(-1))],
        indices_valid_index=np.flatnonzero(fm.indices.reshape(-1) >= 0).astype(np.int64),
        indices_valid_value=fm.indices.reshape(-1)[fm.indices.reshape(-1) >= 0],
        gvalues=gvm,
    )

    # ---- A4 interpolate_voxel_grid forward_cpu (test_interpolate_voxel_grid.py:19-40)
    Bi, Ci, Di, Pi = 3, 4, 32, 128
    vox = rs.uniform(-1, 1, (Bi, Ci, Di, Di, Di)).astype(np.float32)
    # the golden must stay small: commit a seed for vox, not the 1.5 MB array
    vox = np.random.RandomState(7).uniform(-1, 1, (Bi, Ci, Di, Di, Di)).astype(np.float32)
    pti = rs.uniform(0, Di - 1, (Pi, 3)).astype(np.float32)
    pti[:8] = rs.uniform(-2, Di + 1, (8, 3)).astype(np.float32)  # out-of-range corners
    pti[8] = (5.0, 6.0, 7.0)  # exactly on a lattice node
    pti[9] = (31.0, 31.0, 31.0)  # high corner falls outside
    bii = rs.randint(0, Bi, Pi).astype(np.int32)
    vi = itp.InterpolateVoxelGrid()(vox, pti, bii).array
    np.savez_compressed(
        os.path.join(OUT, "ref_interpolate_voxel_grid.npz"),
        vox_seed=np.int64(7), vox_shape=np.array([Bi, Ci, Di, Di, Di]),
        points=pti, batch_indices=bii, values=vi,
    )

    # ---- A5 occupancy_grid_3d: reference golden + random case + backward
    # (test_occupancy_grid_3d.py:13-38)
    og_pts = np.array([[0, 0.05, 0.1], [3.9, 3.95, 4]], dtype=np.float32)
    og = occ.occupancy_grid_3d(og_pts, pitch=1, origin=(0, 0, 0), dims=(5, 5, 5)).array
    pitch_c1 = 0.008705823111730123  # class 2, ros/.../utils/data.h:13
    pts_c1 = (rs.uniform(0, 32, (1000, 3)) * pitch_c1 - 16 * pitch_c1).astype(np.float32)
    og_c1 = occ.occupancy_grid_3d(
        pts_c1, pitch=pitch_c1, origin=(-16 * pitch_c1,) * 3, dims=(32, 32, 32)
    ).array
    og_c1_t2 = occ.occupancy_grid_3d(
        pts_c1[:200], pitch=pitch_c1, origin=(-16 * pitch_c1,) * 3,
        dims=(32, 32, 32), threshold=2,
    ).array
    # hand-written backward of the Function part (occupancy_grid_3d.py:56-74)
    fo = occ.OccupancyGrid3D(pitch=pitch_c1, origin=(-16 * pitch_c1,) * 3, dims=(8, 8, 8))
    p_small = pts_c1[:16]
    dI, dJ, dK = fo.forward((p_small,))
    gI = np.random.RandomState(3).uniform(-1, 1, dI.shape).astype(np.float32)
    gJ = np.random.RandomState(4).uniform(-1, 1, dI.shape).astype(np.float32)
    gK = np.random.RandomState(5).uniform(-1, 1, dI.shape).astype(np.float32)
    gp = fo.backward((p_small,), (gI, gJ, gK))[0]
    np.savez_compressed(
        os.path.join(OUT, "ref_occupancy_grid_3d.npz"),
        known_points=og_pts, known_grid=og,
        c1_points=pts_c1, c1_pitch=pitch_c1, c1_grid=og_c1, c1_grid_thr2=og_c1_t2,
        fn_points=p_small, fn_dI=dI, fn_dJ=dJ, fn_dK=dK, fn_gpoints=gp,
    )

    # ---- A8 rigid transforms
    q = rs.uniform(-1, 1, (5, 4)).astype(np.float32)
    t = rs.uniform(-1, 1, (5, 3)).astype(np.float32)
    Tq = qm.quaternion_matrix(q).array
    Tq1 = qm.quaternion_matrix(q[0]).array
    gR = rs.uniform(-1, 1, (5, 4, 4)).astype(np.float32)
    gq_outer = qm.QuaternionMatrix().backward((None,), (gR,))[0]
    T = tfm.transformation_matrix(q, t).array
    T1 = tfm.transformation_matrix(q[0], t[0]).array
    Tt = tm.translation_matrix(t).array
    Tc = ct.compose_transform(Tq[:, :3, :3], t).array
    pts = rs.uniform(-1, 1, (128, 3)).astype(np.float32)
    tpts = tp.transform_points(pts, T).array
    tpts1 = tp.transform_points(pts, T[0]).array
    np.savez_compressed(
        os.path.join(OUT, "ref_transforms.npz"),
        q=q, t=t, quaternion_matrix=Tq, quaternion_matrix_1d=Tq1,
        gR=gR, gq_outer=gq_outer, transformation_matrix=T,
        transformation_matrix_1d=T1, translation_matrix=Tt, compose_transform=Tc,
        points=pts, transform_points=tpts, transform_points_1=tpts1,
    )

    # ---- A12 functions.average_distance, ADD branch (test_average_distance.py:14-31)
    add = ad.average_distance(pts, T[0], T[1:], symmetric=False).array
    np.savez_compressed(
        os.path.join(OUT, "ref_average_distance.npz"),
        points=pts, transform_true=T[0], transforms_pred=T[1:], add=add,
    )

    # ---- A15 metrics
    errs = np.abs(rs.normal(0, 0.05, 200))
    a1, x1, y1 = auc.ycb_video_add_auc(errs, return_xy=True)
    a2 = auc.ycb_video_add_auc(errs * 10)  # mostly > max_value
    a3 = auc.ycb_video_add_auc(np.full(5, 1.0))  # nothing finite
    a4 = auce.auc_for_errors(errs, 0.1)
    np.savez_compressed(
        os.path.join(OUT, "ref_metrics.npz"),
        errors=errs, add_auc=a1, add_auc_x=x1, add_auc_y=y1,
        add_auc_x10=a2, add_auc_none=np.float64(a3), auc_for_errors=a4,
    )

    # ---- pre-processing callers (next rows): pointcloud_from_depth, masks_to_bboxes, median
    depth = rs.uniform(0.4, 1.2, (24, 32)).astype(np.float32)
    depth[rs.uniform(size=depth.shape) < 0.1] = np.nan
    pc_z = pfd.pointcloud_from_depth(depth, fx=30.0, fy=31.0, cx=15.5, cy=11.5)
    pc_e = pfd.pointcloud_from_depth(
        depth, fx=30.0, fy=31.0, cx=15.5, cy=11.5, depth_type="euclidean"
    )
    masks = np.zeros((3, 24, 32), dtype=bool)
    masks[0, 3:9, 4:20] = True
    masks[1, 10, 11] = True
    bb = m2b.masks_to_bboxes(masks)
    med_in = rs.uniform(-1, 1, (10, 3)).astype(np.float32)
    np.savez_compressed(
        os.path.join(OUT, "ref_preprocess.npz"),
        depth=depth, pc_z=pc_z, pc_euclid=pc_e, masks=masks, bboxes=bb,
        median_in=med_in, median_even=xcupy.median(med_in, axis=0),
        median_odd=xcupy.median(med_in[:9], axis=0),
        median_flat=xcupy.median(med_in),
    )
    # ---- A5 siblings: occupancy_grid_1d / occupancy_grid_2d (functions/geometry/
    # occupancy_grid_1d.py:9-60, occupancy_grid_2d.py:10-75; the latter asserts with
    # collections.Sequence, gone from Python >= 3.10 -> alias it for the import only)
    import collections
    import collections.abc
    if not hasattr(collections, "Sequence"):
        collections.Sequence = collections.abc.Sequence
    o1 = _load(g + ".occupancy_grid_1d", "functions/geometry/occupancy_grid_1d.py")
    o2 = _load(g + ".occupancy_grid_2d", "functions/geometry/occupancy_grid_2d.py")
    p1 = np.array([0.05, 3.9, 2.5, -0.4], dtype=np.float32)      # incl. the module's own self-check points
    m1 = o1.occupancy_grid_1d(p1, pitch=1, origin=0, dimension=5).array
    m1b = o1.occupancy_grid_1d(p1 * 0.1, pitch=0.1, origin=-0.05, dimension=8).array
    p2 = rs.uniform(-0.5, 4.5, (6, 2)).astype(np.float32)
    m2 = o2.occupancy_grid_2d(p2, pitch=1, origin=(0, 0), dimension=(5, 6)).array
    m2b = o2.occupancy_grid_2d(p2, pitch=0.5, origin=(-1.0, 0.5), dimension=(9, 7), threshold=2).array
    f1 = o1.OccupancyGrid1D(pitch=1, origin=0, dimension=5)
    g1 = rs.uniform(-1, 1, (p1.shape[0], 5)).astype(np.float32)
    f1(p1)
    gp1 = f1.backward((p1,), (g1,))[0]
    f2 = o2.OccupancyGrid2D(pitch=0.5, origin=(-1.0, 0.5), dimension=(9, 7))
    g2a = rs.uniform(-1, 1, (7, 9, p2.shape[0])).astype(np.float32)
    g2b = rs.uniform(-1, 1, (7, 9, p2.shape[0])).astype(np.float32)
    d_ik, d_jk = f2(p2)
    gp2 = f2.backward((p2,), (g2a, g2b))[0]
    np.savez_compressed(
        os.path.join(OUT, "ref_occupancy_grid_12d.npz"),
        p1=p1, m1=m1, m1b=m1b, g1=g1, gp1=gp1, p2=p2, m2=m2, m2b=m2b,
        d_ik=d_ik.array, d_jk=d_jk.array, g2a=g2a, g2b=g2b, gp2=gp2,
    )
    for fn in sorted(os.listdir(OUT)):
        print(fn, os.path.getsize(os.path.join(OUT, fn)))


if __name__ == "__main__":
    main()
