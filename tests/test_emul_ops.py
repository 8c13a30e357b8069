"""Kernel SOURCES other than the ICC compiled for the host behind tests/host_emul (fiber emulator)
and checked against the oracle without a GPU: the fused ADD / ADD-S loss (csrc/loss.hip) and
interpolate_voxel_grid with per-item row ranges (csrc/interp.hip)."""
import ctypes
import os
import sys

import numpy as np
import pytest

from oracle import oracle_np as O

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "host_emul"))
import emul  # noqa: E402

pytestmark = pytest.mark.skipif(not emul.available(), reason="g++ not available")
_p, _i32 = ctypes.c_void_p, ctypes.c_int32


def _pose(rs):
    from morefusion_amd.synthetic import random_rotation
    T = np.eye(4, dtype=np.float32)
    T[:3, :3] = random_rotation(rs, 1.0)
    T[:3, 3] = rs.uniform(-0.05, 0.05, 3)
    return T


def test_add_loss_kernel_source_vs_oracle():
    lib = emul.build(["loss.hip"])
    lib.mf_average_distance_fwd.argtypes = [_p, _p, _p, _p, _i32, _i32, _i32, _p, _p, _p]
    lib.mf_average_distance_bwd.argtypes = [_p, _p, _p, _p, _p, _i32, _i32, _i32, _p, _p, _p]
    rs = np.random.RandomState(0)
    B, M, P = 3, 300, 5
    pts = rs.uniform(-0.05, 0.05, (B, M, 3)).astype(np.float32)
    Tt = np.stack([_pose(rs) for _ in range(B)])
    Tp = np.stack([[(_pose(rs) @ np.eye(4)).astype(np.float32) for _ in range(P)] for _ in range(B)])
    for b in range(B):  # predictions near the truth: ADD-S then really picks other points
        for p in range(P):
            Tp[b, p] = Tt[b]
            Tp[b, p, :3, 3] += rs.uniform(-0.01, 0.01, 3).astype(np.float32)
    sym = np.array([0, 1, 1], np.uint8)
    out = np.zeros((B, P), np.float32)
    idx = np.zeros((B, P, M), np.int32)
    assert lib.mf_average_distance_fwd(pts.ctypes.data, Tt.ctypes.data, Tp.ctypes.data, sym.ctypes.data, B, M, P,
                                       out.ctypes.data, idx.ctypes.data, None) == 0
    for b in range(B):
        ref = O.average_distance(pts[b], Tt[b], Tp[b], symmetric=bool(sym[b]))
        np.testing.assert_allclose(out[b], ref, rtol=2e-6, atol=1e-8)
        if sym[b]:  # the arg-min itself, bit for bit (knn/nn.py:48: first minimum); M = 300: the last group of 4 is full,
            pt = O.transform_points(pts[b], Tt[b])  # M2 below: padded
            want = O.nn(pt, O.transform_points(pts[b], Tp[b]).reshape(-1, 3)).reshape(P, M)
            assert np.array_equal(idx[b], want)
    # ties and a tile that is not a multiple of 4: duplicated model points -> equal distances, lowest index wins
    M2 = 37
    pts2 = np.ascontiguousarray(pts[:, :M2]).copy()
    pts2[:, 20:30] = pts2[:, 5:15]
    out2 = np.zeros((B, P), np.float32)
    idx2 = np.zeros((B, P, M2), np.int32)
    sym2 = np.ones(B, np.uint8)
    assert lib.mf_average_distance_fwd(pts2.ctypes.data, Tt.ctypes.data, Tp.ctypes.data, sym2.ctypes.data, B, M2, P,
                                       out2.ctypes.data, idx2.ctypes.data, None) == 0
    for b in range(B):
        pt = O.transform_points(pts2[b], Tt[b])
        want = O.nn(pt, O.transform_points(pts2[b], Tp[b]).reshape(-1, 3)).reshape(P, M2)
        assert np.array_equal(idx2[b], want)
        assert not np.isin(idx2[b], np.arange(20, 30)).any()  # never the later copy
        np.testing.assert_allclose(out2[b], O.average_distance(pts2[b], Tt[b], Tp[b], symmetric=True), rtol=2e-6, atol=1e-8)
    # backward: against central differences of the forward (float64 restatement of the same formula)
    gout = rs.uniform(0.5, 1.5, (B, P)).astype(np.float32)
    for use_idx in (True, False):
        gT = np.full((B, P, 4, 4), 7.0, np.float32)
        assert lib.mf_average_distance_bwd(pts.ctypes.data, Tt.ctypes.data, Tp.ctypes.data, sym.ctypes.data,
                                           gout.ctypes.data, B, M, P, idx.ctypes.data if use_idx else None,
                                           gT.ctypes.data, None) == 0
        assert (gT[:, :, 3] == 0).all()
        for b, p in ((0, 1), (1, 0), (2, 4)):
            num = np.zeros((3, 4))
            for i in range(3):
                for j in range(4):
                    vals = []
                    for sgn in (+1, -1):
                        T = Tp[b, p].astype(np.float64).copy()
                        T[i, j] += sgn * 1e-6
                        true = pts[b].astype(np.float64) @ Tt[b, :3, :3].T.astype(np.float64) + Tt[b, :3, 3]
                        pred = pts[b].astype(np.float64) @ T[:3, :3].T + T[:3, 3]
                        tr = true[idx[b, p]] if sym[b] else true  # arg-min frozen, like the reference
                        vals.append(np.linalg.norm(tr - pred, axis=1).mean())
                    num[i, j] = (vals[0] - vals[1]) / 2e-6 * gout[b, p]
            np.testing.assert_allclose(gT[b, p, :3], num, rtol=2e-3, atol=2e-5)


@pytest.mark.parametrize("channels_first", [0, 1])
def test_interpolate_kernel_source_row_ranges_vs_oracle(channels_first):
    lib = emul.build(["interp.hip"])
    lib.mf_interpolate_voxel_grid_fwd.argtypes = [_p, _p, _p, _p, ctypes.c_int64] + [ctypes.c_int] * 5 + [_p, ctypes.c_int, _p]
    rs = np.random.RandomState(1)
    B, C, X = 3, 6, 8
    counts = [120, 0, 77]
    orphan = 9
    n = sum(counts) + orphan
    bi = np.concatenate([np.full(c, b, np.int32) for b, c in enumerate(counts)] + [np.full(orphan, -2, np.int32)])
    start = np.concatenate([[0], np.cumsum(counts)]).astype(np.int32)
    vox = rs.uniform(-1, 1, (B, C, X, X, X)).astype(np.float32)
    pts = (rs.uniform(-0.15, 1.1, (n, 3)) * X).astype(np.float32)
    valid = bi >= 0
    ref = O.interpolate_voxel_grid(vox, pts[valid], bi[valid], mode="gpu")
    for bs in (None, start):
        out = np.full((C, n) if channels_first else (n, C), 5.0, np.float32)
        assert lib.mf_interpolate_voxel_grid_fwd(vox.ctypes.data, pts.ctypes.data, bi.ctypes.data,
                                                 None if bs is None else bs.ctypes.data, n, B, C, X, X, X,
                                                 out.ctypes.data, channels_first, None) == 0
        got = out.T if channels_first else out
        np.testing.assert_array_equal(got[valid], ref)
        np.testing.assert_array_equal(got[~valid], 0.0)


@pytest.mark.parametrize("C", [1024, 2048])
def test_interpolate_voxel_major_kernel_source_vs_oracle(C):
    """The shapes that select k_interp_fwd_vm<4> / <8> (chunk stored voxel-major in LDS, one 16-byte
    gather per 4 channels): bit-identical to the oracle's GPU-order sums, both output layouts,
    with and without row ranges."""
    lib = emul.build(["interp.hip"])
    lib.mf_interpolate_voxel_grid_fwd.argtypes = [_p, _p, _p, _p, ctypes.c_int64] + [ctypes.c_int] * 5 + [_p, ctypes.c_int, _p]
    rs = np.random.RandomState(2)
    B, X = 2, 8
    counts = [37, 21]
    n = sum(counts) + 3
    bi = np.concatenate([np.full(c, b, np.int32) for b, c in enumerate(counts)] + [np.full(3, 7, np.int32)])
    start = np.concatenate([[0], np.cumsum(counts)]).astype(np.int32)
    vox = rs.uniform(-1, 1, (B, C, X, X, X)).astype(np.float32)
    pts = (rs.uniform(-0.15, 1.1, (n, 3)) * X).astype(np.float32)
    valid = bi < B
    ref = O.interpolate_voxel_grid(vox, pts[valid], bi[valid], mode="gpu")
    for channels_first, bs in ((1, start), (0, None)):
        out = np.full((C, n) if channels_first else (n, C), 5.0, np.float32)
        assert lib.mf_interpolate_voxel_grid_fwd(vox.ctypes.data, pts.ctypes.data, bi.ctypes.data,
                                                 None if bs is None else bs.ctypes.data, n, B, C, X, X, X,
                                                 out.ctypes.data, channels_first, None) == 0
        got = out.T if channels_first else out
        np.testing.assert_array_equal(got[valid], ref)
        np.testing.assert_array_equal(got[~valid], 0.0)


def test_fused_icp_loop_kernel_source_vs_oracle(fixtures3):
    """mf_icp_refine (k_icp over a batch of links + k_icp_step: chain rule, chainer-Adam, next R|t;
    the driver loop of check_iterative_closest_point_link.py:40-70) against the oracle's loop: the
    committed 30-iterate trajectory of fixture 2 and a second link refined in the same batch."""
    from conftest import golden
    from oracle import oracle_c as OC
    lib = emul.build(["occgrid_knn.hip"])
    lib.mf_icp_refine.argtypes = [_p, _p, _p, _p, _i32, _i32, ctypes.c_float, _p, _p, _p, _p, _i32, _i32,
                                  ctypes.c_float, ctypes.c_float, _p, _p, _p]
    g = golden("oracle_icc_icp_trajectories.npz")
    links = []
    for f in (fixtures3[2], fixtures3[0]):
        # argwhere hands back a transposed view: force row-major [T,3]
        target = np.ascontiguousarray((np.argwhere(f["grid_target"] >= 0.5) * f["pitch"] + f["origin"]).astype(np.float32))
        source = np.ascontiguousarray(f["pcd_cad"].astype(np.float32)[:: 4 if f is fixtures3[0] else 1])
        links.append((source, target, O.quaternion_from_matrix(f["transform_init"]).astype(np.float32),
                      f["transform_init"][:3, 3].astype(np.float32)))
    n_iter = 6
    src = np.ascontiguousarray(np.concatenate([k[0] for k in links]))
    tgt = np.ascontiguousarray(np.concatenate([k[1] for k in links]))
    src_off = np.r_[0, np.cumsum([len(k[0]) for k in links])].astype(np.int32)
    tgt_off = np.r_[0, np.cumsum([len(k[1]) for k in links])].astype(np.int32)
    q = np.stack([k[2] for k in links]).copy()
    t = np.stack([k[3] for k in links]).copy()
    m, v = np.zeros((2, 7), np.float32), np.zeros((2, 7), np.float32)
    losses = np.zeros((n_iter, 2), np.float32)
    ws = np.zeros(28 * 2, np.float32)
    rc = lib.mf_icp_refine(src.ctypes.data, src_off.ctypes.data, tgt.ctypes.data, tgt_off.ctypes.data, 2,
                           int(max(len(k[1]) for k in links)), 0.02, q.ctypes.data, t.ctypes.data, m.ctypes.data,
                           v.ctypes.data, n_iter, 0, 0.01, 0.001, losses.ctypes.data, ws.ctypes.data, None)
    assert rc == 0
    np.testing.assert_allclose(losses[:, 0], g["icp_losses"][:n_iter], rtol=1e-4)   # committed golden
    np.testing.assert_allclose(np.r_[q[0], t[0]], g["icp_traj"][n_iter], atol=2e-5)
    # link 1: the oracle's loop at test time
    qi, ti = links[1][2].copy(), links[1][3].copy()
    opt = O.ChainerAdam([qi, ti], [0.01, 0.001])
    for k in range(n_iter):
        loss, gq, gt = OC.icp_loss_grad(links[1][0], links[1][1], qi, ti)
        np.testing.assert_allclose(losses[k, 1], loss, rtol=1e-4)
        opt.update([gq, gt])
    np.testing.assert_allclose(np.r_[q[1], t[1]], np.r_[qi, ti], atol=2e-5)


def test_valid_pixel_order_kernel_source_vs_numpy_where():
    """mf_valid_pixel_order (k_valid_order: shuffle prefix + wave totals + running base) against
    np.where(~isnan(pcd).any(-1)) (model.py:195-196,206): vector-load and scalar paths, a ragged
    tail, an image without valid pixels, +-inf coordinates (valid: only NaN masks a pixel)."""
    lib = emul.build(["preprocess.hip"])
    lib.mf_valid_pixel_order.argtypes = [_p, _i32, _i32, _p, _p, _p]
    rs = np.random.RandomState(3)
    for HW, off in ((4096 + 2048, 0), (5003, 0), (4096, 1), (4096 * 6 + 8, 0), (4096 * 2 + 5, 1)):
        B = 3
        buf = np.zeros(B * HW * 3 + 4, np.float32)
        pcd = buf[off:off + B * HW * 3].reshape(B, HW, 3)   # off = 1: not 16-byte aligned
        pcd[:] = rs.randn(B, HW, 3)
        pcd[0][rs.rand(HW) < 0.6] = np.nan
        pcd[1][:, 1][rs.rand(HW) < 0.3] = np.nan
        pcd[1][5, 2] = np.inf
        pcd[2] = np.nan
        order = np.full((B, HW), -7, np.int32)
        counts = np.full(B, -1, np.int32)
        assert lib.mf_valid_pixel_order(pcd.ctypes.data, B, HW, order.ctypes.data, counts.ctypes.data, None) == 0
        for b in range(B):
            want = np.flatnonzero(~np.isnan(pcd[b]).any(axis=1))
            assert counts[b] == len(want)
            np.testing.assert_array_equal(order[b, :len(want)], want)
            assert (order[b, len(want):] == -7).all()


def test_sparse_conv3_kernel_sources_vs_dense_conv():
    """sparseconv.hip on the host emulator (the fp32 MFMA builtin emulated lane-exactly in the shim):
    points -> chains -> compact GEMM rows -> 8 parity-class MFMA GEMMs -> output-stationary reduce,
    against a dense float64 k4 / s2 / p1 convolution of the oracle's average voxelization; the
    dense-input entry point must give the same bits as the points-fed one."""
    import torch
    from oracle import oracle_np as O_
    lib = emul.build(["sparseconv.hip"])
    i64 = ctypes.c_int64
    lib.mf_sparse_conv3d_workspace_bytes.restype = i64
    lib.mf_sparse_conv3d_workspace_bytes.argtypes = [_i32] * 5 + [i64]
    lib.mf_sparse_conv3d_pack_weights.argtypes = [_p, _i32, _i32, _i32, _i32, _p, _p]
    lib.mf_sparse_conv3d_k4s2_points_fwd.argtypes = [_p, _p, _p, i64] + [ctypes.c_float] * 4 + [_p] * 5 + [_i32] * 6 + [_p]
    lib.mf_sparse_conv3d_k4s2_fwd.argtypes = [_p] * 7 + [_i32] * 6 + [_p]
    rs = np.random.RandomState(5)
    B, Cs, Cout, D, n = 2, 8, 64, 8, 70
    points = rs.uniform(-0.6, D - 0.4, (n, 3)).astype(np.float32)
    points[:10] = points[10:20]                        # shared voxels: means over several points
    values = rs.uniform(-1, 1, (n, Cs)).astype(np.float32)
    bi = np.sort(rs.randint(0, B, n)).astype(np.int32)
    W = (rs.uniform(-1, 1, (Cout, Cs + 3, 4, 4, 4)) * 0.2).astype(np.float32)   # conv weight with extra input channels
    bias = rs.uniform(-0.1, 0.1, Cout).astype(np.float32)
    dense = rs.uniform(-0.1, 0.1, (B, Cout, D // 2, D // 2, D // 2)).astype(np.float32)
    Wp = np.zeros(8 * Cs * 8 * Cout, np.float32)
    assert lib.mf_sparse_conv3d_pack_weights(W.ctypes.data, Cout, Cs, Cs + 3, 2, Wp.ctypes.data, None) == 0
    ws = np.zeros(int(lib.mf_sparse_conv3d_workspace_bytes(B, Cs, Cout, D, n, n)) // 4 + 64, np.float32)
    out = np.full(dense.shape, 7.0, np.float32)
    assert lib.mf_sparse_conv3d_k4s2_points_fwd(values.ctypes.data, points.ctypes.data, bi.ctypes.data, n, 0.0, 0.0, 0.0,
                                                1.0, Wp.ctypes.data, dense.ctypes.data, bias.ctypes.data,
                                                out.ctypes.data, ws.ctypes.data, B, Cs, Cout, D, n, 1, None) == 0
    vox, counts = O_.average_voxelization_3d(values, points, bi, batch_size=B, origin=(0, 0, 0), pitch=1.0,
                                             dimensions=(D, D, D), mode="gpu")
    ref = torch.nn.functional.conv3d(torch.from_numpy(vox).double(), torch.from_numpy(W[:, 2:2 + Cs]).double(),
                                     stride=2, padding=1).numpy()
    ref = np.maximum(ref + dense + bias[None, :, None, None, None], 0)
    np.testing.assert_allclose(out, ref, rtol=1e-4, atol=1e-5)
    assert (out > 0).mean() > 0.2
    out2 = np.full(dense.shape, 7.0, np.float32)
    ws2 = np.zeros_like(ws)
    assert lib.mf_sparse_conv3d_k4s2_fwd(vox.ctypes.data, counts.ctypes.data, Wp.ctypes.data, dense.ctypes.data,
                                         bias.ctypes.data, out2.ctypes.data, ws2.ctypes.data, B, Cs, Cout, D, n, 1,
                                         None) == 0
    np.testing.assert_array_equal(out2, out)


def test_voxelize_kernel_sources_vs_reference_cuda_text():
    """voxelize.hip (chains + wave-per-voxel ordered sums; 64-bit arg-max keys) on the host emulator,
    directly against the output of the REFERENCE's CUDA text (tests/golden/ref_cuda_voxelization.npz):
    average fwd + bwd and max fwd + bwd, bit-exact (round-half-away, intensity ties)."""
    from conftest import golden
    lib = emul.build(["voxelize.hip"])
    i64, f = ctypes.c_int64, ctypes.c_float
    lib.mf_average_voxelization_3d_fwd.argtypes = [_p, _p, _p, i64] + [ctypes.c_int] * 5 + [f] * 4 + [_p] * 6
    lib.mf_average_voxelization_3d_bwd.argtypes = [_p, _p, _p, _p, i64] + [ctypes.c_int] * 5 + [f] * 4 + [_p, _p]
    lib.mf_max_voxelization_3d_fwd.argtypes = [_p, _p, _p, _p, i64] + [ctypes.c_int] * 5 + [f] * 4 + [_p] * 5
    lib.mf_max_voxelization_3d_bwd.argtypes = [_p, _p, i64] + [ctypes.c_int] * 5 + [_p, _p]
    g = golden("ref_cuda_voxelization.npz")
    D, B = int(g["dim"]), int(g["batch_size"])
    values, points, bi = (np.ascontiguousarray(g[k]) for k in ("values", "points", "batch_indices"))
    n, C = values.shape
    o, pitch = [float(v) for v in g["origin"]], float(g["pitch"])
    matrix = np.full((B, C, D, D, D), 9.0, np.float32)
    counts = np.zeros((B, D, D, D), np.int32)
    head = np.zeros(B * D ** 3, np.int32)
    link = np.zeros(n, np.int32)
    nan_flag = np.zeros(1, np.int32)
    assert lib.mf_average_voxelization_3d_fwd(values.ctypes.data, points.ctypes.data, bi.ctypes.data, n, C, B, D, D, D,
                                              *o, pitch, matrix.ctypes.data, counts.ctypes.data, head.ctypes.data,
                                              link.ctypes.data, nan_flag.ctypes.data, None) == 0
    np.testing.assert_array_equal(counts, g["avg_counts"])
    np.testing.assert_array_equal(matrix, g["avg_matrix"])
    assert nan_flag[0] == 0
    gy = np.ascontiguousarray(g["gy"])
    gv = np.full((n, C), 9.0, np.float32)
    assert lib.mf_average_voxelization_3d_bwd(gy.ctypes.data, points.ctypes.data, bi.ctypes.data, counts.ctypes.data,
                                              n, C, B, D, D, D, *o, pitch, gv.ctypes.data, None) == 0
    np.testing.assert_array_equal(gv, g["avg_gvalues"])
    inten = np.ascontiguousarray(g["intensities"])
    mm = np.full((B, C, D, D, D), 9.0, np.float32)
    ind = np.zeros((B, D, D, D), np.int32)
    key = np.zeros(B * D ** 3, np.uint64)
    assert lib.mf_max_voxelization_3d_fwd(values.ctypes.data, points.ctypes.data, bi.ctypes.data, inten.ctypes.data, n, C,
                                          B, D, D, D, *o, pitch, mm.ctypes.data, ind.ctypes.data, key.ctypes.data,
                                          nan_flag.ctypes.data, None) == 0
    np.testing.assert_array_equal(ind, g["max_indices"])
    np.testing.assert_array_equal(mm, g["max_matrix"])
    mgv = np.zeros((n, C), np.float32)
    assert lib.mf_max_voxelization_3d_bwd(gy.ctypes.data, ind.ctypes.data, n, C, B, D, D, D, mgv.ctypes.data, None) == 0
    np.testing.assert_allclose(mgv, g["max_gvalues"], rtol=1e-6, atol=1e-7)


@pytest.mark.parametrize("tag", ["d32_t2", "d32_t3", "d8x12x10_t2"])
def test_tdf_kernel_sources_vs_reference_cuda_text(tag):
    """tdf.hip (two-pass LDS (min, arg-min) tiles) on the host emulator against the reference's K7 / K8
    text: distances and winner indices bit-exact, backward to float-atomic tolerance."""
    from conftest import golden
    lib = emul.build(["tdf.hip"])
    i64, f, ci = ctypes.c_int64, ctypes.c_float, ctypes.c_int
    lib.mf_truncated_distance_function_fwd.argtypes = [_p, i64, f, f, f, f, ci, ci, ci, f, _p, _p, _p]
    lib.mf_truncated_distance_function_bwd.argtypes = [_p, _p, _p, i64, f, f, f, f, ci, ci, ci, f, _p, _p]
    g = golden("ref_cuda_tdf.npz")
    c = {k.split("__", 1)[1]: g[k] for k in g if k.startswith(tag + "__")}
    X, Y, Z = (int(v) for v in c["dims"])
    pts = np.ascontiguousarray(c["points"])
    o = [float(v) for v in c["origin"]]
    tdf = np.zeros((X, Y, Z), np.float32)
    flat = np.zeros((X, Y, Z), np.int32)
    assert lib.mf_truncated_distance_function_fwd(pts.ctypes.data, len(pts), float(c["pitch"]), *o, X, Y, Z,
                                                  float(c["truncation"]), tdf.ctypes.data, flat.ctypes.data, None) == 0
    np.testing.assert_array_equal(tdf, c["matrix"])
    np.testing.assert_array_equal(flat, c["indices"])
    gm = np.ascontiguousarray(c["gmatrix"])
    gp = np.zeros_like(pts)
    assert lib.mf_truncated_distance_function_bwd(gm.ctypes.data, pts.ctypes.data, flat.ctypes.data, len(pts),
                                                  float(c["pitch"]), *o, X, Y, Z, float(c["truncation"]),
                                                  gp.ctypes.data, None) == 0
    np.testing.assert_allclose(gp, c["gpoints"], rtol=2e-5, atol=2e-6)


def test_tdf_kernel_with_more_points_than_one_register_chunk():
    """tdf.hip keeps 16 points per lane in registers: above 4096 points it walks the list in chunks, once per pass.
    6000 points (two chunks, the second ragged) incl. NaN rows, exact duplicates (arg-min ties -> lowest id) and
    points outside the grid, vs the C oracle: distances and winner ids bit-exact."""
    from oracle import oracle_c as OC
    lib = emul.build(["tdf.hip"])
    i64, f, ci = ctypes.c_int64, ctypes.c_float, ctypes.c_int
    lib.mf_truncated_distance_function_fwd.argtypes = [_p, i64, f, f, f, f, ci, ci, ci, f, _p, _p, _p]
    rs = np.random.RandomState(5)
    dims, pitch, origin = (16, 12, 20), 0.01, (-0.08, -0.06, -0.1)
    pts = rs.uniform(-0.1, 0.12, (6000, 3)).astype(np.float32)
    pts[100] = np.nan
    pts[4500:4600] = pts[200:300]          # duplicates across the chunk boundary: the lower id must win
    trunc = float(np.float32(2) * np.float32(pitch))
    tdf = np.zeros(dims, np.float32)
    flat = np.zeros(dims, np.int32)
    assert lib.mf_truncated_distance_function_fwd(pts.ctypes.data, len(pts), pitch, *origin, *dims, trunc,
                                                  tdf.ctypes.data, flat.ctypes.data, None) == 0
    want, want_idx = OC.truncated_distance_function(pts, pitch=pitch, origin=origin, dims=dims, truncation=trunc)
    np.testing.assert_array_equal(tdf, want)
    np.testing.assert_array_equal(flat, want_idx)
    assert (flat >= 0).sum() > 1000


def test_nn_and_interpolate_kernel_sources_vs_reference_cuda_text():
    """k_nn (occgrid_knn.hip) against the reference's RawKernel + argmin, and interp.hip forward /
    backward against its K5 / K6 text (tests/golden/ref_cuda_nn.npz, ref_cuda_interpolate.npz)."""
    from conftest import golden
    i64, ci = ctypes.c_int64, ctypes.c_int
    lib = emul.build(["occgrid_knn.hip"])
    lib.mf_nn.argtypes = [_p, i64, _p, i64, _p, _p, _p]
    g = golden("ref_cuda_nn.npz")
    ref, query = np.ascontiguousarray(g["ref"]), np.ascontiguousarray(g["query"])
    out = np.zeros(len(query), np.int64)
    assert lib.mf_nn(ref.ctypes.data, len(ref), query.ctypes.data, len(query), out.ctypes.data, None, None) == 0
    np.testing.assert_array_equal(out, g["indices"])
    lib = emul.build(["interp.hip"])
    lib.mf_interpolate_voxel_grid_fwd.argtypes = [_p, _p, _p, _p, i64] + [ci] * 5 + [_p, ci, _p]
    lib.mf_interpolate_voxel_grid_bwd.argtypes = [_p, _p, _p, _p, i64] + [ci] * 5 + [_p, ci, _p]
    g = golden("ref_cuda_interpolate.npz")
    vox, pts, bi = (np.ascontiguousarray(g[k]) for k in ("voxelized", "points", "batch_indices"))
    B, C, X = vox.shape[0], vox.shape[1], vox.shape[2]
    n = len(pts)
    vals = np.full((n, C), 3.0, np.float32)
    assert lib.mf_interpolate_voxel_grid_fwd(vox.ctypes.data, pts.ctypes.data, bi.ctypes.data, None, n, B, C, X, X, X,
                                             vals.ctypes.data, 0, None) == 0
    np.testing.assert_array_equal(vals, g["values"])
    gv = np.ascontiguousarray(g["gvalues"])
    gvox = np.full(vox.shape, 3.0, np.float32)
    assert lib.mf_interpolate_voxel_grid_bwd(gv.ctypes.data, pts.ctypes.data, bi.ctypes.data, None, n, B, C, X, X, X,
                                             gvox.ctypes.data, 0, None) == 0
    np.testing.assert_allclose(gvox, g["gvoxelized"], rtol=1e-5, atol=1e-6)


def test_occupancy_grid_and_loss_kernel_sources_vs_reference_outputs():
    """occupancy_grid_3d kernel text against the reference's known-answer grid and BASELINE config 1
    (tests/golden/ref_occupancy_grid_3d.npz, produced by the reference's NumPy code), and the fused
    ADD / ADD-S loss kernels (loss.hip) against the reference's average_distance executed
    (ref_cuda_average_distance.npz: values + gradient to the predicted transforms)."""
    from conftest import golden
    i64, f, ci = ctypes.c_int64, ctypes.c_float, ctypes.c_int
    lib = emul.build(["occgrid_knn.hip"])
    lib.mf_occupancy_grid_3d_fwd.argtypes = [_p, i64, f, f, f, f, ci, ci, ci, f, _p, _p, _p]
    g = golden("ref_occupancy_grid_3d.npz")
    pts = np.ascontiguousarray(g["known_points"])
    grid, dmin = np.zeros((5, 5, 5), np.float32), np.zeros((5, 5, 5), np.float32)
    assert lib.mf_occupancy_grid_3d_fwd(pts.ctypes.data, len(pts), 1.0, 0.0, 0.0, 0.0, 5, 5, 5, 1.0, grid.ctypes.data,
                                        dmin.ctypes.data, None) == 0
    np.testing.assert_array_equal(grid, g["known_grid"])
    p = float(g["c1_pitch"])
    pts = np.ascontiguousarray(g["c1_points"][:200])
    grid, dmin = np.zeros((32,) * 3, np.float32), np.zeros((32,) * 3, np.float32)
    assert lib.mf_occupancy_grid_3d_fwd(pts.ctypes.data, len(pts), p, -16 * p, -16 * p, -16 * p, 32, 32, 32, 2.0,
                                        grid.ctypes.data, dmin.ctypes.data, None) == 0
    np.testing.assert_array_equal(grid, g["c1_grid_thr2"])

    lib = emul.build(["loss.hip"])
    lib.mf_average_distance_fwd.argtypes = [_p, _p, _p, _p, _i32, _i32, _i32, _p, _p, _p]
    lib.mf_average_distance_bwd.argtypes = [_p, _p, _p, _p, _p, _i32, _i32, _i32, _p, _p, _p]
    g = golden("ref_cuda_average_distance.npz")
    pts, Tt, Tp = (np.ascontiguousarray(g[k]) for k in ("points", "transform_true", "transforms_pred"))
    M, P = len(pts), len(Tp)
    for tag, sym in (("add", 0), ("adds", 1)):
        symm = np.array([sym], np.uint8)
        out = np.zeros(P, np.float32)
        idx = np.zeros((P, M), np.int32)
        assert lib.mf_average_distance_fwd(pts.ctypes.data, Tt.ctypes.data, Tp.ctypes.data, symm.ctypes.data, 1, M, P,
                                           out.ctypes.data, idx.ctypes.data, None) == 0
        np.testing.assert_allclose(out, g[f"{tag}_value"], rtol=5e-6, atol=1e-8)
        gout = np.ascontiguousarray(g[f"{tag}_gout"])
        gT = np.zeros((P, 4, 4), np.float32)
        assert lib.mf_average_distance_bwd(pts.ctypes.data, Tt.ctypes.data, Tp.ctypes.data, symm.ctypes.data,
                                           gout.ctypes.data, 1, M, P, idx.ctypes.data, gT.ctypes.data, None) == 0
        want = g[f"{tag}_gT"][:, :3, :]
        np.testing.assert_allclose(gT[:, :3, :], want, rtol=2e-3, atol=2e-4 * float(np.abs(want).max()))


def test_channels_last_sparse_conv3_and_sampler_give_the_channels_first_bits():
    """The channels-last entries of the inference path (sparse conv3 reduce -> [B,Vo,Cout]; trilinear sampler
    reading whole voxel rows and writing into a column block of a wider matrix) against the channels-first
    kernels they stand beside: bit-identical results (same taps / corners in the same order)."""
    lib = emul.build(["sparseconv.hip", "interp.hip"])
    i64, f = ctypes.c_int64, ctypes.c_float
    lib.mf_sparse_conv3d_workspace_bytes.restype = i64
    lib.mf_sparse_conv3d_workspace_bytes.argtypes = [_i32] * 5 + [i64]
    lib.mf_sparse_conv3d_pack_weights.argtypes = [_p, _i32, _i32, _i32, _i32, _p, _p]
    lib.mf_sparse_conv3d_k4s2_points_fwd.argtypes = [_p, _p, _p, i64] + [f] * 4 + [_p] * 5 + [_i32] * 6 + [_p]
    lib.mf_sparse_conv3d_k4s2_points_cl_fwd.argtypes = [_p, i64, _p, _p, i64] + [f] * 4 + [_p] * 5 + [_i32] * 6 + [_p]
    lib.mf_interpolate_voxel_grid_fwd.argtypes = [_p, _p, _p, _p, i64] + [ctypes.c_int] * 5 + [_p, ctypes.c_int, _p]
    lib.mf_interpolate_voxel_grid_cl_fwd.argtypes = [_p, _p, _p, i64] + [ctypes.c_int] * 5 + [_p, i64, _p]
    rs = np.random.RandomState(11)
    B, Cs, Cout, D, n, ld = 2, 8, 256, 8, 90, 20
    points = rs.uniform(-0.6, D - 0.4, (n, 3)).astype(np.float32)
    points[:10] = points[10:20]
    wide = rs.uniform(-1, 1, (n, ld)).astype(np.float32)          # values = columns 4..12 of a wider matrix
    values = np.ascontiguousarray(wide[:, 4:4 + Cs])
    bi = np.sort(rs.randint(0, B, n)).astype(np.int32)
    W = (rs.uniform(-1, 1, (Cout, Cs, 4, 4, 4)) * 0.2).astype(np.float32)
    bias = rs.uniform(-0.1, 0.1, Cout).astype(np.float32)
    Do = D // 2
    dense_cf = rs.uniform(-0.1, 0.1, (B, Cout, Do, Do, Do)).astype(np.float32)
    dense_cl = np.ascontiguousarray(dense_cf.transpose(0, 2, 3, 4, 1))
    Wp = np.zeros(8 * Cs * 8 * Cout, np.float32)
    assert lib.mf_sparse_conv3d_pack_weights(W.ctypes.data, Cout, Cs, Cs, 0, Wp.ctypes.data, None) == 0
    ws = np.zeros(int(lib.mf_sparse_conv3d_workspace_bytes(B, Cs, Cout, D, n, n)) // 4 + 64, np.float32)
    out_cf = np.full(dense_cf.shape, 7.0, np.float32)
    assert lib.mf_sparse_conv3d_k4s2_points_fwd(values.ctypes.data, points.ctypes.data, bi.ctypes.data, n, 0.0, 0.0,
                                                0.0, 1.0, Wp.ctypes.data, dense_cf.ctypes.data, bias.ctypes.data,
                                                out_cf.ctypes.data, ws.ctypes.data, B, Cs, Cout, D, n, 1, None) == 0
    out_cl = np.full(dense_cl.shape, 7.0, np.float32)
    ws[:] = 0
    assert lib.mf_sparse_conv3d_k4s2_points_cl_fwd(wide[:, 4:].ctypes.data, ld, points.ctypes.data, bi.ctypes.data, n,
                                                   0.0, 0.0, 0.0, 1.0, Wp.ctypes.data, dense_cl.ctypes.data,
                                                   bias.ctypes.data, out_cl.ctypes.data, ws.ctypes.data, B, Cs, Cout,
                                                   D, n, 1, None) == 0
    np.testing.assert_array_equal(out_cl.transpose(0, 4, 1, 2, 3), out_cf)
    assert (out_cf > 0).mean() > 0.2

    # sampler: points in the half-resolution frame, incl. out-of-grid corners and an invalid batch index
    pts = (points / 2.0).astype(np.float32)
    pts[3] = (-0.5, 1.0, 1.0)
    pts[4] = (Do - 0.5, Do - 0.2, 0.3)
    bi2 = bi.copy()
    bi2[5] = 9
    ref = np.zeros((n, Cout), np.float32)
    assert lib.mf_interpolate_voxel_grid_fwd(out_cf.ctypes.data, pts.ctypes.data, bi2.ctypes.data, None, n, B, Cout,
                                             Do, Do, Do, ref.ctypes.data, 0, None) == 0
    ldo = Cout + 24
    got = np.full((n, ldo), 5.0, np.float32)
    assert lib.mf_interpolate_voxel_grid_cl_fwd(out_cl.ctypes.data, pts.ctypes.data, bi2.ctypes.data, n, B, Cout, Do,
                                                Do, Do, got[:, 8:].ctypes.data, ldo, None) == 0
    np.testing.assert_array_equal(got[:, 8:8 + Cout], ref)
    assert (got[:, :8] == 5.0).all() and (got[:, 8 + Cout:] == 5.0).all()   # neighbours untouched
    assert (ref[5] == 0).all() and np.abs(ref).sum() > 0
