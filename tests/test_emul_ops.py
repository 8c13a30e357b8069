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
