"""csrc/conv3d.hip (fp32-MFMA implicit GEMM, kernel 4 / stride 2 / pad 1, channels-last) run on the CPU
behind the fiber emulator (tests/host_emul) and checked against torch's dense Conv3d -- the operator the
reference applies there (contrib/singleview_3d/models/model.py:73-74,128,139: Convolution3D(.., 4, 2, pad=1)).
fp32 sums in a different order: tolerance 2e-5 of the largest output."""
import ctypes

import numpy as np
import pytest
import torch

from host_emul import emul

pytestmark = pytest.mark.skipif(not emul.available(), reason="g++ not available")


@pytest.fixture(scope="module")
def lib():
    L = emul.build(["conv3d.hip"])
    i32, p = ctypes.c_int32, ctypes.c_void_p
    L.mf_conv3d_k4s2_pack_weights.argtypes = [p, i32, i32, i32, i32, p, p]
    L.mf_conv3d_k4s2_workspace_bytes.restype = ctypes.c_int64
    L.mf_conv3d_k4s2_workspace_bytes.argtypes = [i32] * 4
    L.mf_conv3d_k4s2_default_split.argtypes = [i32] * 4
    L.mf_conv3d_k4s2_fwd.argtypes = [p] * 6 + [i32] * 6 + [p]
    L.mf_to_channels_last.argtypes = [p, p, i32, i32, ctypes.c_int64, p]
    return L


def _run(L, x_cf, W, bias, add_cl, split, relu, c_off=0, cin=None):
    """x_cf [B,Cin_total,D,D,D] channels-first (torch layout); returns out channels-first."""
    B, _, D = x_cf.shape[:3]
    Cout, w_cin = W.shape[:2]
    cin = cin or w_cin
    # (buffers end at an inaccessible page: an out-of-bounds tap / weight / store read faults here too)
    x_cl = emul.guarded(np.ascontiguousarray(x_cf[:, c_off:c_off + cin].transpose(0, 2, 3, 4, 1)))
    wt = emul.guarded(np.zeros((Cout, 64, cin), np.float32))
    assert L.mf_conv3d_k4s2_pack_weights(emul.ptr(np.ascontiguousarray(W)), Cout, cin, w_cin, c_off, emul.ptr(wt), None) == 0
    np.testing.assert_array_equal(wt, W[:, c_off:c_off + cin].reshape(Cout, cin, 64).transpose(0, 2, 1))
    Do = D // 2
    out = emul.guarded(np.full((B, Do, Do, Do, Cout), np.nan, np.float32))
    nbytes = L.mf_conv3d_k4s2_workspace_bytes(B, Cout, D, split)
    ws = np.zeros(max(nbytes, 4) // 4, np.float32)
    rc = L.mf_conv3d_k4s2_fwd(emul.ptr(x_cl), emul.ptr(wt), emul.ptr(bias), emul.ptr(add_cl), emul.ptr(out),
                              emul.ptr(ws), B, cin, Cout, D, split, int(relu), None)
    assert rc == 0
    return out.transpose(0, 4, 1, 2, 3)


@pytest.mark.parametrize("B,Cin,Cout,D,split,relu", [
    (2, 32, 128, 8, 1, True),     # M = 128: one full tile, every tap, padding on all six faces
    (1, 32, 128, 8, 4, True),     # M = 64 < tile (row guard) + split-K slabs + finish kernel
    (3, 16, 128, 4, 2, False),    # Cin = 16: two taps per K-tile; M = 24; no activation
    (1, 64, 256, 6, 8, True),     # two N tiles, odd output extent (Do = 3)
])
def test_conv3d_k4s2_matches_torch(lib, B, Cin, Cout, D, split, relu):
    rs = np.random.RandomState(B * 100 + Cin)
    x = rs.uniform(-1, 1, (B, Cin, D, D, D)).astype(np.float32)
    x[rs.uniform(size=x.shape) < 0.3] = 0.0   # post-ReLU-like input
    W = (rs.uniform(-1, 1, (Cout, Cin, 4, 4, 4)) / np.sqrt(Cin * 64)).astype(np.float32)
    bias = rs.uniform(-0.2, 0.2, Cout).astype(np.float32)
    got = _run(lib, x, W, bias, None, split, relu)
    ref = torch.nn.functional.conv3d(torch.from_numpy(x).double(), torch.from_numpy(W).double(),
                                     torch.from_numpy(bias).double(), stride=2, padding=1)
    if relu:
        ref = torch.relu(ref)
    ref = ref.numpy()
    assert np.isfinite(got).all()
    assert np.abs(got - ref).max() < 2e-5 * max(1.0, np.abs(ref).max())


def test_conv3d_channel_slice_and_addend(lib):
    """The conv3 use: convolve only channels [c_off, c_off + 16) of a wider weight tensor and add the
    result of the sparse part (``add``, channels-last) before the activation."""
    rs = np.random.RandomState(7)
    B, D, Cout, w_cin, c_off, cin = 1, 8, 128, 24, 8, 16
    x = rs.uniform(0, 1, (B, w_cin, D, D, D)).astype(np.float32)
    W = (rs.uniform(-1, 1, (Cout, w_cin, 4, 4, 4)) / 30).astype(np.float32)
    bias = rs.uniform(-0.2, 0.2, Cout).astype(np.float32)
    add = rs.uniform(-1, 1, (B, D // 2, D // 2, D // 2, Cout)).astype(np.float32)
    for split in (1, 2):
        got = _run(lib, x, W, bias, add, split, True, c_off=c_off, cin=cin)
        ref = torch.nn.functional.conv3d(torch.from_numpy(x[:, c_off:c_off + cin]).double(),
                                         torch.from_numpy(W[:, c_off:c_off + cin]).double(),
                                         torch.from_numpy(bias).double(), stride=2, padding=1)
        ref = torch.relu(ref + torch.from_numpy(add).double().permute(0, 4, 1, 2, 3)).numpy()
        assert np.abs(got - ref).max() < 2e-5 * max(1.0, np.abs(ref).max())


def test_conv3d_rejects_bad_shapes(lib):
    z = np.zeros(4, np.float32)
    for B, Cin, Cout, D, split in [(1, 24, 128, 8, 1), (1, 32, 96, 8, 1), (1, 32, 128, 7, 1), (1, 32, 128, 8, 3)]:
        assert lib.mf_conv3d_k4s2_fwd(emul.ptr(z), emul.ptr(z), None, None, emul.ptr(z), emul.ptr(z),
                                      B, Cin, Cout, D, split, 1, None) != 0


def test_to_channels_last(lib):
    rs = np.random.RandomState(0)
    x = rs.uniform(-1, 1, (2, 19, 45)).astype(np.float32)
    y = np.zeros((2, 45, 19), np.float32)
    assert lib.mf_to_channels_last(emul.ptr(x), emul.ptr(y), 2, 19, 45, None) == 0
    np.testing.assert_array_equal(y, x.transpose(0, 2, 1))


def test_occupancy_convs_match_torch(lib):
    """conv1_occ (1 -> 8, k3 p1) + ReLU + conv2_occ (8 -> 16, k3 dilation 2 p2) + ReLU, channels-last out."""
    i32, p = ctypes.c_int32, ctypes.c_void_p
    lib.mf_occupancy_convs_fwd.argtypes = [p] * 7 + [i32] * 2 + [p]
    rs = np.random.RandomState(3)
    B, D = 2, 6
    grid = (rs.uniform(size=(B, D, D, D)) < 0.4).astype(np.float32)
    W1 = rs.uniform(-0.5, 0.5, (8, 1, 3, 3, 3)).astype(np.float32)
    b1 = rs.uniform(-0.1, 0.1, 8).astype(np.float32)
    W2 = rs.uniform(-0.3, 0.3, (16, 8, 3, 3, 3)).astype(np.float32)
    b2 = rs.uniform(-0.1, 0.1, 16).astype(np.float32)
    w1 = np.ascontiguousarray(W1.transpose(2, 3, 4, 1, 0)).reshape(27, 1, 8)
    w2 = np.ascontiguousarray(W2.transpose(2, 3, 4, 1, 0)).reshape(27, 8, 16)
    h1 = np.zeros((B, D ** 3, 8), np.float32)
    h2 = np.zeros((B, D ** 3, 16), np.float32)
    assert lib.mf_occupancy_convs_fwd(emul.ptr(grid), emul.ptr(w1), emul.ptr(b1), emul.ptr(w2), emul.ptr(b2),
                                      emul.ptr(h1), emul.ptr(h2), B, D, None) == 0
    F = torch.nn.functional
    t1 = torch.relu(F.conv3d(torch.from_numpy(grid)[:, None].double(), torch.from_numpy(W1).double(),
                             torch.from_numpy(b1).double(), padding=1))
    t2 = torch.relu(F.conv3d(t1, torch.from_numpy(W2).double(), torch.from_numpy(b2).double(), padding=2, dilation=2))
    np.testing.assert_allclose(h1.reshape(B, D, D, D, 8).transpose(0, 4, 1, 2, 3), t1.numpy(), atol=2e-6)
    np.testing.assert_allclose(h2.reshape(B, D, D, D, 16).transpose(0, 4, 1, 2, 3), t2.numpy(), atol=5e-6)
    assert (h2 > 0).mean() > 0.2


def test_linear_mfma_gemm_vs_float64():
    """csrc/linear.hip: grouped row-major GEMM + bias + ReLU (the heads' 1x1 convolutions on points-major rows):
    K not a multiple of 32, M not a multiple of 128, N < Npad, column-block inputs / outputs, two groups."""
    L = emul.build(["linear.hip"])
    i32, i64, p = ctypes.c_int32, ctypes.c_int64, ctypes.c_void_p
    L.mf_linear_fwd.argtypes = [p, i64, i32, p, i64, i32, p, i64, p, i64] + [i32] * 7 + [p]
    rs = np.random.RandomState(4)
    M, K, N, Npad, groups = 150, 72, 84, 128, 2
    lda, ldo = 200, 300
    A = rs.uniform(-1, 1, (M, lda)).astype(np.float32)      # group g reads columns [8 + 80 g, 8 + 80 g + K)
    W = np.zeros((groups, Npad, K), np.float32)
    W[:, :N] = rs.uniform(-1, 1, (groups, N, K)) / np.sqrt(K)
    bias = rs.uniform(-0.3, 0.3, (groups, N)).astype(np.float32)
    out = np.full((M, ldo), 9.0, np.float32)                 # group g writes columns [4 + 100 g, 4 + 100 g + N)
    for relu in (1, 0):
        out[:] = 9.0
        assert L.mf_linear_fwd(A[:, 8:].ctypes.data, 80, lda, emul.ptr(W), Npad * K, K, emul.ptr(bias), N,
                               out[:, 4:].ctypes.data, 100, ldo, M, N, Npad, K, groups, relu, None) == 0
        for g in range(groups):
            ref = A[:, 8 + 80 * g:8 + 80 * g + K].astype(np.float64) @ W[g, :N].astype(np.float64).T + bias[g]
            if relu:
                ref = np.maximum(ref, 0)
            np.testing.assert_allclose(out[:, 4 + 100 * g:4 + 100 * g + N], ref, rtol=0, atol=2e-6)
        assert (out[:, :4] == 9.0).all() and (out[:, 4 + N:104] == 9.0).all() and (out[:, 104 + N:] == 9.0).all()
    # an output block that is not 16-byte aligned takes the scalar-store path of the epilogue
    out[:] = 9.0
    assert L.mf_linear_fwd(A[:, 8:].ctypes.data, 80, lda, emul.ptr(W), Npad * K, K, emul.ptr(bias), N,
                           out[:, 5:].ctypes.data, 100, ldo, M, N, Npad, K, groups, 1, None) == 0
    for g in range(groups):
        ref = np.maximum(A[:, 8 + 80 * g:8 + 80 * g + K].astype(np.float64) @ W[g, :N].astype(np.float64).T + bias[g], 0)
        np.testing.assert_allclose(out[:, 5 + 100 * g:5 + 100 * g + N], ref, rtol=0, atol=2e-6)
    assert (out[:, :5] == 9.0).all() and (out[:, 5 + N:105] == 9.0).all()
    # full-height (128 x 128) tiles: taken when >= 256 of them exist (the heads' first layer at batch 8)
    Mb, Nb, Kb = 2200, 1920, 40
    Ab = emul.guarded(rs.uniform(-1, 1, (Mb, Kb)).astype(np.float32))
    Wb = emul.guarded((rs.uniform(-1, 1, (Nb, Kb)) / 6).astype(np.float32))
    bb = rs.uniform(-1, 1, Nb).astype(np.float32)
    ob = emul.guarded(np.zeros((Mb, Nb), np.float32))
    assert L.mf_linear_fwd(Ab.ctypes.data, 0, Kb, Wb.ctypes.data, 0, Kb, emul.ptr(bb), 0, ob.ctypes.data, 0, Nb,
                           Mb, Nb, Nb, Kb, 1, 1, None) == 0
    np.testing.assert_allclose(ob, np.maximum(Ab.astype(np.float64) @ Wb.astype(np.float64).T + bb, 0), atol=2e-6)
    # K = 4 (the padded conv1_pcd): chunks 1..7 of the only K-tile lie past K -- they must not be dereferenced
    # beyond the matrices (buffers end at an inaccessible page)
    A4 = emul.guarded(rs.uniform(-1, 1, (70, 4)).astype(np.float32))
    W4 = emul.guarded(np.concatenate([rs.uniform(-1, 1, (8, 4)), np.zeros((120, 4))]).astype(np.float32))
    b4 = emul.guarded(rs.uniform(-1, 1, 8).astype(np.float32))
    o4 = emul.guarded(np.zeros((70, 8), np.float32))
    assert L.mf_linear_fwd(A4.ctypes.data, 0, 4, W4.ctypes.data, 0, 4, b4.ctypes.data, 0, o4.ctypes.data, 0, 8,
                           70, 8, 128, 4, 1, 1, None) == 0
    np.testing.assert_allclose(o4, np.maximum(A4.astype(np.float64) @ W4[:8].astype(np.float64).T + b4, 0), atol=1e-6)
    z = np.zeros(8, np.float32)
    assert L.mf_linear_fwd(emul.ptr(z), 0, 8, emul.ptr(z), 0, 8, None, 0, emul.ptr(z), 0, 8, 1, 1, 100, 8, 1, 0, None) != 0


def test_point_prep_and_pose_epilogue_vs_numpy():
    """csrc/pointops.hip: the point-wise prologue (camera -> voxel frame, to_center, feature rows, batch indices) and
    epilogue (class selection, chainer's normalize, translation, sigmoid) against the torch expressions of
    contrib/singleview_3d/models/model.py:236,101,262-273 restated in NumPy float32."""
    L = emul.build(["pointops.hip"])
    i32, i64, p, f = ctypes.c_int32, ctypes.c_int64, ctypes.c_void_p, ctypes.c_float
    L.mf_point_prep.argtypes = [p, p, p, p, i32, i32, i32, f, p, p, p, p, p]
    L.mf_pose_epilogue.argtypes = [p, i64, i32, p, p, p, p, i32, i32, i32, p, p, p, p]
    rs = np.random.RandomState(9)
    B, P, Cv = 3, 70, 8
    n = B * P
    pc = rs.uniform(-0.3, 0.9, (B, 3, P)).astype(np.float32)
    vals = rs.uniform(-1, 1, (B, Cv, P)).astype(np.float32)
    origin = rs.uniform(-0.4, 0.2, (B, 3)).astype(np.float32)
    pitch = rs.uniform(0.004, 0.01, B).astype(np.float32)
    pts, tc4 = emul.guarded(np.zeros((n, 3), np.float32)), emul.guarded(np.zeros((n, 4), np.float32))
    xr, bi = emul.guarded(np.zeros((n, Cv), np.float32)), emul.guarded(np.zeros(n, np.int32))
    assert L.mf_point_prep(emul.ptr(pc), emul.ptr(vals), emul.ptr(origin), emul.ptr(pitch), B, P, Cv, 15.5,
                           pts.ctypes.data, tc4.ctypes.data, xr.ctypes.data, bi.ctypes.data, None) == 0
    pv = ((pc - origin[:, :, None]) / pitch[:, None, None]).astype(np.float32)           # model.py:236
    np.testing.assert_array_equal(pts, pv.transpose(0, 2, 1).reshape(n, 3))
    np.testing.assert_array_equal(tc4[:, :3], (np.float32(15.5) - pts).astype(np.float32))
    assert (tc4[:, 3] == 0).all()
    np.testing.assert_array_equal(xr, vals.transpose(0, 2, 1).reshape(n, Cv))
    np.testing.assert_array_equal(bi, np.repeat(np.arange(B), P))

    nf, np4 = 21, 128
    o = emul.guarded(rs.uniform(-2, 2, (n, 3 * np4)).astype(np.float32))
    cid = np.array([1, 21, 7], np.int64)
    rot, trans, conf = (emul.guarded(np.zeros((n, k), np.float32)) for k in (4, 3, 1))
    assert L.mf_pose_epilogue(o.ctypes.data, 3 * np4, np4, emul.ptr(cid), pts.ctypes.data, emul.ptr(origin),
                              emul.ptr(pitch), B, P, nf, rot.ctypes.data, trans.ctypes.data, conf.ctypes.data, None) == 0
    # a background / out-of-range class id gives NaN poses and never reads outside the heads' row
    bad_cid = np.array([0, 22, 7], np.int64)
    r2, t2, c2 = (emul.guarded(np.zeros((n, k), np.float32)) for k in (4, 3, 1))
    assert L.mf_pose_epilogue(o.ctypes.data, 3 * np4, np4, emul.ptr(bad_cid), pts.ctypes.data, emul.ptr(origin),
                              emul.ptr(pitch), B, P, nf, r2.ctypes.data, t2.ctypes.data, c2.ctypes.data, None) == 0
    assert np.isnan(r2[:2 * P]).all() and np.isnan(t2[:2 * P]).all() and np.isnan(c2[:2 * P]).all()
    np.testing.assert_array_equal(r2[2 * P:], rot[2 * P:])
    fg = np.repeat(cid - 1, P)
    rows = np.arange(n)
    q = np.stack([o[rows, 4 * fg + k] for k in range(4)], 1)
    np.testing.assert_allclose(rot, q / (np.linalg.norm(q.astype(np.float64), axis=1, keepdims=True) + 1e-5), atol=2e-7)
    pb = np.repeat(pitch, P)[:, None]
    t = np.stack([o[rows, np4 + 3 * fg + k] for k in range(3)], 1)
    np.testing.assert_allclose(trans, (pts * pb + np.repeat(origin, P, 0)) + t * pb, atol=1e-7)
    np.testing.assert_allclose(conf[:, 0], 1 / (1 + np.exp(-o[rows, 2 * np4 + fg].astype(np.float64))), atol=2e-7)


def test_psp_tail_kernel_vs_torch_formulation():
    """csrc/psp_tail.hip (the last PSPNet level at the sampled pixels, one launch) against the torch formulation
    it replaces (PSPNetExtractor._tail over _tail_taps: gathers + einsum + conv1d + log_softmax), which
    tests/test_host_logic.py pins against the dense decoder of models/dense_fusion/pspnet.py.  NCHW and
    channels-last source maps, image-border pixels (zero padding of the 3x3 convolution) included."""
    from morefusion_amd.models.backbone2d import PSPNetExtractor
    L = emul.build(["psp_tail.hip"])
    i32, i64, p = ctypes.c_int32, ctypes.c_int64, ctypes.c_void_p
    L.mf_psp_tail_fwd.argtypes = [p, i64, i64, i64, i64, p, p, p, p, p, p] + [i32] * 4 + [p, p]
    torch.manual_seed(0)
    net = PSPNetExtractor().eval()
    with torch.no_grad():
        net.up3.prelu.weight.fill_(0.2)
    B, H, W, P = 2, 12, 10, 37
    u2 = torch.randn(B, 64, H, W)
    Ho, Wo = 2 * H, 2 * W
    rs = np.random.RandomState(1)
    pix = rs.randint(0, Ho * Wo, (B, P))
    pix[0, :6] = [0, Wo - 1, (Ho - 1) * Wo, Ho * Wo - 1, Wo, 2 * Wo - 1]   # corners and edges
    pix = torch.from_numpy(pix)
    with torch.no_grad():
        ref = net._tail(u2, net._tail_taps(pix, H, W)).transpose(1, 2).reshape(B * P, 32).numpy()
    w3t = net.up3.conv.weight.detach().permute(2, 3, 1, 0).reshape(9, 64, 64).contiguous().numpy()
    w1t = net.conv1.weight.detach().reshape(32, 64).t().contiguous().numpy()
    b3, b1 = net.up3.conv.bias.detach().numpy(), net.conv1.bias.detach().numpy()
    slope = net.up3.prelu.weight.detach().numpy()
    pixc = np.ascontiguousarray(pix.numpy().reshape(-1).astype(np.int64))
    for fmt in (torch.contiguous_format, torch.channels_last):
        x = u2.contiguous(memory_format=fmt)
        out = emul.guarded(np.zeros((B * P, 32), np.float32))
        assert L.mf_psp_tail_fwd(x.data_ptr(), x.stride(0), x.stride(1), x.stride(2), x.stride(3), emul.ptr(pixc),
                                 emul.ptr(w3t), emul.ptr(b3), emul.ptr(slope), emul.ptr(w1t), emul.ptr(b1), B, P, H, W,
                                 out.ctypes.data, None) == 0
        np.testing.assert_allclose(out, ref, rtol=0, atol=2e-5)
    assert np.abs(np.exp(ref).sum(1) - 1).max() < 1e-5   # rows are log-probabilities
