"""bf16_ops.py (the autograd operators over csrc/gemm_bf16.hip) on the CPU: the emulated library with torch CPU
tensors as device memory, against torch's own float32 autograd on the same bf16-rounded operands
(model.py:73-74,125-139 conv3 / conv4; :76-91,239-258 the 1x1 convolution chains)."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from host_emul import emul

pytestmark = pytest.mark.skipif(not emul.available(), reason="g++ not available")


@pytest.fixture()
def K(monkeypatch):
    from morefusion_amd import _lib
    from morefusion_amd.contrib.singleview_3d.models import bf16_ops
    L = emul.build(["gemm_bf16.hip"])
    for name, (argtypes, restype) in _lib._SIGNATURES.items():
        fn = getattr(L, name, None)
        if fn is not None:
            fn.argtypes, fn.restype = argtypes, restype
    monkeypatch.setattr(_lib, "lib", lambda: L)
    monkeypatch.setattr(_lib, "require_gpu", lambda *a: None)
    monkeypatch.setattr(_lib, "stream_ptr", lambda: None)
    monkeypatch.setattr(_lib, "check", lambda code, what: (_ for _ in ()).throw(RuntimeError(what)) if code else None)
    return bf16_ops


def rel(a, b):
    a, b = a.detach(), b.detach()
    return float((a.float() - b.float()).abs().max() / (b.float().abs().max() + 1e-30))


def test_conv3d_operator_forward_and_gradients(K):
    torch.manual_seed(0)
    B, Cin, Cout, D, w_cin, c_off = 1, 16, 64, 16, 24, 8
    conv = torch.nn.Conv3d(w_cin, Cout, 4, 2, padding=1)
    x = torch.randn(B, D ** 3, Cin).to(torch.bfloat16).requires_grad_(True)
    out = K.conv3d_k4s2(x, conv, D, relu=True, c_off=c_off)
    g = torch.randn_like(out.float()).to(torch.bfloat16)
    out.backward(g)
    # reference: float32 conv of the bf16-rounded operands, output rounded to bf16 before the ReLU mask is taken
    xr = x.detach().float().reshape(B, D, D, D, Cin).permute(0, 4, 1, 2, 3).requires_grad_(True)
    wr = conv.weight.detach().to(torch.bfloat16).float()[:, c_off:c_off + Cin].requires_grad_(True)
    br = conv.bias.detach().clone().requires_grad_(True)
    y = F.relu(F.conv3d(xr, wr, br, stride=2, padding=1))
    y.backward(g.float().reshape(B, D // 2, D // 2, D // 2, Cout).permute(0, 4, 1, 2, 3))
    y_cl = y.detach().permute(0, 2, 3, 4, 1).reshape(B, -1, Cout)
    assert rel(out, y_cl) < 2 ** -7
    assert rel(x.grad, xr.grad.permute(0, 2, 3, 4, 1).reshape(B, -1, Cin)) < 2e-2   # (bf16 output + mask flips at 0)
    assert rel(conv.weight.grad[:, c_off:c_off + Cin], wr.grad) < 2e-2
    assert float(conv.weight.grad[:, :c_off].abs().max()) == 0.0
    assert rel(conv.bias.grad, br.grad) < 2e-2


@pytest.mark.parametrize("narrow", [True, False], ids=["narrow_kernel", "gemm_engine"])
def test_occupancy_branch_operator_chain(K, monkeypatch, narrow):
    """conv1_occ -> conv2_occ (model.py:69-72,120-124) through the general-geometry operator: values and the
    gradients of both layers' parameters vs torch's float32 convolutions.  ``narrow_kernel`` (the default, round 6):
    both forward passes and conv2_occ's data gradient on k_conv_k3_narrow_bf16 (voxels as the MFMA's columns, operands
    straight from global memory) -- three launches, counted; ``gemm_engine`` (MF_NARROW_CONV=0): round 4's path
    through the implicit-GEMM engine."""
    monkeypatch.setenv("MF_NARROW_CONV", "1" if narrow else "0")
    Lh = K._lib.lib()
    calls = []
    real = Lh.mf_conv3d_k3_narrow_bf16
    monkeypatch.setattr(Lh, "mf_conv3d_k3_narrow_bf16", lambda *a: (calls.append(a[5:9]), real(*a))[1], raising=False)
    torch.manual_seed(2)
    B, D = 1, 8
    c1 = torch.nn.Conv3d(1, 8, 3, 1, padding=1)
    c2 = torch.nn.Conv3d(8, 16, 3, 1, padding=2, dilation=2)
    grid = (torch.rand(B, D, D, D) > 0.5).float()
    g8 = torch.zeros(B, D ** 3, 8, dtype=torch.bfloat16)
    g8[:, :, 0] = grid.reshape(B, -1)
    out = K.conv3d(K.conv3d(g8, c1, D), c2, D)
    g = torch.randn(out.shape).to(torch.bfloat16)
    out.backward(g)
    r1, r2 = torch.nn.Conv3d(1, 8, 3, 1, padding=1), torch.nn.Conv3d(8, 16, 3, 1, padding=2, dilation=2)
    with torch.no_grad():
        for a, b in ((r1, c1), (r2, c2)):
            a.weight.copy_(b.weight.to(torch.bfloat16).float())
            a.bias.copy_(b.bias)
    h1 = F.relu(r1(grid[:, None])).to(torch.bfloat16).float()   # (the operator hands bf16 activations on)
    y = F.relu(r2(h1))
    assert rel(out, y.permute(0, 2, 3, 4, 1).reshape(B, -1, 16)) < 2 ** -7
    # gradients of the second layer against autograd on the same (bf16-rounded) intermediate
    h1l = h1.detach().requires_grad_(True)
    y2 = F.relu(r2(h1l))
    y2.backward(g.float().reshape(B, D, D, D, 16).permute(0, 4, 1, 2, 3))
    assert rel(c2.weight.grad, r2.weight.grad) < 2e-2 and rel(c2.bias.grad, r2.bias.grad) < 2e-2
    assert c1.weight.grad is not None and float(c1.weight.grad.abs().sum()) > 0 and c1.weight.grad.shape == (8, 1, 3, 3, 3)
    # first layer: chain the reference's dh1 (masked by its own ReLU) through conv1's weight gradient
    dz1 = (h1l.grad * (h1 > 0)).to(torch.bfloat16).float()
    h1_pre = r1(grid[:, None])
    h1_pre.backward(dz1)
    assert rel(c1.weight.grad, r1.weight.grad) < 3e-2
    # (read channels, written channels, D, dilation) of the narrow kernel's launches: forward 8 -> 8, forward 8 -> 16
    # (dilation 2), data gradient 16 -> 8
    assert calls == ([(8, 8, D, 1), (8, 16, D, 2), (16, 8, D, 2)] if narrow else [])


@pytest.mark.parametrize("n,Kin,N,relu", [(150, 3, 8, True), (130, 64, 63, False), (200, 984, 640, True)])
def test_linear_operator_forward_and_gradients(K, n, Kin, N, relu):
    torch.manual_seed(1)
    conv = torch.nn.Conv1d(Kin, N, 1)
    wide = torch.randn(n, Kin + 16).to(torch.bfloat16) if Kin % 8 == 0 else torch.randn(n, Kin)
    x = wide[:, 8:8 + Kin] if Kin % 8 == 0 else wide          # a column block of a wider matrix
    x = x.detach().requires_grad_(True)
    out = K.linear(x, conv, relu=relu)
    g = torch.randn(n, N).to(torch.bfloat16)
    out.backward(g)
    xr = x.detach().to(torch.bfloat16).float().requires_grad_(True)
    wr = conv.weight.detach().reshape(N, Kin).to(torch.bfloat16).float().requires_grad_(True)
    br = conv.bias.detach().clone().requires_grad_(True)
    y = xr @ wr.t() + br
    y = F.relu(y) if relu else y
    y.backward(g.float())
    assert out.shape == (n, N) and out.dtype == torch.bfloat16
    assert rel(out, y.detach()) < 2 ** -7
    assert rel(x.grad, xr.grad) < 2e-2
    assert rel(conv.weight.grad.reshape(N, Kin), wr.grad) < 2e-2
    assert rel(conv.bias.grad, br.grad) < 2e-2


@pytest.fixture()
def KV(monkeypatch):
    """bf16_ops over the emulated voxelize / interp / gemm sources."""
    from morefusion_amd import _lib
    from morefusion_amd.contrib.singleview_3d.models import bf16_ops
    L = emul.build(["gemm_bf16.hip", "voxelize.hip", "interp.hip"])
    for name, (argtypes, restype) in _lib._SIGNATURES.items():
        fn = getattr(L, name, None)
        if fn is not None:
            fn.argtypes, fn.restype = argtypes, restype
    monkeypatch.setattr(_lib, "lib", lambda: L)
    monkeypatch.setattr(_lib, "require_gpu", lambda *a: None)
    monkeypatch.setattr(_lib, "stream_ptr", lambda: None)
    monkeypatch.setattr(_lib, "check", lambda code, what: (_ for _ in ()).throw(RuntimeError(what)) if code else None)
    return bf16_ops


def test_channels_last_bf16_voxelization_and_sampling_vs_oracle(KV):
    """AverageVoxelizationCL / InterpolateVoxelGridCL (the bf16 training path's voxel ops) against the oracle's
    channels-first float32 restatements of average_voxelization_3d.py:8-113 and interpolate_voxel_grid.py:61-215 on
    the same bf16-rounded inputs: forward within one bf16 rounding, backward within bf16 / float-atomic tolerance.
    Points outside the grid, a pile-up of 70 points in one voxel (the chain fallback), two batch items."""
    from oracle import oracle_np as O
    rs = np.random.RandomState(0)
    B, D, C, n = 2, 8, 16, 300
    pts = rs.uniform(-1.0, D + 0.5, (n, 3)).astype(np.float32)
    pts[100:170] = (3.2, 4.1, 2.9)          # 70 points in one voxel (NaN rows: the reference raises, model.py never has them)
    bi = np.repeat(np.arange(B), n // B).astype(np.int32)
    vals = torch.from_numpy(rs.uniform(-1, 1, (n, C)).astype(np.float32)).to(torch.bfloat16)
    v = vals.clone().requires_grad_(True)
    ldx = C + 8
    x = KV.AverageVoxelizationCL.apply(v, torch.from_numpy(pts), torch.from_numpy(bi), B, D, ldx)
    want, counts = O.average_voxelization_3d(vals.float().numpy(), pts, bi, batch_size=B, origin=(0, 0, 0), pitch=1.0,
                                             dimensions=(D, D, D))
    want_cl = torch.from_numpy(want).reshape(B, C, D ** 3).transpose(1, 2)
    got = x[:, :, :C].float()
    assert float((got - want_cl).abs().max()) <= 2.0 ** -8 * float(want_cl.abs().max()) + 1e-6
    assert int((counts > 0).sum()) > 50
    # backward: gradient rows gathered from the voxels, / count
    gx = torch.zeros(B, D ** 3, ldx, dtype=torch.bfloat16)
    gx[:, :, :C] = torch.from_numpy(rs.uniform(-1, 1, (B, D ** 3, C)).astype(np.float32)).to(torch.bfloat16)
    x.backward(gx)
    gm = gx[:, :, :C].float().transpose(1, 2).reshape(B, C, D, D, D).numpy()
    gv_o = O.average_voxelization_3d_backward(gm, pts, bi, counts, origin=(0, 0, 0), pitch=1.0, dimensions=(D, D, D))
    assert float((v.grad.float() - torch.from_numpy(gv_o)).abs().max()) <= 2.0 ** -8 * float(np.abs(gv_o).max()) + 1e-6

    # sampler
    X, Cs = 8, 16
    vox = torch.from_numpy(rs.uniform(-1, 1, (B, X ** 3, Cs)).astype(np.float32)).to(torch.bfloat16)
    vg = vox.clone().requires_grad_(True)
    sp = rs.uniform(-0.6, X - 0.4, (n, 3)).astype(np.float32)
    sp[7] = np.nan
    out = KV.InterpolateVoxelGridCL.apply(vg, torch.from_numpy(sp), torch.from_numpy(bi), X)
    vox_cf = vox.float().transpose(1, 2).reshape(B, Cs, X, X, X).numpy()
    want = O.interpolate_voxel_grid(vox_cf, sp, bi)
    assert float((out.float() - torch.from_numpy(want)).abs().max()) <= 2.0 ** -8 * float(np.abs(want).max()) + 1e-6
    g = torch.from_numpy(rs.uniform(-1, 1, (n, Cs)).astype(np.float32)).to(torch.bfloat16)
    out.backward(g)
    gv_o = O.interpolate_voxel_grid_backward(g.float().numpy(), sp, bi, (B, Cs, X, X, X))
    gv_cl = torch.from_numpy(gv_o).reshape(B, Cs, X ** 3).transpose(1, 2)
    assert float((vg.grad.float() - gv_cl).abs().max()) <= 2.0 ** -7 * float(gv_cl.abs().max()) + 1e-5
    # the same backward with the items' row offsets (the model passes them: points sorted by item), and the C-ABI's
    # fp32 output: float sums in LDS, only the summation order differs from the oracle
    vg2 = vox.clone().requires_grad_(True)
    bs = torch.tensor([0, n // B, n], dtype=torch.int32)
    KV.InterpolateVoxelGridCL.apply(vg2, torch.from_numpy(sp), torch.from_numpy(bi), X, bs).backward(g)
    assert float((vg2.grad.float() - gv_cl).abs().max()) <= 2.0 ** -7 * float(gv_cl.abs().max()) + 1e-5
    from morefusion_amd import _lib
    g32 = torch.full((B, X ** 3, Cs), 7.0)
    spt, bit = torch.from_numpy(sp), torch.from_numpy(bi)
    for start in (None, bs):
        g32.fill_(7.0)   # every element is written: no zero fill by the caller
        _lib.check(_lib.lib().mf_interpolate_voxel_grid_cl_bf16_bwd(
            g.data_ptr(), Cs, spt.data_ptr(), bit.data_ptr(), start.data_ptr() if start is not None else None, n, B, Cs,
            X, X, X, g32.data_ptr(), 0, _lib.stream_ptr()), "bwd")
        assert float((g32 - gv_cl).abs().max()) <= 1e-5 * float(gv_cl.abs().max()) + 1e-6
    # a 12^3 grid: four voxel ranges of 512, the last one partial; rows in any order (no offsets)
    X2, n2 = 12, 90
    sp2 = rs.uniform(-0.6, X2 - 0.4, (n2, 3)).astype(np.float32)
    bi2 = rs.randint(0, B, n2).astype(np.int32)
    g2 = torch.from_numpy(rs.uniform(-1, 1, (n2, 4)).astype(np.float32)).to(torch.bfloat16)
    want2 = O.interpolate_voxel_grid_backward(g2.float().numpy(), sp2, bi2, (B, 4, X2, X2, X2))
    want2 = torch.from_numpy(want2).reshape(B, 4, X2 ** 3).transpose(1, 2)
    got2 = torch.full((B, X2 ** 3, 4), 7.0)
    sp2t, bi2t = torch.from_numpy(sp2), torch.from_numpy(bi2)
    _lib.check(_lib.lib().mf_interpolate_voxel_grid_cl_bf16_bwd(
        g2.data_ptr(), 4, sp2t.data_ptr(), bi2t.data_ptr(), None, n2, B, 4, X2, X2, X2, got2.data_ptr(), 0,
        _lib.stream_ptr()), "bwd")
    assert float((got2 - want2).abs().max()) <= 1e-5 * float(want2.abs().max()) + 1e-6
    for c_small in (8, 12):   # 8 and 4 channels per workgroup (the network's layers take the 16-channel variant)
        gs = g[:, :c_small].contiguous()
        want_s = O.interpolate_voxel_grid_backward(gs.float().numpy(), sp, bi, (B, c_small, X, X, X))
        want_s = torch.from_numpy(want_s).reshape(B, c_small, X ** 3).transpose(1, 2)
        got_s = torch.full((B, X ** 3, c_small), 7.0)
        _lib.check(_lib.lib().mf_interpolate_voxel_grid_cl_bf16_bwd(
            gs.data_ptr(), c_small, spt.data_ptr(), bit.data_ptr(), bs.data_ptr(), n, B, c_small, X, X, X,
            got_s.data_ptr(), 0, _lib.stream_ptr()), "bwd")
        assert float((got_s - want_s).abs().max()) <= 1e-5 * float(want_s.abs().max()) + 1e-6


@pytest.fixture()
def KS(monkeypatch):
    """bf16_ops over the emulated engines + the sparse conv3 kernels + the voxelization kernels."""
    from morefusion_amd import _lib
    from morefusion_amd.contrib.singleview_3d.models import bf16_ops
    L = emul.build(["gemm_bf16.hip", "sparseconv_bf16.hip", "voxelize.hip"])
    for name, (argtypes, restype) in _lib._SIGNATURES.items():
        fn = getattr(L, name, None)
        if fn is not None:
            fn.argtypes, fn.restype = argtypes, restype
    monkeypatch.setattr(_lib, "lib", lambda: L)
    monkeypatch.setattr(_lib, "require_gpu", lambda *a: None)
    monkeypatch.setattr(_lib, "stream_ptr", lambda: None)
    monkeypatch.setattr(_lib, "check", lambda code, what: (_ for _ in ()).throw(RuntimeError(what)) if code else None)
    return bf16_ops


@pytest.mark.parametrize("with_occ", [True, False])
def test_sparse_conv3_operator_forward_and_all_gradients_vs_dense_float32(KS, with_occ):
    """SparseConv3 (csrc/sparseconv_bf16.hip, round 5): conv3 on the compact rows of the occupied voxels + the dense
    engine on the occupancy channels, against torch's float32 ``conv3d`` over the DENSE grid built from the same
    bf16-rounded operands (model.py:113-128: average_voxelization_3d -> concat -> conv3 -> ReLU): values, and the
    gradients to the point rows (through the voxel means), to the occupancy channels, to the weight (both channel
    ranges) and to the bias.  Points sit on and outside the grid faces, several share a voxel, one class of the
    k4 / s2 / p1 parity split stays empty in the second batch item."""
    torch.manual_seed(1)
    rs = np.random.RandomState(3)
    B, D, Cs, Co, Cout = 2, 8, 16, 8 if with_occ else 0, 256
    n0 = 70
    pts = rs.uniform(-0.4, D - 0.6, (2 * n0, 3)).astype(np.float32)
    pts[:6] = rs.uniform(-2, D + 1, (6, 3))                  # outside the grid
    pts[6:12] = pts[12:18] + rs.uniform(-0.2, 0.2, (6, 3))   # shared voxels
    pts[n0:] = np.round(pts[n0:] / 2) * 2 + 0.1              # item 1: even coordinates only -> one parity class
    pts = np.clip(pts, -3, D + 2).astype(np.float32)
    bi = np.repeat(np.arange(B, dtype=np.int32), n0)
    conv = torch.nn.Conv3d(Cs + Co, Cout, 4, 2, padding=1)
    feat = torch.randn(2 * n0, Cs).to(torch.bfloat16).requires_grad_(True)
    hocc = torch.randn(B, D ** 3, Co).to(torch.bfloat16).requires_grad_(True) if with_occ else None
    out = KS.SparseConv3.apply(feat, hocc, torch.from_numpy(pts), torch.from_numpy(bi), conv.weight, conv.bias, B, D)
    g = torch.randn(out.shape).to(torch.bfloat16)
    out.backward(g)

    # reference: dense grid of bf16-rounded voxel means (fp32 sum in point order, one rounding), float32 convolution
    fr = feat.detach().float().requires_grad_(True)
    idx = np.round(pts).astype(np.int64)
    ok = ((idx >= 0) & (idx < D)).all(1)
    key = bi.astype(np.int64) * D ** 3 + (idx[:, 0] * D + idx[:, 1]) * D + idx[:, 2]
    keys = torch.from_numpy(np.where(ok, key, 0))
    okt = torch.from_numpy(ok)
    cnt = torch.zeros(B * D ** 3).index_add_(0, keys[okt], torch.ones(int(ok.sum())))
    sums = torch.zeros(B * D ** 3, Cs).index_add_(0, keys[okt], fr[okt])
    means = sums / cnt.clamp(min=1)[:, None]
    means_b = means + (means.detach().to(torch.bfloat16).float() - means.detach())   # value rounded, gradient straight through
    x = means_b.reshape(B, D, D, D, Cs)
    if with_occ:
        hr = hocc.detach().float().requires_grad_(True)
        x = torch.cat((x, hr.reshape(B, D, D, D, Co)), dim=4)
    wr = conv.weight.detach().to(torch.bfloat16).float().requires_grad_(True)
    br = conv.bias.detach().clone().requires_grad_(True)
    y_pre = F.conv3d(x.permute(0, 4, 1, 2, 3), wr, br, stride=2, padding=1)
    y_cl = F.relu(y_pre).detach().permute(0, 2, 3, 4, 1).reshape(B, -1, Cout)
    assert rel(out, y_cl) < 2 ** -7
    # The sparse contributions are rounded to bf16 row by row before they are summed, so a pre-activation next to 0
    # can land on the other side of the ReLU than the float32 reference's (each flip moves a whole gradient row by
    # a few per cent of the maximum): the gradient kernels are checked with the OPERATOR's mask, the masks
    # themselves must agree on all but a fraction of the elements.
    mask = (out.detach().float() > 0)
    flips = (mask != (y_cl > 0)).float().mean()
    assert float(flips) < 5e-3
    gz = (g.float() * mask).reshape(B, D // 2, D // 2, D // 2, Cout).permute(0, 4, 1, 2, 3)
    y_pre.backward(gz)
    assert int(ok.sum()) < 2 * n0 and int((cnt > 1).sum()) >= 3
    assert rel(feat.grad, fr.grad) < 1e-2                      # (bf16 gradient rows)
    assert float(feat.grad[~okt].abs().max()) == 0.0           # points outside the grid receive nothing
    assert rel(conv.weight.grad[:, :Cs], wr.grad[:, :Cs]) < 1e-2
    assert rel(conv.bias.grad, br.grad) < 1e-2
    if with_occ:
        assert rel(conv.weight.grad[:, Cs:], wr.grad[:, Cs:]) < 1e-2
        assert rel(hocc.grad, hr.grad) < 1e-2


@pytest.fixture()
def KP(monkeypatch):
    from morefusion_amd import _lib
    from morefusion_amd.contrib.singleview_3d.models import bf16_ops
    L = emul.build(["pointops.hip"])
    for name, (argtypes, restype) in _lib._SIGNATURES.items():
        fn = getattr(L, name, None)
        if fn is not None:
            fn.argtypes, fn.restype = argtypes, restype
    monkeypatch.setattr(_lib, "lib", lambda: L)
    monkeypatch.setattr(_lib, "require_gpu", lambda *a: None)
    monkeypatch.setattr(_lib, "stream_ptr", lambda: None)
    monkeypatch.setattr(_lib, "check", lambda code, what: (_ for _ in ()).throw(RuntimeError(what)) if code else None)
    return bf16_ops


def pose_epilogue_torch(orot, otrn, ocnf, class_id, pts, pitch, origin, B, P, nf):
    """model.py:262-273 as the torch composite the fused operator replaces."""
    ar = torch.arange(B)
    fg = (class_id - 1).long()
    q = orot.reshape(B, P, nf, 4)[ar, :, fg]
    q = q / (q.norm(dim=2, keepdim=True) + 1e-5)
    pc = pts.reshape(B, P, 3) * pitch[:, None, None] + origin[:, None, :]
    t = pc + otrn.reshape(B, P, nf, 3)[ar, :, fg] * pitch[:, None, None]
    c = torch.sigmoid(ocnf).reshape(B, P, nf)[ar, :, fg]
    return q, t, c


def test_training_pose_epilogue_forward_and_gradients(KP):
    """K.PoseEpilogue (csrc/pointops.hip k_pose_epi3_fwd / _bwd) vs the torch composite of model.py:262-273:
    values, the three heads' gradient rows (zeros outside the object's class), NaN for a class id without a head."""
    torch.manual_seed(5)
    B, P, nf = 3, 70, 21
    n = B * P
    orot = torch.randn(n, 4 * nf, requires_grad=True)
    otrn = torch.randn(n, 3 * nf, requires_grad=True)
    ocnf = torch.randn(n, nf, requires_grad=True)
    class_id = torch.tensor([1, 21, 7])
    pts = torch.rand(n, 3) * 32
    pitch = torch.tensor([0.004, 0.0075, 0.01])
    origin = torch.randn(B, 3) * 0.1
    q, t, c = KP.PoseEpilogue.apply(orot, otrn, ocnf, class_id, pts, pitch, origin, B, P, nf)
    gq, gt, gc = torch.randn_like(q), torch.randn_like(t), torch.randn_like(c)
    torch.autograd.backward([q, t, c], [gq, gt, gc])
    got = [x.grad.clone() for x in (orot, otrn, ocnf)]
    for x in (orot, otrn, ocnf):
        x.grad = None
    qr, tr, cr = pose_epilogue_torch(orot, otrn, ocnf, class_id, pts, pitch, origin, B, P, nf)
    torch.autograd.backward([qr, tr, cr], [gq, gt, gc])
    np.testing.assert_allclose(q.detach().numpy(), qr.detach().numpy(), rtol=2e-6, atol=2e-7)
    np.testing.assert_allclose(t.detach().numpy(), tr.detach().numpy(), rtol=2e-6, atol=2e-7)
    np.testing.assert_allclose(c.detach().numpy(), cr.detach().numpy(), rtol=2e-6, atol=2e-7)
    for a, b in zip(got, (orot.grad, otrn.grad, ocnf.grad)):
        np.testing.assert_allclose(a.numpy(), b.numpy(), rtol=2e-5, atol=2e-6)
        assert (a != 0).sum() == (b != 0).sum()
    # background (0) and an id past the heads: NaN outputs, zero gradients, no out-of-range read
    bad = torch.tensor([0, 22, 7])
    q, t, c = KP.PoseEpilogue.apply(orot, otrn, ocnf, bad, pts, pitch, origin, B, P, nf)
    assert torch.isnan(q[:2]).all() and torch.isnan(t[:2]).all() and torch.isnan(c[:2]).all()
    assert torch.isfinite(q[2]).all()
    for x in (orot, otrn, ocnf):
        x.grad = None
    torch.autograd.backward([q, t, c], [gq, gt, gc])
    assert float(orot.grad[:2 * P].abs().max()) == 0.0 and float(orot.grad[2 * P:].abs().max()) > 0


def test_confidence_loss_forward_and_gradients(monkeypatch):
    """functions.loss.confidence_loss (csrc/loss.hip k_conf_loss_fwd / _bwd) vs the torch composite of
    model.py:417-434, incl. non-confident points (conf <= 0) and an object without any (NaN, zero gradients)."""
    from morefusion_amd import _lib
    import importlib
    CL = importlib.import_module("morefusion_amd.functions.loss.confidence_loss")
    L = emul.build(["loss.hip"])
    for name, (argtypes, restype) in _lib._SIGNATURES.items():
        fn = getattr(L, name, None)
        if fn is not None:
            fn.argtypes, fn.restype = argtypes, restype
    monkeypatch.setattr(_lib, "lib", lambda: L)
    monkeypatch.setattr(_lib, "stream_ptr", lambda: None)
    torch.manual_seed(3)
    for B, P, dead in ((5, 333, False), (18, 70, False), (3, 100, True)):
        add = torch.rand(B, P, requires_grad=True)
        conf = (torch.rand(B, P) - 0.2).requires_grad_(True)  # a fifth of the points not confident
        if dead:
            with torch.no_grad():
                conf[1] = -conf[1].abs()
        loss = CL._ConfidenceLoss.apply(add, conf, 0.015)
        loss.backward(torch.tensor(1.7))
        got = add.grad.clone(), conf.grad.clone()
        add.grad = conf.grad = None
        ref = CL.confidence_loss(add, conf, 0.015)  # CPU tensors: the composite
        ref.backward(torch.tensor(1.7))
        if dead:
            assert torch.isnan(loss) and torch.isnan(ref)
            assert float(got[0][1].abs().max()) == 0.0 and float(got[1][1].abs().max()) == 0.0
            continue
        np.testing.assert_allclose(float(loss.detach()), float(ref.detach()), rtol=2e-6)
        np.testing.assert_allclose(got[0].numpy(), add.grad.numpy(), rtol=1e-5, atol=1e-9)
        np.testing.assert_allclose(got[1].numpy(), conf.grad.numpy(), rtol=1e-5, atol=1e-9)


def test_sampled_pspnet_tail_training_rows_vs_torch_formulation(monkeypatch):
    """PSPNetExtractor._tail_rows_bf16 (csrc/psp_tail.hip k_tail_rows_fwd / _bwd + the bf16 GEMM engines + the PReLU
    kernel) vs the float32 torch formulation ``_tail`` (taps, four gathers, einsum, conv1d, log-softmax): the window
    rows bit-for-bit against the same arithmetic in torch, the features and every gradient within bf16 tolerance."""
    from morefusion_amd import _lib
    from morefusion_amd.models import backbone2d, ops2d
    L = emul.build(["psp_tail.hip", "gemm_bf16.hip", "backbone2d.hip"])
    for name, (argtypes, restype) in _lib._SIGNATURES.items():
        fn = getattr(L, name, None)
        if fn is not None:
            fn.argtypes, fn.restype = argtypes, restype
    monkeypatch.setattr(_lib, "lib", lambda: L)
    monkeypatch.setattr(_lib, "require_gpu", lambda *a: None)
    monkeypatch.setattr(_lib, "stream_ptr", lambda: None)
    monkeypatch.setattr(_lib, "check", lambda code, what: (_ for _ in ()).throw(RuntimeError(what)) if code else None)
    torch.manual_seed(11)
    B, H, W, P = 2, 8, 12, 48
    net = backbone2d.PSPNetExtractor()
    with torch.no_grad():
        net.up3.prelu.weight.fill_(0.3)
    u2 = torch.randn(B, 64, H, W).to(torch.bfloat16).float().contiguous(memory_format=torch.channels_last)
    pix = torch.randint(0, 4 * H * W, (B, P))
    pix[0, :4] = torch.tensor([0, 2 * W - 1, (2 * H - 1) * 2 * W, 4 * H * W - 1])   # the four image corners
    # ---- the window rows against the torch taps, in float32 before the bf16 rounding
    rows = ops2d.tail_rows(u2.to(torch.bfloat16), pix)
    taps = net._tail_taps(pix, H, W)
    flat = u2.permute(0, 2, 3, 1).reshape(B, H * W, 64)

    def tap(iy, ix):
        return torch.gather(flat, 1, (iy * W + ix)[:, :, None].expand(B, P * 9, 64))
    ly, lx = taps["ly"][:, :, None], taps["lx"][:, :, None]
    up = (1 - ly) * ((1 - lx) * tap(taps["y0"], taps["x0"]) + lx * tap(taps["y0"], taps["x1"])) + \
        ly * ((1 - lx) * tap(taps["y1"], taps["x0"]) + lx * tap(taps["y1"], taps["x1"]))
    up = (up * taps["valid"][:, :, None]).reshape(B * P, 9, 64).permute(0, 2, 1).reshape(B * P, 576)
    assert torch.equal(rows, up.to(torch.bfloat16))
    # ---- the whole tail and its gradients.  Slope 1: PReLU is the identity, every gradient within the bf16 rounding of
    # its chain; slope 0.3: a pre-activation that rounds across 0 in bf16 takes the other slope (the weight gradients
    # sum only n = 96 such terms here) -> the L2 sense
    def l2(a, b):
        return float((a.detach().float() - b.detach().float()).norm() / b.detach().float().norm())

    for slope, err, tol in ((1.0, rel, 2e-2), (0.3, l2, 6e-2)):
        with torch.no_grad():
            net.up3.prelu.weight.fill_(slope)
        net.zero_grad()
        ua = u2.clone().requires_grad_(True)
        out = net._tail_rows_bf16(ua, pix)                                             # [n, 32]
        g = torch.randn_like(out)
        out.backward(g)
        got = {k: p.grad.clone() for k, p in net.named_parameters() if p.grad is not None}
        got_u = ua.grad.clone()
        net.zero_grad()
        ub = u2.clone().contiguous().requires_grad_(True)                              # NCHW: the gather form of _tail
        ref = net._tail(ub, taps)                                                      # [B, 32, P]
        ref.backward(g.reshape(B, P, 32).transpose(1, 2))
        assert rel(out, ref.transpose(1, 2).reshape(B * P, 32)) < 2e-2
        assert err(got_u, ub.grad) < tol
        assert set(got) == {"up3.conv.weight", "up3.conv.bias", "up3.prelu.weight", "conv1.weight", "conv1.bias"}
        for k, v in got.items():
            assert v.shape == dict(net.named_parameters())[k].shape
            assert err(v, dict(net.named_parameters())[k].grad) < tol, (slope, k)
