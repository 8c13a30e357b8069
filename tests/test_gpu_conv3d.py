"""GPU parity of the hand-written 3-D CNN kernels of round 3 (csrc/conv3d.hip + the channels-last entries of
sparseconv.hip / interp.hip) against the operators the reference applies there -- cuDNN Convolution3D in
contrib/singleview_3d/models/model.py:69-74,120-139, here torch's Conv3d (MIOpen) / float64 CPU -- at the
network's full sizes, and of the channels-last inference path against the channels-first one stage by stage.
Tolerances: exact-fp32 MFMA sums in another order than MIOpen's: 1e-4 of the largest output for the K = 16384
conv4 reduction, 2e-5 for the short ones; bit-equal where the summation order is the same by construction."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

import morefusion_amd as mf  # noqa: E402
from morefusion_amd import _lib  # noqa: E402
from morefusion_amd.contrib.singleview_3d.models import Model  # noqa: E402
from morefusion_amd.contrib.singleview_3d.models.volumetric_cl import ChannelsLastVolumetric  # noqa: E402


def _model(seed=0):
    torch.manual_seed(seed)
    return Model(n_fg_class=21, with_occupancy=True).cuda().eval()


def _cl(x_cf):  # [B,C,D,D,D] -> [B,D^3,C]
    B, C = x_cf.shape[:2]
    return x_cf.reshape(B, C, -1).transpose(1, 2).contiguous()


def _cf(x_cl, D):  # [B,D^3,C] -> [B,C,D,D,D]
    B, _, C = x_cl.shape
    return x_cl.transpose(1, 2).reshape(B, C, D, D, D)


@pytest.mark.parametrize("B", [1, 3, 8])
def test_conv4_implicit_gemm_vs_float64(B):
    """conv4 (256 -> 512 on 16^3) at every split the heuristic can pick, vs a float64 convolution."""
    model = _model()
    vol = ChannelsLastVolumetric(model)
    torch.manual_seed(B)
    h3 = torch.relu(torch.randn(B, 256, 16, 16, 16, device="cuda"))
    with torch.no_grad():
        ref = torch.relu(F.conv3d(h3.double().cpu(), model.conv4.weight.double().cpu(), model.conv4.bias.double().cpu(),
                                  stride=2, padding=1)).float()
    scale = float(ref.abs().max())
    outs = []
    for split in sorted({1, 4, _lib.lib().mf_conv3d_k4s2_default_split(B, 256, 512, 16), 64}):
        got = _cf(vol.conv_k4s2("conv4", model.conv4, _cl(h3), B, 16, cin=256, split=split), 8).cpu()
        assert float((got - ref).abs().max()) < 1e-4 * scale, split
        outs.append(got)
    # run-to-run determinism of the split-K sum (slabs added in slab order, no atomics)
    again = _cf(vol.conv_k4s2("conv4", model.conv4, _cl(h3), B, 16, cin=256, split=64), 8).cpu()
    assert torch.equal(again, outs[-1])
    # and against MIOpen's fp32 convolution (what round 2 shipped)
    with torch.no_grad():
        mi = torch.relu(model.conv4(h3)).cpu()
    assert float((outs[0] - mi).abs().max()) < 2e-4 * scale


def test_conv3_dense_channels_and_occupancy_convs_vs_torch():
    model = _model(1)
    vol = ChannelsLastVolumetric(model)
    b = mf.synthetic.make_singleview_batch(4, seed=11)
    grid = torch.as_tensor(b["grid_nontarget_empty"]).cuda()
    g = grid.float()[:, None]
    h_occ_ref = torch.relu(model.conv2_occ(torch.relu(model.conv1_occ(g))))
    h_occ = vol.occupancy(grid)
    assert float((_cf(h_occ, 32) - h_occ_ref).abs().max()) < 2e-5 * max(1.0, float(h_occ_ref.abs().max()))
    dense_ref = F.conv3d(h_occ_ref.double().cpu(), model.conv3.weight[:, 144:].double().cpu(), None, stride=2, padding=1)
    dense = vol.conv_k4s2("conv3_occ", model.conv3, h_occ, 4, 32, cin=16, c_off=144, relu=False, bias=False)
    assert float((_cf(dense, 16).cpu().double() - dense_ref).abs().max()) < 3e-5 * max(1.0, float(dense_ref.abs().max()))


def test_channels_last_path_equals_channels_first_path_stage_by_stage():
    """Same weights, same inputs: round 2's inference path (channels-first, stock conv4 / occupancy convs)
    vs the channels-last path.  The sparse part of conv3 and both samplers are bit-identical given identical
    inputs; the final per-point outputs agree to the convolution tolerance."""
    model = _model(2)
    B, P = 4, 1000
    b = mf.synthetic.make_singleview_batch(B, seed=5)
    inp = {k: torch.as_tensor(b[k]).cuda() for k in ("class_id", "rgb", "pcd", "pitch", "origin", "grid_nontarget_empty")}
    with torch.no_grad():
        pix = model._select_points(inp["pcd"])
        values, points = model._backbone_features(inp["rgb"], inp["pcd"], pix)
        args = (inp["class_id"], values, points, inp["pitch"].float(), inp["origin"].float(), inp["grid_nontarget_empty"])
        model.channels_last_3d = False
        rot0, trans0, conf0 = model._pose_from_features(*args)
        model.channels_last_3d = True
        rot1, trans1, conf1 = model._pose_from_features(*args)
        assert getattr(model, "_volumetric_cl", None) is not None   # the new path ran
        # stage check: sparse conv3 (no dense part) channels-last vs channels-first, bit-identical
        vol = model._volumetric_cl
        pv = ((points - inp["origin"].float()[:, :, None]) / inp["pitch"].float()[:, None, None])
        feat, pts_k = vol.features(values, points, inp["pitch"].float(), inp["origin"].float(), inp["grid_nontarget_empty"])
        pts = pv.transpose(1, 2).reshape(B * P, 3).contiguous()
        assert torch.equal(pts_k, pts)   # mf_point_prep: the torch expression's bits
        bi = torch.arange(B, dtype=torch.int32, device="cuda").repeat_interleave(P)
        f2 = feat[:, 72:216].contiguous()
        h3_cf = vol._sparse.from_points(f2, pts, bi, batch_size=B, h_dense=None, dim=32)
        h3_cl = vol._sparse.from_points_cl(feat[:, 72:216], feat.stride(0), pts, bi, B, None, 32)
        assert torch.equal(_cf(h3_cl, 16), h3_cf)
        s_cf = mf.functions.interpolate_voxel_grid(h3_cf, pts / 2.0, bi)
        blk = torch.zeros((B * P, 260), device="cuda")
        vol.sample(h3_cl, 16, pts / 2.0, bi, blk[:, 4:], 260)
        assert torch.equal(blk[:, 4:], s_cf) and float(blk[:, :4].abs().sum()) == 0.0
    np.testing.assert_allclose(rot1.cpu().numpy(), rot0.cpu().numpy(), rtol=0, atol=2e-4)
    np.testing.assert_allclose(conf1.cpu().numpy(), conf0.cpu().numpy(), rtol=0, atol=2e-4)
    pitch = inp["pitch"].float().cpu().numpy().reshape(B, 1, 1)
    np.testing.assert_allclose(trans1.cpu().numpy() / pitch, trans0.cpu().numpy() / pitch, rtol=0, atol=2e-4)


@pytest.mark.parametrize("fmt", [torch.contiguous_format, torch.channels_last], ids=["nchw", "nhwc"])
def test_psp_tail_kernel_vs_torch_formulation_on_gpu(fmt):
    """``PSPNetExtractor.forward_sampled_rows`` (csrc/psp_tail.hip, one launch) against ``forward_sampled`` (the torch
    formulation of round 1-2, pinned against the dense decoder of models/dense_fusion/pspnet.py by
    tests/test_host_logic.py) at the network's size: 8 objects x 1000 pixels of a 256^2 image, incl. border pixels."""
    torch.manual_seed(3)
    model = _model(3)
    net = model.pspnet_extractor
    x = torch.randn(8, 512, 32, 32, device="cuda").contiguous(memory_format=fmt)
    pix = torch.randint(0, 256 * 256, (8, 1000), device="cuda")
    pix[:, :4] = torch.tensor([0, 255, 255 * 256, 256 * 256 - 1], device="cuda")
    with torch.no_grad():
        ref = net.forward_sampled(x, pix).transpose(1, 2).reshape(8000, 32)
        got = net.forward_sampled_rows(x, pix)
    assert float((got - ref).abs().max()) < 5e-5 * max(1.0, float(ref.abs().max()))
    assert float((got.exp().sum(1) - 1).abs().max()) < 1e-4
