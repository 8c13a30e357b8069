"""csrc/backbone2d.hip on the MI355X: bilinear resize (align_corners) and single-slope PReLU of PSPNet's decoder
(morefusion/models/dense_fusion/pspnet.py:18-22,50-57), forward and backward, at the decoder's shapes, against
torch's own CUDA operators (whose backward is a float-atomic scatter: compared at 1e-5 relative)."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from morefusion_amd.models import ops2d  # noqa: E402


@pytest.mark.parametrize("dtype,tol", [(torch.float32, 1e-5), (torch.bfloat16, 2.0 ** -7)])
@pytest.mark.parametrize("shape,size", [((2, 256, 64, 64), (128, 128)), ((2, 1024, 32, 32), (64, 64)),
                                        ((2, 512, 6, 6), (32, 32)), ((2, 512, 1, 1), (32, 32))])
@pytest.mark.parametrize("fmt", [torch.channels_last, torch.contiguous_format], ids=["channels_last", "channels_first"])
def test_bilinear_resize_vs_torch(dtype, tol, shape, size, fmt):
    torch.manual_seed(0)
    x = torch.randn(shape, device="cuda").to(dtype).contiguous(memory_format=fmt)
    xa = x.clone(memory_format=torch.preserve_format).requires_grad_(True)
    y = ops2d.upsample_bilinear(xa, size)
    g = torch.randn(y.shape, device="cuda").to(dtype)
    y.backward(g)
    xr = x.float().requires_grad_(True)
    yr = F.interpolate(xr, size, mode="bilinear", align_corners=True)
    yr.backward(g.float())
    assert y.is_contiguous(memory_format=fmt) or shape[2] * shape[3] == 1
    assert float((y.float() - yr).abs().max()) <= tol * max(1.0, float(yr.abs().max()))
    assert float((xa.grad.float() - xr.grad).abs().max()) <= tol * max(1.0, float(xr.grad.abs().max()))


@pytest.mark.parametrize("dtype,tol", [(torch.float32, 1e-6), (torch.bfloat16, 2.0 ** -7)])
def test_prelu_vs_torch(dtype, tol):
    torch.manual_seed(1)
    x = torch.randn(2, 64, 128, 128, device="cuda").to(dtype)
    slope = torch.tensor([0.25], device="cuda", requires_grad=True)
    xa = x.clone().requires_grad_(True)
    y = ops2d.prelu(xa, slope)
    g = torch.randn(y.shape, device="cuda").to(dtype)
    y.backward(g)
    xr = x.float().requires_grad_(True)
    sr = torch.tensor([0.25], device="cuda", requires_grad=True)
    yr = F.prelu(xr, sr)
    yr.backward(g.float())
    assert float((y.float() - yr).abs().max()) <= tol * float(yr.abs().max())
    assert float((xa.grad.float() - xr.grad).abs().max()) <= tol * float(xr.grad.abs().max())
    assert abs(float(slope.grad) - float(sr.grad)) <= 2e-3 * abs(float(sr.grad)) + 1e-3
    # deterministic (a fixed-order block sum, no float atomics)
    xa2 = x.clone().requires_grad_(True)
    s2 = torch.tensor([0.25], device="cuda", requires_grad=True)
    ops2d.prelu(xa2, s2).backward(g)
    assert float(s2.grad) == float(slope.grad)


@pytest.mark.parametrize("dtype,tol", [(torch.float32, 1e-6), (torch.bfloat16, 2.0 ** -7)])
@pytest.mark.parametrize("fmt", [torch.channels_last, torch.contiguous_format], ids=["channels_last", "channels_first"])
@pytest.mark.parametrize("residual", [False, True])
def test_batchnorm_inference_add_relu_vs_torch(dtype, tol, fmt, residual):
    """ops2d.bn_act at ResNet18Extractor's shapes vs torch's BatchNorm2d(eval) (+ add) + ReLU on the GPU."""
    torch.manual_seed(4)
    for shape in ((8, 64, 64, 64), (8, 512, 32, 32)):
        C = shape[1]
        bn = torch.nn.BatchNorm2d(C).cuda().eval()
        with torch.no_grad():
            bn.running_mean.normal_()
            bn.running_var.uniform_(0.3, 2.0)
            bn.weight.normal_()
            bn.bias.normal_()
            x = torch.randn(shape, device="cuda").to(dtype).contiguous(memory_format=fmt)
            idn = torch.randn(shape, device="cuda").to(dtype).contiguous(memory_format=fmt) if residual else None
            assert ops2d.bn_act_supported(x, bn)
            y = ops2d.bn_act(x, bn, identity=idn, relu=True)
            ref = bn(x.float())
            ref = F.relu(ref + idn.float() if residual else ref)
        assert y.is_contiguous(memory_format=fmt)
        assert float((y.float() - ref).abs().max()) <= tol * max(1.0, float(ref.abs().max()))
    assert not ops2d.bn_act_supported(x.requires_grad_(True) if False else x, bn.train())


@pytest.mark.parametrize("dtype", [torch.uint8, torch.float32])
def test_rgb_normalisation_is_bit_equal_to_the_torch_composite(dtype):
    from morefusion_amd.models.backbone2d import ResNet18Extractor as R
    torch.manual_seed(0)
    rgb = torch.randint(0, 256, (8, 256, 256, 3), dtype=torch.uint8, device="cuda").to(dtype)
    import numpy as np
    y = ops2d.normalize_rgb(rgb, R.mean_rgb, R.std_rgb)
    x = rgb.float().permute(0, 3, 1, 2)
    # the reference divides (models/resnet.py:43, NumPy / CuPy true division): bit-equal to that; torch's CUDA
    # `x / 255.0` multiplies by the rounded reciprocal -- the composite differs from both by up to an ulp of the quotient
    xn = x.cpu().numpy()
    ref_np = (xn / np.float32(255.0) - np.asarray(R.mean_rgb, np.float32)[None, :, None, None]) / \
        np.asarray(R.std_rgb, np.float32)[None, :, None, None]
    assert ref_np.dtype == np.float32 and np.array_equal(y.cpu().numpy(), ref_np)
    ref = (x / 255.0 - torch.tensor(R.mean_rgb, device="cuda").view(1, 3, 1, 1)) / \
        torch.tensor(R.std_rgb, device="cuda").view(1, 3, 1, 1)
    assert y.stride() == ref.stride()
    torch.testing.assert_close(y, ref, rtol=0, atol=1e-6)
    net = R().cuda().eval()
    with torch.no_grad():   # the extractor takes the image as it arrives (uint8 or float) and gives the same features
        a = net(rgb.permute(0, 3, 1, 2))
        b = net(x.contiguous())   # NCHW-contiguous float: the composite branch
    assert float((a - b).abs().max()) <= 1e-4 * float(b.abs().max())


def test_resnet18_extractor_inference_runs_on_the_fused_batchnorm_launch(monkeypatch):
    """Under no_grad every BatchNorm of ResNet18Extractor goes through ops2d.bn_act (20 of them, 8 with the residual
    add); with MF_TORCH_BN=1 none does, and the features agree."""
    from morefusion_amd.models.backbone2d import ResNet18Extractor as R
    torch.manual_seed(1)
    net = R().cuda().eval()
    rgb = torch.randint(0, 256, (2, 256, 256, 3), dtype=torch.uint8, device="cuda")
    calls = []
    real = ops2d.bn_act
    monkeypatch.setattr(ops2d, "bn_act", lambda x, bn, identity=None, relu=True:
                        (calls.append(identity is not None), real(x, bn, identity=identity, relu=relu))[1])
    with torch.no_grad():
        a = net(rgb.permute(0, 3, 1, 2))
        assert len(calls) == 20 and sum(calls) == 8
        monkeypatch.setenv("MF_TORCH_BN", "1")
        b = net(rgb.permute(0, 3, 1, 2))
        assert len(calls) == 20
    assert float((a - b).abs().max()) <= 1e-5 * float(b.abs().max())
