"""csrc/backbone2d.hip on the CPU emulator through models/ops2d.py: bilinear resize with align_corners and the
single-slope PReLU of PSPNet's decoder (morefusion/models/dense_fusion/pspnet.py:18-22,50-57), forward and backward,
float32 and bfloat16, against torch's own operators + autograd."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from host_emul import emul

pytestmark = pytest.mark.skipif(not emul.available(), reason="g++ not available")


@pytest.fixture()
def ops(monkeypatch):
    from morefusion_amd import _lib
    from morefusion_amd.models import ops2d
    L = emul.build(["backbone2d.hip"])
    for name, (argtypes, restype) in _lib._SIGNATURES.items():
        fn = getattr(L, name, None)
        if fn is not None:
            fn.argtypes, fn.restype = argtypes, restype
    monkeypatch.setattr(_lib, "lib", lambda: L)
    monkeypatch.setattr(_lib, "require_gpu", lambda *a: None)
    monkeypatch.setattr(_lib, "stream_ptr", lambda: None)
    monkeypatch.setattr(_lib, "check", lambda code, what: (_ for _ in ()).throw(RuntimeError(what)) if code else None)
    return ops2d


@pytest.mark.parametrize("dtype,tol", [(torch.float32, 2e-6), (torch.bfloat16, 2.0 ** -7)])
@pytest.mark.parametrize("shape,size", [((2, 16, 5, 7), (10, 14)), ((1, 8, 1, 1), (6, 6)), ((1, 8, 3, 2), (12, 12)),
                                        ((2, 8, 6, 6), (12, 12)), ((1, 64, 9, 11), (18, 22)), ((1, 64, 8, 8), (13, 16)),
                                        ((1, 64, 1, 1), (8, 8)), ((2, 64, 3, 2), (16, 16))])
@pytest.mark.parametrize("layout", ["channels_first", "channels_last"])
def test_bilinear_resize_forward_and_backward(ops, dtype, tol, shape, size, layout):
    torch.manual_seed(0)
    x = torch.randn(shape).to(dtype)
    if layout == "channels_last":
        x = x.contiguous(memory_format=torch.channels_last)
    xa = x.clone(memory_format=torch.preserve_format).requires_grad_(True)
    y = ops.upsample_bilinear(xa, size)
    if shape[2] * shape[3] > 1:   # (a 1 x 1 map is contiguous in both formats: either result layout is right)
        assert y.is_contiguous(memory_format=torch.channels_last if layout == "channels_last" else torch.contiguous_format)
    g = torch.randn(y.shape).to(dtype)
    y.backward(g)
    xr = x.float().requires_grad_(True)
    yr = F.interpolate(xr, size, mode="bilinear", align_corners=True)
    yr.backward(g.float())
    assert y.shape == yr.shape and y.dtype == dtype
    assert float((y.float() - yr).abs().max()) <= tol * max(1.0, float(yr.abs().max()))
    assert float((xa.grad.float() - xr.grad).abs().max()) <= max(tol, 2e-6) * max(1.0, float(xr.grad.abs().max()))


@pytest.mark.parametrize("dtype,tol", [(torch.float32, 1e-6), (torch.bfloat16, 2.0 ** -7)])
def test_single_slope_prelu_forward_and_backward(ops, dtype, tol):
    torch.manual_seed(1)
    x = torch.randn(2, 16, 9, 11).to(dtype)        # 3168 elements: one ragged workgroup
    slope = torch.tensor([0.25], requires_grad=True)
    xa = x.clone().requires_grad_(True)
    y = ops.prelu(xa, slope)
    g = torch.randn(y.shape).to(dtype)
    y.backward(g)
    xr = x.float().requires_grad_(True)
    sr = torch.tensor([0.25], requires_grad=True)
    yr = F.prelu(xr, sr)
    yr.backward(g.float())
    assert float((y.float() - yr).abs().max()) <= tol * float(yr.abs().max())
    assert float((xa.grad.float() - xr.grad).abs().max()) <= tol * float(xr.grad.abs().max())
    assert abs(float(slope.grad) - float(sr.grad)) <= 1e-4 * abs(float(sr.grad)) + 1e-5


@pytest.mark.parametrize("dtype,tol", [(torch.float32, 1e-6), (torch.bfloat16, 2.0 ** -7)])
@pytest.mark.parametrize("layout", ["channels_first", "channels_last"])
@pytest.mark.parametrize("residual,relu", [(False, True), (True, True), (False, False)])
def test_batchnorm_inference_add_relu_in_one_launch(ops, monkeypatch, dtype, tol, layout, residual, relu):
    """ops2d.bn_act (k_bn_act) vs torch's BatchNorm2d in eval mode (+ add) (+ ReLU): ResNet18Extractor's units
    (models/resnet.py:44; chainercv2 ResUnit)."""
    torch.manual_seed(4)
    B, C, H, W = 2, 16, 6, 4
    bn = torch.nn.BatchNorm2d(C).eval()
    with torch.no_grad():
        bn.running_mean.normal_()
        bn.running_var.uniform_(0.3, 2.0)
        bn.weight.normal_()
        bn.bias.normal_()
    fmt = torch.channels_last if layout == "channels_last" else torch.contiguous_format
    x = torch.randn(B, C, H, W).to(dtype).contiguous(memory_format=fmt)
    idn = torch.randn(B, C, H, W).to(dtype).contiguous(memory_format=fmt) if residual else None
    monkeypatch.setattr(torch.Tensor, "is_cuda", property(lambda self: True))
    with torch.no_grad():
        assert ops.bn_act_supported(x, bn)
        y = ops.bn_act(x, bn, identity=idn, relu=relu)
    monkeypatch.undo()
    with torch.no_grad():
        ref = bn(x.float())
        if residual:
            ref = ref + idn.float()
        if relu:
            ref = F.relu(ref)
    assert y.dtype == dtype and y.shape == x.shape and y.is_contiguous(memory_format=fmt)
    assert float((y.float() - ref).abs().max()) <= tol * max(1.0, float(ref.abs().max()))
    assert not ops.bn_act_supported(x, bn)            # a CPU tensor
    assert not ops.bn_act_supported(x, bn.train())    # training statistics


@pytest.mark.parametrize("dtype", [torch.uint8, torch.float32])
def test_rgb_normalisation_in_one_launch_is_bit_equal_to_the_composite(ops, dtype):
    """ops2d.normalize_rgb (k_rgb_norm) vs ``(rgb / 255 - mean) / std`` (models/resnet.py:33-36), same operation order."""
    from morefusion_amd.models.backbone2d import ResNet18Extractor as R
    torch.manual_seed(0)
    rgb = torch.randint(0, 256, (2, 5, 7, 3), dtype=torch.uint8).to(dtype)
    y = ops.normalize_rgb(rgb, R.mean_rgb, R.std_rgb)
    x = rgb.float().permute(0, 3, 1, 2)
    ref = (x / 255.0 - torch.tensor(R.mean_rgb).view(1, 3, 1, 1)) / torch.tensor(R.std_rgb).view(1, 3, 1, 1)
    assert y.shape == ref.shape and y.stride() == ref.stride()
    assert torch.equal(y, ref)
