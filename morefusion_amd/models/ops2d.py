"""Autograd operators over csrc/backbone2d.hip: the memory-bound maps of PSPNet's decoder -- bilinear resize with
``align_corners`` (morefusion/models/dense_fusion/pspnet.py:18-22,50-56: ``F.resize_images``) and the single-slope
PReLU (:57) -- forward and backward, on channels-last float32 / bfloat16 tensors.

The stock kernels run these far below the HBM roofline (the resize backward is a float-atomic scatter, the PReLU slope
gradient a whole-tensor reduction: 2.8 + 1.3 ms of a 25 ms bf16 training step); here the backward passes are a
deterministic gather and a two-stage block sum.  No fallback: tensors must live on the GPU."""
import os

import torch

from .. import _lib


def _dense(x):
    """[B,C,H,W] in one of the two dense layouts the kernels read in place: (tensor, memory_format).  A tensor that is
    neither channels-first nor channels-last contiguous is copied to the one it is closer to (channels-last when its
    channel stride is 1)."""
    if x.is_contiguous(memory_format=torch.channels_last) and x.shape[1] % 8 == 0 and x.shape[1] > 1:
        return x, torch.channels_last
    if x.is_contiguous():
        return x, torch.contiguous_format
    if x.stride(1) == 1 and x.shape[1] % 8 == 0:
        return x.contiguous(memory_format=torch.channels_last), torch.channels_last
    return x.contiguous(), torch.contiguous_format


def supported(x, n_slope=1):
    """The kernels' own preconditions (callers fall back to the stock op otherwise): a 4-D float32 / bfloat16 CUDA
    tensor, one PReLU slope, an element count that is a multiple of 8 and storage the 16-byte vector accesses can
    address in place (a slice with a storage offset is neither dense nor aligned: `_dense` would copy it into an
    aligned tensor, so only the dense layouts need the pointer check)."""
    if not (x.is_cuda and x.dim() == 4 and x.dtype in (torch.float32, torch.bfloat16) and n_slope == 1):
        return False
    if x.numel() % 8:
        return False
    dense = x.is_contiguous() or x.is_contiguous(memory_format=torch.channels_last)
    return not dense or x.data_ptr() % 16 == 0


class _Upsample(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, Ho, Wo):
        _lib.require_gpu(x)
        x, fmt = _dense(x.detach())
        B, C, H, W = x.shape
        y = torch.empty((B, C, Ho, Wo), dtype=x.dtype, device=x.device, memory_format=fmt)
        bf = int(x.dtype == torch.bfloat16)
        if fmt == torch.channels_last:
            _lib.check(_lib.lib().mf_upsample_bilinear_cl_fwd(x.data_ptr(), y.data_ptr(), B, H, W, Ho, Wo, C, bf,
                                                              _lib.stream_ptr()), "mf_upsample_bilinear_cl_fwd")
        else:
            _lib.check(_lib.lib().mf_upsample_bilinear_cf_fwd(x.data_ptr(), y.data_ptr(), B * C, H, W, Ho, Wo, bf,
                                                              _lib.stream_ptr()), "mf_upsample_bilinear_cf_fwd")
        ctx.geom = (B, C, H, W, Ho, Wo, fmt)
        return y

    @staticmethod
    def backward(ctx, gy):
        B, C, H, W, Ho, Wo, fmt = ctx.geom
        gy = gy.contiguous(memory_format=fmt)
        gx = torch.empty((B, C, H, W), dtype=gy.dtype, device=gy.device, memory_format=fmt)
        bf = int(gy.dtype == torch.bfloat16)
        if fmt == torch.channels_last:
            _lib.check(_lib.lib().mf_upsample_bilinear_cl_bwd(gy.data_ptr(), gx.data_ptr(), B, H, W, Ho, Wo, C, bf,
                                                              _lib.stream_ptr()), "mf_upsample_bilinear_cl_bwd")
        else:
            _lib.check(_lib.lib().mf_upsample_bilinear_cf_bwd(gy.data_ptr(), gx.data_ptr(), B * C, H, W, Ho, Wo, bf,
                                                              _lib.stream_ptr()), "mf_upsample_bilinear_cf_bwd")
        return gx, None, None


class _PReLU(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, slope):
        _lib.require_gpu(x, slope)
        x, fmt = _dense(x.detach())       # element-wise with one slope: either dense layout is read as it lies
        if x.numel() % 8:
            raise ValueError("prelu: the element count must be a multiple of 8")
        a = slope.detach().float().contiguous()
        y = torch.empty_like(x)
        ctx.fmt = fmt
        _lib.check(_lib.lib().mf_prelu_fwd(x.data_ptr(), a.data_ptr(), y.data_ptr(), x.numel(),
                                           int(x.dtype == torch.bfloat16), _lib.stream_ptr()), "mf_prelu_fwd")
        ctx.save_for_backward(x, a)
        ctx.slope_meta = (slope.dtype, slope.shape)
        return y

    @staticmethod
    def backward(ctx, gy):
        x, a = ctx.saved_tensors
        L = _lib.lib()
        gy = gy.to(x.dtype).contiguous(memory_format=ctx.fmt)
        dx = torch.empty_like(x)
        da = torch.empty((1,), dtype=torch.float32, device=x.device)
        ws = torch.empty((max(int(L.mf_prelu_bwd_workspace_floats(x.numel())), 1),), dtype=torch.float32, device=x.device)
        _lib.check(L.mf_prelu_bwd(x.data_ptr(), gy.data_ptr(), a.data_ptr(), dx.data_ptr(), da.data_ptr(), ws.data_ptr(),
                                  x.numel(), int(x.dtype == torch.bfloat16), _lib.stream_ptr()), "mf_prelu_bwd")
        dtype, shape = ctx.slope_meta
        return dx, da.to(dtype).reshape(shape)


def normalize_rgb(rgb_hwc, mean, std):
    """``(rgb / 255 - mean) / std`` of a [B, H, W, 3] image (uint8 or float32, contiguous) in one launch -> float32
    [B, 3, H, W] in channels-last memory (what ``rgb.float().permute(0, 3, 1, 2)`` followed by the three elementwise
    operations gives)."""
    import ctypes
    _lib.require_gpu(rgb_hwc)
    B, H, W, _ = rgb_hwc.shape
    out = torch.empty((B, H, W, 3), dtype=torch.float32, device=rgb_hwc.device)
    m = (ctypes.c_float * 3)(*[float(v) for v in mean])
    s = (ctypes.c_float * 3)(*[float(v) for v in std])
    _lib.check(_lib.lib().mf_rgb_normalize(rgb_hwc.data_ptr(), int(rgb_hwc.dtype == torch.uint8),
                                           ctypes.cast(m, ctypes.c_void_p), ctypes.cast(s, ctypes.c_void_p),
                                           out.data_ptr(), B * H * W, _lib.stream_ptr()), "mf_rgb_normalize")
    return out.permute(0, 3, 1, 2)


def bn_act_supported(x, bn):
    """The fused BatchNorm(inference) kernel's preconditions: no autograd graph to build, BatchNorm in eval mode with
    running statistics and affine parameters, a dense fp32 / bf16 CUDA tensor whose layout gives 8-element runs."""
    if torch.is_grad_enabled() or bn.training or bn.running_mean is None or bn.weight is None:
        return False
    if os.environ.get("MF_TORCH_BN") == "1":  # (A/B knob: torch's three launches)
        return False
    if not (x.is_cuda and x.dim() == 4 and x.dtype in (torch.float32, torch.bfloat16)) or x.data_ptr() % 16:
        return False
    if x.is_contiguous(memory_format=torch.channels_last) and x.shape[1] % 8 == 0 and x.shape[1] > 1:
        return True
    return x.is_contiguous() and (x.shape[2] * x.shape[3]) % 8 == 0


def bn_act(x, bn, identity=None, relu=True):
    """``relu?(bn(x) (+ identity))`` for a BatchNorm2d in eval mode, one launch (csrc/backbone2d.hip k_bn_act)."""
    _lib.require_gpu(x)
    B, C, H, W = x.shape
    cl = not x.is_contiguous()
    fmt = torch.channels_last if cl else torch.contiguous_format
    if identity is not None:
        identity = identity.to(x.dtype).contiguous(memory_format=fmt)
    y = torch.empty_like(x)
    _lib.check(_lib.lib().mf_bn_act_fwd(x.data_ptr(), _lib.ptr(identity), bn.running_mean.data_ptr(),
                                        bn.running_var.data_ptr(), bn.weight.data_ptr(), bn.bias.data_ptr(),
                                        float(bn.eps), y.data_ptr(), x.numel(), C, H * W, int(cl), int(relu),
                                        int(x.dtype == torch.bfloat16), _lib.stream_ptr()), "mf_bn_act_fwd")
    return y


class _TailRows(torch.autograd.Function):
    """The 3 x 3 windows of the (virtually) x2 up-sampled map at the sampled pixels as GEMM rows: u2 [B,64,H,W] bf16
    channels-last, pix [B,P] flat indices into [2H,2W] -> [B*P, 576] bf16 (column c * 9 + ky * 3 + kx: the flattened
    layout of Convolution2D's own weight).  One launch forward (taps + four gathers + blend: ~75 torch launches), three
    backward (zero, patch-wise fp32 atomics, round) instead of four scatter-adds and their glue (~46)."""

    @staticmethod
    def forward(ctx, u2, pix):
        _lib.require_gpu(u2, pix)
        B, C, H, W = u2.shape
        if C != 64:
            raise ValueError("tail_rows: the map must have 64 channels")
        u = u2.detach().to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
        pixc = pix.reshape(-1).to(torch.int64).contiguous()
        P = pix.shape[1]
        rows = torch.empty((B * P, 576), dtype=torch.bfloat16, device=u.device)
        _lib.check(_lib.lib().mf_psp_tail_rows_bf16_fwd(u.data_ptr(), pixc.data_ptr(), B, P, H, W, rows.data_ptr(),
                                                        _lib.stream_ptr()), "mf_psp_tail_rows_bf16_fwd")
        ctx.save_for_backward(pixc)
        ctx.geom = (B, P, H, W, u2.dtype)
        return rows

    @staticmethod
    def backward(ctx, grows):
        (pixc,) = ctx.saved_tensors
        B, P, H, W, dtype = ctx.geom
        g = grows.to(torch.bfloat16).contiguous()
        acc = torch.empty((B, H, W, 64), dtype=torch.float32, device=g.device)
        gu = torch.empty((B, 64, H, W), dtype=torch.bfloat16, device=g.device, memory_format=torch.channels_last)
        _lib.check(_lib.lib().mf_psp_tail_rows_bf16_bwd(g.data_ptr(), pixc.data_ptr(), B, P, H, W, acc.data_ptr(),
                                                        gu.data_ptr(), _lib.stream_ptr()), "mf_psp_tail_rows_bf16_bwd")
        return gu.to(dtype), None


def tail_rows(u2, pix):
    return _TailRows.apply(u2, pix)


def _autocast_dtype(x):
    if x.is_cuda and torch.is_autocast_enabled() and torch.get_autocast_dtype("cuda") == torch.bfloat16:
        return x.to(torch.bfloat16)  # the convolution behind the resize would round to bf16 anyway
    return x


def upsample_bilinear(x, size):
    """``F.interpolate(x, size, mode="bilinear", align_corners=True)`` for [B,C,H,W] on the MI355X; the result has
    the memory format of the input (channels-first or channels-last, read and written in place)."""
    return _Upsample.apply(_autocast_dtype(x), int(size[0]), int(size[1]))


def prelu(x, slope):
    """``F.prelu(x, slope)`` for a single slope."""
    return _PReLU.apply(_autocast_dtype(x), slope)
