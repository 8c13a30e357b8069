"""Autograd operators over the bf16 MFMA kernels of csrc/gemm_bf16.hip: the 3-D convolutions and the 1x1
convolution chains of the pose network, forward AND backward, for bf16 training (BASELINE config 5) and
``--dtype bf16`` inference.

Reference: contrib/singleview_3d/models/model.py:73-74,125-139 (conv3 / conv4, ``L.Convolution3D(.., 4, 2, pad=1)``
+ ReLU) and :59-66,76-91,101-111,239-258 (``L.Convolution1D(.., 1)`` chains), trained by
examples/ycb_video/singleview_3d/train.py:342-369 (cuDNN forward / backward-data / backward-filter).

Layout: activations are bf16, channels-LAST ([B, D^3, C] grids, [n, C] point rows); parameters stay fp32 in the
framework layout and are packed (cast + permuted) into the kernels' k-contiguous bf16 operands on every call --
they change every optimiser step; the pack is ~0.1 ms for conv4's 8.4 M weights.  Gradients of activations are
bf16, gradients of parameters fp32 (what ``torch.autocast`` produces with stock operators).
No fallback: a CPU tensor or a missing libmfhip.so raises.
"""
import torch

from .... import _lib

BF16 = torch.bfloat16


def _bf16c(t):
    return t.detach().to(BF16).contiguous()


def _empty(shape, dtype, like):
    return torch.empty(shape, dtype=dtype, device=like.device)


def relu_mask(y, dy):
    """dz = dy where y > 0 else 0 (bf16); dy bf16 or fp32."""
    L = _lib.lib()
    dz = torch.empty_like(y)
    n = y.numel()
    if n % 8:
        return torch.where(y > 0, dy.to(BF16), torch.zeros_like(y))
    if dy.dtype == torch.float32:
        dy = dy.contiguous()
        _lib.check(L.mf_relu_mask_bf16(y.data_ptr(), None, dy.data_ptr(), dz.data_ptr(), n, _lib.stream_ptr()),
                   "mf_relu_mask_bf16")
    else:
        dy = _bf16c(dy)
        _lib.check(L.mf_relu_mask_bf16(y.data_ptr(), dy.data_ptr(), None, dz.data_ptr(), n, _lib.stream_ptr()),
                   "mf_relu_mask_bf16")
    return dz


class Conv3dK4S2(torch.autograd.Function):
    """``relu?(Convolution3D(Cin, Cout, 4, 2, pad=1)(x))`` on a channels-last bf16 grid:
    x [B, D^3, Cin] -> [B, (D/2)^3, Cout].  ``weight`` fp32 [Cout, w_cin, 4, 4, 4]; the convolution uses its input
    channels [c_off, c_off + Cin) (conv3's occupancy / voxelized channel groups can be fed separately)."""

    @staticmethod
    def forward(ctx, x, weight, bias, D, relu, c_off=0):
        _lib.require_gpu(x, weight)
        L = _lib.lib()
        x = _bf16c(x)
        B, V, Cin = x.shape
        Cout, w_cin = weight.shape[0], weight.shape[1]
        assert V == D ** 3 and weight.shape[2:] == (4, 4, 4) and c_off + Cin <= w_cin
        w = weight.detach().float().contiguous()
        wt = _empty((Cout, 64, Cin), BF16, x)
        need_dx = ctx.needs_input_grad[0]
        wd = _empty((8, Cin, 8, Cout), BF16, x) if need_dx else None
        _lib.check(L.mf_conv3d_k4s2_pack_bf16(w.data_ptr(), Cout, Cin, w_cin, c_off, wt.data_ptr(), _lib.ptr(wd),
                                              _lib.stream_ptr()), "mf_conv3d_k4s2_pack_bf16")
        out = _empty((B, (D // 2) ** 3, Cout), BF16, x)
        b = bias.detach().float().contiguous() if bias is not None else None
        _lib.check(L.mf_conv3d_k4s2_bf16_fwd(x.data_ptr(), wt.data_ptr(), _lib.ptr(b), out.data_ptr(), B, Cin, Cout, D,
                                             int(relu), 0, _lib.stream_ptr()), "mf_conv3d_k4s2_bf16_fwd")
        ctx.save_for_backward(x, wd, out if relu else None)
        ctx.geom = (B, Cin, Cout, D, w_cin, c_off, bool(relu), bias is not None, weight.shape)
        return out

    @staticmethod
    def backward(ctx, dy):
        x, wd, out = ctx.saved_tensors
        B, Cin, Cout, D, w_cin, c_off, relu, has_bias, wshape = ctx.geom
        L = _lib.lib()
        dz = relu_mask(out, dy) if relu else _bf16c(dy)
        dx = dw = db = None
        if ctx.needs_input_grad[0]:
            dx = torch.empty_like(x)
            _lib.check(L.mf_conv3d_k4s2_bf16_dgrad(dz.data_ptr(), wd.data_ptr(), dx.data_ptr(), B, Cin, Cout, D, 0, 0,
                                                   _lib.stream_ptr()), "mf_conv3d_k4s2_bf16_dgrad")
        if ctx.needs_input_grad[1]:
            split = L.mf_conv3d_k4s2_bf16_wgrad_default_split(B, Cin, Cout, D)
            ws = _empty((L.mf_conv3d_k4s2_bf16_wgrad_workspace_bytes(Cin, Cout, split),), torch.uint8, x)
            dw = torch.zeros(wshape, dtype=torch.float32, device=x.device) if w_cin != Cin else \
                _empty(wshape, torch.float32, x)
            _lib.check(L.mf_conv3d_k4s2_bf16_wgrad(dz.data_ptr(), x.data_ptr(), dw.data_ptr(), ws.data_ptr(), B, Cin,
                                                   Cout, D, w_cin, c_off, split, _lib.stream_ptr()),
                       "mf_conv3d_k4s2_bf16_wgrad")
        if has_bias and ctx.needs_input_grad[2]:
            db = dz.reshape(-1, Cout).sum(dim=0, dtype=torch.float32)
        return dx, dw, db, None, None, None


def _wgrad_split(M, N, K, groups):
    tiles = -(-N // 128) * -(-K // 128) * groups
    ktiles = -(-M // 64)
    s = 1
    while tiles * s < 512 and ktiles // (s * 2) >= 8:
        s *= 2
    return s


class Linear(torch.autograd.Function):
    """``relu?(x W^T + b)`` on bf16 point rows: x [n, K] (any row pitch, unit column stride) -> [n, N] bf16.
    ``weight`` fp32 [N, K] or Convolution1D's [N, K, 1].  K is padded to a multiple of 8 with zero columns when
    needed (conv1_pcd: K = 3)."""

    @staticmethod
    def forward(ctx, x, weight, bias, relu):
        _lib.require_gpu(x, weight)
        L = _lib.lib()
        w2 = weight.detach().reshape(weight.shape[0], -1).float()
        N, K = w2.shape
        Kp = -(-K // 8) * 8
        if x.dtype != BF16 or x.stride(1) != 1 or x.stride(0) % 8 or x.data_ptr() % 16 or Kp != K:
            xp = torch.zeros((x.shape[0], Kp), dtype=BF16, device=x.device) if Kp != K else None
            if xp is not None:
                xp[:, :K] = x.detach()
                x = xp
            else:
                x = _bf16c(x)
        else:
            x = x.detach()
        n = x.shape[0]
        wb = _empty((N, Kp), BF16, x)
        _lib.check(L.mf_cast_rows_bf16(w2.contiguous().data_ptr(), K, wb.data_ptr(), Kp, N, K, _lib.stream_ptr()),
                   "mf_cast_rows_bf16")
        out = _empty((n, N), BF16, x)
        b = bias.detach().float().contiguous() if bias is not None else None
        _lib.check(L.mf_linear_bf16(x.data_ptr(), 0, x.stride(0), wb.data_ptr(), 0, Kp, _lib.ptr(b), 0, out.data_ptr(),
                                    0, N, n, N, Kp, 1, int(relu), 0, 0, _lib.stream_ptr()), "mf_linear_bf16")
        ctx.save_for_backward(x, wb, out if relu else None)
        ctx.geom = (n, N, K, Kp, bool(relu), bias is not None, weight.shape)
        return out

    @staticmethod
    def backward(ctx, dy):
        x, wb, out = ctx.saved_tensors
        n, N, K, Kp, relu, has_bias, wshape = ctx.geom
        L = _lib.lib()
        dz = relu_mask(out, dy) if relu else _bf16c(dy)
        Np = -(-N // 8) * 8
        if Np != N:  # the transposed-weight GEMM and the TN engine read dz in 8-column chunks
            dzp = torch.zeros((n, Np), dtype=BF16, device=dz.device)
            dzp[:, :N] = dz
        else:
            dzp = dz
        dx = dw = db = None
        if ctx.needs_input_grad[0]:
            wt = torch.zeros((Kp, Np), dtype=BF16, device=dz.device) if Np != N else _empty((Kp, Np), BF16, dz)
            wt[:, :N] = wb.t()
            dxp = _empty((n, Kp), BF16, dz)
            _lib.check(L.mf_linear_bf16(dzp.data_ptr(), 0, Np, wt.data_ptr(), 0, Np, None, 0, dxp.data_ptr(), 0, Kp, n,
                                        Kp, Np, 1, 0, 0, 0, _lib.stream_ptr()), "mf_linear_bf16 (dgrad)")
            dx = dxp[:, :K]
        if ctx.needs_input_grad[1]:
            split = _wgrad_split(n, Np, Kp, 1)
            dwp = _empty((Np, Kp), torch.float32, dz)
            ws = _empty((split * Np * Kp,), torch.float32, dz) if split > 1 else None
            _lib.check(L.mf_linear_wgrad_bf16(dzp.data_ptr(), 0, Np, x.data_ptr(), 0, x.stride(0), dwp.data_ptr(), 0,
                                              Kp, _lib.ptr(ws), n, Np, Kp, 1, split, _lib.stream_ptr()),
                       "mf_linear_wgrad_bf16")
            dw = dwp[:N, :K].reshape(wshape)
        if has_bias and ctx.needs_input_grad[2]:
            db = dz.sum(dim=0, dtype=torch.float32)
        return dx, dw, db, None


def conv3d_k4s2(x_cl, conv, D, relu=True, c_off=0):
    """``conv``: torch.nn.Conv3d(.., 4, 2, padding=1)."""
    return Conv3dK4S2.apply(x_cl, conv.weight, conv.bias, D, relu, c_off)


def linear(x_rows, conv, relu=True):
    """``conv``: torch.nn.Conv1d(K, N, 1) (or nn.Linear) applied to point rows."""
    return Linear.apply(x_rows, conv.weight, conv.bias, relu)
