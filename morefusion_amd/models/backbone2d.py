"""2-D backbone of the pose network: DenseFusion-style ResNet18 + PSPNet decoder.

Restates morefusion/models/dense_fusion/resnet.py:9-136 and pspnet.py:10-82 with stock
``torch.nn`` layers (dense 2-D convolutions -> MIOpen; not hand-written).
No BatchNorm anywhere (the reference has none); ``F.resize_images`` == bilinear with
align_corners=True; PReLU has one shared slope initialised to 0.25.
"""

import os

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import ops2d


class BasicBlock(nn.Module):
    def __init__(self, in_channels, out_channels, stride, dilate, residual_conv=False):
        super().__init__()
        self.conv1 = nn.Conv2d(in_channels, out_channels, 3, stride, padding=dilate,
                               dilation=dilate, bias=False)
        self.conv2 = nn.Conv2d(out_channels, out_channels, 3, 1, padding=dilate,
                               dilation=dilate, bias=False)
        self.residual_conv = (
            nn.Conv2d(in_channels, out_channels, 1, stride, bias=False) if residual_conv else None)

    def forward(self, x):
        h = self.conv2(F.relu(self.conv1(x)))
        residual = x if self.residual_conv is None else self.residual_conv(x)
        return F.relu(h + residual)


class ResBlock(nn.Sequential):
    def __init__(self, n_layer, in_channels, out_channels, stride, dilate, residual_conv=True):
        blocks = [BasicBlock(in_channels, out_channels, stride, 1, residual_conv=residual_conv)]
        for _ in range(n_layer - 1):
            blocks.append(BasicBlock(out_channels, out_channels, 1, dilate))
        super().__init__(*blocks)


class ResNet18(nn.Module):
    """[B,3,H,W] uint8-range float -> [B,512,H/8,W/8] (dense_fusion/resnet.py:9-58)."""

    mean_rgb = (0.485, 0.456, 0.406)
    std_rgb = (0.229, 0.224, 0.225)

    def __init__(self):
        super().__init__()
        self.conv1 = nn.Conv2d(3, 64, 7, 2, 3, bias=False)
        self.res2 = ResBlock(2, 64, 64, 1, 1, residual_conv=False)
        self.res3 = ResBlock(2, 64, 128, 2, 1)
        self.res4 = ResBlock(2, 128, 256, 1, 2)
        self.res5 = ResBlock(2, 256, 512, 1, 4)
        self.register_buffer("mean", torch.tensor(self.mean_rgb).view(1, 3, 1, 1))
        self.register_buffer("std", torch.tensor(self.std_rgb).view(1, 3, 1, 1))

    def forward(self, x):
        h = (x / 255.0 - self.mean) / self.std
        h = self.conv1(h)
        h = F.max_pool2d(h, 3, 2, 1)
        return self.res5(self.res4(self.res3(self.res2(h))))


class _ConvBlock(nn.Module):
    """chainercv2 ``ConvBlock``: conv (no bias) + BatchNorm (+ ReLU); children ``conv``, ``bn``."""

    def __init__(self, cin, cout, k, stride=1, pad=0, dilate=1, activate=True):
        super().__init__()
        self.conv = nn.Conv2d(cin, cout, k, stride, padding=pad, dilation=dilate, bias=False)
        self.bn = nn.BatchNorm2d(cout, eps=1e-5)
        self.activate = activate

    def forward(self, x):
        h = self.conv(x)
        if ops2d.bn_act_supported(h, self.bn):  # inference: BatchNorm + ReLU in one launch
            return ops2d.bn_act(h, self.bn, relu=self.activate)
        h = self.bn(h)
        return F.relu(h) if self.activate else h


class _ResBody(nn.Module):
    def __init__(self, cin, cout, stride, dilate):
        super().__init__()
        self.conv1 = _ConvBlock(cin, cout, 3, stride, pad=dilate, dilate=dilate)
        self.conv2 = _ConvBlock(cout, cout, 3, 1, pad=dilate, dilate=dilate, activate=False)

    def forward(self, x):
        return self.conv2(self.conv1(x))


class _ResUnit(nn.Module):
    """chainercv2 ``ResUnit`` (basic block): ``body`` + optional ``identity_conv`` + ReLU."""

    def __init__(self, cin, cout, stride, dilate, resize):
        super().__init__()
        self.body = _ResBody(cin, cout, stride, dilate)
        self.identity_conv = _ConvBlock(cin, cout, 1, stride, activate=False) if resize else None

    def forward(self, x):
        identity = x if self.identity_conv is None else self.identity_conv(x)
        last = self.body.conv2
        h = last.conv(self.body.conv1(x))
        if ops2d.bn_act_supported(h, last.bn) and identity.shape == h.shape:
            return ops2d.bn_act(h, last.bn, identity=identity, relu=True)  # BatchNorm + add + ReLU in one launch
        return F.relu(last.bn(h) + identity)


class _InitBlock(nn.Module):
    def __init__(self):
        super().__init__()
        self.conv = _ConvBlock(3, 64, 7, 2, pad=3)

    def forward(self, x):
        return F.max_pool2d(self.conv(x), 3, 2, 1)


class ResNet18Extractor(nn.Module):
    """``morefusion.models.ResNet18Extractor`` (models/resnet.py:7-52): the chainercv2 ImageNet
    ResNet-18 with the strides of stages 3 / 4 removed and their second units dilated by 2 / 4, so
    that [B,3,H,W] -> [B,512,H/8,W/8] like the DenseFusion ResNet18.  BatchNorm always runs in
    inference mode (``using_config('train', False)``, :44) and no gradient flows below ``res2``
    (``h.unchain()``, :47-48).  Child names follow chainercv2 (``init_block/conv/{conv,bn}``,
    ``resN/unitM/body/convK/{conv,bn}``, ``identity_conv``) so that ``serializers`` maps a
    ``pretrained_resnet18=True`` checkpoint onto it; the ImageNet weights themselves are a
    chainercv2 download and not reachable offline (random init here; parity unpinned)."""

    mean_rgb = (0.485, 0.456, 0.406)
    std_rgb = (0.229, 0.224, 0.225)

    def __init__(self):
        super().__init__()
        self.init_block = _InitBlock()
        spec = ((64, 64, 1, 1, False), (64, 128, 2, 1, True), (128, 256, 1, 2, True), (256, 512, 1, 4, True))
        for n, (cin, cout, stride, dilate, resize) in zip((2, 3, 4, 5), spec):
            stage = nn.Module()
            stage.unit1 = _ResUnit(cin, cout, stride, 1, resize)
            stage.unit2 = _ResUnit(cout, cout, 1, dilate, False)
            setattr(self, f"res{n}", stage)
        self.register_buffer("mean", torch.tensor(self.mean_rgb).view(1, 3, 1, 1))
        self.register_buffer("std", torch.tensor(self.std_rgb).view(1, 3, 1, 1))

    def train(self, mode=True):  # BatchNorm statistics are never updated (resnet.py:44)
        super().train(mode)
        for m in self.modules():
            if isinstance(m, nn.BatchNorm2d):
                m.eval()
        return self

    def forward(self, x):
        hwc = x.permute(0, 2, 3, 1)
        if (x.is_cuda and x.shape[1] == 3 and x.dtype in (torch.uint8, torch.float32) and hwc.is_contiguous()
                and not x.requires_grad):
            h = ops2d.normalize_rgb(hwc, self.mean_rgb, self.std_rgb)  # the image as it arrives, one launch
        else:
            h = (x.float() / 255.0 - self.mean) / self.std
        with torch.no_grad():  # unchain_at="res2" (resnet.py:47-48): no gradient reaches these layers -- no graph either
            h = self.init_block(h)
            h = self.res2.unit2(self.res2.unit1(h))
        h = h.detach()
        for n in (3, 4, 5):
            stage = getattr(self, f"res{n}")
            h = stage.unit2(stage.unit1(h))
        return h


class PSPModule(nn.Module):
    def __init__(self, in_channels, out_channels, sizes):
        super().__init__()
        self.sizes = sizes
        self.convs = nn.ModuleList(
            [nn.Conv2d(in_channels, in_channels, 1, bias=False) for _ in sizes])
        self.bottleneck = nn.Conv2d(in_channels * (len(sizes) + 1), out_channels, 1)

    def _pool_matrix(self, H, W, device, dtype):
        """[H*W, sum(size^2)] matrix whose column (size, by, bx) averages that bin's window:
        ``F.avg_pool2d(x, k, k)`` with k = (H // size, W // size) for every size at once (non-overlapping
        windows, remainder rows / columns dropped: pspnet.py's pooling pyramid)."""
        key = (H, W, str(device), dtype)
        cache = self.__dict__.setdefault("_pool_cache", {})
        if key not in cache:
            import numpy as np
            cols = []
            for size in self.sizes:
                kh, kw = H // size, W // size
                for by in range((H - kh) // kh + 1):
                    for bx in range((W - kw) // kw + 1):
                        m = np.zeros((H, W), np.float32)
                        m[by * kh:(by + 1) * kh, bx * kw:(bx + 1) * kw] = 1.0 / (kh * kw)
                        cols.append(m.reshape(-1))
            with torch.inference_mode(False):  # (a constant created under inference_mode could not enter autograd later)
                cache[key] = torch.from_numpy(np.stack(cols, 1)).to(device=device, dtype=dtype)
        return cache[key]

    def __getstate__(self):  # the cached constants are rebuilt on demand: keep copies / pickles of the module lean
        state = dict(self.__dict__)
        state.pop("_pool_cache", None)
        return state

    def _pooled(self, x):
        """The pooling pyramid as ONE GEMM against the constant bin matrix (torch's avg_pool2d kernel walks each
        32 x 32 ... 5 x 5 window with a single thread: measured 0.09-0.28 ms per call, 0.4 ms per predict at
        any batch size; a strided ``mean`` is fast but its reduce kernel faults under hipGraph replay)."""
        B, C, H, W = x.shape
        # the bin weights (1/100, 1/25, ...) and the window sums stay fp32 under autocast: rounded to bf16 they bias
        # every pooled value by up to 0.2 %
        Pm = self._pool_matrix(H, W, x.device, torch.float32)
        # always through the [B, H*W, C] view (free for a channels-last x, one 2 MB copy per object otherwise): the
        # pooled maps come out channels-last, so the 1x1 convolutions and the up-sampling behind them run their
        # NHWC kernels whatever layout MIOpen's solver search left x in (torch's NCHW bilinear kernel: 0.15 ms a call)
        with torch.autocast(x.device.type, enabled=False):
            pooled = torch.matmul(Pm.t(), x.float().permute(0, 2, 3, 1).reshape(B, H * W, C)).to(x.dtype)  # [B, bins, C]
        out, o = [], 0
        for size in self.sizes:
            n_y, n_x = (H - H // size) // (H // size) + 1, (W - W // size) // (W // size) + 1
            out.append(pooled[:, o:o + n_y * n_x].transpose(1, 2).reshape(B, C, n_y, n_x)
                       .contiguous(memory_format=torch.channels_last))
            o += n_y * n_x
        return out

    def branches(self, x):
        """The four pooled-context maps, up-sampled back to [H,W] (global receptive field)."""
        H, W = x.shape[2:]
        hs = []
        for pooled, conv in zip(self._pooled(x), self.convs):
            h = conv(pooled).contiguous(memory_format=torch.channels_last)
            hs.append(ops2d.upsample_bilinear(h, (H, W)) if ops2d.supported(h) else
                      F.interpolate(h, (H, W), mode="bilinear", align_corners=True))
        return hs

    def forward(self, x):
        return F.relu(self.bottleneck(torch.cat(self.branches(x) + [x], dim=1)))


class PSPUpsample(nn.Module):
    def __init__(self, in_channels, out_channels):
        super().__init__()
        self.conv = nn.Conv2d(in_channels, out_channels, 3, 1, padding=1)
        self.prelu = nn.PReLU()

    def forward(self, x):
        H, W = x.shape[2:]
        # the resize and the single-slope PReLU on csrc/backbone2d.hip (memory-bound maps, forward + backward)
        if ops2d.supported(x):
            h = ops2d.upsample_bilinear(x, (H * 2, W * 2))
        else:
            h = F.interpolate(x, (H * 2, W * 2), mode="bilinear", align_corners=True)
        h = self.conv(h)
        if ops2d.supported(h, self.prelu.weight.numel()):
            return ops2d.prelu(h, self.prelu.weight)
        return self.prelu(h)


class PSPNetExtractor(nn.Module):
    """[B,512,h,w] -> log-softmax features [B,32,8h,8w] (dense_fusion/pspnet.py:10-35)."""

    def __init__(self):
        super().__init__()
        self.psp = PSPModule(512, 1024, [1, 2, 3, 6])
        self.up1 = PSPUpsample(1024, 256)
        self.up2 = PSPUpsample(256, 64)
        self.up3 = PSPUpsample(64, 64)
        self.conv1 = nn.Conv2d(64, 32, 1)

    bf16_tail_kernels = True  # the sampled tail under bf16 autocast on the hand-written kernels (_tail_rows_bf16)

    def __getstate__(self):  # the tail kernel's packed weights are a cache
        state = dict(self.__dict__)
        state.pop("_tail_pack", None)
        return state

    def forward(self, x):
        h = F.dropout(self.psp(x), 0.3, self.training)
        h = F.dropout(self.up1(h), 0.15, self.training)
        h = F.dropout(self.up2(h), 0.15, self.training)
        h = self.up3(h)
        return F.log_softmax(self.conv1(h), dim=1)

    def forward_sampled(self, x, pix, plan=None):
        """Same features as ``forward(x)`` gathered at the flat pixel indices ``pix`` [B,P]
        of the full-resolution map -> [B,32,P], WITHOUT materialising the last level.

        The pose network reads only P = 1000 pixels per object of the [B,32,256,256] output
        (model.py:222); the last PSPUpsample (bilinear x2 + 3x3 conv 64->64 at 256^2, 4.8
        GFLOP and >400 MB of activations per 8 objects), the 1x1 head and the log-softmax are
        therefore evaluated at those pixels only: each sampled pixel gathers its 3x3 window
        of the (virtually) up-sampled map -- 4 bilinear taps per window element, same
        align_corners=True source-index arithmetic as ``F.interpolate`` -- and applies the
        same weights.  Mathematically identical; differs only by summation order.
        (Restricting up1/up2 to the outputs the samples depend on as well was built in round 1
        and measured in round 2: 8.13 ms vs 7.96 ms per 8-object predict -- slower, removed.)"""
        h = F.dropout(self.psp(x), 0.3, self.training)
        h = F.dropout(self.up1(h), 0.15, self.training)
        u2 = F.dropout(self.up2(h), 0.15, self.training)  # [B,64,H,W], H = W = 128
        if (plan is None and self.bf16_tail_kernels and u2.is_cuda and u2.shape[1] == 64 and self.conv1.out_channels == 32
                and self.up3.prelu.weight.numel() == 1 and torch.is_autocast_enabled()
                and torch.get_autocast_dtype("cuda") == torch.bfloat16 and os.environ.get("MF_TORCH_TAIL") != "1"):
            B, P = pix.shape
            return self._tail_rows_bf16(u2, pix).reshape(B, P, -1).transpose(1, 2)  # (a view: the rows are the data)
        taps = plan if plan is not None else self.plan(pix, x.shape[2], x.shape[3])
        return self._tail(u2, taps)

    def _tail_rows_bf16(self, u2, pix):
        """``_tail`` under bf16 autocast (training, ``--dtype bf16``) -> rows [B*P, 32] fp32: the windows as GEMM rows
        from one launch (ops2d.tail_rows), up3's 3x3 convolution and the 1x1 head as GEMMs on the bf16 MFMA engines with
        their data / weight gradients (bf16_ops.Linear on the convolutions' own parameters), the single-slope PReLU
        on its kernel, log-softmax in fp32 -- ~12 launches forward + ~25 backward where the torch formulation takes
        ~85 + ~46 (MF_TORCH_TAIL=1 keeps that form for A/B runs)."""
        from ..contrib.singleview_3d.models import bf16_ops as K
        n = pix.numel()
        rows = ops2d.tail_rows(u2, pix)                                               # [n, 576] bf16
        pre = K.Linear.apply(rows, self.up3.conv.weight, self.up3.conv.bias, False)   # [n, 64] bf16
        h = ops2d._PReLU.apply(pre.view(1, 1, n, 64), self.up3.prelu.weight).view(n, 64)
        z = K.Linear.apply(h, self.conv1.weight, self.conv1.bias, False)              # [n, 32] bf16
        return F.log_softmax(z.float(), dim=1)

    def forward_sampled_rows(self, x, pix):
        """``forward_sampled`` through ONE HIP launch (csrc/psp_tail.hip) -> rows [B*P, 32] (the point MLP's
        GEMM input): the taps' index arithmetic, the four gathers, the 3x3 convolution + PReLU, the 1x1
        convolution and the log-softmax of the torch formulation (~55 launches) fused.  Inference, fp32, CUDA."""
        from .. import _lib
        h = self.psp(x)
        h = self.up1(h)
        u2 = self.up2(h)  # [B,64,H,W], H = W = 128; NCHW or channels-last strides, both read in place
        if u2.dtype != torch.float32:
            u2 = u2.float()
        B, C, H, W = u2.shape
        assert C == 64 and self.conv1.out_channels == 32
        key = (self.up3.conv.weight.data_ptr(), self.up3.conv.weight._version, self.conv1.weight.data_ptr(),
               self.conv1.weight._version)
        pack = self.__dict__.get("_tail_pack")
        if pack is None or pack[0] != key:
            w3t = self.up3.conv.weight.detach().float().permute(2, 3, 1, 0).reshape(9, 64, 64).contiguous()
            w1t = self.conv1.weight.detach().float().reshape(32, 64).t().contiguous()
            pack = (key, w3t, w1t)
            self.__dict__["_tail_pack"] = pack
        _, w3t, w1t = pack
        _lib.require_gpu(u2, pix)
        pixc = pix.reshape(-1).to(torch.int64).contiguous()
        P = pix.shape[1]
        out = torch.empty((B * P, 32), dtype=torch.float32, device=u2.device)
        _lib.check(_lib.lib().mf_psp_tail_fwd(
            u2.data_ptr(), u2.stride(0), u2.stride(1), u2.stride(2), u2.stride(3), pixc.data_ptr(), w3t.data_ptr(),
            self.up3.conv.bias.detach().float().data_ptr(), self.up3.prelu.weight.detach().float().data_ptr(),
            w1t.data_ptr(), self.conv1.bias.detach().float().data_ptr(), B, P, H, W, out.data_ptr(),
            _lib.stream_ptr()), "mf_psp_tail_fwd")
        return out

    @staticmethod
    def _tail_taps(pix, H, W):
        """Source taps at the [H,W] level (128^2) of every 3x3 window element of the sampled
        full-resolution pixels: indices, bilinear fractions and the zero-padding mask."""
        B, P = pix.shape
        Ho, Wo = 2 * H, 2 * W
        py, px = pix // Wo, pix % Wo  # [B,P]
        d = torch.arange(-1, 2, device=pix.device)  # (a device-side arange: safe under hipGraph capture)
        yy = (py[:, :, None, None] + d[None, None, :, None]).expand(B, P, 3, 3).reshape(B, P * 9)
        xx = (px[:, :, None, None] + d[None, None, None, :]).expand(B, P, 3, 3).reshape(B, P * 9)
        valid = (yy >= 0) & (yy < Ho) & (xx >= 0) & (xx < Wo)  # zero padding of the 3x3 conv
        yy, xx = yy.clamp(0, Ho - 1), xx.clamp(0, Wo - 1)
        sy = yy.to(torch.float32) * ((H - 1) / (Ho - 1))
        sx = xx.to(torch.float32) * ((W - 1) / (Wo - 1))
        y0, x0 = sy.floor().long(), sx.floor().long()
        y1, x1 = (y0 + 1).clamp(max=H - 1), (x0 + 1).clamp(max=W - 1)
        return dict(P=P, H=H, W=W, valid=valid, y0=y0, x0=x0, y1=y1, x1=x1, ly=sy - y0, lx=sx - x0)

    def _tail(self, u2, taps):
        """up3 (bilinear x2 + 3x3 conv + PReLU), the 1x1 head and log-softmax at the samples."""
        B, C, H, W = u2.shape
        P = taps["P"]
        if u2.is_contiguous(memory_format=torch.channels_last) and not u2.is_contiguous():
            return self._tail_nhwc(u2, taps)
        ly, lx = taps["ly"].to(u2.dtype)[:, None, :], taps["lx"].to(u2.dtype)[:, None, :]
        flat = u2.reshape(B, C, H * W)

        def tap(iy, ix):
            return torch.gather(flat, 2, (iy * W + ix)[:, None, :].expand(B, C, P * 9))

        y0, x0, y1, x1 = taps["y0"], taps["x0"], taps["y1"], taps["x1"]
        up = (1 - ly) * ((1 - lx) * tap(y0, x0) + lx * tap(y0, x1)) + \
            ly * ((1 - lx) * tap(y1, x0) + lx * tap(y1, x1))
        up = (up * taps["valid"][:, None, :]).reshape(B, C, P, 9)
        w = self.up3.conv.weight.reshape(self.up3.conv.out_channels, C, 9)
        h = torch.einsum("bcpk,ock->bop", up, w) + self.up3.conv.bias[None, :, None]
        h = self.up3.prelu(h)
        h = F.conv1d(h, self.conv1.weight.reshape(self.conv1.out_channels, -1, 1), self.conv1.bias)
        return F.log_softmax(h, dim=1)

    def _tail_nhwc(self, u2, taps):
        """``_tail`` for a channels-last ``u2`` (the backbone in NHWC memory format): a tap is C contiguous
        floats -- row gathers from the [B, H*W, C] view instead of C strided element gathers."""
        B, C, H, W = u2.shape
        P = taps["P"]
        ly, lx = taps["ly"].to(u2.dtype)[:, :, None], taps["lx"].to(u2.dtype)[:, :, None]
        flat = u2.permute(0, 2, 3, 1).reshape(B, H * W, C)  # a view of the NHWC storage

        def tap(iy, ix):
            return torch.gather(flat, 1, (iy * W + ix)[:, :, None].expand(B, P * 9, C))

        y0, x0, y1, x1 = taps["y0"], taps["x0"], taps["y1"], taps["x1"]
        up = (1 - ly) * ((1 - lx) * tap(y0, x0) + lx * tap(y0, x1)) + \
            ly * ((1 - lx) * tap(y1, x0) + lx * tap(y1, x1))
        up = (up * taps["valid"][:, :, None]).reshape(B, P, 9, C)
        w = self.up3.conv.weight.reshape(self.up3.conv.out_channels, C, 9)
        h = torch.einsum("bpkc,ock->bop", up, w) + self.up3.conv.bias[None, :, None]
        h = self.up3.prelu(h)
        h = F.conv1d(h, self.conv1.weight.reshape(self.conv1.out_channels, -1, 1), self.conv1.bias)
        return F.log_softmax(h, dim=1)

    def plan(self, pix, H0, W0):
        """What ``forward_sampled`` derives from the sampled pixels alone (the tail's taps);
        [H0,W0] = size of the ResNet output, 1/8 of the image."""
        return self._tail_taps(pix, 4 * H0, 4 * W0)
