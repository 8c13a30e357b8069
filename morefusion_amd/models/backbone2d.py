"""2-D backbone of the pose network: DenseFusion-style ResNet18 + PSPNet decoder.

Restates morefusion/models/dense_fusion/resnet.py:9-136 and pspnet.py:10-82 with stock
``torch.nn`` layers (dense 2-D convolutions -> MIOpen; not hand-written).
No BatchNorm anywhere (the reference has none); ``F.resize_images`` == bilinear with
align_corners=True; PReLU has one shared slope initialised to 0.25.
"""
import torch
import torch.nn as nn
import torch.nn.functional as F


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


class PSPModule(nn.Module):
    def __init__(self, in_channels, out_channels, sizes):
        super().__init__()
        self.sizes = sizes
        self.convs = nn.ModuleList(
            [nn.Conv2d(in_channels, in_channels, 1, bias=False) for _ in sizes])
        self.bottleneck = nn.Conv2d(in_channels * (len(sizes) + 1), out_channels, 1)

    def branches(self, x):
        """The four pooled-context maps, up-sampled back to [H,W] (global receptive field)."""
        H, W = x.shape[2:]
        hs = []
        for size, conv in zip(self.sizes, self.convs):
            k = (H // size, W // size)
            h = conv(F.avg_pool2d(x, k, k))
            hs.append(F.interpolate(h, (H, W), mode="bilinear", align_corners=True))
        return hs

    def forward(self, x):
        return F.relu(self.bottleneck(torch.cat(self.branches(x) + [x], dim=1)))

    def forward_needed(self, x, where):
        """relu(bottleneck(.)) only at the positions ``where`` = (b, y, x) index vectors (the 1x1
        bottleneck is local once the pooled branches exist) -> channels-last [B,H,W,O], zeros
        elsewhere."""
        feats = torch.cat(self.branches(x) + [x], dim=1).permute(0, 2, 3, 1).contiguous()  # [B,H,W,5C]
        b, y, x_ = where
        rows = feats[b, y, x_]  # [n, 5C]
        O = self.bottleneck.out_channels
        wmat = self.bottleneck.weight.reshape(O, -1)
        rows = F.relu(torch.addmm(self.bottleneck.bias.to(rows.dtype), rows, wmat.t().to(rows.dtype)))
        out = rows.new_zeros(feats.shape[:3] + (O,))
        out[b, y, x_] = rows
        return out


class PSPUpsample(nn.Module):
    def __init__(self, in_channels, out_channels):
        super().__init__()
        self.conv = nn.Conv2d(in_channels, out_channels, 3, 1, padding=1)
        self.prelu = nn.PReLU()

    def forward(self, x):
        H, W = x.shape[2:]
        h = F.interpolate(x, (H * 2, W * 2), mode="bilinear", align_corners=True)
        return self.prelu(self.conv(h))


class PSPNetExtractor(nn.Module):
    """[B,512,h,w] -> log-softmax features [B,32,8h,8w] (dense_fusion/pspnet.py:10-35)."""

    def __init__(self):
        super().__init__()
        self.psp = PSPModule(512, 1024, [1, 2, 3, 6])
        self.up1 = PSPUpsample(1024, 256)
        self.up2 = PSPUpsample(256, 64)
        self.up3 = PSPUpsample(64, 64)
        self.conv1 = nn.Conv2d(64, 32, 1)

    def forward(self, x):
        h = F.dropout(self.psp(x), 0.3, self.training)
        h = F.dropout(self.up1(h), 0.15, self.training)
        h = F.dropout(self.up2(h), 0.15, self.training)
        h = self.up3(h)
        return F.log_softmax(self.conv1(h), dim=1)

    def forward_sampled(self, x, pix, sparse_decoder=False, plan=None):
        """Same features as ``forward(x)`` gathered at the flat pixel indices ``pix`` [B,P]
        of the full-resolution map -> [B,32,P], WITHOUT materialising the last level.

        The pose network reads only P = 1000 pixels per object of the [B,32,256,256] output
        (model.py:222); the last PSPUpsample (bilinear x2 + 3x3 conv 64->64 at 256^2, 4.8
        GFLOP and >400 MB of activations per 8 objects), the 1x1 head and the log-softmax are
        therefore evaluated at those pixels only: each sampled pixel gathers its 3x3 window
        of the (virtually) up-sampled map -- 4 bilinear taps per window element, same
        align_corners=True source-index arithmetic as ``F.interpolate`` -- and applies the
        same weights.  Mathematically identical; differs only by summation order.

        ``sparse_decoder=True`` (inference) pushes the same idea through ``up2`` and ``up1``:
        the decoder is local (bilinear x2 + 3x3 conv per level), so the 128^2 and 64^2 outputs
        the sampled pixels depend on form a small neighbourhood of the object mask; only those
        are computed (window gather + one GEMM per level), and so is the 1x1 bottleneck of the
        pyramid module.  The ResNet and the pooled pyramid branches have a global receptive
        field and stay dense."""
        taps = plan if plan is not None else self.plan(pix, x.shape[2], x.shape[3], sparse_decoder)
        if "where" in taps:
            u2 = self._decode_needed(x, taps)
        else:
            h = F.dropout(self.psp(x), 0.3, self.training)
            h = F.dropout(self.up1(h), 0.15, self.training)
            u2 = F.dropout(self.up2(h), 0.15, self.training)  # [B,64,H,W], H = W = 128
        return self._tail(u2, taps)

    @staticmethod
    def _tail_taps(pix, H, W):
        """Source taps at the [H,W] level (128^2) of every 3x3 window element of the sampled
        full-resolution pixels: indices, bilinear fractions and the zero-padding mask."""
        B, P = pix.shape
        Ho, Wo = 2 * H, 2 * W
        py, px = pix // Wo, pix % Wo  # [B,P]
        d = torch.tensor([-1, 0, 1], device=pix.device)
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

    # ---- needed-set decoder (inference) ---------------------------------------------------
    @staticmethod
    def _mark(mask_flat, iy, ix, W, src):
        mask_flat.scatter_add_(1, iy * W + ix, src)

    @staticmethod
    def needed_sets(taps):
        """Boolean maps [B,H2,W2], [B,H1,W1], [B,H0,W0] of the up2 / up1 / bottleneck outputs the
        samples depend on.  up2's set is exact (the tail gathers with the very same indices);
        each lower set is the bilinear source taps of the 3x3-dilated set above it, widened by
        one pixel so that it covers whichever neighbour ``F.interpolate`` picks at an
        exactly-integer source coordinate."""
        H2, W2 = taps["H"], taps["W"]
        B = taps["valid"].shape[0]
        dev = taps["valid"].device
        src = taps["valid"].to(torch.int32)
        m2 = torch.zeros((B, H2 * W2), dtype=torch.int32, device=dev)
        for iy, ix in ((taps["y0"], taps["x0"]), (taps["y0"], taps["x1"]),
                       (taps["y1"], taps["x0"]), (taps["y1"], taps["x1"])):
            PSPNetExtractor._mark(m2, iy, ix, W2, src)
        m2 = (m2 > 0).reshape(B, H2, W2)
        m1 = PSPNetExtractor._source_set(m2)
        return m2, m1, PSPNetExtractor._source_set(m1)

    @staticmethod
    def _source_set(need):
        """Positions of the half-resolution map that a PSPUpsample evaluated on ``need`` reads
        (bilinear taps of the 3x3-dilated set, widened by one pixel as above)."""
        B, H, W = need.shape
        Hs, Ws = H // 2, W // 2
        dev = need.device
        dil = F.max_pool2d(need[:, None].float(), 3, 1, 1)[:, 0] > 0
        gy = torch.arange(H, device=dev, dtype=torch.float32) * ((Hs - 1) / (H - 1))
        gx = torch.arange(W, device=dev, dtype=torch.float32) * ((Ws - 1) / (W - 1))
        y0, x0 = gy.floor().long(), gx.floor().long()
        y1, x1 = (y0 + 1).clamp(max=Hs - 1), (x0 + 1).clamp(max=Ws - 1)
        m = torch.zeros((B, Hs * Ws), dtype=torch.int32, device=dev)
        src = dil.reshape(B, -1).to(torch.int32)
        for iy, ix in ((y0, x0), (y0, x1), (y1, x0), (y1, x1)):
            m.scatter_add_(1, (iy[:, None] * Ws + ix[None, :]).reshape(1, -1).expand(B, -1), src)
        return F.max_pool2d((m > 0).reshape(B, 1, Hs, Ws).float(), 3, 1, 1)[:, 0] > 0

    @staticmethod
    def _sparse_up(dense_cl, where, up):
        """One PSPUpsample evaluated only at ``where`` = (b, y, x) index vectors of its [H,W]
        output.  ``dense_cl`` is the previous level [B,H/2,W/2,C] (channels last; values outside
        its own needed set are never read with a non-zero weight).  Returns the output densified
        to [B,H,W,O]."""
        B, Hs, Ws, C = dense_cl.shape
        H, W = 2 * Hs, 2 * Ws
        up2x = F.interpolate(dense_cl.permute(0, 3, 1, 2), (H, W), mode="bilinear", align_corners=True)
        up2x = up2x.permute(0, 2, 3, 1).contiguous()  # [B,H,W,C]
        b, y, x = where
        d = torch.tensor([-1, 0, 1], device=b.device)
        yy = (y[:, None, None] + d[None, :, None]).expand(-1, 3, 3).reshape(-1, 9)
        xx = (x[:, None, None] + d[None, None, :]).expand(-1, 3, 3).reshape(-1, 9)
        inside = (yy >= 0) & (yy < H) & (xx >= 0) & (xx < W)
        cols = up2x[b[:, None], yy.clamp(0, H - 1), xx.clamp(0, W - 1)]  # [n,9,C]
        cols = (cols * inside[:, :, None].to(cols.dtype)).reshape(-1, 9 * C)
        O = up.conv.out_channels
        wmat = up.conv.weight.permute(0, 2, 3, 1).reshape(O, 9 * C)  # (ky,kx) major, channel minor
        rows = up.prelu(torch.addmm(up.conv.bias.to(cols.dtype), cols, wmat.t().to(cols.dtype)))
        out = rows.new_zeros((B, H, W, O))
        out[b, y, x] = rows
        return out

    def plan(self, pix, H0, W0, sparse_decoder=False):
        """Everything ``forward_sampled`` derives from the sampled pixels alone: the tail's taps
        and, for the needed-set decoder, the (b, y, x) index vectors per level.  ``nonzero`` is a
        host synchronisation -- call this BEFORE queueing the ResNet so that it waits on a few
        tiny index kernels only ([H0,W0] = size of the ResNet output, 1/8 of the image)."""
        taps = self._tail_taps(pix, 4 * H0, 4 * W0)
        if sparse_decoder and not self.training:
            taps["where"] = tuple(torch.nonzero(m, as_tuple=True) for m in self.needed_sets(taps))
        return taps

    def _decode_needed(self, x, taps):
        w2, w1, w0 = taps["where"]
        u1 = self._sparse_up(self.psp.forward_needed(x, w0), w1, self.up1)
        u2 = self._sparse_up(u1, w2, self.up2)
        return u2.permute(0, 3, 1, 2)  # [B,64,H2,W2] view; the tail reshapes (copies) it
