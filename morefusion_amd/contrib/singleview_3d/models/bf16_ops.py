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
import ctypes
import os
import weakref

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


# Packed (cast + permuted) bf16 weight operands, keyed by the weight's storage, in-place version and the pack epoch: an
# eager inference pass (``--dtype bf16``: constant weights) packs once instead of once per call; a training step
# misses once per optimiser step, as before.  Never used while a stream is being captured (the graph must hold the
# pack kernels: the weights change between replays).
# A hipGraph REPLAY that contains an optimiser step changes the parameters on the device WITHOUT bumping
# ``_version``: whoever replays such a graph calls ``invalidate_packs()`` (parallel.DataParallelStep.replay, the
# ``--graph`` training loop) -- the epoch in the key makes every older entry unreachable, an eager forward between
# replayed steps re-packs from the current weights.  One entry per (weight, kind): a miss drops the weight's older
# versions (hundreds of MB of stale conv3 / conv4 packs would otherwise pile up during eager training).
_PACKS = {}
_PACKS_CAP = 256
_PACK_EPOCH = [0]


def invalidate_packs():
    """Forget every cached weight pack (call after the parameters changed behind autograd's back: a graph replay with
    an optimiser step, ``load_state_dict(assign=True)`` on captured storage, a manual ``copy_`` under ``no_grad`` bumps
    the version itself and needs nothing)."""
    _PACK_EPOCH[0] += 1
    _PACKS.clear()


def _cached_pack(weight, key, build):
    """``weight`` must be the module's own Parameter for a hit: the entry holds a weak reference to it and is valid only
    for that very object at that in-place version and pack epoch (an address alone can be reused by another tensor --
    e.g. the per-call ``torch.cat`` of the three heads' first layers lands at the same address every step)."""
    if not isinstance(weight, torch.nn.Parameter) or (weight.is_cuda and torch.cuda.is_current_stream_capturing()):
        return build()
    slot = (id(weight),) + key
    hit = _PACKS.get(slot)
    stamp = (weight._version, _PACK_EPOCH[0])
    if hit is None or hit[0]() is not weight or hit[1] != stamp:
        if len(_PACKS) >= _PACKS_CAP:
            _PACKS.clear()
        hit = _PACKS[slot] = (weakref.ref(weight), stamp, build())  # (replaces the slot's older version)
    return hit[2]


class Conv3d(torch.autograd.Function):
    """``relu?(Convolution3D(x))`` on a channels-last bf16 grid x [B, D^3, Cin] -> [B, Do^3, Cout], geometry
    ``(ks, stride, pad, dil)``: (4, 2, 1, 1) = conv3 / conv4, (3, 1, 1, 1) = conv1_occ, (3, 1, 2, 2) = conv2_occ.
    ``weight`` fp32 [Cout, w_cin, ks, ks, ks]; the convolution uses its input channels [c_off, c_off + Cin), reading
    channels at or beyond w_cin as zeros (conv1_occ's single input channel travels as 8 channels, 7 of them zero)."""

    @staticmethod
    def forward(ctx, x, weight, bias, D, geom, relu, c_off=0):
        _lib.require_gpu(x, weight)
        L = _lib.lib()
        ks, stride, pad, dil = geom
        x = _bf16c(x)
        B, V, Cin = x.shape
        Cout, w_cin = weight.shape[0], weight.shape[1]
        assert V == D ** 3 and tuple(weight.shape[2:]) == (ks, ks, ks)
        Do = (D + 2 * pad - dil * (ks - 1) - 1) // stride + 1
        need_dx = ctx.needs_input_grad[0]
        k4s2 = tuple(geom) == (4, 2, 1, 1)
        if need_dx and not (k4s2 or stride == 1):
            raise NotImplementedError("data gradient: k4/s2/p1 or stride-1 layers")

        # 3 x 3 x 3 / stride 1 / pad = dilation between narrow layers (the occupancy branch: 8 -> 8, 8 -> 16 and the
        # data gradient 16 -> 8): the voxels-as-columns MFMA kernel instead of 8 or 16 valid columns of a GEMM tile
        def narrow(ci, co):
            return (ks == 3 and stride == 1 and pad == dil and ci in (8, 16) and 4 <= co <= 16 and co % 4 == 0
                    and D & (D - 1) == 0 and c_off == 0 and os.environ.get("MF_NARROW_CONV", "1") != "0")
        nar_f, nar_d = narrow(Cin, Cout), need_dx and narrow(Cout, Cin)

        def build_narrow(transpose):
            w = weight.detach().float().contiguous()
            wp = _empty((int(L.mf_conv3d_k3_narrow_bf16_pack_elems(Cout if transpose else Cin)),), BF16, x)
            _lib.check(L.mf_conv3d_k3_narrow_bf16_pack(w.data_ptr(), Cout, Cin, w_cin, c_off, int(transpose), wp.data_ptr(),
                                                       _lib.stream_ptr()), "mf_conv3d_k3_narrow_bf16_pack")
            return wp

        wpd = _cached_pack(weight, ("k3n_d", Cin, x.device.index), lambda: build_narrow(True)) if nar_d else None
        if nar_f:
            wpf = _cached_pack(weight, ("k3n_f", Cin, x.device.index), lambda: build_narrow(False))
            out = _empty((B, Do ** 3, Cout), BF16, x)
            b = bias.detach().float().contiguous() if bias is not None else None
            _lib.check(L.mf_conv3d_k3_narrow_bf16(x.data_ptr(), wpf.data_ptr(), _lib.ptr(b), out.data_ptr(), B, Cin, Cout, D,
                                                  dil, int(relu), _lib.stream_ptr()), "mf_conv3d_k3_narrow_bf16")
            wf = None
            if need_dx and not nar_d:
                wf = _cached_pack(weight, ("conv3d_flipT", Cin, x.device.index), lambda: build()[2])
            ctx.save_for_backward(x, None, wf, out if relu else None, wpd)
            ctx.geom = (B, Cin, Cout, D, Do, ks, stride, pad, dil, w_cin, c_off, bool(relu), bias is not None, weight.shape)
            return out

        def build():
            w = weight.detach().float().contiguous()
            wt = _empty((Cout, ks ** 3, Cin), BF16, x)
            wd = _empty((8, Cin, 8, Cout), BF16, x) if need_dx and k4s2 else None
            wf = _empty((Cin, ks ** 3, Cout), BF16, x) if need_dx and not k4s2 else None
            _lib.check(L.mf_conv3d_bf16_pack(w.data_ptr(), Cout, Cin, w_cin, c_off, ks, wt.data_ptr(), _lib.ptr(wd),
                                             _lib.ptr(wf), _lib.stream_ptr()), "mf_conv3d_bf16_pack")
            return wt, wd, wf

        wt, wd, wf = _cached_pack(weight, ("conv3d", Cin, c_off, bool(need_dx), x.device.index), build)
        out = _empty((B, Do ** 3, Cout), BF16, x)
        b = bias.detach().float().contiguous() if bias is not None else None
        # a layer with too few output tiles for the chip (conv4 at 16 objects) splits its reduction over fp32 slabs
        nws = L.mf_conv3d_bf16_fwd_workspace_bytes(B, Cin, Cout, D, ks, stride, pad, dil)
        ws = _empty((nws,), torch.uint8, x) if nws > 0 else None
        _lib.check(L.mf_conv3d_bf16_fwd_ws(x.data_ptr(), wt.data_ptr(), _lib.ptr(b), out.data_ptr(), _lib.ptr(ws), nws,
                                           B, Cin, Cout, D, ks, stride, pad, dil, int(relu), 0, Cout,
                                           _lib.stream_ptr()), "mf_conv3d_bf16_fwd_ws")
        ctx.save_for_backward(x, wd, None if nar_d else wf, out if relu else None, wpd)
        ctx.geom = (B, Cin, Cout, D, Do, ks, stride, pad, dil, w_cin, c_off, bool(relu), bias is not None, weight.shape)
        return out

    @staticmethod
    def backward(ctx, dy):
        x, wd, wf, out, wpd = ctx.saved_tensors
        B, Cin, Cout, D, Do, ks, stride, pad, dil, w_cin, c_off, relu, has_bias, wshape = ctx.geom
        L = _lib.lib()
        dz = relu_mask(out, dy) if relu else _bf16c(dy)
        dx = dw = db = None
        if ctx.needs_input_grad[0]:
            dx = torch.empty_like(x)
            if wpd is not None:  # narrow 3 x 3 x 3 layer: the same kernel on the transposed / flipped pack
                _lib.check(L.mf_conv3d_k3_narrow_bf16(dz.data_ptr(), wpd.data_ptr(), None, dx.data_ptr(), B, Cout, Cin, D,
                                                      dil, 0, _lib.stream_ptr()), "mf_conv3d_k3_narrow_bf16 (data gradient)")
            elif wd is not None:
                _lib.check(L.mf_conv3d_k4s2_bf16_dgrad(dz.data_ptr(), wd.data_ptr(), dx.data_ptr(), B, Cin, Cout, D, 0,
                                                       0, _lib.stream_ptr()), "mf_conv3d_k4s2_bf16_dgrad")
            else:  # stride 1: dx = conv(dz, flipped / transposed weights), pad' = dil (ks - 1) - pad
                _lib.check(L.mf_conv3d_bf16_fwd(dz.data_ptr(), wf.data_ptr(), None, dx.data_ptr(), B, Cout, Cin, Do, ks,
                                                1, dil * (ks - 1) - pad, dil, 0, 0, Cin, _lib.stream_ptr()),
                           "mf_conv3d_bf16_fwd (data gradient)")
        if ctx.needs_input_grad[1]:
            split = L.mf_conv3d_bf16_wgrad_default_split(B, Cin, Cout, Do, ks)
            ws = _empty((L.mf_conv3d_bf16_wgrad_workspace_bytes(Cin, Cout, ks, split),), torch.uint8, x)
            covered = c_off == 0 and Cin >= w_cin   # every input channel of the weight is written by this call
            dw = _empty(wshape, torch.float32, x) if covered else torch.zeros(wshape, dtype=torch.float32,
                                                                                device=x.device)
            _lib.check(L.mf_conv3d_bf16_wgrad(dz.data_ptr(), x.data_ptr(), dw.data_ptr(), ws.data_ptr(), B, Cin, Cout, D,
                                              ks, stride, pad, dil, w_cin, c_off, split, _lib.stream_ptr()),
                       "mf_conv3d_bf16_wgrad")
        if has_bias and ctx.needs_input_grad[2]:
            db = dz.reshape(-1, Cout).sum(dim=0, dtype=torch.float32)
        return dx, dw, db, None, None, None, None


def _wgrad_split(M, N, K, groups):
    """Slabs of the weight-gradient reduction (csrc/gemm_bf16.hip: the cost model on the tile form that will run)."""
    return int(_lib.lib().mf_linear_wgrad_bf16_default_split(M, N, K, groups))


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

        def build():
            wb_ = _empty((N, Kp), BF16, x)
            _lib.check(L.mf_cast_rows_bf16(w2.contiguous().data_ptr(), K, wb_.data_ptr(), Kp, N, K, _lib.stream_ptr()),
                       "mf_cast_rows_bf16")
            return wb_

        wb = _cached_pack(weight, ("linear", Kp, x.device.index), build)
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
            if Np != N or Kp != K:  # a slice of the padded buffer: dense strides for the gradient buckets (DDP warns
                dw = dw.contiguous()  # "grad strides do not match bucket view strides" and copies otherwise)
        if has_bias and ctx.needs_input_grad[2]:
            db = dz.sum(dim=0, dtype=torch.float32)
        return dx, dw, db, None


def conv3d(x_cl, conv, D, relu=True, c_off=0):
    """``conv``: a torch.nn.Conv3d with a cubic kernel of 3 or 4, stride 1 or 2 (its own padding / dilation)."""
    geom = (conv.kernel_size[0], conv.stride[0], conv.padding[0], conv.dilation[0])
    return Conv3d.apply(x_cl, conv.weight, conv.bias, D, geom, relu, c_off)


def conv3d_k4s2(x_cl, conv, D, relu=True, c_off=0):
    """``conv``: torch.nn.Conv3d(.., 4, 2, padding=1)."""
    return Conv3d.apply(x_cl, conv.weight, conv.bias, D, (4, 2, 1, 1), relu, c_off)


def linear(x_rows, conv, relu=True):
    """``conv``: torch.nn.Conv1d(K, N, 1) (or nn.Linear) applied to point rows."""
    return Linear.apply(x_rows, conv.weight, conv.bias, relu)


class AverageVoxelizationCL(torch.autograd.Function):
    """``average_voxelization_3d`` (origin 0, pitch 1, grid D^3) of bf16 point rows straight into the channels-last
    bf16 conv3 input: values [n, C] -> x [B, D^3, ldx] with columns [0, C) filled (columns [C, ldx) are the caller's:
    conv3's occupancy channels).  Reference: functions/geometry/average_voxelization_3d.py:8-113 as called at
    contrib/singleview_3d/models/model.py:113."""

    @staticmethod
    def forward(ctx, values, points, batch_indices, B, D, ldx):
        _lib.require_gpu(values, points, batch_indices)
        L = _lib.lib()
        values = _bf16c(values)
        n, C = values.shape
        pts, bi = _lib.f32c(points), _lib.i32c(batch_indices)
        V = D ** 3
        x = _empty((B, V, ldx), BF16, values)
        counts = _empty((B * V,), torch.int32, values)
        head = _empty((B * V,), torch.int32, values)
        link = _empty((max(n, 1),), torch.int32, values)
        _lib.check(L.mf_average_voxelization_cl_bf16_fwd(values.data_ptr(), C, pts.data_ptr(), bi.data_ptr(), n, C, B, D,
                                                         x.data_ptr(), ldx, counts.data_ptr(), head.data_ptr(),
                                                         link.data_ptr(), _lib.stream_ptr()),
                   "mf_average_voxelization_cl_bf16_fwd")
        ctx.save_for_backward(pts, bi, counts)
        ctx.geom = (n, C, B, D, ldx)
        return x

    @staticmethod
    def backward(ctx, gx):
        pts, bi, counts = ctx.saved_tensors
        n, C, B, D, ldx = ctx.geom
        gx = _bf16c(gx)
        gv = _empty((n, C), BF16, gx)
        _lib.check(_lib.lib().mf_average_voxelization_cl_bf16_bwd(gx.data_ptr(), ldx, pts.data_ptr(), bi.data_ptr(),
                                                                  counts.data_ptr(), n, C, B, D, gv.data_ptr(), C,
                                                                  _lib.stream_ptr()),
                   "mf_average_voxelization_cl_bf16_bwd")
        return gv, None, None, None, None, None


class SparseConv3(torch.autograd.Function):
    """conv3 of the pose network WITHOUT the dense 160-channel grid (round 5; csrc/sparseconv_bf16.hip):
    ``relu(Convolution3D(160, Cout, 4, 2, pad=1)([average_voxelization_3d(feat2) | h_occ]) + bias)`` where the
    voxelized 144 channels exist only as compact rows of the occupied voxels (<= 1000 of an object's 32768) -- forward,
    data gradient (to the point rows ``feat2`` and to the dense occupancy channels ``h_occ``) and weight gradient.
    Reference: contrib/singleview_3d/models/model.py:73,113-128 (``_voxelize`` -> ``conv3``), trained by
    examples/ycb_video/singleview_3d/train.py:342-369.

    feat2 [n, Cs] bf16 point rows, points [n, 3] float32 in the voxel frame (origin 0, pitch 1), batch_indices [n]
    int32, h_occ [B, D^3, Co] bf16 channels-last or None (Co = weight.shape[1] - Cs), weight fp32
    [Cout, Cs + Co, 4, 4, 4], bias fp32 [Cout] -> [B, (D/2)^3, Cout] bf16."""

    @staticmethod
    def forward(ctx, feat2, h_occ, points, batch_indices, weight, bias, B, D):
        _lib.require_gpu(feat2, points, batch_indices, weight)
        L = _lib.lib()
        feat2 = _bf16c(feat2)
        n, Cs = feat2.shape
        Cout, w_cin = weight.shape[0], weight.shape[1]
        Co = w_cin - Cs
        assert tuple(weight.shape[2:]) == (4, 4, 4) and Co >= 0 and (h_occ is None) == (Co == 0)
        pts, bi = _lib.f32c(points), _lib.i32c(batch_indices)
        Vo, N8 = (D // 2) ** 3, 8 * Cout
        ws = _empty((L.mf_sparse_conv3_bf16_workspace_bytes(n, B, D),), torch.uint8, feat2)
        _lib.check(L.mf_sparse_conv3_bf16_index(pts.data_ptr(), bi.data_ptr(), n, B, D, ws.data_ptr(), _lib.stream_ptr()),
                   "mf_sparse_conv3_bf16_index")
        tabs = (ctypes.c_int64 * 7)()
        L.mf_sparse_conv3_bf16_tables(ws.data_ptr(), n, B, D, tabs)
        t_group, t_range, t_rowmap, t_counts, _, t_head, t_link = (int(v) for v in tabs)
        Mp = int(L.mf_sparse_conv3_bf16_max_rows(n))
        A = torch.zeros((Mp, Cs), dtype=BF16, device=feat2.device)   # pad rows stay zero (the weight gradient sums them)
        _lib.check(L.mf_average_voxelization_rows_bf16_fwd(feat2.data_ptr(), feat2.stride(0), pts.data_ptr(), bi.data_ptr(),
                                                           n, Cs, B, D, t_counts, t_head, t_link, t_rowmap, A.data_ptr(),
                                                           Cs, _lib.stream_ptr()), "mf_average_voxelization_rows_bf16_fwd")
        need_df = ctx.needs_input_grad[0]
        need_do = h_occ is not None and ctx.needs_input_grad[1]
        w32 = weight.detach().float().contiguous()
        Wp = _empty((8, N8, Cs), BF16, feat2)
        Wq = _empty((8, Cs, N8), BF16, feat2) if need_df else None
        _lib.check(L.mf_sparse_conv3_bf16_pack(w32.data_ptr(), Cout, Cs, w_cin, 0, Wp.data_ptr(), _lib.ptr(Wq),
                                               _lib.stream_ptr()), "mf_sparse_conv3_bf16_pack")
        C = _empty((Mp, N8), BF16, feat2)
        _lib.check(L.mf_linear_bf16_tiles(A.data_ptr(), Cs, Wp.data_ptr(), N8 * Cs, Cs, t_group, C.data_ptr(), N8, Mp, N8, Cs,
                                          0, _lib.stream_ptr()), "mf_linear_bf16_tiles")
        dense = wd = None
        if h_occ is not None:  # the dense channels: the dense engine at Cin = Co, fp32 pre-activation, no bias
            h_occ = _bf16c(h_occ)
            wt = _empty((Cout, 64, Co), BF16, feat2)
            _lib.check(L.mf_conv3d_bf16_pack(w32.data_ptr(), Cout, Co, w_cin, Cs, 4, wt.data_ptr(), None, None,
                                             _lib.stream_ptr()), "mf_conv3d_bf16_pack")
            if need_do:  # data-gradient operand of the narrow "columns first" form: [64 Co][Cout]
                wd = _empty((64 * Co, Cout), BF16, feat2)
                _lib.check(L.mf_conv3d_k4s2_bf16_pack_cols(w32.data_ptr(), Cout, Co, w_cin, Cs, wd.data_ptr(),
                                                           _lib.stream_ptr()), "mf_conv3d_k4s2_bf16_pack_cols")
            dense = _empty((B, Vo, Cout), torch.float32, feat2)
            _lib.check(L.mf_conv3d_bf16_fwd(h_occ.data_ptr(), wt.data_ptr(), None, dense.data_ptr(), B, Co, Cout, D, 4, 2, 1,
                                            1, 0, 1, Cout, _lib.stream_ptr()), "mf_conv3d_bf16_fwd (occupancy channels)")
        out = _empty((B, Vo, Cout), BF16, feat2)
        b = bias.detach().float().contiguous() if bias is not None else None
        _lib.check(L.mf_sparse_conv3_bf16_reduce(C.data_ptr(), _lib.ptr(dense), _lib.ptr(b), ws.data_ptr(), n, B, D, Cout, 1,
                                                 out.data_ptr(), _lib.stream_ptr()), "mf_sparse_conv3_bf16_reduce")
        ctx.save_for_backward(A, Wq, wd, h_occ, out, pts, bi, ws)
        ctx.geom = (n, Cs, Co, Cout, w_cin, B, D, Mp, bias is not None, tuple(weight.shape),
                    (t_group, t_range, t_rowmap, t_counts))
        return out

    @staticmethod
    def backward(ctx, dy):
        A, Wq, wd, h_occ, out, pts, bi, ws = ctx.saved_tensors
        n, Cs, Co, Cout, w_cin, B, D, Mp, has_bias, wshape, (t_group, t_range, t_rowmap, t_counts) = ctx.geom
        L = _lib.lib()
        N8 = 8 * Cout
        dz = relu_mask(out, dy)
        dYg = _empty((Mp, N8), BF16, dz)
        _lib.check(L.mf_sparse_conv3_bf16_gather_dy(dz.data_ptr(), ws.data_ptr(), n, B, D, Cout, dYg.data_ptr(),
                                                    _lib.stream_ptr()), "mf_sparse_conv3_bf16_gather_dy")
        dfeat = docc = dw = db = None
        if ctx.needs_input_grad[0]:
            dA = _empty((Mp, Cs), BF16, dz)
            _lib.check(L.mf_linear_bf16_tiles(dYg.data_ptr(), N8, Wq.data_ptr(), Cs * N8, N8, t_group, dA.data_ptr(), Cs, Mp,
                                              Cs, N8, 0, _lib.stream_ptr()), "mf_linear_bf16_tiles (data gradient)")
            dfeat = _empty((n, Cs), BF16, dz)
            _lib.check(L.mf_average_voxelization_rows_bf16_bwd(dA.data_ptr(), Cs, pts.data_ptr(), bi.data_ptr(), t_counts,
                                                               t_rowmap, n, Cs, B, D, dfeat.data_ptr(), Cs,
                                                               _lib.stream_ptr()), "mf_average_voxelization_rows_bf16_bwd")
        if h_occ is not None and ctx.needs_input_grad[1]:
            # columns first: T[o][tap * Co + c] = dz[o] . W[:, c, tap] (one plain GEMM, dz read once), then every
            # input voxel gathers its 8 contributions (the parity-class engine paid a 128-column tile and 2.1 GB of
            # gathered operand reads for these 16 columns: 268 us -> measured below 130)
            Mo = B * (D // 2) ** 3
            T = _empty((Mo, 64 * Co), BF16, dz)
            _lib.check(L.mf_linear_bf16(dz.data_ptr(), 0, Cout, wd.data_ptr(), 0, Cout, None, 0, T.data_ptr(), 0, 64 * Co, Mo,
                                        64 * Co, Cout, 1, 0, 0, 0, _lib.stream_ptr()), "mf_linear_bf16 (occupancy dgrad)")
            docc = torch.empty_like(h_occ)
            _lib.check(L.mf_conv3d_k4s2_bf16_col2im(T.data_ptr(), B, D, Co, docc.data_ptr(), _lib.stream_ptr()),
                       "mf_conv3d_k4s2_bf16_col2im")
        if ctx.needs_input_grad[4]:
            dWp = _empty((8, N8, Cs), torch.float32, dz)
            _lib.check(L.mf_linear_wgrad_bf16_ranges(dYg.data_ptr(), N8, A.data_ptr(), Cs, dWp.data_ptr(), N8 * Cs, Cs,
                                                     t_range, 8, N8, Cs, _lib.stream_ptr()), "mf_linear_wgrad_bf16_ranges")
            dw = _empty(wshape, torch.float32, dz)
            _lib.check(L.mf_sparse_conv3_bf16_unpack_dw(dWp.data_ptr(), Cout, Cs, w_cin, 0, dw.data_ptr(), _lib.stream_ptr()),
                       "mf_sparse_conv3_bf16_unpack_dw")
            if h_occ is not None:
                split = L.mf_conv3d_bf16_wgrad_default_split(B, Co, Cout, D // 2, 4)
                ws2 = _empty((L.mf_conv3d_bf16_wgrad_workspace_bytes(Co, Cout, 4, split),), torch.uint8, dz)
                _lib.check(L.mf_conv3d_bf16_wgrad(dz.data_ptr(), h_occ.data_ptr(), dw.data_ptr(), ws2.data_ptr(), B, Co, Cout,
                                                  D, 4, 2, 1, 1, w_cin, Cs, split, _lib.stream_ptr()),
                           "mf_conv3d_bf16_wgrad (occupancy channels)")
        if has_bias and ctx.needs_input_grad[5]:
            db = dz.reshape(-1, Cout).sum(dim=0, dtype=torch.float32)
        return dfeat, docc, None, None, dw, db, None, None


class PoseEpilogue(torch.autograd.Function):
    """Class selection + ``F.normalize`` + translation + sigmoid on the three heads' outputs (model.py:262-273), one
    launch forward and one backward (csrc/pointops.hip) instead of ~27 / ~47 torch launches (the advanced-indexing
    backward is an index_put with a radix sort, three times).  orot [n, 4 nf], otrn [n, 3 nf], ocnf [n, nf] fp32,
    class_id [B] (1-based), pts [n, 3] voxel frame, pitch [B], origin [B, 3] -> rot [B,P,4], trans [B,P,3], conf [B,P]."""

    @staticmethod
    def forward(ctx, orot, otrn, ocnf, class_id, pts, pitch, origin, B, P, nf):
        _lib.require_gpu(orot, otrn, ocnf, pts)
        L = _lib.lib()
        orot, otrn, ocnf = _lib.f32c(orot), _lib.f32c(otrn), _lib.f32c(ocnf)
        cid = torch.as_tensor(class_id).detach().to(device=orot.device, dtype=torch.int64).contiguous()
        pts_, pit, org = _lib.f32c(pts), _lib.f32c(pitch), _lib.f32c(origin)
        _lib.require_gpu(pit, org)
        rot = _empty((B, P, 4), torch.float32, orot)
        trans = _empty((B, P, 3), torch.float32, orot)
        conf = _empty((B, P), torch.float32, orot)
        _lib.check(L.mf_pose_epilogue_train_fwd(orot.data_ptr(), otrn.data_ptr(), ocnf.data_ptr(), cid.data_ptr(),
                                                pts_.data_ptr(), org.data_ptr(), pit.data_ptr(), B, P, nf, rot.data_ptr(),
                                                trans.data_ptr(), conf.data_ptr(), _lib.stream_ptr()),
                   "mf_pose_epilogue_train_fwd")
        ctx.save_for_backward(orot, ocnf, cid, pit)
        ctx.geom = (B, P, nf)
        return rot, trans, conf

    @staticmethod
    def backward(ctx, grot, gtrans, gconf):
        orot, ocnf, cid, pit = ctx.saved_tensors
        B, P, nf = ctx.geom
        n = B * P
        grot, gtrans, gconf = _lib.f32c(grot), _lib.f32c(gtrans), _lib.f32c(gconf)
        drot = _empty((n, 4 * nf), torch.float32, orot)
        dtrn = _empty((n, 3 * nf), torch.float32, orot)
        dcnf = _empty((n, nf), torch.float32, orot)
        _lib.check(_lib.lib().mf_pose_epilogue_train_bwd(orot.data_ptr(), ocnf.data_ptr(), cid.data_ptr(), pit.data_ptr(),
                                                         grot.data_ptr(), gtrans.data_ptr(), gconf.data_ptr(), B, P, nf,
                                                         drot.data_ptr(), dtrn.data_ptr(), dcnf.data_ptr(),
                                                         _lib.stream_ptr()), "mf_pose_epilogue_train_bwd")
        return drot, dtrn, dcnf, None, None, None, None, None, None, None


class InterpolateVoxelGridCL(torch.autograd.Function):
    """``interpolate_voxel_grid`` on a channels-last bf16 grid: vox [B, X^3, C], points [n, 3] (voxel units),
    batch_indices [n] -> rows [n, C] bf16.  Reference: functions/geometry/interpolate_voxel_grid.py:61-215 as called
    at contrib/singleview_3d/models/model.py:131,141."""

    @staticmethod
    def forward(ctx, vox, points, batch_indices, X, batch_start=None):
        """``batch_start`` (optional, int32 [B + 1]): row offsets of the items when the points are sorted by item --
        the backward then visits only an item's own rows per workgroup."""
        _lib.require_gpu(vox, points, batch_indices)
        vox = _bf16c(vox)
        B, V, C = vox.shape
        assert V == X ** 3
        pts, bi = _lib.f32c(points), _lib.i32c(batch_indices)
        n = pts.shape[0]
        out = _empty((n, C), BF16, vox)
        _lib.check(_lib.lib().mf_interpolate_voxel_grid_cl_bf16_fwd(vox.data_ptr(), pts.data_ptr(), bi.data_ptr(), n, B,
                                                                    C, X, X, X, out.data_ptr(), C, _lib.stream_ptr()),
                   "mf_interpolate_voxel_grid_cl_bf16_fwd")
        ctx.save_for_backward(pts, bi, _lib.i32c(batch_start) if batch_start is not None else None)
        ctx.geom = (n, B, C, X)
        return out

    @staticmethod
    def backward(ctx, g):
        pts, bi, bs = ctx.saved_tensors
        n, B, C, X = ctx.geom
        g = _bf16c(g)
        gv = _empty((B, X ** 3, C), BF16, g)   # every element written by the kernel: bf16 straight from the fp32 sums
        _lib.check(_lib.lib().mf_interpolate_voxel_grid_cl_bf16_bwd(
            g.data_ptr(), C, pts.data_ptr(), bi.data_ptr(), bs.data_ptr() if bs is not None else None, n, B, C, X, X, X,
            gv.data_ptr(), 1, _lib.stream_ptr()), "mf_interpolate_voxel_grid_cl_bf16_bwd")
        return gv, None, None, None, None
