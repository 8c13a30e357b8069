"""interpolate_voxel_grid -- trilinear sampling of [B,C,X,Y,Z] grids at points.

API of morefusion/functions/geometry/interpolate_voxel_grid.py:271-272; kernels
(:170-212, :224-266) replaced by ``mf_interpolate_voxel_grid_{fwd,bwd}``
(morefusion_amd/csrc/interp.hip).  Points are in voxel-index units.
"""
import os

import torch

from ... import _lib


class InterpolateVoxelGrid(torch.autograd.Function):
    @staticmethod
    def forward(ctx, voxelized, points, batch_indices, channels_first, batch_start=None):
        _lib.require_gpu(voxelized, points, batch_indices)
        if voxelized.dtype != torch.float32 or voxelized.ndim != 5:
            raise TypeError("voxelized must be float32 [B, C, X, Y, Z]")
        if points.dtype != torch.float32 or points.ndim != 2 or points.shape[1] != 3:
            raise TypeError("points must be float32 [P, 3]")
        if (batch_indices.dtype != torch.int32 or batch_indices.ndim != 1
                or batch_indices.shape[0] != points.shape[0]):
            raise TypeError("batch_indices must be int32 [P]")
        vox, pts, bi = voxelized.contiguous(), points.contiguous(), batch_indices.contiguous()
        B, C, X, Y, Z = vox.shape
        if batch_start is not None:
            _lib.require_gpu(batch_start)
            if batch_start.dtype != torch.int32 or batch_start.shape != (B + 1,):
                raise TypeError("batch_start must be int32 [B + 1]")
            batch_start = batch_start.contiguous()
        n = pts.shape[0]
        shape = (C, n) if channels_first else (n, C)
        values = torch.empty(shape, dtype=torch.float32, device=vox.device)
        _lib.check(
            _lib.lib().mf_interpolate_voxel_grid_fwd(
                vox.data_ptr(), pts.data_ptr(), bi.data_ptr(), _lib.ptr(batch_start), n, B, C, X, Y, Z,
                values.data_ptr(), int(channels_first), _lib.stream_ptr()),
            "mf_interpolate_voxel_grid_fwd")
        ctx.save_for_backward(pts, bi)
        ctx.batch_start = batch_start
        ctx.meta = (B, C, X, Y, Z, bool(channels_first))
        return values

    @staticmethod
    def backward(ctx, gvalues):
        pts, bi = ctx.saved_tensors
        B, C, X, Y, Z, channels_first = ctx.meta
        gvalues = gvalues.contiguous()
        gvox = torch.empty((B, C, X, Y, Z), dtype=torch.float32, device=gvalues.device)
        _lib.check(
            _lib.lib().mf_interpolate_voxel_grid_bwd(
                gvalues.data_ptr(), pts.data_ptr(), bi.data_ptr(), _lib.ptr(ctx.batch_start),
                pts.shape[0], B, C, X, Y, Z, gvox.data_ptr(), int(channels_first), _lib.stream_ptr()),
            "mf_interpolate_voxel_grid_bwd")
        return gvox, None, None, None, None


def interpolate_voxel_grid(voxelized, points, batch_indices, channels_first=False, batch_start=None):
    """Returns [P, C] like the reference; ``channels_first=True`` returns [C, P]
    (the coalesced layout the pose network consumes, saving a transpose).  ``batch_start``
    (int32 [B+1] row offsets, optional) tells the kernel that the rows of item b are
    ``batch_start[b]:batch_start[b+1]`` (points sorted by batch index, as the pose network's
    are): each workgroup then visits only its own rows.

    Precondition with ``batch_start``: the points ARE sorted by batch item and the offsets agree with
    ``batch_indices`` (the kernels trust the offsets and do not look at ``batch_indices`` then); offsets are
    clamped to ``[0, P]`` on the device, so an inconsistent table cannot read or write out of bounds, but it
    gives wrong values.  Set ``MF_DEBUG_CHECKS=1`` to have the wrapper verify the table (one host sync)."""
    if batch_start is not None and os.environ.get("MF_DEBUG_CHECKS"):
        bs = batch_start.detach().cpu().long()
        bi = batch_indices.detach().cpu().long()
        ok = bool((bs[1:] >= bs[:-1]).all()) and int(bs[0]) >= 0 and int(bs[-1]) <= bi.shape[0]
        if ok:
            expect = torch.repeat_interleave(torch.arange(bs.shape[0] - 1), bs[1:] - bs[:-1])
            ok = torch.equal(bi[int(bs[0]):int(bs[-1])], expect)
        if not ok:
            raise ValueError("interpolate_voxel_grid: batch_start is inconsistent with batch_indices "
                             "(points must be sorted by batch item)")
    return InterpolateVoxelGrid.apply(voxelized, points, batch_indices, channels_first, batch_start)
