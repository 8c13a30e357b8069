"""truncated_distance_function, pseudo_occupancy_voxelization.

API of morefusion/functions/geometry/truncated_distance_function.py:169-213;
kernels (:44-93, :112-157) replaced by ``mf_truncated_distance_function_{fwd,bwd}``
and ``mf_pseudo_occupancy_weights`` (morefusion_amd/csrc/tdf.hip).
"""
import math

import numpy as np
import torch

from ... import _lib


def _ksize(pitch, truncation):
    # truncated_distance_function.py:36-38, evaluated in float32 like cupy.ceil(trunc/pitch)
    ks = int(math.ceil(float(np.float32(truncation) / np.float32(pitch))))
    return ks + 1 if ks % 2 == 0 else ks


class TruncatedDistanceFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, points, pitch, origin, dims, truncation):
        _lib.require_gpu(points)
        if points.ndim != 2 or points.shape[1] != 3:
            raise TypeError("points must be [P, 3]")
        if points.dtype != torch.float32:
            raise TypeError("points must be float32 (the HIP kernels are float32-only)")
        ox, oy, oz = _lib.as_float3(origin)
        X, Y, Z = (int(d) for d in dims)
        pts = points.contiguous()
        tdf = torch.empty((X, Y, Z), dtype=torch.float32, device=pts.device)
        flat = torch.empty((X, Y, Z), dtype=torch.int32, device=pts.device)
        _lib.check(
            _lib.lib().mf_truncated_distance_function_fwd(
                pts.data_ptr(), pts.shape[0], float(pitch), ox, oy, oz, X, Y, Z,
                float(truncation), tdf.data_ptr(), flat.data_ptr(), _lib.stream_ptr()),
            "mf_truncated_distance_function_fwd")
        ctx.save_for_backward(pts, flat)
        ctx.meta = (float(pitch), ox, oy, oz, X, Y, Z, float(truncation))
        ctx.mark_non_differentiable(flat)
        return tdf, flat

    @staticmethod
    def backward(ctx, gtdf, _gflat):
        pts, flat = ctx.saved_tensors
        pitch, ox, oy, oz, X, Y, Z, truncation = ctx.meta
        gtdf = gtdf.contiguous()
        gpoints = torch.zeros_like(pts)
        _lib.check(
            _lib.lib().mf_truncated_distance_function_bwd(
                gtdf.data_ptr(), pts.data_ptr(), flat.data_ptr(), pts.shape[0], pitch, ox, oy, oz,
                X, Y, Z, truncation, gpoints.data_ptr(), _lib.stream_ptr()),
            "mf_truncated_distance_function_bwd")
        return gpoints, None, None, None, None


def _scalar(x):
    return float(x.item()) if isinstance(x, torch.Tensor) else float(x)


def truncated_distance_function(points, *, pitch, origin, dims, truncation, return_indices=False):
    pitch, truncation = _scalar(pitch), _scalar(truncation)
    tdf, flat = TruncatedDistanceFunction.apply(points, pitch, origin, dims, truncation)
    if return_indices:
        K = _ksize(pitch, truncation) ** 3
        # func._indices // ksize**3 (:177); -1 stays -1 under floor division
        return tdf, torch.div(flat, K, rounding_mode="floor")
    return tdf


def pseudo_occupancy_voxelization(points, sdf, *, pitch, origin, dims, threshold=1, sdf_offset=0):
    """Returns (grid_uniform, grid_surface, grid_inside); only the TDF carries
    gradient, the sdf-derived weights are constants (:198-213)."""
    pitch = _scalar(pitch)
    truncation = float(np.float32(threshold) * np.float32(pitch))
    tdf, flat = TruncatedDistanceFunction.apply(points, pitch, origin, dims, truncation)
    _lib.require_gpu(sdf)
    X, Y, Z = tdf.shape
    K = _ksize(pitch, truncation) ** 3
    sdf_c = _lib.f32c(sdf)
    dev = tdf.device
    grids = torch.empty((3, X, Y, Z), dtype=torch.float32, device=dev)
    wsurf = torch.empty((X, Y, Z), dtype=torch.float32, device=dev)
    win = torch.empty((X, Y, Z), dtype=torch.float32, device=dev)
    wmax = torch.empty((1,), dtype=torch.float32, device=dev)
    _lib.check(
        _lib.lib().mf_pseudo_occupancy_weights(
            tdf.detach().data_ptr(), flat.data_ptr(), sdf_c.data_ptr(), X, Y, Z, K, truncation,
            float(sdf_offset), grids.data_ptr(), wsurf.data_ptr(), win.data_ptr(),
            wmax.data_ptr(), _lib.stream_ptr()),
        "mf_pseudo_occupancy_weights")
    if tdf.requires_grad:
        grid = 1 - tdf / truncation  # differentiable replay of :195
        return grid, grid * wsurf, grid * win
    return grids[0], grids[1], grids[2]
