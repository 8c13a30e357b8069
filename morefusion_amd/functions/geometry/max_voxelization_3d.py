"""max_voxelization_3d -- per voxel keep the features of the max-intensity point.

API of morefusion/functions/geometry/max_voxelization_3d.py:188-210; kernels
(:75-134, :153-183) replaced by ``mf_max_voxelization_3d_{fwd,bwd}``.
Deterministic: max intensity, lowest point index among ties (the reference CPU rule).
"""
import torch

from ... import _lib
from .voxelization_3d import check_dimensions, check_inputs


class MaxVoxelization3D(torch.autograd.Function):
    @staticmethod
    def forward(ctx, values, points, batch_indices, intensities, batch_size, origin, pitch,
                dimensions, check_nan):
        check_inputs(values, points, batch_indices)
        _lib.require_gpu(intensities)
        if intensities.dtype != torch.float32 or intensities.shape != (values.shape[0],):
            raise TypeError("intensities must be float32 [P]")
        X, Y, Z = check_dimensions(dimensions)
        ox, oy, oz = _lib.as_float3(origin)
        values_c, points_c = values.contiguous(), points.contiguous()
        bi_c, int_c = batch_indices.contiguous(), intensities.contiguous()
        n, C = values_c.shape
        dev = values.device
        matrix = torch.empty((batch_size, C, X, Y, Z), dtype=torch.float32, device=dev)
        indices = torch.empty((batch_size, X, Y, Z), dtype=torch.int32, device=dev)
        key = torch.empty((batch_size * X * Y * Z,), dtype=torch.int64, device=dev)
        nan_flag = torch.empty((1,), dtype=torch.int32, device=dev) if check_nan else None
        _lib.check(
            _lib.lib().mf_max_voxelization_3d_fwd(
                values_c.data_ptr(), points_c.data_ptr(), bi_c.data_ptr(), int_c.data_ptr(), n, C,
                batch_size, X, Y, Z, ox, oy, oz, float(pitch), matrix.data_ptr(),
                indices.data_ptr(), key.data_ptr(), _lib.ptr(nan_flag), _lib.stream_ptr()),
            "mf_max_voxelization_3d_fwd")
        if check_nan and int(nan_flag.item()):
            raise ValueError("points include nan")
        ctx.save_for_backward(indices)
        ctx.meta = (batch_size, n, (X, Y, Z))
        ctx.mark_non_differentiable(indices)
        return matrix, indices

    @staticmethod
    def backward(ctx, gmatrix, _gind):
        (indices,) = ctx.saved_tensors
        B, n, (X, Y, Z) = ctx.meta
        gmatrix = gmatrix.contiguous()
        C = gmatrix.shape[1]
        gvalues = torch.zeros((n, C), dtype=torch.float32, device=gmatrix.device)
        _lib.check(
            _lib.lib().mf_max_voxelization_3d_bwd(
                gmatrix.data_ptr(), indices.data_ptr(), n, C, B, X, Y, Z, gvalues.data_ptr(),
                _lib.stream_ptr()),
            "mf_max_voxelization_3d_bwd")
        return gvalues, None, None, None, None, None, None, None, None


def max_voxelization_3d(
    values, points, batch_indices, intensities, *, batch_size, origin, pitch, dimensions,
    return_indices=False, check_nan=True,
):
    voxelized, indices = MaxVoxelization3D.apply(
        values, points, batch_indices, intensities, batch_size, origin, pitch, dimensions,
        check_nan)
    if return_indices:
        return voxelized, indices
    return voxelized
