"""average_voxelization_3d -- mean of point features per voxel.

API of morefusion/functions/geometry/average_voxelization_3d.py:223-244; the
forward/backward kernels it launched (:42-118, :146-220) are replaced by
``mf_average_voxelization_3d_{fwd,bwd}`` (morefusion_amd/csrc/voxelize.hip).
"""
import torch

from ... import _lib
from . import _cpu
from .voxelization_3d import check_dimensions, check_inputs


class AverageVoxelization3D(torch.autograd.Function):
    """(values [P,C], points [P,3], batch_indices [P]) -> matrix [B,C,X,Y,Z].

    ``counts`` [B,X,Y,Z] int32 is returned as a second, non-differentiable output.
    No gradient flows to ``points`` (reference: ``return gvalues, None, None``)."""

    @staticmethod
    def forward(ctx, values, points, batch_indices, batch_size, origin, pitch, dimensions,
                check_nan):
        check_inputs(values, points, batch_indices)
        X, Y, Z = check_dimensions(dimensions)
        ox, oy, oz = _lib.as_float3(origin)
        values_c, points_c, bi_c = values.contiguous(), points.contiguous(), batch_indices.contiguous()
        n, C = values_c.shape
        dev = values.device
        matrix = torch.empty((batch_size, C, X, Y, Z), dtype=torch.float32, device=dev)
        counts = torch.empty((batch_size, X, Y, Z), dtype=torch.int32, device=dev)
        head = torch.empty((batch_size * X * Y * Z,), dtype=torch.int32, device=dev)
        link = torch.empty((max(n, 1),), dtype=torch.int32, device=dev)
        nan_flag = torch.empty((1,), dtype=torch.int32, device=dev) if check_nan else None
        _lib.check(
            _lib.lib().mf_average_voxelization_3d_fwd(
                values_c.data_ptr(), points_c.data_ptr(), bi_c.data_ptr(), n, C, batch_size,
                X, Y, Z, ox, oy, oz, float(pitch), matrix.data_ptr(), counts.data_ptr(),
                head.data_ptr(), link.data_ptr(), _lib.ptr(nan_flag), _lib.stream_ptr()),
            "mf_average_voxelization_3d_fwd")
        if check_nan and int(nan_flag.item()):  # host sync, exactly like the reference's
            raise ValueError("points include nan")  # `if cupy.isnan(points).sum()` (:47-48)
        ctx.save_for_backward(points_c, bi_c, counts)
        ctx.meta = (batch_size, ox, oy, oz, float(pitch), (X, Y, Z))
        ctx.mark_non_differentiable(counts)
        return matrix, counts

    @staticmethod
    def backward(ctx, gmatrix, _gcounts):
        points, bi, counts = ctx.saved_tensors
        B, ox, oy, oz, pitch, (X, Y, Z) = ctx.meta
        gmatrix = gmatrix.contiguous()
        n, C = points.shape[0], gmatrix.shape[1]
        gvalues = torch.empty((n, C), dtype=torch.float32, device=gmatrix.device)
        _lib.check(
            _lib.lib().mf_average_voxelization_3d_bwd(
                gmatrix.data_ptr(), points.data_ptr(), bi.data_ptr(), counts.data_ptr(), n, C,
                B, X, Y, Z, ox, oy, oz, pitch, gvalues.data_ptr(), _lib.stream_ptr()),
            "mf_average_voxelization_3d_bwd")
        return gvalues, None, None, None, None, None, None, None


def average_voxelization_3d(
    values, points, batch_indices, *, batch_size, origin, pitch, dimensions,
    return_counts=False, check_nan=True,
):
    """``check_nan=False`` skips the reference's NaN validation (and its host sync).  NumPy arrays / CPU tensors take
    the product's CPU path (``_cpu.average_voxelization_3d`` = ``forward_cpu`` / ``backward_cpu``,
    average_voxelization_3d.py:8-40,120-145: indices round half to even there), CUDA tensors the HIP kernels."""
    if _cpu.is_cpu_input(values, points, batch_indices):
        return _cpu.average_voxelization_3d(values, points, batch_indices, batch_size=batch_size, origin=origin,
                                            pitch=pitch, dimensions=dimensions, return_counts=return_counts)
    voxel, counts = AverageVoxelization3D.apply(
        values, points, batch_indices, batch_size, origin, pitch, dimensions, check_nan)
    if return_counts:
        return voxel, counts
    return voxel
