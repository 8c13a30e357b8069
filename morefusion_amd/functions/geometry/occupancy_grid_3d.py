"""occupancy_grid_3d -- soft occupancy min(relu(thr - min_p |v - p_f|), 1).

API of morefusion/functions/geometry/occupancy_grid_3d.py:77-85.  The reference
materialises three [X,Y,Z,P] tensors; here a single fused min-over-points kernel
(``mf_occupancy_grid_3d_{fwd,bwd}``, morefusion_amd/csrc/occgrid_knn.hip) for CUDA tensors, and the
reference's own expressions in x-plane chunks for NumPy arrays / CPU tensors (``_cpu.py``).
"""
import torch

from ... import _lib
from . import _cpu


class OccupancyGrid3D(torch.autograd.Function):
    @staticmethod
    def forward(ctx, points, pitch, origin, dims, threshold):
        _lib.require_gpu(points)
        if points.dtype != torch.float32 or points.ndim != 2 or points.shape[1] != 3:
            raise TypeError("points must be float32 [P, 3]")
        ox, oy, oz = _lib.as_float3(origin)
        X, Y, Z = (int(d) for d in dims)
        pts = points.contiguous()
        grid = torch.empty((X, Y, Z), dtype=torch.float32, device=pts.device)
        dmin = torch.empty((X, Y, Z), dtype=torch.float32, device=pts.device)
        _lib.check(
            _lib.lib().mf_occupancy_grid_3d_fwd(
                pts.data_ptr(), pts.shape[0], float(pitch), ox, oy, oz, X, Y, Z,
                float(threshold), grid.data_ptr(), dmin.data_ptr(), _lib.stream_ptr()),
            "mf_occupancy_grid_3d_fwd")
        ctx.save_for_backward(pts, dmin)
        ctx.meta = (float(pitch), ox, oy, oz, X, Y, Z, float(threshold))
        return grid

    @staticmethod
    def backward(ctx, ggrid):
        pts, dmin = ctx.saved_tensors
        pitch, ox, oy, oz, X, Y, Z, threshold = ctx.meta
        ggrid = ggrid.contiguous()
        gpoints = torch.zeros_like(pts)
        _lib.check(
            _lib.lib().mf_occupancy_grid_3d_bwd(
                ggrid.data_ptr(), pts.data_ptr(), pts.shape[0], pitch, ox, oy, oz, X, Y, Z,
                threshold, dmin.data_ptr(), gpoints.data_ptr(), _lib.stream_ptr()),
            "mf_occupancy_grid_3d_bwd")
        return gpoints, None, None, None, None


def occupancy_grid_3d(points, *, pitch, origin, dims, threshold=1):
    """Device chosen by the array type, like the reference's ``get_array_module`` (occupancy_grid_3d.py:32): a NumPy
    array or a CPU tensor takes the product's CPU path (``_cpu.occupancy_grid_3d``: the reference's float32
    expressions, BASELINE config 1), a CUDA tensor the fused HIP kernel."""
    if _cpu.is_cpu_input(points):
        return _cpu.occupancy_grid_3d(points, pitch=pitch, origin=origin, dims=dims, threshold=threshold)
    return OccupancyGrid3D.apply(points, pitch, origin, dims, threshold)
