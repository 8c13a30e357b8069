"""Product-owned CPU path of the operator surface (BASELINE config 1: "1k-point cloud -> 32^3
occupancy_grid_3d on CPU/NumPy path").

The reference chooses the device from the array type (``cuda.get_array_module``:
morefusion/functions/geometry/occupancy_grid_3d.py:32; ``forward_cpu`` / ``forward_gpu``:
average_voxelization_3d.py:8-40).  The wrappers of this package do the same: a NumPy array or a CPU
tensor takes the implementations below -- float32 torch-CPU expressions in the order of the
reference's NumPy code, so the results equal its CPU fork bit for bit (voxel indices round half to
even like ``ndarray.round``; the GPU fork rounds half away) -- a CUDA tensor takes the HIP kernels.
This module is part of the product and self-contained: the test-only CPU restatements of the repository are
never imported from here (tests/test_cabi.py enforces it).
"""
import numpy as np
import torch


def is_cpu_input(*xs):
    """True when the call is a CPU call: every array argument is a NumPy array or a CPU tensor."""
    seen = False
    for x in xs:
        if isinstance(x, np.ndarray):
            seen = True
        elif isinstance(x, torch.Tensor):
            if x.is_cuda:
                return False
            seen = True
    return seen


def as_tensor(x):
    """NumPy array -> CPU tensor sharing its memory; tensors pass through."""
    if isinstance(x, np.ndarray):
        return torch.from_numpy(np.ascontiguousarray(x))
    return x


def _float3(v):
    a = np.asarray(v.detach().cpu().numpy() if isinstance(v, torch.Tensor) else v, dtype=np.float32).reshape(-1)
    if a.shape != (3,):
        raise ValueError("origin must have 3 components")
    return torch.from_numpy(a.copy())


class _Sqrt(torch.autograd.Function):
    """Correctly rounded float32 square root (NumPy's: the IEEE instruction).  ``torch.sqrt`` on the CPU goes through
    a vector math library that is off by one ulp in ~0.7 % of float32 inputs -- enough to break bit-equality with
    the reference's NumPy path."""

    @staticmethod
    def forward(ctx, x):
        y = torch.from_numpy(np.sqrt(x.detach().numpy()))
        ctx.save_for_backward(y)
        return y

    @staticmethod
    def backward(ctx, g):
        (y,) = ctx.saved_tensors
        return g / (2 * y)


# ---- occupancy_grid_3d (occupancy_grid_3d.py:31-85) -------------------------------------------
def occupancy_grid_3d(points, *, pitch, origin, dims, threshold=1, x_chunk=4):
    """min(relu(threshold - min_p |v - p_f|), 1) over all voxels v of an [X,Y,Z] grid.

    The reference broadcasts three [X,Y,Z,P] float32 tensors (393 MB at 32^3 x 1000 points); here the
    same expressions run over ``x_chunk`` x-planes at a time (16 MB).  Differentiable w.r.t. ``points``
    through torch autograd (the gradient reaches the arg-min point of each voxel)."""
    points = as_tensor(points)
    if points.dtype != torch.float32 or points.ndim != 2 or points.shape[1] != 3:
        raise TypeError("points must be float32 [P, 3]")  # check_type_forward (:21-29)
    X, Y, Z = (int(d) for d in dims)
    o = _float3(origin)
    p = torch.tensor(float(np.float32(pitch)), dtype=torch.float32)
    pf = (points - o) / p  # a coordinate -> voxel coordinate (:42)
    px, py, pz = pf[:, 0], pf[:, 1], pf[:, 2]
    J = torch.arange(Y, dtype=torch.float32).view(1, Y, 1, 1)
    K = torch.arange(Z, dtype=torch.float32).view(1, 1, Z, 1)
    out = []
    for x0 in range(0, X, x_chunk):
        I = torch.arange(x0, min(x0 + x_chunk, X), dtype=torch.float32).view(-1, 1, 1, 1)
        d_IP, d_JP, d_KP = I - px, J - py, K - pz
        d = _Sqrt.apply(d_IP ** 2 + d_JP ** 2 + d_KP ** 2)  # (:81)
        d_min = d.min(dim=3).values
        m = torch.relu(threshold - d_min)
        out.append(torch.minimum(m, torch.ones_like(m)))
    return torch.cat(out, 0)


# ---- average_voxelization_3d (average_voxelization_3d.py:8-40, :120-145) -------------------------
def _voxel_index_cpu(points, origin, pitch, dimensions):
    """forward_cpu's ``((point - origin) / pitch).round().astype(int)`` and its validity test (:29-30)."""
    o = _float3(origin)
    q = (points - o) / torch.tensor(float(np.float32(pitch)), dtype=torch.float32)
    idx = torch.round(q).to(torch.int64)  # half to even, like ndarray.round
    dims = torch.tensor(dimensions, dtype=torch.int64)
    valid = ((idx >= 0) & (idx < dims)).all(dim=1)
    return idx, valid


class _AverageVoxelization3DCPU(torch.autograd.Function):
    @staticmethod
    def forward(ctx, values, points, batch_indices, batch_size, origin, pitch, dimensions):
        if torch.isnan(points).any():
            raise ValueError("points include nan")  # (:13-14)
        B, (X, Y, Z), C = int(batch_size), dimensions, values.shape[1]
        idx, valid = _voxel_index_cpu(points, origin, pitch, dimensions)
        V = X * Y * Z
        flat = (batch_indices.to(torch.int64) * V + (idx[:, 0] * Y + idx[:, 1]) * Z + idx[:, 2])[valid]
        vals = values[valid]
        counts = torch.bincount(flat, minlength=B * V)
        # sum of the point rows of every voxel IN POINT ORDER (the loop of :24-34): the r-th point of each
        # voxel is added in round r -- vectorised, the same float32 additions in the same order
        order = torch.argsort(flat, stable=True)
        sflat = flat[order]
        start = torch.cumsum(counts, 0) - counts
        rank = torch.arange(sflat.numel()) - start[sflat]
        acc = torch.zeros((B * V, C), dtype=torch.float32)
        for r in range(int(rank.max()) + 1 if rank.numel() else 0):
            sel = rank == r
            acc[sflat[sel]] += vals[order[sel]]
        nz = counts > 0
        acc[nz] /= counts[nz].to(torch.float32)[:, None]  # (:36-37)
        matrix = acc.view(B, X, Y, Z, C).permute(0, 4, 1, 2, 3).contiguous()
        counts = counts.view(B, X, Y, Z).to(torch.int32)
        ctx.save_for_backward(flat, valid, counts)
        ctx.shape = (values.shape[0], C, B, V)
        ctx.mark_non_differentiable(counts)
        return matrix, counts

    @staticmethod
    def backward(ctx, gmatrix, _gcounts):
        flat, valid, counts = ctx.saved_tensors
        P, C, B, V = ctx.shape
        g = gmatrix.permute(0, 2, 3, 4, 1).reshape(B * V, C)
        gvalues = torch.zeros((P, C), dtype=torch.float32)
        gvalues[valid] = g[flat] / counts.view(-1)[flat].to(torch.float32)[:, None]  # (:138-143)
        return gvalues, None, None, None, None, None, None


def average_voxelization_3d(values, points, batch_indices, *, batch_size, origin, pitch, dimensions,
                            return_counts=False):
    from .voxelization_3d import check_dimensions

    dimensions = check_dimensions(dimensions)
    values, points, batch_indices = as_tensor(values), as_tensor(points), as_tensor(batch_indices)
    if values.dtype != torch.float32 or values.ndim != 2:
        raise TypeError("values must be float32 [P, C]")
    if points.dtype != torch.float32 or points.ndim != 2 or points.shape[1] != 3:
        raise TypeError("points must be float32 [P, 3]")
    if points.shape[0] != values.shape[0]:
        raise TypeError("points and values must have the same length")
    if batch_indices.dtype != torch.int32 or batch_indices.ndim != 1 or batch_indices.shape[0] != values.shape[0]:
        raise TypeError("batch_indices must be int32 [P]")
    matrix, counts = _AverageVoxelization3DCPU.apply(values, points, batch_indices, batch_size, origin, pitch, dimensions)
    return (matrix, counts) if return_counts else matrix
