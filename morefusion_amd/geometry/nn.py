"""nn(ref, query) -> index of the nearest ``ref`` row for every ``query`` row.

morefusion/geometry/knn/nn.py:12-55 + knn/cuComputeDistanceGlobal.cu:20-86.
The reference writes the full R x Q squared-distance matrix and runs cupy.argmin
over it; ``mf_nn`` keeps a running (min, arg-min) per query in registers.
Squared L2, ties -> lowest ref index.  Returns int64 [Q].
"""
import torch

from .. import _lib


def nn(ref, query, return_distance=False):
    _lib.require_gpu(ref, query)
    if ref.ndim != 2 or query.ndim != 2 or ref.shape[1] != 3 or query.shape[1] != 3:
        raise TypeError("ref and query must be [R, 3] and [Q, 3]")
    if ref.shape[0] == 0:
        raise ValueError("ref must not be empty")
    r, q = _lib.f32c(ref), _lib.f32c(query)
    out = torch.empty((q.shape[0],), dtype=torch.int64, device=q.device)
    dist = torch.empty((q.shape[0],), dtype=torch.float32, device=q.device) if return_distance else None
    _lib.check(
        _lib.lib().mf_nn(r.data_ptr(), r.shape[0], q.data_ptr(), q.shape[0], out.data_ptr(),
                         _lib.ptr(dist), _lib.stream_ptr()),
        "mf_nn")
    if return_distance:
        return out, dist
    return out
