"""transformation_matrix(quaternion, translation) -> homogeneous 4x4 transform(s).

API of morefusion/functions/geometry/transformation_matrix.py:5-18: a batch ``[N,4]`` /
``[N,3]`` gives ``[N,4,4]``, a single ``[4]`` / ``[3]`` pair gives ``[4,4]``.
"""
from .compose_transform import compose_transform
from .quaternion_matrix import quaternion_matrix


def transformation_matrix(quaternion, translation):
    single = quaternion.ndim == 1
    q = quaternion[None] if single else quaternion
    t = translation[None] if single else translation
    if q.ndim != 2 or q.shape[1] != 4 or t.shape != (q.shape[0], 3):
        raise ValueError(
            f"expected quaternion [N,4] and translation [N,3] (or [4] and [3]), got "
            f"{tuple(quaternion.shape)} and {tuple(translation.shape)}")
    rotation = quaternion_matrix(q)[:, :3, :3]
    T = compose_transform(rotation, t)
    return T[0] if single else T


class _TransformationMatrixFused(__import__("torch").autograd.Function):
    """``transformation_matrix`` of a BATCH of poses as one HIP launch forward and one backward
    (``mf_transformation_matrix_fwd/_bwd``, csrc/pointops.hip): the 16000 predicted poses of a training step cost
    ~85 small torch launches through the composite above.  float32 CUDA tensors, q [N,4], t [N,3]."""

    @staticmethod
    def forward(ctx, q, t):
        import torch

        from ... import _lib
        _lib.require_gpu(q, t)
        q, t = _lib.f32c(q), _lib.f32c(t)
        n = q.shape[0]
        T = torch.empty((n, 4, 4), dtype=torch.float32, device=q.device)
        _lib.check(_lib.lib().mf_transformation_matrix_fwd(q.data_ptr(), t.data_ptr(), n, T.data_ptr(), _lib.stream_ptr()),
                   "mf_transformation_matrix_fwd")
        ctx.save_for_backward(q)
        return T

    @staticmethod
    def backward(ctx, gT):
        import torch

        from ... import _lib
        (q,) = ctx.saved_tensors
        gT = _lib.f32c(gT)
        n = q.shape[0]
        gq = torch.empty((n, 4), dtype=torch.float32, device=q.device)
        gt = torch.empty((n, 3), dtype=torch.float32, device=q.device)
        _lib.check(_lib.lib().mf_transformation_matrix_bwd(q.data_ptr(), gT.data_ptr(), n, gq.data_ptr(), gt.data_ptr(),
                                                           _lib.stream_ptr()), "mf_transformation_matrix_bwd")
        return gq, gt


def transformation_matrix_batch(quaternion, translation):
    """[N,4] / [N,3] float32 CUDA tensors -> [N,4,4] through the fused kernels; anything else takes the composite."""
    import torch
    if (quaternion.is_cuda and quaternion.dtype == torch.float32 and translation.dtype == torch.float32
            and quaternion.ndim == 2 and quaternion.shape[1] == 4 and translation.shape == (quaternion.shape[0], 3)):
        return _TransformationMatrixFused.apply(quaternion, translation)
    return transformation_matrix(quaternion, translation)
