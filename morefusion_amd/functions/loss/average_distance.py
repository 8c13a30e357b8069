"""average_distance -- ADD / ADD-S pose loss.

API of morefusion/functions/loss/average_distance.py:40-85: ``points [M,3]``,
``transform_true [4,4]``, ``transforms_pred [P,4,4]`` -> ``[P]`` mean distances; ``symmetric``
switches to ADD-S (nearest true point per predicted point).  On the MI355X the whole composite
is ONE fused kernel per direction (csrc/loss.hip, ``mf_average_distance_{fwd,bwd}``): nothing of
size P x M is materialised and the ADD-S search runs out of LDS.  ``average_distance_batch`` is
the same op over B objects at once -- what ``Model.loss`` uses instead of the reference's
per-object Python loop (contrib/singleview_3d/models/model.py:406-434).

Gradients flow to ``transforms_pred`` only: the true pose and the model points are data in
every caller of the reference.  CPU tensors take a plain torch composite (ADD only).
"""
import torch

from ... import _lib


class _AverageDistance(torch.autograd.Function):
    @staticmethod
    def forward(ctx, points, T_true, T_pred, symmetric):
        B, M = points.shape[0], points.shape[1]
        P = T_pred.shape[1]
        pts, Tt, Tp = _lib.f32c(points), _lib.f32c(T_true), _lib.f32c(T_pred)
        out = torch.empty((B, P), dtype=torch.float32, device=Tp.device)
        need_grad = T_pred.requires_grad
        any_sym = symmetric is not None
        idx = (torch.empty((B, P, M), dtype=torch.int32, device=Tp.device)
               if (need_grad and any_sym) else None)
        _lib.check(_lib.lib().mf_average_distance_fwd(
            pts.data_ptr(), Tt.data_ptr(), Tp.data_ptr(), _lib.ptr(symmetric), B, M, P,
            out.data_ptr(), _lib.ptr(idx), _lib.stream_ptr()), "mf_average_distance_fwd")
        ctx.save_for_backward(pts, Tt, Tp)
        ctx.extra = (symmetric, idx)
        return out

    @staticmethod
    def backward(ctx, gout):
        pts, Tt, Tp = ctx.saved_tensors
        symmetric, idx = ctx.extra
        B, M, P = pts.shape[0], pts.shape[1], Tp.shape[1]
        g = _lib.f32c(gout)
        gT = torch.empty_like(Tp)
        _lib.check(_lib.lib().mf_average_distance_bwd(
            pts.data_ptr(), Tt.data_ptr(), Tp.data_ptr(), _lib.ptr(symmetric), g.data_ptr(), B, M, P,
            _lib.ptr(idx), gT.data_ptr(), _lib.stream_ptr()), "mf_average_distance_bwd")
        return None, None, gT, None


def _check(points, transform_true, transforms_pred):
    if points.ndim != 3 or points.shape[2] != 3:
        raise ValueError("points must be [B, M, 3]")
    B = points.shape[0]
    if transform_true.shape != (B, 4, 4):
        raise ValueError("transform_true must be [B, 4, 4]")
    if transforms_pred.ndim != 4 or transforms_pred.shape[0] != B or transforms_pred.shape[2:] != (4, 4):
        raise ValueError("transforms_pred must be [B, P, 4, 4]")


def average_distance_batch(points, transform_true, transforms_pred, symmetric=None):
    """points [B,M,3], transform_true [B,4,4], transforms_pred [B,P,4,4], symmetric [B] bool
    (or None) -> [B,P]."""
    _check(points, transform_true, transforms_pred)
    if not transforms_pred.is_cuda:
        if symmetric is not None and bool(torch.as_tensor(symmetric).any()):
            raise RuntimeError("ADD-S needs the HIP nearest-neighbour search: move the tensors to 'cuda' "
                               "(there is no CPU fallback)")
        true = torch.einsum("bij,bmj->bmi", transform_true[:, :3, :3], points) + transform_true[:, None, :3, 3]
        pred = (torch.einsum("bpij,bmj->bpmi", transforms_pred[:, :, :3, :3], points)
                + transforms_pred[:, :, None, :3, 3])
        return torch.linalg.vector_norm(true[:, None] - pred, dim=3).mean(dim=2)
    _lib.require_gpu(points, transform_true, transforms_pred)
    sym = None
    if symmetric is not None:
        sym = torch.as_tensor(symmetric, device=transforms_pred.device).to(torch.uint8).contiguous()
        if sym.shape != (points.shape[0],):
            raise ValueError("symmetric must be [B]")
    return _AverageDistance.apply(points, transform_true, transforms_pred, sym)


def average_distance(points, transform_true, transforms_pred, symmetric=False):
    if points.ndim != 2 or transform_true.shape != (4, 4) or transforms_pred.ndim != 3:
        raise ValueError("expected points [M,3], transform_true [4,4], transforms_pred [P,4,4]")
    sym = None
    if symmetric:
        sym = torch.ones(1, dtype=torch.bool, device=transforms_pred.device)
    return average_distance_batch(points[None], transform_true[None], transforms_pred[None], sym)[0]
