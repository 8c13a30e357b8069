"""The confidence terms of the DenseFusion pose loss and their reduction.

contrib/singleview_3d/models/model.py:417-434 of the reference: per object, over the points whose predicted confidence
is positive, the mean of ``add * conf - lambda * log(conf)``; the loss is the mean over the batch's objects (an object
without a confident point gives NaN, as the reference's mean of nothing does).  On the MI355X one launch forward and
one backward (csrc/loss.hip ``mf_confidence_loss_{fwd,bwd}``) in place of ~28 / ~20 elementwise and reduction
launches; CPU tensors take the torch composite.
"""
import torch

from ... import _lib


class _ConfidenceLoss(torch.autograd.Function):
    @staticmethod
    def forward(ctx, add, conf, lam):
        B, P = add.shape
        a, c = _lib.f32c(add), _lib.f32c(conf)
        loss = torch.empty((), dtype=torch.float32, device=a.device)
        cnt = torch.empty((B,), dtype=torch.int32, device=a.device)
        _lib.check(_lib.lib().mf_confidence_loss_fwd(a.data_ptr(), c.data_ptr(), B, P, float(lam), loss.data_ptr(),
                                                     cnt.data_ptr(), _lib.stream_ptr()), "mf_confidence_loss_fwd")
        ctx.save_for_backward(a, c, cnt)
        ctx.lam = float(lam)
        return loss

    @staticmethod
    def backward(ctx, gloss):
        a, c, cnt = ctx.saved_tensors
        B, P = a.shape
        g = _lib.f32c(gloss)
        dadd, dconf = torch.empty_like(a), torch.empty_like(c)
        _lib.check(_lib.lib().mf_confidence_loss_bwd(a.data_ptr(), c.data_ptr(), cnt.data_ptr(), g.data_ptr(), B, P,
                                                     ctx.lam, dadd.data_ptr(), dconf.data_ptr(), _lib.stream_ptr()),
                   "mf_confidence_loss_bwd")
        return dadd, dconf, None


def confidence_loss(add, confidence, lambda_confidence):
    """add, confidence [B, P] -> scalar."""
    if add.ndim != 2 or add.shape != confidence.shape:
        raise ValueError("add and confidence must be [B, P]")
    if add.is_cuda != confidence.is_cuda:
        raise RuntimeError("add and confidence must be on the same device")
    if not add.is_cuda:
        keep = confidence.detach() > 0
        conf = torch.where(keep, confidence, torch.ones_like(confidence))
        per_point = torch.where(keep, add * conf - lambda_confidence * torch.log(conf), torch.zeros_like(add))
        return (per_point.sum(dim=1) / keep.sum(dim=1)).sum() / add.shape[0]
    _lib.require_gpu(add, confidence)
    return _ConfidenceLoss.apply(add, confidence, lambda_confidence)
