"""IterativeClosestPointLink -- ICP as a differentiable loss (single object).

API of morefusion/contrib/iterative_closest_point_link.py:9-44.  The reference
materialises the [T,S,3] difference tensor; ``mf_icp_loss_grad`` streams the
transformed source through LDS, keeps the arg-min per target point in registers and
reduces the matched squared distances and their pose-gradient moments in one launch.
"""
import numpy as np
import torch

from .. import _lib
from .. import functions as functions_module
from ..geometry.quaternion_from_matrix import quaternion_from_matrix, translation_from_matrix


class _IcpLoss(torch.autograd.Function):
    @staticmethod
    def forward(ctx, R, t, source, target, thresh):
        _lib.require_gpu(R, t, source, target)
        Rt = torch.cat([R.reshape(9), t.reshape(3)]).to(torch.float32).contiguous()
        src, tgt = _lib.f32c(source), _lib.f32c(target)
        out = torch.zeros((16,), dtype=torch.float32, device=Rt.device)
        _lib.check(
            _lib.lib().mf_icp_loss_grad(src.data_ptr(), src.shape[0], tgt.data_ptr(), tgt.shape[0],
                                        Rt.data_ptr(), float(thresh), out.data_ptr(),
                                        _lib.stream_ptr()),
            "mf_icp_loss_grad")
        ctx.save_for_backward(out)
        return out[0].clone()

    @staticmethod
    def backward(ctx, g):
        (out,) = ctx.saved_tensors
        return out[4:13].reshape(3, 3) * g, out[13:16] * g, None, None, None


class IterativeClosestPointLink(torch.nn.Module):
    def __init__(self, transform):
        super().__init__()
        if isinstance(transform, torch.Tensor):
            transform = transform.detach().cpu().numpy()
        quaternion = quaternion_from_matrix(transform).astype(np.float32)
        translation = translation_from_matrix(transform).astype(np.float32)
        self.quaternion = torch.nn.Parameter(torch.from_numpy(quaternion))
        self.translation = torch.nn.Parameter(torch.from_numpy(translation))

    def to_gpu(self, device=None):
        return self.to("cuda" if device is None else f"cuda:{device}")

    def zerograds(self):
        for p in self.parameters():
            p.grad = None

    @property
    def T(self):
        return functions_module.transformation_matrix(self.quaternion, self.translation)

    def forward(self, source, target):
        # source: from cad, target: from depth;  keep = squared distance < 0.02 (:38)
        T = self.T
        return _IcpLoss.apply(T[:3, :3], T[:3, 3], source, target, 0.02)
