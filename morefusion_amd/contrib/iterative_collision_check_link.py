"""IterativeCollisionCheckLink -- joint multi-object collision-based pose refinement.

API of morefusion/contrib/iterative_collision_check_link.py:9-99: parameters
``quaternion [N,4]`` (wxyz) and ``translation [N,3]``; calling the link returns the
scalar loss ``penalty - reward`` with gradients to both.  The ~300 launches of the
reference's forward+backward are three fused HIP kernels (csrc/icc.hip); ``refine()``
additionally runs the whole optimisation loop of
examples/ycb_video/pose_refinement/check_iterative_collision_check_link.py:44-79
on the device as one hipGraph.
"""
import numpy as np
import torch

from ..geometry.quaternion_from_matrix import quaternion_from_matrix, translation_from_matrix
from .icc_batch import IccScenes


class _IccLoss(torch.autograd.Function):
    @staticmethod
    def forward(ctx, quaternion, translation, scenes):
        loss, gq, gt = scenes.loss_grad(quaternion, translation)
        ctx.save_for_backward(gq, gt)
        return loss[0].clone()

    @staticmethod
    def backward(ctx, gloss):
        gq, gt = ctx.saved_tensors
        return gq * gloss, gt * gloss, None


class IterativeCollisionCheckLink(torch.nn.Module):
    def __init__(self, transform, voxel_dim=32, voxel_threshold=2, sdf_offset=0):
        super().__init__()
        self._voxel_dim = voxel_dim
        self._voxel_threshold = voxel_threshold
        self._sdf_offset = sdf_offset
        quaternion, translation = [], []
        for transform_i in transform:
            if isinstance(transform_i, torch.Tensor):
                transform_i = transform_i.detach().cpu().numpy()
            quaternion.append(quaternion_from_matrix(transform_i))
            translation.append(translation_from_matrix(transform_i))
        quaternion = np.stack(quaternion).astype(np.float32)
        translation = np.stack(translation).astype(np.float32)
        self.quaternion = torch.nn.Parameter(torch.from_numpy(quaternion))
        self.translation = torch.nn.Parameter(torch.from_numpy(translation))
        self._scenes = None
        self._scenes_key = None

    # -- chainer.Link conveniences used by the reference's call sites ------------------
    @property
    def xp(self):
        """``link.xp`` of the reference's call sites (``link.xp.asarray(points)``): arrays on this link's device."""
        from ..chainer_compat import link_xp
        return link_xp(self)

    def to_gpu(self, device=None):
        return self.to("cuda" if device is None else f"cuda:{device}")

    def zerograds(self):
        for p in self.parameters():
            p.grad = None

    def _pack(self, points, sdf, pitch, origin, grid_target, grid_nontarget_empty):
        # identity AND in-place version of every tensor argument: per-frame buffers updated in
        # place between calls must not hit a stale packed copy
        def ident(x):
            if isinstance(x, torch.Tensor):
                return (int(x.data_ptr()), int(x._version), tuple(x.shape))
            if isinstance(x, (list, tuple)):
                return tuple(ident(v) for v in x)
            return id(x)
        key = tuple(ident(x) for x in (points, sdf, pitch, origin, grid_target, grid_nontarget_empty))
        if self._scenes is None or self._scenes_key != key:
            self._scenes = IccScenes(
                [dict(points=points, sdf=sdf, pitch=pitch, origin=origin, grid_target=grid_target,
                      grid_nontarget_empty=grid_nontarget_empty)],
                voxel_dim=self._voxel_dim, voxel_threshold=self._voxel_threshold,
                sdf_offset=self._sdf_offset, device=self.quaternion.device)
            self._scenes_key = key
            self._hold = (points, sdf, pitch, origin, grid_target, grid_nontarget_empty)
        return self._scenes

    def forward(self, points, sdf, pitch, origin, grid_target, grid_nontarget_empty):
        if len(points) != self.quaternion.shape[0]:
            raise ValueError("number of point sets != number of poses")
        scenes = self._pack(points, sdf, pitch, origin, grid_target, grid_nontarget_empty)
        return _IccLoss.apply(self.quaternion, self.translation, scenes)

    @torch.no_grad()
    def refine(self, points, sdf, pitch, origin, grid_target, grid_nontarget_empty, n_iter=100,
               alpha=0.01, translation_alpha_scale=0.1, return_history=False):
        """The reference driver's loop (Adam(alpha), translation alpha x0.1, n_iter x
        {forward, backward, update}) fused on the device.  Updates the parameters in
        place; optionally returns (losses [n_iter], trajectory [n_iter,N,7])."""
        scenes = self._pack(points, sdf, pitch, origin, grid_target, grid_nontarget_empty)
        dev = self.quaternion.device
        N = self.quaternion.shape[0]
        if getattr(self, "_adam", None) is None:
            self._adam = [torch.zeros((N, 7), dtype=torch.float32, device=dev) for _ in range(2)]
            self._adam_t = 0
        q = self.quaternion.data.contiguous()
        t = self.translation.data.contiguous()
        losses = traj = None
        if return_history:
            losses = torch.empty((n_iter, 1), dtype=torch.float32, device=dev)
            traj = torch.empty((n_iter, N, 7), dtype=torch.float32, device=dev)
        scenes.refine(q, t, self._adam[0], self._adam[1], n_iter, step0=self._adam_t,
                      alpha_q=alpha, alpha_t=alpha * translation_alpha_scale, losses=losses,
                      traj=traj)
        self._adam_t += n_iter
        self.quaternion.data.copy_(q)
        self.translation.data.copy_(t)
        if return_history:
            return losses[:, 0], traj
