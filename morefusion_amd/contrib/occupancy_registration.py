"""OccupancyRegistration -- single-object pose registration against occupancy grids.

API of morefusion/contrib/occupancy_registration.py:10-139: the
predecessor of ICC.  The soft occupancy grid of the transformed source points is rewarded
for overlapping ``grid_target[0]`` (occupied) and penalised for overlapping
``grid_target[1]`` (or ``max(grid_target[1], grid_target[2])``).  The reference builds
the grid from three dense [X,Y,Z,P] tensors per iteration; here it is the fused HIP
``occupancy_grid_3d`` (csrc/occgrid_knn.hip), so one iteration is a handful of launches.
"""
import numpy as np
import torch

from .. import functions as functions_module
from .. import geometry as geometry_module
from ..geometry.quaternion_from_matrix import quaternion_from_matrix, translation_from_matrix
from ..optimizers import Adam


class OccupancyRegistrationLink(torch.nn.Module):
    def __init__(self, quaternion_init=None, translation_init=None):
        super().__init__()
        if quaternion_init is None:
            quaternion_init = np.array([1, 0, 0, 0], dtype=np.float32)
        if translation_init is None:
            translation_init = np.zeros((3,), dtype=np.float32)
        self.quaternion = torch.nn.Parameter(torch.as_tensor(quaternion_init, dtype=torch.float32))
        self.translation = torch.nn.Parameter(torch.as_tensor(translation_init, dtype=torch.float32))

    @property
    def xp(self):
        """``link.xp`` of the reference's call sites (``link.xp.asarray(points)``): arrays on this link's device."""
        from ..chainer_compat import link_xp
        return link_xp(self)

    def to_gpu(self, device=None):
        return self.to("cuda" if device is None else f"cuda:{device}")

    def cleargrads(self):
        for p in self.parameters():
            p.grad = None

    def forward(self, points_source, grid_target, *, pitch, origin, threshold):
        if grid_target.dtype != torch.float32 or grid_target.shape[0] not in (2, 3):
            raise TypeError("grid_target must be float32 [2|3, X, Y, Z]")
        transform = functions_module.transformation_matrix(self.quaternion, self.translation)
        moved = functions_module.transform_points(points_source, transform)
        grid_source = functions_module.occupancy_grid_3d(
            moved, pitch=pitch, origin=origin, dims=tuple(grid_target.shape[1:]), threshold=threshold)
        occupied = grid_target[0]
        reward = (occupied * grid_source).sum() / occupied.sum()
        unoccupied = grid_target[1] if grid_target.shape[0] == 2 else torch.maximum(grid_target[1], grid_target[2])
        penalty = (unoccupied * grid_source).sum() / grid_source.sum()
        return penalty - reward


class OccupancyRegistration:
    def __init__(self, points_source, grid_target, *, pitch, origin, threshold, transform_init,
                 gpu=0, alpha=0.1):
        if gpu < 0:
            raise RuntimeError("OccupancyRegistration runs on the MI355X (gpu >= 0)")
        transform_init = np.asarray(transform_init)
        link = OccupancyRegistrationLink(quaternion_from_matrix(transform_init).astype(np.float32),
                                         translation_from_matrix(transform_init).astype(np.float32))
        link.to_gpu(gpu)
        dev = link.quaternion.device
        self._points_source = torch.as_tensor(points_source, dtype=torch.float32).to(dev)
        self._grid_target = torch.as_tensor(grid_target, dtype=torch.float32).to(dev)
        self._pitch, self._origin, self._threshold = pitch, origin, threshold
        self._optimizer = Adam(alpha=alpha).setup(link)
        link.translation.update_rule.hyperparam.alpha *= 0.1

    @property
    def _transform(self):
        link = self._optimizer.target
        with torch.no_grad():
            R = functions_module.quaternion_matrix(link.quaternion)[:3, :3]
            return geometry_module.compose_transform(R, link.translation).cpu().numpy()

    def register_iterative(self, iteration=None):
        iteration = 100 if iteration is None else iteration
        yield self._transform
        link = self._optimizer.target
        for _ in range(iteration):
            loss = link(points_source=self._points_source, grid_target=self._grid_target,
                        pitch=self._pitch, origin=self._origin, threshold=self._threshold)
            loss.backward()
            self._optimizer.update()
            link.cleargrads()
            yield self._transform

    def register(self, iteration=None):
        for _ in self.register_iterative(iteration=iteration):
            pass
        return self._transform
