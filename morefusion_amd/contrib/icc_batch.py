"""Device-resident batch of ICC scenes + the fused loss / refinement entry points.

Packs the reference's argument lists
(``link(points, sdf, pitch, origin, grid_target, grid_nontarget_empty)``,
morefusion/contrib/iterative_collision_check_link.py:31-33) for one or many independent
scenes into the flat device arrays ``mfIccBatch`` describes (include/mfhip.h) and
owns the workspace of ``mf_icc_loss_grad`` / ``mf_icc_refine``.
"""
import ctypes
import os

import torch

from .. import _lib


def _stack(x, dtype, device):
    if isinstance(x, torch.Tensor):
        return x.detach().to(device=device, dtype=dtype).contiguous()
    return torch.stack([torch.as_tensor(v) for v in x]).to(device=device, dtype=dtype).contiguous()


class IccScenes:
    """``scenes``: list of dicts with the reference's per-scene arguments
    (points: list of [P_i,3]; sdf: list of [P_i]; pitch [N]; origin [N,3];
    grid_target, grid_nontarget_empty [N,D,D,D])."""

    def __init__(self, scenes, voxel_dim=32, voxel_threshold=2, sdf_offset=0.0, device="cuda", single_pass=None):
        """``single_pass``: None = choose by the no-entry grids' values (binary -> the single-pass kernel);
        False forces the general two-kernel iteration (k_icc_tile -> W -> k_icc_accum) on any grids; True is
        honoured only for {0,1} grids (the polynomial form of the loss holds for those alone)."""
        dev = torch.device(device)
        self._single_pass = single_pass
        if dev.type != "cuda":
            raise RuntimeError("IccScenes lives on the MI355X (device must be 'cuda')")
        pts, sdf, obj_off, scene_off, obj_scene = [], [], [0], [0], []
        pitch, origin, gt, gne = [], [], [], []
        for s, sc in enumerate(scenes):
            n = len(sc["points"])
            if len(sc["sdf"]) != n:
                raise ValueError("points and sdf lists differ in length")
            for i in range(n):
                p = torch.as_tensor(sc["points"][i]).to(dev, torch.float32)
                d = torch.as_tensor(sc["sdf"][i]).to(dev, torch.float32)
                if p.ndim != 2 or p.shape[1] != 3 or d.shape != (p.shape[0],):
                    raise TypeError("points[i] must be [P,3] and sdf[i] [P]")
                pts.append(p)
                sdf.append(d)
                obj_off.append(obj_off[-1] + p.shape[0])
                obj_scene.append(s)
            scene_off.append(scene_off[-1] + n)
            pitch.append(_stack(sc["pitch"], torch.float32, dev).reshape(n))
            origin.append(_stack(sc["origin"], torch.float32, dev).reshape(n, 3))
            gt.append(_stack(sc["grid_target"], torch.float32, dev).reshape(n, voxel_dim, voxel_dim, voxel_dim))
            gne.append(_stack(sc["grid_nontarget_empty"], torch.float32, dev).reshape(n, voxel_dim, voxel_dim, voxel_dim))
        self.device = dev
        self.n_objects, self.n_scenes, self.n_points = len(obj_scene), len(scenes), obj_off[-1]
        self.dim = voxel_dim
        points = torch.cat(pts, 0).contiguous()
        sdf_all = torch.cat(sdf, 0).contiguous()
        self.pts4 = torch.empty((self.n_points, 4), dtype=torch.float32, device=dev)
        L = _lib.lib()
        _lib.check(L.mf_pack_points_sdf(points.data_ptr(), sdf_all.data_ptr(), self.n_points,
                                        self.pts4.data_ptr(), _lib.stream_ptr()),
                   "mf_pack_points_sdf")
        self.obj_off = torch.tensor(obj_off, dtype=torch.int32, device=dev)
        self.scene_off = torch.tensor(scene_off, dtype=torch.int32, device=dev)
        self.obj_scene = torch.tensor(obj_scene, dtype=torch.int32, device=dev)
        self.scene_off_host = scene_off
        self.pitch = torch.cat(pitch).contiguous()
        self.origin = torch.cat(origin).contiguous()
        self.grid_target = torch.cat(gt).contiguous()
        self.grid_ne = torch.cat(gne).contiguous()
        self.desc = _lib.IccBatch(
            self.pts4.data_ptr(), self.obj_off.data_ptr(), self.scene_off.data_ptr(),
            self.obj_scene.data_ptr(), self.pitch.data_ptr(), self.origin.data_ptr(),
            self.grid_target.data_ptr(), self.grid_ne.data_ptr(), self.n_objects, self.n_scenes,
            self.n_points, voxel_dim,
            max(scene_off[i + 1] - scene_off[i] for i in range(len(scenes))),
            float(voxel_threshold), float(sdf_offset),
            # {0,1} no-entry grids (bool cast to float32, what every caller of the reference passes)
            # take the single-pass kernel; checked once here (pack time, not in the loop)
            int(bool(((self.grid_ne == 0) | (self.grid_ne == 1)).all())))
        self.desc.flags = 0  # (reserved)
        nbytes = L.mf_icc_workspace_bytes(ctypes.byref(self.desc))
        if nbytes < 0:
            raise ValueError("mf_icc: invalid batch descriptor (objects per scene <= 128 with {0,1} no-entry grids, <= 64 "
                             "with other values; dim <= 64)")
        self.ws = torch.empty((nbytes,), dtype=torch.uint8, device=dev)
        self.prepare()

    def prepare(self):
        """(Re-)derive what depends on the point / grid arrays: bounding spheres, sum(grid_target),
        tables, and whether ``grid_ne`` is {0,1}-valued (selects the single-pass kernel; other values,
        e.g. OctoMap probabilities, take the two-kernel path).  Call again after updating ``pts4`` /
        ``grid_target`` / ``grid_ne`` / ``pitch`` / ``origin`` in place (one host sync for the flag)."""
        self.desc.grid_ne_binary = int(bool(((self.grid_ne == 0) | (self.grid_ne == 1)).all()))
        if self._single_pass is False:
            self.desc.grid_ne_binary = 0
        elif self._single_pass and not self.desc.grid_ne_binary:
            raise ValueError("single_pass=True needs {0,1}-valued no-entry grids")
        _lib.check(_lib.lib().mf_icc_prepare(ctypes.byref(self.desc), self.ws.data_ptr(), _lib.stream_ptr()),
                   "mf_icc_prepare")

    def loss_grad(self, q, t):
        """q [O,4], t [O,3] float32 cuda -> (loss [S], gq [O,4], gt [O,3])."""
        _lib.require_gpu(q, t)
        q, t = _lib.f32c(q), _lib.f32c(t)
        loss = torch.empty((self.n_scenes,), dtype=torch.float32, device=self.device)
        gq = torch.empty((self.n_objects, 4), dtype=torch.float32, device=self.device)
        gt = torch.empty((self.n_objects, 3), dtype=torch.float32, device=self.device)
        _lib.check(
            _lib.lib().mf_icc_loss_grad(ctypes.byref(self.desc), q.data_ptr(), t.data_ptr(),
                                        loss.data_ptr(), gq.data_ptr(), gt.data_ptr(),
                                        self.ws.data_ptr(), _lib.stream_ptr()),
            "mf_icc_loss_grad")
        return loss, gq, gt

    def refine(self, q, t, adam_m, adam_v, n_iter, step0=0, alpha_q=0.01, alpha_t=0.001,
               losses=None, traj=None):
        """In-place: q [O,4], t [O,3], adam_m/adam_v [O,7] (contiguous float32 cuda).
        One hipGraph launch for the whole n_iter loop; no host synchronisation."""
        for x in (q, t, adam_m, adam_v):
            _lib.require_gpu(x)
            if x.dtype != torch.float32 or not x.is_contiguous():
                raise TypeError("refine() needs contiguous float32 state tensors")
        _lib.check(
            _lib.lib().mf_icc_refine(ctypes.byref(self.desc), q.data_ptr(), t.data_ptr(),
                                     adam_m.data_ptr(), adam_v.data_ptr(), int(n_iter), int(step0),
                                     float(alpha_q), float(alpha_t), _lib.ptr(losses),
                                     _lib.ptr(traj), self.ws.data_ptr(), _lib.stream_ptr()),
            "mf_icc_refine")
