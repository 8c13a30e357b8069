"""IterativeClosestPointLink -- ICP as a differentiable loss (single object).

API of morefusion/contrib/iterative_closest_point_link.py:9-44.  The reference
materialises the [T,S,3] difference tensor; ``mf_icp_loss_grad`` streams the
transformed source through LDS, keeps the arg-min per target point in registers and
reduces the matched squared distances and their pose-gradient moments in one launch.
``refine()`` / ``icp_refine()`` run the reference driver's whole loop
(examples/ycb_video/pose_refinement/check_iterative_closest_point_link.py:40-70: one link per
instance, summed loss, chainer Adam with the translations' alpha scaled) on the device:
``mf_icp_refine``, two launches per iteration, no autograd round trip, no host synchronisation.
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

    @property
    def T(self):
        return functions_module.transformation_matrix(self.quaternion, self.translation)

    def forward(self, source, target):
        # source: from cad, target: from depth;  keep = squared distance < 0.02 (:38)
        T = self.T
        return _IcpLoss.apply(T[:3, :3], T[:3, 3], source, target, 0.02)

    @torch.no_grad()
    def refine(self, source, target, n_iter=100, alpha=0.01, translation_alpha_scale=0.1, return_history=False,
               reset_optimizer=False):
        """The driver's loop for this one link, fused on the device; updates the parameters in place."""
        return icp_refine([self], [source], [target], n_iter=n_iter, alpha=alpha,
                          translation_alpha_scale=translation_alpha_scale, return_history=return_history,
                          reset_optimizer=reset_optimizer)


@torch.no_grad()
def icp_refine(links, sources, targets, n_iter=100, alpha=0.01, translation_alpha_scale=0.1, thresh=0.02,
               return_history=False, reset_optimizer=False):
    """``links``: IterativeClosestPointLink list (the driver's ChainList), ``sources`` / ``targets``: one
    [S_l,3] / [T_l,3] CUDA tensor per link.  Runs n_iter x {loss + gradient of every link, Adam step}
    as ``mf_icp_refine`` and writes the refined poses back into the links.  Optionally returns the
    per-iteration losses [n_iter, L] (their sum over L is the driver's loss).  The Adam moments and step
    count live on the links between calls (as an optimizer object would hold them);
    ``reset_optimizer=True`` starts from a fresh optimizer like a re-run of the reference driver
    (check_iterative_closest_point_link.py:40-45 builds a new ``Adam`` per run)."""
    L = len(links)
    if not (L == len(sources) == len(targets)):
        raise ValueError("one source and one target point set per link")
    dev = links[0].quaternion.device
    _lib.require_gpu(*[k.quaternion for k in links], *[k.translation for k in links], *sources, *targets)
    src = torch.cat([_lib.f32c(s) for s in sources]).contiguous()
    tgt = torch.cat([_lib.f32c(t) for t in targets]).contiguous()
    src_off = torch.tensor(np.r_[0, np.cumsum([s.shape[0] for s in sources])], dtype=torch.int32, device=dev)
    tgt_off = torch.tensor(np.r_[0, np.cumsum([t.shape[0] for t in targets])], dtype=torch.int32, device=dev)
    q = torch.stack([k.quaternion.data for k in links]).float().contiguous()
    t = torch.stack([k.translation.data for k in links]).float().contiguous()
    state = []
    for k in links:  # Adam moments persist on the link across calls, like an optimizer would hold them
        if reset_optimizer or getattr(k, "_adam", None) is None:
            k._adam = [torch.zeros(7, dtype=torch.float32, device=dev) for _ in range(2)]
            k._adam_t = 0
        state.append(k)
    if len({k._adam_t for k in links}) != 1:
        raise ValueError("links with different optimiser step counts cannot share one fused loop")
    m = torch.stack([k._adam[0] for k in links]).contiguous()
    v = torch.stack([k._adam[1] for k in links]).contiguous()
    losses = torch.empty((n_iter, L), dtype=torch.float32, device=dev) if return_history else None
    ws = torch.empty((28 * L,), dtype=torch.float32, device=dev)
    _lib.check(_lib.lib().mf_icp_refine(
        src.data_ptr(), src_off.data_ptr(), tgt.data_ptr(), tgt_off.data_ptr(), L,
        max(int(x.shape[0]) for x in targets), float(thresh), q.data_ptr(), t.data_ptr(), m.data_ptr(),
        v.data_ptr(), int(n_iter), int(links[0]._adam_t), float(alpha), float(alpha * translation_alpha_scale),
        _lib.ptr(losses), ws.data_ptr(), _lib.stream_ptr()), "mf_icp_refine")
    for i, k in enumerate(links):
        k.quaternion.data.copy_(q[i])
        k.translation.data.copy_(t[i])
        k._adam = [m[i].clone(), v[i].clone()]
        k._adam_t += n_iter
    return losses
